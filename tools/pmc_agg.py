"""Aggregate a rocprofv3 `--pmc` counter_collection CSV on the GPU box: per kernel name -> dispatches, mean counter value.
Usage: python tools/pmc_agg.py <dir with *_counter_collection.csv> <out.csv>   (the raw CSV is deleted afterwards: it is
tens of MB and gpurun only copies small files back)."""
import csv
import glob
import os
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in files:
    with open(f, newline='') as fh:
        for row in csv.DictReader(fh):
            name = row.get('Kernel_Name') or row.get('Kernel Name') or ''
            ctr = row.get('Counter_Name') or row.get('Counter Name') or ''
            val = float(row.get('Counter_Value') or row.get('Counter Value') or 0)
            a = acc[name][ctr]
            a[0] += 1
            a[1] += val
    os.remove(f)
with open(out, 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'counter', 'dispatches', 'mean_value', 'sum_value'])
    for name in sorted(acc, key=lambda n: -sum(v[1] for v in acc[n].values())):
        for ctr, (n, s) in sorted(acc[name].items()):
            w.writerow([name[:160], ctr, n, s / max(n, 1), s])
print(f'{len(files)} file(s), {len(acc)} kernels -> {out}')
