"""Concurrency summary of a rocprofv3 --kernel-trace CSV (the lanes' kernels with their real start / end stamps): how busy each hardware queue is, how many kernels
run at once, how much of the wall time no kernel runs, and per kernel family the time it takes UNDER concurrency next to its launch count -- to be read beside the
serialised per-kernel averages of --stats.

    python tools/trace_overlap.py <kernel_trace.csv> [out.json] [--skip-frac 0.35]      (the first part of the trace = start-up / warm-up is skipped)"""
import csv
import json
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    skip = 0.35
    for i, a in enumerate(sys.argv):
        if a == '--skip-frac':
            skip = float(sys.argv[i + 1])
    rows = []
    with open(args[0]) as f:
        rd = csv.DictReader(f)
        for r in rd:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']))
    rows.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    cut = t_lo + int((t_hi - t_lo) * skip)
    rows = [r for r in rows if r[0] >= cut]
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    wall = t_hi - t_lo
    ev = []
    per_q = defaultdict(int)
    fam = defaultdict(lambda: [0, 0])
    for s, e, q, name in rows:
        ev.append((s, 1)); ev.append((e, -1))
        per_q[q] += e - s
        key = name.split('<')[0].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '').replace('dpipe_pipe::', '')[:48]
        fam[key][0] += e - s; fam[key][1] += 1
    ev.sort()
    hist = defaultdict(int)
    level, last = 0, t_lo
    for t, d in ev:
        hist[level] += t - last
        last = t
        level += d
    out = {'kernels': len(rows), 'wall_ms': round(wall / 1e6, 2), 'sum_kernel_ms': round(sum(e - s for s, e, _, _ in rows) / 1e6, 2),
           'mean_concurrency': round(sum(e - s for s, e, _, _ in rows) / wall, 3),
           'wall_share_by_kernels_in_flight': {str(k): round(v / wall, 4) for k, v in sorted(hist.items())},
           'queue_busy_share': {q: round(v / wall, 4) for q, v in sorted(per_q.items())},
           'families_ms_and_launches': {k: [round(v[0] / 1e6, 2), v[1], round(v[0] / v[1] / 1e3, 1)] for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:24]}}
    js = json.dumps(out, indent=1)
    print(js)
    if len(args) > 1:
        open(args[1], 'w').write(js)


if __name__ == '__main__':
    main()
