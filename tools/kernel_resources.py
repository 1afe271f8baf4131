"""Compile every csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and tabulate registers / spills / LDS / occupancy per
kernel (no GPU needed).  Usage: python tools/kernel_resources.py [out.csv]"""
import csv
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusion_pipe_amd import build  # noqa: E402
CSRC = ROOT / 'diffusion_pipe_amd' / 'csrc'
FIELDS = {'VGPRs': 'vgprs', 'AGPRs': 'agprs', 'SGPRs': 'sgprs', 'ScratchSize [bytes/lane]': 'scratch_bytes', 'Occupancy [waves/SIMD]': 'occupancy',
          'SGPRs Spill': 'sgpr_spill', 'VGPRs Spill': 'vgpr_spill', 'LDS Size [bytes/block]': 'lds_bytes'}


def main():
    rows = []
    for src in sorted(CSRC.glob('*.hip')):
        # the build's own flags (diffusion_pipe_amd/build.py), per-file extras included
        r = subprocess.run([build.HIPCC, *build.CFLAGS, *build.EXTRA_CFLAGS.get(src.name, []), '-c', str(src), '-o', '/dev/null',
                            '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, cwd=CSRC)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r'remark: (?:\s*)Function Name: (\S+)', line)
            if m:
                name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = {'file': src.name, 'kernel': re.sub(r'\(anonymous namespace\)::', '', name)[:140]}
                rows.append(cur)
                continue
            m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)', line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    out = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / 'profiles' / 'r2_kernel_resources.csv')
    cols = ['file', 'kernel'] + list(FIELDS.values())
    with open(out, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=cols)
        w.writeheader()
        for r in rows:
            w.writerow({c: r.get(c, '') for c in cols})
    spilled = [r for r in rows if r.get('vgpr_spill')]
    print(f'{len(rows)} kernels -> {out}; VGPR spills in {len(spilled)}:')
    for r in spilled:
        print('  ', r['file'], r['kernel'][:100], r['vgpr_spill'])


if __name__ == '__main__':
    main()
