"""Per-kernel summary of a `tools/run_gpu.sh sqpmc` pass (SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE, aggregated by tools/pmc_agg.py):

    python tools/sq_summary.py <agg.csv> [out.json] [top]

mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024): the MFMA pipes' busy cycles (32 per v_mfma_f32_32x32x16_bf16, summed over the chip's 1 024 SIMDs,
MI355X_MICROARCH.md) over the kernel's duration in cycles (SQ_BUSY_CYCLES is summed over the 32 shader engines: / 32 reproduces the launch durations of the --stats pass)
times the SIMD count.  The wave-cycle shares (parked at s_waitcnt / barrier, issue-stalled, of which on the LDS pipe, issuing) are fractions of SQ_WAVE_CYCLES and disjoint
(guide, "rocprofv3 PMC slots")."""
import csv
import json
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    d = OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault(r['kernel'], {})[r['counter']] = (int(r['dispatches']), float(r['mean_value']))
    rows = []
    for k, c in d.items():
        g = lambda n: c.get(n, (0, 0.0))[1]          # noqa: E731
        n, wc, busy = c.get('SQ_WAVE_CYCLES', (0, 0))[0], g('SQ_WAVE_CYCLES'), g('SQ_BUSY_CYCLES')
        if wc == 0 or busy == 0 or k.startswith('void at::') or 'rocclr' in k:
            continue
        rows.append({'kernel': k[:110], 'dispatches': n, 'kernel_cycles': round(busy / 32), 'mfma_busy_pct': round(100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (busy * 32), 1),
                     'wave_parked_waitcnt_barrier': round(g('SQ_WAIT_ANY') / wc, 3), 'wave_issue_stalled': round(g('SQ_WAIT_INST_ANY') / wc, 3),
                     'of_which_lds_pipe': round(g('SQ_WAIT_INST_LDS') / wc, 3), 'wave_issuing': round(g('SQ_ACTIVE_INST_ANY') / wc, 3),
                     'valu_insts_per_mfma_busy_cycle': round(g('SQ_INSTS_VALU') / max(g('SQ_VALU_MFMA_BUSY_CYCLES'), 1), 3) if g('SQ_VALU_MFMA_BUSY_CYCLES') else None,
                     'weight': busy * n})
    rows.sort(key=lambda r: -r['weight'])
    for r in rows:
        r.pop('weight')
    rows = rows[:top]
    for r in rows:
        print(f"{r['kernel'][:78]:78s} n={r['dispatches']:5d} cyc={r['kernel_cycles']:8d} mfma {r['mfma_busy_pct']:5.1f}%  parked {r['wave_parked_waitcnt_barrier']:.2f} stalled {r['wave_issue_stalled']:.2f} (lds {r['of_which_lds_pipe']:.2f}) issuing {r['wave_issuing']:.2f}")
    if len(sys.argv) > 2:
        json.dump(rows, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
