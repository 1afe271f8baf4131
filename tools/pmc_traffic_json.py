"""profiles/r*_pmc_gemm_traffic.json from the two `tools/pmc_agg.py` summaries of the FETCH_SIZE / WRITE_SIZE passes over `tools/gemm_replay.py <trace> <div>`
(separate rocprofv3 --pmc passes, `tools/gpu_profile.sh`): HBM-side bytes per GEMM launch next to the algorithmic bytes of the same launch list.
FETCH_SIZE is in KB and reports half the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section: doubled here); WRITE_SIZE in KB as reported.

    python tools/pmc_traffic_json.py <fetch_agg.csv> <write_agg.csv> <trace.json> <div> <out.json>"""
import csv
import json
import sys

sys.path.insert(0, '.')
from tools import gemm_replay  # noqa: E402


def total(path, counter):
    kb, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if 'gemm_pipe_kernel' in row['kernel'] or 'gemm_pipe_group_kernel' in row['kernel'] or 'gemm_kernel' in row['kernel']:
            if row['counter'] == counter:
                kb += float(row['sum_value']); n += int(row['dispatches'])
    return kb, n


def main():
    fetch_csv, write_csv, trace, div, out = sys.argv[1:6]
    fkb, fn = total(fetch_csv, 'FETCH_SIZE')
    wkb, wn = total(write_csv, 'WRITE_SIZE')
    uniq = json.load(open(trace))
    rd = wr = n = 0
    for g in gemm_replay.groups(uniq):                 # a grouped call (dpipe_gemm_group) is `grp_l` launches (normally one) for all its problems
        c = max(1, g[0]['count'] // int(div))
        for d in g:
            dd = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}
            r, w = gemm_replay.algorithmic_bytes(dd)
            rd += r * c; wr += w * c
        n += c * ((g[0].get('grp_l') or len(g)) if len(g) > 1 else 1)
    fetch_b, write_b = 2.0 * fkb * 1024, wkb * 1024
    res = {'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/gemm_replay.py {trace} {div} (one micro-batch of the step\'s GEMM launch list, '
                     'eager, HBM-cold operands); FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md, WRITE_SIZE (KB) as reported',
           'launches': fn, 'launches_write_pass': wn, 'launches_expected': n,
           'fetch_bytes_per_launch': round(fetch_b / max(fn, 1)), 'write_bytes_per_launch': round(write_b / max(wn, 1)),
           'hbm_bytes_per_launch': round(fetch_b / max(fn, 1) + write_b / max(wn, 1)),
           'algorithmic_read_bytes_per_launch': round(rd / n), 'algorithmic_write_bytes_per_launch': round(wr / n),
           'fetch_over_algorithmic_reads': round(fetch_b / max(fn, 1) / (rd / n), 2), 'write_over_algorithmic_writes': round(write_b / max(wn, 1) / (wr / n), 2)}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
