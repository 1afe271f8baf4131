"""Per-descriptor ledger of the step's GEMM launch list: every UNIQUE descriptor of a saved trace (bench.py --save-gemm-trace; profiles/r*_gemm_trace_sdxl_step.json)
timed HBM-cold inside a hipGraph (operands rotate through an arena larger than L2 + Infinity Cache), with the dispatcher's own choice (`auto`), a set of forced
tile / split-K configurations (dpipe_gemm_ex `tile_hint`) and hipBLASLt through torch.matmul where the descriptor is a plain 2-D product.

    python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json [out.jsonl] [--hints auto,2001,...] [--min-share 0.0]

One JSON line per descriptor: shape, count per step, us per launch per configuration, share of the step's GEMM time; last line = totals (ms per step by
configuration: `auto`, `best` = per-descriptor minimum over the forced configurations, `torch`)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gemm_replay  # noqa: E402

DEFAULT_HINTS = {'auto': 0, 't64': 2001, 't64k2': 2002, 't64k4': 2004, 't128': 3001, 't128k2': 3002, 't128k3': 3003, 't128r2': 4001, 't128r2k2': 4002,
                 't256': 7001, 't256k2': 7002, 't256h': 9001, 't64k8': 2008, 't64k16': 2016,
                 # (hints 5000 / 6000 / 8000 / 10000 / 11000 -- 256 x 128, 64^2 3-deep, 128^2 5-deep, skinny 128 x 64, 4-wave half-K-step 128^2 -- named tiles removed in round 5)
                 't128v': 12001, 't128vk2': 12002, 't128vk3': 12003}      # round 5: register-staged 128^2 tile     # 4-wave 128^2 on a 3-deep ring of half K-steps (3 workgroups per CU); deeper 64^2 splits
# (round 3 also timed hints 12000 - 14000: T128Q4, T128H5, T128H4 -- profiles/r3k_*, r3q_*; removed from the library)
# (round 3 also timed four-wave tiles under hints 11000 - 14000: profiles/r3_gemm_desc_ledger_4wave_tiles_negative.jsonl; removed from the library)


def time_desc(d, hint, device, arena, ops, budget_bytes=600 << 20):
    """us per launch of descriptor d with tile_hint `hint`: nbuf launches over distinct arena slices in one graph."""
    rd, wr = gemm_replay.algorithmic_bytes(d)
    nbuf = max(4, min(128, budget_bytes // max(rd + wr, 1)))
    dd = dict(d, tile=hint)
    dd.pop('count', None)
    lst = [dd] * nbuf
    arena.off = 0
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        gemm_replay.issue(lst[:2], arena, ops)
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    arena.off = 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        gemm_replay.issue(lst, arena, ops)
    g.replay()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / (3 * nbuf) * 1e3


def time_torch(d, device, arena, budget_bytes=600 << 20):
    if d['bo'] * d['bi'] != 1 or d['dt'] != 0 or d['out_f32']:
        return None
    M, N, K = d['M'], d['N'], d['K']
    rd, wr = gemm_replay.algorithmic_bytes(d)
    nbuf = max(4, min(128, budget_bytes // max(rd + wr, 1)))
    arena.off = 0
    ops_ = []
    for _ in range(nbuf):
        a = arena.take((K * M) * 2, torch.bfloat16)[:K * M].view((K, M) if d['ta'] else (M, K))
        b = arena.take((K * N) * 2, torch.bfloat16)[:K * N].view((N, K) if d['tb'] else (K, N))
        c = arena.take((M * N) * 2, torch.bfloat16)[:M * N].view(M, N)
        ops_.append((a.t() if d['ta'] else a, b.t() if d['tb'] else b, c))

    def run():
        for a, b, c in ops_:
            if d['acc']:
                c.addmm_(a, b)
            else:
                torch.matmul(a, b, out=c)
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        run()
    g.replay()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / (3 * nbuf) * 1e3


def main():
    from diffusion_pipe_amd import ops
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = dict(a[2:].split('=', 1) if '=' in a else (a[2:], '1') for a in sys.argv[1:] if a.startswith('--'))
    uniq = json.load(open(args[0]))
    out = open(args[1], 'w') if len(args) > 1 else None
    hints = DEFAULT_HINTS
    if 'hints' in opts:
        named = {v: k for k, v in DEFAULT_HINTS.items()}
        hints = {('auto' if h in ('0', 'auto') else named.get(int(h), f'h{h}')): (0 if h in ('0', 'auto') else int(h)) for h in opts['hints'].split(',')}
    skip_torch = 'no-torch' in opts
    min_gflop = float(opts.get('min-gflop', 0))
    uniq = [d for d in uniq if gemm_replay.flops({k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}) / 1e9 >= min_gflop]
    dev = torch.device('cuda:0')
    arena = gemm_replay.Arena(dev, 3 << 30)
    ops.WS_LANE = 'desc-timing'
    rows = []
    for d in uniq:
        d = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}
        rec = {'ta': d['ta'], 'tb': d['tb'], 'M': d['M'], 'N': d['N'], 'K': d['K'], 'batch': d['bo'] * d['bi'], 'count': d['count'], 'acc': int(d['acc']),
               'colsum': int(d['colsum']), 'bias': int(d['bias']), 'res': int(d['res']), 'gflop': round(gemm_replay.flops(d) / 1e9, 3), 'us': {}}
        for name, hint in hints.items():
            if (7000 <= hint < 8000 or 9000 <= hint < 10000) and (d['M'] < 512 or d['N'] < 512):
                continue                     # 256^2 tiles: not for slivers
            try:
                rec['us'][name] = round(time_desc(d, hint, dev, arena, ops), 2)
            except Exception as e:       # configuration not eligible for this descriptor
                rec['us'][name] = None
        if not skip_torch:
            try:
                t = time_torch(d, dev, arena)
                rec['us']['torch'] = round(t, 2) if t is not None else None
            except Exception:
                rec['us']['torch'] = None
        rows.append(rec)
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + '\n'); out.flush()
    tot = {}
    forced = [n for n in hints if n != 'auto']
    for r in rows:
        us = r['us']
        tot['auto'] = tot.get('auto', 0.0) + r['count'] * (us.get('auto') or 0.0)
        cands = [us[n] for n in forced if us.get(n)] + ([us['auto']] if us.get('auto') else [])
        tot['best'] = tot.get('best', 0.0) + r['count'] * (min(cands) if cands else 0.0)
        tot['torch_or_auto'] = tot.get('torch_or_auto', 0.0) + r['count'] * (us.get('torch') or us.get('auto') or 0.0)
    summ = {'summary_ms_per_step': {k: round(v / 1e3, 2) for k, v in tot.items()}, 'launches': sum(r['count'] for r in rows),
            'flops_per_step_T': round(sum(r['count'] * r['gflop'] for r in rows) / 1e3, 2)}
    print(json.dumps(summ), flush=True)
    if out:
        out.write(json.dumps(summ) + '\n')


if __name__ == '__main__':
    main()
