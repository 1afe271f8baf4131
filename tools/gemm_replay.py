"""Re-issue a recorded GEMM launch list (ops.GEMM_TRACE descriptors: shapes, leading dimensions, batch strides, epilogue flags) over scratch
operands -- as ONE hipGraph timed with HIP events (bench.py's `roofline`: in-graph, back-to-back launches exactly as the training step replays
them, average launch duration = graph time / launches) or eagerly (`python tools/gemm_replay.py <trace.json>`: a rocprofv3 `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` target; the counter collector cannot attribute launches inside a replayed hipGraph).

Operands rotate through a scratch arena much larger than the 8 x 4 MiB L2s + the 256 MiB Infinity Cache, so every launch finds its weights
HBM-cold as in training (each weight is read once per micro-batch pass); element values are small random bf16."""
import json
import sys
from collections import OrderedDict

import torch

ESZ = {0: 2, 1: 4}      # hip.BF16, hip.F32


def _extent(rows, cols, ld, d, s):
    return (rows - 1) * ld + cols + (d['bo'] - 1) * s[0] + (d['bi'] - 1) * s[1]


def operand_elems(d):
    """elements spanned by A, B, C of one descriptor (the bytes a launch must at least touch are algorithmic_bytes())."""
    ra, ca = (d['K'], d['M']) if d['ta'] else (d['M'], d['K'])
    rb, cb = (d['N'], d['K']) if d['tb'] else (d['K'], d['N'])
    return (_extent(ra, ca, d['lda'], d, d['sa']), _extent(rb, cb, d['ldb'], d, d['sb']), _extent(d['M'], d['N'], d['ldc'], d, d['sc']))


def flops(d):
    return 2.0 * d['M'] * d['N'] * d['K'] * d['bo'] * d['bi']


def algorithmic_bytes(d):
    """every operand read once, the result written once (+ read once when accumulating / adding a residual)"""
    batch = d['bo'] * d['bi']
    e = ESZ[d['dt']]
    oe = 4 if d['out_f32'] else e
    rd = (d['M'] * d['K'] + d['N'] * d['K']) * e * batch + (d['N'] * e if d['bias'] else 0)
    rd += d['M'] * d['N'] * oe * batch * (int(d['acc']) + int(d['res']))
    wr = d['M'] * d['N'] * oe * batch + (d['M'] * e * batch if d['colsum'] else 0)
    return rd, wr


class Arena:
    def __init__(self, device, nbytes):
        n = nbytes // 2
        self.buf = (torch.randn(n, device=device, dtype=torch.float32) * 0.05).to(torch.bfloat16) if n <= (1 << 28) else \
            torch.cat([(torch.randn(1 << 28, device=device, dtype=torch.float32) * 0.05).to(torch.bfloat16) for _ in range((n + (1 << 28) - 1) >> 28)])[:n]
        self.bytes = self.buf.view(torch.uint8)
        self.off = 0

    def take(self, nbytes, dtype):
        nbytes = (nbytes + 255) & ~255
        if self.off + nbytes > self.bytes.numel():
            self.off = 0
        v = self.bytes[self.off:self.off + nbytes].view(dtype)
        self.off += nbytes
        return v


def _operands(d, arena):
    from diffusion_pipe_amd import hip
    dt = torch.bfloat16 if d['dt'] == hip.BF16 else torch.float32
    odt = torch.float32 if d['out_f32'] else dt
    ea, eb, ec = operand_elems(d)
    a = arena.take(ea * ESZ[d['dt']], dt)[:ea]
    b = arena.take(eb * ESZ[d['dt']], dt)[:eb]
    osz = 4 if d['out_f32'] else ESZ[d['dt']]
    c = arena.take(ec * osz, odt)[:ec]
    bias = arena.take(d['N'] * ESZ[d['dt']], dt)[:d['N']] if d['bias'] else None
    er = (d['M'] - 1) * d['ldr'] + d['N'] + (d['bo'] - 1) * d['sc'][0] + (d['bi'] - 1) * d['sc'][1] if d['res'] else 0
    res = arena.take(er * osz, odt)[:er] if d['res'] else None
    cs = arena.take(d['M'] * d['bo'] * d['bi'] * ESZ[d['dt']], dt)[:d['M'] * d['bo'] * d['bi']] if d['colsum'] else None
    return a, b, c, bias, res, cs


def groups(trace):
    """the launch list as the step issued it: [[d], [d0, d1], ...] -- consecutive descriptors recorded by one ops.gemm_group call (`grp` = 0 .. `grp_n` - 1) form
    one entry, every other descriptor an entry of its own"""
    out, i = [], 0
    while i < len(trace):
        n = trace[i].get('grp_n') or 1
        if n > 1 and trace[i].get('grp') == 0 and all(j < len(trace) and trace[j].get('grp') == j - i for j in range(i, i + n)):
            out.append(trace[i:i + n]); i += n
        else:
            out.append([trace[i]]); i += 1
    return out


def launch_count(trace):
    """kernel launches of the list: 1 per single descriptor, what dpipe_gemm_group reported (`grp_l`) per grouped call"""
    return sum((g[0].get('grp_l') or len(g)) if len(g) > 1 else 1 for g in groups(trace))


def issue(trace, arena, ops):
    """launch every descriptor once on the current stream over fresh arena slices (grouped calls as grouped calls)"""
    for g in groups(trace):
        if len(g) == 1:
            d = g[0]
            a, b, c, bias, res, cs = _operands(d, arena)
            ops.gemm(a, b, d['ta'], d['tb'], d['M'], d['N'], d['K'], c, lda=d['lda'], ldb=d['ldb'], ldc=d['ldc'], batch_outer=d['bo'], batch_inner=d['bi'],
                     stride_a=d['sa'], stride_b=d['sb'], stride_c=d['sc'], bias=bias, act=d['act'], alpha=d['alpha'], accumulate=d['acc'], tile_hint=d['tile'],
                     residual=res, ldr=d['ldr'], colsum=cs, colsum_accumulate=d['colsum_acc'])
            continue
        probs = []
        for d in g:
            a, b, c, bias, res, cs = _operands(d, arena)
            ra, ca = (d['K'], d['M']) if d['ta'] else (d['M'], d['K'])
            rb, cb = (d['N'], d['K']) if d['tb'] else (d['K'], d['N'])
            probs.append({'a': torch.as_strided(a, (ra, ca), (d['lda'], 1)), 'b': torch.as_strided(b, (rb, cb), (d['ldb'], 1)), 'ta': d['ta'], 'tb': d['tb'],
                          'M': d['M'], 'N': d['N'], 'K': d['K'], 'out': torch.as_strided(c, (d['M'], d['N']), (d['ldc'], 1)), 'lda': d['lda'], 'ldb': d['ldb'], 'ldc': d['ldc'],
                          'bias': bias, 'act': d['act'], 'alpha': d['alpha'], 'acc': d['acc'],
                          'res': torch.as_strided(res, (d['M'], d['N']), (d['ldr'], 1)) if res is not None else None, 'ldr': d['ldr'],
                          'colsum': cs, 'colsum_acc': d['colsum_acc']})
        if ops.gemm_group(probs) is None:
            raise RuntimeError('gemm_replay: a recorded grouped call is not eligible on replay')


def time_in_graph(trace, device, reps=3, arena_bytes=3 << 30):
    """-> dict(ms per replay of the launch list, launches, flops, algorithmic read / write bytes).  One hipGraph of every launch, replayed
    `reps` times on the current stream between two HIP events recorded on that same stream."""
    from diffusion_pipe_amd import ops
    trace = [d for d in trace]
    arena = Arena(device, arena_bytes)
    ops.WS_LANE = 'roofline'
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        issue(trace[:64], arena, ops)                      # warm-up (workspace allocation, module load)
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    arena.off = 0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        issue(trace, arena, ops)
    ops.WS_LANE = None
    graph.replay()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize(device)
    rd = sum(algorithmic_bytes(d)[0] for d in trace)
    wr = sum(algorithmic_bytes(d)[1] for d in trace)
    return {'ms': e0.elapsed_time(e1) / reps, 'launches': launch_count(trace), 'problems': len(trace), 'flops': sum(flops(d) for d in trace), 'read_bytes': rd, 'write_bytes': wr}


def time_concurrent(trace, device, lanes, reps=2, arena_bytes=2 << 30):
    """The same launch list as `lanes` hipGraphs replayed AT THE SAME TIME on `lanes` HIP streams (each its own operand arena and split-K workspace) -- the
    way the engine's micro-batch lanes run it.  -> dict(ms for all lanes' lists, launches, flops) of ONE round (every lane replays its list once)."""
    from diffusion_pipe_amd import ops
    main = torch.cuda.current_stream(device)
    streams = [main] + [torch.cuda.Stream(device) for _ in range(lanes - 1)]        # lane 0 on the caller's stream, as the engine runs it (4 hardware queues)
    graphs = []
    for li, st in enumerate(streams):
        arena = Arena(device, arena_bytes)
        ops.WS_LANE = ('roofline-lane', li)
        side = torch.cuda.Stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            issue(trace[:64], arena, ops)
        main.wait_stream(side)
        torch.cuda.synchronize(device)
        arena.off = 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):       # captured on torch's capture stream; a graph replays on whatever stream launches it
            issue(trace, arena, ops)
        graphs.append((g, arena))
    ops.WS_LANE = None

    def round_():
        for st in streams[1:]:
            st.wait_stream(main)            # fork BEFORE the caller's stream receives lane 0's graph
        for st, (g, _) in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
        for st in streams[1:]:
            main.wait_stream(st)
    round_()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps):
        round_()
    e1.record(main)
    torch.cuda.synchronize(device)
    return {'ms': e0.elapsed_time(e1) / reps, 'launches': launch_count(trace) * lanes, 'flops': sum(flops(d) for d in trace) * lanes, 'lanes': lanes}


def unique_with_counts(trace):
    """unique launch-list entries with their counts, flattened: the members of a grouped call stay adjacent (`grp` / `grp_n` / `grp_l` keys) and carry the count of
    their group, so per-descriptor tools can still walk the list and `groups()` can rebuild the calls"""
    u = OrderedDict()
    for g in groups(trace):
        k = json.dumps(g, sort_keys=True)
        u[k] = u.get(k, 0) + 1
    return [dict(d, count=n) for k, n in u.items() for d in json.loads(k)]


def main():
    """eager replay of a saved unique-descriptor list (bench.py --save-gemm-trace): each descriptor `reps` times over rotating operands"""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffusion_pipe_amd import ops
    path, div = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    uniq = json.load(open(path))
    dev = torch.device('cuda:0')
    arena = Arena(dev, 3 << 30)
    total = 0
    for g in groups(uniq):
        n = max(1, g[0]['count'] // div)         # the step's own launch mix: every entry as often as the step issues it (/ div)
        g = [{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k != 'count'} for d in g]
        for _ in range(n):
            issue(g, arena, ops)
        total += n
    torch.cuda.synchronize()
    print(f'{len(uniq)} unique GEMM descriptors, {total} eager calls')


if __name__ == '__main__':
    main()
