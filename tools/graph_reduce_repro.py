"""Repro attempt for DESIGN.md section 2's "ATen broadcast reduction returns garbage under hipGraph REPLAY" (round 5: the gradient of `conv1(h) + temb[:, :, None, None]`,
an ATen `sum` over the pixels of a channels-last tensor, came back with inf in single elements once the stacked SDXL step was replayed as hipGraphs on two lanes).
PURE PyTorch: no kernel of this repo runs here, so a mismatch below is ROCm / PyTorch's, and a clean sheet points back at this repo's graphs (pool sharing, lanes).

Every scenario captures the reduction, replays it `REPS` times and compares each result BITWISE with the eager result of the same input (ATen's reduction is
deterministic for a fixed launch configuration):
  single        one graph, one stream
  produced      the reduced tensor is produced inside the graph (elementwise kernel -> reduction), as autograd does
  autograd      the reduction is autograd's own: (x + t[:, :, None, None]).backward(gy) captured, t.grad compared
  shared_pool   two graphs (a "store" and an "accumulate" flavour) captured into ONE memory pool, replayed alternately (engine.py: a lane's two graphs)
  two_lanes     two graphs with private pools replayed CONCURRENTLY on two streams (the engine's lanes)
  forked        the capture itself forks onto a second stream and joins (multi-stream capture)
  lanes_shared  two lanes, each with a shared-pool graph pair, replayed concurrently -- the engine's exact arrangement
    python tools/graph_reduce_repro.py [out.json]"""
import json
import sys

import torch

REPS = 150
dev = torch.device('cuda:0')


def make(B, C, HW, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    gy = torch.randn(B, C, HW, HW, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return gy


def reduce_(gy):
    return gy.sum(dim=(2, 3))


def check(name, shape, results, want, report):
    bad = sum(0 if torch.equal(r, want) else 1 for r in results)
    nonfinite = sum(0 if bool(torch.isfinite(r).all()) else 1 for r in results)
    report.append({'scenario': name, 'shape': list(shape), 'replays': len(results), 'mismatching_replays': bad, 'non_finite_replays': nonfinite})
    print(report[-1], flush=True)


def capture(fn, pool=None, stream=None):
    g = torch.cuda.CUDAGraph()
    s = stream or torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        fn()                                                    # warm-up outside the capture
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, pool=pool, stream=s):
        out = fn()
    return g, out


def main():
    report = []
    for (B, C, HW) in [(4, 1280, 32), (4, 640, 64), (2, 320, 128), (8, 1280, 32)]:
        shape = (B, C, HW, HW)
        gy = make(B, C, HW, 1)
        want = reduce_(gy).clone()
        torch.cuda.synchronize()
        # single
        g, out = capture(lambda: reduce_(gy))
        res = []
        for _ in range(REPS):
            g.replay(); res.append(out.clone())
        torch.cuda.synchronize(); check('single', shape, res, want, report)
        # produced inside the graph
        src = gy.clone()
        g, out = capture(lambda: reduce_(src * 1.0))
        res = []
        for _ in range(REPS):
            g.replay(); res.append(out.clone())
        torch.cuda.synchronize(); check('produced', shape, res, want, report)
        # autograd's own reduction
        x = torch.zeros_like(gy).requires_grad_(True)
        t = torch.zeros(B, C, device=dev, dtype=torch.bfloat16, requires_grad=True)

        def bw():
            t.grad = None
            (x + t[:, :, None, None]).backward(gy)
            return t.grad
        g, out = capture(bw)
        want_t = bw().clone()
        res = []
        for _ in range(REPS):
            g.replay(); res.append(out.clone())
        torch.cuda.synchronize(); check('autograd', shape, res, want_t, report)
        # two graphs in one pool, alternately
        pool = torch.cuda.graph_pool_handle()
        acc = torch.zeros(B, C, device=dev, dtype=torch.float32)

        def store():
            acc.copy_(reduce_(gy * 1.0)); return acc

        def accum():
            acc.add_(reduce_(gy * 1.0)); return acc
        g1, _ = capture(store, pool=pool)
        g2, _ = capture(accum, pool=pool)
        res = []
        for _ in range(REPS // 3):
            g1.replay(); g2.replay(); g2.replay(); res.append(acc.clone())
        w3 = (want.float() * 3)
        torch.cuda.synchronize(); check('shared_pool', shape, res, w3, report)
        # two lanes concurrently
        gyb = make(B, C, HW, 2)
        wantb = reduce_(gyb).clone()
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ga, oa = capture(lambda: reduce_(gy * 1.0))
        gb, ob = capture(lambda: reduce_(gyb * 1.0))
        ra, rb = [], []
        for _ in range(REPS):
            with torch.cuda.stream(sa):
                ga.replay(); ra.append(oa.clone())
            with torch.cuda.stream(sb):
                gb.replay(); rb.append(ob.clone())
        torch.cuda.synchronize(); check('two_lanes.a', shape, ra, want, report); check('two_lanes.b', shape, rb, wantb, report)
        # capture that forks onto a second stream
        side = torch.cuda.Stream(dev)

        def forked():
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                a = reduce_(gy * 1.0)
            b = reduce_(gyb * 1.0)
            cur.wait_stream(side)
            return a + b
        g, out = capture(forked)
        res = []
        for _ in range(REPS):
            g.replay(); res.append(out.clone())
        torch.cuda.synchronize(); check('forked', shape, res, want + wantb, report)
        # the engine's arrangement: two lanes, each a shared-pool pair, concurrent
        lanes = []
        for li, src_ in enumerate((gy, gyb)):
            pool = torch.cuda.graph_pool_handle()
            acc_ = torch.zeros(B, C, device=dev, dtype=torch.float32)
            s1, _ = capture(lambda a=acc_, s=src_: a.copy_(reduce_(s * 1.0)), pool=pool)
            s2, _ = capture(lambda a=acc_, s=src_: a.add_(reduce_(s * 1.0)), pool=pool)
            lanes.append((torch.cuda.Stream(dev), s1, s2, acc_))
        ra, rb = [], []
        for _ in range(REPS // 2):
            for (st, s1, s2, a), sink in zip(lanes, (ra, rb)):
                with torch.cuda.stream(st):
                    s1.replay(); s2.replay(); sink.append(a.clone())
        torch.cuda.synchronize()
        check('lanes_shared.a', shape, ra, want.float() * 2, report); check('lanes_shared.b', shape, rb, wantb.float() * 2, report)
    # long: ONE capture holding a chain of 60 (matmul -> channels-last view -> pixel sum) links with their temporaries freed and reused inside the capture --
    # the shape of a lane graph (hundreds of kernels, the reduction's staging buffer / semaphores allocated from the capture's private pool between other buffers)
    B, C, HW = 4, 640, 64
    w = (torch.randn(C, C, device=dev) / C ** 0.5).to(torch.bfloat16)
    x0 = make(B, C, HW, 7)

    def chain():
        acc = torch.zeros(B, C, device=dev, dtype=torch.float32)
        x = x0
        for i in range(60):
            y = torch.matmul(x.permute(0, 2, 3, 1).reshape(-1, C), w).view(B, HW, HW, C).permute(0, 3, 1, 2)          # channels-last [B, C, H, W]
            acc = acc + y.sum(dim=(2, 3)).float() * (1.0 / (HW * HW))
            x = (y * 0.5 + x0 * 0.5)
        return acc
    want_c = chain().clone()
    torch.cuda.synchronize()
    g, out = capture(chain)
    res = []
    for _ in range(40):
        g.replay(); res.append(out.clone())
    torch.cuda.synchronize(); check('long_chain', (B, C, HW, HW), res, want_c, report)
    bad = [r for r in report if r['mismatching_replays'] or r['non_finite_replays']]
    print(f'{len(bad)} of {len(report)} scenario / shape pairs misbehaved', flush=True)
    if len(sys.argv) > 1:
        json.dump({'torch': torch.__version__, 'hip': torch.version.hip, 'device': torch.cuda.get_device_name(0), 'misbehaving': bad, 'all': report}, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
