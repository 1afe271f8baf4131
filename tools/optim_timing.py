"""Step-end kernels against the HBM roofline: the fused AdamW passes (dpipe_adamw_sumsq / dpipe_adamw_step, 1 and 3 gradient lanes) and the 8-bit block-wise AdamW
(dpipe_adamw8bit_multi, with and without the Kahan shift buffer) on ~1 GiB of bf16 parameters split into SDXL-like tensors; HIP events around 10 steps.
Bytes moved per parameter (reads + writes): AdamW 2 (L + 3) x 2 B + L x 2 B for the norm pass; 8-bit: p 2+2, g 2, codes 2 x (1+1), absmax ~0 = 10 B (+ 4 B shift).
    python tools/optim_timing.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pipe_amd import optim  # noqa: E402


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device('cuda:0')
    shapes = [(10240, 1280)] * 24 + [(1280, 1280)] * 96 + [(1280, 5120)] * 16 + [(1280,)] * 200 + [(640, 640, 3, 3)] * 8
    n = sum(int(torch.tensor(s).prod()) for s in shapes)
    for kind, kahan in (('adamw8bit', False), ('adamw8bitkahan', True)):
        ps = [torch.nn.Parameter((torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)) for s in shapes]
        for p in ps:
            p.grad = (torch.randn_like(p.float()) * 0.01).to(torch.bfloat16)
        opt = optim.AdamW8bit(ps, lr=1e-4, betas=(0.9, 0.99), weight_decay=0.01, kahan=kahan)
        ms = timed(opt.step)
        big = sum(p.numel() for p in ps if p.numel() >= 4096)
        bpp = 14 if kahan else 10
        print(json.dumps({'op': kind, 'parameters': n, 'tensors': len(shapes), 'ms_per_step': round(ms, 3), 'bytes_per_parameter': bpp,
                          'TB_per_s': round(big * bpp / ms / 1e9, 3), 'frac_of_6.3TBps': round(big * bpp / ms / 1e9 / 6.3, 3)}), flush=True)
        del opt, ps
    for lanes in (1, 3):
        ps = [torch.nn.Parameter((torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)) for s in shapes]
        grads = [{id(p): (torch.randn_like(p.float()) * 0.01).to(torch.bfloat16) for p in ps} for _ in range(lanes)]
        opt = optim.FusedAdamW(ps, lr=1e-4, betas=(0.9, 0.99), weight_decay=0.01)

        def step():
            total = opt.grads_sumsq(grads)
            opt.fused_update(grads, total, 1.0, zero_grads=True)
        ms = timed(step)
        bpp = 2 * lanes + 2 * (lanes + 3) * 2          # norm pass reads L lanes; update pass reads L + 3 and writes L + 3 tensors
        print(json.dumps({'op': 'fused_adamw_step_end', 'lanes': lanes, 'parameters': n, 'ms_per_step': round(ms, 3), 'bytes_per_parameter': bpp,
                          'TB_per_s': round(n * bpp / ms / 1e9, 3), 'frac_of_6.3TBps': round(n * bpp / ms / 1e9 / 6.3, 3)}), flush=True)
        del opt, ps, grads


if __name__ == '__main__':
    main()
