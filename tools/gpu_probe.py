"""Micro-benchmarks run on the GPU box: per-kernel timings next to the PyTorch-ROCm library op of the same shape.
Writes JSON lines to stdout; used to steer kernel work, not part of the product path."""
import json
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def gemm_probe():
    dev = torch.device('cuda')
    shapes = [  # (M, N, K) of SDXL / Flux / Wan linears at bs=1
        (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560),
        (4608, 9216, 3072), (4608, 3072, 12288), (9216, 5120, 5120), (9216, 13824, 5120), (8192, 8192, 8192)]
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) / K ** 0.5
        gy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        rec = {'probe': 'gemm', 'M': M, 'N': N, 'K': K}
        for name, fn, ref in [
            ('fwd', lambda: ops.mm(x, w, False, True), lambda: torch.nn.functional.linear(x, w)),
            ('dgrad', lambda: ops.mm(gy, w, False, False), lambda: gy @ w),
            ('wgrad', lambda: ops.mm(gy, x, True, False), lambda: gy.t() @ x),
        ]:
            for hint in (64, 128):
                us = timeit(lambda: (ops.mm(x, w, False, True, tile_hint=hint) if name == 'fwd' else
                                     ops.mm(gy, w, False, False, tile_hint=hint) if name == 'dgrad' else
                                     ops.mm(gy, x, True, False, tile_hint=hint)))
                rec[f'{name}_t{hint}_us'] = round(us, 1)
                rec[f'{name}_t{hint}_TF'] = round(fl / us / 1e6, 1)
            us_ref = timeit(ref)
            rec[f'{name}_torch_us'] = round(us_ref, 1)
            rec[f'{name}_torch_TF'] = round(fl / us_ref / 1e6, 1)
        print(json.dumps(rec), flush=True)


def ew_probe():
    dev = torch.device('cuda')
    x = torch.randn(4096, 5120, device=dev, dtype=torch.bfloat16)
    w = torch.ones(5120, device=dev, dtype=torch.bfloat16)
    n = x.numel() * 2
    for name, fn, passes in [
        ('rmsnorm_fwd', lambda: ops.rms_norm(x, w), 2),
        ('lnmod_fwd', lambda: ops.layer_norm_modulate(x), 2),
        ('gelu_fwd', lambda: ops.gelu_tanh(x), 2),
        ('gated_residual_fwd', lambda: ops.gated_residual(x, x, None), 3),
    ]:
        with torch.no_grad():
            us = timeit(fn)
        print(json.dumps({'probe': name, 'us': round(us, 1), 'GBps': round(passes * n / us / 1e3, 1)}), flush=True)


if __name__ == '__main__':
    print(json.dumps({'device': torch.cuda.get_device_name(0)}))
    which = sys.argv[1:] or ['gemm', 'ew']
    if 'ew' in which:
        ew_probe()
    if 'gemm' in which:
        gemm_probe()
    if 'attn' in which:
        from tools.attn_probe import attn_probe
        attn_probe()
