import json

import torch

from diffusion_pipe_amd import ops
from tools.gpu_probe import timeit


def attn_probe():
    dev = torch.device('cuda')
    for (B, Sq, Sk, H, D) in [(1, 4096, 4096, 10, 64), (1, 1024, 1024, 20, 64), (1, 4096, 77, 10, 64), (1, 1024, 77, 20, 64),
                              (1, 4608, 4608, 24, 128), (1, 9216, 9216, 40, 128)]:
        q = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        k = torch.randn(B, Sk, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        v = torch.randn(B, Sk, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        go = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16)
        fl = 4.0 * B * H * Sq * Sk * D
        with torch.no_grad():
            us_f = timeit(lambda: ops.attention(q, k, v, impl='flash'), iters=10, warmup=3)
            us_t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)), iters=10, warmup=3)
        o = ops.attention(q, k, v, impl='flash')
        us_b = timeit(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True), iters=5, warmup=2)
        qt, kt, vt = (t.detach().transpose(1, 2).requires_grad_(True) for t in (q, k, v))
        ot = torch.nn.functional.scaled_dot_product_attention(qt, kt, vt)
        got = go.transpose(1, 2)
        us_bt = timeit(lambda: torch.autograd.grad(ot, (qt, kt, vt), got, retain_graph=True), iters=5, warmup=2)
        print(json.dumps({'probe': 'attn', 'B': B, 'Sq': Sq, 'Sk': Sk, 'H': H, 'D': D, 'fwd_us': round(us_f, 1), 'fwd_TF': round(fl / us_f / 1e6, 1),
                          'torch_fwd_us': round(us_t, 1), 'torch_fwd_TF': round(fl / us_t / 1e6, 1),
                          'bwd_us': round(us_b, 1), 'bwd_TF': round(2.5 * fl / us_b / 1e6, 1), 'torch_bwd_us': round(us_bt, 1)}), flush=True)
