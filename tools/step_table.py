"""Per-signature table of the GEMM / attention launches of one SDXL micro-batch (fwd + bwd): count, mean launch time of
the HIP kernel and of the PyTorch-ROCm library op of the same shape.  Steers kernel work; not part of the product."""
import json
import sys
from collections import Counter

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def main():
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    dev = torch.device('cuda:0')
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'gradient_accumulation_steps': 1, 'gradient_clipping': 1.0}, device=dev)
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=1e-6), [p for p in module.parameters()])
    feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=1, latent_hw=128, seed=1))
    micro = split_batch((feats, label), 1)
    engine.train_batch(iter(micro))
    ops.GEMM_TRACE, ops.ATTN_TRACE = [], []
    engine.train_batch(iter(micro))
    gt, at = Counter(ops.GEMM_TRACE), Counter(ops.ATTN_TRACE)
    ops.GEMM_TRACE = ops.ATTN_TRACE = None
    torch.cuda.synchronize()
    del engine, module, work
    torch.cuda.empty_cache()
    rows = []
    for sig, cnt in gt.items():
        dt, ta, tb, M, N, K, batch, has_bias, act, accumulate, out_f32, hint = sig
        tdt = torch.bfloat16 if dt == 0 else torch.float32
        a = torch.randn((batch, K, M) if ta else (batch, M, K), device=dev, dtype=tdt)
        b = torch.randn((batch, N, K) if tb else (batch, K, N), device=dev, dtype=tdt)
        c = torch.zeros((batch, M, N), device=dev, dtype=torch.float32 if out_f32 else tdt)
        bias = torch.randn(N, device=dev, dtype=tdt) if has_bias else None

        def mine():
            ops.gemm(a, b, ta, tb, M, N, K, c, lda=a.shape[2], ldb=b.shape[2], ldc=N, batch_outer=batch, batch_inner=1,
                     stride_a=(a.shape[1] * a.shape[2], 0), stride_b=(b.shape[1] * b.shape[2], 0), stride_c=(M * N, 0),
                     bias=bias, act=act, accumulate=bool(accumulate), tile_hint=hint)
        aa = a.transpose(1, 2) if ta else a
        bb = b.transpose(1, 2) if tb else b
        us = timeit(mine, iters=10, warmup=3)
        ust = timeit(lambda: torch.bmm(aa, bb), iters=10, warmup=3)
        fl = 2.0 * M * N * K * batch
        rows.append({'op': 'gemm', 'ta': ta, 'tb': tb, 'M': M, 'N': N, 'K': K, 'batch': batch, 'bias': has_bias, 'act': act, 'cnt': cnt,
                     'us': round(us, 1), 'torch_us': round(ust, 1), 'TF': round(fl / us / 1e6, 1), 'tot_ms': round(cnt * us / 1e3, 3),
                     'torch_tot_ms': round(cnt * ust / 1e3, 3)})
    for sig, cnt in at.items():
        B, Sq, Sk, H, D, causal = sig
        q = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        k = torch.randn(B, Sk, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        v = torch.randn(B, Sk, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        go = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            us_f = timeit(lambda: ops.attention(q, k, v, impl='flash', causal=bool(causal)), iters=10, warmup=3)
            us_t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=bool(causal)), iters=10, warmup=3)
        o = ops.attention(q, k, v, impl='flash', causal=bool(causal))
        us_b = timeit(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True), iters=5, warmup=2)
        qt, kt, vt = (t.detach().transpose(1, 2).requires_grad_(True) for t in (q, k, v))
        ot = torch.nn.functional.scaled_dot_product_attention(qt, kt, vt, is_causal=bool(causal))
        us_bt = timeit(lambda: torch.autograd.grad(ot, (qt, kt, vt), go.transpose(1, 2), retain_graph=True), iters=5, warmup=2)
        rows.append({'op': 'attn', 'B': B, 'Sq': Sq, 'Sk': Sk, 'H': H, 'D': D, 'causal': causal, 'cnt': cnt, 'fwd_us': round(us_f, 1), 'torch_fwd_us': round(us_t, 1),
                     'bwd_us': round(us_b, 1), 'torch_bwd_us': round(us_bt, 1), 'tot_ms': round(cnt * (us_f + us_b) / 1e3, 3),
                     'torch_tot_ms': round(cnt * (us_t + us_bt) / 1e3, 3)})
    rows.sort(key=lambda r: -r['tot_ms'])
    for r in rows:
        print(json.dumps(r), flush=True)
    for op in ('gemm', 'attn'):
        sel = [r for r in rows if r['op'] == op]
        print(json.dumps({'summary': op, 'launch_sigs': len(sel), 'tot_ms': round(sum(r['tot_ms'] for r in sel), 2),
                          'torch_tot_ms': round(sum(r['torch_tot_ms'] for r in sel), 2)}), flush=True)


main()
