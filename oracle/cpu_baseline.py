"""ORACLE / CPU baseline (never imported by the product): times the reference-style fp32 eager path on the host
cores for bench.py's `cpu_baseline` object.

The reference cannot execute on CPU, so the baseline is the oracle restatement
(`"kind": "port"`).  A full SDXL step in fp32 needs ~14 GB of weights and 40+ s on 8 cores, so a BOUNDED sample is
timed: the UNet's second up-block (three resnet + transformer layers at 1/2 latent resolution, 640 channels,
cross-attention to 77 x 2048 text states) forward + backward at the real shapes, then scaled to the whole step by
its share of the step's algorithmic FLOPs.  The sample and the scaling are reported next to the number.
"""
import os
import time

import torch

from . import sdxl_ref


def sdxl_cpu_baseline(cfg, latent_hw=128, budget_s=20.0):
    # many-core hosts (the GPU box has 256 hardware threads) run these mid-sized fp32 ops fastest on a subset
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    ch = list(cfg.block_out_channels)
    temb, g = ch[0] * 4, cfg.norm_groups
    hw = latent_hw // 2 if len(ch) >= 3 else latent_hw
    li = 1 if len(ch) >= 3 else 0                      # UNet level of the sampled block
    out_c, prev_c, in_c = ch[li], ch[min(li + 1, len(ch) - 1)], ch[max(li - 1, 0)]
    depth = cfg.transformer_layers[li]
    n = cfg.layers_per_block + 1
    torch.manual_seed(0)
    layers = []
    for j in range(n):
        skip = in_c if j == n - 1 else out_c
        rin = prev_c if j == 0 else out_c
        resnet = sdxl_ref.ResnetBlock2D(rin + skip, out_c, temb, g)
        attn = sdxl_ref.Transformer2DModel(cfg.num_heads[li], 64, out_c, depth, cfg.cross_attention_dim, g) if depth > 0 else None
        layers.append(sdxl_ref.UpBlockInnerLayer(resnet, attn))
    skips = [torch.randn(1, in_c, hw, hw)] + [torch.randn(1, out_c, hw, hw) for _ in range(n - 1)]
    hidden = torch.randn(1, prev_c, hw, hw, requires_grad=True)
    emb, ctx = torch.randn(1, temb), torch.randn(1, 77, cfg.cross_attention_dim)

    def run_layer(j):
        """forward + backward of pipeline layer j of the block on its own inputs (real shapes)"""
        h = torch.randn(1, prev_c if j == 0 else out_c, hw, hw, requires_grad=True)
        x = (h, torch.zeros(1, dtype=torch.long), emb, ctx, *skips[:n - j], torch.tensor(False))
        layers[j](x)[0].square().mean().backward()

    # algorithmic FLOPs of the sample and of the whole step (same 2 x MAC accounting as the GPU side)
    def lin(i, o, t):
        return 2.0 * i * o * t

    def conv(ci, co, k, r):
        return 2.0 * ci * co * k * k * r * r

    def resnet_f(ci, co):
        return conv(ci, co, 3, hw) + conv(co, co, 3, hw) + lin(temb, co, 1) + (conv(ci, co, 1, hw) if ci != co else 0)

    s = hw * hw
    tr = 0.0
    if depth > 0:
        per = 4 * lin(out_c, out_c, s) + 4.0 * s * s * out_c + 2 * lin(out_c, out_c, s) + 2 * lin(cfg.cross_attention_dim, out_c, 77) \
            + 4.0 * s * 77 * out_c + lin(out_c, 8 * out_c, s) + lin(4 * out_c, out_c, s)
        tr = 2 * lin(out_c, out_c, s) + depth * per
    layer_flops = [3.0 * (resnet_f((prev_c if j == 0 else out_c) + (in_c if j == n - 1 else out_c), out_c) + tr) for j in range(n)]
    # time layers one by one until the budget is used up (always at least one)
    t0 = time.perf_counter()
    best, sample_flops, runs = 0.0, 0.0, 0
    for j in range(n):
        t1 = time.perf_counter()
        run_layer(j)
        dt = time.perf_counter() - t1
        best += dt
        sample_flops += layer_flops[j]
        runs += 1
        if (time.perf_counter() - t0) + dt > budget_s:
            break
    return {'sample_seconds': round(best, 3), 'sample_tflop': round(sample_flops / 1e12, 3), 'cores': threads, 'kind': 'port',
            'cpu_tflops': round(sample_flops / best / 1e12, 3), 'runs': runs,
            'sample': f'oracle fp32 eager fwd+bwd of {runs} of the {n} layers of UNet up-block {li} ([resnet + transformer depth {depth}] at {hw}x{hw}, {out_c} ch), '
                      f'{threads} threads; value = 1 image / (sample_seconds * step_flops / sample_flops)',
            '_sample_flops': sample_flops}
