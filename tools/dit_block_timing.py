"""One real-width DiT block, forward + backward, timed as a hipGraph replay on the product kernels: Wan2.1-14B's block (dim 5120, ffn 13824, 40 heads of 128,
cross attention to 512 text tokens) at several video token counts -- the unit BASELINE configs 3 / 4 repeat 40 times per micro-batch.
Reports ms per block (fwd + bwd incl. weight gradients), algorithmic TFLOP/s and the fraction of the dense bf16 MFMA peak.
Run on the GPU box: python tools/dit_block_timing.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pipe_amd.workloads import wan  # noqa: E402
from tools.kernel_timing import graph_time  # noqa: E402

PEAK = 2500.0   # dense bf16 TFLOP/s (MI355X_MICROARCH.md)


def wan_block_flops(S, L, d, ffn):
    """forward FLOPs of one WanAttentionBlock: self attention (q, k, v, o projections + S x S attention), cross attention (q, o on S tokens; k, v on L text
    tokens; S x L attention), FFN; backward = 2 x forward (dgrad + wgrad for every GEMM, 2.5 x for attention counted as 2 x here)."""
    proj = 2 * S * d * d
    return 4 * proj + 4 * S * S * d + 2 * proj + 2 * 2 * L * d * d + 4 * S * L * d + 2 * 2 * S * d * ffn


def main():
    dev = torch.device('cuda:0')
    dim, ffn, heads, L = 5120, 13824, 40, 512
    d = dim // heads
    block = wan.WanAttentionBlock(dim, ffn, heads, cross_attn_norm=True, eps=1e-6).to(dev, torch.bfloat16)
    for p in block.parameters():
        torch.nn.init.normal_(p, std=0.02)
    freqs = torch.cat([wan.rope_params(1024, d - 4 * (d // 6)), wan.rope_params(1024, 2 * (d // 6)), wan.rope_params(1024, 2 * (d // 6))], dim=1)
    grids = [(9, 32, 16), (9, 32, 32), (21, 30, 52)]
    if len(sys.argv) > 1:
        grids = [grids[int(sys.argv[1])]]          # one grid only (profiling runs)
    for grid in grids:       # 4 608 tokens; 9 216 (512 x 512 x 33 frames); 32 760 (480 x 832 x 81 frames, Wan's own default)
        S = grid[0] * grid[1] * grid[2]
        cos, sin = (t.to(dev) for t in wan.rope_tables(freqs, grid))
        x = (torch.randn(1, S, dim, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
        ctx = torch.randn(1, L, dim, device=dev).to(torch.bfloat16).requires_grad_(True)
        e = (torch.randn(1, 1, 6, dim, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
        gy = (torch.randn(1, S, dim, device=dev) / S).to(torch.bfloat16)

        def fwd_bwd():
            block(x, e, cos, sin, ctx).backward(gy)

        with torch.no_grad():
            f_us = graph_time(lambda: block(x, e, cos, sin, ctx), n=3, reps=3)
        fb_us = graph_time(fwd_bwd, n=3, reps=3)
        fl = wan_block_flops(S, L, dim, ffn)
        print(json.dumps({'block': 'wan2.1-14b', 'tokens': S, 'grid': grid, 'fwd_ms': round(f_us / 1e3, 2), 'fwd_bwd_ms': round(fb_us / 1e3, 2),
                          'fwd_TF': round(fl / f_us / 1e6, 1), 'fwd_bwd_TF': round(3 * fl / fb_us / 1e6, 1), 'mfu_fwd_bwd': round(3 * fl / fb_us / 1e6 / PEAK, 3),
                          'x40_blocks_s': round(40 * fb_us / 1e6, 2)}), flush=True)
        del x, ctx, e, gy
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
