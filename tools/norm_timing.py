"""Per-launch timing of the LayerNorm / GroupNorm kernels at the SDXL shapes (hipGraph of 20 calls, HIP events): forward and forward + backward.
Run on the GPU box: python tools/norm_timing.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pipe_amd import ops  # noqa: E402
from tools.kernel_timing import graph_time  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    for rows, cols in [(1024, 1280), (4096, 640), (77, 1280), (77, 768)]:
        x = torch.randn(1, rows, cols, device=dev, dtype=torch.bfloat16)
        gy = torch.randn_like(x)
        w = torch.ones(cols, device=dev, dtype=torch.bfloat16, requires_grad=True)
        b = torch.zeros(cols, device=dev, dtype=torch.bfloat16, requires_grad=True)
        xr = x.clone().requires_grad_(True)
        f = graph_time(lambda: ops.layer_norm_modulate(x, w, b, None, None, 1e-5))
        fb = graph_time(lambda: ops.layer_norm_modulate(xr, w, b, None, None, 1e-5).backward(gy))
        print(json.dumps({'op': 'layer_norm', 'rows': rows, 'cols': cols, 'fwd_us': round(f, 1), 'bwd_us': round(fb - f, 1)}), flush=True)
    for C, H in [(320, 128), (640, 64), (1280, 32), (1920, 32), (2560, 32)]:
        x = torch.randn(1, C, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = torch.randn_like(x)
        w = torch.ones(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
        b = torch.zeros(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
        xr = x.clone().requires_grad_(True)
        f = graph_time(lambda: ops.group_norm_nhwc(x, 32, w, b, 1e-5, act='silu'))
        fb = graph_time(lambda: ops.group_norm_nhwc(xr, 32, w, b, 1e-5, act='silu').backward(gy))
        print(json.dumps({'op': 'group_norm_nhwc+silu', 'C': C, 'HW': H * H, 'fwd_us': round(f, 1), 'bwd_us': round(fb - f, 1)}), flush=True)


if __name__ == '__main__':
    main()
