"""Eager launches for rocprofv3 --pmc passes over the flash-attention kernels (SDXL self / cross attention at head dim 64, Flux-sized head dim 128) and the
implicit-GEMM convolution instances (CONV = 1 forward / dgrad, CONV = 2 wgrad) on SDXL UNet shapes -- the SQ / LDS counter evidence round 2's review asked for.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \\
        -f csv -d out -o pmc -- python tools/attn_conv_pmc_probe.py ; python tools/pmc_agg.py out out_agg.csv"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pipe_amd import nn as dnn, ops  # noqa: E402

dev = torch.device('cuda:0')
bf16 = torch.bfloat16
# (Sq, Sk, heads, head dim)
for (Sq, Sk, H, D) in [(1024, 1024, 20, 64), (4096, 4096, 10, 64), (1024, 77, 20, 64), (4096, 77, 10, 64), (4608, 4608, 24, 128)]:
    q = torch.randn(1, Sq, H, D, device=dev, dtype=bf16, requires_grad=True)
    k, v = (torch.randn(1, Sk, H, D, device=dev, dtype=bf16, requires_grad=True) for _ in range(2))
    go = torch.randn(1, Sq, H, D, device=dev, dtype=bf16)
    for _ in range(3):
        o = ops.attention(q, k, v, impl='flash')
        o.backward(go)
    torch.cuda.synchronize()
# (H = W, Cin, Cout, kernel, stride)
for (hw, cin, cout, ksz, stride) in [(32, 1280, 1280, 3, 1), (64, 640, 640, 3, 1), (128, 320, 320, 3, 1), (64, 640, 640, 3, 2)]:
    conv = dnn.Conv2d(cin, cout, ksz, stride=stride, padding=ksz // 2).to(dev, bf16)
    x = torch.randn(1, cin, hw, hw, device=dev).to(bf16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(3):
        y = conv(x)
        y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
print('done')
