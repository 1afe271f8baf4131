"""Eager flash-attention launches (SDXL self-attention shapes) for a rocprofv3 --pmc pass."""
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
for (S, H) in [(1024, 20), (4096, 10)]:
    q, k, v = (torch.randn(1, S, H, 64, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    go = torch.randn(1, S, H, 64, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        o = ops.attention(q, k, v, impl='flash')
        o.backward(go)
    torch.cuda.synchronize()
