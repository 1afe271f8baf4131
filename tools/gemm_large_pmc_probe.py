"""Eager launches of one tile configuration of the pipelined GEMM on DiT-sized shapes, for rocprofv3 --pmc passes
(hipGraph replays crash the counter collector).  DPIPE_PROBE_HINT selects the configuration (tile_hint of dpipe_gemm_ex)."""
import os
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
hint = int(os.environ.get('DPIPE_PROBE_HINT', '7001'))
for (ta, tb, M, N, K) in [(0, 1, 8192, 8192, 8192), (0, 1, 4096, 4096, 4096), (0, 0, 4608, 3072, 9216), (1, 0, 13824, 5120, 9216)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        ops.mm(a, b, bool(ta), bool(tb), out=out, tile_hint=hint)
    torch.cuda.synchronize()
    print(ta, tb, M, N, K)
