"""Eager launches of the pipelined GEMM on the step's dominant shapes with HBM-cold weights, for a rocprofv3 --pmc pass
(FETCH_SIZE / WRITE_SIZE per dispatch; hipGraph replays crash the counter collector on this ROCm, eager launches do not).
Each shape: 8 launches over 8 distinct weight buffers after a 512 MiB cache flush."""
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (ta, tb, M, N, K) in [(0, 1, 1024, 1280, 1280), (0, 1, 1024, 10240, 1280), (0, 0, 1024, 5120, 1280), (1, 0, 10240, 1280, 1024), (0, 1, 8192, 8192, 8192)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    ws = [torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16) for _ in range(8)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    flush.zero_()
    for w in ws:
        ops.mm(a, w, bool(ta), bool(tb), out=out)
    torch.cuda.synchronize()
    print(ta, tb, M, N, K, 'algorithmic bytes', (M * K + N * K + M * N) * 2)
