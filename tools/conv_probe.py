"""MIOpen convolution timings at the SDXL shapes (bf16), NCHW vs channels_last, fwd and fwd+bwd."""
import json
import sys

import torch

sys.path.insert(0, '.')
from tools.gpu_probe import timeit  # noqa: E402

dev = torch.device('cuda')
shapes = [(320, 320, 128, 3), (640, 640, 64, 3), (1280, 1280, 32, 3), (2560, 1280, 32, 3), (1920, 1280, 32, 3), (1920, 640, 64, 3), (1280, 640, 64, 3),
          (960, 640, 64, 3), (960, 320, 128, 3), (640, 320, 128, 3), (2560, 1280, 32, 1), (960, 320, 128, 1), (320, 640, 64, 3), (640, 1280, 32, 3)]
for cin, cout, hw, k in shapes:
    rec = {'cin': cin, 'cout': cout, 'hw': hw, 'k': k, 'gflop_fwd': round(2 * cin * cout * k * k * hw * hw / 1e9, 1)}
    for fmt in ('nchw', 'nhwc'):
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2).to(dev, torch.bfloat16)
        x = torch.randn(1, cin, hw, hw, device=dev, dtype=torch.bfloat16, requires_grad=True)
        if fmt == 'nhwc':
            conv = conv.to(memory_format=torch.channels_last)
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.no_grad():
            rec[f'{fmt}_fwd_us'] = round(timeit(lambda: conv(x), iters=10, warmup=3), 1)

        def fb():
            y = conv(x)
            y.backward(y.detach())
        rec[f'{fmt}_fwdbwd_us'] = round(timeit(fb, iters=5, warmup=3), 1)
    print(json.dumps(rec), flush=True)
