"""In-graph timing of the implicit-GEMM convolution (forward / dgrad / wgrad) on the SDXL UNet's convolution shapes under forced tile configurations,
next to torch's bf16 channels-last convolution (MIOpen), HBM-cold weights (the graph cycles through enough weight copies to exceed the Infinity Cache).
Prints JSON lines: us per launch and TFLOP/s."""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from diffusion_pipe_amd import nn as dnn, ops  # noqa: E402
from tools.kernel_timing import graph_time     # noqa: E402

# (H = W, Cin, Cout, k, stride, upsample, count per micro-batch forward)
SHAPES = [(32, 1280, 1280, 3, 1, 1, 8), (32, 2560, 1280, 3, 1, 1, 3), (32, 1920, 1280, 3, 1, 1, 1), (64, 640, 640, 3, 1, 1, 8), (64, 1280, 640, 3, 1, 1, 2),
          (64, 960, 640, 3, 1, 1, 1), (64, 1920, 640, 3, 1, 1, 1), (128, 320, 320, 3, 1, 1, 9), (128, 640, 320, 3, 1, 1, 3), (128, 960, 320, 3, 1, 1, 1),
          (128, 320, 320, 3, 2, 1, 1), (64, 640, 640, 3, 2, 1, 1), (32, 1280, 1280, 3, 1, 2, 1), (64, 640, 640, 3, 1, 2, 1), (64, 320, 640, 1, 1, 1, 1)]
HINTS = [('auto', 0), ('t64', 2001), ('t128', 3001), ('t128r2', 4001), ('t128k2', 3002), ('t128k3', 3003), ('t128r2k2', 4002)]      # [forward, dgrad, wgrad] us per launch each
# (round 5: the calls follow the current C ABI again -- `flags` / `out_f32` arguments added in round 3 had turned every measurement of this tool into an exception)


def main():
    dev = torch.device('cuda:0')
    for (hw, cin, cout, k, stride, ups, cnt) in SHAPES:
        pad = k // 2
        wbytes = cout * cin * k * k * 2
        nbuf = max(2, min(64, -(-600_000_000 // wbytes)))
        convs = [dnn.Conv2d(cin, cout, k, stride=stride, padding=pad).to(dev, torch.bfloat16) for _ in range(nbuf)]
        x = torch.randn(1, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y0 = convs[0](x, upsample=ups)
        gy = torch.randn_like(y0)
        ho = y0.shape[2]
        fl = 2.0 * ho * ho * cout * cin * k * k
        rec = {'op': 'conv', 'hw': hw, 'cin': cin, 'cout': cout, 'k': k, 'stride': stride, 'ups': ups, 'cnt': cnt, 'gflop': round(fl / 1e9, 2)}
        from diffusion_pipe_amd.hip import check, lib, ptr, stream
        xv = ops.nhwc_view(x)
        gyv = ops.nhwc_view(gy)
        y = torch.empty_like(gyv)
        hi = hw * ups
        dx = torch.empty((1, hi, hi, cin), device=dev, dtype=torch.bfloat16)
        dws = [torch.zeros_like(c.weight) for c in convs]
        db = torch.zeros(cout, device=dev, dtype=torch.bfloat16)
        ws = ops._splitk_workspace(dev)
        for name, hint in HINTS:
            def fwd():
                for c in convs:
                    check(lib().dpipe_conv2d_fwd(ptr(xv), cin, ptr(c.weight), ptr(c.bias), None, cout, ptr(y), cout, 1, hw, hw, cin, cout, k, k, stride, pad, ups, 0, 0,
                                                 ptr(ws), ws.numel(), hint, stream()), 'fwd')

            def dgrad():
                for c in convs:
                    check(lib().dpipe_conv2d_dgrad(ptr(gyv), cout, ptr(c.weight), ptr(dx), cin, 1, hi, hi, cin, cout, k, k, stride, pad, 0, ptr(ws), ws.numel(), hint, stream()), 'dgrad')

            def wgrad():
                for c, dw in zip(convs, dws):
                    check(lib().dpipe_conv2d_wgrad(ptr(gyv), cout, ptr(xv), cin, ptr(dw), ptr(db), 1, hw, hw, cin, cout, k, k, stride, pad, ups, 1, 1, 0,
                                                   ptr(ws), ws.numel(), hint, stream()), 'wgrad')
            res = []
            for fn in (fwd, dgrad, wgrad):
                try:
                    res.append(round(graph_time(fn, n=1, reps=3) / nbuf, 1))
                except Exception as e:
                    res.append(None)
                    rec.setdefault('errors', set()).add(repr(e)[:120])
            rec[name] = res
        xt = x if ups == 1 else F.interpolate(x, scale_factor=2.0, mode='nearest').contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            rec['torch_fwd'] = round(graph_time(lambda: [F.conv2d(xt, c.weight, c.bias, stride=stride, padding=pad) for c in convs], n=1, reps=3) / nbuf, 1)
        rec['auto_TF'] = [round(fl / u / 1e6) if u else None for u in rec['auto']]
        if 'errors' in rec:
            rec['errors'] = sorted(rec['errors'])
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
