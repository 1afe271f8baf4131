"""Debug probe (round 5): nn.Linear forward + backward at tiny M under hipGraph capture / replay with fresh inputs, against an fp32 reference; FUSE_GRAD_ACCUM as in the engine."""
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    ops.FUSE_GRAD_ACCUM = True
    for M in (1, 2, 3, 4):
        for (fin, fout) in ((1280, 320), (1280, 640), (1280, 1280), (2816, 1280), (1280, 5120)):
            lin = torch.nn.Linear(fin, fout).to(dev, torch.bfloat16)
            lin.weight.grad = torch.zeros_like(lin.weight); lin.bias.grad = torch.zeros_like(lin.bias)
            x = torch.randn(M, fin, device=dev, dtype=torch.bfloat16, requires_grad=True)
            gy = torch.randn(M, fout, device=dev, dtype=torch.bfloat16)
            gx = torch.zeros_like(x)

            def body():
                xx = x.detach().requires_grad_(True)
                y = ops.linear(xx, lin.weight, lin.bias)
                y.backward(gy)
                gx.copy_(xx.grad)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body(); body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            worst = {}
            for rep in range(3):
                with torch.no_grad():
                    x.copy_(torch.randn_like(x)); gy.copy_(torch.randn_like(gy))
                    lin.weight.grad.zero_(); lin.bias.grad.zero_()
                # junk in freed memory: the graph's private pool is not zeroed between replays -- poison the caching allocator's free blocks too
                junk = torch.full((64 << 20,), float('nan'), device=dev, dtype=torch.float32); del junk
                g.replay()
                torch.cuda.synchronize()
                want_gx = gy.float() @ lin.weight.float()
                want_gw = gy.float().t() @ x.float()
                want_gb = gy.float().sum(0)
                for nm, got, want in (('dx', gx, want_gx), ('dW', lin.weight.grad, want_gw), ('db', lin.bias.grad, want_gb)):
                    err = ((got.float() - want).abs().max() / want.abs().max().clamp_min(1e-6)).item() if bool(torch.isfinite(got).all()) else float('nan')
                    worst[nm] = max(worst.get(nm, 0.0), err) if err == err else float('nan')
            print(f'M={M} {fin}->{fout}: ' + ' '.join(f'{k} {v:.3e}' for k, v in worst.items()), flush=True)


if __name__ == '__main__':
    main()
