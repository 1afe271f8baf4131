"""Debug probe (round 5): CLIP-G (text_encoder_2) forward + backward at batch B under hipGraph capture / replay with fresh token ids; reports non-finite gradients per variant
(which output feeds the loss) to localise the stacked-step NaN.    python tools/clip_graph_debug.py [B]"""
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402
from diffusion_pipe_amd.workloads import sdxl  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device('cuda:0')
    cfg = sdxl.SDXLConfig()
    torch.manual_seed(0)
    for which in ('te2', 'te1'):
        te = sdxl.CLIPTextModel(cfg.te2 if which == 'te2' else cfg.te1).to(dev, torch.bfloat16)
        c = te.config
        names = {id(p): n for n, p in te.named_parameters()}
        for variant in ('hidden', 'pooled', 'both'):
            if variant != 'hidden' and which == 'te1':
                continue
            ids = torch.randint(1000, 40000, (B, 77), device=dev)
            ids[:, 0] = c.bos; ids[:, -1] = c.eos
            static_ids = ids.clone()

            def body():
                hidden, pooled = te(static_ids, want_pooled=variant != 'hidden')
                loss = 0
                if variant != 'pooled':
                    loss = loss + hidden.float().square().mean()
                if variant != 'hidden':
                    loss = loss + pooled.float().square().mean()
                loss.backward()
            for p in te.parameters():
                p.grad = None
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body(); body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            for rep in range(3):
                for p in te.parameters():
                    if p.grad is not None:
                        p.grad.zero_()
                static_ids.copy_(torch.randint(1000, 40000, (B, 77), device=dev)); static_ids[:, 0] = c.bos; static_ids[:, -1] = c.eos
                g.replay()
                torch.cuda.synchronize()
                bad = [names[id(p)] for p in te.parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
                print(f'{which} B={B} loss on {variant:6s} replay {rep}: {len(bad)} non-finite of {len(names)}: {bad[-4:]}', flush=True)
            del g


if __name__ == '__main__':
    main()
