"""Debug probe (round 5): the full-size SDXL step with stacked micro-batches returned NaN (profiles/r5a_*).  Runs the pipeline layers eagerly on a batch of B samples and on
each sample alone, reports per layer whether outputs are finite and how far sample 0 of the batched pass is from the single-sample pass, then does the same for every
parameter gradient after loss + backward.

    python tools/stack_debug.py [B]"""
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd.data import split_batch, stack_micro_batches  # noqa: E402
from diffusion_pipe_amd.workloads import sdxl  # noqa: E402


def flat(x):
    return [t for t in (x if isinstance(x, (tuple, list)) else (x,)) if torch.is_tensor(t) and t.is_floating_point()]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device('cuda:0')
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    torch.manual_seed(1234)
    feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=B, latent_hw=128, seed=100))
    singles = [tuple(tuple(t.to(dev) for t in part) for part in mb) for mb in split_batch((feats, label), B)]      # exactly what the engine receives
    feats, label = stack_micro_batches(singles, B)[0]                                                               # ... and what it stacks them into
    layers = work.to_layers()
    loss_fn = work.get_loss_fn()

    def run(f, l, names=None):
        x = f
        outs = []
        for i, layer in enumerate(layers):
            x = layer(x)
            outs.append([t.detach() for t in flat(x)])
        loss = loss_fn(x, l)
        return outs, loss

    params = [p for m in work.modules().values() for p in m.parameters()]
    for p in params:
        p.grad = None
    outs_b, loss_b = run(feats, label)
    loss_b.backward()
    torch.cuda.synchronize()
    gb = {n: p.grad.detach().clone() for k, m in work.modules().items() for n, p in ((f'{k}.{n_}', p_) for n_, p_ in m.named_parameters()) if p.grad is not None}
    print('batched loss', float(loss_b), flush=True)
    for p in params:
        p.grad = None
    outs_0, loss_0 = run(*singles[0])
    print('single-sample-0 loss', float(loss_0), flush=True)
    for i, (ob, o0) in enumerate(zip(outs_b, outs_0)):
        fin = all(bool(torch.isfinite(t).all()) for t in ob)
        worst = 0.0
        for tb, t0 in zip(ob, o0):
            if tb.shape[0] == B and t0.shape[0] == 1 and tb.shape[1:] == t0.shape[1:]:
                d = (tb[:1].float() - t0.float()).abs().max().item() / (t0.float().abs().max().item() + 1e-6)
                worst = max(worst, d)
        print(f'layer {i:2d} {type(layers[i]).__name__:24s} finite={fin} sample0 max rel diff vs single pass {worst:.3e}', flush=True)
    bad = [n for n, g in gb.items() if not bool(torch.isfinite(g).all())]
    print('parameters with non-finite gradients (batched):', len(bad), 'of', len(gb))
    for n in bad[:40]:
        print('  ', n)


if __name__ == '__main__':
    main()
