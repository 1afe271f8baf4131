"""Debug probe (round 5): which parameter gradients go non-finite when the full-size SDXL step runs STACKED micro-batches under hipGraph (eager: finite, tools/stack_debug.py).
    python tools/stack_debug_graph.py [stack] [lanes]"""
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd.data import split_batch  # noqa: E402
from diffusion_pipe_amd.engine import ManualPipelineModule, initialize  # noqa: E402
from diffusion_pipe_amd.workloads import sdxl  # noqa: E402


def main():
    stack = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    gas = stack * lanes
    dev = torch.device('cuda:0')
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 0.0, 'steps_per_print': 1 << 30,
                                                         'hip_graph': True, 'graph_lanes': lanes, 'stack_micro_batches': stack, 'store_first_micro_batch': False}, device=dev)
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), [p for p in module.parameters()])
    names = {id(p): f'{k}.{n}' for k, m in work.modules().items() for n, p in m.named_parameters()}
    # gradients entering the text encoders: copied into persistent buffers by hooks (the copy kernels are captured with the graph)
    probes = {}
    init_layer = module.forward_funcs[0]
    orig_cond = init_layer.get_text_conditioning

    def cond(input_ids, input_ids_2):
        e1, _ = init_layer.get_prompt_embeds(input_ids, init_layer.text_encoder, False)
        e2, pooled = init_layer.get_prompt_embeds(input_ids_2, init_layer.text_encoder_2, True)
        for nm, t in (('e1', e1), ('e2', e2), ('pooled', pooled)):
            buf = probes.setdefault(nm, torch.zeros_like(t))
            t.register_hook(lambda g, b=buf: (b.copy_(g), None)[1])
        return torch.cat([e1, e2], dim=-1), pooled
    init_layer.get_text_conditioning = cond

    def wrap(mod, nm):
        orig = mod.forward

        def fwd(x, *a, **k):
            if torch.is_tensor(x) and x.requires_grad:
                buf = probes.setdefault(nm + '.in', torch.zeros_like(x))
                x.register_hook(lambda g, b=buf: (b.copy_(g), None)[1])
            y = orig(x, *a, **k)
            buf = probes.setdefault(nm + '.out', torch.zeros_like(y))
            y.register_hook(lambda g, b=buf: (b.copy_(g), None)[1])
            return y
        mod.forward = fwd
    ae = init_layer.add_embedding
    wrap(ae, 'add_embedding')
    for nm, sub in ae.named_children():
        wrap(sub, 'add_embedding.' + nm)
    wrap(init_layer.time_embedding, 'time_embedding')
    n_res = 0
    for mname, mod in work.unet.named_modules():
        if type(mod).__name__ == 'ResnetBlock2D' and n_res < 40:
            wrap(mod.time_emb_proj, f'res{n_res:02d}.time_emb_proj({mname})')
            n_res += 1
    torch.manual_seed(1234)
    for step in range(2):
        feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=128, seed=100 + step))
        micro = [tuple(tuple(t.to(dev) for t in part) for part in mb) for mb in split_batch((feats, label), gas)]
        # peek at the gradients BEFORE the optimizer's zero_grad: hook the optimizer step
        seen = {}
        orig = engine.optimizer.step

        def spy(*a, **k):
            for p in module.parameters():
                if p.grad is not None:
                    seen[id(p)] = bool(torch.isfinite(p.grad).all())
            return orig(*a, **k)
        engine.optimizer.step = spy
        loss = engine.train_batch(iter(micro))
        torch.cuda.synchronize()
        engine.optimizer.step = orig
        bad = [names[i] for i, ok in seen.items() if not ok]
        print(f'step {step}: loss {float(loss):.6f}, {len(bad)} / {len(seen)} parameters with non-finite gradients', flush=True)
        for n in bad[:3]:
            print('    ', n)
        for nm, b in probes.items():
            fin = torch.isfinite(b)
            cols = (~fin).any(dim=tuple(range(b.dim() - 1))).nonzero().flatten().tolist()
            rows = (~fin).reshape(-1, b.shape[-1]).any(dim=1).nonzero().flatten().tolist()
            if bool(fin.all()) and b[fin].abs().max().item() < 1e3:
                continue
            print(f'    grad into {nm} {tuple(b.shape)}: finite={bool(fin.all())}, bad columns {cols[:6]}..{cols[-3:]} ({len(cols)}), bad rows {rows[:6]}..{rows[-3:]} ({len(rows)}), absmax finite part {b[fin].abs().max().item() if fin.any() else None}', flush=True)


if __name__ == '__main__':
    main()
