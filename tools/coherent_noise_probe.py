"""CPU probe (no GPU): WHICH bf16 roundings of the SDXL forward move the global gradient norm?  The oracle's fp32 eager step is evaluated with forward hooks that round the
outputs of chosen module groups to bf16 (straight-through gradient), on the bench's own parity samples (same seeds as bench.py's pool):

    python tools/coherent_noise_probe.py <out.jsonl> [n_samples] [groups ...]

groups: none (fp32 baseline), all (every Linear / Conv2d / GroupNorm / LayerNorm output = the bf16 activation path), temb (time_embedding, add_embedding, time_emb_proj:
the per-channel addends shared by every pixel), ctx (text-encoder outputs + attn2.to_k / to_v), all-temb (all but the addends), all-temb-ctx."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pipe_amd.data import split_batch          # noqa: E402
from diffusion_pipe_amd.workloads import sdxl            # noqa: E402
from oracle import eager_step, sdxl_ref                  # noqa: E402


def bench_samples(cfg, n, gas=8, latent=128):
    w = sdxl.SDXLWorkload.__new__(sdxl.SDXLWorkload)
    w.cfg, w.v_pred = cfg, False
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, cfg.num_train_timesteps, dtype=torch.float32) ** 2
    w.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    torch.manual_seed(1234)
    pool = []
    for s in range(3):
        feats, label = w.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=latent, seed=100 + s))
        pool.append(split_batch((feats, label), gas))
    return [pool[j // gas][j % gas] for j in range(n)]


def classify(name, mod):
    g = set()
    leaf = isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.GroupNorm, torch.nn.LayerNorm))
    if leaf:
        g.add('all')
    is_temb = leaf and ('time_emb_proj' in name or 'time_embedding' in name or 'add_embedding' in name)
    is_ctx = (leaf and ('attn2.to_k' in name or 'attn2.to_v' in name)) or (leaf and name.startswith('te'))
    if is_temb:
        g.add('temb')
    if is_ctx:
        g.add('ctx')
    if leaf and not is_temb:
        g.add('all-temb')
    if leaf and not is_temb and not is_ctx:
        g.add('all-temb-ctx')
    return g


def main():
    out_path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
    groups = sys.argv[3:] or ['none', 'all', 'temb', 'all-temb']
    cfg = sdxl.SDXLConfig()
    torch.set_num_threads(os.cpu_count())
    ref = sdxl_ref.SDXLRef(cfg, seed=0)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(torch.bfloat16).float())              # the product's weights are bf16 values
    layers = ref.to_layers()
    named = [(f'unet.{k}', m) for k, m in ref.unet.named_modules()] + [(f'te1.{k}', m) for k, m in ref.text_encoder.named_modules()] + \
            [(f'te2.{k}', m) for k, m in ref.text_encoder_2.named_modules()]
    active = {'g': 'none'}

    def mk(gs):
        def hook(_m, _i, o):
            return o.to(torch.bfloat16).to(torch.float32) if active['g'] in gs else None
        return hook
    for name, m in named:
        gs = classify(name, m)
        if gs:
            m.register_forward_hook(mk(gs))
    samples = bench_samples(cfg, n)
    base = {}
    with open(out_path, 'a') as f:
        for si, mb in enumerate(samples):
            for g in groups:
                active['g'] = g
                for p in ref.parameters():
                    p.grad = None
                t0 = time.perf_counter()
                loss, norm = eager_step.eager_train_step(layers, eager_step.sdxl_loss_fn(), [mb], None, gradient_clipping=0.0, params=ref.parameters())
                loss, norm = float(loss), float(norm)
                if g == 'none':
                    base[si] = (loss, norm)
                row = {'sample': si, 'timestep': int(mb[0][1][0]), 'group': g, 'loss': loss, 'grad_norm': norm, 'seconds': round(time.perf_counter() - t0, 1)}
                if si in base:
                    row['loss_rel'] = (loss - base[si][0]) / base[si][0]
                    row['grad_norm_rel'] = (norm - base[si][1]) / base[si][1]
                print(json.dumps(row), flush=True)
                f.write(json.dumps(row) + '\n'); f.flush()


if __name__ == '__main__':
    main()
