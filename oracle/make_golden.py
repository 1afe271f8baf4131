"""ORACLE tooling (test infrastructure): mint golden vectors from the REFERENCE'S OWN code.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  The reference's in-tree Wan DiT
(models/wan/model.py -- the one place where the hot-path block arithmetic K1..K7 lives in the reference tree, SURVEY.md
section 2a row 7) is imported unmodified on CPU: `diffusers` (absent offline) is stubbed with the three names the file
imports, and `flash_attention` (CUDA-only flash-attn) is rebound to the reference's own SDPA fallback
models/wan/attention.py:128-174 `attention(..., dtype=torch.float32)` (exact for unpadded batches).

    python -m oracle.make_golden            # writes tests/golden/wan_block_fp32.safetensors (+ .json manifest)

The vectors pin (a) oracle/blocks_ref.py on CPU (tests/test_golden_cpu.py) and (b) the HIP-kernel Wan block of
diffusion_pipe_amd/workloads/wan.py on the MI355X (tests/test_gpu_wan.py).
"""
import json
import os
import sys
import types

import torch

REF = '/root/reference'
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

CASE = dict(dim=128, ffn_dim=256, num_heads=2, grid=(2, 6, 8), ctx_len=40, eps=1e-6, seed=1234)


def import_reference_wan():
    """models.wan.model from /root/reference with the minimal import shims (no reference source is copied)."""
    if not os.path.isdir(REF):
        raise SystemExit(f'{REF} not found: golden vectors are minted in the build container only')
    import torch.nn as nn
    cfg = types.ModuleType('diffusers.configuration_utils')

    class ConfigMixin:
        pass

    def register_to_config(fn):
        return fn
    cfg.ConfigMixin, cfg.register_to_config = ConfigMixin, register_to_config
    mu = types.ModuleType('diffusers.models.modeling_utils')
    mu.ModelMixin = nn.Module
    for name, mod in (('diffusers', types.ModuleType('diffusers')), ('diffusers.configuration_utils', cfg),
                      ('diffusers.models', types.ModuleType('diffusers.models')), ('diffusers.models.modeling_utils', mu)):
        sys.modules.setdefault(name, mod)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    attn = importlib.import_module('models.wan.attention')
    model = importlib.import_module('models.wan.model')
    model.flash_attention = lambda q, k, v, k_lens=None, window_size=(-1, -1): attn.attention(q, k, v, k_lens=None, dtype=torch.float32)
    return model


def main():
    m = import_reference_wan()
    c = CASE
    torch.manual_seed(c['seed'])
    dim, heads = c['dim'], c['num_heads']
    block = m.WanAttentionBlock('default', dim, c['ffn_dim'], heads, (-1, -1), True, True, c['eps']).float()
    head = m.Head(dim, 16, (1, 2, 2), c['eps']).float()
    with torch.no_grad():       # non-trivial norm weights (the constructors leave them at one / zero)
        for mod in (block.self_attn.norm_q, block.self_attn.norm_k, block.cross_attn.norm_q, block.cross_attn.norm_k, block.norm3):
            mod.weight.uniform_(0.5, 1.5)
        block.norm3.bias.uniform_(-0.2, 0.2)
    f, h, w = c['grid']
    S = f * h * w
    d = dim // heads
    freqs = torch.cat([m.rope_params(1024, d - 4 * (d // 6)), m.rope_params(1024, 2 * (d // 6)), m.rope_params(1024, 2 * (d // 6))], dim=1)
    x = torch.randn(1, S, dim, requires_grad=True)
    e = (torch.randn(1, 1, 6, dim) * 0.5).requires_grad_(True)
    ctx = torch.randn(1, c['ctx_len'], dim, requires_grad=True)
    e_head = torch.randn(1, 1, dim, requires_grad=True)
    seq_lens = torch.tensor([S])
    grid_sizes = torch.tensor([[f, h, w]])
    wy = torch.randn(1, S, dim)
    wh = torch.randn(1, S, 64)
    y = block(x, e, seq_lens, grid_sizes, freqs, ctx, None)
    out = head(y, e_head)
    loss = (y * wy).sum() + (out * wh).sum()
    loss.backward()
    t = m.sinusoidal_embedding_1d(256, torch.tensor([17.0, 500.0, 999.0]))
    tensors = {'in.x': x, 'in.e': e, 'in.context': ctx, 'in.e_head': e_head, 'in.wy': wy, 'in.wh': wh,
               'in.freqs_re': freqs.real, 'in.freqs_im': freqs.imag,
               'out.y': y, 'out.head': out, 'out.loss': loss.reshape(1), 'out.sinusoidal_256': t,
               'grad.x': x.grad, 'grad.e': e.grad, 'grad.context': ctx.grad, 'grad.e_head': e_head.grad}
    for n, p in block.named_parameters():
        tensors[f'block.{n}'] = p
        tensors[f'grad.block.{n}'] = p.grad
    for n, p in head.named_parameters():
        tensors[f'head.{n}'] = p
        tensors[f'grad.head.{n}'] = p.grad
    from safetensors.torch import save_file
    os.makedirs(OUT_DIR, exist_ok=True)
    save_file({k: v.detach().contiguous().float() for k, v in tensors.items()}, os.path.join(OUT_DIR, 'wan_block_fp32.safetensors'))
    json.dump({'case': c, 'source': 'models/wan/model.py:WanAttentionBlock.forward (:277-312), Head.forward (:332-343), rope_params/rope_apply (:29-67), '
                                  'sinusoidal_embedding_1d (:15-25); attention = models/wan/attention.py:128-174 in fp32',
               'torch': torch.__version__, 'tensors': sorted(tensors)}, open(os.path.join(OUT_DIR, 'wan_block_fp32.json'), 'w'), indent=1)
    print(f'wrote {len(tensors)} tensors, loss = {loss.item():.6f}')


if __name__ == '__main__':
    main()
