"""GPU: `diffusion_pipe_amd.adopt` on a module tree with the reference's Wan class / attribute / parameter names (tests/wan_standin.py: forwards = the pinned
oracle functions in fp32 ATen arithmetic).  The adopted tree -- same Parameter objects, HIP kernels behind the reference's call signature -- must reproduce the
stand-in's outputs and gradients: exact-fp32 kernel mode within north_star's 1e-3, bf16 within 4e-2 of each tensor's scale (tests/test_gpu_wan.py's bounds)."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def _case(gpu, dtype):
    from tests import wan_standin as ws
    torch.manual_seed(3)
    dim, ffn, heads, grid, L = 256, 512, 2, (2, 6, 8), 24
    blocks = [ws.WanAttentionBlock('default', dim, ffn, heads, cross_attn_norm=True), ws.WanAttentionBlock('default', dim, ffn, heads, cross_attn_norm=False)]
    head = ws.Head(dim, 16, (1, 2, 2))
    tree = nn.ModuleList([nn.ModuleList([ws.TransformerLayer(b) for b in blocks]), nn.ModuleDict({'head': head})]).to(gpu)
    S = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(5)
    ins = {'x': torch.randn(1, S, dim, generator=g), 'e0': torch.randn(1, 1, 6, dim, generator=g) * 0.3, 'eh': torch.randn(1, 1, dim, generator=g) * 0.3,
           'ctx': torch.randn(1, L, dim, generator=g), 'wy': torch.randn(1, S, 64, generator=g)}
    freqs = ws.rope_freqs(dim // heads).to(gpu)
    grid_sizes = torch.tensor([list(grid)], dtype=torch.long, device=gpu)
    seq_lens = torch.tensor([S], dtype=torch.long, device=gpu)
    return tree, ins, freqs, grid_sizes, seq_lens


def _run(tree, ins, freqs, grid_sizes, seq_lens, gpu, dtype):
    x, e0, eh, ctx = (ins[k].to(gpu, dtype).requires_grad_(True) for k in ('x', 'e0', 'eh', 'ctx'))
    t = (x, eh, e0, seq_lens, grid_sizes, freqs, ctx)
    for layer in tree[0]:
        t = layer(t)
    out = tree[1]['head'](t[0], eh)
    loss = (out.float() * ins['wy'].to(gpu)).sum()
    loss.backward()
    torch.cuda.synchronize()
    grads = {'x': x.grad, 'e0': e0.grad, 'eh': eh.grad, 'ctx': ctx.grad, **{n: p.grad for n, p in tree.named_parameters()}}
    return out.detach(), float(loss), grads


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
def test_adopted_wan_tree_matches_the_standin(gpu, dtype, tol):
    from diffusion_pipe_amd import adopt as adopt_mod
    from diffusion_pipe_amd import nn as dnn
    tree, ins, freqs, grid_sizes, seq_lens = _case(gpu, dtype)
    want_out, want_loss, want_g = _run(tree, ins, freqs, grid_sizes, seq_lens, gpu, torch.float32)            # the stand-in itself, fp32 ATen
    mine = copy.deepcopy(tree).to(dtype)
    for p in mine.parameters():
        p.grad = None
    names = [n for n, _ in mine.named_parameters()]
    ptrs = {n: p.data_ptr() for n, p in mine.named_parameters()}
    report = adopt_mod.adopt(mine)
    assert sum(1 for v in report.values() if v[1] == 'AdoptedWanAttentionBlock') == 2 and ('Head', 'AdoptedWanHead') in report.values()
    assert [n for n, _ in mine.named_parameters()] == names and all(p.data_ptr() == ptrs[n] for n, p in mine.named_parameters())
    assert not [m for m in mine.modules() if type(m) is nn.Linear]
    assert isinstance(mine[0][0].block.self_attn.q, dnn.Linear)
    out, loss, g = _run(mine, ins, freqs, grid_sizes, seq_lens, gpu, dtype)
    assert _rel(out, want_out) < tol
    assert abs(loss - want_loss) / abs(want_loss) < tol
    for k in want_g:
        assert g[k] is not None, k
        assert _rel(g[k], want_g[k]) < tol, k
    # the rotary tables were built once per grid and are served from the cache afterwards (what a hipGraph capture relies on)
    blk = mine[0][0].block
    assert len(blk._rope_cache) == 1
    _run(mine, ins, freqs, grid_sizes, seq_lens, gpu, dtype)
    assert len(blk._rope_cache) == 1
