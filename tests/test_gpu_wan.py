"""GPU: the HIP-kernel Wan DiT block (diffusion_pipe_amd/workloads/wan.py, every op through the C ABI) against golden
vectors minted from the REFERENCE'S OWN models/wan/model.py (oracle/make_golden.py).  Forward outputs, loss and every
gradient (inputs, modulation, all weights).  Tolerances: exact-fp32 kernel mode 1e-3 relative (north_star's bound; fp32
MFMA / VALU kernels, different summation order); bf16 training mode 4e-2 of each tensor's scale vs the fp32 vectors."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'wan_block_fp32.safetensors')
CASE = dict(dim=128, ffn_dim=256, num_heads=2, grid=(2, 6, 8), eps=1e-6)


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
def test_wan_block_matches_reference_vectors(gpu, dtype, tol):
    from diffusion_pipe_amd import ops
    from diffusion_pipe_amd.workloads import wan
    g = load_file(GOLD)
    c = CASE
    block = wan.WanAttentionBlock(c['dim'], c['ffn_dim'], c['num_heads'], cross_attn_norm=True, eps=c['eps'])
    head = wan.Head(c['dim'], 16, (1, 2, 2), c['eps'])
    block.load_state_dict({k[len('block.'):]: v for k, v in g.items() if k.startswith('block.')})
    head.load_state_dict({k[len('head.'):]: v for k, v in g.items() if k.startswith('head.')})
    block.to(gpu, dtype)
    head.to(gpu, dtype)
    freqs = torch.complex(g['in.freqs_re'], g['in.freqs_im'])
    cos, sin = (t.to(gpu) for t in wan.rope_tables(freqs, c['grid']))
    x, ctx = (g[k].to(gpu, dtype).requires_grad_(True) for k in ('in.x', 'in.context'))
    e, eh = (g[k].to(gpu, dtype).requires_grad_(True) for k in ('in.e', 'in.e_head'))
    y = block(x, e, cos, sin, ctx)
    out = head(y, eh)
    loss = (y.float() * g['in.wy'].to(gpu)).sum() + (out.float() * g['in.wh'].to(gpu)).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(y, g['out.y']) < tol and _rel(out, g['out.head']) < tol
    assert abs(loss.item() - g['out.loss'].item()) / abs(g['out.loss'].item()) < tol
    for name, t in (('x', x), ('e', e), ('context', ctx), ('e_head', eh)):
        assert _rel(t.grad, g[f'grad.{name}']) < tol, name
    for k, v in block.named_parameters():
        assert _rel(v.grad, g[f'grad.block.{k}']) < tol, k
    for k, v in head.named_parameters():
        assert _rel(v.grad, g[f'grad.head.{k}']) < tol, k
    # timestep features (K7): arguments reach ~1e3 rad, where one fp32 ulp of the phase is 6e-5
    t = ops.sinusoidal_embedding(torch.tensor([17.0, 500.0, 999.0], device=gpu), 256)
    assert torch.allclose(t.cpu(), g['out.sinusoidal_256'], atol=5e-4)
