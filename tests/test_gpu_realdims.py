"""GPU: block arithmetic at the REAL widths of BASELINE configs 3 and 4 (head_dim 128) against tests/golden/realdims.json
(oracle/make_golden_realdims.py): the Wan2.1-14B DiT block (dim 5120, ffn 13824, 40 heads, 4 608 tokens, 512 text tokens) vs the reference's OWN
`WanAttentionBlock` run on CPU; one Flux double + one single stream block at 3072 = 24 x 128 (4 096 image + 512 text tokens) vs the oracle's restatement.
Weights and inputs are rebuilt from the generator's seeds; compared: loss, and (sum |t|) checksums of the output and of every gradient.

Tolerances: exact-fp32 kernel mode 1e-3 on the loss (north_star's bound) and 5e-3 on each checksum; bf16 mode (the flash-attention / LDS-DMA GEMM path that
trains) 3e-2 on the loss and 8e-2 on each checksum (+ an absolute floor of 1e-6 of the largest checksum for gradients that are analytically ~0)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'realdims.json')))


def _check(got, want, tol, floor, what):
    err = abs(float(got.detach().double().abs().sum()) - want[0])
    assert err <= tol * want[0] + floor, (what, float(got.detach().double().abs().sum()), want[0])


@pytest.mark.parametrize('dtype,ltol,ctol', [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 3e-2, 8e-2)])
def test_wan14b_width_block_matches_the_reference_block(gpu, dtype, ltol, ctol):
    from diffusion_pipe_amd.workloads import wan
    from oracle.make_golden_realdims import WAN, wan_case
    ref = G['wan14b_block']
    c = WAN
    state, inp = wan_case()
    assert abs(float(sum(v.double().abs().sum() for v in state.values())) - ref['state_checksum']) <= 1e-9 * ref['state_checksum']
    block = wan.WanAttentionBlock(c['dim'], c['ffn_dim'], c['num_heads'], cross_attn_norm=True, eps=c['eps'])
    block.load_state_dict(state)
    block.to(gpu, dtype)
    d = c['dim'] // c['num_heads']
    freqs = torch.cat([wan.rope_params(1024, d - 4 * (d // 6)), wan.rope_params(1024, 2 * (d // 6)), wan.rope_params(1024, 2 * (d // 6))], dim=1)
    cos, sin = (t.to(gpu) for t in wan.rope_tables(freqs, c['grid']))
    x, ctx = (inp[k].to(gpu, dtype).requires_grad_(True) for k in ('x', 'context'))
    e = inp['e'].to(gpu, dtype).requires_grad_(True)
    y = block(x, e, cos, sin, ctx)
    loss = (y.float() * inp['wy'].to(gpu)).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref['loss']) <= ltol * abs(ref['loss']), (loss.item(), ref['loss'])
    floor = 1e-6 * max(v[0] for v in ref['param_grads'].values())
    _check(y, ref['y'], ctol, 0.0, 'y')
    _check(x.grad, ref['grad_x'], ctol, floor, 'grad x')
    _check(e.grad, ref['grad_e'], ctol, floor, 'grad e')
    _check(ctx.grad, ref['grad_context'], ctol, floor, 'grad context')
    for n, p in block.named_parameters():
        _check(p.grad, ref['param_grads'][n], ctol, floor, n)


@pytest.mark.parametrize('dtype,ltol,ctol', [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 3e-2, 8e-2)])
def test_flux_width_double_and_single_block_match_the_oracle(gpu, dtype, ltol, ctol):
    from oracle.make_golden_realdims import flux_case
    ref = G['flux_blocks']
    cfg, work, feats, target = flux_case()
    assert abs(float(sum(v.double().abs().sum() for v in work.transformer.state_dict().values())) - ref['state_checksum']) <= 1e-9 * ref['state_checksum']
    work.transformer.to(gpu, dtype)
    x = tuple(t.to(gpu) for t in feats)
    for layer in work.to_layers():
        x = layer(x)
    loss = work.get_loss_fn()(x, (target.to(gpu), torch.tensor([], device=gpu)))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref['loss']) <= ltol * ref['loss'], (loss.item(), ref['loss'])
    _check(x, ref['out'], ctol, 0.0, 'output')
    floor = 1e-6 * max(v[0] for v in ref['param_grads'].values())
    for n, p in work.transformer.named_parameters():
        if n in ref['param_grads']:
            _check(p.grad, ref['param_grads'][n], ctol, floor, n)
