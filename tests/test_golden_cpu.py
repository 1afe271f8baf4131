"""CPU: the oracle's Wan-block restatement (oracle/blocks_ref.py) against golden vectors minted from the reference's own
models/wan/model.py (oracle/make_golden.py; the reference tree is not needed to run this test)."""
import os

import torch
from safetensors.torch import load_file

from oracle import blocks_ref as br

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'wan_block_fp32.safetensors')
CASE = dict(dim=128, ffn_dim=256, num_heads=2, grid=(2, 6, 8), eps=1e-6)


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def test_oracle_wan_block_matches_reference_vectors():
    g = load_file(GOLD)
    p = {k[len('block.'):]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith('block.')}
    ph = {k[len('head.'):]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith('head.')}
    x, e, ctx, eh = (g[k].clone().requires_grad_(True) for k in ('in.x', 'in.e', 'in.context', 'in.e_head'))
    cos, sin = br.rope_tables(g['in.freqs_re'], g['in.freqs_im'], CASE['grid'])
    y = br.wan_block(p, x, e, ctx, CASE['num_heads'], cos, sin, CASE['eps'])
    out = br.wan_head(ph, y, eh, CASE['eps'])
    loss = (y * g['in.wy']).sum() + (out * g['in.wh']).sum()
    loss.backward()
    assert _rel(y, g['out.y']) < 1e-5 and _rel(out, g['out.head']) < 1e-5
    assert abs(loss.item() - g['out.loss'].item()) / abs(g['out.loss'].item()) < 1e-5
    for name, t in (('x', x), ('e', e), ('context', ctx), ('e_head', eh)):
        assert _rel(t.grad, g[f'grad.{name}']) < 1e-4, name
    for k, v in p.items():
        assert _rel(v.grad, g[f'grad.block.{k}']) < 1e-4, k
    for k, v in ph.items():
        assert _rel(v.grad, g[f'grad.head.{k}']) < 1e-4, k


def test_oracle_sinusoidal_embedding_matches_reference_vectors():
    g = load_file(GOLD)
    got = br.sinusoidal_embedding_1d(256, torch.tensor([17.0, 500.0, 999.0]))
    assert torch.allclose(got, g['out.sinusoidal_256'], atol=1e-6)


WAN_MODEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'wan_model_fp32.safetensors')


def test_oracle_wan_forward_matches_reference_whole_model_step():
    """oracle/blocks_ref.wan_forward (+ the default loss) against the reference's own WanModel driven through the reference's own
    pipeline layers, prepare_inputs and loss (oracle/make_golden_wan_model.py): output, loss and every parameter gradient."""
    import json
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import wan
    from oracle import blocks_ref as br, eager_step
    g = load_file(WAN_MODEL)
    meta = json.load(open(WAN_MODEL.replace('.safetensors', '.json')))
    cfg = wan.tiny_wan_config()
    assert (cfg.dim, cfg.ffn_dim, cfg.text_dim, cfg.num_heads, cfg.num_layers, cfg.text_len) == tuple(meta['config'][k] for k in
                                                                                                    ('dim', 'ffn_dim', 'text_dim', 'num_heads', 'num_layers', 'text_len'))
    p = {k[len('param.'):]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith('param.')}
    out = br.wan_forward(p, cfg, g['prep.x_t'], g['prep.t'], g['in.text_embeddings'], g['in.seq_lens'])
    assert torch.allclose(out, g['out'], rtol=1e-4, atol=1e-5)
    loss = eager_step.default_loss_fn()(out, (g['prep.target'], torch.tensor([])))
    assert abs(loss.item() - g['loss'].item()) / g['loss'].item() < 1e-6
    loss.backward()
    for k, v in p.items():
        want = g[f'grad.{k}']
        assert v.grad is not None, k
        assert (v.grad - want).abs().max() <= 1e-4 * want.abs().max() + 1e-8, k
    # the product's prepare_inputs draws the same x_t / t / target from the same RNG state
    work = wan.WanWorkload(cfg, dtype=torch.float32)
    torch.manual_seed(meta['seed_prepare_inputs'])
    feats, (target, mask) = work.prepare_inputs({'latents': g['in.latents'], 'mask': None, 'text_embeddings': g['in.text_embeddings'],
                                                 'seq_lens': g['in.seq_lens']})
    assert torch.equal(feats[0], g['prep.x_t']) and torch.equal(feats[2], g['prep.t']) and torch.equal(target, g['prep.target']) and mask is None
