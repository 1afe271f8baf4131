"""ORACLE fixture generator (test infrastructure): block arithmetic at the REAL widths of BASELINE configs 3 and 4.

  * Wan2.1-14B DiT block (dim 5120, ffn 13824, 40 heads of 128, cross-attention to 512 text tokens) on a (9, 32, 16) token grid = 4 608 tokens: the
    REFERENCE'S OWN `WanAttentionBlock` (models/wan/model.py:277-312, imported unmodified as in oracle/make_golden.py, attention =
    models/wan/attention.py:128-174 in fp32) -- forward, a weighted-sum loss, gradients of inputs and of every parameter.
  * Flux.1-dev width (3072 = 24 heads of 128, rotary axes (16, 56, 56), 4096-wide text states): one double-stream + one single-stream block inside the
    reference's pipeline layers, 4 096 image + 512 text tokens, through oracle/flux_ref.py (diffusers restated: parity unpinned) -- output, loss, gradients.

  * HunyuanVideo width (BASELINE config 5: 3072 = 24 x 128, rope (16, 56, 56), token refiner over 4096-wide LLM states): one double + one single stream
    block between the real embedders / refiner / final layer, 2 880 video + 256 text tokens (66 padded), through oracle/hv_ref.py (parity unpinned).

Nothing large is stored: weights and inputs are rebuilt from seeds by `wan_case()` / `flux_case()` (shared with the GPU test), the JSON holds the loss and
(sum |t|, sum t, seeded projection <t, r>, ||t||_2) checksums (oracle/checksums.py) of the outputs and of every gradient.  ~15 GB of host memory, a few minutes of CPU.

    python oracle/make_golden_realdims.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, '..', 'tests', 'golden', 'realdims.json')

WAN = dict(dim=5120, ffn_dim=13824, num_heads=40, grid=(9, 32, 16), ctx_len=512, eps=1e-6, seed=77)


def checksum(t, name):
    from oracle.checksums import checksum4
    return checksum4(t, name)           # [sum |t|, sum t, <t, r(name)>, ||t||_2]


def seeded_state(shapes, seed):
    """{name: tensor} with fan-in scaled weights, small biases, norm scales around one -- the same stream for generator and test."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in shapes.items():
        if name == 'modulation':
            out[name] = torch.randn(shape, generator=g) * 0.3
        elif name.endswith('.weight') and len(shape) >= 2:
            out[name] = torch.randn(shape, generator=g) / (shape[-1] ** 0.5)
        elif name.endswith('.weight'):
            out[name] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=g)
    return out


def wan_case():
    """-> (block state dict, inputs dict) of the Wan-14B-width block case; shapes from the product module (same parameter names as the reference's)."""
    from diffusion_pipe_amd.workloads import wan
    c = WAN
    with torch.device('meta'):
        block = wan.WanAttentionBlock(c['dim'], c['ffn_dim'], c['num_heads'], cross_attn_norm=True, eps=c['eps'])
    state = seeded_state({n: tuple(p.shape) for n, p in block.named_parameters()}, c['seed'])
    g = torch.Generator().manual_seed(c['seed'] + 1)
    f, h, w = c['grid']
    S = f * h * w
    inputs = {'x': torch.randn(1, S, c['dim'], generator=g), 'e': torch.randn(1, 1, 6, c['dim'], generator=g) * 0.5,
              'context': torch.randn(1, c['ctx_len'], c['dim'], generator=g), 'wy': torch.randn(1, S, c['dim'], generator=g) / S}
    return state, inputs


def flux_case():
    """-> (FluxConfig, product workload on CPU fp32, prepared features, target): Flux.1-dev width, 1 + 1 blocks, 4 096 image + 512 text tokens."""
    from diffusion_pipe_amd.workloads import flux
    cfg = flux.FluxConfig(num_layers=1, num_single_layers=1)
    work = flux.FluxWorkload(cfg, model_config={'guidance': 3.5}, dtype=torch.float32, seed=31)
    torch.manual_seed(32)
    feats, (target, _) = work.prepare_inputs(flux.synthetic_flux_batch(cfg, batch_size=1, latent_hw=(128, 128), text_tokens=512, seed=33))
    return cfg, work, feats, target


HV = dict(latent_thw=(5, 48, 48), text_tokens=256, valid_text=(190,), seed=41)


def hv_case():
    """-> (HunyuanVideoConfig, oracle transformer, product workload on CPU fp32 with the same weights, prepared features, label): HunyuanVideo's real width
    (3072 = 24 heads of 128, rope (16, 56, 56), 4096-wide LLM states, 768-wide pooled CLIP vector), 1 double + 1 single stream block, 2 880 video tokens +
    256 text tokens of which 66 are padding."""
    from diffusion_pipe_amd.workloads import hunyuan_video as hv
    from oracle import hv_ref
    cfg = hv.HunyuanVideoConfig(mm_double_blocks_depth=1, mm_single_blocks_depth=1)
    tr = hv_ref.HYVideoDiffusionTransformer(cfg, seed=HV['seed'])
    work = hv.HunyuanVideoWorkload(cfg, model_config={}, dtype=torch.float32, seed=HV['seed'] + 1)
    work.transformer.load_state_dict(tr.state_dict())
    torch.manual_seed(HV['seed'] + 2)
    feats, label = work.prepare_inputs(hv.synthetic_hv_batch(cfg, batch_size=1, latent_thw=HV['latent_thw'], text_tokens=HV['text_tokens'], valid_text=HV['valid_text'],
                                                             seed=HV['seed'] + 3))
    return cfg, tr, work, feats, label


def hv_reference_forward(tr, cfg, f):
    """the reference's layer sequence (models/hunyuan_video.py:483-680) over the oracle transformer, as tests/test_gpu_hv.py spells it"""
    from oracle import hv_ref
    x_t, t, pe1, m1, pe2, fc, fs, gd = f
    vec = tr.time_in(t) + tr.vector_in(pe2) + tr.guidance_in(gd)
    img, txt = tr.img_in(x_t), tr.txt_in(pe1, t, m1)
    cu = hv_ref.get_cu_seqlens(m1, img.shape[1])
    mx = img.shape[1] + txt.shape[1]
    for b in tr.double_blocks:
        img, txt = b(img, txt, vec, cu, cu, mx, mx, (fc[0], fs[0]))
    x = torch.cat([img, txt], 1)
    for b in tr.single_blocks:
        x = b(x, vec, txt.shape[1], cu, cu, mx, mx, (fc[0], fs[0]))
    out = tr.final_layer(x[:, :img.shape[1]], vec)
    _, _, T, H, W = x_t.shape
    return tr.unpatchify(out, T // cfg.patch_size[0], H // cfg.patch_size[1], W // cfg.patch_size[2])


def hv_section():
    cfg, tr, work, feats, label = hv_case()
    out = hv_reference_forward(tr, cfg, feats)
    loss = ((out - label[0]) ** 2).mean()
    loss.backward()
    print('hunyuan-video blocks: loss', float(loss), flush=True)
    return {'case': HV, 'source': 'oracle/hv_ref.py + oracle/blocks_ref.py (hyvideo transformer restated; block dataflow pinned by models/hunyuan_image_modeling.py, wrappers by '
                                  'models/hunyuan_video.py:413-492)', 'loss': float(loss), 'out': checksum(out, 'out'),
            'param_grads': {n: checksum(p.grad, n) for n, p in tr.named_parameters() if p.grad is not None},
            'state_checksum': float(sum(v.double().abs().sum() for v in tr.state_dict().values()))}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'hv':         # add / refresh the HunyuanVideo section only (the Wan / Flux sections take ~15 GB and minutes)
        gold = json.load(open(OUT))
        gold['hv_blocks'] = hv_section()
        with open(OUT, 'w') as fh:
            json.dump(gold, fh)
        return
    gold = {'torch': torch.__version__}
    # ---- Wan-14B-width block: the reference's own class
    from oracle.make_golden import import_reference_wan
    m = import_reference_wan()
    c = WAN
    state, inp = wan_case()
    block = m.WanAttentionBlock('default', c['dim'], c['ffn_dim'], c['num_heads'], (-1, -1), True, True, c['eps']).float()
    block.load_state_dict(state)
    d = c['dim'] // c['num_heads']
    freqs = torch.cat([m.rope_params(1024, d - 4 * (d // 6)), m.rope_params(1024, 2 * (d // 6)), m.rope_params(1024, 2 * (d // 6))], dim=1)
    x, e, ctx = (inp[k].clone().requires_grad_(True) for k in ('x', 'e', 'context'))
    f, h, w = c['grid']
    y = block(x, e, torch.tensor([f * h * w]), torch.tensor([[f, h, w]]), freqs, ctx, None)
    loss = (y * inp['wy']).sum()
    loss.backward()
    gold['wan14b_block'] = {'case': c, 'source': 'models/wan/model.py:277-312 WanAttentionBlock (imported), attention = models/wan/attention.py:128-174 in fp32',
                            'loss': float(loss), 'y': checksum(y, 'y'), 'grad_x': checksum(x.grad, 'grad_x'), 'grad_e': checksum(e.grad, 'grad_e'), 'grad_context': checksum(ctx.grad, 'grad_context'),
                            'param_grads': {n: checksum(p.grad, n) for n, p in block.named_parameters()},
                            'state_checksum': float(sum(v.double().abs().sum() for v in state.values()))}
    print('wan14b block: loss', float(loss), flush=True)
    del block, x, e, ctx, y
    # ---- Flux-width double + single block through the oracle's restatement
    from oracle import flux_ref
    cfg, work, feats, target = flux_case()
    ref = flux_ref.FluxRef(cfg, seed=1)
    ref.transformer.load_state_dict(work.transformer.state_dict())
    xx = tuple(t.clone() for t in feats)
    for layer in ref.to_layers():
        xx = layer(xx)
    loss = ((xx - target) ** 2).mean()
    loss.backward()
    gold['flux_blocks'] = {'source': 'oracle/flux_ref.py (diffusers FluxTransformer2DModel restated; wrappers pinned by models/flux.py:456-548)', 'loss': float(loss), 'out': checksum(xx, 'out'),
                           'param_grads': {n: checksum(p.grad, n) for n, p in ref.transformer.named_parameters() if p.grad is not None},
                           'state_checksum': float(sum(v.double().abs().sum() for v in work.transformer.state_dict().values()))}
    print('flux blocks: loss', float(loss), flush=True)
    del ref, work, xx
    gold['hv_blocks'] = hv_section()
    with open(OUT, 'w') as fh:
        json.dump(gold, fh)


if __name__ == '__main__':
    main()
