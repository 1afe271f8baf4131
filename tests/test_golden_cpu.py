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


def test_oracle_sdxl_layers_match_the_reference_wrapper_classes():
    """oracle/sdxl_ref.py's 23 pipeline layers against the reference's OWN wrapper classes (models/sdxl.py:654-995, lifted and run over
    the same restated blocks by oracle/make_golden_sdxl_layers.py): every stage-boundary tuple layout (skip stack push / pop,
    forward_upsample_size), the conditioning (timestep + text-time embedding, chunked prompt encoding) and the final output."""
    import json
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import sdxl
    from oracle import sdxl_ref
    from oracle.make_golden_sdxl_layers import weight_checksum
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sdxl_layers')
    meta, g = json.load(open(base + '.json')), load_file(base + '.safetensors')
    cfg = sdxl.tiny_config()
    ref = sdxl_ref.SDXLRef(cfg, seed=meta['seed'])
    assert abs(weight_checksum(ref) - meta['weight_checksum']) <= 1e-9 * meta['weight_checksum']          # same seeded weights as the generator
    layers = ref.to_layers()
    assert [type(l).__name__ for l in layers] == meta['layer_names']
    assert [type(l).__name__ for l in sdxl.SDXLWorkload(cfg, dtype=torch.float32).to_layers()] == meta['layer_names']    # product: same 23 layers
    for tag, rec in meta['cases'].items():
        x = (g[f'{tag}.latents'].clone(), g[f'{tag}.timesteps'], g[f'{tag}.ids1'], g[f'{tag}.ids2'], g[f'{tag}.add_time_ids'])
        for i, layer in enumerate(layers):
            x = layer(x)
            got = [list(t.shape) for t in x] if isinstance(x, tuple) else list(x.shape)
            assert got == rec['layouts'][i], (tag, i, type(layer).__name__)
            if i == 0:
                assert torch.allclose(x[2], g[f'{tag}.emb'], rtol=1e-5, atol=1e-6) and torch.allclose(x[3], g[f'{tag}.encoder_hidden_states'], rtol=1e-5, atol=1e-6)
                assert bool(x[-1]) == rec['forward_upsample_size']
        out, ts = x
        assert torch.allclose(out, g[f'{tag}.out'], rtol=1e-4, atol=1e-5) and torch.equal(ts, g[f'{tag}.out_ts'])


def test_oracle_flux_layers_match_the_reference_wrapper_classes():
    """oracle/flux_ref.py's pipeline layers against the reference's OWN Flux wrapper classes and prepare_inputs (models/flux.py:323-404,
    456-548, lifted and run over the same restated blocks by oracle/make_golden_flux_layers.py); the product's prepare_inputs / layer list too."""
    import json
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import flux
    from oracle import flux_ref
    from oracle.make_golden_flux_layers import weight_checksum
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'flux_layers')
    meta, g = json.load(open(base + '.json')), load_file(base + '.safetensors')
    cfg = flux.tiny_flux_config()
    ref = flux_ref.FluxRef(cfg, seed=meta['seed'])
    assert abs(weight_checksum(ref) - meta['weight_checksum']) <= 1e-9 * meta['weight_checksum']
    work = flux.FluxWorkload(cfg, model_config={'guidance': 3.5}, dtype=torch.float32)
    want_names = meta['layer_names']
    assert [type(l).__name__ for l in work.to_layers()] == want_names and len(ref.to_layers()) == len(want_names)
    torch.manual_seed(meta['seed_prepare_inputs'])
    feats, (target, mask) = work.prepare_inputs({'latents': g['latents'], 'mask': None, 't5_embed': g['t5_embed'], 'clip_embed': g['clip_embed']})
    for i, f in enumerate(feats):
        assert torch.allclose(f.float(), g[f'feature.{i}'].float(), rtol=1e-6, atol=1e-6), i
    assert torch.allclose(target, g['target'], rtol=1e-6, atol=1e-6) and mask is None
    x = tuple(g[f'feature.{i}'].clone() for i in range(8))
    for i, layer in enumerate(ref.to_layers()):
        x = layer(x)
        got = [list(v.shape) for v in x] if isinstance(x, tuple) else list(x.shape)
        assert got == meta['layouts'][i], (i, type(layer).__name__)
        if i == 0:
            assert torch.allclose(x[2], g['temb'], rtol=1e-5, atol=1e-6) and torch.equal(x[3], g['freqs_cos']) and torch.equal(x[4], g['freqs_sin'])
    assert torch.allclose(x, g['out'], rtol=1e-4, atol=1e-5)
    assert abs(((x - g['target']) ** 2).mean().item() - meta['loss']) / meta['loss'] < 1e-5


def test_oracle_mmdit_blocks_match_the_in_tree_reference_blocks():
    """oracle/blocks_ref.mm_double_block / mm_single_block against the reference's in-tree models/hunyuan_image_modeling.py blocks
    (imported unmodified, hyimage leaf helpers stubbed: oracle/make_golden_mmdit.py): outputs on the valid rows, loss, input and
    parameter gradients, with 5 of 12 text tokens of one sample padded."""
    import json
    from safetensors.torch import load_file
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mmdit_blocks_fp32')
    meta, g = json.load(open(base + '.json')), load_file(base + '.safetensors')
    heads, Si, St = meta['heads'], meta['img_tokens'], meta['txt_tokens']
    text_len = g['in.text_len']
    valid = (torch.arange(St)[None, :] < text_len[:, None])[:, :, None].float()
    cos, sin = g['in.cos_half'], g['in.sin_half']

    def close(a, b, tol=1e-4):
        return (a - b).abs().max() <= tol * b.abs().max() + 1e-7

    # double-stream block
    p = {k[len('double.param.'):]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith('double.param.')}
    img, txt, vec = (g[f'in.{n}'].clone().requires_grad_(True) for n in ('img', 'txt', 'vec'))
    o_img, o_txt = br.mm_double_block(p, img, txt, vec, heads, cos, sin, text_len)
    assert close(o_img, g['double.out_img']) and close(o_txt * valid, g['double.out_txt'] * valid)
    loss = (o_img * g['in.w_img']).sum() + (o_txt * g['in.w_txt'] * valid).sum()
    assert abs(loss.item() - meta['double_loss']) / abs(meta['double_loss']) < 1e-5
    loss.backward()
    assert close(img.grad, g['double.grad.img']) and close(vec.grad, g['double.grad.vec'])
    assert close(txt.grad * valid, g['double.grad.txt'] * valid)                       # padded text rows: don't-care (masked out of the loss)
    for k, v in p.items():
        assert close(v.grad, g[f'double.pgrad.{k}'], 2e-4), k
    # single-stream block
    p = {k[len('single.param.'):]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith('single.param.')}
    x = torch.cat([g['in.img'], g['in.txt']], dim=1).requires_grad_(True)
    vec = g['in.vec'].clone().requires_grad_(True)
    rows = torch.cat([torch.ones(2, Si, 1), valid], dim=1)
    out = br.mm_single_block(p, x, vec, St, heads, cos, sin, text_len)
    assert close(out * rows, g['single.out'] * rows)
    loss = (out * torch.cat([g['in.w_img'], g['in.w_txt'] * valid], dim=1)).sum()
    assert abs(loss.item() - meta['single_loss']) / abs(meta['single_loss']) < 1e-5
    loss.backward()
    assert close(x.grad * rows, g['single.grad.x'] * rows) and close(vec.grad, g['single.grad.vec'])
    for k, v in p.items():
        assert close(v.grad, g[f'single.pgrad.{k}'], 2e-4), k


def test_oracle_clip_text_encoders_match_hf_transformers():
    """oracle/sdxl_ref.CLIPTextModel (both SDXL encoder geometries) against the real HF transformers CLIPTextModel /
    CLIPTextModelWithProjection (oracle/make_golden_clip.py): penultimate hidden state, first output (projected pooled embedding for
    encoder 2; for encoder 1 the reference never reads it), loss and every parameter gradient."""
    import json
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import sdxl
    from oracle import sdxl_ref
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'clip_encoders_fp32')
    meta, g = json.load(open(base + '.json')), load_file(base + '.safetensors')
    cfg = sdxl.tiny_config()
    for tag, c in (('te1', cfg.te1), ('te2', cfg.te2)):
        model = sdxl_ref.CLIPTextModel(c)
        sd = {k[len(f'{tag}.param.'):]: v for k, v in g.items() if k.startswith(f'{tag}.param.')}
        if c.proj_dim is None:
            sd = {k: v for k, v in sd.items() if not k.startswith('text_projection.')}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (tag, missing, unexpected)
        hidden_states, pooled = model(g[f'{tag}.ids'])
        assert len(hidden_states) == meta['encoders'][tag]['hidden_states']
        penult = hidden_states[-2]
        assert torch.allclose(penult, g[f'{tag}.penultimate'], rtol=1e-4, atol=1e-5)
        loss = (penult * g[f'{tag}.w1']).sum()
        if tag == 'te2':
            assert torch.allclose(pooled, g[f'{tag}.first'], rtol=1e-4, atol=1e-5)
            loss = loss + (pooled * g[f'{tag}.w2']).sum()
        assert abs(loss.item() - g[f'{tag}.loss'].item()) <= 1e-5 * abs(g[f'{tag}.loss'].item())
        loss.backward()
        scale = max(g[f'{tag}.grad.{k}'].abs().max().item() for k, _ in model.named_parameters())
        for k, p in model.named_parameters():
            want = g[f'{tag}.grad.{k}']
            got = p.grad if p.grad is not None else torch.zeros_like(p)          # layers past the penultimate state get no gradient in encoder 1
            # (analytically-zero gradients such as k_proj.bias are pure rounding noise: absolute floor at 1e-5 of the largest gradient)
            assert (got - want).abs().max() <= 2e-4 * want.abs().max() + 1e-5 * scale, (tag, k)


def test_hunyuan_video_host_logic_matches_the_reference_pipeline_code():
    """prepare_inputs (models/hunyuan_video.py:413-481), get_rotary_pos_embed (:35-81), the layer list of to_layers (:483-492, with the bare
    concatenate_hidden_states callable) and the cu_seqlens hand-off, against the reference's own code lifted over the oracle transformer
    (oracle/make_golden_hv_layers.py).  Bit-identical: it is all host-side index / RNG logic."""
    import json
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import hunyuan_video as hv
    from oracle import hv_ref
    from oracle.make_golden_hv_layers import weight_checksum
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hv_layers')
    meta, g = json.load(open(base + '.json')), load_file(base + '.safetensors')
    cfg = hv.tiny_hv_config()
    tr = hv_ref.HYVideoDiffusionTransformer(cfg, seed=meta['seed'])
    assert abs(weight_checksum(tr) - meta['weight_checksum']) <= 1e-9 * meta['weight_checksum']
    work = hv.HunyuanVideoWorkload(cfg, model_config=meta['model_config'], dtype=torch.float32)
    layers = work.to_layers()
    assert [getattr(l, '__name__', type(l).__name__) for l in layers] == meta['layer_names']
    assert layers[3] is hv.concatenate_hidden_states and not isinstance(layers[3], torch.nn.Module)
    b = {k[len('batch.'):]: v for k, v in g.items() if k.startswith('batch.')}
    torch.manual_seed(meta['seed_prepare_inputs'])
    feats, (target, mask) = work.prepare_inputs(b)
    for i, f in enumerate(feats):
        assert f.dtype == g[f'feature.{i}'].dtype and torch.equal(f, g[f'feature.{i}']), i
    assert torch.equal(target, g['target']) and torch.equal(mask, g['label_mask'])
    torch.manual_seed(meta['seed_quantile'])
    fq, (_, mq) = work.prepare_inputs(dict(b, mask=None), timestep_quantile=0.3)
    assert torch.equal(fq[1], g['quantile.t']) and torch.equal(fq[0], g['quantile.x_t']) and mq is None
    # the InitialLayer hand-off pieces that are pure host logic
    assert torch.equal(hv.get_cu_seqlens(b['prompt_attention_mask_1'], g['initial.img'].shape[1]), g['initial.cu_seqlens'])
    cos, sin = hv.get_rotary_pos_embed(cfg, (3 - 1) * 4 + 1, 8 * 8, 12 * 8)
    assert torch.equal(cos, g['initial.freqs_cos']) and torch.equal(sin, g['initial.freqs_sin'])
    assert hv._text_len(g['initial.cu_seqlens'], g['initial.img'].shape[1], int(g['initial.max_seqlen'])).tolist() == [12, 7]
    # the product's transformer carries the reference package's parameter names (checkpoints map 1:1)
    assert set(dict(work.transformer.named_parameters())) == set(dict(tr.named_parameters()))
