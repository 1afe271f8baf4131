"""ORACLE (test infrastructure, never imported by the product): plain-PyTorch fp32 restatement of the reference's in-tree
Wan DiT block arithmetic, one function per reference routine, each citing the lines it follows.

PINNED: tests/test_golden_cpu.py checks every function here against vectors minted from the reference's own
models/wan/model.py (oracle/make_golden.py -> tests/golden/wan_block_fp32.safetensors).
"""

import torch
import torch.nn.functional as F


def sinusoidal_embedding_1d(dim, position):
    """models/wan/model.py:15-25: [cos | sin] of position * 10000^(-i/half)."""
    half = dim // 2
    position = position.to(torch.float32)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, dtype=torch.float32) / half))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_tables(freqs_re, freqs_im, grid):
    """cos / sin [S, d/2] of models/wan/model.py:40-58: the complex table [1024, d/2] is split [c - 2(c//3), c//3, c//3]
    along the frequency axis and indexed by (frame, row, column)."""
    f, h, w = grid
    c = freqs_re.shape[1]
    sizes = [c - 2 * (c // 3), c // 3, c // 3]

    def expand(t):
        a, b, cc = t.split(sizes, dim=1)
        return torch.cat([a[:f].view(f, 1, 1, -1).expand(f, h, w, -1), b[:h].view(1, h, 1, -1).expand(f, h, w, -1),
                          cc[:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, -1)
    return expand(freqs_re), expand(freqs_im)


def rope_apply(x, cos, sin):
    """models/wan/model.py:47-61: complex multiply of consecutive pairs (x[2i] + i x[2i+1]) by (cos + i sin).  x: [B, S, H, D]."""
    xr, xi = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    out = torch.stack([xr * c - xi * s, xr * s + xi * c], dim=-1)
    return out.flatten(3)


def rms_norm(x, weight, eps):
    """models/wan/model.py:70-86."""
    return (x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)) * weight


def layer_norm(x, eps, weight=None, bias=None):
    """models/wan/model.py:89-99 (fp32 LayerNorm, optional affine)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def sdpa(q, k, v):
    """models/wan/attention.py:159-174 (unmasked scaled-dot-product attention), [B, S, H, D] layout."""
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return o.transpose(1, 2)


def self_attention(p, prefix, x, heads, cos, sin, eps):
    """models/wan/model.py:127-156."""
    B, S, C = x.shape
    d = C // heads
    q = rms_norm(F.linear(x, p[f'{prefix}.q.weight'], p[f'{prefix}.q.bias']), p[f'{prefix}.norm_q.weight'], eps).view(B, S, heads, d)
    k = rms_norm(F.linear(x, p[f'{prefix}.k.weight'], p[f'{prefix}.k.bias']), p[f'{prefix}.norm_k.weight'], eps).view(B, S, heads, d)
    v = F.linear(x, p[f'{prefix}.v.weight'], p[f'{prefix}.v.bias']).view(B, S, heads, d)
    o = sdpa(rope_apply(q, cos, sin), rope_apply(k, cos, sin), v).flatten(2)
    return F.linear(o, p[f'{prefix}.o.weight'], p[f'{prefix}.o.bias'])


def cross_attention(p, prefix, x, context, heads, eps):
    """models/wan/model.py:161-181."""
    B, S, C = x.shape
    d = C // heads
    q = rms_norm(F.linear(x, p[f'{prefix}.q.weight'], p[f'{prefix}.q.bias']), p[f'{prefix}.norm_q.weight'], eps).view(B, S, heads, d)
    k = rms_norm(F.linear(context, p[f'{prefix}.k.weight'], p[f'{prefix}.k.bias']), p[f'{prefix}.norm_k.weight'], eps).view(B, -1, heads, d)
    v = F.linear(context, p[f'{prefix}.v.weight'], p[f'{prefix}.v.bias']).view(B, -1, heads, d)
    return F.linear(sdpa(q, k, v).flatten(2), p[f'{prefix}.o.weight'], p[f'{prefix}.o.bias'])


def wan_block(p, x, e, context, heads, cos, sin, eps):
    """models/wan/model.py:277-312.  p: dict of the block's parameters (reference names); e: [B, 1, 6, C]."""
    e = (p['modulation'].unsqueeze(0) + e).chunk(6, dim=2)
    y = self_attention(p, 'self_attn', layer_norm(x, eps) * (1 + e[1].squeeze(2)) + e[0].squeeze(2), heads, cos, sin, eps)
    x = x + y * e[2].squeeze(2)
    n3 = layer_norm(x, eps, p.get('norm3.weight'), p.get('norm3.bias')) if 'norm3.weight' in p else x
    x = x + cross_attention(p, 'cross_attn', n3, context, heads, eps)
    h = layer_norm(x, eps) * (1 + e[4].squeeze(2)) + e[3].squeeze(2)
    y = F.linear(F.gelu(F.linear(h, p['ffn.0.weight'], p['ffn.0.bias']), approximate='tanh'), p['ffn.2.weight'], p['ffn.2.bias'])
    return x + y * e[5].squeeze(2)


def wan_head(p, x, e, eps):
    """models/wan/model.py:332-343.  e: [B, 1, C]."""
    e = (p['modulation'].unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)
    return F.linear(layer_norm(x, eps) * (1 + e[1].squeeze(2)) + e[0].squeeze(2), p['head.weight'], p['head.bias'])


# ---- MMDiT blocks (Flux / HunyuanVideo / HunyuanImage).  PARITY UNPINNED: the dataflow follows the reference's in-tree
# models/hunyuan_image_modeling.py:61-345, its helpers are restated from the un-vendored hyimage package. -----------------
def modulate(x, shift, scale):
    """[3P] hyimage modulate_layers.modulate: x * (1 + scale[:, None]) + shift[:, None]."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def apply_gate(x, gate):
    """[3P] hyimage modulate_layers.apply_gate."""
    return x * gate.unsqueeze(1)


def apply_rotary_interleaved(x, cos, sin):
    """[3P] hyimage posemb_layers.apply_rotary_emb(use_real=True): x * cos + rotate_half(x) * sin on interleaved pairs, with
    the [S, d] tables given here as their distinct halves [S, d/2]."""
    c = cos.repeat_interleave(2, dim=-1)[None, :, None, :]
    s = sin.repeat_interleave(2, dim=-1)[None, :, None, :]
    xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * c + rot * s).to(x.dtype)


def _mm_attention(q, k, v, kv_len):
    """models/hunyuan_image_modeling.py:20-58 (flash_attn_no_pad with the padded text keys masked)."""
    B, S = q.shape[0], k.shape[1]
    mask = None
    if kv_len is not None:
        mask = (torch.arange(S)[None, :] < kv_len[:, None])[:, None, None, :]
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask)
    return o.transpose(1, 2).reshape(B, q.shape[1], -1)


def _stream_qkv(p, s, x, shift, scale, heads, cos, sin):
    h = modulate(layer_norm(x, 1e-6), shift, scale)
    qkv = F.linear(h, p[f'{s}_attn_qkv.weight'], p.get(f'{s}_attn_qkv.bias'))
    q, k, v = qkv.view(qkv.shape[0], qkv.shape[1], 3, heads, -1).unbind(2)
    q, k = rms_norm(q, p[f'{s}_attn_q_norm.weight'], 1e-6), rms_norm(k, p[f'{s}_attn_k_norm.weight'], 1e-6)
    if cos is not None:
        q, k = apply_rotary_interleaved(q, cos, sin), apply_rotary_interleaved(k, cos, sin)
    return q, k, v


def _mlp(p, prefix, x):
    return F.linear(F.gelu(F.linear(x, p[f'{prefix}.fc1.weight'], p[f'{prefix}.fc1.bias']), approximate='tanh'), p[f'{prefix}.fc2.weight'], p[f'{prefix}.fc2.bias'])


def mm_double_block(p, img, txt, vec, heads, cos=None, sin=None, text_len=None):
    """models/hunyuan_image_modeling.py:148-240."""
    i = F.linear(F.silu(vec), p['img_mod.linear.weight'], p['img_mod.linear.bias']).chunk(6, dim=-1)
    t = F.linear(F.silu(vec), p['txt_mod.linear.weight'], p['txt_mod.linear.bias']).chunk(6, dim=-1)
    iq, ik, iv = _stream_qkv(p, 'img', img, i[0], i[1], heads, cos, sin)
    tq, tk, tv = _stream_qkv(p, 'txt', txt, t[0], t[1], heads, None, None)
    Si = img.shape[1]
    kv_len = text_len + Si if text_len is not None else None
    attn = _mm_attention(torch.cat([iq, tq], 1), torch.cat([ik, tk], 1), torch.cat([iv, tv], 1), kv_len)
    img_attn, txt_attn = attn[:, :Si], attn[:, Si:]
    img = img + apply_gate(F.linear(img_attn, p['img_attn_proj.weight'], p.get('img_attn_proj.bias')), i[2])
    img = img + apply_gate(_mlp(p, 'img_mlp', modulate(layer_norm(img, 1e-6), i[3], i[4])), i[5])
    txt = txt + apply_gate(F.linear(txt_attn, p['txt_attn_proj.weight'], p.get('txt_attn_proj.bias')), t[2])
    txt = txt + apply_gate(_mlp(p, 'txt_mlp', modulate(layer_norm(txt, 1e-6), t[3], t[4])), t[5])
    return img, txt


def mm_single_block(p, x, vec, txt_len, heads, cos=None, sin=None, text_len=None):
    """models/hunyuan_image_modeling.py:305-345."""
    hidden = x.shape[-1]
    shift, scale, gate = F.linear(F.silu(vec), p['modulation.linear.weight'], p['modulation.linear.bias']).chunk(3, dim=-1)
    h = F.linear(modulate(layer_norm(x, 1e-6), shift, scale), p['linear1.weight'], p['linear1.bias'])
    qkv, mlp = torch.split(h, [3 * hidden, h.shape[-1] - 3 * hidden], dim=-1)
    q, k, v = qkv.reshape(qkv.shape[0], qkv.shape[1], 3, heads, -1).unbind(2)
    q, k = rms_norm(q, p['q_norm.weight'], 1e-6), rms_norm(k, p['k_norm.weight'], 1e-6)
    Si = x.shape[1] - txt_len
    if cos is not None:
        q = torch.cat([apply_rotary_interleaved(q[:, :Si], cos, sin), q[:, Si:]], 1)
        k = torch.cat([apply_rotary_interleaved(k[:, :Si], cos, sin), k[:, Si:]], 1)
    kv_len = text_len + Si if text_len is not None else None
    attn = _mm_attention(q, k, v, kv_len)
    out = F.linear(torch.cat([attn, F.gelu(mlp, approximate='tanh')], dim=2), p['linear2.weight'], p['linear2.bias'])
    return x + apply_gate(out, gate)


# ---- whole Wan t2v model as the pipeline runs it (models/wan/wan.py:414-546, models/wan/model.py:449-518) -----------------
def wan_forward(p, cfg, x_t, t, text_embeddings, seq_lens):
    """p: dict of WanModel parameters (reference names); equal grids, cached text embeddings.  Returns [B, C, F, H, W]."""
    B = x_t.shape[0]
    x = F.conv3d(x_t, p['patch_embedding.weight'], p['patch_embedding.bias'], stride=cfg.patch_size)
    grid = tuple(x.shape[2:])
    x = x.flatten(2).transpose(1, 2)
    e = sinusoidal_embedding_1d(cfg.freq_dim, t.flatten())
    e = F.linear(F.silu(F.linear(e, p['time_embedding.0.weight'], p['time_embedding.0.bias'])), p['time_embedding.2.weight'], p['time_embedding.2.bias']).unsqueeze(1)
    e0 = F.linear(F.silu(e), p['time_projection.1.weight'], p['time_projection.1.bias']).unflatten(2, (6, cfg.dim))
    ctx = torch.stack([torch.cat([u[:n], u.new_zeros(cfg.text_len - n, u.shape[1])]) for u, n in zip(text_embeddings, seq_lens.tolist())])
    ctx = F.linear(F.gelu(F.linear(ctx, p['text_embedding.0.weight'], p['text_embedding.0.bias']), approximate='tanh'),
                   p['text_embedding.2.weight'], p['text_embedding.2.bias'])
    d = cfg.dim // cfg.num_heads
    theta = lambda n: torch.outer(torch.arange(1024, dtype=torch.float32), 1.0 / torch.pow(10000.0, torch.arange(0, n, 2, dtype=torch.float32) / n))  # noqa: E731
    ang = torch.cat([theta(d - 4 * (d // 6)), theta(2 * (d // 6)), theta(2 * (d // 6))], dim=1)          # models/wan/model.py:29-37,478-483
    cos, sin = rope_tables(torch.cos(ang), torch.sin(ang), grid)
    for i in range(cfg.num_layers):
        bp = {k[len(f'blocks.{i}.'):]: v for k, v in p.items() if k.startswith(f'blocks.{i}.')}
        x = wan_block(bp, x, e0, ctx, cfg.num_heads, cos, sin, cfg.eps)
    hp = {k[len('head.'):]: v for k, v in p.items() if k.startswith('head.')}
    x = wan_head(hp, x, e.squeeze(1).unsqueeze(1), cfg.eps)
    f, h, w = grid
    pt, ph, pw = cfg.patch_size
    u = x.view(B, f, h, w, pt, ph, pw, cfg.out_dim)
    return torch.einsum('bfhwpqrc->bcfphqwr', u).reshape(B, cfg.out_dim, f * pt, h * ph, w * pw)
