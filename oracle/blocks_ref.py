"""ORACLE (test infrastructure, never imported by the product): plain-PyTorch fp32 restatement of the reference's in-tree
Wan DiT block arithmetic, one function per reference routine, each citing the lines it follows.

PINNED: tests/test_golden_cpu.py checks every function here against vectors minted from the reference's own
models/wan/model.py (oracle/make_golden.py -> tests/golden/wan_block_fp32.safetensors).
"""
import math

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
