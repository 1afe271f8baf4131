"""Test scaffolding for `diffusion_pipe_amd.adopt` where /root/reference is absent (the GPU box): a module tree with the reference's class names, attribute names
and parameter names for the Wan DiT (models/wan/model.py:70-99,102-181,237-343) whose forwards evaluate the pinned oracle functions of oracle/blocks_ref.py
(fp32 ATen arithmetic: the checker).  Not a copy of the reference's modules -- only their names and call signatures, which is what `adopt` keys on."""
import torch
from torch import nn

from oracle import blocks_ref as br


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return br.rms_norm(x.float(), self.weight, self.eps).type_as(x)


class WanLayerNorm(nn.LayerNorm):
    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)

    def forward(self, x):
        return super().forward(x.float()).type_as(x)


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        super().__init__()
        self.dim, self.num_heads, self.head_dim, self.window_size, self.qk_norm, self.eps = dim, num_heads, dim // num_heads, window_size, qk_norm, eps
        self.q, self.k, self.v, self.o = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.norm_q, self.norm_k = WanRMSNorm(dim, eps=eps), WanRMSNorm(dim, eps=eps)


class WanCrossAttention(WanSelfAttention):
    pass


class WanAttentionBlock(nn.Module):
    def __init__(self, cross_attn_type, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True, cross_attn_norm=False, eps=1e-6):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads, self.window_size, self.qk_norm, self.cross_attn_norm, self.eps = dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanCrossAttention(dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate='tanh'), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens):
        grid = tuple(int(v) for v in grid_sizes[0].tolist())
        cos, sin = br.rope_tables(freqs.real.float(), freqs.imag.float(), grid)
        return br.wan_block(dict(self.named_parameters()), x, e, context, self.num_heads, cos.to(x.device), sin.to(x.device), self.eps)


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, out_dim * patch_size[0] * patch_size[1] * patch_size[2])
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e):
        return br.wan_head(dict(self.named_parameters()), x, e, self.eps)


class TransformerLayer(nn.Module):
    """call shape of models/wan/wan.py:514-529"""

    def __init__(self, block):
        super().__init__()
        self.block = block

    def forward(self, inputs):
        x, e, e0, seq_lens, grid_sizes, freqs, context = inputs
        return (self.block(x, e0, seq_lens, grid_sizes, freqs, context, None), e, e0, seq_lens, grid_sizes, freqs, context)


def rope_freqs(head_dim, max_len=1024):
    """the complex table of models/wan/model.py:28-37,478-483: cat of three rope_params tables"""
    def params(dim):
        f = torch.outer(torch.arange(max_len, dtype=torch.float64), 1.0 / torch.pow(10000, torch.arange(0, dim, 2, dtype=torch.float64) / dim))
        return torch.polar(torch.ones_like(f), f)
    d = head_dim
    return torch.cat([params(d - 4 * (d // 6)), params(2 * (d // 6)), params(2 * (d // 6))], dim=1)
