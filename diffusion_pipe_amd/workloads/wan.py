"""Wan2.1 DiT block on the MI355X kernels (BASELINE config 4's block arithmetic; the one hot-path block that lives in
the reference tree: models/wan/model.py:70-181,237-343, parameter names kept so checkpoints map 1:1).

Every op below is a HIP kernel of libdpipe_hip.so: Linear = MFMA GEMM (K1/K6), WanRMSNorm (K2), RoPE on interleaved
pairs (K3), flash / unfused attention (K4), LayerNorm fused with the AdaLN scale/shift and the gated residual (K5),
GELU(tanh) (K6), sinusoidal timestep features (K7).  Parity: tests/test_gpu_wan.py against vectors minted from the
reference's own code (oracle/make_golden.py).
"""
import torch
from torch import nn

from .. import nn as dnn
from .. import ops


def rope_tables(freqs, grid):
    """cos / sin tables [S, d/2] (fp32) for one (frames, height, width) grid from the reference's complex table
    [1024, d/2] = cat(rope_params(d - 4(d//6)), rope_params(2(d//6)), rope_params(2(d//6))) (models/wan/model.py:40-58,478-483)."""
    f, h, w = grid
    c = freqs.shape[1]
    a, b, cc = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    t = torch.cat([a[:f].view(f, 1, 1, -1).expand(f, h, w, -1), b[:h].view(1, h, 1, -1).expand(f, h, w, -1),
                   cc[:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, -1)
    return t.real.float().contiguous(), t.imag.float().contiguous()


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, eps=1e-6):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = dnn.Linear(dim, dim), dnn.Linear(dim, dim), dnn.Linear(dim, dim), dnn.Linear(dim, dim)
        self.norm_q, self.norm_k = dnn.RMSNorm(dim, eps=eps), dnn.RMSNorm(dim, eps=eps)
        self.attn_impl = 'auto'

    def forward(self, x, cos, sin):
        B, S, _ = x.shape
        n, d = self.num_heads, self.head_dim
        q = ops.rope(self.norm_q(self.q(x)).view(B, S, n, d), cos, sin, interleaved=True)
        k = ops.rope(self.norm_k(self.k(x)).view(B, S, n, d), cos, sin, interleaved=True)
        v = self.v(x).view(B, S, n, d)
        return self.o(ops.attention(q, k, v, impl=self.attn_impl).reshape(B, S, n * d))


class WanCrossAttention(WanSelfAttention):
    def forward(self, x, context, context_lens=None):
        B, S, _ = x.shape
        n, d = self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(B, S, n, d)
        k = self.norm_k(self.k(context)).view(B, -1, n, d)
        v = self.v(context).view(B, -1, n, d)
        return self.o(ops.attention(q, k, v, kv_len=context_lens, impl=self.attn_impl).reshape(B, S, n * d))


class FFN(nn.Sequential):
    """nn.Sequential(Linear, GELU(tanh), Linear): indices 0 and 2 carry the parameters, like the reference."""

    def __init__(self, dim, ffn_dim):
        super().__init__(dnn.Linear(dim, ffn_dim), dnn.GELU(approximate='tanh'), dnn.Linear(ffn_dim, dim))


class WanAttentionBlock(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, cross_attn_norm=True, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.norm1 = dnn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.self_attn = WanSelfAttention(dim, num_heads, eps)
        self.norm3 = dnn.LayerNorm(dim, eps=eps, elementwise_affine=True) if cross_attn_norm else None
        self.cross_attn = WanCrossAttention(dim, num_heads, eps)
        self.norm2 = dnn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.ffn = FFN(dim, ffn_dim)
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, e, cos, sin, context, context_lens=None):
        """x: [B, S, C]; e: [B, 1, 6, C] (fp32 modulation of the timestep); cos / sin: rope_tables()."""
        m = (self.modulation.unsqueeze(0) + e).chunk(6, dim=2)                 # six [B, 1, 1, C]
        shift1, scale1, gate1, shift2, scale2, gate2 = (t.reshape(t.shape[0], -1) for t in m)
        y = self.self_attn(self.norm1(x, scale=scale1, shift=shift1), cos, sin)        # LN * (1 + scale) + shift, one pass
        x = ops.gated_residual(x, y, gate1)                                              # x + y * gate
        n3 = self.norm3(x) if self.norm3 is not None else x
        x = ops.gated_residual(x, self.cross_attn(n3, context, context_lens))
        y = self.ffn(self.norm2(x, scale=scale2, shift=shift2))
        return ops.gated_residual(x, y, gate2)


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.norm = dnn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.head = dnn.Linear(dim, out_dim * patch_size[0] * patch_size[1] * patch_size[2])
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e):
        """e: [B, 1, C]."""
        m = (self.modulation.unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)
        shift, scale = (t.reshape(t.shape[0], -1) for t in m)
        return self.head(self.norm(x, scale=scale, shift=shift))
