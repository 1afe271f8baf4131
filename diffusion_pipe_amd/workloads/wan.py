"""Wan2.1 DiT block on the MI355X kernels (BASELINE config 4's block arithmetic; the one hot-path block that lives in
the reference tree: models/wan/model.py:70-181,237-343, parameter names kept so checkpoints map 1:1).

Every op below is a HIP kernel of libdpipe_hip.so: Linear = MFMA GEMM (K1/K6), WanRMSNorm (K2), RoPE on interleaved
pairs (K3), flash / unfused attention (K4), LayerNorm fused with the AdaLN scale/shift and the gated residual (K5),
GELU(tanh) (K6), sinusoidal timestep features (K7).  Parity: tests/test_gpu_wan.py against vectors minted from the
reference's own code (oracle/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F
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
        # K2 + K3 in one pass each: RMSNorm over the whole token, RoPE per head, q / k written once (models/wan/model.py:124-125,139-140)
        q = ops.rms_norm_rope(self.q(x).view(B, S, n, d), self.norm_q.weight, cos, sin, self.norm_q.eps, per_head=False)
        k = ops.rms_norm_rope(self.k(x).view(B, S, n, d), self.norm_k.weight, cos, sin, self.norm_k.eps, per_head=False)
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
        # with_skip: the residual branch takes the norm node's alias of x; its gradient is added inside the LayerNorm backward kernel
        h, x = self.norm1(x, scale=scale1, shift=shift1, with_skip=True)                 # LN * (1 + scale) + shift, one pass
        x = ops.gated_residual(x, self.self_attn(h, cos, sin), gate1)                    # x + y * gate
        if self.norm3 is not None:
            n3, x = self.norm3(x, with_skip=True)
        else:
            n3 = x
        x = ops.gated_residual(x, self.cross_attn(n3, context, context_lens))
        h, x = self.norm2(x, scale=scale2, shift=shift2, with_skip=True)
        return ops.gated_residual(x, self.ffn(h), gate2)


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


# ================================================================================== whole model + pipeline layer wrappers
from dataclasses import dataclass  # noqa: E402
from typing import Tuple  # noqa: E402

from ..data import get_t_distribution, sample_t, slice_t_distribution  # noqa: E402


@dataclass
class WanConfig:
    """models/wan/configs.py:54-75 (t2v-14B: dim 5120, ffn 13824, 40 heads, 40 layers); the test config keeps head_dim 64."""
    in_dim: int = 16
    dim: int = 5120
    ffn_dim: int = 13824
    freq_dim: int = 256
    text_dim: int = 4096
    out_dim: int = 16
    num_heads: int = 40
    num_layers: int = 40
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    text_len: int = 512
    eps: float = 1e-6
    cross_attn_norm: bool = True


def tiny_wan_config():
    return WanConfig(dim=128, ffn_dim=256, text_dim=64, num_heads=2, num_layers=2, text_len=24)


def rope_params(max_seq_len, dim, theta=10000.0):
    """models/wan/model.py:29-37 (complex table [max_seq_len, dim/2])."""
    freqs = torch.outer(torch.arange(max_seq_len, dtype=torch.float32),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    return torch.polar(torch.ones_like(freqs), freqs)


class PatchEmbed3d(nn.Module):
    """nn.Conv3d(in, dim, kernel = stride = patch) (models/wan/model.py:449-450) as patchify + one MFMA GEMM; the parameter keeps
    the Conv3d shape [dim, in, pt, ph, pw] so checkpoints load unchanged."""

    def __init__(self, in_dim, dim, patch):
        super().__init__()
        self.patch = patch
        self.weight = nn.Parameter(torch.empty(dim, in_dim, *patch))
        self.bias = nn.Parameter(torch.zeros(dim))
        nn.init.xavier_uniform_(self.weight.flatten(1))

    def forward(self, x):
        B, C, F_, H, W = x.shape
        pt, ph, pw = self.patch
        f, h, w = F_ // pt, H // ph, W // pw
        patches = x.view(B, C, f, pt, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, f * h * w, C * pt * ph * pw)
        return ops.linear(patches.to(self.weight.dtype), self.weight.flatten(1), self.bias), (f, h, w)


class WanModel(nn.Module):
    def __init__(self, c: WanConfig):
        super().__init__()
        self.config = c
        self.patch_embedding = PatchEmbed3d(c.in_dim, c.dim, c.patch_size)
        self.text_embedding = nn.Sequential(dnn.Linear(c.text_dim, c.dim), dnn.GELU(approximate='tanh'), dnn.Linear(c.dim, c.dim))
        self.time_embedding = nn.Sequential(dnn.Linear(c.freq_dim, c.dim), dnn.SiLU(), dnn.Linear(c.dim, c.dim))
        self.time_projection = nn.Sequential(dnn.SiLU(), dnn.Linear(c.dim, c.dim * 6))
        self.blocks = nn.ModuleList([WanAttentionBlock(c.dim, c.ffn_dim, c.num_heads, c.cross_attn_norm, c.eps) for _ in range(c.num_layers)])
        self.head = Head(c.dim, c.out_dim, c.patch_size, c.eps)
        d = c.dim // c.num_heads
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)), rope_params(1024, 2 * (d // 6))], dim=1)

    def unpatchify(self, x, grid):
        """models/wan/model.py:495-518 for equal grids: [B, L, out * prod(patch)] -> [B, out, F, H, W]."""
        c = self.config.out_dim
        f, h, w = grid
        pt, ph, pw = self.config.patch_size
        u = x.view(x.shape[0], f, h, w, pt, ph, pw, c)
        return torch.einsum('bfhwpqrc->bcfphqwr', u).reshape(x.shape[0], c, f * pt, h * ph, w * pw)


def make_contiguous(*values):
    return tuple(x.contiguous() if torch.is_tensor(x) else x for x in values)


class InitialLayer(nn.Module):
    """models/wan/wan.py:414-511 for t2v with cached text embeddings: patch embedding, time embedding / projection, text
    embedding (padded to text_len with zeros, all positions attended like the reference), rotary tables of the grid."""

    def __init__(self, model: WanModel):
        super().__init__()
        self.patch_embedding, self.time_embedding = model.patch_embedding, model.time_embedding
        self.text_embedding, self.time_projection = model.text_embedding, model.time_projection
        self.freqs, self.freq_dim, self.dim, self.text_len = model.freqs, model.config.freq_dim, model.config.dim, model.config.text_len

    def forward(self, inputs):
        for item in inputs:
            if torch.is_floating_point(item):
                item.requires_grad_(True)
        x, y, t, text_embeddings, seq_lens, clip_fea = inputs
        wdtype = self.patch_embedding.weight.dtype
        x, grid = self.patch_embedding(x)
        e = self.time_embedding(ops.sinusoidal_embedding(t.flatten(), self.freq_dim).to(wdtype)).unsqueeze(1)        # [B, 1, C]
        e0 = self.time_projection(e).unflatten(2, (6, self.dim))                                                         # [B, 1, 6, C]
        valid = (torch.arange(text_embeddings.shape[1], device=text_embeddings.device)[None, :] < seq_lens[:, None]).unsqueeze(-1)
        ctx = torch.where(valid, text_embeddings, torch.zeros_like(text_embeddings))
        if ctx.shape[1] < self.text_len:
            ctx = torch.cat([ctx, ctx.new_zeros(ctx.shape[0], self.text_len - ctx.shape[1], ctx.shape[2])], dim=1)
        context = self.text_embedding(ctx.to(wdtype))
        if self.freqs.device != x.device:
            self.freqs = self.freqs.to(x.device)
        cos, sin = rope_tables(self.freqs, grid)
        grid_sizes = torch.stack([torch.full((x.shape[0],), g, dtype=torch.long, device=x.device) for g in grid], dim=1)   # fills: capture-safe
        return make_contiguous(x, e, e0, seq_lens, grid_sizes, cos, sin, context)


class TransformerLayer(nn.Module):
    """models/wan/wan.py:514-529."""

    def __init__(self, block):
        super().__init__()
        self.block = block

    def forward(self, inputs):
        x, e, e0, seq_lens, grid_sizes, cos, sin, context = inputs
        x = self.block(x, e0, cos, sin, context)
        return make_contiguous(x, e, e0, seq_lens, grid_sizes, cos, sin, context)


class FinalLayer(nn.Module):
    """models/wan/wan.py:532-546."""

    def __init__(self, model: WanModel, grid_of):
        super().__init__()
        self.head, self.model, self.grid_of = model.head, [model], grid_of

    def forward(self, inputs):
        x, e, e0, seq_lens, grid_sizes, cos, sin, context = inputs
        x = self.head(x, e)
        return self.model[0].unpatchify(x, self.grid_of(x.shape[1]))


class WanWorkload:
    """Adapter-API subset over a randomly initialised Wan t2v model: to_layers() = 1 + num_layers + 1 layers
    (models/wan/wan.py:377-384), flow-matching prepare_inputs (:332-375), default loss (models/base.py:418-436)."""
    name = 'wan'
    checkpointable_layers = ['TransformerLayer']

    def __init__(self, config: WanConfig, model_config=None, dtype=torch.bfloat16, seed=0, device='cpu'):
        self.cfg = config
        self.model_config = model_config or {}
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        with torch.device(device):               # a GPU workload is initialised in HBM (14 B parameters); 'cpu' = the seeded host stream
            self.transformer = WanModel(config)
        torch.random.set_rng_state(state)
        self.transformer.to(dtype=dtype)
        for n, p in self.transformer.named_parameters():
            p.original_name = n
        self.t_dist = get_t_distribution(self.model_config)
        self._grid = None

    adapter_target_modules = ['WanAttentionBlock']

    def configure_adapter(self, adapter_config):
        """LoRA on every Linear inside the `adapter_target_modules` blocks (models/base.py:262-297, models/wan/wan.py:70)."""
        if adapter_config.get('type', 'lora') != 'lora':
            raise NotImplementedError(f"Adapter type {adapter_config['type']} is not implemented")
        inside = set()
        for name, module in self.transformer.named_modules():
            if module.__class__.__name__ in self.adapter_target_modules:
                inside.update(f'{name}.{n}' for n, sub in module.named_modules() if n)
        wrapped = dnn.apply_lora(self.transformer, rank=adapter_config['rank'], alpha=adapter_config['alpha'],
                                 dropout=adapter_config.get('dropout', 0.0), dtype=adapter_config.get('dtype'),
                                 target=lambda name, module: name in inside)
        for n, p in self.transformer.named_parameters():
            p.original_name = n
        return wrapped

    def to_layers(self):
        m = self.transformer
        return [InitialLayer(m)] + [TransformerLayer(b) for b in m.blocks] + [FinalLayer(m, lambda L: self._grid)]

    def prepare_inputs(self, inputs, timestep_quantile=None):
        latents = inputs['latents'].float()
        mask = inputs.get('mask')
        bs, _, frames, h, w = latents.shape
        pt, ph, pw = self.cfg.patch_size
        self._grid = (frames // pt, h // ph, w // pw)
        if mask is not None:
            mask = F.interpolate(mask.unsqueeze(1), size=(h, w), mode='nearest-exact').unsqueeze(2)     # [B, 1, 1, h, w] on the latent grid
        t = self.t_dist
        if shift := self.model_config.get('shift', None):
            t = (t * shift) / (1 + (shift - 1) * t)
        elif self.model_config.get('flux_shift', False):                # resolution-dependent shift (utils/common.py:114-121)
            slope = (1.15 - 0.5) / (4096 - 256)
            mu = slope * ((h // 2) * (w // 2)) + (0.5 - slope * 256)
            t = math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** 1.0)
        t = slice_t_distribution(t, min_t=self.model_config.get('min_t', 0.0), max_t=self.model_config.get('max_t', 1.0))
        t = sample_t(t, bs, quantile=timestep_quantile)
        x_1 = latents
        x_0 = torch.randn_like(x_1)
        te = t.view(-1, 1, 1, 1, 1)
        x_t = (1 - te) * x_1 + te * x_0
        target = x_0 - x_1
        return (x_t, None, t * 1000, inputs['text_embeddings'], inputs['seq_lens'], None), (target, mask)

    def get_loss_fn(self):
        def loss_fn(output, label):
            target, mask = label
            return ops.fused_loss(output, target, mask if mask.numel() > 0 else None)
        return loss_fn

    def get_param_groups(self, parameters):
        return [{'params': list(parameters)}]

    # ---- saved files (models/wan/wan.py:258-265, models/base.py:367-386)
    def save_adapter(self, save_dir, peft_state_dict, adapter_config=None):
        from ..formats import save_comfyui_adapter
        save_comfyui_adapter(save_dir, peft_state_dict, adapter_config)

    def save_model(self, save_dir, state_dict):
        from ..formats import save_plain_model
        save_plain_model(save_dir, state_dict)

    def load_adapter_weights(self, adapter_path):
        from ..formats import load_comfyui_adapter
        return load_comfyui_adapter(self.transformer, adapter_path)


def synthetic_wan_batch(cfg: WanConfig, batch_size=1, frames=2, latent_hw=(12, 16), text_tokens=20, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {'latents': torch.randn(batch_size, cfg.in_dim, frames, *latent_hw, generator=g), 'mask': None,
            'text_embeddings': torch.randn(batch_size, text_tokens, cfg.text_dim, generator=g),
            'seq_lens': torch.full((batch_size,), text_tokens - 3, dtype=torch.long)}
