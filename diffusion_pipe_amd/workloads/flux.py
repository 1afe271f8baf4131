"""Flux (BASELINE config 3: Flux.1-dev, LoRA, pp = 2) on the MI355X kernels: diffusers' FluxTransformer2DModel blocks with
diffusers' module / parameter names (state dicts interchange), the reference adapter's layer wrappers
(models/flux.py:456-548: EmbeddingWrapper, TransformerWrapper, SingleTransformerWrapper, OutputWrapper -- same stage-boundary
tuple `(hidden, encoder, temb, freqs_cos, freqs_sin, img_seq_len)`), `prepare_inputs` (models/flux.py:323-394: logit-normal /
uniform t, shift / flux_shift, x_t = (1-t) x1 + t x0, target = x0 - x1, 2x2 patchify, position ids, guidance vector) and
`to_layers()` = 1 + num_layers + num_single_layers + 1 (:396-404).

Per double block: two AdaLN-Zero modulations (SiLU -> Linear -> 6 chunks), LayerNorm fused with scale / shift (K5), one fused
QKV GEMM per stream (K1), per-head RMSNorm of q / k (K2), RoPE over the joint [text ; image] sequence (K3), flash attention
(K4), output projections, gated residuals (K5), GELU-tanh feed-forward (K6).  Single block: the same on the concatenated
sequence with the parallel MLP branch and one output GEMM over [attention | mlp].
The diffusers package is absent from the image: the block arithmetic is restated from its published definition (parity unpinned);
the wrappers, to_layers and prepare_inputs are pinned by the reference's own code (oracle/make_golden_flux_layers.py,
oracle/make_golden_reflogic.py; oracle side: oracle/flux_ref.py).
"""
import math
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import nn as dnn
from .. import ops


def make_contiguous(*values):
    return tuple(x.contiguous() if torch.is_tensor(x) else x for x in values)


@dataclass
class FluxConfig:
    """configs/flux_dev_config.json of the reference."""
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)

    @property
    def dim(self):
        return self.num_attention_heads * self.attention_head_dim


def tiny_flux_config():
    """Same topology, head_dim 64 (axes 8 + 28 + 28), 2 double + 3 single blocks: parity tests the oracle finishes in seconds."""
    return FluxConfig(in_channels=16, num_layers=2, num_single_layers=3, attention_head_dim=64, num_attention_heads=2,
                      joint_attention_dim=96, pooled_projection_dim=48, axes_dims_rope=(8, 28, 28))


# --------------------------------------------------------------------------------------------------------- embeddings
class TextProjection(nn.Module):
    """PixArtAlphaTextProjection(act_fn='silu')."""

    def __init__(self, in_features, hidden):
        super().__init__()
        self.linear_1, self.act_1, self.linear_2 = dnn.Linear(in_features, hidden), dnn.SiLU(), dnn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(self.act_1(self.linear_1(x)))


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    def __init__(self, dim, pooled_dim, guidance=True):
        super().__init__()
        self.time_proj = dnn.Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0.0)
        self.timestep_embedder = dnn.TimestepEmbedding(256, dim)
        if guidance:
            self.guidance_embedder = dnn.TimestepEmbedding(256, dim)
        self.text_embedder = TextProjection(pooled_dim, dim)
        self.guidance = guidance

    def forward(self, timestep, guidance, pooled):
        dt = self.text_embedder.linear_1.weight.dtype
        emb = self.timestep_embedder(self.time_proj(timestep).to(dt))
        if self.guidance:
            emb = emb + self.guidance_embedder(self.time_proj(guidance).to(dt))
        return emb + self.text_embedder(pooled.to(dt))


class FluxPosEmbed(nn.Module):
    """cos / sin tables [S, sum(axes_dim)] in fp32 from float64 angles, each frequency repeated for its (2i, 2i+1) pair."""

    def __init__(self, theta, axes_dim):
        super().__init__()
        self.theta, self.axes_dim = theta, axes_dim

    def forward(self, ids):
        cos, sin = [], []
        pos = ids.double()
        for i, d in enumerate(self.axes_dim):
            freqs = 1.0 / (self.theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=ids.device) / d))
            ang = pos[:, i, None] * freqs[None]
            cos.append(ang.cos().repeat_interleave(2, dim=1))
            sin.append(ang.sin().repeat_interleave(2, dim=1))
        return torch.cat(cos, dim=-1).float(), torch.cat(sin, dim=-1).float()


def _half_tables(cos, sin):
    """The kernels take one angle per (2i, 2i+1) pair: [S, D] boundary tables -> [S, D/2]."""
    return cos[:, 0::2].contiguous(), sin[:, 0::2].contiguous()


# ---------------------------------------------------------------------------------------------------------- attention
def _project3(x, a, b, c):
    """Three projections of x as ONE GEMM when the layers are plain Linears (no adapter around them) -> [B, S, 3, inner]."""
    B, S, _ = x.shape
    if all(type(m) is dnn.Linear for m in (a, b, c)):
        return ops.fused_linear(x, [a.weight, b.weight, c.weight], [a.bias, b.bias, c.bias]).view(B, S, 3, -1)
    return torch.stack([a(x), b(x), c(x)], dim=2)


class FluxAttention(nn.Module):
    def __init__(self, dim, heads, head_dim, added_kv=False, pre_only=False, eps=1e-6):
        super().__init__()
        self.heads, self.head_dim = heads, head_dim
        inner = heads * head_dim
        self.to_q, self.to_k, self.to_v = dnn.Linear(dim, inner), dnn.Linear(dim, inner), dnn.Linear(dim, inner)
        self.norm_q, self.norm_k = dnn.RMSNorm(head_dim, eps=eps), dnn.RMSNorm(head_dim, eps=eps)
        if not pre_only:
            self.to_out = nn.ModuleList([dnn.Linear(inner, dim), dnn.Identity()])
        if added_kv:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = dnn.Linear(dim, inner), dnn.Linear(dim, inner), dnn.Linear(dim, inner)
            self.norm_added_q, self.norm_added_k = dnn.RMSNorm(head_dim, eps=eps), dnn.RMSNorm(head_dim, eps=eps)
            self.to_add_out = dnn.Linear(inner, dim)

    def _qkv(self, x, projections, norm_q, norm_k, cos, sin, offset):
        """per-head RMSNorm of q / k fused with the rotation (K2 + K3, one pass over each, read straight from the fused projection's output); `offset`: row of the
        [text ; image] angle tables at which this stream's tokens start"""
        B, S, _ = x.shape
        q, k, v = _project3(x, *projections).view(B, S, 3, self.heads, self.head_dim).unbind(2)
        return (ops.rms_norm_rope(q, norm_q.weight, cos, sin, norm_q.eps, per_head=True, token_offset=offset),
                ops.rms_norm_rope(k, norm_k.weight, cos, sin, norm_k.eps, per_head=True, token_offset=offset), v)

    def forward(self, hidden, encoder, cos, sin):
        """-> attention output over [text ; image] tokens, [B, L + S, inner] (text first)."""
        L = encoder.shape[1] if encoder is not None else 0
        q, k, v = self._qkv(hidden, (self.to_q, self.to_k, self.to_v), self.norm_q, self.norm_k, cos, sin, L)
        if encoder is not None:
            eq, ek, ev = self._qkv(encoder, (self.add_q_proj, self.add_k_proj, self.add_v_proj), self.norm_added_q, self.norm_added_k, cos, sin, 0)
            q, k, v = torch.cat([eq, q], dim=1), torch.cat([ek, k], dim=1), torch.cat([ev, v], dim=1)
        o = ops.attention(q, k, v.contiguous())
        return o.reshape(o.shape[0], o.shape[1], -1)


# -------------------------------------------------------------------------------------------------------------- blocks
class AdaLayerNormZero(nn.Module):
    """emb = linear(silu(temb)) -> n chunks (n = 6: shift / scale / gate of attention and of the MLP; n = 3: single blocks);
    returns LN(x) * (1 + scale) + shift, the norm node's alias of x for the residual branch (its gradient is then added inside
    the LayerNorm backward kernel) and the remaining chunks."""

    def __init__(self, dim, chunks=6):
        super().__init__()
        self.silu, self.linear, self.chunks = dnn.SiLU(), dnn.Linear(dim, chunks * dim), chunks
        self.norm = dnn.LayerNorm(dim, eps=1e-6, elementwise_affine=False)

    def forward(self, x, temb):
        parts = self.linear(self.silu(temb)).chunk(self.chunks, dim=1)
        y, skip = self.norm(x, scale=parts[1], shift=parts[0], with_skip=True)
        return (y, skip, *parts[2:])


class GELUProjection(nn.Module):
    """diffusers GELU(dim_in, dim_out, approximate='tanh'): `.proj` Linear then GELU."""

    def __init__(self, dim, inner):
        super().__init__()
        self.proj, self.act = dnn.Linear(dim, inner), dnn.GELU(approximate='tanh')

    def forward(self, x):
        return self.act(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GELUProjection(dim, 4 * dim), dnn.Identity(), dnn.Linear(4 * dim, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim):
        super().__init__()
        self.norm1, self.norm1_context = AdaLayerNormZero(dim), AdaLayerNormZero(dim)
        self.attn = FluxAttention(dim, heads, head_dim, added_kv=True)
        self.norm2 = dnn.LayerNorm(dim, eps=1e-6, elementwise_affine=False)
        self.ff = FeedForward(dim)
        self.norm2_context = dnn.LayerNorm(dim, eps=1e-6, elementwise_affine=False)
        self.ff_context = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb):
        cos, sin = _half_tables(*image_rotary_emb)
        h, hidden_states, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden_states, temb)
        c, encoder_hidden_states, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder_hidden_states, temb)
        L = encoder_hidden_states.shape[1]
        o = self.attn(h, c, cos, sin)
        x = ops.gated_residual(hidden_states, self.attn.to_out[0](o[:, L:].contiguous()), gate_msa)
        n, x = self.norm2(x, scale=scale_mlp, shift=shift_mlp, with_skip=True)
        x = ops.gated_residual(x, self.ff(n), gate_mlp)
        e = ops.gated_residual(encoder_hidden_states, self.attn.to_add_out(o[:, :L].contiguous()), c_gate_msa)
        n, e = self.norm2_context(e, scale=c_scale_mlp, shift=c_shift_mlp, with_skip=True)
        e = ops.gated_residual(e, self.ff_context(n), c_gate_mlp)
        return e, x


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim):
        super().__init__()
        self.norm = AdaLayerNormZero(dim, chunks=3)
        self.proj_mlp, self.act_mlp = dnn.Linear(dim, 4 * dim), dnn.GELU(approximate='tanh')
        self.proj_out = dnn.Linear(5 * dim, dim)
        self.attn = FluxAttention(dim, heads, head_dim, pre_only=True)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb):
        cos, sin = _half_tables(*image_rotary_emb)
        L = encoder_hidden_states.shape[1]
        x = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        n, x, gate = self.norm(x, temb)
        mlp = self.act_mlp(self.proj_mlp(n))
        attn = self.attn(n, None, cos, sin)
        x = ops.gated_residual(x, self.proj_out(torch.cat([attn, mlp], dim=2)), gate)
        return x[:, :L], x[:, L:]


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.silu, self.linear = dnn.SiLU(), dnn.Linear(dim, 2 * dim)
        self.norm = dnn.LayerNorm(dim, eps=1e-6, elementwise_affine=False)

    def forward(self, x, conditioning):
        scale, shift = self.linear(self.silu(conditioning.to(self.linear.weight.dtype))).chunk(2, dim=1)      # scale first
        return self.norm(x, scale=scale, shift=shift)


class FluxTransformer2DModel(nn.Module):
    def __init__(self, c: FluxConfig):
        super().__init__()
        self.config = c
        dim = c.dim
        self.pos_embed = FluxPosEmbed(10000, c.axes_dims_rope)
        self.time_text_embed = CombinedTimestepGuidanceTextProjEmbeddings(dim, c.pooled_projection_dim, c.guidance_embeds)
        self.context_embedder = dnn.Linear(c.joint_attention_dim, dim)
        self.x_embedder = dnn.Linear(c.in_channels, dim)
        self.transformer_blocks = nn.ModuleList(FluxTransformerBlock(dim, c.num_attention_heads, c.attention_head_dim)
                                                for _ in range(c.num_layers))
        self.single_transformer_blocks = nn.ModuleList(FluxSingleTransformerBlock(dim, c.num_attention_heads, c.attention_head_dim)
                                                       for _ in range(c.num_single_layers))
        self.norm_out = AdaLayerNormContinuous(dim)
        self.proj_out = dnn.Linear(dim, c.in_channels)


# ------------------------------------------------------------------------- pipeline layers (models/flux.py:456-548)
class EmbeddingWrapper(nn.Module):
    def __init__(self, x_embedder, time_text_embed, context_embedder, pos_embed):
        super().__init__()
        self.x_embedder, self.time_text_embed = x_embedder, time_text_embed
        self.context_embedder, self.pos_embed = context_embedder, pos_embed

    def forward(self, inputs):
        for item in inputs:
            if torch.is_floating_point(item):
                item.requires_grad_(True)
        hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance, img_seq_len = inputs
        wdtype = self.x_embedder.weight.dtype
        hidden_states = self.x_embedder(hidden_states.to(wdtype))
        temb = self.time_text_embed(timestep.float() * 1000, guidance.float() * 1000, pooled_projections)
        encoder_hidden_states = self.context_embedder(encoder_hidden_states.to(wdtype))
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        freqs_cos, freqs_sin = self.pos_embed(torch.cat((txt_ids, img_ids), dim=0))
        return make_contiguous(hidden_states, encoder_hidden_states, temb, freqs_cos, freqs_sin, img_seq_len)


class TransformerWrapper(nn.Module):
    def __init__(self, block, block_idx=0):
        super().__init__()
        self.block, self.block_idx = block, block_idx

    def forward(self, inputs):
        hidden_states, encoder_hidden_states, temb, freqs_cos, freqs_sin, img_seq_len = inputs
        encoder_hidden_states, hidden_states = self.block(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                          temb=temb, image_rotary_emb=(freqs_cos, freqs_sin))
        return make_contiguous(hidden_states, encoder_hidden_states, temb, freqs_cos, freqs_sin, img_seq_len)


class SingleTransformerWrapper(TransformerWrapper):
    pass


class OutputWrapper(nn.Module):
    def __init__(self, norm_out, proj_out, image_tokens_of):
        super().__init__()
        self.norm_out, self.proj_out, self.image_tokens_of = norm_out, proj_out, image_tokens_of

    def forward(self, inputs):
        hidden_states, encoder_hidden_states, temb, freqs_cos, freqs_sin, img_seq_len = inputs
        # the reference reads img_seq_len[0].item() here (a device sync per micro-batch); the host already knows the count from
        # prepare_inputs, which keeps this layer capturable in a hipGraph
        n = self.image_tokens_of(img_seq_len)
        hidden_states = hidden_states[:, :n, ...]
        return self.proj_out(self.norm_out(hidden_states, temb))


# --------------------------------------------------------------------------------------------------------- the adapter
def patchify(x):
    """'b c (h ph) (w pw) -> b (h w) (c ph pw)' with ph = pw = 2 (models/flux.py:377-378)."""
    b, c, H, W = x.shape
    return x.reshape(b, c, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(b, (H // 2) * (W // 2), c * 4)


def prepare_latent_image_ids(h, w):
    ids = torch.zeros(h, w, 3)
    ids[..., 1] = torch.arange(h, dtype=torch.float32)[:, None]
    ids[..., 2] = torch.arange(w, dtype=torch.float32)[None, :]
    return ids.reshape(h * w, 3)


class FluxWorkload:
    name = 'flux'
    checkpointable_layers = ['TransformerWrapper', 'SingleTransformerWrapper']
    adapter_target_modules = ['FluxTransformerBlock', 'FluxSingleTransformerBlock']

    def __init__(self, config: FluxConfig = None, model_config=None, dtype=torch.bfloat16, seed=0, device='cpu'):
        self.cfg = config or FluxConfig()
        self.model_config = dict(model_config or {})
        self.model_config.setdefault('guidance', 1.0)          # train.py:113
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        with torch.device(device):               # a GPU workload is initialised in HBM (12 B parameters: seconds instead of minutes of host RNG); 'cpu' = the seeded host stream
            self.transformer = FluxTransformer2DModel(self.cfg)
        torch.random.set_rng_state(state)
        self.transformer.to(dtype=dtype)
        for n, p in self.transformer.named_parameters():
            p.original_name = n
        self._image_tokens = None

    def configure_adapter(self, adapter_config):
        """LoRA on every Linear inside the double / single blocks (models/base.py:262-297, models/flux.py:163)."""
        if adapter_config.get('type', 'lora') != 'lora':
            raise NotImplementedError(f"Adapter type {adapter_config['type']} is not implemented")
        inside = set()
        for name, module in self.transformer.named_modules():
            if module.__class__.__name__ in self.adapter_target_modules:
                inside.update(f'{name}.{n}' for n, _ in module.named_modules() if n)
        wrapped = dnn.apply_lora(self.transformer, rank=adapter_config['rank'], alpha=adapter_config['alpha'],
                                 dropout=adapter_config.get('dropout', 0.0), dtype=adapter_config.get('dtype'),
                                 target=lambda name, module: name in inside)
        for n, p in self.transformer.named_parameters():
            p.original_name = n
        return wrapped

    def to_layers(self):
        t = self.transformer
        layers = [EmbeddingWrapper(t.x_embedder, t.time_text_embed, t.context_embedder, t.pos_embed)]
        layers += [TransformerWrapper(b, i) for i, b in enumerate(t.transformer_blocks)]
        layers += [SingleTransformerWrapper(b, i) for i, b in enumerate(t.single_transformer_blocks)]
        layers.append(OutputWrapper(t.norm_out, t.proj_out, lambda img_seq_len: self._image_tokens))
        return layers

    def sample_timesteps(self, bs, image_tokens, timestep_quantile=None):
        mc = self.model_config
        method = mc.get('timestep_sample_method', 'logit_normal')
        if method == 'logit_normal':
            dist = torch.distributions.normal.Normal(0, 1)
        elif method == 'uniform':
            dist = torch.distributions.uniform.Uniform(0, 1)
        else:
            raise NotImplementedError()
        t = dist.icdf(torch.full((bs,), float(timestep_quantile))) if timestep_quantile is not None else dist.sample((bs,))
        if method == 'logit_normal':
            t = torch.sigmoid(t * mc.get('sigmoid_scale', 1.0))
        if shift := mc.get('shift', None):
            t = (t * shift) / (1 + (shift - 1) * t)
        elif mc.get('flux_shift', False):
            slope = (1.15 - 0.5) / (4096 - 256)
            mu = slope * image_tokens + (0.5 - slope * 256)
            t = math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** 1.0)
        return t

    def prepare_inputs(self, inputs, timestep_quantile=None):
        latents = inputs['latents'].float()
        clip_embed, t5_embed, mask = inputs['clip_embed'], inputs['t5_embed'], inputs.get('mask')
        bs, c, h, w = latents.shape
        if mask is not None:
            mask = mask.unsqueeze(1).expand((-1, c, -1, -1))
            mask = patchify(F.interpolate(mask, size=(h, w), mode='nearest-exact'))
        img_ids = prepare_latent_image_ids(h // 2, w // 2).unsqueeze(0).repeat((bs, 1, 1))
        txt_ids = torch.zeros(bs, t5_embed.shape[1], 3)
        self._image_tokens = (h // 2) * (w // 2)
        t = self.sample_timesteps(bs, self._image_tokens, timestep_quantile)
        x_1 = latents
        x_0 = torch.randn_like(x_1)
        te = t.view(-1, 1, 1, 1)
        x_t = patchify((1 - te) * x_1 + te * x_0)
        target = patchify(x_0 - x_1)
        guidance_vec = torch.full((bs,), float(self.model_config['guidance']), dtype=torch.float32)
        img_seq_len = torch.tensor(x_t.shape[1]).repeat((bs,))
        return (x_t, t5_embed, clip_embed, t, img_ids, txt_ids, guidance_vec, img_seq_len), (target, mask)

    def get_loss_fn(self):
        def loss_fn(output, label):
            target, mask = label
            return ops.fused_loss(output, target, mask if mask.numel() > 0 else None)
        return loss_fn

    def get_param_groups(self, parameters):
        return [{'params': list(parameters)}]

    # ---- saved files (models/flux.py:231-290)
    def save_adapter(self, save_dir, peft_state_dict):
        """LoRA -> diffusers-format pytorch_lora_weights.safetensors."""
        from ..formats import save_flux_diffusers_lora
        save_flux_diffusers_lora(save_dir, peft_state_dict)

    def save_model(self, save_dir, diffusers_sd):
        """Full fine-tune -> one BFL-layout model.safetensors."""
        from ..formats import save_flux_bfl
        save_flux_bfl(save_dir, diffusers_sd)


def synthetic_flux_batch(cfg: FluxConfig, batch_size=1, latent_hw=(16, 16), text_tokens=24, seed=0):
    """SURVEY 8(d) config 3 shapes: latents randn[B, C/4, h, w], t5_embed randn[B, T, 4096], clip_embed randn[B, 768]."""
    g = torch.Generator().manual_seed(seed)
    return {'latents': torch.randn(batch_size, cfg.in_channels // 4, *latent_hw, generator=g), 'mask': None,
            't5_embed': torch.randn(batch_size, text_tokens, cfg.joint_attention_dim, generator=g),
            'clip_embed': torch.randn(batch_size, cfg.pooled_projection_dim, generator=g)}
