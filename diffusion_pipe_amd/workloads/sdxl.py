"""SDXL on the MI355X engine: UNet2DCondition + both CLIP text encoders, built from the HIP-backed modules of
diffusion_pipe_amd.nn, and the layer-per-stage wrappers of the reference adapter.

What follows the reference (models/sdxl.py): the adapter surface `to_layers()` (:591-602, 23 layers: 1 + 3 + 3 + 2 +
2 + 4 + 4 + 3 + 1), `prepare_inputs()` (:538-579), `get_loss_fn()` (:632-651 with the SNR weights of :281-355),
`get_param_groups()` (:604-630) and the wrappers InitialLayer ... FinalLayer (:654-995) that carry the UNet skip stack
through the pipeline as a flat tuple `(hidden, timesteps, emb, encoder_hidden_states, *skips, forward_upsample_size)`.
What stands in for third-party code absent from the reference snapshot ([3P] diffusers UNet2DConditionModel, HF
CLIPTextModel; parity unpinned): the module definitions below, with the libraries' parameter names so real
checkpoints map 1:1.  No checkpoints exist offline: weights are random-initialised (seeded).
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import nn as dnn
from .. import ops


def make_contiguous(*values):
    """Dense tensors for the stage boundary; feature maps ([B, C, H, W] logical shape, as in the reference's tuples) stay channels-last
    in memory: the UNet below runs NHWC (implicit-GEMM convolutions and token-major transformer blocks share one layout)."""
    def dense(x):
        if not torch.is_tensor(x):
            return x
        if x.dim() == 4 and x.is_floating_point():
            return x.contiguous(memory_format=torch.channels_last)
        return x.contiguous()
    return tuple(dense(x) for x in values)


# ----------------------------------------------------------------------------------------------------- configs
@dataclass
class CLIPConfig:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    act: str = 'quick_gelu'
    vocab: int = 49408
    max_pos: int = 77
    proj_dim: Optional[int] = None          # CLIPTextModelWithProjection when set
    bos: int = 49406
    eos: int = 49407
    pad: int = 49407


@dataclass
class SDXLConfig:
    in_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    transformer_layers: Tuple[int, ...] = (0, 2, 10)     # 0 = block without attention
    num_heads: Tuple[int, ...] = (5, 10, 20)
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    norm_groups: int = 32
    te1: CLIPConfig = field(default_factory=CLIPConfig)
    te2: CLIPConfig = field(default_factory=lambda: CLIPConfig(hidden=1280, layers=32, heads=20, mlp=5120, act='gelu', proj_dim=1280, pad=0))
    num_train_timesteps: int = 1000
    vae_scale_factor: int = 8

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    @property
    def add_embed_in(self):
        return self.te2.proj_dim + 6 * self.addition_time_embed_dim


def tiny_config():
    """Scaled-down SDXL with the same topology (23 layers, skip stack, cross-attention) for parity tests."""
    return SDXLConfig(block_out_channels=(64, 128, 256), transformer_layers=(0, 1, 2), num_heads=(1, 2, 4), cross_attention_dim=192,
                      addition_time_embed_dim=32,
                      te1=CLIPConfig(hidden=64, layers=2, heads=1, mlp=128, vocab=1000, bos=998, eos=999, pad=999),
                      te2=CLIPConfig(hidden=128, layers=2, heads=2, mlp=256, act='gelu', vocab=1000, proj_dim=64, bos=998, eos=999, pad=0))


# ------------------------------------------------------------------------------------------------- CLIP encoders
class CLIPAttention(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.heads, self.head_dim = c.heads, c.hidden // c.heads
        self.q_proj, self.k_proj = dnn.Linear(c.hidden, c.hidden), dnn.Linear(c.hidden, c.hidden)
        self.v_proj, self.out_proj = dnn.Linear(c.hidden, c.hidden), dnn.Linear(c.hidden, c.hidden)

    def forward(self, x, residual=None):
        B, S, C = x.shape
        plain = all(type(m) is dnn.Linear for m in (self.q_proj, self.k_proj, self.v_proj))    # False once an adapter wraps them
        if plain and ops.flash_eligible(self.q_proj.weight.dtype, self.head_dim):      # fused QKV GEMM + packed causal attention
            qkv = ops.fused_linear(x, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight],
                                   [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias])
            return self.out_proj(ops.attention_packed(qkv, None, self.heads, self.head_dim, causal=True), residual)
        q = self.q_proj(x).view(B, S, self.heads, self.head_dim)
        k = self.k_proj(x).view(B, S, self.heads, self.head_dim)
        v = self.v_proj(x).view(B, S, self.heads, self.head_dim)
        return self.out_proj(ops.attention(q, k, v, causal=True).reshape(B, S, C), residual)


class CLIPMLP(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.fc1, self.fc2 = dnn.Linear(c.hidden, c.mlp), dnn.Linear(c.mlp, c.hidden)
        self.act = dnn.QuickGELU() if c.act == 'quick_gelu' else dnn.GELU()

    def forward(self, x, residual=None):
        return self.fc2(self.act(self.fc1(x)), residual)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.layer_norm1, self.self_attn = dnn.LayerNorm(c.hidden, eps=1e-5), CLIPAttention(c)
        self.layer_norm2, self.mlp = dnn.LayerNorm(c.hidden, eps=1e-5), CLIPMLP(c)

    def forward(self, x):
        # with_skip: the residual branch consumes the norm node's alias of x, so its gradient is added inside the LayerNorm
        # backward kernel instead of by an autograd accumulation launch
        h, x = self.layer_norm1(x, with_skip=True)
        x = self.self_attn(h, residual=x)
        h, x = self.layer_norm2(x, with_skip=True)
        return self.mlp(h, residual=x)


class CLIPEmbeddings(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.token_embedding = nn.Embedding(c.vocab, c.hidden)
        self.position_embedding = nn.Embedding(c.max_pos, c.hidden)

    def forward(self, input_ids):
        pos = torch.arange(input_ids.shape[1], device=input_ids.device)
        return self.token_embedding(input_ids) + self.position_embedding(pos)[None]


class CLIPEncoder(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(c) for _ in range(c.layers)])


class CLIPTextTransformer(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.embeddings, self.encoder = CLIPEmbeddings(c), CLIPEncoder(c)
        self.final_layer_norm = dnn.LayerNorm(c.hidden, eps=1e-5)


class CLIPTextModel(nn.Module):
    """Returns (penultimate hidden state, pooled / projected embedding or None) -- what InitialLayer consumes
    (`hidden_states[-2]` and `prompt_embeds[0]`, models/sdxl.py:766-777)."""

    def __init__(self, c: CLIPConfig):
        super().__init__()
        self.config = c
        self.text_model = CLIPTextTransformer(c)
        self.text_projection = dnn.Linear(c.hidden, c.proj_dim, bias=False) if c.proj_dim else None

    def forward(self, input_ids, want_pooled=False):
        tm = self.text_model
        x = tm.embeddings(input_ids)
        penultimate = None
        n = len(tm.encoder.layers)
        for i, layer in enumerate(tm.encoder.layers):
            x = layer(x)
            if i == n - 2:
                penultimate = x
        if n == 1:
            penultimate = tm.embeddings(input_ids)
        pooled = None
        if want_pooled:
            last = tm.final_layer_norm(x)
            eos_pos = input_ids.to(torch.int).argmax(dim=-1)             # EOS carries the largest token id
            pooled = last.gather(1, eos_pos.long()[:, None, None].expand(-1, 1, last.shape[-1])).squeeze(1)
            if self.text_projection is not None:
                pooled = self.text_projection(pooled)
        return penultimate, pooled


# ------------------------------------------------------------------------------------------------------- UNet
_DEBUG_ATEN_TEMB_ADD = os.environ.get('DPIPE_DEBUG_ATEN_TEMB_ADD', '0') == '1'      # repro switch for the round-5 graph-replay corruption (never set by the product)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_ch, out_ch, temb_ch, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = dnn.GroupNorm(groups, in_ch, eps=eps)
        self.conv1 = dnn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.time_emb_proj = dnn.Linear(temb_ch, out_ch)
        self.norm2 = dnn.GroupNorm(groups, out_ch, eps=eps)
        self.conv2 = dnn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.nonlinearity = dnn.SiLU()
        self.conv_shortcut = dnn.Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else None

    def forward(self, x, temb):
        x = x.contiguous(memory_format=torch.channels_last)
        if (temb.dtype == torch.float32 and x.dtype == torch.bfloat16 and x.is_cuda and ops.PRECISE_ADDENDS and type(self.time_emb_proj) is dnn.Linear
                and type(self.conv1) is dnn.Conv2d and self.conv1.weight.shape[0] % 4 == 0
                and ops.conv2d_eligible(x.dtype, self.conv1.weight, self.conv1.stride, self.conv1.padding, self.conv1.dilation, self.conv1.groups)):
            # round 6: the addend time_emb_proj(silu(emb)) + conv1.bias is computed from the fp32 embedding with fp32 accuracy and enters conv1's epilogue as a bf16
            # hi / lo pair (one row per sample) -- it shifts every pixel of a channel, so its rounding is a coherent error, not noise
            pair = ops.precise_row_linear(temb, self.time_emb_proj.weight, self.time_emb_proj.bias, extra=self.conv1.bias, act='silu', pair=True)
            h, x = self.norm1(x, act='silu', with_skip=True)
            h = self.conv1(h, extra_bias=pair)
            if self.conv_shortcut is not None:
                x = self.conv_shortcut(x)
            return self.conv2(self.norm2(h, act='silu'), residual=x)
        t = self.time_emb_proj(self.nonlinearity(temb.to(x.dtype) if temb.dtype != x.dtype else temb))
        h, x = self.norm1(x, act='silu', with_skip=True)          # GroupNorm + SiLU in one pass; the skip branch's gradient is folded into its backward
        if x.shape[0] == 1:        # batch 1: the time-embedding addend is one value per channel -> it joins conv1's bias vector in the epilogue
            h = self.conv1(h, extra_bias=t.reshape(-1).to(x.dtype))
        elif (not _DEBUG_ATEN_TEMB_ADD and x.dtype == torch.bfloat16 and x.is_cuda and t.shape[1] % 4 == 0
              and ops.conv2d_eligible(x.dtype, self.conv1.weight, self.conv1.stride, self.conv1.padding, self.conv1.dilation, self.conv1.groups)):
            h = self.conv1(h, extra_bias=t.to(x.dtype))           # batch > 1 (stacked micro-batches): one addend row per sample in conv1's epilogue (round 5)
        elif _DEBUG_ATEN_TEMB_ADD:
            h = self.conv1(h) + t[:, :, None, None]                # DEBUG ONLY (tools/stack_debug_graph.py, tools/oob_guard_probe.py): the round-5 form whose ATen reduction went bad under replay
        else:
            h = ops.add_sample_channel_bias(self.conv1(h), t)      # (its dt is summed per sample by this repo's column_sum, not by ATen's broadcast reduction)
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return self.conv2(self.norm2(h, act='silu'), residual=x)       # the block's "x + h" rides conv2's epilogue


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, cross_dim):
        super().__init__()
        self.norm1, self.attn1 = dnn.LayerNorm(dim), dnn.Attention(dim, None, heads, head_dim)
        self.norm2, self.attn2 = dnn.LayerNorm(dim), dnn.Attention(dim, cross_dim, heads, head_dim)
        self.norm3, self.ff = dnn.LayerNorm(dim), dnn.FeedForward(dim)

    def forward(self, x, encoder_hidden_states, kv=None):
        # the three "x + f(norm(x))" adds ride the epilogue of f's output projection (GEMM `residual`)
        h, x = self.norm1(x, with_skip=True)            # (and their gradients ride the LayerNorm backward kernels)
        x = self.attn1(h, residual=x)
        h, x = self.norm2(x, with_skip=True)
        x = self.attn2(h, encoder_hidden_states, residual=x, kv=kv)
        h, x = self.norm3(x, with_skip=True)
        return self.ff(h, residual=x)


BATCH_CONTEXT_KV = os.environ.get('DPIPE_BATCH_CONTEXT_KV', '1') == '1'     # A/B switch: per-block K / V projections of the text context


class Transformer2DModel(nn.Module):
    """use_linear_projection=True variant (SDXL)."""

    def __init__(self, heads, head_dim, in_ch, num_layers, cross_dim, groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = dnn.GroupNorm(groups, in_ch, eps=1e-6)
        self.proj_in = dnn.Linear(in_ch, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, head_dim, cross_dim) for _ in range(num_layers)])
        self.proj_out = dnn.Linear(inner, in_ch)

    def forward(self, x, encoder_hidden_states):
        B, C, H, W = x.shape
        x = x.contiguous(memory_format=torch.channels_last)          # [B, HW, C] memory: the token-major view below is free
        h, residual = self.norm(x, with_skip=True)
        residual = residual.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        ctx = encoder_hidden_states.to(h.dtype)
        kvs = [None] * len(self.transformer_blocks)
        if BATCH_CONTEXT_KV and len(self.transformer_blocks) > 1 and all(b.attn2.kv_batchable() for b in self.transformer_blocks):
            # every block projects the SAME 77 text tokens to its keys / values: one GEMM for all of them (forward, and one dgrad + one wgrad in the
            # backward) instead of one tiny M = 77 launch per block -- 70 -> 11 launches per pass over the UNet; the blocks consume column slices in place
            ws = [w for b in self.transformer_blocks for w in (b.attn2.to_k.weight, b.attn2.to_v.weight)]
            kvs = ops.split_columns(ops.fused_linear(ctx, ws, None), [2 * b.attn2.heads * b.attn2.dim_head for b in self.transformer_blocks])
        for blk, kv in zip(self.transformer_blocks, kvs):
            h = blk(h, ctx, kv)
        h = self.proj_out(h, residual)                               # "+ residual" in the projection's epilogue
        return h.reshape(B, H, W, C).permute(0, 3, 1, 2)


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = dnn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x.contiguous(memory_format=torch.channels_last))


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = dnn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x, output_size=None):
        x = x.contiguous(memory_format=torch.channels_last)
        if output_size is None or tuple(output_size) == (2 * x.shape[2], 2 * x.shape[3]):
            return self.conv(x, upsample=2)              # nearest 2x up-sampling folded into the convolution's gather
        x = F.interpolate(x, size=output_size, mode='nearest')
        return self.conv(x)


class _Block(nn.Module):
    """Container with the diffusers attribute names (resnets / attentions / downsamplers / upsamplers)."""

    def __init__(self, resnets, attentions=None, downsamplers=None, upsamplers=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if attentions is not None:
            self.attentions = nn.ModuleList(attentions)
        self.downsamplers = nn.ModuleList(downsamplers) if downsamplers else None
        self.upsamplers = nn.ModuleList(upsamplers) if upsamplers else None


class UNet2DConditionModel(nn.Module):
    def __init__(self, c: SDXLConfig):
        super().__init__()
        self.config = c
        ch, temb, g, head_dim = c.block_out_channels, c.time_embed_dim, c.norm_groups, 64
        self.conv_in = dnn.Conv2d(c.in_channels, ch[0], 3, padding=1)
        self.time_proj = dnn.Timesteps(ch[0], flip_sin_to_cos=True, downscale_freq_shift=0)
        self.time_embedding = dnn.TimestepEmbedding(ch[0], temb)
        self.add_time_proj = dnn.Timesteps(c.addition_time_embed_dim, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.add_embedding = dnn.TimestepEmbedding(c.add_embed_in, temb)

        def tr(i, channels):
            return Transformer2DModel(c.num_heads[i], head_dim, channels, c.transformer_layers[i], c.cross_attention_dim, g)

        down = []
        out = ch[0]
        for i, oc in enumerate(ch):
            inp, out = out, oc
            last = i == len(ch) - 1
            resnets = [ResnetBlock2D(inp if j == 0 else out, out, temb, g) for j in range(c.layers_per_block)]
            attns = [tr(i, out) for _ in range(c.layers_per_block)] if c.transformer_layers[i] > 0 else None
            down.append(_Block(resnets, attns, downsamplers=None if last else [Downsample2D(out)]))
        self.down_blocks = nn.ModuleList(down)

        mid_i = len(ch) - 1
        self.mid_block = _Block([ResnetBlock2D(ch[-1], ch[-1], temb, g), ResnetBlock2D(ch[-1], ch[-1], temb, g)], [tr(mid_i, ch[-1])])

        up = []
        rev = list(reversed(ch))
        out = rev[0]
        for i, oc in enumerate(rev):
            prev, out = out, oc
            inp = rev[min(i + 1, len(ch) - 1)]
            last = i == len(ch) - 1
            n = c.layers_per_block + 1
            resnets = []
            for j in range(n):
                skip = inp if j == n - 1 else out
                rin = prev if j == 0 else out
                resnets.append(ResnetBlock2D(rin + skip, out, temb, g))
            li = len(ch) - 1 - i
            attns = [tr(li, out) for _ in range(n)] if c.transformer_layers[li] > 0 else None
            up.append(_Block(resnets, attns, upsamplers=None if last else [Upsample2D(out)]))
        self.up_blocks = nn.ModuleList(up)
        self.num_upsamplers = len(ch) - 1

        self.conv_norm_out = dnn.GroupNorm(g, ch[0], eps=1e-5)
        self.conv_act = dnn.SiLU()
        self.conv_out = dnn.Conv2d(ch[0], c.in_channels, 3, padding=1)


# -------------------------------------------------------------------------------------- pipeline layer wrappers
# (Round 3 negative result, profiles/r3l_parallel_text_encoders_negative.jsonl: CLIP-L on a forked stream next to CLIP-G inside the lane's graph -- slower; the switch and
#  its side streams were removed in round 5.)


class InitialLayer(nn.Module):
    """Text conditioning (both CLIP encoders run inside stage 0 and are trained), time / added-condition embedding,
    conv_in (models/sdxl.py:654-784)."""

    def __init__(self, unet, text_encoder, text_encoder_2):
        super().__init__()
        self.clip_skip = None
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.time_proj, self.time_embedding = unet.time_proj, unet.time_embedding
        self.add_time_proj, self.add_embedding = unet.add_time_proj, unet.add_embedding
        self.conv_in = unet.conv_in
        self.num_upsamplers = unet.num_upsamplers
        self.max_len = text_encoder.config.max_pos

    def forward(self, inputs):
        for t in inputs:
            if torch.is_floating_point(t):
                t.requires_grad_(True)
        sample, timestep, input_ids, input_ids_2, add_time_ids = inputs
        factor = 2 ** self.num_upsamplers
        forward_upsample_size = torch.full((), any(d % factor != 0 for d in sample.shape[-2:]), dtype=torch.bool, device=sample.device)

        encoder_hidden_states, pooled = self.get_text_conditioning(input_ids, input_ids_2)
        wdtype = self.conv_in.weight.dtype
        # round 6: with plain bf16 embedding MLPs on the GPU the time / added-condition embedding stays fp32 from the sinusoidal features to the addend of every
        # ResnetBlock2D (ops.precise_row_linear); otherwise (fp32 model, adapters on these layers, DPIPE_PRECISE_ADDENDS=0) it takes the model dtype as before
        precise = sample.is_cuda and wdtype == torch.bfloat16 and self.time_embedding.precise_ok() and self.add_embedding.precise_ok()
        t_emb = self.time_proj(timestep.expand(sample.shape[0]))
        time_embeds = self.add_time_proj(add_time_ids.flatten()).reshape(sample.shape[0], -1)
        add_embeds = torch.cat([pooled.float(), time_embeds], dim=-1)
        if not precise:
            t_emb, add_embeds = t_emb.to(wdtype), add_embeds.to(wdtype)
        emb = self.time_embedding(t_emb) + self.add_embedding(add_embeds)
        sample = self.conv_in(sample.to(wdtype))
        return make_contiguous(sample, timestep, emb, encoder_hidden_states, sample, forward_upsample_size)

    def get_text_conditioning(self, input_ids, input_ids_2):
        e1, _ = self.get_prompt_embeds(input_ids, self.text_encoder, False)
        e2, pooled = self.get_prompt_embeds(input_ids_2, self.text_encoder_2, True)
        return torch.cat([e1, e2], dim=-1), pooled

    def get_prompt_embeds(self, input_ids, text_encoder, want_pooled):
        """Arbitrary-length ids in chunks of max_len - 2, BOS prepended, first pad replaced by EOS (models/sdxl.py:745-784)."""
        c = text_encoder.config
        bs, device = input_ids.shape[0], input_ids.device
        embeds, pooled = [], None
        for i, chunk in enumerate(torch.split(input_ids, self.max_len - 2, dim=-1)):
            chunk = torch.cat([torch.full((bs, 1), c.bos, device=device), chunk, torch.full((bs, 1), c.pad, device=device)], dim=-1)
            first_pad = torch.argmax((chunk == c.pad).to(torch.int32), dim=-1)
            chunk.scatter_(1, first_pad[:, None], c.eos)        # chunk[b, first_pad[b]] = EOS without a host-side scalar copy
            hidden, p = text_encoder(chunk, want_pooled=want_pooled and i == 0)
            if i == 0:
                pooled = p
            embeds.append(hidden)
        return torch.cat(embeds, dim=1), pooled


def _unpack(inputs):
    hidden, timesteps, emb, ctx, *skips, fus = inputs
    return hidden, timesteps, emb, ctx, tuple(skips), fus


class DownBlockInnerLayer(nn.Module):
    def __init__(self, resnet, attn):
        super().__init__()
        self.resnet, self.attn = resnet, attn

    def forward(self, inputs):
        h, ts, emb, ctx, skips, fus = _unpack(inputs)
        h = self.resnet(h, emb)
        if self.attn is not None:
            h = self.attn(h, ctx)
        return make_contiguous(h, ts, emb, ctx, *skips, h, fus)


class DownsamplerLayer(nn.Module):
    def __init__(self, downsamplers):
        super().__init__()
        self.downsamplers = downsamplers

    def forward(self, inputs):
        h, ts, emb, ctx, skips, fus = _unpack(inputs)
        for d in self.downsamplers:
            h = d(h)
        return make_contiguous(h, ts, emb, ctx, *skips, h, fus)


class MidBlockInnerLayer(nn.Module):
    def __init__(self, resnet, attn):
        super().__init__()
        self.resnet, self.attn = resnet, attn

    def forward(self, inputs):
        h, ts, emb, ctx, skips, fus = _unpack(inputs)
        # models/sdxl.py:817-823 applies resnet then attn for every mid layer (layer 0 has attn=None)
        h = self.resnet(h, emb)
        if self.attn is not None:
            h = self.attn(h, ctx)
        return make_contiguous(h, ts, emb, ctx, *skips, fus)


class UpBlockInnerLayer(nn.Module):
    def __init__(self, resnet, attn):
        super().__init__()
        self.resnet, self.attn = resnet, attn

    def forward(self, inputs):
        h, ts, emb, ctx, skips, fus = _unpack(inputs)
        h = torch.cat([h, skips[-1]], dim=1)
        h = self.resnet(h, emb)
        if self.attn is not None:
            h = self.attn(h, ctx)
        return make_contiguous(h, ts, emb, ctx, *skips[:-1], fus)


class UpsamplerLayer(nn.Module):
    def __init__(self, upsamplers, is_final_block):
        super().__init__()
        self.upsamplers, self.is_final_block = upsamplers, is_final_block

    def forward(self, inputs):
        h, ts, emb, ctx, skips, fus = _unpack(inputs)
        # The reference reads the `forward_upsample_size` flag on the host (one sync per call) and only then passes
        # the explicit size; nearest-neighbour resize to the skip tensor's size is the same map as scale_factor=2
        # whenever the flag is false, so the size is always passed and the GPU->CPU sync disappears.
        size = skips[-1].shape[2:] if (not self.is_final_block and len(skips) > 0) else None
        for u in self.upsamplers:
            h = u(h, size)
        return make_contiguous(h, ts, emb, ctx, *skips, fus)


class FinalLayer(nn.Module):
    def __init__(self, unet):
        super().__init__()
        self.conv_norm_out, self.conv_act, self.conv_out = unet.conv_norm_out, unet.conv_act, unet.conv_out

    def forward(self, inputs):
        h, ts, emb, ctx, skips, fus = _unpack(inputs)
        return self.conv_out(self.conv_norm_out(h, act='silu')), ts


# -------------------------------------------------------------------------------------------------- the adapter
class SDXLWorkload:
    """Adapter-API subset the engine needs over a randomly initialised SDXL."""
    name = 'sdxl'
    checkpointable_layers = ['InitialLayer', 'DownBlockInnerLayer', 'DownsamplerLayer', 'MidBlockInnerLayer', 'UpBlockInnerLayer',
                             'UpsamplerLayer', 'FinalLayer']

    def __init__(self, config: Optional[SDXLConfig] = None, model_config=None, dtype=torch.bfloat16, seed=0, device='cpu'):
        self.cfg = config or SDXLConfig()
        self.model_config = model_config or {}
        self.train_config = {}            # the run's TOML dict (`config['optimizer']['lr']` is the base LR of get_param_groups)
        self.v_pred = self.model_config.get('v_pred', False)
        self.min_snr_gamma = self.model_config.get('min_snr_gamma', None)
        self.debiased_estimation_loss = self.model_config.get('debiased_estimation_loss', None)
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        with torch.device(device):
            self.unet = UNet2DConditionModel(self.cfg)
            self.text_encoder = CLIPTextModel(self.cfg.te1)
            self.text_encoder_2 = CLIPTextModel(self.cfg.te2)
        torch.random.set_rng_state(gen_state)
        for prefix, mod in (('unet', self.unet), ('text_encoder', self.text_encoder), ('text_encoder_2', self.text_encoder_2)):
            mod.to(dtype)
            for n, p in mod.named_parameters():
                p.original_name = f'{prefix}.{n}'
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, self.cfg.num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        alpha, sigma = self.alphas_cumprod.sqrt(), (1.0 - self.alphas_cumprod).sqrt()
        self.all_snr = (alpha / sigma) ** 2

    def modules(self):
        return {'unet': self.unet, 'text_encoder': self.text_encoder, 'text_encoder_2': self.text_encoder_2}

    def configure_adapter(self, adapter_config):
        """LoRA on every Linear of the UNet's down / mid / up blocks and of both text encoders (models/sdxl.py:431-459);
        everything else is frozen, adapter tensors take `adapter_config['dtype']`.  -> {component: wrapped layer names}."""
        if adapter_config.get('type', 'lora') != 'lora':
            raise NotImplementedError(f"Adapter type {adapter_config['type']} is not implemented")
        kw = dict(rank=adapter_config['rank'], alpha=adapter_config['alpha'], dropout=adapter_config.get('dropout', 0.0),
                  dtype=adapter_config.get('dtype'))
        in_blocks = lambda name, module: name.split('.')[0] in ('down_blocks', 'mid_block', 'up_blocks')
        wrapped = {}
        for prefix, top, target in (('unet', self.unet, in_blocks), ('text_encoder', self.text_encoder, None),
                                    ('text_encoder_2', self.text_encoder_2, None)):
            wrapped[prefix] = dnn.apply_lora(top, target=target, **kw)
            for n, p in top.named_parameters():
                p.original_name = f'{prefix}.{n}'
        return wrapped

    # ---- saved files (models/sdxl.py:465-525)
    def save_adapter(self, save_dir, peft_state_dict):
        """LoRA -> kohya-format lora.safetensors."""
        from ..formats import save_sdxl_kohya_lora
        save_sdxl_kohya_lora(save_dir, peft_state_dict)

    def set_vae_state_dict(self, vae_state_dict):
        """The base checkpoint's VAE weights (diffusers AutoencoderKL names): not part of the training step, but part of every single-file checkpoint
        the reference writes (models/sdxl.py:503-522 always converts and embeds self.vae.state_dict())."""
        self.vae_state_dict = dict(vae_state_dict)

    def _vae_for_save(self):
        """-> (diffusers-named VAE state dict or None, already-LDM-named `first_stage_model.*` tensors or None).  Sources, in order: set_vae_state_dict();
        the `first_stage_model.*` tensors of the single-file checkpoint the run was configured with (`[model] checkpoint_path`, models/sdxl.py:386),
        read lazily at the first save."""
        if getattr(self, 'vae_state_dict', None) is not None:
            return self.vae_state_dict, None
        path = self.model_config.get('checkpoint_path')
        if path and os.path.isfile(path) and str(path).endswith('.safetensors'):
            from safetensors import safe_open
            with safe_open(path, framework='pt', device='cpu') as f:
                ldm = {k: f.get_tensor(k) for k in f.keys() if k.startswith('first_stage_model.')}
            if ldm:
                return None, ldm
        return None, None

    _NO_VAE = ('SDXL save_model: no VAE weights to embed -- the reference always writes first_stage_model.* into model.safetensors.  Give the workload the '
               "base checkpoint's VAE (set_vae_state_dict(...) or [model] checkpoint_path = a single-file .safetensors), or pass allow_missing_vae=True "
               'to write a UNet + text-encoder-only file on purpose.')

    def check_save_sources(self, is_adapter=False):
        """Called by saver.Saver at construction: a full fine-tune that could not write its single-file checkpoint fails before the first step, not at the
        first save.  Honours model_config['allow_missing_vae']."""
        if is_adapter or self.model_config.get('allow_missing_vae', False):
            return
        vae, ldm = self._vae_for_save()
        if vae is None and ldm is None:
            raise RuntimeError(self._NO_VAE)

    def save_model(self, save_dir, diffusers_sd, vae_state_dict=None, allow_missing_vae=False):
        """Full fine-tune -> single-file SDXL checkpoint: UNet, both text encoders AND the VAE, like the reference's (models/sdxl.py:487-525; ComfyUI, Forge and
        diffusers' from_single_file expect the complete ldm file).  The saver passes two arguments (utils/saver.py:106), so the VAE comes from the workload
        (_vae_for_save); a checkpoint without `first_stage_model.*` keys is only written when the caller asks for it explicitly."""
        from ..formats import save_sdxl_ldm
        ldm_vae = None
        if vae_state_dict is None:
            vae_state_dict, ldm_vae = self._vae_for_save()
        if vae_state_dict is None and ldm_vae is None and not (allow_missing_vae or self.model_config.get('allow_missing_vae', False)):
            raise RuntimeError(self._NO_VAE)
        save_sdxl_ldm(save_dir, diffusers_sd, vae_state_dict, ldm_vae)

    def to_layers(self):
        unet = self.unet
        layers = [InitialLayer(unet, self.text_encoder, self.text_encoder_2)]
        for block in unet.down_blocks:
            attns = getattr(block, 'attentions', [None] * len(block.resnets))
            layers += [DownBlockInnerLayer(r, a) for r, a in zip(block.resnets, attns)]
            if block.downsamplers is not None:
                layers.append(DownsamplerLayer(block.downsamplers))
        mid = unet.mid_block
        layers.append(MidBlockInnerLayer(mid.resnets[0], None))
        layers += [MidBlockInnerLayer(r, a) for a, r in zip(mid.attentions, mid.resnets[1:])]
        for i, block in enumerate(unet.up_blocks):
            attns = getattr(block, 'attentions', [None] * len(block.resnets))
            layers += [UpBlockInnerLayer(r, a) for r, a in zip(block.resnets, attns)]
            if block.upsamplers is not None:
                layers.append(UpsamplerLayer(block.upsamplers, i == len(unet.up_blocks) - 1))
        layers.append(FinalLayer(unet))
        return layers

    def prepare_inputs(self, inputs, timestep_quantile=None):
        latents = inputs['latents'].float()
        mask = inputs.get('mask')
        input_ids, input_ids_2 = inputs['input_ids'], inputs['input_ids_2']
        bs, _, h, w = latents.shape
        if mask is not None:
            mask = F.interpolate(mask.unsqueeze(1), size=(h, w), mode='nearest-exact')
        noise = torch.randn_like(latents)
        T = self.cfg.num_train_timesteps
        if timestep_quantile is not None:
            timesteps = torch.full((bs,), int(timestep_quantile * T))
        else:
            timesteps = torch.randint(0, T, (bs,))
        a = self.alphas_cumprod[timesteps].view(-1, 1, 1, 1)
        noisy = a.sqrt() * latents + (1 - a).sqrt() * noise
        target = (a.sqrt() * noise - (1 - a).sqrt() * latents) if self.v_pred else noise
        ph, pw = h * self.cfg.vae_scale_factor, w * self.cfg.vae_scale_factor
        add_time_ids = torch.tensor([[ph, pw, 0, 0, ph, pw]], dtype=torch.float32).expand(bs, -1)
        return (noisy, timesteps, input_ids, input_ids_2, add_time_ids), (target, mask)

    def snr_row_weights(self, timesteps):
        """Per-sample loss weights of models/sdxl.py:333-355; None when neither option is set."""
        if self.min_snr_gamma is None and self.debiased_estimation_loss is None:
            return None
        if self.all_snr.device != timesteps.device:
            self.all_snr = self.all_snr.to(timesteps.device)        # one-time move (keeps graph capture free of H2D copies)
        snr = self.all_snr[timesteps]
        w = torch.ones_like(snr)
        if self.min_snr_gamma is not None:
            m = torch.minimum(snr, torch.full_like(snr, self.min_snr_gamma))
            w = w * (m / (snr + 1) if self.v_pred else m / snr)
        if self.debiased_estimation_loss is not None:
            s = torch.minimum(snr, torch.ones_like(snr) * 1000)
            w = w * (1 / (s + 1) if self.v_pred else 1 / torch.sqrt(s))
        return w.float()

    def get_loss_fn(self):
        def loss_fn(output, label):
            output, timesteps = output
            target, mask = label
            return ops.fused_loss(output, target, mask if mask.numel() > 0 else None, self.snr_row_weights(timesteps), per_sample=True)
        return loss_fn

    def get_param_groups(self, parameters):
        groups = {'unet.': [], 'text_encoder.': [], 'text_encoder_2.': []}
        for p in parameters:
            for prefix in groups:
                if p.original_name.startswith(prefix):
                    groups[prefix].append(p)
                    break
            else:
                raise RuntimeError(f'Unexpected parameter: {p.original_name}')
        base_lr = self.train_config.get('optimizer', {}).get('lr', self.model_config.get('lr'))
        out = []
        for prefix, key in (('unet.', 'unet_lr'), ('text_encoder.', 'text_encoder_1_lr'), ('text_encoder_2.', 'text_encoder_2_lr')):
            g = {'params': groups[prefix]}
            lr = self.model_config.get(key, base_lr)
            if lr is not None:
                g['lr'] = lr
            out.append(g)
        return out


def synthetic_batch(cfg: SDXLConfig, batch_size=1, latent_hw=128, seed=0, ids_len=75):
    """SURVEY 8(d) inputs: latents randn[B,4,h,w], ids uniform in [1000, 40000) (no pad tokens), mask None."""
    g = torch.Generator().manual_seed(seed)
    hi = min(40000, cfg.te1.vocab - 3)
    lo = min(1000, hi // 2)
    return {'latents': torch.randn(batch_size, cfg.in_channels, latent_hw, latent_hw, generator=g),
            'input_ids': torch.randint(lo, hi, (batch_size, ids_len), generator=g),
            'input_ids_2': torch.randint(lo, hi, (batch_size, ids_len), generator=g), 'mask': None}


# ------------------------------------------------------------------------------------------ algorithmic FLOP model
def layer_forward_flops(cfg: SDXLConfig, latent_hw=128, batch=1, text_tokens=77):
    """Algorithmic forward FLOPs (2 x MACs of every conv / linear + 4*Sq*Sk*C per attention core) of each of the
    pipeline layers returned by SDXLWorkload.to_layers(), in order.  Used for roofline / MFU reporting
    (the surveyor's 6.76 TFLOP forward is re-derived here from the config)."""
    ch, temb = cfg.block_out_channels, cfg.time_embed_dim

    def conv(cin, cout, k, hw):
        return 2.0 * cin * cout * k * k * hw * hw * batch

    def lin(i, o, tokens):
        return 2.0 * i * o * tokens * batch

    def resnet(cin, cout, hw):
        f = conv(cin, cout, 3, hw) + conv(cout, cout, 3, hw) + lin(temb, cout, 1)
        return f + (conv(cin, cout, 1, hw) if cin != cout else 0.0)

    def transformer(c, depth, hw):
        s = hw * hw
        f = 2 * lin(c, c, s)
        per = 4 * lin(c, c, s) + 4.0 * s * s * c * batch                                    # self-attention
        per += 2 * lin(c, c, s) + 2 * lin(cfg.cross_attention_dim, c, text_tokens) + 4.0 * s * text_tokens * c * batch   # cross
        per += lin(c, 8 * c, s) + lin(4 * c, c, s)                                           # GEGLU feed-forward
        return f + depth * per

    def clip(c: CLIPConfig):
        s = text_tokens
        per = 4 * lin(c.hidden, c.hidden, s) + 4.0 * s * s * c.hidden * batch + 2 * lin(c.hidden, c.mlp, s)
        return c.layers * per + (lin(c.hidden, c.proj_dim, 1) if c.proj_dim else 0.0)

    flops = []
    hw = latent_hw
    flops.append(clip(cfg.te1) + clip(cfg.te2) + lin(ch[0], temb, 1) + lin(temb, temb, 1) + lin(cfg.add_embed_in, temb, 1) + lin(temb, temb, 1)
                 + conv(cfg.in_channels, ch[0], 3, hw))
    out = ch[0]
    for i, oc in enumerate(ch):
        inp, out = out, oc
        for j in range(cfg.layers_per_block):
            f = resnet(inp if j == 0 else out, out, hw)
            if cfg.transformer_layers[i] > 0:
                f += transformer(out, cfg.transformer_layers[i], hw)
            flops.append(f)
        if i != len(ch) - 1:
            hw //= 2
            flops.append(conv(out, out, 3, hw))
    mi = len(ch) - 1
    flops.append(resnet(ch[-1], ch[-1], hw))
    flops.append(resnet(ch[-1], ch[-1], hw) + transformer(ch[-1], cfg.transformer_layers[mi], hw))
    rev = list(reversed(ch))
    out = rev[0]
    for i, oc in enumerate(rev):
        prev, out = out, oc
        inp = rev[min(i + 1, len(ch) - 1)]
        n = cfg.layers_per_block + 1
        li = len(ch) - 1 - i
        for j in range(n):
            skip = inp if j == n - 1 else out
            f = resnet((prev if j == 0 else out) + skip, out, hw)
            if cfg.transformer_layers[li] > 0:
                f += transformer(out, cfg.transformer_layers[li], hw)
            flops.append(f)
        if i != len(ch) - 1:
            hw *= 2
            flops.append(conv(out, out, 3, hw))
    flops.append(conv(ch[0], cfg.in_channels, 3, hw))
    return flops


def train_step_flops(cfg: SDXLConfig, latent_hw=128, batch=1, text_tokens=77):
    """Full fine-tune: backward = 2 x forward for trained weights (activation-checkpoint recompute not counted)."""
    return 3.0 * sum(layer_forward_flops(cfg, latent_hw, batch, text_tokens))
