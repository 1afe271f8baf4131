"""nn.Module building blocks whose arithmetic runs in the HIP kernels (ops.py).  Parameter names follow the
libraries the reference adapters wrap (diffusers / HF transformers), so state dicts line up with theirs.

Compute dtype = parameter dtype (bf16 for training, fp32 for the exact-parity mode): there is no autocast layer;
inputs are cast to the weight dtype at the first GEMM of a block, norms keep fp32 statistics.
Convolutions (implicit GEMM on the MFMA tile kernel, channels-last) and GroupNorm are kernels of this repo as well, in both modes.
"""
import math

import torch
from torch import nn

from . import ops

import os as _os
_NORM_SKIP = _os.environ.get('DPIPE_NORM_SKIP', '1') == '1'      # A/B switch: fold the bypass branch's gradient into the norm backward kernels


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, residual=None):
        return ops.linear(x, self.weight, self.bias, residual)

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'


class LoRALinear(nn.Module):
    """LoRA-wrapped Linear, the restatement of peft's `lora.Linear` the reference trains adapters with
    (models/base.py:272-297, models/sdxl.py:431-459; peft is an un-vendored dependency, requirements.txt:9):

        y = base_layer(x) + lora_B(lora_A(dropout(x))) * (alpha / r)

    Module / parameter names follow peft (`base_layer`, `lora_A.<adapter>`, `lora_B.<adapter>`), so a peft state dict
    loads unchanged.  lora_A uses nn.Linear's default init, lora_B starts at zero.  The base weight is frozen, so its wgrad
    GEMM disappears from the backward pass; the up-projection adds into the base output inside its GEMM epilogue."""

    def __init__(self, base_layer, rank, alpha, dropout=0.0, dtype=None, adapter_name='default'):
        super().__init__()
        dev = base_layer.weight.device
        dtype = dtype or base_layer.weight.dtype
        self.base_layer = base_layer
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.r, self.lora_alpha, self.scaling, self.adapter_name = rank, alpha, alpha / rank, adapter_name
        self.dropout_p = float(dropout)
        self.lora_A = nn.ModuleDict({adapter_name: Linear(self.in_features, rank, bias=False, device=dev, dtype=dtype)})
        self.lora_B = nn.ModuleDict({adapter_name: Linear(rank, self.out_features, bias=False, device=dev, dtype=dtype)})
        nn.init.zeros_(self.lora_B[adapter_name].weight)
        for p in base_layer.parameters():
            p.requires_grad_(False)

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def forward(self, x, residual=None):
        y = self.base_layer(x, residual)
        A, B = self.lora_A[self.adapter_name], self.lora_B[self.adapter_name]
        h = x
        if self.dropout_p > 0.0 and self.training:
            h = torch.nn.functional.dropout(h, self.dropout_p)
        h = A(h)
        if self.scaling != 1.0:
            h = h * self.scaling
        if B.weight.dtype == y.dtype:
            return B(h, y)                    # y + h B^T: the add rides the up-projection's epilogue
        return y + B(h).to(y.dtype)

    def extra_repr(self):
        return f'r={self.r}, alpha={self.lora_alpha}, dropout={self.dropout_p}'


def apply_lora(root, rank, alpha, dropout=0.0, dtype=None, target=None, adapter_name='default'):
    """Wrap Linear layers under `root` with LoRALinear and freeze everything else (what `peft.get_peft_model` /
    diffusers' `add_adapter` do for the reference).  `target(name, module) -> bool` restricts the wrapped layers (the
    reference targets every nn.Linear inside the adapter's `adapter_target_modules` classes, models/base.py:262-270).
    Returns the names of the wrapped layers."""
    sites = []
    for name, module in root.named_modules():
        for child_name, child in module.named_children():
            full = f'{name}.{child_name}' if name else child_name
            if type(child) is Linear and (target is None or target(full, child)):
                sites.append((module, child_name, child, full))
    for p in root.parameters():
        p.requires_grad_(False)
    for parent, child_name, child, _ in sites:
        parent._modules[child_name] = LoRALinear(child, rank, alpha, dropout, dtype, adapter_name)
    return [full for *_, full in sites]


def lora_state_dict(root, adapter_name='default'):
    """The adapter's tensors keyed like peft's `get_peft_model_state_dict` (adapter name stripped)."""
    tag = f'.{adapter_name}.'
    return {k.replace(tag, '.'): v for k, v in root.state_dict().items() if '.lora_A.' in k or '.lora_B.' in k}


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5, elementwise_affine=True, bias=True, device=None, dtype=None):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype)) if elementwise_affine else None
        self.bias = nn.Parameter(torch.zeros(dim, device=device, dtype=dtype)) if (elementwise_affine and bias) else None

    def forward(self, x, scale=None, shift=None, with_skip=False):
        """Optionally fused AdaLN modulation: LN(x) * (1 + scale) + shift.  with_skip -> (y, x'): hand x' to the residual branch
        around this norm (pre-norm blocks), so that branch's gradient is added inside the LayerNorm backward kernel."""
        if with_skip and not _NORM_SKIP:
            return ops.layer_norm_modulate(x, self.weight, self.bias, scale, shift, self.eps), x
        return ops.layer_norm_modulate(x, self.weight, self.bias, scale, shift, self.eps, with_skip)


class GroupNorm(nn.Module):
    """nn.GroupNorm (same parameter names) on the HIP kernels, optionally fused with the SiLU that follows it."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels, device=device, dtype=dtype)) if affine else None
        self.bias = nn.Parameter(torch.zeros(num_channels, device=device, dtype=dtype)) if affine else None

    def forward(self, x, act=None, with_skip=False):
        nhwc = x.dim() == 4 and not x.is_contiguous() and ops.is_channels_last(x)      # channels-last UNet (csrc/groupnorm_nhwc.hip)
        fn = ops.group_norm_nhwc if nhwc else ops.group_norm
        if with_skip and not _NORM_SKIP:
            return fn(x, self.num_groups, self.weight, self.bias, self.eps, act), x
        return fn(x, self.num_groups, self.weight, self.bias, self.eps, act, with_skip)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameter names / logical shapes, so checkpoints map 1:1) whose weight is STORED channels-last and whose bf16
    forward / backward run as implicit GEMM on the MFMA tile kernel (csrc/conv_pipe.hip) over channels-last activations.  Returns a
    channels-last [B, Cout, Ho, Wo] tensor.  `upsample=2` folds diffusers' nearest 2x Upsample2D into the convolution's gather;
    `residual` / `extra_bias` ride the epilogue.  fp32 tensors (exact-parity mode) take the same kernels as three bf16 hi / lo split launches
    accumulated in fp32 (ops._Conv2dNHWCFn); the UNet's 4-channel conv_in / conv_out run as a tiny im2col GEMM / a Cout-padded tile.  Grouped /
    dilated / non-square-padded convolutions (none of which the reference's models have on this path) raise: there is no library fallback."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self.weight.dim() == 4 and not self.weight.permute(0, 2, 3, 1).is_contiguous():
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        return out

    def forward(self, x, upsample=1, residual=None, extra_bias=None):
        bias = self.bias
        if extra_bias is not None and extra_bias.dim() == 3:
            # the bf16 hi / lo pair of an fp32 addend that already CONTAINS this convolution's bias (ops.precise_row_linear(..., extra=self.bias, pair=True))
            if not (x.is_cuda and x.dtype == torch.bfloat16 and ops.conv2d_eligible(x.dtype, self.weight, self.stride, self.padding, self.dilation, self.groups)):
                from .hip import DpipeHipError
                raise DpipeHipError('Conv2d: a hi / lo bias pair needs the bf16 implicit-GEMM path')
            return ops.conv2d_nhwc(x, self.weight, extra_bias, self.stride[0], self.padding[0], upsample, residual)
        if extra_bias is not None and extra_bias.dim() == 2:
            # one addend row per sample ([B, Cout]: the time embedding of a resnet at batch > 1, i.e. under micro-batch stacking): a per-sample bias in the epilogue
            bias = ops.bias_plus_sample(bias, extra_bias)
        elif extra_bias is not None:                    # per-channel addend folded into the bias vector (time embedding of a resnet, batch 1)
            bias = extra_bias if bias is None else bias + extra_bias
        plain = self.padding_mode == 'zeros' and not isinstance(self.padding, str)
        if x.is_cuda and plain and ops.conv2d_eligible(x.dtype, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return ops.conv2d_nhwc(x, self.weight, bias, self.stride[0], self.padding[0], upsample, residual)
        Cout, Cin, kh, kw = self.weight.shape
        ours = x.is_cuda and plain and x.dtype == self.weight.dtype and x.dtype in (torch.bfloat16, torch.float32) and self.groups == 1 \
            and tuple(self.dilation) == (1, 1) and self.stride[0] == self.stride[1] and self.stride[0] in (1, 2) and self.padding[0] == self.padding[1]
        if ours and Cin % 64 == 0:
            # few output channels (the UNet's conv_out: 320 -> 4): zero-pad Cout to one 64-wide tile, run the implicit-GEMM kernels, slice
            pad = -Cout % 64
            w = torch.nn.functional.pad(self.weight, (0, 0, 0, 0, 0, 0, 0, pad)).contiguous(memory_format=torch.channels_last)
            b = torch.nn.functional.pad(bias, (0, pad)) if bias is not None else None
            y = ops.conv2d_nhwc(x, w, b, self.stride[0], self.padding[0], upsample, None)[:, :Cout].contiguous(memory_format=torch.channels_last)
            return y if residual is None else y + residual
        if ours and upsample == 1 and Cin * kh * kw <= 512:
            # few input channels (the UNet's conv_in: 4 -> 320): the im2col matrix is tiny ([pixels, Cin kh kw] padded to one or two 64-wide
            # K-steps), so the convolution is one MFMA GEMM over it; unfold / pad and their adjoints are ATen kernels on a few hundred KB
            B, _, H, W = x.shape
            col = torch.nn.functional.unfold(x.contiguous(), (kh, kw), padding=self.padding, stride=self.stride)        # [B, Cin kh kw, L]
            K = Cin * kh * kw
            col = torch.nn.functional.pad(col.transpose(1, 2), (0, -K % 64))                                            # [B, L, K64]
            w2 = torch.nn.functional.pad(self.weight.reshape(Cout, K), (0, -K % 64))
            Ho = (H + 2 * self.padding[0] - kh) // self.stride[0] + 1
            y = ops.linear(col, w2, bias).view(B, Ho, -1, Cout).permute(0, 3, 1, 2)                                     # channels-last [B, Cout, Ho, Wo]
            return y if residual is None else y + residual
        # no library fallback on the product path: a convolution none of this repo's kernels takes (grouped, dilated, non-square padding / stride,
        # a CPU tensor) is an error, never a silent MIOpen call (the reference's models have none of these on the hot path)
        from .hip import DpipeHipError
        raise DpipeHipError(f'Conv2d({Cin}, {Cout}, kernel {kh}x{kw}, stride {tuple(self.stride)}, padding {self.padding}, dilation {tuple(self.dilation)}, '
                            f'groups {self.groups}, {x.dtype} on {x.device}): no HIP kernel of libdpipe_hip takes this convolution')


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6, elementwise_affine=True, device=None, dtype=None):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype)) if elementwise_affine else None

    def forward(self, x):
        return ops.rms_norm(x, self.weight, self.eps)


class SiLU(nn.Module):
    def forward(self, x):
        return ops.silu(x)


class GELU(nn.Module):
    def __init__(self, approximate='none'):
        super().__init__()
        self.approximate = approximate

    def forward(self, x):
        return ops.gelu_tanh(x) if self.approximate == 'tanh' else ops.gelu(x)


class QuickGELU(nn.Module):
    def forward(self, x):
        return ops.quick_gelu(x)


class Identity(nn.Module):
    def forward(self, x):
        return x


class GEGLU(nn.Module):
    """diffusers.models.activations.GEGLU: proj to 2*inner, hidden * gelu(gate)."""

    def __init__(self, dim_in, dim_out, device=None, dtype=None):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2, device=device, dtype=dtype)

    def forward(self, x):
        return ops.geglu(self.proj(x))


class FeedForward(nn.Module):
    """diffusers FeedForward(activation_fn='geglu'): net = [GEGLU, Dropout, Linear]."""

    def __init__(self, dim, mult=4, device=None, dtype=None):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner, device=device, dtype=dtype), Identity(), Linear(inner, dim, device=device, dtype=dtype)])

    def forward(self, x, residual=None):
        g, out = self.net[0], self.net[2]
        if type(g) is GEGLU and type(out) is Linear and type(self.net[1]) is Identity:
            # GEGLU's multiply + the output Linear as one autograd node: the GEGLU backward rides the Linear's dgrad epilogue (ops.geglu_linear, round 6)
            return ops.geglu_linear(g.proj(x), out.weight, out.bias, residual)
        x = g(x)
        return out(x, residual)                 # net[1] is the (identity) dropout slot; the block's residual add rides the epilogue


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention (self- or cross-attention, no qk-norm, no mask)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True, device=None, dtype=None):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        kw = dict(device=device, dtype=dtype)
        self.to_q = Linear(query_dim, inner, bias=bias, **kw)
        self.to_k = Linear(kv_dim, inner, bias=bias, **kw)
        self.to_v = Linear(kv_dim, inner, bias=bias, **kw)
        self.to_out = nn.ModuleList([Linear(inner, query_dim, bias=out_bias, **kw), Identity()])
        self.attn_impl = 'auto'
        self.fuse_projections = True      # fused QKV / KV GEMMs + packed flash attention in bf16 (fp32 parity mode: separate)

    def kv_batchable(self):
        """True when this (cross-) attention's K / V projections may be computed outside, batched with other blocks' (Transformer2DModel)."""
        return (type(self.to_k) is Linear and type(self.to_v) is Linear and self.to_k.bias is None and self.to_v.bias is None and self.fuse_projections
                and self.attn_impl == 'auto' and ops.flash_eligible(self.to_q.weight.dtype, self.dim_head))

    def forward(self, hidden_states, encoder_hidden_states=None, residual=None, kv=None):
        """kv: this block's packed [B, Sk, 2 H D] key / value projection of the context, computed by the caller for several blocks at once."""
        B, S, _ = hidden_states.shape
        H, D = self.heads, self.dim_head
        if kv is not None:
            return self.to_out[0](ops.attention_packed(self.to_q(hidden_states), kv, H, D), residual)
        plain = type(self.to_q) is Linear and type(self.to_k) is Linear and type(self.to_v) is Linear     # no adapter wrapped around them
        fused = plain and self.fuse_projections and self.attn_impl == 'auto' and ops.flash_eligible(self.to_q.weight.dtype, D)
        if fused and encoder_hidden_states is None:         # self attention: one QKV GEMM, packed attention
            qkv = ops.fused_linear(hidden_states, [self.to_q.weight, self.to_k.weight, self.to_v.weight],
                                   [self.to_q.bias, self.to_k.bias, self.to_v.bias])
            o = ops.attention_packed(qkv, None, H, D)
            return self.to_out[0](o, residual)
        if fused:                                           # cross attention: Q GEMM + one KV GEMM on the context
            q = self.to_q(hidden_states)
            kv = ops.fused_linear(encoder_hidden_states, [self.to_k.weight, self.to_v.weight], [self.to_k.bias, self.to_v.bias])
            o = ops.attention_packed(q, kv, H, D)
            return self.to_out[0](o, residual)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.to_q(hidden_states).view(B, S, H, D)
        k = self.to_k(ctx).view(B, ctx.shape[1], H, D)
        v = self.to_v(ctx).view(B, ctx.shape[1], H, D)
        o = ops.attention(q, k, v, impl=self.attn_impl)
        return self.to_out[0](o.reshape(B, S, H * D), residual)


class Timesteps(nn.Module):
    """diffusers Timesteps / get_timestep_embedding: fp32 sinusoidal features, [cos | sin] when flip_sin_to_cos."""

    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0, scale=1.0):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos = num_channels, flip_sin_to_cos
        self.downscale_freq_shift, self.scale = downscale_freq_shift, scale

    def forward(self, timesteps):
        return ops.sinusoidal_embedding(timesteps, self.num_channels, 10000.0, sin_first=not self.flip_sin_to_cos,
                                        downscale_shift=self.downscale_freq_shift, scale=self.scale)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, device=None, dtype=None):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim, device=device, dtype=dtype)
        self.act = SiLU()
        self.linear_2 = Linear(time_embed_dim, time_embed_dim, device=device, dtype=dtype)

    def precise_ok(self):
        """both projections are plain bf16 Linear layers (no adapter wrapped around them): fp32 rows may take ops.precise_row_linear"""
        return (ops.PRECISE_ADDENDS and type(self.linear_1) is Linear and type(self.linear_2) is Linear and type(self.act) is SiLU
                and self.linear_1.weight.dtype == torch.bfloat16 and self.linear_1.weight.is_cuda)

    def forward(self, sample, condition=None):
        if sample.dtype == torch.float32 and self.precise_ok():
            # fp32 rows in, fp32 embedding out (round 6): what leaves here is added to every pixel of a channel by each ResnetBlock2D -- a coherent error if rounded to bf16
            h = ops.precise_row_linear(sample, self.linear_1.weight, self.linear_1.bias)
            return ops.precise_row_linear(h, self.linear_2.weight, self.linear_2.bias, act='silu')
        return self.linear_2(self.act(self.linear_1(sample)))
