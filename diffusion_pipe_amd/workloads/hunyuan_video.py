"""HunyuanVideo t2v (BASELINE config 5: full fine-tune, pp = 8, activation offload) on the MI355X kernels.

What follows the reference (models/hunyuan_video.py): `prepare_inputs` (:413-481: logit-normal / uniform t with sigmoid_scale and shift,
x_t = (1 - t) x1 + t x0, target = x0 - x1, mask resized to the latent grid, rotary tables for the latent's (frames, rows, columns),
guidance x 1000, t x 1000), `get_rotary_pos_embed` (:35-81), `to_layers()` = 1 + 20 double + concatenate + 40 single + 1 (:483-492) and the
layer wrappers (:544-680) with their stage-boundary tuples
    (img, txt, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args)   before the concatenation,
    (x,        vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args)   after it.
The reference's wrappers read max_seqlen / txt_seq_len / img_seq_len / unpatchify_args back to the host (`.item()`, one sync per block);
here the same tensors ride along for the next stage but every length a block needs comes from tensor SHAPES (image tokens = rows of the
rotary table, text tokens = the rest) and the valid-text count stays a device tensor derived from cu_seqlens -- no sync, graph-capturable.

What stands in for the un-vendored `hyvideo` package (empty submodule in the snapshot; parity unpinned, restated from the published
tencent/HunyuanVideo modules with their parameter names so checkpoints map 1:1): TimestepEmbedder, MLPEmbedder, PatchEmbed (a Conv3d with
kernel = stride = patch, executed as one GEMM over the patchified latent), SingleTokenRefiner, FinalLayer, unpatchify.  The double / single
stream blocks are workloads/mmdit.py (dataflow pinned against models/hunyuan_image_modeling.py).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import nn as dnn
from .. import ops
from . import mmdit


def make_contiguous(*values):
    return tuple(x.contiguous() if torch.is_tensor(x) else x for x in values)


@dataclass
class HunyuanVideoConfig:
    """HYVideo-T/2-cfgdistill (hyvideo/modules/models.py HUNYUAN_VIDEO_CONFIG)."""
    in_channels: int = 16
    out_channels: int = 16
    hidden_size: int = 3072
    heads_num: int = 24
    mlp_width_ratio: float = 4.0
    mm_double_blocks_depth: int = 20
    mm_single_blocks_depth: int = 40
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    rope_dim_list: Tuple[int, int, int] = (16, 56, 56)
    text_states_dim: int = 4096
    text_states_dim_2: int = 768
    guidance_embed: bool = True
    refiner_depth: int = 2


def tiny_hv_config():
    """Same topology at head_dim 64 (rope 8 + 28 + 28): 2 double + 3 single blocks, for parity tests the oracle finishes in seconds."""
    return HunyuanVideoConfig(in_channels=4, out_channels=4, hidden_size=128, heads_num=2, mm_double_blocks_depth=2, mm_single_blocks_depth=3,
                              rope_dim_list=(8, 28, 28), text_states_dim=96, text_states_dim_2=48)


# ------------------------------------------------------------------------------------------------------------ rotary tables
def get_nd_rotary_pos_embed(rope_dim_list, sizes, theta=256.0):
    """[3P] hyvideo posemb_layers.get_nd_rotary_pos_embed(use_real=True): cos / sin [S, head_dim], every frequency repeated for its pair."""
    grids = torch.meshgrid(*[torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in sizes], indexing='ij')
    cos, sin = [], []
    for dim, g in zip(rope_dim_list, grids):
        freqs = 1.0 / theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)
        ang = torch.outer(g.reshape(-1), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1))
        sin.append(ang.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


def get_rotary_pos_embed(cfg, video_length, height, width):
    """models/hunyuan_video.py:35-81 (884 VAE: latent = ((frames - 1) / 4 + 1, height / 8, width / 8), rope_theta 256)."""
    latents_size = [(video_length - 1) // 4 + 1, height // 8, width // 8]
    assert all(s % p == 0 for s, p in zip(latents_size, cfg.patch_size)), \
        f'Latent size(last 3 dimensions) should be divisible by patch size({list(cfg.patch_size)}), but got {latents_size}.'
    rope_sizes = [s // p for s, p in zip(latents_size, cfg.patch_size)]
    head_dim = cfg.hidden_size // cfg.heads_num
    rope_dim_list = list(cfg.rope_dim_list) if cfg.rope_dim_list is not None else [head_dim // 3] * 3
    assert sum(rope_dim_list) == head_dim, 'sum(rope_dim_list) should equal to head_dim of attention layer'
    return get_nd_rotary_pos_embed(rope_dim_list, rope_sizes, theta=256.0)


def get_cu_seqlens(text_mask, img_len):
    """[3P] hyvideo attenion.get_cu_seqlens without its host loop: [2B + 1] int32, per sample the end of the valid (image + valid text) run and
    the end of the padded slot."""
    B, max_len = text_mask.shape[0], text_mask.shape[1] + img_len
    base = torch.arange(B, device=text_mask.device, dtype=torch.int64) * max_len
    cu = torch.zeros(2 * B + 1, dtype=torch.int32, device=text_mask.device)
    cu[1::2] = (base + text_mask.sum(dim=1) + img_len).to(torch.int32)
    cu[2::2] = (base + max_len).to(torch.int32)
    return cu


# ------------------------------------------------------------------------------------------------------------------ modules
class TimestepEmbedder(nn.Module):
    def __init__(self, hidden, freq_dim=256):
        super().__init__()
        self.freq_dim = freq_dim
        self.mlp = nn.Sequential(dnn.Linear(freq_dim, hidden), dnn.SiLU(), dnn.Linear(hidden, hidden))

    def forward(self, t):
        feats = ops.sinusoidal_embedding(t, self.freq_dim, 10000.0, sin_first=False)          # [cos | sin], fp32
        return self.mlp(feats.to(self.mlp[0].weight.dtype))


class MLPEmbedder(nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.in_layer, self.silu, self.out_layer = dnn.Linear(in_dim, hidden), dnn.SiLU(), dnn.Linear(hidden, hidden)

    def forward(self, x):
        return self.out_layer(self.silu(self.in_layer(x)))


class TextProjection(nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.linear_1, self.act_1, self.linear_2 = dnn.Linear(in_dim, hidden), dnn.SiLU(), dnn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(self.act_1(self.linear_1(x)))


class _Conv3dParams(nn.Module):
    """holds `weight` [out, in, pt, ph, pw] / `bias` under the Conv3d names"""

    def __init__(self, in_chans, hidden, patch):
        super().__init__()
        ref = nn.Conv3d(in_chans, hidden, kernel_size=tuple(patch), stride=tuple(patch))
        self.weight, self.bias = nn.Parameter(ref.weight.detach().clone()), nn.Parameter(ref.bias.detach().clone())


class PatchEmbed(nn.Module):
    """Conv3d(kernel = stride = patch) -> [B, tokens, hidden]: non-overlapping patches, so the convolution is ONE GEMM over the patchified latent."""

    def __init__(self, patch_size, in_chans, hidden):
        super().__init__()
        self.patch_size = tuple(patch_size)
        self.proj = _Conv3dParams(in_chans, hidden, patch_size)

    def forward(self, x):
        B, C, T, H, W = x.shape
        pt, ph, pw = self.patch_size
        p = x.view(B, C, T // pt, pt, H // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, -1, C * pt * ph * pw)
        return ops.linear(p.to(self.proj.weight.dtype), self.proj.weight.view(self.proj.weight.shape[0], -1), self.proj.bias)


class _RefinerMLP(nn.Module):
    def __init__(self, hidden, mlp_hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = dnn.Linear(hidden, mlp_hidden), dnn.SiLU(), dnn.Linear(mlp_hidden, hidden)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class IndividualTokenRefinerBlock(nn.Module):
    def __init__(self, hidden, heads, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        d = hidden // heads
        self.norm1 = dnn.LayerNorm(hidden, eps=1e-6)
        self.self_attn_qkv = dnn.Linear(hidden, 3 * hidden)
        self.self_attn_q_norm, self.self_attn_k_norm = dnn.LayerNorm(d, eps=1e-6), dnn.LayerNorm(d, eps=1e-6)
        self.self_attn_proj = dnn.Linear(hidden, hidden)
        self.norm2 = dnn.LayerNorm(hidden, eps=1e-6)
        self.mlp = _RefinerMLP(hidden, int(hidden * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(dnn.SiLU(), dnn.Linear(hidden, 2 * hidden))

    def forward(self, x, c, text_len):
        """text_len: int32 [B] valid tokens (padding last).  Valid queries see the valid keys, as under the reference's mask; what the
        padded rows hold differs from the reference (there: key 0 only) and is never read by a valid token downstream."""
        gate_msa, gate_mlp = self.adaLN_modulation(c).chunk(2, dim=1)
        B, L, _ = x.shape
        q, k, v = self.self_attn_qkv(self.norm1(x)).view(B, L, 3, self.heads, -1).unbind(2)
        q, k = self.self_attn_q_norm(q.contiguous()), self.self_attn_k_norm(k.contiguous())
        o = ops.attention(q, k, v.contiguous(), kv_len=text_len).reshape(B, L, -1)
        x = ops.gated_residual(x, self.self_attn_proj(o), gate_msa)
        return ops.gated_residual(x, self.mlp(self.norm2(x)), gate_mlp)


class IndividualTokenRefiner(nn.Module):
    def __init__(self, hidden, heads, depth):
        super().__init__()
        self.blocks = nn.ModuleList([IndividualTokenRefinerBlock(hidden, heads) for _ in range(depth)])

    def forward(self, x, c, text_len):
        for blk in self.blocks:
            x = blk(x, c, text_len)
        return x


class SingleTokenRefiner(nn.Module):
    def __init__(self, in_dim, hidden, heads, depth=2):
        super().__init__()
        self.input_embedder = dnn.Linear(in_dim, hidden)
        self.t_embedder = TimestepEmbedder(hidden)
        self.c_embedder = TextProjection(in_dim, hidden)
        self.individual_token_refiner = IndividualTokenRefiner(hidden, heads, depth)

    def forward(self, x, t, mask=None):
        dt = self.input_embedder.weight.dtype
        timestep_aware = self.t_embedder(t)
        text_len = None
        if mask is None:
            context = x.float().mean(dim=1)
        else:
            mf = mask.float().unsqueeze(-1)
            context = (x.float() * mf).sum(dim=1) / mf.sum(dim=1)
            text_len = mask.sum(dim=1).to(torch.int32)
        c = timestep_aware + self.c_embedder(context.to(dt))
        return self.individual_token_refiner(self.input_embedder(x.to(dt)), c, text_len)


class FinalLayer(nn.Module):
    def __init__(self, hidden, patch_size, out_channels):
        super().__init__()
        self.norm_final = dnn.LayerNorm(hidden, eps=1e-6, elementwise_affine=False)
        self.linear = dnn.Linear(hidden, patch_size[0] * patch_size[1] * patch_size[2] * out_channels)
        self.adaLN_modulation = nn.Sequential(dnn.SiLU(), dnn.Linear(hidden, 2 * hidden))

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)
        return self.linear(self.norm_final(x, scale=scale, shift=shift))


class HYVideoDiffusionTransformer(nn.Module):
    def __init__(self, cfg: HunyuanVideoConfig):
        super().__init__()
        h = cfg.hidden_size
        self.config = cfg
        self.patch_size, self.rope_dim_list, self.hidden_size, self.heads_num = list(cfg.patch_size), list(cfg.rope_dim_list), h, cfg.heads_num
        self.unpatchify_channels = cfg.out_channels
        self.guidance_embed, self.text_projection, self.use_attention_mask = cfg.guidance_embed, 'single_refiner', True
        self.img_in = PatchEmbed(cfg.patch_size, cfg.in_channels, h)
        self.txt_in = SingleTokenRefiner(cfg.text_states_dim, h, cfg.heads_num, depth=cfg.refiner_depth)
        self.time_in = TimestepEmbedder(h)
        self.vector_in = MLPEmbedder(cfg.text_states_dim_2, h)
        self.guidance_in = TimestepEmbedder(h) if cfg.guidance_embed else None
        self.double_blocks = nn.ModuleList([mmdit.MMDoubleStreamBlock(h, cfg.heads_num, cfg.mlp_width_ratio) for _ in range(cfg.mm_double_blocks_depth)])
        self.single_blocks = nn.ModuleList([mmdit.MMSingleStreamBlock(h, cfg.heads_num, cfg.mlp_width_ratio) for _ in range(cfg.mm_single_blocks_depth)])
        self.final_layer = FinalLayer(h, cfg.patch_size, cfg.out_channels)

    def unpatchify(self, x, t, h, w):
        c = self.unpatchify_channels
        pt, ph, pw = self.patch_size
        x = x.reshape(x.shape[0], t, h, w, c, pt, ph, pw)
        return torch.einsum('nthwcopq->nctohpwq', x).reshape(x.shape[0], c, t * pt, h * ph, w * pw)


# ------------------------------------------------------------------------------------------------------ pipeline layer wrappers
def _half_tables(cos, sin):
    """[S, head_dim] boundary tables (every frequency twice) -> the kernels' [S, head_dim / 2]"""
    return cos[:, 0::2].contiguous(), sin[:, 0::2].contiguous()


def _text_len(cu_seqlens, img_len, max_len):
    """valid text tokens per sample, as a device tensor (the reference's blocks hand cu_seqlens to flash-attn varlen)"""
    B = (cu_seqlens.numel() - 1) // 2
    return (cu_seqlens[1::2].to(torch.int64) - torch.arange(B, device=cu_seqlens.device) * max_len - img_len).to(torch.int32)


class InitialLayer(nn.Module):
    """models/hunyuan_video.py:544-615."""

    def __init__(self, transformer):
        super().__init__()
        self.transformer = [transformer]
        self.time_in, self.vector_in = transformer.time_in, transformer.vector_in
        self.guidance_embed, self.guidance_in = transformer.guidance_embed, transformer.guidance_in
        self.img_in, self.text_projection, self.txt_in = transformer.img_in, transformer.text_projection, transformer.txt_in
        self._consts = {}

    def _const(self, values, device):
        """small integer tensors of the stage tuple (token counts, unpatchify grid): built once per value (a host -> device copy cannot be
        captured into a hipGraph) and constant for a given input shape"""
        key = (tuple(values) if isinstance(values, (tuple, list)) else values, str(device))
        if key not in self._consts:
            self._consts[key] = torch.tensor(values, device=device)
        return self._consts[key]

    def forward(self, inputs):
        for item in inputs:
            if torch.is_floating_point(item):
                item.requires_grad_(True)
        x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance = inputs
        tr = self.transformer[0]
        _, _, ot, oh, ow = x.shape
        tt, th, tw = ot // tr.patch_size[0], oh // tr.patch_size[1], ow // tr.patch_size[2]
        unpatchify_args = self._const([tt, th, tw], x.device)
        assert freqs_cos.ndim == 3
        freqs_cos, freqs_sin = freqs_cos[0], freqs_sin[0]
        dt = self.vector_in.in_layer.weight.dtype
        vec = self.time_in(t) + self.vector_in(text_states_2.to(dt))
        if self.guidance_embed:
            if guidance is None:
                raise ValueError("Didn't get guidance strength for guidance distilled model.")
            vec = vec + self.guidance_in(guidance)
        img = self.img_in(x)
        if self.text_projection == 'linear':
            txt = self.txt_in(text_states.to(dt))
        elif self.text_projection == 'single_refiner':
            txt = self.txt_in(text_states, t, text_mask if tr.use_attention_mask else None)
        else:
            raise NotImplementedError(f'Unsupported text_projection: {self.text_projection}')
        txt_seq_len, img_seq_len = txt.shape[1], img.shape[1]
        cu_seqlens = get_cu_seqlens(text_mask, img_seq_len)
        max_seqlen = self._const(img_seq_len + txt_seq_len, img.device)
        txt_seq_len = self._const(txt_seq_len, img.device)
        img_seq_len = self._const(img_seq_len, img.device)
        return make_contiguous(img, txt, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args)


class DoubleBlock(nn.Module):
    """models/hunyuan_video.py:618-633 (block swap offloader hooks: out of scope)."""

    def __init__(self, block, block_idx, offloader=None):
        super().__init__()
        self.block, self.block_idx = block, block_idx

    def forward(self, inputs):
        img, txt, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args = inputs
        cos, sin = _half_tables(freqs_cos, freqs_sin)
        text_len = _text_len(cu_seqlens, img.shape[1], img.shape[1] + txt.shape[1])
        img, txt = self.block(img, txt, vec, cos, sin, text_len)
        return make_contiguous(img, txt, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args)


def concatenate_hidden_states(inputs):
    """models/hunyuan_video.py:636-639: a bare callable in the layer list."""
    img, txt, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args = inputs
    x = torch.cat((img, txt), 1)
    return x, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args


class SingleBlock(nn.Module):
    """models/hunyuan_video.py:642-657."""

    def __init__(self, block, block_idx, offloader=None):
        super().__init__()
        self.block, self.block_idx = block, block_idx

    def forward(self, inputs):
        x, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args = inputs
        cos, sin = _half_tables(freqs_cos, freqs_sin)
        img_len = freqs_cos.shape[0]                                   # image tokens = rows of the rotary table
        text_len = _text_len(cu_seqlens, img_len, x.shape[1])
        x = self.block(x, vec, x.shape[1] - img_len, cos, sin, text_len)
        return make_contiguous(x, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args)


class OutputLayer(nn.Module):
    """models/hunyuan_video.py:659-680.  The (frames, rows, columns) of the token grid are read from the device tensor once per input shape
    (eager warm-up); under hipGraph capture the cached triple is used -- all micro-batches of a captured shape share it."""

    def __init__(self, transformer):
        super().__init__()
        self.transformer = [transformer]
        self.final_layer = transformer.final_layer
        self._grid = {}

    def forward(self, inputs):
        x, vec, cu_seqlens, max_seqlen, freqs_cos, freqs_sin, txt_seq_len, img_seq_len, unpatchify_args = inputs
        img_len = freqs_cos.shape[0]
        key = (tuple(x.shape), img_len)
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            if key not in self._grid:
                raise RuntimeError('OutputLayer: token grid unknown for this shape (run one eager step before capturing)')
            tt, th, tw = self._grid[key]
        else:
            tt, th, tw = (int(v) for v in unpatchify_args.tolist())
            self._grid[key] = (tt, th, tw)
        img = self.final_layer(x[:, :img_len, ...].contiguous(), vec)
        return self.transformer[0].unpatchify(img, tt, th, tw)


# ---------------------------------------------------------------------------------------------------------------- the adapter
class HunyuanVideoWorkload:
    """Adapter-API subset the engine needs over a randomly initialised HunyuanVideo transformer."""
    name = 'hunyuan-video'
    checkpointable_layers = ['DoubleBlock', 'SingleBlock']
    adapter_target_modules = ['MMDoubleStreamBlock', 'MMSingleStreamBlock']

    def __init__(self, config=None, model_config=None, dtype=torch.bfloat16, seed=0, device='cpu'):
        self.cfg = config or HunyuanVideoConfig()
        self.model_config = model_config or {}
        self.train_config = {}
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        with torch.device(device):
            self.transformer = HYVideoDiffusionTransformer(self.cfg)
        torch.random.set_rng_state(state)
        self.transformer.to(dtype)
        for n, p in self.transformer.named_parameters():
            p.original_name = n

    def modules(self):
        return {'transformer': self.transformer}

    def prepare_inputs(self, inputs, timestep_quantile=None):
        """models/hunyuan_video.py:413-481."""
        latents = inputs['latents'].float()
        prompt_embeds_1, prompt_attention_mask_1 = inputs['prompt_embeds_1'], inputs['prompt_attention_mask_1']
        prompt_embeds_2, mask = inputs['prompt_embeds_2'], inputs['mask']
        bs, channels, num_frames, h, w = latents.shape
        if mask is not None:
            mask = F.interpolate(mask.unsqueeze(1), size=(h, w), mode='nearest-exact').unsqueeze(2)
        guidance_expand = torch.tensor([self.model_config.get('guidance', 1.0)] * bs, dtype=torch.float32) * 1000
        method = self.model_config.get('timestep_sample_method', 'logit_normal')
        if method == 'logit_normal':
            dist = torch.distributions.normal.Normal(0, 1)
        elif method == 'uniform':
            dist = torch.distributions.uniform.Uniform(0, 1)
        else:
            raise NotImplementedError()
        if timestep_quantile is not None:
            t = dist.icdf(torch.full((bs,), timestep_quantile, device=latents.device))
        else:
            t = dist.sample((bs,)).to(latents.device)
        if method == 'logit_normal':
            t = torch.sigmoid(t * self.model_config.get('sigmoid_scale', 1.0))
        if shift := self.model_config.get('shift', None):
            t = (t * shift) / (1 + (shift - 1) * t)
        x_1 = latents
        x_0 = torch.randn_like(x_1)
        t_expanded = t.view(-1, 1, 1, 1, 1)
        x_t = (1 - t_expanded) * x_1 + t_expanded * x_0
        target = x_0 - x_1
        freqs_cos, freqs_sin = get_rotary_pos_embed(self.cfg, (num_frames - 1) * 4 + 1, h * 8, w * 8)
        freqs_cos, freqs_sin = freqs_cos.expand(bs, -1, -1), freqs_sin.expand(bs, -1, -1)
        return (x_t, t * 1000, prompt_embeds_1, prompt_attention_mask_1, prompt_embeds_2, freqs_cos, freqs_sin, guidance_expand), (target, mask)

    def to_layers(self):
        tr = self.transformer
        layers = [InitialLayer(tr)]
        layers += [DoubleBlock(b, i) for i, b in enumerate(tr.double_blocks)]
        layers.append(concatenate_hidden_states)
        layers += [SingleBlock(b, i) for i, b in enumerate(tr.single_blocks)]
        layers.append(OutputLayer(tr))
        return layers

    def get_loss_fn(self):
        def loss_fn(output, label):
            target, mask = label
            return ops.fused_loss(output, target, mask if mask.numel() > 0 else None)
        return loss_fn

    def get_param_groups(self, parameters):
        return [{'params': list(parameters)}]

    def configure_adapter(self, adapter_config):
        """LoRA on the Linears inside the double / single stream blocks (adapter_target_modules, models/base.py:262-270)."""
        if adapter_config.get('type', 'lora') != 'lora':
            raise NotImplementedError(f"Adapter type {adapter_config['type']} is not implemented")
        inside = set()
        for name, module in self.transformer.named_modules():
            if module.__class__.__name__ in self.adapter_target_modules:
                inside.update(f'{name}.{n}' for n, sub in module.named_modules() if n)
        wrapped = dnn.apply_lora(self.transformer, rank=adapter_config['rank'], alpha=adapter_config['alpha'], dropout=adapter_config.get('dropout', 0.0),
                                 dtype=adapter_config.get('dtype'), target=lambda name, module: name in inside)
        for n, p in self.transformer.named_parameters():
            p.original_name = n
        return wrapped

    def save_adapter(self, save_dir, peft_state_dict, adapter_config=None):
        from ..formats import save_comfyui_adapter
        save_comfyui_adapter(save_dir, peft_state_dict, adapter_config)

    def save_model(self, save_dir, state_dict):
        from ..formats import save_plain_model
        save_plain_model(save_dir, state_dict)


def synthetic_hv_batch(cfg: HunyuanVideoConfig, batch_size=1, latent_thw=(3, 8, 8), text_tokens=12, valid_text=(9,), seed=0):
    """SURVEY 8(d) config 5 shapes: latents randn[B, 16, F, h, w], llm embeds [B, T, 4096] + mask, clip pooled [B, 768]."""
    g = torch.Generator().manual_seed(seed)
    T, H, W = latent_thw
    valid = [valid_text[i % len(valid_text)] for i in range(batch_size)]
    mask = torch.zeros(batch_size, text_tokens, dtype=torch.int64)
    for i, n in enumerate(valid):
        mask[i, :n] = 1
    return {'latents': torch.randn(batch_size, cfg.in_channels, T, H, W, generator=g),
            'prompt_embeds_1': torch.randn(batch_size, text_tokens, cfg.text_states_dim, generator=g), 'prompt_attention_mask_1': mask,
            'prompt_embeds_2': torch.randn(batch_size, cfg.text_states_dim_2, generator=g), 'mask': None}
