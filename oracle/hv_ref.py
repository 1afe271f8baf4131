"""ORACLE (test infrastructure, never imported by the product): plain-PyTorch fp32 restatement of the HunyuanVideo transformer as the
reference's pipeline layers drive it (models/hunyuan_video.py:544-680 wrappers over `transformer.time_in / vector_in / guidance_in / img_in /
txt_in / double_blocks / single_blocks / final_layer / unpatchify`).  The transformer itself lives in the un-vendored `hyvideo` package
(empty submodule in the snapshot; tencent/HunyuanVideo, hyvideo/modules/{models,embed_layers,token_refiner,posemb_layers,attenion}.py):
every class below is restated from its published definition -- PARITY UNPINNED -- except the double / single stream block dataflow, which
is oracle/blocks_ref.py's (pinned against the in-tree models/hunyuan_image_modeling.py blocks).  What IS pinned by the reference's own
code: the wrappers, to_layers and prepare_inputs (lifted and run over this module tree: oracle/make_golden_hv_layers.py).

Block call signatures are the ones the reference's wrappers use:
    double(img, txt, vec, cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv, (freqs_cos, freqs_sin)) -> img, txt
    single(x, vec, txt_len, cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv, (freqs_cos, freqs_sin)) -> x
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import blocks_ref as br


# ---- [3P] hyvideo/modules/posemb_layers.py ------------------------------------------------------------------------------------
def get_nd_rotary_pos_embed(rope_dim_list, sizes, theta=10000.0, use_real=True, theta_rescale_factor=1.0):
    """cos / sin [S, sum(rope_dim_list)] of an n-d grid: per axis positions 0 .. size-1 (linspace(0, size, size + 1)[:size]), frequencies
    1 / theta^(2i / dim), each value repeated twice (interleaved real layout), axes concatenated along the feature dim."""
    assert use_real
    grids = torch.meshgrid(*[torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in sizes], indexing='ij')
    cos, sin = [], []
    for dim, g in zip(rope_dim_list, grids):
        freqs = 1.0 / (theta * theta_rescale_factor) ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)
        ang = torch.outer(g.reshape(-1), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1))
        sin.append(ang.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


# ---- [3P] hyvideo/modules/attenion.py -----------------------------------------------------------------------------------------
def get_cu_seqlens(text_mask, img_len):
    """[2B + 1] int32: per sample the end of its valid (image + valid text) run and the end of its padded slot."""
    B = text_mask.shape[0]
    text_len = text_mask.sum(dim=1)
    max_len = text_mask.shape[1] + img_len
    cu = torch.zeros([2 * B + 1], dtype=torch.int32, device=text_mask.device)
    for i in range(B):
        cu[2 * i + 1] = i * max_len + text_len[i] + img_len
        cu[2 * i + 2] = (i + 1) * max_len
    return cu


def text_len_from_cu_seqlens(cu_seqlens, img_len, max_len):
    B = (cu_seqlens.numel() - 1) // 2
    return (cu_seqlens[1::2].to(torch.int64) - torch.arange(B, device=cu_seqlens.device) * max_len - img_len)


# ---- [3P] hyvideo/modules/embed_layers.py -------------------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden, freq_dim=256):
        super().__init__()
        self.freq_dim = freq_dim
        self.mlp = nn.Sequential(nn.Linear(freq_dim, hidden), nn.SiLU(), nn.Linear(hidden, hidden))

    def forward(self, t):
        return self.mlp(timestep_embedding(t, self.freq_dim))


class MLPEmbedder(nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.in_layer, self.silu, self.out_layer = nn.Linear(in_dim, hidden), nn.SiLU(), nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.out_layer(self.silu(self.in_layer(x)))


class TextProjection(nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.linear_1, self.act_1, self.linear_2 = nn.Linear(in_dim, hidden), nn.SiLU(), nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(self.act_1(self.linear_1(x)))


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, hidden):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, hidden, kernel_size=tuple(patch_size), stride=tuple(patch_size))

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


# ---- [3P] hyvideo/modules/token_refiner.py ------------------------------------------------------------------------------------
class IndividualTokenRefinerBlock(nn.Module):
    def __init__(self, hidden, heads, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        d = hidden // heads
        self.norm1 = nn.LayerNorm(hidden, eps=1e-6)
        self.self_attn_qkv = nn.Linear(hidden, 3 * hidden)
        self.self_attn_q_norm, self.self_attn_k_norm = nn.LayerNorm(d, eps=1e-6), nn.LayerNorm(d, eps=1e-6)
        self.self_attn_proj = nn.Linear(hidden, hidden)
        self.norm2 = nn.LayerNorm(hidden, eps=1e-6)
        self.mlp = nn.ModuleDict({'fc1': nn.Linear(hidden, int(hidden * mlp_ratio)), 'fc2': nn.Linear(int(hidden * mlp_ratio), hidden)})
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden, 2 * hidden))

    def forward(self, x, c, attn_mask):
        gate_msa, gate_mlp = self.adaLN_modulation(c).chunk(2, dim=1)
        B, L, _ = x.shape
        q, k, v = self.self_attn_qkv(self.norm1(x)).view(B, L, 3, self.heads, -1).unbind(2)
        q, k = self.self_attn_q_norm(q), self.self_attn_k_norm(k)
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=attn_mask).transpose(1, 2).reshape(B, L, -1)
        x = x + br.apply_gate(self.self_attn_proj(o), gate_msa)
        return x + br.apply_gate(self.mlp['fc2'](F.silu(self.mlp['fc1'](self.norm2(x)))), gate_mlp)


class IndividualTokenRefiner(nn.Module):
    def __init__(self, hidden, heads, depth):
        super().__init__()
        self.blocks = nn.ModuleList([IndividualTokenRefinerBlock(hidden, heads) for _ in range(depth)])

    def forward(self, x, c, mask):
        attn_mask = None
        if mask is not None:
            B, L = mask.shape
            m1 = mask.view(B, 1, 1, L).repeat(1, 1, L, 1)
            attn_mask = (m1 & m1.transpose(2, 3)).bool()
            attn_mask[:, :, :, 0] = True            # padded queries keep one key: no NaN rows
        for blk in self.blocks:
            x = blk(x, c, attn_mask)
        return x


class SingleTokenRefiner(nn.Module):
    def __init__(self, in_dim, hidden, heads, depth=2):
        super().__init__()
        self.input_embedder = nn.Linear(in_dim, hidden)
        self.t_embedder = TimestepEmbedder(hidden)
        self.c_embedder = TextProjection(in_dim, hidden)
        self.individual_token_refiner = IndividualTokenRefiner(hidden, heads, depth)

    def forward(self, x, t, mask=None):
        timestep_aware = self.t_embedder(t)
        if mask is None:
            context = x.mean(dim=1)
        else:
            mf = mask.float().unsqueeze(-1)
            context = (x * mf).sum(dim=1) / mf.sum(dim=1)
        c = timestep_aware + self.c_embedder(context)
        return self.individual_token_refiner(self.input_embedder(x), c, mask.bool() if mask is not None else None)


# ---- [3P] hyvideo/modules/models.py -------------------------------------------------------------------------------------------
class _BlockParams(nn.Module):
    """parameter container with hyvideo's names; arithmetic = oracle/blocks_ref.py"""

    def pdict(self):
        return dict(self.named_parameters())


class MMDoubleStreamBlock(_BlockParams):
    def __init__(self, hidden, heads, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        d, mh = hidden // heads, int(hidden * mlp_ratio)
        for s in ('img', 'txt'):
            setattr(self, f'{s}_mod', nn.ModuleDict({'linear': nn.Linear(hidden, 6 * hidden)}))
            setattr(self, f'{s}_attn_qkv', nn.Linear(hidden, 3 * hidden))
            setattr(self, f'{s}_attn_q_norm', nn.ParameterDict({'weight': nn.Parameter(torch.ones(d))}))
            setattr(self, f'{s}_attn_k_norm', nn.ParameterDict({'weight': nn.Parameter(torch.ones(d))}))
            setattr(self, f'{s}_attn_proj', nn.Linear(hidden, hidden))
            setattr(self, f'{s}_mlp', nn.ModuleDict({'fc1': nn.Linear(hidden, mh), 'fc2': nn.Linear(mh, hidden)}))

    def forward(self, img, txt, vec, cu_q, cu_kv, max_q, max_kv, freqs_cis):
        cos, sin = freqs_cis
        text_len = text_len_from_cu_seqlens(cu_q, img.shape[1], int(max_q))
        return br.mm_double_block(self.pdict(), img, txt, vec, self.heads, cos[:, 0::2], sin[:, 0::2], text_len)


class MMSingleStreamBlock(_BlockParams):
    def __init__(self, hidden, heads, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        d, mh = hidden // heads, int(hidden * mlp_ratio)
        self.linear1, self.linear2 = nn.Linear(hidden, 3 * hidden + mh), nn.Linear(hidden + mh, hidden)
        self.q_norm = nn.ParameterDict({'weight': nn.Parameter(torch.ones(d))})
        self.k_norm = nn.ParameterDict({'weight': nn.Parameter(torch.ones(d))})
        self.modulation = nn.ModuleDict({'linear': nn.Linear(hidden, 3 * hidden)})

    def forward(self, x, vec, txt_len, cu_q, cu_kv, max_q, max_kv, freqs_cis):
        cos, sin = freqs_cis
        text_len = text_len_from_cu_seqlens(cu_q, x.shape[1] - txt_len, int(max_q))
        return br.mm_single_block(self.pdict(), x, vec, txt_len, self.heads, cos[:, 0::2], sin[:, 0::2], text_len)


class FinalLayer(nn.Module):
    def __init__(self, hidden, patch_size, out_channels):
        super().__init__()
        self.linear = nn.Linear(hidden, patch_size[0] * patch_size[1] * patch_size[2] * out_channels)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden, 2 * hidden))

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)
        return self.linear(br.modulate(br.layer_norm(x, 1e-6), shift, scale))


class HYVideoDiffusionTransformer(nn.Module):
    def __init__(self, cfg, seed=0):
        super().__init__()
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        h = cfg.hidden_size
        self.patch_size, self.rope_dim_list, self.hidden_size, self.heads_num = list(cfg.patch_size), list(cfg.rope_dim_list), h, cfg.heads_num
        self.unpatchify_channels = cfg.out_channels
        self.guidance_embed, self.text_projection, self.use_attention_mask = cfg.guidance_embed, 'single_refiner', True
        self.img_in = PatchEmbed(cfg.patch_size, cfg.in_channels, h)
        self.txt_in = SingleTokenRefiner(cfg.text_states_dim, h, cfg.heads_num, depth=cfg.refiner_depth)
        self.time_in = TimestepEmbedder(h)
        self.vector_in = MLPEmbedder(cfg.text_states_dim_2, h)
        self.guidance_in = TimestepEmbedder(h) if cfg.guidance_embed else None
        self.double_blocks = nn.ModuleList([MMDoubleStreamBlock(h, cfg.heads_num, cfg.mlp_width_ratio) for _ in range(cfg.mm_double_blocks_depth)])
        self.single_blocks = nn.ModuleList([MMSingleStreamBlock(h, cfg.heads_num, cfg.mlp_width_ratio) for _ in range(cfg.mm_single_blocks_depth)])
        self.final_layer = FinalLayer(h, cfg.patch_size, cfg.out_channels)
        for n, p in self.named_parameters():        # non-trivial norm scales and modulation so every path carries signal
            if n.endswith('norm.weight') and p.dim() == 1:
                nn.init.normal_(p, 1.0, 0.1)
        torch.random.set_rng_state(state)

    def unpatchify(self, x, t, h, w):
        c = self.unpatchify_channels
        pt, ph, pw = self.patch_size
        x = x.reshape(x.shape[0], t, h, w, c, pt, ph, pw)
        return torch.einsum('nthwcopq->nctohpwq', x).reshape(x.shape[0], c, t * pt, h * ph, w * pw)
