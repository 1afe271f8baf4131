"""MMDiT double-stream / single-stream blocks (Flux, HunyuanVideo, HunyuanImage: BASELINE configs 3 and 5) on the MI355X
kernels.  Dataflow and parameter names follow the reference's in-tree restatement with fused QKV,
models/hunyuan_image_modeling.py:61-345 (its helpers ModulateDiT / MLP / RMSNorm / modulate / apply_gate / apply_rotary_emb
live in the un-vendored hyimage package and are restated from their published definitions; the block dataflow itself is pinned:
the oracle side, oracle/blocks_ref.py:mm_double_block / mm_single_block, is tested against vectors from the reference file imported
unmodified -- oracle/make_golden_mmdit.py).

Per block: AdaLN modulation from `vec` (SiLU -> Linear), LayerNorm fused with scale/shift (K5), ONE fused QKV GEMM per stream
(K1), per-head RMSNorm of q and k (K2), RoPE on the image tokens (K3), joint attention over [image ; text] tokens with the
text padding masked by a key count (K4), gated residuals (K5), GELU-tanh MLP (K6).
"""
import torch
from torch import nn

from .. import nn as dnn
from .. import ops


class ModulateDiT(nn.Module):
    """act -> Linear(hidden, factor * hidden) on the conditioning vector [B, hidden]."""

    def __init__(self, hidden, factor):
        super().__init__()
        self.act = dnn.SiLU()
        self.linear = dnn.Linear(hidden, factor * hidden)

    def forward(self, vec):
        return self.linear(self.act(vec))


class MLP(nn.Module):
    def __init__(self, hidden, mlp_hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = dnn.Linear(hidden, mlp_hidden), dnn.GELU(approximate='tanh'), dnn.Linear(mlp_hidden, hidden)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def _qkv_heads(qkv, heads):
    """[B, S, 3 H d] -> q, k, v views [B, S, H, d] (models/hunyuan_image_modeling.py:181-182)."""
    B, S, _ = qkv.shape
    q, k, v = qkv.view(B, S, 3, heads, -1).unbind(2)
    return q, k, v


def joint_attention(img_qkv, txt_qkv, text_len=None):
    """Attention over the concatenated [image ; text] sequence (models/hunyuan_image_modeling.py:20-58).  text_len: int32
    [B] valid text tokens (the reference passes a boolean text mask; padding sits at the end of the text)."""
    q, k, v = (torch.cat([a, b], dim=1) for a, b in zip(img_qkv, txt_qkv))
    kv_len = None
    if text_len is not None:
        kv_len = (text_len.to(torch.int32) + img_qkv[0].shape[1]).contiguous()
    o = ops.attention(q, k, v, kv_len=kv_len)
    return o.reshape(o.shape[0], o.shape[1], -1)


class MMDoubleStreamBlock(nn.Module):
    def __init__(self, hidden_size, heads_num, mlp_width_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.heads_num = heads_num
        d = hidden_size // heads_num
        mlp_hidden = int(hidden_size * mlp_width_ratio)
        for s in ('img', 'txt'):
            setattr(self, f'{s}_mod', ModulateDiT(hidden_size, 6))
            setattr(self, f'{s}_norm1', dnn.LayerNorm(hidden_size, eps=1e-6, elementwise_affine=False))
            setattr(self, f'{s}_attn_qkv', dnn.Linear(hidden_size, 3 * hidden_size, bias=qkv_bias))
            setattr(self, f'{s}_attn_q_norm', dnn.RMSNorm(d, eps=1e-6))
            setattr(self, f'{s}_attn_k_norm', dnn.RMSNorm(d, eps=1e-6))
            setattr(self, f'{s}_attn_proj', dnn.Linear(hidden_size, hidden_size, bias=qkv_bias))
            setattr(self, f'{s}_norm2', dnn.LayerNorm(hidden_size, eps=1e-6, elementwise_affine=False))
            setattr(self, f'{s}_mlp', MLP(hidden_size, mlp_hidden))

    def _stream_qkv(self, s, x, shift, scale, cos, sin):
        h, skip = getattr(self, f'{s}_norm1')(x, scale=scale, shift=shift, with_skip=True)
        q, k, v = _qkv_heads(getattr(self, f'{s}_attn_qkv')(h), self.heads_num)
        qn, kn = getattr(self, f'{s}_attn_q_norm'), getattr(self, f'{s}_attn_k_norm')
        if cos is not None:       # image stream: per-head RMSNorm + RoPE in one pass over q and k (K2 + K3), read straight from the fused QKV output
            q = ops.rms_norm_rope(q, qn.weight, cos, sin, qn.eps, per_head=True)
            k = ops.rms_norm_rope(k, kn.weight, cos, sin, kn.eps, per_head=True)
        else:
            q, k = qn(q), kn(k)
        return (q, k, v), skip

    def forward(self, img, txt, vec, cos=None, sin=None, text_len=None):
        """img [B, Si, C], txt [B, St, C], vec [B, C]; cos / sin: fp32 [Si, d/2] rotary tables of the image tokens."""
        i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = self.img_mod(vec).chunk(6, dim=-1)
        t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = self.txt_mod(vec).chunk(6, dim=-1)
        iq, img = self._stream_qkv('img', img, i_sh1, i_sc1, cos, sin)         # img / txt: the norm nodes' aliases for the residual adds
        tq, txt = self._stream_qkv('txt', txt, t_sh1, t_sc1, None, None)
        attn = joint_attention(iq, tq, text_len)
        Si = img.shape[1]
        img_attn, txt_attn = attn[:, :Si].contiguous(), attn[:, Si:].contiguous()
        img = ops.gated_residual(img, self.img_attn_proj(img_attn), i_g1)
        n, img = self.img_norm2(img, scale=i_sc2, shift=i_sh2, with_skip=True)       # residual gradient folded into the LayerNorm backward
        img = ops.gated_residual(img, self.img_mlp(n), i_g2)
        txt = ops.gated_residual(txt, self.txt_attn_proj(txt_attn), t_g1)
        n, txt = self.txt_norm2(txt, scale=t_sc2, shift=t_sh2, with_skip=True)
        txt = ops.gated_residual(txt, self.txt_mlp(n), t_g2)
        return img, txt


class MMSingleStreamBlock(nn.Module):
    def __init__(self, hidden_size, heads_num, mlp_width_ratio=4.0):
        super().__init__()
        self.hidden_size, self.heads_num = hidden_size, heads_num
        d = hidden_size // heads_num
        self.mlp_hidden_dim = int(hidden_size * mlp_width_ratio)
        self.linear1 = dnn.Linear(hidden_size, 3 * hidden_size + self.mlp_hidden_dim)       # QKV || MLP-in, one GEMM
        self.linear2 = dnn.Linear(hidden_size + self.mlp_hidden_dim, hidden_size)
        self.q_norm, self.k_norm = dnn.RMSNorm(d, eps=1e-6), dnn.RMSNorm(d, eps=1e-6)
        self.pre_norm = dnn.LayerNorm(hidden_size, eps=1e-6, elementwise_affine=False)
        self.mlp_act = dnn.GELU(approximate='tanh')
        self.modulation = ModulateDiT(hidden_size, 3)

    def forward(self, x, vec, txt_len, cos=None, sin=None, text_len=None):
        """x [B, Si + St, C] with the text tokens last; txt_len = St (python int, as in the reference)."""
        shift, scale, gate = self.modulation(vec).chunk(3, dim=-1)
        n, x = self.pre_norm(x, scale=scale, shift=shift, with_skip=True)
        h = self.linear1(n)
        qkv, mlp = torch.split(h, [3 * self.hidden_size, self.mlp_hidden_dim], dim=-1)
        q, k, v = _qkv_heads(qkv.contiguous(), self.heads_num)
        Si = x.shape[1] - txt_len
        if cos is not None:       # one pass over q and k: per-head RMSNorm of every token, rotation of the Si image tokens only (no slice / concatenate copies)
            q = ops.rms_norm_rope(q, self.q_norm.weight, cos, sin, self.q_norm.eps, per_head=True, rope_tokens=Si)
            k = ops.rms_norm_rope(k, self.k_norm.weight, cos, sin, self.k_norm.eps, per_head=True, rope_tokens=Si)
        else:
            q, k = self.q_norm(q), self.k_norm(k)
        kv_len = (text_len.to(torch.int32) + Si).contiguous() if text_len is not None else None
        o = ops.attention(q.contiguous(), k.contiguous(), v, kv_len=kv_len)
        attn = o.reshape(o.shape[0], o.shape[1], -1)
        out = self.linear2(torch.cat([attn, self.mlp_act(mlp.contiguous())], dim=2))
        return ops.gated_residual(x, out, gate)
