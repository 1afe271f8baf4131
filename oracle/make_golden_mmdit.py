"""ORACLE fixture generator (test infrastructure): the reference's IN-TREE MMDiT blocks -- models/hunyuan_image_modeling.py
(MMDoubleStreamBlock, MMSingleStreamBlock and its joint `attention`), imported UNMODIFIED on CPU.  The file imports seven small helpers
from the un-vendored `hyimage` package; those are supplied as stub modules restated from their published definitions ([3P], named
below: ModulateDiT, MLP, RMSNorm via get_norm_layer, modulate, apply_gate, apply_rotary_emb, flash_attn_no_pad -> masked SDPA).
So the block DATAFLOW (fused QKV split, norm / rope placement, joint [image ; text] attention with the padded-text mask, gate /
residual order, the single block's parallel MLP) is the reference's own code; only the leaf helpers are restatements.

Writes tests/golden/mmdit_blocks_fp32.safetensors (+ .json): weights, inputs, outputs and all gradients of one double and one single
block (hidden 64, 2 heads, text padded 5 of 12 tokens).  Pins oracle/blocks_ref.py:mm_double_block / mm_single_block.

    python oracle/make_golden_mmdit.py
"""
import importlib
import json
import os
import sys
import types

import torch
import torch.nn.functional as F
from safetensors.torch import save_file
from torch import nn

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')


# ------------------------------------------------------------------------------------------------ [3P] hyimage helper stubs
class ModulateDiT(nn.Module):
    def __init__(self, hidden_size, factor, act_layer, dtype=None, device=None):
        super().__init__()
        self.act = act_layer()
        self.linear = nn.Linear(hidden_size, factor * hidden_size, bias=True, dtype=dtype, device=device)

    def forward(self, x):
        return self.linear(self.act(x))


class MLP(nn.Module):
    def __init__(self, in_channels, hidden_channels=None, out_features=None, act_layer=nn.GELU, bias=True, dtype=None, device=None):
        super().__init__()
        self.fc1 = nn.Linear(in_channels, hidden_channels, bias=bias, dtype=dtype, device=device)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_channels, out_features or in_channels, bias=bias, dtype=dtype, device=device)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class RMSNorm(nn.Module):
    def __init__(self, dim, elementwise_affine=True, eps=1e-6, dtype=None, device=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype, device=device)) if elementwise_affine else None

    def forward(self, x):
        y = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x)
        return y * self.weight if self.weight is not None else y


def modulate(x, shift=None, scale=None):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def apply_gate(x, gate=None, tanh=False):
    return x * gate.unsqueeze(1)


def apply_rotary_emb(xq, xk, freqs_cis, head_first=False):
    """freqs_cis = (cos, sin), each [S, D] with every angle repeated for its (2i, 2i+1) pair; x: [B, S, H, D]."""
    cos, sin = freqs_cis
    cos, sin = cos[None, :, None, :], sin[None, :, None, :]

    def rot(x):
        xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
        return torch.stack([-xi, xr], dim=-1).flatten(3)
    return (xq.float() * cos + rot(xq) * sin).type_as(xq), (xk.float() * cos + rot(xk) * sin).type_as(xk)


def flash_attn_no_pad(qkv, key_padding_mask, causal=False, dropout_p=0.0, softmax_scale=None, deterministic=False):
    """qkv: [B, S, 3, H, D]; key_padding_mask: [B, S] True = valid.  Padded keys masked out; padded query rows return 0 (unpad / pad)."""
    q, k, v = (t.transpose(1, 2) for t in qkv.unbind(2))
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=key_padding_mask[:, None, None, :], scale=softmax_scale).transpose(1, 2)
    return o * key_padding_mask[:, :, None, None].to(o.dtype)


def import_reference_mmdit():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    for pkg in ('hyimage', 'hyimage.models', 'hyimage.models.hunyuan', 'hyimage.models.hunyuan.modules'):
        mod(pkg)
    base = 'hyimage.models.hunyuan.modules.'
    mod(base + 'flash_attn_no_pad', flash_attn_no_pad=flash_attn_no_pad)
    mod(base + 'activation_layers', get_activation_layer=lambda name: {'silu': nn.SiLU, 'gelu_tanh': lambda: nn.GELU(approximate='tanh'), 'gelu': nn.GELU}[name])
    mod(base + 'mlp_layers', MLP=MLP, LinearWarpforSingle=None)
    mod(base + 'modulate_layers', ModulateDiT=ModulateDiT, apply_gate=apply_gate, modulate=modulate)
    mod(base + 'norm_layers', get_norm_layer=lambda name: {'rms': RMSNorm, 'layer': nn.LayerNorm}[name])
    mod(base + 'posemb_layers', apply_rotary_emb=apply_rotary_emb)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location('ref_hunyuan_image_modeling', os.path.join(REF, 'models', 'hunyuan_image_modeling.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    m = import_reference_mmdit()
    torch.manual_seed(2024)
    hidden, heads, Si, St, valid = 64, 2, 40, 12, [12, 7]
    dbl = m.MMDoubleStreamBlock(hidden, heads, 4.0, qkv_bias=True).float()
    sgl = m.MMSingleStreamBlock(hidden, heads, 4.0).float()
    with torch.no_grad():
        for blk in (dbl, sgl):
            for n, p in blk.named_parameters():
                if 'norm' in n:
                    p.add_(torch.randn_like(p) * 0.1)            # informative q / k norm weights
    g = torch.Generator().manual_seed(7)
    img = torch.randn(2, Si, hidden, generator=g, requires_grad=True)
    txt = torch.randn(2, St, hidden, generator=g, requires_grad=True)
    vec = torch.randn(2, hidden, generator=g, requires_grad=True)
    d = hidden // heads
    ang = torch.randn(Si, d // 2, generator=g)
    cos, sin = ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)
    text_mask = torch.arange(St)[None, :] < torch.tensor(valid)[:, None]
    w_img, w_txt = torch.randn(2, Si, hidden, generator=g), torch.randn(2, St, hidden, generator=g)

    o_img, o_txt = dbl(img, txt, vec, freqs_cis=(cos, sin), text_mask=text_mask)
    valid_rows = text_mask[:, :, None].float()
    loss_d = (o_img * w_img).sum() + (o_txt * w_txt * valid_rows).sum()        # padded text rows are don't-care
    loss_d.backward()
    tensors = {'in.img': img.detach(), 'in.txt': txt.detach(), 'in.vec': vec.detach(), 'in.cos_half': ang.cos(), 'in.sin_half': ang.sin(),
               'in.text_len': torch.tensor(valid), 'in.w_img': w_img, 'in.w_txt': w_txt,
               'double.out_img': o_img.detach(), 'double.out_txt': o_txt.detach(), 'double.loss': loss_d.detach().reshape(1),
               'double.grad.img': img.grad.clone(), 'double.grad.txt': txt.grad.clone(), 'double.grad.vec': vec.grad.clone()}
    tensors.update({f'double.param.{n}': p.detach().clone() for n, p in dbl.named_parameters()})
    tensors.update({f'double.pgrad.{n}': p.grad.clone() for n, p in dbl.named_parameters()})

    x = torch.cat([img.detach(), txt.detach()], dim=1).requires_grad_(True)
    vec2 = vec.detach().clone().requires_grad_(True)
    w_x = torch.cat([w_img, w_txt * valid_rows], dim=1)
    o_x = sgl(x, vec2, St, freqs_cis=(cos, sin), text_mask=text_mask)
    loss_s = (o_x * w_x).sum()
    loss_s.backward()
    tensors.update({'single.out': o_x.detach(), 'single.loss': loss_s.detach().reshape(1), 'single.grad.x': x.grad.clone(), 'single.grad.vec': vec2.grad.clone()})
    tensors.update({f'single.param.{n}': p.detach().clone() for n, p in sgl.named_parameters()})
    tensors.update({f'single.pgrad.{n}': p.grad.clone() for n, p in sgl.named_parameters()})

    os.makedirs(OUT, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'mmdit_blocks_fp32.safetensors'))
    meta = {'hidden': hidden, 'heads': heads, 'img_tokens': Si, 'txt_tokens': St, 'valid_text': valid, 'torch': torch.__version__,
            'generated_from': 'models/hunyuan_image_modeling.py (imported; hyimage leaf helpers stubbed)',
            'double_loss': float(loss_d), 'single_loss': float(loss_s)}
    with open(os.path.join(OUT, 'mmdit_blocks_fp32.json'), 'w') as fh:
        json.dump(meta, fh, indent=1)
    print(meta)


if __name__ == '__main__':
    main()
