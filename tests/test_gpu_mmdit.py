"""GPU: MMDiT double-stream / single-stream blocks on the HIP kernels (workloads/mmdit.py) vs the oracle's plain-PyTorch
restatement (oracle/blocks_ref.py; dataflow of models/hunyuan_image_modeling.py:61-345, helper definitions unpinned) on the
same seeded weights: outputs and every gradient.  fp32 kernel mode < 1e-3, bf16 training mode < 4e-2 of each tensor's scale.
Flux / HunyuanVideo geometry scaled down (head_dim 128 kept): hidden 256 = 2 heads x 128, 96 image + 24 text tokens
(5 of them padding in sample 1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def _tables(S, d, gen):
    ang = torch.rand(S, d // 2, generator=gen) * 6.28
    return torch.cos(ang), torch.sin(ang)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize('kind', ['double', 'single'])
def test_mmdit_block_matches_oracle(gpu, dtype, tol, kind):
    from diffusion_pipe_amd.workloads import mmdit
    from oracle import blocks_ref as br
    g = torch.Generator().manual_seed(5)
    hidden, heads, Si, St, B = 256, 2, 96, 24, 2
    torch.manual_seed(11)
    block = mmdit.MMDoubleStreamBlock(hidden, heads) if kind == 'double' else mmdit.MMSingleStreamBlock(hidden, heads)
    with torch.no_grad():
        for n, p in block.named_parameters():
            if n.endswith('norm.weight'):
                p.uniform_(0.5, 1.5)
    ref_p = {n: p.detach().clone().requires_grad_(True) for n, p in block.named_parameters()}
    block.to(gpu, dtype)
    cos, sin = _tables(Si, hidden // heads, g)
    # the exact-fp32 attention path (GEMM + softmax kernels) has no key-length masking: fp32 runs unpadded, bf16 (flash) padded
    text_len = torch.tensor([St, St - 5]) if dtype == torch.bfloat16 else None
    tl_gpu = text_len.to(gpu) if text_len is not None else None
    valid = text_len if text_len is not None else torch.tensor([St, St])
    img, txt, vec = torch.randn(B, Si, hidden, generator=g), torch.randn(B, St, hidden, generator=g), torch.randn(B, hidden, generator=g)
    w1, w2 = torch.randn(B, Si, hidden, generator=g), torch.randn(B, St, hidden, generator=g)

    def leaf(t, dev=None, dt=None):
        t = t.clone() if dev is None else t.to(dev, dt)
        return t.requires_grad_(True)

    if kind == 'double':
        xi, xt, xv = leaf(img, gpu, dtype), leaf(txt, gpu, dtype), leaf(vec, gpu, dtype)
        oi, ot = block(xi, xt, xv, cos.to(gpu), sin.to(gpu), tl_gpu)
        # padded text rows of sample 1 are excluded from the loss (their attention rows see garbage-free but unused values)
        tmask = (torch.arange(St)[None, :] < valid[:, None]).float()[..., None]
        ((oi.float() * w1.to(gpu)).sum() + (ot.float() * (w2 * tmask).to(gpu)).sum()).backward()
        ri, rt, rv = leaf(img), leaf(txt), leaf(vec)
        qi, qt = br.mm_double_block(ref_p, ri, rt, rv, heads, cos, sin, text_len)
        ((qi * w1).sum() + (qt * w2 * tmask).sum()).backward()
        outs = [(oi, qi), (ot * tmask.to(gpu), qt * tmask), (xi.grad, ri.grad), (xt.grad, rt.grad), (xv.grad, rv.grad)]
    else:
        x = torch.cat([img, txt], dim=1)
        w = torch.cat([w1, w2], dim=1)
        smask = torch.cat([torch.ones(B, Si), (torch.arange(St)[None, :] < valid[:, None]).float()], dim=1)[..., None]
        xx, xv = leaf(x, gpu, dtype), leaf(vec, gpu, dtype)
        o = block(xx, xv, St, cos.to(gpu), sin.to(gpu), tl_gpu)
        (o.float() * (w * smask).to(gpu)).sum().backward()
        rx, rv = leaf(x), leaf(vec)
        q = br.mm_single_block(ref_p, rx, rv, St, heads, cos, sin, text_len)
        (q * w * smask).sum().backward()
        outs = [(o * smask.to(gpu), q * smask), (xx.grad, rx.grad), (xv.grad, rv.grad)]
    torch.cuda.synchronize()
    for a, b in outs:
        assert _rel(a, b) < tol
    for n, p in block.named_parameters():
        assert _rel(p.grad, ref_p[n].grad) < tol, n
