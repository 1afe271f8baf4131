"""GPU: the fp32 per-channel addend path of round 6 (ops.precise_row_linear + the hi / lo bias pair of the implicit-GEMM convolution) against plain PyTorch fp64 / fp32
references of the same arithmetic (diffusers TimestepEmbedding / ResnetBlock2D.time_emb_proj + `hidden_states + temb[:, :, None, None]` behind models/sdxl.py:797-865).
Tolerances: the addend itself must be fp32-accurate (1e-5 of its scale -- a bf16 row would be ~2e-3 off); gradients are bf16 quantities (2e-2 of each tensor's scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize('R', [1, 4])
@pytest.mark.parametrize('act', [None, 'silu'])
@pytest.mark.parametrize('pair', [False, True])
def test_precise_row_linear_forward_is_fp32_accurate_and_backward_matches(gpu, R, act, pair):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(11)
    K, N = 1280, 640
    x = (torch.randn(R, K, generator=g) * 1.5).to(gpu).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, torch.bfloat16).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to(gpu, torch.bfloat16).requires_grad_(True)
    e = (torch.randn(N, generator=g) * 0.1).to(gpu, torch.bfloat16).requires_grad_(True)
    y = ops.precise_row_linear(x, w, b, extra=e, act=act, pair=pair)
    xd = x.detach().double().requires_grad_(True)
    wd, bd, ed = (t.detach().double().requires_grad_(True) for t in (w, b, e))
    want = F.linear(F.silu(xd) if act else xd, wd, bd) + ed
    got = (y[0].float() + y[1].float()) if pair else y
    assert got.shape == (R, N) and (y.dtype == torch.bfloat16) == pair
    assert _rel(got, want) < (3e-5 if pair else 1e-5)                  # (the pair re-rounds its lo half: 2^-17)
    # what a bf16 row would have been: two orders of magnitude further away
    naive = F.linear((F.silu(x) if act else x).to(torch.bfloat16), w, b).float()
    assert _rel(naive + e.float(), want) > 20 * _rel(got, want)
    go = torch.randn(R, N, generator=g).to(gpu)
    if pair:
        y.backward(go.to(torch.bfloat16).reshape(1, R, N).expand(2, R, N))          # what the convolution hands back: both rows = the gradient of their sum
        go = go.to(torch.bfloat16).float()
    else:
        y.backward(go)
    want.backward(go.double())
    torch.cuda.synchronize()
    assert x.grad.dtype == torch.float32 and _rel(x.grad, xd.grad) < 2e-2
    assert _rel(w.grad, wd.grad) < 2e-2 and _rel(b.grad, bd.grad) < 2e-2 and _rel(e.grad, ed.grad) < 2e-2


def test_precise_row_linear_accumulates_into_grad_buffers(gpu):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(12)
    K, N = 320, 1280
    x = torch.randn(1, K, generator=g).to(gpu).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, torch.bfloat16).requires_grad_(True)
    b = torch.zeros(N, device=gpu, dtype=torch.bfloat16, requires_grad=True)
    go = torch.randn(1, N, generator=g).to(gpu)
    ops.precise_row_linear(x, w, b).backward(go)
    w1, b1 = w.grad.clone(), b.grad.clone()
    ops.precise_row_linear(x, w, b).backward(go)                       # second micro-batch: fused accumulation into the persistent .grad buffers
    torch.cuda.synchronize()
    assert _rel(w.grad, 2 * w1.float()) < 1e-2 and _rel(b.grad, 2 * b1.float()) < 1e-2


@pytest.mark.parametrize('B', [1, 3])
def test_conv_takes_the_addend_as_a_hi_lo_pair(gpu, B):
    """conv1(h) + (time_emb_proj(silu(emb)) + conv1.bias)[:, :, None, None] with the addend as a pair: forward equal to the fp32 reference up to the OUTPUT rounding;
    gradients of the convolution bias, the projection and the embedding agree with the reference."""
    from diffusion_pipe_amd import nn as dnn, ops
    torch.manual_seed(5)
    Cin, Cout, K, hw = 64, 128, 256, 16
    conv = dnn.Conv2d(Cin, Cout, 3, padding=1).to(gpu, torch.bfloat16)
    proj = dnn.Linear(K, Cout).to(gpu, torch.bfloat16)
    x = torch.randn(B, Cin, hw, hw, device=gpu).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    emb = (torch.randn(B, K, device=gpu) * 2).requires_grad_(True)
    pair = ops.precise_row_linear(emb, proj.weight, proj.bias, extra=conv.bias, act='silu', pair=True)
    assert pair.shape == (2, B, Cout)
    y = conv(x, extra_bias=pair)
    go = torch.randn_like(y)
    y.backward(go)
    torch.cuda.synchronize()
    xd, ed = x.detach().double().requires_grad_(True), emb.detach().double().requires_grad_(True)
    wd, bd = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    pw, pb = proj.weight.detach().double().requires_grad_(True), proj.bias.detach().double().requires_grad_(True)
    want = F.conv2d(xd, wd, bd, padding=1) + F.linear(F.silu(ed), pw, pb)[:, :, None, None]
    want.backward(go.double())
    assert _rel(y, want) < 6e-3                                        # one bf16 rounding of the output
    # the addend reached the accumulator in fp32: the per-channel MEAN error over the pixels is far below one bf16 ulp of the addend
    mean_err = (y.double() - want).mean(dim=(0, 2, 3)).abs().max().item()
    assert mean_err < 2e-4 * want.abs().max().item()
    assert _rel(x.grad, xd.grad) < 2e-2 and _rel(conv.weight.grad, wd.grad) < 2e-2
    assert _rel(conv.bias.grad, bd.grad) < 2e-2 and _rel(proj.bias.grad, pb.grad) < 2e-2
    assert _rel(proj.weight.grad, pw.grad) < 2e-2 and _rel(emb.grad, ed.grad) < 2e-2
