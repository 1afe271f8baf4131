"""GPU parity tests of the HIP kernels (call through the C ABI via diffusion_pipe_amd.ops).

Each kernel is compared with a plain PyTorch fp32 evaluation of the same op on the same seeded inputs.
Tolerances: fp32 kernels 1e-4 relative (exact-fp32 MFMA / fp32 VALU math, different summation order);
bf16 kernels: inputs are rounded to bf16 first, the reference is computed in fp32 from those rounded inputs,
and the result must agree to bf16 output rounding (rel 1.6e-2 of the output scale).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    """max |a-b| relative to the scale of the reference (floored so analytically-zero results compare absolutely)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-3)).item()


def _worst_elem(got, want):
    """max over elements of |got - want| / (rms(want) + |want|); the denominator is floored at 1e-3 for gradients that are analytically ~0 (one-key softmax)"""
    got, want = got.detach().float(), want.detach().float()
    rms = want.pow(2).mean().sqrt()
    return ((got - want).abs() / (rms + want.abs()).clamp_min(1e-3)).max().item()


def _tol(dtype):
    return 1.6e-2 if dtype == torch.bfloat16 else 2e-4


def test_tr16_probe_pins_lds_transpose_read_semantics(gpu):
    from diffusion_pipe_amd import hip
    src = torch.arange(256, dtype=torch.int16, device=gpu)
    out = torch.full((256,), -1, dtype=torch.int16, device=gpu)
    hip.check(hip.lib().dpipe_tr16_probe(hip.ptr(src), hip.ptr(out), hip.stream()))
    torch.cuda.synchronize()
    lane = torch.arange(64).repeat_interleave(4)
    j = torch.arange(4).repeat(64)
    expect = (64 * (lane // 16) + 16 * j + (lane % 16)).to(torch.int16)
    got = out.cpu()
    assert torch.equal(got, expect), f'ds_read_b64_tr_b16 mapping differs:\n{got.view(64, 4)[:20]}'


GEMM_SHAPES = [(128, 128, 64), (256, 384, 192), (100, 72, 40), (77, 130, 64), (33, 1000, 520), (1024, 1280, 1280)]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('trans', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('shape', GEMM_SHAPES)
@pytest.mark.parametrize('tile', [64, 128])
def test_gemm_matches_fp32_matmul(gpu, dtype, trans, shape, tile):
    from diffusion_pipe_amd import ops
    ta, tb = trans
    M, N, K = shape
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(gpu, dtype)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(gpu, dtype)
    ref = (a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())
    out = ops.mm(a, b, ta, tb, tile_hint=tile)
    torch.cuda.synchronize()
    assert out.shape == (M, N)
    assert _rel_err(out, ref) < _tol(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_gemm_bias_act_accumulate_batched(gpu, dtype):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(200, 96, generator=g).to(gpu, dtype)
    w = torch.randn(160, 96, generator=g).to(gpu, dtype)
    bias = torch.randn(160, generator=g).to(gpu, dtype)
    ref = F.gelu(a.float() @ w.float().t() + bias.float(), approximate='tanh')
    out = ops.mm(a, w, False, True, bias=bias, act='gelu_tanh')
    assert _rel_err(out, ref) < _tol(dtype)
    # accumulate into an fp32 / same-dtype buffer
    c0 = torch.randn(200, 160, generator=g).to(gpu, dtype)
    c = c0.clone()
    ops.gemm(a, w, False, True, 200, 160, 96, c, lda=96, ldb=96, ldc=160, accumulate=True)
    assert _rel_err(c, c0.float() + a.float() @ w.float().t()) < _tol(dtype)
    # two-level batch with strides: [B, S, H, D] attention-style operands
    B, S, H, D = 2, 72, 3, 64
    q = torch.randn(B, S, H, D, generator=g).to(gpu, dtype)
    k = torch.randn(B, S, H, D, generator=g).to(gpu, dtype)
    p = torch.zeros(B, H, S, S, device=gpu, dtype=dtype)
    ops.gemm(q, k, False, True, S, S, D, p, lda=H * D, ldb=H * D, ldc=S, batch_outer=B, batch_inner=H,
             stride_a=(S * H * D, D), stride_b=(S * H * D, D), stride_c=(H * S * S, S * S))
    ref = torch.einsum('bqhd,bkhd->bhqk', q.float(), k.float())
    assert _rel_err(p, ref) < _tol(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_linear_autograd(gpu, dtype):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 150, 192, generator=g).to(gpu, dtype).requires_grad_(True)
    w = (torch.randn(320, 192, generator=g) / 14).to(gpu, dtype).requires_grad_(True)
    b = torch.randn(320, generator=g).to(gpu, dtype).requires_grad_(True)
    gy = torch.randn(2, 150, 320, generator=g).to(gpu, dtype)
    y = ops.linear(x, w, b)
    y.backward(gy)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    yr.backward(gy.float())
    tol = _tol(dtype)
    assert _rel_err(y, yr) < tol
    assert _rel_err(x.grad, xr.grad) < tol
    assert _rel_err(w.grad, wr.grad) < tol
    assert _rel_err(b.grad, br.grad) < tol


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('act', ['gelu_tanh', 'gelu_erf', 'silu'])
def test_activations(gpu, dtype, act):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 1000 + 3, generator=g) * 2).to(gpu, dtype).requires_grad_(True)
    gy = torch.randn(37, 1003, generator=g).to(gpu, dtype)
    fn = {'gelu_tanh': ops.gelu_tanh, 'gelu_erf': ops.gelu, 'silu': ops.silu}[act]
    rf = {'gelu_tanh': lambda t: F.gelu(t, approximate='tanh'), 'gelu_erf': F.gelu, 'silu': F.silu}[act]
    y = fn(x)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    yr = rf(xr)
    yr.backward(gy.float())
    assert _rel_err(y, yr) < _tol(dtype)
    assert _rel_err(x.grad, xr.grad) < _tol(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_geglu(gpu, dtype):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 50, 2 * 640, generator=g).to(gpu, dtype).requires_grad_(True)
    gy = torch.randn(3, 50, 640, generator=g).to(gpu, dtype)
    y = ops.geglu(x)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    h, gate = xr.chunk(2, dim=-1)
    yr = h * F.gelu(gate)
    yr.backward(gy.float())
    assert _rel_err(y, yr) < _tol(dtype)
    assert _rel_err(x.grad, xr.grad) < _tol(dtype)


@pytest.mark.parametrize('dtype,gdtype', [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
def test_gated_residual(gpu, dtype, gdtype):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(6)
    B, S, D = 2, 700, 256
    x = torch.randn(B, S, D, generator=g).to(gpu, dtype).requires_grad_(True)
    y = torch.randn(B, S, D, generator=g).to(gpu, dtype).requires_grad_(True)
    gate = torch.randn(B, 1, D, generator=g).to(gpu, gdtype).requires_grad_(True)
    go = torch.randn(B, S, D, generator=g).to(gpu, dtype)
    out = ops.gated_residual(x, y, gate)
    out.backward(go)
    xr, yr, gr = (t.detach().float().requires_grad_(True) for t in (x, y, gate))
    outr = xr + yr * gr
    outr.backward(go.float())
    tol = _tol(dtype)
    assert _rel_err(out, outr) < tol
    assert _rel_err(x.grad, xr.grad) < tol
    assert _rel_err(y.grad, yr.grad) < tol
    assert _rel_err(gate.grad, gr.grad) < (3e-2 if gdtype == torch.bfloat16 else 2e-3)
    # plain residual (gate=None)
    out2 = ops.gated_residual(x.detach(), y.detach(), None)
    assert _rel_err(out2, x.detach().float() + y.detach().float()) < tol


@pytest.mark.parametrize('dtype,wdtype', [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
@pytest.mark.parametrize('cols', [128, 1536])
def test_rmsnorm(gpu, dtype, wdtype, cols):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 333, cols, generator=g).to(gpu, dtype).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(gpu, wdtype).requires_grad_(True)
    gy = torch.randn(3, 333, cols, generator=g).to(gpu, dtype)
    y = ops.rms_norm(x, w, 1e-6)
    y.backward(gy)
    xr, wr = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True)
    yr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6) * wr
    yr.backward(gy.float())
    tol = _tol(dtype)
    assert _rel_err(y, yr) < tol
    assert _rel_err(x.grad, xr.grad) < tol
    assert _rel_err(w.grad, wr.grad) < (3e-2 if wdtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize('dtype,wdtype,mdtype', [(torch.bfloat16, torch.bfloat16, torch.float32), (torch.bfloat16, torch.float32, torch.bfloat16),
                                                 (torch.float32, torch.float32, torch.float32)])
@pytest.mark.parametrize('affine,mod', [(False, True), (True, False), (True, True), (False, False)])
def test_layernorm_modulate(gpu, dtype, wdtype, mdtype, affine, mod):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(9)
    B, S, D = 2, 515, 384
    x = torch.randn(B, S, D, generator=g).to(gpu, dtype).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(gpu, wdtype).requires_grad_(True) if affine else None
    beta = (0.1 * torch.randn(D, generator=g)).to(gpu, wdtype).requires_grad_(True) if affine else None
    scale = (0.3 * torch.randn(B, 1, D, generator=g)).to(gpu, mdtype).requires_grad_(True) if mod else None
    shift = (0.3 * torch.randn(B, 1, D, generator=g)).to(gpu, mdtype).requires_grad_(True) if mod else None
    gy = torch.randn(B, S, D, generator=g).to(gpu, dtype)
    y = ops.layer_norm_modulate(x, gamma, beta, scale, shift, 1e-6)
    y.backward(gy)
    f = lambda t: None if t is None else t.detach().float().requires_grad_(True)
    xr, gr, br, sr, hr = f(x), f(gamma), f(beta), f(scale), f(shift)
    n = F.layer_norm(xr, (D,), gr, br, 1e-6)
    yr = n * (1 + sr) + hr if mod else n
    yr.backward(gy.float())
    tol = _tol(dtype)
    assert _rel_err(y, yr) < tol
    assert _rel_err(x.grad, xr.grad) < tol
    if affine:
        assert _rel_err(gamma.grad, gr.grad) < (3e-2 if wdtype == torch.bfloat16 else 2e-3)
        assert _rel_err(beta.grad, br.grad) < (3e-2 if wdtype == torch.bfloat16 else 2e-3)
    if mod:
        assert _rel_err(scale.grad, sr.grad) < (3e-2 if mdtype == torch.bfloat16 else 2e-3)
        assert _rel_err(shift.grad, hr.grad) < (3e-2 if mdtype == torch.bfloat16 else 2e-3)


def test_batched_context_kv_projection_matches_per_block_projections(gpu):
    """Cross attention of several transformer blocks over the same text context: ONE K / V projection GEMM for all blocks + column slices consumed in
    place by the packed flash attention (ops.split_columns, strided K / V views) vs one projection per block -- outputs and every gradient (context,
    all projection weights, queries) agree to bf16 rounding."""
    from diffusion_pipe_amd import nn as dnn, ops
    torch.manual_seed(4)
    B, S, L, C, H, D, NB = 2, 300, 77, 256, 4, 64, 3
    attns = [dnn.Attention(H * D, C, H, D).to(gpu, torch.bfloat16) for _ in range(NB)]
    assert all(a.kv_batchable() for a in attns)
    x = torch.randn(B, S, H * D, device=gpu).to(torch.bfloat16)
    ctx = torch.randn(B, L, C, device=gpu).to(torch.bfloat16)
    gy = torch.randn(B, S, H * D, device=gpu).to(torch.bfloat16)

    def run(batched):
        for a in attns:
            for p in a.parameters():
                p.grad = None
        xi, ci = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
        kvs = [None] * NB
        if batched:
            ws = [w for a in attns for w in (a.to_k.weight, a.to_v.weight)]
            kvs = ops.split_columns(ops.fused_linear(ci, ws, None), [2 * H * D] * NB)
            assert kvs[1].stride(1) == NB * 2 * H * D and kvs[1].data_ptr() != kvs[0].data_ptr()
        h = xi
        for a, kv in zip(attns, kvs):
            h = a(h, ci, residual=h, kv=kv)
        h.backward(gy)
        return h.detach(), xi.grad, ci.grad, [p.grad.clone() for a in attns for p in a.parameters()]

    o0, gx0, gc0, gp0 = run(False)
    o1, gx1, gc1, gp1 = run(True)
    assert _rel_err(o1, o0) < 1.6e-2 and _rel_err(gx1, gx0) < 2e-2 and _rel_err(gc1, gc0) < 2e-2
    for a, b in zip(gp1, gp0):
        assert _rel_err(a, b) < 2e-2


LN_FUSED_CASES = [  # (B, S, D, affine, mod): shapes that take the fused backward (parameter-gradient partials inside the dx pass) and its fall-backs
    (1, 1024, 1280, True, False), (1, 4096, 640, True, False), (1, 77, 1280, True, False), (1, 77, 768, True, False), (2, 1024, 640, True, False),
    (1, 1024, 1536, False, True), (2, 1024, 512, False, True), (2, 515, 384, False, True), (1, 1030, 2048, True, False), (1, 256, 3072, False, True)]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('case', LN_FUSED_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_layernorm_backward_fused_parameter_gradients(gpu, dtype, case):
    """LayerNorm backward at the SDXL / CLIP shapes: dx (+ the bypass gradient) and the column sums (dgamma, dbeta) or (dscale, dshift) from ONE pass over
    x / dy plus the slab sum, per-element bound, twice in a row with fused accumulation into existing .grad buffers (the graph path's form)."""
    from diffusion_pipe_amd import ops
    B, S, D, affine, mod = case
    if dtype == torch.float32 and D > 1024:
        pytest.skip('fp32 rows beyond the register cache take the unfused path (covered by test_layernorm_modulate)')
    g = torch.Generator().manual_seed(B * 31 + S + D)
    x = torch.randn(B, S, D, generator=g).to(gpu, dtype).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(gpu, dtype).requires_grad_(True) if affine else None
    beta = (0.1 * torch.randn(D, generator=g)).to(gpu, dtype).requires_grad_(True) if affine else None
    scale = (0.3 * torch.randn(B, 1, D, generator=g)).to(gpu, dtype).requires_grad_(True) if mod else None
    shift = (0.3 * torch.randn(B, 1, D, generator=g)).to(gpu, dtype).requires_grad_(True) if mod else None
    gy = torch.randn(B, S, D, generator=g).to(gpu, dtype)
    gskip = torch.randn(B, S, D, generator=g).to(gpu, dtype)
    f = lambda t: None if t is None else t.detach().float().requires_grad_(True)
    xr, gr, br, sr, hr = f(x), f(gamma), f(beta), f(scale), f(shift)
    n = F.layer_norm(xr, (D,), gr, br, 1e-5)
    yr = n * (1 + sr) + hr if mod else n
    (yr * gy.float()).sum().backward()
    old = ops.FUSE_GRAD_ACCUM
    try:
        ops.FUSE_GRAD_ACCUM = True
        for _ in range(2):       # the second pass adds into the .grad buffers the first one created
            y, xs = ops.layer_norm_modulate(x, gamma, beta, scale, shift, 1e-5, with_skip=True)
            torch.autograd.backward([y, xs], [gy, gskip])
    finally:
        ops.FUSE_GRAD_ACCUM = old
    rt = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-4

    def close(got, want, what, k=1.0):
        got, want = got.float(), want.float()
        tol = k * rt * (want.pow(2).mean().sqrt().clamp_min(1e-6) + want.abs())
        bad = (got - want).abs() > tol
        assert not bad.any(), f'{what}: {int(bad.sum())} / {bad.numel()} off, worst {(got - want).abs().max().item():.4g}'
    close(x.grad, 2 * (xr.grad + gskip.float()), 'dx')
    # column sums over up to 4 096 rows of bf16-rounded terms, accumulated twice in the parameter dtype: 4 x the per-element bound
    if affine:
        close(gamma.grad, 2 * gr.grad, 'dgamma', 4.0)
        close(beta.grad, 2 * br.grad, 'dbeta', 4.0)
    if mod:
        close(scale.grad, 2 * sr.grad, 'dscale', 4.0)
        close(shift.grad, 2 * hr.grad, 'dshift', 4.0)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('interleaved', [True, False])
def test_rope(gpu, dtype, interleaved):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(10)
    B, S, H, D = 2, 130, 5, 128
    x = torch.randn(B, S, H, D, generator=g).to(gpu, dtype).requires_grad_(True)
    ang = torch.rand(S, D // 2, generator=g) * 6.28
    cos, sin = ang.cos().to(gpu), ang.sin().to(gpu)
    gy = torch.randn(B, S, H, D, generator=g).to(gpu, dtype)
    y = ops.rope(x, cos, sin, interleaved)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    if interleaved:
        xc = torch.view_as_complex(xr.reshape(B, S, H, D // 2, 2))
        fr = torch.polar(torch.ones_like(ang), ang).to(gpu).view(1, S, 1, D // 2)
        yr = torch.view_as_real(xc * fr).flatten(3)
    else:
        a, b = xr[..., :D // 2], xr[..., D // 2:]
        c, s = cos.view(1, S, 1, -1), sin.view(1, S, 1, -1)
        yr = torch.cat([a * c - b * s, a * s + b * c], dim=-1)
    yr.backward(gy.float())
    assert _rel_err(y, yr) < _tol(dtype)
    assert _rel_err(x.grad, xr.grad) < _tol(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('per_sample', [False, True])
@pytest.mark.parametrize('kind', ['mse', 'huber', 'smooth_l1'])
def test_fused_loss(gpu, dtype, per_sample, kind):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(12)
    shape = (3, 4, 64, 66)
    out = torch.randn(shape, generator=g).to(gpu, dtype).requires_grad_(True)
    target = torch.randn(shape, generator=g).to(gpu)
    mask = (torch.rand((3, 1, 64, 66), generator=g) > 0.3).float().to(gpu)
    w = torch.rand(3, generator=g).to(gpu) if per_sample else None
    param = 0.7
    loss = ops.fused_loss(out, target, mask, w, per_sample=per_sample, kind=kind, param=param)
    (loss * 1.5).backward()
    o = out.detach().float().requires_grad_(True)
    if kind == 'mse':
        el = F.mse_loss(o, target, reduction='none')
    elif kind == 'huber':
        el = F.huber_loss(o, target, reduction='none', delta=param)
    else:
        el = F.smooth_l1_loss(o, target, reduction='none', beta=param)
    el = el * mask
    ref = (el.mean([1, 2, 3]) * w).mean() if per_sample else el.mean()
    (ref * 1.5).backward()
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-5
    assert _rel_err(out.grad, o.grad) < _tol(dtype)
    # no mask, empty-mask tensor behaves as "no mask" (utils/dataset.py:1277-1279 turns None into an empty tensor)
    l2 = ops.fused_loss(out.detach(), target, torch.tensor([], device=gpu), None, per_sample=False, kind='mse')
    assert abs(l2.item() - F.mse_loss(out.detach().float(), target).item()) < 1e-5 * l2.item() + 1e-7


def test_grad_norm_and_clip(gpu):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(13)
    grads = [torch.randn(n, generator=g).to(gpu, dt) for n, dt in
             [(70001, torch.bfloat16), (5, torch.bfloat16), (1 << 17, torch.float32), (333, torch.float32), (1280 * 1280, torch.bfloat16)]]
    ref_ss = sum(x.float().norm(2).square() for x in grads)
    ss = ops.grads_sumsq(grads)
    assert abs(ss.item() - ref_ss.item()) / ref_ss.item() < 1e-5
    max_norm = 1.0
    coef = min(1.0, max_norm / (math.sqrt(ref_ss.item()) + 1e-6))
    expect = [x.float() * coef for x in grads]
    ops.grads_clip_scale_(grads, ss, max_norm)
    for x, e in zip(grads, expect):
        assert _rel_err(x, e) < _tol(x.dtype)
    # norm below the threshold: gradients untouched bit for bit
    small = [torch.full((1000,), 1e-4, device=gpu, dtype=torch.bfloat16)]
    before = small[0].clone()
    ops.grads_clip_scale_(small, ops.grads_sumsq(small), 1.0)
    assert torch.equal(small[0], before)


def test_sinusoidal_and_flow_match(gpu):
    from diffusion_pipe_amd import ops
    t = torch.tensor([0.0, 17.0, 999.0], device=gpu)
    e = ops.sinusoidal_embedding(t, 256)
    half = 128
    sinusoid = torch.outer(t.float(), torch.pow(10000, -torch.arange(half, device=gpu).float().div(half)))
    ref = torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)
    assert (e - ref).abs().max().item() < 2e-3      # fp32 trig of arguments up to 1e3
    g = torch.Generator().manual_seed(2)
    x1, x0 = torch.randn(2, 16, 8, 8, generator=g).to(gpu), torch.randn(2, 16, 8, 8, generator=g).to(gpu)
    tt = torch.tensor([0.25, 0.9], device=gpu)
    xt, tgt = ops.flow_match_prep(x1, x0, tt)
    tv = tt.view(-1, 1, 1, 1)
    assert torch.allclose(xt, (1 - tv) * x1 + tv * x0, atol=1e-6)
    assert torch.allclose(tgt, x0 - x1, atol=1e-6)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_unfused_attention_path(gpu, dtype):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(14)
    B, Sq, Sk, H, D = 2, 200, 77, 3, 64
    q = torch.randn(B, Sq, H, D, generator=g).to(gpu, dtype).requires_grad_(True)
    k = torch.randn(B, Sk, H, D, generator=g).to(gpu, dtype).requires_grad_(True)
    v = torch.randn(B, Sk, H, D, generator=g).to(gpu, dtype).requires_grad_(True)
    go = torch.randn(B, Sq, H, D, generator=g).to(gpu, dtype)
    o = ops.attention(q, k, v, impl='unfused')
    o.backward(go)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    orf = F.scaled_dot_product_attention(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2)).transpose(1, 2)
    orf.backward(go.float())
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-4
    assert _rel_err(o, orf) < tol
    assert _rel_err(q.grad, qr.grad) < tol
    assert _rel_err(k.grad, kr.grad) < tol
    assert _rel_err(v.grad, vr.grad) < tol


def _sdpa_ref(q, k, v, kv_len=None):
    """fp32 reference on [B, S, H, D] tensors with optional per-batch key-length mask."""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    mask = None
    if kv_len is not None:
        mask = (torch.arange(Sk, device=q.device)[None, :] < kv_len[:, None].to(q.device)).view(B, 1, 1, Sk)
    return F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask).transpose(1, 2)


ATTN_CASES = [  # (B, Sq, Sk, H, D, kv_len)
    (2, 200, 200, 3, 64, None), (2, 200, 200, 3, 128, None), (1, 300, 77, 5, 64, None), (2, 130, 96, 2, 128, [77, 50]),
    (1, 1024, 1024, 4, 128, None), (1, 64, 1, 2, 64, None), (1, 33, 513, 2, 128, None),
    # SDXL cross attention (one key block per head -> query-split dK/dV + reduce), many-head self attention (128-query
    # workgroups), ragged query-slice counts
    (1, 1024, 77, 20, 64, None), (1, 4096, 77, 10, 64, None), (2, 1000, 100, 3, 128, [77, 100]), (1, 1024, 1024, 20, 64, None),
    (3, 700, 77, 30, 64, None),
    # enough 256-row workgroups (>= 192) for the 8-wave long-sequence kernels (forward LDS-DMA ring, head-dim-128 dQ), ragged tails and key masks
    (1, 3100, 3100, 16, 128, None), (2, 1601, 900, 28, 64, [900, 333]), (2, 1700, 700, 16, 128, [650, 700]),
    (2, 1300, 1700, 14, 128, [1650, 1700])]   # >= 192 256-key workgroups: dV / dK as two 8-wave kernels


@pytest.mark.parametrize('case', ATTN_CASES)
def test_flash_attention_fwd_bwd(gpu, case):
    from diffusion_pipe_amd import ops
    B, Sq, Sk, H, D, kvl = case
    g = torch.Generator().manual_seed(B * 1000 + Sq + Sk + D)
    dt = torch.bfloat16
    q = torch.randn(B, Sq, H, D, generator=g).to(gpu, dt).requires_grad_(True)
    k = torch.randn(B, Sk, H, D, generator=g).to(gpu, dt).requires_grad_(True)
    v = torch.randn(B, Sk, H, D, generator=g).to(gpu, dt).requires_grad_(True)
    go = torch.randn(B, Sq, H, D, generator=g).to(gpu, dt)
    kv_len = torch.tensor(kvl, dtype=torch.int32, device=gpu) if kvl is not None else None
    o = ops.attention(q, k, v, kv_len=kv_len, impl='flash')
    o.backward(go)
    torch.cuda.synchronize()
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    orf = _sdpa_ref(qr, kr, vr, kv_len)
    orf.backward(go.float())
    # bf16 P / dS operands inside the kernel: 2e-2 of the output scale
    assert _rel_err(o, orf) < 2e-2
    assert _rel_err(q.grad, qr.grad) < 3e-2
    assert _rel_err(k.grad, kr.grad) < 3e-2
    assert _rel_err(v.grad, vr.grad) < 3e-2
    # ... and PER ELEMENT (a max-abs over the global max hides errors in small outputs): |err| / (rms(ref) + |ref|), the bounds of
    # tests/test_gpu_attention_long.py (3 x the observed worst element there: output 0.0085, gradients 0.017)
    assert _worst_elem(o, orf) < 0.025, _worst_elem(o, orf)
    for got, want in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)):
        assert _worst_elem(got, want) < 0.05, _worst_elem(got, want)
    if kvl is not None:   # masked keys receive exactly zero gradient
        for bi, n in enumerate(kvl):
            if n < Sk:
                assert k.grad[bi, n:].abs().max().item() == 0.0 and v.grad[bi, n:].abs().max().item() == 0.0


@pytest.fixture
def staged_attention_kernels():
    """Select the register-staged attention kernels (the fallback when K / V extents exceed the 32-bit buffer offsets of the LDS-DMA kernels) through the
    C ABI's dpipe_set_option; unset again afterwards."""
    from diffusion_pipe_amd import hip
    ids = (hip.OPT_ATTN_FWD_DMA, hip.OPT_ATTN_BWD_DMA)
    for i in ids:
        hip.check(hip.lib().dpipe_set_option(i, 0), 'set_option')
    assert hip.lib().dpipe_get_option(hip.OPT_ATTN_FWD_DMA) == 0
    yield
    for i in ids:
        hip.check(hip.lib().dpipe_set_option(i, -1), 'set_option')


@pytest.mark.parametrize('case', [ATTN_CASES[0], ATTN_CASES[1], ATTN_CASES[3], ATTN_CASES[7], ATTN_CASES[9], ATTN_CASES[12], ATTN_CASES[15]])
def test_flash_attention_register_staged_fallback(gpu, staged_attention_kernels, case):
    test_flash_attention_fwd_bwd(gpu, case)


@pytest.mark.parametrize('dma', [0, 1])
def test_flash_attention_kernel_families_agree(gpu, dma):
    """Same inputs through both kernel families: outputs and gradients within bf16 rounding of each other (and of the fp32 reference, checked above)."""
    from diffusion_pipe_amd import hip, ops
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(2, 520, 6, 128, generator=g).to(gpu, torch.bfloat16).requires_grad_(True) for _ in range(3))
    go = torch.randn(2, 520, 6, 128, generator=g).to(gpu, torch.bfloat16)
    try:
        for i in (hip.OPT_ATTN_FWD_DMA, hip.OPT_ATTN_BWD_DMA):
            hip.check(hip.lib().dpipe_set_option(i, dma), 'set_option')
        o = ops.attention(q, k, v, impl='flash', causal=True)
        o.backward(go)
    finally:
        for i in (hip.OPT_ATTN_FWD_DMA, hip.OPT_ATTN_BWD_DMA):
            hip.check(hip.lib().dpipe_set_option(i, -1), 'set_option')
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    orf = F.scaled_dot_product_attention(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2), is_causal=True).transpose(1, 2)
    orf.backward(go.float())
    assert _rel_err(o, orf) < 2e-2
    for got, want in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)):
        assert _rel_err(got, want) < 3e-2


def test_flash_attention_strided_views_and_rescale_spike(gpu):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(77)
    B, S, H, D = 1, 260, 2, 64
    qkv = torch.randn(B, S, 3, H, D, generator=g).to(gpu, torch.bfloat16)
    # a key late in the sequence that dominates one query row forces the online-softmax rescale branch
    qkv[0, 5, 0] = 4.0
    qkv[0, 200, 1] = 4.0
    qkv = qkv.requires_grad_(True)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]      # strided views, head dim contiguous
    go = torch.randn(B, S, H, D, generator=g).to(gpu, torch.bfloat16)
    o = ops.attention(q, k, v, impl='flash')
    o.backward(go)
    ref = qkv.detach().float().requires_grad_(True)
    orf = _sdpa_ref(ref[:, :, 0], ref[:, :, 1], ref[:, :, 2])
    orf.backward(go.float())
    assert _rel_err(o, orf) < 2e-2
    assert _rel_err(qkv.grad, ref.grad) < 3e-2


def test_flash_matches_unfused_hip_path(gpu):
    """The two HIP attention implementations (flash MFMA kernel, GEMM+softmax kernels) agree with each other."""
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(21)
    q, k, v = (torch.randn(1, 320, 4, 128, generator=g).to(gpu, torch.bfloat16) for _ in range(3))
    a = ops.attention(q, k, v, impl='flash')
    b = ops.attention(q, k, v, impl='unfused')
    assert _rel_err(a, b) < 2e-2


@pytest.mark.parametrize('impl,dtype', [('flash', torch.bfloat16), ('unfused', torch.bfloat16), ('unfused', torch.float32)])
@pytest.mark.parametrize('S', [77, 200])
def test_causal_attention(gpu, impl, dtype, S):
    """Causal masking used by the CLIP text encoders that run inside SDXL's first pipeline layer."""
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(31 + S)
    B, H, D = 2, 3, 64
    q, k, v = (torch.randn(B, S, H, D, generator=g).to(gpu, dtype).requires_grad_(True) for _ in range(3))
    go = torch.randn(B, S, H, D, generator=g).to(gpu, dtype)
    o = ops.attention(q, k, v, impl=impl, causal=True)
    o.backward(go)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    orf = F.scaled_dot_product_attention(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2), is_causal=True).transpose(1, 2)
    orf.backward(go.float())
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-4
    assert _rel_err(o, orf) < tol
    for a, b in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)):
        assert _rel_err(a, b) < tol


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_quick_gelu(gpu, dtype):
    from diffusion_pipe_amd import ops
    x = torch.linspace(-6, 6, 4096).to(gpu, dtype).requires_grad_(True)
    y = ops.quick_gelu(x)
    y.sum().backward()
    xr = x.detach().float().requires_grad_(True)
    yr = xr * torch.sigmoid(1.702 * xr)
    yr.sum().backward()
    assert _rel_err(y, yr) < _tol(dtype) and _rel_err(x.grad, xr.grad) < _tol(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('shape', [(1, 320), (77, 1280), (1024, 1280), (4096, 640), (333, 200)])
def test_column_sum_and_accumulate(gpu, dtype, shape):
    from diffusion_pipe_amd import ops
    rows, cols = shape
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).to(gpu, dtype)
    ref = x.float().sum(0)
    out = ops.column_sum(x)
    assert out.dtype == dtype and _rel_err(out, ref) < _tol(dtype)
    acc0 = torch.randn(cols, generator=g).to(gpu, dtype)
    acc = acc0.clone()
    ops.column_sum(x, out=acc)
    assert _rel_err(acc, acc0.float() + ref) < _tol(dtype)
    # row-strided view (a column block of a wider matrix)
    wide = torch.randn(rows, cols + 64, generator=g).to(gpu, dtype)
    view = wide[:, 32:32 + cols] if dtype == torch.float32 else wide[:, 64:64 + cols]
    assert _rel_err(ops.column_sum(view), view.float().sum(0)) < _tol(dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_fused_gradient_accumulation_matches_autograd_accumulation(gpu, dtype):
    """Two micro-batches through Linear + LayerNorm + RMSNorm: with ops.FUSE_GRAD_ACCUM the second backward adds into the
    existing .grad inside the kernels; the result must equal plain autograd accumulation (up to one bf16 rounding)."""
    from diffusion_pipe_amd import nn as dnn, ops

    def run(fuse):
        torch.manual_seed(3)
        lin, ln, rms = dnn.Linear(192, 320).to(gpu, dtype), dnn.LayerNorm(320).to(gpu, dtype), dnn.RMSNorm(320).to(gpu, dtype)
        ops.FUSE_GRAD_ACCUM = fuse
        try:
            for mb in range(3):
                x = torch.randn(2, 150, 192, generator=torch.Generator().manual_seed(mb)).to(gpu, dtype)
                (rms(ln(lin(x))).float() * torch.linspace(-1, 1, 320, device=gpu)).sum().backward()
        finally:
            ops.FUSE_GRAD_ACCUM = False
        return [p.grad.clone() for m in (lin, ln, rms) for p in m.parameters()]

    for a, b in zip(run(True), run(False)):
        assert _rel_err(a, b) < (2e-2 if dtype == torch.bfloat16 else 1e-4)


@pytest.mark.parametrize('cross', [False, True])
@pytest.mark.parametrize('fuse_accum', [False, True])
def test_fused_qkv_projection_and_packed_attention_match_separate_path(gpu, cross, fuse_accum):
    """nn.Attention with fused QKV (self) / KV (cross) GEMMs + packed flash attention vs three separate Linear layers +
    plain flash attention on the same weights: outputs, input gradients and every parameter gradient over 3 micro-batches."""
    from diffusion_pipe_amd import nn as dnn, ops

    def run(fused):
        torch.manual_seed(9)
        attn = dnn.Attention(640, 2048 if cross else None, heads=10, dim_head=64).to(gpu, torch.bfloat16)
        attn.fuse_projections = fused
        ops.FUSE_GRAD_ACCUM = fuse_accum
        outs = []
        try:
            for mb in range(3):
                g = torch.Generator().manual_seed(mb)
                x = torch.randn(2, 200, 640, generator=g).to(gpu, torch.bfloat16).requires_grad_(True)
                c = torch.randn(2, 77, 2048, generator=g).to(gpu, torch.bfloat16).requires_grad_(True) if cross else None
                y = attn(x, c)
                (y.float() * torch.linspace(-1, 1, 640, device=gpu)).sum().backward()
                outs += [y.detach(), x.grad] + ([c.grad] if cross else [])
        finally:
            ops.FUSE_GRAD_ACCUM = False
        return outs + [p.grad.clone() for p in attn.parameters()]

    for a, b in zip(run(True), run(False)):
        assert a.shape == b.shape
        assert _rel_err(a, b) < 2e-2


def test_clip_attention_fused_causal_matches_reference(gpu):
    from diffusion_pipe_amd.workloads import sdxl
    c = sdxl.CLIPConfig(hidden=768, layers=1, heads=12, mlp=3072)
    torch.manual_seed(3)
    attn = sdxl.CLIPAttention(c).to(gpu, torch.bfloat16)
    x = torch.randn(2, 77, 768, generator=torch.Generator().manual_seed(1)).to(gpu, torch.bfloat16).requires_grad_(True)
    y = attn(x)
    y.float().square().mean().backward()
    xr = x.detach().float().requires_grad_(True)
    w = {n: p.detach().float() for n, p in attn.named_parameters()}
    q, k, v = (F.linear(xr, w[f'{n}_proj.weight'], w[f'{n}_proj.bias']).view(2, 77, 12, 64).transpose(1, 2) for n in 'qkv')
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(2, 77, 768)
    yr = F.linear(o, w['out_proj.weight'], w['out_proj.bias'])
    yr.square().mean().backward()
    assert _rel_err(y, yr) < 3e-2 and _rel_err(x.grad, xr.grad) < 4e-2


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('tokens', [150, 4096])
def test_linear_residual_epilogue_and_fused_bias_gradient(gpu, dtype, tokens):
    """y = x W^T + b + r in one GEMM; backward: db rides the wgrad GEMM (bf16: column sums of the A fragments, through the
    split-K slabs when the token count makes the wgrad split), dr = dy.  Two micro-batches with in-kernel accumulation."""
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(21 + tokens)
    w = (torch.randn(320, 192, generator=g) / 14).to(gpu, dtype).requires_grad_(True)
    b = torch.randn(320, generator=g).to(gpu, dtype).requires_grad_(True)
    wr, br = (t.detach().float().requires_grad_(True) for t in (w, b))
    ops.FUSE_GRAD_ACCUM = ops.FUSE_BIAS_GRAD = True
    try:
        for mb in range(2):
            x = torch.randn(1, tokens, 192, generator=g).to(gpu, dtype).requires_grad_(True)
            r = torch.randn(1, tokens, 320, generator=g).to(gpu, dtype).requires_grad_(True)
            gy = (torch.randn(1, tokens, 320, generator=g) / 8).to(gpu, dtype)
            y = ops.linear(x, w, b, r)
            y.backward(gy)
            xr, rr = (t.detach().float().requires_grad_(True) for t in (x, r))
            yr = F.linear(xr, wr, br) + rr
            yr.backward(gy.float())
            tol = _tol(dtype)
            assert _rel_err(y, yr) < tol and _rel_err(x.grad, xr.grad) < tol and _rel_err(r.grad, rr.grad) < 1e-6
    finally:
        ops.FUSE_GRAD_ACCUM = ops.FUSE_BIAS_GRAD = False
    assert _rel_err(w.grad, wr.grad) < _tol(dtype) * 1.5
    assert _rel_err(b.grad, br.grad) < _tol(dtype) * 1.5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('act', [None, 'silu'])
@pytest.mark.parametrize('shape', [(2, 64, 16, 16), (1, 320, 32, 32), (1, 1280, 32, 32), (1, 960, 64, 64), (3, 32, 8, 8)])
def test_group_norm_fused_silu(gpu, dtype, act, shape):
    """nn.GroupNorm(32, C) (+ SiLU) forward / backward on NCHW vs PyTorch fp32, incl. in-kernel parameter-gradient accumulation."""
    from diffusion_pipe_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).to(gpu, dtype).requires_grad_(True)
    w = (torch.rand(C, generator=g) + 0.5).to(gpu, dtype).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.3).to(gpu, dtype).requires_grad_(True)
    gy = torch.randn(N, C, H, W, generator=g).to(gpu, dtype)
    y = ops.group_norm(x, 32, w, b, 1e-5, act)
    y.backward(gy)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    yr = F.group_norm(xr, 32, wr, br, 1e-5)
    if act == 'silu':
        yr = F.silu(yr)
    yr.backward(gy.float())
    tol = _tol(dtype)
    assert _rel_err(y, yr) < tol and _rel_err(x.grad, xr.grad) < tol * 1.5
    assert _rel_err(w.grad, wr.grad) < tol * 1.5 and _rel_err(b.grad, br.grad) < tol * 1.5
    # second micro-batch accumulates into the existing .grad inside the kernel
    ops.FUSE_GRAD_ACCUM = True
    try:
        ops.group_norm(x, 32, w, b, 1e-5, act).backward(gy)
    finally:
        ops.FUSE_GRAD_ACCUM = False
    assert _rel_err(w.grad, 2 * wr.grad) < tol * 2 and _rel_err(b.grad, 2 * br.grad) < tol * 2


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_norms_fold_the_bypass_gradient_into_their_backward(gpu, dtype, tol):
    """`with_skip`: y, x' = norm(x); out = f(y) + g(x').  x has one autograd consumer (the norm node) and the bypass gradient is
    added inside the dx kernel (dpipe_lnmod_bwd gx_add / dpipe_groupnorm_bwd dx_add) -- same x.grad as the plain two-consumer graph."""
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(3)
    # LayerNorm + AdaLN modulation, [B, S, C]
    x = torch.randn(2, 70, 128, generator=g)
    gamma, beta = torch.randn(128, generator=g), torch.randn(128, generator=g)
    scale, shift = torch.randn(2, 128, generator=g) * 0.1, torch.randn(2, 128, generator=g) * 0.1
    wy, ws = torch.randn(2, 70, 128, generator=g), torch.randn(2, 70, 128, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (128,), gamma, beta, 1e-5) * (1 + scale[:, None]) + shift[:, None]
    ((yr * wy).sum() + (xr * xr * ws).sum()).backward()
    xg = x.to(gpu, dtype).requires_grad_(True)
    y, xs = ops.layer_norm_modulate(xg, gamma.to(gpu, dtype), beta.to(gpu, dtype), scale.to(gpu, dtype), shift.to(gpu, dtype), 1e-5, with_skip=True)
    assert xs.data_ptr() == xg.data_ptr()
    ((y.float() * wy.to(gpu)).sum() + (xs.float() ** 2 * ws.to(gpu)).sum()).backward()
    assert _rel_err(y, yr) < tol and _rel_err(xg.grad, xr.grad) < tol
    # only the bypass is used: the gradient passes through untouched
    xg2 = x.to(gpu, dtype).requires_grad_(True)
    _, xs2 = ops.layer_norm_modulate(xg2, None, None, None, None, 1e-5, with_skip=True)
    (xs2.float() * ws.to(gpu)).sum().backward()
    assert _rel_err(xg2.grad, ws) < tol
    # GroupNorm + SiLU, NCHW
    x = torch.randn(2, 64, 12, 16, generator=g)
    gw, gb = torch.randn(64, generator=g), torch.randn(64, generator=g)
    wy, ws = torch.randn(2, 64, 12, 16, generator=g), torch.randn(2, 64, 12, 16, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.silu(torch.nn.functional.group_norm(xr, 8, gw, gb, 1e-5))
    ((yr * wy).sum() + (xr * ws).sum()).backward()
    xg = x.to(gpu, dtype).requires_grad_(True)
    y, xs = ops.group_norm(xg, 8, gw.to(gpu, dtype), gb.to(gpu, dtype), 1e-5, 'silu', with_skip=True)
    ((y.float() * wy.to(gpu)).sum() + (xs.float() * ws.to(gpu)).sum()).backward()
    assert _rel_err(y, yr) < tol and _rel_err(xg.grad, xr.grad) < tol


# ------------------------------------------------------------------------------------------- K2 + K3 fused: RMSNorm -> RoPE in one pass
def _norm_rope_reference(x, w, cos, sin, eps, per_head, token_offset, rope_tokens):
    """float64: rope(rms_norm(x) * w) with interleaved pairs, norm over D (per head) or H * D (whole token), rotation of tokens < rope_tokens only"""
    B, S, H, D = x.shape
    x = x.double()
    if per_head:
        n = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w.double().view(1, 1, 1, D)
    else:
        flat = x.reshape(B, S, H * D)
        n = (flat * torch.rsqrt(flat.pow(2).mean(-1, keepdim=True) + eps) * w.double().view(1, 1, H * D)).view(B, S, H, D)
    c = cos.double()[token_offset:token_offset + S].view(1, S, 1, D // 2)
    s = sin.double()[token_offset:token_offset + S].view(1, S, 1, D // 2)
    a, b = n[..., 0::2], n[..., 1::2]
    rot = torch.stack([a * c - b * s, a * s + b * c], dim=-1).reshape(B, S, H, D)
    rt = S if rope_tokens is None else rope_tokens
    keep = (torch.arange(S, device=x.device) < rt).view(1, S, 1, 1)
    return torch.where(keep, rot, n)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('case', ['wan_full_row', 'flux_per_head_offset', 'hv_partial_rope_strided', 'tiny_head64'])
def test_rms_norm_rope_fused_matches_float64_and_the_two_kernel_route(gpu, dtype, case):
    """ops.rms_norm_rope (csrc/norm.hip rmsnorm_rope_*: SURVEY 2b's "RMSNorm -> RoPE -> Q/K write-out in one pass") -- forward, input gradient and weight gradient
    against a float64 reference; and against the unfused composition (rms_norm, rope [, slice + concatenate]) it replaces.  Cases: Wan (norm over the whole token, weight
    [H D]); Flux (per-head norm, image tokens at an offset of the [text ; image] tables); HunyuanVideo's single-stream form (strided q out of a fused QKV projection,
    rotation of the image tokens only); head dim 64."""
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(31)
    B, S, H, D, per_head, off, rt, strided = {'wan_full_row': (2, 150, 5, 128, False, 0, None, False), 'flux_per_head_offset': (1, 200, 3, 128, True, 37, None, True),
                                               'hv_partial_rope_strided': (2, 96, 4, 128, True, 0, 70, True), 'tiny_head64': (1, 33, 2, 64, True, 0, None, False)}[case]
    qkv = torch.randn(B, S, 3 * H * D, generator=g).to(gpu, dtype)
    w = (1.0 + 0.2 * torch.randn(D if per_head else H * D, generator=g)).to(gpu, dtype).requires_grad_(True)
    ang = torch.rand(off + S, D // 2, generator=g) * 6.28
    cos, sin = ang.cos().float().to(gpu), ang.sin().float().to(gpu)
    gy = torch.randn(B, S, H, D, generator=g).to(gpu, dtype)
    src = qkv.clone().requires_grad_(True)
    q = src.view(B, S, 3, H, D).unbind(2)[1] if strided else src.view(B, S, 3, H, D)[:, :, 1].contiguous()
    y = ops.rms_norm_rope(q, w, cos, sin, 1e-6, per_head=per_head, token_offset=off, rope_tokens=rt)
    y.backward(gy)
    torch.cuda.synchronize()
    gq, gw = src.grad.view(B, S, 3, H, D)[:, :, 1].clone(), w.grad.clone()
    assert src.grad.view(B, S, 3, H, D)[:, :, 0].abs().max().item() == 0                      # the other slices of the projection receive nothing
    # float64 reference
    xd = qkv.view(B, S, 3, H, D)[:, :, 1].double().requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    yd = _norm_rope_reference(xd, wd, cos, sin, 1e-6, per_head, off, rt)
    yd.backward(gy.double())
    tol = 2.5e-2 if dtype == torch.bfloat16 else 2e-5
    assert _worst_elem(y, yd) < tol, _worst_elem(y, yd)
    assert _worst_elem(gq, xd.grad) < tol * 2, _worst_elem(gq, xd.grad)
    assert _rel_err(gw, wd.grad) < tol, _rel_err(gw, wd.grad)
    # the two-kernel route it replaces
    old, ops.FUSE_NORM_ROPE = ops.FUSE_NORM_ROPE, False
    try:
        src2 = qkv.clone().requires_grad_(True)
        w2 = w.detach().clone().requires_grad_(True)
        q2 = src2.view(B, S, 3, H, D).unbind(2)[1]
        y2 = ops.rms_norm_rope(q2, w2, cos, sin, 1e-6, per_head=per_head, token_offset=off, rope_tokens=rt)
        y2.backward(gy)
    finally:
        ops.FUSE_NORM_ROPE = old
    torch.cuda.synchronize()
    assert _worst_elem(y, y2) < tol and _rel_err(gw, w2.grad) < tol
    assert _worst_elem(gq, src2.grad.view(B, S, 3, H, D)[:, :, 1]) < tol * 2


@pytest.mark.parametrize('k_in', [72, 20])
def test_linear_backward_under_store_mode_stores_once_even_when_the_fused_launch_is_refused(gpu, k_in):
    """`_LinearFn.backward` asked `ops._acc` for the weight / bias buffers while it BUILT the grouped dgrad + wgrad launch; when that launch is refused
    (rc -2: in_features % 8 != 0 is not eligible for the LDS-DMA kernel with fused column sums) the unfused fall-back asked again, got "accumulate" and added into the
    stale buffer inside a first-micro-batch (store) graph.  The flag is now drawn once per target: with GRAD_STORE on, pre-filled .grad buffers must come back holding
    exactly this micro-batch's gradient -- k_in = 72 takes the grouped launch, k_in = 20 the refused one."""
    from diffusion_pipe_amd import ops
    torch.manual_seed(3)
    lin = torch.nn.Linear(k_in, 48).to(gpu, torch.bfloat16)
    x = torch.randn(64, k_in, device=gpu, dtype=torch.bfloat16, requires_grad=True)
    gy = torch.randn(64, 48, device=gpu, dtype=torch.bfloat16)
    want_w = gy.float().t() @ x.detach().float()
    want_b = gy.float().sum(0)
    old_fuse, old_store = ops.FUSE_GRAD_ACCUM, ops.GRAD_STORE
    try:
        ops.FUSE_GRAD_ACCUM, ops.GRAD_STORE = True, {}
        lin.weight.grad = torch.full_like(lin.weight, 1000.0)        # what a previous optimizer step left in the lane's accumulators
        lin.bias.grad = torch.full_like(lin.bias, 1000.0)
        ops.linear(x, lin.weight, lin.bias).backward(gy)
        assert lin.weight.grad.data_ptr() in ops.GRAD_STORE and lin.bias.grad.data_ptr() in ops.GRAD_STORE
        assert _rel_err(lin.weight.grad, want_w) < _tol(torch.bfloat16), 'stale accumulator leaked into the stored weight gradient'
        assert _rel_err(lin.bias.grad, want_b) < _tol(torch.bfloat16), 'stale accumulator leaked into the stored bias gradient'
        x.grad = None
        ops.linear(x, lin.weight, lin.bias).backward(gy)              # second micro-batch inside the same graph: accumulates
        assert _rel_err(lin.weight.grad, 2 * want_w) < _tol(torch.bfloat16)
        assert _rel_err(lin.bias.grad, 2 * want_b) < _tol(torch.bfloat16)
    finally:
        ops.FUSE_GRAD_ACCUM, ops.GRAD_STORE = old_fuse, old_store
