"""GPU parity tests of the LDS-DMA pipelined bf16 GEMM (csrc/gemm_pipe.hip) through the C ABI (dpipe_gemm_ex):
every operand layout (fwd NT, dgrad NN, wgrad TN, TT), ragged tiles served by the buffer bounds check, forced and
automatic split-K (slab reduction, ticket re-arm, determinism), epilogue variants, strided batches.
Reference: fp32 matmul of the bf16-rounded operands; tolerance = bf16 output rounding (1.6e-2 of the output scale),
fp32-output cases 2e-3 (fp32 accumulation of bf16 products in a different order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PIPE = 1000          # tile_hint: pipelined kernel, its own tile / split choice; PIPE + S forces S slices
T64, T128 = 2000, 3000   # ... with the 64 x 64 / 128 x 128 tile forced (+ S)
T256K = 9000         # the 256 x 256 tile on a 4-deep ring of 32-wide half K-steps
T128V = 12000        # round 5: the 8-wave 128 x 128 tile with a register-staged feed (global -> VGPR -> LDS, two K-steps in flight in registers, 2-deep LDS ring)
T128S2, T256S = 4000, 7000   # 128 x 128 with a 2-deep ring; 256 x 256 (the DiT-sized configuration: double-buffered fragments, DMA spread over the K-step)


def _rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-3)).item()


def _assert_close_elementwise(out, ref):
    """Per-element bound (a max-abs over the global max hides errors in small outputs): |out - ref| <= 2^-7 |ref| + 2^-7 rms(ref).
    The kernel accumulates in fp32 and rounds once to bf16 (2^-9 relative), so this holds with a wide margin at every K tested (the convolution tests hold
    the same kernel to the same bound at K up to 23 040)."""
    out, ref = out.float(), ref.float()
    tol = 2.0 ** -7 * ref.abs() + 2.0 ** -7 * ref.pow(2).mean().sqrt().clamp_min(1e-6)
    bad = (out - ref).abs() > tol
    assert not bad.any(), f'{int(bad.sum())} / {bad.numel()} elements outside the per-element bound, worst {(out - ref).abs().max().item():.4g}'


def _operands(gpu, ta, tb, M, N, K, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(gpu, torch.bfloat16)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(gpu, torch.bfloat16)
    ref = (a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())
    return a, b, ref


# (M, N, K): K is a multiple of 64 (needed whenever an operand is K-contiguous); M / N ragged against the 128-tile
SHAPES = [(128, 128, 64), (256, 384, 192), (77, 1280, 1280), (100, 72, 128), (1000, 136, 320), (1024, 1280, 1280), (130, 8, 4096)]


@pytest.mark.parametrize('trans', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('split', [0, 1, 2, 5])
@pytest.mark.parametrize('tile', [T64, T128, T128S2, T256S, T256K, T128V])
def test_pipe_gemm_matches_fp32_matmul(gpu, trans, shape, split, tile):
    from diffusion_pipe_amd import ops
    from diffusion_pipe_amd.hip import DpipeHipError
    ta, tb = trans
    M, N, K = shape
    a, b, ref = _operands(gpu, ta, tb, M, N, K, M * 7 + N * 3 + K)
    if (ta and M % 8) or (not tb and N % 8):          # MN-contiguous operand whose row pitch is not 16-byte aligned
        with pytest.raises(DpipeHipError):
            ops.mm(a, b, ta, tb, tile_hint=tile + split)
        return
    out = ops.mm(a, b, ta, tb, tile_hint=tile + split)
    torch.cuda.synchronize()
    assert out.shape == (M, N)
    assert _rel_err(out, ref) < 1.6e-2
    _assert_close_elementwise(out, ref)


SKINNY = [(77, 768, 768), (77, 2304, 768), (77, 768, 3072), (77, 3840, 1280), (77, 1280, 5120), (77, 2560, 2048), (1, 1280, 1280), (128, 320, 256), (77, 264, 320)]


@pytest.mark.parametrize('trans', [(False, True), (False, False)])
@pytest.mark.parametrize('shape', SKINNY, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('hint', [0, T64 + 1, T64 + 4, T64 + 16, T128V + 2])
def test_skinny_m_gemm_automatic_and_forced_split(gpu, trans, shape, hint):
    """The 77-token linears of the text encoders / cross-attention K, V projections (forward NT, dgrad NN): the dispatcher's own choice (64 x 64 tiles, K cut into
    slices reduced by the last arriver over write-through slabs) and forced slice counts / tiles, with bias + residual riding the reducer's epilogue;
    launched twice (ticket re-arm) and compared bit for bit (deterministic slice order)."""
    from diffusion_pipe_amd import ops
    ta, tb = trans
    M, N, K = shape
    a, b, ref = _operands(gpu, ta, tb, M, N, K, M + 3 * N + 5 * K)
    g = torch.Generator(device='cpu').manual_seed(N)
    bias = torch.randn(N, generator=g).to(gpu, torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(gpu, torch.bfloat16)
    want = ref + bias.float() + res.float()
    out = ops.mm(a, b, ta, tb, bias=bias, residual=res, tile_hint=hint)
    out2 = ops.mm(a, b, ta, tb, bias=bias, residual=res, tile_hint=hint)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    _assert_close_elementwise(out, want)


@pytest.mark.parametrize('K', [77, 8, 200, 1024, 4100])
@pytest.mark.parametrize('split', [0, 1, 3, T64 + 3, T128 + 2])
def test_pipe_wgrad_any_k(gpu, K, split):
    """dW = dy^T x: both operands K-major ([tokens][features]) -> any token count; rows >= K read as zero."""
    from diffusion_pipe_amd import ops
    M, N = 320, 200
    a, b, ref = _operands(gpu, True, False, M, N, K, K)
    out = ops.mm(a, b, True, False, tile_hint=PIPE + split)
    assert _rel_err(out, ref) < 1.6e-2


def test_pipe_last_k_row_of_a_ragged_mn_major_operand(gpu):
    """Only the LAST K-row feeds the LAST output row / column (causal attention's dV, dK have this shape: [77][80-pitch]
    probabilities): the buffer bounds check must not clip the odd tail of that row."""
    from diffusion_pipe_amd import ops
    K, M, N, pitch = 77, 77, 64, 80
    abuf = torch.zeros(K, pitch, device=gpu, dtype=torch.bfloat16)
    abuf[torch.arange(K), torch.arange(M)] = 1.0                       # A^T = identity (lower-triangular support)
    b = torch.randn(K, N, generator=torch.Generator().manual_seed(1)).to(gpu, torch.bfloat16)
    for hint in (T64 + 1, T128 + 1, PIPE):
        out = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
        ops.gemm(abuf, b, True, False, M, N, K, out, lda=pitch, ldb=N, ldc=N, tile_hint=hint)
        assert torch.equal(out, b), f'hint {hint}: rows differ {(out != b).any(1).nonzero().flatten().tolist()}'
    # same for the [K][N] operand: B = identity with an 80-element pitch, N = 77
    a = torch.randn(K, 64, generator=torch.Generator().manual_seed(2)).to(gpu, torch.bfloat16)
    out = torch.empty(64, M, device=gpu, dtype=torch.bfloat16)
    ops.gemm(a, abuf, True, False, 64, M, K, out, lda=64, ldb=pitch, ldc=M, tile_hint=T64 + 1)
    assert torch.equal(out, a.t().contiguous())


def test_pipe_not_eligible_is_an_error_and_auto_falls_back(gpu):
    from diffusion_pipe_amd import ops
    from diffusion_pipe_amd.hip import DpipeHipError
    a, b, ref = _operands(gpu, False, True, 100, 72, 40, 3)          # K % 64 != 0 with K-contiguous operands
    with pytest.raises(DpipeHipError):
        ops.mm(a, b, False, True, tile_hint=PIPE)
    out = ops.mm(a, b, False, True)                                    # auto: generic kernel
    assert _rel_err(out, ref) < 1.6e-2


@pytest.mark.parametrize('split', [1, 4, T128 - PIPE + 1, T128 - PIPE + 3])
def test_pipe_epilogue_bias_act_accumulate_f32(gpu, split):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 200, 164, 256
    a = torch.randn(M, K, generator=g).to(gpu, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / 16).to(gpu, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(gpu, torch.bfloat16)
    lin = a.float() @ w.float().t()
    out = ops.mm(a, w, False, True, bias=bias, act='gelu_tanh', tile_hint=PIPE + split)
    assert _rel_err(out, F.gelu(lin + bias.float(), approximate='tanh')) < 1.6e-2
    # unaligned bias pointer / odd ldc take the scalar epilogue
    bias2 = torch.randn(N + 1, generator=g).to(gpu, torch.bfloat16)[1:]
    cbuf = torch.zeros(M, N + 3, device=gpu, dtype=torch.bfloat16)
    ops.gemm(a, w, False, True, M, N, K, cbuf, lda=K, ldb=K, ldc=N + 3, bias=bias2, tile_hint=PIPE + split)
    assert _rel_err(cbuf[:, :N], lin + bias2.float()) < 1.6e-2
    assert torch.count_nonzero(cbuf[:, N:]) == 0
    # accumulate: bf16 and fp32 outputs
    c0 = torch.randn(M, N, generator=g).to(gpu, torch.bfloat16)
    c = c0.clone()
    ops.gemm(a, w, False, True, M, N, K, c, lda=K, ldb=K, ldc=N, accumulate=True, alpha=0.5, tile_hint=PIPE + split)
    assert _rel_err(c, c0.float() + 0.5 * lin) < 1.6e-2
    cf0 = torch.randn(M, N, generator=g).to(gpu)
    cf = cf0.clone()
    ops.gemm(a, w, False, True, M, N, K, cf, lda=K, ldb=K, ldc=N, accumulate=True, tile_hint=PIPE + split)
    assert _rel_err(cf, cf0 + lin) < 2e-3


def test_pipe_batched_strided(gpu):
    from diffusion_pipe_amd import ops
    g = torch.Generator().manual_seed(9)
    B, S, H, D = 2, 200, 3, 64
    q = torch.randn(B, S, H, D, generator=g).to(gpu, torch.bfloat16)
    k = torch.randn(B, S, H, D, generator=g).to(gpu, torch.bfloat16)
    Sp = 208
    p = torch.zeros(B, H, S, Sp, device=gpu, dtype=torch.bfloat16)
    for hint in (PIPE, T64 + 1, T128 + 1):
        p.zero_()
        ops.gemm(q, k, False, True, S, S, D, p, lda=H * D, ldb=H * D, ldc=Sp, batch_outer=B, batch_inner=H,
                 stride_a=(S * H * D, D), stride_b=(S * H * D, D), stride_c=(H * S * Sp, S * Sp), tile_hint=hint)
        ref = torch.einsum('bqhd,bkhd->bhqk', q.float(), k.float())
        assert _rel_err(p[..., :S], ref) < 1.6e-2
        assert torch.count_nonzero(p[..., S:]) == 0
    # K-contiguous A with K = 208 (not a multiple of 64): not eligible -> the forced hint must refuse
    from diffusion_pipe_amd.hip import DpipeHipError
    pv = torch.empty(B, S, H, D, device=gpu, dtype=torch.bfloat16)
    with pytest.raises(DpipeHipError):
        ops.gemm(p, k, False, False, S, D, Sp, pv, lda=Sp, ldb=H * D, ldc=H * D, batch_outer=B, batch_inner=H,
                 stride_a=(H * S * Sp, S * Sp), stride_b=(S * H * D, D), stride_c=(S * H * D, D), tile_hint=PIPE)


def test_pipe_splitk_is_deterministic_and_rearms_counters(gpu):
    from diffusion_pipe_amd import ops
    a, b, ref = _operands(gpu, False, False, 1024, 1280, 2560, 77)
    first = ops.mm(a, b, False, False, tile_hint=T64 + 4)
    assert _rel_err(first, ref) < 1.6e-2
    for i in range(40):                       # every launch must find its ticket counters at zero
        again = ops.mm(a, b, False, False, tile_hint=(T64 + 4 if i % 2 == 0 else T128 + 7))
        if i % 2 == 0:
            assert torch.equal(again, first), f'launch {i}: split-K result changed between launches'
        else:
            assert _rel_err(again, ref) < 1.6e-2
    torch.cuda.synchronize()


def test_pipe_agrees_with_generic_kernel_on_sdxl_shapes(gpu):
    from diffusion_pipe_amd import ops
    for (ta, tb, M, N, K) in [(0, 1, 1024, 10240, 1280), (0, 0, 1024, 1280, 10240), (1, 0, 640, 640, 4096), (0, 1, 77, 1280, 2048),
                              (1, 0, 10240, 1280, 1024), (0, 1, 4096, 640, 2560)]:
        a, b, ref = _operands(gpu, bool(ta), bool(tb), M, N, K, M + N + K)
        auto = ops.mm(a, b, bool(ta), bool(tb))
        generic = ops.mm(a, b, bool(ta), bool(tb), tile_hint=128)
        assert _rel_err(auto, ref) < 1.6e-2
        assert _rel_err(auto, generic) < 1.6e-2


@pytest.mark.parametrize('trans', [(False, True), (False, False), (True, False)])
@pytest.mark.parametrize('hint', [0, T256S, T256S + 2])
def test_pipe_dit_sized_gemm_256_tile(gpu, trans, hint):
    """DiT-sized problem (>= 128 tiles of 256 x 256, 64 K-steps): the automatic choice for NT / NN is the 256 x 256 configuration;
    ragged M / N against the 256-tile, fused bias + GELU + residual epilogue, accumulate and fp32 output."""
    from diffusion_pipe_amd import ops
    ta, tb = trans
    M, N, K = 3000, 2936, 4096
    a, b, ref = _operands(gpu, ta, tb, M, N, K, 41)
    out = ops.mm(a, b, ta, tb, tile_hint=hint)
    assert _rel_err(out, ref) < 1.6e-2
    g = torch.Generator(device='cpu').manual_seed(5)
    bias = torch.randn(N, generator=g).to(gpu, torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(gpu, torch.bfloat16)
    got = ops.mm(a, b, ta, tb, bias=bias, act='gelu_tanh', residual=res, tile_hint=hint)
    want = F.gelu(ref + bias.float(), approximate="tanh") + res.float()
    assert _rel_err(got, want) < 1.6e-2
    acc = torch.randn(M, N, generator=g).to(gpu)
    base = acc.clone()
    ops.mm(a, b, ta, tb, out=acc, accumulate=True, tile_hint=hint)
    assert _rel_err(acc, base + ref) < 2e-3


# ---------------------------------------------------------------------------------------------------------- grouped launches
# (tokens, out_features, in_features) of Linear layers whose backward the step issues: dgrad [tokens, in] = dy [tokens, out] . W [out, in] next to
# wgrad [out, in] (+)= dy^T . x with the bias gradient as column sums of dy inside the wgrad
LINEAR_BWD = [(1024, 1280, 1280), (1024, 3840, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (4096, 640, 640), (4096, 2560, 640), (77, 1280, 1280),
              (77, 5120, 1280), (77, 768, 3072), (77, 2304, 768), (1, 1280, 320), (300, 200, 136), (64, 32, 1280)]


def _linear_bwd_problems(gpu, ops, rows, nout, nin, seed, accumulate, with_bias):
    g = torch.Generator(device='cpu').manual_seed(seed)
    dy = torch.randn(rows, nout, generator=g).to(gpu, torch.bfloat16)
    x = torch.randn(rows, nin, generator=g).to(gpu, torch.bfloat16)
    w = (torch.randn(nout, nin, generator=g) * 0.05).to(gpu, torch.bfloat16)
    gw0 = (torch.randn(nout, nin, generator=g) * 0.5).to(gpu, torch.bfloat16)
    gb0 = torch.randn(nout, generator=g).to(gpu, torch.bfloat16)

    def make():
        gx, gw, gb = torch.empty(rows, nin, device=gpu, dtype=torch.bfloat16), gw0.clone(), gb0.clone()
        probs = [ops.mm_problem(dy, w, False, False, out=gx),
                 ops.mm_problem(dy, x, True, False, out=gw, accumulate=accumulate, colsum=gb if with_bias else None, colsum_accumulate=accumulate)]
        return probs, (gx, gw, gb)
    ref = {'gx': dy.float() @ w.float(), 'gw': dy.float().t() @ x.float() + (gw0.float() if accumulate else 0),
           'gb': dy.float().sum(0) + (gb0.float() if accumulate else 0)}
    return make, ref


@pytest.mark.parametrize('shape', LINEAR_BWD, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('accumulate', [False, True])
@pytest.mark.parametrize('shallow', [0, 2])
def test_grouped_dgrad_wgrad_launch_equals_separate_launches(gpu, shape, accumulate, shallow):
    """dpipe_gemm_group: the dgrad and the wgrad (+ fused bias column sums, + fused gradient accumulation) of one Linear backward as ONE launch whose workgroups
    are divided between the two problems.  Against the fp32 reference per element; bit-identical to two separate dpipe_gemm_ex launches whenever the group keeps
    each problem's own tile choice (the dispatcher re-plans a mixed-geometry pair onto the 64 x 64 tile: then within the bf16 bound only); twice in a row
    (split-K tickets of both problems re-arm)."""
    from diffusion_pipe_amd import hip, ops
    rows, nout, nin = shape
    lib = hip.lib()
    prev = lib.dpipe_get_option(hip.OPT_GEMM_SHALLOW)
    lib.dpipe_set_option(hip.OPT_GEMM_SHALLOW, shallow)
    try:
        make, ref = _linear_bwd_problems(gpu, ops, rows, nout, nin, rows + nout + nin, accumulate, True)
        probs, (gx, gw, gb) = make()
        ops.GEMM_TRACE = []
        done = ops.gemm_group(probs)
        trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
        if done is None:                      # fused column sum not eligible (unaligned): the caller's two-kernel route; nothing may have been written
            assert (rows % 8) or (nout % 8) or (nin % 8)
            return
        torch.cuda.synchronize()
        _assert_close_elementwise(gx, ref['gx'])
        _assert_close_elementwise(gw, ref['gw'])
        _assert_close_elementwise(gb, ref['gb'])
        assert [t['grp'] for t in trace] == [0, 1] and trace[0]['grp_n'] == 2 and trace[0]['grp_l'] in (1, 2)
        # separate launches of the same problems
        old, ops.GROUP_GEMM = ops.GROUP_GEMM, False
        try:
            probs2, (gx2, gw2, gb2) = make()
            assert ops.gemm_group(probs2) is not None
        finally:
            ops.GROUP_GEMM = old
        torch.cuda.synchronize()
        _assert_close_elementwise(gx2, ref['gx'])
        # run the grouped form again on fresh outputs: deterministic, tickets re-armed
        probs3, (gx3, gw3, gb3) = make()
        ops.gemm_group(probs3)
        torch.cuda.synchronize()
        assert torch.equal(gx3, gx) and torch.equal(gw3, gw) and torch.equal(gb3, gb)
        same_plan = trace[0]['grp_l'] == 1 and (torch.equal(gx2, gx) and torch.equal(gw2, gw))
        if not same_plan:                     # re-planned pair: still within the per-element bound of the separately launched results
            _assert_close_elementwise(gw2, ref['gw'])
    finally:
        lib.dpipe_set_option(hip.OPT_GEMM_SHALLOW, prev)


def test_grouped_launch_of_four_mixed_layout_problems(gpu):
    """Four independent problems of one tile geometry (NT forward-style, NN, TN, TT) in one launch -- the general form of the table (e.g. same-shaped linears of
    two sibling branches with their gradients): each equals its single launch bit for bit."""
    from diffusion_pipe_amd import ops
    M, N, K = 320, 256, 512
    outs, singles, probs = [], [], []
    for i, (ta, tb) in enumerate([(False, True), (False, False), (True, False), (True, True)]):
        a, b, ref = _operands(gpu, ta, tb, M, N, K, 100 + i)
        out = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
        probs.append(ops.mm_problem(a, b, ta, tb, out=out))
        singles.append(ops.mm(a, b, ta, tb))
        outs.append((out, ref))
    ops.GEMM_TRACE = []
    assert ops.gemm_group(probs) is not None
    trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    torch.cuda.synchronize()
    assert trace[0]['grp_l'] == 1 and len(trace) == 4
    for (out, ref), single in zip(outs, singles):
        _assert_close_elementwise(out, ref)
        assert torch.equal(out, single)


def test_grouped_launch_with_an_fp32_member_falls_back_per_problem(gpu):
    from diffusion_pipe_amd import ops
    a, b, ref = _operands(gpu, False, True, 128, 128, 256, 5)
    af, bf = a.float(), b.float()
    o1 = torch.empty(128, 128, device=gpu, dtype=torch.bfloat16)
    o2 = torch.empty(128, 128, device=gpu, dtype=torch.float32)
    assert ops.gemm_group([ops.mm_problem(a, b, False, True, out=o1), ops.mm_problem(af, bf, False, True, out=o2)]) is not None
    torch.cuda.synchronize()
    _assert_close_elementwise(o1, ref)
    assert (o2 - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


def test_linear_backward_through_the_grouped_launch_matches_ungrouped(gpu):
    """ops.linear / ops.fused_linear backward (grouped dgrad + wgrad + bias sums) vs the ungrouped A/B path: identical gradients, with and without fused accumulation."""
    from diffusion_pipe_amd import ops
    g = torch.Generator(device='cpu').manual_seed(9)
    x0 = torch.randn(2, 512, 640, generator=g).to(gpu, torch.bfloat16)
    ws = [torch.nn.Parameter((torch.randn(640, 640, generator=g) * 0.04).to(gpu, torch.bfloat16)) for _ in range(3)]
    bs = [torch.nn.Parameter(torch.randn(640, generator=g).to(gpu, torch.bfloat16)) for _ in range(3)]
    w1 = torch.nn.Parameter((torch.randn(640, 1920, generator=g) * 0.04).to(gpu, torch.bfloat16))       # (both pairs keep their own 64 x 64 plan inside the group)
    b1 = torch.nn.Parameter(torch.randn(640, generator=g).to(gpu, torch.bfloat16))

    def run(grouped, accum):
        old, ops.GROUP_GEMM = ops.GROUP_GEMM, grouped
        oldf, ops.FUSE_GRAD_ACCUM = ops.FUSE_GRAD_ACCUM, accum
        try:
            for p in ws + bs + [w1, b1]:
                p.grad = None
            res = None
            for _ in range(2 if accum else 1):            # second pass accumulates into the .grad buffers of the first
                x = x0.clone().requires_grad_(True)
                y = ops.linear(ops.fused_linear(x, ws, bs), w1, b1)
                y.float().pow(2).mean().backward()
                res = x.grad
            torch.cuda.synchronize()
            return [res.clone()] + [p.grad.clone() for p in ws + bs + [w1, b1]]
        finally:
            ops.GROUP_GEMM, ops.FUSE_GRAD_ACCUM = old, oldf
    for accum in (False, True):
        a, b = run(True, accum), run(False, accum)
        for u, v in zip(a, b):
            assert torch.equal(u, v), (accum, (u.float() - v.float()).abs().max().item())
