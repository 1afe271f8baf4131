"""GPU: the fused step end (csrc/optim.hip through dpipe_adamw_sumsq / dpipe_adamw_step, optim.FusedAdamW) against
torch.optim.AdamW -- the optimizer the reference constructs (train.py:672-678) -- and the reference's clip composition
(utils/patches.py:175-246).  fp32 parameters: 1e-6 relative after 5 steps.  bf16 parameters: each step is compared with the
fp32 update of the same bf16 state rounded once to bf16 (the kernel computes in fp32 and rounds once: <= 1 bf16 ulp)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(7,), (1000, 33), (64, 64), (3, 5, 7), (65537,), (1,)]          # ragged / unaligned tails, a chunk boundary (65536)


def _params(dtype, gpu, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(gpu, dtype)) for s in SHAPES]


def test_fused_adamw_fp32_matches_torch_adamw(gpu):
    from diffusion_pipe_amd.optim import FusedAdamW
    mine = _params(torch.float32, gpu)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in mine]
    kw = dict(lr=3e-3, betas=(0.9, 0.95), eps=1e-8)
    groups = lambda ps: [{'params': ps[:3], 'weight_decay': 0.1}, {'params': ps[3:], 'weight_decay': 0.0, 'lr': 1e-3}]
    opt, ropt = FusedAdamW(groups(mine), **kw), torch.optim.AdamW(groups(ref), foreach=False, **kw)
    g = torch.Generator().manual_seed(1)
    for _ in range(5):
        for p, r in zip(mine, ref):
            r.grad = torch.randn(r.shape, generator=g)
            p.grad = r.grad.to(gpu)
        opt.step()
        ropt.step()
    for p, r in zip(mine, ref):
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-6, atol=1e-7)
        assert torch.allclose(opt.state[p]['exp_avg_sq'].cpu(), ropt.state[r]['exp_avg_sq'], rtol=2e-6, atol=1e-12)
        assert opt.state[p]['step'] == 5.0 and p.grad is not None           # step() leaves the gradients alone
    # optimizer checkpoints interchange with torch.optim.AdamW's
    ropt2 = torch.optim.AdamW(groups([torch.nn.Parameter(p.detach().clone()) for p in mine]), **kw)
    ropt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    assert float(ropt2.state[ropt2.param_groups[0]['params'][0]]['step']) == 5.0


@pytest.mark.parametrize('lanes', [1, 3])
def test_fused_step_end_lanes_clip_zero_bf16(gpu, lanes):
    from diffusion_pipe_amd.optim import FusedAdamW
    ps = _params(torch.bfloat16, gpu, seed=3)
    opt = FusedAdamW([{'params': ps[:2], 'weight_decay': 0.05}, {'params': ps[2:], 'weight_decay': 0.0}], lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    g = torch.Generator().manual_seed(5)
    max_norm = 1.0
    for step in range(1, 4):
        lane_grads = [{id(p): (torch.randn(p.shape, generator=g) * 0.05).to(gpu, torch.bfloat16) for p in ps} for _ in range(lanes)]
        # fp32 reference from the current bf16 state
        before = [(p.detach().float().cpu(), opt.state[p]['exp_avg'].float().cpu() if opt.state[p] else torch.zeros(p.shape),
                   opt.state[p]['exp_avg_sq'].float().cpu() if opt.state[p] else torch.zeros(p.shape)) for p in ps]
        gsum = [sum(lane[id(p)].float().cpu() for lane in lane_grads) for p in ps]
        want_sumsq = sum((x.double() ** 2).sum() for x in gsum)
        coef = min(1.0, max_norm / (want_sumsq.sqrt().item() + 1e-6))
        total = opt.grads_sumsq(lane_grads)
        assert abs(total.item() - want_sumsq.item()) / want_sumsq.item() < 1e-5
        opt.fused_update(lane_grads, total, max_norm, zero_grads=True)
        torch.cuda.synchronize()
        for i, p in enumerate(ps):
            wd = 0.05 if i < 2 else 0.0
            p0, m0, v0 = before[i]
            gr = gsum[i] * coef
            pw = p0 * (1 - 1e-2 * wd)
            m1 = m0 + (1 - 0.9) * (gr - m0)
            v1 = 0.99 * v0 + (1 - 0.99) * gr * gr
            pw = pw - (1e-2 / (1 - 0.9 ** step)) * m1 / (v1.sqrt() / (1 - 0.99 ** step) ** 0.5 + 1e-8)
            # one bf16 ulp (8 significant bits) at the scale of the operands: the clip coefficient comes from an fp32 device
            # reduction (1e-6 relative to the host's), which can flip a rounding, and m / p are differences of nearby terms
            scales = (p0.abs(), torch.maximum(m0.abs(), gr.abs()), torch.maximum(v0, gr * gr))
            for (got, want), scale in zip(((p.detach(), pw), (opt.state[p]['exp_avg'], m1), (opt.state[p]['exp_avg_sq'], v1)), scales):
                got, want_bf = got.float().cpu(), want.to(torch.bfloat16).float()
                ulp = torch.maximum(want.abs(), scale).clamp_min(1e-30) * 2.0 ** -7
                assert ((got - want_bf).abs() <= ulp).all()
                assert (got == want_bf).float().mean() > 0.98                       # fp32 arithmetic, one rounding: almost always identical
            assert all(not lane[id(p)].any() for lane in lane_grads)                 # every lane zeroed in the same pass


def test_engine_lanes_with_fused_step_end_match_torch_adamw_path(gpu):
    """SDXL (tiny) on the hipGraph path with 2 lanes: FusedAdamW (lanes summed / clipped / applied / zeroed by the fused passes) vs
    torch.optim.AdamW (foreach lane sum, clip kernels, torch step): same loss trajectory and gradient norms over 3 steps."""
    from diffusion_pipe_amd import optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    gas = 4
    out = {}
    for use_hip in (True, False):
        work = sdxl.SDXLWorkload(cfg, model_config={'min_snr_gamma': 5.0}, dtype=torch.bfloat16, seed=2)
        work.train_config = {'optimizer': {'type': 'adamw', 'lr': 2e-4, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
        module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
        engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas,
                                                             'gradient_clipping': 0.5, 'hip_graph': True, 'graph_lanes': 2}, device=gpu)
        params = [p for p in module.parameters() if p.requires_grad]
        engine._configure_optimizer(optim.make_optimizer_factory(work.train_config, work, gas, use_hip_adamw=use_hip), params)
        assert isinstance(engine.optimizer, optim.FusedAdamW) == use_hip
        torch.manual_seed(7)
        micro = split_batch(work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=3, ids_len=75)), gas)
        losses, norms = [], []
        for _ in range(3):
            losses.append(engine.train_batch(iter(copy.deepcopy(micro))).item())
            norms.append(engine.get_global_grad_norm().item())
        out[use_hip] = (losses, norms)
    (l1, n1), (l0, n0) = out[True], out[False]
    assert l1[0] == pytest.approx(l0[0], rel=1e-3) and n1[0] == pytest.approx(n0[0], rel=5e-3)      # step 1: identical weights (bf16 run-to-run noise only)
    assert l1[2] < l1[0] and l0[2] < l0[0]                                                         # both trajectories descend
    for a, b in zip(l1 + n1, l0 + n0):
        assert a == pytest.approx(b, rel=3e-2)


def test_fused_adamw_keeps_the_bias_correction_per_parameter(gpu):
    """A parameter that first receives a gradient on a later step uses ITS step count (torch.optim.AdamW semantics), not its group's."""
    from diffusion_pipe_amd.optim import FusedAdamW
    mine = _params(torch.float32, gpu, seed=9)[:3]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in mine]
    kw = dict(lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    opt, ropt = FusedAdamW(mine, **kw), torch.optim.AdamW(ref, foreach=False, **kw)
    g = torch.Generator().manual_seed(2)
    for step in range(4):
        for i, (p, r) in enumerate(zip(mine, ref)):
            late = i == 1 and step < 2                       # parameter 1 joins at step 2
            r.grad = None if late else torch.randn(r.shape, generator=g)
            p.grad = None if late else r.grad.to(gpu)
        opt.step()
        ropt.step()
    assert [opt.state[p]['step'] for p in mine] == [4.0, 2.0, 4.0]
    for p, r in zip(mine, ref):
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-6, atol=1e-7)


def test_fused_adamw_kahan_matches_the_reference_sequence_and_tracks_fp32(gpu):
    """kahan=True (optimizers/generic_optim.py:486-497): shift += update; old = p; p += shift; shift += old - p with bf16 roundings -- op-for-op
    against a torch emulation, and the property it exists for: with updates far below one bf16 ulp the compensated parameters follow the fp32
    trajectory while plain bf16 AdamW stalls."""
    from diffusion_pipe_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shape = (257, 33)
    p0 = (torch.randn(shape) * 2).to(torch.bfloat16)
    kw = dict(lr=1e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0)
    pk = torch.nn.Parameter(p0.clone().to(gpu))
    pp = torch.nn.Parameter(p0.clone().to(gpu))
    pf = torch.nn.Parameter(p0.float())
    ok, op, of = FusedAdamW([pk], kahan=True, **kw), FusedAdamW([pp], **kw), torch.optim.AdamW([pf], foreach=False, **kw)
    rnd = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(1)
    for step in range(1, 41):
        grad = (torch.randn(shape, generator=g) * 0.1 + 0.5).to(torch.bfloat16)
        pk.grad, pp.grad, pf.grad = grad.to(gpu), grad.to(gpu), grad.float()
        ok.step(); op.step(); of.step()
    shift = ok.state[pk]['shift']
    assert shift.dtype == torch.bfloat16 and shift.shape == pk.shape
    true = pf.detach()
    err_kahan = ((pk.detach().float().cpu() + shift.float().cpu()) - true).abs().max().item()
    err_plain = (pp.detach().float().cpu() - true).abs().max().item()
    moved = (true - p0.float()).abs().max().item()
    assert moved < 1e-3 and err_kahan < 0.1 * moved, (moved, err_kahan)          # 40 steps of ~1e-5: the compensated sum keeps them
    assert err_plain > 0.5 * moved, (moved, err_plain)                           # plain bf16 parameters barely move (1 ulp at |p| ~ 2 is 1.6e-2)
    # one step op-for-op: from a fresh state the update and its compensation are exactly the reference's sequence
    q = torch.nn.Parameter(p0.clone().to(gpu))
    oq = FusedAdamW([q], kahan=True, **kw)
    grad = (torch.randn(shape, generator=g)).to(torch.bfloat16)
    q.grad = grad.to(gpu)
    oq.step()
    gf, old = grad.float(), p0.float()
    m1, v1 = (1 - 0.9) * gf, (1 - 0.99) * gf * gf
    new = old - (1e-5 / (1 - 0.9)) * m1 / (v1.sqrt() / (1 - 0.99) ** 0.5 + 1e-8)
    s = rnd(rnd(new - old))
    pn = rnd(old + s)
    sh = rnd(s + rnd(old - pn))
    # (fp32 contraction order may move an update across a bf16 rounding boundary for a handful of elements; the compensation absorbs it)
    same = (q.detach().float().cpu() == pn) & (oq.state[q]['shift'].float().cpu() == sh)
    assert same.float().mean().item() > 0.99
    assert torch.allclose(q.detach().float().cpu() + oq.state[q]['shift'].float().cpu(), pn + sh, rtol=0, atol=2e-7 + 4e-3 * (pn + sh - old).abs().max().item())


@pytest.mark.parametrize('dtype,kahan,n', [(torch.bfloat16, False, 8192), (torch.bfloat16, True, 5000), (torch.float32, False, 4096 + 77), (torch.bfloat16, True, 1000)])
def test_adamw8bit_kernel_matches_the_restated_library_algorithm(gpu, dtype, kahan, n):
    """optim.AdamW8bit (dpipe_adamw8bit_step) vs oracle/adam8bit_ref.py (bitsandbytes' published block-wise 8-bit Adam; the library itself is absent, so this
    pins the kernel to the restatement only).  Every step starts from the kernel's own state copied into the oracle, so each step is compared on identical
    inputs: block absmax 1e-6 relative; codes and parameters identical except where fp32 contraction order moves a value across a rounding / code boundary
    (< 0.5 % of the elements, by one code / one ulp).  n = 1000 < min_8bit_size runs the fp32-moment path.  One gradient element is inf at step 4."""
    import numpy as np
    from diffusion_pipe_amd import optim
    from oracle import adam8bit_ref as ref
    name = 'bf16' if dtype == torch.bfloat16 else 'f32'
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g).to(dtype)
    grads = [(torch.randn(n, generator=g) * (0.5 + i)).to(dtype) for i in range(6)]
    if n >= 4096:
        grads[3][17] = float('inf')          # (the fp32-moment path of small tensors has no non-finite guard, as in the library)
    p = torch.nn.Parameter(p0.clone().to(gpu))
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    opt = optim.AdamW8bit([p], kahan=kahan, **kw)
    pr = [p0.float().numpy().copy()]
    oref = ref.AdamW8bitRef(pr, kahan=kahan, dtype=name, **kw)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -22
    for it, gr in enumerate(grads):
        p.grad = gr.to(gpu)
        opt.step()
        oref.step([gr.float().numpy()])
        torch.cuda.synchronize()
        st, sr = opt.state[p], oref.state[0]
        got_p = p.detach().float().cpu().numpy()
        close = np.abs(got_p - pr[0]) <= ulp * np.abs(pr[0]) + (1e-12 if dtype == torch.bfloat16 else 2.5e-7)      # fp32: the kernel's fused multiply-adds round once
        assert close.all() and (got_p == pr[0]).mean() > (0.995 if dtype == torch.bfloat16 else 0.7), (it, float(np.abs(got_p - pr[0]).max()), float((got_p == pr[0]).mean()))
        if it == 3 and n >= 4096:
            assert np.isfinite(got_p).all()
        if kahan:
            got_s = st['shift'].float().cpu().numpy()
            assert (np.abs(got_s - sr['shift']) <= ulp * np.abs(sr['shift']) + ulp * np.abs(pr[0]) * 1.01 + 1e-12).all()
        if n >= 4096:
            assert st['state1'].dtype == torch.uint8 and st['absmax1'].numel() == -(-n // 256)
            for k in ('absmax1', 'absmax2'):
                np.testing.assert_allclose(st[k].cpu().numpy(), sr[k], rtol=2e-6, atol=0)
            for k, rk in (('state1', 'c1'), ('state2', 'c2')):
                d = np.abs(st[k].cpu().numpy().astype(np.int32) - sr[rk].astype(np.int32))
                assert d.max() <= 1 and (d == 0).mean() > 0.995, (it, k, int(d.max()), float((d == 0).mean()))
            if it == 3:
                assert sr['c1'][17] == st['state1'][17].item()          # the inf gradient reset that element's moments (code of 0)
            # re-synchronise the oracle with the kernel's state
            sr['c1'], sr['c2'] = st['state1'].cpu().numpy().copy(), st['state2'].cpu().numpy().copy()
            sr['absmax1'], sr['absmax2'] = st['absmax1'].cpu().numpy().copy(), st['absmax2'].cpu().numpy().copy()
        else:
            assert st['state1'].dtype == torch.float32
            np.testing.assert_allclose(st['state1'].cpu().numpy(), sr['m'], rtol=1e-5, atol=1e-7)
            sr['m'], sr['v'] = st['state1'].cpu().numpy().copy(), st['state2'].cpu().numpy().copy()
        pr[0][:] = got_p
        if kahan:
            sr['shift'] = st['shift'].float().cpu().numpy().copy()


@pytest.mark.parametrize('kahan', [False, True])
def test_adamw8bit_multi_tensor_launch_equals_the_single_tensor_kernel(gpu, kahan):
    """optim.AdamW8bit.step() = ONE dpipe_adamw8bit_multi launch over every 8-bit tensor of a group (8 elements per thread, half-wave absmax) vs the
    one-tensor kernel dpipe_adamw8bit_step on copies of the same state: ragged sizes (tail lanes, partial quantisation blocks), a channels-last 4-D tensor,
    three steps -- identical up to rare fused-multiply-add rounding flips (>= 99.9 % of codes / parameters bit-equal, the rest one code / one ulp)."""
    from diffusion_pipe_amd import hip, optim
    g = torch.Generator().manual_seed(3)
    shapes = [(8192,), (5000,), (4096 + 77,), (300, 41), (64, 33, 3, 3), (2048 * 5 + 8,)]
    ps = []
    for sh in shapes:
        t = torch.randn(*sh, generator=g).to(torch.bfloat16).to(gpu)
        if len(sh) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t))
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    opt = optim.AdamW8bit(ps, kahan=kahan, **kw)
    q1, q2 = opt._maps(gpu)
    ref = [{'p': p.detach().clone(), 'c1': torch.zeros(p.numel(), dtype=torch.uint8, device=gpu), 'c2': torch.zeros(p.numel(), dtype=torch.uint8, device=gpu),
            'a1': torch.zeros(-(-p.numel() // 256), device=gpu), 'a2': torch.zeros(-(-p.numel() // 256), device=gpu),
            's': torch.zeros_like(p) if kahan else None} for p in ps]
    for step in range(1, 4):
        for p, r in zip(ps, ref):
            gr = (torch.randn(p.shape, generator=g) * step).to(torch.bfloat16).to(gpu)
            if p.dim() == 4:
                gr = gr.contiguous(memory_format=torch.channels_last)
            p.grad = gr
            hip.check(hip.lib().dpipe_adamw8bit_step(hip.ptr(r['p']), hip.ptr(gr), hip.ptr(r['c1']), hip.ptr(r['c2']), hip.ptr(r['a1']), hip.ptr(r['a2']), hip.ptr(q1), hip.ptr(q2),
                                                     hip.ptr(r['s']), p.numel(), kw['lr'], 0.9, 0.99, kw['eps'], kw['weight_decay'], step, 1.0, hip.BF16, hip.stream()), 'single')
        opt.step()
        torch.cuda.synchronize()
        assert len(opt._tables) == 1
        for p, r in zip(ps, ref):
            st = opt.state[p]
            assert st['state1'].dtype == torch.uint8
            same_p = (p.detach().view(torch.int16) == r['p'].view(torch.int16)).float().mean().item()
            d1 = (st['state1'].int() - r['c1'].int()).abs()
            d2 = (st['state2'].int() - r['c2'].int()).abs()
            assert same_p > 0.999 and d1.max().item() <= 1 and d2.max().item() <= 1 and (d1 == 0).float().mean().item() > 0.999, (step, tuple(p.shape), same_p)
            assert torch.allclose(st['absmax1'], r['a1'], rtol=2e-6, atol=0) and torch.allclose(st['absmax2'], r['a2'], rtol=2e-6, atol=0)
            # keep the two trajectories on identical state (a one-code flip would otherwise propagate)
            r['p'].copy_(p.detach()); r['c1'].copy_(st['state1']); r['c2'].copy_(st['state2']); r['a1'].copy_(st['absmax1']); r['a2'].copy_(st['absmax2'])
            if kahan:
                r['s'].copy_(st['shift'])


def test_adamw8bit_trains_like_fp32_adamw(gpu):
    """60 steps on a quadratic with raw bf16 parameters: the 8-bit (Kahan) optimizer reaches the fp32 AdamW trajectory's error level, with 2.03 bytes of
    moment state per parameter."""
    from diffusion_pipe_amd import optim
    g = torch.Generator().manual_seed(3)
    target = torch.randn(3, 4096, generator=g)
    p8 = torch.nn.Parameter(torch.zeros(3, 4096, dtype=torch.bfloat16, device=gpu))
    pf = torch.nn.Parameter(torch.zeros(3, 4096))
    o8 = optim.AdamW8bit([p8], lr=5e-2, betas=(0.9, 0.99), weight_decay=0.0, kahan=True)
    of = torch.optim.AdamW([pf], lr=5e-2, betas=(0.9, 0.99), weight_decay=0.0)
    tg = target.to(gpu)
    for _ in range(60):
        p8.grad = (p8.detach().float() - tg).to(torch.bfloat16)
        pf.grad = pf.detach() - target
        o8.step(); of.step()
    e8 = (p8.detach().float().cpu() - target).abs().mean().item()
    ef = (pf.detach() - target).abs().mean().item()
    assert e8 < ef + 0.03, (e8, ef)
    st = o8.state[p8]
    moment_bytes = st['state1'].numel() + st['state2'].numel() + 4 * (st['absmax1'].numel() + st['absmax2'].numel())
    assert moment_bytes / p8.numel() < 2.04


def test_engine_step_with_the_8bit_optimizer(gpu):
    """SDXL (tiny) on the hipGraph path (2 lanes), optimizer.type = 'adamw8bitkahan' as in the reference's TOML: the engine sums the lanes, clips, then
    AdamW8bit.step() runs the 8-bit kernel on every parameter tensor; the loss descends like the fused bf16 AdamW run and large tensors hold uint8 state."""
    from diffusion_pipe_amd import optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    gas = 4
    out = {}
    for kind in ('adamw8bitkahan', 'adamw'):
        work = sdxl.SDXLWorkload(cfg, model_config={'min_snr_gamma': 5.0}, dtype=torch.bfloat16, seed=2)
        work.train_config = {'optimizer': {'type': kind, 'lr': 2e-4, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
        module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
        engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas,
                                                             'gradient_clipping': 0.5, 'hip_graph': True, 'graph_lanes': 2}, device=gpu)
        params = [p for p in module.parameters() if p.requires_grad]
        engine._configure_optimizer(optim.make_optimizer_factory(work.train_config, work, gas), params)
        torch.manual_seed(7)
        micro = split_batch(work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=3, ids_len=75)), gas)
        out[kind] = [engine.train_batch(iter(copy.deepcopy(micro))).item() for _ in range(4)]
        if kind == 'adamw8bitkahan':
            assert isinstance(engine.optimizer, optim.AdamW8bit)
            states = [(p, engine.optimizer.state[p]) for p in params if len(engine.optimizer.state[p])]       # (parameters that never saw a gradient hold no state)
            assert len(states) > 0.9 * len(params)
            kinds = {st['state1'].dtype for p, st in states if p.numel() >= 4096}
            assert kinds == {torch.uint8} and all('shift' in st for _, st in states)
    l8, lf = out['adamw8bitkahan'], out['adamw']
    assert l8[0] == pytest.approx(lf[0], rel=1e-3)
    assert l8[3] < l8[0] and lf[3] < lf[0]
    for a, b in zip(l8, lf):
        assert a == pytest.approx(b, rel=5e-2)
