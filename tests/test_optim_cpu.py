"""Optimizer / LR-schedule factory and the LoRA wrapping structure -- host logic only (no kernels run)."""
import copy
import math

import pytest
import torch

from diffusion_pipe_amd import nn as dnn
from diffusion_pipe_amd import optim
from diffusion_pipe_amd.workloads import sdxl


def _workload():
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), model_config={'unet_lr': 4e-5, 'text_encoder_2_lr': 1e-6}, dtype=torch.float32)
    work.train_config = {'optimizer': {'type': 'adamw', 'lr': 2e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
    return work


def test_stage_without_trainable_parameters_gets_noop_optimizer():
    work = _workload()
    opt = optim.make_optimizer_factory(work.train_config, work, global_batch_size=4, device_is_gpu=False)([])
    assert isinstance(opt, optim.DummyOptimizer) and opt.param_groups == [] and opt.state_dict() == {}
    opt.step()
    opt.zero_grad()


def test_param_groups_component_lrs_and_weight_decay_split():
    work = _workload()
    params = [p for m in work.modules().values() for p in m.parameters()]
    opt = optim.make_optimizer_factory(work.train_config, work, global_batch_size=4, device_is_gpu=False)(params)
    assert isinstance(opt, torch.optim.AdamW)
    # 3 components x (decay, no-decay), decay half first; component order unet, text_encoder, text_encoder_2
    got = [(g['lr'], g['weight_decay'], {p.original_name.split('.')[0] for p in g['params']}, {p.ndim == 1 for p in g['params']})
           for g in opt.param_groups]
    assert got == [(4e-5, 0.01, {'unet'}, {False}), (4e-5, 0, {'unet'}, {True}),
                   (2e-5, 0.01, {'text_encoder'}, {False}), (2e-5, 0, {'text_encoder'}, {True}),
                   (1e-6, 0.01, {'text_encoder_2'}, {False}), (1e-6, 0, {'text_encoder_2'}, {True})]
    assert sum(len(g['params']) for g in opt.param_groups) == len(params)
    assert all(g['betas'] == (0.9, 0.99) for g in opt.param_groups)
    with pytest.raises(RuntimeError, match='Unexpected parameter'):
        stray = torch.nn.Parameter(torch.zeros(2, 2))
        stray.original_name = 'vae.x'
        work.get_param_groups([stray])


def test_beta2_half_life_and_unavailable_optimizers():
    work = _workload()
    cfg = {'optimizer': dict(work.train_config['optimizer'], beta2_half_life=2000)}
    params = list(work.unet.parameters())
    opt = optim.make_optimizer_factory(cfg, work, global_batch_size=16, device_is_gpu=False)(params)
    assert math.isclose(opt.param_groups[0]['betas'][1], 0.5 ** (16 / 2000))
    assert cfg['optimizer']['betas'] == [0.9, 0.99]          # the caller's config is not mutated between stages
    for kind in ('adamw_optimi', 'StableAdamW'):
        with pytest.raises(NotImplementedError, match='not available'):
            optim.make_optimizer_factory({'optimizer': {'type': kind, 'lr': 1e-4}}, work, 4)(params)
    # the 8-bit AdamW is a HIP kernel: constructed for GPU stages, refused (not silently replaced) elsewhere
    with pytest.raises(NotImplementedError, match='GPU stages only'):
        optim.make_optimizer_factory({'optimizer': {'type': 'AdamW8bit', 'lr': 1e-4}}, work, 4, device_is_gpu=False)(params)
    for kind, kahan in (('AdamW8bit', False), ('AdamW8bitKahan', True)):
        o8 = optim.make_optimizer_factory({'optimizer': {'type': kind, 'lr': 1e-4, 'betas': [0.9, 0.99], 'weight_decay': 0.01}}, work, 4)(params)
        assert isinstance(o8, optim.AdamW8bit) and o8.kahan is kahan and {g['weight_decay'] for g in o8.param_groups} == {0.01, 0}
    sgd = optim.make_optimizer_factory({'optimizer': {'type': 'SGD', 'lr': 1e-3, 'momentum': 0.9}}, work, 4)(params)
    assert isinstance(sgd, torch.optim.SGD)


def test_lr_schedules():
    p = torch.nn.Parameter(torch.zeros(3))

    def lrs(config, steps):
        opt = torch.optim.SGD([p], lr=1.0)
        sched = optim.make_lr_scheduler(opt, config, steps_per_epoch=10)
        out = []
        for _ in range(steps):
            out.append(opt.param_groups[0]['lr'])
            opt.step()
            sched.step()
        return out
    assert lrs({'warmup_steps': 0}, 3) == [1.0, 1.0, 1.0]
    # LinearLR(start_factor=1/w, total_iters=w): factor_k = 1/w + (1 - 1/w) k / w -- the reference's warm-up, as torch defines it
    assert lrs({'warmup_steps': 4}, 6) == pytest.approx([0.25, 0.4375, 0.625, 0.8125, 1.0, 1.0])
    lin = lrs({'lr_scheduler': 'linear', 'epochs': 1, 'warmup_steps': 0}, 11)
    assert lin[0] == 1.0 and lin[5] == pytest.approx(0.5) and lin[10] == pytest.approx(0.0)
    cos = lrs({'lr_scheduler': 'cosine', 'epochs': 2, 'warmup_steps': 0}, 21)
    assert cos[10] == pytest.approx(0.5 + 0.5e-6, rel=1e-4) and cos[20] == pytest.approx(1e-6)
    with pytest.raises(NotImplementedError):
        optim.make_lr_scheduler(torch.optim.SGD([p], lr=1.0), {'lr_scheduler': 'exotic'}, 10)


def test_lora_wrapping_structure_follows_the_reference_targets():
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.bfloat16)
    n_linear = {k: sum(type(m) is dnn.Linear for m in mod.modules()) for k, mod in work.modules().items()}
    wrapped = work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 8, 'dropout': 0.0, 'dtype': torch.float32})
    # text encoders: every Linear; UNet: only inside down / mid / up blocks (time / add embeddings stay plain)
    assert len(wrapped['text_encoder']) == n_linear['text_encoder'] and len(wrapped['text_encoder_2']) == n_linear['text_encoder_2']
    assert 0 < len(wrapped['unet']) < n_linear['unet']
    assert all(n.split('.')[0] in ('down_blocks', 'mid_block', 'up_blocks') for n in wrapped['unet'])
    assert type(work.unet.time_embedding.linear_1) is dnn.Linear
    trainable = {p.original_name: p for m in work.modules().values() for p in m.parameters() if p.requires_grad}
    assert len(trainable) == 2 * sum(len(v) for v in wrapped.values())
    assert all(('.lora_A.default.' in n or '.lora_B.default.' in n) and p.dtype == torch.float32 for n, p in trainable.items())
    name = 'unet.' + wrapped['unet'][0]
    lora = work.unet.get_submodule(wrapped['unet'][0])
    assert lora.base_layer.weight.dtype == torch.bfloat16 and not lora.base_layer.weight.requires_grad
    assert torch.count_nonzero(lora.lora_B['default'].weight) == 0 and lora.scaling == 2.0
    sd = dnn.lora_state_dict(work.unet)
    assert f"{wrapped['unet'][0]}.lora_A.weight" in sd and len(sd) == 2 * len(wrapped['unet'])
    with pytest.raises(NotImplementedError):
        work.configure_adapter({'type': 'lokr', 'rank': 4, 'alpha': 4})


def test_flux_prepare_inputs_host_logic_matches_oracle():
    """models/flux.py:323-394: patchify, position ids, timestep transforms, tuple layout (host tensors only)."""
    from diffusion_pipe_amd.workloads import flux
    from oracle import flux_ref
    cfg = flux.tiny_flux_config()
    batch = flux.synthetic_flux_batch(cfg, batch_size=3, latent_hw=(8, 12), text_tokens=10, seed=3)
    assert torch.equal(flux.patchify(batch['latents']), flux_ref.patchify(batch['latents']))
    assert torch.equal(flux.prepare_latent_image_ids(4, 6), flux_ref.latent_image_ids(4, 6))
    for mc in ({}, {'shift': 3.0}, {'flux_shift': True}, {'timestep_sample_method': 'uniform'}, {'sigmoid_scale': 1.3, 'guidance': 3.5}):
        work = flux.FluxWorkload(cfg, model_config=mc, dtype=torch.float32)
        torch.manual_seed(11)
        feats, (target, mask) = work.prepare_inputs(batch)
        x_t, t5, clip, t, img_ids, txt_ids, guidance, img_seq_len = feats
        # replay the RNG stream: t draw first, then the noise
        torch.manual_seed(11)
        method = mc.get('timestep_sample_method', 'logit_normal')
        z = torch.distributions.normal.Normal(0, 1).sample((3,)) if method == 'logit_normal' else torch.distributions.uniform.Uniform(0, 1).sample((3,))
        want_t = flux_ref.timestep_transform(z, method, mc.get('sigmoid_scale', 1.0), mc.get('shift'), mc.get('flux_shift', False), image_tokens=24)
        x0 = torch.randn_like(batch['latents'])
        te = want_t.view(-1, 1, 1, 1)
        assert torch.allclose(t, want_t, atol=1e-7)
        assert torch.allclose(x_t, flux_ref.patchify((1 - te) * batch['latents'] + te * x0), atol=1e-6)
        assert torch.allclose(target, flux_ref.patchify(x0 - batch['latents']), atol=1e-6)
        assert x_t.shape == (3, 24, 16) and img_ids.shape == (3, 24, 3) and txt_ids.shape == (3, 10, 3) and not txt_ids.any()
        assert guidance.tolist() == [float(mc.get('guidance', 1.0))] * 3 and img_seq_len.tolist() == [24] * 3 and mask is None
    # eval quantile: deterministic t = sigmoid(icdf(q))
    work = flux.FluxWorkload(cfg, dtype=torch.float32)
    t = work.prepare_inputs(batch, timestep_quantile=0.5)[0][3]
    assert torch.allclose(t, torch.full((3,), 0.5))
    assert len(work.to_layers()) == 1 + cfg.num_layers + cfg.num_single_layers + 1


def test_dynamic_8bit_maps_and_the_8bit_adam_restatement():
    """The two 256-entry code maps (product = oracle construction), the nearest-code quantiser, and the oracle's 8-bit AdamW against fp32 AdamW:
    same trajectory within the quantisation error of the moments (the library is absent: nothing here pins it, see oracle/adam8bit_ref.py)."""
    import numpy as np
    from oracle import adam8bit_ref as ref
    q1, q2 = ref.create_dynamic_map(True), ref.create_dynamic_map(False)
    for got, want in ((optim.create_dynamic_map(True).numpy(), q1), (optim.create_dynamic_map(False).numpy(), q2)):
        assert got.shape == (256,) and np.allclose(got, want, rtol=2e-7, atol=0)
    assert (np.diff(q1) > 0).all() and (np.diff(q2) >= 0).all() and q1[0] < -0.99 and q1[-1] == 1.0 and q2[0] == 0.0 and q2[-1] == 1.0
    assert q1[127] == 0.0 and np.allclose(q1[128:255], -q1[126::-1])           # 127 negative codes, zero, 127 positive codes, 1.0
    x = np.random.default_rng(0).uniform(-1, 1, 4000).astype(np.float32)
    k = ref.quantize_nearest(q1, x)
    assert (np.abs(q1[k] - x) <= np.abs(q1[None, :] - x[:, None]).min(axis=1) + 1e-9).all()
    assert (ref.quantize_nearest(q1, q1) == np.arange(256)).all()                # codes are fixed points
    # trajectory vs fp32 AdamW on a quadratic: p -> target
    rng = np.random.default_rng(1)
    target = rng.normal(size=8192).astype(np.float32)
    p8 = [np.zeros(8192, np.float32)]
    pk = [np.zeros(8192, np.float32)]
    o8 = ref.AdamW8bitRef(p8, lr=5e-2, betas=(0.9, 0.99), weight_decay=0.0, dtype='f32')
    ok = ref.AdamW8bitRef(pk, lr=5e-2, betas=(0.9, 0.99), weight_decay=0.0, kahan=True, dtype='bf16')
    pt = torch.zeros(8192, requires_grad=True)
    ot = torch.optim.AdamW([pt], lr=5e-2, betas=(0.9, 0.99), weight_decay=0.0)
    for _ in range(60):
        o8.step([p8[0] - target]); ok.step([ref.round_to(pk[0] - target, 'bf16')])
        pt.grad = (pt.detach() - torch.from_numpy(target)); ot.step()
    err32 = float((pt.detach() - torch.from_numpy(target)).abs().mean())
    assert abs(float(np.abs(p8[0] - target).mean()) - err32) < 0.02 and float(np.abs(pk[0] - target).mean()) < err32 + 0.03
    assert float(np.abs(p8[0] - pt.detach().numpy()).max()) < 0.15               # 8-bit moments: same path within a few per cent of the step sizes


def test_adamw8bit_state_dict_round_trip_keeps_the_state_dtypes():
    """torch's Optimizer.load_state_dict would cast uint8 codes / fp32 absmax to the parameter dtype; AdamW8bit reloads them as saved (resume must be lossless)."""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(64, 80).to(torch.bfloat16)), torch.nn.Parameter(torch.randn(100).to(torch.bfloat16))]
    a = optim.AdamW8bit(ps, lr=1e-3, kahan=True)
    for p in ps:                                                  # (the step itself is a HIP kernel; fill the state by hand)
        st = a._init_state(p)
        st['step'] = 7
        if st['state1'].dtype == torch.uint8:
            st['state1'].random_(0, 256); st['state2'].random_(0, 256)
            st['absmax1'].uniform_(1e-4, 1e-3); st['absmax2'].uniform_(1e-9, 1e-7)
        else:
            st['state1'].normal_(); st['state2'].uniform_()
        st['shift'].normal_(std=1e-3)
    sd = copy.deepcopy(a.state_dict())
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    b = optim.AdamW8bit(qs, lr=1e-3, kahan=True)
    b.load_state_dict(sd)
    for p, q in zip(ps, qs):
        sa, sb = a.state[p], b.state[q]
        assert sb['step'] == 7 and set(sa) == set(sb)
        for k, v in sa.items():
            if torch.is_tensor(v):
                assert sb[k].dtype == v.dtype and torch.equal(sb[k], v), k
    big = b.state[qs[0]]
    assert big['state1'].dtype == torch.uint8 and big['absmax1'].dtype == torch.float32 and big['qmap1'].dtype == torch.float32 and big['shift'].dtype == torch.bfloat16
    assert b.state[qs[1]]['state1'].dtype == torch.float32                                      # < min_8bit_size: fp32 moments


@pytest.mark.parametrize('kahan', [False, True])
def test_adamw8bit_small_tensor_path_matches_the_oracle(kahan):
    """Tensors below min_8bit_size keep fp32 moments (the library's 32-bit path); the product updates them with multi-tensor ops -- device independent, so
    checked here on CPU against oracle/adam8bit_ref.py over 5 steps with bf16 parameters (parameters identical but for rare one-ulp flips)."""
    import numpy as np
    from oracle import adam8bit_ref as ref
    g = torch.Generator().manual_seed(5)
    shapes = [(100,), (33, 7), (4095,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(torch.bfloat16)) for s in shapes]
    opt = optim.AdamW8bit(ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01, kahan=kahan)
    pr = [p.detach().float().numpy().reshape(-1).copy() for p in ps]
    oref = ref.AdamW8bitRef(pr, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01, kahan=kahan, dtype='bf16')
    for _ in range(5):
        grads = [torch.randn(s, generator=g).to(torch.bfloat16) for s in shapes]
        items = []
        for p, gr in zip(ps, grads):
            st = opt.state[p] if len(opt.state[p]) else opt._init_state(p)
            st['step'] += 1
            items.append((p, gr, st))
        optim.AdamW8bit._step_fp32_moments(items, 1e-2, 0.9, 0.99, 1e-8, 0.01)
        oref.step([gr.float().numpy().reshape(-1) for gr in grads])
        for p, r, st, sr in zip(ps, pr, [opt.state[p] for p in ps], oref.state):
            got = p.detach().float().numpy().reshape(-1)
            assert (np.abs(got - r) <= 2.0 ** -7 * np.abs(r) + 1e-12).all() and (got == r).mean() > 0.99
            np.testing.assert_allclose(st['state1'].numpy().reshape(-1), sr['m'], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(st['state2'].numpy().reshape(-1), sr['v'], rtol=1e-5, atol=1e-10)
            r[:] = got                                                     # re-synchronise (a one-ulp flip must not compound)
            sr['m'], sr['v'] = st['state1'].numpy().reshape(-1).copy(), st['state2'].numpy().reshape(-1).copy()
            if kahan:
                gs = st['shift'].float().numpy().reshape(-1)
                assert (np.abs(gs - sr['shift']) <= 2.0 ** -7 * (np.abs(sr['shift']) + np.abs(r)) + 1e-12).all()
                sr['shift'] = gs.copy()


def test_grad_store_first_touch_protocol():
    """ops._acc / ops.GRAD_STORE (first-micro-batch-stores graphs): outside store mode a fused gradient write into a persistent buffer accumulates; inside, the FIRST
    write to a buffer stores and registers its byte span, later writes to the same buffer accumulate; a fresh output (no buffer) never accumulates."""
    from diffusion_pipe_amd import ops
    a, b = torch.zeros(8), torch.zeros(4, 4)
    assert ops.GRAD_STORE is None
    assert ops._acc(a) is True and ops._acc(None) is False
    ops.GRAD_STORE = {}
    try:
        assert ops._acc(a) is False and ops._acc(a) is True and ops._acc(a) is True          # first touch stores, then accumulates
        assert ops._acc(b) is False and ops._acc(None) is False
        assert ops.GRAD_STORE == {a.data_ptr(): 32, b.data_ptr(): 64}
        assert ops._acc_all(True, a, b) == 1 and ops._acc_all(False, a) == 0
        c, d = torch.zeros(3), torch.zeros(3)
        assert ops._acc_all(True, c, d) == 0 and ops._acc_all(True, c, d) == 1               # gamma + beta written together: both registered, one answer
        assert c.data_ptr() in ops.GRAD_STORE and d.data_ptr() in ops.GRAD_STORE
    finally:
        ops.GRAD_STORE = None
    assert ops._acc(b) is True


def test_direct_quantiser_of_the_8bit_adamw_kernel_is_exact_on_the_dynamic_maps():
    """csrc/optim.hip `direct_code` (round 5) restated in numpy fp32: decade from one logarithm (floor(log2|x| * 0.30103 + 7), clamped to 0 .. 6), slot from the per-decade
    table {lo, scale, slots - 1, first position} by one subtract, one multiply-add, a clamp and a round, then the nearest of the three map entries around that guess (ties ->
    lower index; the kernel's LDS copy of the map carries a -3e38 / +3e38 sentinel either side, the same thing as clipping the candidate indices here).  Must equal the
    oracle's nearest-code search on both of bitsandbytes' dynamic maps for every code, every midpoint between codes +- a few ulps, the decade borders, values one ulp past
    +-1 (x = moment * (1 / absmax) can land there) and two million log-uniform / uniform points; the guess itself must stay within one index of the answer (that is what
    makes three entries enough) -- also when the hardware logarithm (v_log_f32, 1 ulp) is off by +-4 ulps or +-3e-6 absolute."""
    import numpy as np
    from oracle.adam8bit_ref import create_dynamic_map, quantize_nearest
    f = np.float32
    rng = np.random.default_rng(0)
    inv = np.array([1e6, 1e5, 1e4, 1e3, 1e2, 1e1, 1e0], dtype=f)
    for signed in (True, False):
        q = create_dynamic_map(signed)
        base = 1 if signed else 2
        slots = np.array([base << i for i in range(7)], np.int32)
        lo = (f(0.1) / inv).astype(f)
        scale = (slots.astype(f) * inv * f(1.0 / 0.9)).astype(f)

        def guess(x, log_ulps=0, log_abs=0.0):
            a = np.abs(x)
            with np.errstate(divide='ignore'):
                lg = np.log2(a.astype(np.float64)).astype(f)
            if log_ulps:
                fin = np.isfinite(lg)
                lg = np.where(fin, lg + f(log_ulps) * np.spacing(np.where(fin, np.abs(lg), f(1))).astype(f), lg).astype(f)
            lg = (lg + f(log_abs)).astype(f)
            i = np.clip(np.floor(lg * f(0.30103) + f(7.0)), 0, 6).astype(np.int32)          # (-inf for a = 0 clips to 0)
            t = np.clip((a - lo[i]) * scale[i] - f(0.5), 0, (slots[i] - 1).astype(f))
            pos = slots[i] - base + np.rint(t).astype(np.int32)
            return np.where(x >= 0, 128 + pos, 126 - pos) if signed else 1 + pos

        mid = ((q[:-1].astype(np.float64) + q[1:]) / 2).astype(f)
        pts = [q, np.array([0, 1, 1e-7, 1e-8, 0.1, 0.01, 1e-3, 1e-4, 1e-5, 1e-6, 0.09999999, 0.100000001, 1.0000001, 1.0000002], dtype=f)]
        for d in (0, 1, -1, 2, -2, 7, -7):
            pts += [(mid.view(np.int32) + d).view(f), (q.view(np.int32) + d).view(f)]
        lu = (10 ** rng.uniform(-9, 0, 1_000_000)).astype(f)
        pts += [lu, rng.uniform(-1 if signed else 0, 1, 1_000_000).astype(f)] + ([-lu, np.array([-1.0000001, -1.0000002], dtype=f)] if signed else [])
        x = np.concatenate(pts)
        x = x[np.isfinite(x)].astype(f)
        x = x[np.abs(x) <= f(1.000001)]
        if not signed:
            x = x[x >= 0]
        want = quantize_nearest(q, x).astype(np.int32)
        for log_ulps, log_abs in ((0, 0.0), (4, 0.0), (-4, 0.0), (0, 3e-6), (0, -3e-6)):
            k = guess(x, log_ulps, log_abs)
            assert k.min() >= 0 and k.max() <= 255
            assert np.abs(k - want).max() <= 1, (signed, log_ulps, log_abs)
            cand = np.stack([np.clip(k - 1, 0, 255), np.clip(k, 0, 255), np.clip(k + 1, 0, 255)], 1)
            got = cand[np.arange(len(x)), np.argmin(np.abs(q[cand] - x[:, None]), 1)]      # argmin: the first (= lowest index) of equal distances
            assert np.array_equal(got, want), (signed, log_ulps, log_abs)
        # the kernel's in-workgroup check that a launch's maps ARE these maps rests on this closed form (csrc/optim.hip dyn_map_value)
        idx = np.arange(256)
        if signed:
            pos = np.where(idx < 127, 126 - idx, idx - 128)
        else:
            pos = idx - 1
        pos = np.clip(pos, 0, None)
        dec = np.floor(np.log2(pos + base)).astype(np.int32) - (0 if signed else 1)
        nper = (1 << dec) if signed else (2 << dec)
        val = 10.0 ** (dec - 6) * (0.1 + 0.9 * ((pos + base - nper) + 0.5) / nper)
        if signed:
            val = np.where(idx < 127, -val, val); val[127] = 0
        else:
            val[0] = 0
        val[255] = 1
        assert np.allclose(val, q, rtol=1e-5, atol=1e-12)


def test_gelu_erf_approximation():
    """csrc/dpipe_common.h `gelu_erf_cdf` (round 6): the branch-free Phi(x) the exact-GELU kernels evaluate instead of ocml's erff -- Abramowitz-Stegun 7.1.26 with the
    complementary form on the negative side -- restated here in fp32 numpy and held against the fp64 erf: the bounds quoted in the header."""
    import math

    import numpy as np
    from scipy.special import erf
    f32 = np.float32
    x = np.linspace(-12, 12, 400001).astype(f32)
    z = np.abs(x) * f32(0.7071067811865476)
    t = f32(1) / (f32(1) + f32(0.3275911) * z)
    E = np.exp(-(f32(0.5) * x * x)).astype(f32)
    P = t * (f32(0.254829592) + t * (f32(-0.284496736) + t * (f32(1.421413741) + t * (f32(-1.453152027) + t * f32(1.061405429)))))
    half = f32(0.5) * P * E
    cdf = np.where(x >= 0, f32(1) - half, half).astype(f32)
    xd = x.astype(np.float64)
    cdf_t = 0.5 * (1 + erf(xd / math.sqrt(2)))
    assert np.abs(cdf - cdf_t).max() <= 3.0e-7
    assert np.abs(x * cdf - xd * cdf_t).max() <= 4.3e-7
    d = cdf + x * f32(0.3989422804014327) * E
    d_t = cdf_t + xd * np.exp(-0.5 * xd * xd) / math.sqrt(2 * math.pi)
    assert np.abs(d - d_t).max() <= 3.2e-7
    assert (cdf[x < 0] >= 0).all() and (cdf <= 1).all()          # no cancellation on the negative side: Phi stays in [0, 1]
