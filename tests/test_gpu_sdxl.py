"""End-to-end parity of the train_batch hot path on the SDXL-shaped workload (BASELINE config family 1/2, scaled
down so the CPU oracle finishes in seconds): the MI355X engine (HIP kernels through the C ABI) vs the oracle's
sequential fp32 eager step on identical seeded weights and micro-batches.

Tolerances: exact-fp32 kernel mode -> loss and global grad-norm within 1e-3 relative (north_star's bound);
bf16 training mode (the mode the reference itself trains in) -> 3e-2 (bf16 has 8 mantissa bits; compared with the
fp32 oracle, not with a bf16 oracle)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dtype, gpu, gas=2, latent_hw=32):
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    from oracle import eager_step, sdxl_ref
    cfg = sdxl.tiny_config()
    mc = {'min_snr_gamma': 5.0}
    ref = sdxl_ref.SDXLRef(cfg, seed=1)
    work = sdxl.SDXLWorkload(cfg, model_config=mc, dtype=torch.float32, seed=2)
    for k, m in work.modules().items():
        m.load_state_dict(ref.modules()[k].state_dict())
        m.to(dtype)
    torch.manual_seed(7)
    batch = sdxl.synthetic_batch(cfg, batch_size=2 * gas, latent_hw=latent_hw, seed=3, ids_len=75)
    from diffusion_pipe_amd.data import split_batch
    feats, label = work.prepare_inputs(batch)
    micro = split_batch((feats, label), gas)
    layers = work.to_layers()
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': gas,
                                                         'gradient_clipping': 1.0}, device=gpu)
    params = [p for p in module.parameters() if p.requires_grad]
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), params)      # lr 0: inspect grads / norm only
    snr = eager_step.all_snr(eager_step.ddpm_alphas_cumprod())
    ref_loss_fn = eager_step.sdxl_loss_fn(snr_table=snr, min_snr_gamma=5.0)
    return engine, work, ref, micro, ref_loss_fn, eager_step


def _run(dtype, gpu):
    engine, work, ref, micro, ref_loss_fn, eager_step = _setup(dtype, gpu)
    want_loss, want_norm = eager_step.eager_train_step(ref.to_layers(), ref_loss_fn, copy.deepcopy(micro), None, gradient_clipping=1.0, params=ref.parameters())
    ref_grads = {f'{k}.{n}': p.grad.clone() for k, m in ref.modules().items() for n, p in m.named_parameters() if p.grad is not None}
    # hook the product's grads before the optimizer step zeroes them
    got_grads = {}
    orig = engine._exec_optimizer_step

    def spy(*a, **kw):
        engine.clip_fp32_gradients()
        for k, m in work.modules().items():
            for n, p in m.named_parameters():
                if p.grad is not None:
                    got_grads[f'{k}.{n}'] = p.grad.detach().float().cpu().clone()
        engine._gradient_clipping = 0.0
        orig(engine, *a, **kw)
    engine._INSTRUCTION_MAP = dict(engine._INSTRUCTION_MAP)
    from diffusion_pipe_amd.engine import schedule as sched
    engine._INSTRUCTION_MAP[sched.OptimizerStep] = lambda self, **kw: spy(**kw)
    loss = engine.train_batch(iter(micro)).item()
    norm = engine.get_global_grad_norm().item()
    return loss, norm, want_loss.item(), want_norm.item(), got_grads, ref_grads


def test_sdxl_step_fp32_matches_oracle_1e3(gpu):
    loss, norm, want_loss, want_norm, got, ref = _run(torch.float32, gpu)
    assert abs(loss - want_loss) / abs(want_loss) < 1e-3, (loss, want_loss)
    assert abs(norm - want_norm) / want_norm < 1e-3, (norm, want_norm)
    # per-parameter gradients after clipping (clip coefficient included)
    assert set(got) == set(ref)            # the same parameters receive gradients (unused ones stay None on both sides)
    worst = 0.0
    for k, g in ref.items():
        denom = g.abs().max().clamp_min(1e-6)
        worst = max(worst, ((got[k] - g).abs().max() / denom).item())
    assert worst < 5e-3, worst


def test_sdxl_step_bf16_close_to_fp32_oracle(gpu):
    loss, norm, want_loss, want_norm, got, ref = _run(torch.bfloat16, gpu)
    assert abs(loss - want_loss) / abs(want_loss) < 3e-2, (loss, want_loss)
    assert abs(norm - want_norm) / want_norm < 5e-2, (norm, want_norm)


def test_sdxl_eval_batch_and_activation_checkpointing(gpu):
    """eval_batch == mean micro-batch loss; per-layer activation checkpointing leaves loss and grads unchanged."""
    from functools import partial
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    engine, work, ref, micro, ref_loss_fn, eager_step = _setup(torch.float32, gpu)
    want = eager_step.eager_eval(ref.to_layers(), ref_loss_fn, micro).item()
    got = engine.eval_batch(iter(micro), num_micro_batches=len(micro)).item()
    assert abs(got - want) / abs(want) < 1e-3
    base = engine.train_batch(iter(micro)).item()
    ckpt = partial(torch.utils.checkpoint.checkpoint, use_reentrant=False)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='uniform', loss_fn=work.get_loss_fn(),
                                  activation_checkpoint_interval=1, checkpointable_layers=work.checkpointable_layers,
                                  activation_checkpoint_func=ckpt)
    eng2, _, _, _ = initialize(model=module, config={'gradient_accumulation_steps': len(micro), 'gradient_clipping': 1.0}, device=gpu)
    eng2._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), [p for p in module.parameters()])
    assert abs(eng2.train_batch(iter(micro)).item() - base) / abs(base) < 1e-5
    assert abs(eng2.get_global_grad_norm().item() - engine.get_global_grad_norm().item()) / engine.get_global_grad_norm().item() < 1e-4
    # checkpoint inputs parked in pinned host memory (the reference's activation_checkpointing = 'unsloth'); threshold lowered
    # so that the tiny model's stage inputs actually travel over PCIe
    from diffusion_pipe_amd.engine import offloaded_checkpoint
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='uniform', loss_fn=work.get_loss_fn(),
                                  activation_checkpoint_interval=1, checkpointable_layers=work.checkpointable_layers,
                                  activation_checkpoint_func=partial(offloaded_checkpoint, threshold=1000))
    eng3, _, _, _ = initialize(model=module, config={'gradient_accumulation_steps': len(micro), 'gradient_clipping': 1.0}, device=gpu)
    eng3._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), [p for p in module.parameters()])
    assert abs(eng3.train_batch(iter(micro)).item() - base) / abs(base) < 1e-5
    assert abs(eng3.get_global_grad_norm().item() - engine.get_global_grad_norm().item()) / engine.get_global_grad_norm().item() < 1e-4


def test_concurrent_micro_batch_lanes_match_sequential_graph_path(gpu):
    """graph_lanes = 2 / 4: micro-batches replay concurrently on separate streams into per-lane gradient accumulators that
    are summed before clip / step -- loss and global gradient norm must agree with the one-lane path on every step
    (a lane sharing scratch memory with another lane would show up here as run-to-run noise)."""
    import os
    from diffusion_pipe_amd import hip
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    gas = 4

    def run(lanes):
        work = sdxl.SDXLWorkload(cfg, model_config={'min_snr_gamma': 5.0}, dtype=torch.bfloat16, seed=2, device=gpu)
        module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
        engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                             'hip_graph': True, 'graph_lanes': lanes}, device=gpu)
        engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=1e-3), [p for p in module.parameters()])
        if os.environ.get('DPIPE_GEMM_SHALLOW') is None:
            # the engine picks the GEMM ring depth from the number of graphs it replays concurrently (C-ABI option DPIPE_OPT_GEMM_SHALLOW): deep rings for one
            # lane, the two-workgroups-per-CU ring of the 128^2 tile from two lanes on -- and a later engine in the same process may change an earlier one's choice
            assert engine.gemm_shallow_rings == (2 if lanes > 1 else 0)
            assert hip.lib().dpipe_get_option(hip.OPT_GEMM_SHALLOW) == engine.gemm_shallow_rings
        res = []
        for step in range(4):
            torch.manual_seed(100 + step)
            feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=10 + step))
            loss = engine.train_batch(iter(split_batch((feats, label), gas)))
            res.append((loss.item(), engine.get_global_grad_norm().item()))
        return res

    base = run(1)
    for lanes in (2, 4):
        got = run(lanes)
        for (l0, n0), (l1, n1) in zip(base, got):
            assert abs(l1 - l0) / abs(l0) < 5e-3, (lanes, base, got)
            assert abs(n1 - n0) / n0 < 1e-2, (lanes, base, got)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_clip_text_encoders_match_hf_transformers_vectors(gpu, dtype, tol):
    """The HIP-kernel CLIP text encoders (both SDXL geometries: quick_gelu / gelu + projection, causal attention) against vectors from
    the real HF transformers CLIPTextModel / CLIPTextModelWithProjection (oracle/make_golden_clip.py): penultimate hidden state,
    projected pooled embedding, loss, all parameter gradients as one vector."""
    import os
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import sdxl
    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'clip_encoders_fp32.safetensors'))
    cfg = sdxl.tiny_config()
    for tag, c in (('te1', cfg.te1), ('te2', cfg.te2)):
        model = sdxl.CLIPTextModel(c)
        sd = {k[len(f'{tag}.param.'):]: v for k, v in g.items() if k.startswith(f'{tag}.param.')}
        if c.proj_dim is None:
            sd = {k: v for k, v in sd.items() if not k.startswith('text_projection.')}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (tag, missing, unexpected)
        model.to(gpu, dtype)
        penult, pooled = model(g[f'{tag}.ids'].to(gpu), want_pooled=(tag == 'te2'))
        want_p = g[f'{tag}.penultimate']
        assert ((penult.float().cpu() - want_p).abs().max() / want_p.abs().max()).item() < tol
        loss = (penult.float() * g[f'{tag}.w1'].to(gpu)).sum()
        if tag == 'te2':
            want_e = g[f'{tag}.first']
            assert ((pooled.float().cpu() - want_e).abs().max() / want_e.abs().max()).item() < tol
            loss = loss + (pooled.float() * g[f'{tag}.w2'].to(gpu)).sum()
        assert abs(loss.item() - g[f'{tag}.loss'].item()) / abs(g[f'{tag}.loss'].item()) < tol
        loss.backward()
        err2 = ref2 = 0.0
        for k, p in model.named_parameters():
            want = g[f'{tag}.grad.{k}']
            got = p.grad.float().cpu() if p.grad is not None else torch.zeros_like(want)
            err2, ref2 = err2 + (got - want).double().pow(2).sum().item(), ref2 + want.double().pow(2).sum().item()
        assert (err2 / ref2) ** 0.5 < tol, tag


def test_host_offloaded_checkpointing_runs_inside_the_hipgraph_path(gpu):
    """BASELINE config 5's "activation offload to host DRAM" on the engine's fast path: offloaded_checkpoint (the reference's unsloth_checkpoint
    contract, utils/unsloth_utils.py:24-79 hooked up at train.py:586-603) captured into the per-lane micro-batch graphs -- pinned buffers from the
    per-lane pool, D2H / H2D as memcpy nodes -- must reproduce the plain graph path's loss and gradient norm on every step (replays included)."""
    from functools import partial
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize, offloaded_checkpoint
    from diffusion_pipe_amd.engine import offload
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    gas = 4

    def run(ckpt):
        work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=2, device=gpu)
        kw = {}
        if ckpt:
            kw = dict(activation_checkpoint_interval=1, checkpointable_layers=work.checkpointable_layers,
                      activation_checkpoint_func=partial(offloaded_checkpoint, threshold=1000))
        module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='uniform', loss_fn=work.get_loss_fn(), dynamic_shape=True, **kw)
        engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                             'hip_graph': True, 'graph_lanes': 2}, device=gpu)
        engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=1e-3), [p for p in module.parameters()])
        out = []
        for step in range(3):
            torch.manual_seed(50 + step)
            feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=20 + step))
            loss = engine.train_batch(iter(split_batch((feats, label), gas)))
            out.append((loss.item(), engine.get_global_grad_norm().item()))
        return out

    base = run(False)
    got = run(True)
    assert any(k[0] == ('lane', 0) for k in offload._FREE) and any(k[0] == ('lane', 1) for k in offload._FREE)      # per-lane pinned pools were used
    for (l0, n0), (l1, n1) in zip(base, got):
        assert abs(l1 - l0) / abs(l0) < 2e-3 and abs(n1 - n0) / n0 < 5e-3, (base, got)


def test_lane_streams_are_probed_to_run_concurrently(gpu):
    """engine.concurrent_streams: HIP streams share 4 hardware queues, and which queue a new stream lands on depends on how many streams the process created
    before -- two lanes on one queue serialise (16.4 instead of 20.6 images/s on the full-size step).  The engine therefore PROBES its lane streams: spin
    kernels on the caller's stream and on every returned stream together must take about as long as one alone, whatever was created beforehand."""
    import time
    from diffusion_pipe_amd.engine.engine import concurrent_streams
    junk = [torch.cuda.Stream(gpu) for _ in range(3)]                  # shift the round-robin the way an application's own streams would
    main = torch.cuda.current_stream(gpu)
    picked = concurrent_streams(gpu, 3, main)
    assert len(picked) == 3 and len({s.cuda_stream for s in picked} | {main.cuda_stream}) == 4

    def spin(streams, cycles=3_000_000):
        torch.cuda.synchronize(gpu)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(gpu)
        return time.perf_counter() - t0
    spin([main])
    one = min(spin([main]) for _ in range(3))
    four = min(spin([main] + picked) for _ in range(3))
    assert four < 1.6 * one, (one, four)
    del junk


def test_first_micro_batch_stores_graphs_equal_the_zeroing_path_bit_for_bit(gpu):
    """`store_first_micro_batch` (engine default on the fused-step-end path): each lane's first micro-batch of a step replays a graph whose gradient kernels STORE into
    the lane's accumulators (ops.GRAD_STORE), later micro-batches accumulate, and the fused step end zeroes nothing.  bf16(0 + x) = bf16(x), so the trajectory --
    losses, gradient norms and every parameter after four AdamW steps, GAS 6 on 2 lanes (three micro-batches per lane and step) -- is IDENTICAL to the path that
    zeroes the accumulators at the step end and accumulates everywhere."""
    from diffusion_pipe_amd import optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    gas = 6

    def run(store_first):
        work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=2, device=gpu)
        module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
        engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                             'hip_graph': True, 'graph_lanes': 2, 'store_first_micro_batch': store_first}, device=gpu)
        work.train_config = {'optimizer': {'type': 'adamw', 'lr': 1e-3, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
        engine._configure_optimizer(optim.make_optimizer_factory(work.train_config, work, global_batch_size=gas), [p for p in module.parameters() if p.requires_grad])
        assert engine._fused_step_end()
        res = []
        for step in range(4):
            torch.manual_seed(100 + step)
            feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=10 + step))
            loss = engine.train_batch(iter(split_batch((feats, label), gas)))
            res.append((loss.item(), engine.get_global_grad_norm().item()))
        has_first = all(e.get('graph_first') is not None for lane in engine._lanes for e in lane['graphs'].values())
        params = {f'{k}.{n}': p.detach().clone() for k, m in work.modules().items() for n, p in m.named_parameters()}
        return res, params, has_first

    on, p_on, first_on = run(True)
    off, p_off, first_off = run(False)
    assert first_on and not first_off
    assert on == off, (on, off)
    assert all(torch.equal(p_on[k], p_off[k]) for k in p_on)


def test_stacked_micro_batches_match_the_unstacked_graph_path(gpu):
    """`stack_micro_batches: 2 / 4`: the step's four micro-batches of one sample run as two passes of two / one pass of four (hipGraph, lanes) -- loss and global gradient
    norm of every step agree with the unstacked path (same samples and loss terms; other GEMM shapes, so bf16 rounding differs: the pp = 2 tests' tolerances)."""
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    gas = 4

    def run(stack, lanes):
        work = sdxl.SDXLWorkload(cfg, model_config={'min_snr_gamma': 5.0}, dtype=torch.bfloat16, seed=2, device=gpu)
        module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
        engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                             'hip_graph': True, 'graph_lanes': lanes, 'stack_micro_batches': stack}, device=gpu)
        engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=1e-3), [p for p in module.parameters()])
        assert engine.micro_batches == gas // stack and engine.gradient_accumulation_steps() == gas
        res = []
        for step in range(3):
            torch.manual_seed(100 + step)
            feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=10 + step))
            micro = [tuple(tuple(t.to(gpu) for t in part) for part in mb) for mb in split_batch((feats, label), gas)]      # resident in HBM, like the bench's pool
            loss = engine.train_batch(iter(micro))
            res.append((loss.item(), engine.get_global_grad_norm().item()))
        return res

    base = run(1, 2)
    for stack, lanes in ((2, 2), (4, 1)):
        got = run(stack, lanes)
        for (l0, n0), (l1, n1) in zip(base, got):
            assert abs(l1 - l0) / abs(l0) < 2e-2, (stack, base, got)
            assert abs(n1 - n0) / n0 < 3e-2, (stack, base, got)


def test_stacked_micro_batches_at_full_size_stay_finite_under_graph_replay(gpu):
    """Round 5 regression: at FULL size the stacked step (batch > 1 per pass) went NaN from the first graph REPLAY on -- ATen's broadcast reduction behind
    `conv1(h) + temb[:, :, None, None]` returned garbage in single elements of dt under replay (the tiny configuration never showed it); the addend's gradient is now
    summed by this repo's column_sum (ops.add_sample_channel_bias).  Three optimizer steps of the real SDXL UNet + text encoders, two samples stacked per pass, hipGraph:
    every loss finite and every parameter finite afterwards; the first (capture) step's loss equals the eager stacked evaluation of the same batch."""
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=gpu)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': 2, 'gradient_clipping': 1.0, 'steps_per_print': 1 << 30,
                                                         'hip_graph': True, 'graph_lanes': 1, 'stack_micro_batches': 2}, device=gpu)
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=1e-6), [p for p in module.parameters()])
    torch.manual_seed(1234)
    losses = []
    for step in range(3):
        feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=2, latent_hw=128, seed=100 + step))
        micro = [tuple(tuple(t.to(gpu) for t in part) for part in mb) for mb in split_batch((feats, label), 2)]
        losses.append(engine.train_batch(iter(micro)).item())
        norm = engine.get_global_grad_norm().item()
        assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e3 and norm == norm and norm < 1e6, (step, losses, norm)
    assert all(bool(torch.isfinite(p).all()) for p in module.parameters())


def test_round5_graph_replay_corruption_is_rocm_packet_capture_not_this_repo(gpu, record_property):
    """Round 6 root cause of the round-5 stacked-step NaN.  The old arithmetic -- `conv1(h) + temb[:, :, None, None]`, whose backward is an ATen
    reduction with a hipMemsetAsync'ed semaphore buffer -- is kept behind a debug switch (DPIPE_DEBUG_ATEN_TEMB_ADD=1, never set by the product) and replayed as ONE
    single-lane hipGraph at full size by tools/stack_debug_graph.py:
      * with ROCm's graph packet capture DISABLED (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) every gradient is finite -- asserted: the kernels, pools and capture logic of this
        repo are sound (the guard-band / poison probe, tools/oob_guard_probe.py, finds no out-of-bounds write and no uninitialised read either);
      * with packet capture ENABLED (the ROCm 7.2 default) the first replay returns garbage from that ATen reduction (profiles/r6_graph_replay_corruption_packet_capture_ab.txt:
        531 of 2 375 parameters non-finite) -- recorded, not asserted: it is ROCm's behaviour and may change.  The product path holds no such node any more."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outcomes = {}
    for capture in ('0', '1'):
        env = dict(os.environ, DPIPE_PRECISE_ADDENDS='0', DPIPE_DEBUG_ATEN_TEMB_ADD='1', DEBUG_CLR_GRAPH_PACKET_CAPTURE=capture)
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'stack_debug_graph.py'), '4', '1'], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        bad = [int(m) for m in re.findall(r'step \d+: loss \S+, (\d+) / \d+ parameters with non-finite gradients', r.stdout)]
        assert len(bad) == 2, r.stdout[-2000:]
        outcomes[capture] = bad
        record_property(f'non_finite_parameters_per_step_packet_capture_{capture}', bad)
    print('non-finite parameters per step, packet capture off / on:', outcomes)
    assert outcomes['0'] == [0, 0], outcomes
    assert outcomes['1'][0] == 0, outcomes              # the eager / capture step is clean either way
