"""BASELINE config 2 at FULL size on the MI355X: SDXL 1024x1024 (latent 128), the full `SDXLConfig()`, micro-batch 1, GAS 6, AdamW, clip 1.0 --
the configuration `bench.py` times.  Regression test for the round-1 driver-bench hang (hipGraph replays of 3 concurrent lanes queued
across step boundaries wedged the queue): >= 12 optimizer steps are enqueued back to back with NO host synchronisation by the caller,
the 3-lane graph path must finish, stay finite and follow the 1-lane path's trajectory (same math up to the summation order of the
lanes' bf16 gradient accumulators)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 12
GAS = 6


def _engine(gpu, lanes, pool_seed=100):
    from diffusion_pipe_amd import optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=gpu)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': GAS, 'gradient_clipping': 1.0,
                                                         'hip_graph': True, 'graph_lanes': lanes}, device=gpu)
    work.train_config = {'optimizer': {'type': 'adamw', 'lr': 1e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
    engine._configure_optimizer(optim.make_optimizer_factory(work.train_config, work, global_batch_size=GAS), [p for p in module.parameters() if p.requires_grad])
    torch.manual_seed(1234)
    pool = []
    for s in range(3):
        feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=GAS, latent_hw=128, seed=pool_seed + s))
        pool.append(split_batch((feats, label), GAS))
    return engine, pool


def _trajectory(gpu, lanes):
    engine, pool = _engine(gpu, lanes)
    losses, norms = [], []
    for i in range(STEPS):                                   # no .item(), no synchronize: device scalars only
        engine.reset_activation_shape()
        losses.append(engine.train_batch(iter(pool[i % len(pool)])))
        norms.append(engine.get_global_grad_norm())
    torch.cuda.synchronize()
    out = [float(l.item()) for l in losses], [float(n.item()) for n in norms]
    del engine, pool
    torch.cuda.empty_cache()
    return out


def test_full_size_sdxl_three_lanes_twelve_unsynchronised_steps_match_one_lane(gpu):
    loss3, norm3 = _trajectory(gpu, 3)
    assert all(math.isfinite(v) for v in loss3 + norm3), (loss3, norm3)
    loss1, norm1 = _trajectory(gpu, 1)
    assert all(math.isfinite(v) for v in loss1 + norm1), (loss1, norm1)
    for i, (a, b) in enumerate(zip(loss3, loss1)):
        assert abs(a - b) <= 2e-2 * abs(b) + 1e-4, f'step {i}: loss {a} (3 lanes) vs {b} (1 lane)'
    for i, (a, b) in enumerate(zip(norm3, norm1)):
        assert abs(a - b) <= 5e-2 * abs(b) + 1e-4, f'step {i}: grad norm {a} (3 lanes) vs {b} (1 lane)'


def test_full_size_sdxl_micro_batch_matches_the_cpu_oracle_golden(gpu):
    """BASELINE config 2, one full-size micro-batch, against tests/golden/sdxl_fullsize.json (oracle/make_golden_fullsize.py: the oracle's fp32
    eager path on the host, same seeded weights and prepared input):
      * exact-fp32 kernel mode: loss and global gradient norm within 1e-3 relative (north_star's bound), every parameter's sum |g| within 5e-3 (+ 1e-8 of the model total);
      * the timed path (bf16, hipGraph, 3 lanes replaying the micro-batch): loss within 3e-2, gradient norm within 5e-2."""
    import json
    import os
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from oracle.make_golden_fullsize import build, weight_checksum
    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sdxl_fullsize.json')))
    cfg, work, micro = build()
    assert abs(weight_checksum(work.modules()) - meta['weight_checksum']) <= 1e-9 * meta['weight_checksum'], 'seeded weights differ from the generator run'
    for m in work.modules().values():
        m.to(gpu)
    feats, label = micro[0]
    # ---- exact-fp32 kernels, eager
    x = tuple(t.to(gpu) for t in feats)
    for layer in work.to_layers():
        x = layer(x)
    loss = work.get_loss_fn()(x, tuple(t.to(gpu) for t in label))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - meta['loss']) / meta['loss'] < 1e-3, (loss.item(), meta['loss'])
    sums, sq = {}, 0.0
    for k, m in work.modules().items():
        for n, p in m.named_parameters():
            if p.grad is not None:
                g = p.grad.double()
                sums[f'{k}.{n}'] = float(g.abs().sum())
                sq += float((g * g).sum())
    assert abs(sq ** 0.5 - meta['grad_norm']) / meta['grad_norm'] < 1e-3, (sq ** 0.5, meta['grad_norm'])
    assert len(sums) == meta['parameters_with_grad']
    # per parameter: sum |g| within 5e-3 relative, plus an absolute floor of 1e-8 of the model's total for parameters whose gradient is
    # analytically zero (attention key biases: softmax is invariant to them) and therefore pure rounding noise on both sides
    total = sum(v[0] for v in meta['grad_checksums'].values())
    excess = {k: abs(v - meta['grad_checksums'][k][0]) - (5e-3 * meta['grad_checksums'][k][0] + 1e-8 * total) for k, v in sums.items()}
    worst = max(excess, key=excess.get)
    assert excess[worst] <= 0, (worst, sums[worst], meta['grad_checksums'][worst][0], total)
    # ---- the timed path: bf16, hipGraph, 3 lanes (the same micro-batch on every lane: same mean loss, same averaged gradient)
    for m in work.modules().values():
        for p in m.parameters():
            p.grad = None
        m.to(torch.bfloat16)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': 3, 'gradient_clipping': 1e9,
                                                         'hip_graph': True, 'graph_lanes': 3}, device=gpu)
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), [p for p in module.parameters() if p.requires_grad])
    loss = engine.train_batch(iter([micro[0]] * 3)).item()
    norm = engine.get_global_grad_norm().item()
    assert abs(loss - meta['loss']) / meta['loss'] < 3e-2, (loss, meta['loss'])
    assert abs(norm - meta['grad_norm']) / meta['grad_norm'] < 5e-2, (norm, meta['grad_norm'])
