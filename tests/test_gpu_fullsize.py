"""BASELINE config 2 at FULL size on the MI355X: SDXL 1024x1024 (latent 128), the full `SDXLConfig()`, micro-batch 1, GAS 8 on 4 lanes, AdamW, clip 1.0 --
the configuration `bench.py` times.  Regression test for the round-1 driver-bench hang (hipGraph replays of concurrent lanes queued
across step boundaries wedged the queue): >= 12 optimizer steps are enqueued back to back with NO host synchronisation by the caller,
the 4-lane graph path (lane 0 on the caller's stream, the engine's default) must finish, stay finite and follow the 1-lane path's trajectory (same math up to the summation order of the
lanes' bf16 gradient accumulators)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 12
GAS = 8
LANES = 4


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


def test_full_size_sdxl_four_lanes_twelve_unsynchronised_steps_match_one_lane(gpu):
    loss3, norm3 = _trajectory(gpu, LANES)
    assert all(math.isfinite(v) for v in loss3 + norm3), (loss3, norm3)
    loss1, norm1 = _trajectory(gpu, 1)
    assert all(math.isfinite(v) for v in loss1 + norm1), (loss1, norm1)
    for i, (a, b) in enumerate(zip(loss3, loss1)):
        assert abs(a - b) <= 2e-2 * abs(b) + 1e-4, f'step {i}: loss {a} ({LANES} lanes) vs {b} (1 lane)'
    for i, (a, b) in enumerate(zip(norm3, norm1)):
        assert abs(a - b) <= 5e-2 * abs(b) + 1e-4, f'step {i}: grad norm {a} ({LANES} lanes) vs {b} (1 lane)'


# Bounds of the golden comparison.  fp32 = the exact-parity kernel mode (north_star: 1e-3 on loss and gradient norm); bf16 = the TIMED path (bf16 kernels,
# hipGraph, 4 lanes).  Per parameter, against the golden's [sum |g|, sum g, <g, r>, ||g||_2] rows (oracle/checksums.py): abs-sum and L2 norm relative,
# signed sum relative to sum |g|, projection error in units of ||g_ref|| / sqrt(12) (a sample of the tensor's relative L2 error, sign / placement included).
# Three levels per kind: `max` over all 2 375 parameters, the 99th percentile `q99`, and for the projection the AGGREGATE sqrt(12 sum_p dproj_p^2) / ||g||
# = an estimate of the relative L2 error of the whole gradient vector.  Observed on MI355X (round 3): fp32 path loss 0 / norm 8.3e-5 / proj max 5.9e-5;
# bf16 path loss 1.7e-4, norm 2.2e-4, per-parameter max abs_sum 0.083, l2 0.092, proj 0.84 -- all three on the self-attention to_q / to_k weights of the
# deepest transformer blocks, whose gradient is a small difference of bf16-rounded softmax terms (dS = P o (dP - delta)); bounds ~3 x observed.
FP32_BOUNDS = dict(loss=1e-3, norm=1e-3, abs_sum=5e-3, signed_sum=5e-3, proj=1.5e-2, l2=5e-3, agg=5e-3)
# Round 4 (the attention backward's delta reads the forward's fp32 O, csrc/attention.hip): bf16 per-parameter max abs_sum 0.0062, signed_sum 0.0050, proj 0.084, l2 0.0058
# (two runs: 0.0055 / 0.0043 / 0.079 / 0.0059), q99 0.0035 / 0.0021 / 0.058 / 0.0037, aggregate 0.0093 -- the to_q / to_k outliers of round 3 (0.083 / 0.84 sigma) are gone.
BF16_BOUNDS = dict(loss=6e-4, norm=2e-3, abs_sum=0.02, signed_sum=1.5e-2, proj=0.3, l2=0.02, agg=0.026)
# 99th percentile over the parameters (observed bf16: abs_sum 0.035, signed_sum 0.002, proj 0.18, l2 0.032; fp32: 2.6e-6, 1.4e-6, 3.6e-5, 2.7e-6; whole-gradient L2 error
# estimate 0.0086 / 5.3e-6)
FP32_Q99 = dict(abs_sum=1e-4, signed_sum=1e-4, proj=1e-3, l2=1e-4)
BF16_Q99 = dict(abs_sum=0.012, signed_sum=6e-3, proj=0.18, l2=0.012)


def _compare_grad_rows(rows, meta, bounds, what, q99=None):
    """rows: {name: checksum4 row} of this run.  Returns the observed error statistics; asserts the bounds (+ an absolute floor of 1e-8 of the model's
    total sum |g| for parameters whose gradient is analytically zero -- attention key biases: softmax is invariant to them -- and therefore rounding noise)."""
    from oracle.checksums import relative_errors
    gold = meta['grad_checksums']
    assert len(rows) == meta['parameters_with_grad'], (len(rows), meta['parameters_with_grad'])
    total = sum(v[0] for v in gold.values())
    floor = 1e-8 * total
    kinds = ('abs_sum', 'signed_sum', 'proj', 'l2')
    errs = {k: [] for k in kinds}
    worst = {k: (0.0, '') for k in kinds}
    bad, agg_num, agg_den = [], 0.0, 0.0
    for name, got in rows.items():
        ref = gold[name]
        agg_num += 12.0 * (got[2] - ref[2]) ** 2
        agg_den += ref[3] ** 2
        if ref[0] <= floor and got[0] <= 4 * floor:
            continue
        e = dict(zip(kinds, relative_errors(got, ref)))
        for k, v in e.items():
            errs[k].append(v)
            if v > worst[k][0]:
                worst[k] = (v, name)
            slack = floor / max(ref[0], 1e-300) if k != 'proj' else floor / max(ref[3] / 12 ** 0.5, 1e-300)
            if v > bounds[k] + slack:
                bad.append((name, k, v))
    stats = {}
    for k in kinds:
        xs = sorted(errs[k])
        stats[k] = {'median': xs[len(xs) // 2], 'q99': xs[int(0.99 * (len(xs) - 1))], 'max': xs[-1]}
    stats['agg'] = (agg_num / max(agg_den, 1e-300)) ** 0.5
    print(f'{what}: per-parameter errors ' + ', '.join(f'{k} median {stats[k]["median"]:.3g} / q99 {stats[k]["q99"]:.3g} / max {stats[k]["max"]:.3g} ({worst[k][1]})'
                                                        for k in kinds) + f'; whole-gradient relative L2 error estimate {stats["agg"]:.3g}')
    assert not bad, f'{what}: {len(bad)} parameter checks beyond the bounds, e.g. {bad[:5]}'
    assert stats['agg'] <= bounds['agg'], (what, stats['agg'])
    for k, b in (q99 or {}).items():
        assert stats[k]['q99'] <= b, (what, k, 'q99', stats[k]['q99'], b)
    return stats


class _ChecksumOptimizer(torch.optim.Optimizer):
    """Leaves the parameters alone; step() records the checksum rows of the step's (lane-summed) gradients -- they are zeroed right after."""

    def __init__(self, params, names):
        super().__init__(params, {})
        self.names, self.rows = names, {}

    def step(self, closure=None):
        from oracle.checksums import checksum4
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    self.rows[self.names[id(p)]] = checksum4(p.grad, self.names[id(p)])


def test_full_size_sdxl_micro_batch_matches_the_cpu_oracle_golden(gpu, record_property):
    """BASELINE config 2, one full-size micro-batch, against tests/golden/sdxl_fullsize.json (oracle/make_golden_fullsize.py: the oracle's fp32
    eager path on the host, same seeded weights and prepared input) -- loss, global gradient norm and, for every one of the 2 375 parameters with a
    gradient, sum |g|, sum g, a seeded projection <g, r> and ||g||_2:
      * exact-fp32 kernel mode (fp32 MFMA GEMM, unfused attention, the implicit-GEMM convolution as bf16 hi / lo split launches): FP32_BOUNDS;
      * the timed path (bf16, hipGraph, 4 lanes replaying the micro-batch): BF16_BOUNDS."""
    import json
    import os
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from oracle.checksums import checksum4
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
    rows, sq = {}, 0.0
    for k, m in work.modules().items():
        for n, p in m.named_parameters():
            if p.grad is not None:
                rows[f'{k}.{n}'] = checksum4(p.grad, f'{k}.{n}')
                sq += rows[f'{k}.{n}'][3] ** 2
    e_loss, e_norm = abs(loss.item() - meta['loss']) / meta['loss'], abs(sq ** 0.5 - meta['grad_norm']) / meta['grad_norm']
    print(f'fp32 kernel path vs oracle: loss rel. error {e_loss:.3g}, gradient-norm rel. error {e_norm:.3g}')
    assert e_loss < FP32_BOUNDS['loss'], (loss.item(), meta['loss'])
    assert e_norm < FP32_BOUNDS['norm'], (sq ** 0.5, meta['grad_norm'])
    w32 = _compare_grad_rows(rows, meta, FP32_BOUNDS, 'fp32 kernel path', FP32_Q99)
    # ---- the timed path: bf16, hipGraph, 4 lanes (the same micro-batch on every lane: same mean loss, same averaged gradient)
    names = {}
    for k, m in work.modules().items():
        for p in m.parameters():
            p.grad = None
        m.to(torch.bfloat16)
        for n, p in m.named_parameters():
            names[id(p)] = f'{k}.{n}'
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': LANES, 'gradient_clipping': 1e9,
                                                         'hip_graph': True, 'graph_lanes': LANES}, device=gpu)
    opt = engine._configure_optimizer(lambda ps: _ChecksumOptimizer(ps, names), [p for p in module.parameters() if p.requires_grad])
    loss = engine.train_batch(iter([micro[0]] * LANES)).item()
    norm = engine.get_global_grad_norm().item()
    e_loss16, e_norm16 = abs(loss - meta['loss']) / meta['loss'], abs(norm - meta['grad_norm']) / meta['grad_norm']
    print(f'timed path (bf16, hipGraph, {LANES} lanes) vs oracle: loss rel. error {e_loss16:.3g}, gradient-norm rel. error {e_norm16:.3g}')
    assert e_loss16 < BF16_BOUNDS['loss'], (loss, meta['loss'])
    assert e_norm16 < BF16_BOUNDS['norm'], (norm, meta['grad_norm'])
    w16 = _compare_grad_rows(opt.rows, meta, BF16_BOUNDS, 'timed bf16 path', BF16_Q99)
    for k, v in (('fp32_loss', e_loss), ('fp32_norm', e_norm), ('bf16_loss', e_loss16), ('bf16_norm', e_norm16)):
        record_property(k, v)
    for tag, w in (('fp32', w32), ('bf16', w16)):
        record_property(f'{tag}_agg', w['agg'])
        for k in ('abs_sum', 'signed_sum', 'proj', 'l2'):
            record_property(f'{tag}_{k}_max', w[k]['max'])
            record_property(f'{tag}_{k}_q99', w[k]['q99'])
