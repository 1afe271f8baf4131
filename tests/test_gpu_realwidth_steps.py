"""GPU: one `train_batch` STEP of BASELINE configs 3 / 4 / 5 at their REAL width (depth-truncated), the way bench.py --workload runs them -- bf16 kernels,
LoRA rank 32 on every block Linear where the config trains adapters (Flux, Wan), full fine-tune with activation checkpointing + host-offloaded
checkpoints for HunyuanVideo, hipGraph + 2 micro-batch lanes (Flux, Wan), gradient clipping 1.0 -- against tests/golden/realwidth_steps.json: the
oracle's fp32 eager step on the host over the same seeded weights and micro-batch (oracle/make_golden_realwidth_steps.py; VERDICT round 3 item 5:
step-level parity existed only at toy width).

Compared: mean loss, pre-clip global gradient norm, and per TRAINED parameter the [sum |g|, sum g, <g, r>, ||g||_2] checksum rows (oracle/checksums.py).
Bounds = ~3 x the errors observed on MI355X (printed and recorded as test properties)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'realwidth_steps.json')

# (loss, norm, per-parameter max of: abs_sum, signed_sum (of sum |g|), projection (in sigma = ||g|| / sqrt(12)), l2; whole-gradient L2 error estimate).
# Observed on MI355X (round 4, profiles/r4e_realwidth_steps.txt):
#   flux  loss 1.0e-5  norm 5.2e-4  abs_sum 0.0043  signed_sum 0.0039  proj 0.029  l2 0.0034  agg 0.0085
#   wan   loss 3.1e-5  norm 7.5e-4  abs_sum 0.0023  signed_sum 0.0015  proj 0.015  l2 0.0025  agg 0.0046
#   hv    loss 4.5e-6  norm 1.3e-5  abs_sum 0.0037  signed_sum 0.0065  proj 0.022  l2 0.0036  agg 0.0093
_B = dict(loss=3e-4, norm=2.5e-3, abs_sum=0.015, signed_sum=0.02, proj=0.1, l2=0.012, agg=0.03)
BOUNDS = {'flux': _B, 'wan': _B, 'hv': _B}


class _Rows(torch.optim.Optimizer):
    """Leaves the parameters alone; step() records the checksum rows of the step's (lane-summed, clipped) gradients."""

    def __init__(self, params, names):
        super().__init__(params, {})
        self.names, self.rows = names, {}

    def step(self, closure=None):
        from oracle.checksums import checksum4
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    self.rows[self.names[id(p)]] = checksum4(p.grad, self.names[id(p)])


def _run_step(gpu, work, micro, gold, *, lanes, graph, ckpt=None, name_map=None):
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from oracle.make_golden_realwidth_steps import state_checksum
    assert abs(state_checksum(work.transformer) - gold['state_checksum']) <= 1e-7 * gold['state_checksum'], 'seeded weights differ from the generator run'
    work.transformer.to(gpu, torch.bfloat16)
    names = {id(p): (name_map(n) if name_map else n) for n, p in work.transformer.named_parameters()}
    kwargs = {}
    if ckpt is not None:
        kwargs = dict(activation_checkpoint_interval=1, checkpointable_layers=work.checkpointable_layers, activation_checkpoint_func=ckpt)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True, **kwargs)
    gas = max(2, lanes)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                         'hip_graph': graph, 'graph_lanes': lanes}, device=gpu)
    opt = engine._configure_optimizer(lambda ps: _Rows(ps, names), [p for p in module.parameters() if p.requires_grad])
    loss = norm = None
    for _ in range(2 if graph else 1):                   # graph path: the second step replays the captured lanes
        opt.rows = {}
        loss = engine.train_batch(iter([micro] * gas)).item()          # the same micro-batch on every lane: mean loss = its loss, summed g / GAS = its gradient
        norm = engine.get_global_grad_norm().item()
    torch.cuda.synchronize()
    coef = min(1.0, 1.0 / (norm + 1e-6))                  # undo the clip the engine applied before optimizer.step (utils/patches.py:222-246)
    rows = {k: [v / coef for v in r] for k, r in opt.rows.items()}
    return loss, norm, rows


def _check(case, loss, norm, rows, gold, record_property):
    from oracle.checksums import relative_errors
    b = BOUNDS[case]
    e_loss, e_norm = abs(loss - gold['loss']) / gold['loss'], abs(norm - gold['grad_norm']) / gold['grad_norm']
    ref = gold['param_grads']
    assert set(rows) == set(ref), (sorted(set(rows) ^ set(ref))[:6], len(rows), len(ref))
    kinds = ('abs_sum', 'signed_sum', 'proj', 'l2')
    worst = {k: (0.0, '') for k in kinds}
    num = den = 0.0
    floor = 1e-8 * sum(r[0] for r in ref.values())       # parameters whose gradient is analytically zero (attention key biases: softmax is invariant to them) hold
    for n, got in rows.items():                          # rounding noise on both sides: skipped below an absolute floor of 1e-8 of the model's total sum |g|
        num += 12.0 * (got[2] - ref[n][2]) ** 2
        den += ref[n][3] ** 2
        if ref[n][0] <= floor and got[0] <= 4 * floor:
            continue
        for k, v in zip(kinds, relative_errors(got, ref[n])):
            if v > worst[k][0]:
                worst[k] = (v, n)
    agg = (num / max(den, 1e-300)) ** 0.5
    print(f'{case} real-width step vs oracle: loss rel. error {e_loss:.3g}, gradient-norm rel. error {e_norm:.3g}; per-parameter max ' +
          ', '.join(f'{k} {worst[k][0]:.3g} ({worst[k][1]})' for k in kinds) + f'; whole-gradient relative L2 error estimate {agg:.3g}')
    for k, v in (('loss', e_loss), ('norm', e_norm), ('agg', agg)):
        record_property(f'{case}_{k}', v)
    for k in kinds:
        record_property(f'{case}_{k}_max', worst[k][0])
    assert e_loss < b['loss'], (loss, gold['loss'])
    assert e_norm < b['norm'], (norm, gold['grad_norm'])
    for k in kinds:
        assert worst[k][0] < b[k], (k, worst[k])
    assert agg < b['agg'], agg


def test_flux_real_width_lora_step_matches_oracle(gpu, record_property):
    from oracle.make_golden_realwidth_steps import flux_case
    gold = json.load(open(GOLD))['flux']
    _, work, micro = flux_case()
    loss, norm, rows = _run_step(gpu, work, micro, gold, lanes=2, graph=True)
    _check('flux', loss, norm, rows, gold, record_property)


def test_wan_real_width_lora_step_matches_oracle(gpu, record_property):
    from oracle.make_golden_realwidth_steps import wan_case
    gold = json.load(open(GOLD))['wan']
    _, work, micro = wan_case()
    loss, norm, rows = _run_step(gpu, work, micro, gold, lanes=2, graph=True)
    _check('wan', loss, norm, rows, gold, record_property)


def test_hunyuan_video_real_width_full_finetune_step_with_offloaded_checkpoints_matches_oracle(gpu, record_property):
    from diffusion_pipe_amd.engine import offload
    from oracle.make_golden_realwidth_steps import hv_case
    gold = json.load(open(GOLD))['hv']
    _, work, _tr, micro = hv_case()
    del _tr
    before = sum(len(v) for v in offload._FREE.values())
    loss, norm, rows = _run_step(gpu, work, micro, gold, lanes=1, graph=False, ckpt=offload.offloaded_checkpoint)
    assert sum(len(v) for v in offload._FREE.values()) > before, 'no checkpoint was parked in pinned host memory (8.8 M-element image tokens >= OFFLOAD_THRESHOLD)'
    _check('hv', loss, norm, rows, gold, record_property)
