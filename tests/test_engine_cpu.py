"""Engine host logic on CPU (gloo): pp=1, pp=2 and dp=2 runs of the 1F1B engine reproduce the oracle's sequential
eager step (loss, parameters after the optimizer step, clipped gradients).  world_size-2 processes over 127.0.0.1."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from diffusion_pipe_amd.engine import ManualPipelineModule, PipelineModule, initialize
from diffusion_pipe_amd import data as dpdata
from oracle import eager_step as oracle

D = 16


class First(nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(D, D)

    def forward(self, inputs):
        for t in inputs:
            if torch.is_floating_point(t):
                t.requires_grad_(True)          # adapter convention (models/sdxl.py:677-679)
        x, ids = inputs
        return torch.tanh(self.lin(x)), ids, x * 0.5   # carries an int tensor and a float skip tensor downstream


class Mid(nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(D, D)

    def forward(self, inputs):
        x, ids, skip = inputs
        return torch.tanh(self.lin(x)) + 0.1 * skip, ids, skip


class Last(nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(D, D)

    def forward(self, inputs):
        x, ids, skip = inputs
        return self.lin(x + skip) * (1 + ids.float().mean() * 0)


def make_layers(seed=0, n_mid=4):
    torch.manual_seed(seed)
    return [First()] + [Mid() for _ in range(n_mid)] + [Last()]


def make_batches(n, bs, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.randn(bs, D, generator=g)
        ids = torch.randint(0, 10, (bs, 3), generator=g)
        target = torch.randn(bs, D, generator=g)
        out.append(((x, ids), (target, torch.tensor([]))))
    return out


def oracle_run(steps, gas, clip, batches_per_step, n_mid=4):
    layers = make_layers(n_mid=n_mid)
    params = [p for l in layers for p in l.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-2)
    losses = []
    for s in range(steps):
        loss, _ = oracle.eager_train_step(layers, oracle.default_loss_fn(), batches_per_step[s], opt, gradient_clipping=clip, params=params)
        losses.append(loss.item())
    return losses, [p.detach().clone() for p in params]


def engine_run(steps, gas, clip, batches_for_rank, num_stages, partition_method='uniform', split=None, scope='global', n_mid=4, extra=None):
    layers = make_layers(n_mid=n_mid)
    all_params = [p for l in layers for p in l.parameters()]
    module = ManualPipelineModule(layers=layers, num_stages=num_stages, partition_method=partition_method,
                                  manual_partition_split=split, loss_fn=oracle.default_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': gas,
                                                         'gradient_clipping': clip, 'clip_norm_scope': scope, **(extra or {})}, device='cpu')
    engine.grad_kernels = oracle.TorchGradKernels
    local = [p for p in module.parameters() if p.requires_grad]
    engine._configure_optimizer(lambda ps: torch.optim.AdamW(ps, lr=1e-2), local)
    losses = []
    for s in range(steps):
        engine.reset_activation_shape()
        it = iter(batches_for_rank[s]) if (engine.is_first_stage() or engine.is_last_stage()) else None
        losses.append(engine.train_batch(it).item())
    return losses, all_params, engine


def test_engine_pp1_matches_oracle():
    steps, gas = 3, 4
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches)
    got_l, got_p, engine = engine_run(steps, gas, 0.5, batches, num_stages=1)
    assert got_l == pytest.approx(want_l, rel=1e-6)
    for a, b in zip(got_p, want_p):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    # eval path: forward-only schedule returns the mean micro-batch loss
    ev = engine.eval_batch(iter(batches[0]), num_micro_batches=gas).item()
    assert ev == pytest.approx(oracle.eager_eval(list(engine.module.forward_funcs), oracle.default_loss_fn(), batches[0]).item(), rel=1e-6)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        steps, gas = 2, 4
        if mode == 'pp2':
            batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=2, partition_method='manual', split=[2], scope='global')
            assert engine.module.parts == [0, 2, 6]
        elif mode == 'pp2_dsclip':
            batches = [make_batches(gas, 2, 100)]
            losses, params, engine = engine_run(1, gas, 0.05, batches, num_stages=2, partition_method='uniform', scope='deepspeed')
        elif mode == 'pp4':          # 4 stages x 1 replica: fill / steady state / drain of the 1F1B order over three stage boundaries
            batches = [make_batches(gas + 2, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, gas + 2, 0.5, batches, num_stages=4, partition_method='uniform', scope='global')
            assert engine.module.parts == [0, 2, 4, 5, 6]
        elif mode == 'pp8':          # the BASELINE pp = 8 depth: 10 layers over 8 stages, 16 micro-batches
            batches = [make_batches(16, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, 16, 0.5, batches, num_stages=8, partition_method='uniform', scope='global', n_mid=8)
        elif mode == 'pp2_stack2':   # stack_micro_batches: 2 -- the four micro-batches of a step run as two passes of twice the size on both stages
            batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=2, partition_method='manual', split=[2], scope='global', extra={'stack_micro_batches': 2})
            assert engine.micro_batches == 2 and engine.gradient_accumulation_steps() == 4 and engine.train_batch_size() == 8
        elif mode == 'pp2_stack2_lanes2':   # 8 micro-batches -> 4 passes of two, on two interleaved pipeline lanes
            batches = [make_batches(8, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, 8, 0.5, batches, num_stages=2, partition_method='manual', split=[2], scope='global',
                                                extra={'stack_micro_batches': 2, 'pipe_lanes': 2})
        elif mode == 'dp2_stack2':
            batches = [make_batches(2 * gas, 2, 100 + s)[rank * gas:(rank + 1) * gas] for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=1, extra={'stack_micro_batches': 2})
        elif mode == 'pp2_lanes2':   # two interleaved 1F1B streams per stage (pipe_lanes): micro-batches {0, 2} on lane 0, {1, 3} on lane 1
            batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=2, partition_method='manual', split=[2], scope='global', extra={'pipe_lanes': 2})
            assert engine.pipe_lanes == 2 and len(engine._pipe_lane_state) == 2
        elif mode == 'pp4_lanes3':   # 7 micro-batches over 3 lanes (3 + 2 + 2: lanes of unequal length drain at different ticks), 4 stages
            batches = [make_batches(7, 2, 100 + s) for s in range(steps)]
            losses, params, engine = engine_run(steps, 7, 0.5, batches, num_stages=4, partition_method='uniform', scope='global', extra={'pipe_lanes': 3})
            assert len(engine._pipe_lane_state) == 3
        elif mode == 'pp2dp2_lanes2':
            d = engine_dp_rank(rank)
            batches = [make_batches(2 * gas, 2, 100 + s)[d * gas:(d + 1) * gas] for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=2, partition_method='manual', split=[3], scope='global', extra={'pipe_lanes': 2})
        elif mode == 'pp2dp2':       # 2 stages x 2 replicas: replica d sees its own micro-batches, gradients averaged over the DP group
            d = engine_dp_rank(rank)
            batches = [make_batches(2 * gas, 2, 100 + s)[d * gas:(d + 1) * gas] for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=2, partition_method='manual', split=[3], scope='global')
            assert engine.grid.get_data_parallel_rank() == d and engine.dp_world_size == 2
        elif mode == 'pp2_loader':
            _loader_worker(rank, outdir)
            return
        elif mode == 'flat_dp4':
            _flat_reduce_worker(rank, world, outdir)
            return
        elif mode in ('dp2_overlap', 'dp2_overlap_direct', 'dp2_no_overlap'):
            # the data-parallel average under the tail of the step's last backward (engine/overlap.py): marks at layer boundaries start the collectives of the
            # late layers' gradients while the backward is still in the early ones ('dp2_no_overlap': the same run with the marks off)
            batches = [make_batches(2 * gas, 2, 100 + s)[rank * gas:(rank + 1) * gas] for s in range(steps)]
            extra = {'dp_overlap': mode != 'dp2_no_overlap', 'dp_overlap_min_bytes': 0, 'dp_overlap_marks': 3, 'dp_bucket_bytes': 1500}
            if mode == 'dp2_overlap_direct':
                extra['dp_direct_min_bytes'] = 0
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=1, extra=extra)
            overlap = dict(engine.overlap_report, boundaries=list(engine._marks.boundaries) if engine._marks is not None else None)
        elif mode == 'dp4_overlap':
            gas = 2
            batches = [make_batches(4 * gas, 2, 100 + s)[rank * gas:(rank + 1) * gas] for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=1, extra={'dp_overlap_min_bytes': 0, 'dp_overlap_marks': 3, 'dp_bucket_bytes': 1500})
            overlap = dict(engine.overlap_report, boundaries=list(engine._marks.boundaries))
        elif mode == 'pp2dp2_overlap':   # 2 stages x 2 replicas, two pipeline lanes, marks on both stages
            d = engine_dp_rank(rank)
            batches = [make_batches(2 * gas, 2, 100 + s)[d * gas:(d + 1) * gas] for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=2, partition_method='manual', split=[3], scope='global',
                                                extra={'pipe_lanes': 2, 'dp_overlap_min_bytes': 0, 'dp_overlap_marks': 2})
            overlap = dict(engine.overlap_report, boundaries=list(engine._marks.boundaries) if engine._marks is not None else None)
        else:  # dp2: each replica sees its own micro-batches (dp2_direct: every gradient averaged in place, no staging bucket)
            batches = [make_batches(2 * gas, 2, 100 + s)[rank * gas:(rank + 1) * gas] for s in range(steps)]
            losses, params, engine = engine_run(steps, gas, 0.5, batches, num_stages=1, extra={'dp_direct_min_bytes': 0} if mode == 'dp2_direct' else None)
            assert engine.dp_direct_min_bytes == (0 if mode == 'dp2_direct' else 1 << 20)
        start, stop = engine.module.local_layer_range()
        torch.save({'losses': losses, 'range': (start, stop), 'params': [p.detach() for p in params], 'overlap': locals().get('overlap')}, os.path.join(outdir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def _loader_worker(rank, outdir):
    """The reference's data feed under pp = 2 (train.py:164-173, utils/dataset.py:1341-1405): both end stages pull the SAME dataset
    batches but draw their own noise; the first stage's noise-dependent target is handed to the last stage (`broadcast_target`), the
    iterator for a step is pre-pulled on the end stages only, epochs advance when the dataset wraps."""
    steps, gas = 3, 2
    g = torch.Generator().manual_seed(5)
    dataset = [{'x': torch.randn(2 * gas, D, generator=g), 'ids': torch.randint(0, 10, (2 * gas, 3), generator=g)} for _ in range(2)]
    noise_gen = torch.Generator().manual_seed(1000 + rank)              # each process draws different noise, like the reference

    def prepare_inputs(batch, timestep_quantile=None):
        noise = torch.randn(batch['x'].shape, generator=noise_gen)
        return (batch['x'] + 0.1 * noise, batch['ids']), (noise - batch['x'], None)        # (features), (target, mask)

    layers = make_layers()
    module = ManualPipelineModule(layers=layers, num_stages=2, partition_method='uniform', loss_fn=oracle.default_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': gas,
                                                         'gradient_clipping': 0.5, 'clip_norm_scope': 'global'}, device='cpu')
    engine.grad_kernels = oracle.TorchGradKernels
    engine._configure_optimizer(lambda ps: torch.optim.AdamW(ps, lr=1e-2), [p for p in module.parameters()])
    loader = dpdata.MicroBatchLoader(dataset, engine, gas, prepare_inputs)
    seen, losses, epochs = [], [], []
    for _ in range(steps):
        it = dpdata.get_data_iterator_for_step(loader, engine)
        micro = list(it)
        seen.append(micro)
        engine.reset_activation_shape()
        losses.append(engine.train_batch(iter(micro)).item())
        loader.sync_epoch()
        epochs.append(loader.epoch)
    torch.save({'losses': losses, 'seen': seen, 'epochs': epochs}, os.path.join(outdir, f'r{rank}.pt'))


def _flat_case(rank, lane):
    """gradients of replica `rank`, lane `lane`: seeded, different everywhere"""
    torch.manual_seed(7)
    layers = make_layers()
    params = [p for l in layers for p in l.parameters()]
    g = torch.Generator().manual_seed(1000 * rank + lane)
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    return layers, params


def _flat_reduce_worker(rank, world, outdir):
    """pp = 1, dp = world: two lanes of persistent gradients per replica in flat arenas -> engine._reduce_flat (lane sum + DP average, several buckets
    per arena, in place) -> every parameter's .grad view holds the average over replicas of the sum over lanes."""
    from diffusion_pipe_amd.engine.engine import flatten_grads
    layers, params = _flat_case(rank, 0)
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=None)
    engine, _, _, _ = initialize(model=module, config={'gradient_accumulation_steps': 2, 'dp_bucket_bytes': 256, 'dp_overlap': outdir.endswith('marked'),
                                                         'dp_overlap_min_bytes': 0, 'dp_overlap_marks': 2}, device='cpu')
    assert engine.dp_world_size == world and engine.flat_grads
    base = flatten_grads(params)
    assert list(base) == [torch.float32] and base[torch.float32].numel() >= sum(p.numel() for p in params)
    for p in params:
        assert p.grad.untyped_storage().data_ptr() == base[torch.float32].untyped_storage().data_ptr()
    _, lane1_params = _flat_case(rank, 1)
    other = flatten_grads(lane1_params)
    if engine._marks is not None:
        # as engine._reduce_flat_marked does under the last backward: the arena's tail behind the LAST mark first (lane sum + average), the rest -- arena[:offset] --
        # through _reduce_flat(stop=...)
        bounds = engine._marks.arena_bounds(params, base)
        off = bounds[max(bounds)][torch.float32]
        assert 0 < off < base[torch.float32].numel()
        tail = base[torch.float32][off:]
        tail.add_(other[torch.float32][off:])
        engine._dp_reduce_(tail, engine.grid.get_data_parallel_group())
        engine._reduce_flat(base, [other], stop={torch.float32: off})
    else:
        engine._reduce_flat(base, [other])
    engine._exec_reduce_grads(skip_storages={base[torch.float32].untyped_storage().data_ptr()})     # nothing left outside the arena: a no-op
    torch.save({'grads': [p.grad.clone() for p in params]}, os.path.join(outdir, f'r{rank}.pt'))


@pytest.mark.parametrize('marked', [False, True])
def test_flat_gradient_arenas_lane_sum_and_dp_average_gloo_ws4(marked):
    """marked: the arena's tail behind the last progress mark is summed and averaged first, the head through `_reduce_flat(stop=...)` -- the split the overlapped
    path makes (engine._reduce_flat_marked)"""
    world = 4
    with tempfile.TemporaryDirectory(suffix='marked' if marked else '') as d:
        mp.spawn(_worker, args=(world, _free_port(), 'flat_dp4', d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f'r{r}.pt')) for r in range(world)]
    want = None
    for r in range(world):
        tot = [a.grad + b.grad for a, b in zip(_flat_case(r, 0)[1], _flat_case(r, 1)[1])]
        want = tot if want is None else [w + t for w, t in zip(want, tot)]
    want = [w / world for w in want]
    for r in res:
        for a, b in zip(r['grads'], want):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_flatten_grads_keeps_packed_blocks_strides_and_values():
    """engine.flatten_grads: gradients that share one storage (the packed Q / K / V weight gradients of a fused projection) stay back to back, a
    channels-last convolution weight gradient keeps its strides, values survive, a second call is a no-op, late arrivals stay outside."""
    from diffusion_pipe_amd.engine.engine import flatten_grads
    from diffusion_pipe_amd import ops
    q, k, v = (nn.Parameter(torch.randn(8, 16)) for _ in range(3))
    conv = nn.Parameter(torch.randn(4, 6, 3, 3).contiguous(memory_format=torch.channels_last))
    bias = nn.Parameter(torch.randn(5))
    half = nn.Parameter(torch.randn(7, 3).to(torch.bfloat16))
    packed = torch.randn(24, 16)
    q.grad, k.grad, v.grad = packed[0:8], packed[8:16], packed[16:24]
    conv.grad = torch.randn(4, 6, 3, 3).contiguous(memory_format=torch.channels_last)
    bias.grad, half.grad = torch.randn(5), torch.randn(7, 3).to(torch.bfloat16)
    params = [q, k, v, conv, bias, half]
    before = [p.grad.clone() for p in params]
    arenas = flatten_grads(params)
    assert set(arenas) == {torch.float32, torch.bfloat16}
    for p, b in zip(params, before):
        assert torch.equal(p.grad, b) and p.grad.stride() == b.stride()
        assert p.grad.untyped_storage().data_ptr() == arenas[p.grad.dtype].untyped_storage().data_ptr()
        assert p.grad.data_ptr() % 16 == 0
    assert ops.packed_view([q.grad, k.grad, v.grad]) is not None          # still one [24, 16] block
    assert conv.grad.is_contiguous(memory_format=torch.channels_last)
    again = flatten_grads(params, arenas)
    assert all(again[dt] is arenas[dt] for dt in arenas)
    late = nn.Parameter(torch.randn(3))
    late.grad = torch.randn(3)
    after = flatten_grads(params + [late], arenas)
    assert after[torch.float32] is arenas[torch.float32]
    assert late.grad.untyped_storage().data_ptr() != arenas[torch.float32].untyped_storage().data_ptr()
    arenas[torch.float32].zero_()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in (q, k, v, conv, bias))


def engine_dp_rank(rank):
    """[3P] PipeDataParallelTopology(axes = pipe, data): rank = stage * num_dp + dp_rank (2 x 2 grid here)."""
    return rank % 2


def _spawn(mode, world=2):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), mode, d), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f'r{r}.pt')) for r in range(world)]


def _stage_params(res, layer_param_counts):
    """Assemble the full parameter list from the stage that owns each layer."""
    out = []
    idx = 0
    for layer, n in enumerate(layer_param_counts):
        owner = next(r for r in res if r['range'][0] <= layer < r['range'][1])
        out += owner['params'][idx:idx + n]
        idx += n
    return out


def test_engine_pp2_gloo_matches_oracle():
    steps, gas = 2, 4
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches)
    res = _spawn('pp2')
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-6)      # loss broadcast from the last stage to every rank
    got = _stage_params(res, [2] * 6)
    for a, b in zip(got, want_p):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_engine_pp2_deepspeed_clip_scope_counts_stage0_only():
    """Reference quirk: with pp>1 DeepSpeed's clip only counts pipeline-rank-0 parameters."""
    steps, gas, clip = 1, 4, 0.05
    batches = [make_batches(gas, 2, 100)]
    layers = make_layers()
    params = [p for l in layers for p in l.parameters()]
    for f, l in batches[0]:
        (oracle.default_loss_fn()(oracle.run_layers(layers, f), l) / gas).backward()
    stage0 = [p for l in layers[:3] for p in l.parameters()]
    norm0 = torch.stack([p.grad.float().norm(2) for p in stage0]).square().sum().sqrt()
    coef = min(1.0, clip / (norm0.item() + 1e-6))
    opt = torch.optim.AdamW(params, lr=1e-2)
    for p in params:
        p.grad.mul_(coef)
    opt.step()
    res = _spawn('pp2_dsclip')
    got = _stage_params(res, [2] * 6)
    for a, b in zip(got, params):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_engine_dp2_gloo_matches_oracle():
    steps, gas = 2, 4
    # two replicas x gas micro-batches == one process with 2*gas micro-batches (mean of means with equal sizes)
    batches = [make_batches(2 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 2 * gas, 0.5, batches)
    res = _spawn('dp2')
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
        for a, b in zip(r['params'], want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_engine_dp2_in_place_average_of_large_gradients_matches_staged_buckets():
    """gradients outside an arena are averaged in place from `dp_direct_min_bytes` on (no concatenation, no copy back); with the
    threshold at 0 every gradient takes that route -- same losses and parameters as the staged buckets of the default."""
    steps, gas = 2, 4
    batches = [make_batches(2 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 2 * gas, 0.5, batches)
    res = _spawn('dp2_direct')
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
        for a, b in zip(r['params'], want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_backward_marks_fire_in_descending_order_when_the_later_layers_gradients_are_final():
    """engine/overlap.py: a mark at layer boundary j fires inside the backward, after every gradient of the layers >= j has its final value and before the
    gradients of the earlier layers exist -- the property the overlapped data-parallel average rests on.  Checked on a chain with skip tensors (created in layer 0,
    consumed by every later layer: their hooks must not fire the marks early) over two accumulated micro-batches."""
    from diffusion_pipe_amd.engine.overlap import BackwardMarks
    layers = make_layers(n_mid=4)
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=oracle.default_loss_fn(), dynamic_shape=True)
    marks = BackwardMarks(module, max_marks=3, min_bytes=0)
    assert marks.boundaries and all(1 <= j < len(layers) for j in marks.boundaries) and marks.boundaries == sorted(marks.boundaries)
    params = [p for l in layers for p in l.parameters()]
    seen = []

    def sink(j):
        seen.append((j, {id(p): (None if p.grad is None else p.grad.clone()) for p in params}))
    batches = make_batches(2, 2, 5)
    for i, ((x, ids), (target, _)) in enumerate(batches):
        marks.sink = sink if i == 1 else None          # the LAST micro-batch reports
        marks.begin()
        out = module((x, ids))
        (oracle.default_loss_fn()(out, (target, torch.tensor([]))) / 2).backward()
    marks.sink = None
    assert [j for j, _ in seen] == sorted(marks.boundaries, reverse=True)
    for j, snap in seen:
        for p in params:
            layer = marks.layer_of[id(p)]
            if layer >= j:
                assert snap[id(p)] is not None and torch.equal(snap[id(p)], p.grad), (j, layer)        # final when the mark fired
        early = [p for p in params if marks.layer_of[id(p)] < j - 1]
        assert early and all(not torch.equal(snap[id(p)], p.grad) for p in early)                      # ... and the backward had not reached the early layers
    # arena geometry: gradients re-homed in parameter order -> the layers >= j are a suffix of the arena
    from diffusion_pipe_amd.engine.engine import flatten_grads
    arenas = flatten_grads(params)
    bounds = marks.arena_bounds(params, arenas)
    flat = arenas[torch.float32]
    for j in marks.boundaries:
        off = bounds[j][torch.float32]
        behind = sum(p.numel() for p in params if marks.layer_of[id(p)] >= j)
        assert 0 < off < flat.numel() and flat.numel() - off >= behind
        assert all((p.grad.storage_offset() >= off) == (marks.layer_of[id(p)] >= j) for p in params)
    marks.close()
    seen.clear()
    marks.sink = sink
    marks.begin()
    out = module(batches[0][0])
    oracle.default_loss_fn()(out, (batches[0][1][0], torch.tensor([]))).backward()
    assert not seen                                     # closed: no hooks left on the layers


def test_backward_marks_choose_boundaries_by_gradient_bytes_and_refuse_arenas_out_of_layer_order():
    """`BackwardMarks._choose`: at most max_marks boundaries, ascending, each with about total / (max_marks + 1) bytes of gradients between it and the next one, none
    at layer 0 and none with nothing in front of it; `arena_bounds` names an offset only for an arena whose gradients lie in layer order (else nothing may start early)."""
    from diffusion_pipe_amd.engine.overlap import BackwardMarks
    from diffusion_pipe_amd.engine.engine import flatten_grads
    for n_mid, max_marks in ((4, 3), (9, 2), (9, 6), (1, 6), (4, 0)):
        layers = make_layers(n_mid=n_mid)
        module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=None)
        marks = BackwardMarks(module, max_marks=max_marks, min_bytes=0)
        b = marks.boundaries
        per_layer = (D * D + D) * 4
        assert len(b) <= max_marks and b == sorted(set(b)) and all(1 <= j < len(layers) for j in b)
        want = (len(layers) * per_layer) // (max_marks + 1) if max_marks else 0
        edges = b + [len(layers)]
        for lo, hi in zip(edges, edges[1:]):
            assert (hi - lo) * per_layer >= want                 # every marked range carries its share
            assert (hi - lo - 1) * per_layer < max(want, 1)      # ... and not a layer more than needed
        marks.close()
    # a huge min_bytes: no boundary qualifies -> the engine switches the overlap off
    marks = BackwardMarks(module, max_marks=3, min_bytes=1 << 30)
    assert marks.boundaries == []
    # arena geometry: in layer order -> suffix offsets; gradients re-homed in REVERSED parameter order -> no offsets for that arena
    layers = make_layers(n_mid=4)
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=None)
    marks = BackwardMarks(module, max_marks=3, min_bytes=0)
    params = [p for l in layers for p in l.parameters()]
    for p in params:
        p.grad = torch.ones_like(p)
    ordered = marks.arena_bounds(params, flatten_grads(params))
    assert all(torch.float32 in ordered[j] for j in marks.boundaries)
    offs = [ordered[j][torch.float32] for j in marks.boundaries]
    assert offs == sorted(offs) and len(set(offs)) == len(offs)
    for p in params:
        p.grad = torch.ones_like(p)
    shuffled = marks.arena_bounds(params, flatten_grads(list(reversed(params))))
    assert all(shuffled[j] == {} for j in marks.boundaries)
    marks.close()


def test_engine_dp4_eager_overlap_matches_oracle():
    """four replicas (gloo ring: the bucket a value travels in may change its summation order, so against the oracle with a tolerance, not bitwise)"""
    steps, gas = 2, 2
    batches = [make_batches(4 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 4 * gas, 0.5, batches)
    res = _spawn('dp4_overlap', world=4)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
        for a, b in zip(r['params'], want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        assert r['overlap']['early_collectives'] > 0 and r['overlap']['marks'] == [4, 2]


def test_engine_dp2_average_under_the_last_backward_matches_the_reduction_after_it():
    """VERDICT round 5 'what's missing' 2 / item 7 (utils/patches.py:153-156, train.py:843-844): with `dp_overlap` the collectives of the late layers' gradients are
    started from inside the step's last backward (asynchronous gloo operations here; the communication stream / event-record nodes on the GPU).  Same losses and
    BITWISE the same parameters as the reduction after the backward (two replicas: the sum of two addends does not depend on the bucket a value travels in), on the
    staged-bucket and on the in-place route; the report says what started early."""
    steps, gas = 2, 4
    batches = [make_batches(2 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 2 * gas, 0.5, batches)
    plain = _spawn('dp2_no_overlap')
    for mode in ('dp2_overlap', 'dp2_overlap_direct'):
        res = _spawn(mode)
        for r, q in zip(res, plain):
            assert r['losses'] == pytest.approx(want_l, rel=1e-5)
            for a, b, c in zip(r['params'], want_p, q['params']):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
                assert torch.equal(a, c)
            ov = r['overlap']
            assert ov['path'] == 'eager' and ov['marks'] == sorted(ov['boundaries'], reverse=True) == [4, 2]        # six equal layers, two of them behind each mark
            assert ov['early_collectives'] >= (2 if mode == 'dp2_overlap' else 8) and 0 < ov['early_bytes'] < ov['total_bytes']
            assert ov['early_bytes'] == 4 * (D * D + D) * 4                                                          # layers 2 .. 5 left under the backward
    assert all(q['overlap']['boundaries'] is None and not q['overlap'].get('marks') for q in plain)


def test_engine_pp2_dp2_lanes_with_overlapped_average_match_oracle():
    steps, gas = 2, 4
    batches = [make_batches(2 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 2 * gas, 0.5, batches)
    res = _spawn('pp2dp2_overlap', world=4)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
    for replica in (0, 1):                                   # ranks {0, 2} hold replica 0's stages, {1, 3} replica 1's
        got = _stage_params([res[replica], res[2 + replica]], [2] * 6)
        for a, b in zip(got, want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert all(r['overlap'].get('early_collectives', 0) > 0 and r['overlap']['path'] == 'eager' for r in res), [r['overlap'] for r in res]


def test_checkpoint_roundtrip(tmp_path):
    steps, gas = 1, 2
    batches = [make_batches(gas, 2, 5)]
    _, params, engine = engine_run(steps, gas, 0.0, batches, num_stages=1)
    engine.save_checkpoint(str(tmp_path), client_state={'step': 1, 'examples': 4}, save_latest=True, exclude_frozen_parameters=True)
    assert (tmp_path / 'latest').read_text() == 'global_step1'
    _, params2, engine2 = engine_run(0, gas, 0.0, [], num_stages=1)
    path, client = engine2.load_checkpoint(str(tmp_path), load_module_strict=False)
    assert client == {'step': 1, 'examples': 4} and path.endswith('global_step1')
    for a, b in zip(params, params2):
        assert torch.equal(a, b)
    assert engine2.global_steps == 1


def test_microbatch_loader_epoch_tracking():
    class FakeEngine:
        is_pipe_parallel = False
    dataset = [{'i': i} for i in range(3)]

    def prepare_inputs(batch, timestep_quantile=None):
        x = torch.full((4, 2), float(batch['i']))
        return (x, None), (x + 1, None)

    loader = dpdata.MicroBatchLoader(dataset, FakeEngine(), 2, prepare_inputs)
    assert len(loader) == 6
    seen = []
    for _ in range(6):
        (f, l) = next(loader)
        seen.append(f[0][0, 0].item())
        assert f[1].numel() == 0 and l[1].numel() == 0
    assert seen == [0, 0, 1, 1, 2, 2]
    assert loader.epoch == 2                         # epoch advances as soon as the final micro-batch is returned
    assert next(loader)[0][0][0, 0].item() == 0


def test_offloaded_checkpoint_recomputes_like_plain_autograd():
    """activation_checkpoint_func = offloaded_checkpoint (the reference's unsloth_checkpoint contract, utils/unsloth_utils.py:
    24-79): same outputs and gradients as running the function directly; `no_backward` arguments come back as None."""
    from diffusion_pipe_amd.engine import offloaded_checkpoint
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 8)
    seen = []

    def fn(x, aux, flag):
        seen.append(flag is None)
        y = torch.tanh(lin(x)) * (aux if aux is not None else 1.0)
        return y, aux

    x = torch.randn(4, 8, requires_grad=True)
    aux = torch.full((4, 8), 2.0, requires_grad=True)
    flag = torch.ones(1)
    flag.no_backward = True
    y, _ = offloaded_checkpoint(fn, x, aux, flag)
    y.square().sum().backward()
    got = (x.grad.clone(), aux.grad.clone(), lin.weight.grad.clone())
    x.grad = aux.grad = None
    lin.weight.grad = None
    y2, _ = fn(x, aux, flag)
    y2.square().sum().backward()
    assert torch.allclose(y, y2)
    for a, b in zip(got, (x.grad, aux.grad, lin.weight.grad)):
        assert torch.allclose(a, b, atol=1e-6)
    assert seen[:2] == [False, True]        # forward saw the flag, the recomputation got None in its place


def test_pp2_data_feed_hands_first_stage_targets_to_the_last_stage():
    first, last = _spawn('pp2_loader')
    # the last stage trains against the FIRST stage's noise-dependent targets, its own draws are discarded
    for mb_first, mb_last in zip(sum(first['seen'], []), sum(last['seen'], [])):
        assert torch.equal(mb_first[1][0], mb_last[1][0]) and mb_last[1][1].numel() == 0          # None mask -> empty tensor
        assert not torch.equal(mb_first[0][0], mb_last[0][0])                                      # features: independent noise
    # oracle: sequential steps on the first stage's features and targets
    layers = make_layers()
    params = [p for l in layers for p in l.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-2)
    want = [oracle.eager_train_step(layers, oracle.default_loss_fn(), micro, opt, gradient_clipping=0.5, params=params)[0].item()
            for micro in first['seen']]
    assert first['losses'] == pytest.approx(want, rel=1e-6) and last['losses'] == pytest.approx(want, rel=1e-6)
    # 2 dataset batches x gas = 2 micro-batches each: step 3 starts epoch 2 on both end stages (one micro-batch is always pre-pulled)
    assert first['epochs'] == last['epochs'] and first['epochs'][0] == 1 and first['epochs'][-1] == 2


def test_engine_pp4_gloo_matches_oracle():
    steps, gas = 2, 6
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches)
    res = _spawn('pp4', world=4)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-6)
    got = _stage_params(res, [2] * 6)
    for a, b in zip(got, want_p):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_engine_pp2_dp2_gloo_matches_oracle():
    """The 2 x 2 grid the reference runs on 4 GPUs with pipeline_stages = 2: equals one process over all 2 * gas micro-batches."""
    steps, gas = 2, 4
    batches = [make_batches(2 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 2 * gas, 0.5, batches)
    res = _spawn('pp2dp2', world=4)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
    for replica in (0, 1):                                   # ranks {0, 2} hold replica 0's stages, {1, 3} replica 1's
        got = _stage_params([res[replica], res[2 + replica]], [2] * 6)
        for a, b in zip(got, want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_engine_pp8_gloo_matches_oracle():
    steps, gas = 2, 16
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches, n_mid=8)
    res = _spawn('pp8', world=8)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-6)
    got = _stage_params(res, [2] * 10)
    for a, b in zip(got, want_p):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_engine_pp2_two_pipeline_lanes_gloo_match_oracle():
    """`pipe_lanes: 2`: two interleaved 1F1B instruction streams per stage reproduce the sequential oracle step (same sums, other summation order)."""
    steps, gas = 2, 4
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches)
    res = _spawn('pp2_lanes2')
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-6)
    got = _stage_params(res, [2] * 6)
    for a, b in zip(got, want_p):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_engine_pp4_three_uneven_pipeline_lanes_gloo_match_oracle():
    steps, gas = 2, 7
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches)
    res = _spawn('pp4_lanes3', world=4)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-6)
    got = _stage_params(res, [2] * 6)
    for a, b in zip(got, want_p):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_engine_pp2_dp2_two_pipeline_lanes_gloo_match_oracle():
    steps, gas = 2, 4
    batches = [make_batches(2 * gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, 2 * gas, 0.5, batches)
    res = _spawn('pp2dp2_lanes2', world=4)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
    for replica in (0, 1):
        got = _stage_params([res[replica], res[2 + replica]], [2] * 6)
        for a, b in zip(got, want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_stack_micro_batches_is_the_inverse_of_split_batch():
    g = torch.Generator().manual_seed(0)
    feats = (torch.randn(8, 4, 3, generator=g), torch.randint(0, 9, (8, 5), generator=g), None)
    label = (torch.randn(8, 4, generator=g), None)
    micro = dpdata.split_batch((feats, label), 8)
    stacked = dpdata.stack_micro_batches(micro, 4)
    assert len(stacked) == 2
    for i, (f, l) in enumerate(stacked):
        assert torch.equal(f[0], feats[0][4 * i:4 * i + 4]) and torch.equal(f[1], feats[1][4 * i:4 * i + 4]) and f[2].numel() == 0
        assert torch.equal(l[0], label[0][4 * i:4 * i + 4]) and l[1].numel() == 0
    assert len(dpdata.stack_micro_batches(micro, 3)) == 3 and dpdata.stack_micro_batches(micro, 3)[2][0][0].shape[0] == 2     # ragged tail: a smaller last pass
    it = dpdata.StackedIterator(iter(micro), 2)
    assert [b[0][0].shape[0] for b in it] == [2, 2, 2, 2]
    single = dpdata.stack_micro_batches([(torch.ones(1, 2), torch.zeros(1, 2)), (torch.ones(1, 2) * 2, torch.zeros(1, 2))], 2)      # bare-tensor features / labels
    assert single[0][0].shape == (2, 2) and single[0][1].shape == (2, 2)


def test_engine_pp1_stacked_micro_batches_match_oracle():
    """`stack_micro_batches: 2` (and 4 = the whole step as one pass): the step over UNSTACKED micro-batches in the oracle = the engine's stacked passes"""
    steps, gas = 3, 4
    batches = [make_batches(gas, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, gas, 0.5, batches)
    for k in (2, 4):
        got_l, got_p, engine = engine_run(steps, gas, 0.5, batches, num_stages=1, extra={'stack_micro_batches': k})
        assert engine.micro_batches == gas // k and engine.gradient_accumulation_steps() == gas
        assert got_l == pytest.approx(want_l, rel=1e-6)
        ev = engine.eval_batch(iter(batches[0])).item()              # eval pulls the iterator's own (unstacked) micro-batches, all GAS of them
        assert ev == pytest.approx(oracle.eager_eval(list(engine.module.forward_funcs), oracle.default_loss_fn(), batches[0]).item(), rel=1e-6)
        for a, b in zip(got_p, want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        engine_run(1, gas, 0.5, batches, num_stages=1, extra={'stack_micro_batches': 3})


@pytest.mark.parametrize('mode,world,total', [('pp2_stack2', 2, 4), ('pp2_stack2_lanes2', 2, 8), ('dp2_stack2', 2, 8)])
def test_engine_stacked_micro_batches_gloo_match_oracle(mode, world, total):
    steps = 2
    batches = [make_batches(total, 2, 100 + s) for s in range(steps)]
    want_l, want_p = oracle_run(steps, total, 0.5, batches)
    res = _spawn(mode, world=world)
    for r in res:
        assert r['losses'] == pytest.approx(want_l, rel=1e-5)
    if mode.startswith('pp2'):
        got = _stage_params(res, [2] * 6)
        for a, b in zip(got, want_p):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    else:
        for r in res:
            for a, b in zip(r['params'], want_p):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
