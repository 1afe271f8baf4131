"""Bit-exact integer / index logic: product (diffusion_pipe_amd.engine, .data) vs the scalar oracle restatement
(oracle/intlogic.py), plus the reference's own worked examples.  CPU only."""
import math
import random

import numpy as np
import pytest

from diffusion_pipe_amd import data
from diffusion_pipe_amd.engine import module as pm
from diffusion_pipe_amd.engine import schedule as ps
from diffusion_pipe_amd.engine.topology import PipeDataParallelTopology
from oracle import intlogic as ol


def _names(steps):
    return [[(c.name, c.kwargs.get('buffer_id')) for c in step] for step in steps]


def test_train_schedule_worked_example_s2_m4():
    """S=2, M=4, stage 0 starts [Load0 Fwd0] [SendAct0] [Load1 Fwd1] [RecvGrad0 SendAct1 Bwd0]."""
    steps = _names(ps.TrainSchedule(micro_batches=4, stages=2, stage_id=0).steps())
    assert len(steps) == 2 * (4 + 2 - 1)
    assert steps[0] == [('LoadMicroBatch', 0), ('ForwardPass', 0)]
    assert steps[1] == [('SendActivation', 0)]
    assert steps[2] == [('LoadMicroBatch', 1), ('ForwardPass', 1)]
    assert steps[3] == [('RecvGrad', 0), ('SendActivation', 1), ('BackwardPass', 0)]
    assert steps[-1][-3:] == [('ReduceTiedGrads', None), ('ReduceGrads', None), ('OptimizerStep', None)]


@pytest.mark.parametrize('stages', [1, 2, 3, 4, 8])
@pytest.mark.parametrize('mbs', [1, 2, 4, 7, 16])
def test_schedules_match_oracle(stages, mbs):
    for stage in range(stages):
        assert _names(ps.TrainSchedule(mbs, stages, stage).steps()) == ol.train_schedule(mbs, stages, stage)
        assert _names(ps.InferenceSchedule(mbs, stages, stage).steps()) == ol.inference_schedule(mbs, stages, stage)


@pytest.mark.parametrize('stages', [2, 4, 8])
def test_train_schedule_is_a_valid_pipeline(stages):
    """Size-independent properties: every micro-batch is forwarded then backwarded exactly once per stage, sends
    and receives pair up between neighbours in the same step order."""
    M = 8
    per_stage = [ol.train_schedule(M, stages, s) for s in range(stages)]
    for s, steps in enumerate(per_stage):
        flat = [c for st in steps for c in st]
        assert sum(1 for c in flat if c[0] == 'ForwardPass') == M
        assert sum(1 for c in flat if c[0] == 'BackwardPass') == M
        if s + 1 < stages:
            sends = [i for i, st in enumerate(steps) for c in st if c[0] == 'SendActivation']
            recvs = [i for i, st in enumerate(per_stage[s + 1]) for c in st if c[0] == 'RecvActivation']
            assert sends == recvs                      # same step index on both sides => no deadlock
            gs = [i for i, st in enumerate(per_stage[s + 1]) for c in st if c[0] == 'SendGrad']
            gr = [i for i, st in enumerate(steps) for c in st if c[0] == 'RecvGrad']
            assert gs == gr


def test_partition_uniform_and_balanced_match_oracle():
    rng = random.Random(0)
    for _ in range(300):
        n, m = rng.randint(1, 40), rng.randint(1, 9)
        assert pm.partition_uniform(n, m) == ol.partition_uniform(n, m)
        w = [rng.choice([0, 1, 5, 1000, 123456789, 2_600_000_000]) * rng.randint(0, 3) for _ in range(n)]
        assert pm.partition_balanced(w, m) == ol.partition_balanced(w, m), (w, m)


def test_partition_layer_counts_of_the_baseline_models():
    # SDXL 23 layers (models/sdxl.py:591-602), Flux 59, Wan-14B 42, HunyuanVideo 63
    for L in (23, 59, 42, 63):
        for S in (1, 2, 4, 8):
            parts = pm.partition_uniform(L, S)
            assert parts[0] == 0 and parts[-1] == L and len(parts) == S + 1
            assert all(b >= a for a, b in zip(parts, parts[1:]))
            assert ol.manual_partition(L, S, parts[1:-1]) == parts


def test_topology_rank_layout():
    topo = PipeDataParallelTopology(num_pp=4, num_dp=2)
    for s in range(4):
        for d in range(2):
            assert topo.get_rank(s, d) == ol.rank_of(s, d, 4, 2)
            assert topo.get_coord(topo.get_rank(s, d)) == (s, d)
    assert topo.get_axis_comm_lists('pipe') == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert topo.get_axis_comm_lists('data') == [[0, 1], [2, 3], [4, 5], [6, 7]]


def test_bucket_arithmetic_matches_oracle():
    rng = random.Random(1)
    ars = data.make_ar_buckets(0.5, 2.0, 7)
    assert np.array_equal(ars, ol.dedup_and_sort(np.geomspace(0.5, 2.0, num=7)))
    fb = np.array([1, 33, 65])
    for _ in range(500):
        w, h, frames = rng.randint(64, 4000), rng.randint(64, 4000), rng.choice([1, 1, 20, 33, 50, 65, 200])
        log_ar = np.log(w / h)
        for is_video in (False, True):
            assert data.find_closest_ar_bucket(log_ar, frames, is_video, ars, fb) == ol.find_closest_ar_bucket(log_ar, frames, is_video, ars, fb)
        res, mult = rng.choice([512, 768, 1024]), rng.choice([16, 32, 64])
        for ar in ars:
            assert data.size_bucket_for(ar, frames, res, mult) == ol.size_bucket(ar, frames, res, mult)
    sbs = [(1024, 1024, 1), (1280, 768, 1), (768, 1280, 1), (512, 512, 33), (640, 384, 65)]
    for _ in range(200):
        log_ar, frames = math.log(rng.uniform(0.3, 3.0)), rng.choice([1, 10, 33, 40, 65, 100])
        for is_video in (False, True):
            a, b = data.find_closest_size_bucket(log_ar, frames, is_video, sbs), ol.find_closest_size_bucket(log_ar, frames, is_video, sbs)
            assert (a is None and b is None) or np.array_equal(a, b)
    for x in [0.5, 1.5, 2.5, 1023.9, 1040.0, 1008.0, 24.0, 8.0]:
        assert data.round_to_nearest_multiple(x, 16) == ol.round_to_nearest_multiple(x, 16)
    assert data.round_to_nearest_multiple(24.0, 16) == 32 and data.round_to_nearest_multiple(8.0, 16) == 0   # banker's rounding
    assert data.seed_from_hash('some/path.png') == ol.seed_from_hash('some/path.png')
    a, b = list(range(50)), list(range(50))
    data.shuffle_with_seed(a, 7); ol.shuffle_with_seed(b, 7)
    assert a == b


def test_iteration_order_and_dp_slices():
    state = random.getstate()
    for lens, gbs in [([5, 9, 3], 4), ([100], 8), ([1, 1, 1], 4), ([0, 7], 2)]:
        got = data.batched_iteration_order(lens, gbs)
        want = ol.iteration_order(lens, gbs)
        assert [tuple(r) for r in got.tolist()] == want
        assert len(got) % gbs == 0
    assert random.getstate() == state               # global RNG untouched (utils/dataset.py:41-45)
    for idx in range(3):
        covered = []
        for r in range(4):
            s, e = data.dp_batch_slice(idx, 8, r, 4)
            assert (s, e) == ol.dp_slice(idx, 8, r, 4)
            covered += list(range(s, e))
        assert covered == list(range(idx * 8, idx * 8 + 8))
    assert data.pick_global_batch_size((1024, 1024, 1), {None: 4}) == 4
    d = {512: 8, 1024: 2}
    for sb in [(512, 512, 1), (1280, 768, 1), (768, 768, 1)]:
        assert data.pick_global_batch_size(sb, d) == ol.pick_global_batch_size(sb, d)


def test_split_batch_edge_cases():
    import torch
    feats = (torch.arange(8.).view(4, 2), None, torch.arange(4))
    label = (torch.ones(4, 3), None)
    mbs = data.split_batch((feats, label), 2)
    assert len(mbs) == 2
    (f0, l0), (f1, l1) = mbs
    assert torch.equal(f0[0], feats[0][:2]) and torch.equal(f1[2], feats[2][2:])
    assert f0[1].numel() == 0 and l1[1].numel() == 0        # None -> empty tensor
    assert all(torch.is_tensor(t) for t in f0 + l0)


def test_partition_methods_on_real_layer_lists():
    """PipelineModule._partition_layers on the SDXL (23 layers) and Flux layer lists (single process; the 4-stage topology is swapped in
    after construction): 'parameters' = the balanced DP over ALL parameters of each layer (train.py:81-90), 'type:<regex>' balances the
    matching layers, 'uniform' splits by count, the reference's 'manual' honours explicit boundaries."""
    import torch
    from diffusion_pipe_amd.engine import ManualPipelineModule, PipelineModule
    from diffusion_pipe_amd.workloads import flux, sdxl

    def parts(module, method, stages, rank=0):
        module._topo, module.global_rank = PipeDataParallelTopology(num_pp=stages, num_dp=1), rank
        module._partition_layers(method)
        return module.parts, (module._local_start, module._local_stop)
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32)
    layers = work.to_layers()
    counts = [sum(p.numel() for p in l.parameters()) for l in layers]
    m = PipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=work.get_loss_fn())
    assert m.parts == [0, 23]
    got, bounds = parts(m, 'parameters', 4, rank=2)
    assert got == ol.partition_balanced(counts, 4) and got[0] == 0 and got[-1] == 23 and bounds == (got[2], got[3])
    assert parts(m, 'uniform', 4)[0] == ol.partition_uniform(23, 4)
    marks = [1 if type(l).__name__ == 'UpBlockInnerLayer' else 0 for l in layers]
    assert parts(m, 'type:UpBlockInner', 4)[0] == ol.partition_balanced(marks, 4) and sum(marks) == 9
    manual = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform')
    manual.manual_partition_split = [3, 9, 17]
    assert parts(manual, 'manual', 4, rank=3) == ([0, 3, 9, 17, 23], (17, 23)) and ol.manual_partition(23, 4, [3, 9, 17]) == [0, 3, 9, 17, 23]
    fl = flux.FluxWorkload(flux.tiny_flux_config(), dtype=torch.float32).to_layers()
    fm = PipelineModule(layers=fl, num_stages=1, partition_method='uniform')
    assert parts(fm, 'type:transformerwrapper', 2)[0] == ol.partition_balanced([1 if 'TransformerWrapper' in type(l).__name__ else 0 for l in fl], 2)
    with pytest.raises(NotImplementedError):
        parts(m, 'profile', 4)


# ---- pipeline lanes (engine `pipe_lanes`): wire order and deadlock freedom of L interleaved 1F1B streams per stage
def _lane_wire_ops(stages, micro_batches, lanes):
    """per stage: the P2P operations in ISSUE order, as (op, peer stage, message id); a message id names (kind, lane, micro-batch of that lane)"""
    from diffusion_pipe_amd.engine import schedule as sched
    counts = sched.lane_micro_batches(micro_batches, lanes)
    seqs = []
    for stage in range(stages):
        scheds = [sched.TrainSchedule(micro_batches=n, stages=stages, stage_id=stage) for n in counts]
        held = [dict() for _ in counts]                  # lane -> {pipe buffer id: micro-batch it holds}
        loaded = [0] * len(counts)
        ops = []
        for lane, cmds in sched.interleave_lanes(scheds):
            for cmd in cmds:
                b = cmd.kwargs.get('buffer_id')
                if isinstance(cmd, sched.RecvActivation):
                    held[lane][b] = loaded[lane]; loaded[lane] += 1
                    ops.append(('recv', stage - 1, ('act', lane, held[lane][b])))
                elif isinstance(cmd, sched.LoadMicroBatch) and stage == 0:
                    held[lane][b] = loaded[lane]; loaded[lane] += 1
                elif isinstance(cmd, sched.SendActivation):
                    ops.append(('send', stage + 1, ('act', lane, held[lane][b])))
                elif isinstance(cmd, sched.RecvGrad):
                    ops.append(('recv', stage + 1, ('grad', lane, held[lane][b])))
                elif isinstance(cmd, sched.SendGrad):
                    ops.append(('send', stage - 1, ('grad', lane, held[lane][b])))
        seqs.append(ops)
    return counts, seqs


@pytest.mark.parametrize('stages,micro_batches,lanes', [(2, 4, 2), (2, 16, 3), (3, 7, 2), (4, 7, 3), (4, 32, 3), (8, 64, 3), (8, 9, 4), (5, 5, 5), (2, 3, 8)])
def test_interleaved_pipeline_lanes_keep_the_wire_order_and_cannot_deadlock(stages, micro_batches, lanes):
    counts, seqs = _lane_wire_ops(stages, micro_batches, lanes)
    assert sum(counts) == micro_batches
    # (1) per ordered pair of neighbours the k-th message sent is the k-th message the receiver expects (in-order links carry no tags)
    for a in range(stages):
        for b in (a - 1, a + 1):
            if 0 <= b < stages:
                sent = [m for op, peer, m in seqs[a] if op == 'send' and peer == b]
                expected = [m for op, peer, m in seqs[b] if op == 'recv' and peer == a]
                assert sent == expected, (a, b)
    # every micro-batch of every lane crosses every boundary once in each direction
    for a in range(stages - 1):
        acts = {m for op, peer, m in seqs[a] if op == 'send' and peer == a + 1}
        assert acts == {('act', lane, k) for lane, n in enumerate(counts) for k in range(n)}
    # (2) blocking receives + asynchronous sends (gloo / the host-staged link): runs to completion
    pos = [0] * stages
    chan = {}
    progressed = True
    while progressed:
        progressed = False
        for s in range(stages):
            while pos[s] < len(seqs[s]):
                op, peer, m = seqs[s][pos[s]]
                if op == 'send':
                    chan.setdefault((s, peer), []).append(m)
                elif chan.get((peer, s)):
                    assert chan[(peer, s)].pop(0) == m
                else:
                    break
                pos[s] += 1
                progressed = True
    assert all(pos[s] == len(seqs[s]) for s in range(stages)), 'deadlock under blocking receives'
    # (3) full rendezvous, every stage strictly in issue order (the most conservative model of RCCL send / recv kernels on ONE communication stream: an
    # operation retires only together with its partner, and nothing of a stage overtakes it): runs to completion
    pos = [0] * stages
    progressed = True
    while progressed:
        progressed = False
        for s in range(stages):
            if pos[s] == len(seqs[s]):
                continue
            op, peer, m = seqs[s][pos[s]]
            if pos[peer] < len(seqs[peer]):
                pop, ppeer, pm = seqs[peer][pos[peer]]
                if ppeer == s and pm == m and pop != op:
                    pos[s] += 1; pos[peer] += 1
                    progressed = True
    assert all(pos[s] == len(seqs[s]) for s in range(stages)), 'deadlock under rendezvous in issue order'
