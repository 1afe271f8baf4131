"""Size-independent properties of the integer / index logic (hypothesis): the 1F1B instruction stream is a valid pipeline for any
(stages, micro-batches), pipe buffers are never overwritten while their micro-batch is in flight, balanced partitions are optimal,
micro-batch splitting and the cache round-trip are lossless."""
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from diffusion_pipe_amd import data
from diffusion_pipe_amd.cache import Cache
from diffusion_pipe_amd.engine import module as pm
from diffusion_pipe_amd.engine import schedule as ps


def _steps(cls, mbs, stages, stage):
    return [[(c.name, c.kwargs.get('buffer_id')) for c in st_] for st_ in cls(mbs, stages, stage).steps()]


@settings(max_examples=60, deadline=None, derandomize=True)
@given(stages=st.integers(1, 8), mbs=st.integers(1, 24))
def test_train_schedule_is_a_valid_pipeline_for_any_shape(stages, mbs):
    per_stage = [_steps(ps.TrainSchedule, mbs, stages, s) for s in range(stages)]
    assert len({len(s) for s in per_stage}) == 1 and len(per_stage[0]) == 2 * (mbs + stages - 1)
    for s, steps in enumerate(per_stage):
        flat = [c for step in steps for c in step]
        fwd = [b for n, b in flat if n == 'ForwardPass']
        bwd = [b for n, b in flat if n == 'BackwardPass']
        assert len(fwd) == len(bwd) == mbs
        # a pipe buffer holds one micro-batch from its forward to its backward: no forward may reuse it in between
        live = set()
        order_f, order_b = [], []
        for n, b in flat:
            if n == 'ForwardPass':
                assert b not in live, (s, b)
                live.add(b)
                order_f.append(b)
            elif n == 'BackwardPass':
                assert b in live
                live.remove(b)
                order_b.append(b)
        assert not live and order_f == order_b                      # backwards retire micro-batches in forward order
        assert len(live) <= ps.TrainSchedule(mbs, stages, s).num_pipe_buffers()
        # the step end appears once, last, on every stage
        assert [n for n, _ in steps[-1]][-3:] == ['ReduceTiedGrads', 'ReduceGrads', 'OptimizerStep']
        assert sum(1 for n, _ in flat if n == 'OptimizerStep') == 1
        # neighbours post matching transfers in the same step (the in-order communication stream cannot deadlock)
        if s + 1 < stages:
            nxt = per_stage[s + 1]
            for i, (a, b) in enumerate(zip(steps, nxt)):
                assert sum(n == 'SendActivation' for n, _ in a) == sum(n == 'RecvActivation' for n, _ in b), i
                assert sum(n == 'RecvGrad' for n, _ in a) == sum(n == 'SendGrad' for n, _ in b), i
        loads = sum(1 for n, _ in flat if n == 'LoadMicroBatch')
        assert loads == (mbs if s in (0, stages - 1) else 0)


@settings(max_examples=60, deadline=None, derandomize=True)
@given(stages=st.integers(1, 8), mbs=st.integers(1, 24))
def test_inference_schedule_covers_every_micro_batch_once(stages, mbs):
    for s in range(stages):
        flat = [c for step in _steps(ps.InferenceSchedule, mbs, stages, s) for c in step]
        assert sum(1 for n, _ in flat if n == 'ForwardPass') == mbs
        assert not any(n in ('BackwardPass', 'OptimizerStep', 'SendGrad', 'RecvGrad') for n, _ in flat)


@settings(max_examples=80, deadline=None, derandomize=True)
@given(weights=st.lists(st.integers(0, 50), min_size=1, max_size=9), parts=st.integers(1, 5))
def test_balanced_partition_is_a_valid_partition(weights, parts):
    """DeepSpeed's `partition_balanced` is a DP on (heaviest - lightest) part weight -- not the bottleneck optimum -- so the properties
    are structural: contiguous cover, no empty stage once there are enough layers, agreement with the oracle restatement."""
    from oracle import intlogic as ol
    got = pm.partition_balanced(weights, parts)
    assert got == ol.partition_balanced(weights, parts)
    assert got[0] == 0 and got[-1] == len(weights) and len(got) == parts + 1
    if len(weights) >= parts:
        assert all(a < b for a, b in zip(got, got[1:]))
    else:
        assert all(a <= b for a, b in zip(got, got[1:]))


@settings(max_examples=40, deadline=None, derandomize=True)
@given(pieces=st.integers(1, 6), rows_per=st.integers(1, 4), cols=st.integers(1, 5))
def test_split_batch_is_lossless(pieces, rows_per, cols):
    n = pieces * rows_per
    f = (torch.arange(n * cols, dtype=torch.float32).view(n, cols), torch.arange(n))
    l = (torch.arange(n * cols, dtype=torch.float32).view(n, cols) + 0.5, None)
    out = data.split_batch((f, l), pieces)
    assert len(out) == pieces
    assert torch.equal(torch.cat([p[0][0] for p in out]), f[0]) and torch.equal(torch.cat([p[0][1] for p in out]), f[1])
    assert torch.equal(torch.cat([p[1][0] for p in out]), l[0]) and all(p[1][1].numel() == 0 for p in out)


@settings(max_examples=15, deadline=None, derandomize=True)
@given(sizes=st.lists(st.integers(1, 400), min_size=1, max_size=12), shard_kb=st.integers(1, 4))
def test_cache_round_trip(tmp_path_factory, sizes, shard_kb):
    d = tmp_path_factory.mktemp('cache')
    cache = Cache(d, 'fp', shard_size_gb=shard_kb * 1e-6)
    items = [{'x': torch.arange(n, dtype=torch.float32) * (i + 1), 'caption': f'c{i}', 'mask': None} for i, n in enumerate(sizes)]
    for it in items:
        cache.add(it)
    cache.finalize_current_shard()
    cache.close()
    again = Cache(d, 'fp')
    assert len(again) == len(items)
    for i in reversed(range(len(items))):
        got = again[i]
        assert torch.equal(got['x'], items[i]['x']) and got['caption'] == items[i]['caption'] and got['mask'] is None
    again.close()


def _comm_ops(cls, mbs, stages, stage):
    kinds = {'SendActivation': ('send', +1, 'act'), 'RecvActivation': ('recv', -1, 'act'), 'SendGrad': ('send', -1, 'grad'), 'RecvGrad': ('recv', +1, 'grad')}
    return [(kinds[c.name][0], stage + kinds[c.name][1], kinds[c.name][2]) for step in cls(mbs, stages, stage).steps() for c in step if c.name in kinds]


def test_instruction_order_is_deadlock_free_under_blocking_rendezvous():
    """RCCL point-to-point semantics on ONE in-order communication stream per rank: a send / recv at the head of a rank's queue completes
    only when the peer's queue head is the matching recv / send.  Every stage's transfer sequence (train and eval schedules) must drain
    for every pipeline depth the metric names (1..8) and any micro-batch count -- gloo's non-blocking isend would hide an ordering bug."""
    for stages in range(1, 9):
        for mbs in list(range(1, 20)) + [24, 32, 48]:
            for cls in (ps.TrainSchedule, ps.InferenceSchedule):
                queues = [_comm_ops(cls, mbs, stages, s) for s in range(stages)]
                heads = [0] * stages
                progress = True
                while progress:
                    progress = False
                    for a in range(stages):
                        if heads[a] >= len(queues[a]):
                            continue
                        kind, b, what = queues[a][heads[a]]
                        assert 0 <= b < stages
                        if heads[b] < len(queues[b]):
                            kb, pb, wb = queues[b][heads[b]]
                            if pb == a and wb == what and {kind, kb} == {'send', 'recv'}:
                                heads[a] += 1
                                heads[b] += 1
                                progress = True
                assert all(h == len(q) for h, q in zip(heads, queues)), (cls.__name__, stages, mbs)
