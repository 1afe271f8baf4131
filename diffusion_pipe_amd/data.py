"""Host-side data plumbing of the hot path: micro-batch splitting, first->last stage target hand-off, and the integer
bucket / iteration-order logic.  Pure index arithmetic: results are kept
bit-exact with the reference (pinned against oracle/intlogic.py by tests/test_intlogic.py).
"""
import hashlib
import math
import random

import numpy as np
import torch
import torch.distributed as dist

ROUND_DECIMAL_DIGITS = 3


# ------------------------------------------------------------------------------------------------- micro-batches
def split_batch(batch, pieces):
    """(features, label) tuples of tensors -> list of `pieces` micro-batches along dim 0; None -> empty tensor
    (utils/dataset.py:1273-1281)."""
    features, label = batch
    split_size = features[0].size(0) // pieces

    def parts(t):
        return torch.split(t, split_size) if t is not None else [torch.tensor([])] * pieces

    f_parts = [parts(t) for t in features]
    l_parts = [parts(t) for t in label]
    return [(tuple(fp[i] for fp in f_parts), tuple(lp[i] for lp in l_parts)) for i in range(pieces)]


def stack_micro_batches(micro_batches, k):
    """The inverse of `split_batch` over runs of k consecutive micro-batches: [(features, label)] * n -> [(features, label)] * ceil(n / k), every tensor
    concatenated along dim 0 (the axis split_batch cut, utils/dataset.py:1273-1281); the empty tensors that stand for None stay one empty tensor.  Used by the
    engine option `stack_micro_batches`: k micro-batches of size b run as ONE pass of size k b -- same samples, same per-sample loss terms, a k-times larger M in
    every GEMM (one read of the weights per k samples)."""
    def cat(ts):
        if not torch.is_tensor(ts[0]):
            return ts[0]
        if all(t.numel() == 0 for t in ts):
            return ts[0]
        return torch.cat(list(ts), 0)

    def merge(parts):
        if torch.is_tensor(parts[0]) or not isinstance(parts[0], (tuple, list)):
            return cat(parts)
        return tuple(cat([p[j] for p in parts]) for j in range(len(parts[0])))

    out = []
    for i in range(0, len(micro_batches), k):
        group = micro_batches[i:i + k]
        out.append(group[0] if len(group) == 1 else (merge([g[0] for g in group]), merge([g[1] for g in group])))
    return out


class StackedIterator:
    """Iterator view for `stack_micro_batches`: every next() pulls k micro-batches from the underlying iterator and returns them stacked."""

    def __init__(self, it, k):
        self.it, self.k = it, k

    def __iter__(self):
        return self

    def __next__(self):
        return stack_micro_batches([next(self.it) for _ in range(self.k)], self.k)[0]


def broadcast_target(target, engine, device=None):
    """Send the noise-dependent target from the first to the last pipeline stage (utils/dataset.py:1387-1405)."""
    if not engine.is_pipe_parallel:
        return target
    assert engine.is_first_stage() or engine.is_last_stage()
    grid = engine.grid
    src, dst = grid.stage_to_global(0), grid.stage_to_global(engine.num_stages - 1)
    target = target.to(device or engine.device)
    if engine.is_first_stage():
        dist.send(target, dst)
    else:
        dist.recv(target, src)
    return target


def get_data_iterator_for_step(dataloader, engine, num_micro_batches=None):
    """Pre-pull the step's micro-batches on the first / last stage only (train.py:164-173)."""
    n = num_micro_batches or engine.gradient_accumulation_steps()      # micro-batches the iterator hands over (not passes: engine `stack_micro_batches`)
    if not (engine.is_first_stage() or engine.is_last_stage()):
        return None
    it = iter(dataloader)
    return iter([next(it) for _ in range(n)])


class MicroBatchLoader:
    """Endless iterator of micro-batches with epoch tracking -- the PipelineDataLoader contract
    (utils/dataset.py:1302-1435) over any re-iterable `dataset` of already-collated batches.

    `prepare_inputs(batch, timestep_quantile=...) -> (features, (targets..., mask))` is the adapter hook."""

    def __init__(self, dataset, engine, gradient_accumulation_steps, prepare_inputs):
        if len(dataset) == 0:
            raise RuntimeError('Processed dataset was empty.')
        self.dataset = dataset
        self.engine = engine
        self.gradient_accumulation_steps = gradient_accumulation_steps
        self.prepare_inputs = prepare_inputs
        self.eval_quantile = None
        self.skip_first_n_batches = 0
        self.iter_called = False
        self.reset()

    def reset(self):
        self.epoch = 1
        self.num_batches_pulled = 0
        self.next_micro_batch = None
        self.data = self._pull()

    def set_eval_quantile(self, quantile):
        self.eval_quantile = quantile

    def __iter__(self):
        self.iter_called = True
        return self

    def __len__(self):
        return len(self.dataset) * self.gradient_accumulation_steps

    def __next__(self):
        if self.next_micro_batch is None:
            self.next_micro_batch = next(self.data)
        ret = self.next_micro_batch
        try:
            self.next_micro_batch = next(self.data)
        except StopIteration:
            self.skip_first_n_batches = 0
            self.data = self._pull()
            self.num_batches_pulled = 0
            self.next_micro_batch = None
            self.epoch += 1
        return ret

    def _pull(self):
        for idx in range(self.skip_first_n_batches, len(self.dataset)):
            batch = self.dataset[idx]
            features, label = self.prepare_inputs(batch, timestep_quantile=self.eval_quantile)
            *targets, mask = label
            label = (*[broadcast_target(t, self.engine) for t in targets], mask)
            self.num_batches_pulled += 1
            yield from split_batch((features, label), self.gradient_accumulation_steps)

    def sync_epoch(self):
        if not (dist.is_available() and dist.is_initialized()):
            return
        result = [None] * dist.get_world_size()
        dist.all_gather_object(result, self.epoch)
        self.epoch = max(result)

    def state_dict(self):
        return {'epoch': self.epoch, 'num_batches_pulled': self.num_batches_pulled}

    def load_state_dict(self, state_dict):
        assert not self.iter_called
        self.epoch = state_dict['epoch']
        self.num_batches_pulled = state_dict['num_batches_pulled'] - 1   # one micro-batch is always pre-pulled
        self.skip_first_n_batches = self.num_batches_pulled
        self.data = self._pull()


# ---------------------------------------------------------------------------------------------- bucket arithmetic
def round_to_nearest_multiple(x, multiple):
    """Python banker's rounding of x / multiple (utils/common.py:106-107)."""
    return int(round(x / multiple) * multiple)


def round_down_to_multiple(x, multiple):
    return int((x // multiple) * multiple)


def dedup_and_sort(values):
    # round() is applied to the elements as given (numpy scalars round the numpy way, Python floats the Python way)
    return np.array(sorted({round(v, ROUND_DECIMAL_DIGITS) for v in values}))


def seed_from_hash(item):
    return int(hashlib.md5(str(item).encode()).hexdigest(), 16) % int(1e9)


def shuffle_with_seed(items, seed=None):
    """In-place shuffle that leaves the global `random` state untouched (utils/dataset.py:41-45)."""
    state = random.getstate()
    random.seed(seed)
    random.shuffle(items)
    random.setstate(state)


def make_ar_buckets(min_ar, max_ar, num_ar_buckets):
    return dedup_and_sort(np.geomspace(min_ar, max_ar, num=num_ar_buckets))


def size_bucket_for(ar, frames, resolution, round_to_multiple):
    """(w, h, frames) of an aspect-ratio bucket at a square-equivalent resolution (utils/dataset.py:419-425)."""
    area = resolution ** 2
    w = math.sqrt(area * ar)
    h = area / w
    return (round_to_nearest_multiple(w, round_to_multiple), round_to_nearest_multiple(h, round_to_multiple), frames)


def find_closest_ar_bucket(log_ar, frames, is_video, ars, frame_buckets):
    """Nearest AR in log space + a frame bucket no longer than the clip (utils/dataset.py:838-852).
    Returns (ar, frame_bucket) or None.  Index semantics (argmin over the non-negative differences, then indexing
    the full frame_buckets array with it) follow the reference exactly."""
    ars = np.asarray(ars)
    frame_buckets = np.asarray(frame_buckets)
    i = int(np.argmin(np.abs(log_ar - np.log(ars))))
    diffs = frames - frame_buckets
    fits = diffs[diffs >= 0]
    if len(fits) == 0:
        return None
    j = int(np.argmin(fits))
    if is_video and frame_buckets[j] == 1:
        return None
    return (ars[i], frame_buckets[j])


def find_closest_size_bucket(log_ar, frames, is_video, size_buckets):
    """First explicit (w, h, frames) bucket, in order of AR distance (stable), that the clip can fill
    (utils/dataset.py:854-871)."""
    size_buckets = np.asarray(size_buckets)
    log_ars = np.log(size_buckets[:, 0] / size_buckets[:, 1])
    order = np.argsort(np.abs(log_ar - log_ars), kind='stable')
    for sb in size_buckets[order]:
        if is_video and sb[-1] == 1:
            continue
        if frames >= sb[-1]:
            return sb
    return None


def pick_global_batch_size(size_bucket, batch_size_by_resolution):
    """{None: bs} or {resolution: bs}: nearest resolution to sqrt(w*h) wins, first on ties (utils/dataset.py:362-375)."""
    if None in batch_size_by_resolution:
        return batch_size_by_resolution[None]
    bucket_size = math.sqrt(size_bucket[-2] * size_bucket[-3])
    best, best_diff = None, float('inf')
    for size, bs in batch_size_by_resolution.items():
        diff = abs(size - bucket_size)
        if diff < best_diff:
            best, best_diff = bs, diff
    return best


def batched_iteration_order(dataset_lengths, global_batch_size):
    """Global (dataset_idx, item_idx) order of a size bucket spanning several datasets, truncated to whole global
    batches (utils/dataset.py:347-361,386-390)."""
    order = [i for i, n in enumerate(dataset_lengths) for _ in range(n)]
    shuffle_with_seed(order, 0)
    seen = [0] * len(dataset_lengths)
    pairs = []
    for d in order:
        pairs.append((d, seen[d]))
        seen[d] += 1
    keep = (len(pairs) // global_batch_size) * global_batch_size
    return np.array(pairs[:keep], dtype=np.int64).reshape(-1, 2)


def dp_batch_slice(batch_idx, global_batch_size, dp_rank, dp_world_size):
    """[start, end) of data-parallel rank `dp_rank` inside global batch `batch_idx` (utils/dataset.py:381-384)."""
    per_rank = global_batch_size // dp_world_size
    start = batch_idx * global_batch_size + dp_rank * per_rank
    return start, start + per_rank


# ------------------------------------------------------------------------------------------------ timestep helpers
def get_t_distribution(model_config):
    """10 000-quantile table of the timestep distribution (utils/common.py:124-147)."""
    method = model_config.get('timestep_sample_method', 'logit_normal')
    if method == 'logit_normal':
        d = torch.distributions.normal.Normal(0, 1)
    elif method == 'uniform':
        d = torch.distributions.uniform.Uniform(0, 1)
    else:
        raise NotImplementedError()
    n = 10_000
    t = d.icdf(torch.linspace(1 / n, 1 - 1 / n, n))
    if method == 'logit_normal':
        t = torch.sigmoid(t * model_config.get('sigmoid_scale', 1.0))
    return t


def slice_t_distribution(t, min_t=0.0, max_t=1.0):
    return t[torch.searchsorted(t, min_t).item():torch.searchsorted(t, max_t).item()]


def sample_t(t, batch_size, quantile=None):
    if quantile is not None:
        i = (torch.full((batch_size,), quantile) * len(t)).to(torch.int32)
    else:
        i = torch.randint(0, len(t), size=(batch_size,))
    return t[i]
