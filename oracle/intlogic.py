"""ORACLE (test infrastructure, never imported by the product): scalar restatement of the reference's integer /
index logic on the train_batch path.  Plain Python loops, one function per reference routine, each citing the lines
it follows.  DeepSpeed-owned pieces (not vendored in /root/reference, pinned deepspeed==0.18.4 in
requirements.txt:1) are restated from the published algorithm and marked [3P, parity unpinned].
"""
import hashlib
import math
import random

import numpy as np


# ---- layer -> stage partition --------------------------------------------------------------------------------
def manual_partition(num_layers, num_stages, partition_split):
    """utils/pipeline.py:16-25: boundaries = [0] + split + [L]; len(split) must be num_stages - 1."""
    assert len(partition_split) == num_stages - 1
    return [0] + list(partition_split) + [num_layers]


def default_partition_split(num_layers, num_stages):
    """train.py:607: config.get('partition_split', [len(layers) / num_stages]) (a float, as in the reference)."""
    return [num_layers / num_stages]


def partition_uniform(num_items, num_parts):
    """[3P] deepspeed.runtime.utils.partition_uniform."""
    parts = [0] * (num_parts + 1)
    if num_items <= num_parts:
        for p in range(num_parts + 1):
            parts[p] = min(p, num_items)
        return parts
    chunksize = num_items // num_parts
    residual = num_items - (chunksize * num_parts)
    parts = [p * chunksize for p in range(num_parts + 1)]
    for i in range(residual):
        for q in range(i + 1, num_parts + 1):
            parts[q] += 1
    return parts


def partition_balanced(weights, num_parts):
    """[3P] deepspeed.runtime.utils.partition_balanced: linear-partition DP, objective (max part - min part),
    sequential scan over the split point with a '>=' update (ties -> latest split)."""
    n, m = len(weights), num_parts
    if n <= m:
        return partition_uniform(n, m)
    inf = float('inf')
    dp_max = [[inf] * (m + 1) for _ in range(n + 1)]
    dp_min = [[inf] * (m + 1) for _ in range(n + 1)]
    dp_cost = [[inf] * (m + 1) for _ in range(n + 1)]
    position = [[0] * (m + 1) for _ in range(n + 1)]
    prefix = [0.0] * (n + 1)
    for i, w in enumerate(weights):
        prefix[i + 1] = prefix[i] + float(w)
    dp_max[0][0] = 0
    dp_cost[0][0] = 0
    for i in range(1, n + 1):
        for j in range(1, min(i, m) + 1):
            for k in range(i):
                seg = prefix[i] - prefix[k]
                max_sum = max(dp_max[k][j - 1], seg)
                min_sum = min(dp_min[k][j - 1], seg)
                cost = max_sum - min_sum
                if dp_cost[i][j] >= cost:
                    dp_cost[i][j] = cost
                    dp_max[i][j] = max_sum
                    dp_min[i][j] = min_sum
                    position[i][j] = k
    parts = [n]
    for i in reversed(range(1, m + 1)):
        parts.append(position[parts[-1]][i])
    parts.reverse()
    return parts


# ---- 1F1B schedule ---------------------------------------------------------------------------------------------
def _step_to_micro_batch(step_id, stage_id, stages):
    """[3P] deepspeed.runtime.pipe.schedule.TrainSchedule._step_to_micro_batch and its four helpers."""
    even_step, even_stage = step_id % 2 == 0, stage_id % 2 == 0
    if even_step and even_stage:
        return step_id // 2 - stage_id // 2, True
    if not even_step and not even_stage:
        return (step_id - 1) // 2 - stage_id // 2, True
    if even_step and not even_stage:
        return step_id // 2 - stages + (stage_id + 1) // 2, False
    return ((step_id - 1) // 2) - stages + 1 + stage_id // 2, False


def train_schedule(micro_batches, stages, stage_id):
    """utils/patches.py:113-160 (the reference's patched TrainSchedule.steps) as a list of steps, each a list of
    (instruction_name, buffer_id or None)."""
    num_buffers = max(2, min(stages - stage_id, micro_batches))      # [3P] TrainSchedule.num_pipe_buffers
    valid_mb = lambda mb: 0 <= mb < micro_batches
    valid_stage = lambda s: 0 <= s < stages
    prev_stage, next_stage = stage_id - 1, stage_id + 1
    out = []
    prev_mb = -1
    total_steps = 2 * (micro_batches + stages - 1)
    for step_id in range(total_steps):
        mb, is_forward = _step_to_micro_batch(step_id, stage_id, stages)
        prev_buffer = prev_mb % num_buffers if valid_mb(prev_mb) else None
        curr_buffer = mb % num_buffers if valid_mb(mb) else None
        cmds = []
        if stage_id == 0 or stage_id == stages - 1:
            if is_forward and valid_mb(mb):
                cmds.append(('LoadMicroBatch', curr_buffer))
        if is_forward:
            if valid_mb(prev_mb) and valid_stage(prev_stage):
                cmds.append(('SendGrad', prev_buffer))
            if valid_mb(mb) and valid_stage(prev_stage):
                cmds.append(('RecvActivation', curr_buffer))
        else:
            if valid_mb(mb) and valid_stage(next_stage):
                cmds.append(('RecvGrad', curr_buffer))
            if valid_mb(prev_mb) and valid_stage(next_stage):
                cmds.append(('SendActivation', prev_buffer))
        if valid_mb(mb):
            cmds.append(('ForwardPass' if is_forward else 'BackwardPass', curr_buffer))
        if step_id == total_steps - 1:
            cmds += [('ReduceTiedGrads', None), ('ReduceGrads', None), ('OptimizerStep', None)]
        prev_mb = mb
        out.append(cmds)
    return out


def inference_schedule(micro_batches, stages, stage_id):
    """[3P] deepspeed.runtime.pipe.schedule.InferenceSchedule.steps."""
    out = []
    valid_mb = lambda mb: 0 <= mb < micro_batches
    for step_id in range(micro_batches + stages - 1):
        cmds = []
        mb = step_id - stage_id
        if stage_id % 2 == 0:
            recv_buf, send_buf = step_id % 2, (step_id + 1) % 2
        else:
            recv_buf, send_buf = (step_id + 1) % 2, step_id % 2
        if stage_id == 0 or stage_id == stages - 1:
            if valid_mb(mb):
                cmds.append(('LoadMicroBatch', recv_buf))
        if stage_id % 2 == 0:
            if stage_id + 1 < stages and valid_mb(mb - 1):
                cmds.append(('SendActivation', send_buf))
            if stage_id - 1 >= 0 and valid_mb(mb):
                cmds.append(('RecvActivation', recv_buf))
        else:
            if stage_id - 1 >= 0 and valid_mb(mb):
                cmds.append(('RecvActivation', recv_buf))
            if stage_id + 1 < stages and valid_mb(mb - 1):
                cmds.append(('SendActivation', send_buf))
        if valid_mb(mb):
            cmds.append(('ForwardPass', recv_buf))
        out.append(cmds)
    return out


# ---- topology --------------------------------------------------------------------------------------------------
def rank_of(stage, dp_rank, num_stages, num_dp):
    """[3P] PipeDataParallelTopology(axes=['pipe','data']): row-major rank = pipe * num_dp + data."""
    return stage * num_dp + dp_rank


# ---- bucket / iteration-order arithmetic ---------------------------------------------------------------------------
ROUND_DECIMAL_DIGITS = 3


def round_to_nearest_multiple(x, multiple):
    """utils/common.py:106-107."""
    return int(round(x / multiple) * multiple)


def dedup_and_sort(values):
    """utils/dataset.py:74-78."""
    values = set(round(x, ROUND_DECIMAL_DIGITS) for x in values)
    values = list(values)
    values.sort()
    return np.array(values)


def seed_from_hash(item):
    """utils/dataset.py:81-82."""
    return int(hashlib.md5(str.encode(str(item))).hexdigest(), 16) % int(1e9)


def shuffle_with_seed(l, seed=None):
    """utils/dataset.py:41-45."""
    rng_state = random.getstate()
    random.seed(seed)
    random.shuffle(l)
    random.setstate(rng_state)


def size_bucket(ar, frames, res, round_to_multiple):
    """utils/dataset.py:419-425."""
    area = res ** 2
    w = math.sqrt(area * ar)
    h = area / w
    w = round_to_nearest_multiple(w, round_to_multiple)
    h = round_to_nearest_multiple(h, round_to_multiple)
    return (w, h, frames)


def find_closest_ar_bucket(log_ar, frames, is_video, ars, frame_buckets):
    """utils/dataset.py:838-852."""
    log_ars = np.log(ars)
    i = np.argmin(np.abs(log_ar - log_ars))
    diffs = frames - frame_buckets
    positive_diffs = diffs[diffs >= 0]
    if len(positive_diffs) == 0:
        return None
    j = np.argmin(positive_diffs)
    if is_video and frame_buckets[j] == 1:
        return None
    return (ars[i], frame_buckets[j])


def find_closest_size_bucket(log_ar, frames, is_video, size_buckets):
    """utils/dataset.py:854-871 with self.log_ars = log(w / h) of the explicit buckets (utils/dataset.py:495-496,508)."""
    log_ars = np.log(np.array([w / h for w, h, _ in size_buckets]))
    ar_diffs = np.abs(log_ar - log_ars)
    candidates = np.asarray(size_buckets)[np.argsort(ar_diffs, kind='stable')]
    found = False
    for sb in candidates:
        if is_video and sb[-1] == 1:
            continue
        if frames >= sb[-1]:
            found = True
            break
    if not found:
        return None
    return sb


def iteration_order(dataset_lengths, global_batch_size):
    """utils/dataset.py:347-361 + _make_divisible_by (:386-390)."""
    order = []
    for i, n in enumerate(dataset_lengths):
        order.extend([i] * n)
    shuffle_with_seed(order, 0)
    cumulative = [0] * len(dataset_lengths)
    for k, d in enumerate(order):
        order[k] = (d, cumulative[d])
        cumulative[d] += 1
    new_length = (len(order) // global_batch_size) * global_batch_size
    return order[:new_length]


def pick_global_batch_size(size_bucket_, batch_size_dict):
    """utils/dataset.py:362-375."""
    if None in batch_size_dict:
        return batch_size_dict[None]
    bucket_size = math.sqrt(size_bucket_[-2] * size_bucket_[-3])
    min_diff = float('inf')
    chosen = None
    for size, bs in batch_size_dict.items():
        diff = abs(size - bucket_size)
        if diff < min_diff:
            min_diff = diff
            chosen = bs
    return chosen


def dp_slice(idx, global_batch_size, dp_rank, dp_world):
    """utils/dataset.py:381-384."""
    batch_size = global_batch_size // dp_world
    start_idx = idx * global_batch_size + dp_rank * batch_size
    return start_idx, start_idx + batch_size
