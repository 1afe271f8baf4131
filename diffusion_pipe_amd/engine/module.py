"""PipelineModule / ManualPipelineModule: layer list -> per-stage layer range.

Drop-in for the construction the reference performs at train.py:605-617
(`ManualPipelineModule(layers=..., num_stages=..., partition_method=..., manual_partition_split=..., loss_fn=...,
dynamic_shape=True, activation_checkpoint_interval=1, checkpointable_layers=..., activation_checkpoint_func=...)`),
i.e. for DeepSpeed's `PipelineModule` plus utils/pipeline.py:11-53.  The layer -> stage boundary logic is integer
arithmetic and is kept bit-exact with the reference (tests pin it against oracle/intlogic.py).
"""
import re

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from .topology import PipeDataParallelTopology, PipelineParallelGrid


def partition_uniform(num_items, num_parts):
    """Boundaries of `num_parts` contiguous chunks; the first (num_items % num_parts) chunks get one extra item."""
    if num_items <= num_parts:
        return [min(p, num_items) for p in range(num_parts + 1)]
    chunk, residual = divmod(num_items, num_parts)
    return [p * chunk + min(p, residual) for p in range(num_parts + 1)]


def partition_balanced(weights, num_parts):
    """Linear-partition DP minimising (heaviest part - lightest part); ties resolved toward the LATEST split point,
    as DeepSpeed's `partition_balanced` does (the reference's default partition_method='parameters')."""
    n, m = len(weights), num_parts
    if n <= m:
        return partition_uniform(n, m)
    prefix = np.zeros(n + 1, dtype=np.float64)
    prefix[1:] = np.cumsum(np.asarray(weights, dtype=np.float64))
    INF = np.inf
    best_max = np.full((n + 1, m + 1), INF)
    best_min = np.full((n + 1, m + 1), INF)
    best_cost = np.full((n + 1, m + 1), INF)
    split = np.zeros((n + 1, m + 1), dtype=np.int64)
    best_max[0, 0] = 0.0
    best_cost[0, 0] = 0.0
    for i in range(1, n + 1):
        seg = prefix[i] - prefix[:i]                       # weight of layers k..i-1 for every split point k < i
        for j in range(1, min(i, m) + 1):
            cand_max = np.maximum(best_max[:i, j - 1], seg)
            cand_min = np.minimum(best_min[:i, j - 1], seg)
            cost = cand_max - cand_min                      # inf - x = inf, inf - inf = nan for unreachable states
            cost = np.where(np.isnan(cost), INF, cost)
            lo = cost.min()
            if not (best_cost[i, j] >= lo):
                continue
            k = int(np.flatnonzero(cost == lo)[-1])         # last minimiser == sequential ">=" update order
            best_cost[i, j], best_max[i, j], best_min[i, j], split[i, j] = lo, cand_max[k], cand_min[k], k
    parts = [n]
    for j in range(m, 0, -1):
        parts.append(int(split[parts[-1], j]))
    parts.reverse()
    return parts


def _layer_name(layer):
    if isinstance(layer, nn.Module):
        return layer.__class__.__name__
    return getattr(layer, '__name__', layer.__class__.__name__)


class PipelineModule(nn.Module):
    def __init__(self, layers, num_stages=None, topology=None, loss_fn=None, seed_layers=False, seed_fn=None,
                 base_seed=1234, partition_method='parameters', activation_checkpoint_interval=0,
                 activation_checkpoint_func=None, checkpointable_layers=None, dynamic_shape=False):
        super().__init__()
        if num_stages is None and topology is None:
            raise RuntimeError('must provide num_stages or topology')
        self.micro_offset = 0
        self.loss_fn = loss_fn
        self.checkpointable_layers = checkpointable_layers
        if checkpointable_layers is not None:
            assert isinstance(checkpointable_layers, list), 'param `checkpointable_layers` must be type of list.'
        self.dynamic_shape = dynamic_shape
        self.activation_checkpoint_interval = activation_checkpoint_interval
        self.activation_checkpoint_func = activation_checkpoint_func or torch.utils.checkpoint.checkpoint

        initialized = dist.is_available() and dist.is_initialized()
        self.global_rank = dist.get_rank() if initialized else 0
        self.world_size = dist.get_world_size() if initialized else 1
        if topology is not None:
            self._topo = topology
            self.num_stages = topology.get_dim('pipe')
        else:
            self.num_stages = num_stages
            if self.world_size % num_stages != 0:
                raise RuntimeError(f'num_stages ({num_stages}) must divide distributed world size ({self.world_size})')
            self._topo = PipeDataParallelTopology(num_pp=num_stages, num_dp=self.world_size // num_stages)
        self.stage_id = self._topo.get_coord(self.global_rank).pipe
        self._grid = PipelineParallelGrid(topology=self._topo, global_rank=self.global_rank, world_size=self.world_size)

        self._layer_specs = list(layers)
        self._num_layers = len(self._layer_specs)
        self._local_start = 0
        self._local_stop = None
        self.parts = None
        self._partition_layers(method=partition_method)

        self.forward_funcs = []
        self.fwd_map = {}
        self._build()

    # ------------------------------------------------------------------------------------------- partitioning
    def _count_layer_params(self):
        """Per-layer parameter counts.  The reference monkeypatches DeepSpeed to count ALL parameters, trainable
        or not (train.py:81-90); that is the behaviour kept here."""
        counts = [0] * len(self._layer_specs)
        for idx, layer in enumerate(self._layer_specs):
            if isinstance(layer, nn.Module):
                counts[idx] = sum(p.numel() for p in layer.parameters())
        return counts

    def _find_layer_type(self, layername):
        regex = re.compile(layername, re.IGNORECASE)
        return [idx for idx, layer in enumerate(self._layer_specs) if regex.search(_layer_name(layer))]

    def _set_bounds(self, start=None, stop=None):
        self._local_start = start
        self._local_stop = stop

    def _partition_layers(self, method='uniform'):
        num_stages = self._topo.get_dim('pipe')
        stage_id = self._topo.get_coord(self.global_rank).pipe
        method = method.lower()
        if method == 'uniform':
            self.parts = partition_uniform(num_items=len(self._layer_specs), num_parts=num_stages)
        elif method == 'parameters':
            self.parts = partition_balanced(weights=self._count_layer_params(), num_parts=num_stages)
        elif method.startswith('type:'):
            layertype = method.split(':')[1]
            binary_weights = [0] * len(self._layer_specs)
            for idx in self._find_layer_type(layertype):
                binary_weights[idx] = 1
            self.parts = partition_balanced(weights=binary_weights, num_parts=num_stages)
        elif method == 'profile':
            raise NotImplementedError(f'Partitioning method {method} not implemented.')
        else:
            raise NotImplementedError(f'Partitioning method {method} not implemented.')
        if self.global_rank == 0:
            self._print_partition(num_stages)
        self._set_bounds(start=self.parts[stage_id], stop=self.parts[stage_id + 1])

    def _print_partition(self, num_stages):
        for stage in range(num_stages):
            start, stop = self.parts[stage], self.parts[stage + 1]
            print(f'stage={stage} layers={stop - start}')
            for idx, layer in enumerate(self._layer_specs[start:stop]):
                print(f'    {idx + start:2d}: {_layer_name(layer)}')
        if self.loss_fn:
            print(f'  loss: {getattr(self.loss_fn, "__name__", self.loss_fn.__class__.__name__)}')

    # ------------------------------------------------------------------------------------------- local layers
    def _build(self):
        specs = self._layer_specs
        for local_idx, layer in enumerate(specs[self._local_start:self._local_stop]):
            layer_idx = local_idx + self._local_start
            if isinstance(layer, nn.Module):
                name = str(layer_idx)
                self.forward_funcs.append(layer)
                self.fwd_map.update({name: len(self.forward_funcs) - 1})
                self.add_module(name, layer)
            else:
                self.forward_funcs.append(layer)      # bare callable (models/hunyuan_video.py:488)
        for p in self.parameters():
            p.ds_pipe_replicated = False

    def _is_checkpointable(self, funcs):
        if self.checkpointable_layers is not None:
            return all(f.__class__.__name__ in self.checkpointable_layers for f in funcs)
        params = [f.parameters() for f in funcs if isinstance(f, nn.Module)]
        return any(len(list(p)) > 0 for p in params)

    def forward(self, forward_input):
        self.micro_offset += 1

        def exec_range_func(start, end):
            def exec_func(*inputs):
                if len(inputs) == 1:
                    inputs = inputs[0]
                for layer in self.forward_funcs[start:end]:
                    inputs = layer(inputs)
                return inputs
            return exec_func

        if self.activation_checkpoint_interval == 0:
            return exec_range_func(0, len(self.forward_funcs))(forward_input)
        num_layers = len(self.forward_funcs)
        x = forward_input
        for start_idx in range(0, num_layers, self.activation_checkpoint_interval):
            end_idx = min(start_idx + self.activation_checkpoint_interval, num_layers)
            funcs = self.forward_funcs[start_idx:end_idx]
            if not isinstance(x, tuple):
                x = (x,)
            if self._is_checkpointable(funcs):
                x = self.activation_checkpoint_func(exec_range_func(start_idx, end_idx), *x)
            else:
                x = exec_range_func(start_idx, end_idx)(*x)
        return x

    # --------------------------------------------------------------------------------------------- accessors
    def topology(self):
        return self._topo

    def mpu(self):
        return self._grid

    def num_pipeline_stages(self):
        return self._topo.get_dim('pipe')

    def local_layer_range(self):
        return self._local_start, self._local_stop

    def ckpt_layer_path(self, ckpt_dir, local_layer_idx):
        idx = local_layer_idx + self._local_start
        return f'{ckpt_dir}/layer_{idx:02d}-model_states.pt'


class ManualPipelineModule(PipelineModule):
    """partition_method='manual' with explicit boundaries (utils/pipeline.py:11-53)."""

    def __init__(self, *args, manual_partition_split=None, **kwargs):
        self.manual_partition_split = manual_partition_split
        super().__init__(*args, **kwargs)

    def _partition_layers(self, method='uniform'):
        if method.lower() == 'manual' and self.manual_partition_split is not None:
            num_stages = self._topo.get_dim('pipe')
            stage_id = self._topo.get_coord(self.global_rank).pipe
            num_partitions = len(self.manual_partition_split)
            assert num_partitions == num_stages - 1, \
                f'partition_split must be length {num_stages-1} (pipeline_stages-1), was actually {num_partitions}'
            total_layers = len(self._layer_specs)
            self.parts = [0] + list(self.manual_partition_split) + [total_layers]
            if self.global_rank == 0:
                self._print_partition(num_stages)
            self._set_bounds(start=self.parts[stage_id], stop=self.parts[stage_id + 1])
        else:
            super()._partition_layers(method)
