"""Process topology for hybrid pipeline x data parallelism: one process per MI355X, RCCL groups over xGMI.

Mirrors the surface of DeepSpeed's `PipeDataParallelTopology` / `PipelineParallelGrid` that the reference reaches
into (train.py:632,821-838; utils/saver.py:59-60; utils/dataset.py:1389-1398; utils/patches.py:208-234).
Axes are ['pipe', 'data'], so global rank = stage_id * dp_world + dp_rank: a pipeline's stages sit `dp_world`
ranks apart and each data-parallel group is a contiguous run of ranks.
"""
from collections import namedtuple

import torch.distributed as dist

ProcessCoord = namedtuple('ProcessCoord', ['pipe', 'data'])


class PipeDataParallelTopology:
    def __init__(self, num_pp, num_dp):
        assert num_pp >= 1 and num_dp >= 1
        self.num_pp, self.num_dp = num_pp, num_dp
        self.axes = ['pipe', 'data']
        self.dims = [num_pp, num_dp]

    def world_size(self):
        return self.num_pp * self.num_dp

    def get_dim(self, axis):
        return {'pipe': self.num_pp, 'data': self.num_dp}.get(axis, 0)

    def get_rank(self, pipe, data):
        assert 0 <= pipe < self.num_pp and 0 <= data < self.num_dp
        return pipe * self.num_dp + data

    def get_coord(self, rank):
        assert 0 <= rank < self.world_size()
        return ProcessCoord(pipe=rank // self.num_dp, data=rank % self.num_dp)

    def get_axis_list(self, axis, idx):
        """All global ranks whose coordinate on `axis` equals idx."""
        if axis == 'pipe':
            return [self.get_rank(idx, d) for d in range(self.num_dp)]
        if axis == 'data':
            return [self.get_rank(p, idx) for p in range(self.num_pp)]
        return []

    def get_axis_comm_lists(self, axis):
        """Groups of ranks that differ only along `axis` (the communicator lists for that axis)."""
        if axis == 'pipe':
            return [[self.get_rank(p, d) for p in range(self.num_pp)] for d in range(self.num_dp)]
        if axis == 'data':
            return [[self.get_rank(p, d) for d in range(self.num_dp)] for p in range(self.num_pp)]
        return []


class PipelineParallelGrid:
    """Rank bookkeeping + process groups.  Also serves as the `mpu` object handed to optimizers / clipping."""

    def __init__(self, topology=None, global_rank=None, world_size=None):
        initialized = dist.is_available() and dist.is_initialized()
        self.global_rank = global_rank if global_rank is not None else (dist.get_rank() if initialized else 0)
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if initialized else 1)
        self._topo = topology or PipeDataParallelTopology(1, self.world_size)
        assert self._topo.world_size() == self.world_size, \
            f'topology covers {self._topo.world_size()} ranks but the world has {self.world_size}'
        self.pipe_parallel_size = self._topo.get_dim('pipe')
        self.data_parallel_size = self._topo.get_dim('data')
        self.model_parallel_size = 1
        self.slice_parallel_size = 1
        coord = self._topo.get_coord(self.global_rank)
        self.stage_id = coord.pipe
        self.data_parallel_id = coord.data

        # Every rank must create every group, in the same order (torch.distributed contract).
        self.dp_group = None
        self.dp_groups = self._topo.get_axis_comm_lists('data')
        self.pp_group_ranks = None
        self.pp_proc_group = None
        self.pipe_groups = self._topo.get_axis_comm_lists('pipe')
        self.dp_proc_group = None
        for ranks in self.dp_groups:
            grp = dist.new_group(ranks=ranks) if initialized and self.world_size > 1 else None
            if self.global_rank in ranks:
                self.dp_group, self.dp_proc_group = ranks, grp
        for ranks in self.pipe_groups:
            grp = dist.new_group(ranks=ranks) if initialized and self.world_size > 1 else None
            if self.global_rank in ranks:
                self.pp_group_ranks, self.pp_proc_group = ranks, grp
        # reference code reads `grid.pp_group` as the list of global ranks of this pipeline (train.py:822)
        self.pp_group = self.pp_group_ranks
        # "model parallel" group in DeepSpeed's grid = all stages of one pipeline (no tensor slicing here)
        self.ds_model_proc_group = self.pp_proc_group
        self.ds_model_rank = self.stage_id
        self.ds_model_world_size = self.pipe_parallel_size

    # --- stage helpers -------------------------------------------------------------------------------------
    def get_stage_id(self):
        return self.stage_id

    def get_data_parallel_id(self):
        return self.data_parallel_id

    def is_first_stage(self):
        return self.stage_id == 0

    def is_last_stage(self):
        return self.stage_id == self.pipe_parallel_size - 1

    def stage_to_global(self, stage_id, **kwargs):
        return self._topo.get_rank(stage_id, self.data_parallel_id)

    def topology(self):
        return self._topo

    # --- mpu interface -----------------------------------------------------------------------------------
    def get_global_rank(self):
        return self.global_rank

    def get_pipe_parallel_rank(self):
        return self.stage_id

    def get_pipe_parallel_world_size(self):
        return self.pipe_parallel_size

    def get_pipe_parallel_group(self):
        return self.pp_proc_group

    def get_data_parallel_rank(self):
        return self.data_parallel_id

    def get_data_parallel_world_size(self):
        return self.data_parallel_size

    def get_data_parallel_group(self):
        return self.dp_proc_group

    def get_model_parallel_rank(self):
        return self.ds_model_rank

    def get_model_parallel_world_size(self):
        return self.ds_model_world_size

    def get_model_parallel_group(self):
        return self.ds_model_proc_group

    def get_slice_parallel_rank(self):
        return 0

    def get_slice_parallel_world_size(self):
        return 1

    def get_slice_parallel_group(self):
        return None
