"""MI355X-native pipeline-parallel training engine (drop-in for the DeepSpeed surface diffusion-pipe uses)."""
from .engine import PipelineEngine, initialize
from .offload import offloaded_checkpoint
from .module import ManualPipelineModule, PipelineModule, partition_balanced, partition_uniform
from .schedule import InferenceSchedule, TrainSchedule
from .topology import PipeDataParallelTopology, PipelineParallelGrid

__all__ = ['PipelineEngine', 'initialize', 'PipelineModule', 'ManualPipelineModule', 'partition_uniform', 'partition_balanced',
           'TrainSchedule', 'InferenceSchedule', 'offloaded_checkpoint', 'PipeDataParallelTopology', 'PipelineParallelGrid']
