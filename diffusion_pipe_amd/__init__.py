"""diffusion_pipe_amd: MI355X-native pipeline-parallel training step for diffusion-pipe model adapters.

Layout (only what the train_batch hot path needs):
  csrc/      HIP kernels for gfx950 + the C ABI (include/dpipe_hip.h) -> libdpipe_hip.so
  hip.py     ctypes binding (fails loudly when the library is missing)
  ops.py     autograd operators over the C ABI
  nn.py      nn.Module building blocks on those operators
  engine/    1F1B scheduler, P2P, DP reduce, clip, optimizer step (DeepSpeed PipelineEngine surface)
  workloads/ SDXL / Wan / Flux-shaped layer lists following the reference adapters' to_layers()
  compat/    `deepspeed` import shim so the reference's train.py / utils run unchanged on this engine
"""
__version__ = '0.1.0'
