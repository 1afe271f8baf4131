"""torch.profiler view of one timed bench step (kernel table + host/GPU time), run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
sys.argv = ['bench.py', '--steps', '1', '--warmup', '1', '--no-cpu-baseline']
import bench  # noqa: E402


def main():
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    dev = torch.device('cuda:0')
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'gradient_accumulation_steps': 2, 'gradient_clipping': 1.0}, device=dev)
    engine._configure_optimizer(lambda ps: torch.optim.AdamW(ps, lr=1e-5, fused=True), [p for p in module.parameters()])
    feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=2, latent_hw=128, seed=1))
    micro = split_batch((feats, label), 2)
    for _ in range(2):
        engine.train_batch(iter(micro))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.train_batch(iter(micro))
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f'unprofiled step (GAS=2): host-side issue {host*1e3:.1f} ms, wall {wall*1e3:.1f} ms')
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        engine.train_batch(iter(micro))
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=70))


main()
