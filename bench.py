#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: training images/sec (whole node), SDXL 1024x1024, bs=1 per stage, pp = N.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one optimizer step of the hot path (`engine.train_batch`): GAS micro-batches of one 1024x1024 image each
(latents [1,4,128,128], 75-token prompts through both trained CLIP text encoders) through the SDXL UNet split over N
pipeline stages, 1F1B schedule, fused loss, gradient clip, AdamW -- full fine-tune, bf16, synthetic data, random
weights.  GAS = 6 * N so per-GPU work is constant as N grows ("weak"); at N = 1 the six micro-batches of a step run on
three concurrent hipGraph lanes (same-box sweep: GAS 4 / 2 lanes 11.65, GAS 6 / 2 lanes 12.08, GAS 6 / 3 lanes 13.63, GAS 8 / 4
lanes 11.86 images/s).  Prints ONE JSON line on rank 0.

Extra objects: `roofline` (dominant kernel = the hand-written MFMA GEMM, timed with HIP events around every launch of
the last timed step) and `cpu_baseline` (the oracle's fp32 eager path on the host cores, bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--gas', type=int, default=0, help='micro-batches per step (default 6 * gpus)')
    ap.add_argument('--config', default='full', choices=['full', 'tiny'])
    ap.add_argument('--latent', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--activation-checkpointing', action='store_true')
    ap.add_argument('--partition', default='parameters')
    ap.add_argument('--no-graph', action='store_true', help='disable hipGraph capture of the micro-batch fwd+bwd')
    ap.add_argument('--lanes', type=int, default=3, help='concurrent micro-batch lanes of the single-stage hipGraph path')
    ap.add_argument('--test-single-device', action='store_true',
                    help='TEST ONLY: all ranks share cuda:0, collectives over gloo, stage payloads staged through the host')
    ap.add_argument('--torch-adamw', action='store_true', help='A/B switch: torch.optim.AdamW(fused=True) + separate lane-sum / clip / zero passes')
    ap.add_argument('--parallel-wgrad', action='store_true', help='fork wgrad onto a side stream (A/B switch; measured slower)')
    return ap.parse_args()


def gemm_roofline(trace):
    """(flops, ms, launches, per-kernel-config breakdown) of the GEMM launches of one step from their HIP-event pairs."""
    tot_flops = tot_ms = 0.0
    for (dt, ta, tb, M, N, K, batch), e0, e1 in trace:
        tot_ms += e0.elapsed_time(e1)
        tot_flops += 2.0 * M * N * K * batch
    return tot_flops, tot_ms, len(trace)


def main():
    args = parse()
    # watchdog: a run that stops making progress (a wedged collective, a kernel that never retires) dumps every thread's Python
    # stack and exits instead of holding the GPU until the caller's clock runs out
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(int(os.environ.get('DPIPE_BENCH_WATCHDOG_S', '1500')), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    if args.test_single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.test_single_device:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    if os.environ.get('DPIPE_MIOPEN_BENCHMARK', '0') == '1':
        torch.backends.cudnn.benchmark = True          # MIOpen exhaustive solver search per convolution shape (A/B switch)
    from diffusion_pipe_amd import hip, ops
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    hip.lib()

    cfg = sdxl.SDXLConfig() if args.config == 'full' else sdxl.tiny_config()
    latent = args.latent if args.config == 'full' else 32
    pp = world
    gas = args.gas or 6 * world
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=device)
    layers = work.to_layers()
    kwargs = {}
    if args.activation_checkpointing:
        from functools import partial
        kwargs = dict(activation_checkpoint_interval=1, checkpointable_layers=work.checkpointable_layers,
                      activation_checkpoint_func=partial(torch.utils.checkpoint.checkpoint, use_reentrant=False))
    module = ManualPipelineModule(layers=layers, num_stages=pp, partition_method=args.partition, loss_fn=work.get_loss_fn(),
                                  dynamic_shape=True, **kwargs)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas,
                                                         'gradient_clipping': 1.0, 'steps_per_print': 1 << 30, 'hip_graph': not args.no_graph,
                                                         'parallel_wgrad': args.parallel_wgrad, 'p2p_via_host': args.test_single_device, 'graph_lanes': args.lanes}, device=device)
    params = [p for p in module.parameters() if p.requires_grad]

    # the reference's optimizer construction (train.py:650-815): AdamW on the raw bf16 parameters, per-component groups split into
    # weight-decay / no-weight-decay halves; on the GPU the step end (lane sum + clip + update + zero) runs as the fused HIP passes
    from diffusion_pipe_amd import optim
    work.train_config = {'optimizer': {'type': 'adamw', 'lr': 1e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
    make_opt = optim.make_optimizer_factory(work.train_config, work, global_batch_size=gas, use_hip_adamw=not args.torch_adamw)
    engine._configure_optimizer(make_opt, params)

    # synthetic data: a pool of distinct pre-pulled steps (the reference pre-pulls every step's micro-batches, train.py:164-173)
    torch.manual_seed(1234)
    pool = []
    for s in range(3):
        feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=latent, seed=100 + s))
        pool.append(split_batch((feats, label), gas))
    needs_data = engine.is_first_stage() or engine.is_last_stage()

    def one_step(i):
        engine.reset_activation_shape()
        return engine.train_batch(iter(pool[i % len(pool)]) if needs_data else None)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for i in range(args.warmup):
        loss = one_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    gnorm = engine.get_global_grad_norm()
    gnorm = float(gnorm.item()) if gnorm is not None else float('nan')

    # --- roofline of the dominant kernel (the MFMA GEMM of dpipe_gemm: every Linear forward / dgrad / wgrad).  Launches inside
    # a replayed hipGraph cannot be bracketed, so ONE extra step runs eagerly on the same data with every GEMM launch
    # bracketed by two HIP events recorded on the launch stream (ops.GEMM_TRACE); achieved = sum(2 M N K) / sum(elapsed).
    # Each kernel then runs alone on the chip with operands as cold as in training (all other kernels of the step run in
    # between), which is the per-kernel figure; the graph path additionally overlaps two micro-batches (config.lanes).
    ops.GEMM_TRACE = []
    was_graph, was_stage = engine.use_graph, engine.use_stage_graphs
    engine.use_graph = engine.use_stage_graphs = False
    if was_graph or was_stage:
        for p_ in module.parameters():
            p_.grad = None
    one_step(0)
    engine.use_graph, engine.use_stage_graphs = was_graph, was_stage
    trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    torch.cuda.synchronize()
    g_flops, g_ms, launches = gemm_roofline(trace)
    rl = torch.tensor([g_flops, g_ms, launches], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(rl)
    g_flops, g_ms, launches = rl.tolist()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        images = gas * 1 * engine.dp_world_size
        value = images / (elapsed / args.steps)
        step_flops = sdxl.train_step_flops(cfg, latent_hw=latent, batch=1) * gas
        achieved = g_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        peak = 2500.0
        out = {
            'metric': 'training images/sec (whole node), SDXL 1024² bs=1/stage, pp=1/2/4/8',
            'value': round(value, 4), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'SDXL {latent * 8}x{latent * 8} full fine-tune (UNet + both CLIP text encoders trained), micro-batch 1 per stage, '
                                   f'pp={pp}, GAS={gas}, AdamW, clip 1.0' + (' [tiny test config]' if args.config != 'full' else ''),
                       'global_batch': images, 'parallelism': f'pp{pp}', 'gradient_accumulation_steps': gas,
                       'activation_checkpointing': bool(args.activation_checkpointing), 'partition': module.parts, 'hip_graph': bool(engine.use_graph or engine.use_stage_graphs),
                       'concurrent_micro_batch_lanes': engine.graph_lanes},
            'loss': float(loss.item()), 'grad_norm': float(gnorm),
            'step_tflop_algorithmic': round(step_flops / 1e12, 2),
            'mfu_vs_bf16_mfma_peak': round(step_flops / (elapsed / args.steps) / (peak * 1e12 * world), 5),
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_pipe_kernel<bf16> (dpipe_gemm_ex: every Linear forward / dgrad / wgrad)', 'achieved': round(achieved, 2), 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 5), 'traffic': None,
                         'launches_per_step': int(launches), 'avg_launch_us': round(g_ms * 1e3 / max(launches, 1), 2),
                         'gemm_gpu_ms_per_step': round(g_ms / world, 2),
                         'note': 'per-launch HIP-event times of one eager step (kernels run alone); graph replays overlap the lanes'},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_baseline import sdxl_cpu_baseline
            cb = sdxl_cpu_baseline(cfg, latent_hw=latent, budget_s=20.0)
            per_image = sdxl.train_step_flops(cfg, latent_hw=latent, batch=1)
            cb['value'] = round(1.0 / (cb['sample_seconds'] * per_image / cb.pop('_sample_flops')), 6)
            cb['unit'] = 'images/s'
            out['cpu_baseline'] = cb
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
