#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: training images/sec (whole node), SDXL 1024x1024, bs=1 per stage, pp = N.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one optimizer step of the hot path (`engine.train_batch`): GAS micro-batches of one 1024x1024 image each
(latents [1,4,128,128], 75-token prompts through both trained CLIP text encoders) through the SDXL UNet split over N
pipeline stages, 1F1B schedule, fused loss, gradient clip, AdamW -- full fine-tune, bf16, synthetic data resident in HBM before the
timed region, random weights.  GAS = 6 * N so per-GPU work is constant as N grows ("weak"); at N = 1 the six micro-batches of a step
run on three concurrent hipGraph lanes.  The host waits for the end of step n - 1 before it enqueues step n (engine
`max_steps_in_flight = 1`).  Prints ONE JSON line on rank 0.

Extra objects: `roofline` (dominant kernel = the hand-written MFMA GEMM: the step's recorded GEMM launch list replayed as one
GEMM-only hipGraph between HIP events, `traffic` from the committed rocprofv3 --pmc passes over the same list) and `cpu_baseline`
(the oracle's fp32 eager path on the host cores, one whole image, N=1 only).  Watchdogs: wall clock (DPIPE_BENCH_WATCHDOG_S, 900 s)
and progress (DPIPE_BENCH_STALL_S, 90 s without a retired graph replay -> exit 3 with the stuck launch named).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    # pp > 1 launches forward / backward stage graphs on two streams with P2P in between and cannot drain the queue per micro-batch:
    # take the runtime's per-node dispatch path there instead of the AQL-packet-capture path that wedges under multi-stream run-ahead
    # (engine.max_steps_in_flight).  Must be set before the HIP runtime initialises.
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

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
    ap.add_argument('--steps-in-flight', type=int, default=1, help='bound on the host run-ahead in optimizer steps (0 = unbounded; see engine.max_steps_in_flight)')
    ap.add_argument('--save-gemm-trace', default='', help='write the step\'s unique GEMM descriptors (+ counts) as JSON (input of tools/gemm_replay.py)')
    ap.add_argument('--sync-each-step', action='store_true', help='bisect switch: device synchronize after every step')
    ap.add_argument('--host-inputs', action='store_true', help='hand every micro-batch over as pageable host tensors (the loader contract); '
                    'default: the synthetic pool is resident in HBM before the timed region')
    return ap.parse_args()


def start_progress_monitor(engine_mod, stall_s):
    """Watchdog thread: polls the HIP events the engine records after every graph replay / step end (engine.TRACE).  If events
    are pending and none completes for `stall_s` seconds, report the last retired and the first pending label plus
    `rocm-smi --showuse`, and exit(3) -- a wedged queue must not hold the GPU until the caller's clock runs out."""
    import subprocess
    import threading
    engine_mod.TRACE = []
    verbose = os.environ.get('DPIPE_TRACE_STEPS', '0') == '1'

    def run():
        last_done, last_change, idx = None, time.monotonic(), 0
        while True:
            time.sleep(2.0)
            tr = engine_mod.TRACE
            if tr is None:
                return
            progressed = False
            while idx < len(tr) and tr[idx][1].query():
                last_done, idx, progressed = tr[idx][0], idx + 1, True
            now = time.monotonic()
            if progressed or idx >= len(tr):
                last_change = now
            if verbose:
                print(f'[trace] t={now:.1f} retired={idx}/{len(tr)} last={last_done} pending={tr[idx][0] if idx < len(tr) else None}', file=sys.stderr, flush=True)
            if idx < len(tr) and now - last_change > stall_s:
                print(f'[watchdog] no HIP event retired for {stall_s:.0f} s: last retired {last_done}, first pending {tr[idx][0]}, '
                      f'{len(tr) - idx} pending', file=sys.stderr, flush=True)
                try:
                    print(subprocess.run(['rocm-smi', '--showuse'], capture_output=True, text=True, timeout=20).stdout, file=sys.stderr, flush=True)
                except Exception as e:       # noqa: BLE001
                    print(f'[watchdog] rocm-smi failed: {e}', file=sys.stderr, flush=True)
                os._exit(3)

    th = threading.Thread(target=run, daemon=True, name='dpipe-progress-monitor')
    th.start()
    return th


def main():
    args = parse()
    # watchdogs: (1) wall clock -- dump every thread's Python stack and exit; (2) progress (start_progress_monitor) -- exit as soon
    # as the GPU stops retiring the step's graph replays
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(int(os.environ.get('DPIPE_BENCH_WATCHDOG_S', '900')), exit=True)
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
                                                         'parallel_wgrad': args.parallel_wgrad, 'p2p_via_host': args.test_single_device, 'graph_lanes': args.lanes,
                                                         'max_steps_in_flight': args.steps_in_flight}, device=device)
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
    cpu_sample = pool[0][0]                           # one micro-batch (host tensors) for the cpu_baseline leg
    if not args.host_inputs:
        # inputs resident in HBM before the timed region starts; the engine copies them into each lane's static graph inputs (D2D)
        pool = [[tuple(tuple(t.to(device) for t in part) for part in mb) for mb in step] for step in pool]
        torch.cuda.synchronize()
    from diffusion_pipe_amd.engine import engine as engine_mod
    start_progress_monitor(engine_mod, float(os.environ.get('DPIPE_BENCH_STALL_S', '90')))

    def one_step(i):
        engine.reset_activation_shape()
        return engine.train_batch(iter(pool[i % len(pool)]) if needs_data else None)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    trace_steps = os.environ.get('DPIPE_TRACE_STEPS', '0') == '1'
    for i in range(args.warmup):
        loss = one_step(i)
        if args.sync_each_step:
            torch.cuda.synchronize()
        if trace_steps:
            print(f'[trace] warm-up step {i} enqueued at {time.monotonic():.2f}', file=sys.stderr, flush=True)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(args.warmup + i)
        if args.sync_each_step:
            torch.cuda.synchronize()
        if trace_steps:
            print(f'[trace] timed step {i} enqueued at {time.monotonic():.2f}', file=sys.stderr, flush=True)
    fence()
    engine_mod.TRACE = None
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    gnorm = engine.get_global_grad_norm()
    gnorm = float(gnorm.item()) if gnorm is not None else float('nan')

    # --- roofline of the dominant kernel (the MFMA GEMM behind dpipe_gemm_ex: every Linear forward / dgrad / wgrad; 48 % of the step's
    # kernel time, profiles/).  Launches inside the step's replayed hipGraphs cannot be bracketed one by one, so: ONE extra step runs
    # eagerly with ops.GEMM_TRACE recording the descriptor of every GEMM launch; tools/gemm_replay.py then captures exactly that launch
    # list (same shapes, leading dimensions, epilogues, split-K workspace; operands rotating through a 3 GiB arena = HBM-cold weights) into a
    # GEMM-only hipGraph and replays it between two HIP events on the replay stream.  achieved = sum(2 M N K) / graph time;
    # average launch duration = graph time / launches (includes the ~1 us in-graph dispatch gap; rocprofv3's per-kernel average of the
    # same command, committed under profiles/, is the cross-check).
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
    from tools import gemm_replay
    if args.save_gemm_trace and rank == 0:
        with open(args.save_gemm_trace, 'w') as f:
            json.dump(gemm_replay.unique_with_counts(trace), f)
    rt = gemm_replay.time_in_graph(trace, device, reps=3) if trace else {'ms': 0.0, 'launches': 0, 'flops': 0.0, 'read_bytes': 0, 'write_bytes': 0}
    rl = torch.tensor([rt['flops'], rt['ms'], rt['launches'], rt['read_bytes'] + rt['write_bytes']], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(rl)
    g_flops, g_ms, launches, g_bytes = rl.tolist()
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'r2_pmc_gemm_traffic.json')      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this launch list
    if os.path.isfile(tpath):
        with open(tpath) as f:
            traffic = json.load(f)

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
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 5),
                         'traffic': traffic['hbm_bytes_per_launch'] if traffic else None,
                         'launches_per_step': int(launches), 'avg_launch_us': round(g_ms * 1e3 / max(launches, 1), 2),
                         'algorithmic_gflop_per_launch': round(g_flops / max(launches, 1) / 1e9, 3),
                         'algorithmic_bytes_per_launch': round(g_bytes / max(launches, 1)),
                         'gemm_gpu_ms_per_step': round(g_ms / world, 2),
                         'traffic_detail': traffic,
                         'method': 'GEMM-only hipGraph of the step\'s recorded launch list, HIP events on the replay stream, HBM-cold operands'},
        }
        if world == 1 and not args.no_cpu_baseline:
            # parity of the TIMED path, in every run: the product's current weights go to the host as fp32, the GPU engine (bf16 kernels,
            # hipGraph, lanes -- exactly what was timed) evaluates ONE step whose micro-batches are all `cpu_sample`, and the oracle's fp32
            # eager path evaluates the same micro-batch on the same weights inside the cpu_baseline leg: loss and pre-clip gradient norm
            # must agree (the step's mean loss over identical micro-batches = that micro-batch's loss; GAS x (g / GAS) = g)
            state = {k: {n: v.detach().to('cpu', torch.float32) for n, v in m.state_dict().items()} for k, m in work.modules().items()}
            engine.reset_activation_shape()
            p_loss = engine.train_batch(iter([cpu_sample] * gas))
            p_norm = engine.get_global_grad_norm()
            torch.cuda.synchronize()
            p_loss, p_norm = float(p_loss.item()), float(p_norm.item())
            from oracle.cpu_baseline import sdxl_cpu_baseline
            out['cpu_baseline'] = cb = sdxl_cpu_baseline(cfg, latent_hw=latent, micro_batch=cpu_sample, state=state)
            out['parity'] = {'loss_gpu': p_loss, 'loss_cpu': cb['loss'], 'loss_rel': abs(p_loss - cb['loss']) / abs(cb['loss']),
                             'grad_norm_gpu': p_norm, 'grad_norm_cpu': cb['grad_norm'], 'grad_norm_rel': abs(p_norm - cb['grad_norm']) / cb['grad_norm'],
                             'what': 'timed path (bf16 kernels, hipGraph, lanes) vs the oracle fp32 eager path: same weights (the product state dict after the '
                                     'timed steps), same micro-batch; pre-clip global gradient norm'}
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
