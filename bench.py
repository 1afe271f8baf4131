#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: training images/sec (whole node), SDXL 1024x1024, bs=1 per stage, pp = N.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either that bare command -- it re-execs itself under torch.distributed.run, one rank per GPU -- or the launcher form
     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one optimizer step of the hot path (`engine.train_batch`): GAS micro-batches of one 1024x1024 image each
(latents [1,4,128,128], 75-token prompts through both trained CLIP text encoders) through the SDXL UNet split over N
pipeline stages, 1F1B schedule, fused loss, gradient clip, AdamW -- full fine-tune, bf16, synthetic data resident in HBM before the
timed region, random weights.  GAS = 8 * N so per-GPU work is constant as N grows ("weak"); at N = 1 the eight micro-batches of a step
run on four concurrent hipGraph lanes (rounds 1 - 2 and most of round 3: GAS 6 on three lanes; `--lanes 3 --gas 6` reproduces that line).  The host waits for the end of step n - 2 before it enqueues step n (engine
`max_steps_in_flight`: 2 on the probed lane path since round 4's 2 000-step soak, 1 under pp > 1).  Prints ONE JSON line on rank 0.

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

# Parity bounds asserted after the line is printed (exit code 4 when missed).  Round 6: tightened to what 16-sample statistics measured on two weight states and two
# boxes (profiles/r6b_*, r6d_*), with the REFERENCE'S OWN bf16 arithmetic evaluated beside it (`parity.reference_bf16`: the oracle model under torch.autocast with bf16 weights
# and gradients -- what models/sdxl.py:387,636,675-988 executes):
#  * the HIP kernels in their exact-fp32 mode, same weights and micro-batch, vs the oracle's fp32 eager path: loss and pre-clip gradient norm within north_star's 1e-3
#    (observed 0 / 3e-5, also on the sample where the bf16 path strays furthest);
#  * the TIMED bf16 path over 16 distinct micro-batches: loss within 1e-3 (observed <= 8e-5); gradient norm |mean| and median within 5e-4 (observed |mean| 8e-5 .. 1.4e-4,
#    median 1.9e-4), worst single sample within 4e-3 (observed 1.5e-3 / 1.8e-3 with the fp32 per-channel addends of round 6, 3.1e-3 without them on the same weights; the
#    yardstick's own worst on those samples: 3.0e-3 with matmul convolutions, 5.7e-3 through MIOpen, where its MEAN sits at -3.7e-3).  One bf16 evaluation of this
#    random-initialised network cannot be held to 1e-3 on every sample -- the gradient norm follows the mean of the residual almost one to one and
#    the reference's own bf16 step is no closer; the statistic is.
PARITY_BOUND = 1e-3
PARITY_BOUND_BF16_NORM = 4e-3           # worst single sample of the parity leg (round 5: 1e-2)
PARITY_BOUND_BF16_NORM_MEAN = 5e-4      # |mean| and median of the signed gradient-norm error over the parity samples (round 5: 1e-3)


def _argv_int(flag, default):
    for i, a in enumerate(sys.argv):
        if a == flag and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
        if a.startswith(flag + '='):
            return int(a.split('=', 1)[1])
    return default


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-exec this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>` -- one rank per GPU; every rank
    re-enters this file with RANK / LOCAL_RANK / WORLD_SIZE set, so what follows (the graph-dispatch choice below included, which must precede HIP
    initialisation) runs exactly as under an external launcher.  Done before `import torch`: nothing of this process survives the exec."""
    n = _argv_int('--gpus', 1)
    if n <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver (RCCL / tensor sharing across ranks)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


if __name__ == '__main__':
    _self_launch()

if int(os.environ.get('WORLD_SIZE', '1')) > 1 and _argv_int('--pp', 0) != 1 and os.environ.get('DPIPE_PP_PACKET_CAPTURE', '0') != '1':
    # pp > 1 launches forward / backward stage graphs on two streams with P2P in between: take the runtime's per-node dispatch path there instead of the
    # AQL-packet-capture path that wedged under multi-stream run-ahead in round 2.  DPIPE_PP_PACKET_CAPTURE=1 keeps the packet-capture
    # path (the engine drains the queue once per optimizer step at every pp: engine.max_steps_in_flight).  Must be set before the HIP runtime initialises.
    # Pure data parallelism (--pp 1) runs the single-stage lane path of the N = 1 line and keeps packet capture.
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--gas', type=int, default=0, help='micro-batches per step per replica (default 8 * pp)')
    ap.add_argument('--pp', type=int, default=0, help='pipeline stages (default: = gpus, the BASELINE metric\'s pp = N line); gpus / pp data-parallel replicas of the pipeline')
    ap.add_argument('--p2p', default='auto', choices=['auto', 'rccl', 'torch'], help='stage-to-stage link (engine p2p_backend)')
    ap.add_argument('--config', default='full', choices=['full', 'tiny'])
    ap.add_argument('--workload', default='sdxl', choices=['sdxl', 'flux', 'wan', 'hv'],
                    help='sdxl = BASELINE config 2 (the metric\'s configuration, the default); flux / wan / hv = BASELINE configs 3 / 4 / 5 as real steps on '
                         'this GPU (pp = 1): Flux.1-dev 1024x1024 LoRA, Wan2.1-14B t2v 512x512x33f LoRA, HunyuanVideo 720p x 65f full fine-tune with '
                         'checkpointing + host-offloaded activations')
    ap.add_argument('--lora-rank', type=int, default=32)
    ap.add_argument('--full-ft', action='store_true', help='flux / wan: train every weight instead of LoRA adapters (activation checkpointing on)')
    ap.add_argument('--latent', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the cpu_baseline / parity leg (and the other_configs leg)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the bounded sdxl-stacked / flux / wan / hv steps the default run appends as `other_configs`')
    ap.add_argument('--no-synced-loop', action='store_true', help='skip the second timed region (a host read of the loss after every step: `value_synced_loop`)')
    ap.add_argument('--parity-workers', type=int, default=int(os.environ.get('DPIPE_BENCH_PARITY_WORKERS', '0')),
                    help='child processes evaluating the host oracle\'s extra parity samples (0: one background thread -- the default: on the GPU box five workers took 100 s per sample each against 25 s alone, '
                         'its container schedules ~32 cores whatever os.cpu_count() says; -1: 5 / 3 when the host has >= 192 / 96 hardware threads and >= 320 / 192 GB available)')
    ap.add_argument('--fp32-leg', action='store_true', help='run the exact-fp32 kernel leg of `parity` even under --light (the stacked child run of the default line)')
    ap.add_argument('--no-reference-bf16', action='store_true', help='skip `parity.reference_bf16` (the oracle model under bf16 autocast on the GPU: what the reference itself evaluates)')
    ap.add_argument('--parity-cpu-samples', type=int, default=int(os.environ.get('DPIPE_BENCH_PARITY_CPU_SAMPLES', '8')),
                    help='how many of the parity samples the HOST oracle evaluates (25 s each on the GPU box, not parallelisable there); the others are compared with the same '
                         'oracle code evaluated in fp32 on the GPU, whose agreement with the host evaluation on the common samples is reported')
    ap.add_argument('--parity-samples', type=int, default=int(os.environ.get('DPIPE_BENCH_PARITY_SAMPLES', '16')),
                    help='distinct micro-batches of the parity leg (timed path vs the oracle on the final weights; each costs ~20 s of host time, bounded by --parity-budget)')
    ap.add_argument('--parity-budget', type=float, default=float(os.environ.get('DPIPE_BENCH_PARITY_BUDGET_S', '240')), help='host seconds the oracle may spend on parity samples beyond the first')
    ap.add_argument('--light', action='store_true', help='the timed steps + the parity object only: no roofline replay legs, no fp32-kernel leg, no other_configs (the `other_configs.sdxl_stacked` child run)')
    ap.add_argument('--activation-checkpointing', action='store_true')
    ap.add_argument('--partition', default='parameters')
    ap.add_argument('--no-graph', action='store_true', help='disable hipGraph capture of the micro-batch fwd+bwd')
    ap.add_argument('--lanes', type=int, default=0, help='concurrent micro-batch lanes of the single-stage hipGraph path (default: 4 for sdxl, 2 for flux / wan, 1 for hv)')
    ap.add_argument('--test-single-device', action='store_true',
                    help='TEST ONLY: all ranks share cuda:0, collectives over gloo, stage payloads staged through the host')
    ap.add_argument('--pipe-lanes', type=int, default=int(os.environ.get('DPIPE_PIPE_LANES', '0')),
                    help='pp > 1: interleaved 1F1B instruction streams per stage (engine `pipe_lanes`; 1 = the reference\'s single stream).  Default 2 under pp > 1: two stages '
                         'sharing one MI355X 14.79 vs 11.70 images/s (profiles/r5o_bench_pp2_single_device_pipe_lanes.jsonl); the engine\'s own default stays 1')
    ap.add_argument('--stack', type=int, default=int(os.environ.get('DPIPE_STACK', '1')),
                    help='run K consecutive micro-batches of a step as ONE pass of K x the size (engine `stack_micro_batches`; same samples, same gradient; default 1)')
    ap.add_argument('--torch-adamw', action='store_true', help='A/B switch: torch.optim.AdamW(fused=True) + separate lane-sum / clip / zero passes')
    ap.add_argument('--steps-in-flight', type=int, default=-1, help='bound on the host run-ahead in optimizer steps (0 = unbounded; default: the engine\'s own choice, '
                    '2 on the probed single-stage lane path, else 1 -- see engine.max_steps_in_flight)')
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
    engine_mod.TRACE_TIMING = bool(os.environ.get('DPIPE_STEP_TIMELINE', ''))
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


def build_dit_workload(args, device):
    """-> (workload, label, make_batch(bs, seed), module kwargs, full_ft, default gas, use hipGraph)"""
    tiny = args.config == 'tiny'
    bf16 = torch.bfloat16
    adapter = {'type': 'lora', 'rank': args.lora_rank, 'alpha': args.lora_rank, 'dtype': bf16}
    kwargs, full_ft = {}, args.full_ft
    if args.workload == 'flux':
        from diffusion_pipe_amd.workloads import flux as W
        cfg = W.tiny_flux_config() if tiny else W.FluxConfig()
        work = W.FluxWorkload(cfg, model_config={'guidance': 1.0}, dtype=bf16, seed=0, device=device)
        hw, tt = ((16, 16), 24) if tiny else ((128, 128), 512)
        make = lambda bs, seed: W.synthetic_flux_batch(cfg, batch_size=bs, latent_hw=hw, text_tokens=tt, seed=seed)
        label = f'Flux.1-dev {hw[0] * 8}x{hw[1] * 8} ({cfg.num_layers} double + {cfg.num_single_layers} single MMDiT blocks, dim {cfg.dim}, {hw[0] * hw[1] // 4} image + {tt} text tokens)'
        gas, graph = 2, True
    elif args.workload == 'wan':
        from diffusion_pipe_amd.workloads import wan as W
        cfg = W.tiny_wan_config() if tiny else W.WanConfig()
        work = W.WanWorkload(cfg, dtype=bf16, seed=0, device=device)
        fr, hw, tt = (2, (12, 16), 20) if tiny else (9, (64, 64), 512)
        make = lambda bs, seed: W.synthetic_wan_batch(cfg, batch_size=bs, frames=fr, latent_hw=hw, text_tokens=tt, seed=seed)
        label = (f'Wan2.1-14B t2v {hw[0] * 8}x{hw[1] * 8}x{(fr - 1) * 4 + 1}f ({cfg.num_layers} DiT blocks, dim {cfg.dim}, ffn {cfg.ffn_dim}, '
                 f'{fr * hw[0] * hw[1] // 4} video + {tt} text tokens)')
        gas, graph = 2, True
    else:
        from diffusion_pipe_amd.workloads import hunyuan_video as W
        from diffusion_pipe_amd.engine.offload import offloaded_checkpoint
        cfg = W.tiny_hv_config() if tiny else W.HunyuanVideoConfig()
        work = W.HunyuanVideoWorkload(cfg, model_config={'guidance': 1.0}, dtype=bf16, seed=0, device=device)
        thw, tt = ((3, 8, 8), 12) if tiny else ((17, 90, 160), 256)
        make = lambda bs, seed: W.synthetic_hv_batch(cfg, batch_size=bs, latent_thw=thw, text_tokens=tt, valid_text=(tt,), seed=seed)
        label = (f'HunyuanVideo t2v {thw[1] * 8}x{thw[2] * 8}x{(thw[0] - 1) * 4 + 1}f ({cfg.mm_double_blocks_depth} double + {cfg.mm_single_blocks_depth} single '
                 f'MMDiT blocks, dim {cfg.hidden_size}, {thw[0] * thw[1] * thw[2] // 4} video + {tt} text tokens)')
        full_ft, gas, graph = True, 1, False
        if tiny:
            from diffusion_pipe_amd.engine import offload as _off
            _off.OFFLOAD_THRESHOLD = 1024              # the tiny test config must exercise the host-offload path too
    if full_ft:
        # full fine-tune of a 12 - 14 B transformer at video token counts: per-layer activation checkpointing; HunyuanVideo (config 5) additionally parks the
        # checkpoints in host DRAM (the reference's activation_checkpointing = 'unsloth': train.py:586-603, utils/unsloth_utils.py:24-79)
        from functools import partial
        fn = offloaded_checkpoint if args.workload == 'hv' else partial(torch.utils.checkpoint.checkpoint, use_reentrant=False)
        kwargs = dict(activation_checkpoint_interval=1, checkpointable_layers=work.checkpointable_layers, activation_checkpoint_func=fn)
    else:
        work.configure_adapter(adapter)
    return work, label, make, kwargs, full_ft, gas, graph


def run_dit_workload(args, device, world, rank):
    print(json.dumps(measure_dit_workload(args, device, world, light=args.light)), flush=True)


def measure_dit_workload(args, device, world, light=False):
    """BASELINE configs 3 / 4 / 5 as real `train_batch` steps on ONE MI355X (pp = 1; 288 GB of HBM hold what the reference spreads over 2 / 4 / 8 GPUs):
    real depth, real token counts, synthetic latents / text embeddings resident in HBM, random weights, AdamW.  Prints the same JSON line: samples/s,
    the step's algorithmic FLOPs (from the traced forward: GEMM 2 M N K + attention 4 Sq Sk D H; x 3 full fine-tune, x 2 LoRA, recompute not counted),
    `roofline` for the step's dominant kernel family (MFMA GEMM or flash attention, whichever takes longer in the replay of the step's own launch list),
    peak HBM use, and a bounded `cpu_baseline` (one oracle block).  `light` (the `other_configs` leg of the default run): the timed steps only -- no replay legs,
    no FLOP trace, no cpu_baseline -- and the dict is returned instead of printed."""
    import gc
    from diffusion_pipe_amd import hip, ops, optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.engine import engine as engine_mod, offload as offload_mod
    hip.lib()
    assert world == 1, 'the DiT workloads run pp = 1 on one GPU (bench.py --workload sdxl is the multi-GPU line)'
    t_build = time.perf_counter()
    work, label, make, kwargs, full_ft, gas_default, graph = build_dit_workload(args, device)
    graph = graph and not args.no_graph
    gas = args.gas or gas_default
    lanes = args.lanes or (2 if (graph and args.workload in ('flux', 'wan') and gas >= 2) else 1)      # measured: Flux 2.49 / 3.09 / 2.89 samples/s at 1 / 2 / 3 lanes, Wan-14B 0.85 / 0.94 at 1 / 2
    n_params = sum(p.numel() for p in work.transformer.parameters())
    n_train = sum(p.numel() for p in work.transformer.parameters() if p.requires_grad)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method=args.partition, loss_fn=work.get_loss_fn(), dynamic_shape=True, **kwargs)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                         'steps_per_print': 1 << 30, 'hip_graph': graph, 'graph_lanes': lanes,
                                                         'stack_micro_batches': args.stack if gas % max(1, args.stack) == 0 else 1,
                                                         **({'max_steps_in_flight': args.steps_in_flight} if args.steps_in_flight >= 0 else {})}, device=device)
    work.train_config = {'optimizer': {'type': 'adamw', 'lr': 1e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
    engine._configure_optimizer(optim.make_optimizer_factory(work.train_config, work, global_batch_size=gas), [p for p in module.parameters() if p.requires_grad])
    torch.manual_seed(1234)
    pool = []
    for s_ in range(2):
        feats, label_t = work.prepare_inputs(make(gas, 100 + s_))
        step = split_batch((feats, label_t), gas)
        pool.append([tuple(tuple(t.to(device) for t in part) for part in mb) for mb in step])
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    start_progress_monitor(engine_mod, float(os.environ.get('DPIPE_BENCH_STALL_S', '600')))

    def one_step(i):
        engine.reset_activation_shape()
        return engine.train_batch(iter(pool[i % len(pool)]))

    trace_in_timed = not graph            # eager launches pass through ops.gemm / ops.attention anyway: record the last timed step's launch list
    loss = None
    for i in range(args.warmup):
        loss = one_step(i)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if trace_in_timed and i == args.steps - 1:
            ops.GEMM_TRACE, ops.ATTN_TRACE = [], []
        loss = one_step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    engine_mod.TRACE = None
    free_b, total_b = torch.cuda.mem_get_info(device)
    peak_hbm = max(torch.cuda.max_memory_reserved(device), total_b - free_b)      # the hipGraph pools hold the saved activations: reserved, not "allocated"
    gnorm = engine.get_global_grad_norm()
    gnorm = float(gnorm.item()) if gnorm is not None else float('nan')
    if light:
        value = gas / (elapsed / args.steps)
        out = {'metric': f'training samples/sec, one MI355X (pp=1), BASELINE config {dict(flux=3, wan=4, hv=5)[args.workload]}', 'value': round(value, 5), 'unit': 'samples/s',
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'dtype': 'bf16', 'data': 'synthetic',
               'config': {'workload': label + (', full fine-tune' if full_ft else f', LoRA rank {args.lora_rank} on every Linear of the blocks') + f', micro-batch 1, pp=1, GAS={gas}, AdamW, clip 1.0',
                          'hip_graph': bool(graph), 'concurrent_micro_batch_lanes': lanes, 'parameters': n_params, 'trainable_parameters': n_train},
               'loss': float(loss.item()), 'grad_norm': gnorm, 'peak_hbm_gb': round(peak_hbm / 2 ** 30, 2)}
        engine_mod.TRACE = None
        del engine, module, pool, loss
        work.transformer = None
        ops.release_caches()
        gc.collect()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info(device)
        out['hbm_in_use_after_release_gb'] = round((total_b - free_b) / 2 ** 30, 2)
        return out
    if not trace_in_timed:
        # one eager step records the launch list the graphs replay; the lanes' graphs (and their static activation pools) are not needed any more
        engine._lanes = []
        import gc as _gc
        _gc.collect()
        torch.cuda.empty_cache()
        ops.GEMM_TRACE, ops.ATTN_TRACE = [], []
        was = engine.use_graph
        engine.use_graph = False
        for p_ in module.parameters():
            p_.grad = None
        one_step(0)
        engine.use_graph = was
        torch.cuda.synchronize()
    gemm_trace, attn_trace, ops.GEMM_TRACE, ops.ATTN_TRACE = ops.GEMM_TRACE, ops.ATTN_TRACE, None, None
    pinned = sum(t.numel() * t.element_size() for bufs in offload_mod._FREE.values() for t in bufs)

    # algorithmic FLOPs of one sample's forward: the layers once, no autograd, no checkpoint wrapper
    ops.GEMM_TRACE, ops.ATTN_TRACE = [], []
    with torch.no_grad():
        x = pool[0][0][0]
        x = x[0] if len(x) == 1 else x
        for layer in module.forward_funcs:
            x = layer(x)
    torch.cuda.synchronize()
    from tools import attn_replay, gemm_replay
    fwd_gemm = sum(gemm_replay.flops(d) for d in ops.GEMM_TRACE)
    fwd_attn = sum(attn_replay.flops_fwd(e) for e in ops.ATTN_TRACE)
    ops.GEMM_TRACE, ops.ATTN_TRACE = None, None
    del x
    sample_flops = (3.0 if full_ft else 2.0) * (fwd_gemm + fwd_attn)

    # free the training state before the replay legs allocate their scratch (static graph pools, optimizer state, gradients)
    value = gas / (elapsed / args.steps)
    loss_v = float(loss.item())
    del engine, module, pool
    work.transformer = None
    gc.collect()
    torch.cuda.empty_cache()
    g_rt = gemm_replay.time_in_graph(gemm_trace, device, reps=1 if sum(gemm_replay.flops(d) for d in gemm_trace) > 5e14 else 3, arena_bytes=8 << 30)
    a_rt = attn_replay.time_list(attn_trace, device, reps=1 if sample_flops > 1e15 else 3)
    peak = 2500.0
    g_tf = g_rt['flops'] / (g_rt['ms'] * 1e-3) / 1e12 if g_rt['ms'] > 0 else 0.0
    a_tf = a_rt['flops'] / (a_rt['ms'] * 1e-3) / 1e12 if a_rt['ms'] > 0 else 0.0
    dom_attn = a_rt['ms'] > g_rt['ms']
    roof = {'bound': 'mfma', 'peak': peak, 'unit': 'TFLOP/s', 'traffic': None,
            'gemm': {'kernel': 'gemm_pipe_kernel<bf16> (dpipe_gemm_ex)', 'achieved': round(g_tf, 1), 'frac': round(g_tf / peak, 4), 'launches_per_step': g_rt['launches'],
                     'gpu_ms_per_step': round(g_rt['ms'], 2), 'avg_launch_us': round(g_rt['ms'] * 1e3 / max(g_rt['launches'], 1), 1)},
            'attention': {'kernel': 'attn_fwd_dma / attn_bwd_dq_dma / attn_bwd_dkv_dma (dpipe_attn_fwd / dpipe_attn_bwd)', 'achieved': round(a_tf, 1), 'frac': round(a_tf / peak, 4),
                          'calls_per_step': a_rt['calls'], 'gpu_ms_per_step': round(a_rt['ms'], 2), 'per_shape': a_rt['per_shape'],
                          'flop_count': 'forward 4 Sq Sk D H, backward 2.5 x forward (the kernels recompute S three times: 64 MFMAs issued per 40 counted)'},
            'method': 'the step\'s recorded GEMM launch list replayed as one GEMM-only hipGraph (HBM-cold operands) and every attention shape of the step replayed '
                      'forward (+ backward) in a hipGraph, HIP events on the replay stream; dominant = the family with the larger replay time'}
    dom = roof['attention' if dom_attn else 'gemm']
    roof.update({'kernel': dom['kernel'], 'achieved': dom['achieved'], 'frac': dom['frac']})
    ms_per_step = elapsed / args.steps * 1e3
    out = {'metric': f'training samples/sec, one MI355X (pp=1), BASELINE config {dict(flux=3, wan=4, hv=5)[args.workload]}',
           'value': round(value, 5), 'unit': 'samples/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': label + (', full fine-tune' if full_ft else f', LoRA rank {args.lora_rank} on every Linear of the blocks') +
                      f', micro-batch 1, pp=1, GAS={gas}, AdamW, clip 1.0' + (' [tiny test config]' if args.config != 'full' else ''),
                      'global_batch': gas, 'parallelism': 'pp1', 'gradient_accumulation_steps': gas, 'hip_graph': bool(graph), 'concurrent_micro_batch_lanes': lanes,
                      'activation_checkpointing': ('host-offloaded (unsloth)' if args.workload == 'hv' else True) if full_ft else False,
                      'parameters': n_params, 'trainable_parameters': n_train},
           'loss': loss_v, 'grad_norm': gnorm, 'build_seconds': round(t_build, 1),
           'sample_tflop_algorithmic': round(sample_flops / 1e12, 1), 'forward_tflop': {'gemm': round(fwd_gemm / 1e12, 2), 'attention': round(fwd_attn / 1e12, 2)},
           'mfu_vs_bf16_mfma_peak': round(sample_flops * value / (peak * 1e12), 5),
           'peak_hbm_gb': round(peak_hbm / 2 ** 30, 2), 'pinned_host_gb': round(pinned / 2 ** 30, 2), 'roofline': roof}
    if not args.no_cpu_baseline:
        from oracle.cpu_baseline import dit_block_cpu_baseline
        out['cpu_baseline'] = dit_block_cpu_baseline(args.workload, sample_flops) if args.config == 'full' else None
    return out


def main():
    args = parse()
    # watchdogs: (1) wall clock -- dump every thread's Python stack and exit; (2) progress (start_progress_monitor) -- exit as soon
    # as the GPU stops retiring the step's graph replays
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(int(os.environ.get('DPIPE_BENCH_WATCHDOG_S', '1500')), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        # (a bare `python bench.py --gpus N` never gets here: _self_launch() re-execs it under torch.distributed.run before torch is imported)
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (self-launching) or torch.distributed.run --nproc-per-node N')
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

    from diffusion_pipe_amd import hip, ops
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    hip.lib()

    if args.workload != 'sdxl':
        return run_dit_workload(args, device, world, rank)
    args.lanes = args.lanes or 4           # 4 lanes with lane 0 on the caller's stream = the 4 hardware queues HIP streams map onto (engine._train_batch_graphed)
    cfg = sdxl.SDXLConfig() if args.config == 'full' else sdxl.tiny_config()
    latent = args.latent if args.config == 'full' else 32
    pp = args.pp or world
    if world % pp:
        raise SystemExit(f'--pp {pp} does not divide {world} ranks')
    dp = world // pp
    gas = args.gas or 8 * pp               # two micro-batches per lane and step
    args.pipe_lanes = args.pipe_lanes or (2 if pp > 1 else 1)
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=device)
    if os.environ.get('DPIPE_BENCH_ZERO_WEIGHTS', '0') == '1':
        # diagnostic: every weight zero -> every MFMA operand zero -> the same kernels and launches at a fraction of the switching power.  The
        # step time of this run next to the normal one says how much of the step is the chip holding its power budget (MI355X_MICROARCH.md, DVFS give-back).  Not a benchmark.
        with torch.no_grad():
            for m_ in work.modules().values():
                for p_ in m_.parameters():
                    p_.zero_()
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
                                                         'p2p_via_host': args.test_single_device, 'graph_lanes': args.lanes, 'p2p_backend': args.p2p,
                                                         'stage_fwd_streams': int(os.environ.get('DPIPE_STAGE_FWD_STREAMS', '1' if args.test_single_device else '2')),
                                                         'pipe_lanes': args.pipe_lanes, 'stack_micro_batches': args.stack,
                                                         **({'max_steps_in_flight': args.steps_in_flight} if args.steps_in_flight >= 0 else {})}, device=device)
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
    cpu_samples = [pool[j // gas][j % gas] for j in range(max(1, min(args.parity_samples, 3 * gas)))]      # ... and the parity leg's distinct micro-batches (host tensors)
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
    elapsed = time.perf_counter() - t0
    # The loop the boundary promises: the reference reads the loss on the host after EVERY train_batch (`model_engine.train_batch(iterator).item()`, train.py:918;
    # this repo's train_loop.py:54 does the same), which drains the queue and forfeits the host run-ahead the region above enjoys (engine.max_steps_in_flight = 2).
    # Timed the same way (K steps between two fences), reported beside `value` as `value_synced_loop`.
    synced_elapsed = None
    if not args.no_synced_loop:
        t1 = time.perf_counter()
        for i in range(args.steps):
            loss = one_step(args.warmup + args.steps + i)
            loss.item()
        fence()
        synced_elapsed = time.perf_counter() - t1
    timeline_path = os.environ.get('DPIPE_STEP_TIMELINE', '')
    if timeline_path and engine_mod.TRACE_TIMING and rank == 0:
        # per-step timeline of the timed steps: GPU time (ms, relative to the step's first event) at which every lane's replay started / ended, the lanes
        # joined and the step end retired, plus the host clock at which each launch was issued
        tr = [e for e in engine_mod.TRACE if len(e) == 3]
        steps_ = {}
        for label, ev, host in tr:
            steps_.setdefault(label[1] if label[0] != 'step_end' else label[1], []).append((label, ev, host))
        rows = []
        for st in sorted(steps_)[-min(6, len(steps_)):]:
            evs = steps_[st]
            ev0, h0 = evs[0][1], evs[0][2]
            rows.append({'step': st, 'events': [{'label': list(l), 'gpu_ms': round(ev0.elapsed_time(e), 3), 'host_ms': round((h - h0) * 1e3, 3)} for l, e, h in evs]})
        for r in rows:
            r['gpu_ms_total'] = r['events'][-1]['gpu_ms']
        with open(timeline_path, 'w') as f:
            json.dump(rows, f, indent=1)
    engine_mod.TRACE = None
    own_elapsed = elapsed
    t = torch.tensor([elapsed, synced_elapsed or 0.0], device=device, dtype=torch.float64)
    comm_ranks, comm_backend = 1, None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, device=device, dtype=torch.float32)       # rank count as the communicator itself sees it (one contribution per rank)
        dist.all_reduce(ones)
        comm_ranks, comm_backend = int(ones.item()), dist.get_backend()
    elapsed, synced_elapsed = t[0].item(), (t[1].item() if synced_elapsed is not None else None)
    free_b, total_b = torch.cuda.mem_get_info(device)
    peak_hbm = max(torch.cuda.max_memory_reserved(device), total_b - free_b)      # after the timed steps, before the roofline / parity legs allocate; hipGraph pools are "reserved"
    gnorm = engine.get_global_grad_norm()
    engine_norm_missing = gnorm is None          # (a stage without trainable parameters reports no norm: not a divergence)
    gnorm = float(gnorm.item()) if gnorm is not None else float('nan')

    # --- roofline of the dominant kernel (the MFMA GEMM behind dpipe_gemm_ex: every Linear forward / dgrad / wgrad; 48 % of the step's
    # kernel time, profiles/).  Launches inside the step's replayed hipGraphs cannot be bracketed one by one, so: ONE extra step runs
    # eagerly with ops.GEMM_TRACE recording the descriptor of every GEMM launch; tools/gemm_replay.py then captures exactly that launch
    # list (same shapes, leading dimensions, epilogues, split-K workspace; operands rotating through a 3 GiB arena = HBM-cold weights) into a
    # GEMM-only hipGraph and replays it between two HIP events on the replay stream.  achieved = sum(2 M N K) / graph time;
    # average launch duration = graph time / launches (includes the ~1 us in-graph dispatch gap; rocprofv3's per-kernel average of the
    # same command, committed under profiles/, is the cross-check).
    trace = []
    if not args.light:
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
    # ... and the same list the way the step runs it: one graph per micro-batch lane, all lanes replaying at once (the lanes exist because a single list leaves
    # CUs idle; this is the dominant kernel's sustained rate in the product's execution mode, reported NEXT TO the single-stream figure, never instead of it)
    rc = None
    if trace and world == 1 and engine.graph_lanes > 1:
        passes = engine.micro_batches                                                     # graph replays per step (GAS, or GAS / K with --stack K)
        per_lane = trace[:len(trace) // passes * max(1, passes // engine.graph_lanes)]    # a lane's share of the step: passes / lanes replays
        rc = gemm_replay.time_concurrent(per_lane, device, engine.graph_lanes)
    rl = torch.tensor([rt['flops'], rt['ms'], rt['launches'], rt['read_bytes'] + rt['write_bytes']], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(rl)
    g_flops, g_ms, launches, g_bytes = rl.tolist()
    traffic = None
    tpath = next((q for q in (os.path.join(ROOT, 'profiles', f) for f in ('r6_pmc_gemm_traffic.json', 'r5_pmc_gemm_traffic.json', 'r4_pmc_gemm_traffic.json', 'r3_pmc_gemm_traffic.json')) if os.path.isfile(q)), '')
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this launch list: tools/run_gpu.sh pmc; the newest committed pass wins)
    if os.path.isfile(tpath):
        with open(tpath) as f:
            traffic = json.load(f)

    parity_failed = False
    gemm_policy = (getattr(engine, 'gemm_shallow_rings', None), getattr(engine, 'gemm_big_tiles', None))
    # Round 6: a multi-GPU run must be diagnosable from its JSON line alone (nothing above one GPU has ever been timed by the builder).  Every rank contributes: its stage,
    # the layers it holds, the link it ended up with and how `auto` was negotiated, its own timed-region clock, peak HBM, and -- measured now, after the steps that
    # matter -- the GPU time of one micro-batch of its captured forward + backward graphs replayed alone (`stage_ms`): gas x stage_ms / ms_per_step is the stage's busy
    # fraction (> 1 is possible with pipe lanes: two instruction streams overlap on the chip).
    per_rank = None
    if world > 1:
        own = {'rank': rank, 'stage': engine.stage_id, 'dp_rank': engine.grid.get_data_parallel_rank(), 'layers': [int(module.parts[engine.stage_id]), int(module.parts[engine.stage_id + 1])],
               'params_m': round(sum(p_.numel() for p_ in module.parameters()) / 1e6, 1), 'own_ms_per_step': round(own_elapsed / args.steps * 1e3, 2),
               'peak_hbm_gb': round(peak_hbm / 2 ** 30, 2), **engine.link_report(), 'stream_probe': engine.stream_probe}
        if engine.is_data_parallel:
            # the data-parallel average under the backward's tail (engine/overlap.py): what the last step started early, and by how much it led the lanes' join
            own['dp_overlap'] = dict(engine.overlap_report or {}, lead_ms_first_last=engine.overlap_lead_ms())
        try:
            sm = engine.stage_replay_ms()
            own['stage_ms'] = round(sm, 3) if sm is not None else None
            own['busy_frac'] = round(gas * sm / (elapsed / args.steps * 1e3), 4) if sm else None
        except Exception as e:                                  # noqa: BLE001 -- a diagnostic never fails the run
            own['stage_ms_error'] = repr(e)[:200]
        per_rank = [None] * world
        dist.all_gather_object(per_rank, own)
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
            'value_synced_loop': round(images / (synced_elapsed / args.steps), 4) if synced_elapsed else None,
            'ms_per_step_synced_loop': round(synced_elapsed / args.steps * 1e3, 3) if synced_elapsed else None,
            'synced_loop': 'the same K steps timed again with a host read of the loss after every train_batch (the reference loop: train.py:918); `value` = K train_batch '
                           'calls between two fences, the host running up to engine.max_steps_in_flight optimizer steps ahead',
            'config': {'workload': f'SDXL {latent * 8}x{latent * 8} full fine-tune (UNet + both CLIP text encoders trained), micro-batch 1 per stage, '
                                   f'pp={pp}, GAS={gas}, AdamW, clip 1.0' + (f', {args.stack} micro-batches stacked per pass' if args.stack > 1 else '') +
                                   (' [tiny test config]' if args.config != 'full' else ''),
                       'global_batch': images, 'parallelism': f'pp{pp}' + (f' x dp{dp}' if dp > 1 else ''), 'gradient_accumulation_steps': gas,
                       'stage_link': type(engine.link).__name__ if engine.link is not None else None, 'per_rank': per_rank,
                       'rccl_ranks': comm_ranks if comm_backend == 'nccl' else 0, 'process_group': comm_backend,
                       'graph_packet_capture': os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '1') != '0',
                       'activation_checkpointing': bool(args.activation_checkpointing), 'partition': module.parts, 'hip_graph': bool(engine.use_graph or engine.use_stage_graphs),
                       'concurrent_micro_batch_lanes': engine.graph_lanes, 'pipe_lanes': engine.pipe_lanes, 'micro_batch_stacking': engine.stack_micro_batches, 'stream_probe': engine.stream_probe,
                       'max_steps_in_flight': engine.max_steps_in_flight},
            'loss': float(loss.item()), 'grad_norm': float(gnorm), 'peak_hbm_gb': round(peak_hbm / 2 ** 30, 2),
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
                         'gemm_shallow_rings': gemm_policy[0], 'gemm_big_tiles': gemm_policy[1],
                         'method': 'GEMM-only hipGraph of the step\'s recorded launch list, HIP events on the replay stream, HBM-cold operands'},
        }
        if rc:
            c_ach = rc['flops'] / (rc['ms'] * 1e-3) / 1e12
            out['roofline']['concurrent_lanes'] = {'lanes': rc['lanes'], 'achieved': round(c_ach, 2), 'frac': round(c_ach / peak, 5), 'ms': round(rc['ms'], 2),
                                                   'launches': rc['launches'],
                                                   'method': 'the same launch list split over the engine\'s micro-batch lanes: one GEMM-only hipGraph per lane, all '
                                                             'lanes replayed at once on their own streams (how the timed step runs them)'}
        if world == 1 and not args.no_cpu_baseline:
            # parity of the TIMED path, in every run: the product's current weights go to the host as fp32, the GPU engine (bf16 kernels,
            # hipGraph, lanes -- exactly what was timed) evaluates ONE step whose micro-batches are all `cpu_sample`, and the oracle's fp32
            # eager path evaluates the same micro-batch on the same weights inside the cpu_baseline leg: loss and pre-clip gradient norm
            # must agree (the step's mean loss over identical micro-batches = that micro-batch's loss; GAS x (g / GAS) = g)
            state = {k: {n: v.detach().to('cpu', torch.float32) for n, v in m.state_dict().items()} for k, m in work.modules().items()}
            # DPIPE_BENCH_PARITY_DETAIL=1: per-parameter view of the same comparison (tools/parity_report.py) -- the fused step end's update is replaced by a recorder of the
            # fp32 lane sums' checksum rows, the oracle returns the rows of its gradients; the family table goes to stderr
            detail = os.environ.get('DPIPE_BENCH_PARITY_DETAIL', '0') == '1' and hasattr(engine.optimizer, 'fused_update')
            gpu_rows = None
            if detail:
                from tools.parity_report import record_fused_rows
                gpu_rows = record_fused_rows(engine, {id(p_): f'{k}.{n}' for k, m in work.modules().items() for n, p_ in m.named_parameters()})
            # Round 5: the comparison is a STATISTIC.  One bf16 evaluation of this random-initialised network lands anywhere in a +-3e-3 band around the oracle's
            # gradient norm (the norm follows the residual's mean), so the leg evaluates `--parity-samples` DISTINCT micro-batches on the same
            # final weights (learning rate 0 for these steps: nothing moves between the samples; every step's GAS micro-batches are one sample repeated) and the
            # oracle evaluates as many of them as its host-time budget allows; reported: per-sample signed errors, their mean and sigma.
            for g_ in engine.optimizer.param_groups:
                g_['lr'] = 0.0
            gpu_l, gpu_n = [], []
            for smp in cpu_samples:
                engine.reset_activation_shape()
                l_ = engine.train_batch(iter([smp] * gas))
                n_ = engine.get_global_grad_norm()
                torch.cuda.synchronize()
                gpu_l.append(float(l_.item())); gpu_n.append(float(n_.item()))
            p_loss, p_norm = gpu_l[0], gpu_n[0]
            # Round 6: the yardstick.  The same weights and the same micro-batches through the ORACLE model with bf16 weights under torch.autocast on this
            # GPU -- what the reference's own step evaluates (models/sdxl.py:387,636,675-988) -- so the timed path's distance from the fp32 oracle can be read beside the
            # reference's own bf16 distance from it.  Checker only (ATen / MIOpen kernels), after the timed region.
            ref16 = None
            if not args.no_reference_bf16:
                try:
                    from oracle.gpu_reference_bf16 import sdxl_reference_bf16
                    t_r = time.perf_counter()
                    ref16 = sdxl_reference_bf16(cfg, state, cpu_samples, device, library_conv=os.environ.get('DPIPE_BENCH_YARDSTICK_MIOPEN', '0') == '1') + (round(time.perf_counter() - t_r, 1),)
                except Exception as e:                              # noqa: BLE001 -- reported in the line
                    ref16 = repr(e)[:300]
            groups16 = None
            if os.environ.get('DPIPE_BENCH_PARITY_GROUPS', '0') == '1':         # diagnostic: which bf16 roundings does the gradient norm react to (fp32 oracle on the GPU + rounding hooks)
                try:
                    from oracle.gpu_reference_bf16 import sdxl_rounding_groups
                    groups16 = sdxl_rounding_groups(cfg, state, cpu_samples, device)
                except Exception as e:                              # noqa: BLE001
                    groups16 = {'error': repr(e)[:300]}
            gpu32 = None
            n_cpu = max(1, min(args.parity_cpu_samples, len(cpu_samples)))
            try:                                                    # the oracle's fp32 path on the GPU: the reference value of the samples the host does not evaluate + a cross-check on those it does
                from oracle.gpu_reference_bf16 import sdxl_oracle_fp32_on_gpu
                gpu32 = sdxl_oracle_fp32_on_gpu(cfg, state, cpu_samples, device)
            except Exception as e:                                  # noqa: BLE001 -- then the statistic holds the host's samples only
                gpu32 = repr(e)[:300]
                n_cpu = len(cpu_samples)
            workers = args.parity_workers
            if workers < 0:
                try:
                    avail_gb = int([ln for ln in open('/proc/meminfo') if ln.startswith('MemAvailable')][0].split()[1]) / 2 ** 20
                except Exception:                                   # noqa: BLE001
                    avail_gb = 0
                workers = (5 if (os.cpu_count() or 1) >= 192 and avail_gb >= 320 else 3) if (os.cpu_count() or 1) >= 96 and avail_gb >= 192 else 0
            from oracle.cpu_baseline import sdxl_cpu_baseline
            # the oracle times its first sample alone (= `cpu_baseline`), then evaluates the other parity samples on a background thread (~25 s of host time each) while this
            # process goes on with the GPU legs below; `finish_parity()` joins it
            out['cpu_baseline'] = cb = sdxl_cpu_baseline(cfg, latent_hw=latent, micro_batch=cpu_sample, state=state, per_parameter=detail,
                                                         extra_micro_batches=cpu_samples[1:n_cpu], extra_budget_s=args.parity_budget, extra_async=not detail,
                                                         extra_workers=0 if detail else workers)
            if detail:
                from tools.parity_report import family_table
                family_table(gpu_rows, cb.pop('rows'), out=lambda line: print(line, file=sys.stderr, flush=True))
            # ... and the same micro-batch on the same weights through this repo's kernels in their exact-fp32 mode (fp32 MFMA GEMM, fp32 split convolution, unfused
            # attention; eager): north_star's 1e-3 bound is asserted on THIS comparison -- it isolates the kernels' arithmetic from bf16 rounding noise
            f32_leg = {}
            eval32 = None
            gc_ = __import__('gc')
            try:
                if args.light and not args.fp32_leg:
                    raise RuntimeError('skipped (--light)')
                del engine, module
                ops.release_caches()                  # the fused step end's pointer tables keep parameters / states / lane gradients alive
                gc_.collect(); torch.cuda.empty_cache()
                w32 = [sdxl.SDXLWorkload(cfg, dtype=torch.float32, seed=0, device=device)]
                for k, m in w32[0].modules().items():
                    m.load_state_dict({n: v.to(device) for n, v in state[k].items()})

                def eval32(smp):
                    for m in w32[0].modules().values():
                        for p_ in m.parameters():
                            p_.grad = None
                    x32 = tuple(t.to(device) for t in smp[0])
                    for layer in w32[0].to_layers():
                        x32 = layer(x32)
                    l32 = w32[0].get_loss_fn()(x32, tuple(t.to(device) for t in smp[1]))
                    l32.backward()
                    torch.cuda.synchronize()
                    n32 = float(sum(float(p_.grad.double().pow(2).sum()) for m in w32[0].modules().values() for p_ in m.parameters() if p_.grad is not None) ** 0.5)
                    return float(l32.item()), n32

                l32, n32 = eval32(cpu_sample)
                f32_leg = {'loss_gpu': l32, 'grad_norm_gpu': n32, 'loss_rel': abs(l32 - cb['loss']) / abs(cb['loss']),
                           'grad_norm_rel': abs(n32 - cb['grad_norm']) / cb['grad_norm'], 'bound': PARITY_BOUND,
                           'what': 'the same weights and micro-batch through the HIP kernels in exact-fp32 mode (eager) vs the oracle fp32 eager path'}
            except Exception as e:                                  # noqa: BLE001 -- reported, and counted as a parity failure below
                f32_leg = {'error': repr(e)[:300]}
                eval32 = None

            def finish_parity():
                """join the oracle's background samples, turn the two lists into the `parity` object, then the fp32-kernel evaluation of the worst bf16 sample"""
                cpu_l, cpu_n = cb.pop('extra_join')() if 'extra_join' in cb else (cb['loss_all'], cb['grad_norm_all'])
                cb.pop('loss_all', None); cb.pop('grad_norm_all', None)
                n_host = len(cpu_n)
                cross = None
                if isinstance(gpu32, tuple):
                    # samples the host oracle did not evaluate take the SAME oracle code's fp32 evaluation on the GPU as their reference; on the common samples the two evaluations
                    # of the oracle are compared with each other (`oracle_gpu_fp32_vs_host`)
                    cross = {'grad_norm_rel_max': max(abs(g - c) / c for g, c in zip(gpu32[1], cpu_n)), 'loss_rel_max': max(abs(g - c) / abs(c) for g, c in zip(gpu32[0], cpu_l)),
                             'common_samples': n_host}
                    cpu_l, cpu_n = list(cpu_l) + list(gpu32[0][n_host:]), list(cpu_n) + list(gpu32[1][n_host:])
                n_s = len(cpu_n)
                e_n = [(g - c) / c for g, c in zip(gpu_n, cpu_n)]
                e_l = [abs(g - c) / abs(c) for g, c in zip(gpu_l, cpu_l)]
                mean_n = sum(e_n) / n_s
                sig_n = (sum((e - mean_n) ** 2 for e in e_n) / max(n_s - 1, 1)) ** 0.5
                out['parity'] = {'loss_gpu': p_loss, 'loss_cpu': cb['loss'], 'loss_rel': abs(p_loss - cb['loss']) / abs(cb['loss']),
                                 'grad_norm_gpu': p_norm, 'grad_norm_cpu': cb['grad_norm'], 'grad_norm_rel': abs(p_norm - cb['grad_norm']) / cb['grad_norm'],
                                 'samples': n_s, 'grad_norm_rel_signed': [round(e, 6) for e in e_n], 'grad_norm_rel_mean': mean_n, 'grad_norm_rel_sigma': sig_n,
                                 'grad_norm_rel_median': sorted(abs(e) for e in e_n)[n_s // 2], 'grad_norm_rel_max': max(abs(e) for e in e_n), 'loss_rel_max': max(e_l),
                                 'what': 'timed path (bf16 kernels, hipGraph, lanes) vs the oracle fp32 eager path: same weights (the product state dict after the '
                                         'timed steps), `samples` distinct micro-batches (entry 0 = the fields above); pre-clip global gradient norm, signed relative error.  The first '
                                         '`oracle_evaluated_on.host_cpu_fp32` samples are referenced to the oracle evaluated on the host cores, the others to the same oracle code '
                                         'evaluated in fp32 on the GPU (`oracle_gpu_fp32_vs_host`: how far those two evaluations of the oracle are apart on the common samples)',
                                 'bounds': {'loss_rel_max': PARITY_BOUND, 'grad_norm_rel_max': PARITY_BOUND_BF16_NORM, 'grad_norm_rel_mean': PARITY_BOUND_BF16_NORM_MEAN,
                                            'grad_norm_rel_median': PARITY_BOUND_BF16_NORM_MEAN},
                                 'fp32_kernels': f32_leg,
                                 'oracle_evaluated_on': {'host_cpu_fp32': n_host, 'gpu_fp32_same_oracle_code': n_s - n_host},
                                 'oracle_gpu_fp32_vs_host': cross if cross is not None else ({'error': gpu32} if gpu32 is not None else None)}
                if isinstance(ref16, tuple):
                    r_n = [(g - c) / c for g, c in zip(ref16[1], cpu_n)]
                    r_l = [abs(g - c) / abs(c) for g, c in zip(ref16[0], cpu_l)]
                    r_mean = sum(r_n) / len(r_n)
                    out['parity']['reference_bf16'] = {
                        'grad_norm_rel_signed': [round(e, 6) for e in r_n], 'grad_norm_rel_mean': r_mean,
                        'grad_norm_rel_sigma': (sum((e - r_mean) ** 2 for e in r_n) / max(len(r_n) - 1, 1)) ** 0.5,
                        'grad_norm_rel_median': sorted(abs(e) for e in r_n)[len(r_n) // 2], 'grad_norm_rel_max': max(abs(e) for e in r_n), 'loss_rel_max': max(r_l),
                        'seconds': ref16[2], 'convolutions': 'MIOpen' if os.environ.get('DPIPE_BENCH_YARDSTICK_MIOPEN', '0') == '1' else 'unfold + matmul (bf16 products, fp32 accumulation)',
                        'what': 'the yardstick: the oracle model with bf16 weights under torch.autocast(bfloat16) on this GPU (ATen kernels; loss in fp32, gradients in bf16 = '
                                'what the reference\'s step evaluates, models/sdxl.py:387,636,675-988) vs the oracle fp32 eager path, same weights, same micro-batches'}
                elif ref16 is not None:
                    out['parity']['reference_bf16'] = {'error': ref16}
                if groups16 is not None and 'none' in groups16:
                    out['parity']['rounding_groups'] = {g: [round((v - b) / b, 6) for v, b in zip(vals, groups16['none'])] for g, vals in groups16.items() if g != 'none'}
                    out['parity']['rounding_groups']['fp32_on_gpu_vs_cpu_oracle'] = [round((v - c) / c, 7) for v, c in zip(groups16['none'], cpu_n)]
                elif groups16 is not None:
                    out['parity']['rounding_groups'] = groups16
                worst = max(range(n_s), key=lambda j: abs(e_n[j]))
                if worst != 0 and eval32 is not None:
                    # the sample on which the bf16 path strayed furthest: is that distance bf16 rounding (the fp32 kernels land on the oracle) or a kernel's arithmetic?
                    try:
                        lw, nw = eval32(cpu_samples[worst])
                        f32_leg['worst_bf16_sample'] = {'index': worst, 'bf16_grad_norm_rel': e_n[worst], 'loss_rel': abs(lw - cpu_l[worst]) / abs(cpu_l[worst]),
                                                        'grad_norm_rel': abs(nw - cpu_n[worst]) / cpu_n[worst]}
                    except Exception as e:                      # noqa: BLE001
                        f32_leg['worst_bf16_sample'] = {'error': repr(e)[:300]}
                if eval32 is not None:
                    w32.clear()
                gc_.collect(); torch.cuda.empty_cache()
            engine = module = None
        parity_done = False
        if world == 1 and args.config == 'full' and not args.no_other_configs and not args.no_cpu_baseline and not args.light:
            # BASELINE configs 3 and 4 as real steps on this GPU, bounded (2 warm-up + 4 timed steps each, the SDXL state freed first): a driver-timed number for the DiT
            # workloads rides the default line (`--workload flux|wan|hv` gives the full record with roofline legs and cpu_baseline).  Never fatal.
            import copy
            import gc
            engine = module = None
            state = None                          # 10 GB of fp32 host copies of the weights: the fp32 leg has loaded them, the oracle holds its own
            del work, pool, layers, params, make_opt
            ops.release_caches()
            gc.collect()
            torch.cuda.empty_cache()
            others = {}
            import subprocess
            # (1) BASELINE config 5 (HunyuanVideo 720p x 65 frames, full fine-tune, host-offloaded checkpoints: 10.9 PFLOP per step, ~22 s) as ONE warm-up + ONE timed step in a
            # child process of its own (128 GB of HBM and 25 GB of pinned host memory that must be gone again when it ends, whatever happens inside)
            t_wl = time.perf_counter()
            try:
                cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'hv', '--steps', '1', '--warmup', '1', '--light']
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
                others['hv'] = json.loads(line[-1]) if line else {'error': f'child rc {r.returncode}: ' + r.stderr[-300:]}
            except Exception as e:                          # noqa: BLE001
                others['hv'] = {'error': repr(e)[:300]}
            others['hv']['wall_seconds'] = round(time.perf_counter() - t_wl, 1)
            finish_parity()                                 # the oracle's background samples are (nearly) done by now; frees the fp32 model before the 232 GB Wan step
            parity_done = True
            # (2) the same SDXL step with 4 micro-batches STACKED per pass on 2 lanes (engine `stack_micro_batches`): the reference's own lever for
            # bigger GEMMs is micro_batch_size_per_gpu (train.py:396-400) -- same samples, same loss terms, every GEMM with 4 x the M.  NOT the headline (BASELINE's metric is
            # quoted at bs = 1 per stage): a second, labelled line with its own parity object, measured by a child run of this file on the freed GPU.
            t_wl = time.perf_counter()
            try:
                # (round 6: the stacked line carries the same kind of parity object as the headline -- eight samples (one through the host oracle, seven through the same oracle in
                # fp32 on the GPU) and the yardstick; the exact-fp32 kernel leg evaluates single samples and is not repeated here)
                cmd = [sys.executable, os.path.abspath(__file__), '--stack', '4', '--lanes', '2', '--steps', '10', '--warmup', '3', '--light', '--parity-samples', '8',
                       '--parity-cpu-samples', '1', '--parity-budget', '40', '--no-synced-loop']
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
                if line:
                    d = json.loads(line[-1])
                    others['sdxl_stacked'] = {k: d.get(k) for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'data', 'loss', 'grad_norm', 'peak_hbm_gb',
                                                                       'mfu_vs_bf16_mfma_peak', 'parity')}
                    others['sdxl_stacked']['config'] = d['config']
                    others['sdxl_stacked']['note'] = 'micro_batch_stacking = 4: NOT the bs = 1 headline; exit code of the child run (4 = parity miss): %d' % r.returncode
                else:
                    others['sdxl_stacked'] = {'error': f'child rc {r.returncode}: ' + r.stderr[-300:]}
            except Exception as e:                          # noqa: BLE001
                others['sdxl_stacked'] = {'error': repr(e)[:300]}
            others['sdxl_stacked']['wall_seconds'] = round(time.perf_counter() - t_wl, 1)
            for wl in ('flux', 'wan'):
                a2 = copy.copy(args)
                a2.workload, a2.steps, a2.warmup, a2.gas, a2.lanes, a2.full_ft, a2.stack = wl, 4, 2, 0, 0, False, 1
                t_wl = time.perf_counter()
                try:
                    others[wl] = measure_dit_workload(a2, device, 1, light=True)
                except Exception as e:                      # noqa: BLE001
                    others[wl] = {'error': repr(e)[:300]}
                ops.release_caches()                        # (outside the handler: the failed call's frames -- and the tensors they hold -- are gone by now)
                gc.collect()
                torch.cuda.empty_cache()
                others[wl]['wall_seconds'] = round(time.perf_counter() - t_wl, 1)
            out['other_configs'] = others
        if world == 1 and not args.no_cpu_baseline and not parity_done:
            finish_parity()
        print(json.dumps(out), flush=True)
        par = out.get('parity')
        if par and args.config == 'full':
            # north_star's bound on the timed path: loss and pre-clip gradient norm within 1e-3 (relative) of the oracle's fp32 eager path.  The line above is
            # printed either way; a run that misses the bound exits non-zero so a parity regression cannot ship behind a good throughput number.
            bad = [f'timed bf16 path {k} = {par[k]:.3e} > {b:g}' for k, b in (('loss_rel_max', PARITY_BOUND), ('grad_norm_rel_max', PARITY_BOUND_BF16_NORM)) if not (par[k] <= b)]
            if par['samples'] >= 4:
                bad += [f'timed bf16 path {k} over {par["samples"]} samples = {abs(par[k]):.3e} > {PARITY_BOUND_BF16_NORM_MEAN:g}'
                        for k in ('grad_norm_rel_mean', 'grad_norm_rel_median') if not abs(par[k]) <= PARITY_BOUND_BF16_NORM_MEAN]
            f32 = par.get('fp32_kernels') or {}
            if args.light and not args.fp32_leg:
                pass
            elif 'error' in f32 or not f32:
                bad.append(f'fp32-kernel leg did not run: {f32.get("error")}')
            else:
                bad += [f'fp32 kernels {k} = {f32[k]:.3e} > {PARITY_BOUND:g}' for k in ('loss_rel', 'grad_norm_rel') if not (f32[k] <= PARITY_BOUND)]
                w_ = f32.get('worst_bf16_sample')
                if w_ and 'error' not in w_:
                    bad += [f'fp32 kernels on the worst bf16 sample {k} = {w_[k]:.3e} > {PARITY_BOUND:g}' for k in ('loss_rel', 'grad_norm_rel') if not (w_[k] <= PARITY_BOUND)]
            if bad:
                print('[bench] PARITY FAILURE (vs the oracle fp32 eager path): ' + '; '.join(bad), file=sys.stderr, flush=True)
                parity_failed = True
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    import math
    if rank == 0 and not (math.isfinite(float(loss.item())) and (math.isfinite(gnorm) or engine_norm_missing)):
        # (round 5: a stacked A/B looked 20 % faster for a whole call -- on NaN losses.  A throughput line over a diverged computation is not a measurement.)
        print('[bench] NON-FINITE loss / gradient norm in the timed steps: the throughput above is not a measurement', file=sys.stderr, flush=True)
        sys.exit(5)
    if parity_failed:
        sys.exit(4)


if __name__ == '__main__':
    main()
