"""PipelineEngine: the hand-written replacement for DeepSpeed's PipelineEngine on the train_batch hot path.

One process per MI355X.  `train_batch` runs the 1F1B instruction stream of `schedule.TrainSchedule`: per micro-batch
forward / backward through the local layers (whose arithmetic runs in the HIP kernels of libdpipe_hip.so),
activation / gradient exchange with the neighbour stages over RCCL P2P on a side stream (`p2p.StageLink`),
data-parallel gradient all-reduce in large flat buckets, on-device gradient clipping without a host sync, optimizer
step.  The public surface is the one the reference drives: train.py:608-627,817-823,
862,894,916-918,183; utils/saver.py:59-128; utils/dataset.py:1389-1401.

Semantics restated from DeepSpeed 0.18.4 (not vendored in the reference; see DESIGN.md "unpinned"):
  - loss of each micro-batch is scaled by 1/GAS for backward; gradients accumulate in the parameter dtype;
  - returned loss = mean over micro-batches, averaged over data-parallel replicas, broadcast from the last stage;
  - step end: ReduceTiedGrads (none) -> ReduceGrads (DP average) -> clip -> optimizer.step -> zero_grad -> lr step.
"""
import os
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import schedule as sched
from .module import PipelineModule
from .p2p import HostStagedLink, RcclLink, StageLink


# Progress tracing (bench.py / tools: DPIPE_TRACE_STEPS=1): a list that receives (label, HIP event) after every graph replay and
# step end -- recorded on the launch stream, never waited on here, so tracing does not change the execution order.  A monitor
# thread polls event.query() to report which replay a wedged queue stopped at.
TRACE = None
TRACE_TIMING = False        # True (bench.py DPIPE_STEP_TIMELINE): timing events + the host clock of the launch, for a per-step timeline of the lanes


def _trace(label, stream):
    if TRACE is not None:
        if TRACE_TIMING:
            import time
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            TRACE.append((label, ev, time.perf_counter()))
        else:
            TRACE.append((label, stream.record_event()))


def _capture_mode():
    """hipGraph capture error mode: with a process group alive, RCCL's watchdog thread issues (harmless) event queries while
    this thread captures -- only calls of the capturing thread may invalidate the capture then."""
    return 'thread_local' if (TRACE is not None or (dist.is_available() and dist.is_initialized())) else 'global'


def _is_float(t):
    return torch.is_tensor(t) and t.is_floating_point()


def _as_list(x):
    return list(x) if isinstance(x, (tuple, list)) else [x]


def concurrent_streams(device, n, main=None, candidates=12, report=None):
    """-> n HIP streams whose work overlaps with `main`'s (default: the current stream) and with each other's.

    The runtime multiplexes HIP streams onto a handful of hardware queues (4 on MI355X / ROCm 7.2 whatever GPU_MAX_HW_QUEUES says): two streams that land on
    the same queue run their kernels one after the other, and which queue a new stream gets depends on how many streams the process created before.  Each
    candidate is therefore PROBED: a spin kernel (torch.cuda._sleep) on the candidate next to one on every stream already chosen must take about as long as one
    alone (median of 3 samples < 1.5 x); candidates that double the time share a queue with a chosen stream and are dropped.  Falls back to fresh streams when
    the probe is unavailable or no candidate passes (fewer queues than lanes: the caller still works, two lanes serialise).

    `report` (a dict, filled in place; the engine keeps it as `engine.stream_probe` and bench.py prints it): how many streams were wanted / passed the probe /
    were padded with unprobed fresh streams, the spin time alone and every candidate's ratio -- so a run whose lanes share a queue says so."""
    import statistics
    import time
    main = main or torch.cuda.current_stream(device)
    report = report if report is not None else {}
    report.update({'wanted': n, 'probed_ok': 0, 'unprobed_fallback': 0, 'candidates_tried': 0, 'ratios': [], 'spin_ms': None, 'error': None})
    fresh = lambda: [torch.cuda.Stream(device) for _ in range(n)]
    if n <= 0:
        return []
    try:
        def spin(streams, cycles):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for st in streams:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize(device)
            return time.perf_counter() - t0
        cycles = 200_000
        spin([main], cycles)                                    # warm-up (module load)
        t1 = statistics.median(spin([main], cycles) for _ in range(3))
        while t1 < 1.5e-3 and cycles < (1 << 30):               # ~2 ms per spin: far above launch / synchronize jitter
            cycles *= 2
            t1 = statistics.median(spin([main], cycles) for _ in range(3))
        report['spin_ms'] = round(t1 * 1e3, 3)
        chosen = []
        # normal-priority candidates first; when they run out of hardware queues (4 on ROCm 7.2) and more lanes are wanted, HIGH-priority candidates: the runtime
        # keeps a separate pool of hardware queues per stream priority, so a 5th .. 8th lane CAN get a queue of its own (DPIPE_LANE_PRIORITY_STREAMS=1).  OFF by default:
        # measured in round 6 (profiles/r6q_bench_lanes_priority_streams.jsonl, same box), 8 lanes that genuinely overlap (probe ratios 1.03 - 1.08) run the SDXL step at
        # 20.6 images/s, 6 lanes at 15.2, 8 lanes sharing the four normal queues at 22.6 -- against 23.3 with four lanes: the chip is full at four, the queue count is not the bound
        passes = [0] + ([-1] if n > 3 and os.environ.get('DPIPE_LANE_PRIORITY_STREAMS', '0') != '0' else [])
        report['priority_streams'] = 0
        for prio in passes:
            for _ in range(candidates):
                if len(chosen) == n:
                    break
                cand = torch.cuda.Stream(device, priority=prio) if prio else torch.cuda.Stream(device)
                group = [main] + chosen + [cand]
                t = statistics.median(spin(group, cycles) for _ in range(3))
                report['candidates_tried'] += 1
                report['ratios'].append(round(t / t1, 2))
                if t < 1.5 * t1:                                    # all of them overlapped; a shared queue gives >= 2 x
                    chosen.append(cand)
                    report['priority_streams'] += int(prio != 0)
                elif prio:
                    break                                           # (the priority pool is exhausted too)
        report['probed_ok'] = len(chosen)
        report['unprobed_fallback'] = n - len(chosen)
        if len(chosen) < n and int(os.environ.get('RANK', '0')) == 0:
            print(f'[dpipe] stream probe: only {len(chosen)} of {n} extra streams overlap with the caller\'s (ratios {report["ratios"]}); '
                  f'{n - len(chosen)} unprobed stream(s) may share a hardware queue and serialise', flush=True)
        return chosen if len(chosen) == n else chosen + [torch.cuda.Stream(device) for _ in range(n - len(chosen))]
    except Exception as e:                                      # noqa: BLE001 -- a probe must never stop training
        report.update({'error': repr(e), 'unprobed_fallback': n})
        return fresh()


def flatten_grads(params, arenas=None):
    """Re-home the existing `.grad` tensors of `params` into ONE flat buffer per dtype (a "gradient arena") and return {dtype: flat tensor}.

    The data-parallel reduction and the lane summation then run on a handful of large contiguous views instead of thousands of tensors, with no
    `torch.cat` staging copy and no copy back.  Gradients that share a storage -- the packed weight / bias gradients of a
    fused QKV projection (ops.pack_parameters) -- move as one block, so their back-to-back layout survives.  Strides are preserved (channels-last
    convolution weights keep channels-last gradients).  Gradients already inside `arenas` are left alone; gradients that appear later (a new tuple
    layout reaching a parameter for the first time) simply stay outside and are handled one by one."""
    arenas = dict(arenas or {})
    inside = {dt: (a.untyped_storage().data_ptr()) for dt, a in arenas.items()}
    blocks, sizes = {}, {}
    for p in params:
        g = p.grad
        if g is None or inside.get(g.dtype) == g.untyped_storage().data_ptr():
            continue
        if g.dtype in arenas:
            continue                                   # late arrival: an arena of this dtype exists already (its size is baked into captured graphs)
        st = g.untyped_storage()
        if st.data_ptr() not in blocks:
            nel = st.nbytes() // g.element_size()
            off = sizes.get(g.dtype, 0)
            blocks[st.data_ptr()] = (g.dtype, off)
            sizes[g.dtype] = off + (nel + 127) // 128 * 128
    if not blocks:
        return arenas
    dev = next(p.grad.device for p in params if p.grad is not None)
    for dt, n in sizes.items():
        arenas[dt] = torch.zeros(n, dtype=dt, device=dev)
    with torch.no_grad():
        for p in params:
            g = p.grad
            if g is None:
                continue
            hit = blocks.get(g.untyped_storage().data_ptr())
            if hit is None or hit[0] != g.dtype:
                continue
            view = torch.as_strided(arenas[g.dtype], g.shape, g.stride(), hit[1] + g.storage_offset())
            view.copy_(g)
            p.grad = view
    return arenas


_SHALLOW_SET_BY_ENGINE = False      # the process-wide GEMM ring option currently holds an engine's 'auto' choice (a later engine may replace it)
_BIG_SET_BY_ENGINE = False          # the same for the 128^2-tile threshold

class PipelineEngine:
    def __init__(self, module, config, args=None, optimizer=None, lr_scheduler=None, model_parameters=None, device=None):
        assert isinstance(module, PipelineModule), 'model must be a PipelineModule'
        self.module = module
        self._config = dict(config or {})
        self.args = args
        self.grid = module.mpu()
        self.mpu = self.grid
        self.global_rank = self.grid.global_rank
        self.world_size = self.grid.world_size
        self.num_stages = self.grid.pipe_parallel_size
        self.stage_id = self.grid.get_stage_id()
        self.prev_stage = self.stage_id - 1
        self.next_stage = self.stage_id + 1
        self.dp_world_size = self.grid.data_parallel_size
        self.is_pipe_parallel = self.num_stages > 1
        self.is_data_parallel = self.dp_world_size > 1

        self.micro_batch_size = int(self._config.get('train_micro_batch_size_per_gpu', 1))
        self.micro_batches = int(self._config.get('gradient_accumulation_steps', 1))
        # `stack_micro_batches` = k (default 1): train_batch still pulls gradient_accumulation_steps micro-batches of train_micro_batch_size_per_gpu samples from the
        # iterator, but runs k consecutive ones as ONE pass of k x the size (data.stack_micro_batches = the inverse of the reference's split_batch along dim 0).  Same
        # samples, same per-sample loss terms, same gradient and mean loss up to fp summation order (the loss is a mean over equal-sized samples; nothing in the path
        # mixes samples: GroupNorm / LayerNorm / attention are per sample) -- and every weight is read once per k samples, every GEMM has a k-times larger M.  What 288 GB
        # of HBM per GPU are for; the reference's 24 GB cards are why its configs say micro-batch 1.  From here on self.micro_batches counts PASSES per step.
        self.stack_micro_batches = max(1, int(self._config.get('stack_micro_batches', 1)))
        self._user_micro_batches = self.micro_batches
        if self.stack_micro_batches > 1:
            if self.micro_batches % self.stack_micro_batches:
                raise ValueError(f'stack_micro_batches = {self.stack_micro_batches} does not divide gradient_accumulation_steps = {self.micro_batches}')
            self.micro_batches //= self.stack_micro_batches
        self._gradient_clipping = float(self._config.get('gradient_clipping', 0.0))
        self._steps_per_print = int(self._config.get('steps_per_print', 10))
        self.train_batch_size_ = self.micro_batch_size * self._user_micro_batches * self.dp_world_size

        if device is None:
            if torch.cuda.is_available():
                local_rank = int(os.environ.get('LOCAL_RANK', 0))
                torch.cuda.set_device(local_rank)
                device = torch.device('cuda', local_rank)
            else:
                device = torch.device('cpu')
        self.device = torch.device(device)
        self.module.to(self.device)

        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.client_optimizer = optimizer
        self.communication_data_type = None
        self._support_torch_style_backward = True
        self.first_last_stage_group = None
        self.global_steps = 0
        self.global_samples = 0
        self.skipped_steps = 0
        # Reference-faithful gradient-norm scope under pipeline parallelism: DeepSpeed's clip only counts
        # parameters of pipeline rank 0 (utils/patches.py:212-215).  'global' = true global norm.
        self.clip_norm_scope = self._config.get('clip_norm_scope', 'deepspeed')
        self.clip_grad_fn = None           # optional whole-function override (the reference patches this, patches.py:429)
        self.grad_kernels = None           # provider of grads_sumsq / grads_clip_scale_; None = HIP kernels (ops.py)
        self.dp_bucket_bytes = int(self._config.get('dp_bucket_bytes', 512 << 20))
        self.dp_direct_min_bytes = int(self._config.get('dp_direct_min_bytes', 1 << 20))     # gradients outside an arena: averaged in place from this size on (below: staged buckets)
        # flat gradient arenas (one buffer per dtype per lane / stage) whenever gradients are persistent (hipGraph paths) and replicas exist to
        # reduce over; 'flat_grads': True forces them on a single replica as well (tests, lane summation in one launch per dtype)
        self.flat_grads = bool(self._config.get('flat_grads', self.dp_world_size > 1))
        self._stage_arena = {}
        self._dp_stream = None
        self._dp_avg_ok = None             # ReduceOp.AVG capability of the data-parallel backend (probed on first use)
        self._probe_report = {}            # engine.concurrent_streams fills it: how many lane / stage streams passed the hardware-queue probe
        self._capturing_buffer = 0
        self._last_grad_norm = None

        # hipGraph mode: one captured graph per micro-batch shape replaces ~10^4 per-op launches (static shapes only;
        # single-stage for now -- P2P stays outside graphs).  Gradients then live in persistent buffers.
        want_graph = bool(self._config.get('hip_graph', False)) and self.device.type == 'cuda'
        self.use_graph = want_graph and not self.is_pipe_parallel
        # pipeline-parallel form: per (pipe buffer, tuple layout) one forward graph and one backward graph of this stage's
        # layers ("slots": 1F1B keeps up to num_pipe_buffers micro-batches in flight, each needs its own saved
        # activations); P2P, the schedule and the step end stay outside the graphs.
        self.use_stage_graphs = want_graph and self.is_pipe_parallel
        self._graphs = {}
        self._stage_slots = {}
        self._slot_of_buffer = {}           # pipe buffer id -> the slot its last micro-batch ran in (target of in-place receives)
        # stage-graph mode runs the forward half of the schedule (Load / RecvActivation / Forward / SendActivation) and the
        # backward half (RecvGrad / Backward / SendGrad) on two HIP streams: in 1F1B steady state a stage's next forward and
        # its pending backward belong to different micro-batches, so their graphs overlap on the GPU (same effect as the
        # concurrent lanes of the single-stage path).  Per-slot events order a slot's forward before its backward and its
        # backward before the slot is reused.
        # `stage_fwd_streams` (default 2; not on the last stage, whose forward feeds its own backward at once and accumulates the loss): forwards of
        # consecutive micro-batches alternate between two streams (pipe buffer id parity), so in the fill phase and whenever the previous stage runs
        # ahead TWO forward graphs and one backward graph share the GPU -- the concurrency the three micro-batch lanes give the single-stage path.
        # Backwards stay on ONE stream: their wgrad kernels accumulate into the same .grad buffers.
        # Pipeline lanes (`pipe_lanes` = L, default 1): L independent 1F1B instruction streams per stage, micro-batch i of a step on lane i % L, interleaved tick by
        # tick (_train_batch_pipe_lanes).  1F1B lets a stage hold (stages - stage id) micro-batches, of which ONE forward and ONE backward can overlap -- the last
        # stage of a deep pipeline computes one micro-batch at a time, on a chip that needs four to be full (the single-stage lane path).  With L lanes every stage
        # has L micro-batches in the same phase at once, each lane on a HIP stream of its own with its own slots, gradient accumulators and loss scalar; the
        # lanes' accumulators meet in the (fused) step end exactly like the single-stage lanes'.  Same sums as one 1F1B stream up to fp summation order.
        self.pipe_lanes = max(1, int(self._config.get('pipe_lanes', 1))) if self.is_pipe_parallel else 1
        self._pipe_lane_state = []
        self._pipe_lane_streams = []
        self._cur_pipe_lane = None
        n_fwd = max(1, int(self._config.get('stage_fwd_streams', 2))) if (self.use_stage_graphs and self.stage_id != self.num_stages - 1 and self.pipe_lanes == 1) else 1
        self._fwd_streams = [torch.cuda.Stream(self.device) for _ in range(n_fwd)] if self.use_stage_graphs else []
        self._fwd_stream = self._fwd_streams[0] if self._fwd_streams else None
        # the backward half runs on the caller's stream unless `stage_bwd_own_stream` is set: with two forward streams and the link's communication stream a
        # stage then uses exactly the 4 hardware queues HIP streams are multiplexed onto (see the lane path: a 5th stream shares a queue and serialises)
        self._bwd_stream = torch.cuda.Stream(self.device) if (self.use_stage_graphs and self._config.get('stage_bwd_own_stream', False)) else None
        self._streams_forked = False
        self._g_total_loss = None
        # Concurrent micro-batch lanes (single-stage graph path): at micro-batch 1 most kernels of the step fill well under
        # half of the 256 CUs, so `graph_lanes` micro-batches replay at the same time on separate HIP streams, each lane
        # accumulating into its own gradient buffers (288 GB HBM: +5 GB per lane for SDXL); the lanes' gradients are summed
        # once before ReduceGrads / clip / optimizer.  Same math as sequential accumulation up to fp summation order.
        self.graph_lanes = max(1, int(self._config.get('graph_lanes', 1))) if self.use_graph else 1     # bench: 4 (lane 0 on the caller's stream)
        # `store_first_micro_batch` (default on; DPIPE_STORE_FIRST=0 for the A/B): every lane owns a second graph per tuple layout for the FIRST micro-batch it runs in
        # a step, whose parameter-gradient kernels STORE into the lane's accumulators instead of adding (ops.GRAD_STORE) -- the fused step end then has nothing to zero
        # (4 lanes x 5.2 GB of writes per SDXL step).  Used on the fused-step-end, single-replica path only; the two graphs of a lane share one memory pool.
        self.store_first = bool(self._config.get('store_first_micro_batch', os.environ.get('DPIPE_STORE_FIRST', '1') == '1')) and self.use_graph
        self._lanes = []
        self._lane_streams = None
        # GEMM ring-depth policy (C-ABI option DPIPE_OPT_GEMM_SHALLOW): with >= 2 graphs replaying concurrently (micro-batch lanes, or forward + backward stage
        # graphs) the 128^2 GEMM tile runs on its 2-deep 64 KiB ring, so a workgroup of another lane fits the same CU -- each launch is ~7 % slower alone, the
        # step 2 % faster (MI355X, 3 lanes: 19.34 vs 18.93 images/s).  `gemm_shallow_rings`: 'auto' (default) | 0 | 1 | 2 | 3; an explicit DPIPE_GEMM_SHALLOW /
        # dpipe_set_option wins over 'auto'.
        if self.device.type == 'cuda':
            from .. import hip as _hip
            want = self._config.get('gemm_shallow_rings', 'auto')
            concurrent = self.graph_lanes if self.use_graph else ((self.pipe_lanes if self.pipe_lanes > 1 else len(self._fwd_streams) + 1) if self.use_stage_graphs else 1)
            if want == 'auto':
                global _SHALLOW_SET_BY_ENGINE
                if _SHALLOW_SET_BY_ENGINE or _hip.lib().dpipe_get_option(_hip.OPT_GEMM_SHALLOW) < 0:      # never override the user's explicit choice
                    _hip.check(_hip.lib().dpipe_set_option(_hip.OPT_GEMM_SHALLOW, 2 if concurrent >= 2 else 0), 'set_option')
                    _SHALLOW_SET_BY_ENGINE = True
            else:
                _hip.check(_hip.lib().dpipe_set_option(_hip.OPT_GEMM_SHALLOW, int(want)), 'set_option')
            self.gemm_shallow_rings = _hip.lib().dpipe_get_option(_hip.OPT_GEMM_SHALLOW)
            # GEMM tile-size policy (C-ABI option DPIPE_OPT_GEMM_BIG_TILES): the dispatcher's 128^2-vs-64^2 threshold (128 tiles) is the isolated-launch optimum -- a launch of
            # < 128 tiles fills more CUs as 64^2 tiles.  With >= 2 graphs replaying concurrently the chip is full either way and the CU time per FLOP decides: a 64^2 tile
            # moves twice the operand bytes per FLOP through the CU's global -> LDS path.  Same box, 4 lanes: 20.85 vs 20.15 / 20.20 images/s with the threshold at 16
            # (profiles/r4f_bench_big_tiles.jsonl); `gemm_big_tiles`: 'auto' (default) | int; an explicit DPIPE_GEMM_BIG_TILES / dpipe_set_option wins over 'auto'.
            want_big = self._config.get('gemm_big_tiles', 'auto')
            if want_big == 'auto':
                global _BIG_SET_BY_ENGINE
                if _BIG_SET_BY_ENGINE or _hip.lib().dpipe_get_option(_hip.OPT_GEMM_BIG_TILES) < 0:
                    _hip.check(_hip.lib().dpipe_set_option(_hip.OPT_GEMM_BIG_TILES, 16 if concurrent >= 2 else 128), 'set_option')
                    _BIG_SET_BY_ENGINE = True
            else:
                _hip.check(_hip.lib().dpipe_set_option(_hip.OPT_GEMM_BIG_TILES, int(want_big)), 'set_option')
            self.gemm_big_tiles = _hip.lib().dpipe_get_option(_hip.OPT_GEMM_BIG_TILES)
        # Bounded host run-ahead.  train_batch returns a device scalar, so a tight loop could queue optimizer steps without limit: before enqueuing step n the host
        # waits for the end of step n - max_steps_in_flight (0 = unbounded).  History: in round 2 hipGraph launches of >= 2 lanes queued ACROSS a step boundary wedged a
        # lane's queue within 2 - 13 steps in 6 of 6 runs (HISTORY.md section 2a) and the default became 1 -- which the reference's loop implies anyway (.item() after
        # every train_batch, train.py:918).  Round 3 found the cause of the wedge's trigger (lanes sharing one of the 4 hardware queues with the caller's stream: a lane
        # graph of step n + 1 can sit ahead of step n's step end in the same in-order queue) and removed it (lane 0 on the caller's stream, the others probed onto queues
        # of their own): 175 steps of run-ahead finished; round 4 soaked 400 + 1 600 more (22.7 images/s against 22.0: the host's launch latency of step n + 1 hides under
        # step n's tail).  Default since: 2 on the single-stage lane path WHEN every lane stream passed the probe, 1 otherwise (pipeline stages with P2P between graphs,
        # unprobed lanes, DPIPE_LANE_STREAM_PROBE=0); an explicit config value wins.
        self._steps_in_flight_explicit = 'max_steps_in_flight' in self._config
        self.max_steps_in_flight = int(self._config.get('max_steps_in_flight', 2 if (self.use_graph and not self.is_pipe_parallel) else 1))
        self._step_done = []
        if self.device.type == 'cuda' and self._config.get('fuse_grad_accumulation', True):
            from .. import ops as _ops
            _ops.FUSE_GRAD_ACCUM = True     # wgrad / bias / norm-weight kernels add straight into existing .grad buffers
        comm_stream = None
        if self.use_stage_graphs and self.device.type == 'cuda' and os.environ.get('DPIPE_LANE_STREAM_PROBE', '1') != '0':
            # forward streams + the link's communication stream next to the caller's stream (backward half): streams probed to sit on distinct hardware
            # queues -- chosen BEFORE the link is built, so its self-test and every later transfer run on the stream it keeps
            want_comm = not self._config.get('p2p_via_host', False)
            if self.pipe_lanes > 1:        # lane 0 runs on the caller's stream, lanes 1 .. L - 1 and the link's stream on probed queues of their own (4 hardware queues: L <= 3 next to a link)
                sts = concurrent_streams(self.device, self.pipe_lanes - 1 + (1 if want_comm else 0), report=self._probe_report)
                self._pipe_lane_streams = sts[:self.pipe_lanes - 1]
            else:
                sts = concurrent_streams(self.device, len(self._fwd_streams) + (1 if want_comm else 0), report=self._probe_report)
                self._fwd_streams = sts[:len(self._fwd_streams)]
                self._fwd_stream = self._fwd_streams[0]
            if want_comm:
                comm_stream = sts[-1]
        self.link = self._make_link(comm_stream) if self.is_pipe_parallel else None
        self.loss = None
        self.total_loss = None
        self.agg_train_loss = None
        self._data_iter = None
        self._eval_mode = False
        self.pipe_buffers = {}
        self._force_grad_boundary = False
        # Data-parallel average UNDER the tail of the step's last backward (engine/overlap.py): `dp_overlap` (default on with replicas),
        # `dp_overlap_marks` layer boundaries at most, none with fewer than `dp_overlap_min_bytes` of gradients behind it.
        self.dp_overlap = bool(self._config.get('dp_overlap', True)) and self.is_data_parallel
        self._marks = None
        self._fwd_count = self._bwd_count = 0
        self._early = None                 # eager path: what the marks of the running last backward have started ({'done': ids, 'pending': [...], ...})
        self.overlap_report = {}           # last step: marks that fired, collectives / bytes started before the backward had finished
        self._overlap_events = None
        self._mark_err = None
        self._stage_lane = {'id': -1, 'arena': {}, 'grads': None}      # stage-graph path without pipeline lanes: the 'lane' the stage's progress marks belong to
        self._stage_marked_stop = None
        self.dp_mark_timeout_ms = int(self._config.get('dp_mark_timeout_ms', 20000))
        if self.dp_overlap:
            from .overlap import BackwardMarks
            self._marks = BackwardMarks(self.module, max_marks=int(self._config.get('dp_overlap_marks', 6)),
                                        min_bytes=int(self._config.get('dp_overlap_min_bytes', 1 << 20)))
            if not self._marks.boundaries:
                self._marks, self.dp_overlap = None, False
        if self.is_data_parallel:
            self._broadcast_model()

    def _make_link(self, comm_stream=None):
        """Stage-to-stage endpoint.  GPU default ('auto'): the C-ABI RCCL link (csrc/comm.hip dpipe_send / dpipe_recv: one grouped operation per tuple,
        receives straight into the slot buffers) when EVERY rank of the world brings it up and passes its self-test, else torch.distributed's isend /
        irecv on the same RCCL backend.  The decision is `RcclLink.negotiate`: phased (local probe, connect, self-test), each phase closed by a world
        all-reduce(MIN) that every rank reaches -- so the two ends of a boundary can never pick different links and an asymmetric failure cannot strand
        a neighbour inside a collective.  `comm_stream`: the link's communication stream, chosen (probed) by the caller before the link uses it."""
        if self._config.get('p2p_via_host', False) and self.device.type == 'cuda':
            return HostStagedLink(self.grid, self.device)
        backend = self._config.get('p2p_backend', 'auto' if self.device.type == 'cuda' else 'torch')
        if self.device.type != 'cuda' or backend == 'torch':
            return StageLink(self.grid, self.device, comm_stream=comm_stream)
        if backend == 'rccl':
            return RcclLink(self.grid, self.device, comm_stream=comm_stream)
        link = RcclLink.negotiate(self.grid, self.device, comm_stream=comm_stream)
        return link if link is not None else StageLink(self.grid, self.device, comm_stream=comm_stream)

    # ------------------------------------------------------------------------------------------------ config
    def train_micro_batch_size_per_gpu(self):
        return self.micro_batch_size

    def gradient_accumulation_steps(self):
        return self._user_micro_batches

    def train_batch_size(self):
        return self.train_batch_size_

    def gradient_clipping(self):
        return self._gradient_clipping

    def steps_per_print(self):
        return self._steps_per_print

    def is_first_stage(self):
        return self.stage_id == 0

    def is_last_stage(self):
        return self.stage_id == self.num_stages - 1

    def is_gradient_accumulation_boundary(self):
        return True

    def reset_activation_shape(self):
        """Forget cached stage-boundary tuple layouts; the reference calls this before every batch (train.py:916)."""
        if self.link is not None:
            self.link.reset()

    def train(self, mode=True):
        self.module.train(mode)
        return self

    def eval(self):
        self.module.train(False)
        return self

    def get_global_grad_norm(self):
        return self._last_grad_norm

    def link_report(self):
        """what this rank's stage link is and how `p2p_backend: 'auto'` was decided (RcclLink.negotiate: the phase a fallback was agreed at and this rank's own error,
        if any) -- gathered per rank into the bench line so a multi-GPU run is diagnosable from its JSON alone"""
        rep = {'link': type(self.link).__name__ if self.link is not None else None}
        neg = getattr(RcclLink, 'last_negotiation', None)
        if neg is not None and self.link is not None and not isinstance(self.link, HostStagedLink):
            rep.update(negotiated=dict(neg))
        return rep

    def stage_replay_ms(self, reps=3):
        """GPU time of ONE micro-batch of this stage's captured forward + backward graphs replayed back to back on the current stream, no exchange, no waiting on a
        neighbour (HIP events around `reps` replays of one captured slot).  micro_batches x this / step time = the stage's busy fraction: what the 1F1B bubble, the
        link and the slower neighbours leave of it.  Replays accumulate garbage into the gradient buffers: call it after the last step that matters.  None when the
        stage holds no captured slot (eager path, single-stage lane path)."""
        if self.device.type != 'cuda' or not self._stage_slots:
            return None
        slot = next(iter(self._stage_slots.values()))
        torch.cuda.synchronize(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        slot['fwd'].replay(); slot['bwd'].replay()
        e0.record()
        for _ in range(reps):
            slot['fwd'].replay(); slot['bwd'].replay()
        e1.record()
        torch.cuda.synchronize(self.device)
        return e0.elapsed_time(e1) / reps

    @property
    def stream_probe(self):
        """Outcome of the hardware-queue probe for this engine's lane / stage streams (engine.concurrent_streams `report`); {} before the probe ran."""
        return dict(self._probe_report)

    # -------------------------------------------------------------------------------------------- optimizer
    def _configure_optimizer(self, client_optimizer, model_parameters):
        """client_optimizer: an Optimizer or a factory `params -> Optimizer` (train.py:817)."""
        if callable(client_optimizer) and not isinstance(client_optimizer, torch.optim.Optimizer):
            self.optimizer = client_optimizer(model_parameters)
        else:
            self.optimizer = client_optimizer
        self.client_optimizer = self.optimizer
        return self.optimizer

    def _broadcast_model(self):
        """Replicas start from DP-rank 0's trainable parameters (utils/patches.py:163-172)."""
        src = self.grid.dp_group[0]
        for _, p in self.module.named_parameters():
            if torch.is_tensor(p) and p.requires_grad:
                dist.broadcast(p.data, src, group=self.grid.get_data_parallel_group())

    def _trainable_params(self):
        return [p for p in self.module.parameters() if p.requires_grad]

    # ---------------------------------------------------------------------------------------- batch drivers
    def _reserve_buffers(self, n):
        self.pipe_buffers = {k: [None] * n for k in ('inputs', 'labels', 'outputs', 'grads', 'slot')}

    def train_batch(self, data_iter=None):
        if not torch._C.is_grad_enabled():
            raise RuntimeError('train_batch() requires gradients enabled. Use eval_batch() instead.')
        self.module.train()
        self._eval_mode = False
        self.total_loss = None
        self._data_iter = data_iter
        self._fwd_count = self._bwd_count = 0
        self._early = None
        self.overlap_report = {}
        self._overlap_events = None
        self._stage_marked_stop = None
        if self.stack_micro_batches > 1 and data_iter is not None:
            from ..data import StackedIterator
            self._data_iter = StackedIterator(data_iter, self.stack_micro_batches)
        self._throttle()
        if self.use_graph:
            self._train_batch_graphed()
        else:
            if self.use_stage_graphs:
                if self._g_total_loss is None:
                    self._g_total_loss = torch.zeros((), device=self.device, dtype=torch.float32)
                self._g_total_loss.zero_()
            if self.pipe_lanes > 1 and self.micro_batches > 1:
                self._train_batch_pipe_lanes()
            else:
                schedule = sched.TrainSchedule(micro_batches=self.micro_batches, stages=self.num_stages, stage_id=self.stage_id)
                self._reserve_buffers(schedule.num_pipe_buffers())
                self._exec_schedule(schedule)
                if self.use_stage_graphs and self.is_last_stage():
                    self.total_loss = self._g_total_loss
        self.agg_train_loss = self._aggregate_total_loss(self.micro_batches)
        self.global_samples += self.train_batch_size_
        if self.link is not None:
            self.link.flush()
        self._data_iter = None
        if self.device.type == 'cuda' and self.max_steps_in_flight > 0:
            self._step_done.append(torch.cuda.current_stream(self.device).record_event())
        return self.agg_train_loss

    def _throttle(self):
        while self.max_steps_in_flight > 0 and len(self._step_done) >= self.max_steps_in_flight:
            self._step_done.pop(0).synchronize()

    def eval_batch(self, data_iter, return_logits=False, compute_loss=True, reduce_output='avg', bcast_loss=True,
                   num_micro_batches=None):
        self.module.eval()
        self._eval_mode = True
        self.total_loss = None
        self._data_iter = data_iter
        micro_batches = self._user_micro_batches if num_micro_batches is None else num_micro_batches     # eval never stacks: the iterator's own micro-batches
        schedule = sched.InferenceSchedule(micro_batches=micro_batches, stages=self.num_stages, stage_id=self.stage_id)
        self._reserve_buffers(schedule.num_pipe_buffers())
        with torch.no_grad():
            self._exec_schedule(schedule)
            out = self._aggregate_total_loss(micro_batches)
        if self.link is not None:
            self.link.flush()
        self._data_iter = None
        self._eval_mode = False
        return out

    _FWD_INSTR = (sched.LoadMicroBatch, sched.RecvActivation, sched.ForwardPass, sched.SendActivation)
    _BWD_INSTR = (sched.RecvGrad, sched.BackwardPass, sched.SendGrad)

    def _exec_schedule(self, pipe_schedule):
        two_streams = self.use_stage_graphs and not self._eval_mode
        if two_streams:
            main = torch.cuda.current_stream(self.device)
            for st in self._fwd_streams:
                st.wait_stream(main)                    # previous optimizer step / data preparation
            if self._bwd_stream is not None:
                self._bwd_stream.wait_stream(main)
            self._streams_forked = True
        for step_cmds in pipe_schedule:
            for cmd in step_cmds:
                handler = self._INSTRUCTION_MAP.get(type(cmd))
                if handler is None:
                    raise RuntimeError(f'{self.__class__.__name__} does not understand instruction {cmd!r}')
                if two_streams and isinstance(cmd, self._FWD_INSTR):
                    with torch.cuda.stream(self._fwd_streams[cmd.kwargs.get('buffer_id', 0) % len(self._fwd_streams)]):
                        handler(self, **cmd.kwargs)
                elif two_streams and isinstance(cmd, self._BWD_INSTR) and self._bwd_stream is not None:
                    with torch.cuda.stream(self._bwd_stream):
                        handler(self, **cmd.kwargs)
                elif two_streams and isinstance(cmd, self._BWD_INSTR):
                    handler(self, **cmd.kwargs)             # backward half on the caller's stream
                else:
                    self._join_streams()                # step end (reduce / clip / optimizer) sees both halves finished
                    handler(self, **cmd.kwargs)
        self._join_streams()

    def _join_streams(self):
        if self._streams_forked:
            main = torch.cuda.current_stream(self.device)
            for st in self._fwd_streams:
                main.wait_stream(st)
            if self._bwd_stream is not None:
                main.wait_stream(self._bwd_stream)
            self._streams_forked = False

    # ------------------------------------------------------------------------------------------- pipeline lanes
    _STEP_END_INSTR = (sched.ReduceTiedGrads, sched.ReduceGrads, sched.OptimizerStep)

    def _pipe_lanes_for(self, n):
        while len(self._pipe_lane_state) < n:
            li = len(self._pipe_lane_state)
            stream = None                                   # lane 0 (and every lane of the eager path): the caller's stream
            if self.use_stage_graphs and li > 0:
                stream = self._pipe_lane_streams[li - 1] if li - 1 < len(self._pipe_lane_streams) else torch.cuda.Stream(self.device)
            self._pipe_lane_state.append({'id': li, 'stream': stream, 'grads': {}, 'arena': {}, 'slot_of_buffer': {},
                                          'loss': torch.zeros((), device=self.device, dtype=torch.float32) if self.use_stage_graphs else None})
        return self._pipe_lane_state[:n]

    def _train_batch_pipe_lanes(self):
        """One optimizer step as L interleaved 1F1B streams (`pipe_lanes`).  Lane l runs the reference's patched TrainSchedule (utils/patches.py:113-160) over the
        micro-batches l, l + L, l + 2 L, ... of the step; tick t of every lane is issued before tick t + 1 of any.  Why this cannot deadlock and keeps the wire order:
        the schedule's step -> micro-batch arithmetic does not depend on the micro-batch count, so tick t means the same thing on every lane, and every send of the
        schedule is received by the neighbour in the SAME tick -- neighbours therefore post (tick, lane, instruction)-ordered sequences that are complementary
        message by message, which is all an in-order link (RCCL on one communication stream, gloo's FIFO per pair) needs.  The end stages pull the whole step from
        the iterator first (as train.py:164-173 pre-pulls a step), so (features, label) pair i reaches the same lane on the first and on the last stage.
        ReduceTiedGrads / ReduceGrads / OptimizerStep close every lane's stream and run ONCE, after all lanes drained."""
        import contextlib
        counts = sched.lane_micro_batches(self.micro_batches, self.pipe_lanes)
        L = len(counts)
        lanes = self._pipe_lanes_for(L)
        graphed = self.use_stage_graphs
        main = torch.cuda.current_stream(self.device) if graphed else None
        ends = self.is_first_stage() or self.is_last_stage()
        batches = [self._next_batch() for _ in range(self.micro_batches)] if ends else None
        schedules = []
        for lane, n in zip(lanes, counts):
            schedule = sched.TrainSchedule(micro_batches=n, stages=self.num_stages, stage_id=self.stage_id)
            lane['pipe_buffers'] = {k: [None] * schedule.num_pipe_buffers() for k in ('inputs', 'labels', 'outputs', 'grads', 'slot')}
            lane['data_iter'] = iter(batches[lane['id']::L]) if ends else None
            if graphed:
                lane['loss'].zero_()
                if lane['stream'] is not None:
                    lane['stream'].wait_stream(main)          # previous optimizer step / data preparation
            schedules.append(schedule)
        own_slots = self._slot_of_buffer
        try:
            for li, cmds in sched.interleave_lanes(schedules):
                lane = lanes[li]
                self.pipe_buffers, self._slot_of_buffer, self._data_iter, self._cur_pipe_lane = lane['pipe_buffers'], lane['slot_of_buffer'], lane['data_iter'], lane
                with (torch.cuda.stream(lane['stream']) if lane['stream'] is not None else contextlib.nullcontext()):
                    for cmd in cmds:
                        if isinstance(cmd, self._STEP_END_INSTR):
                            continue
                        handler = self._INSTRUCTION_MAP.get(type(cmd))
                        if handler is None:
                            raise RuntimeError(f'{self.__class__.__name__} does not understand instruction {cmd!r}')
                        handler(self, **cmd.kwargs)
        finally:
            self._slot_of_buffer, self._cur_pipe_lane, self._data_iter = own_slots, None, None
            for lane in lanes:
                lane['data_iter'] = None
        if not graphed:          # eager path: every lane ran on the caller's stream and autograd accumulated into the one set of .grad tensors
            self._exec_reduce_tied_grads()
            self._exec_reduce_grads()
            self._exec_optimizer_step()
            return
        base = lanes[0]
        flat_ok = bool(base['arena']) and all(set(l['arena']) == set(base['arena']) and all(l['arena'][d].numel() == base['arena'][d].numel() for d in base['arena'])
                                              for l in lanes[1:])
        marked_stop = None
        if self._marks is not None and flat_ok and self.is_data_parallel:
            # every instruction of the step is issued: the communication stream follows the progress marks of each lane's last backward replay
            marked_stop = self._reduce_flat_marked(lanes, [p for p in self.module.parameters() if p.requires_grad])
        for lane in lanes[1:]:
            main.wait_stream(lane['stream'])
        if marked_stop is not None and self._overlap_events is not None:
            joined = torch.cuda.Event(enable_timing=True)
            joined.record(main)
            self._overlap_events[2] = joined
        for p in self.module.parameters():
            p.grad = base['grads'].get(id(p))
        if self.is_last_stage():
            for lane in lanes[1:]:
                base['loss'].add_(lane['loss'])
            self.total_loss = base['loss']
        self._exec_reduce_tied_grads()
        if self._fused_step_end() and not self.is_data_parallel:
            self._exec_optimizer_step(lane_grads=[lane['grads'] for lane in lanes])      # lanes summed, clipped, applied and zeroed in one pass
            return
        covered = {dt: a.untyped_storage().data_ptr() for dt, a in base['arena'].items()} if flat_ok else {}
        for lane in lanes[1:]:
            ks = [k for k in lane['grads'] if k in base['grads']]
            if flat_ok:       # gradients outside the arenas one by one; the arenas themselves in _reduce_flat (bucket-wise lane sums + all-reduce, minus what ran under the backward)
                ks = [k for k in ks if covered.get(base['grads'][k].dtype) != base['grads'][k].untyped_storage().data_ptr()]
            if ks:
                torch._foreach_add_([base['grads'][k] for k in ks], [lane['grads'][k] for k in ks])
        if flat_ok:
            self._reduce_flat(base['arena'], [l['arena'] for l in lanes[1:]], stop=marked_stop)
        self._exec_reduce_grads(skip_storages=set(covered.values()) if flat_ok else set())
        self._exec_optimizer_step()                              # zeroes lane 0's buffers (p.grad)
        for lane in lanes[1:]:
            if lane['grads']:
                torch._foreach_zero_(list(lane['grads'].values()))

    # ------------------------------------------------------------------------------------------- hipGraph path
    def _train_batch_graphed(self):
        """Single-stage 1F1B degenerates to [Load, Forward, Backward] x GAS, then reduce / clip / step; each
        micro-batch's forward + loss + backward is ONE graph replay.  With graph_lanes = K > 1 micro-batch i replays on
        lane i % K (own stream, own static buffers, own gradient accumulators); lanes are joined and summed at the end."""
        from .. import ops as _ops
        K = min(self.graph_lanes, self.micro_batches)
        main = torch.cuda.current_stream(self.device)
        # (Round 5 negative result, profiles/r5a_bench_stacking_lanes_cu_masks.jsonl: every lane on a CU-masked stream confined to its own pair of XCDs -- 15.91 vs 22.76 images/s,
        #  same box: a lane's small launches can no longer spill onto the CUs the other lanes leave idle.  Removed.)
        if len(self._lanes) < K and self._lane_streams is None:
            self._lane_streams = concurrent_streams(self.device, self.graph_lanes - 1, main, report=self._probe_report) if os.environ.get('DPIPE_LANE_STREAM_PROBE', '1') != '0' else []
            probed = os.environ.get('DPIPE_LANE_STREAM_PROBE', '1') != '0' and os.environ.get('DPIPE_LANE0_MAIN', '1') != '0' and \
                (self.graph_lanes == 1 or (self._probe_report.get('unprobed_fallback', 1) == 0 and self._probe_report.get('error') is None))
            if not probed and not self._steps_in_flight_explicit:
                self.max_steps_in_flight = 1          # lanes that may share a hardware queue with the caller's stream: no run-ahead across the step boundary
        while len(self._lanes) < K:
            # Lane 0 replays on the CALLER'S stream, lanes 1 .. K - 1 on streams of their own (probed to be concurrent: concurrent_streams).  The runtime multiplexes HIP streams onto 4 hardware queues
            # (GPU_MAX_HW_QUEUES = 8 / 16 changes nothing measurable): with K lane streams NEXT TO an idle caller's stream, the 4th lane shares a queue with
            # another one and the two serialise -- 4 lanes on own streams 16.4 images/s, the same 4 lanes with lane 0 on the caller's stream 20.6 (MI355X,
            # round 3, profiles/r3t_*, r3u_*); 5 lanes fall back to 17.8.  DPIPE_LANE0_MAIN=0 restores a separate stream for lane 0 (A/B).
            li = len(self._lanes)
            if li == 0 and os.environ.get('DPIPE_LANE0_MAIN', '1') != '0':
                st = main
            else:
                st = self._lane_streams[li - 1] if 0 < li <= len(self._lane_streams) else torch.cuda.Stream(self.device)
            self._lanes.append({'id': len(self._lanes), 'stream': st, 'graphs': {}, 'grads': {}, 'arena': {},
                                'loss': torch.zeros((), device=self.device, dtype=torch.float32)})
        lanes = self._lanes[:K]
        params = list(self.module.parameters())
        # every micro-batch of the step is pulled BEFORE the first replay is enqueued (the reference's loader pre-pulls a step the same way, train.py:164-173):
        # lane 0 replays on the caller's stream, so an iterator that produces device tensors lazily on that stream (device-side prepare_inputs, H2D from
        # pinned memory) would otherwise be queued BEHIND lane 0's whole graph while lanes 1 .. K - 1 copy from its outputs right away
        batches = [self._next_batch() for _ in range(self.micro_batches)]
        for lane in lanes:
            lane['loss'].zero_()
            lane['stream'].wait_stream(main)          # ordered after the previous step end AND after whatever the iterator enqueued on the caller's stream
        store_first = self.store_first and self._fused_step_end() and not self.is_data_parallel
        started = set()
        for i in range(self.micro_batches):
            lane = lanes[i % K]
            feats, labels = batches[i]
            feats = (feats,) if torch.is_tensor(feats) else tuple(feats)
            labels = (labels,) if torch.is_tensor(labels) else tuple(labels)
            sig = tuple((tuple(t.shape), t.dtype) for t in feats + labels)
            entry = lane['graphs'].get(sig)
            if entry is None:
                torch.cuda.synchronize(self.device)            # capture with every lane idle
                entry = self._capture_micro_batch(lane, params, feats, labels)
                lane['graphs'][sig] = entry
                lane['stream'].wait_stream(main)
            with torch.cuda.stream(lane['stream']):
                for dst, src in zip(entry['inputs'] + entry['labels'], feats + labels):
                    if src.numel() > 0:
                        dst.copy_(src, non_blocking=True)
                if TRACE_TIMING:
                    _trace(('launch', self.global_steps, i, lane['id']), lane['stream'])
                first = store_first and lane['id'] not in started and entry.get('graph_first') is not None
                started.add(lane['id'])
                if self._marks is not None:
                    # this replay's number: what its progress marks will hold (a mark written by an earlier replay can never satisfy a wait for this one)
                    lane['marked_step'] = bool(entry.get('marked')) and 'gen' in lane
                    lane['fired_step'] = set(entry.get('fired', ()))
                    if lane['marked_step']:
                        lane['gen_host'] += 1
                        lane['gen'].fill_(lane['gen_host'])
                (entry['graph_first'] if first else entry['graph']).replay()
                _trace(('replay', self.global_steps, i, lane['id']), lane['stream'])
        base = lanes[0]
        flat_ok = bool(base['arena']) and all(set(l['arena']) == set(base['arena']) and all(l['arena'][d].numel() == base['arena'][d].numel() for d in base['arena'])
                                              for l in lanes[1:])
        marked_stop = None
        if self._marks is not None and flat_ok and self.is_data_parallel:
            # every replay of the step is launched: the communication stream follows the lanes' progress marks and averages the gradients of the late layers
            # over the replicas while the graphs are still in the backward of the early ones
            marked_stop = self._reduce_flat_marked(lanes, params)
        for lane in lanes:
            main.wait_stream(lane['stream'])
        if TRACE_TIMING:
            _trace(('lanes_joined', self.global_steps), main)
        if marked_stop is not None and self._overlap_events is not None:
            joined = torch.cuda.Event(enable_timing=True)
            joined.record(main)
            self._overlap_events[2] = joined
        _ops.WS_LANE = None
        if self._fused_step_end() and not self.is_data_parallel:
            # the lanes' accumulators go straight into the fused step end: summed, clipped, applied and zeroed in one pass
            for lane in lanes[1:]:
                base['loss'].add_(lane['loss'])
            for p in params:
                p.grad = base['grads'].get(id(p))
            self.total_loss = base['loss']
            self._exec_reduce_tied_grads()
            # nothing to zero when every lane's next first micro-batch stores (all of this step's lanes hold a first-micro-batch graph for every layout they captured)
            keep = store_first and all(e.get('graph_first') is not None for lane in lanes for e in lane['graphs'].values())
            self._exec_optimizer_step(lane_grads=[lane['grads'] for lane in lanes], zero_lane_grads=not keep)
            _trace(('step_end', self.global_steps - 1), main)
            return
        # lane 0 owns the step's gradients; add the other lanes' accumulators and losses into it
        covered = set()
        if flat_ok:
            covered = {dt: (a.untyped_storage().data_ptr()) for dt, a in base['arena'].items()}
        for lane in lanes[1:]:
            ks = [k for k in lane['grads'] if k in base['grads']]
            if flat_ok:       # gradients outside the arenas (late arrivals) one by one; the arenas themselves in _reduce_flat (bucket-wise, overlapped with the all-reduce)
                ks = [k for k in ks if covered.get(base['grads'][k].dtype) != base['grads'][k].untyped_storage().data_ptr()]
            if ks:
                torch._foreach_add_([base['grads'][k] for k in ks], [lane['grads'][k] for k in ks])
            base['loss'].add_(lane['loss'])
        for p in params:
            p.grad = base['grads'].get(id(p))
        self.total_loss = base['loss']
        self._exec_reduce_tied_grads()
        if flat_ok:
            self._reduce_flat(base['arena'], [l['arena'] for l in lanes[1:]], stop=marked_stop)
        self._exec_reduce_grads(skip_storages=set(covered.values()) if flat_ok else None)
        self._exec_optimizer_step()                              # zeroes lane 0's buffers (p.grad)
        for lane in lanes[1:]:
            if flat_ok:
                for a in lane['arena'].values():
                    a.zero_()
                rest = [g for g in lane['grads'].values() if covered.get(g.dtype) is None or g.untyped_storage().data_ptr() != lane['arena'][g.dtype].untyped_storage().data_ptr()]
                if rest:
                    torch._foreach_zero_(rest)
            elif lane['grads']:
                torch._foreach_zero_(list(lane['grads'].values()))
        _trace(('step_end', self.global_steps - 1), main)

    def _capture_micro_batch(self, lane, params, feats, labels):
        from .. import ops as _ops
        static_in = tuple(t.clone().detach().to(self.device) for t in feats)
        static_lab = tuple(t.clone().detach().to(self.device) for t in labels)
        single = len(static_in) == 1
        single_label = len(static_lab) == 1
        for p in params:                                  # this lane's accumulators become the parameters' .grad
            p.grad = lane['grads'].get(id(p))
        _ops.WS_LANE = lane['id'] if self.graph_lanes > 1 else None
        from . import offload as _offload
        _offload.POOL_TAG = ('lane', lane['id'])          # host-offloaded checkpoints: pinned buffers private to this lane's graph
        saved = {k: g.clone() for k, g in lane['grads'].items()}      # micro-batches this lane already accumulated
        saved_loss = lane['loss'].clone()

        ckpt = getattr(self.module, 'activation_checkpoint_interval', 0) > 0

        def leaf(t):       # as _exec_load_micro_batch: Function-style checkpoint wrappers only build a backward node when an input needs one
            t = t.detach()
            return t.requires_grad_(True) if (ckpt and t.is_floating_point()) else t

        def body():
            if self._marks is not None:
                self._marks.begin()
            x = leaf(static_in[0]) if single else tuple(leaf(t) for t in static_in)
            out = self.module(x)
            loss = self.module.loss_fn(out, static_lab[0] if single_label else static_lab)
            lane['loss'].add_(loss.detach().to(torch.float32))
            (loss / self.micro_batches).backward()

        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        want_first = self.store_first and self._fused_step_end() and not self.is_data_parallel
        spans = None
        with torch.cuda.stream(side):
            body()                                  # eager warm-up (library autotuning, allocator pools, .grad buffers)
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        if self.flat_grads:                          # the warm-up created this lane's .grad buffers: re-home them into one flat arena per dtype before
            lane['arena'] = flatten_grads(params, lane['arena'])      # their addresses are baked into the graph
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # ... the second pass runs in store mode: it finds the gradient buffers a fused kernel first-touches (ops.GRAD_STORE), by address -- hence AFTER the
            # re-homing above (spans recorded before flatten_grads named the freed buffers, every arena view counted as "not covered" and was zeroed at
            # the head of the store graph: safe, but the optimisation silently off).  Still before this lane's graph pool exists: nothing added to the memory peak
            if want_first:
                _ops.GRAD_STORE = {}
            body()
            if want_first:
                spans, _ops.GRAD_STORE = sorted(_ops.GRAD_STORE.items()), None
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        if self._marks is not None:
            # data-parallel replicas: the backward's progress marks become event-record nodes of this lane's graph (engine/overlap.py, _reduce_flat_marked)
            self._marks.sink = self._mark_sink(lane)
        fired = ()
        try:
            with torch.cuda.graph(graph, capture_error_mode=_capture_mode()):
                body()
            if self._marks is not None:
                fired = self._marks.fired()         # the boundaries whose mark IS a node of this graph (a boundary without a non-leaf float input has none)
        finally:
            if self._marks is not None:
                self._marks.sink = None
        graph_first = None
        if want_first:
            # the lane's first-micro-batch-of-a-step graph: the gradient buffers no fused kernel first-touches (autograd's own accumulation: embedding tables; found by
            # the store-mode warm-up pass above) are zeroed at the head of the graph, everything else is stored into by its first gradient kernel
            def covered(g):
                a, b = g.data_ptr(), g.data_ptr() + g.numel() * g.element_size()
                return any(lo <= a and b <= lo + n for lo, n in spans)
            unhandled = [p.grad for p in params if p.grad is not None and not covered(p.grad)]
            _ops.GRAD_STORE = {}
            graph_first = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_first, pool=graph.pool(), capture_error_mode=_capture_mode()):
                if unhandled:
                    torch._foreach_zero_(unhandled)
                body()
            _ops.GRAD_STORE = None
        # undo the warm-up's side effects: gradient buffers stay allocated (persistent, addresses baked into the
        # graph); parameters that receive no gradient keep grad = None like the eager path.
        for p in params:
            if p.grad is not None:
                k = id(p)
                if k in saved:
                    p.grad.copy_(saved[k])
                else:
                    p.grad.zero_()
                    lane['grads'][k] = p.grad
        lane['loss'].copy_(saved_loss)
        _ops.WS_LANE = None
        _offload.POOL_TAG = None
        return {'graph': graph, 'graph_first': graph_first, 'inputs': static_in, 'labels': static_lab, 'marked': self._marks is not None, 'fired': fired}

    # ------------------------------------------------------------------------- hipGraph path, pipeline stages
    def _stage_slot(self, buffer_id, inputs, labels):
        ins = tuple(_as_list(inputs))
        labs = tuple(_as_list(labels)) if labels is not None else ()
        lane_id = self._cur_pipe_lane['id'] if self._cur_pipe_lane is not None else -1
        sig = (lane_id, buffer_id, torch.is_tensor(inputs), torch.is_tensor(labels), tuple((tuple(t.shape), t.dtype) for t in ins + labs))
        slot = self._stage_slots.get(sig)
        if slot is None:
            self._capturing_buffer = buffer_id
            slot = self._capture_stage_slot(ins, labs, torch.is_tensor(inputs), torch.is_tensor(labels))
            self._stage_slots[sig] = slot
        return slot

    def _capture_stage_slot(self, ins, labs, single_in, single_lab):
        """Capture this stage's forward and backward for one pipe buffer.  The forward graph writes the stage outputs
        (and keeps the saved activations) in the slot's private pool; the backward graph reads the gradients of the
        outputs from static buffers, accumulates parameter gradients in place (fused into the wgrad / column-sum
        kernels) and leaves the input gradients in static tensors for SendGrad."""
        first, last = self.is_first_stage(), self.is_last_stage()
        torch.cuda.synchronize(self.device)               # capture with both halves of the schedule idle
        from . import offload as _offload
        _offload.POOL_TAG = ('slot', len(self._stage_slots))      # host-offloaded checkpoints: pinned buffers private to this slot's graphs
        params = [p for p in self.module.parameters() if p.requires_grad]
        lane = self._cur_pipe_lane                                # pipeline lanes: this lane's accumulators become the parameters' .grad, its scalar collects the loss
        if lane is not None:
            for p in params:
                p.grad = lane['grads'].get(id(p))
        loss_acc = lane['loss'] if lane is not None else self._g_total_loss
        saved_grads = {id(p): p.grad.clone() for p in params if p.grad is not None}     # micro-batches already accumulated
        saved_loss = loss_acc.clone()

        def make_inputs():
            xs = tuple(t.detach().clone() for t in ins)
            if not first:
                for t in xs:
                    t.requires_grad_(t.is_floating_point())
            return xs
        static_lab = tuple(t.detach().clone() for t in labs)

        def forward(xs):
            out = self.module(xs[0] if single_in else xs)
            if last:
                loss = self.module.loss_fn(out, static_lab[0] if single_lab else static_lab) if self.module.loss_fn is not None else out
                loss_acc.add_(loss.detach().to(torch.float32))
                return loss
            return out

        def backward(out, gouts):
            if last:
                (out / self.micro_batches).backward()
            else:
                torch.autograd.backward(tensors=[t for t in _as_list(out) if _is_float(t)], grad_tensors=gouts)

        def grads_like(out):
            return [torch.zeros_like(t, memory_format=torch.contiguous_format) for t in _as_list(out) if _is_float(t)]

        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):                      # eager warm-up: library autotuning, allocator pools, .grad buffers
                xs = make_inputs()
                out = forward(xs)
                backward(out, None if last else grads_like(out))
                del xs, out
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        if self.flat_grads:
            if lane is not None:
                lane['arena'] = flatten_grads(params, lane['arena'])
            else:
                self._stage_arena = flatten_grads(params, self._stage_arena)

        from .. import ops as _ops
        static_in = make_inputs()
        fwd_graph, bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # forward graphs of the two forward streams and the backward graph replay at the same time: they must not share split-K ticket counters / slabs.
        # A pipeline lane replays its forward and backward graphs on ONE stream: one workspace per lane.
        _ops.WS_LANE = ('pipe-lane', lane['id']) if lane is not None else ('stage-fwd', self._capturing_buffer % max(1, len(self._fwd_streams)))
        fired = ()
        if self._marks is not None:
            # data-parallel replicas: the forward capture registers the layer-boundary hooks, the backward capture turns each into a `mark = gen` kernel node of the
            # backward graph (engine/overlap.py); the communication stream follows them after the step's last backward replay (_reduce_flat_marked)
            self._marks.begin()
            self._marks.sink = self._mark_sink(lane if lane is not None else self._stage_lane)
        try:
            with torch.cuda.graph(fwd_graph, capture_error_mode=_capture_mode()):
                out = forward(static_in)
            static_gout = None if last else grads_like(out)
            _ops.WS_LANE = ('pipe-lane', lane['id']) if lane is not None else 'stage-bwd'
            with torch.cuda.graph(bwd_graph, pool=fwd_graph.pool(), capture_error_mode=_capture_mode()):
                backward(out, static_gout)
            if self._marks is not None:
                fired = self._marks.fired()
        finally:
            if self._marks is not None:
                self._marks.sink = None
        _ops.WS_LANE = None
        # undo the side effects of warm-up / capture on the accumulators
        for p in params:
            if p.grad is not None:
                if id(p) in saved_grads:
                    p.grad.copy_(saved_grads[id(p)])
                else:
                    p.grad.zero_()
        loss_acc.copy_(saved_loss)
        if lane is not None:
            for p in params:
                if p.grad is not None:
                    lane['grads'][id(p)] = p.grad                # persistent: their addresses are baked into this lane's backward graphs
        _offload.POOL_TAG = None
        return {'fwd': fwd_graph, 'bwd': bwd_graph, 'inputs': static_in, 'labels': static_lab, 'out': out, 'gout': static_gout,
                'single_in': single_in, 'fwd_done': None, 'bwd_done': None, 'marked': self._marks is not None, 'fired': fired}

    def _exec_forward_pass_graphed(self, buffer_id):
        inputs = self.pipe_buffers['inputs'][buffer_id]
        labels = self.pipe_buffers['labels'][buffer_id] if self.is_last_stage() else None
        slot = self._stage_slot(buffer_id, inputs, labels)
        cur = torch.cuda.current_stream(self.device)
        if slot['bwd_done'] is not None:
            cur.wait_event(slot['bwd_done'])            # the slot's previous micro-batch finished its backward
        with torch.no_grad():
            for dst, src in zip(slot['inputs'], _as_list(inputs)):
                if src.numel() > 0 and src.data_ptr() != dst.data_ptr():          # received in place: nothing to copy
                    dst.copy_(src, non_blocking=True)
            if labels is not None:
                for dst, src in zip(slot['labels'], _as_list(labels)):
                    if src.numel() > 0:
                        dst.copy_(src, non_blocking=True)
        slot['fwd'].replay()
        slot['fwd_done'] = cur.record_event()
        self._slot_of_buffer[buffer_id] = slot
        self.pipe_buffers['inputs'][buffer_id] = slot['inputs'][0] if slot['single_in'] else slot['inputs']   # their .grad feeds SendGrad
        self.pipe_buffers['outputs'][buffer_id] = slot['out']
        self.pipe_buffers['slot'][buffer_id] = slot

    def _exec_backward_pass_graphed(self, buffer_id):
        slot = self.pipe_buffers['slot'][buffer_id]
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot['fwd_done'])                # this micro-batch's forward (other stream) has run
        if not self.is_last_stage():
            grads = self.pipe_buffers['grads'][buffer_id]
            assert len(grads) == len(slot['gout']), \
                f'stage {self.stage_id}: {len(slot["gout"])} floating-point outputs but {len(grads)} received gradients'
            with torch.no_grad():
                for dst, src in zip(slot['gout'], grads):
                    if src.data_ptr() != dst.data_ptr():
                        dst.copy_(src, non_blocking=True)
            self.pipe_buffers['grads'][buffer_id] = None
        self._bwd_count += 1
        mlane = None
        if self._marks is not None:
            mlane = self._cur_pipe_lane if self._cur_pipe_lane is not None else self._stage_lane
            mlane['marked_step'] = bool(slot.get('marked')) and 'gen' in mlane
            mlane['fired_step'] = set(slot.get('fired', ()))
            if mlane['marked_step']:
                mlane['gen_host'] += 1                  # this replay's number: what its progress marks will hold
                mlane['gen'].fill_(mlane['gen_host'])
        slot['bwd'].replay()
        slot['bwd_done'] = cur.record_event()
        self.pipe_buffers['outputs'][buffer_id] = None
        if mlane is self._stage_lane and mlane is not None and self._bwd_count == self.micro_batches and self.is_data_parallel and self._stage_arena:
            # the step's last backward is launched: average the late layers' gradients over the replicas while it is still running
            self._stage_lane['arena'] = self._stage_arena
            self._stage_marked_stop = self._reduce_flat_marked([self._stage_lane], [p for p in self.module.parameters() if p.requires_grad])

    # ----------------------------------------------------------------------------------------- instructions
    def _next_batch(self):
        if self._data_iter is None:
            raise RuntimeError('first / last stage need a data iterator')
        return next(self._data_iter)

    def _exec_load_micro_batch(self, buffer_id):
        batch = self._next_batch()
        if self.is_first_stage():
            feats = batch[0]
            if torch.is_tensor(feats):
                loaded = feats.clone().detach().to(self.device)
            else:
                loaded = tuple(x.clone().detach().to(self.device) for x in feats)
            if getattr(self.module, 'activation_checkpoint_interval', 0) > 0 and not self._eval_mode:
                # like DeepSpeed's _exec_load_micro_batch under activation checkpointing: Function-style (reentrant) checkpoint
                # wrappers -- unsloth_checkpoint / offloaded_checkpoint -- only build a backward node when an input needs one
                for x in _as_list(loaded):
                    if x.is_floating_point():
                        x.requires_grad_(True)
            self.pipe_buffers['inputs'][buffer_id] = loaded
        if self.is_last_stage():
            labels = batch[1]
            if torch.is_tensor(labels):
                loaded = labels.to(self.device)
            elif isinstance(labels, (tuple, list)):
                loaded = tuple(x.to(self.device).detach() for x in labels)
            else:
                loaded = labels
            self.pipe_buffers['labels'][buffer_id] = loaded

    def _exec_forward_pass(self, buffer_id):
        if self.use_stage_graphs and not self._eval_mode:
            return self._exec_forward_pass_graphed(buffer_id)
        inputs = self.pipe_buffers['inputs'][buffer_id]
        if self._marks is not None and not self._eval_mode:
            # the step's LAST forward registers the backward progress marks (its backward is the step's last one: 1F1B runs a stage's backwards in forward order)
            self._fwd_count += 1
            self._marks.sink = None
            if self._fwd_count == self.micro_batches:
                self._marks.begin()
                self._marks.sink = self._eager_mark
        outputs = self.module(inputs)
        if self.is_last_stage():
            if self.module.loss_fn is not None:
                labels = self.pipe_buffers['labels'][buffer_id]
                self.loss = self.module.loss_fn(outputs, labels)
            else:
                self.loss = outputs
            if torch.is_tensor(self.loss):
                det = self.loss.detach()
                self.total_loss = det.clone() if self.total_loss is None else self.total_loss + det
            self.pipe_buffers['outputs'][buffer_id] = self.loss
        else:
            self.pipe_buffers['outputs'][buffer_id] = outputs

    def _exec_backward_pass(self, buffer_id):
        if self.use_stage_graphs:
            return self._exec_backward_pass_graphed(buffer_id)
        outputs = self.pipe_buffers['outputs'][buffer_id]
        self._bwd_count += 1
        if self.is_last_stage():
            (outputs / self.micro_batches).backward()
        else:
            out_tensors = [t for t in _as_list(outputs) if _is_float(t)]
            grad_tensors = self.pipe_buffers['grads'][buffer_id]
            assert len(out_tensors) == len(grad_tensors), \
                f'stage {self.stage_id}: {len(out_tensors)} floating-point outputs but {len(grad_tensors)} received gradients'
            torch.autograd.backward(tensors=out_tensors, grad_tensors=grad_tensors)
            self.pipe_buffers['grads'][buffer_id] = None
        self.pipe_buffers['outputs'][buffer_id] = None

    def _exec_send_activations(self, buffer_id):
        self.link.send_tuple(self.pipe_buffers['outputs'][buffer_id], self.next_stage, tag='act')
        if self._eval_mode:
            self.pipe_buffers['outputs'][buffer_id] = None

    def _exec_recv_activations(self, buffer_id):
        slot = self._slot_of_buffer.get(buffer_id) if (self.use_stage_graphs and not self._eval_mode) else None
        if slot is not None:         # receive straight into the slot's static inputs (no staging copy); ordered after the slot's previous backward
            recvd = self.link.recv_tuple(self.prev_stage, tag='act', into=slot['inputs'], after=slot['bwd_done'])
        else:
            recvd = self.link.recv_tuple(self.prev_stage, tag='act')
        if not self._eval_mode:
            for t in _as_list(recvd):
                t.requires_grad = t.is_floating_point()
        self.pipe_buffers['inputs'][buffer_id] = recvd

    def _exec_send_grads(self, buffer_id):
        inputs = self.pipe_buffers['inputs'][buffer_id]
        grads = []
        for t in _as_list(inputs):
            if _is_float(t):
                grads.append(t.grad if t.grad is not None else torch.zeros_like(t))
        self.link.send_plain(grads, self.prev_stage)
        self.pipe_buffers['inputs'][buffer_id] = None

    def _exec_recv_grads(self, buffer_id):
        outputs = self.pipe_buffers['outputs'][buffer_id]
        templates = [t for t in _as_list(outputs) if _is_float(t)]
        slot = self.pipe_buffers['slot'][buffer_id] if self.use_stage_graphs else None
        if slot is not None and slot['gout'] is not None:        # straight into the backward graph's static output-gradient buffers
            self.pipe_buffers['grads'][buffer_id] = self.link.recv_like(templates, self.next_stage, into=slot['gout'], after=slot['bwd_done'])
        else:
            self.pipe_buffers['grads'][buffer_id] = self.link.recv_like(templates, self.next_stage)

    def _exec_reduce_tied_grads(self):
        pass   # the reference's adapters register no tied layers

    def _dp_reduce_(self, chunk, group, async_op=False):
        """in-place data-parallel AVERAGE of one contiguous bucket.  One all-reduce with the averaging folded in (ReduceOp.AVG) where the backend has it (RCCL;
        the gloo of this torch build), else sum + scale; `communication_data_type` (DeepSpeed's knob, honoured by the bucketed path too) reduces a cast copy
        of the bucket and writes the result back.  `async_op` (host-side backends under the backward's tail): the collective is started and a `finish()` that
        waits for it and completes the average is returned."""
        comm_dt = self.communication_data_type
        buf = chunk if (comm_dt is None or comm_dt == chunk.dtype) else chunk.to(comm_dt)
        if self._dp_avg_ok is None:
            # decided ONCE, from the backend's name, before any collective is issued (probing ReduceOp.AVG by catching the exception of a live
            # collective lets a single failing rank issue a second all-reduce its peers never post -- a hang instead of an error).  RCCL / NCCL have AVG; gloo
            # takes sum + scale (exact for the power-of-two world sizes the CPU tests use, and never a capability question).
            self._dp_avg_ok = str(dist.get_backend(group)).lower() == 'nccl'
        avg = self._dp_avg_ok
        work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=async_op) if avg else dist.all_reduce(buf, group=group, async_op=async_op)

        def finish():
            if async_op:
                work.wait()
            if not avg:
                buf.div_(self.dp_world_size)
            if buf is not chunk:
                chunk.copy_(buf)
        if async_op:
            return finish
        finish()
        return None

    def _reduce_flat(self, base, others, stop=None):
        """Lane summation + data-parallel average over flat gradient arenas, bucket by bucket: bucket k's lane sum runs on the compute
        stream, its all-reduce on the communication stream behind an event -- so the all-reduce of bucket k overlaps the summation of bucket k + 1 and
        the xGMI links start moving data as soon as the first bucket is summed.  No staging concatenation, no copy back: the buckets are views of the
        arenas the wgrad kernels accumulated into.  base: {dtype: flat}; others: the other lanes' arenas (same layout); `stop` ({dtype: element count}):
        only arena[:stop] -- the tail was reduced under the backward (_reduce_flat_marked)."""
        group = self.grid.get_data_parallel_group() if self.is_data_parallel else None
        cuda = self.device.type == 'cuda'
        if cuda and group is not None and self._dp_stream is None:
            self._dp_stream = torch.cuda.Stream(self.device)
        cur = torch.cuda.current_stream(self.device) if cuda else None
        for dt, flat in base.items():
            step = max(1, self.dp_bucket_bytes // flat.element_size())
            end = flat.numel() if stop is None else min(flat.numel(), stop.get(dt, flat.numel()))
            for off in range(0, end, step):
                chunk = flat[off:min(off + step, end)]
                for o in others:
                    chunk.add_(o[dt][off:min(off + step, end)])
                if group is None:
                    continue
                if cuda:
                    ev = cur.record_event()
                    with torch.cuda.stream(self._dp_stream):
                        self._dp_stream.wait_event(ev)
                        self._dp_reduce_(chunk, group)
                else:
                    self._dp_reduce_(chunk, group)
        if cuda and group is not None:
            cur.wait_stream(self._dp_stream)

    def _lane_marks(self, lane):
        """The lane's progress marks: one device word per boundary (written by a kernel node of every graph the lane captures, include/dpipe_hip.h C5), the word
        that names the replay in flight (`gen`, set on the lane's stream before every replay) and its host copy."""
        if 'marks' not in lane:
            lane['marks'] = torch.zeros(len(self._marks.boundaries), dtype=torch.int32, device=self.device)
            lane['gen'] = torch.zeros(1, dtype=torch.int32, device=self.device)
            lane['gen_host'] = 0
        return lane['marks']

    def _mark_sink(self, lane):
        """Sink of BackwardMarks while a lane graph is being captured: boundary j -> a `mark = gen` kernel node on the capturing stream."""
        from .. import hip as _hip
        words = self._lane_marks(lane)
        index = {j: i for i, j in enumerate(self._marks.boundaries)}

        def sink(j):
            _hip.check(_hip.lib().dpipe_mark_post(words.data_ptr() + 4 * index[j], lane['gen'].data_ptr(), torch.cuda.current_stream(self.device).cuda_stream), 'mark_post')
        return sink

    def _reduce_flat_marked(self, lanes, params):
        """The data-parallel average UNDER the tail of the lanes' last replays (hipGraph lane path; called when every replay of the step has been launched, before
        the lanes are joined).  For the boundaries j in descending order: the communication stream waits for mark j of EVERY lane -- recorded inside the lane's graph
        once the backward has left layer j, i.e. arena[offset_j:] of that lane is final -- then sums the lanes' arena ranges into lane 0's and averages them over the
        replicas, while the graphs are still running the backward of the earlier layers.  Returns {dtype: element count still to do} for _reduce_flat, or None
        when the arenas are not laid out in layer order (nothing started)."""
        from .. import hip as _hip
        base = lanes[0]
        key = tuple(sorted((str(dt), l['arena'][dt].data_ptr()) for l in lanes for dt in l['arena']))
        cached = getattr(self, '_marked_bounds', None)
        if cached is None or cached[0] != key:
            per_lane = [self._marks.arena_bounds(params, l['arena'], grad_of=(lambda p, l=l: l['grads'].get(id(p))) if l.get('grads') else None) for l in lanes]
            bounds = per_lane[0] if all(b == per_lane[0] for b in per_lane[1:]) else None
            self._marked_bounds = cached = (key, bounds)
        bounds = cached[1]
        if not bounds or not any(bounds.values()) or any('marks' not in l or not l.get('marked_step', False) for l in lanes):
            return None
        if self._mark_err is None:
            self._mark_err = torch.zeros(1, dtype=torch.int32).pin_memory()       # written by a waiting kernel that gave up (a lost mark), read here one step later
        elif int(self._mark_err[0]) != 0:
            raise RuntimeError('data-parallel overlap: a progress mark of an earlier step never arrived (dpipe_mark_wait timed out); gradients of that step were reduced early')
        index = {j: i for i, j in enumerate(self._marks.boundaries)}
        group = self.grid.get_data_parallel_group()
        if self._dp_stream is None:
            self._dp_stream = torch.cuda.Stream(self.device)
        dps = self._dp_stream
        stop = {dt: a.numel() for dt, a in base['arena'].items()}
        report = {'path': 'graph lanes' if self.use_graph else 'stage graphs', 'marks': [], 'early_collectives': 0, 'early_bytes': 0, 'total_bytes': sum(a.numel() * a.element_size() for a in base['arena'].values())}
        lib = _hip.lib()
        usable = set.intersection(*[l.get('fired_step', set()) for l in lanes])
        for j in sorted(bounds, reverse=True):
            offs = {dt: off for dt, off in bounds[j].items() if off < stop[dt]}
            if not offs or j not in usable:
                continue
            for l in lanes:       # the replay numbered gen_host is the lane's last one of this step: behind its mark j the lane's arena[offset_j:] is final
                _hip.check(lib.dpipe_mark_wait(l['marks'].data_ptr() + 4 * index[j], l['gen_host'] & 0xffffffff, self._mark_err.data_ptr(), self.dp_mark_timeout_ms,
                                               dps.cuda_stream), 'mark_wait')
            with torch.cuda.stream(dps):
                for dt, off in offs.items():
                    flat = base['arena'][dt]
                    step = max(1, self.dp_bucket_bytes // flat.element_size())
                    for o0 in range(off, stop[dt], step):
                        o1 = min(o0 + step, stop[dt])
                        chunk = flat[o0:o1]
                        for l in lanes[1:]:
                            chunk.add_(l['arena'][dt][o0:o1])
                        self._dp_reduce_(chunk, group)
                        report['early_collectives'] += 1
                        report['early_bytes'] += (o1 - o0) * flat.element_size()
                    stop[dt] = off
            report['marks'].append(j)
            if len(report['marks']) == 1:
                first_done = torch.cuda.Event(enable_timing=True)
                first_done.record(dps)
        if report['marks']:
            last_done = torch.cuda.Event(enable_timing=True)
            last_done.record(dps)
            self._overlap_events = [first_done, last_done, None]
        self.overlap_report = report
        return stop

    def overlap_lead_ms(self):
        """(hipGraph lane path, after a step; synchronises) how long before the lanes' graphs had all finished the average of the FIRST / LAST marked gradient range
        was complete on the communication stream: positive = that much of the reduction ran under the backward."""
        ev = self._overlap_events
        if not ev or ev[2] is None:
            return None
        ev[2].synchronize()
        return round(ev[0].elapsed_time(ev[2]), 3), round(ev[1].elapsed_time(ev[2]), 3)

    def _eager_mark(self, j):
        """Sink of BackwardMarks on the eager path (called from inside the step's last backward): the gradients of the layers >= j are final -- start their average."""
        if self._bwd_count != self.micro_batches or not self.is_data_parallel:
            return
        st = self._early
        if st is None:
            st = self._early = {'done': set(), 'finish': [], 'marks': [], 'collectives': 0, 'bytes': 0, 'streamed': False}
        layer_of = self._marks.layer_of
        grads = []
        for p in self._trainable_params():
            if p.grad is not None and id(p) not in st['done'] and layer_of.get(id(p), -1) >= j:
                st['done'].add(id(p))
                grads.append(p.grad)
        st['marks'].append(j)
        if grads:
            self._reduce_grad_tensors(grads, self.grid.get_data_parallel_group(), early=st)

    def _reduce_grad_tensors(self, grads, group, early=None):
        """Average gradient tensors over the replicas in place: a gradient of dp_direct_min_bytes or more as its own collective, the small ones through staged buckets.
        `early` (the eager overlap state): the collectives are STARTED here -- on the communication stream behind an event (GPU) or as asynchronous operations
        (gloo) -- and completed by _exec_reduce_grads."""
        cuda = self.device.type == 'cuda'
        by_dtype = OrderedDict()
        for g in grads:
            dt = self.communication_data_type or g.dtype
            by_dtype.setdefault((dt, g.dtype), []).append(g)
        async_op = early is not None and not cuda
        ctx = None
        if early is not None and cuda:
            if self._dp_stream is None:
                self._dp_stream = torch.cuda.Stream(self.device)
            ev = torch.cuda.current_stream(self.device).record_event()
            self._dp_stream.wait_event(ev)
            ctx = torch.cuda.stream(self._dp_stream)
            ctx.__enter__()
            early['streamed'] = True
        try:
            for (comm_dt, _), gs in by_dtype.items():
                # (round 6) a gradient of dp_direct_min_bytes or more is averaged IN PLACE, as its own collective: no concatenation, no copy back --
                # staging is for the small tensors only (biases, norm weights: a collective each would be latency-bound).  Every rank walks the same parameter list, so the
                # sequence of collectives is the same everywhere.
                small = []
                for g in gs:
                    if g.is_contiguous() and g.numel() * g.element_size() >= self.dp_direct_min_bytes:
                        fin = self._dp_reduce_(g.view(-1), group, async_op=async_op)
                        if early is not None:
                            early['collectives'] += 1; early['bytes'] += g.numel() * g.element_size()
                            if fin is not None:
                                early['finish'].append(fin)
                    else:
                        small.append(g)
                bucket, size = [], 0
                for g in small + [None]:
                    if g is not None:
                        bucket.append(g); size += g.numel() * g.element_size()
                    if bucket and (g is None or size >= self.dp_bucket_bytes):
                        flat = torch.cat([b.reshape(-1).to(comm_dt) for b in bucket])
                        fin = self._dp_reduce_(flat, group, async_op=async_op)

                        def copy_back(flat=flat, bucket=bucket, fin=fin):
                            if fin is not None:
                                fin()
                            off = 0
                            for b in bucket:
                                b.copy_(flat[off:off + b.numel()].view_as(b)); off += b.numel()
                        if async_op:
                            early['finish'].append(copy_back)
                        else:
                            copy_back()
                        if early is not None:
                            early['collectives'] += 1; early['bytes'] += size
                        bucket, size = [], 0
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def _exec_reduce_grads(self, skip_storages=None):
        """Data-parallel gradient average.  Persistent-gradient paths keep the gradients in flat arenas (flatten_grads) and reduce those in
        place, in buckets sized for xGMI / 288 GB HBM; gradients outside an arena (eager path: autograd allocates them per step) are averaged in place when
        large and bucketed through a staging concatenation when small.  What the marks of the step's last backward already started (eager path, _eager_mark)
        is completed first and not repeated."""
        if self._marks is not None:
            self._marks.sink = None
        if not self.is_data_parallel:
            return
        group = self.grid.get_data_parallel_group()
        if self._overlap_events is not None and self._overlap_events[2] is None and self.device.type == 'cuda':
            joined = torch.cuda.Event(enable_timing=True)       # stage-graph path: the step's last backward has been joined into the caller's stream by now
            joined.record(torch.cuda.current_stream(self.device))
            self._overlap_events[2] = joined
        early, self._early = self._early, None
        done = set()
        if early is not None:
            for fin in early['finish']:
                fin()
            if early['streamed']:
                torch.cuda.current_stream(self.device).wait_stream(self._dp_stream)
            done = early['done']
            self.overlap_report = {'path': 'eager', 'marks': early['marks'], 'early_collectives': early['collectives'], 'early_bytes': early['bytes'],
                                   'total_bytes': sum(p.grad.numel() * p.grad.element_size() for p in self._trainable_params() if p.grad is not None)}
        if skip_storages is None and self._stage_arena:
            self._reduce_flat(self._stage_arena, [], stop=self._stage_marked_stop)
            skip_storages = {a.untyped_storage().data_ptr() for a in self._stage_arena.values()}
        grads = [p.grad for p in self._trainable_params()
                 if p.grad is not None and id(p) not in done and not (skip_storages and p.grad.untyped_storage().data_ptr() in skip_storages)]
        if grads:
            self._reduce_grad_tensors(grads, group)

    def clip_fp32_gradients(self):
        """Reference `clip_grad_norm_` (utils/patches.py:175-246) on the HIP multi-tensor kernels, no host sync."""
        params = [p for p in self.module.parameters() if p.grad is not None]
        if self.clip_grad_fn is not None:
            self._last_grad_norm = self.clip_grad_fn(params, self._gradient_clipping, mpu=self.mpu)
            return
        ops = self.grad_kernels
        if ops is None:
            from .. import ops          # HIP multi-tensor kernels; raises on CPU tensors (no fallback)
        grads = [p.grad for p in params]
        counted = grads
        if self.is_pipe_parallel and self.clip_norm_scope == 'deepspeed' and self.stage_id != 0:
            counted = []
        if counted:
            sumsq = ops.grads_sumsq(counted)
        else:
            sumsq = torch.zeros((), device=self.device, dtype=torch.float32)
        if self.is_pipe_parallel:
            dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=self.grid.get_model_parallel_group())
        if self.is_data_parallel:
            norm = sumsq.sqrt() / float(self.dp_world_size)
            dist.all_reduce(norm, group=self.grid.get_data_parallel_group())
            sumsq = norm * norm
        if grads:
            ops.grads_clip_scale_(grads, sumsq, self._gradient_clipping)
        self._last_grad_norm = sumsq.sqrt()

    def _fused_step_end(self):
        """True when the optimizer brings the fused HIP step end (optim.FusedAdamW) and nothing overrides the clip."""
        return (self.optimizer is not None and hasattr(self.optimizer, 'fused_update') and self.clip_grad_fn is None
                and self.grad_kernels is None and self.device.type == 'cuda')

    def _fused_optimizer_step(self, lane_grads, zero_lane_grads=True):
        """Norm of the (lane-summed) gradients -> the reference's cross-stage / DP composition of the scalar
        (utils/patches.py:222-239) -> one pass: clip, AdamW, zero."""
        opt = self.optimizer
        persistent = self.use_graph or self.use_stage_graphs
        sumsq = None
        if self._gradient_clipping > 0.0:
            counted = not (self.is_pipe_parallel and self.clip_norm_scope == 'deepspeed' and self.stage_id != 0)
            sumsq = opt.grads_sumsq(lane_grads) if counted else None
            if sumsq is None:
                sumsq = torch.zeros((), device=self.device, dtype=torch.float32)
            if self.is_pipe_parallel:
                dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=self.grid.get_model_parallel_group())
            if self.is_data_parallel:
                norm = sumsq.sqrt() / float(self.dp_world_size)
                dist.all_reduce(norm, group=self.grid.get_data_parallel_group())
                sumsq = norm * norm
            self._last_grad_norm = sumsq.sqrt()
        opt.fused_update(lane_grads, sumsq, self._gradient_clipping, zero_grads=persistent and zero_lane_grads)

    def _exec_optimizer_step(self, lr_kwargs=None, lane_grads=None, zero_lane_grads=True):
        if self._fused_step_end():
            self._fused_optimizer_step(lane_grads, zero_lane_grads)
            if not (self.use_graph or self.use_stage_graphs):
                for p in self.module.parameters():
                    p.grad = None
            if self.lr_scheduler is not None:
                self.lr_scheduler.step(**(lr_kwargs or {}))
            self.global_steps += 1
            return
        if self._gradient_clipping > 0.0:
            self.clip_fp32_gradients()
        if self.optimizer is not None:
            self.optimizer.step()
        if self.use_graph or self.use_stage_graphs:
            grads = [p.grad for p in self.module.parameters() if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)         # buffers are referenced by the captured graphs: zero, never free
        else:
            if self.optimizer is not None:
                self.optimizer.zero_grad()
            for p in self.module.parameters():
                p.grad = None
        if self.lr_scheduler is not None:
            self.lr_scheduler.step(**(lr_kwargs or {}))
        self.global_steps += 1

    _INSTRUCTION_MAP = {
        sched.OptimizerStep: _exec_optimizer_step,
        sched.ReduceGrads: _exec_reduce_grads,
        sched.ReduceTiedGrads: _exec_reduce_tied_grads,
        sched.LoadMicroBatch: _exec_load_micro_batch,
        sched.ForwardPass: _exec_forward_pass,
        sched.BackwardPass: _exec_backward_pass,
        sched.SendActivation: _exec_send_activations,
        sched.RecvActivation: _exec_recv_activations,
        sched.SendGrad: _exec_send_grads,
        sched.RecvGrad: _exec_recv_grads,
    }

    # ------------------------------------------------------------------------------------------------ loss
    def _aggregate_total_loss(self, micro_batches):
        """mean over micro-batches -> mean over DP replicas -> broadcast from the last stage."""
        if self.is_last_stage():
            if self.total_loss is None:
                loss = torch.zeros((), device=self.device, dtype=torch.float32)
            else:
                loss = (self.total_loss / micro_batches).to(torch.float32)
            if self.is_data_parallel:
                loss = loss.clone()
                dist.all_reduce(loss, group=self.grid.get_data_parallel_group())
                loss /= self.dp_world_size
        else:
            loss = torch.zeros((), device=self.device, dtype=torch.float32)
        if self.is_pipe_parallel:
            loss = loss.clone().reshape(1)
            src = self.grid.stage_to_global(self.num_stages - 1)
            dist.broadcast(loss, src=src, group=self.grid.get_pipe_parallel_group())
            loss = loss.reshape(())
        return loss

    # ------------------------------------------------------------------------------------------ checkpoints
    def _ckpt_tag(self):
        return f'global_step{self.global_steps}'

    def save_checkpoint(self, save_dir, tag=None, client_state=None, save_latest=True, exclude_frozen_parameters=False):
        """Layout: save_dir/<tag>/layer_XX-model_states.pt (one per local layer, written by DP rank 0) and
        mp_rank_XX_model_states.pt (optimizer, lr scheduler, client_state) + save_dir/latest (utils/saver.py:118-128)."""
        tag = tag or self._ckpt_tag()
        path = os.path.join(save_dir, tag)
        os.makedirs(path, exist_ok=True)
        if self.grid.get_data_parallel_rank() == 0:
            start, _ = self.module.local_layer_range()
            for local_idx, layer in enumerate(self.module.forward_funcs):
                if not isinstance(layer, torch.nn.Module):
                    continue
                sd = OrderedDict()
                trainable = {n for n, p in layer.named_parameters() if p.requires_grad}
                for k, v in layer.state_dict().items():
                    if exclude_frozen_parameters and k not in trainable:
                        continue
                    sd[k] = v.detach().cpu()
                torch.save(sd, os.path.join(path, f'layer_{local_idx + start:02d}-model_states.pt'))
            state = {
                'optimizer': self.optimizer.state_dict() if self.optimizer is not None else None,
                'lr_scheduler': self.lr_scheduler.state_dict() if self.lr_scheduler is not None else None,
                'global_steps': self.global_steps, 'global_samples': self.global_samples,
                'client_state': dict(client_state or {}), 'num_stages': self.num_stages,
            }
            torch.save(state, os.path.join(path, f'mp_rank_{self.stage_id:02d}_model_states.pt'))
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        if save_latest and self.global_rank == 0:
            with open(os.path.join(save_dir, 'latest'), 'w') as f:
                f.write(tag)
        return True

    def load_checkpoint(self, load_dir, tag=None, load_module_strict=True, load_optimizer_states=True,
                        load_lr_scheduler_states=True, load_module_only=False):
        if tag is None:
            latest = os.path.join(load_dir, 'latest')
            if not os.path.isfile(latest):
                return None, None
            with open(latest) as f:
                tag = f.read().strip()
        path = os.path.join(load_dir, tag)
        start, _ = self.module.local_layer_range()
        for local_idx, layer in enumerate(self.module.forward_funcs):
            if not isinstance(layer, torch.nn.Module):
                continue
            fn = os.path.join(path, f'layer_{local_idx + start:02d}-model_states.pt')
            if os.path.isfile(fn):
                layer.load_state_dict(torch.load(fn, map_location='cpu'), strict=load_module_strict)
        state = torch.load(os.path.join(path, f'mp_rank_{self.stage_id:02d}_model_states.pt'), map_location='cpu', weights_only=False)
        if not load_module_only:
            if load_optimizer_states and self.optimizer is not None and state.get('optimizer') is not None:
                self.optimizer.load_state_dict(state['optimizer'])
            if load_lr_scheduler_states and self.lr_scheduler is not None and state.get('lr_scheduler') is not None:
                self.lr_scheduler.load_state_dict(state['lr_scheduler'])
        self.global_steps = state.get('global_steps', 0)
        self.global_samples = state.get('global_samples', 0)
        return path, state.get('client_state', {})


def initialize(args=None, model=None, optimizer=None, model_parameters=None, training_data=None, lr_scheduler=None,
               config=None, config_params=None, device=None, **kwargs):
    """Same call shape and return tuple as `deepspeed.initialize` (train.py:623-627):
    returns (engine, optimizer, training_dataloader=None, lr_scheduler)."""
    cfg = config if config is not None else config_params
    engine = PipelineEngine(module=model, config=cfg, args=args, optimizer=optimizer, lr_scheduler=lr_scheduler,
                            model_parameters=model_parameters, device=device)
    return engine, engine.optimizer, None, engine.lr_scheduler
