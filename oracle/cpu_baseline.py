"""ORACLE / CPU baseline (test infrastructure; never imported by the product): times the reference-style fp32 eager path on the host
cores for bench.py's `cpu_baseline` object.

The reference cannot execute on CPU (flash-attn / DeepSpeed / CUDA-only), so the baseline is the oracle
restatement (`"kind": "port"`): `oracle/sdxl_ref.SDXLRef` at the FULL BASELINE configuration (2.6 B parameters, fp32) driven by
`oracle/eager_step.eager_train_step` -- sequential `to_layers()` + loss + backward, the path the parity tests compare the HIP engine with.
BOUNDED sample: ONE micro-batch (= one 1024 x 1024 image) of the step's GAS, forward + loss + backward + clip, no optimizer step;
images/s = 1 / sample_seconds.  (Round 1 timed three layers of one up-block and scaled by FLOPs; this is a whole image.)
"""
import os
import time

import torch

from . import eager_step, sdxl_ref


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown.  The GPU box reports 256 hardware threads and a quota of 16."""
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(p)
    except Exception:                                    # noqa: BLE001
        pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / p
    except Exception:                                    # noqa: BLE001
        return None


def default_threads():
    """threads of the oracle's CPU legs: DPIPE_CPU_BASELINE_THREADS if set; else at most 32 (these mid-sized fp32 ops run fastest on a subset of a many-core host) and at most
    the container's CPU quota -- on the GPU box (256 hardware threads, quota 16) 16 threads evaluate one sample in 18.6 s, 32 threads in 26.4 - 26.9 s (call r6i: throttled
    OpenMP teams wait at every barrier)."""
    if 'DPIPE_CPU_BASELINE_THREADS' in os.environ:
        return max(1, min(os.cpu_count() or 1, int(os.environ['DPIPE_CPU_BASELINE_THREADS'])))
    n = min(os.cpu_count() or 1, 32)
    q = cpu_quota_cores()
    if q:
        n = max(1, min(n, int(q + 0.999)))
    return n


def _bf16_exact(v):
    return v.is_floating_point() and bool((v.to(torch.bfloat16).to(v.dtype) == v).all())


def spawn_parity_workers(cfg, state, micro_batches, workers, threads):
    """Evaluate `micro_batches` on `state` in `workers` child PROCESSES (round 6: 16 parity samples at ~25 s of host time each do not fit one background thread inside the
    bench's few minutes; the GPU box has 256 hardware threads and the oracle runs fastest on ~32).  The weights travel once through /dev/shm (as bf16 where that is exact:
    the product's parameters are bf16 values), worker w takes samples w, w + workers, ...  -> join() returning ([loss...], [grad_norm...]) in sample order."""
    import subprocess
    import sys
    path = f'/dev/shm/dpipe_parity_{os.getpid()}.pt'
    packed = {k: {n: (v.to(torch.bfloat16) if _bf16_exact(v) else v) for n, v in m.items()} for k, m in state.items()}
    torch.save({'cfg': cfg, 'state': packed, 'mbs': [(tuple(t.cpu() for t in mb[0]), tuple(t.cpu() for t in mb[1])) for mb in micro_batches]}, path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, '-m', 'oracle.cpu_baseline', path, str(w), str(workers), str(threads)], stdout=subprocess.PIPE, text=True, cwd=root)
             for w in range(min(workers, len(micro_batches)))]

    def join():
        import json
        got = {}
        try:
            for pr in procs:
                out, _ = pr.communicate()
                for ln in out.splitlines():
                    if ln.startswith('{"parity_worker"'):
                        got.update({int(k): v for k, v in json.loads(ln)['results'].items()})
        finally:
            if os.path.exists(path):
                os.remove(path)
        idx = sorted(got)
        if idx != list(range(len(idx))):                     # a worker died: keep the contiguous prefix (the statistic says how many samples it holds)
            idx = [i for i in range(len(micro_batches)) if all(j in got for j in range(i + 1))]
        return [got[i][0] for i in idx], [got[i][1] for i in idx]
    return join


def _parity_worker_main(path, w, workers, threads):
    import json
    torch.set_num_threads(threads)
    blob = torch.load(path, weights_only=False)
    ref = sdxl_ref.SDXLRef(blob['cfg'], seed=0)
    for k, m in ref.modules().items():
        m.load_state_dict({n: v.to(torch.float32) if v.is_floating_point() else v for n, v in blob['state'][k].items()})
    layers = ref.to_layers()
    res = {}
    for i in range(w, len(blob['mbs']), workers):
        for p_ in ref.parameters():
            p_.grad = None
        l_, n_ = eager_step.eager_train_step(layers, eager_step.sdxl_loss_fn(), [blob['mbs'][i]], None, gradient_clipping=1.0, params=ref.parameters())
        res[i] = [float(l_), float(n_)]
    print(json.dumps({'parity_worker': w, 'results': res}), flush=True)


def sdxl_cpu_baseline(cfg, latent_hw=128, threads=None, micro_batch=None, state=None, per_parameter=False, extra_micro_batches=(), extra_budget_s=0.0, extra_async=False,
                      extra_workers=0):
    """-> cpu_baseline dict.  `micro_batch`: one prepared (features, label) pair as the engine receives it.  `state`: {module name: state dict}
    of the PRODUCT's weights (host tensors) -- loaded into the oracle so that its loss / gradient norm are comparable with the GPU path's on the
    same micro-batch (bench.py's `parity` object); None = the oracle's own seeded initialisation.
    `extra_micro_batches` (round 5: parity as a statistic): further micro-batches evaluated on the same weights AFTER the timed sample, as many as fit
    `extra_budget_s` seconds of host time (each costs about `sample_seconds`); their losses / pre-clip gradient norms come back as `loss_all` / `grad_norm_all`
    (entry 0 = the timed sample).  `extra_async`: the extra samples are evaluated on a background thread (the caller goes on with GPU work; torch's CPU kernels release
    the GIL) -- the dict then carries `extra_join`, a callable that waits for the thread and returns (loss_all, grad_norm_all); the timed first sample is never concurrent
    with anything.  `extra_workers` > 0 (round 6): the extra samples go to that many child processes instead (`spawn_parity_workers`), started AFTER the timed sample; this
    process frees its own copy of the model first."""
    # many-core hosts (the GPU box has 256 hardware threads) run these mid-sized fp32 ops fastest on a subset
    threads = threads or default_threads()
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    extra_thread = None
    try:
        t_build = time.perf_counter()
        ref = sdxl_ref.SDXLRef(cfg, seed=0)
        if state is not None:
            for k, m in ref.modules().items():
                m.load_state_dict({n: v.to(torch.float32) for n, v in state[k].items()})
        layers = ref.to_layers()
        assert micro_batch is not None, 'pass one prepared (features, label) micro-batch'
        micro_batch = (tuple(t.cpu() for t in micro_batch[0]), tuple(t.cpu() for t in micro_batch[1]))
        t_build = time.perf_counter() - t_build
        t0 = time.perf_counter()
        loss, norm = eager_step.eager_train_step(layers, eager_step.sdxl_loss_fn(), [micro_batch], None, gradient_clipping=1.0, params=ref.parameters())
        dt = time.perf_counter() - t0
        loss_all, norm_all = [float(loss)], [float(norm)]

        def run_extras():
            t_extra = time.perf_counter()
            for mb in extra_micro_batches:
                if time.perf_counter() - t_extra + dt > extra_budget_s or per_parameter:       # the next sample would overrun the budget
                    break
                for p_ in ref.parameters():
                    p_.grad = None
                mb = (tuple(t.cpu() for t in mb[0]), tuple(t.cpu() for t in mb[1]))
                l_, n_ = eager_step.eager_train_step(layers, eager_step.sdxl_loss_fn(), [mb], None, gradient_clipping=1.0, params=ref.parameters())
                loss_all.append(float(l_)); norm_all.append(float(n_))
            return loss_all, norm_all
        worker_join = None
        if extra_workers > 0 and extra_micro_batches and not per_parameter:
            del ref, layers
            worker_join = spawn_parity_workers(cfg, state, list(extra_micro_batches), extra_workers, threads)
        elif extra_async and extra_micro_batches and not per_parameter:
            import threading
            extra_thread = threading.Thread(target=run_extras, name='oracle-parity-samples', daemon=True)
            extra_thread.start()
        else:
            run_extras()
        rows = None
        if per_parameter:          # tools/parity_probe.py: [sum |g|, sum g, <g, r>, ||g||_2] of every parameter's PRE-clip gradient (the step clipped in place)
            from .checksums import checksum4
            coef = min(1.0, 1.0 / (float(norm) + 1e-6))
            rows = {f'{k}.{n}': [v / coef for v in checksum4(p.grad, f'{k}.{n}')] for k, m in ref.modules().items() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        if extra_thread is None:
            torch.set_num_threads(prev)

    def extra_join():
        if worker_join is not None:
            l_, n_ = worker_join()
            return loss_all + l_, norm_all + n_
        extra_thread.join()
        torch.set_num_threads(prev)
        return loss_all, norm_all
    return {**({'rows': rows} if per_parameter else {}), **({'extra_join': extra_join} if extra_thread is not None or worker_join is not None else {}), 'value': round(1.0 / dt, 6), 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'sample_seconds': round(dt, 3),
            'host': {'hardware_threads': os.cpu_count(), 'cgroup_cpu_quota_cores': cpu_quota_cores(), 'threads_used': threads},
            'build_seconds': round(t_build, 1), 'loss': float(loss), 'grad_norm': float(norm), 'loss_all': loss_all, 'grad_norm_all': norm_all, 'weights': 'product state dict' if state is not None else 'oracle seed 0',
            'sample': f'oracle fp32 eager path (oracle/sdxl_ref.py + eager_step.py): ONE whole micro-batch = one {latent_hw * 8}x{latent_hw * 8} image '
                      f'through all 23 pipeline layers + loss + backward + clip (1 of the step\'s micro-batches, no optimizer step), {threads} threads'}


def dit_block_cpu_baseline(kind, step_flops_per_sample, threads=None):
    """cpu_baseline dict for the DiT workloads (BASELINE configs 3 - 5), where a whole fp32 step does not fit the bound of ~10 - 30 s of host work
    (14 B / 13 B parameters in fp32 exceed the host memory of the build box, a HunyuanVideo step is ~11 PFLOP): ONE transformer
    block of the oracle, forward + backward in fp32, at the workload's real width -- Wan / Flux at the real token count, HunyuanVideo at 1/8 of its
    61 456 tokens -- gives the host's sustained FLOP rate on this arithmetic; samples/s = that rate / the step's algorithmic FLOPs per sample.  The
    extrapolation is stated in `sample`."""
    import torch.nn.functional as F
    from . import blocks_ref as br, flux_ref
    threads = threads or default_threads()
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    try:
        if kind == 'wan':
            dim, ffn, heads, S, L = 5120, 13824, 40, 9216, 512
            d = dim // heads
            shapes = {'modulation': (1, 6, dim), 'norm3.weight': (dim,), 'norm3.bias': (dim,), 'ffn.0.weight': (ffn, dim), 'ffn.0.bias': (ffn,),
                      'ffn.2.weight': (dim, ffn), 'ffn.2.bias': (dim,)}
            for a in ('self_attn', 'cross_attn'):
                for n in 'qkvo':
                    shapes[f'{a}.{n}.weight'], shapes[f'{a}.{n}.bias'] = (dim, dim), (dim,)
                shapes[f'{a}.norm_q.weight'], shapes[f'{a}.norm_k.weight'] = (dim,), (dim,)
            p = {k: (torch.randn(s, generator=g) / (s[-1] ** 0.5 if len(s) > 1 else 4.0)).requires_grad_(True) for k, s in shapes.items()}
            x = torch.randn(1, S, dim, generator=g).requires_grad_(True)
            e, ctx = torch.randn(1, 1, 6, dim, generator=g) * 0.3, torch.randn(1, L, dim, generator=g)
            ang = torch.rand(S, d // 2, generator=g) * 6.28
            cos, sin = torch.cos(ang), torch.sin(ang)
            fwd = 2 * S * dim * dim * 6 + 2 * 2 * L * dim * dim + 4 * S * S * dim + 4 * S * L * dim + 2 * 2 * S * dim * ffn
            what = f'ONE Wan2.1-14B block (dim {dim}, ffn {ffn}, {heads} heads, {S} video + {L} text tokens = the real shape; 1 of 40 layers)'

            def run():
                br.wan_block(p, x, e, ctx, heads, cos, sin, 1e-6).square().mean().backward()
        elif kind == 'flux':
            dim, heads, hd, S, L = 3072, 24, 128, 4096, 512
            blk = flux_ref.FluxTransformerBlock(dim, heads, hd)
            x = torch.randn(1, S, dim, generator=g).requires_grad_(True)
            c = torch.randn(1, L, dim, generator=g).requires_grad_(True)
            temb = torch.randn(1, dim, generator=g)
            ang = torch.rand(S + L, hd // 2, generator=g) * 6.28
            rot = (torch.cos(ang).repeat_interleave(2, dim=1), torch.sin(ang).repeat_interleave(2, dim=1))
            T = S + L
            fwd = 2 * T * dim * dim * 4 + 4 * T * T * dim + 2 * 2 * T * dim * 4 * dim + 2 * 2 * 6 * dim * dim
            what = f'ONE Flux.1-dev double-stream block (dim {dim}, {heads} x {hd}, {S} image + {L} text tokens = the real shape; 1 of 19 + 38 blocks)'

            def run():
                e_out, x_out = blk(x, c, temb, rot)
                (x_out.square().mean() + e_out.square().mean()).backward()
        elif kind == 'hv':
            from . import hv_ref
            dim, heads, S, L = 3072, 24, 7680, 256
            blk = hv_ref.MMSingleStreamBlock(dim, heads, 4.0)
            T = S + L
            x = torch.randn(1, T, dim, generator=g).requires_grad_(True)
            vec = torch.randn(1, dim, generator=g)
            ang = torch.rand(S, dim // heads // 2, generator=g) * 6.28
            cos, sin = torch.cos(ang), torch.sin(ang)
            pd = blk.pdict()
            fwd = 2 * T * dim * (3 * dim + 4 * dim) + 4 * T * T * dim + 2 * T * 5 * dim * dim + 2 * 3 * dim * dim
            what = (f'ONE HunyuanVideo single-stream block (dim {dim}, {heads} x 128) at {S} video + {L} text tokens = 1/8 of the 61 456-token sequence '
                    f'of config 5 (attention is 64 x smaller than at full length: the full-length block alone is ~180 TFLOP); 1 of 20 + 40 blocks')

            def run():
                br.mm_single_block(pd, x, vec, L, heads, cos, sin, None).square().mean().backward()
        else:
            raise ValueError(kind)
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    sample_flops = 3.0 * fwd
    rate = sample_flops / dt
    return {'value': rate / step_flops_per_sample, 'unit': 'samples/s', 'cores': threads, 'kind': 'port', 'sample_seconds': round(dt, 3),
            'host_tflops': round(rate / 1e12, 3), 'sample_tflop': round(sample_flops / 1e12, 2),
            'sample': f'oracle fp32 eager path, {what}, forward + backward, {threads} threads; value = measured host FLOP rate / the step\'s '
                      f'algorithmic FLOPs per sample ({step_flops_per_sample / 1e12:.1f} TFLOP) -- an extrapolation, the whole fp32 step does not fit the bound'}


if __name__ == '__main__':          # python -m oracle.cpu_baseline <blob> <worker> <workers> <threads>   (a parity worker of spawn_parity_workers)
    import sys
    _parity_worker_main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
