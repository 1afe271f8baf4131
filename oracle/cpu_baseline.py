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


def sdxl_cpu_baseline(cfg, latent_hw=128, threads=None, micro_batch=None, state=None):
    """-> cpu_baseline dict.  `micro_batch`: one prepared (features, label) pair as the engine receives it.  `state`: {module name: state dict}
    of the PRODUCT's weights (host tensors) -- loaded into the oracle so that its loss / gradient norm are comparable with the GPU path's on the
    same micro-batch (bench.py's `parity` object); None = the oracle's own seeded initialisation."""
    # many-core hosts (the GPU box has 256 hardware threads) run these mid-sized fp32 ops fastest on a subset
    threads = threads or min(os.cpu_count() or 1, int(os.environ.get('DPIPE_CPU_BASELINE_THREADS', '32')))
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
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
    finally:
        torch.set_num_threads(prev)
    return {'value': round(1.0 / dt, 6), 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'sample_seconds': round(dt, 3),
            'build_seconds': round(t_build, 1), 'loss': float(loss), 'grad_norm': float(norm), 'weights': 'product state dict' if state is not None else 'oracle seed 0',
            'sample': f'oracle fp32 eager path (oracle/sdxl_ref.py + eager_step.py): ONE whole micro-batch = one {latent_hw * 8}x{latent_hw * 8} image '
                      f'through all 23 pipeline layers + loss + backward + clip (1 of the step\'s micro-batches, no optimizer step), {threads} threads'}
