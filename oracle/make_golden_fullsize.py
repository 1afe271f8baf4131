"""ORACLE fixture generator (test infrastructure): BASELINE config 2 at FULL size through the reference-style CPU path -- SDXL 1024x1024
(latent 128), the full `SDXLConfig()` (2.6 B parameters), ONE micro-batch of one image: oracle/sdxl_ref.py driven by
oracle/eager_step.eager_train_step (sequential to_layers() + SDXL loss + backward, fp32) on seeded weights and a seeded prepared input.
Records loss, the global gradient norm and per-parameter gradient checksums (sum |g|, sum g, a seeded projection <g, r>, ||g||_2:
oracle/checksums.py) in tests/golden/sdxl_fullsize.json -- no
tensors: weights and inputs are rebuilt from the seeds (a weight checksum guards the RNG stream).  ~25 GB of host memory, minutes of CPU.

    python oracle/make_golden_fullsize.py
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import eager_step, sdxl_ref                             # noqa: E402
from oracle.checksums import checksum4                               # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
WEIGHT_SEED, DATA_SEED, PREP_SEED = 0, 100, 1234


def build(device='cpu'):
    """-> (product workload with fp32 weights on `device`, its prepared one-image micro-batch on the host)"""
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.float32, seed=WEIGHT_SEED, device=device)
    torch.manual_seed(PREP_SEED)
    feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=1, latent_hw=128, seed=DATA_SEED))
    return cfg, work, split_batch((feats, label), 1)


def weight_checksum(modules):
    return float(sum(p.detach().double().abs().sum() for m in modules.values() for p in m.parameters()))


def main():
    t0 = time.time()
    cfg, work, micro = build()
    ref = sdxl_ref.SDXLRef(cfg, seed=1)
    for k, m in ref.modules().items():
        m.load_state_dict({n: v.detach().contiguous() for n, v in work.modules()[k].state_dict().items()})
    wsum = weight_checksum(ref.modules())
    del work
    print(f'built in {time.time() - t0:.0f} s, weight checksum {wsum:.6e}', flush=True)
    t0 = time.time()
    loss, norm = eager_step.eager_train_step(ref.to_layers(), eager_step.sdxl_loss_fn(), micro, None, gradient_clipping=0.0, params=ref.parameters())
    print(f'step in {time.time() - t0:.0f} s: loss {loss.item():.6f} grad norm {norm.item():.6f}', flush=True)
    grads = {}
    for k, m in ref.modules().items():
        for n, p in m.named_parameters():
            if p.grad is not None:
                grads[f'{k}.{n}'] = checksum4(p.grad, f'{k}.{n}')      # [sum |g|, sum g, <g, r>, ||g||_2]
    meta = {'generated_by': 'oracle/make_golden_fullsize.py (oracle/sdxl_ref.py + oracle/eager_step.py; reference call sites models/sdxl.py:591-602,632-651)',
            'seeds': {'weights': WEIGHT_SEED, 'data': DATA_SEED, 'prepare_inputs': PREP_SEED}, 'torch': torch.__version__, 'weight_checksum': wsum,
            'loss': float(loss), 'grad_norm': float(norm), 'parameters_with_grad': len(grads), 'grad_checksums': grads}
    with open(os.path.join(OUT, 'sdxl_fullsize.json'), 'w') as fh:
        json.dump(meta, fh)
    print('wrote', len(grads), 'gradient checksums')


if __name__ == '__main__':
    main()
