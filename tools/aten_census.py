"""Which ATen kernels does one SDXL micro-batch still launch, and from where?  Runs ONE full-size micro-batch (forward + loss + backward, eager, bf16) under a
TorchDispatchMode that counts every ATen op on CUDA tensors together with the innermost frame of this repo that issued it.  The product's own kernels go through
ctypes (invisible here), so what is listed is exactly the glue left to PyTorch: candidates for fusion into existing epilogues.

    python tools/aten_census.py [tiny]      -> JSON lines: {"op", "count", "numel_total", "site"} sorted by count"""
import collections
import json
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SKIP = ('aten.view', 'aten._unsafe_view', 'aten.detach', 'aten.alias', 'aten.t.', 'aten.transpose', 'aten.permute', 'aten.expand', 'aten.slice', 'aten.select', 'aten.unsqueeze',
        'aten.squeeze', 'aten.as_strided', 'aten.reshape', 'aten.split', 'aten.unbind', 'aten.chunk', 'aten.empty', 'aten.new_empty', 'aten.is_', 'aten.sym_', 'aten.stride',
        'aten.size', 'aten.lift_fresh', 'aten._local_scalar_dense', 'aten.record_stream', 'aten.set_', 'aten.unfold.', 'aten.empty_like', 'aten.empty_strided')


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.numel = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            ts = [a for a in list(args) + ([out] if torch.is_tensor(out) else list(out) if isinstance(out, (tuple, list)) else []) if torch.is_tensor(a)]
            if any(t.is_cuda for t in ts):
                site = '?'
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if ROOT in fr.filename and 'aten_census' not in fr.filename:
                        site = f'{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}'
                        break
                self.rows[(name, site)] += 1
                self.numel[(name, site)] += max((t.numel() for t in ts), default=0)
        return out


def main():
    from diffusion_pipe_amd import ops
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import sdxl
    tiny = len(sys.argv) > 1 and sys.argv[1] == 'tiny'
    dev = torch.device('cuda:0')
    cfg = sdxl.tiny_config() if tiny else sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    layers = work.to_layers()
    loss_fn = work.get_loss_fn()
    torch.manual_seed(0)
    feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=1, latent_hw=32 if tiny else 128, seed=1))
    (f, l), = split_batch((feats, label), 1)
    f, l = tuple(t.to(dev) for t in f), tuple(t.to(dev) for t in l)
    ops.FUSE_GRAD_ACCUM = True

    def step():
        x = f
        for layer in layers:
            x = layer(x)
        loss_fn(x, l).backward()
    step()                         # creates the .grad buffers: the counted pass accumulates into them like every micro-batch of the graph path
    torch.cuda.synchronize()
    with Census() as c:
        step()
    torch.cuda.synchronize()
    total = sum(c.rows.values())
    for (name, site), n in c.rows.most_common(60):
        print(json.dumps({'op': name, 'count': n, 'numel_total': c.numel[(name, site)], 'site': site}))
    print(json.dumps({'total_aten_ops_on_cuda_tensors': total, 'distinct': len(c.rows)}))


if __name__ == '__main__':
    main()
