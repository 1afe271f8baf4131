"""Guard-band / poison probe (round 6, root-causing DESIGN.md section 2's graph-replay corruption): does any kernel of this repo WRITE outside its output buffer, or READ
memory it never wrote?  A pure-PyTorch repro of the suspected ATen reduction is clean in every graph arrangement (tools/graph_reduce_repro.py), and the round-5 corruption
appeared with ONE lane too, under replay only -- the signature of a deterministic out-of-bounds write (or an uninitialised read) whose victim depends on the memory layout:
harmless wherever the eager allocator happens to put things, fatal at the fixed addresses of a graph's private pool.

While the probe is active every `torch.empty / empty_like / new_empty` on the GPU (= every output and workspace buffer ops.py hands to a kernel) is carved out of a larger
allocation with 4 KiB guard bands of 0xA5 on both sides and a payload pre-filled with the bit pattern 0x7F7F... (3.4e38 as bf16 or fp32: any read of a never-written
element turns the step non-finite).  One full-size micro-batch (batch = `stack`: the stacked form that went bad, or 1) runs eagerly through the product's own layers,
loss and backward; then every guard band is checked and the loss / every gradient is checked for finiteness.

    [DPIPE_PRECISE_ADDENDS=0 DPIPE_DEBUG_ATEN_TEMB_ADD=1] python tools/oob_guard_probe.py [stack=4] [out.json]"""
import json
import math
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = 4096
REAL_EMPTY, REAL_EMPTY_LIKE, REAL_NEW_EMPTY = torch.empty, torch.empty_like, torch.Tensor.new_empty
REG = []
ACTIVE = [False]


def _site():
    for fr in reversed(traceback.extract_stack()[:-3]):
        if 'diffusion_pipe_amd' in fr.filename and 'oob_guard_probe' not in fr.filename:
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}'
    return '?'


def _carve(nbytes, device):
    pad = (-nbytes) % 256
    raw = REAL_EMPTY(nbytes + pad + 2 * G, dtype=torch.uint8, device=device)
    raw[:G].fill_(0xA5)
    raw[G + nbytes:].fill_(0xA5)
    raw[G:G + nbytes].fill_(0x7F)
    REG.append((raw, nbytes, _site()))
    return raw[G:G + nbytes]


def _is_cuda(device):
    return device is not None and torch.device(device).type == 'cuda'


def guarded_empty(*size, **kw):
    device = kw.get('device')
    if not ACTIVE[0] or not _is_cuda(device) or kw.get('pin_memory') or kw.get('out') is not None:
        return REAL_EMPTY(*size, **kw)
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    size = tuple(int(s) for s in size)
    dtype = kw.get('dtype') or torch.get_default_dtype()
    n = math.prod(size) if size else 1
    if n == 0:
        return REAL_EMPTY(*size, **kw)
    flat = _carve(n * torch.empty((), dtype=dtype).element_size(), device).view(dtype)
    if kw.get('memory_format') == torch.channels_last and len(size) == 4:
        t = flat.view(size[0], size[2], size[3], size[1]).permute(0, 3, 1, 2)
    else:
        t = flat.view(size)
    return t.requires_grad_() if kw.get('requires_grad') else t


def guarded_empty_like(x, **kw):
    if not ACTIVE[0] or not x.is_cuda or 'device' in kw or 'layout' in kw:
        return REAL_EMPTY_LIKE(x, **kw)
    dtype = kw.get('dtype') or x.dtype
    flat = _carve(x.numel() * torch.empty((), dtype=dtype).element_size(), x.device).view(dtype) if x.numel() else None
    if flat is None:
        return REAL_EMPTY_LIKE(x, **kw)
    mf = kw.get('memory_format', torch.preserve_format)
    dense = x.dim() > 0 and x.numel() == sum((s - 1) * st for s, st in zip(x.shape, x.stride())) + 1
    if mf == torch.preserve_format and dense and x.dim() > 0 and not x.is_contiguous():
        return flat.as_strided(x.shape, x.stride())
    if mf == torch.channels_last and x.dim() == 4:
        return flat.view(x.shape[0], x.shape[2], x.shape[3], x.shape[1]).permute(0, 3, 1, 2)
    return flat.view(x.shape)


def guarded_new_empty(self, *size, **kw):
    if not ACTIVE[0] or not (self.is_cuda if 'device' not in kw else _is_cuda(kw['device'])):
        return REAL_NEW_EMPTY(self, *size, **kw)
    kw.setdefault('dtype', self.dtype)
    kw.setdefault('device', self.device)
    return guarded_empty(*size, **kw)


def check_guards():
    bad = []
    for raw, nbytes, site in REG:
        head_ok = bool((raw[:G] == 0xA5).all())
        tail_ok = bool((raw[G + nbytes:] == 0xA5).all())
        if not (head_ok and tail_ok):
            tail = raw[G + nbytes:]
            first = int((tail != 0xA5).nonzero()[0]) if not tail_ok else None
            bad.append({'site': site, 'payload_bytes': nbytes, 'head_intact': head_ok, 'tail_intact': tail_ok, 'first_bad_tail_byte': first,
                        'bad_tail_bytes': int((tail != 0xA5).sum()) if not tail_ok else 0, 'bad_head_bytes': int((raw[:G] != 0xA5).sum()) if not head_ok else 0})
    return bad


def main():
    stack = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    out_path = sys.argv[2] if len(sys.argv) > 2 else ''
    from diffusion_pipe_amd import ops
    from diffusion_pipe_amd.workloads import sdxl
    dev = torch.device('cuda:0')
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    layers, loss_fn = work.to_layers(), work.get_loss_fn()
    params = [p for m in work.modules().values() for p in m.parameters()]
    names = {id(p): f'{k}.{n}' for k, m in work.modules().items() for n, p in m.named_parameters()}
    torch.manual_seed(1234)
    from diffusion_pipe_amd.data import split_batch
    feats, label = split_batch(work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=stack, latent_hw=128, seed=100)), 1)[0]      # (None mask -> empty tensor, as the engine receives it)
    report = {'stack': stack, 'precise_addends': ops.PRECISE_ADDENDS, 'aten_temb_add': sdxl._DEBUG_ATEN_TEMB_ADD, 'runs': []}
    torch.empty, torch.empty_like, torch.Tensor.new_empty = guarded_empty, guarded_empty_like, guarded_new_empty
    try:
        for run in range(2):                      # run 0: cold (the lazily created workspaces are guarded too); run 1: warm caches
            ACTIVE[0] = True
            for p in params:
                p.grad = None
            x = tuple(t.to(dev) for t in feats)
            for layer in layers:
                x = layer(x)
            loss = loss_fn(x, tuple(t.to(dev) for t in label))
            loss.backward()
            torch.cuda.synchronize()
            ACTIVE[0] = False
            bad = check_guards()
            nonfinite = [names[id(p)] for p in params if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            row = {'run': run, 'loss': float(loss), 'loss_finite': bool(torch.isfinite(loss)), 'guarded_allocations': len(REG),
                   'guarded_gib': round(sum(r[0].numel() for r in REG) / 2 ** 30, 2), 'guard_violations': bad[:40], 'n_guard_violations': len(bad),
                   'non_finite_gradients': nonfinite[:20], 'n_non_finite_gradients': len(nonfinite)}
            report['runs'].append(row)
            print(json.dumps({k: v for k, v in row.items() if k != 'guard_violations'}), flush=True)
            for b in bad[:40]:
                print('   VIOLATION', json.dumps(b), flush=True)
            del x, loss
            REG.clear()
            torch.cuda.empty_cache()
    finally:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = REAL_EMPTY, REAL_EMPTY_LIKE, REAL_NEW_EMPTY
    if out_path:
        json.dump(report, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
