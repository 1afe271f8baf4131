"""Where along the network does the bf16 path's signal drift in SCALE?  (diagnostics; run on the GPU box)

The same full-size SDXL micro-batch runs twice through the product's layers -- once in the exact-fp32 kernel mode (known to match the oracle to 1e-4 on loss and
gradient norm: tests/test_gpu_fullsize.py), once in the timed bf16 mode -- on identical (bf16-representable) weights.  Forward hooks on the resnets, transformer
blocks, attentions, feed-forwards and CLIP layers record every module's output and (through a tensor hook) the gradient that arrives at that output; for each the
projection coefficient  c = <bf16, fp32> / <fp32, fp32> - 1  and the relative L2 error are printed in execution order.  Unbiased rounding noise leaves c at
+-1e-5; a systematic shrink shows as c drifting negative along the backward pass.      python tools/scale_probe.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def first_float(x):
    if torch.is_tensor(x):
        return x if x.is_floating_point() else None
    if isinstance(x, (tuple, list)):
        for t in x:
            f = first_float(t)
            if f is not None:
                return f
    return None


def run(work, feats, label, classes, dev):
    rec, order = {}, []

    def hook(name):
        def fn(mod, inp, out):
            t = first_float(out)
            if t is None:
                return
            key = name if name not in rec else f'{name}#{sum(1 for k in rec if k.startswith(name))}'
            rec[key] = {'out': t.detach().float().clone()}
            order.append(key)
            if t.requires_grad:
                t.register_hook(lambda g, key=key: rec[key].__setitem__('grad', g.detach().float().clone()))
        return fn
    handles = []
    for k, m in work.modules().items():
        for n, sub in m.named_modules():
            if isinstance(sub, classes):
                handles.append(sub.register_forward_hook(hook(f'{k}.{n}')))
    x = tuple(t.to(dev) for t in feats)
    for layer in work.to_layers():
        x = layer(x)
    loss = work.get_loss_fn()(x, tuple(t.to(dev) for t in label))
    loss.backward()
    torch.cuda.synchronize()
    for h in handles:
        h.remove()
    grads = {f'{k}.{n}': p.grad.detach().float().clone() for k, m in work.modules().items() for n, p in m.named_parameters() if p.grad is not None}
    return float(loss), rec, order, grads


def coef(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    den = float((b * b).sum())
    if den == 0.0:
        return 0.0, 0.0
    return float((a * b).sum()) / den - 1.0, float((a - b).norm()) / den ** 0.5


def main():
    from diffusion_pipe_amd import nn as dnn
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import sdxl
    dev = torch.device('cuda', 0)
    cfg = sdxl.SDXLConfig()
    classes = (sdxl.ResnetBlock2D, sdxl.BasicTransformerBlock, sdxl.Transformer2DModel, sdxl.CLIPEncoderLayer, dnn.Attention, dnn.FeedForward, sdxl.Downsample2D,
               sdxl.Upsample2D)
    w16 = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    torch.manual_seed(1234)
    feats, label = w16.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=1, latent_hw=128, seed=100))
    (feats, label), = split_batch((feats, label), 1)
    dc_only = os.environ.get('SCALE_PROBE_DC_ONLY', '0') == '1'
    l32 = l16 = wavg = float('nan')
    rows, r16, r32, order, g16, g32 = [], {}, {}, [], {}, {}
    if not dc_only:
        w32 = sdxl.SDXLWorkload(cfg, dtype=torch.float32, seed=0, device=dev)
        for k, m in w32.modules().items():
            m.load_state_dict({n: v.float() for n, v in w16.modules()[k].state_dict().items()})
        l32, r32, order, g32 = run(w32, feats, label, classes, dev)
        del w32
        torch.cuda.empty_cache()
        l16, r16, _, g16 = run(w16, feats, label, classes, dev)
        print(f'loss fp32 {l32:.7f} bf16 {l16:.7f}')
    print(f'{"module (forward order)":78s} {"out c-1":>10s} {"out err":>9s} {"grad c-1":>10s} {"grad err":>9s}')
    for key in order:
        a, b = r16.get(key), r32.get(key)
        if a is None or b is None or a['out'].shape != b['out'].shape:
            continue
        co, eo = coef(a['out'], b['out'])
        cg, eg = coef(a['grad'], b['grad']) if ('grad' in a and 'grad' in b) else (float('nan'), float('nan'))
        rows.append({'module': key, 'out_scale_minus_1': co, 'out_rel_err': eo, 'grad_scale_minus_1': cg, 'grad_rel_err': eg})
        print(f'{key[:78]:78s} {co:10.5f} {eo:9.5f} {cg:10.5f} {eg:9.5f}')
    fam = {}
    for n in g32:
        if n in g16:
            c, e = coef(g16[n], g32[n])
            fam[n] = (c, e, float(g32[n].norm()))
    tot = sum(v[2] ** 2 for v in fam.values()) or 1.0
    worst = sorted(fam.items(), key=lambda kv: kv[1][0] * kv[1][2] ** 2)[:15]
    print('parameter gradients with the largest norm-weighted negative scale:')
    for n, (c, e, nn_) in worst:
        print(f'  {n:90s} c-1 {c:9.5f} err {e:8.5f} share of |g|^2 {nn_ ** 2 / tot:.5f}')
    if fam:
        wavg = sum(v[0] * v[2] ** 2 for v in fam.values()) / tot
        print(f'norm^2-weighted mean scale - 1 over all parameter gradients: {wavg:.6f}')
    # ---- how strongly does the gradient norm react to the MEAN of the residual?  The same bf16 step with the target shifted by a constant of +-1e-3 rms(out - target)
    # (0.1 % of the residual's rms, all in its DC component): if the network's response to the mean residual dominates the gradient, the rounding noise of the bf16
    # forward -- whose DC component is of this order -- moves the global gradient norm by the same relative amount, with either sign, in every parameter family at once.
    def norm_for(shift):
        for m in w16.modules().values():
            for p_ in m.parameters():
                p_.grad = None
        x = tuple(t.to(dev) for t in feats)
        for layer in w16.to_layers():
            x = layer(x)
        out = x[0] if isinstance(x, tuple) else x
        lab = tuple(t.to(dev) for t in label)
        rms = float((out.detach().float() - lab[0].float()).pow(2).mean().sqrt())
        lab = (lab[0] + shift * rms,) + tuple(lab[1:])
        w16.get_loss_fn()(x, lab).backward()
        torch.cuda.synchronize()
        return float(sum(float(p_.grad.float().pow(2).sum()) for m in w16.modules().values() for p_ in m.parameters() if p_.grad is not None) ** 0.5), rms
    n0, rms = norm_for(0.0)
    dc = {}
    for sh in (1e-3, -1e-3, 1e-2, -1e-2):
        n1, _ = norm_for(sh)
        dc[str(sh)] = n1 / n0 - 1.0
        print(f'target shifted by {sh:+.0e} x rms(out - target) = {sh * rms:+.3e}: global gradient norm changes by {n1 / n0 - 1.0:+.5f}')
    if len(sys.argv) > 1:
        json.dump({'loss_fp32': l32, 'loss_bf16': l16, 'modules': rows, 'weighted_param_grad_scale_minus_1': wavg, 'grad_norm_response_to_target_dc_shift': dc},
                  open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
