"""FULL-DEPTH parity of BASELINE configs 3 / 4 / 5, block by block ("what's missing" 4: the step-level checks of these configurations run 2 + 2 / 2 / 1 + 1
blocks because the CPU oracle cannot hold 59 / 42 / 63 layers; a chained per-block check can).

The product transformer is built at REAL width and FULL depth in bf16 on the GPU (random weights from a seed, as every bench of this repo) and run layer by layer on one
micro-batch of the configuration's real shape.  For every transformer block k the ORACLE block (oracle/flux_ref.py, oracle/blocks_ref.py, oracle/hv_ref.py: fp32 ATen
arithmetic -- evaluated on the same GPU: this is the checker, not the product) receives block k's parameters (as fp32) and THE PRODUCT PATH'S OWN INPUT of block k -- so the
inputs carry whatever the depth does to the activations (growth of the residual stream, outlier channels) -- and both sides run forward + backward against the same seeded
output gradient:

    forward   rel = || y_product - y_oracle ||_F / || y_oracle ||_F        of every tensor the block rewrites
    dgrad     the same for the gradient reaching the block's inputs
    wgrad     | ||g_product|| - ||g_oracle|| | / ||g_oracle|| over ALL of the block's trained parameters, and the worst single tensor's relative Frobenius error

then the chain continues with the PRODUCT's output.  Written as JSON (one row per block + maxima); HunyuanVideo runs at 1/8 of config 5's tokens (the oracle's fp32 score
matrix of 61 456 tokens is 362 GB), Flux and Wan at the full token count of configs 3 / 4.

    python tools/chained_block_parity.py flux|wan|hv [out.json] [--blocks N]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = torch.device('cuda:0')


def _fro(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _to32(t):
    return t.detach().float() if torch.is_tensor(t) and t.is_floating_point() else t


def run_pair(name, product_layer, product_module, oracle_fn, oracle_params, inputs, changed, seed):
    """One block.  product_layer(tuple) -> tuple; oracle_fn(fp32 tuple) -> tuple of the `changed` tensors; oracle_params {name: fp32 leaf} keyed like product_module's
    named_parameters.  -> (row, product output tuple)."""
    pin = tuple((t.detach().requires_grad_(True) if (torch.is_tensor(t) and t.is_floating_point() and i in changed) else t) for i, t in enumerate(inputs))
    for p in product_module.parameters():
        p.grad = None
    pout = product_layer(pin)
    oin = tuple((_to32(t).requires_grad_(True) if (torch.is_tensor(t) and t.is_floating_point() and i in changed) else _to32(t)) for i, t in enumerate(inputs))
    with torch.device(DEV):                     # the oracle's helpers build their index / mask tensors with bare factory calls
        oout = oracle_fn(oin)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    row = {'block': name}
    gs_p, gs_o, outs_p, outs_o = [], [], [], []
    for j, i in enumerate(changed):
        yp, yo = pout[i], oout[j]
        row[f'fwd_rel_{j}'] = _fro(yp, yo)
        row[f'out_rms_{j}'] = float(yo.double().pow(2).mean().sqrt())
        g = torch.randn(yo.shape, device=DEV, generator=gen) * 1e-3
        gs_o.append(g); gs_p.append(g.to(yp.dtype)); outs_p.append(yp); outs_o.append(yo)
    torch.autograd.backward(outs_p, gs_p)
    torch.autograd.backward(outs_o, [g.to(torch.bfloat16).float() for g in gs_o])          # the very gradient values the product received
    for j, i in enumerate(changed):
        row[f'dgrad_rel_{j}'] = _fro(pin[i].grad, oin[i].grad)
    pg = {n: p.grad for n, p in product_module.named_parameters() if p.requires_grad}
    sq_p = sq_o = 0.0
    worst, worst_name = 0.0, None
    for n, g in pg.items():
        if g is None or n not in oracle_params or oracle_params[n].grad is None:
            continue
        go = oracle_params[n].grad
        sq_p += float(g.double().pow(2).sum()); sq_o += float(go.double().pow(2).sum())
        e = _fro(g, go)
        if e > worst:
            worst, worst_name = e, n
    row['wgrad_norm_rel'] = abs(sq_p ** 0.5 - sq_o ** 0.5) / max(sq_o ** 0.5, 1e-30)
    row['wgrad_worst_tensor_rel'], row['wgrad_worst_tensor'] = worst, worst_name
    row['finite'] = all(bool(torch.isfinite(pout[i]).all()) for i in changed)
    out = tuple(t.detach() if torch.is_tensor(t) else t for t in pout)
    for p in product_module.parameters():
        p.grad = None
    return row, out


def oracle_copy(product_block, oracle_block):
    """oracle block (fp32, GPU) with the product block's parameter VALUES; -> {name: leaf}"""
    oracle_block.to(DEV, torch.float32)
    oracle_block.load_state_dict({k: v.detach().float() for k, v in product_block.state_dict().items()})
    return dict(oracle_block.named_parameters())


# ------------------------------------------------------------------------------------------------------------------ the three families
def flux_chain(max_blocks):
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import flux
    from oracle import flux_ref
    cfg = flux.FluxConfig()
    work = flux.FluxWorkload(cfg, model_config={'guidance': 1.0}, dtype=torch.bfloat16, seed=51, device=DEV)
    torch.manual_seed(54)
    feats, _ = split_batch(work.prepare_inputs(flux.synthetic_flux_batch(cfg, batch_size=1, latent_hw=(128, 128), text_tokens=512, seed=55)), 1)[0]
    layers = work.to_layers()
    with torch.no_grad():
        x = layers[0](tuple(t.to(DEV) for t in feats))
    dim = cfg.num_attention_heads * cfg.attention_head_dim
    blocks = [(f'double.{i}', layers[1 + i], flux_ref.FluxTransformerBlock) for i in range(cfg.num_layers)] + \
             [(f'single.{i}', layers[1 + cfg.num_layers + i], flux_ref.FluxSingleTransformerBlock) for i in range(cfg.num_single_layers)]
    what = f'Flux.1-dev (config 3): {cfg.num_layers} double + {cfg.num_single_layers} single blocks, dim {dim}, {x[0].shape[1]} image + {x[1].shape[1]} text tokens'
    for name, layer, ocls in blocks[:max_blocks]:
        ob = ocls(dim, cfg.num_attention_heads, cfg.attention_head_dim)
        params = oracle_copy(layer.block, ob)
        wrap = flux_ref.BlockWrapper(ob)
        yield what, name, layer, layer.block, (lambda t, w=wrap: w(t)[:2]), params, x, (0, 1)
        x = yield


def wan_chain(max_blocks):
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import wan
    from oracle import blocks_ref as br
    cfg = wan.WanConfig()
    work = wan.WanWorkload(cfg, dtype=torch.bfloat16, seed=61, device=DEV)
    torch.manual_seed(64)
    feats, _ = split_batch(work.prepare_inputs(wan.synthetic_wan_batch(cfg, batch_size=1, frames=9, latent_hw=(64, 64), text_tokens=512, seed=65)), 1)[0]
    layers = work.to_layers()
    with torch.no_grad():
        x = layers[0](tuple(t.to(DEV) for t in feats))
    what = f'Wan2.1-14B (config 4): {cfg.num_layers} blocks, dim {cfg.dim}, {x[0].shape[1]} video + {x[7].shape[1]} text tokens'
    for i in range(min(cfg.num_layers, max_blocks)):
        layer = layers[1 + i]
        params = {n: p.detach().float().requires_grad_(True) for n, p in layer.block.named_parameters()}

        def ofn(t, params=params):
            xx, e, e0, seq_lens, grid_sizes, cos, sin, context = t
            return (br.wan_block(params, xx, e0, context, cfg.num_heads, cos, sin, cfg.eps),)
        yield what, f'block.{i}', layer, layer.block, ofn, params, x, (0,)
        x = yield


def hv_chain(max_blocks):
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import hunyuan_video as hv
    from oracle import hv_ref
    cfg = hv.HunyuanVideoConfig()
    work = hv.HunyuanVideoWorkload(cfg, model_config={'guidance': 1.0}, dtype=torch.bfloat16, seed=72, device=DEV)
    torch.manual_seed(73)
    feats, _ = split_batch(work.prepare_inputs(hv.synthetic_hv_batch(cfg, batch_size=1, latent_thw=(17, 32, 60), text_tokens=256, valid_text=(190,), seed=74)), 1)[0]
    layers = work.to_layers()
    with torch.no_grad():
        x = layers[0](tuple(t.to(DEV) for t in feats))
    h, heads = cfg.hidden_size, cfg.heads_num
    nd, ns = cfg.mm_double_blocks_depth, cfg.mm_single_blocks_depth
    what = f'HunyuanVideo (config 5 at 1/8 of its tokens): {nd} double + {ns} single blocks, dim {h}, {x[0].shape[1]} video + {x[1].shape[1]} text tokens'
    done = 0
    for i in range(nd):
        if done >= max_blocks:
            return
        layer = layers[1 + i]
        ob = hv_ref.MMDoubleStreamBlock(h, heads, cfg.mlp_width_ratio)
        params = oracle_copy(layer.block, ob)

        def ofn(t, ob=ob):
            img, txt, vec, cu, mx, fc, fs = t[:7]
            return ob(img, txt, vec, cu, cu, mx, mx, (fc, fs))
        yield what, f'double.{i}', layer, layer.block, ofn, params, x, (0, 1)
        x = yield
        done += 1
    x = layers[1 + nd](x)                       # concatenate_hidden_states
    for i in range(ns):
        if done >= max_blocks:
            return
        layer = layers[2 + nd + i]
        ob = hv_ref.MMSingleStreamBlock(h, heads, cfg.mlp_width_ratio)
        params = oracle_copy(layer.block, ob)

        def ofn(t, ob=ob):
            xx, vec, cu, mx, fc, fs = t[:6]
            return (ob(xx, vec, xx.shape[1] - fc.shape[0], cu, cu, mx, mx, (fc, fs)),)
        yield what, f'single.{i}', layer, layer.block, ofn, params, x, (0,)
        x = yield
        done += 1


def main():
    which = sys.argv[1]
    args = [a for a in sys.argv[2:] if not a.startswith('--')]
    out_path = args[0] if args else ''
    max_blocks = int(sys.argv[sys.argv.index('--blocks') + 1]) if '--blocks' in sys.argv else 10 ** 6
    chain = {'flux': flux_chain, 'wan': wan_chain, 'hv': hv_chain}[which](max_blocks)
    rows, what, t0 = [], None, time.perf_counter()
    try:
        item = next(chain)
        while True:
            what, name, layer, module, ofn, params, x, changed = item
            row, x = run_pair(name, layer, module, ofn, params, x, changed, seed=1000 + len(rows))
            rows.append(row)
            print(json.dumps(row), flush=True)
            del params, ofn, item
            torch.cuda.empty_cache()
            next(chain)
            item = chain.send(x)
    except StopIteration:
        pass
    keys = sorted({k for r in rows for k in r if k.startswith(('fwd_rel', 'dgrad_rel', 'wgrad_'))} - {'wgrad_worst_tensor'})
    summary = {'workload': what, 'blocks': len(rows), 'seconds': round(time.perf_counter() - t0, 1), 'all_finite': all(r['finite'] for r in rows),
               'max': {k: max(r[k] for r in rows if k in r) for k in keys}, 'median': {k: sorted(r[k] for r in rows if k in r)[len([r for r in rows if k in r]) // 2] for k in keys},
               'out_rms_first_last': [rows[0].get('out_rms_0'), rows[-1].get('out_rms_0')] if rows else None}
    print(json.dumps(summary), flush=True)
    if out_path:
        json.dump({'summary': summary, 'rows': rows}, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
