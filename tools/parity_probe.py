"""Where does the timed path's gradient differ from the oracle's?  Reproduces bench.py's `parity` leg (SDXL full size, 4 lanes, GAS 8, fused AdamW, W training steps
first), then compares the PRE-clip gradient of every parameter -- [sum |g|, sum g, <g, r>, ||g||_2] rows (oracle/checksums.py) -- between the GPU engine and the
oracle's fp32 eager path on the same weights and micro-batch, and prints the contribution of each module family to the difference of the squared global norm.
Test infrastructure (imports oracle/).   python tools/parity_probe.py [train_steps] [out.json]"""
import json
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Rows(torch.optim.Optimizer):
    def __init__(self, params, names):
        super().__init__(params, {})
        self.names, self.rows = names, {}

    def step(self, closure=None):
        from oracle.checksums import checksum4
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    self.rows[self.names[id(p)]] = checksum4(p.grad, self.names[id(p)])


def family(name):
    name = re.sub(r'\.\d+\.', '.N.', name)
    for pat, fam in ((r'attn1\.to_[qk]', 'unet self-attn to_q/to_k'), (r'attn1\.to_v', 'unet self-attn to_v'), (r'attn1\.to_out', 'unet self-attn to_out'),
                     (r'attn2\.to_q', 'unet cross-attn to_q'), (r'attn2\.to_[kv]', 'unet cross-attn to_k/to_v'), (r'attn2\.to_out', 'unet cross-attn to_out'),
                     (r'\.ff\.', 'unet feed-forward'), (r'norm[123]\.', 'unet block LayerNorm'), (r'proj_in|proj_out', 'unet transformer proj_in/out'),
                     (r'resnets.*conv|conv_shortcut|downsamplers|upsamplers|conv_in|conv_out', 'unet convolutions'), (r'resnets.*norm|conv_norm_out|attentions\.N\.norm', 'unet GroupNorm'),
                     (r'time_emb|time_embedding|add_embedding', 'unet time / add embeddings'), (r'text_encoder.*(q_proj|k_proj)', 'CLIP q/k_proj'),
                     (r'text_encoder.*(v_proj|out_proj)', 'CLIP v/out_proj'), (r'text_encoder.*mlp', 'CLIP mlp'), (r'text_encoder.*(layer_norm|final_layer_norm)', 'CLIP LayerNorm'),
                     (r'text_encoder.*embed', 'CLIP embeddings'), (r'text_projection', 'CLIP text_projection')):
        if re.search(pat, name):
            return fam
    return 'other'


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    out_path = sys.argv[2] if len(sys.argv) > 2 else ''
    from diffusion_pipe_amd import optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    from oracle.cpu_baseline import sdxl_cpu_baseline
    from oracle.checksums import relative_errors
    dev = torch.device('cuda', 0)
    gas, lanes = 8, 4
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0, 'steps_per_print': 1 << 30,
                                                         'hip_graph': True, 'graph_lanes': lanes}, device=dev)
    params = [p for p in module.parameters() if p.requires_grad]
    work.train_config = {'optimizer': {'type': 'adamw', 'lr': 1e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
    fused = optim.make_optimizer_factory(work.train_config, work, global_batch_size=gas)
    engine._configure_optimizer(fused, params)
    torch.manual_seed(1234)
    pool = []
    for s in range(3):
        feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=128, seed=100 + s))
        pool.append(split_batch((feats, label), gas))
    sample = pool[0][0]
    for i in range(steps):
        engine.reset_activation_shape()
        engine.train_batch(iter(pool[i % len(pool)]))
    torch.cuda.synchronize()
    names = {id(p): f'{k}.{n}' for k, m in work.modules().items() for n, p in m.named_parameters()}
    if os.environ.get('PROBE_FUSED', '0') == '1':
        # bench.py's own path: the fused step end reads the lanes' bf16 accumulators (sum over lanes in fp32 inside adamw_sumsq / adamw_step).  Record the rows of the
        # fp32 lane sum right before the fused update consumes (and zeroes) the accumulators; the update itself is skipped (weights stay what the oracle gets)
        from oracle.checksums import checksum4
        opt = engine.optimizer
        byid = {id(p): p for p in params}
        seen = {}

        def fake_update(lane_grads=None, total_sumsq=None, max_norm=0.0, zero_grads=True):
            for pid, p in byid.items():
                gs = [lg[pid] for lg in lane_grads if pid in lg]
                if gs:
                    tot = gs[0].float().clone()
                    for g_ in gs[1:]:
                        tot += g_.float()
                    seen[names[pid]] = checksum4(tot, names[pid])
        opt.fused_update = fake_update
        engine.reset_activation_shape()
        loss = engine.train_batch(iter([sample] * gas)).item()
        norm = engine.get_global_grad_norm().item()
        torch.cuda.synchronize()
        gpu = seen
        print(f'fused step end: adamw_sumsq norm {norm:.6f}; sqrt(sum of per-parameter ||fp32 lane sum||^2) {sum(r[3] ** 2 for r in seen.values()) ** 0.5:.6f}')
    else:
        rec = Rows(params, names)
        engine.optimizer = rec
        engine.reset_activation_shape()
        loss = engine.train_batch(iter([sample] * gas)).item()
        norm = engine.get_global_grad_norm().item()
        torch.cuda.synchronize()
        coef = min(1.0, 1.0 / (norm + 1e-6))
        gpu = {k: [v / coef for v in r] for k, r in rec.rows.items()}
    state = {k: {n: v.detach().to('cpu', torch.float32) for n, v in m.state_dict().items()} for k, m in work.modules().items()}
    cpu = sdxl_cpu_baseline(cfg, latent_hw=128, micro_batch=sample, state=state, per_parameter=True)
    ref = cpu['rows']
    print(f'after {steps} training steps: loss gpu {loss:.7f} cpu {cpu["loss"]:.7f}; grad norm gpu {norm:.6f} cpu {cpu["grad_norm"]:.6f} rel {abs(norm - cpu["grad_norm"]) / cpu["grad_norm"]:.3e}')
    fam = {}
    per = []
    for n, r in ref.items():
        g = gpu.get(n)
        if g is None:
            continue
        f = fam.setdefault(family(n), [0.0, 0.0, 0.0, 0])
        f[0] += g[3] ** 2; f[1] += r[3] ** 2; f[2] += 12.0 * (g[2] - r[2]) ** 2; f[3] += 1
        per.append((g[3] ** 2 - r[3] ** 2, n, g[3], r[3], relative_errors(g, r)))
    tot_g, tot_r = sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values())
    print(f'sum of per-parameter norms^2: gpu {tot_g ** 0.5:.6f} cpu {tot_r ** 0.5:.6f}')
    print(f'{"family":36s} {"n":>5s} {"share of |g|^2":>14s} {"norm gpu/cpu - 1":>17s} {"d(norm^2) / |g|^2":>18s} {"L2 err estimate":>16s}')
    table = []
    for k, (a, b, e, c) in sorted(fam.items(), key=lambda kv: -abs(kv[1][0] - kv[1][1])):
        row = {'family': k, 'tensors': c, 'share': b / tot_r, 'norm_ratio_minus_1': (a / b) ** 0.5 - 1, 'dnorm2_over_total': (a - b) / tot_r, 'l2_err_estimate': (e / b) ** 0.5}
        table.append(row)
        print(f'{k:36s} {c:5d} {row["share"]:14.4f} {row["norm_ratio_minus_1"]:17.5f} {row["dnorm2_over_total"]:18.6f} {row["l2_err_estimate"]:16.4f}')
    per.sort(key=lambda t: -abs(t[0]))
    print('largest per-parameter contributions to the norm^2 difference:')
    for d, n, g, r, e in per[:25]:
        print(f'  {d / tot_r:+.6f}  {n:95s} norm gpu {g:.5f} cpu {r:.5f}  errs abs {e[0]:.4f} signed {e[1]:.4f} proj {e[2]:.3f} l2 {e[3]:.4f}')
    if out_path:
        json.dump({'train_steps': steps, 'loss_gpu': loss, 'loss_cpu': cpu['loss'], 'grad_norm_gpu': norm, 'grad_norm_cpu': cpu['grad_norm'], 'families': table,
                   'top': [{'name': n, 'dnorm2_over_total': d / tot_r, 'norm_gpu': g, 'norm_cpu': r} for d, n, g, r, _ in per[:40]]}, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
