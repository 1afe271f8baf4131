"""Per-parameter-family comparison of two sets of gradient checksum rows (oracle/checksums.py) -- shared by tools/parity_probe.py and bench.py's
DPIPE_BENCH_PARITY_DETAIL leg.  Test infrastructure."""
import re


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



def family_table(gpu, ref, out=print, top=25):
    """gpu / ref: {parameter name: [sum |g|, sum g, <g, r>, ||g||_2]}.  Prints each family's share of the squared norm, norm ratio, contribution to the difference of
    the squared global norm and projection-based L2 error estimate, then the parameters contributing most.  -> (table rows, top parameter rows)"""
    from oracle.checksums import relative_errors
    fam, per = {}, []
    for n, r in ref.items():
        g = gpu.get(n)
        if g is None:
            continue
        f = fam.setdefault(family(n), [0.0, 0.0, 0.0, 0])
        f[0] += g[3] ** 2; f[1] += r[3] ** 2; f[2] += 12.0 * (g[2] - r[2]) ** 2; f[3] += 1
        per.append((g[3] ** 2 - r[3] ** 2, n, g[3], r[3], relative_errors(g, r)))
    tot_g, tot_r = sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values())
    out(f'sqrt(sum of per-parameter norms^2): gpu {tot_g ** 0.5:.6f} reference {tot_r ** 0.5:.6f}; parameters missing on the gpu side: {len([n for n in ref if n not in gpu])}')
    out(f'{"family":36s} {"n":>5s} {"share of |g|^2":>14s} {"norm gpu/ref - 1":>17s} {"d(norm^2) / |g|^2":>18s} {"L2 err estimate":>16s}')
    table = []
    for k, (a, b, e, c) in sorted(fam.items(), key=lambda kv: -abs(kv[1][0] - kv[1][1])):
        row = {'family': k, 'tensors': c, 'share': b / tot_r, 'norm_ratio_minus_1': (a / b) ** 0.5 - 1, 'dnorm2_over_total': (a - b) / tot_r, 'l2_err_estimate': (e / b) ** 0.5}
        table.append(row)
        out(f'{k:36s} {c:5d} {row["share"]:14.4f} {row["norm_ratio_minus_1"]:17.5f} {row["dnorm2_over_total"]:18.6f} {row["l2_err_estimate"]:16.4f}')
    per.sort(key=lambda t: -abs(t[0]))
    out('largest per-parameter contributions to the norm^2 difference:')
    for d, n, g, r, e in per[:top]:
        out(f'  {d / tot_r:+.6f}  {n:95s} norm gpu {g:.5f} ref {r:.5f}  errs abs {e[0]:.4f} signed {e[1]:.4f} proj {e[2]:.3f} l2 {e[3]:.4f}')
    return table, [{'name': n, 'dnorm2_over_total': d / tot_r, 'norm_gpu': g, 'norm_ref': r} for d, n, g, r, _ in per[:40]]


def record_fused_rows(engine, names):
    """Replace the fused step end's update by a recorder: -> dict that receives {name: checksum row of the fp32 lane sum} at the next train_batch (the weights stay
    untouched; the lanes' accumulators are zeroed so later steps start clean)."""
    import torch
    from oracle.checksums import checksum4
    seen = {}

    def fake_update(lane_grads=None, total_sumsq=None, max_norm=0.0, zero_grads=True):
        lanes = lane_grads if lane_grads is not None else [{pid: p.grad for pid, p in byid.items() if p.grad is not None}]
        for pid in byid:
            gs = [lg[pid] for lg in lanes if pid in lg]
            if gs:
                tot = gs[0].float().clone()
                for g_ in gs[1:]:
                    tot += g_.float()
                seen[names[pid]] = checksum4(tot, names[pid])
                if zero_grads:
                    for g_ in gs:
                        g_.zero_()
    byid = {id(p): p for group in engine.optimizer.param_groups for p in group['params']}
    engine.optimizer.fused_update = fake_update
    return seen
