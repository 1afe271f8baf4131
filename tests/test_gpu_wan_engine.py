"""GPU: the engine driving a second model family end to end -- Wan t2v (DiT, flow matching: BASELINE config 4 scaled down,
head_dim 64 kept) through to_layers() / prepare_inputs() / get_loss_fn(), hipGraph path with 2 lanes -- against the oracle's
sequential fp32 composition (oracle/blocks_ref.wan_forward, whose block arithmetic is pinned by the reference's own vectors)
on identical weights and micro-batches: loss and global gradient norm (north_star: 1e-3 relative in fp32)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
def test_wan_train_batch_matches_oracle(gpu, dtype, tol):
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import wan
    from oracle import blocks_ref as br, eager_step
    cfg = wan.tiny_wan_config()
    gas = 4
    work = wan.WanWorkload(cfg, dtype=torch.float32, seed=3)
    ref_p = {n: p.detach().clone().requires_grad_(True) for n, p in work.transformer.named_parameters()}
    work.transformer.to(gpu, dtype)
    torch.manual_seed(5)
    feats, label = work.prepare_inputs(wan.synthetic_wan_batch(cfg, batch_size=gas, seed=7))
    micro = split_batch((feats, label), gas)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='uniform', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                         'hip_graph': dtype == torch.bfloat16, 'graph_lanes': 2}, device=gpu)
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), [p for p in module.parameters()])
    loss = engine.train_batch(iter(micro)).item()
    norm = engine.get_global_grad_norm().item()
    # oracle: sequential fp32 step, loss / GAS per micro-batch, reference clip_grad_norm_
    loss_fn = eager_step.default_loss_fn()
    total = 0.0
    for (f, lab) in micro:
        x_t, _, t, te, sl, _ = f
        out = br.wan_forward(ref_p, cfg, x_t, t, te, sl)
        l = loss_fn(out, lab)
        (l / gas).backward()
        total += l.item()
    want_loss = total / gas
    want_norm = eager_step.clip_grad_norm_(list(ref_p.values()), 1.0).item()
    assert abs(loss - want_loss) / abs(want_loss) < tol, (loss, want_loss)
    assert abs(norm - want_norm) / want_norm < tol * 1.5, (norm, want_norm)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_wan_model_matches_reference_whole_model_vectors(gpu, dtype, tol):
    """The HIP-kernel Wan model (to_layers() = Initial + blocks + Final, default loss) against vectors from the reference's own
    WanModel driven through the reference's own pipeline layers, prepare_inputs and loss (oracle/make_golden_wan_model.py):
    output, loss, every parameter gradient."""
    import os
    from safetensors.torch import load_file
    from diffusion_pipe_amd.workloads import wan
    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'wan_model_fp32.safetensors'))
    cfg = wan.tiny_wan_config()
    work = wan.WanWorkload(cfg, dtype=torch.float32)
    work.transformer.load_state_dict({k[len('param.'):]: v for k, v in g.items() if k.startswith('param.')}, strict=True)
    work.transformer.to(gpu, dtype)
    work.prepare_inputs({'latents': g['in.latents'], 'mask': None, 'text_embeddings': g['in.text_embeddings'], 'seq_lens': g['in.seq_lens']})   # sets the grid
    dev = lambda t: t.to(gpu)                                                                                                                    # noqa: E731
    x = (dev(g['prep.x_t']), torch.tensor([], device=gpu), dev(g['prep.t']), dev(g['in.text_embeddings']), dev(g['in.seq_lens']), torch.tensor([], device=gpu))
    for layer in work.to_layers():
        x = layer(x)
    loss = work.get_loss_fn()(x, (dev(g['prep.target']), torch.tensor([], device=gpu)))
    loss.backward()
    want_out, want_loss = g['out'], g['loss'].item()
    assert ((x.float().cpu() - want_out).abs().max() / want_out.abs().max()).item() < tol
    assert abs(loss.item() - want_loss) / want_loss < tol
    worst, err2, ref2 = 0.0, 0.0, 0.0
    for n, p in work.transformer.named_parameters():
        want = g[f'grad.{n}']
        assert p.grad is not None, n
        d = p.grad.float().cpu() - want
        worst = max(worst, (d.abs().max() / want.abs().max().clamp_min(1e-9)).item())
        err2, ref2 = err2 + d.double().pow(2).sum().item(), ref2 + want.double().pow(2).sum().item()
    if dtype == torch.float32:
        assert worst < 5e-3, worst                     # every parameter's gradient, max-norm relative
    assert (err2 / ref2) ** 0.5 < tol                  # all gradients as one vector (bf16: rounding noise averages out)
