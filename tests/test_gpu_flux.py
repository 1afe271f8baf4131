"""GPU: Flux (BASELINE config 3) -- the HIP-kernel double / single stream blocks, the reference's layer wrappers and
prepare_inputs driven by the engine, against oracle/flux_ref.py (diffusers' FluxTransformer2DModel restated; diffusers is
absent from the image -> parity unpinned) on identical weights and micro-batches.

Tolerances: exact-fp32 kernel mode 1e-3 relative on outputs / loss / global gradient norm (north_star's bound) and 5e-3 of each
gradient's max per parameter; bf16 training mode 4e-2 against the fp32 oracle."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


def _pair(cfg, dtype, gpu, seed=1):
    from diffusion_pipe_amd.workloads import flux
    from oracle import flux_ref
    work = flux.FluxWorkload(cfg, dtype=torch.float32, seed=seed)
    ref = flux_ref.FluxRef(cfg, seed=seed + 1)
    ref.transformer.load_state_dict(work.transformer.state_dict())
    work.transformer.to(gpu, dtype)
    return work, ref


def _run_layers(layers, feats):
    x = feats
    for layer in layers:
        x = layer(x)
    return x


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize('case', ['tiny', 'head128'])
def test_flux_forward_backward_matches_oracle(gpu, dtype, tol, case):
    from diffusion_pipe_amd.workloads import flux
    cfg = flux.tiny_flux_config()
    if case == 'head128':            # Flux.1-dev's head geometry: head_dim 128, rotary axes (16, 56, 56)
        cfg = flux.FluxConfig(in_channels=16, num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=2,
                              joint_attention_dim=64, pooled_projection_dim=32)
    work, ref = _pair(cfg, dtype, gpu)
    torch.manual_seed(5)
    feats, (target, _) = work.prepare_inputs(flux.synthetic_flux_batch(cfg, batch_size=2, latent_hw=(16, 24), text_tokens=19, seed=3))
    want = _run_layers(ref.to_layers(), tuple(f.clone() for f in feats))
    want_loss = ((want - target) ** 2).mean()
    want_loss.backward()
    got = _run_layers(work.to_layers(), tuple(f.to(gpu) for f in feats))
    loss = work.get_loss_fn()(got, (target.to(gpu), torch.tensor([], device=gpu)))
    loss.backward()
    torch.cuda.synchronize()
    assert got.shape == want.shape == (2, 8 * 12, cfg.in_channels)
    assert _rel(got, want) < tol
    assert abs(loss.item() - want_loss.item()) / want_loss.item() < tol
    ref_grads = {n: p.grad for n, p in ref.transformer.named_parameters()}
    worst = max(_rel(p.grad, ref_grads[n]) for n, p in work.transformer.named_parameters())
    assert worst < (5e-3 if dtype == torch.float32 else 8e-2), worst


@pytest.mark.parametrize('dtype,tol,lora', [(torch.float32, 1e-3, False), (torch.bfloat16, 4e-2, False), (torch.float32, 1e-3, True)])
def test_flux_train_batch_matches_oracle(gpu, dtype, tol, lora):
    """engine.train_batch over to_layers() (hipGraph path, 2 lanes) vs the oracle's sequential step: mean loss, pre-clip norm.
    `lora`: the BASELINE config-3 mode -- adapters on every Linear of the double / single blocks, base weights frozen."""
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import flux
    from oracle import eager_step, lora_ref
    cfg = flux.tiny_flux_config()
    gas = 4
    work, ref = _pair(cfg, torch.float32, 'cpu', seed=4)
    if lora:
        names = work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 4, 'dtype': torch.float32})
        blocks = lambda name, module: name.startswith(('transformer_blocks.', 'single_transformer_blocks.'))
        assert lora_ref.apply_lora_ref(ref.transformer, 4, 4, target=blocks) == names
        gen = torch.Generator().manual_seed(9)
        for n, p in work.transformer.named_parameters():
            if '.lora_B.' in n:
                p.data.normal_(0, 0.05, generator=gen)
        ref.transformer.load_state_dict(work.transformer.state_dict())
    work.transformer.to(gpu, dtype)
    torch.manual_seed(5)
    micro = split_batch(work.prepare_inputs(flux.synthetic_flux_batch(cfg, batch_size=gas, latent_hw=(16, 16), text_tokens=12, seed=7)), gas)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(),
                                  dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas,
                                                         'gradient_clipping': 1.0, 'hip_graph': True, 'graph_lanes': 2}, device=gpu)
    params = [p for p in module.parameters() if p.requires_grad]
    assert len(params) == (2 * len(names) if lora else len(list(work.transformer.parameters())))
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), params)
    ref_params = [p for p in ref.parameters() if p.requires_grad]
    want_loss, want_norm = eager_step.eager_train_step(ref.to_layers(), eager_step.default_loss_fn(), copy.deepcopy(micro), None,
                                                       gradient_clipping=1.0, params=ref_params)
    for _ in range(2):                                   # second step replays the captured graphs (lr = 0: same weights)
        loss = engine.train_batch(iter(copy.deepcopy(micro))).item()
        norm = engine.get_global_grad_norm().item()
        assert abs(loss - want_loss.item()) / want_loss.item() < tol, (loss, want_loss.item())
        assert abs(norm - want_norm.item()) / want_norm.item() < 1.5 * tol, (norm, want_norm.item())
