"""LoRA training through the HIP path (SURVEY.md 8(d) configs 1 and 3 train adapters, not full weights) vs the oracle's
peft restatement (oracle/lora_ref.py; peft itself is absent from the image -> parity unpinned).

Tolerances: fp32 exact-kernel mode 1e-3 relative (north_star's bound) on outputs / loss / global grad norm, 5e-3 of each
gradient's max on per-parameter gradients; bf16 mode 3e-2 against the fp32 oracle (8 mantissa bits)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize('rank,alpha,with_res', [(4, 4.0, False), (32, 32.0, True), (8, 20.0, True)])
def test_lora_linear_matches_peft_restatement(gpu, dtype, tol, rank, alpha, with_res):
    from diffusion_pipe_amd import nn as dnn
    from oracle import lora_ref
    torch.manual_seed(rank)
    ref = lora_ref.LoRALinearRef(torch.nn.Linear(192, 320), rank, alpha)
    torch.nn.init.normal_(ref.lora_B['default'].weight, std=0.05)
    mod = dnn.LoRALinear(dnn.Linear(192, 320, dtype=torch.float32), rank, alpha)
    mod.load_state_dict(ref.state_dict())
    mod.to(device=gpu, dtype=dtype)
    assert not mod.base_layer.weight.requires_grad and mod.lora_A['default'].weight.requires_grad
    x = torch.randn(2, 130, 192)
    res = torch.randn(2, 130, 320) if with_res else None
    xr = x.clone().requires_grad_(True)
    want = ref(xr) + (res if with_res else 0)
    g = torch.randn_like(want)
    want.backward(g)
    xg = x.to(device=gpu, dtype=dtype).requires_grad_(True)
    got = mod(xg, res.to(device=gpu, dtype=dtype) if with_res else None)
    got.backward(g.to(device=gpu, dtype=dtype))
    assert _rel(got, want) < tol
    assert _rel(xg.grad, xr.grad) < tol
    assert _rel(mod.lora_A['default'].weight.grad, ref.lora_A['default'].weight.grad) < tol
    assert _rel(mod.lora_B['default'].weight.grad, ref.lora_B['default'].weight.grad) < tol
    assert mod.base_layer.weight.grad is None and mod.base_layer.bias.grad is None
    # second, independent check: the adapter equals a Linear with the merged weight W + (alpha/r) B A
    merged = torch.nn.functional.linear(x, lora_ref.merged_weight(ref), ref.base_layer.bias) + (res if with_res else 0)
    assert _rel(got, merged) < tol


def _lora_setup(dtype, gpu, hip_graph, gas=2):
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    from oracle import eager_step, lora_ref, sdxl_ref
    cfg = sdxl.tiny_config()
    adapter = {'type': 'lora', 'rank': 4, 'alpha': 8, 'dropout': 0.0, 'dtype': torch.float32}
    ref = sdxl_ref.SDXLRef(cfg, seed=1)
    work = sdxl.SDXLWorkload(cfg, model_config={'min_snr_gamma': 5.0}, dtype=torch.float32, seed=2)
    wrapped = work.configure_adapter(adapter)
    in_blocks = lambda name, module: name.split('.')[0] in ('down_blocks', 'mid_block', 'up_blocks')
    gen = torch.Generator().manual_seed(11)
    for k, m in ref.modules().items():
        names = lora_ref.apply_lora_ref(m, adapter['rank'], adapter['alpha'], target=in_blocks if k == 'unet' else None)
        assert names == wrapped[k]                       # the same layers carry adapters on both sides
        for n, p in m.named_parameters():
            if '.lora_B.' in n:                          # lora_B = 0 would zero every lora_A gradient: start off-init
                p.data.normal_(0, 0.05, generator=gen)
        work.modules()[k].load_state_dict(m.state_dict())
        work.modules()[k].to(dtype)
    batch = sdxl.synthetic_batch(cfg, batch_size=2 * gas, latent_hw=32, seed=3, ids_len=75)
    torch.manual_seed(7)
    micro = split_batch(work.prepare_inputs(batch), gas)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(),
                                  dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': gas,
                                                         'gradient_clipping': 1.0, 'hip_graph': hip_graph}, device=gpu)
    params = [p for p in module.parameters() if p.requires_grad]
    assert params and all('.lora_' in p.original_name for p in params)
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), params)
    snr = eager_step.all_snr(eager_step.ddpm_alphas_cumprod())
    return engine, work, ref, micro, eager_step.sdxl_loss_fn(snr_table=snr, min_snr_gamma=5.0), eager_step


@pytest.mark.parametrize('dtype,tol,hip_graph', [(torch.float32, 1e-3, False), (torch.float32, 1e-3, True), (torch.bfloat16, 4e-2, True)])
def test_sdxl_lora_step_matches_oracle(gpu, dtype, tol, hip_graph):
    engine, work, ref, micro, ref_loss_fn, eager_step = _lora_setup(dtype, gpu, hip_graph)
    ref_params = [p for p in ref.parameters() if p.requires_grad]
    want_loss, want_norm = eager_step.eager_train_step(ref.to_layers(), ref_loss_fn, copy.deepcopy(micro), None,
                                                       gradient_clipping=1.0, params=ref_params)
    for _ in range(2 if hip_graph else 1):               # step 2 replays the captured graphs (lr = 0: same weights)
        loss = engine.train_batch(iter(copy.deepcopy(micro))).item()
        norm = engine.get_global_grad_norm().item()
        assert abs(loss - want_loss.item()) / abs(want_loss.item()) < tol, (loss, want_loss.item())
        assert abs(norm - want_norm.item()) / want_norm.item() < 2 * tol, (norm, want_norm.item())
    frozen = [p for m in work.modules().values() for n, p in m.named_parameters() if '.lora_' not in n]
    assert frozen and all(p.grad is None for p in frozen)


def test_wan_lora_block_matches_oracle(gpu):
    """Adapters inside WanAttentionBlock (the reference's adapter_target_modules for Wan): block output and adapter
    gradients vs the oracle block with the peft restatement around the same Linears."""
    from diffusion_pipe_amd.workloads import wan
    from oracle import blocks_ref, lora_ref
    cfg = wan.tiny_wan_config()
    work = wan.WanWorkload(cfg, dtype=torch.float32, seed=5)
    names = work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 4})
    assert len(names) == 10 * cfg.num_layers
    blk = work.transformer.blocks[0]
    gen = torch.Generator().manual_seed(3)
    for n, p in blk.named_parameters():
        if '.lora_B.' in n:
            p.data.normal_(0, 0.05, generator=gen)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    dim, heads = cfg.dim, cfg.num_heads
    S, L = 96, 20
    x, ctx = torch.randn(1, S, dim, generator=gen), torch.randn(1, L, dim, generator=gen)
    e = torch.randn(1, 1, 6, dim, generator=gen) * 0.1
    ang = torch.randn(S, dim // heads // 2, generator=gen)
    cos, sin = ang.cos(), ang.sin()
    merged = {k.replace('.base_layer', ''): v for k, v in sd.items() if '.lora_' not in k}
    for n in ('self_attn.q', 'self_attn.k', 'self_attn.v', 'self_attn.o', 'cross_attn.q', 'cross_attn.k', 'cross_attn.v', 'cross_attn.o',
              'ffn.0', 'ffn.2'):                 # alpha / r = 1: merged weight W + B A
        merged[f'{n}.weight'] = merged[f'{n}.weight'] + sd[f'{n}.lora_B.default.weight'] @ sd[f'{n}.lora_A.default.weight']
    want = blocks_ref.wan_block(merged, x, e, ctx, heads, cos, sin, cfg.eps)
    blk.to(gpu)
    got = blk(x.to(gpu), e.to(gpu), cos.to(gpu), sin.to(gpu), ctx.to(gpu))
    assert _rel(got, want) < 1e-3
    got.sum().backward()
    assert all(p.grad is not None for n, p in blk.named_parameters() if '.lora_' in n)
    assert all(p.grad is None for n, p in blk.named_parameters() if '.lora_' not in n)
