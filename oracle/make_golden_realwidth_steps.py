"""ORACLE fixture generator (test infrastructure): one `train_batch` STEP of BASELINE configs 3 / 4 / 5 at their REAL width, depth-truncated so the oracle's
fp32 eager path finishes on the host (configs 3 - 5 had step-level parity only at toy width, block-level parity only at real width).

  * flux  -- Flux.1-dev width (3072 = 24 heads of 128, 4096-wide text states): 2 double-stream + 2 single-stream blocks between the real embedders / output
             layers, 4 096 image + 512 text tokens (1024 x 1024), LoRA rank 32 on every Linear of the blocks (the config-3 training mode; lora_B seeded
             non-zero so the adapters' gradients are not trivial);
  * wan   -- Wan2.1-14B width (dim 5120, ffn 13824, 40 heads of 128): 2 DiT blocks, 4 608 video tokens (9 x 64 x 32 latent: half of config 4's 9 216 -- the
             oracle's unfused fp32 attention keeps every score matrix for its backward) + 512 text tokens, LoRA rank 32;
  * hv    -- HunyuanVideo width (3072 = 24 x 128, token refiner over 4096-wide LLM states): 1 double + 1 single stream block, 2 880 video + 256 text
             tokens of which 66 are padding (config 5's 61 456 tokens need 362 GB of fp32 scores on the host), FULL fine-tune.

Each case: the oracle's `eager_step.eager_train_step` (loss / GAS, the reference's clip_grad_norm_, utils/patches.py:175-246) over ONE micro-batch with
gradient clipping 1.0 -> mean loss, pre-clip global gradient norm, and the [sum |g|, sum g, <g, r>, ||g||_2] checksums (oracle/checksums.py) of every TRAINED
parameter's gradient before clipping.  Nothing large is stored: weights, adapters and inputs are rebuilt from seeds by `flux_case()` / `wan_case()` /
`hv_case()`, which tests/test_gpu_realwidth_steps.py imports to build the identical product workload (a weight checksum guards the seeds).

    python oracle/make_golden_realwidth_steps.py [flux] [wan] [hv]          (~30 GB of host memory, minutes of CPU per case)
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, '..', 'tests', 'golden', 'realwidth_steps.json')

RANK = 32
FLUX = dict(num_layers=2, num_single_layers=2, latent_hw=(128, 128), text_tokens=512, seed=51)
WAN = dict(num_layers=2, frames=9, latent_hw=(64, 32), text_tokens=512, seed=61)
HV = dict(double=1, single=1, latent_thw=(5, 48, 48), text_tokens=256, valid_text=(190,), seed=71)


def _seed_lora_b(module, seed):
    """peft starts lora_B at zero (the adapters then receive a zero gradient through A): give it seeded values so every adapter tensor is exercised"""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if '.lora_B.' in n:
                p.normal_(0, 0.02, generator=gen)


def state_checksum(module):
    return float(sum(v.double().abs().sum() for v in module.state_dict().values()))


def flux_case():
    """-> (cfg, product workload on CPU fp32 with LoRA configured, one micro-batch (features, label))"""
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import flux
    c = FLUX
    cfg = flux.FluxConfig(num_layers=c['num_layers'], num_single_layers=c['num_single_layers'])
    work = flux.FluxWorkload(cfg, model_config={'guidance': 1.0}, dtype=torch.float32, seed=c['seed'])
    torch.manual_seed(c['seed'] + 1)
    work.configure_adapter({'type': 'lora', 'rank': RANK, 'alpha': RANK, 'dtype': torch.float32})
    _seed_lora_b(work.transformer, c['seed'] + 2)
    torch.manual_seed(c['seed'] + 3)
    micro = split_batch(work.prepare_inputs(flux.synthetic_flux_batch(cfg, batch_size=1, latent_hw=c['latent_hw'], text_tokens=c['text_tokens'], seed=c['seed'] + 4)), 1)
    return cfg, work, micro[0]


def wan_case():
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import wan
    c = WAN
    cfg = wan.WanConfig(num_layers=c['num_layers'])
    work = wan.WanWorkload(cfg, dtype=torch.float32, seed=c['seed'])
    torch.manual_seed(c['seed'] + 1)
    work.configure_adapter({'type': 'lora', 'rank': RANK, 'alpha': RANK, 'dtype': torch.float32})
    _seed_lora_b(work.transformer, c['seed'] + 2)
    torch.manual_seed(c['seed'] + 3)
    micro = split_batch(work.prepare_inputs(wan.synthetic_wan_batch(cfg, batch_size=1, frames=c['frames'], latent_hw=c['latent_hw'], text_tokens=c['text_tokens'],
                                                                    seed=c['seed'] + 4)), 1)
    return cfg, work, micro[0]


def hv_case():
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.workloads import hunyuan_video as hv
    from oracle import hv_ref
    c = HV
    cfg = hv.HunyuanVideoConfig(mm_double_blocks_depth=c['double'], mm_single_blocks_depth=c['single'])
    tr = hv_ref.HYVideoDiffusionTransformer(cfg, seed=c['seed'])
    work = hv.HunyuanVideoWorkload(cfg, model_config={'guidance': 1.0}, dtype=torch.float32, seed=c['seed'] + 1)
    work.transformer.load_state_dict(tr.state_dict())
    torch.manual_seed(c['seed'] + 2)
    feats, label = work.prepare_inputs(hv.synthetic_hv_batch(cfg, batch_size=1, latent_thw=c['latent_thw'], text_tokens=c['text_tokens'], valid_text=c['valid_text'],
                                                             seed=c['seed'] + 3))
    micro = split_batch((feats, label), 1)
    return cfg, work, tr, micro[0]


def _grad_rows(named_params):
    from oracle.checksums import checksum4
    return {n: checksum4(p.grad, n) for n, p in named_params if p.requires_grad and p.grad is not None}


def _step(layers, micro, named_params, label_of=None):
    """the oracle step over one micro-batch; gradient rows are taken BEFORE the clip scales the gradients (clip 1e30 first, then the reference formula by hand)"""
    from oracle import eager_step
    named_params = list(named_params)
    params = [p for _, p in named_params if p.requires_grad]
    loss, norm = eager_step.eager_train_step(layers, eager_step.default_loss_fn(), [micro], None, gradient_clipping=0.0, params=params)
    rows = _grad_rows(named_params)
    return float(loss), float(norm), rows


def flux_section():
    from oracle import flux_ref, lora_ref
    cfg, work, micro = flux_case()
    ref = flux_ref.FluxRef(cfg, seed=1)
    blocks = lambda name, module: name.startswith(('transformer_blocks.', 'single_transformer_blocks.'))      # noqa: E731
    lora_ref.apply_lora_ref(ref.transformer, RANK, RANK, target=blocks)
    ref.transformer.load_state_dict(work.transformer.state_dict())
    loss, norm, rows = _step(ref.to_layers(), micro, ref.transformer.named_parameters())
    print('flux step: loss', loss, 'grad norm', norm, 'trained tensors', len(rows), flush=True)
    return {'case': FLUX, 'rank': RANK, 'source': 'oracle/flux_ref.py + oracle/lora_ref.py + oracle/eager_step.py (diffusers blocks / peft restated: parity unpinned; wrappers pinned by '
                                                  'models/flux.py:396-404,456-548)', 'loss': loss, 'grad_norm': norm, 'param_grads': rows, 'state_checksum': state_checksum(work.transformer)}


def wan_section():
    from oracle import blocks_ref as br, lora_ref  # noqa: F401
    cfg, work, micro = wan_case()
    # the oracle's Wan forward is functional over a {name: tensor} dict (oracle/blocks_ref.wan_forward): LoRA enters as merged-weight arithmetic
    # y = x W^T + (x A^T) B^T alpha / r, written out per Linear so that A and B are leaves with their own gradients
    params = {n: p.detach().clone().requires_grad_(p.requires_grad) for n, p in work.transformer.named_parameters()}
    lin = {}
    for n in params:
        if '.lora_A.' in n:
            base = n.split('.lora_A.')[0]
            lin[base] = (params[base + '.base_layer.weight'], params.get(base + '.base_layer.bias'), params[n], params[n.replace('.lora_A.', '.lora_B.')])
    flat = {}
    for n, p in params.items():
        if '.base_layer.' in n or '.lora_A.' in n or '.lora_B.' in n:
            continue
        flat[n] = p
    scale = 1.0                                                # alpha / r = 1
    for base, (w, b, a, bb) in lin.items():
        flat[base + '.weight'] = w + scale * (bb @ a)          # merged weight: d/dA and d/dB flow through the sum; W itself is frozen (requires_grad False)
        if b is not None:
            flat[base + '.bias'] = b

    def layer(f):
        x_t, _, t, te, sl, _ = f
        return br.wan_forward(flat, cfg, x_t, t, te, sl)
    named = [(n, p) for n, p in params.items()]
    loss, norm, rows = _step([layer], micro, named)
    print('wan step: loss', loss, 'grad norm', norm, 'trained tensors', len(rows), flush=True)
    return {'case': WAN, 'rank': RANK, 'source': 'oracle/blocks_ref.wan_forward (block arithmetic pinned by the reference\'s own WanAttentionBlock vectors) over LoRA-merged weights '
                                                 'W + (alpha / r) B A (peft restated: parity unpinned) + oracle/eager_step.py', 'loss': loss, 'grad_norm': norm, 'param_grads': rows,
            'state_checksum': state_checksum(work.transformer)}


def hv_section():
    from oracle.make_golden_realdims import hv_reference_forward
    cfg, work, tr, micro = hv_case()
    loss, norm, rows = _step([lambda f: hv_reference_forward(tr, cfg, f)], micro, tr.named_parameters())
    print('hv step: loss', loss, 'grad norm', norm, 'trained tensors', len(rows), flush=True)
    return {'case': HV, 'source': 'oracle/hv_ref.py + oracle/blocks_ref.py + oracle/eager_step.py (hyvideo transformer restated: parity unpinned; wrappers pinned by '
                                  'models/hunyuan_video.py:413-492)', 'loss': loss, 'grad_norm': norm, 'param_grads': rows, 'state_checksum': state_checksum(work.transformer)}


def main():
    which = sys.argv[1:] or ['flux', 'wan', 'hv']
    gold = json.load(open(OUT)) if os.path.isfile(OUT) else {}
    gold['torch'] = torch.__version__
    for name in which:
        gold[name] = {'flux': flux_section, 'wan': wan_section, 'hv': hv_section}[name]()
        with open(OUT, 'w') as fh:
            json.dump(gold, fh)


if __name__ == '__main__':
    main()
