"""ORACLE fixture generator (test infrastructure): the reference's OWN Flux pipeline-layer classes (models/flux.py:456-548:
EmbeddingWrapper, TransformerWrapper, SingleTransformerWrapper, OutputWrapper) and FluxPipeline.to_layers (:396-404), lifted with `ast`
and executed on CPU over the oracle's restated diffusers blocks (presented under the class name the wrapper tests for,
`CombinedTimestepGuidanceTextProjEmbeddings`; [3P] FluxPosEmbed = oracle rope_tables), fed by the reference's own prepare_inputs
(:323-394, lifted).  Pins the stage-boundary tuple layout, the x1000 timestep / guidance scaling, the [text ; image] id order and the
final slice for oracle/flux_ref.py's wrappers.  Writes tests/golden/flux_layers.{json,safetensors}.

    python oracle/make_golden_flux_layers.py
"""
import json
import math
import os
import sys

import torch
import torch.nn.functional as F
from safetensors.torch import save_file
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import flux_ref                                          # noqa: E402
from oracle.make_golden_reflogic import lift                         # noqa: E402
from oracle.make_golden_wan_model import lift_classes                # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
SEED = 3


def weight_checksum(ref):
    return float(sum(p.detach().double().abs().sum() for p in ref.parameters()))


def main():
    from einops import rearrange
    from diffusion_pipe_amd.workloads import flux
    cfg = flux.tiny_flux_config()
    ref = flux_ref.FluxRef(cfg, seed=SEED)
    t = ref.transformer

    class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):          # the class name the reference's wrapper branches on
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, timestep, guidance, pooled):
            return self.inner(timestep, guidance, pooled)

    make_contiguous, _ = lift('models/base.py', 'make_contiguous', namespace={'torch': torch})
    ns = lift_classes('models/flux.py', {'EmbeddingWrapper', 'TransformerWrapper', 'SingleTransformerWrapper', 'OutputWrapper'},
                      {'nn': nn, 'torch': torch, 'make_contiguous': make_contiguous})
    to_layers, where = lift('models/flux.py', 'to_layers', cls='FluxPipeline', namespace=ns)
    off = type('Off', (), {'wait_for_block': staticmethod(lambda i: None), 'submit_move_blocks_forward': staticmethod(lambda i: None)})
    shim = type('T', (), {})()
    shim.x_embedder, shim.context_embedder = t.x_embedder, t.context_embedder
    shim.time_text_embed = CombinedTimestepGuidanceTextProjEmbeddings(t.time_text_embed)
    shim.pos_embed = lambda ids: flux_ref.rope_tables(ids, cfg.axes_dims_rope)
    shim.transformer_blocks, shim.single_transformer_blocks = t.transformer_blocks, t.single_transformer_blocks
    shim.norm_out, shim.proj_out = t.norm_out, t.proj_out
    owner = type('FluxPipelineStub', (), {'transformer': shim, 'offloader_double': off, 'offloader_single': off})()
    layers = to_layers(owner)

    def _ids(bs, h, w, device, dtype):
        ids = torch.zeros(h, w, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(h)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(w)[None, :]
        return ids.reshape(h * w, 3).to(device=device, dtype=dtype)
    tsh, _ = lift('models/flux.py', 'time_shift', namespace={'math': math, 'torch': torch})
    glf, _ = lift('models/flux.py', 'get_lin_function')
    prep, where_prep = lift('models/flux.py', 'prepare_inputs', cls='FluxPipeline', namespace={
        'torch': torch, 'F': F, 'rearrange': rearrange, 'get_lin_function': glf, 'time_shift': tsh})
    pstub = type('S', (), {})()
    pstub.model_config, pstub.is_flex2, pstub._prepare_latent_image_ids = {'guidance': 3.5}, False, _ids

    g = torch.Generator().manual_seed(12)
    batch = {'latents': torch.randn(2, cfg.in_channels // 4, 12, 16, generator=g), 'mask': None,
             't5_embed': torch.randn(2, 14, cfg.joint_attention_dim, generator=g), 'clip_embed': torch.randn(2, cfg.pooled_projection_dim, generator=g)}
    torch.manual_seed(6)
    features, (target, _) = prep(pstub, batch)
    x = tuple(f.clone() for f in features)
    layouts = []
    tensors = {}
    for i, layer in enumerate(layers):
        x = layer(x)
        layouts.append([list(v.shape) for v in x] if isinstance(x, tuple) else list(x.shape))
        if i == 0:
            tensors['temb'], tensors['freqs_cos'], tensors['freqs_sin'] = x[2].detach().clone(), x[3].clone(), x[4].clone()
    loss = ((x.float() - target) ** 2).mean()
    tensors.update({'latents': batch['latents'], 't5_embed': batch['t5_embed'], 'clip_embed': batch['clip_embed'], 'target': target,
                    'out': x.detach().clone(), 'loss': loss.detach().reshape(1)})
    for i, f in enumerate(features):
        tensors[f'feature.{i}'] = f.clone()
    meta = {'generated_from': {'layers': 'models/flux.py:456-548 (lifted)', 'to_layers': where, 'prepare_inputs': where_prep}, 'seed': SEED,
            'seed_prepare_inputs': 6, 'weight_checksum': weight_checksum(ref), 'torch': torch.__version__,
            'layer_names': [type(l).__name__ for l in layers], 'layouts': layouts, 'loss': float(loss)}
    os.makedirs(OUT, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'flux_layers.safetensors'))
    with open(os.path.join(OUT, 'flux_layers.json'), 'w') as fh:
        json.dump(meta, fh)
    print(meta['layer_names'], meta['loss'], meta['weight_checksum'])


if __name__ == '__main__':
    main()
