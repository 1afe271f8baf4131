"""ORACLE fixture generator (test infrastructure): the reference's OWN HunyuanVideo pipeline-layer code -- InitialLayer, DoubleBlock,
concatenate_hidden_states, SingleBlock, OutputLayer (models/hunyuan_video.py:544-680), HunyuanVideoPipeline.to_layers (:483-492),
prepare_inputs (:413-481) and get_rotary_pos_embed (:35-81) -- lifted with `ast` and executed on CPU over the oracle's restatement of the
un-vendored hyvideo transformer (oracle/hv_ref.py; [3P] get_nd_rotary_pos_embed / get_cu_seqlens = the oracle's restatements).  Pins the
stage-boundary tuple layouts, the bare-callable concatenate layer, the x1000 timestep / guidance scaling, the rotary-table geometry, the
text-mask -> cu_seqlens hand-off and the final image-token slice + unpatchify.  Writes tests/golden/hv_layers.{json,safetensors}.

    python oracle/make_golden_hv_layers.py
"""
import ast
import json
import os
import sys

import torch
import torch.nn.functional as F
from safetensors.torch import save_file
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import hv_ref                                            # noqa: E402
from oracle.make_golden_reflogic import REF, lift                    # noqa: E402
from oracle.make_golden_wan_model import lift_classes                # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
SEED = 5
MODEL_CONFIG = {'guidance': 6.0, 'timestep_sample_method': 'logit_normal', 'sigmoid_scale': 1.2, 'shift': 3.0}


def weight_checksum(tr):
    return float(sum(p.detach().double().abs().sum() for p in tr.parameters()))


def batch(cfg, seed=12):
    g = torch.Generator().manual_seed(seed)
    mask = torch.zeros(2, 12, dtype=torch.int64)
    mask[0, :12] = 1
    mask[1, :7] = 1                                                   # 5 of 12 text tokens of sample 1 are padding
    return {'latents': torch.randn(2, cfg.in_channels, 3, 8, 12, generator=g), 'mask': torch.rand(2, 64, 96, generator=g).round(),
            'prompt_embeds_1': torch.randn(2, 12, cfg.text_states_dim, generator=g), 'prompt_attention_mask_1': mask,
            'prompt_embeds_2': torch.randn(2, cfg.text_states_dim_2, generator=g)}


def main():
    from diffusion_pipe_amd.workloads import hunyuan_video as hv
    cfg = hv.tiny_hv_config()
    tr = hv_ref.HYVideoDiffusionTransformer(cfg, seed=SEED)
    make_contiguous, _ = lift('models/base.py', 'make_contiguous', namespace={'torch': torch})
    ns = lift_classes('models/hunyuan_video.py', {'InitialLayer', 'DoubleBlock', 'SingleBlock', 'OutputLayer'},
                      {'nn': nn, 'torch': torch, 'make_contiguous': make_contiguous, 'get_cu_seqlens': hv_ref.get_cu_seqlens})
    concat, w_cat = lift('models/hunyuan_video.py', 'concatenate_hidden_states', namespace={'torch': torch})
    ns['concatenate_hidden_states'] = concat
    to_layers, w_layers = lift('models/hunyuan_video.py', 'to_layers', cls='HunyuanVideoPipeline', namespace=ns)
    rope, w_rope = lift('models/hunyuan_video.py', 'get_rotary_pos_embed', namespace={'get_nd_rotary_pos_embed': hv_ref.get_nd_rotary_pos_embed})
    prep, w_prep = lift('models/hunyuan_video.py', 'prepare_inputs', cls='HunyuanVideoPipeline', namespace={'torch': torch, 'F': F, 'get_rotary_pos_embed': rope})
    off = type('Off', (), {'wait_for_block': staticmethod(lambda i: None), 'submit_move_blocks_forward': staticmethod(lambda i: None)})
    owner = type('HunyuanVideoPipelineStub', (), {'transformer': tr, 'offloader_double': off, 'offloader_single': off, 'model_config': MODEL_CONFIG})()
    layers = to_layers(owner)
    b = batch(cfg)
    torch.manual_seed(8)
    features, (target, mask) = prep(owner, b)
    torch.manual_seed(9)
    features_q, _ = prep(owner, dict(b, mask=None), timestep_quantile=0.3)
    x = tuple(f.clone() for f in features)
    layouts, tensors = [], {}
    for i, layer in enumerate(layers):
        x = layer(x)
        layouts.append([[list(v.shape), str(v.dtype)] for v in x] if isinstance(x, tuple) else [list(x.shape), str(x.dtype)])
        if i == 0:
            for j, name in enumerate(('img', 'txt', 'vec', 'cu_seqlens', 'max_seqlen', 'freqs_cos', 'freqs_sin', 'txt_seq_len', 'img_seq_len', 'unpatchify_args')):
                tensors[f'initial.{name}'] = x[j].detach().clone()
    loss = (F.mse_loss(x.float(), target, reduction='none') * mask).mean()         # models/base.py:418-436 default loss with a mask
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in tr.named_parameters() if p.grad is not None}
    pick = ['time_in.mlp.0.weight', 'txt_in.individual_token_refiner.blocks.1.self_attn_qkv.weight', 'txt_in.c_embedder.linear_1.weight', 'img_in.proj.weight',
            'double_blocks.0.img_attn_qkv.weight', 'double_blocks.1.txt_mlp.fc2.weight', 'single_blocks.2.linear1.weight', 'final_layer.linear.weight', 'guidance_in.mlp.2.bias']
    for n in pick:
        tensors[f'grad.{n}'] = grads[n]
    tensors.update({f'batch.{k}': v for k, v in b.items()})
    tensors.update({'target': target, 'label_mask': mask, 'out': x.detach().clone(), 'loss': loss.detach().reshape(1)})
    for i, f in enumerate(features):
        tensors[f'feature.{i}'] = f.clone()
    tensors['quantile.t'], tensors['quantile.x_t'] = features_q[1].clone(), features_q[0].clone()
    gsq = float(sum((g.double() ** 2).sum() for g in grads.values()))
    meta = {'generated_from': {'layers': 'models/hunyuan_video.py:544-680 (lifted)', 'concatenate_hidden_states': w_cat, 'to_layers': w_layers, 'prepare_inputs': w_prep,
                               'get_rotary_pos_embed': w_rope},
            'seed': SEED, 'seed_prepare_inputs': 8, 'seed_quantile': 9, 'model_config': MODEL_CONFIG, 'weight_checksum': weight_checksum(tr), 'torch': torch.__version__,
            'layer_names': [getattr(l, '__name__', type(l).__name__) for l in layers], 'layouts': layouts, 'loss': float(loss), 'grad_norm': gsq ** 0.5,
            'grads_with_value': len(grads), 'parameters': sum(1 for _ in tr.parameters())}
    os.makedirs(OUT, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'hv_layers.safetensors'))
    with open(os.path.join(OUT, 'hv_layers.json'), 'w') as fh:
        json.dump(meta, fh)
    print(meta['layer_names'], meta['loss'], meta['grad_norm'], meta['weight_checksum'])


if __name__ == '__main__':
    main()
