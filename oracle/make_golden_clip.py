"""ORACLE fixture generator (test infrastructure): the SDXL text encoders from the REAL library -- HF transformers' CLIPTextModel and
CLIPTextModelWithProjection (the classes the reference loads behind models/sdxl.py:385-396, both trained in SDXL full fine-tuning,
docs/supported_models.md:50) -- instantiated on CPU with the tiny geometries of `workloads.sdxl.tiny_config()` and random weights.
transformers is installed in this image (version recorded in the manifest; the reference leaves it unpinned, requirements.txt:4).
`eos_token_id = 2` as in the published SDXL encoder configs, i.e. pooled output = hidden state at argmax(input_ids).

Writes tests/golden/clip_encoders_fp32.safetensors (+ .json): weights (keys in the SDXL checkpoint layout `text_model.*`), ids, the
penultimate hidden state (what models/sdxl.py:779-784 takes), the first output (text_embeds for encoder 2) and all gradients.
Pins oracle/sdxl_ref.py:CLIPTextModel on CPU and (next) the HIP-kernel encoders on the MI355X.

    python oracle/make_golden_clip.py
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, '..', 'tests', 'golden')


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from diffusion_pipe_amd.workloads import sdxl
    cfg = sdxl.tiny_config()
    tensors, meta = {}, {'transformers': transformers.__version__, 'torch': torch.__version__, 'encoders': {}}
    g = torch.Generator().manual_seed(31)
    for tag, c, klass in (('te1', cfg.te1, CLIPTextModel), ('te2', cfg.te2, CLIPTextModelWithProjection)):
        hf_cfg = CLIPTextConfig(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers,
                                num_attention_heads=c.heads, max_position_embeddings=c.max_pos, hidden_act=c.act, layer_norm_eps=1e-5,
                                projection_dim=c.proj_dim or c.hidden, bos_token_id=c.bos, eos_token_id=2, pad_token_id=c.pad)
        torch.manual_seed(100 + len(tag) + (1 if tag == 'te2' else 0))
        model = klass(hf_cfg).float().train()
        ids = torch.randint(3, c.vocab - 3, (2, c.max_pos), generator=g)
        out = model(ids, output_hidden_states=True)
        penult, first = out.hidden_states[-2], out[0]
        w1, w2 = torch.randn(penult.shape, generator=g), torch.randn(first.shape, generator=g)
        # encoder 1: only the penultimate hidden state is consumed (models/sdxl.py:779-784); encoder 2 also feeds its projected pooled output
        loss = (penult * w1).sum() + ((first * w2).sum() if tag == 'te2' else 0.0)
        loss.backward()
        for k, v in model.state_dict().items():
            key = k if k.startswith(('text_model.', 'text_projection.')) else 'text_model.' + k
            tensors[f'{tag}.param.{key}'] = v.detach().clone()
        for k, p in model.named_parameters():
            key = k if k.startswith(('text_model.', 'text_projection.')) else 'text_model.' + k
            tensors[f'{tag}.grad.{key}'] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone()
        tensors.update({f'{tag}.ids': ids, f'{tag}.penultimate': penult.detach(), f'{tag}.first': first.detach(), f'{tag}.w1': w1, f'{tag}.w2': w2,
                        f'{tag}.loss': loss.detach().reshape(1)})
        meta['encoders'][tag] = {'class': klass.__name__, 'hidden_states': len(out.hidden_states), 'act': c.act, 'loss': float(loss.detach())}
    os.makedirs(OUT, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'clip_encoders_fp32.safetensors'))
    with open(os.path.join(OUT, 'clip_encoders_fp32.json'), 'w') as fh:
        json.dump(meta, fh, indent=1)
    print(meta)


if __name__ == '__main__':
    main()
