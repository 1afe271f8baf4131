"""Saved-model / adapter file formats of the reference's adapters -- the conversions that sit between the
engine's parameters (keyed by `original_name`) and the files other tools load:

  * ComfyUI-style LoRA file of the Wan / generic adapters (models/wan/wan.py:258-262, models/base.py:703-728): peft keys with the adapter
    name removed, prefixed `diffusion_model.`, `adapter_model.safetensors` + `adapter_config.json`; loading strips a `transformer.` /
    `diffusion_model.` prefix and re-inserts `.default` before the trailing `.weight`.
  * Full fine-tune of Wan: `model.safetensors` with the parameters' own names (models/wan/wan.py:264-265).
  * Full fine-tune of Flux: diffusers parameter names -> the original BFL single-file layout (models/flux.py:22-113,257-290): fused qkv
    (and qkv + mlp for single blocks) concatenated in order, adaLN scale / shift halves of the final layer swapped.
Everything here is host-side byte shuffling on CPU tensors; file bytes are identical to the reference's for the same state dict (tests)."""
import json
import os
import re
from pathlib import Path

import torch
from safetensors.torch import load_file, save_file


# ------------------------------------------------------------------------------------------------ ComfyUI-style LoRA files
def comfyui_adapter_state_dict(peft_state_dict):
    return {'diffusion_model.' + k: v for k, v in peft_state_dict.items()}


def save_comfyui_adapter(save_dir, peft_state_dict, adapter_config=None):
    """peft_state_dict: {'<module>.lora_A.weight': tensor, ...} (adapter name already stripped, utils/saver.py:74)."""
    save_dir = Path(save_dir)
    os.makedirs(save_dir, exist_ok=True)
    if adapter_config is not None:      # the fields peft.LoraConfig.save_pretrained writes that a loader needs to rebuild the adapter
        cfg = {'peft_type': 'LORA', 'r': adapter_config['rank'], 'lora_alpha': adapter_config['alpha'], 'lora_dropout': adapter_config.get('dropout', 0.0),
               'bias': 'none', 'target_modules': sorted(adapter_config.get('target_modules', []))}
        with open(save_dir / 'adapter_config.json', 'w') as fh:
            json.dump(cfg, fh, indent=2, sort_keys=True)
    save_file(comfyui_adapter_state_dict({k: v.contiguous() for k, v in peft_state_dict.items()}), save_dir / 'adapter_model.safetensors', metadata={'format': 'pt'})


def load_comfyui_adapter(model, adapter_path, adapter_name='default'):
    """Load a LoRA file into a model whose Linears were wrapped by nn.apply_lora (models/base.py:709-728)."""
    files = list(Path(adapter_path).glob('*.safetensors'))
    if len(files) == 0:
        raise RuntimeError(f'No safetensors file found in {adapter_path}')
    if len(files) > 1:
        raise RuntimeError(f'Multiple safetensors files found in {adapter_path}')
    names = {n for n, _ in model.named_parameters()}
    state = {}
    for k, v in load_file(files[0]).items():
        k = re.sub(r'^(transformer|diffusion_model)\.', '', k)
        k = re.sub(r'\.weight$', f'.{adapter_name}.weight', k)
        if k not in names:
            raise RuntimeError(f'modified_state_dict key {k} is not in the model parameters')
        state[k] = v
    model.load_state_dict(state, strict=False)
    return sorted(state)


def save_plain_model(save_dir, state_dict):
    save_dir = Path(save_dir)
    os.makedirs(save_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in state_dict.items()}, save_dir / 'model.safetensors', metadata={'format': 'pt'})


# ------------------------------------------------------------------------------------------------ Flux: diffusers -> BFL
_IO = (('in_layer', 'linear_1'), ('out_layer', 'linear_2'))
_FLUX_GLOBAL = [(f'{bfl}.{b}', f'time_text_embed.{dif}.{d}') for bfl, dif in (('time_in', 'timestep_embedder'), ('vector_in', 'text_embedder'),
                                                                            ('guidance_in', 'guidance_embedder')) for b, d in _IO]
_FLUX_GLOBAL += [('txt_in', 'context_embedder'), ('img_in', 'x_embedder'), ('final_layer.linear', 'proj_out'), ('final_layer.adaLN_modulation.1', 'norm_out.linear')]
_FLUX_DOUBLE = [('img_mod.lin', ['norm1.linear']), ('txt_mod.lin', ['norm1_context.linear']),
                ('img_attn.qkv', ['attn.to_q', 'attn.to_k', 'attn.to_v']), ('txt_attn.qkv', ['attn.add_q_proj', 'attn.add_k_proj', 'attn.add_v_proj']),
                ('img_mlp.0', ['ff.net.0.proj']), ('img_mlp.2', ['ff.net.2']), ('txt_mlp.0', ['ff_context.net.0.proj']), ('txt_mlp.2', ['ff_context.net.2']),
                ('img_attn.proj', ['attn.to_out.0']), ('txt_attn.proj', ['attn.to_add_out'])]
_FLUX_DOUBLE_SCALES = [('img_attn.norm.query_norm', 'attn.norm_q'), ('img_attn.norm.key_norm', 'attn.norm_k'),
                       ('txt_attn.norm.query_norm', 'attn.norm_added_q'), ('txt_attn.norm.key_norm', 'attn.norm_added_k')]
_FLUX_SINGLE = [('modulation.lin', ['norm.linear']), ('linear1', ['attn.to_q', 'attn.to_k', 'attn.to_v', 'proj_mlp']), ('linear2', ['proj_out'])]
_FLUX_SINGLE_SCALES = [('norm.query_norm', 'attn.norm_q'), ('norm.key_norm', 'attn.norm_k')]


def _swap_halves(t):
    a, b = t.chunk(2, dim=0)
    return torch.cat([b, a], dim=0)


def flux_diffusers_to_bfl(state_dict):
    """{diffusers FluxTransformer2DModel key: tensor} -> {BFL key: tensor}.  Unknown keys raise KeyError, like the reference."""
    sd = dict(state_dict)
    out = {}

    def take(key):
        if key not in sd:
            raise KeyError(key)
        return sd.pop(key)

    def group(bfl, parts, kinds=('weight', 'bias')):
        for kind in kinds:
            present = [f'{p}.{kind}' in sd for p in parts]
            if not any(present):
                continue
            tensors = [take(f'{p}.{kind}') for p in parts]
            out[f'{bfl}.{kind}'] = tensors[0] if len(tensors) == 1 else torch.cat(tensors)
    for bfl, dif in _FLUX_GLOBAL:
        group(bfl, [dif])
    blocks = sorted({int(m.group(1)) for k in sd for m in [re.match(r'transformer_blocks\.(\d+)\.', k)] if m})
    for b in blocks:
        for bfl, parts in _FLUX_DOUBLE:
            group(f'double_blocks.{b}.{bfl}', [f'transformer_blocks.{b}.{p}' for p in parts])
        for bfl, dif in _FLUX_DOUBLE_SCALES:
            out[f'double_blocks.{b}.{bfl}.scale'] = take(f'transformer_blocks.{b}.{dif}.weight')
    singles = sorted({int(m.group(1)) for k in sd for m in [re.match(r'single_transformer_blocks\.(\d+)\.', k)] if m})
    for b in singles:
        for bfl, parts in _FLUX_SINGLE:
            group(f'single_blocks.{b}.{bfl}', [f'single_transformer_blocks.{b}.{p}' for p in parts])
        for bfl, dif in _FLUX_SINGLE_SCALES:
            out[f'single_blocks.{b}.{bfl}.scale'] = take(f'single_transformer_blocks.{b}.{dif}.weight')
    if sd:
        raise KeyError(f'Key not found in diffusers_to_bfl_map: {sorted(sd)[0]}')
    for k in ('final_layer.adaLN_modulation.1.weight', 'final_layer.adaLN_modulation.1.bias'):
        if k in out:
            out[k] = _swap_halves(out[k])       # diffusers keeps (scale, shift), the BFL layout (shift, scale)
    return out


def save_flux_bfl(save_dir, diffusers_state_dict):
    save_dir = Path(save_dir)
    os.makedirs(save_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in flux_diffusers_to_bfl(diffusers_state_dict).items()}, save_dir / 'model.safetensors', metadata={'format': 'pt'})
