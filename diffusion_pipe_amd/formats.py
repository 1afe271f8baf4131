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


# ------------------------------------------------------------------------------------------------ SDXL full fine-tune -> single-file (ldm) layout
# models/sdxl.py:16-278 (the reference's copy of diffusers' convert_diffusers_to_original_sdxl.py) and its save_model (:489-525).  The
# reference drives the renaming with ordered string replacement; here the diffusers names are parsed structurally (block / layer
# indices), which gives the same names for every UNet / VAE / CLIP parameter (tests compare against the reference's own functions).
_SDXL_UNET_TOP = {
    'time_embedding.linear_1': 'time_embed.0', 'time_embedding.linear_2': 'time_embed.2', 'conv_in': 'input_blocks.0.0', 'conv_norm_out': 'out.0',
    'conv_out': 'out.2', 'add_embedding.linear_1': 'label_emb.0.0', 'add_embedding.linear_2': 'label_emb.0.2',
}
_SDXL_RESNET_PARTS = {'norm1': 'in_layers.0', 'conv1': 'in_layers.2', 'norm2': 'out_layers.0', 'conv2': 'out_layers.3',
                      'time_emb_proj': 'emb_layers.1', 'conv_shortcut': 'skip_connection'}


def sdxl_unet_key_to_ldm(key):
    """diffusers UNet2DConditionModel parameter name -> ldm `model.diffusion_model.*` name (without that prefix)."""
    stem, _, leaf = key.rpartition('.')
    if stem in _SDXL_UNET_TOP:
        return f'{_SDXL_UNET_TOP[stem]}.{leaf}'

    def resnet_tail(tail):
        part, _, rest = tail.partition('.')
        return f'{_SDXL_RESNET_PARTS.get(part, part)}.{rest}'
    m = re.match(r'^down_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.*)$', key)
    if m:
        i, kind, j, tail = int(m[1]), m[2], int(m[3]), m[4]
        return f'input_blocks.{3 * i + j + 1}.0.{resnet_tail(tail)}' if kind == 'resnets' else f'input_blocks.{3 * i + j + 1}.1.{tail}'
    m = re.match(r'^down_blocks\.(\d+)\.downsamplers\.0\.conv\.(.*)$', key)
    if m:
        return f'input_blocks.{3 * (int(m[1]) + 1)}.0.op.{m[2]}'
    m = re.match(r'^up_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.*)$', key)
    if m:
        i, kind, j, tail = int(m[1]), m[2], int(m[3]), m[4]
        return f'output_blocks.{3 * i + j}.0.{resnet_tail(tail)}' if kind == 'resnets' else f'output_blocks.{3 * i + j}.1.{tail}'
    m = re.match(r'^up_blocks\.(\d+)\.upsamplers\.0\.(.*)$', key)
    if m:                                            # SDXL's up blocks 0 / 1 both carry attention, so the up-sampler is always sub-module 2
        i, tail = int(m[1]), m[2]
        sub = 2 if (i > 0 or tail.startswith('conv.')) else 1
        return f'output_blocks.{3 * i + 2}.{sub}.{tail}'
    m = re.match(r'^mid_block\.attentions\.0\.(.*)$', key)
    if m:
        return f'middle_block.1.{m[1]}'
    m = re.match(r'^mid_block\.resnets\.(\d+)\.(.*)$', key)
    if m:
        return f'middle_block.{2 * int(m[1])}.{resnet_tail(m[2])}'
    return key


def sdxl_vae_key_to_ldm(key):
    """diffusers AutoencoderKL parameter name -> ldm `first_stage_model.*` name (models/sdxl.py:183-226)."""
    k = key
    m = re.match(r'^(encoder|decoder)\.(down_blocks|up_blocks)\.(\d+)\.resnets\.(\d+)\.(.*)$', k)
    if m:
        side, i, j, tail = m[1], int(m[3]), int(m[4]), m[5]
        k = f'{side}.down.{i}.block.{j}.{tail}' if m[2] == 'down_blocks' else f'{side}.up.{3 - i}.block.{j}.{tail}'
    m = re.match(r'^(encoder|decoder)\.down_blocks\.(\d+)\.downsamplers\.0\.(.*)$', k)
    if m:
        k = f'{m[1]}.down.{m[2]}.downsample.{m[3]}'
    m = re.match(r'^(encoder|decoder)\.up_blocks\.(\d+)\.upsamplers\.0\.(.*)$', k)
    if m:
        k = f'{m[1]}.up.{3 - int(m[2])}.upsample.{m[3]}'
    m = re.match(r'^(encoder|decoder)\.mid_block\.resnets\.(\d+)\.(.*)$', k)
    if m:
        k = f'{m[1]}.mid.block_{int(m[2]) + 1}.{m[3]}'
    k = k.replace('conv_shortcut', 'nin_shortcut').replace('conv_norm_out', 'norm_out')
    if 'attentions' in key:
        k = k.replace('mid_block.attentions.0.', 'mid.attn_1.')
        for hf, sd in (('group_norm.', 'norm.'), ('to_q.', 'q.'), ('to_k.', 'k.'), ('to_v.', 'v.'), ('to_out.0.', 'proj_out.')):
            k = k.replace(hf, sd)
    return k


def sdxl_vae_to_ldm(vae_state_dict):
    out = {}
    for key, v in vae_state_dict.items():
        k = sdxl_vae_key_to_ldm(key)
        if any(f'mid.attn_1.{w}.weight' in k for w in ('q', 'k', 'v', 'proj_out')) and v.ndim != 1:
            v = v.reshape(*v.shape, 1, 1)            # HF linear attention weights -> the ldm VAE's 1x1 convolutions
        out[k] = v
    return out


_OPENCLIP_RENAMES = (('text_model.encoder.layers.', 'transformer.resblocks.'), ('layer_norm1', 'ln_1'), ('layer_norm2', 'ln_2'), ('.fc1.', '.c_fc.'), ('.fc2.', '.c_proj.'),
                     ('.self_attn', '.attn'), ('text_model.final_layer_norm.', 'ln_final.'), ('text_model.embeddings.token_embedding.weight', 'token_embedding.weight'),
                     ('text_model.embeddings.position_embedding.weight', 'positional_embedding'))
_OPENCLIP_RE = re.compile('|'.join(re.escape(hf) for hf, _ in _OPENCLIP_RENAMES))
_OPENCLIP_MAP = dict(_OPENCLIP_RENAMES)


def sdxl_openclip_to_ldm(text_enc_dict):
    """HF CLIPTextModelWithProjection names -> open_clip names; q / k / v projections fuse into in_proj_weight / in_proj_bias (models/sdxl.py:228-272)."""
    rename = lambda k: _OPENCLIP_RE.sub(lambda m: _OPENCLIP_MAP[m.group(0)], k)
    out, fused = {}, {}
    for k, v in text_enc_dict.items():
        m = re.match(r'^(.*\.self_attn)\.([qkv])_proj\.(weight|bias)$', k)
        if m:
            fused.setdefault((m[1], m[3]), {})[m[2]] = v
            continue
        out[rename(k)] = v
    for kind in ('weight', 'bias'):                  # the reference emits every fused weight before every fused bias
        for (pre, what), parts in fused.items():
            if what != kind:
                continue
            if set(parts) != set('qkv'):
                raise RuntimeError('CORRUPTED MODEL: one of the q-k-v values for the text encoder was missing')
            out[f'{rename(pre)}.in_proj_{what}'] = torch.cat([parts['q'], parts['k'], parts['v']])
    return out


def sdxl_diffusers_to_ldm(diffusers_sd, vae_state_dict=None, ldm_vae=None):
    """{'unet.*', 'text_encoder.*', 'text_encoder_2.*'} (parameters' original_name keys) -> single-file SDXL state dict (models/sdxl.py:489-523).
    vae_state_dict: diffusers AutoencoderKL names (converted like the reference does); ldm_vae: tensors already named `first_stage_model.*` (taken verbatim
    from the base single-file checkpoint), inserted at the same position so the key order equals the reference writer's."""
    unet, te1, te2 = {}, {}, {}
    for name, p in diffusers_sd.items():
        if name.startswith('unet.'):
            unet[name[len('unet.'):]] = p
        elif name.startswith('text_encoder.'):
            te1[name[len('text_encoder.'):]] = p
        elif name.startswith('text_encoder_2.'):
            te2[name[len('text_encoder_2.'):]] = p
        else:
            raise RuntimeError(f'Unexpected parameter: {name}')
    out = {'model.diffusion_model.' + sdxl_unet_key_to_ldm(k): v for k, v in unet.items()}
    if vae_state_dict is not None:
        out.update({'first_stage_model.' + k: v for k, v in sdxl_vae_to_ldm(vae_state_dict).items()})
    elif ldm_vae is not None:
        out.update(ldm_vae)
    out.update({'conditioner.embedders.0.transformer.' + k: v for k, v in te1.items()})
    te2 = {'conditioner.embedders.1.model.' + k: v for k, v in sdxl_openclip_to_ldm(te2).items()}
    proj = 'conditioner.embedders.1.model.text_projection'
    if proj + '.weight' in te2:
        te2[proj] = te2.pop(proj + '.weight').T.contiguous()
    out.update(te2)
    return out


def save_sdxl_ldm(save_dir, diffusers_sd, vae_state_dict=None, ldm_vae=None):
    save_dir = Path(save_dir)
    os.makedirs(save_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in sdxl_diffusers_to_ldm(diffusers_sd, vae_state_dict, ldm_vae).items()}, save_dir / 'model.safetensors', metadata={'format': 'pt'})


# ------------------------------------------------------------------------------------------------ SDXL LoRA -> kohya file, Flux LoRA -> diffusers file
def peft_to_kohya(peft_state_dict):
    """peft LoRA names (adapter name stripped, `unet.` / `text_encoder.` / `text_encoder_2.` prefixes) -> kohya-ss names, the conversion the reference
    reaches through diffusers.utils.state_dict_utils.convert_state_dict_to_kohya (models/sdxl.py:465-474; diffusers is absent offline: restated from
    the published utility -- parity unpinned): lora_A / lora_B -> lora_down / lora_up, component prefix -> lora_unet / lora_te1 / lora_te2, every dot
    but the last two -> '_', and one `<module>.alpha` = rank per lora_down."""
    out = {}
    for key, weight in peft_state_dict.items():
        k = key
        for hf, kohya in (('lora_A', 'lora_down'), ('lora_B', 'lora_up')):
            if hf in k:
                k = k.replace(hf, kohya)
                break
        if 'text_encoder_2.' in k:
            k = k.replace('text_encoder_2.', 'lora_te2.')
        elif 'text_encoder.' in k:
            k = k.replace('text_encoder.', 'lora_te1.')
        elif 'unet' in k:
            k = k.replace('unet', 'lora_unet')
        elif 'lora_magnitude_vector' in k:
            k = k.replace('lora_magnitude_vector', 'dora_scale')
        k = k.replace('.', '_', k.count('.') - 2)
        out[k] = weight
        if 'lora_down' in k:
            out[f'{k.split(".")[0]}.alpha'] = torch.tensor(len(weight))
    return out


def save_sdxl_kohya_lora(save_dir, peft_state_dict):
    save_dir = Path(save_dir)
    os.makedirs(save_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in peft_to_kohya(peft_state_dict).items()}, save_dir / 'lora.safetensors', metadata={'format': 'pt'})


def save_flux_diffusers_lora(save_dir, peft_state_dict):
    """Flux LoRA file as diffusers' FluxPipeline.save_lora_weights(transformer_lora_layers=...) writes it (models/flux.py:231-236; diffusers absent
    offline, restated: every key prefixed `transformer.`, file pytorch_lora_weights.safetensors) -- parity unpinned."""
    save_dir = Path(save_dir)
    os.makedirs(save_dir, exist_ok=True)
    save_file({'transformer.' + k: v.contiguous() for k, v in peft_state_dict.items()}, save_dir / 'pytorch_lora_weights.safetensors', metadata={'format': 'pt'})
