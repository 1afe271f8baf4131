"""ORACLE fixture generator (test infrastructure): saved-file formats produced by the REFERENCE'S OWN code, lifted with `ast` at
generation time -- FluxPipeline.save_model with its module-level BFL mapping (models/flux.py:19-113,257-290), WanPipeline.save_adapter /
save_model (models/wan/wan.py:258-265) and BasePipeline.load_adapter_weights (models/base.py:367-386) -- run on seeded tiny state dicts.
Records the produced files' SHA-256 and key / shape listings in tests/golden/formats.json (no tensors: the inputs are rebuilt from the seed).

    python oracle/make_golden_formats.py
"""
import ast
import hashlib
import json
import os
import re
import sys
import tempfile
from pathlib import Path

import safetensors
import safetensors.torch
import torch
from safetensors.torch import load_file, save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden_reflogic import REF, lift                    # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')


def sha(path):
    return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def flux_state_dict():
    from diffusion_pipe_amd.workloads import flux
    work = flux.FluxWorkload(flux.tiny_flux_config(), dtype=torch.float32, seed=11)
    return {k: v.detach().clone() for k, v in work.transformer.state_dict().items()}


def wan_lora_state_dict():
    """peft-style adapter tensors as utils/saver.py:66-75 hands them to save_adapter (adapter name stripped)."""
    from diffusion_pipe_amd.workloads import wan
    work = wan.WanWorkload(wan.tiny_wan_config(), dtype=torch.float32, seed=12)
    work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 4})
    g = torch.Generator().manual_seed(13)
    sd = {}
    for n, p in work.transformer.named_parameters():
        if p.requires_grad:
            sd[n.replace('.default', '')] = torch.randn(p.shape, generator=g)
    return sd


def sdxl_state_dict():
    """{original_name: tensor} of a tiny SDXL (same module tree / parameter names as diffusers' UNet2DConditionModel + both CLIP encoders)."""
    from diffusion_pipe_amd.workloads import sdxl
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=21)
    return {p.original_name: p.detach().clone().contiguous() for m in work.modules().values() for p in m.parameters()}


def fake_vae_state_dict():
    """diffusers AutoencoderKL parameter NAMES touching every renaming rule of models/sdxl.py:183-226 (tiny seeded tensors)."""
    g = torch.Generator().manual_seed(22)
    names = []
    for side, blocks, nres in (('encoder', 'down_blocks', 2), ('decoder', 'up_blocks', 3)):
        for i in range(4):
            for j in range(nres):
                for leaf in ('norm1.weight', 'conv1.weight', 'conv1.bias', 'norm2.bias', 'conv2.weight'):
                    names.append(f'{side}.{blocks}.{i}.resnets.{j}.{leaf}')
            names.append(f'{side}.{blocks}.{i}.resnets.0.conv_shortcut.weight')
            if i < 3:
                names.append(f'{side}.{blocks}.{i}.{"downsamplers" if side == "encoder" else "upsamplers"}.0.conv.weight')
        for j in range(2):
            names += [f'{side}.mid_block.resnets.{j}.norm1.weight', f'{side}.mid_block.resnets.{j}.conv2.bias']
        names += [f'{side}.mid_block.attentions.0.group_norm.weight', f'{side}.mid_block.attentions.0.group_norm.bias']
        for proj in ('to_q', 'to_k', 'to_v', 'to_out.0'):
            names += [f'{side}.mid_block.attentions.0.{proj}.weight', f'{side}.mid_block.attentions.0.{proj}.bias']
        names += [f'{side}.conv_in.weight', f'{side}.conv_norm_out.weight', f'{side}.conv_out.bias']
    names += ['quant_conv.weight', 'post_quant_conv.bias']
    sd = {}
    for n in names:
        shape = (8, 8) if ('to_' in n and n.endswith('weight')) else ((8, 4, 3, 3) if n.endswith('conv.weight') or 'conv1.weight' in n or 'conv2.weight' in n else (8,))
        sd[n] = torch.randn(shape, generator=g)
    return sd


def sdxl_peft_state_dict():
    """peft-style LoRA tensors of the tiny SDXL as utils/saver.py:66-75 hands them to save_adapter."""
    from diffusion_pipe_amd.workloads import sdxl
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=23)
    work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 4})
    g = torch.Generator().manual_seed(24)
    return {p.original_name.replace('.default', ''): torch.randn(p.shape, generator=g)
            for m in work.modules().values() for p in m.parameters() if p.requires_grad}


def main():
    gold = {'torch': torch.__version__, 'safetensors': safetensors.__version__}
    # ---- SDXL full fine-tune -> single-file (ldm) checkpoint: the reference's save_model with its module-level conversion tables / functions
    full = os.path.join(REF, 'models', 'sdxl.py')
    tree = ast.parse(open(full).read(), filename=full)
    first = next(n.lineno for n in tree.body if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == 'unet_conversion_map' for t in n.targets))
    last = next(n.end_lineno for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'convert_openai_text_enc_state_dict')
    nodes = [n for n in tree.body if first <= n.lineno and n.end_lineno <= last]
    ns = {'torch': torch, 're': re, 'save_file': save_file, 'print': lambda *a, **k: None}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), full, 'exec'), ns)
    sdxl_save_model, where = lift('models/sdxl.py', 'save_model', cls='SDXLPipeline', namespace=ns)
    with tempfile.TemporaryDirectory() as d:
        vae = fake_vae_state_dict()
        stub = type('S', (), {'vae': type('V', (), {'state_dict': staticmethod(lambda: vae)})()})()
        sdxl_save_model(stub, Path(d), sdxl_state_dict())
        f = os.path.join(d, 'model.safetensors')
        sd = load_file(f)
        gold['sdxl_ldm'] = {'generated_from': f'models/sdxl.py:{first}-{last} + {where}', 'sha256': sha(f), 'keys': {k: list(v.shape) for k, v in sorted(sd.items())}}
    # ---- Flux full fine-tune -> BFL file
    full = os.path.join(REF, 'models', 'flux.py')
    tree = ast.parse(open(full).read(), filename=full)
    consts = [n for n in tree.body if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in ('NUM_DOUBLE_BLOCKS', 'NUM_SINGLE_BLOCKS', 'BFL_TO_DIFFUSERS_MAP')
                                                                       for t in n.targets)]
    mapfn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'make_diffusers_to_bfl_map')
    ns = {'torch': torch, 'save_file': save_file, 'logger': type('L', (), {'error': staticmethod(lambda *a: None)})}
    exec(compile(ast.Module(body=consts + [mapfn], type_ignores=[]), full, 'exec'), ns)
    save_model, where = lift('models/flux.py', 'save_model', cls='FluxPipeline', namespace=ns)
    with tempfile.TemporaryDirectory() as d:
        save_model(None, Path(d), flux_state_dict())
        f = os.path.join(d, 'model.safetensors')
        sd = load_file(f)
        gold['flux_bfl'] = {'generated_from': where, 'sha256': sha(f), 'keys': {k: list(v.shape) for k, v in sorted(sd.items())}}
    # ---- Wan adapter / full model files and the adapter loader
    wan_save_adapter, w1 = lift('models/wan/wan.py', 'save_adapter', cls='WanPipeline', namespace={'safetensors': safetensors})
    wan_save_model, w2 = lift('models/wan/wan.py', 'save_model', cls='WanPipeline', namespace={'safetensors': safetensors})
    load_adapter, w3 = lift('models/base.py', 'load_adapter_weights', cls='BasePipeline', namespace={
        'safetensors': safetensors, 're': re, 'Path': Path, 'is_main_process': lambda: False})
    lora = wan_lora_state_dict()
    with tempfile.TemporaryDirectory() as d:
        stub = type('S', (), {'peft_config': type('P', (), {'save_pretrained': staticmethod(lambda path: None)})})()
        wan_save_adapter(stub, Path(d), lora)
        f = os.path.join(d, 'adapter_model.safetensors')
        gold['wan_adapter'] = {'generated_from': w1, 'sha256': sha(f), 'keys': sorted(load_file(f))[:4], 'count': len(lora)}
        # the reference's loader maps the file back onto peft parameter names
        from diffusion_pipe_amd.workloads import wan
        work = wan.WanWorkload(wan.tiny_wan_config(), dtype=torch.float32, seed=12)
        work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 4})
        loaded = {}
        tr = work.transformer
        orig = tr.load_state_dict
        tr.load_state_dict = lambda sd, strict=True: loaded.update(sd)
        load_adapter(type('S', (), {'transformer': tr})(), d)
        tr.load_state_dict = orig
        gold['wan_adapter']['loader'] = {'generated_from': w3, 'keys': sorted(loaded)[:4], 'count': len(loaded)}
    with tempfile.TemporaryDirectory() as d:
        from diffusion_pipe_amd.workloads import wan
        work = wan.WanWorkload(wan.tiny_wan_config(), dtype=torch.float32, seed=12)
        full_sd = {n: p.detach().clone() for n, p in work.transformer.named_parameters()}
        wan_save_model(None, Path(d), full_sd)
        gold['wan_model'] = {'generated_from': w2, 'sha256': sha(os.path.join(d, 'model.safetensors')), 'count': len(full_sd)}
    with open(os.path.join(OUT, 'formats.json'), 'w') as fh:
        json.dump(gold, fh, indent=1)
    print({k: (v if not isinstance(v, dict) else {kk: (vv if kk != 'keys' else len(vv)) for kk, vv in v.items()}) for k, v in gold.items()})


if __name__ == '__main__':
    main()
