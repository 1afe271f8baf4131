"""Saved-file formats: diffusion_pipe_amd.formats against files written by the reference's own save functions
(oracle/make_golden_formats.py lifts FluxPipeline.save_model + its BFL mapping, WanPipeline.save_adapter / save_model and
BasePipeline.load_adapter_weights) -- byte-identical safetensors files (SHA-256) for the same seeded state dicts."""
import hashlib
import json
import os

import pytest
import safetensors
import torch
from safetensors.torch import load_file

from diffusion_pipe_amd import formats
from oracle.make_golden_formats import flux_state_dict, wan_lora_state_dict

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'formats.json')))
same_build = torch.__version__ == G['torch'] and safetensors.__version__ == G['safetensors']


def _sha(path):
    return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def test_flux_full_model_is_written_in_the_bfl_layout(tmp_path):
    sd = flux_state_dict()
    from diffusion_pipe_amd.workloads import flux
    work = flux.FluxWorkload(flux.tiny_flux_config(), dtype=torch.float32)
    work.save_model(tmp_path, sd)
    got = load_file(tmp_path / 'model.safetensors')
    assert {k: list(v.shape) for k, v in sorted(got.items())} == G['flux_bfl']['keys']
    # fused projections are concatenated in q, k, v (, mlp) order; the final adaLN halves are swapped to (shift, scale)
    assert torch.equal(got['double_blocks.1.img_attn.qkv.weight'], torch.cat([sd[f'transformer_blocks.1.attn.to_{x}.weight'] for x in 'qkv']))
    assert torch.equal(got['single_blocks.2.linear1.bias'], torch.cat([sd[f'single_transformer_blocks.2.{x}.bias'] for x in ('attn.to_q', 'attn.to_k', 'attn.to_v', 'proj_mlp')]))
    scale, shift = sd['norm_out.linear.weight'].chunk(2, dim=0)
    assert torch.equal(got['final_layer.adaLN_modulation.1.weight'], torch.cat([shift, scale]))
    if same_build:
        assert _sha(tmp_path / 'model.safetensors') == G['flux_bfl']['sha256']
    with pytest.raises(KeyError):
        formats.flux_diffusers_to_bfl(dict(sd, **{'transformer_blocks.0.unknown.weight': torch.zeros(1)}))


def test_wan_adapter_and_model_files(tmp_path):
    from diffusion_pipe_amd.workloads import wan
    lora = wan_lora_state_dict()
    work = wan.WanWorkload(wan.tiny_wan_config(), dtype=torch.float32, seed=12)
    names = work.configure_adapter({'type': 'lora', 'rank': 4, 'alpha': 4})
    work.save_adapter(tmp_path / 'a', lora, {'rank': 4, 'alpha': 4, 'target_modules': names})
    saved = load_file(tmp_path / 'a' / 'adapter_model.safetensors')
    assert len(saved) == G['wan_adapter']['count'] and sorted(saved)[:4] == G['wan_adapter']['keys']
    assert all(k.startswith('diffusion_model.') and '.default' not in k for k in saved)
    cfg = json.load(open(tmp_path / 'a' / 'adapter_config.json'))
    assert cfg['r'] == 4 and cfg['lora_alpha'] == 4 and cfg['peft_type'] == 'LORA' and len(cfg['target_modules']) == len(names)
    if same_build:
        assert _sha(tmp_path / 'a' / 'adapter_model.safetensors') == G['wan_adapter']['sha256']
    # loading maps the file back onto the adapter parameters (strip prefix, re-insert the adapter name)
    os.remove(tmp_path / 'a' / 'adapter_config.json')
    loaded = work.load_adapter_weights(tmp_path / 'a')
    assert len(loaded) == G['wan_adapter']['loader']['count'] and loaded[:4] == G['wan_adapter']['loader']['keys']
    params = dict(work.transformer.named_parameters())
    assert all(torch.equal(params[k], lora[k.replace('.default', '')]) for k in loaded)
    # full fine-tune: the parameters' own names
    work2 = wan.WanWorkload(wan.tiny_wan_config(), dtype=torch.float32, seed=12)
    full = {n: p.detach().clone() for n, p in work2.transformer.named_parameters()}
    work2.save_model(tmp_path / 'm', full)
    assert len(load_file(tmp_path / 'm' / 'model.safetensors')) == G['wan_model']['count']
    if same_build:
        assert _sha(tmp_path / 'm' / 'model.safetensors') == G['wan_model']['sha256']
    with pytest.raises(RuntimeError, match='not in the model parameters'):
        from safetensors.torch import save_file
        save_file({'diffusion_model.blocks.9.nope.lora_A.weight': torch.zeros(1)}, str(tmp_path / 'bad.safetensors'))
        (tmp_path / 'b').mkdir()
        os.replace(tmp_path / 'bad.safetensors', tmp_path / 'b' / 'bad.safetensors')
        work.load_adapter_weights(tmp_path / 'b')


def test_sdxl_full_model_is_written_in_the_single_file_ldm_layout(tmp_path):
    """SDXLWorkload.save_model vs the reference's own save_model + conversion tables (models/sdxl.py:24-276,487-525, lifted): same key set,
    same shapes, byte-identical file for UNet + VAE + both text encoders (fused open_clip in_proj, transposed text_projection)."""
    from oracle.make_golden_formats import fake_vae_state_dict, sdxl_state_dict
    from diffusion_pipe_amd.workloads import sdxl
    sd, vae = sdxl_state_dict(), fake_vae_state_dict()
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=21)
    work.save_model(tmp_path, sd, vae_state_dict=vae)
    got = load_file(tmp_path / 'model.safetensors')
    assert {k: list(v.shape) for k, v in sorted(got.items())} == G['sdxl_ldm']['keys']
    assert torch.equal(got['model.diffusion_model.input_blocks.4.0.in_layers.2.weight'], sd['unet.down_blocks.1.resnets.0.conv1.weight'])
    assert torch.equal(got['model.diffusion_model.output_blocks.2.2.conv.weight'], sd['unet.up_blocks.0.upsamplers.0.conv.weight'])
    assert torch.equal(got['conditioner.embedders.1.model.transformer.resblocks.1.attn.in_proj_bias'],
                       torch.cat([sd[f'text_encoder_2.text_model.encoder.layers.1.self_attn.{x}_proj.bias'] for x in 'qkv']))
    assert torch.equal(got['conditioner.embedders.1.model.text_projection'], sd['text_encoder_2.text_projection.weight'].T)
    assert got['first_stage_model.encoder.mid.attn_1.q.weight'].shape == (8, 8, 1, 1)
    if same_build:
        assert _sha(tmp_path / 'model.safetensors') == G['sdxl_ldm']['sha256']
    with pytest.raises(RuntimeError):
        formats.sdxl_diffusers_to_ldm({'vae.x': torch.zeros(1)})


def test_sdxl_save_model_as_the_saver_calls_it_embeds_the_vae(tmp_path):
    """utils/saver.py:106 calls model.save_model(save_dir, state_dict) with TWO arguments; the reference's SDXLPipeline.save_model then embeds
    self.vae.state_dict() (models/sdxl.py:503-522).  The workload must produce the same complete key set from the VAE it holds (set_vae_state_dict), or
    from the base single-file checkpoint's first_stage_model.* tensors, and refuse to write a VAE-less file silently."""
    from oracle.make_golden_formats import fake_vae_state_dict, sdxl_state_dict
    from diffusion_pipe_amd.workloads import sdxl
    sd, vae = sdxl_state_dict(), fake_vae_state_dict()
    # 1. VAE held by the workload: two-argument call == the golden's three-argument file
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=21)
    work.set_vae_state_dict(vae)
    work.save_model(tmp_path / 'a', sd)
    got = load_file(tmp_path / 'a' / 'model.safetensors')
    assert {k: list(v.shape) for k, v in sorted(got.items())} == G['sdxl_ldm']['keys']
    if same_build:
        assert _sha(tmp_path / 'a' / 'model.safetensors') == G['sdxl_ldm']['sha256']
    # 2. VAE read lazily from the configured base checkpoint (already in ldm naming): same key set, tensors verbatim
    work2 = sdxl.SDXLWorkload(sdxl.tiny_config(), model_config={'checkpoint_path': str(tmp_path / 'a' / 'model.safetensors')}, dtype=torch.float32, seed=21)
    work2.save_model(tmp_path / 'b', sd)
    got2 = load_file(tmp_path / 'b' / 'model.safetensors')
    assert sorted(got2) == sorted(got)
    assert all(torch.equal(got2[k], got[k]) for k in got if k.startswith('first_stage_model.'))
    assert list(load_file(tmp_path / 'b' / 'model.safetensors').keys()) == list(got.keys())
    # 3. nothing to embed: loud failure, unless the caller opts in to a UNet + text-encoder-only file
    work3 = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=21)
    with pytest.raises(RuntimeError, match='no VAE weights'):
        work3.save_model(tmp_path / 'c', sd)
    work3.save_model(tmp_path / 'c', sd, allow_missing_vae=True)
    assert not any(k.startswith('first_stage_model.') for k in load_file(tmp_path / 'c' / 'model.safetensors'))


def test_sdxl_and_flux_adapter_files(tmp_path):
    """kohya LoRA naming (restated from diffusers' published convert_state_dict_to_kohya; diffusers absent -> parity unpinned): structure checks."""
    from oracle.make_golden_formats import sdxl_peft_state_dict
    from diffusion_pipe_amd.workloads import flux, sdxl
    peft_sd = sdxl_peft_state_dict()
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=23)
    work.save_adapter(tmp_path / 'k', peft_sd)
    got = load_file(tmp_path / 'k' / 'lora.safetensors')
    downs = [k for k in got if k.endswith('.lora_down.weight')]
    assert len(downs) * 2 == len(peft_sd) and len(got) == 3 * len(downs)                      # down, up, alpha per wrapped Linear
    assert {k.split('_')[0] + '_' + k.split('_')[1] for k in got} == {'lora_unet', 'lora_te1', 'lora_te2'}
    key = 'unet.down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.lora_A.weight'
    want = 'lora_unet_down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q'
    assert torch.equal(got[want + '.lora_down.weight'], peft_sd[key]) and torch.equal(got[want + '.lora_up.weight'], peft_sd[key.replace('lora_A', 'lora_B')])
    assert got[want + '.alpha'].item() == 4 and got[want + '.alpha'].dtype == torch.int64
    assert 'lora_te2_text_model_encoder_layers_0_self_attn_q_proj.lora_down.weight' in got
    assert all(k.count('.') <= 2 for k in got)
    fwork = flux.FluxWorkload(flux.tiny_flux_config(), dtype=torch.float32)
    lora = {'transformer_blocks.0.attn.to_q.lora_A.weight': torch.randn(4, 8), 'transformer_blocks.0.attn.to_q.lora_B.weight': torch.randn(8, 4)}
    fwork.save_adapter(tmp_path / 'f', lora)
    got = load_file(tmp_path / 'f' / 'pytorch_lora_weights.safetensors')
    assert sorted(got) == sorted('transformer.' + k for k in lora)


def test_saver_rejects_an_adapter_without_save_methods_before_training(tmp_path):
    from diffusion_pipe_amd.saver import Saver
    with pytest.raises(NotImplementedError):
        Saver(None, {}, True, tmp_path, object(), None, None, None)


def test_saver_checks_the_sdxl_vae_source_before_training(tmp_path):
    """a full fine-tune with nothing to embed as `first_stage_model.*` must fail when the Saver is built, not at the first checkpoint."""
    from diffusion_pipe_amd.saver import Saver
    from diffusion_pipe_amd.workloads import sdxl
    from oracle.make_golden_formats import fake_vae_state_dict
    work = sdxl.SDXLWorkload(sdxl.tiny_config(), dtype=torch.float32, seed=21)
    with pytest.raises(RuntimeError, match='no VAE weights'):
        Saver(None, {}, False, tmp_path, work, None, None, None)
    Saver(None, {}, True, tmp_path, work, None, None, None)                  # LoRA runs write no VAE
    work.set_vae_state_dict(fake_vae_state_dict())
    Saver(None, {}, False, tmp_path, work, None, None, None)
    opt_in = sdxl.SDXLWorkload(sdxl.tiny_config(), model_config={'allow_missing_vae': True}, dtype=torch.float32, seed=21)
    Saver(None, {}, False, tmp_path, opt_in, None, None, None)
