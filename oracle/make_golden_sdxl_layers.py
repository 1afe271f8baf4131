"""ORACLE fixture generator (test infrastructure): the reference's OWN SDXL pipeline-layer classes (models/sdxl.py:654-995:
InitialLayer ... FinalLayer, Unet{Down,Mid,Up}BlockLayer.to_layers and SDXLPipeline.to_layers :591-602), lifted out of the file with `ast`
at generation time and executed on CPU over the oracle's restated diffusers / CLIP blocks, which are presented to them through a shim
that speaks the diffusers API the wrappers call ([3P], restated: `unet.get_time_embed / get_aug_embed / process_encoder_hidden_states`,
`Transformer2DModel(..., return_dict=False)[0]`, `CLIPTextModel(ids, output_hidden_states=True)`, tokenizer special ids).

What this pins: the in-tree dataflow of the 23 layers -- skip-stack push / pop order, `forward_upsample_size`, the chunked BOS / EOS /
pad prompt handling, mid-block order, the stage-boundary tuple layout -- for oracle/sdxl_ref.py's wrappers (CPU test) and, through them,
the product.  Writes tests/golden/sdxl_layers.{json,safetensors} (outputs and tuple layouts only; weights are rebuilt from the seed and
guarded by a checksum).

    python oracle/make_golden_sdxl_layers.py
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import sdxl_ref                                          # noqa: E402
from oracle.make_golden_reflogic import lift                         # noqa: E402
from oracle.make_golden_wan_model import lift_classes                # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
SEED = 1
CASES = {'hw32': (32, 32), 'hw18x22_forward_upsample_size': (18, 22)}


def tiny_config():
    from diffusion_pipe_amd.workloads import sdxl
    return sdxl.tiny_config()


def weight_checksum(ref):
    return float(sum(p.detach().double().abs().sum() for p in ref.parameters()))


# ------------------------------------------------------------------------------------------------- [3P] API shim
class _AttnShim(nn.Module):
    def __init__(self, tr):
        super().__init__()
        self.tr = tr

    def forward(self, hidden_states, encoder_hidden_states=None, return_dict=True):
        return (self.tr(hidden_states, encoder_hidden_states),)


class _BlockShim:
    def __init__(self, block):
        self.resnets = block.resnets
        if hasattr(block, 'attentions'):
            self.attentions = [_AttnShim(a) for a in block.attentions]
        self.downsamplers, self.upsamplers = block.downsamplers, block.upsamplers


class _TEOut:
    def __init__(self, hidden_states, pooled):
        self.hidden_states, self._first = tuple(hidden_states), pooled

    def __getitem__(self, i):
        assert i == 0
        return self._first


class _TEShim(nn.Module):
    def __init__(self, te):
        super().__init__()
        self.te = te

    def forward(self, input_ids, output_hidden_states=False):
        hidden_states, pooled = self.te(input_ids)
        return _TEOut(hidden_states, pooled)


class _Tokenizer:
    def __init__(self, c):
        self.bos_token_id, self.eos_token_id, self.pad_token_id, self.model_max_length = c.bos, c.eos, c.pad, c.max_pos


class _UNetShim:
    def __init__(self, unet):
        c = unet.cfg
        self.time_proj = lambda t: sdxl_ref.get_timestep_embedding(t, c.block_out_channels[0], True, 0)
        self.add_time_proj = lambda t: sdxl_ref.get_timestep_embedding(t, c.addition_time_embed_dim, True, 0)
        self._time_embedding, self.add_embedding = unet.time_embedding, unet.add_embedding
        self.time_embedding = lambda x, cond=None: self._time_embedding(x)
        self.time_embed_act, self.encoder_hid_proj = None, None
        self.conv_in, self.conv_norm_out, self.conv_act, self.conv_out = unet.conv_in, unet.conv_norm_out, nn.SiLU(), unet.conv_out
        self.down_blocks = [_BlockShim(b) for b in unet.down_blocks]
        self.mid_block = _BlockShim(unet.mid_block)
        self.up_blocks = [_BlockShim(b) for b in unet.up_blocks]
        self.num_upsamplers = unet.num_upsamplers

    def get_time_embed(self, sample, timestep):
        return self.time_proj(timestep.expand(sample.shape[0])).to(dtype=sample.dtype)

    def get_aug_embed(self, emb, encoder_hidden_states, added_cond_kwargs):
        text_embeds, time_ids = added_cond_kwargs['text_embeds'], added_cond_kwargs['time_ids']
        time_embeds = self.add_time_proj(time_ids.flatten()).reshape((text_embeds.shape[0], -1))
        return self.add_embedding(torch.concat([text_embeds, time_embeds], dim=-1).to(emb.dtype))

    def process_encoder_hidden_states(self, encoder_hidden_states, added_cond_kwargs):
        return encoder_hidden_states


def main():
    cfg = tiny_config()
    ref = sdxl_ref.SDXLRef(cfg, seed=SEED)
    make_contiguous, _ = lift('models/base.py', 'make_contiguous', namespace={'torch': torch})
    names = {'InitialLayer', 'DownBlockInnerLayer', 'MidBlockInnerLayer', 'UpBlockInnerLayer', 'DownsamplerLayer', 'UpsamplerLayer',
             'UnetDownBlockLayer', 'UnetMidBlockLayer', 'UnetUpBlockLayer', 'FinalLayer'}
    ns = lift_classes('models/sdxl.py', names, {'nn': nn, 'torch': torch, 'make_contiguous': make_contiguous})
    to_layers, where = lift('models/sdxl.py', 'to_layers', cls='SDXLPipeline', namespace=ns)
    pipe = type('Pipe', (), {})()
    pipe.unet = _UNetShim(ref.unet)
    pipe.text_encoder, pipe.text_encoder_2 = _TEShim(ref.text_encoder), _TEShim(ref.text_encoder_2)
    pipe.tokenizer, pipe.tokenizer_2 = _Tokenizer(cfg.te1), _Tokenizer(cfg.te2)
    owner = type('SDXLPipelineStub', (), {'diffusers_pipeline': pipe})()
    layers = to_layers(owner)

    meta = {'generated_from': {'layers': 'models/sdxl.py:654-995 (lifted)', 'to_layers': where}, 'seed': SEED, 'weight_checksum': weight_checksum(ref),
            'torch': torch.__version__, 'layer_names': [type(l).__name__ for l in layers], 'cases': {}}
    tensors = {}
    g = torch.Generator().manual_seed(8)
    for tag, (h, w) in CASES.items():
        latents = torch.randn(2, cfg.in_channels, h, w, generator=g)
        timesteps = torch.randint(0, 1000, (2,), generator=g)
        ids1 = torch.randint(1, cfg.te1.vocab - 3, (2, 75 if tag == 'hw32' else 100), generator=g)      # 100 ids: two 75-token chunks
        ids2 = torch.randint(1, cfg.te2.vocab - 3, (2, ids1.shape[1]), generator=g)
        add_time_ids = torch.tensor([[h * 8.0, w * 8.0, 0, 0, h * 8.0, w * 8.0]]).expand(2, -1).contiguous()
        x = (latents.clone(), timesteps, ids1, ids2, add_time_ids)
        layouts = []
        for i, layer in enumerate(layers):
            x = layer(x)
            layouts.append([list(t.shape) for t in x] if isinstance(x, tuple) else list(x.shape))
            if i == 0:
                tensors[f'{tag}.emb'], tensors[f'{tag}.encoder_hidden_states'] = x[2].detach().clone(), x[3].detach().clone()
        out, ts = x
        for k, v in (('latents', latents), ('timesteps', timesteps), ('ids1', ids1), ('ids2', ids2), ('add_time_ids', add_time_ids), ('out', out.detach()), ('out_ts', ts)):
            tensors[f'{tag}.{k}'] = v.clone()
        meta['cases'][tag] = {'layouts': layouts, 'forward_upsample_size': bool(layouts[0] and h % 4 != 0 or w % 4 != 0)}
    os.makedirs(OUT, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'sdxl_layers.safetensors'))
    with open(os.path.join(OUT, 'sdxl_layers.json'), 'w') as fh:
        json.dump(meta, fh)
    print(meta['layer_names'], len(layers), {k: v['forward_upsample_size'] for k, v in meta['cases'].items()}, meta['weight_checksum'])


if __name__ == '__main__':
    main()
