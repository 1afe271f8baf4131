"""`diffusion_pipe_amd.adopt` on CPU: structure only (the substitutes' arithmetic needs the GPU: tests/test_gpu_adopt.py).  (1) The REFERENCE's own Wan model
and pipeline layers (imported / lifted from /root/reference, skipped where absent): every nn.Linear / WanRMSNorm / WanLayerNorm / WanAttentionBlock / Head under
`to_layers()` is substituted, the parameters are the same objects, the state dict is unchanged.  (2) The same on the stand-in tree (runs everywhere)."""
import os

import pytest
import torch
from torch import nn

from diffusion_pipe_amd import adopt as adopt_mod
from diffusion_pipe_amd import nn as dnn
from diffusion_pipe_amd.workloads import wan as pwan

REF = '/root/reference'


def _check_tree(root_modules, params_before, sd_before, sd_after):
    left = [(n, type(m).__name__) for r in root_modules for n, m in r.named_modules()
            if type(m) in (nn.Linear, nn.LayerNorm, nn.GroupNorm, nn.GELU, nn.SiLU) or type(m).__name__ in ('WanRMSNorm', 'WanLayerNorm')]
    assert left == [], left
    params_after = {id(p) for r in root_modules for p in r.parameters()}
    assert params_after == params_before                                      # the very same Parameter objects: nothing copied, nothing dropped
    assert list(sd_before) == list(sd_after)
    assert all(sd_before[k].data_ptr() == sd_after[k].data_ptr() for k in sd_before)


def test_adopt_standin_tree_shares_every_parameter():
    from tests import wan_standin as ws
    torch.manual_seed(0)
    blocks = nn.ModuleList([ws.WanAttentionBlock('default', 64, 128, 4, cross_attn_norm=True), ws.WanAttentionBlock('default', 64, 128, 4, cross_attn_norm=False)])
    head = ws.Head(64, 4, (1, 2, 2))
    layers = [ws.TransformerLayer(b) for b in blocks] + [head]
    model = nn.ModuleList([blocks, head])
    before = {id(p) for p in model.parameters()}
    sd0 = model.state_dict()
    report = adopt_mod.adopt(layers)
    assert report['layers.0.block'] == ('WanAttentionBlock', 'AdoptedWanAttentionBlock')
    assert isinstance(layers[0].block, pwan.WanAttentionBlock) and layers[1].block.norm3 is None
    assert type(layers[0].block.self_attn.q) is dnn.Linear and type(layers[0].block.self_attn.norm_q) is dnn.RMSNorm
    # the layer list's root Head is a root: it is left alone (a root cannot be replaced in place); its leaves are substituted
    assert type(layers[2].head) is dnn.Linear and type(layers[2].norm) is dnn.LayerNorm
    _check_tree([nn.ModuleList(layers)], before, sd0, nn.ModuleList([nn.ModuleList([l.block for l in layers[:2]]), layers[2]]).state_dict())
    # the product path fails loudly without the GPU: no CPU fallback behind a substitute
    from diffusion_pipe_amd.hip import DpipeHipError
    with pytest.raises((DpipeHipError, RuntimeError)):
        layers[0].block.self_attn.q(torch.randn(2, 64))
    # second call: nothing left to do
    assert adopt_mod.adopt(layers) == {}


def test_adopt_leaf_rules():
    conv_ok, conv_grouped = nn.Conv2d(8, 8, 3, padding=1), nn.Conv2d(8, 8, 3, padding=1, groups=2)

    class MyLinear(nn.Linear):
        pass
    m = nn.ModuleDict({'a': nn.Linear(4, 4), 'b': MyLinear(4, 4), 'c': conv_ok, 'd': conv_grouped, 'e': nn.GroupNorm(2, 8), 'f': nn.LayerNorm([4, 4]), 'g': nn.SiLU()})
    w = conv_ok.weight
    rep = adopt_mod.adopt(m)
    assert set(rep) == {'a', 'c', 'e', 'g'}
    assert type(m['b']) is MyLinear and m['d'] is conv_grouped and type(m['f']) is nn.LayerNorm      # subclasses / unsupported shapes stay what they are
    assert m['c'].weight is w and w.permute(0, 2, 3, 1).is_contiguous()


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_adopt_reference_wan_layers():
    from oracle.make_golden import import_reference_wan
    from oracle.make_golden_reflogic import lift, lift_classes
    from oracle.make_golden_wan_model import CFG
    m = import_reference_wan()
    torch.manual_seed(0)
    model = m.WanModel(**CFG).float()
    make_contiguous, _ = lift('models/base.py', 'make_contiguous', namespace={'torch': torch})
    ns = lift_classes('models/wan/wan.py', {'InitialLayer', 'TransformerLayer', 'FinalLayer'},
                      {'nn': nn, 'torch': torch, 'make_contiguous': make_contiguous, 'sinusoidal_embedding_1d': m.sinusoidal_embedding_1d})
    to_layers, _ = lift('models/wan/wan.py', 'to_layers', cls='WanPipeline', namespace=ns)
    pipe = type('Pipe', (), {})()
    pipe.transformer, pipe.cache_text_embeddings = model, True
    pipe.offloader = type('Off', (), {'wait_for_block': staticmethod(lambda i: None), 'submit_move_blocks_forward': staticmethod(lambda i: None)})
    layers = to_layers(pipe)
    before = {id(p) for p in model.parameters()}
    names_before = {id(p): n for n, p in model.named_parameters()}
    sd0 = {k: v for k, v in model.state_dict().items()}
    report = adopt_mod.adopt(layers)
    n_blocks = len(model.blocks)
    assert sum(1 for v in report.values() if v == ('WanAttentionBlock', 'AdoptedWanAttentionBlock')) == n_blocks
    assert ('Head', 'AdoptedWanHead') in report.values()
    root = nn.ModuleList(layers)
    left = [(n, type(mod).__name__) for n, mod in root.named_modules() if type(mod) is nn.Linear or type(mod).__name__ in ('WanRMSNorm', 'WanLayerNorm', 'WanAttentionBlock')]
    assert left == [], left
    after = {id(p) for p in root.parameters()}
    assert after == before, sorted(names_before[i] for i in before - after)
    # the adapter's own model still reports the same state dict, over the same storage
    sd1 = model.state_dict()
    assert list(sd0) == list(sd1) and all(sd0[k].data_ptr() == sd1[k].data_ptr() for k in sd0)
    # the layers' view of the parameters carries the reference's names (block i of the model == layers[i + 1].block)
    for i in range(n_blocks):
        assert {n for n, _ in layers[i + 1].block.named_parameters()} == {n for n, _ in model.blocks[i].named_parameters()}
        assert all(p is dict(model.blocks[i].named_parameters())[n] for n, p in layers[i + 1].block.named_parameters())
