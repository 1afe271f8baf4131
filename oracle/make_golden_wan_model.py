"""ORACLE fixture generator (test infrastructure): the Wan t2v WHOLE-MODEL training step from the reference's own code --
`models/wan/model.py:WanModel` imported unmodified (shims of oracle/make_golden.py) and driven through the reference's own pipeline
layers `InitialLayer / TransformerLayer / FinalLayer` and `WanPipeline.to_layers / prepare_inputs` (models/wan/wan.py:332-384,414-546),
which are lifted out of the file with `ast` at generation time (the module itself imports deepspeed-dependent code) -- then the
default loss (models/base.py:418-436) and a backward pass.  Writes tests/golden/wan_model_fp32.safetensors (+ .json): the weights,
the inputs prepare_inputs produced, the output of every pipeline layer boundary that matters (final output), the loss and every
parameter gradient.  Pins oracle/blocks_ref.py:wan_forward on CPU and (next) the HIP-kernel model on the MI355X.

    python oracle/make_golden_wan_model.py
"""
import json
import os
import sys

import torch
import torch.nn.functional as F
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden import REF, import_reference_wan          # noqa: E402
from oracle.make_golden_reflogic import lift, lift_classes         # noqa: E402, F401

OUT = os.path.join(HERE, '..', 'tests', 'golden')
CFG = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=24, in_dim=16, dim=128, ffn_dim=256, freq_dim=256, text_dim=64, out_dim=16,
           num_heads=2, num_layers=2, window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6)


def main():
    m = import_reference_wan()
    torch.manual_seed(4321)
    model = m.WanModel(**CFG).float()
    with torch.no_grad():                      # the reference zero-initialises nothing here, but make every tensor informative
        for n, p in model.named_parameters():
            if p.abs().sum() == 0:
                p.normal_(0, 0.02)
    make_contiguous, _ = lift('models/base.py', 'make_contiguous', namespace={'torch': torch})
    ns = lift_classes('models/wan/wan.py', {'InitialLayer', 'TransformerLayer', 'FinalLayer'},
                      {'nn': torch.nn, 'torch': torch, 'make_contiguous': make_contiguous, 'sinusoidal_embedding_1d': m.sinusoidal_embedding_1d})
    to_layers, where_layers = lift('models/wan/wan.py', 'to_layers', cls='WanPipeline', namespace=ns)
    gtd, _ = lift('utils/common.py', 'get_t_distribution', namespace={'torch': torch})
    std, _ = lift('utils/common.py', 'slice_t_distribution', namespace={'torch': torch})
    smp, _ = lift('utils/common.py', 'sample_t', namespace={'torch': torch})
    prep, where_prep = lift('models/wan/wan.py', 'prepare_inputs', cls='WanPipeline', namespace={
        'torch': torch, 'F': F, 'slice_t_distribution': std, 'sample_t': smp})
    loss_factory, where_loss = lift('models/base.py', 'get_loss_fn', cls='BasePipeline', namespace={'torch': torch, 'F': F})

    pipe = type('Pipe', (), {})()
    pipe.transformer, pipe.cache_text_embeddings, pipe.model_type, pipe.model_config = model, True, 't2v', {}
    pipe.t_dist = gtd({})
    pipe.offloader = type('Off', (), {'wait_for_block': staticmethod(lambda i: None), 'submit_move_blocks_forward': staticmethod(lambda i: None)})
    pipe.config = {}
    layers = to_layers(pipe)

    g = torch.Generator().manual_seed(99)
    batch = {'latents': torch.randn(2, 16, 2, 12, 16, generator=g), 'mask': None,
             'text_embeddings': torch.randn(2, 20, CFG['text_dim'], generator=g), 'seq_lens': torch.tensor([17, 20])}
    torch.manual_seed(5)
    features, (target, mask) = prep(pipe, batch)
    # the pipeline hands None through as empty tensors (utils/dataset.py:1277-1279)
    x = tuple(torch.tensor([]) if t is None else t for t in features)
    for layer in layers:
        x = layer(x)
    out = x
    loss = loss_factory(pipe)(out, (target, torch.tensor([])))
    loss.backward()

    tensors = {f'param.{n}': p.detach().clone() for n, p in model.named_parameters()}
    tensors.update({f'grad.{n}': p.grad.detach().clone() for n, p in model.named_parameters()})
    tensors.update({'in.latents': batch['latents'], 'in.text_embeddings': batch['text_embeddings'], 'in.seq_lens': batch['seq_lens'],
                    'prep.x_t': features[0], 'prep.t': features[2], 'prep.target': target, 'out': out.detach(), 'loss': loss.detach().reshape(1)})
    os.makedirs(OUT, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'wan_model_fp32.safetensors'))
    meta = {'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in CFG.items()}, 'torch': torch.__version__,
            'generated_from': {'model': 'models/wan/model.py (imported)', 'layers': 'models/wan/wan.py:414-546 (lifted)', 'to_layers': where_layers,
                               'prepare_inputs': where_prep, 'loss': where_loss}, 'num_layers': len(layers), 'loss': float(loss),
            'params': len(list(model.parameters())), 'seed_prepare_inputs': 5}
    with open(os.path.join(OUT, 'wan_model_fp32.json'), 'w') as fh:
        json.dump(meta, fh, indent=1)
    print(meta)


if __name__ == '__main__':
    main()
