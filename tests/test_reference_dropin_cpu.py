"""Drop-in check of the engine boundary: functions and classes of the REFERENCE'S training loop, lifted out of
/root/reference at test time with `ast` and executed unmodified, drive THIS repository's engine, loader and module -- the reference's
own `evaluate_single` / `get_data_iterator_for_step` (train.py:167-195) and its `Saver` (utils/saver.py:47-128: save_full_model,
save_adapter, save_checkpoint).  Skipped where the reference tree is absent (the GPU box); CPU only."""
import os

import pytest
import torch
from torch import nn

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')

from diffusion_pipe_amd import data as dpdata                          # noqa: E402
from diffusion_pipe_amd import evaluate as dpeval                      # noqa: E402
from diffusion_pipe_amd.engine import ManualPipelineModule, initialize  # noqa: E402
from oracle import eager_step as oracle                                # noqa: E402

D = 8


class Toy(nn.Module):
    def __init__(self, first=False):
        super().__init__()
        self.lin, self.first = nn.Linear(D, D), first

    def forward(self, x):
        if self.first:
            x = x[0] if isinstance(x, tuple) else x
        return torch.tanh(self.lin(x))


def _engine(trainable_only_last=False):
    torch.manual_seed(0)
    layers = [Toy(first=True), Toy(), Toy()]
    for i, layer in enumerate(layers):
        for n, p in layer.named_parameters():
            p.original_name = f'blocks.{i}.{n}'
            if trainable_only_last and i < 2:
                p.requires_grad_(False)
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=oracle.default_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': 2, 'gradient_clipping': 1.0},
                                 device='cpu')
    engine.grad_kernels = oracle.TorchGradKernels
    engine._configure_optimizer(lambda ps: torch.optim.AdamW(ps, lr=1e-2), [p for p in module.parameters() if p.requires_grad])
    return engine, module, layers


def _dataset():
    g = torch.Generator().manual_seed(4)
    return [{'x': torch.randn(4, D, generator=g)} for _ in range(3)]


def _prepare_inputs(batch, timestep_quantile=None):
    q = 0.5 if timestep_quantile is None else timestep_quantile
    noise = torch.randn(batch['x'].shape) if timestep_quantile is None else torch.full_like(batch['x'], q)
    return (batch['x'] * (1 - q) + noise * q,), (noise - batch['x'], None)


def test_reference_evaluate_single_drives_this_engine():
    from oracle.make_golden_reflogic import lift
    get_it, _ = lift('train.py', 'get_data_iterator_for_step')
    ref_eval, where = lift('train.py', 'evaluate_single', namespace={'get_data_iterator_for_step': get_it})
    assert where.startswith('train.py:')
    engine, module, layers = _engine()
    loader = dpdata.MicroBatchLoader(_dataset(), engine, 2, _prepare_inputs)
    for q in (0.1, 0.5, 0.9):
        want = ref_eval(engine, loader, 2, q)                          # the reference's own loop over this engine / loader
        got = dpeval.evaluate_single(engine, loader, 2, q)
        assert got == pytest.approx(want, rel=1e-7)
        # and both equal the oracle's mean loss over the same micro-batches
        loader.set_eval_quantile(q)
        micro = [next(iter(loader)) for _ in range(6)]
        loader.reset()
        per_step = [oracle.eager_eval(layers, oracle.default_loss_fn(), micro[i:i + 2]).item() for i in (0, 2, 4)]
        assert got == pytest.approx(sum(per_step) / 3, rel=1e-6)
    torch.manual_seed(123)
    before = torch.random.get_rng_state()
    out = dpeval.evaluate(engine, {'eval0': loader}, 2)
    assert torch.equal(torch.random.get_rng_state(), before)          # training RNG stream untouched
    assert len(out) == 10 and out['eval0/loss'] == pytest.approx(sum(out[f'eval0/loss_quantile_{q:.2f}'] for q in dpeval.TIMESTEP_QUANTILES_FOR_EVAL) / 9)


def test_reference_saver_drives_this_engine(tmp_path):
    from oracle.make_golden_reflogic import lift, lift_classes
    conv, _ = lift('utils/saver.py', 'convert_state_dict_dtype')
    dist_stub = type('dist', (), {'barrier': staticmethod(lambda: None)})
    logger = type('L', (), {'warning': staticmethod(lambda *a: None)})
    ns = lift_classes('utils/saver.py', {'Saver'}, {'Path': __import__('pathlib').Path, 'os': os, 'shutil': __import__('shutil'), 'torch': torch,
                                                   'dist': dist_stub, 'logger': logger, 'convert_state_dict_dtype': conv, 'is_main_process': lambda: True,
                                                   'print': lambda *a, **k: None})
    Saver = ns['Saver']
    saved = {}

    class Model:                                                          # the adapter hooks the Saver calls
        def save_model(self, save_dir, state_dict):
            saved['model'] = dict(state_dict)

        def save_adapter(self, save_dir, state_dict):
            saved['adapter'] = dict(state_dict)
    cfg_file = tmp_path / 'run.toml'
    cfg_file.write_text('x = 1\n')
    args = type('A', (), {'config': str(cfg_file)})

    # full fine-tune: every parameter, keyed by original_name, cast to save_dtype
    engine, module, layers = _engine()
    loader = dpdata.MicroBatchLoader(_dataset(), engine, 2, _prepare_inputs)
    s = Saver(args, {'save_dtype': torch.bfloat16}, False, tmp_path / 'out', Model(), loader, engine, module)
    s.save_model('step1')
    want = {p.original_name: p for p in module.parameters()}
    assert saved['model'].keys() == want.keys() and all(v.dtype == torch.bfloat16 for v in saved['model'].values())
    assert all(torch.equal(saved['model'][k], want[k].detach().to(torch.bfloat16)) for k in want)
    assert (tmp_path / 'out' / 'step1' / 'run.toml').exists() and not (tmp_path / 'out' / 'step1' / 'tmp').exists()

    # adapter: only trainable parameters
    engine2, module2, _ = _engine(trainable_only_last=True)
    s2 = Saver(args, {}, True, tmp_path / 'out2', Model(), loader, engine2, module2)
    s2.save_model('epoch1')
    assert set(saved['adapter']) == {'blocks.2.lin.weight', 'blocks.2.lin.bias'}

    # training-state checkpoint through the reference's call (save_latest, exclude_frozen_parameters, client_state with the loader state)
    it = iter(loader)
    engine.train_batch(iter([next(it), next(it)]))
    s.save_checkpoint(step=7, examples=28)
    engine3, module3, _ = _engine()
    path, client = engine3.load_checkpoint(str(tmp_path / 'out'))
    assert client['step'] == 7 and client['examples'] == 28 and client['custom_loader'] == loader.state_dict()
    for a, b in zip(module3.parameters(), module.parameters()):
        assert torch.equal(a, b)
    assert engine3.optimizer.state_dict()['state'].keys() == engine.optimizer.state_dict()['state'].keys()


def test_reference_clip_grad_norm_as_the_engine_clip_override():
    """`engine.clip_grad_fn` (the hook the reference fills with its patched clip_grad_norm_, utils/patches.py:175-246,429): the lifted
    reference function, given this engine's grid as `mpu`, produces the same step as the engine's own kernels-based clip."""
    import math
    from oracle.make_golden_reflogic import lift
    acc = type('Acc', (), {'current_device_name': staticmethod(lambda: 'cpu'), 'FloatTensor': staticmethod(lambda v: torch.tensor(v, dtype=torch.float32))})
    dist_stub = type('dist', (), {'all_reduce': staticmethod(lambda t, op=None, group=None: None), 'get_world_size': staticmethod(lambda group=None: 1),
                                  'ReduceOp': type('R', (), {'MAX': 0, 'SUM': 1})})
    groups_stub = type('groups', (), {'_get_data_parallel_group': staticmethod(lambda: None)})
    ds_stub = type('deepspeed', (), {'runtime': type('rt', (), {'utils': type('u', (), {'is_model_parallel_parameter': staticmethod(lambda p: False)})})})
    ref_clip, _ = lift('utils/patches.py', 'clip_grad_norm_', namespace={'torch': torch, 'inf': math.inf, 'get_accelerator': lambda: acc, 'dist': dist_stub,
                                                                        'groups': groups_stub, 'deepspeed': ds_stub})
    results = []
    for use_reference in (False, True):
        engine, module, layers = _engine()
        engine._gradient_clipping = 0.05
        if use_reference:
            engine.clip_grad_fn = ref_clip
        loader = dpdata.MicroBatchLoader(_dataset(), engine, 2, lambda b, timestep_quantile=None: _prepare_inputs(b, 0.3))
        it = iter(loader)
        loss = engine.train_batch(iter([next(it), next(it)])).item()
        results.append((loss, float(engine.get_global_grad_norm()), [p.detach().clone() for p in module.parameters()]))
    (l0, n0, p0), (l1, n1, p1) = results
    assert l0 == pytest.approx(l1, rel=1e-7) and n0 == pytest.approx(n1, rel=1e-6) and n0 > 0.05
    for a, b in zip(p0, p1):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)


def test_reference_patched_schedule_runs_on_this_train_schedule_class():
    """utils/patches.py:419 assigns `TrainSchedule.steps = train_schedule_steps`: the lifted function, bound to THIS repository's
    TrainSchedule (its index helpers and instruction classes), yields exactly the engine's native instruction stream."""
    from diffusion_pipe_amd.engine import schedule as ps
    from oracle.make_golden_reflogic import lift
    names = ('LoadMicroBatch', 'ForwardPass', 'BackwardPass', 'SendActivation', 'RecvActivation', 'SendGrad', 'RecvGrad', 'ReduceTiedGrads', 'ReduceGrads',
             'OptimizerStep')
    fn, _ = lift('utils/patches.py', 'train_schedule_steps', namespace={n: getattr(ps, n) for n in names})
    for stages, mbs in ((1, 3), (2, 4), (4, 8), (8, 16), (3, 2)):
        for stage in range(stages):
            native = [list(step) for step in ps.TrainSchedule(mbs, stages, stage).steps()]
            patched = [list(step) for step in fn(ps.TrainSchedule(mbs, stages, stage))]
            assert patched == native, (stages, mbs, stage)


def test_reference_wan_adapter_layers_train_through_this_engine():
    """The reference's own WanModel (imported) and pipeline layers (lifted) as the `layers=` of this repository's ManualPipelineModule:
    one train_batch on CPU reproduces the golden loss and gradient norm of the reference's plain eager step (tests/golden/wan_model_fp32).
    Layers that stay PyTorch modules run unchanged -- the engine only needs nn.Modules / callables (SURVEY.md 8(b) B1 conventions)."""
    import json
    from safetensors.torch import load_file
    from oracle.make_golden import import_reference_wan
    from oracle.make_golden_reflogic import lift, lift_classes
    from oracle.make_golden_wan_model import CFG
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'wan_model_fp32')
    g, meta = load_file(base + '.safetensors'), json.load(open(base + '.json'))
    m = import_reference_wan()
    model = m.WanModel(**CFG).float()
    model.load_state_dict({k[len('param.'):]: v for k, v in g.items() if k.startswith('param.')}, strict=False)
    make_contiguous, _ = lift('models/base.py', 'make_contiguous', namespace={'torch': torch})
    ns = lift_classes('models/wan/wan.py', {'InitialLayer', 'TransformerLayer', 'FinalLayer'},
                      {'nn': nn, 'torch': torch, 'make_contiguous': make_contiguous, 'sinusoidal_embedding_1d': m.sinusoidal_embedding_1d})
    to_layers, _ = lift('models/wan/wan.py', 'to_layers', cls='WanPipeline', namespace=ns)
    pipe = type('Pipe', (), {})()
    pipe.transformer, pipe.cache_text_embeddings = model, True
    pipe.offloader = type('Off', (), {'wait_for_block': staticmethod(lambda i: None), 'submit_move_blocks_forward': staticmethod(lambda i: None)})
    layers = to_layers(pipe)
    loss_fn, _ = lift('models/base.py', 'get_loss_fn', cls='BasePipeline', namespace={'torch': torch, 'F': torch.nn.functional})
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='parameters', loss_fn=loss_fn(type('S', (), {'config': {}})()),
                                  dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': 1, 'gradient_clipping': 1e9},
                                 device='cpu')
    engine.grad_kernels = oracle.TorchGradKernels
    engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=0.0), [p for p in module.parameters()])
    empty = torch.tensor([])
    micro = ((g['prep.x_t'], empty, g['prep.t'], g['in.text_embeddings'], g['in.seq_lens'], empty), (g['prep.target'], empty))
    loss = engine.train_batch(iter([micro])).item()
    assert loss == pytest.approx(meta['loss'], rel=1e-6)
    want_norm = sum(v.double().pow(2).sum().item() for k, v in g.items() if k.startswith('grad.')) ** 0.5
    assert float(engine.get_global_grad_norm()) == pytest.approx(want_norm, rel=1e-5)


def test_product_saver_writes_what_the_reference_saver_writes(tmp_path):
    """diffusion_pipe_amd.saver.Saver next to the reference's own Saver (lifted) over the same engine: identical merged state dicts handed
    to the adapter hooks, the same files on disk, the same epoch / step bookkeeping."""
    from diffusion_pipe_amd.saver import Saver as Mine
    from oracle.make_golden_reflogic import lift, lift_classes
    conv, _ = lift('utils/saver.py', 'convert_state_dict_dtype')
    ns = lift_classes('utils/saver.py', {'Saver'}, {'Path': __import__('pathlib').Path, 'os': os, 'shutil': __import__('shutil'), 'torch': torch, 'sys': __import__('sys'),
                                                   'dist': type('dist', (), {'barrier': staticmethod(lambda: None)}), 'logger': type('L', (), {'warning': staticmethod(lambda *a: None)}),
                                                   'convert_state_dict_dtype': conv, 'is_main_process': lambda: True, 'print': lambda *a, **k: None,
                                                   'need_to_checkpoint': lambda config, epoch=None: epoch is not None and epoch % config.get('checkpoint_every_n_epochs', 10 ** 9) == 0})
    cfg_file = tmp_path / 'run.toml'
    cfg_file.write_text('x = 1\n')
    args = type('A', (), {'config': str(cfg_file)})
    config = {'save_dtype': torch.bfloat16, 'save_every_n_epochs': 1, 'checkpoint_every_n_epochs': 2, 'epochs': 3, 'save_every_n_steps': 5}
    outs = {}
    for tag, klass in (('ref', ns['Saver']), ('mine', Mine)):
        got = {}
        model = type('M', (), {'save_model': lambda self, d, sd, got=got: got.update(model=dict(sd)), 'save_adapter': lambda self, d, sd, got=got: got.update(adapter=dict(sd))})()
        engine, module, _ = _engine(trainable_only_last=True)
        loader = dpdata.MicroBatchLoader(_dataset(), engine, 2, _prepare_inputs)
        for is_adapter in (False, True):
            s = klass(args, config, is_adapter, tmp_path / f'{tag}{int(is_adapter)}', model, loader, engine, module)
            s.save_model('final')
            it = iter(loader)
            for _ in range(6):                                  # one full pass: the loader rolls over to epoch 2
                next(it)
            got[f'epoch{int(is_adapter)}'] = s.process_epoch(1, 4, 16)
            got[f'step{int(is_adapter)}'] = s.process_step(5, 20)
            loader.reset()
        got['files'] = sorted(str(p.relative_to(tmp_path)).replace(tag, 'X', 1) for p in tmp_path.rglob('*') if p.is_file() and str(p.relative_to(tmp_path)).startswith(tag))
        outs[tag] = got
    ref, mine = outs['ref'], outs['mine']
    assert ref['model'].keys() == mine['model'].keys() and all(torch.equal(ref['model'][k], mine['model'][k]) for k in ref['model'])
    assert ref['adapter'].keys() == mine['adapter'].keys() == {'blocks.2.lin.weight', 'blocks.2.lin.bias'}
    for k in ('epoch0', 'epoch1', 'step0', 'step1'):
        assert ref[k] == mine[k], k
    assert ref['files'] == mine['files'] and len(mine['files']) >= 4          # config copies of the saved models, no tmp/ left behind
