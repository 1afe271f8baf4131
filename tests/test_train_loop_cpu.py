"""The loop around train_batch (diffusion_pipe_amd/train_loop.py; train.py:864-975) on CPU with plain PyTorch layers: epochs end where the
dataset wraps, evaluation / save / checkpoint cadence, and -- the property that matters -- stopping after a checkpoint and resuming
reproduces the uninterrupted run exactly (parameters, optimizer state, LR schedule, loader position)."""
import torch
from torch import nn

from diffusion_pipe_amd import optim
from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
from diffusion_pipe_amd.train_loop import run_training
from oracle import eager_step as oracle

D = 8


class Layer(nn.Module):
    def __init__(self, first=False):
        super().__init__()
        self.lin, self.first = nn.Linear(D, D), first

    def forward(self, x):
        if self.first:
            x = x[0] if isinstance(x, tuple) else x
        return torch.tanh(self.lin(x))


class Adapter:
    def __init__(self):
        self.saved = []
        self.gen = torch.Generator().manual_seed(0)

    def prepare_inputs(self, batch, timestep_quantile=None):
        q = 0.5 if timestep_quantile is None else timestep_quantile
        return (batch['x'] * (1 - q),), (batch['y'], None)

    def get_param_groups(self, params):
        return [{'params': list(params)}]

    def save_model(self, save_dir, state_dict):
        self.saved.append((str(save_dir), sorted(state_dict)))


def _setup(tmp_path, tag):
    torch.manual_seed(0)
    layers = [Layer(first=True), Layer(), Layer()]
    for i, l in enumerate(layers):
        for n, p in l.named_parameters():
            p.original_name = f'blocks.{i}.{n}'
    module = ManualPipelineModule(layers=layers, num_stages=1, partition_method='uniform', loss_fn=oracle.default_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 2, 'gradient_accumulation_steps': 2, 'gradient_clipping': 1.0},
                                 device='cpu')
    engine.grad_kernels = oracle.TorchGradKernels
    adapter = Adapter()
    cfg = {'optimizer': {'type': 'adamw', 'lr': 5e-3, 'betas': [0.9, 0.99], 'weight_decay': 0.01}}
    engine._configure_optimizer(optim.make_optimizer_factory(cfg, adapter, 4, device_is_gpu=False), [p for p in module.parameters()])
    engine.lr_scheduler = optim.make_lr_scheduler(engine.optimizer, {'lr_scheduler': 'linear', 'epochs': 3, 'warmup_steps': 2}, steps_per_epoch=3)
    g = torch.Generator().manual_seed(1)
    train = [{'x': torch.randn(4, D, generator=g), 'y': torch.randn(4, D, generator=g)} for _ in range(3)]          # 3 steps per epoch
    evald = {'eval0': [{'x': torch.randn(2, D, generator=g), 'y': torch.randn(2, D, generator=g)} for _ in range(2)]}
    return adapter, engine, module, train, evald, tmp_path / tag


def test_epochs_eval_and_save_cadence(tmp_path):
    adapter, engine, module, train, evald, root = _setup(tmp_path, 'a')
    logged = []
    config = {'epochs': 2, 'eval_every_n_epochs': 1, 'eval_before_first_step': True, 'eval_gradient_accumulation_steps': 1, 'save_every_n_epochs': 1,
              'checkpoint_every_n_epochs': 1}
    out = run_training(adapter, engine, module, train, config, root, eval_data=evald, log=lambda n, v, s: logged.append((n, s)))
    assert out['step'] == 6 and len(out['losses']) == 6                                  # 2 epochs x 3 steps, stops when epoch 3 would start
    assert sorted(out['evals']) == [0, 3, 6] and len(out['evals'][3]) == 10            # before the first step + once per finished epoch; 9 quantiles + mean
    assert [s.rsplit('/', 1)[-1] for s, _ in adapter.saved] == ['epoch1', 'epoch2'] and len(adapter.saved[0][1]) == 6
    assert (root / 'latest').exists() and ('train/epoch_loss', 1) in logged and ('train/epoch_loss', 2) in logged
    assert out['losses'][-1] < out['losses'][0]


def test_resume_reproduces_the_uninterrupted_run(tmp_path):
    config = {'epochs': 3, 'checkpoint_every_n_minutes': 0}             # time-based checkpoints with a zero interval: one after every step but the first
    adapter, engine, module, train, _, root = _setup(tmp_path, 'full')
    full = run_training(adapter, engine, module, train, {'epochs': 3}, root)
    want = [p.detach().clone() for p in module.parameters()]
    assert full['step'] == 9
    # first process: stops after step 4 (first step of epoch 2); its last checkpoint holds the engine, optimizer, schedule and loader position
    adapter1, engine1, module1, train1, _, root1 = _setup(tmp_path, 'part')
    part = run_training(adapter1, engine1, module1, train1, config, root1, max_steps=4)
    assert part['losses'] == full['losses'][:4]
    # second process: fresh engine, resume
    adapter2, engine2, module2, train2, _, _ = _setup(tmp_path, 'part')
    rest = run_training(adapter2, engine2, module2, train2, {'epochs': 3}, root1, resume=True)
    assert rest['step'] == 9 and rest['losses'] == full['losses'][4:]
    for a, b in zip(module2.parameters(), want):
        assert torch.equal(a, b)
    assert engine2.lr_scheduler.get_last_lr() == engine.lr_scheduler.get_last_lr()


def test_resume_from_an_epoch_boundary_checkpoint_keeps_the_reference_quirk(tmp_path):
    """A checkpoint written by process_epoch stores the loader right after its roll-over (num_batches_pulled == 0); load_state_dict turns
    that into skip = -1 and the reference's sampler then starts at dataset index -1 -- one replayed batch (pinned against the reference's
    own PipelineDataLoader in tests/test_reflogic_cpu.py).  The loop keeps that behaviour: the resumed epoch has one extra step."""
    config = {'epochs': 2, 'checkpoint_every_n_epochs': 1}
    adapter1, engine1, module1, train1, _, root1 = _setup(tmp_path, 'b')
    part = run_training(adapter1, engine1, module1, train1, config, root1, max_steps=3)
    assert part['step'] == 3
    adapter2, engine2, module2, train2, _, _ = _setup(tmp_path, 'b')
    rest = run_training(adapter2, engine2, module2, train2, config, root1, resume=True)
    assert len(rest['losses']) == 4 and rest['step'] == 7              # batch -1 (= the last one) once more, then the epoch's three batches
