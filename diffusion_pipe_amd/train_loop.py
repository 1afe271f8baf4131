"""The loop around `train_batch` (train.py:864-975 of the reference): resume from the newest checkpoint (engine, optimizer, LR schedule,
loader position), one `train_batch` per step over the step's pre-pulled micro-batches, epoch bookkeeping, periodic evaluation, model
saves and training-state checkpoints, stop after `config['epochs']`.  Everything model-specific stays behind the adapter API
(`prepare_inputs`, `to_layers`, `get_loss_fn`, `get_param_groups`, `save_model` / `save_adapter`); logging back-ends are a callback."""
import torch

from . import evaluate as _eval
from .data import MicroBatchLoader, get_data_iterator_for_step
from .saver import Saver


def run_training(model, model_engine, pipeline_model, train_data, config, save_root, args=None, eval_data=None, is_adapter=False,
                 resume=False, log=None, max_steps=None):
    """-> {'step': last completed step, 'epoch': ..., 'losses': [...], 'evals': {step: {...}}}.

    `model`: the adapter (`prepare_inputs`, `save_model` / `save_adapter`); `train_data` / `eval_data[name]`: re-iterable datasets of collated
    batches; `config` keys as in the reference's TOML: epochs, eval_every_n_steps / _epochs, eval_before_first_step,
    eval_gradient_accumulation_steps, save_every_n_steps / _epochs, checkpoint_every_n_epochs / _minutes, force_constant_lr."""
    log = log or (lambda name, value, step: None)
    gas = model_engine.gradient_accumulation_steps()
    global_batch = model_engine.train_micro_batch_size_per_gpu() * gas * model_engine.grid.get_data_parallel_world_size()
    loader = MicroBatchLoader(train_data, model_engine, gas, model.prepare_inputs)
    optimizer = model_engine.optimizer
    step, examples = 1, global_batch
    if resume:                                                    # train.py:868-890
        load_path, client_state = model_engine.load_checkpoint(str(save_root), load_module_strict=False,
                                                               load_lr_scheduler_states='force_constant_lr' not in config)
        assert load_path is not None, f'no checkpoint under {save_root}'
        loader.load_state_dict(client_state['custom_loader'])
        step = client_state['step'] + 1
        examples = client_state.get('examples', client_state['step'] * global_batch) + global_batch
    if 'force_constant_lr' in config:
        model_engine.lr_scheduler = torch.optim.lr_scheduler.ConstantLR(optimizer, factor=1.0)
        for pg in optimizer.param_groups:
            pg['lr'] = config['force_constant_lr']
    eval_gas = config.get('eval_gradient_accumulation_steps', 1)
    eval_loaders = {name: MicroBatchLoader(ds, model_engine, eval_gas, model.prepare_inputs) for name, ds in (eval_data or {}).items()}
    saver = Saver(args, config, is_adapter, save_root, model, loader, model_engine, pipeline_model)
    if max_steps is None and 'max_steps' in config:
        max_steps = config['max_steps']
    out = {'losses': [], 'evals': {}}

    def run_eval(at):
        if eval_loaders:
            out['evals'][at] = _eval.evaluate(model_engine, eval_loaders, eval_gas)
            for k, v in out['evals'][at].items():
                log(k, v, at)
    epoch = loader.epoch
    if config.get('eval_before_first_step') and not resume:
        run_eval(0)
    epoch_loss, num_steps = 0.0, 0
    while True:
        model_engine.reset_activation_shape()
        loss = model_engine.train_batch(get_data_iterator_for_step(loader, model_engine)).item()
        out['losses'].append(loss)
        epoch_loss, num_steps = epoch_loss + loss, num_steps + 1
        loader.sync_epoch()
        new_epoch, checkpointed, saved = saver.process_epoch(epoch, step, examples)
        finished_epoch = new_epoch != epoch
        log('train/loss', loss, step)
        every_steps, every_epochs = config.get('eval_every_n_steps'), config.get('eval_every_n_epochs')
        if (every_steps and step % every_steps == 0) or (finished_epoch and every_epochs and epoch % every_epochs == 0):
            run_eval(step)
        if finished_epoch:
            log('train/epoch_loss', epoch_loss / num_steps, epoch)
            epoch_loss, num_steps = 0.0, 0
            if new_epoch is None:
                final_model_name = f'epoch{epoch}'
                break
            epoch = new_epoch
        checkpointed, saved = saver.process_step(step, examples)
        if max_steps is not None and step >= max_steps:
            final_model_name = f'step{step}'
            break
        step += 1
        examples += global_batch
    # final training-state checkpoint and model, unless the last iteration just wrote them (train.py:969-973)
    if not checkpointed:
        saver.save_checkpoint(step, examples)
    if not saved:
        saver.save_model(final_model_name)
    out.update(step=step, epoch=epoch, final_model_name=final_model_name)
    return out
