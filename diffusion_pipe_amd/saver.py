"""Model / adapter / training-state saving around the engine (utils/saver.py:47-177 of the reference).

Every pipeline stage of data-parallel replica 0 writes the parameters it owns (keyed by `original_name`) into
`<save_root>/<name>/tmp/state_dict_<stage>.bin`; after a barrier, stage 0 merges the pieces and hands one state dict to the adapter's
`save_adapter` (trainable parameters, peft's adapter name stripped) or `save_model` (every parameter), copies the run's config file next
to it and removes `tmp/`.  Training-state checkpoints go through `engine.save_checkpoint` with the loader state as client state."""
import os
import shutil
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist


def _barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def _is_main():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class Saver:
    def __init__(self, args, config, is_adapter, save_root, model, train_dataloader, model_engine, pipeline_model):
        self.args, self.config, self.is_adapter = args, config, is_adapter
        self.save_root = Path(save_root)
        self.model, self.train_dataloader = model, train_dataloader
        self.model_engine, self.pipeline_model = model_engine, pipeline_model
        self._last_checkpoint_time = None
        # fail before training starts, on every rank alike, rather than with an AttributeError after the barriers of the first save
        need = 'save_adapter' if is_adapter else 'save_model'
        if not callable(getattr(model, need, None)):
            raise NotImplementedError(f'{type(model).__name__} has no {need}(save_dir, state_dict): the run could not save its weights')
        # ... and anything else the adapter needs at save time (SDXL full fine-tune: the base checkpoint's VAE to embed) is checked NOW, not at the first
        # checkpoint hours into the run
        check = getattr(model, 'check_save_sources', None)
        if callable(check):
            check(is_adapter)

    # ------------------------------------------------------------------------------------------------ model files
    def _gather_and_save(self, name, select, finish):
        dp_id = self.model_engine.grid.get_data_parallel_rank()
        stage_id = self.model_engine.grid.get_pipe_parallel_rank()
        save_dir = self.save_root / name
        tmp_dir = save_dir / 'tmp'
        if dp_id == 0 and stage_id == 0:
            os.makedirs(tmp_dir, exist_ok=False)
        _barrier()
        if dp_id == 0:
            part = select()
            if 'save_dtype' in self.config:
                part = {k: v.to(device='cpu', dtype=self.config['save_dtype']) for k, v in part.items()}
            torch.save(part, tmp_dir / f'state_dict_{stage_id}.bin')
        _barrier()
        if dp_id == 0 and stage_id == 0:
            state_dict = {}
            for path in tmp_dir.glob('*.bin'):
                state_dict.update(torch.load(path, weights_only=True, map_location='cpu'))
            finish(save_dir, state_dict)
            if getattr(self.args, 'config', None):
                shutil.copy(self.args.config, save_dir)
            shutil.rmtree(tmp_dir)

    def save_adapter(self, name):
        def select():
            out = {}
            for _, p in self.pipeline_model.named_parameters():
                if p.requires_grad and hasattr(p, 'original_name'):
                    out[p.original_name.replace('.default', '').replace('.modules_to_save', '')] = p.detach()
            return out
        self._gather_and_save(name, select, self.model.save_adapter)

    def save_full_model(self, name):
        self._gather_and_save(name, lambda: {p.original_name: p.detach() for p in self.pipeline_model.parameters() if hasattr(p, 'original_name')},
                              self.model.save_model)

    def save_model(self, name):
        if self.is_adapter:
            self.save_adapter(name)
        else:
            self.save_full_model(name)

    # ------------------------------------------------------------------------------------------------ training state
    def save_checkpoint(self, step, examples):
        self.model_engine.save_checkpoint(self.save_root, client_state={'step': step, 'examples': examples, 'custom_loader': self.train_dataloader.state_dict()},
                                          save_latest=True, exclude_frozen_parameters=True)

    def need_to_checkpoint(self, epoch=None):
        if epoch is not None:
            if 'checkpoint_every_n_epochs' in self.config and epoch % self.config['checkpoint_every_n_epochs'] == 0:
                self._last_checkpoint_time = time.time()
                return True
            return False
        if 'checkpoint_every_n_minutes' not in self.config:
            return False
        decision = [False]
        if _is_main():                      # rank 0 owns the clock, everyone follows its decision
            now = time.time()
            if self._last_checkpoint_time is None:
                self._last_checkpoint_time = now
            elif (now - self._last_checkpoint_time) / 60 > self.config['checkpoint_every_n_minutes']:
                decision[0] = True
                self._last_checkpoint_time = now
        if dist.is_available() and dist.is_initialized():
            dist.broadcast_object_list(decision, src=0)
        return decision[0]

    def process_epoch(self, epoch, step, examples):
        checkpointed = saved = False
        if self.train_dataloader.epoch != epoch:
            if self.need_to_checkpoint(epoch):
                self.save_checkpoint(step, examples)
                checkpointed = True
            if 'save_every_n_epochs' in self.config and epoch % self.config['save_every_n_epochs'] == 0:
                self.save_model(f'epoch{epoch}')
                saved = True
            epoch = self.train_dataloader.epoch
            if epoch > self.config['epochs']:
                return None, checkpointed, saved
        return epoch, checkpointed, saved

    def process_step(self, step, examples):
        checkpointed = saved = False
        manual_save = manual_quit = False
        for signal, quits in (('save', False), ('save_quit', True)):          # files the user drops into save_root
            f = self.save_root / signal
            if f.exists() and f.is_file():
                manual_save, manual_quit = True, quits
                _barrier()
                if _is_main():
                    os.remove(f)
                break
        if 'save_every_n_steps' in self.config and step % self.config['save_every_n_steps'] == 0:
            self.save_model(f'step{step}')
            saved = True
        if self.need_to_checkpoint() or manual_save:
            self.save_checkpoint(step, examples)
            checkpointed = True
        if manual_quit:
            sys.exit()
        return checkpointed, saved
