"""Evaluation pass of the training loop (train.py:176-242): the forward-only schedule at nine fixed
timestep quantiles per evaluation dataset, RNG isolated and re-seeded per rank so every evaluation sees the same noise.

    losses = evaluate(model_engine, {'eval0': eval_loader}, eval_gradient_accumulation_steps)
    # {'eval0/loss_quantile_0.10': ..., ..., 'eval0/loss': mean over the quantiles}

`eval_loader` is a `data.MicroBatchLoader` (the PipelineDataLoader contract: set_eval_quantile / sync_epoch / epoch / reset)."""
import random
from contextlib import contextmanager

import numpy as np
import torch
import torch.distributed as dist

from .data import get_data_iterator_for_step

TIMESTEP_QUANTILES_FOR_EVAL = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]


@contextmanager
def isolate_rng():
    """Snapshot / restore the python, numpy and torch (CPU + current device) generators (utils/isolate_rng.py)."""
    py, npy, cpu = random.getstate(), np.random.get_state(), torch.random.get_rng_state()
    dev = torch.cuda.get_rng_state() if torch.cuda.is_available() else None
    try:
        yield
    finally:
        random.setstate(py)
        np.random.set_state(npy)
        torch.random.set_rng_state(cpu)
        if dev is not None:
            torch.cuda.set_rng_state(dev)


def evaluate_single(model_engine, eval_dataloader, eval_gradient_accumulation_steps, quantile):
    """One pass over the evaluation set at one timestep quantile: mean of the per-step losses until the loader wraps (train.py:176-195)."""
    eval_dataloader.set_eval_quantile(quantile)
    total, count = 0.0, 0
    while True:
        model_engine.reset_activation_shape()
        iterator = get_data_iterator_for_step(eval_dataloader, model_engine, num_micro_batches=eval_gradient_accumulation_steps)
        total += model_engine.eval_batch(iterator, num_micro_batches=eval_gradient_accumulation_steps).item()
        eval_dataloader.sync_epoch()
        count += 1
        if eval_dataloader.epoch == 2:
            break
    eval_dataloader.reset()
    return total / count


def evaluate(model_engine, eval_dataloaders, eval_gradient_accumulation_steps, quantiles=None):
    """-> {'<name>/loss_quantile_<q>': loss, '<name>/loss': mean}; the training RNG streams are untouched (train.py:198-242)."""
    out = {}
    if len(eval_dataloaders) == 0:
        return out
    quantiles = TIMESTEP_QUANTILES_FOR_EVAL if quantiles is None else quantiles
    with torch.no_grad(), isolate_rng():
        seed = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        random.seed(seed)
        torch.manual_seed(seed)
        np.random.seed(seed)
        for name, loader in eval_dataloaders.items():
            losses = []
            for q in quantiles:
                loss = evaluate_single(model_engine, loader, eval_gradient_accumulation_steps, q)
                losses.append(loss)
                out[f'{name}/loss_quantile_{q:.2f}'] = loss
            out[f'{name}/loss'] = sum(losses) / len(losses)
    return out
