"""ORACLE / checker leg (test infrastructure; never imported by the product): what the REFERENCE's own bf16 evaluation of a micro-batch looks like.

The reference trains SDXL with bf16 weights (`[model] dtype = 'bfloat16'`, models/sdxl.py:387) and runs every pipeline layer's forward under
`@torch.autocast('cuda', dtype=AUTOCAST_DTYPE)` (models/sdxl.py:675,794,810,825,842,857,874,906,937,988) with the loss computed with autocast disabled
(models/sdxl.py:636); gradients accumulate in the parameters' dtype (bf16) and the clip norm is taken over them in fp32 (utils/patches.py:175-246).
This file evaluates exactly that -- the oracle restatement `oracle/sdxl_ref.SDXLRef` with bf16 weights on the GPU through ATen / MIOpen kernels under autocast -- on
the micro-batches bench.py's `parity` leg uses, so that the distance of this repo's timed bf16 path from the fp32 oracle can be read next to the distance of
the reference's own bf16 path from the same fp32 oracle.  ATen is allowed here: this is the checker, not the product.
"""
import torch

from . import eager_step, sdxl_ref


def sdxl_reference_bf16(cfg, state, micro_batches, device, dtype=torch.bfloat16):
    """-> (losses, pre-clip global gradient norms) of `micro_batches` = [(features, label), ...] (host tensors) evaluated ONE at a time on `state` =
    {module name: state dict} (the product's weights).  Eager, sequential `to_layers()`, one backward per micro-batch (GAS = 1: the parity leg's steps hold one sample)."""
    torch.backends.cudnn.benchmark = False                     # MIOpen immediate mode: no per-shape find pass on a fresh box
    with torch.device(device):                                  # the restatement builds its index / mask helpers with bare factory calls
        ref = sdxl_ref.SDXLRef(cfg, seed=0)
        for k, m in ref.modules().items():
            m.to(dtype)
            m.load_state_dict({n: v.to(device=device, dtype=dtype) for n, v in state[k].items()})
        layers = ref.to_layers()
        loss_fn = eager_step.sdxl_loss_fn()
        params = ref.parameters()
        losses, norms = [], []
        for feats, label in micro_batches:
            for p in params:
                p.grad = None
            x = tuple(t.to(device) for t in feats)
            for layer in layers:
                with torch.autocast('cuda', dtype=dtype):
                    x = layer(x)
            with torch.autocast('cuda', enabled=False):
                loss = loss_fn(x, tuple(t.to(device) for t in label))
            loss.backward()
            sq = torch.stack([p.grad.detach().float().norm(2) for p in params if p.grad is not None]).square().sum()
            losses.append(float(loss.item())); norms.append(float(sq.sqrt().item()))
            del x, loss
    del ref, layers, params
    torch.cuda.empty_cache()
    return losses, norms
