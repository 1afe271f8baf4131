"""Activation checkpointing with the checkpoint inputs parked in host DRAM (BASELINE config 5: "activation offload to host DRAM").

Same contract as the reference's `unsloth_checkpoint(function, *args)` (utils/unsloth_utils.py:24-79, hooked up at
train.py:586-603 as `activation_checkpoint_func`): the forward of `function` runs without an autograd graph, tensor
arguments with at least OFFLOAD_THRESHOLD elements are kept on the host instead of in HBM, the backward brings them back,
re-runs `function` with gradients enabled and back-propagates; arguments flagged `no_backward` are not kept at all and are
passed as None to the recomputation.

MI355X structure: the device->host copy runs on a dedicated copy stream into PINNED buffers (PCIe Gen5 x16, 63 GB/s) behind an
event, so it overlaps the forward of the following layers; the backward's host->device copy is issued on the same stream and the
compute stream waits on its event only.  With 288 GB of HBM the engine's default is plain (on-device) checkpointing or none;
this path exists for the video-sized activations of config 5.

hipGraph capture (the engine's fast path): a host allocation cannot be captured, so the pinned buffers come from a persistent pool keyed by
(POOL_TAG, shape, dtype) -- POOL_TAG is set by the engine to the lane / pipeline slot being warmed up or captured, because graphs of different
lanes / slots replay concurrently and must not share host buffers.  The eager warm-up passes that precede every capture allocate the buffers
(acquire in forward order, release in backward), the capture pass re-acquires the same ones, and their addresses are baked into the graph's
D2H / H2D memcpy nodes.  Under capture the copy stream is joined right after each copy is issued (the graph's static memory plan may hand the
source block to the next allocation on the capture stream), so the copies overlap compute only on the eager path.
"""
import torch

OFFLOAD_THRESHOLD = 5_000_000      # elements; same default as the reference (10 MB of bf16)
_COPY_STREAMS = {}
POOL_TAG = None                     # set by the engine around warm-up / capture / eager execution of one lane or pipeline slot
_FREE = {}                          # (tag, shape, dtype) -> [pinned host tensors not in use]


def _acquire(shape, dtype):
    key = (POOL_TAG, tuple(shape), dtype)
    free = _FREE.setdefault(key, [])
    if free:
        return free.pop(), key
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('offloaded_checkpoint: no pinned host buffer available during hipGraph capture (the eager warm-up must run the same '
                           f'checkpoint sequence first): {key}')
    return torch.empty(tuple(shape), dtype=dtype, device='cpu', pin_memory=True), key


def _release(host, key):
    _FREE.setdefault(key, []).append(host)


def _copy_stream(device):
    s = _COPY_STREAMS.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device)
        _COPY_STREAMS[device.index] = s
    return s


class _OffloadedCheckpoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, function, threshold, *args):
        kept, kept_idx, host_events = [], [], []
        for i, x in enumerate(args):
            if getattr(x, 'no_backward', False):
                continue
            if torch.is_tensor(x) and x.is_cuda and x.numel() >= threshold:
                cs = _copy_stream(x.device)
                cur = torch.cuda.current_stream(x.device)
                capturing = torch.cuda.is_current_stream_capturing()
                cs.wait_stream(cur)
                host, key = _acquire(x.shape, x.dtype)
                with torch.cuda.stream(cs):
                    host.copy_(x, non_blocking=True)
                    host_events.append((len(kept), None if capturing else cs.record_event(), x.device, key))
                if capturing:
                    cur.wait_stream(cs)                 # static memory plan: the source block must not be reused before the copy ran
                else:
                    x.record_stream(cs)
                kept.append(host)
            else:
                kept.append(x)
            kept_idx.append(i)
        with torch.no_grad():
            output = function(*args)
        tensors = [t for t in kept if torch.is_tensor(t)]
        ctx.save_for_backward(*tensors)
        ctx.layout = [('t', None) if torch.is_tensor(t) else ('v', t) for t in kept]
        ctx.function, ctx.kept_idx, ctx.num_args, ctx.host_events = function, kept_idx, len(args), host_events
        return output

    @staticmethod
    def backward(ctx, *grads):
        saved = list(ctx.saved_tensors)
        kept = [saved.pop(0) if kind == 't' else val for kind, val in ctx.layout]
        device_of = {pos: dev for pos, _, dev, _ in ctx.host_events}
        done_of = {pos: ev for pos, ev, _, _ in ctx.host_events}
        key_of = {pos: key for pos, _, _, key in ctx.host_events}
        args = [None] * ctx.num_args
        for pos, (i, x) in enumerate(zip(ctx.kept_idx, kept)):
            if pos in device_of:                                    # parked on the host: bring it back on the copy stream
                dev = device_of[pos]
                cs = _copy_stream(dev)
                host = x
                if done_of[pos] is not None:
                    cs.wait_event(done_of[pos])
                cs.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cs):
                    x = host.to(dev, non_blocking=True)
                torch.cuda.current_stream(dev).wait_stream(cs)
                if not torch.cuda.is_current_stream_capturing():
                    x.record_stream(torch.cuda.current_stream(dev))
                _release(host, key_of[pos])             # stream order (and the graph's node order) keeps the H2D ahead of the buffer's next D2H
            if torch.is_tensor(x):
                x = x.detach()
                if torch.is_floating_point(x):
                    x.requires_grad_(True)
            args[i] = x
        with torch.enable_grad():
            outputs = ctx.function(*args)
        outputs = outputs if isinstance(outputs, (tuple, list)) else (outputs,)
        out_t, grad_t = [], []
        for out, grad in zip(outputs, grads):
            if torch.is_tensor(out) and out.requires_grad:
                out_t.append(out)
                grad_t.append(grad)
        torch.autograd.backward(out_t, grad_t)
        return (None, None) + tuple(a.grad if torch.is_tensor(a) else None for a in args)


def offloaded_checkpoint(function, *args, threshold=None):
    """Drop-in for the reference's `unsloth_checkpoint` as `activation_checkpoint_func`."""
    return _OffloadedCheckpoint.apply(function, OFFLOAD_THRESHOLD if threshold is None else threshold, *args)
