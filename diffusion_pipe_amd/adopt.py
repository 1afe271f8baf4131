"""`adopt(model_or_layers)`: run a REFERENCE adapter's own modules on the HIP kernels, in place, with the parameters shared.

`north_star`: "existing model adapters drop in unchanged".  A reference adapter object (models/base.py:262-303: `to_layers()` returns nn.Modules built from
the adapter's transformer, e.g. models/wan/wan.py:414-546 around models/wan/model.py:237-312) reaches this engine as plain PyTorch modules, whose arithmetic would
run on ATen.  `adopt` walks such a module tree (or the list `to_layers()` returned) and substitutes, IN the tree:

  * block-level: classes named `WanAttentionBlock` / `Head` of models/wan/model.py -> `workloads.wan.WanAttentionBlock` / `Head` behind the reference's call
    signature (`block(x, e0, seq_lens, grid_sizes, freqs, context, context_lens)`, models/wan/wan.py:525; `head(x, e)`, :543) -- fused QKV-norm-RoPE, flash
    attention, LayerNorm + modulation, gated residuals;
  * leaf-level, everywhere else: `torch.nn.Linear` -> `nn.Linear`, `torch.nn.LayerNorm` (and `WanLayerNorm`) -> `nn.LayerNorm`, `WanRMSNorm` -> `nn.RMSNorm`,
    `torch.nn.GroupNorm` -> `nn.GroupNorm`, `torch.nn.Conv2d` (plain, square) -> `nn.Conv2d`, `torch.nn.GELU / SiLU` -> the HIP activations.

Every substitute holds the SAME `nn.Parameter` objects as the module it replaces (names unchanged), so `state_dict()`, `load_state_dict()`, optimizer parameter
groups, `p.original_name`, LoRA wrapping by name and the reference's `save_model / save_adapter` see no difference.  Nothing is copied; modules without a
substitute stay what they are (the engine only needs nn.Modules).  Returns a report {qualified name: (old class, new class)}.
"""
import functools

import torch
from torch import nn as tnn

from . import nn as dnn

_ROPE_CACHE_MAX = 16


def _share(dst, src):
    """dst (freshly built on the meta device) takes src's Parameter / buffer OBJECTS name by name; -> dst."""
    src_params, src_bufs = dict(src.named_parameters(recurse=True)), dict(src.named_buffers(recurse=True))
    for name, _ in list(dst.named_parameters(recurse=True)):
        if name not in src_params:
            raise KeyError(f'adopt: {type(src).__name__} has no parameter {name!r} for {type(dst).__name__}')
        mod, leaf = _owner(dst, name)
        if tuple(mod._parameters[leaf].shape) != tuple(src_params[name].shape):
            raise ValueError(f'adopt: parameter {name!r}: {tuple(src_params[name].shape)} does not fit {tuple(mod._parameters[leaf].shape)}')
        mod._parameters[leaf] = src_params[name]
    for name, _ in list(dst.named_buffers(recurse=True)):
        if name in src_bufs:
            mod, leaf = _owner(dst, name)
            mod._buffers[leaf] = src_bufs[name]
    extra = set(src_params) - set(dict(dst.named_parameters(recurse=True)))
    if extra:
        raise KeyError(f'adopt: {type(dst).__name__} would drop parameters {sorted(extra)} of {type(src).__name__}')
    dst.train(src.training)
    return dst


def _owner(root, dotted):
    *path, leaf = dotted.split('.')
    for p in path:
        root = getattr(root, p)
    return root, leaf


def _meta(build):
    with torch.device('meta'):
        return build()


# ------------------------------------------------------------------------------------------------------------------ leaf substitutes
def _linear(m):
    return _share(_meta(lambda: dnn.Linear(m.in_features, m.out_features, bias=m.bias is not None)), m)


def _layer_norm(m):
    if len(m.normalized_shape) != 1:
        return None
    return _share(_meta(lambda: dnn.LayerNorm(m.normalized_shape[0], eps=m.eps, elementwise_affine=m.elementwise_affine, bias=m.bias is not None)), m)


def _rms_norm(m):
    return _share(_meta(lambda: dnn.RMSNorm(m.dim, eps=m.eps)), m)


def _group_norm(m):
    return _share(_meta(lambda: dnn.GroupNorm(m.num_groups, m.num_channels, eps=m.eps, affine=m.affine)), m)


def _conv2d(m):
    plain = m.groups == 1 and tuple(m.dilation) == (1, 1) and m.padding_mode == 'zeros' and not isinstance(m.padding, str) and m.kernel_size[0] == m.kernel_size[1] \
        and m.stride[0] == m.stride[1] and m.stride[0] in (1, 2) and m.padding[0] == m.padding[1]
    if not plain:
        return None
    new = _share(_meta(lambda: dnn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, stride=m.stride, padding=m.padding, bias=m.bias is not None)), m)
    if m.weight.device.type != 'meta':
        m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)          # the storage format nn.Conv2d keeps (same values, same logical shape)
    return new


def _leaf(m):
    """-> substitute for one module, or None.  Exact torch types only (a subclass may carry behaviour of its own), plus the reference's named norm classes."""
    t, name = type(m), type(m).__name__
    if t is tnn.Linear:
        return _linear(m)
    if t is tnn.LayerNorm or name == 'WanLayerNorm':              # models/wan/model.py:87-99: nn.LayerNorm computed in fp32 (the HIP kernel's statistics are fp32)
        return _layer_norm(m)
    if name == 'WanRMSNorm' and hasattr(m, 'dim') and hasattr(m, 'eps'):      # models/wan/model.py:70-84
        return _rms_norm(m)
    if t is tnn.GroupNorm:
        return _group_norm(m)
    if t is tnn.Conv2d:
        return _conv2d(m)
    if t is tnn.GELU:
        return dnn.GELU(approximate=m.approximate)
    if t is tnn.SiLU:
        return dnn.SiLU()
    return None


# ------------------------------------------------------------------------------------------------------------------ Wan blocks behind the reference's signature
@functools.lru_cache(maxsize=None)
def _adopted_wan_classes():
    from .workloads import wan

    class AdoptedWanAttentionBlock(wan.WanAttentionBlock):
        """workloads.wan.WanAttentionBlock called the way models/wan/wan.py:525 calls the reference's block: (x, e0, seq_lens, grid_sizes, freqs, context, context_lens).
        The rotary tables of the (frames, height, width) grid are built once per grid and reused -- reading `grid_sizes` is a host synchronisation, which a hipGraph
        capture cannot contain: the engine's eager warm-up passes fill the cache, captures hit it.  Sequences are taken at full length (`seq_lens` = the padded length,
        as for the unpadded batches the reference builds at micro-batch 1 and as `workloads.wan` assumes)."""

        def _tables(self, grid_sizes, freqs, tokens):
            cache = self.__dict__.setdefault('_rope_cache', {})
            key = (grid_sizes.data_ptr(), freqs.data_ptr(), tokens)
            capturing = grid_sizes.is_cuda and torch.cuda.is_current_stream_capturing()
            if capturing and key in cache:
                return cache[key][1:]
            if capturing:
                raise RuntimeError('adopt: rotary tables of a new grid requested inside a hipGraph capture (run one eager pass with this shape first)')
            grid = tuple(int(v) for v in grid_sizes[0].tolist())
            hit = cache.get(key)
            if hit is None or hit[0] != grid:
                if len(cache) >= _ROPE_CACHE_MAX:
                    cache.clear()
                cos, sin = wan.rope_tables(freqs, grid)
                hit = cache[key] = (grid, cos.to(grid_sizes.device), sin.to(grid_sizes.device))
            return hit[1:]

        def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens=None):
            cos, sin = self._tables(grid_sizes, freqs, x.shape[1])
            return super().forward(x, e, cos, sin, context, context_lens)

    class AdoptedWanHead(wan.Head):
        pass                                     # models/wan/model.py:312-343 `head(x, e)`: same signature

    return AdoptedWanAttentionBlock, AdoptedWanHead


def _wan_block(m):
    if not all(hasattr(m, a) for a in ('self_attn', 'cross_attn', 'ffn', 'modulation', 'norm1', 'norm2', 'num_heads', 'dim', 'ffn_dim')):
        return None
    if type(m.cross_attn).__name__ != 'WanCrossAttention' or not getattr(m, 'qk_norm', True) or tuple(getattr(m, 'window_size', (-1, -1))) != (-1, -1):
        return None                              # i2v cross attention / windows / no qk-norm: leaf substitution still applies inside
    Block, _ = _adopted_wan_classes()
    return _share(_meta(lambda: Block(m.dim, m.ffn_dim, m.num_heads, cross_attn_norm=bool(getattr(m, 'cross_attn_norm', False)), eps=m.eps)), m)


def _wan_head(m):
    if not all(hasattr(m, a) for a in ('norm', 'head', 'modulation', 'dim', 'out_dim', 'patch_size')) or not isinstance(m.head, tnn.Linear):
        return None
    _, Head = _adopted_wan_classes()
    return _share(_meta(lambda: Head(m.dim, m.out_dim, tuple(m.patch_size), eps=m.eps)), m)


_BLOCKS = {'WanAttentionBlock': _wan_block, 'Head': _wan_head}


# ------------------------------------------------------------------------------------------------------------------ the walk
def adopt(model_or_layers, blocks=True, leaves=True):
    """Substitute in place (see the module docstring).  `model_or_layers`: an nn.Module, or the list / tuple `to_layers()` returned (callables that are not
    nn.Modules are skipped; a module reachable from several layers -- e.g. FinalLayer's `model` -- is substituted once).  -> {name: (old class, new class)}."""
    roots = [('', model_or_layers)] if isinstance(model_or_layers, tnn.Module) else [(f'layers.{i}', m) for i, m in enumerate(model_or_layers) if isinstance(m, tnn.Module)]
    report, done = {}, {}

    def visit(prefix, mod):
        for child_name, child in list(mod._modules.items()):
            if child is None:
                continue
            full = f'{prefix}.{child_name}' if prefix else child_name
            if id(child) in done:                                     # the same module under a second parent: point it at the substitute made before
                if done[id(child)] is not None:
                    mod._modules[child_name] = done[id(child)]
                continue
            if getattr(type(child), '__module__', '').startswith('diffusion_pipe_amd'):
                done[id(child)] = None
                continue
            new = None
            if blocks and type(child).__name__ in _BLOCKS:
                new = _BLOCKS[type(child).__name__](child)
            if new is None and leaves:
                new = _leaf(child)
            done[id(child)] = new
            if new is not None:
                mod._modules[child_name] = new
                report[full] = (type(child).__name__, type(new).__name__)
            else:
                visit(full, child)

    for prefix, root in roots:
        visit(prefix, root)
    return report


class _CallableModule(type(torch)):
    """`from diffusion_pipe_amd import adopt; adopt(model.to_layers())` and `diffusion_pipe_amd.adopt.adopt(...)` are the same call."""

    def __call__(self, *args, **kwargs):
        return adopt(*args, **kwargs)


import sys as _sys  # noqa: E402
_sys.modules[__name__].__class__ = _CallableModule
