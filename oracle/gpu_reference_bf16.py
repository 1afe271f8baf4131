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


class _ConvAsMatmul:
    """Inside this context nn.Conv2d evaluates as unfold + matmul (+ bias add on the rounded output, which is also what PyTorch-ROCm's MIOpen path does: convolution, then
    `output.add_(bias)`).  Same arithmetic class as the library convolution under autocast -- bf16 products, fp32 accumulation, one rounding of the result -- without
    MIOpen, which compiles its kernels at first use on a fresh box (round 6, first run of this leg: 215 s for 16 samples, almost all of it MIOpen's run-time compiles)."""

    def __enter__(self):
        import torch.nn as nn
        import torch.nn.functional as F
        self._orig = nn.Conv2d._conv_forward

        def conv_forward(mod, x, weight, bias):
            if mod.groups != 1 or tuple(mod.dilation) != (1, 1) or isinstance(mod.padding, str) or mod.padding_mode != 'zeros':
                return self._orig(mod, x, weight, bias)
            B, _, H, W = x.shape
            kh, kw = weight.shape[2:]
            Ho = (H + 2 * mod.padding[0] - kh) // mod.stride[0] + 1
            Wo = (W + 2 * mod.padding[1] - kw) // mod.stride[1] + 1
            cols = F.unfold(x, (kh, kw), padding=mod.padding, stride=mod.stride)            # [B, Cin kh kw, Ho Wo]
            y = torch.matmul(weight.flatten(1), cols.to(weight.dtype) if cols.dtype != weight.dtype and not torch.is_autocast_enabled() else cols)
            y = y.view(B, weight.shape[0], Ho, Wo)
            return y if bias is None else y + bias.to(y.dtype)[None, :, None, None]
        nn.Conv2d._conv_forward = conv_forward
        return self

    def __exit__(self, *exc):
        import torch.nn as nn
        nn.Conv2d._conv_forward = self._orig
        return False


def sdxl_reference_bf16(cfg, state, micro_batches, device, dtype=torch.bfloat16, library_conv=False):
    """-> (losses, pre-clip global gradient norms) of `micro_batches` = [(features, label), ...] (host tensors) evaluated ONE at a time on `state` =
    {module name: state dict} (the product's weights).  Eager, sequential `to_layers()`, one backward per micro-batch (GAS = 1: the parity leg's steps hold one sample)."""
    import contextlib
    # `library_conv`: the convolutions through MIOpen as a PyTorch-ROCm user of the reference would run them (slow on a fresh box: run-time kernel compiles)
    with torch.device(device), (contextlib.nullcontext() if library_conv else _ConvAsMatmul()):                # (torch.device: the restatement builds its index / mask helpers with bare factory calls)
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


def _group_of(name, mod):
    """rounding groups of tools/coherent_noise_probe.py (which bf16 roundings move the gradient norm?)"""
    import torch.nn as nn
    g = set()
    leaf = isinstance(mod, (nn.Linear, nn.Conv2d, nn.GroupNorm, nn.LayerNorm))
    if not leaf:
        return g
    temb = 'time_emb_proj' in name or 'time_embedding' in name or 'add_embedding' in name
    ctx = 'attn2.to_k' in name or 'attn2.to_v' in name or name.startswith('te')
    g.add('all')
    g.add('temb' if temb else 'all-temb')
    if ctx:
        g.add('ctx')
    if isinstance(mod, (nn.GroupNorm, nn.LayerNorm)):
        g.add('norms')
    if isinstance(mod, nn.Conv2d):
        g.add('convs')
    if isinstance(mod, nn.Linear) and not temb:
        g.add('linears')
    return g


def sdxl_oracle_fp32_on_gpu(cfg, state, micro_batches, device):
    """The ORACLE's fp32 eager path (oracle/sdxl_ref.py + eager_step's loss) evaluated on the GPU through ATen in fp32 -- the same code the host evaluates for `cpu_baseline`,
    on another processor.  -> (losses, pre-clip gradient norms).  bench.py uses it for the parity samples beyond what the host cores can evaluate inside the run (the GPU
    box's container gives ~32 effective cores: 25 s per sample, no speed-up from parallel workers) and reports its agreement with the host evaluation on the samples both
    see (observed <= 4.2e-5 on the gradient norm, 16 samples: profiles/r6b_*)."""
    prev_tf32 = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    losses, norms = [], []
    with torch.device(device), _ConvAsMatmul():
        ref = sdxl_ref.SDXLRef(cfg, seed=0)
        for k, m in ref.modules().items():
            m.load_state_dict({n: v.to(device=device, dtype=torch.float32) for n, v in state[k].items()})
        layers, loss_fn, params = ref.to_layers(), eager_step.sdxl_loss_fn(), ref.parameters()
        for feats, label in micro_batches:
            for p in params:
                p.grad = None
            x = tuple(t.to(device) for t in feats)
            for layer in layers:
                x = layer(x)
            loss = loss_fn(x, tuple(t.to(device) for t in label))
            loss.backward()
            sq = torch.stack([p.grad.detach().double().norm(2) for p in params if p.grad is not None]).square().sum()
            losses.append(float(loss.item())); norms.append(float(sq.sqrt().item()))
            del x, loss
    torch.backends.cuda.matmul.allow_tf32 = prev_tf32
    del ref, layers, params
    torch.cuda.empty_cache()
    return losses, norms


def sdxl_rounding_groups(cfg, state, micro_batches, device, groups=('none', 'all', 'temb', 'all-temb', 'ctx', 'norms', 'convs', 'linears')):
    """Diagnostic (DPIPE_BENCH_PARITY_GROUPS=1): the oracle model in FP32 on the GPU with forward hooks that round the outputs of one module group to bf16
    (straight-through gradient), on the bench's own weights and parity samples -> {group: [gradient norm per sample]}.  Says which roundings of a bf16 forward the
    global gradient norm of this network reacts to."""
    prev_tf32 = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    active = {'g': 'none'}
    with torch.device(device), _ConvAsMatmul():
        ref = sdxl_ref.SDXLRef(cfg, seed=0)
        for k, m in ref.modules().items():
            m.load_state_dict({n: v.to(device=device, dtype=torch.float32) for n, v in state[k].items()})
        named = [(f'unet.{k}', m) for k, m in ref.unet.named_modules()] + [(f'te1.{k}', m) for k, m in ref.text_encoder.named_modules()] + \
                [(f'te2.{k}', m) for k, m in ref.text_encoder_2.named_modules()]

        def mk(gs):
            def hook(_m, _i, o):
                return o.to(torch.bfloat16).to(torch.float32) if active['g'] in gs else None
            return hook
        for name, m in named:
            gs = _group_of(name, m)
            if gs:
                m.register_forward_hook(mk(gs))
        layers, loss_fn, params = ref.to_layers(), eager_step.sdxl_loss_fn(), ref.parameters()
        out = {g: [] for g in groups}
        for feats, label in micro_batches:
            for g in groups:
                active['g'] = g
                for p in params:
                    p.grad = None
                x = tuple(t.to(device) for t in feats)
                for layer in layers:
                    x = layer(x)
                loss_fn(x, tuple(t.to(device) for t in label)).backward()
                sq = torch.stack([p.grad.detach().float().norm(2) for p in params if p.grad is not None]).square().sum()
                out[g].append(float(sq.sqrt().item()))
                del x
    torch.backends.cuda.matmul.allow_tf32 = prev_tf32
    del ref, layers, params
    torch.cuda.empty_cache()
    return out
