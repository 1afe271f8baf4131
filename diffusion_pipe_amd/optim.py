"""Optimizer / LR-schedule factory of the training step.

Mirrors what the reference's `train.py` builds around `model_engine._configure_optimizer`:
  * `get_optimizer(model_parameters)` (train.py:650-815): stages without trainable parameters get a no-op optimizer
    (`DummyOptimizer`, train.py:61-76); `beta2_half_life` -> beta2 = 0.5 ** (global_batch / half_life); the adapter's
    `get_param_groups()` decides per-component learning rates (models/sdxl.py:604-630); every group is then split into a
    weight-decay group and a no-weight-decay group (1-D parameters and `llm_adapter.embed*`, train.py:791-813).
  * LR schedule (train.py:849-862): constant / linear / cosine, optional linear warm-up chained with SequentialLR.
The update itself stays a PyTorch optimizer on the raw bf16 parameters, as in the reference (no fp32 master weights:
the reference's ds_config has no fp16 / bf16 / ZeRO section, train.py:423-429).  On ROCm the `fused=True` multi-tensor
AdamW is the fast path; bitsandbytes' 8-bit block-wise AdamW (`adamw8bit`, `adamw8bitkahan`) is a HIP kernel of this repo (`AdamW8bit`,
csrc/optim.hip); optimizers that only exist as other CUDA extensions in the reference (optimi, torchao offload) are reported as unavailable instead of
being silently replaced.
"""
from collections import defaultdict

import torch

_TORCH_OPTIMIZERS = {
    'adamw': torch.optim.AdamW,        # replaced by FusedAdamW on a GPU stage (make_optimizer_factory)
    'adamwkahan': torch.optim.AdamW,   # FusedAdamW(kahan=True): AdamW with the reference optimizers' Kahan summation for bf16 parameters (GPU stages only)
    'sgd': torch.optim.SGD,
    'adam': torch.optim.Adam,
}
_NEEDS_EXTERNAL = {
    'adamw_optimi': 'optimi', 'stableadamw': 'optimi', 'offload': 'torchao', 'automagic': None, 'genericoptim': None,
}


class DummyOptimizer(torch.optim.Optimizer):
    """Optimizer of a pipeline stage that owns no trainable parameter (e.g. a stage of frozen layers under LoRA)."""

    def __init__(self):
        self.state = defaultdict(dict)
        self.param_groups = []

    def step(self, closure=None):
        pass

    def zero_grad(self, set_to_none=True):
        pass

    def state_dict(self):
        return {}

    def load_state_dict(self, state_dict):
        pass


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (decoupled weight decay, bias correction, no amsgrad) whose whole step end runs in two HIP
    multi-tensor passes (csrc/optim.hip): the gradient norm of the lane-summed gradients, then ONE pass that sums the engine's
    concurrent micro-batch gradient accumulators, applies the clip coefficient, updates (p, exp_avg, exp_avg_sq) in fp32
    arithmetic and zeroes the accumulators.  State keys / dtypes are torch.optim.AdamW's (`step`, `exp_avg`, `exp_avg_sq` in
    the parameter dtype -- raw bf16 parameters, no fp32 master copy, as the reference trains: train.py:423-429), so
    optimizer checkpoints interchange with the reference's.

    `step()` is the plain torch contract (reads p.grad, leaves it alone).  The engine uses the split form instead:
    `grads_sumsq(lanes)` -> (cross-stage / DP reduction of the scalar by the engine) -> `fused_update(lanes, total, max_norm)`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, kahan=False):
        """kahan: compensated summation for bf16 parameters, the reference optimizers' treatment of raw-bf16 training (optimizers/generic_optim.py:
        486-497, automagic.py:309-320, adamw_8bit.py): one extra `shift` state tensor per bf16 parameter (same key and dtype as the reference's)."""
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.kahan = bool(kahan)

    def _buckets(self, lane_grads_of):
        """[(group, dtype, step, params, exp_avgs, exp_avg_sqs, lanes, Parameters)] over parameters that have a gradient; parameters of one group are
        bucketed by dtype AND by their own step count (torch.optim.AdamW keeps the bias correction per parameter: a parameter that first
        receives a gradient on a later step, or joins a group after a resume, must not borrow its neighbours' step)."""
        out = []
        for group in self.param_groups:
            by_dtype = {}
            for p in group['params']:
                lanes = lane_grads_of(p)
                if lanes is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0.0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)      # channels-last conv weights keep their layout
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                b = by_dtype.setdefault((p.dtype, float(st['step'])), ([], [], [], [[] for _ in lanes], []))
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not dense:
                    raise RuntimeError('FusedAdamW needs dense (row-major or channels-last) parameters')
                b[0].append(p.data)
                b[1].append(st['exp_avg'])
                b[2].append(st['exp_avg_sq'])
                for lane_list, g in zip(b[3], lanes):
                    lane_list.append(g)
                b[4].append(p)                    # the Parameter itself: optimizer state is looked up by it (never by data_ptr: empty / aliased storages collide)
            out += [(group, dt, step, *b) for (dt, step), b in by_dtype.items()]
        return out

    @staticmethod
    def _lanes_from(lane_grads):
        if lane_grads is None:
            return lambda p: None if p.grad is None else [p.grad]

        def of(p):
            gs = [lane.get(id(p)) for lane in lane_grads]
            if all(g is None for g in gs):
                return None
            if any(g is None for g in gs):
                raise RuntimeError('FusedAdamW: a parameter has a gradient accumulator in some lanes only')
            return gs
        return of

    @torch.no_grad()
    def grads_sumsq(self, lane_grads=None, only=None):
        """fp32 device scalar: sum over parameters of ||sum over lanes of g||^2.  lane_grads: list of {id(p): grad} dicts (one per
        lane) or None = [p.grad].  `only(p) -> bool` restricts the counted parameters."""
        from . import ops
        of = self._lanes_from(lane_grads)
        if only is not None:
            inner = of
            of = lambda p: inner(p) if only(p) else None
        total = None
        for _, _, _, ps, ms, vs, lanes, _ in self._buckets(of):
            if total is None:
                total = torch.empty((), device=ps[0].device, dtype=torch.float32)
                ops.adamw_grads_sumsq(ps, ms, vs, lanes, total, accumulate=False)
            else:
                ops.adamw_grads_sumsq(ps, ms, vs, lanes, total, accumulate=True)
        return total

    @torch.no_grad()
    def fused_update(self, lane_grads=None, total_sumsq=None, max_norm=0.0, zero_grads=True):
        from . import ops
        updated = []
        for group, dt, step, ps, ms, vs, lanes, owners in self._buckets(self._lanes_from(lane_grads)):
            beta1, beta2 = group['betas']
            shifts = None
            if self.kahan and dt == torch.bfloat16:
                shifts = []
                for p, pd in zip(owners, ps):
                    st = self.state[p]
                    if 'shift' not in st:
                        st['shift'] = torch.zeros_like(pd, memory_format=torch.preserve_format)
                    shifts.append(st['shift'])
            updated += owners
            ops.adamw_step(ps, ms, vs, lanes, lr=group['lr'], beta1=beta1, beta2=beta2, eps=group['eps'], weight_decay=group['weight_decay'],
                           step=step + 1.0, total_sumsq=total_sumsq, max_norm=max_norm, zero_grads=zero_grads, shifts=shifts)
        for p in updated:
            self.state[p]['step'] = float(self.state[p]['step']) + 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.fused_update(None, None, 0.0, zero_grads=False)
        return loss



def create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8):
    """The 256 code values of bitsandbytes' dynamic 8-bit data type (`bitsandbytes.functional.create_dynamic_map`, the same construction with the same
    torch calls): signed for the first Adam moment, unsigned for the second.  Sorted fp32 tensor in [-1, 1] / [0, 1]."""
    data = []
    non_sign_bits = total_bits - 1
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    i = 0
    for i in range(max_exponent_bits):
        fraction_items = int(2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        boundaries = torch.linspace(0.1, 1, fraction_items)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    if additional_items > 0:
        boundaries = torch.linspace(0.1, 1, additional_items + 1)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    data.append(0)
    data.append(1.0)
    assert len(data) == 2 ** total_bits
    data.sort()
    return torch.tensor(data, dtype=torch.float32)


class AdamW8bit(torch.optim.Optimizer):
    """The reference's `adamw8bit` (bitsandbytes.optim.AdamW8bit, train.py:673-676) and `adamw8bitkahan` (optimizers/adamw_8bit.py:6-124) on MI355X:
    block-wise 8-bit Adam moments (uint8 codes of two 256-entry dynamic maps + one fp32 absmax per 256 elements: 2.03 bytes of optimizer state per
    parameter instead of AdamW's 4 in bf16 / 8 in fp32), decoupled weight decay, updated by one HIP kernel per parameter tensor (dpipe_adamw8bit_step).
    State keys follow bitsandbytes (`state1`, `state2`, `absmax1`, `absmax2`, `qmap1`, `qmap2`, `step`; `shift` for the Kahan variant); parameters with
    fewer than `min_8bit_size` elements keep fp32 moments (the library's optimizer_update_32bit), updated here with torch ops.
    bitsandbytes is not available offline: the algorithm follows its published kernels as restated in oracle/adam8bit_ref.py (parity unpinned)."""

    BLOCK = 256

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, kahan=False, min_8bit_size=4096):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.kahan, self.min_8bit_size = bool(kahan), int(min_8bit_size)
        self._qmaps = {}
        self._tables = {}

    def _maps(self, device):
        if device not in self._qmaps:
            self._qmaps[device] = (create_dynamic_map(True).to(device), create_dynamic_map(False).to(device))
        return self._qmaps[device]

    def _init_state(self, p):
        st = self.state[p]
        st['step'] = 0
        n = p.numel()
        if n < self.min_8bit_size:
            st['state1'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
            st['state2'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
        else:
            blocks = -(-n // self.BLOCK)
            st['state1'] = torch.zeros(n, dtype=torch.uint8, device=p.device)
            st['state2'] = torch.zeros(n, dtype=torch.uint8, device=p.device)
            st['absmax1'] = torch.zeros(blocks, dtype=torch.float32, device=p.device)
            st['absmax2'] = torch.zeros(blocks, dtype=torch.float32, device=p.device)
            st['qmap1'], st['qmap2'] = self._maps(p.device)
        if self.kahan:
            st['shift'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    @torch.no_grad()
    def _step_fp32_moments(items, lr, b1, b2, eps, wd):
        """The library's 32-bit path (optimizer_update_32bit, ADAM) for the tensors below min_8bit_size -- biases, norm weights: ~1 500 of SDXL's 2 600
        tensors -- as multi-tensor (`torch._foreach_*`) passes instead of a dozen launches per tensor.  items: [(p, grad, state)] whose state already counts
        this step.  Update and weight decay land in `shift` for the Kahan variant, each rounded to the parameter dtype, then the compensated add."""
        by_step = {}
        for it in items:
            by_step.setdefault(it[2]['step'], []).append(it)
        for t, its in by_step.items():
            ps, ms, vs = [p for p, _, _ in its], [st['state1'] for _, _, st in its], [st['state2'] for _, _, st in its]
            shifts = [st.get('shift') for _, _, st in its]
            kahan = shifts[0] is not None
            targets = shifts if kahan else ps
            gs = [torch.empty_like(m) for m in ms]
            torch._foreach_copy_(gs, [g for _, g, _ in its])                  # fp32 gradients
            torch._foreach_mul_(ms, b1); torch._foreach_add_(ms, gs, alpha=1.0 - b1)
            torch._foreach_mul_(vs, b2); torch._foreach_addcmul_(vs, gs, gs, value=1.0 - b2)
            c1 = 1.0 - b1 ** t
            c2 = (1.0 - b2 ** t) ** 0.5
            denom = torch._foreach_sqrt(vs)
            torch._foreach_add_(denom, eps * c2)
            upd = torch._foreach_div(ms, denom)
            torch._foreach_mul_(upd, -lr * c2 / c1)
            torch._foreach_copy_(gs, targets)                                 # (gs reused as the fp32 working copy of the updated tensor)
            torch._foreach_add_(gs, upd)
            torch._foreach_copy_(targets, gs)                                 # rounds to the parameter dtype
            if wd > 0:
                torch._foreach_copy_(gs, targets)
                torch._foreach_mul_(gs, 1.0 - lr * wd)
                torch._foreach_copy_(targets, gs)
            if kahan:
                bufs = [torch.empty_like(p) for p in ps]
                torch._foreach_copy_(bufs, ps)
                torch._foreach_add_(ps, shifts)
                torch._foreach_sub_(bufs, ps)
                torch._foreach_add_(shifts, bufs)

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict casts every tensor of the state to the parameter's dtype (codes, absmax and maps would become bf16):
        reload those from the checkpoint with their own dtypes (bitsandbytes overrides load_state_dict for the same reason); `shift` follows the parameter."""
        from itertools import chain
        super().load_state_dict(state_dict)
        saved_ids = list(chain.from_iterable(g['params'] for g in state_dict['param_groups']))
        params = list(chain.from_iterable(g['params'] for g in self.param_groups))
        for sid, p in zip(saved_ids, params):
            for key, value in state_dict['state'].get(sid, {}).items():
                if torch.is_tensor(value) and key != 'shift':
                    self.state[p][key] = value.detach().clone().to(device=p.device)
            st = self.state[p]
            if 'qmap1' in st:                                   # one shared copy of the two maps per device
                st['qmap1'], st['qmap2'] = self._maps(p.device)
            if 'step' in st and torch.is_tensor(st['step']):
                st['step'] = int(st['step'].item())

    @torch.no_grad()
    def step(self, closure=None):
        from . import hip
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            small, big = [], {}                                 # tensors below min_8bit_size: fp32 moments, updated together by multi-tensor ops
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError('AdamW8bit: the 8-bit step is a HIP kernel (no CPU fallback)')
                # element-wise over the storage order: any dense layout (row-major, or the channels-last storage of this repo's convolution weights) as long
                # as the gradient shares it (the reference instead forces both to row-major, adamw_8bit.py:19-20)
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    raise RuntimeError('AdamW8bit: parameters must be dense (row-major or channels-last)')
                grad = p.grad if p.grad.stride() == p.stride() else torch.empty_like(p).copy_(p.grad)
                st = self.state[p] if len(self.state[p]) else self._init_state(p)
                st['step'] += 1
                t = st['step']
                shift = st.get('shift')
                if st['state1'].dtype == torch.uint8:
                    big.setdefault((p.dtype, int(t)), []).append((p, grad, st))
                    continue
                small.append((p, grad, st))
            for (dt, t), items in big.items():                  # one multi-tensor launch per (dtype, step count) of the group
                self._step_8bit(items, dt, t, lr, b1, b2, eps, wd)
            if small:
                self._step_fp32_moments(small, lr, b1, b2, eps, wd)
        return loss

    CHUNK = 2048                                                # elements per workgroup of dpipe_adamw8bit_multi: 8 quantisation blocks

    def _step_8bit(self, items, dt, t, lr, b1, b2, eps, wd):
        """Every 8-bit tensor of a group in ONE launch (dpipe_adamw8bit_multi).  Pointer / chunk tables live on the device, cached on the parameter / state
        addresses; the gradient pointer row is re-uploaded only when a gradient buffer moved (never, with the engine's persistent gradient buffers)."""
        from . import hip
        dev = items[0][0].device
        key = (dt, self.kahan) + tuple(x.data_ptr() for p, g, st in items for x in (p, st['state1']))
        gkey = tuple(g.data_ptr() for _, g, _ in items)
        tab = self._tables.get(key)
        if tab is None:
            i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)
            ctens, coff = [], []
            for i, (p, _, _) in enumerate(items):
                for off in range(0, p.numel(), self.CHUNK):
                    ctens.append(i); coff.append(off)
            tab = {'p': i64([p.data_ptr() for p, _, _ in items]), 'g': None, 'gkey': None,
                   'c1': i64([st['state1'].data_ptr() for *_, st in items]), 'c2': i64([st['state2'].data_ptr() for *_, st in items]),
                   'a1': i64([st['absmax1'].data_ptr() for *_, st in items]), 'a2': i64([st['absmax2'].data_ptr() for *_, st in items]),
                   's': i64([st['shift'].data_ptr() for *_, st in items]) if self.kahan else None,
                   'n': i64([p.numel() for p, _, _ in items]), 'ctens': torch.tensor(ctens, dtype=torch.int32, device=dev), 'coff': i64(coff),
                   'nchunks': len(ctens), 'keep': [(p, st) for p, _, st in items]}     # parameters and states only: a table never pins a gradient buffer
            if len(self._tables) > 16:
                self._tables.clear()
            self._tables[key] = tab
        if tab['gkey'] != gkey:                                 # gradient buffers moved (no persistent arena): one small pointer upload, the rest stays
            tab['g'], tab['gkey'] = torch.tensor(gkey, dtype=torch.int64, device=dev), gkey
        q1, q2 = self._maps(dev)
        hip.check(hip.lib().dpipe_adamw8bit_multi(hip.ptr(tab['p']), hip.ptr(tab['g']), hip.ptr(tab['c1']), hip.ptr(tab['c2']), hip.ptr(tab['a1']), hip.ptr(tab['a2']),
                                                  hip.ptr(tab['s']), hip.ptr(tab['n']), hip.ptr(tab['ctens']), hip.ptr(tab['coff']), tab['nchunks'], hip.ptr(q1), hip.ptr(q2),
                                                  float(lr), float(b1), float(b2), float(eps), float(wd), int(t), 1.0, hip.dtype_code(dt), hip.stream()), 'adamw8bit_multi')


def computed_beta2(global_batch_size, beta2_half_life):
    """beta2 such that the second-moment EMA halves every `beta2_half_life` examples (train.py:658-663)."""
    return 0.5 ** (global_batch_size / beta2_half_life)


def split_weight_decay(param_groups):
    """Each group -> [group with >= 2-D params, group with 1-D params and weight_decay = 0]; empty halves are dropped and
    the with-decay half comes first (train.py:791-813)."""
    out = []
    for pg in param_groups:
        pg = dict(pg)
        params = pg.pop('params')
        no_wd = [p for p in params if p.ndim == 1 or getattr(p, 'original_name', '').startswith('llm_adapter.embed')]
        no_wd_ids = {id(p) for p in no_wd}
        wd = [p for p in params if id(p) not in no_wd_ids]
        if wd:
            out.append(dict(pg, params=wd))
        if no_wd:
            out.append(dict(pg, params=no_wd, weight_decay=0))
    return out


def _optimizer_class(optim_type):
    key = optim_type.lower()
    if key in _TORCH_OPTIMIZERS:
        return _TORCH_OPTIMIZERS[key]
    if key in _NEEDS_EXTERNAL:
        dep = _NEEDS_EXTERNAL[key]
        raise NotImplementedError(f"optimizer type '{optim_type}' is not available in this ROCm build"
                                  + (f" (the reference takes it from '{dep}', a CUDA-only extension)" if dep else ''))
    if hasattr(torch.optim, optim_type):
        return getattr(torch.optim, optim_type)
    raise NotImplementedError(f'unknown optimizer type: {optim_type}')


def make_optimizer_factory(config, workload, global_batch_size, device_is_gpu=True, use_hip_adamw=True):
    """-> get_optimizer(model_parameters), the callable handed to `engine._configure_optimizer` (train.py:817).

    `config['optimizer']` is the reference's TOML table ({type, lr, betas, weight_decay, eps, beta2_half_life?, ...});
    `workload.get_param_groups(params)` supplies the per-component groups (parameters carry `.original_name`)."""
    optim_config = dict(config['optimizer'])

    def get_optimizer(model_parameters):
        model_parameters = list(model_parameters)
        if len(model_parameters) == 0:
            return DummyOptimizer()
        cfg = dict(optim_config)
        optim_type = cfg.pop('type')
        if cfg.pop('gradient_release', False):
            raise NotImplementedError('gradient_release (one optimizer step per micro-batch) is outside the train_batch hot path')
        half_life = cfg.pop('beta2_half_life', None)
        if half_life:
            betas = list(cfg['betas'])
            assert len(betas) == 2
            betas[1] = computed_beta2(global_batch_size, half_life)
            cfg['betas'] = betas
        if 'betas' in cfg:
            cfg['betas'] = tuple(cfg['betas'])
        klass = _optimizer_class(optim_type) if optim_type.lower() not in ('adamw8bit', 'adamw8bitkahan') else None
        plain_adamw = klass is torch.optim.AdamW and not set(cfg) - {'lr', 'betas', 'eps', 'weight_decay'}
        if optim_type.lower() in ('adamw8bit', 'adamw8bitkahan'):
            # train.py:673-686: bitsandbytes.optim.AdamW8bit / optimizers/adamw_8bit.AdamW8bitKahan -> the HIP kernel of this repo
            if not device_is_gpu:
                raise NotImplementedError(f"optimizer type '{optim_type}' is a HIP kernel: GPU stages only")
            extra = set(cfg) - {'lr', 'betas', 'eps', 'weight_decay', 'min_8bit_size'}
            if extra:
                raise NotImplementedError(f"optimizer type '{optim_type}': unsupported options {sorted(extra)}")
            groups = split_weight_decay(workload.get_param_groups(model_parameters))
            return AdamW8bit(groups, kahan=optim_type.lower() == 'adamw8bitkahan', **cfg)
        if optim_type.lower() == 'adamwkahan':
            if not (plain_adamw and device_is_gpu and use_hip_adamw):
                raise NotImplementedError("optimizer type 'adamwkahan' runs on the fused HIP step end of a GPU stage only")
            cfg['kahan'] = True
        if plain_adamw and device_is_gpu and use_hip_adamw:
            klass = FusedAdamW                                        # lane sum + clip + update + zero in two HIP passes
        elif klass in (torch.optim.AdamW, torch.optim.Adam) and 'fused' not in cfg and 'foreach' not in cfg:
            cfg['fused' if device_is_gpu else 'foreach'] = True       # one multi-tensor launch chain per step
        groups = split_weight_decay(workload.get_param_groups(model_parameters))
        return klass(groups, **cfg)

    return get_optimizer


def make_lr_scheduler(optimizer, config, steps_per_epoch):
    """constant | linear | cosine (+ linear warm-up), train.py:849-862."""
    kind = config.get('lr_scheduler', 'constant')
    sched = torch.optim.lr_scheduler
    if kind == 'constant':
        main = sched.ConstantLR(optimizer, factor=1.0)
    elif kind == 'linear':
        main = sched.LinearLR(optimizer, start_factor=1.0, end_factor=0.0, total_iters=config['epochs'] * steps_per_epoch)
    elif kind == 'cosine':
        main = sched.CosineAnnealingLR(optimizer, T_max=config['epochs'] * steps_per_epoch, eta_min=1e-6)
    else:
        raise NotImplementedError(f'Unknown lr_scheduler: {kind}')
    warmup = config.get('warmup_steps', 0)
    if warmup > 0:
        ramp = sched.LinearLR(optimizer, start_factor=1 / warmup, total_iters=warmup)
        main = sched.SequentialLR(optimizer, schedulers=[ramp, main], milestones=[warmup])
    return main
