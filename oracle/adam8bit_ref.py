"""TEST INFRASTRUCTURE ONLY (see oracle/README.md) -- numpy restatement of the 8-bit block-wise AdamW the reference selects with
`optimizer.type = 'adamw8bit'` / `'adamw8bitkahan'` (train.py:673-686 -> bitsandbytes.optim.AdamW8bit, optimizers/adamw_8bit.py:6-124).

**Parity unpinned**: the algorithm lives in bitsandbytes (requirements.txt:14, no version pin), which is neither vendored under /root/reference nor
installed here, so nothing in this file can be checked against the library's output.  It restates the library's published algorithm:

  * quantisation maps: `bitsandbytes.functional.create_dynamic_map(signed=True)` for the first moment, `(signed=False)` for the second
    (Optimizer8bit.fill_qmap): 256 sorted fp32 code values in [-1, 1] / [0, 1];
  * state: uint8 codes + one fp32 absmax per block of 256 elements (Optimizer2State.init_state, block_wise=True); parameters with fewer than
    `min_8bit_size` = 4096 elements keep fp32 moments (optimizer_update_32bit);
  * step (csrc/kernels.cu kOptimizerStatic8bit2StateBlockwise, ADAM): dequantise, m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, new block absmax,
    p += step_size * m / (sqrt(v) + correction2 * eps) with step_size = -lr * correction2 / correction1, correction1 = 1 - b1^t,
    correction2 = sqrt(1 - b2^t), then p *= 1 - lr * weight_decay; requantise m / absmax1, v / absmax2 to the NEAREST code (ties keep the
    search pivot: measure zero), and move the first-moment code one step if its sign differs from m's;
  * the reference's Kahan variant (optimizers/adamw_8bit.py:15,43,119-124) hands the library a `shift` buffer (parameter dtype) in place of
    the parameter -- update AND weight decay land in `shift` -- then p' = p + shift, shift' = shift + (p - p') in the parameter's dtype.

All arithmetic below is fp32 (numpy float32), as in the kernel."""
import numpy as np
import torch

BLOCK = 256
MIN_8BIT_SIZE = 4096


def create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8):
    """bitsandbytes.functional.create_dynamic_map: dynamic-exponent 8-bit data type (Dettmers et al., 8-bit optimizers via block-wise quantization)."""
    data = []
    non_sign_bits = total_bits - 1          # (the library subtracts 1 in both the signed and the unsigned case)
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    i = 0
    for i in range(max_exponent_bits):
        fraction_items = int(2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        boundaries = torch.linspace(0.1, 1, fraction_items)          # the library builds the map with torch (fp32 linspace)
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
    return np.asarray(data, dtype=np.float32)


def quantize_nearest(qmap, x):
    """index of the code nearest to x (x in [-1, 1]); on an exact midpoint the lower code wins, as good as any (the kernel keeps its search pivot)."""
    x = np.asarray(x, dtype=np.float32)
    hi = np.clip(np.searchsorted(qmap, x, side='left'), 1, len(qmap) - 1)
    lo = hi - 1
    take_hi = (qmap[hi] - x) < (x - qmap[lo])
    return np.where(take_hi, hi, lo).astype(np.uint8)


def _pad_blocks(a, fill=0):
    n = a.size
    nb = -(-n // BLOCK)
    out = np.full(nb * BLOCK, fill, dtype=a.dtype)
    out[:n] = a.reshape(-1)
    return out.reshape(nb, BLOCK), n


def adam8bit_blockwise_step(p, g, c1, c2, absmax1, absmax2, qmap1, qmap2, step, lr, beta1, beta2, eps, weight_decay, gnorm_scale=1.0):
    """One kOptimizerStatic8bit2StateBlockwise step on flat fp32 views.  p: the tensor the library updates (the parameter, or the reference's Kahan
    `shift` buffer) as fp32 values; c1 / c2: uint8 codes; absmax1 / absmax2: fp32 [blocks].  Returns (p', c1', c2', absmax1', absmax2') -- p' in fp32
    (the caller rounds to the parameter dtype: the kernel rounds after the update and again after the weight-decay product)."""
    f = np.float32
    n = p.size
    P, _ = _pad_blocks(p.astype(f))
    G, _ = _pad_blocks(g.astype(f))
    C1, _ = _pad_blocks(c1.astype(np.uint8))
    C2, _ = _pad_blocks(c2.astype(np.uint8))
    valid = (np.arange(P.size).reshape(P.shape) < n)
    finite = np.isfinite(G)
    gv = G * f(gnorm_scale)
    m = qmap1[C1] * absmax1.astype(f)[:, None]
    v = qmap2[C2] * absmax2.astype(f)[:, None]
    m = np.where(finite, m * f(beta1) + f(1.0 - beta1) * gv, f(0))
    v = np.where(finite, v * f(beta2) + f(1.0 - beta2) * gv * gv, f(0))
    m = np.where(valid, m, f(0)); v = np.where(valid, v, f(0))
    new1 = np.abs(m).max(axis=1).astype(f)
    new2 = np.abs(v).max(axis=1).astype(f)
    correction1 = f(1.0) - f(beta1) ** f(step)
    correction2 = np.sqrt(f(1.0) - f(beta2) ** f(step)).astype(f)
    step_size = f(-lr) * correction2 / correction1
    upd = P + step_size * (m / (np.sqrt(v) + correction2 * f(eps)))
    return_p = np.where(finite, upd, P)
    wd_factor = f(1.0) - f(lr) * f(weight_decay)
    with np.errstate(divide='ignore', invalid='ignore'):
        x1 = np.where(new1[:, None] > 0, m / new1[:, None], f(0))
        x2 = np.where(new2[:, None] > 0, v / new2[:, None], f(0))
    q1 = quantize_nearest(qmap1, x1).astype(np.int32)
    q2 = quantize_nearest(qmap2, x2)
    flip = np.signbit(qmap1[q1]) != np.signbit(m)
    q1 = np.where(flip & (m > 0), q1 + 1, np.where(flip & ~(m > 0), q1 - 1, q1)).clip(0, 255).astype(np.uint8)
    return (return_p.reshape(-1)[:n], finite.reshape(-1)[:n], wd_factor, q1.reshape(-1)[:n], q2.reshape(-1)[:n], new1, new2)


def round_to(x, dtype):
    """fp32 -> parameter dtype -> fp32 ('bf16' = round to nearest even on the upper 16 bits, 'f32' = identity)."""
    x = np.asarray(x, dtype=np.float32)
    if dtype == 'f32':
        return x
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    out[np.isnan(x)] = np.nan
    return out


class AdamW8bitRef:
    """Reference optimizer over numpy arrays (one flat fp32 array per parameter holding values exactly representable in `dtype`).
    kahan=False: bitsandbytes.optim.AdamW8bit; kahan=True: optimizers/adamw_8bit.py AdamW8bitKahan."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, kahan=False, dtype='bf16'):
        self.params = params
        self.lr, self.betas, self.eps, self.wd, self.kahan, self.dtype = lr, betas, eps, weight_decay, kahan, dtype
        self.qmap1, self.qmap2 = create_dynamic_map(True), create_dynamic_map(False)
        self.state = [None] * len(params)

    def _init(self, p):
        n = p.size
        st = {'step': 0}
        if n < MIN_8BIT_SIZE:
            st['m'] = np.zeros(n, np.float32); st['v'] = np.zeros(n, np.float32)
        else:
            nb = -(-n // BLOCK)
            st['c1'] = np.zeros(n, np.uint8); st['c2'] = np.zeros(n, np.uint8)
            st['absmax1'] = np.zeros(nb, np.float32); st['absmax2'] = np.zeros(nb, np.float32)
        if self.kahan:
            st['shift'] = np.zeros(n, np.float32)
        return st

    def step(self, grads):
        f = np.float32
        b1, b2 = self.betas
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if self.state[i] is None:
                self.state[i] = self._init(p)
            st = self.state[i]
            st['step'] += 1
            t = st['step']
            target = st['shift'] if self.kahan else p          # the tensor the library kernel updates
            if 'c1' in st:
                upd, finite, wdf, c1, c2, a1, a2 = adam8bit_blockwise_step(target, g, st['c1'], st['c2'], st['absmax1'], st['absmax2'], self.qmap1, self.qmap2,
                                                                           t, self.lr, b1, b2, self.eps, self.wd)
                st['c1'], st['c2'], st['absmax1'], st['absmax2'] = c1, c2, a1, a2
                new = round_to(upd, self.dtype)
                if self.wd > 0:
                    new = np.where(finite, round_to(new * wdf, self.dtype), new)
            else:                                              # kOptimizer32bit2State
                gv = g.astype(f)
                st['m'] = st['m'] * f(b1) + f(1.0 - b1) * gv
                st['v'] = st['v'] * f(b2) + f(1.0 - b2) * gv * gv
                c1 = f(1.0) - f(b1) ** f(t)
                c2 = np.sqrt(f(1.0) - f(b2) ** f(t)).astype(f)
                new = round_to(target + (f(-self.lr) * c2 / c1) * (st['m'] / (np.sqrt(st['v']) + f(self.eps) * c2)), self.dtype)
                if self.wd > 0:
                    new = round_to(new * (f(1.0) - f(self.lr) * f(self.wd)), self.dtype)
            if self.kahan:
                st['shift'] = new
                buf = p.copy()
                p[:] = round_to(p + st['shift'], self.dtype)
                st['shift'] = round_to(st['shift'] + round_to(buf - p, self.dtype), self.dtype)
            else:
                p[:] = new
