"""ctypes binding of libdpipe_hip.so (the C-ABI HIP layer, include/dpipe_hip.h).

The product path has no CPU fallback: if the shared object is missing or a symbol is absent this module raises,
and every wrapper checks the return code of the C call (`check`).  PyTorch is used only for device memory and the
current HIP stream.
"""
import ctypes
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p, POINTER
from pathlib import Path

import torch

import os

# DPIPE_HIP_LIB: another build of the same library (A/B kernel timings against a previous build); the default is the in-tree .so
LIB_PATH = Path(os.environ.get('DPIPE_HIP_LIB') or Path(__file__).resolve().parent / 'libdpipe_hip.so')

BF16, F32 = 0, 1
ABI_VERSION = 10                     # DPIPE_ABI_VERSION of include/dpipe_hip.h this binding was written against
CONV_OUT_F32, CONV_ACCUMULATE, CONV_BIAS_PER_SAMPLE, CONV_BIAS_HILO = 1, 2, 4, 8  # dpipe_conv2d_fwd / _dgrad `flags`
OPT_ATTN_FWD_DMA, OPT_ATTN_BWD_DMA, OPT_ATTN_DQ8, OPT_ATTN_DKV_SPLIT, OPT_GEMM_SHALLOW, OPT_GEMM_BIG_TILES = 0, 1, 2, 3, 4, 5     # dpipe_set_option ids (include/dpipe_hip.h)
ACT = {None: 0, 'none': 0, 'gelu_tanh': 1, 'gelu': 2, 'gelu_erf': 2, 'silu': 3, 'quick_gelu': 4}
ACT_GEGLU_BWD = 16                   # DPIPE_ACT_GEGLU_BWD: flag on top of an activation code (the GEGLU backward in a dgrad GEMM's epilogue, ABI 9)
LOSS_KIND = {'mse': 0, 'huber': 1, 'smooth_l1': 2}

P, I, L, F = c_void_p, c_int, c_long, c_float


class GemmDesc(ctypes.Structure):
    """dpipe_gemm_desc of include/dpipe_hip.h (one problem of dpipe_gemm_group), field for field."""
    _fields_ = [('dtype', I), ('transA', I), ('transB', I), ('M', I), ('N', I), ('K', I),
                ('A', P), ('lda', L), ('B', P), ('ldb', L), ('C', P), ('ldc', L),
                ('bias', P), ('act', I), ('alpha', F), ('accumulate', I), ('out_f32', I),
                ('residual', P), ('ldr', L), ('colsum', P), ('colsum_accumulate', I)]


# name -> (restype, argtypes); mirrors include/dpipe_hip.h one to one.
_SIGNATURES = {
    'dpipe_version': (I, []),
    'dpipe_last_error': (c_char_p, []),
    'dpipe_set_option': (I, [I, I]),
    'dpipe_get_option': (I, [I]),
    'dpipe_device_info': (I, [I, POINTER(c_int), c_char_p, I]),
    'dpipe_comm_unique_id': (I, [P]),
    'dpipe_comm_init': (I, [POINTER(c_void_p), I, I, P]),
    'dpipe_comm_destroy': (I, [P]),
    'dpipe_mark_post': (I, [P, P, P]),
    'dpipe_mark_wait': (I, [P, ctypes.c_uint, P, I, P]),
    'dpipe_group_start': (I, []),
    'dpipe_group_end': (I, []),
    'dpipe_send': (I, [P, P, L, I, P]),
    'dpipe_recv': (I, [P, P, L, I, P]),
    'dpipe_loss_workspace_floats': (I, [L, L]),
    'dpipe_loss_fwd': (I, [P, I, P, P, P, L, L, I, F, P, P, P, P]),
    'dpipe_loss_bwd': (I, [P, I, P, P, P, P, L, L, I, F, P, P]),
    'dpipe_act_fwd': (I, [P, P, L, I, I, P]),
    'dpipe_upsample2x_adjoint': (I, [P, P, I, I, I, I, I, P]),
    'dpipe_rowsplit_fwd': (I, [P, P, L, I, P]),
    'dpipe_rowsplit_bwd': (I, [P, P, P, L, I, P]),
    'dpipe_rowcombine_fwd': (I, [P, P, P, P, P, I, I, P]),
    'dpipe_rowcombine_bwd': (I, [P, I, P, P, I, P, I, I, I, P]),
    'dpipe_act_bwd': (I, [P, P, P, L, I, I, P]),
    'dpipe_geglu_fwd': (I, [P, P, L, L, I, I, P]),
    'dpipe_geglu_bwd': (I, [P, P, P, L, L, I, I, P]),
    'dpipe_gated_residual_fwd': (I, [P, P, P, P, L, L, L, I, I, P]),
    'dpipe_gated_residual_slabs': (I, [L]),
    'dpipe_gated_residual_bwd': (I, [P, P, P, P, P, P, L, L, L, I, I, P]),
    'dpipe_sinusoidal_embed': (I, [P, P, L, I, F, I, F, F, P]),
    'dpipe_flow_match_prep': (I, [P, P, P, P, P, L, L, P]),
    'dpipe_multi_sumsq': (I, [P, P, P, P, I, I, P, P, P]),
    'dpipe_multi_clip_scale': (I, [P, P, P, P, I, I, P, F, P]),
    'dpipe_adamw_sumsq': (I, [P, I, P, P, P, I, I, P, P, I, P]),
    'dpipe_adamw_step': (I, [P, P, P, P, I, P, P, P, I, I, F, F, F, F, F, F, F, P, F, I, P]),
    'dpipe_adamw_step_kahan': (I, [P, P, P, P, P, I, P, P, P, I, I, F, F, F, F, F, F, F, P, F, I, P]),
    'dpipe_adamw8bit_step': (I, [P, P, P, P, P, P, P, P, P, L, F, F, F, F, F, I, F, I, P]),
    'dpipe_adamw8bit_multi': (I, [P, P, P, P, P, P, P, P, P, P, I, P, P, F, F, F, F, F, I, F, I, P]),
    'dpipe_rmsnorm_fwd': (I, [P, P, P, P, L, I, F, I, I, P]),
    'dpipe_norm_slabs': (I, [L]),
    'dpipe_rmsnorm_bwd': (I, [P, P, P, P, P, P, P, L, I, I, I, I, P]),
    'dpipe_rmsnorm_rope_fwd': (I, [P, P, P, P, P, P, L, I, I, L, I, L, L, L, F, I, I, P]),
    'dpipe_rmsnorm_rope_bwd': (I, [P, P, P, P, P, P, P, P, P, L, I, I, L, I, L, L, L, I, I, I, P]),
    'dpipe_colsum': (I, [P, L, I, L, P, P, I, I, I, P]),
    'dpipe_lnmod_fwd': (I, [P, P, P, P, P, P, P, P, L, I, L, F, I, I, I, P]),
    'dpipe_lnmod_workspace_floats': (I, [L, I, L]),
    'dpipe_lnmod_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, L, I, I, I, I, P, P]),
    'dpipe_groupnorm_workspace_floats': (L, [L, I, L, I]),
    'dpipe_groupnorm_fwd': (I, [P, P, P, P, P, P, P, L, I, L, I, F, I, I, I, P]),
    'dpipe_groupnorm_bwd': (I, [P, P, P, P, P, P, P, P, P, P, L, I, L, I, I, I, I, I, P, P]),
    'dpipe_conv2d_fwd': (I, [P, L, P, P, P, L, P, L] + [I] * 12 + [P, L, I, P]),
    'dpipe_conv2d_dgrad': (I, [P, L, P, P, L] + [I] * 10 + [P, L, I, P]),
    'dpipe_conv2d_wgrad': (I, [P, L, P, L, P, P] + [I] * 13 + [P, L, I, P]),
    'dpipe_groupnorm_nhwc_workspace_floats': (L, [L, I, L, I]),
    'dpipe_groupnorm_nhwc_fwd': (I, [P, P, P, P, P, P, P, L, I, L, I, F, I, I, I, P]),
    'dpipe_groupnorm_nhwc_bwd': (I, [P, P, P, P, P, P, P, P, P, P, L, I, L, I, I, I, I, I, P, P]),
    'dpipe_rope': (I, [P, P, P, P, L, L, L, I, I, I, I, P]),
    'dpipe_softmax_fwd': (I, [P, P, L, I, L, F, I, I, P]),
    'dpipe_softmax_bwd': (I, [P, P, P, L, I, L, F, I, P]),
    'dpipe_transpose': (I, [P, P, I, I, L, L, L, L, I, I, P]),
    'dpipe_gemm': (I, [I, I, I, I, I, I, P, L, P, L, P, L, I, I, L, L, L, L, L, L, P, I, F, I, I, I, P]),
    'dpipe_gemm_ex': (I, [I, I, I, I, I, I, P, L, P, L, P, L, I, I, L, L, L, L, L, L, P, I, F, I, I, I, P, L, P, L, P, I, P]),
    'dpipe_gemm_group': (I, [POINTER(GemmDesc), I, P, L, POINTER(c_int), P]),
    'dpipe_gemm_group_plan': (I, [P, I, L, P, P, P]),
    'dpipe_tr16_probe': (I, [P, P, P]),
    'dpipe_attn_fwd': (I, [P, P, P, P, P, P, I, I, I, I, I] + [L] * 12 + [F, I, P, P]),
    'dpipe_attn_bwd_partial_floats': (L, [I, I, I, I, I]),
    'dpipe_attn_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I] + [L] * 24 + [F, I, P, L, P, P]),
}

_lib = None


class DpipeHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises loudly when the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise DpipeHipError(
            f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
            f'Build it with `python -m diffusion_pipe_amd.build`.')
    handle = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise DpipeHipError(f'{LIB_PATH} does not export {name}; rebuild the extension') from e
        fn.restype = res
        fn.argtypes = args
    got = handle.dpipe_version()
    if got != ABI_VERSION:         # e.g. an older build named by DPIPE_HIP_LIB: its entry points would read shifted arguments
        raise DpipeHipError(f'{LIB_PATH} has C-ABI version {got}, this binding needs {ABI_VERSION}; rebuild the extension')
    _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what=''):
    if rc != 0:
        msg = lib().dpipe_last_error()
        raise DpipeHipError(f'{what} failed (rc={rc}): {msg.decode() if msg else ""}')


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise DpipeHipError(f'unsupported dtype {dt}: the HIP layer computes in bf16 or fp32')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DpipeHipError('HIP kernels need device tensors; got a CPU tensor (there is no CPU fallback)')


def aligned16(t):
    return t.data_ptr() % 16 == 0


def device_info(dev=0):
    cu = c_int(0)
    buf = ctypes.create_string_buffer(256)
    check(lib().dpipe_device_info(dev, ctypes.byref(cu), buf, 256), 'dpipe_device_info')
    return cu.value, buf.value.decode()
