"""CPU: the C-ABI library loads and exports exactly what include/dpipe_hip.h declares (no compute calls without a GPU),
the ctypes table mirrors the header, and the product refuses to run without the HIP layer (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'dpipe_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'^\s*(?:int|long|const char\*)\s+(dpipe_\w+)\s*\(', text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from diffusion_pipe_amd import hip
    lib = hip.lib()                                  # built in-tree by __graft_entry__.build(); raises if missing
    declared = _header_symbols()
    assert len(declared) >= 30
    raw = ctypes.CDLL(str(hip.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name), f'{name} declared in include/dpipe_hip.h but not exported'
    assert sorted(hip.exported_symbols()) == declared, 'ctypes signature table and header differ'
    assert lib.dpipe_version() >= 1


def test_argument_errors_are_reported_without_a_gpu():
    from diffusion_pipe_amd import hip
    lib = hip.lib()
    rc = lib.dpipe_gemm(0, 0, 1, 0, 16, 16, None, 16, None, 16, None, 16, 1, 1, 0, 0, 0, 0, 0, 0, None, 0, 1.0, 0, 0, 0, None)
    assert rc != 0 and b'dpipe_gemm' in lib.dpipe_last_error()


def test_no_cpu_fallback():
    from diffusion_pipe_amd import ops
    from diffusion_pipe_amd.hip import DpipeHipError
    x = torch.randn(4, 8)
    w = torch.randn(8, 8)
    with pytest.raises(DpipeHipError):
        ops.linear(x, w)
    with pytest.raises(DpipeHipError):
        ops.rms_norm(x)
