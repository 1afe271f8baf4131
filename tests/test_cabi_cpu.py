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


def test_header_constants_match_the_python_binding():
    """ABI version, option ids and flag bits are #defines in the header and plain ints in hip.py: a drift between the two is a silent wrong call."""
    from diffusion_pipe_amd import hip
    text = open(os.path.join(ROOT, 'include', 'dpipe_hip.h')).read()
    defs = {k: int(v) for k, v in re.findall(r'^#define\s+(DPIPE_\w+)\s+(-?\d+)\b', text, flags=re.M)}
    assert defs['DPIPE_ABI_VERSION'] == hip.ABI_VERSION == hip.lib().dpipe_version()
    opts = {'DPIPE_OPT_ATTN_FWD_DMA': hip.OPT_ATTN_FWD_DMA, 'DPIPE_OPT_ATTN_BWD_DMA': hip.OPT_ATTN_BWD_DMA, 'DPIPE_OPT_ATTN_DQ8': hip.OPT_ATTN_DQ8,
            'DPIPE_OPT_ATTN_DKV_SPLIT': hip.OPT_ATTN_DKV_SPLIT, 'DPIPE_OPT_GEMM_SHALLOW': hip.OPT_GEMM_SHALLOW, 'DPIPE_OPT_GEMM_BIG_TILES': hip.OPT_GEMM_BIG_TILES}
    for name, val in opts.items():
        assert defs[name] == val, name
    assert sorted(opts.values()) == list(range(defs['DPIPE_OPTION_COUNT'])), 'an option id of the header has no binding in hip.py'
    assert defs['DPIPE_CONV_OUT_F32'] == hip.CONV_OUT_F32 and defs['DPIPE_CONV_ACCUMULATE'] == hip.CONV_ACCUMULATE
    # options are process-wide state reachable without a GPU: set / get / unset round-trips, unknown ids are refused
    lib = hip.lib()
    assert lib.dpipe_set_option(hip.OPT_GEMM_SHALLOW, 2) == 0 and lib.dpipe_get_option(hip.OPT_GEMM_SHALLOW) == 2
    assert lib.dpipe_set_option(hip.OPT_GEMM_SHALLOW, -1) == 0 and lib.dpipe_get_option(hip.OPT_GEMM_SHALLOW) == int(os.environ.get('DPIPE_GEMM_SHALLOW', '-1'))
    assert lib.dpipe_set_option(defs['DPIPE_OPTION_COUNT'], 1) != 0


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




def test_debug_ablation_switches_are_never_set_by_the_product():
    """csrc/runtime.hip reads DPIPE_DEBUG_ABLATE / DPIPE_DEBUG_GEMM_KDIV (skip a kernel class's launches / shorten the GEMM K loops: the ablation census of DESIGN.md
    section 4.1c -- results are garbage by construction).  Nothing under the package or in bench.py may set or even name them: only tools/run_gpu.sh does."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    offenders = []
    for path in list((root / 'diffusion_pipe_amd').rglob('*.py')) + [root / 'bench.py', root / '__graft_entry__.py']:
        text = path.read_text()
        if 'DPIPE_DEBUG_ABLATE' in text or 'DPIPE_DEBUG_GEMM_KDIV' in text:
            offenders.append(str(path))
    assert not offenders, offenders
    assert 'DPIPE_DEBUG_ABLATE' in (root / 'tools' / 'run_gpu.sh').read_text()
