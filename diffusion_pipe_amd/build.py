"""Build libdpipe_hip.so (the C-ABI HIP layer) for gfx950 with hipcc, in-tree.

`python -m diffusion_pipe_amd.build` compiles every csrc/*.hip into objects under csrc/_build/ and links
diffusion_pipe_amd/libdpipe_hip.so.  hipcc cross-compiles without a GPU, so this runs in the CPU container;
the resulting .so travels with the repo snapshot to the GPU box.  Only stale objects are rebuilt.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / 'csrc'
BUILD = CSRC / '_build'
LIB_PATH = PKG_DIR / 'libdpipe_hip.so'
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CFLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-ffp-contract=fast']
# The GEMM / convolution template's `#pragma unroll` nests over the accumulator tiles must ALL unroll (a loop left rolled indexes the accumulators dynamically and they
# move to scratch): the 256^2 tiles exceed LLVM's default pragma-unroll cost cap (16 384).  Round 5: with the default cap the 8-wave 256^2 kernels (the DiT-sized GEMMs of
# Flux / Wan / HunyuanVideo) carried 100 - 324 bytes of scratch per lane, the 4-wave 256^2 tile 1 088; with the cap raised: 0 - 8.  Every other instantiation is unchanged
# (same register counts, profiles/r5_kernel_resources.csv).
_UNROLL = ['-mllvm', '-pragma-unroll-threshold=200000']
EXTRA_CFLAGS = {'gemm_pipe.hip': _UNROLL, 'gemm_pipe_group.hip': _UNROLL, 'gemm_pipe_256.hip': _UNROLL, 'conv_pipe.hip': _UNROLL}


def _sources():
    return sorted(CSRC.glob('*.hip'))


def _deps_mtime():
    hdrs = list(CSRC.glob('*.h')) + [PKG_DIR.parent / 'include' / 'dpipe_hip.h']
    return max(h.stat().st_mtime for h in hdrs if h.exists())


def _compile(src: Path, force: bool) -> Path:
    obj = BUILD / (src.stem + '.o')
    newest = max(src.stat().st_mtime, _deps_mtime())
    if not force and obj.exists() and obj.stat().st_mtime >= newest:
        return obj
    cmd = [HIPCC, *CFLAGS, *EXTRA_CFLAGS.get(src.name, []), '-c', str(src), '-o', str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile and link the HIP library; returns the path of the shared object."""
    BUILD.mkdir(exist_ok=True)
    srcs = _sources()
    if not srcs:
        raise RuntimeError(f'no HIP sources under {CSRC}')
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest_obj:
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', str(LIB_PATH), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'[dpipe build] {LIB_PATH} ({LIB_PATH.stat().st_size // 1024} KiB, {len(objs)} objects, arch {ARCH})')
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
