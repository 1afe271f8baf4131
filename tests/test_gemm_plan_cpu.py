"""The GEMM dispatcher's tile policy, held through `dpipe_gemm_group_plan` (C ABI 8: the plan of a call without launching anything -- host code only, so this runs on the
CPU box).  The headline's GEMM time depends on these rules:
  * the register-staged 128^2 tile (code 132) is what the dispatcher picks for forward (NT) / dgrad (NN) problems whose workgroups walk >= 8 K-steps -- and never for the
    wgrad layout (TN, A is MN-contiguous), which stays on the LDS-DMA rings;
  * under the engine's lane policy (shallow rings + big tiles from 16 tiles on) the wgrad of such a Linear lands on the 2-deep ring (129) and the dgrad + wgrad pair leaves
    as ONE launch; without the lane policy a (132, 128) pair is re-planned onto the 3-deep 128^2 ring -- not unified onto 64^2 tiles."""
import ctypes

import pytest

from diffusion_pipe_amd import hip

WS = 64 << 20
BASE = 0x10000            # fake, 16-byte aligned operand addresses: the planner looks at alignment only


def _desc(ta, tb, M, N, K):
    lda = M if ta else K
    ldb = K if tb else N
    return hip.GemmDesc(hip.BF16, ta, tb, M, N, K, BASE, lda, BASE * 2, ldb, BASE * 3, N, None, 0, 1.0, 0, 0, None, 0, None, 0)


def _plan(descs):
    arr = (hip.GemmDesc * len(descs))(*descs)
    tiles, splits, launches = (ctypes.c_int * len(descs))(), (ctypes.c_int * len(descs))(), ctypes.c_int(0)
    rc = hip.lib().dpipe_gemm_group_plan(arr, len(descs), WS, tiles, splits, ctypes.byref(launches))
    assert rc == 0, hip.lib().dpipe_last_error()
    return list(tiles), list(splits), launches.value


@pytest.fixture
def options():
    lib = hip.lib()
    saved = [(o, lib.dpipe_get_option(o)) for o in (hip.OPT_GEMM_SHALLOW, hip.OPT_GEMM_BIG_TILES)]

    def set_(shallow, big):
        lib.dpipe_set_option(hip.OPT_GEMM_SHALLOW, shallow)
        lib.dpipe_set_option(hip.OPT_GEMM_BIG_TILES, big)
    yield set_
    for o, v in saved:
        lib.dpipe_set_option(o, v)


# (tokens, in_features, out_features) of SDXL Linear layers at micro-batch 1: attention projections / FF at 64 x 64 and 32 x 32 latents
LINEARS = [(4096, 640, 640), (4096, 640, 5120), (4096, 2560, 640), (1024, 1280, 1280), (1024, 1280, 10240), (1024, 5120, 1280)]


@pytest.mark.parametrize('tokens,fin,fout', LINEARS)
def test_register_staged_tile_is_chosen_for_forward_and_dgrad_not_for_wgrad(options, tokens, fin, fout):
    options(2, 16)                                              # the engine's policy with >= 2 lanes (engine.py: gemm_shallow_rings, gemm_big_tiles)
    fwd, _, _ = _plan([_desc(0, 1, tokens, fout, fin)])         # y = x W^T
    dgrad, _, _ = _plan([_desc(0, 0, tokens, fin, fout)])       # dx = dy W
    wgrad, _, _ = _plan([_desc(1, 0, fout, fin, tokens)])       # dW = dy^T x
    assert fwd == [132] and dgrad == [132], (fwd, dgrad)        # K-contiguous A, >= 8 K-steps per workgroup
    assert wgrad[0] in (128, 129), wgrad                        # the wgrad layout never takes the register-staged tile
    # ... and the backward pair leaves as one launch (the 132 member shares T128R2's geometry)
    tiles, _, launches = _plan([_desc(0, 0, tokens, fin, fout), _desc(1, 0, fout, fin, tokens)])
    assert launches == 1 and tiles[0] == 132 and tiles[1] == 129, (tiles, launches)


def test_short_k_walks_stay_on_the_dma_rings(options):
    options(2, 16)
    tiles, splits, _ = _plan([_desc(0, 1, 4096, 640, 320)])     # 5 K-steps: below the register-staged tile's 8
    assert tiles == [129] and splits == [1]
    # split tiles count the K-steps a workgroup walks, not the problem's
    tiles, splits, _ = _plan([_desc(0, 0, 1024, 1280, 10240)])  # 160 K-steps in three slices
    assert tiles == [132] and splits[0] >= 2 and 160 // splits[0] >= 8


def test_mixed_pair_without_lane_policy_keeps_128_tiles(options):
    """single-lane engines (no shallow-ring option): the wgrad of a 128 .. 255-tile Linear plans onto the 3-deep ring (128), its dgrad onto the register-staged tile (132) --
    the pair must stay on 128^2 tiles as one T128 launch (round 5 unified it onto 64^2 tiles: slower, and a different summation order)"""
    options(0, 128)
    M, fin, fout = 4096, 640, 5120                              # the GEGLU projection's backward: 160 tiles (dgrad), 200 tiles (wgrad)
    alone_d, _, _ = _plan([_desc(0, 0, M, fin, fout)])
    alone_w, _, _ = _plan([_desc(1, 0, fout, fin, M)])
    if not (alone_d == [132] and alone_w == [128]):
        pytest.skip(f'this shape no longer plans as the (132, 128) pair: {alone_d}, {alone_w}')
    tiles, _, launches = _plan([_desc(0, 0, M, fin, fout), _desc(1, 0, fout, fin, M)])
    assert tiles == [128, 128] and launches == 1, (tiles, launches)


def test_pair_with_a_small_member_still_unifies_onto_64(options):
    options(0, 128)
    tiles, _, launches = _plan([_desc(0, 0, 1024, 1280, 1280), _desc(1, 0, 1280, 1280, 1024)])
    assert launches == 1 and len(set(128 if t in (128, 129, 132) else t for t in tiles)) == 1, (tiles, launches)
