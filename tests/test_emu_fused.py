"""
CPU tests of kernel LOGIC through the thread-block emulator (tests/emu): neurite_amd/csrc/fused.hip is compiled for the
host and its C entry point runs on 256 OS threads per block (real barriers, wave shuffles through per-wave buffers, LDS
atomics as host atomics).  First the x-march tile kernel -- validated on hardware by the -m gpu tests -- is checked against
the oracle, which validates the emulator; then the experimental de-duplicating schedule (tune bit 30), which has not run on
hardware yet, must reproduce the tile kernel's result bit for bit (same voxels per block, same accumulation order) on
smooth, ragged and incoherent inputs (hash overflow -> whole-window direct path), with and without fill.
This is not a substitute for the GPU parity tests: the emulator does not model timing or hardware memory ordering.
"""

import ctypes
import os
import sys

import numpy as np
import pytest

from oracle import np_oracle as npo

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

F = np.float32
XM = 3 | (2 << 4) | (3 << 8) | (1 << 14)            # x-march, 4 x 8 (y, z) patches
DEDUP = 1 << 30


@pytest.fixture(scope='module')
def emu():
    if not build_emu.available():
        pytest.skip('host clang++ of the ROCm toolchain not available')
    lib = ctypes.CDLL(build_emu.build())
    lib.nrt_warp_dice_workspace_bytes.restype = ctypes.c_size_t
    lib.nrt_warp_dice_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.nrt_warp_dice_soft_f32.restype = ctypes.c_int
    lib.nrt_warp_dice_soft_f32.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.emu_take_max_shmem.restype = ctypes.c_size_t
    return lib


def warp_dice(lib, mov, trf, fix, tune, fill=None):
    B, S, L = mov.shape[0], mov.shape[1:4], mov.shape[-1]
    shape = (ctypes.c_int * 3)(*S)
    mov, trf, fix = (np.ascontiguousarray(a, F) for a in (mov, trf, fix))
    nws = lib.nrt_warp_dice_workspace_bytes(shape, L, B, tune)
    assert nws > 0
    ws = np.zeros(nws + 64, np.uint8)
    sums, dice, mm = np.zeros((B, 3, L), F), np.zeros((B, L), F), np.zeros(4, F)
    wsp = (ws.ctypes.data + 15) & ~15
    rc = lib.nrt_warp_dice_soft_f32(mov.ctypes.data, trf.ctypes.data, fix.ctypes.data, None, shape, shape, L, B,
                                    int(np.prod(S)) * 3, 1, int(fill is not None), float(fill or 0.0), 0.0, sums.ctypes.data,
                                    dice.ctypes.data, mm.ctypes.data, tune, wsp, nws, None)
    assert rc == 0, rc
    return dice, sums, mm


def npo_rows(trf, S):
    """corner row indices [X, Y, Z, 8] of a shift field (what the kernel hashes)"""
    grid = np.stack(np.meshgrid(*[np.arange(n) for n in S], indexing='ij'), -1).astype(F)
    p = np.clip(grid + trf, 0, np.asarray(S, F) - 1)
    f0 = np.floor(p).astype(np.int64)
    f1 = np.minimum(f0 + 1, np.asarray(S) - 1)
    out = []
    for c in range(8):
        ix = (f1 if c & 4 else f0)[..., 0]; iy = (f1 if c & 2 else f0)[..., 1]; iz = (f1 if c & 1 else f0)[..., 2]
        out.append((ix * S[1] + iy) * S[2] + iz)
    return np.stack(out, -1)


CASES = [
    # batch, shape, field sigma, x segments
    (2, (12, 8, 16), 1.5, 1),           # two 4 x 8 patches per axis, three full windows
    (1, (10, 7, 13), 2.0, 2),           # ragged patches, ragged last window of each segment
    (1, (8, 16, 16), -1.0, 1),          # sigma < 0: locations uniform over the volume -> ~800 distinct rows per window,
                                        # the 512-slot table overflows and the windows take the direct path
    (1, (5, 4, 8), 0.0, 1),             # zero displacement: every corner pair collapses onto few rows
]


@pytest.mark.parametrize('B,S,sigma,nseg', CASES)
def test_dedup_schedule_matches_tile_kernel_and_oracle(emu, B, S, sigma, nseg):
    rng = np.random.default_rng(5)
    L = 32
    mov = rng.random((B,) + S + (L,)).astype(F)
    fix = rng.random((B,) + S + (L,)).astype(F)
    if sigma < 0:
        grid = np.stack(np.meshgrid(*[np.arange(n) for n in S], indexing='ij'), -1).astype(F)
        trf = (rng.random((B,) + S + (3,)) * (np.asarray(S) - 1) - grid).astype(F)
        rows = npo_rows(trf[0], S)
        assert len(np.unique(rows[:4, :4, :8])) > 512                            # first window of the first patch: the table must overflow
    else:
        trf = rng.normal(0, sigma, (B,) + S + (3,)).astype(F) if sigma else np.zeros((B,) + S + (3,), F)
    tune = XM | (nseg << 16)
    for fill in (None, 0.0):
        d_ref = npo.dice(fix, npo.spatial_transformer(mov, trf, fill_value=fill), check_input_limits=False)
        emu.emu_take_max_shmem()
        d_tile, s_tile, mm_tile = warp_dice(emu, mov, trf, fix, tune, fill)
        assert emu.emu_take_max_shmem() == 75 * 1024                           # the x-march tile kernel ran (its LDS padding)
        np.testing.assert_allclose(d_tile, d_ref, rtol=1e-5)                   # the emulator runs the validated kernel correctly
        d_dd, s_dd, mm_dd = warp_dice(emu, mov, trf, fix, tune | DEDUP, fill)
        assert emu.emu_take_max_shmem() == 512 * 128 + 512 * 4                 # warp_dice_dedup ran (row buffer + table)
        assert np.array_equal(s_dd, s_tile) and np.array_equal(d_dd, d_tile)     # same partial sums, bit for bit
        assert np.array_equal(mm_dd, mm_tile)
