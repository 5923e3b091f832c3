#!/usr/bin/env python3
"""CPU model: the wave-private row cache when a wave visits SEVERAL 2 x 4 sub-patches per x-plane (zig-zag march).

Counts per voxel on the bench field: rows fetched by a wave (cache misses + orphans), orphan references, the share of passes
whose fetch list exceeds 32 entries, and the distinct rows a BLOCK of 2 x 2 waves needs (what HBM would deliver if the block's
waves shared perfectly through L1 / L2 and nothing were shared between blocks).
    python tools/lab/wc_zigzag_sim.py
"""
import itertools, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurite_amd import synth

S = 160
nb = int(os.environ.get('NB', '24'))
trf = synth.smooth_displacement(102, S).numpy()
grid = np.stack(np.meshgrid(*[np.arange(S, dtype=np.float32)] * 3, indexing='ij'), -1)
loc = grid + trf
mx = np.float32(S - 1)
l0 = np.clip(np.floor(loc), 0, mx); l1 = np.clip(l0 + 1, 0, mx)
i0 = l0.astype(np.int64); i1 = l1.astype(np.int64)
I = np.stack([i0, i1], 0)          # [2, X, Y, Z, 3]


def refs(x, y, z):
    out = []
    for cx, cy, cz in itertools.product((0, 1), repeat=3):
        out.append((int(I[cx, x, y, z, 0]), int(I[cy, x, y, z, 1]), int(I[cz, x, y, z, 2])))
    return out


def run(subs, hbits, wave_grid=(2, 2), seed=0):
    """subs: list of (dy, dz) offsets of the 2 x 4 sub-patches a wave visits per x-plane; the wave's patch is their union.
    wave_grid: waves of a block along (y, z)."""
    rng = np.random.default_rng(seed)
    bx, by, bz = hbits
    ys = [d[0] for d in subs]; zs = [d[1] for d in subs]
    wy, wz = max(ys) + 2, max(zs) + 4                      # wave patch
    By, Bz = wy * wave_grid[0], wz * wave_grid[1]
    def h(r):
        return ((r[0] & ((1 << bx) - 1)) << (by + bz)) | ((r[1] & ((1 << by) - 1)) << bz) | (r[2] & ((1 << bz) - 1))
    tv = tf = to = np_ = ngt32 = 0
    blk_rows = 0
    for _ in range(nb):
        Y0 = int(rng.integers(0, S // By)) * By; Z0 = int(rng.integers(0, S // Bz)) * Bz
        blockset = set()
        for wyi in range(wave_grid[0]):
            for wzi in range(wave_grid[1]):
                y0 = Y0 + wyi * wy; z0 = Z0 + wzi * wz
                tags = {}
                for x in range(S):
                    for (dy, dz) in subs:
                        rows = []
                        for b in range(2):
                            for c in range(4):
                                rows += refs(x, y0 + dy + b, z0 + dz + c)
                        tv += 8; np_ += 1
                        claim = {}
                        for r in rows:
                            if tags.get(h(r)) != r: claim[h(r)] = r
                        for k, r in claim.items(): tags[k] = r
                        no = sum(1 for r in rows if tags.get(h(r)) != r)
                        # orphans that name the same row still fetch separately in the kernel
                        n = len(claim) + no
                        tf += n; to += no
                        ngt32 += n > 32
                        blockset.update(rows)
        blk_rows += len(blockset)
    return {'subs': subs, 'hash_bits': hbits, 'wave_patch': (wy, wz), 'block_patch': (By, Bz),
            'fetched_per_voxel': round(tf / tv, 3), 'orphans_per_voxel': round(to / tv, 3), 'mean_n': round(tf / np_, 2),
            'p_n_gt32': round(ngt32 / np_, 3), 'block_distinct_rows_per_voxel': round(blk_rows / (nb * By * Bz * S), 3)}


cases = [
    ([(0, 0)], (2, 2, 3)),                       # shipped
    ([(0, 0), (2, 0)], (2, 2, 3)),               # y zig-zag, wave 4 x 4, block 8 x 8
    ([(0, 0), (2, 0)], (2, 3, 2)),
    ([(0, 0), (2, 0)], (1, 3, 3)),
    ([(0, 0), (0, 4)], (2, 2, 3)),               # z zig-zag, wave 2 x 8, block 4 x 16
    ([(0, 0), (0, 4)], (2, 1, 4)),
    ([(0, 0), (0, 4)], (1, 2, 4)),
    ([(0, 0), (2, 0), (2, 4), (0, 4)], (1, 3, 3)),   # wave 4 x 8, block 8 x 16
    ([(0, 0), (2, 0), (2, 4), (0, 4)], (1, 2, 4)),
]
import ast
if os.environ.get('WC_CASES'): cases = ast.literal_eval(os.environ['WC_CASES'])
for subs, hb in cases:
    print(json.dumps(run(subs, hb)), flush=True)
