#!/usr/bin/env python3
"""CPU model of route (a): the wave's row cache as a FIFO ring filled by LDS-DMA (rows land in consecutive ring rows in fetch order; a
direct-mapped tag table maps a row id to its ring position).  A reference hits when its tag names its row AND the row is young enough not
to be overwritten before the pass that uses it has blended: age + n(this pass) + RESERVE <= R, where age = rows allocated since the row was
fetched and RESERVE is what the next pass may allocate before this one blends (64 worst case).  Counts rows fetched per voxel on the bench
field for ring sizes / reserves, next to the shipped direct-mapped cache (128 slots + 16 overflow).
    python tools/lab/wc_fifo_sim.py"""
import itertools, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurite_amd import synth
S = 160
nw = int(os.environ.get('NW', '60'))
trf = synth.smooth_displacement(102, S).numpy()
grid = np.stack(np.meshgrid(*[np.arange(S, dtype=np.float32)] * 3, indexing='ij'), -1)
loc = grid + trf
mx = np.float32(S - 1)
l0 = np.clip(np.floor(loc), 0, mx); l1 = np.clip(l0 + 1, 0, mx)
I = np.stack([l0.astype(np.int64), l1.astype(np.int64)], 0)


def refs(x, y, z):
    return [(int(I[cx, x, y, z, 0]), int(I[cy, x, y, z, 1]), int(I[cz, x, y, z, 2])) for cx, cy, cz in itertools.product((0, 1), repeat=3)]


def h(r):
    return ((r[0] & 3) << 5) | ((r[1] & 3) << 3) | (r[2] & 7)


def run(R, reserve, seed=0):
    rng = np.random.default_rng(seed)
    tv = tf = npass = gt32 = 0
    for _ in range(nw):
        y0 = int(rng.integers(0, S // 2)) * 2; z0 = int(rng.integers(0, S // 4)) * 4
        tags = {}            # slot -> (row, position counter)
        head = 0
        for x in range(S):
            rows = []
            for b in range(2):
                for c in range(4):
                    rows += refs(x, y0 + b, z0 + c)
            uniq = []
            for r in rows:
                if r not in uniq:
                    uniq.append(r)
            # first estimate of n: misses under the liveness rule need n itself -- iterate once (n only shrinks the window)
            def misses(n):
                return [r for r in uniq if not (tags.get(h(r), (None, 0))[0] == r and (head - tags[h(r)][1]) + n + reserve <= R)]
            n = len(misses(0))
            m = misses(n)
            n = len(m)
            # slot conflicts inside the pass: every missing row gets its own ring row, the tag keeps the last writer
            for k, r in enumerate(m):
                tags[h(r)] = (r, head + k)
            head += n
            tv += 8; tf += n; npass += 1; gt32 += n > 32
    return {'ring_rows': R, 'reserve': reserve, 'fetched_per_voxel': round(tf / tv, 3), 'mean_n': round(tf / npass, 2), 'p_n_gt32': round(gt32 / npass, 3)}


for R, res in ((144, 64), (144, 40), (144, 32), (128, 64), (128, 32), (192, 64), (256, 64), (100000, 0)):
    print(json.dumps(run(R, res)), flush=True)
