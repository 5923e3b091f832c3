#!/usr/bin/env python3
"""CPU model of a wave-private, direct-mapped LDS row cache for the 32-channel linear gather (no GPU needed).

A wave handles 8 output voxels per pass (a px x py x pz patch) and marches along x; each voxel needs the 8 corner rows of
floor(loc).  The model counts, on the BENCH field (synth.smooth_displacement, sigma 3), per voxel:
  distinct  rows a pass needs after de-duplication inside the pass (what per-pass de-duplication alone would fetch)
  fetched   rows that miss in the wave's cache (direct-mapped, `slots` rows, hash of the low bits of the row coordinates)
  orphans   corner references whose slot is claimed by a different row of the SAME pass (they bypass the cache)
    python tools/wc_sim.py [--waves 400]
"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurite_amd import synth        # noqa: E402

S = 160
nw = int(sys.argv[sys.argv.index('--waves') + 1]) if '--waves' in sys.argv else 300
trf = synth.smooth_displacement(102, S).numpy()
grid = np.stack(np.meshgrid(*[np.arange(S, dtype=np.float32)] * 3, indexing='ij'), -1)
loc = grid + trf
mx = np.float32(S - 1)
l0 = np.clip(np.floor(loc), 0, mx)
l1 = np.clip(l0 + 1, 0, mx)
i0 = l0.astype(np.int32)
i1 = l1.astype(np.int32)
rng = np.random.default_rng(0)


def run(patch, hbits, xlen=160):
    px, py, pz = patch
    bx, by, bz = hbits
    slots = 1 << (bx + by + bz)
    tot_vox = tot_distinct = tot_fetch = tot_orph = 0
    for _ in range(nw):
        y0 = int(rng.integers(0, S // py)) * py
        z0 = int(rng.integers(0, S // pz)) * pz
        tags = np.full(slots, -1, np.int64)
        for x0 in range(0, xlen, px):
            vs = [(x0 + a, y0 + b, z0 + c) for a in range(px) for b in range(py) for c in range(pz)]
            rows = []
            for (x, y, z) in vs:
                for cx, cy, cz in itertools.product((0, 1), repeat=3):
                    ix = (i1 if cx else i0)[x, y, z, 0]
                    iy = (i1 if cy else i0)[x, y, z, 1]
                    iz = (i1 if cz else i0)[x, y, z, 2]
                    rows.append((int(ix), int(iy), int(iz)))
            uniq = set(rows)
            tot_vox += len(vs)
            tot_distinct += len(uniq)
            # phase 1: every distinct row that misses writes its tag; the LAST writer of a slot wins (any fixed order will do)
            claim = {}
            for r in sorted(uniq):
                h = ((r[0] & ((1 << bx) - 1)) << (by + bz)) | ((r[1] & ((1 << by) - 1)) << bz) | (r[2] & ((1 << bz) - 1))
                rid = (r[0] * S + r[1]) * S + r[2]
                if tags[h] != rid:
                    claim[h] = rid
            for h, rid in claim.items():
                tags[h] = rid
                tot_fetch += 1
            # phase 2: a reference whose slot now holds another row is an orphan (fetched past the cache)
            for r in rows:
                h = ((r[0] & ((1 << bx) - 1)) << (by + bz)) | ((r[1] & ((1 << by) - 1)) << bz) | (r[2] & ((1 << bz) - 1))
                rid = (r[0] * S + r[1]) * S + r[2]
                if tags[h] != rid:
                    tot_orph += 1
    return {'patch': patch, 'hash_bits': hbits, 'slots': slots, 'distinct_per_voxel': round(tot_distinct / tot_vox, 3),
            'fetched_per_voxel': round(tot_fetch / tot_vox, 3), 'orphan_refs_per_voxel': round(tot_orph / tot_vox, 3)}


import ast
cases = ast.literal_eval(os.environ.get('WC_CASES', '[((1,2,4),(2,2,3))]'))
for patch, hb in cases:
    print(json.dumps(run(tuple(patch), tuple(hb))), flush=True)


def histogram(patch=(1, 2, 4), hbits=(2, 2, 3), waves=60):
    """distribution of the fetch-list length n = loaders + orphans per pass (what the kernel's rare paths see)"""
    px, py, pz = patch
    bx, by, bz = hbits
    slots = 1 << (bx + by + bz)
    ns, orph = [], []
    for _ in range(waves):
        y0 = int(rng.integers(0, S // py)) * py
        z0 = int(rng.integers(0, S // pz)) * pz
        tags = np.full(slots, -1, np.int64)
        for x0 in range(0, S, px):
            rows = []
            for (x, y, z) in [(x0 + a, y0 + b, z0 + c) for a in range(px) for b in range(py) for c in range(pz)]:
                for cx, cy, cz in itertools.product((0, 1), repeat=3):
                    rows.append((int((i1 if cx else i0)[x, y, z, 0]), int((i1 if cy else i0)[x, y, z, 1]), int((i1 if cz else i0)[x, y, z, 2])))
            hs = [((r[0] & ((1 << bx) - 1)) << (by + bz)) | ((r[1] & ((1 << by) - 1)) << bz) | (r[2] & ((1 << bz) - 1)) for r in rows]
            rids = [(r[0] * S + r[1]) * S + r[2] for r in rows]
            claim = {}
            for h, rid in zip(hs, rids):
                if tags[h] != rid:
                    claim[h] = rid                      # last writer wins
            for h, rid in claim.items():
                tags[h] = rid
            no = sum(1 for h, rid in zip(hs, rids) if tags[h] != rid)
            ns.append(len(claim) + no)
            orph.append(no)
    ns, orph = np.array(ns), np.array(orph)
    return {'mean_n': float(ns.mean()), 'p_n_gt16': float((ns > 16).mean()), 'p_n_gt24': float((ns > 24).mean()), 'p_n_gt32': float((ns > 32).mean()),
            'p_n_gt40': float((ns > 40).mean()), 'p_orph_gt8': float((orph > 8).mean()), 'p_orph_gt16': float((orph > 16).mean()), 'max_n': int(ns.max())}


if '--hist' in sys.argv:
    print(json.dumps(histogram()))
