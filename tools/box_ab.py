#!/usr/bin/env python3
"""Few-channel linear warps (C = 1..4, 4 x 160^3, bench field): box form (variant 11, source box of a tile staged in LDS) against tile
form (variant 8, corners through the texture unit), plus VecInt and a registration-like (gentle) field.        (GPU box)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')


def timeit(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


S, B = 160, 4
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
for name, f in (('bench field', flow), ('gentle field (x 0.2)', flow * 0.2), ('rough U(-20, 20)', (torch.rand_like(flow) - 0.5) * 40)):
    for C in (1, 2, 3, 4):
        vol = torch.randn(B, S, S, S, C, device=dev)
        row = {'field': name, 'C': C}
        outs = {}
        for variant in (8, 11):
            st = ne.layers.SpatialTransformer()
            st._variant = variant
            ms = timeit(lambda: ne.deferred.materialize(st([vol, f])))
            outs[variant] = ne.deferred.materialize(st([vol, f]))
            nbytes = B * S ** 3 * (8 * C + 12)
            row['ms_v%d' % variant] = round(ms, 4)
            row['frac_v%d' % variant] = round(nbytes / ms / 1e6 / 8000, 3)
        row['same_bits'] = bool(torch.equal(outs[8], outs[11]))
        print(json.dumps(row), flush=True)
vi = ne.layers.VecInt(int_steps=7)
ms = timeit(lambda: vi(flow), n=5)
nbytes = 7 * B * S ** 3 * 36
print(json.dumps({'op': 'VecInt 7 steps 160^3 (library default form)', 'ms': round(ms, 4), 'frac': round(nbytes / ms / 1e6 / 8000, 3)}))
