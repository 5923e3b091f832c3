#!/usr/bin/env python3
"""Stand-alone 32-channel linear warp: z-run register kernel (variant 3) against the wave-cache kernel (variant 10) over displacement
fields of rising steepness, with the statistic the library's field probe uses to choose between them: the fraction of z-neighbour
output pairs whose corner planes can be re-used (same floor in x and y, floor in z advanced by one).
    python tools/interpn_field_sweep.py [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne                  # noqa: E402
from neurite_amd import deferred, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(B, 160, 32, device=dev)
deferred.enabled = False


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def reuse_fraction(f):
    g = torch.stack(torch.meshgrid(*[torch.arange(160, device=dev, dtype=torch.float32)] * 3, indexing='ij'), -1)
    fl = torch.floor(torch.clamp(g + f, 0, 159)).int()
    a, b = fl[:, :, :, :-1], fl[:, :, :, 1:]
    ok = (a[..., 0] == b[..., 0]) & (a[..., 1] == b[..., 1]) & (b[..., 2] == a[..., 2] + 1)
    return float(ok.float().mean())


for scale in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.65, 0.8, 1.0, 1.5):
    f = (trf * scale).contiguous()
    row = {'batch': B, 'field_scale': scale, 'mean_abs_dudz': round(float((f[:, :, :, 1:] - f[:, :, :, :-1]).abs().mean()), 4),
           'reuse_fraction': round(reuse_fraction(f), 4)}
    for name, variant, tune in (('zrun', 3, 20 | (1 << 16)), ('wc', 10, 0), ('default', 0, 0)):
        st = ne.layers.SpatialTransformer()
        st._variant, st._tune = variant, tune
        row['ms_' + name] = round(timeit(lambda: st([mov, f])), 4)
    print(json.dumps(row), flush=True)
