"""Few-channel linear warp, variant 8 (lean tile) against variant 9 (wave-autonomous LDS box), on fields of different gradient.
The default bench field (synth.smooth_displacement: 20^3 control points, sigma 3) changes by ~0.5 voxel per voxel."""
import json, sys, torch
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

S, B = 160, 4
fields = {'coarse20_sigma3 (bench)': dict(coarse=20, sigma=3.0), 'coarse10_sigma3': dict(coarse=10, sigma=3.0),
          'coarse5_sigma3': dict(coarse=5, sigma=3.0), 'coarse5_sigma1': dict(coarse=5, sigma=1.0)}
for name, kw in fields.items():
    flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev, **kw) for b in range(B)])
    grad = float((flow[:, 1:] - flow[:, :-1]).abs().mean())
    for C in (1, 2, 3, 4):
        vol = torch.randn(B, S, S, S, C, device=dev)
        row = {'field': name, 'mean_abs_gradient': round(grad, 3), 'C': C}
        for variant in (8, 9):
            st = ne.layers.SpatialTransformer()
            st._variant = variant
            ms = timeit(lambda: ne.deferred.materialize(st([vol, flow])))
            nbytes = B * S ** 3 * (8 * C + 12)
            row['v%d_ms' % variant] = round(ms, 4)
            row['v%d_frac' % variant] = round(nbytes / ms / 1e6 / 8000, 3)
        print(json.dumps(row))
