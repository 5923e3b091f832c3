"""Headline pipeline (SpatialTransformer + Dice, 4 x 160^3 x 32 one-hot float32) against the gradient of the displacement field.
The bench field is SURVEY.md 8d's (20^3 control points, sigma 3); the others keep sigma and coarsen the control grid."""
import json, torch
import neurite_amd as ne
from neurite_amd import synth, fused
dev = torch.device('cuda:0')

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

S, B, L = 160, 4, 32
mov, fix, _ = synth.cfg2_batch(B, S, L, dev)
fields = {'coarse20_sigma3 (bench)': dict(coarse=20, sigma=3.0), 'coarse10_sigma3': dict(coarse=10, sigma=3.0),
          'coarse5_sigma3': dict(coarse=5, sigma=3.0), 'coarse5_sigma1': dict(coarse=5, sigma=1.0)}
st = ne.layers.SpatialTransformer()
for name, kw in fields.items():
    trf = torch.stack([synth.smooth_displacement(102 + 3 * b, S, device=dev, **kw) for b in range(B)])
    grad = float((trf[:, 1:] - trf[:, :-1]).abs().mean())
    ms_f = timeit(lambda: fused.warp_dice(mov, trf, fix))
    ne.deferred.enabled = False
    ms_w = timeit(lambda: st([mov, trf]))
    ne.deferred.enabled = True
    nvox = B * S ** 3
    print(json.dumps({'field': name, 'mean_abs_gradient': round(grad, 3),
                      'fused_ms': round(ms_f, 4), 'fused_Mvox_s': round(nvox / ms_f / 1e3, 1), 'fused_frac': round(nvox * 268 / ms_f / 1e6 / 8000, 3),
                      'warp_ms': round(ms_w, 4), 'warp_frac': round(nvox * 268 / ms_w / 1e6 / 8000, 3)}))
