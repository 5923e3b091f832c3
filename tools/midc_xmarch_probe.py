"""mid-channel warps (C = 8, 16): the fused kernel on the x-march schedule with the warped volume written, against the drop-in interpn"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
B, S = 4, 160


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


trf = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
st = ne.layers.SpatialTransformer(interp_method='linear')
ne.deferred.enabled = False
for C in (16, 8, 32):
    mov = torch.rand(B, S, S, S, C, device=dev)
    fix = torch.rand(B, S, S, S, C, device=dev)
    print('C=%d interpn (drop-in) %.4f ms' % (C, timeit(lambda: st([mov, trf]))), flush=True)
    G = C // 4
    lyz = {8: (2, 3), 4: (3, 3), 2: (3, 4)}[G]
    for lry, lrz in ((3, 2), (2, 2)):
        tune = 3 | (lyz[0] << 4) | (lyz[1] << 8) | (1 << 14) | (lry << 24) | (lrz << 27)
        try:
            t0 = timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=tune))
            t1 = timeit(lambda: ne.fused.warp_dice(mov, trf, fix, return_warped=True, _tune=tune))
            w = ne.fused.warp_dice(mov, trf, fix, return_warped=True, _tune=tune)[1]
            same = bool(torch.equal(w, st([mov, trf])))
            print('C=%d x-march patch %dx%d region %dx%d: fused %.4f ms, fused + store %.4f ms, warped bit-identical %s'
                  % (C, 1 << lyz[0], 1 << lyz[1], 1 << lry, 1 << lrz, t0, t1, same), flush=True)
        except Exception as e:      # noqa
            print('C=%d tune %x failed: %s' % (C, tune, str(e)[:120]), flush=True)
    del mov, fix
