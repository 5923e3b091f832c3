"""fused kernel with the warped volume written (x-march schedule + store) against the stand-alone interpn kernel"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(4, 160, 32, device=dev, seed0=100)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


st = ne.layers.SpatialTransformer(interp_method='linear')
ne.deferred.enabled = False
print('interpn (drop-in)          %.4f ms' % timeit(lambda: st([mov, trf])))
print('fused, no store            %.4f ms' % timeit(lambda: ne.fused.warp_dice(mov, trf, fix)))
print('fused + store (x-march)    %.4f ms' % timeit(lambda: ne.fused.warp_dice(mov, trf, fix, return_warped=True)))
