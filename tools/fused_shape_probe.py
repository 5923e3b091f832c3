"""L1 set-conflict probe for the fused warp + Dice kernel: the same bench field and maps, the moving volume embedded in a buffer
whose y / z extents are padded (row strides no longer multiples of 64 lines), and the patch shapes of the x-march.
usage: python tools/fused_shape_probe.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
from neurite_amd import synth

dev = torch.device('cuda:0')
B, S, L = 4, 160, 32
mov = torch.stack([synth.one_hot_volume(1 + b, S, L, dev) for b in range(B)])
fix = torch.stack([synth.one_hot_volume(101 + b, S, L, dev) for b in range(B)])
trf = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])


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


XM = 1 << 14
REG = (3 << 24) | (2 << 27)
ref = ne.fused.warp_dice(mov, trf, fix)
for (sy, sz) in (((160, 160), (160, 161), (161, 161), (162, 176), (160, 176), (164, 168), (160, 192)) if '--shapes' in sys.argv else ()):
    big = torch.zeros((B, S, sy, sz, L), dtype=torch.float32, device=dev)
    big[:, :, :S, :S] = mov
    t = timeit(lambda: ne.fused.warp_dice(big, trf, fix))
    d = ne.fused.warp_dice(big, trf, fix)
    print(json.dumps({'vol_shape': [S, sy, sz], 'y_stride_mod64': sz % 64, 'x_stride_mod64': (sy * sz) % 64, 'ms': round(t, 4),
                      'same_dice_as_160': bool(torch.equal(d, ref)) if (sy, sz) == (160, 160) else float((d - ref).abs().max())}), flush=True)
    del big
for lty, ltz, lry, lrz in ((2, 3, 3, 2), (3, 2, 2, 3), (3, 2, 3, 2), (2, 3, 2, 3), (2, 3, 2, 2), (2, 3, 3, 3), (2, 3, 3, 2)):
    tune = 3 | (lty << 4) | (ltz << 8) | XM | (lry << 24) | (lrz << 27)
    try:
        t = timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=tune))
        print(json.dumps({'patch_yz': [1 << lty, 1 << ltz], 'region_patches_yz': [1 << lry, 1 << lrz], 'ms': round(t, 4)}), flush=True)
    except Exception as ex:
        print(json.dumps({'patch_yz': [1 << lty, 1 << ltz], 'error': str(ex)[:100]}), flush=True)
