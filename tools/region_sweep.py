"""fused warp + Dice on the bench tensors against the region shape of the x-march schedule (tune bits 24-26 = log2 patches per region
along y, 27-29 along z): ms per launch.   python tools/region_sweep.py [batch]   (GPU box)"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(B, 160, 32, device=dev)
base = 3 | (2 << 4) | (3 << 8) | (1 << 14)


def timeit(fn, n=40, warm=8):
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


for rep in range(2):
    for lry, lrz in ((3, 2), (3, 3), (4, 2), (4, 3), (2, 2), (2, 3), (3, 1), (4, 1), (5, 2)):
        t = base | (lry << 24) | (lrz << 27)
        ms = timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=t))
        print(json.dumps({'batch': B, 'region_patches_y_z': [1 << lry, 1 << lrz], 'region_voxels_y_z': [4 << lry, 8 << lrz], 'ms': round(ms, 4),
                          'frac': round(B * 160 ** 3 * 268 / ms / 1e6 / 8000, 4)}), flush=True)
