"""batch 1 (BASELINE config 2 as written): x segments of the x-march schedule, fused kernel"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(1, 160, 32, device=dev, seed0=100)


def timeit(fn, n=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


base = 3 | (2 << 4) | (3 << 8) | (1 << 14) | (3 << 24) | (2 << 27)
for rep in range(2):
    for nseg in (0, 2, 3, 4, 5, 6, 8, 10, 16):
        t = timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=base | (nseg << 16)))
        print('nseg %2d (0 = auto): %.4f ms per step (kernel + second stage)' % (nseg, t), flush=True)
