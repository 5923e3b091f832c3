#!/usr/bin/env python3
"""fused warp + Dice, ms per volume against the batch: register kernel (tune bit 30) and wave-cache kernel (bit 29).   (GPU box)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')


def timeit(fn, n=20, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for batch in (1, 2, 4, 8, 16, 32):
    mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
    row = {'batch': batch}
    for name, bit in (('reg', 1 << 30), ('wc', 1 << 29)):
        ms = timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=bit), n=10 if batch >= 16 else 20)
        row['ms_' + name] = round(ms, 4)
        row['ms_per_volume_' + name] = round(ms / batch, 4)
        row['frac_' + name] = round(batch * 160 ** 3 * 268 / ms / 1e9 / 8.0, 4)
    print(json.dumps(row), flush=True)
    del mov, fix, trf
    torch.cuda.empty_cache()
