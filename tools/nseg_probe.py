#!/usr/bin/env python3
"""x segments of the x-march schedule at the bench batch: blocks = 800 patches x batch x nseg over 512 resident block slots (2 per CU):
3200 blocks run in 6.25 rounds of full-length marches -- the last round is a quarter full.   python tools/nseg_probe.py  (GPU box)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
XM = 3 | (2 << 4) | (3 << 8) | (1 << 14) | (3 << 24) | (2 << 27)
WC, NO_WC = 1 << 29, 1 << 30


def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for batch in (4, 1):
    mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
    for rep in range(2):
        for nseg in (0, 1, 2, 4, 5, 8, 10):
            row = {'batch': batch, 'nseg': nseg, 'blocks': 800 * batch * max(nseg, 1), 'rep': rep}
            for name, bit in (('reg', NO_WC), ('wc', WC)):
                tune = (XM | (nseg << 16) | bit) if nseg else bit
                row['ms_' + name] = round(timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=tune)), 4)
            print(json.dumps(row), flush=True)
