#!/usr/bin/env python3
"""What a plain device copy of the stand-alone warp's sizes achieves on this box (torch copy_: 2.1 GB read + 2.1 GB written), next to the
warp itself (5.12 GB of HBM-side traffic per launch): the rate at which the part moves a read + write mix."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(4, 160, 32, device=dev)
out = torch.empty_like(mov)
st = ne.layers.SpatialTransformer(interp_method='linear')
ne.deferred.enabled = False


def timeit(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
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


res = {}
for rep in range(2):
    c = timeit(lambda: out.copy_(mov))
    a = timeit(lambda: torch.add(mov, fix, out=out))
    w = timeit(lambda: st([mov, trf]))
    res['rep%d' % rep] = {'copy_ms': round(c, 4), 'copy_TBs': round(2 * mov.numel() * 4 / c / 1e9, 3),
                          'add_2reads_1write_ms': round(a, 4), 'add_TBs': round(3 * mov.numel() * 4 / a / 1e9, 3),
                          'warp_ms': round(w, 4), 'warp_TBs_on_5.117GB': round(5.117 / w, 3)}
print(json.dumps(res))
