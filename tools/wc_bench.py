#!/usr/bin/env python3
"""Wave-cache fused kernel (csrc/fused_wc.h) against the register kernel (tune bit 30) on the bench tensors: event-timed ms per launch,
Dice agreement.   python tools/wc_bench.py [--batch 4] [--fields bench,zero,rough]   (GPU box)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neurite_amd as ne                   # noqa: E402
from neurite_amd import synth             # noqa: E402

NO_WC, WC = 1 << 30, 1 << 29
dev = torch.device('cuda:0')
batches = [int(v) for v in (sys.argv[sys.argv.index('--batch') + 1] if '--batch' in sys.argv else '4,1').split(',')]
fields = (sys.argv[sys.argv.index('--fields') + 1] if '--fields' in sys.argv else 'bench,zero,rough').split(',')


def timeit(fn, n=30, warm=5):
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


for batch in batches:
    mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
    for fname in fields:
        f = trf if fname == 'bench' else (torch.zeros_like(trf) if fname == 'zero' else synth.cfg2_batch(batch, 160, 32, device=dev, rough=True)[2])
        row = {'batch': batch, 'field': fname}
        d = {}
        for name, tune in (('wc', WC), ('reg', NO_WC)):
            for store in (False, True):
                key = name + ('_store' if store else '')
                row['ms_' + key] = round(timeit(lambda: ne.fused.warp_dice(mov, f, fix, return_warped=store, _tune=tune)), 4)
            d[name] = ne.fused.warp_dice(mov, f, fix, _tune=tune)
        row['max_abs_dice_diff'] = float((d['wc'] - d['reg']).abs().max())
        # the stand-alone warp: the z-run register kernel (variant 3, the default until round 4), the wave-cache kernel (variant 10), the library default
        from neurite_amd import deferred
        keep, deferred.enabled = deferred.enabled, False
        for name, variant, tune in (('interpn_zrun', 3, 20 | (1 << 16)), ('interpn_wc', 10, 0), ('interpn_default', 0, 0)):
            st = ne.layers.SpatialTransformer()
            st._variant, st._tune = variant, tune
            row['ms_' + name] = round(timeit(lambda: st([mov, f])), 4)
        deferred.enabled = keep
        nvox = batch * 160 ** 3
        row['frac_wc'] = round(nvox * 268 / row['ms_wc'] / 1e9 / 8.0, 4)
        row['frac_reg'] = round(nvox * 268 / row['ms_reg'] / 1e9 / 8.0, 4)
        print(json.dumps(row), flush=True)
