#!/usr/bin/env python3
"""Times macro variants of the fused warp + Dice kernel in ONE process (tools/fused_variants.py builds them as
tools/lab/libnrt_fused_<name>.so; each is a full copy of the product library with fused.o rebuilt under -DNRT_FUSED_<NAME>).

    FUSED_VARIANTS="EXP=0 EXP=6 ..." python tools/fused_variants.py --build        (here)
    FUSED_VARIANTS="EXP=0 EXP=6 ..." python tools/fused_probe.py [--batch 4] [--zero] (GPU box)

Prints one JSON line per variant: ms per launch on the bench batch (event timing, 30 launches), and with --zero the same on a zero
displacement field.  The EXP >= 6 variants compute wrong results on purpose (they drop or alias corner rows); EXP=0 is the product."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurite_amd import _lib, synth        # noqa: E402

LAB = os.path.join(ROOT, 'tools', 'lab')
VARIANTS = os.environ.get('FUSED_VARIANTS', 'EXP=0').split()
batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4
TUNE = int(os.environ.get('PROBE_TUNE', '0'))          # e.g. 1 << 29: the wave-cache kernel
size = int(sys.argv[sys.argv.index('--size') + 1]) if '--size' in sys.argv else 160
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(batch, size, 32, device=dev)
fields = {'bench': trf}
if '--zero' in sys.argv:
    fields['zero'] = torch.zeros_like(trf)
if '--rough' in sys.argv:
    fields['rough'] = synth.cfg2_batch(batch, size, 32, device=dev, rough=True)[2]
S = _lib.ints(list(mov.shape[1:-1]))
sums = torch.empty((batch, 3, 32), dtype=torch.float32, device=dev)
dice = torch.empty((batch, 32), dtype=torch.float32, device=dev)


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


for rep in range(2):
    for k in VARIANTS:
        path = os.path.join(LAB, 'libnrt_fused_%s.so' % k.replace('=', '').replace(',', '_'))
        h = C.CDLL(path)
        for name in ('nrt_warp_dice_workspace_bytes', 'nrt_warp_dice_soft_f32'):
            res, args = _lib._SIGNATURES[name]
            getattr(h, name).restype, getattr(h, name).argtypes = res, args
        nws = h.nrt_warp_dice_workspace_bytes(S, 32, batch, TUNE)
        ws = torch.empty(int(nws), dtype=torch.uint8, device=dev)
        row = {'variant': k, 'batch': batch, 'rep': rep}
        for fname, f in fields.items():
            def run():
                rc = h.nrt_warp_dice_soft_f32(_lib.ptr(mov), _lib.ptr(f), _lib.ptr(fix), None, S, S, 32, batch, f[0].numel(), 1, 0, 0.0, 0.0,
                                              _lib.ptr(sums), _lib.ptr(dice), None, TUNE, _lib.ptr(ws), nws, _lib.stream_ptr(dev))
                assert rc == 0, rc
            row['ms_' + fname] = round(timeit(run), 4)
        print(json.dumps(row), flush=True)
