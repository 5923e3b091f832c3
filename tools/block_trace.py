#!/usr/bin/env python3
"""Per-block timeline of one launch of the fused register kernel (lab build with -DNRT_FUSED_TRACE: FUSED_VARIANTS="TRACE=1"
python tools/fused_variants.py --build).   python tools/block_trace.py [--batch 4]     (GPU box)
Prints, per XCD: blocks, first start, last end, mean / max block duration; the launch's makespan against sum(durations) / slots."""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurite_amd import _lib, synth
batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4
dev = torch.device('cuda:0')
h = C.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libnrt_fused_TRACE1.so'))
for name in ('nrt_warp_dice_workspace_bytes', 'nrt_warp_dice_soft_f32'):
    res, args = _lib._SIGNATURES[name]
    getattr(h, name).restype, getattr(h, name).argtypes = res, args
h.nrt_debug_set_trace.argtypes = [C.c_void_p]
mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
S = _lib.ints([160] * 3)
sums = torch.empty((batch, 3, 32), dtype=torch.float32, device=dev)
dice = torch.empty((batch, 32), dtype=torch.float32, device=dev)
for tune, label in ((1 << 30, 'mixed (auto)'), ((3 | (2 << 4) | (3 << 8) | (1 << 14) | (3 << 24) | (2 << 27) | (1 << 16)) | (1 << 30), 'whole columns (nseg 1)'),
                    ((3 | (2 << 4) | (3 << 8) | (1 << 14) | (3 << 24) | (2 << 27) | (2 << 16)) | (1 << 30), 'halves (nseg 2)')):
    nws = h.nrt_warp_dice_workspace_bytes(S, 32, batch, tune)
    ws = torch.empty(int(nws), dtype=torch.uint8, device=dev)
    buf = torch.zeros((40000, 4), dtype=torch.int64, device=dev)
    assert h.nrt_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0

    def run():
        rc = h.nrt_warp_dice_soft_f32(_lib.ptr(mov), _lib.ptr(trf), _lib.ptr(fix), None, S, S, 32, batch, trf[0].numel(), 1, 0, 0.0, 0.0,
                                      _lib.ptr(sums), _lib.ptr(dice), None, tune, _lib.ptr(ws), nws, _lib.stream_ptr(dev))
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf.zero_()
    run()
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    start, end, xcc = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2] & 15          # microseconds
    dur = end - start
    xlen = t[:, 3] & 0xffffffff
    out = {'schedule': label, 'batch': batch, 'blocks': int(len(t)), 'makespan_us': round(float(end.max()), 1),
           'sum_dur_over_512_slots_us': round(float(dur.sum() / 512), 1), 'whole_block_us_mean': round(float(dur[xlen == 160].mean()), 1) if (xlen == 160).any() else None,
           'whole_block_us_p5_p95': [round(float(np.percentile(dur[xlen == 160], q)), 1) for q in (5, 95)] if (xlen == 160).any() else None,
           'per_xcc': []}
    for k in sorted(set(xcc.tolist())):
        m = xcc == k
        out['per_xcc'].append({'xcc': int(k), 'blocks': int(m.sum()), 'last_end_us': round(float(end[m].max()), 1), 'busy_us_per_slot': round(float(dur[m].sum() / 64), 1),
                               'mean_dur_us': round(float(dur[m].mean()), 1)})
    # how many blocks run at once over time (every 20 us)
    grid = np.arange(0, end.max(), 20.0)
    out['resident_blocks_every_20us'] = [int(((start <= g) & (end > g)).sum()) for g in grid]
    print(json.dumps(out), flush=True)
