"""warp of 8- / 16- / 64-channel volumes (feature maps) at 4 x 160^3: kernel variants 2 (rows) and 5 (tiles, several shapes)"""
import json, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
S, B = 160, 4
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def T(lx, ly, lz, zo=0): return lx | (ly << 4) | (lz << 8) | (zo << 12)
for C in (8, 16, 64):
    vol = torch.randn(B, S, S, S, C, device=dev)
    ref = None
    for variant, tune in ((2, 0), (2, 1), (2, 2), (2, 8), (5, 0), (5, T(2, 2, 4)), (5, T(3, 3, 3)), (5, T(2, 3, 4, 1)), (5, T(1, 2, 5)), (5, T(0, 2, 5))):
        st = ne.layers.SpatialTransformer(); st._variant, st._tune = variant, tune
        try:
            out = st([vol, flow])
        except Exception as e:
            print(json.dumps({'C': C, 'variant': variant, 'tune': tune, 'error': str(e)[:60]})); continue
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out if ref is None else ref
        ms = timeit(lambda: st([vol, flow]))
        nbytes = B * S ** 3 * (8 * C + 12)
        print(json.dumps({'C': C, 'variant': variant, 'tune': tune, 'ms': round(ms, 4), 'frac': round(nbytes / ms / 1e6 / 8000, 3), 'same': same}))
    del vol, ref, out
