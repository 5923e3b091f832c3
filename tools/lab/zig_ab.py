#!/usr/bin/env python3
"""A/B of the zig-zag march (NRT_WC_ZIG=1, fused_wc.h ZIG) against the shipped 4 x 8 columns: run once per setting (the switch is read once per
process), compare the printed checksums / Dice values across runs.   NRT_WC_ZIG=0|1 python tools/lab/zig_ab.py [quick]"""
import json, os, sys, time, zlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurite_amd as ne
from neurite_amd import synth, _lib
dev = torch.device('cuda:0')
out = {'zig': os.environ.get('NRT_WC_ZIG', '0'), 'region': os.environ.get('NRT_WC_ZIG_REGION', '')}
st = ne.layers.SpatialTransformer(interp_method='linear')


def crc(t):
    return zlib.crc32(t.detach().cpu().numpy().tobytes())


def timeit(fn, n, ns):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    keepalive = [None] * ns
    def run(m):
        for k in range(m):
            with torch.cuda.stream(streams[k % ns]):
                keepalive[k % ns] = fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        run(2 * ns); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return round(best, 4)


# 1. odd shapes (edge masks on both sub-patches), absolute locations with fill, shift mode, linspace (Resize)
g = torch.Generator(device='cpu').manual_seed(3)
B, S, O = 7, (30, 90, 100), (33, 100, 90)
vol = torch.randn((B,) + S + (32,), generator=g).to(dev)
trf = (torch.randn((B,) + S + (3,), generator=g) * 2.5).to(dev)
fix = torch.rand((B,) + S + (32,), generator=g).to(dev)
ne.deferred.enabled = False
w = st([vol, trf])
out['odd_shift_warp_crc'] = crc(w)
out['odd_shift_dice'] = [round(float(v), 7) for v in ne.fused.warp_dice(vol, trf, fix).flatten()[:6]]
stf = ne.layers.SpatialTransformer(interp_method='linear', fill_value=0.25)
out['odd_fill_warp_crc'] = crc(stf([vol, trf * 4]))
vol2 = torch.randn((B,) + (20, 52, 44) + (32,), generator=g).to(dev)
out['resize_crc'] = crc(ne.layers.Resize(2.0)(vol2))
out['kernel_name_B4'] = _lib.lib().nrt_warp_dice_kernel_name(_lib.ints([160] * 3), _lib.ints([160] * 3), 32, 4, _lib.LOC_SHIFT, 0, 0, 0, 0).decode()
del vol, trf, fix, w, vol2
torch.cuda.empty_cache()
# 2. the bench tensors
for batch in ([4] if 'quick' in sys.argv else [4, 8, 32]):
    mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
    d = ne.fused.warp_dice(mov, trf, fix)
    w = st([mov, trf])
    r = {'dice_mean': float(d.double().mean()), 'dice_crc': crc(d), 'warp_crc': crc(w[:2])}
    del w
    V = batch * 160 ** 3 * 268 / 1e6 / 8000.0
    for ns in (1, 3):
        n = max(6, 48 // batch)
        f = timeit(lambda: ne.fused.warp_dice(mov, trf, fix), n, ns)
        s = timeit(lambda: st([mov, trf]), n, ns)
        r['fused_ms_%d' % ns], r['fused_frac_%d' % ns] = f, round(V / f, 4)
        r['warp_ms_%d' % ns], r['warp_frac_%d' % ns] = s, round(V / s, 4)
    out['B%d' % batch] = r
    del mov, fix, trf
    torch.cuda.empty_cache()
print(json.dumps(out))
