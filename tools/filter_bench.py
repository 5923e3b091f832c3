"""Separable passes (GaussianBlur) and min-max normalisation at 4 x 160^3 x 1: ms per pass and fraction of the HBM roof
(8 bytes per element and pass; min-max: 12 bytes per element)."""
import json, os, torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import neurite_amd as ne
dev = torch.device('cuda:0')

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

S, B = 160, 4
img = torch.randn(B, S, S, S, 1, device=dev)
nb = img.numel() * 8
for sigma in (1.0, 3.0):
    k = ne.utils.gaussian_kernel([sigma], separate=True)
    k = (k[0] if isinstance(k, list) else k).to(dev)
    for ax, name in ((0, 'x'), (1, 'y'), (2, 'z')):
        ms = timeit(lambda: ne.utils.separable_conv(img, k, axis=ax, batched=True))
        print(json.dumps({'op': 'pass %s sigma=%g (W=%d)' % (name, sigma, k.numel()), 'ms': round(ms, 4), 'frac': round(nb / ms / 1e6 / 8000, 3)}))
    blur = ne.layers.GaussianBlur(sigma=sigma)
    ms = timeit(lambda: blur(img))
    print(json.dumps({'op': 'GaussianBlur sigma=%g C=1 (3 passes)' % sigma, 'ms': round(ms, 4), 'frac': round(3 * nb / ms / 1e6 / 8000, 3)}))
ms = timeit(lambda: ne.utils.minmax_norm(img, axis=(1, 2, 3, 4)))
print(json.dumps({'op': 'minmax_norm per batch entry', 'ms': round(ms, 4), 'frac': round(img.numel() * 12 / ms / 1e6 / 8000, 3)}))
img4 = torch.randn(B, S, S, S, 4, device=dev)
blur = ne.layers.GaussianBlur(sigma=2.0)
ms = timeit(lambda: blur(img4), n=5)
print(json.dumps({'op': 'GaussianBlur sigma=2 C=4 (3 passes)', 'ms': round(ms, 4), 'frac': round(3 * img4.numel() * 8 / ms / 1e6 / 8000, 3)}))
