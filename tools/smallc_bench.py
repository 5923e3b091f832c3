"""Warp / resize of few-channel volumes (images, flow fields): ms and fraction of the HBM roof (algorithmic bytes)."""
import json, torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
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
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
for C in (1, 2, 3, 4, 8, 16):
    vol = torch.randn(B, S, S, S, C, device=dev)
    for method in ('linear', 'nearest'):
        st = ne.layers.SpatialTransformer(interp_method=method)
        import os
        st._variant = int(os.environ.get('NRT_SMALLC_VARIANT', '0')) if (method == 'linear' and C <= 4) else 0
        ms = timeit(lambda: ne.deferred.materialize(st([vol, flow])))      # a deferred warp (C = 4 * 2^k) is evaluated here
        nbytes = B * S ** 3 * (8 * C + 12)
        print(json.dumps({'op': 'warp', 'C': C, 'method': method, 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1),
                          'frac': round(nbytes / ms / 1e6 / 8000, 3)}))
# Resize x2 of a half-resolution flow field (RescaleTransform / labels_to_image, models.py:802-804)
for C in (3, 1, 2, 4):
    rs = ne.layers.Resize(2)
    half = torch.randn(B, 80, 80, 80, C, device=dev)
    ms = timeit(lambda: rs(half))
    nbytes = B * (80 ** 3 + 160 ** 3) * 4 * C
    print(json.dumps({'op': 'resize x2 C=%d' % C, 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1), 'frac': round(nbytes / ms / 1e6 / 8000, 3)}))
vi = ne.layers.VecInt(int_steps=7)
ms = timeit(lambda: vi(flow), n=5)
nbytes = 7 * B * S ** 3 * 36
print(json.dumps({'op': 'VecInt 7 steps 160^3', 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1), 'frac': round(nbytes / ms / 1e6 / 8000, 3)}))
# synthesis front-end: Gaussian blur (3 separable passes), min-max normalisation, Perlin-like noise
img = torch.randn(B, S, S, S, 1, device=dev)
for sigma in (1.0, 3.0):
    blur = ne.layers.GaussianBlur(sigma=sigma)
    ms = timeit(lambda: blur(img))
    nbytes = 3 * 2 * img.numel() * 4
    print(json.dumps({'op': 'GaussianBlur sigma=%g C=1 (3 passes)' % sigma, 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1),
                      'frac': round(nbytes / ms / 1e6 / 8000, 3)}))
ms = timeit(lambda: ne.utils.minmax_norm(img, axis=(1, 2, 3, 4)))
nbytes = 3 * img.numel() * 4
print(json.dumps({'op': 'minmax_norm per batch entry', 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1), 'frac': round(nbytes / ms / 1e6 / 8000, 3)}))
ms = timeit(lambda: ne.augment.draw_perlin((S, S, S, 1), scales=(8, 16, 32), max_std=1.0, seed=1), n=5)
print(json.dumps({'op': 'draw_perlin 160^3 scales (8,16,32)', 'ms': round(ms, 4)}))
# mutual information of two images (registration loss), forward and forward + backward
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    mi = ne.metrics.MutualInformation(nb_bins=16)
ya = img * 0.7 + 0.3 * torch.randn_like(img)
ms = timeit(lambda: mi.volumes(img, ya))
print(json.dumps({'op': 'MutualInformation.volumes nb_bins=16', 'ms': round(ms, 4), 'GBs': round(2 * img.numel() * 4 / ms / 1e6, 1)}))
xg = img.clone().requires_grad_()
def mi_step():
    xg.grad = None
    (-mi.volumes(xg, ya).sum()).backward()
ms = timeit(mi_step)
print(json.dumps({'op': 'MutualInformation.volumes forward + backward', 'ms': round(ms, 4)}))
# labels_to_image (neurite/tf/models.py:649-918): 160^3 label map with 32 labels -> warped labels (one-hot) + synthetic image
import warnings
lab = synth.one_hot_volume(1, S, 32, dev).argmax(-1)[None, ..., None].to(torch.int32).repeat(B, 1, 1, 1, 1)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    gen = ne.models.labels_to_image((S, S, S), list(range(32)))
gen = gen.to(dev) if hasattr(gen, 'to') else gen
ms = timeit(lambda: gen(lab), n=5)
print(json.dumps({'op': 'labels_to_image 160^3, 32 labels, batch %d (image + one-hot labels)' % B, 'ms': round(ms, 3),
                  'ms_per_volume': round(ms / B, 3)}))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    gen2 = ne.models.labels_to_image_new(list(range(32)), in_shape=(S, S, S), aff_shift=10, aff_rotate=10, aff_scale=0.1, aff_shear=0.05)
labf = lab.to(torch.float32)
ms = timeit(lambda: gen2(labf), n=5)
print(json.dumps({'op': 'labels_to_image_new 160^3, 32 labels, batch %d, affine + warp + bias + noise + blur (defaults)' % B,
                  'ms': round(ms, 3), 'ms_per_volume': round(ms / B, 3)}))
