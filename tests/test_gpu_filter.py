"""
GPU parity tests of the synthesis front-end pieces (separable_conv / GaussianBlur / minmax_norm / draw_perlin) against the
golden vectors produced by the reference's own source (tests/golden/filter_small.npz) and against the oracle.
Tolerance 1e-5 (the order of the tap accumulation in tf.nn.convolution is unspecified).
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import bits_equal, load_golden
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def test_gaussian_kernel_golden():
    g = load_golden('filter_small')
    for tag, kw in (('gk_iso3', dict(sigma=[1.5, 1.5, 1.5])), ('gk_aniso', dict(sigma=[0.7, 2.0])),
                    ('gk_win', dict(sigma=[1.0, 2.0], windowsize=[5, 4])), ('gk_xy', dict(sigma=[1.0, 2.0], indexing='xy')),
                    ('gk_tiny', dict(sigma=0))):
        np.testing.assert_allclose(N(ne.utils.gaussian_kernel(**kw)), g[tag + '__joint'], rtol=2e-6, atol=1e-9)
        ks = ne.utils.gaussian_kernel(separate=True, **kw)
        ks = ks if isinstance(ks, list) else [ks]
        for i, k in enumerate(ks):
            np.testing.assert_allclose(N(k), g['%s__sep%d' % (tag, i)], rtol=2e-6, atol=1e-9)
    with pytest.raises(ValueError, match='differ in length'):
        ne.utils.gaussian_kernel([1.0, 2.0], windowsize=[3])
    k = ne.utils.gaussian_kernel([2.0, 2.0], random=True, min_sigma=0.5, separate=True, seed=3)
    assert all(abs(float(x.sum()) - 1) < 1e-5 for x in k)


def test_separable_conv_blur_minmax_golden(dev):
    g = load_golden('filter_small')
    x3, xb, x2 = G(g['sc_x3'], dev), G(g['sc_xb'], dev), G(g['sc_x2'], dev)
    k3, k5, k4 = [g['sc_k3_%d' % i] for i in range(3)], g['sc_k5'], g['sc_k4']
    sc = ne.utils.separable_conv
    cases = {
        'sc_all_axes': sc(x3, k3),
        'sc_single_kernel': sc(x3, k5),
        'sc_axis1': sc(x3, torch.from_numpy(k5), axis=1),
        'sc_axes02_valid': sc(x3, [k5, k4], axis=[0, 2], padding='VALID'),
        'sc_even_same': sc(x3, k4),
        'sc_stride2': sc(x3, k5, strides=2),
        'sc_stride_list': sc(x3, [k5, k4], axis=[1, 2], strides=[2, 3]),
        'sc_dil2': sc(x3, k5, dilations=2),
        'sc_batched': sc(xb, k3, batched=True),
        'sc_2d': sc(x2, k5),
        'blur': ne.layers.GaussianBlur(sigma=[1.0, 0.0, 2.0])(xb),
        'blur_iso': ne.layers.GaussianBlur(sigma=1.3)(xb),
        'mm_all': ne.utils.minmax_norm(xb),
        'mm_per_batch': ne.utils.minmax_norm(xb, axis=(1, 2, 3, 4)),
        'mm_per_batch_feature': ne.utils.minmax_norm(xb, axis=(1, 2, 3)),
        'mm_const': ne.utils.minmax_norm(torch.full((3, 4, 2), 2.5, device=dev)),
    }
    for tag, got in cases.items():
        want = g[tag + '__out']
        assert tuple(got.shape) == want.shape, tag
        np.testing.assert_allclose(N(got), want, rtol=1e-5, atol=2e-6, err_msg=tag)
    # error behaviour of the reference
    with pytest.raises(AssertionError, match='non-spatial axis'):
        sc(x3, k5, axis=3)
    with pytest.raises(AssertionError, match='number of kernels'):
        sc(x3, [k5, k4])
    with pytest.raises(ValueError):
        ne.layers.GaussianBlur(sigma=[1, 2])(xb)
    with pytest.raises(ValueError):
        ne.layers.GaussianBlur(sigma=1, isotropic=True)
    assert ne.layers.GaussianBlur(sigma=0)(xb) is xb
    assert ne.layers.GaussianBlur(sigma=1.5).get_config()['sigma'] == 1.5
    with pytest.raises(NotImplementedError):
        ne.utils.minmax_norm(xb, axis=(1, 3))


def test_blur_larger_shapes_vs_oracle(dev):
    rng = np.random.default_rng(3)
    for shape, sigma in (((1, 40, 33, 47, 1), [2.0, 1.0, 3.0]), ((2, 24, 24, 24, 4), 1.5), ((3, 65, 50, 2), [0.8, 2.2])):
        x = rng.standard_normal(shape).astype(F)
        got = N(ne.layers.GaussianBlur(sigma=sigma)(G(x, dev)))
        np.testing.assert_allclose(got, npo.gaussian_blur(x, sigma), rtol=1e-5, atol=2e-6)
        mm = N(ne.utils.minmax_norm(G(x, dev), axis=tuple(range(1, x.ndim))))
        np.testing.assert_allclose(mm, npo.minmax_norm(x, axis=tuple(range(1, x.ndim))), rtol=1e-6, atol=1e-7)
        assert mm.min() == 0.0 and mm.max() == 1.0


def test_fast_separable_passes_equal_plain_kernels(dev, monkeypatch):
    """conv1d_axis_rows / conv1d_inner_lds (csrc/filter.hip: R outputs per lane, taps from the scalar cache, LDS-staged rows for the
    innermost axis) add the taps of an output in the order of the plain kernel: bit-identical results, borders included; and the
    vectorised min-max passes equal the oracle"""
    rng = np.random.default_rng(11)
    cases = [((2, 40, 36, 64, 1), [3.0, 1.0, 2.0]), ((1, 21, 40, 136, 1), [1.0, 2.5, 3.0]), ((3, 9, 70, 33, 1), 1.2),
             ((2, 24, 20, 16, 8), 1.5), ((1, 130, 260, 1), [2.0, 4.0]), ((4, 8, 8, 300, 1), [0.0, 0.0, 6.0])]
    for shape, sigma in cases:
        x = G(rng.standard_normal(shape).astype(F), dev)
        monkeypatch.delenv('NRT_CONV1D_GENERIC', raising=False)
        fast = N(ne.layers.GaussianBlur(sigma=sigma)(x))
        monkeypatch.setenv('NRT_CONV1D_GENERIC', '1')
        plain = N(ne.layers.GaussianBlur(sigma=sigma)(x))
        monkeypatch.delenv('NRT_CONV1D_GENERIC', raising=False)
        assert bits_equal(fast, plain), (shape, sigma)
        np.testing.assert_allclose(fast, npo.gaussian_blur(N(x), sigma), rtol=1e-5, atol=2e-6)
    # even widths, VALID padding, widths above the run length, non-finite taps (skipped at the borders by both forms)
    x = G(rng.standard_normal((2, 33, 48, 72, 1)).astype(F), dev)
    ks = [rng.standard_normal(w).astype(F) for w in (4, 11, 30)]
    ks[1][3] = np.inf
    for padding in ('SAME', 'VALID'):
        monkeypatch.delenv('NRT_CONV1D_GENERIC', raising=False)
        fast = N(ne.utils.separable_conv(x, ks, batched=True, padding=padding))
        monkeypatch.setenv('NRT_CONV1D_GENERIC', '1')
        plain = N(ne.utils.separable_conv(x, ks, batched=True, padding=padding))
        monkeypatch.delenv('NRT_CONV1D_GENERIC', raising=False)
        assert bits_equal(fast, plain), padding
    for shape in ((3, 50, 40, 36, 1), (2, 1001), (5, 7, 9, 11, 1)):
        x = rng.standard_normal(shape).astype(F)
        ax = tuple(range(1, x.ndim))
        np.testing.assert_allclose(N(ne.utils.minmax_norm(G(x, dev), axis=ax)), npo.minmax_norm(x, axis=ax), rtol=1e-6, atol=1e-7)


def test_draw_perlin(dev):
    out = ne.augment.draw_perlin((32, 32, 32, 2), scales=(1, 2, 4, 8), max_std=1.0, seed=5)
    assert out.shape == (32, 32, 32, 2) and out.is_cuda and bool(torch.isfinite(out).all())
    again = ne.augment.draw_perlin((32, 32, 32, 2), scales=(1, 2, 4, 8), max_std=1.0, seed=5)
    assert torch.equal(out, again)                                        # reproducible for a seed
    other = ne.augment.draw_perlin((32, 32, 32, 2), scales=(1, 2, 4, 8), max_std=1.0, seed=6)
    assert not torch.equal(out, other)
    # coarse levels are smooth: neighbouring voxels of the scale-8 level alone differ little compared with its range
    coarse = ne.augment.draw_perlin((32, 32, 32, 1), scales=8, min_std=1.0, max_std=1.0, seed=1)
    d = (coarse[1:] - coarse[:-1]).abs().mean()
    assert float(d) < 0.5 * float(coarse.std())
    odd = ne.augment.draw_perlin((10, 13, 9, 1), scales=(3, 5), seed=2)   # sample shapes ceil(n / scale), zoom = n / sample
    assert odd.shape == (10, 13, 9, 1)
