"""
GPU parity tests of the VoxelMorph companion layers (VecInt, RescaleTransform, ComposeTransform, AffineToDenseShift)
against the NumPy oracle.  The warp+add kernel follows the oracle op-for-op, so the integration and the dense
composition are BIT-EXACT; paths that go through the affine glue (a float32 matmul) are held to 1e-5.
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import bits_equal
from oracle import grad_oracle as go
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def G(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_() if grad else t


def N(t):
    return t.detach().cpu().numpy()


def smooth_field(rng, shape, D, amp):
    coarse = rng.standard_normal(tuple(max(2, s // 4) for s in shape) + (D,)).astype(F)
    zoom = [shape[d] / coarse.shape[d] for d in range(len(shape))]
    f = npo.resize(coarse, zoom) if all(int(coarse.shape[d] * zoom[d]) == shape[d] for d in range(len(shape))) else None
    if f is None or f.shape[:-1] != tuple(shape):
        f = rng.standard_normal(tuple(shape) + (D,)).astype(F)
    return (f * F(amp)).astype(F)


@pytest.mark.parametrize('shape', [(12, 10, 8), (16, 12), (20,)])
@pytest.mark.parametrize('method,steps', [('ss', 5), ('ss', 0), ('quadrature', 4)])
def test_vecint(dev, shape, method, steps):
    rng = np.random.default_rng(len(shape) * 10 + steps)
    D, B = len(shape), 2
    vel = np.stack([smooth_field(rng, shape, D, 3.0) for _ in range(B)], 0)
    if method == 'quadrature' and steps == 0:
        pytest.skip('nb_steps >= 1')
    steps = max(steps, 1) if method == 'quadrature' else steps
    out = N(ne.layers.VecInt(method=method, int_steps=steps)(G(vel, dev)))
    for b in range(B):
        ref = npo.integrate_vec(vel[b], method, steps)
        assert bits_equal(out[b], ref), 'VecInt %s b%d: max diff %g' % (method, b, np.abs(out[b] - ref).max())
    # un-batched functional form
    out1 = N(ne.utils.integrate_vec(G(vel[0], dev), method=method, nb_steps=steps))
    assert bits_equal(out1, npo.integrate_vec(vel[0], method, steps))


def test_vecint_xy_and_errors(dev):
    rng = np.random.default_rng(5)
    vel = smooth_field(rng, (10, 9, 8), 3, 2.0)[None]
    out = N(ne.layers.VecInt(indexing='xy', int_steps=3)(G(vel, dev)))
    ref = npo.integrate_vec(vel[0][..., [1, 0, 2]], 'ss', 3)
    assert bits_equal(out[0], ref)
    with pytest.raises(Exception, match='transform ndims'):
        ne.layers.VecInt()(G(vel[..., :2], dev))
    with pytest.raises(ValueError):
        ne.utils.integrate_vec(G(vel[0], dev), method='euler', nb_steps=2)
    with pytest.raises(NotImplementedError):
        ne.utils.integrate_vec(G(vel[0], dev), method='ode', nb_steps=2)
    assert ne.layers.VecInt(int_steps=3).get_config()['int_steps'] == 3


def test_vecint_backward(dev):
    """gradient of the integrated field wrt the velocity field through the warp backward kernels"""
    rng = np.random.default_rng(7)
    shape, steps = (8, 7, 6), 3
    vel = smooth_field(rng, shape, 3, 2.0) + F(0.013)
    w = rng.standard_normal(shape + (3,)).astype(F)
    v = G(vel[None], dev, True)
    out = ne.layers.VecInt(int_steps=steps)(v)
    (out[0] * G(w, dev)).sum().backward()
    vo = torch.from_numpy(vel).double().requires_grad_()
    x = vo / (2 ** steps)
    for _ in range(steps):
        x = x + go.transform(x, x)
    (x * torch.from_numpy(w).double()).sum().backward()
    np.testing.assert_allclose(N(out[0]), x.detach().numpy(), rtol=1e-4, atol=1e-5)
    want = vo.grad.numpy()
    assert np.abs(N(v.grad[0]) - want).max() / np.abs(want).max() < 1e-3


@pytest.mark.parametrize('factor', [2, 0.5])
def test_rescale_transform(dev, factor):
    rng = np.random.default_rng(11)
    B, shape = 2, (8, 6, 10)
    trf = rng.standard_normal((B,) + shape + (3,)).astype(F)
    layer = ne.layers.RescaleTransform(factor)
    out = N(layer(G(trf, dev)))
    for b in range(B):
        assert bits_equal(out[b], npo.rescale_dense_transform(trf[b], factor))
    assert tuple(layer.compute_output_shape(trf.shape)) == out.shape
    aff = rng.standard_normal((B, 3, 4)).astype(F)
    oa = N(ne.layers.RescaleTransform(factor)(G(aff, dev)))
    want = aff.copy()
    want[..., -1] *= F(factor)
    assert bits_equal(oa, want)
    assert layer.get_config()['zoom_factor'] == factor


def test_compose_transform(dev):
    rng = np.random.default_rng(13)
    B, shape = 2, (9, 8, 7)
    d1 = np.stack([smooth_field(rng, shape, 3, 2.0) for _ in range(B)], 0)
    d2 = np.stack([smooth_field(rng, shape, 3, 2.0) for _ in range(B)], 0)
    d3 = np.stack([smooth_field(rng, shape, 3, 1.0) for _ in range(B)], 0)
    a1 = (np.eye(3, 4, dtype=F)[None] + rng.standard_normal((B, 3, 4)).astype(F) * F(0.05))
    a2 = (np.eye(3, 4, dtype=F)[None] + rng.standard_normal((B, 3, 4)).astype(F) * F(0.05))
    layer = ne.layers.ComposeTransform()
    out = N(layer([G(d1, dev), G(d2, dev), G(d3, dev)]))                   # dense o dense o dense: bit-exact
    for b in range(B):
        assert bits_equal(out[b], npo.compose([d1[b], d2[b], d3[b]]))
    out = N(layer([G(a1, dev), G(a2, dev)]))                               # affine o affine stays affine
    assert out.shape == (B, 3, 4)
    for b in range(B):
        np.testing.assert_allclose(out[b], npo.compose([a1[b], a2[b]]), rtol=1e-5, atol=1e-6)
    out = N(layer([G(a1, dev), G(d2, dev)]))                               # affine o dense
    for b in range(B):
        np.testing.assert_allclose(out[b], npo.compose([a1[b], d2[b]]), rtol=1e-4, atol=2e-4)
    out = N(layer([G(d1, dev), G(a2, dev)]))                               # dense o affine
    for b in range(B):
        np.testing.assert_allclose(out[b], npo.compose([d1[b], a2[b]]), rtol=1e-4, atol=2e-4)
    with pytest.raises(ValueError):
        layer([G(d1, dev)])
    with pytest.raises(ValueError):
        ne.utils.compose([G(d1[0], dev), G(d2[0], dev)], indexing='xy')
    # composition is what warping twice does: warp(warp(v, d1), d2) == warp(v, compose([d1, d2])) up to interpolation
    # of the field itself -- exact for integer shifts
    s1 = np.zeros(shape + (3,), F); s1[..., 0] = 1
    s2 = np.zeros(shape + (3,), F); s2[..., 1] = -2
    c = N(ne.utils.compose([G(s1, dev), G(s2, dev)]))
    assert np.array_equal(c, s1 + s2)


def test_affine_to_dense_shift_layer(dev):
    rng = np.random.default_rng(17)
    B, shape = 2, (6, 7, 5)
    aff = (np.eye(3, 4, dtype=F)[None] + rng.standard_normal((B, 3, 4)).astype(F) * F(0.1))
    layer = ne.layers.AffineToDenseShift(shape)
    out = N(layer(G(aff, dev)))
    assert out.shape == (B,) + shape + (3,) == tuple(layer.compute_output_shape(aff.shape))
    for b in range(B):
        np.testing.assert_allclose(out[b], npo.affine_to_dense_shift(aff[b], shape), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ne.layers.AffineToDenseShift(shape)(G(np.zeros((B, 3, 3), F), dev))


def test_affine_to_dense_shift_kernel(dev):
    """csrc/vxm.hip (one kernel, batched) == the torch form (used under autograd and on the host) == the oracle: 2-D and 3-D,
    with and without the centre shift, square [D + 1, D + 1] matrices, the affine input of SpatialTransformer"""
    rng = np.random.default_rng(23)
    for shape in ((33, 20, 17), (40, 27), (160, 160, 160)):
        D = len(shape)
        aff = (np.eye(D, D + 1, dtype=F)[None] + rng.standard_normal((3, D, D + 1)).astype(F) * F(0.05))
        aff[..., -1] += rng.standard_normal((3, D)).astype(F) * 4
        for center in (True, False):
            got = N(ne.utils.affine_to_dense_shift(G(aff, dev), shape, shift_center=center))
            assert got.shape == (3,) + shape + (D,)
            g = G(aff, dev).requires_grad_()
            via_torch = ne.utils.affine_to_dense_shift(g, shape, shift_center=center)
            assert via_torch.requires_grad                       # the differentiable form
            scale = max(shape)
            np.testing.assert_allclose(got, N(via_torch), rtol=1e-5, atol=2e-6 * scale)
            if np.prod(shape) < 1e5:
                for b in range(3):
                    np.testing.assert_allclose(got[b], npo.affine_to_dense_shift(aff[b], shape, shift_center=center), rtol=1e-5, atol=1e-5)
                    one = N(ne.utils.affine_to_dense_shift(G(aff[b], dev), shape, shift_center=center))
                    assert np.array_equal(one, got[b])
        sq = np.concatenate([aff, np.tile(np.eye(D + 1, dtype=F)[-1:], (3, 1, 1))], 1)
        assert np.array_equal(N(ne.utils.affine_to_dense_shift(G(sq, dev), shape)), N(ne.utils.affine_to_dense_shift(G(aff, dev), shape)))
    # SpatialTransformer on affines: the dense field is formed on the device, the warp equals the oracle's
    vol = rng.standard_normal((2, 12, 10, 9, 3)).astype(F)
    aff = (np.eye(3, 4, dtype=F)[None] + rng.standard_normal((2, 3, 4)).astype(F) * F(0.05))
    out = N(ne.layers.SpatialTransformer()([G(vol, dev), G(aff, dev)]))
    np.testing.assert_allclose(out, npo.spatial_transformer(vol, aff), rtol=1e-4, atol=1e-4)
