"""
GPU parity tests of soft_quantize / MutualInformation (csrc/mi.hip) against the golden vectors produced by the reference's
own source (tests/golden/mi_small.npz), the oracle at larger sizes, and the float64 autograd oracle for the gradients.
"""

import contextlib
import io

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import load_golden
from neurite_amd.errors import InvalidArgumentError
from oracle import grad_oracle as go
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32
R = dict(rtol=5e-5, atol=5e-6)


def G(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_() if grad else t


def N(t):
    return t.detach().cpu().numpy()


def MI(**kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return ne.metrics.MutualInformation(**kw)


def test_mi_golden(dev):
    g = load_golden('mi_small')
    x, y, p3, q3, p16, q16 = [G(g[k], dev) for k in ('x', 'y', 'p3', 'q3', 'p16', 'q16')]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        mi16 = ne.metrics.MutualInformation()
    assert np.isclose(mi16.soft_bin_alpha, g['alpha16'], rtol=1e-6) and buf.getvalue().strip() != ''     # the reference prints alpha
    np.testing.assert_allclose(N(mi16.volumes(x, y)), g['volumes16__out'], **R)
    np.testing.assert_allclose(N(MI(nb_bins=8, min_clip=0.1, max_clip=0.9).volumes(x, y)), g['volumes8clip__out'], **R)
    np.testing.assert_allclose(N(MI(nb_bins=16, soft_bin_alpha=50.0).volumes(x, y)), g['volumes_alpha50__out'], **R)
    np.testing.assert_allclose(N(mi16.channelwise(p3, q3)), g['channelwise__out'], **R)
    np.testing.assert_allclose(N(mi16.segs(p16, q16)), g['segs__out'], **R)
    np.testing.assert_allclose(N(mi16.maps(p16, p16)), g['maps_self__out'], **R)
    np.testing.assert_allclose(N(mi16.volume_seg(x, p16)), g['volume_seg__out'], **R)
    np.testing.assert_allclose(N(mi16.volume_seg(p16, y)), g['seg_volume__out'], **R)
    s = G(g['sq_in'], dev)
    np.testing.assert_allclose(N(ne.utils.soft_quantize(s, nb_bins=5, alpha=3.0)), g['sq_nb5__out'], rtol=2e-6)
    np.testing.assert_allclose(N(ne.utils.soft_quantize(s, bin_centers=np.array([0.1, 0.5, 0.9], F), nb_bins=None, alpha=2.0)),
                               g['sq_centers__out'], rtol=2e-6)
    np.testing.assert_allclose(N(ne.utils.soft_quantize(s, nb_bins=4, alpha=1.5, min_clip=0.2, max_clip=0.8, return_log=True)),
                               g['sq_log_clip__out'], rtol=2e-6, atol=1e-7)
    # error behaviour
    with pytest.raises(InvalidArgumentError):
        mi16.volumes(p3, q3)
    with pytest.raises(InvalidArgumentError):
        mi16.maps(p16, q16[..., :8])
    with pytest.raises(InvalidArgumentError):
        mi16.maps(p16 - 1.0, q16)
    with pytest.raises(InvalidArgumentError):
        mi16.volume_seg(x, y)
    with pytest.raises(AssertionError):
        ne.utils.soft_quantize(s, bin_centers=[0.1, 0.5], nb_bins=3)
    with pytest.raises(AssertionError):
        ne.metrics.MutualInformation(bin_centers=np.array([0., 1.], F), nb_bins=2)


def test_mi_larger_vs_oracle(dev):
    rng = np.random.default_rng(17)
    x = rng.random((2, 33, 30, 35, 1)).astype(F)
    y = (np.sin(3 * x) + 0.2 * rng.standard_normal(x.shape)).astype(F)
    for nb in (16, 24, 32, 5):
        got = N(MI(nb_bins=nb).volumes(G(x, dev), G(y, dev)))
        np.testing.assert_allclose(got, npo.mi_volumes(x, y, nb_bins=nb), rtol=2e-4, atol=2e-5)
    p = rng.random((2, 20, 16, 24, 3)).astype(F)
    q = (0.6 * p + 0.4 * rng.random(p.shape)).astype(F)
    np.testing.assert_allclose(N(MI(nb_bins=16).channelwise(G(p, dev), G(q, dev))), npo.mi_channelwise(p, q), rtol=2e-4, atol=2e-5)
    # given bin centres (the reference's own constructor path asserts in soft_quantize; supported here)
    cen = np.linspace(0, 1, 12).astype(F)
    mi = MI(bin_centers=cen)
    xq = npo.soft_quantize(x[..., 0], cen, None, mi.soft_bin_alpha)
    yq = npo.soft_quantize(np.clip(y, 0, 1)[..., 0], cen, None, mi.soft_bin_alpha)
    np.testing.assert_allclose(N(mi.volumes(G(x, dev), G(np.clip(y, 0, 1), dev))), npo.mi_maps(xq, yq), rtol=2e-4, atol=2e-5)
    # probability maps with a voxel count that is not a multiple of 32 and an odd label count
    pm = rng.random((2, 7, 9, 5, 11)).astype(F)
    qm = rng.random((2, 7, 9, 5, 11)).astype(F)
    np.testing.assert_allclose(N(MI().maps(G(pm, dev), G(qm, dev))), npo.mi_maps(pm, qm), rtol=2e-4, atol=2e-6)


def test_mi_backward(dev):
    """-MI as a registration loss: gradient wrt both images (bin centres constant) vs float64 autograd"""
    rng = np.random.default_rng(23)
    x = rng.random((2, 9, 8, 10, 2)).astype(F)
    y = (0.5 * x + 0.5 * rng.random(x.shape)).astype(F)
    for kw in (dict(nb_bins=16), dict(nb_bins=20, min_clip=0.15, max_clip=0.85)):
        mi = MI(**kw)
        xg, yg = G(x, dev, True), G(y, dev, True)
        val = mi.channelwise(xg, yg)
        w = rng.standard_normal(tuple(val.shape)).astype(F)
        (-(val * G(w, dev)).sum()).backward()
        nb = kw['nb_bins']
        cx = torch.from_numpy(npo.tf_linspace(x.min(), x.max(), nb)).double()
        cy = torch.from_numpy(npo.tf_linspace(y.min(), y.max(), nb)).double()
        xo, yo = torch.from_numpy(x).double().requires_grad_(), torch.from_numpy(y).double().requires_grad_()
        ref = go.mi_channelwise(xo, yo, cx, cy, float(mi.soft_bin_alpha), kw.get('min_clip', float('-inf')),
                                kw.get('max_clip', float('inf')))
        (-(ref * torch.from_numpy(w).double()).sum()).backward()
        np.testing.assert_allclose(N(val), ref.detach().numpy(), rtol=2e-4, atol=2e-5)
        for got, want, nm in ((xg.grad, xo.grad, 'x'), (yg.grad, yo.grad, 'y')):
            want = want.numpy()
            err = np.abs(N(got) - want).max() / np.abs(want).max()
            assert err < 2e-3, (nm, err)
    # volume_seg: the image side differentiates through soft_quantize (csrc/mi.hip: soft_quantize_bwd), the maps through the
    # 1x1 contractions; oracle = float64 autograd of exp(-alpha (x - c)^2) -> MutualInformation.maps, centres held constant
    mi = MI()
    xv = G(x[..., :1], dev, True)
    seg = rng.dirichlet(np.ones(mi.nb_bins), x.shape[:-1])           # maps() wants as many labels as bins (metrics.py:249)
    val = mi.volume_seg(xv, G(seg.astype(F), dev))
    (-val.sum()).backward()
    cen = ne.utils._bin_centers(xv.detach()[..., 0].contiguous(), None, mi.nb_bins).cpu().double()
    xo = torch.from_numpy(x[..., 0]).double().requires_grad_()
    xq = torch.exp(-float(mi.soft_bin_alpha) * (xo[..., None] - cen) ** 2)
    ref = go.mi_maps(xq, torch.from_numpy(seg).double())
    (-ref.sum()).backward()
    np.testing.assert_allclose(N(val), ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    want = xo.grad.numpy()[..., None]
    err = np.abs(N(xv.grad) - want).max() / np.abs(want).max()
    assert err < 2e-3, err


def test_mi_backward_mfma_equals_scalar_kernel(dev, monkeypatch):
    """the matrix-core backward (mi_joint_bwd_mfma: G wy and G^T wx as 16x16x4 MFMAs, 32 exponentials per voxel) against the
    one-thread-per-voxel kernel (NRT_MI_BWD_SCALAR=1): ragged voxel counts, 5 / 16 / 32 bins, clipping, one or both gradients"""
    rng = np.random.default_rng(29)
    for shape, kw, both in (((2, 13, 11, 7, 1), dict(nb_bins=16), True), ((1, 37, 5, 3, 3), dict(nb_bins=5), True),
                            ((3, 20, 20, 20, 1), dict(nb_bins=32, min_clip=0.1, max_clip=0.9), False),
                            ((1, 1, 1, 1, 1), dict(nb_bins=16), True)):
        x = rng.random(shape).astype(F)
        y = (0.6 * x + 0.4 * rng.random(shape)).astype(F)
        grads = {}
        for mode in ('0', '1'):
            monkeypatch.setenv('NRT_MI_BWD_SCALAR', mode)
            with contextlib.redirect_stdout(io.StringIO()):
                mi = MI(**kw)
            xg, yg = G(x, dev, True), G(y, dev, both)
            (-mi.channelwise(xg, yg).sum()).backward()
            grads[mode] = (N(xg.grad), N(yg.grad) if both else None)
        monkeypatch.delenv('NRT_MI_BWD_SCALAR')
        for a, b in zip(grads['0'], grads['1']):
            if a is not None:
                assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30), (shape, kw)


def test_soft_quantize_backward(dev):
    """d soft_quantize / d x (bin centres constant; clip passes the gradient on the closed range) vs float64 autograd, plain and
    return_log forms"""
    import torch
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.2, 1.2, (2, 9, 7, 5)).astype(F)
    cen = np.array([0.0, 0.3, 0.55, 1.0], F)
    for ret_log in (False, True):
        for lo, hi in ((-np.inf, np.inf), (0.1, 0.9)):
            xg = G(x, dev, True)
            out = ne.utils.soft_quantize(xg, bin_centers=cen, nb_bins=None, alpha=2.5, min_clip=lo, max_clip=hi, return_log=ret_log)
            w = rng.standard_normal(out.shape)
            (out.double() * G(w, dev)).sum().backward()
            xo = torch.from_numpy(x).double().requires_grad_()
            xc = torch.clamp(xo, lo, hi)
            lg = -2.5 * (xc[..., None] - torch.from_numpy(cen).double()) ** 2
            ref = lg if ret_log else torch.exp(lg)
            (ref * torch.from_numpy(w)).sum().backward()
            np.testing.assert_allclose(N(out), ref.detach().numpy(), rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(N(xg.grad), xo.grad.numpy(), rtol=2e-4, atol=1e-5 * np.abs(xo.grad.numpy()).max())


def test_mi_maps_backward(dev):
    """-MI between probability maps (segs / maps) as a loss: gradient wrt both maps vs float64 autograd; voxel counts that
    are / are not multiples of 32, odd label counts"""
    rng = np.random.default_rng(29)
    for shape in ((2, 8, 8, 8, 16), (1, 7, 9, 5, 5), (3, 33, 40)):
        p = rng.dirichlet(np.ones(shape[-1]), shape[:-1]).astype(F)
        q = (0.6 * p + 0.4 * rng.dirichlet(np.ones(shape[-1]), shape[:-1])).astype(F)
        pg, qg = G(p, dev, True), G(q, dev, True)
        val = MI().maps(pg, qg)
        w = rng.standard_normal(tuple(val.shape)).astype(F)
        (-(val * G(w, dev)).sum()).backward()
        po, qo = torch.from_numpy(p).double().requires_grad_(), torch.from_numpy(q).double().requires_grad_()
        ref = go.mi_maps(po, qo)
        (-(ref * torch.from_numpy(w).double()).sum()).backward()
        np.testing.assert_allclose(N(val), ref.detach().numpy(), rtol=2e-4, atol=2e-6)
        for got, want, nm in ((pg.grad, po.grad, 'x'), (qg.grad, qo.grad, 'y')):
            want = want.numpy()
            assert got.shape == want.shape
            err = np.abs(N(got) - want).max() / np.abs(want).max()
            assert err < 5e-4, (shape, nm, err)
    # only one side needs a gradient (a fixed atlas prior against a predicted segmentation)
    pg = G(p, dev, True)
    MI().segs(pg, G(q, dev)).sum().backward()
    po = torch.from_numpy(p).double().requires_grad_()
    go.mi_maps(po, torch.from_numpy(q).double()).sum().backward()
    assert np.abs(N(pg.grad) - po.grad.numpy()).max() / np.abs(po.grad.numpy()).max() < 5e-4
