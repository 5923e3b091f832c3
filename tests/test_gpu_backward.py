"""
GPU parity tests of the backward kernels (csrc/backward.hip) against the gradient oracle (oracle/grad_oracle.py:
the reference's forward graphs restated in float64 torch, differentiated by autograd with TF's gradient rules;
pinned against finite differences in tests/test_oracle.py).
Tolerance: 1e-4 relative to the gradient scale (float32 products and sums vs float64).
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import synth
from oracle import grad_oracle as go
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32
TOL = 1e-4


def G(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_() if grad else t


def N(t):
    return t.detach().cpu().numpy()


def D64(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).double()
    return t.requires_grad_() if grad else t


def close(got, want, what=''):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max()) / scale
    assert got.shape == want.shape and err < TOL, '%s: max err / scale = %.3g' % (what, err)


def rand_loc(rng, S, O, kinks=True):
    """absolute locations: mostly inside, some outside on both sides, some exactly on integers / the borders"""
    D = len(S)
    loc = np.stack([rng.uniform(-1.5, S[d] + 0.5, size=O) for d in range(D)], -1).astype(F)
    if kinks:
        flat = loc.reshape(-1, D)
        n = flat.shape[0]
        for k in range(0, n, 7):
            flat[k, k % D] = np.float32(rng.integers(0, S[k % D]))
        for k in range(3, n, 11):
            flat[k, k % D] = np.float32(S[k % D] - 1)
        for k in range(5, n, 13):
            flat[k, k % D] = 0.0
    return loc


@pytest.mark.parametrize('S,C,O', [((9, 8, 7), 4, (6, 5, 7)), ((9, 8, 7), 32, (5, 6, 7)), ((9, 8, 7), 3, (4, 5, 6)),
                                    ((9, 8, 7), 1, (4, 5, 6)), ((11, 9), 2, (7, 8)), ((11, 9), 8, (7, 8)), ((13,), 3, (9,))])
@pytest.mark.parametrize('fill', [None, 0.5])
def test_interpn_backward(dev, S, C, O, fill):
    rng = np.random.default_rng(hash((S, C, fill is None)) % 2**32)
    vol = rng.standard_normal(S + (C,)).astype(F)
    loc = rand_loc(rng, S, O)
    w = rng.standard_normal(O + (C,)).astype(F)
    v, l = G(vol, dev, True), G(loc, dev, True)
    out = ne.utils.interpn(v, l, fill_value=fill)
    (out * G(w, dev)).sum().backward()
    vo, lo = D64(vol, True), D64(loc, True)
    ref = go.interpn(vo, lo, fill)
    (ref * D64(w)).sum().backward()
    close(N(out), ref.detach().numpy(), 'forward')
    close(N(l.grad), lo.grad.numpy(), 'grad_loc')
    close(N(v.grad), vo.grad.numpy(), 'grad_vol')


def test_interpn_backward_partial_requires(dev):
    rng = np.random.default_rng(3)
    S, C, O = (6, 7, 8), 8, (5, 5, 5)
    vol, loc = rng.standard_normal(S + (C,)).astype(F), rand_loc(rng, S, O)
    lo = D64(loc, True)
    vo = D64(vol, True)
    go.interpn(vo, lo).sum().backward()
    l = G(loc, dev, True)
    ne.utils.interpn(G(vol, dev), l).sum().backward()
    close(N(l.grad), lo.grad.numpy(), 'loc only')
    v = G(vol, dev, True)
    ne.utils.interpn(v, G(loc, dev)).sum().backward()
    close(N(v.grad), vo.grad.numpy(), 'vol only')
    # no channel axis, list-of-tensors loc (utils.py:106-107, 119-120)
    v1 = G(vol[..., 0], dev, True)
    ll = [G(loc[..., d], dev, True) for d in range(3)]
    ne.utils.interpn(v1, ll).sum().backward()
    vo1, lo1 = D64(vol[..., :1], True), D64(loc, True)
    go.interpn(vo1, lo1).sum().backward()
    close(N(v1.grad), vo1.grad.numpy()[..., 0], 'squeezed vol')
    for d in range(3):
        close(N(ll[d].grad), lo1.grad.numpy()[..., d], 'list loc %d' % d)


def test_nearest_backward(dev):
    """nearest interpolation (utils.py:193-204): the volume receives tf.gather's scatter-add (masked where a fill value applies),
    the locations no gradient (tf.round); 3-D / 2-D, interpn / SpatialTransformer / Resize"""
    rng = np.random.default_rng(12)
    for S, Cc, O in (((5, 6, 4), 2, (4, 7, 5)), ((9, 7), 3, (6, 8))):
        D = len(S)
        vol = rng.standard_normal(S + (Cc,)).astype(F)
        loc = rng.uniform(-1.5, max(S) + 0.5, O + (D,)).astype(F)
        w = rng.standard_normal(O + (Cc,)).astype(F)
        for fill in (None, 0.5):
            v = G(vol, dev, True)
            l = G(loc, dev, True)
            out = ne.utils.interpn(v, l, interp_method='nearest', fill_value=fill)
            (out * G(w, dev)).sum().backward()
            idx = [np.clip(np.rint(loc[..., d]).astype(np.int64), 0, S[d] - 1) for d in range(D)]
            keep = np.ones(O, bool) if fill is None else ~np.any([(loc[..., d] < 0) | (loc[..., d] > S[d] - 1) for d in range(D)], 0)
            want = np.zeros(S + (Cc,), np.float64)
            np.add.at(want, tuple(i[keep] for i in idx), w[keep].astype(np.float64))
            close(N(v.grad), want, 'nearest grad_vol %s' % (fill,))
            assert float(l.grad.abs().max()) == 0.0
    # layers: SpatialTransformer(nearest) and Resize(nearest) are differentiable wrt the volume
    vol = rng.standard_normal((2, 6, 5, 7, 3)).astype(F)
    trf = rng.normal(0, 1.5, (2, 6, 5, 7, 3)).astype(F)
    v = G(vol, dev, True)
    ne.layers.SpatialTransformer(interp_method='nearest')([v, G(trf, dev)]).square().sum().backward()
    vo = D64(vol, True)
    grid = np.stack(np.meshgrid(*[np.arange(s, dtype=F) for s in (6, 5, 7)], indexing='ij'), -1)
    tot = 0
    for b in range(2):
        loc = (grid + trf[b]).astype(F)
        ii = [torch.from_numpy(np.clip(np.rint(loc[..., d]).astype(np.int64), 0, s - 1)) for d, s in enumerate((6, 5, 7))]
        tot = tot + vo[b][ii[0], ii[1], ii[2]].square().sum()
    tot.backward()
    close(N(v.grad), vo.grad.numpy(), 'SpatialTransformer nearest grad_vol')
    v = G(vol, dev, True)
    ne.layers.Resize(2, interp_method='nearest')(v).sum().backward()
    assert abs(float(v.grad.sum()) - 2 * 12 * 10 * 14 * 3) < 1e-3


def _shift_oracle(vol, shift, fill=None):
    """float32 location (grid + shift, one rounding) like the kernel, then float64 interpolation"""
    O = shift.shape[:-1]
    grid = np.stack(np.meshgrid(*[np.arange(s, dtype=F) for s in O], indexing='ij'), -1)
    loc32 = (grid + shift).astype(F)
    lo = D64(loc32, True)
    vo = D64(vol, True)
    return go.interpn(vo, lo, fill), vo, lo


@pytest.mark.parametrize('C', [32, 5])
@pytest.mark.parametrize('fill', [None, 0.0])
def test_spatial_transformer_backward(dev, C, fill):
    rng = np.random.default_rng(5 + C)
    B, S = 2, (12, 10, 9)
    vol = rng.standard_normal((B,) + S + (C,)).astype(F)
    shift = (rng.standard_normal((B,) + S + (3,)) * 2.5).astype(F)
    shift[0, :3] = 0                                          # identity region: every location on a kink
    w = rng.standard_normal((B,) + S + (C,)).astype(F)
    v, s = G(vol, dev, True), G(shift, dev, True)
    out = ne.layers.SpatialTransformer(fill_value=fill)([v, s])
    (out * G(w, dev)).sum().backward()
    for b in range(B):
        ref, vo, lo = _shift_oracle(vol[b], shift[b], fill)
        (ref * D64(w[b])).sum().backward()
        close(N(out[b]), ref.detach().numpy(), 'fwd')
        close(N(s.grad[b]), lo.grad.numpy(), 'grad_shift b%d' % b)
        close(N(v.grad[b]), vo.grad.numpy(), 'grad_vol b%d' % b)


def test_spatial_transformer_backward_variants(dev):
    rng = np.random.default_rng(9)
    B, S, C = 3, (8, 9, 10), 4
    vol = rng.standard_normal((B,) + S + (C,)).astype(F)
    w = rng.standard_normal((B,) + S + (C,)).astype(F)
    # single_transform: one field shared by the batch, gradient summed over the batch
    shift = (rng.standard_normal((1,) + S + (3,)) * 2).astype(F)
    v, s = G(vol, dev, True), G(shift, dev, True)
    out = ne.layers.SpatialTransformer(single_transform=True)([v, s])
    (out * G(w, dev)).sum().backward()
    gsum = 0
    for b in range(B):
        ref, vo, lo = _shift_oracle(vol[b], shift[0])
        (ref * D64(w[b])).sum().backward()
        gsum = gsum + lo.grad.numpy()
        close(N(v.grad[b]), vo.grad.numpy(), 'single grad_vol')
    close(N(s.grad[0]), gsum, 'single grad_shift')
    # 'xy' indexing: first two flow components swapped
    shift = (rng.standard_normal((B,) + S + (3,)) * 2).astype(F)
    s = G(shift, dev, True)
    out = ne.layers.SpatialTransformer(indexing='xy')([G(vol, dev), s])
    (out * G(w, dev)).sum().backward()
    for b in range(B):
        sw = shift[b][..., [1, 0, 2]]
        ref, vo, lo = _shift_oracle(vol[b], sw)
        (ref * D64(w[b])).sum().backward()
        close(N(s.grad[b]), lo.grad.numpy()[..., [1, 0, 2]], 'xy grad_shift')
    # affine transform: gradient reaches the matrix through the dense-shift glue
    mat = (np.eye(3, 4)[None] + rng.standard_normal((B, 3, 4)) * 0.05).astype(F)
    m = G(mat, dev, True)
    out = ne.layers.SpatialTransformer()([G(vol, dev), m])
    (out * G(w, dev)).sum().backward()
    for b in range(B):
        mo = D64(mat[b], True)
        mesh = torch.stack(torch.meshgrid(*[torch.arange(n, dtype=torch.float64) for n in S], indexing='ij'), -1)
        cen = mesh - torch.tensor([(n - 1) / 2 for n in S], dtype=torch.float64)
        loc = cen @ mo[:, :3].T + mo[:, 3] - cen + mesh
        ref = go.interpn(D64(vol[b]), loc)
        (ref * D64(w[b])).sum().backward()
        got, want = N(m.grad[b]), mo.grad.numpy()
        assert np.abs(got - want).max() / np.abs(want).max() < 2e-3      # float32 affine glue moves a few kinks


def test_transform_and_resize_backward(dev):
    rng = np.random.default_rng(11)
    S, C = (9, 7), 4
    vol = rng.standard_normal(S + (C,)).astype(F)
    shift = (rng.standard_normal(S + (2,)) * 1.5).astype(F)
    v, s = G(vol, dev, True), G(shift, dev, True)
    ne.utils.transform(v, s).square().sum().backward()
    ref, vo, lo = _shift_oracle(vol, shift)
    ref.square().sum().backward()
    close(N(s.grad), lo.grad.numpy(), 'transform grad_shift')
    close(N(v.grad), vo.grad.numpy(), 'transform grad_vol')
    # Resize layer / utils.resize: gradient wrt the volume only (linspace grid is constant)
    B, S3 = 2, (6, 5, 4)
    x = rng.standard_normal((B,) + S3 + (4,)).astype(F)
    xv = G(x, dev, True)
    out = ne.layers.Resize(2)(xv)
    w = rng.standard_normal(tuple(out.shape)).astype(F)
    (out * G(w, dev)).sum().backward()
    O = tuple(out.shape[1:-1])
    lin = [npo.tf_linspace(0., S3[d] - 1., O[d]) for d in range(3)]
    loc = np.stack(np.meshgrid(*lin, indexing='ij'), -1)
    for b in range(B):
        xo = D64(x[b], True)
        (go.interpn(xo, D64(loc)) * D64(w[b])).sum().backward()
        close(N(xv.grad[b]), xo.grad.numpy(), 'resize grad_vol')


@pytest.mark.parametrize('eps', [0., 0.1])
def test_soft_dice_backward(dev, eps):
    rng = np.random.default_rng(13)
    B, S, L = 2, (7, 6, 5), 8
    t = rng.uniform(0, 1, (B,) + S + (L,)).astype(F)
    p = rng.uniform(0, 1, (B,) + S + (L,)).astype(F)
    t[..., 3] = 0; p[..., 3] = 0                                # empty label: divide_no_nan
    wl = rng.uniform(0.5, 1.5, (B, L)).astype(F)
    tt, pt = G(t, dev, True), G(p, dev, True)
    m = ne.metrics.Dice(weights=wl, laplace_smoothing=eps)
    loss = -m.mean_dice(tt, pt)
    loss.backward()
    to, po = D64(t, True), D64(p, True)
    ref = -(go.soft_dice(to, po, eps) * D64(wl)).mean()
    ref.backward()
    close(float(loss.detach()), float(ref.detach()), 'loss')
    close(N(pt.grad), po.grad.numpy(), 'grad_pred')
    close(N(tt.grad), to.grad.numpy(), 'grad_true')
    # odd label count (scalar path), only y_pred tracked, losses.Dice wrapper
    t3, p3 = t[..., :3].copy(), p[..., :3].copy()
    p3t = G(p3, dev, True)
    ne.losses.Dice().mean_loss(G(t3, dev), p3t).backward()
    po3 = D64(p3, True)
    (-go.soft_dice(D64(t3), po3).mean()).backward()
    close(N(p3t.grad), po3.grad.numpy(), 'grad_pred L=3')


@pytest.mark.parametrize('L', [8, 5])
def test_soft_dice_normalize_backward(dev, L):
    """Dice(normalize=True) (metrics.py:434-436): gradients wrt the RAW maps through the per-voxel divide_no_nan normalisation,
    against float64 autograd; a voxel whose labels sum to zero gets no gradient"""
    rng = np.random.default_rng(19 + L)
    B, S = 2, (6, 5, 7)
    t = rng.uniform(0, 2, (B,) + S + (L,)).astype(F)
    p = rng.uniform(0, 3, (B,) + S + (L,)).astype(F)
    p[0, 0, 0, 0] = 0                                           # divide_no_nan: normalised to all-zero
    t[1, 2, 3, 4] = 0
    wl = rng.uniform(0.5, 1.5, (1, L)).astype(F)
    tt, pt = G(t, dev, True), G(p, dev, True)
    m = ne.metrics.Dice(weights=wl, normalize=True, check_input_limits=False, laplace_smoothing=0.1)
    loss = -m.mean_dice(tt, pt)
    loss.backward()
    to, po = D64(t, True), D64(p, True)

    def norm(x):
        s = x.sum(-1, keepdim=True)
        z = s == 0
        return torch.where(z, torch.zeros_like(x), x / torch.where(z, torch.ones_like(s), s))
    ref = -(go.soft_dice(norm(to), norm(po), 0.1) * D64(wl)).mean()
    ref.backward()
    close(float(loss.detach()), float(ref.detach()), 'loss')
    close(N(pt.grad), po.grad.numpy(), 'grad_pred')
    close(N(tt.grad), to.grad.numpy(), 'grad_true')
    assert float(pt.grad[0, 0, 0, 0].abs().max()) == 0.0


@pytest.mark.parametrize('logits,ls,reduction', [(False, 0., 'auto'), (False, 0.1, 'sum'), (True, 0., 'auto'),
                                                 (True, 0.1, 'none'), (False, 0., 'none')])
@pytest.mark.parametrize('C', [6, 8, 32])
def test_cce_backward(dev, logits, ls, reduction, C):
    rng = np.random.default_rng(17)
    B, S = 2, (5, 6, 7)
    lab = rng.integers(0, C, (B,) + S)
    t = np.eye(C, dtype=F)[lab]
    x = rng.standard_normal((B,) + S + (C,)).astype(F) if logits else rng.uniform(0, 1, (B,) + S + (C,)).astype(F)
    if not logits:
        x[0, 0, 0, :2] = 0.
        x[0, 0, 0, :2, -1] = 1.                                  # clipped probabilities: zero gradient there
    wl = rng.uniform(0.5, 2, C).astype(F)
    sw = rng.uniform(0.5, 2, (B,)).astype(F)
    xt = G(x, dev, True)
    loss = ne.losses.CategoricalCrossentropy(wl, from_logits=logits, label_smoothing=ls, reduction=reduction)
    use_sw = reduction != 'auto'
    got = loss.cce(G(t, dev), xt, sample_weight=sw if use_sw else None)
    xo = D64(x, True)
    pv = go.cce_per_voxel(D64(t), xo, D64(wl), logits, ls)
    if use_sw:
        pv = pv * D64(sw)[:, None, None, None]
    ref = pv if reduction == 'none' else (pv.sum() if reduction == 'sum' else pv.mean())
    close(N(got), ref.detach().numpy(), 'fwd')
    up = rng.standard_normal(tuple(ref.shape)).astype(F) if reduction == 'none' else np.float32(1.7)
    (got * G(np.asarray(up), dev)).sum().backward()
    (ref * D64(np.asarray(up))).sum().backward()
    close(N(xt.grad), xo.grad.numpy(), 'grad_pred')


def test_registration_loss_backward_end_to_end(dev):
    """-mean Dice(fixed, warp(moving, flow)) + CCE: gradient wrt the flow through both kernels."""
    rng = np.random.default_rng(19)
    B, S, L = 2, (16, 16, 16), 32
    mov = np.stack([synth.one_hot_volume(40 + b, size=S[0], nb_labels=L).numpy() for b in range(B)], 0).astype(F)
    assert mov.shape == (B,) + S + (L,)
    fix = np.roll(mov, 1, axis=2)
    flow = (rng.standard_normal((B,) + S + (3,)) * 1.2).astype(F)
    f = G(flow, dev, True)
    warped = ne.layers.SpatialTransformer()([G(mov, dev), f])
    loss = -ne.metrics.Dice(check_input_limits=False).mean_dice(G(fix, dev), warped) \
        + 0.1 * ne.losses.CategoricalCrossentropy()(G(fix, dev), warped + 0.01)
    loss.backward()
    tot = 0
    grads = []
    for b in range(B):
        ref, vo, lo = _shift_oracle(mov[b], flow[b])
        d = go.soft_dice(D64(fix[b:b + 1]), ref[None]).sum() / (B * L)
        c = go.cce_per_voxel(D64(fix[b]), ref + 0.01).sum() / (B * np.prod(S))
        l = -d + 0.1 * c
        l.backward()
        tot = tot + float(l.detach())
        grads.append(lo.grad.numpy())
    close(float(loss.detach()), tot, 'loss')
    close(N(f.grad), np.stack(grads, 0), 'grad_flow')


@pytest.mark.parametrize('L,fill,eps', [(32, None, 0.), (8, 0.0, 0.1), (4, None, 0.)])
def test_fused_warp_dice_backward(dev, L, fill, eps):
    """ne.fused.warp_dice backward == SpatialTransformer -> Dice backward == oracle."""
    rng = np.random.default_rng(23 + L)
    B, S = 2, (14, 12, 10)
    mov = np.eye(L, dtype=F)[rng.integers(0, L, (B,) + S)]
    fix = np.eye(L, dtype=F)[rng.integers(0, L, (B,) + S)]
    flow = (rng.standard_normal((B,) + S + (3,)) * 2.0).astype(F)
    flow[1, :2] = 0
    wl = rng.uniform(0.5, 1.5, (B, L)).astype(F)
    f = G(flow, dev, True)
    d = ne.fused.warp_dice(G(mov, dev), f, G(fix, dev), fill_value=fill, laplace_smoothing=eps)
    (-(d * G(wl, dev)).mean()).backward()
    f2 = G(flow, dev, True)
    warped = ne.layers.SpatialTransformer(fill_value=fill)([G(mov, dev), f2])
    d2 = ne.metrics.Dice(check_input_limits=False, laplace_smoothing=eps).dice(G(fix, dev), warped)
    (-(d2 * G(wl, dev)).mean()).backward()
    close(N(f.grad), N(f2.grad), 'fused vs unfused')
    for b in range(B):
        ref, vo, lo = _shift_oracle(mov[b], flow[b], fill)
        dd = go.soft_dice(D64(fix[b:b + 1]), ref[None], eps)
        (-(dd * D64(wl[b:b + 1])).sum() / (B * L)).backward()
        close(N(d[b]), dd.detach().numpy()[0], 'dice')
        close(N(f.grad[b]), lo.grad.numpy(), 'grad_flow b%d' % b)
    # single transform and 'xy' indexing
    f1 = G(flow[:1], dev, True)
    d = ne.fused.warp_dice(G(mov, dev), f1, G(fix, dev), single_transform=True, indexing='xy')
    d.sum().backward()
    f3 = G(flow[:1], dev, True)
    warped = ne.layers.SpatialTransformer(single_transform=True, indexing='xy')([G(mov, dev), f3])
    ne.metrics.Dice(check_input_limits=False).dice(G(fix, dev), warped).sum().backward()
    close(N(f1.grad), N(f3.grad), 'single/xy')
    with pytest.raises(NotImplementedError):
        ne.fused.warp_dice(G(mov, dev, True), f1, G(fix, dev), single_transform=True)


def test_backward_x_march_schedule_ragged(dev):
    """The backward gathers take the forward's x-march block schedule for 32 channels above ~500 patches: a ragged shape"""
    rng = np.random.default_rng(31)
    B, S, L = 6, (20, 50, 61), 32
    mov = rng.random((B,) + S + (L,)).astype(F)
    fix = rng.random((B,) + S + (L,)).astype(F)
    flow = (rng.standard_normal((B,) + S + (3,)) * 2.0).astype(F)
    wl = rng.uniform(0.5, 1.5, (B, L)).astype(F)
    f = G(flow, dev, True)
    d = ne.fused.warp_dice(G(mov, dev), f, G(fix, dev))
    (-(d * G(wl, dev)).mean()).backward()
    f2 = G(flow, dev, True)
    warped = ne.layers.SpatialTransformer()([G(mov, dev), f2])
    d2 = ne.metrics.Dice(check_input_limits=False).dice(G(fix, dev), warped)
    (-(d2 * G(wl, dev)).mean()).backward()
    close(N(f.grad), N(f2.grad), 'fused vs unfused')
    for b in (0, B - 1):
        ref, vo, lo = _shift_oracle(mov[b], flow[b])
        dd = go.soft_dice(D64(fix[b:b + 1]), ref[None])
        (-(dd * D64(wl[b:b + 1])).sum() / (B * L)).backward()
        close(N(f.grad[b]), lo.grad.numpy(), 'grad_flow b%d' % b)


@pytest.mark.parametrize('fill', [None, 0.0])
def test_grad_loc_x_march_absolute_locations(dev, fill):
    """d out / d loc of utils.interpn at 32 channels on the x-march schedule (one volume, 23 x 23 patches >= 512): absolute
    locations, a source volume of another shape, locations outside the volume (clipped: zero gradient on that axis; with a
    fill value the voxel is dead), odd x extent -- the software-pipelined kernel (warp_dice_bwd_xm<ABSOLUTE, false>) against
    the float64 oracle"""
    rng = np.random.default_rng(77)
    S, C = (17, 92, 180), 32
    mov = rng.standard_normal((19, 40, 33, C)).astype(F)
    w = rng.standard_normal(S + (C,)).astype(F)
    scale = np.array([18.0 / 16, 39.0 / 91, 32.0 / 179], F)
    grid = np.stack(np.meshgrid(*[np.arange(n, dtype=F) for n in S], indexing='ij'), -1) * scale
    loc = (grid + rng.standard_normal(S + (3,)).astype(F) * 1.5).astype(F)       # some locations leave the volume
    l = G(loc, dev, True)
    out = ne.utils.interpn(G(mov, dev), l, fill_value=fill)
    (out * G(w, dev)).sum().backward()
    vo, lo = D64(mov), D64(loc, True)
    (go.interpn(vo, lo, fill) * D64(w)).sum().backward()
    close(N(l.grad), lo.grad.numpy(), 'grad_loc absolute fill=%r' % (fill,))


@pytest.mark.parametrize('fill', [None, 0.0])
def test_grad_loc_wave_cache_and_register_kernels_agree(dev, fill, monkeypatch):
    """d loss / d loc at 32 channels on the x-march schedule has two kernels: the wave-cache gather (fused_wc.h, BWD; default) and
    the register-pipelined one (warp_dice_bwd_xm; NRT_BWD_WC=0).  Same arithmetic in the same order: identical bits, for the fused
    warp + Dice and for the plain warp, on a ragged shape with locations outside the volume, and both agree with the oracle"""
    rng = np.random.default_rng(91)
    B, S, L = 3, (21, 50, 117), 32
    mov = rng.random((B,) + S + (L,)).astype(F)
    fix = rng.random((B,) + S + (L,)).astype(F)
    flow = (rng.standard_normal((B,) + S + (3,)) * 2.5).astype(F)
    flow[1, 5:9] = 0                                                     # whole planes on the grid (weights exactly 0 / 1)
    wl = rng.uniform(0.5, 1.5, (B, L)).astype(F)
    w = rng.standard_normal((B,) + S + (L,)).astype(F)
    got = {}
    for wc in ('1', '0'):
        monkeypatch.setenv('NRT_BWD_WC', wc)
        f = G(flow, dev, True)
        d = ne.fused.warp_dice(G(mov, dev), f, G(fix, dev), fill_value=fill, laplace_smoothing=0.05)
        (-(d * G(wl, dev)).mean()).backward()
        f2 = G(flow, dev, True)
        out = ne.layers.SpatialTransformer(fill_value=fill)([G(mov, dev), f2])
        (out * G(w, dev)).sum().backward()
        got[wc] = (N(f.grad), N(f2.grad))
    assert np.array_equal(got['1'][0], got['0'][0]), 'fused: %g' % np.abs(got['1'][0] - got['0'][0]).max()
    assert np.array_equal(got['1'][1], got['0'][1]), 'plain: %g' % np.abs(got['1'][1] - got['0'][1]).max()
    for b in (0, 1):
        ref, vo, lo = _shift_oracle(mov[b], flow[b], fill)
        dd = go.soft_dice(D64(fix[b:b + 1]), ref[None], 0.05)
        (-(dd * D64(wl[b:b + 1])).sum() / (B * L)).backward()
        close(got['1'][0][b], lo.grad.numpy(), 'fused grad_flow b%d' % b)
        ref, vo, lo = _shift_oracle(mov[b], flow[b], fill)
        (ref * D64(w[b])).sum().backward()
        close(got['1'][1][b], lo.grad.numpy(), 'plain grad_flow b%d' % b)


@pytest.mark.parametrize('X', [18, 19, 20])
def test_grad_loc_wave_cache_three_states_every_march_length(dev, X, monkeypatch):
    """the backward forms of the wave-cache gather keep three pass states with one shared set of registers for the rows 32.. of a long fetch
    list (fused_wc.h: NST, XSH): marches of 3 k, 3 k + 1, 3 k + 2 planes, coherent and incoherent fields, bit for bit against the
    register-pipelined kernel (NRT_BWD_WC=0) that shares the gradient arithmetic"""
    rng = np.random.default_rng(77 + X)
    B, S, L = 4, (X, 64, 64), 32
    mov = rng.random((B,) + S + (L,)).astype(F)
    fix = rng.random((B,) + S + (L,)).astype(F)
    flow = (rng.standard_normal((B,) + S + (3,)) * 1.5).astype(F)
    flow[3] = rng.uniform(-30, 30, S + (3,)).astype(F)                   # long fetch lists / the fallback of the cache
    w = rng.standard_normal((B,) + S + (L,)).astype(F)
    got = {}
    for wc in ('1', '0'):
        monkeypatch.setenv('NRT_BWD_WC', wc)
        f = G(flow, dev, True)
        d = ne.fused.warp_dice(G(mov, dev), f, G(fix, dev), laplace_smoothing=0.05)
        (-d.mean()).backward()
        f2 = G(flow, dev, True)
        out = ne.layers.SpatialTransformer()([G(mov, dev), f2])
        (out * G(w, dev)).sum().backward()
        got[wc] = (N(f.grad), N(f2.grad))
    assert np.array_equal(got['1'][0], got['0'][0]), 'fused: %g' % np.abs(got['1'][0] - got['0'][0]).max()
    assert np.array_equal(got['1'][1], got['0'][1]), 'plain: %g' % np.abs(got['1'][1] - got['0'][1]).max()


def test_fused_backward_at_bench_size(dev, monkeypatch):
    """BASELINE config 2 / 4 size (160^3 x 32 one-hot maps, the bench's smooth field), two volumes: properties of d (-mean Dice) / d field
    that do not need the (slow) oracle -- (i) the wave-cache and the register-pipelined kernels give the same bits; (ii) the directional
    derivative along a random +-1 direction matches the central difference of the FORWARD kernel (the loss is piecewise smooth in the
    field: 2 h = 2 % of the voxels cross a cell face per axis, hence the 2 % tolerance), also along the gradient itself; (iii) the gradient of the plain warp is linear in
    the incoming gradient"""
    B, S, L = 2, 160, 32
    mov, fix, trf = synth.cfg2_batch(B, S, L, device=dev, seed0=100)
    got = {}
    for wc in ('1', '0'):
        monkeypatch.setenv('NRT_BWD_WC', wc)
        f = trf.clone().requires_grad_(True)
        (-ne.fused.warp_dice(mov, f, fix).mean()).backward()
        got[wc] = f.grad.clone()
    monkeypatch.delenv('NRT_BWD_WC')
    assert torch.equal(got['1'], got['0']), float((got['1'] - got['0']).abs().max())
    g = got['1']
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    d = (torch.randint(0, 2, trf.shape, device=dev, generator=gen).float() * 2 - 1)
    h = 1e-2
    for direction in (d, g / g.abs().max()):                       # (measured: 0.35 % and 0.25 % apart)
        with torch.no_grad():
            lp = -ne.fused.warp_dice(mov, trf + h * direction, fix).double().mean()
            lm = -ne.fused.warp_dice(mov, trf - h * direction, fix).double().mean()
        fd = float((lp - lm) / (2 * h))
        an = float((g.double() * direction.double()).sum())
        assert abs(an) > 1e-6 and abs(fd - an) <= 0.02 * abs(an), (fd, an)
    # plain warp: d out / d field is linear in grad_out (a location outside the volume has gradient 0 either way)
    w1 = torch.randn(mov.shape, device=dev, generator=gen)
    w2 = torch.randn(mov.shape, device=dev, generator=gen)
    st = ne.layers.SpatialTransformer()
    grads = []
    for w in (w1, w2, w1 + 2 * w2):
        f = trf.clone().requires_grad_(True)
        (st([mov, f]) * w).sum().backward()
        grads.append(f.grad)
    lin = grads[0] + 2 * grads[1]
    assert float((grads[2] - lin).abs().max()) <= 1e-4 * float(lin.abs().max())


@pytest.mark.parametrize('C', [1, 3, 5, 8])
@pytest.mark.parametrize('mode', ['1', '0'])
def test_grad_vol_few_channels(dev, C, mode, monkeypatch):
    """d out / d vol at few channels, 3-D (the backward of VecInt / compose and of warped few-channel network outputs): the
    counting-sort merge over 4 x 4 x 8 tiles (interpn_bwd_vol_sort_any, default) and the per-element scatter
    (NRT_BWD_VOL_SORT_ANY=0) against the float64 oracle: smooth field (many duplicate rows), rough field (every pair its own row),
    fill value (masked voxels), output extents that are no multiples of the tile, a source volume of another shape"""
    monkeypatch.setenv('NRT_BWD_VOL_SORT_ANY', mode)
    rng = np.random.default_rng(41 + C)
    B, S = 2, (9, 14, 21)
    mov = rng.standard_normal((B, 11, 13, 17, C)).astype(F)
    w = rng.standard_normal((B,) + S + (C,)).astype(F)
    grid_scale = np.array([10.0 / 8, 12.0 / 13, 16.0 / 20], F)
    for kind, fill in (('smooth', None), ('smooth', 0.0), ('rough', None)):
        flow = (rng.standard_normal((B,) + S + (3,)) * (1.5 if kind == 'smooth' else 12.0)).astype(F)
        grid = np.stack(np.meshgrid(*[np.arange(n, dtype=F) for n in S], indexing='ij'), -1) * grid_scale
        loc = (grid[None] + flow).astype(F)
        v = G(mov, dev, True)
        out = torch.stack([ne.utils.interpn(v[b], G(loc[b], dev), fill_value=fill) for b in range(B)])
        (out * G(w, dev)).sum().backward()
        for b in range(B):
            vo, lo = D64(mov[b], True), D64(loc[b])
            (go.interpn(vo, lo, fill) * D64(w[b])).sum().backward()
            close(N(v.grad[b]), vo.grad.numpy(), 'grad_vol %s fill=%r b%d' % (kind, fill, b))


@pytest.mark.parametrize('fill', [None, 0.0])
@pytest.mark.parametrize('dedup', ['0', '1', '2'])
def test_grad_vol_row_accumulator_kernel(dev, fill, dedup, monkeypatch):
    """d out / d vol at 32 channels under the x-march schedule: the plain scatter (NRT_BWD_VOL_DEDUP=0), the experimental LDS row-accumulator
    table (interpn_bwd_vol_dedup, env NRT_BWD_VOL_DEDUP=1): duplicate rows merged on chip, rows that find no slot go to memory
    directly, the table is flushed when it fills; and the counting-sort merge (interpn_bwd_vol_sort, NRT_BWD_VOL_DEDUP=2, the default) -- against
    the float64 oracle; smooth field (heavy re-use), rough field (every
    pair its own row: the direct path and many flushes)"""
    monkeypatch.setenv('NRT_BWD_VOL_DEDUP', dedup)
    rng = np.random.default_rng(37)
    B, S, L = 3, (18, 47, 60), 32
    mov = rng.standard_normal((B,) + S + (L,)).astype(F)
    w = rng.standard_normal((B,) + S + (L,)).astype(F)
    for kind in ('smooth', 'rough'):
        flow = (rng.standard_normal((B,) + S + (3,)) * (1.5 if kind == 'smooth' else 25.0)).astype(F)
        v = G(mov, dev, True)
        f = G(flow, dev, True)
        out = ne.layers.SpatialTransformer(fill_value=fill)([v, f])
        (out * G(w, dev)).sum().backward()
        for b in (0, B - 1):
            ref, vo, lo = _shift_oracle(mov[b], flow[b], fill)
            (ref * D64(w[b])).sum().backward()
            close(N(v.grad[b]), vo.grad.numpy(), 'grad_vol %s b%d' % (kind, b))
            close(N(f.grad[b]), lo.grad.numpy(), 'grad_flow %s b%d' % (kind, b))
