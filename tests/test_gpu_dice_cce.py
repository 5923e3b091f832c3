"""
GPU parity tests of Dice (soft / hard) and label-weighted categorical cross-entropy against the golden
vectors of the reference's own source and against the oracle.
Tolerances: hard / one-hot Dice BIT-EXACT (integer counting); soft Dice and CCE 1e-5 relative
(float32 reduction order is unspecified in TensorFlow; the oracle accumulates in float64).
"""

import os
import warnings

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import bits_equal, load_golden
from neurite_amd import synth
from oracle import c_oracle as co
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32
RTOL = 1e-5


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def test_dice_golden(dev):
    g = load_golden('dice_small')
    L = g['oh_t'].shape[-1]
    t, p = G(g['pr_t'], dev), G(g['pr_p'], dev)
    oh_t, oh_p = G(g['oh_t'], dev), G(g['oh_p'], dev)
    lt, lp = G(g['lab_t'], dev), G(g['lab_p'], dev)
    assert bits_equal(N(ne.metrics.Dice().dice(oh_t, oh_p)), g['soft_onehot'])
    np.testing.assert_allclose(N(ne.metrics.SoftDice().dice(t, p)), g['soft_prob'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.metrics.SoftDice(laplace_smoothing=0.1).dice(t, p)), g['soft_prob_laplace'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.metrics.SoftDice(normalize=True).dice(G(g['unn_t'], dev), G(g['unn_p'], dev))),
                               g['soft_prob_normalize'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.metrics.Dice().mean_dice(t, p)), g['mean_soft_prob'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.metrics.Dice(weights=g['w']).mean_dice(t, p)), g['mean_soft_prob_w'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.metrics.Dice(weights=G(g['w'], dev)).mean_dice(t, p)), g['mean_soft_prob_w'], rtol=RTOL)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert bits_equal(N(ne.metrics.HardDice(L, input_type='prob').dice(t, p)), g['hard_prob'])
        assert any('hard* dice' in str(x.message) for x in w)            # metrics.py:455
        assert bits_equal(N(ne.metrics.Dice(dice_type='hard', input_type='prob').dice(t, p)), g['hard_prob_nolabels'])
    assert bits_equal(N(ne.metrics.HardDice(L).dice(lt, lp)), g['hard_label'])
    assert bits_equal(N(ne.metrics.HardDice(L, laplace_smoothing=1.0).dice(lt, lp)), g['hard_label_laplace'])
    assert bits_equal(N(ne.metrics.HardDice(L).dice(lt.to(torch.int64), lp.to(torch.uint8))), g['hard_label'])
    np.testing.assert_allclose(N(ne.losses.Dice().loss(t, p)), g['loss_soft_prob'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.losses.Dice().mean_loss(t, p)), g['mean_loss_soft_prob'], rtol=RTOL)
    assert bits_equal(N(ne.losses.HardDice(L).loss(lt, lp)), g['loss_hard_label'])
    with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):    # metrics.py:439-444
        ne.metrics.Dice().dice(t, G(g['bad'], dev))
    np.testing.assert_allclose(N(ne.metrics.Dice(check_input_limits=False).dice(t, G(g['bad'], dev))),
                               g['not_checked_bad'], rtol=RTOL)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        dl = ne.metrics.Dice().loss(t, p)                                  # deprecated path, :512-519
        assert any('deprecated' in str(x.message) for x in w)
    np.testing.assert_allclose(N(dl), -g['mean_soft_prob'], rtol=RTOL)
    with pytest.raises(TypeError, match='integer label ids'):
        ne.metrics.HardDice(L).dice(t[..., 0], p[..., 0])
    with pytest.raises(AssertionError, match='weights should be a matrix'):
        ne.metrics.Dice(weights=np.ones(L, F)).mean_dice(t, p)


def test_dice_on_16bit_probability_maps(dev):
    """neurite/tf/metrics.py:415-482 is dtype-agnostic (BASELINE config 5 is a bfloat16 pipeline).  Here 16-bit maps are STORAGE:
    the kernels widen to float32 (exact) and run the float32 arithmetic, results are float32 -- so they are bit-identical to the
    float32 kernels on the widened maps, and agree with the reference's own float16 evaluation (reference through the golden shim:
    float16 products and quotients, wide accumulation) to float16 resolution."""
    g = load_golden('dice_small')
    L = g['oh_t'].shape[-1]
    ht, hp = G(g['f16_t'], dev), G(g['f16_p'], dev)
    assert ht.dtype == torch.float16
    tol = dict(rtol=3e-3, atol=1e-3)                                       # float16: 2^-11 per rounded product / quotient
    np.testing.assert_allclose(N(ne.metrics.SoftDice().dice(ht, hp)), g['f16_soft_prob'].astype(F), **tol)
    np.testing.assert_allclose(N(ne.metrics.SoftDice(laplace_smoothing=0.1).dice(ht, hp)), g['f16_soft_prob_laplace'].astype(F), **tol)
    np.testing.assert_allclose(N(ne.metrics.Dice().mean_dice(ht, hp)), g['f16_mean_soft_prob'].astype(F), **tol)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        np.testing.assert_allclose(N(ne.metrics.HardDice(L, input_type='prob').dice(ht, hp)), g['f16_hard_prob'].astype(F), **tol)
    rng = np.random.default_rng(31)
    for Lk, S in ((32, (9, 10, 11)), (8, (7, 5, 6)), (5, (6, 5, 7)), (20, (4, 9, 3))):          # vector kernels and the generic ones
        t32 = torch.from_numpy(rng.random((2,) + S + (Lk,)).astype(F)).to(dev)
        p32 = torch.from_numpy(rng.random((2,) + S + (Lk,)).astype(F)).to(dev)
        for dt in (torch.bfloat16, torch.float16):
            t, p = t32.to(dt), p32.to(dt)
            tw, pw = t.float(), p.float()
            for kw in (dict(), dict(laplace_smoothing=0.3), dict(normalize=True)):
                m = ne.metrics.Dice(check_input_limits=False, **kw)
                d = m.dice(t, p)
                assert d.dtype == torch.float32 and bits_equal(N(d), N(m.dice(tw, pw))), (Lk, dt, kw)
                assert bits_equal(N(m.mean_dice(t, p)), N(m.mean_dice(tw, pw)))
            assert bits_equal(N(ne.metrics.Dice().dice(t, p)), N(ne.metrics.Dice().dice(tw, pw)))      # range asserts on the widened extrema
            assert bits_equal(N(ne.losses.Dice(check_input_limits=False).loss(t, p)), N(ne.losses.Dice(check_input_limits=False).loss(tw, pw)))
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                for kw in (dict(), dict(check_input_limits=False), dict(laplace_smoothing=1.0)):
                    h = ne.metrics.HardDice(Lk, input_type='prob', **kw)
                    assert bits_equal(N(h.dice(t, p)), N(h.dice(tw, pw))), (Lk, dt, kw)
            # mixed storage: widened to float32
            assert bits_equal(N(ne.metrics.Dice(check_input_limits=False).dice(t, pw)), N(ne.metrics.Dice(check_input_limits=False).dice(tw, pw)))
        # range assert fires on 16-bit maps too
        bad = p32.clone()
        bad[0, 0, 0, 0, 0] = 1.5
        with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):
            ne.metrics.Dice().dice(t32.bfloat16(), bad.bfloat16())
    # gradients: float32 arithmetic, returned in the maps' dtype
    t = torch.rand(1, 5, 6, 7, 8, device=dev).bfloat16()
    p = torch.rand(1, 5, 6, 7, 8, device=dev).bfloat16().requires_grad_()
    ne.metrics.Dice(check_input_limits=False).mean_dice(t, p).backward()
    p2 = p.detach().float().requires_grad_()
    ne.metrics.Dice(check_input_limits=False).mean_dice(t.float(), p2).backward()
    assert p.grad.dtype == torch.bfloat16 and bits_equal(N(p.grad.float()), N(p2.grad.bfloat16().float()))
    with pytest.raises(NotImplementedError, match='float64'):
        ne.metrics.Dice().dice(t.double(), t.double())


@pytest.mark.parametrize('L', [1, 3, 4, 5, 8, 12, 16, 20, 24, 32, 36, 64, 100, 128, 132, 252, 256, 260, 300])
def test_soft_dice_label_counts(dev, L):
    rng = np.random.default_rng(L)
    B, V = 2, 3001
    t = rng.random((B, V, L)).astype(F)
    p = rng.random((B, V, L)).astype(F)
    for normalize in (False, True):
        for eps in (0., 0.5):
            got = N(ne.metrics.Dice(normalize=normalize, laplace_smoothing=eps).dice(G(t, dev), G(p, dev)))
            want = npo.dice(t, p, normalize=normalize, laplace_smoothing=eps)
            np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=str((normalize, eps)))
    sums, d, mm = ne.metrics.dice_partial_sums(G(t, dev), G(p, dev))
    ref = np.stack(npo.dice_sums(t, p), 1)
    np.testing.assert_allclose(N(sums), ref, rtol=RTOL)
    assert N(mm).tolist() == [t.min(), t.max(), p.min(), p.max()]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hard = N(ne.metrics.Dice(dice_type='hard', input_type='prob').dice(G(t, dev), G(p, dev)))
        assert bits_equal(hard, npo.dice(t, p, dice_type='hard'))
        # ties -> lowest label (tf.argmax)
        tt = np.zeros((1, 50, L), F)
        pp = np.zeros((1, 50, L), F)
        pp[0, :, L - 1] = 0
        hard0 = N(ne.metrics.Dice(dice_type='hard', input_type='prob').dice(G(tt, dev), G(pp, dev)))
        assert hard0[0, 0] == 1 and np.all(hard0[0, 1:] == 0)


def test_hard_label_dice_many_labels_and_out_of_range(dev):
    rng = np.random.default_rng(12)
    for L in (2, 33, 2036, 20000):
        t = rng.integers(-2, L + 3, (3, 7, 9, 11)).astype(np.int32)
        p = np.where(rng.random(t.shape) < 0.6, t, rng.integers(-2, L + 3, t.shape)).astype(np.int32)
        got = N(ne.metrics.HardDice(L).dice(G(t, dev), G(p, dev)))
        want = npo.dice(t, p, dice_type='hard', input_type='max_label', nb_labels=L)
        assert bits_equal(got, want), L


def test_hard_dice_single_pass_paths(dev):
    """round 2: the hard-from-probabilities kernel returns the extrema of its inputs from the counting pass (the range asserts
    no longer cost a soft pass), and the label-map kernel counts per distinct label of a wave and reduces block histograms
    as rows; both equal the oracle bit for bit, piecewise-constant and random maps, ragged sizes, with and without workspace"""
    import ctypes
    from neurite_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(21)
    # ---- probabilities: extrema + counts ----
    for shape in ((2, 17, 13, 11, 32), (1, 40, 40, 40, 8), (3, 5, 5, 5, 4)):
        t = rng.random(shape).astype(F)
        p = rng.random(shape).astype(F)
        got = N(ne.metrics.HardDice(shape[-1], input_type='prob').dice(G(t, dev), G(p, dev)))
        assert bits_equal(got, npo.dice(t, p, dice_type='hard', input_type='prob', nb_labels=shape[-1]))
        tt, pp = G(t, dev), G(p, dev)
        B, L = shape[0], shape[-1]
        V = t.size // (B * L)
        counts = torch.empty((B, 3, L), dtype=torch.int64, device=dev)
        d = torch.empty((B, L), dtype=torch.float32, device=dev)
        mm = torch.empty((4,), dtype=torch.float32, device=dev)
        nws = lib.nrt_dice_workspace_bytes(V, L, B)
        ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
        rc = lib.nrt_dice_hard_prob_minmax_f32(_lib.ptr(tt), _lib.ptr(pp), V, L, B, 0.0, _lib.ptr(counts), _lib.ptr(d), _lib.ptr(mm),
                                               _lib.ptr(ws), nws, _lib.stream_ptr(dev))
        assert rc == 0
        assert N(mm).tolist() == [t.min(), t.max(), p.min(), p.max()]
        assert bits_equal(N(d), got)
    bad = rng.random((1, 6, 6, 6, 8)).astype(F)
    bad[0, 3, 3, 3, 5] = 1.5
    with pytest.raises(ne.metrics.InvalidArgumentError):
        ne.metrics.HardDice(8, input_type='prob').dice(G(bad, dev), G(np.abs(bad) / 2, dev))
    # 20 / 24 labels: lane-groups of 8 with 5 / 6 real lanes (the extrema come from the counting pass too); 22: the generic kernel
    for L in (20, 24, 22):
        t = rng.random((2, 9, 9, 9, L)).astype(F)
        p = rng.random((2, 9, 9, 9, L)).astype(F)
        assert bits_equal(N(ne.metrics.HardDice(L, input_type='prob').dice(G(t, dev), G(p, dev))),
                          npo.dice(t, p, dice_type='hard', input_type='prob', nb_labels=L))
        bad = p.copy()
        bad[1, 8, 8, 8, L - 1] = 1.25
        with pytest.raises(ne.metrics.InvalidArgumentError):
            ne.metrics.HardDice(L, input_type='prob').dice(G(t, dev), G(bad, dev))
    # ---- label maps ----
    blobs = synth.one_hot_volume(5, 48, 8, dev).argmax(-1).to(torch.int32)
    other = synth.one_hot_volume(6, 48, 8, dev).argmax(-1).to(torch.int32)
    cases = [(N(blobs)[None], N(other)[None], 8),
             (rng.integers(0, 40, (2, 33, 17, 5)).astype(np.int32), rng.integers(-1, 41, (2, 33, 17, 5)).astype(np.int32), 40),
             (rng.integers(0, 3, (1, 1027)).astype(np.int32), rng.integers(0, 3, (1, 1027)).astype(np.int32), 3)]
    for t, p, L in cases:
        want = npo.dice(t, p, dice_type='hard', input_type='max_label', nb_labels=L)
        assert bits_equal(N(ne.metrics.HardDice(L).dice(G(t, dev), G(p, dev))), want), L
        tt, pp = G(t, dev), G(p, dev)
        B = t.shape[0]
        V = t.size // B
        counts = torch.empty((B, 3, L), dtype=torch.int64, device=dev)
        d = torch.empty((B, L), dtype=torch.float32, device=dev)
        rc = lib.nrt_dice_hard_label_i32(_lib.ptr(tt), _lib.ptr(pp), V, L, B, 0.0, _lib.ptr(counts), _lib.ptr(d), None, 0,
                                         _lib.stream_ptr(dev))                    # no workspace: the global-atomic form
        assert rc == 0 and bits_equal(N(d), want), L


def test_dice_deterministic_and_empty(dev):
    t = torch.rand(2, 40, 40, 40, 32, device=dev)
    p = torch.rand(2, 40, 40, 40, 32, device=dev)
    a = N(ne.metrics.Dice().dice(t, p))
    for _ in range(3):
        assert bits_equal(N(ne.metrics.Dice().dice(t, p)), a)            # no float atomics anywhere
    z = torch.zeros(1, 4, 4, 4, 32, device=dev)
    assert float(ne.metrics.Dice().dice(z, z).abs().max()) == 0           # divide_no_nan
    e = torch.zeros(2, 0, 32, device=dev)
    assert tuple(ne.metrics.Dice(check_input_limits=False).dice(e, e).shape) == (2, 32)


def test_full_size_dice_cfg2(dev):
    """160^3 x 32 one-hot volumes: every sum is an integer < 2^24, so any summation order is exact."""
    mov, fix, trf = synth.cfg2_batch(2, 160, 32, device=dev, seed0=1)
    got = N(ne.metrics.Dice().dice(fix, mov))
    sums, _ = co.dice_sums(N(fix), N(mov))
    assert bits_equal(got, co.dice_from_sums(sums))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hard = N(ne.metrics.HardDice(32, input_type='prob').dice(fix, mov))
    assert bits_equal(hard, got)                                           # one-hot: hard == soft
    # warped (soft) probabilities at full size vs float64 sums
    warped = ne.layers.SpatialTransformer()([mov, trf])
    # a tri-linearly warped one-hot map overshoots 1.0 by an ulp at a few voxels (the float32 weights of a
    # cell sum to 1 +- 2^-23), so the reference's default range assert (metrics.py:443-444) fires on it;
    # the pipeline therefore runs with check_input_limits=False -- and the default must still raise
    D = ne.metrics.Dice(check_input_limits=False)
    got = N(D.dice(fix, warped))
    sums, mm = co.dice_sums(N(fix), N(warped))
    np.testing.assert_allclose(got, co.dice_from_sums(sums), rtol=RTOL)
    assert mm[2] >= 0 and 1.0 < mm[3] <= 1.0 + 1e-6
    with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):
        ne.metrics.Dice().dice(fix, warped)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hard = N(ne.metrics.HardDice(32, input_type='prob', check_input_limits=False).dice(fix, warped))
    c = co.dice_hard_counts_prob(N(fix), N(warped))
    assert bits_equal(hard, co.dice_from_sums(c.astype(np.float64)))
    # label-map pipeline of models.py:806-807: nearest warp of the label volume, hard Dice on ids
    lab_m = torch.argmax(mov, -1).to(torch.float32).unsqueeze(-1)
    lab_f = torch.argmax(fix, -1).to(torch.int32)
    wl = ne.layers.SpatialTransformer('nearest', fill_value=0)([lab_m, trf])[..., 0].to(torch.int32)
    got = N(ne.metrics.HardDice(32).dice(lab_f, wl))
    c = co.dice_hard_counts_label(N(lab_f), N(wl), 32)
    assert bits_equal(got, co.dice_from_sums(c.astype(np.float64)))
    # sharding invariants (the multi-GPU path): per-entry dice is independent of the batch split,
    # and spatially split partial sums add up to the whole
    d_all = D.dice(fix, warped)
    d_0 = D.dice(fix[:1], warped[:1])
    assert bits_equal(N(d_all[:1]), N(d_0))
    s_a, _, _ = ne.metrics.dice_partial_sums(fix[:, :80], warped[:, :80])
    s_b, _, _ = ne.metrics.dice_partial_sums(fix[:, 80:], warped[:, 80:])
    s_all, _, _ = ne.metrics.dice_partial_sums(fix, warped)
    np.testing.assert_allclose(N(s_a + s_b), N(s_all), rtol=1e-6)
    d_split = ne.distributed.dice_from_sums(ne.distributed.reduce_dice_sums(s_a + s_b))
    np.testing.assert_allclose(N(d_split), N(d_all), rtol=1e-6)
    m = ne.distributed.all_reduce_mean_dice(d_all)
    np.testing.assert_allclose(float(m), float(D.mean_dice(fix, warped)), rtol=1e-6)


def test_fused_warp_dice(dev):
    """Fused SpatialTransformer+Dice == the two-kernel pipeline: warped bit-identical, Dice to 1e-6."""
    rng = np.random.default_rng(5)
    for (B, S, So, L) in ((2, (19, 14, 27), (19, 14, 27), 32), (1, (9, 8, 12), (7, 11, 5), 8), (2, (12, 12, 12), (12, 12, 12), 4),
                          (1, (10, 9, 17), (10, 9, 17), 64)):
        mov = rng.random((B,) + S + (L,)).astype(F)
        fix = rng.random((B,) + So + (L,)).astype(F)
        trf = rng.normal(0, 2.5, (B,) + So + (3,)).astype(F)
        for fill in (None, 0.0):
            for tune in (0, 1 | (1 << 4) | (3 << 8), 3 | (3 << 4) | (3 << 8) | (1 << 12), (1 << 13) | (5 << 16), (1 << 13) | (1 << 12)):
                d, w, s = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=fill, return_warped=True,
                                             return_sums=True, laplace_smoothing=0.25, _tune=tune)
                w_ref = npo.spatial_transformer(mov, trf, fill_value=fill)
                assert bits_equal(N(w), w_ref), (S, L, fill, tune)
                np.testing.assert_allclose(N(s), np.stack(npo.dice_sums(fix, w_ref), 1), rtol=RTOL)
                np.testing.assert_allclose(N(d), npo.dice(fix, w_ref, laplace_smoothing=0.25, check_input_limits=False), rtol=RTOL)
                d2 = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=fill, laplace_smoothing=0.25, _tune=tune)
                assert bits_equal(N(d2), N(d))                      # with / without writing `warped`
    # full size, against the unfused HIP pipeline and the C oracle
    mov, fix, trf = synth.cfg2_batch(2, 160, 32, device=dev, seed0=1)
    d, w = ne.fused.warp_dice(mov, trf, fix, return_warped=True)
    w2 = ne.layers.SpatialTransformer()([mov, trf])
    assert torch.equal(w, w2)
    d2 = ne.metrics.Dice(check_input_limits=False).dice(fix, w2)
    np.testing.assert_allclose(N(d), N(d2), rtol=1e-6)
    sums, _ = co.dice_sums(N(fix), N(w2))
    np.testing.assert_allclose(N(d), co.dice_from_sums(sums), rtol=RTOL)
    with pytest.raises(ne.errors.InvalidArgumentError):
        ne.fused.warp_dice(mov, trf, fix, check_input_limits=True)
    with pytest.raises(NotImplementedError):
        ne.fused.warp_dice(mov[..., :5], trf, fix[..., :5])


# ------------------------------------------------------------------------------------------- CCE
def test_cce_golden(dev):
    g = load_golden('cce_small')
    t, p, z, w, sw = G(g['t'], dev), G(g['p'], dev), G(g['z'], dev), g['w'], G(g['sw'], dev)
    C = ne.metrics.CategoricalCrossentropy
    np.testing.assert_allclose(N(C()(t, p)), g['plain'], rtol=RTOL)
    np.testing.assert_allclose(N(C(label_weights=w).cce(t, p)), g['weighted'], rtol=RTOL)
    np.testing.assert_allclose(N(C(label_weights=list(w))(t, p)), g['weighted'], rtol=RTOL)
    np.testing.assert_allclose(N(C(label_weights=w, label_smoothing=0.1)(t, p)), g['weighted_smooth'], rtol=RTOL)
    np.testing.assert_allclose(N(C(label_weights=w, from_logits=True)(t, z)), g['weighted_logits'], rtol=RTOL)
    np.testing.assert_allclose(N(C(label_weights=w)(t, p, sample_weight=sw)), g['weighted_sw'], rtol=RTOL)
    np.testing.assert_allclose(N(ne.losses.CategoricalCrossentropy(label_weights=w).loss(t, p)), g['loss_weighted'], rtol=RTOL)
    pv = N(C(label_weights=w, reduction='none')(t, p))
    np.testing.assert_allclose(pv, npo.cce_per_voxel(g['t'], g['p'], w), rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(N(C(label_weights=w, reduction='sum')(t, p)), pv.sum(), rtol=RTOL)


@pytest.mark.parametrize('C', [2, 4, 6, 12, 16, 20, 32, 64, 100, 252, 260])
def test_cce_channel_counts_and_bf16(dev, C):
    rng = np.random.default_rng(C)
    N_ = (2, 13, 11, 7)
    lab = rng.integers(0, C, N_)
    t = np.eye(C, dtype=F)[lab]
    p = (rng.random(N_ + (C,)) + 0.01).astype(F)
    w = (rng.random(C) + 0.3).astype(F)
    z = (rng.standard_normal(N_ + (C,)) * 2).astype(F)
    for kw in ({}, {'label_smoothing': 0.2}):
        got = float(ne.metrics.CategoricalCrossentropy(label_weights=w, **kw)(G(t, dev), G(p, dev)))
        np.testing.assert_allclose(got, npo.cce(t, p, w, **kw), rtol=RTOL)
        got = float(ne.metrics.CategoricalCrossentropy(label_weights=w, from_logits=True, **kw)(G(t, dev), G(z, dev)))
        np.testing.assert_allclose(got, npo.cce(t, z, w, from_logits=True, **kw), rtol=RTOL)
    # bf16 inputs: arithmetic in float32 on the bf16-rounded values (tolerance from SURVEY A.9: 1e-3)
    tb, pb = G(t, dev).to(torch.bfloat16), G(p, dev).to(torch.bfloat16)
    got = float(ne.metrics.CategoricalCrossentropy(label_weights=w)(tb, pb))
    want = npo.cce(N(tb.float()), N(pb.float()), w)
    np.testing.assert_allclose(got, want, rtol=1e-4)


def test_cce_cfg5_size_bf16(dev):
    """BASELINE config 5 loss stage: [1, 94, 94, 94, 16] bf16, inverse-frequency label weights."""
    rng = np.random.default_rng(6)
    S, C = (94, 94, 94), 16
    lab = rng.integers(0, C, S)
    t = torch.from_numpy(np.eye(C, dtype=F)[lab])[None].to(dev)
    p = torch.softmax(torch.randn((1,) + S + (C,), device=dev), -1)
    freq = np.bincount(lab.ravel(), minlength=C) / lab.size
    w = (1.0 / freq)
    w = (w / w.sum()).astype(F)
    tb, pb = t.to(torch.bfloat16), p.to(torch.bfloat16)
    got = float(ne.metrics.WeightedCategoricalCrossentropy(label_weights=w)(tb, pb))
    want = co.wcce(N(tb.float()), N(pb.float()), w)
    np.testing.assert_allclose(got, want, rtol=1e-4)
    got32 = float(ne.metrics.CategoricalCrossentropy(label_weights=w)(t, p))
    np.testing.assert_allclose(got32, co.wcce(N(t), N(p), w), rtol=RTOL)


def test_fused_x_march_schedule_ragged(dev):
    """The x-march block schedule (default for 32 labels above ~500 patches) on shapes that are not multiples of the 4 x 8
    patch, the 8 x 4 patch region or the x segment: bit-identical `warped`, Dice equal to the tile schedule and the oracle."""
    rng = np.random.default_rng(77)
    B, S, L = 6, (36, 50, 61), 32
    mov = rng.random((B,) + S + (L,)).astype(F)
    fix = rng.random((B,) + S + (L,)).astype(F)
    trf = rng.normal(0, 2.0, (B,) + S + (3,)).astype(F)
    w_ref = npo.spatial_transformer(mov, trf, fill_value=0.0)
    d_ref = npo.dice(fix, w_ref, check_input_limits=False)
    xm = 3 | (2 << 4) | (3 << 8) | (1 << 14)
    tunes = (0,                                  # auto -> x-march (13 x 8 patches x 6 volumes >= 512)
             3 | (3 << 4) | (4 << 8),            # 8 x 8 x 16 tiles
             xm,                                 # x-march, no regions, one segment
             xm | (3 << 16),                     # three x segments (36 = 3 x 12)
             xm | (5 << 16) | (2 << 24) | (1 << 27),     # five segments (ragged last), 4 x 2 regions
             3 | (3 << 4) | (3 << 8) | (1 << 14) | (1 << 24) | (3 << 27),        # 8 x 8 patches (two passes per plane), 2 x 8 regions
             )
    for tune in tunes:
        d, w = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=0.0, return_warped=True, _tune=tune)
        assert bits_equal(N(w), w_ref), tune
        np.testing.assert_allclose(N(d), d_ref, rtol=RTOL, err_msg=str(tune))
        d2 = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=0.0, _tune=tune)
        assert bits_equal(N(d2), N(d)), tune


def test_fused_wave_cache_kernel(dev):
    """The wave-private LDS row cache form of the fused kernel (csrc/fused_wc.h: default for 32 float32 labels on the x-march schedule)
    against the oracle and against the register kernel (tune bit 30) on coherent fields (cache hits across passes), moderately rough
    ones (orphans in the overflow rows), incoherent ones (more orphans than the overflow holds: register path per pass), locations
    outside the volume, fill values, shapes that are not multiples of the patch, x segments."""
    rng = np.random.default_rng(404)
    NO_WC, WC = 1 << 30, 1 << 29
    xm = 3 | (2 << 4) | (3 << 8) | (1 << 14) | WC
    for (B, S, So) in ((2, (21, 18, 29), (21, 18, 29)), (1, (40, 33, 47), (37, 30, 41)), (3, (16, 8, 8), (16, 5, 9))):
        mov = rng.random((B,) + S + (32,)).astype(F)             # (positive maps: no cancellation in the Dice sums)
        fix = rng.random((B,) + So + (32,)).astype(F)
        fields = {
            'zero': np.zeros((B,) + So + (3,), F),
            'smooth': np.stack([N(synth.smooth_displacement(11 + b, max(So), sigma=2.0, coarse=4, device='cpu'))[:So[0], :So[1], :So[2]]
                                for b in range(B)]),
            'steep': np.stack([N(synth.smooth_displacement(31 + b, max(So), sigma=4.0, coarse=max(2, max(So) // 4), device='cpu'))
                               [:So[0], :So[1], :So[2]] for b in range(B)]),
            'iid0.7': rng.normal(0, 0.7, (B,) + So + (3,)).astype(F),
            'iid3': rng.normal(0, 3.0, (B,) + So + (3,)).astype(F),
            'incoherent': rng.uniform(-max(S), max(S), (B,) + So + (3,)).astype(F),
        }
        for name, trf in fields.items():
            trf = np.ascontiguousarray(trf, F)
            for fill in (None, 0.25):
                w_ref = npo.spatial_transformer(mov, trf, fill_value=fill)
                d_ref = npo.dice(fix, w_ref, check_input_limits=False)
                for tune in (xm, xm | (3 << 16) | (1 << 24) | (1 << 27)):
                    d, w, s = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=fill, return_warped=True,
                                                 return_sums=True, _tune=tune)
                    assert bits_equal(N(w), w_ref), (S, name, fill, tune)
                    np.testing.assert_allclose(N(d), d_ref, rtol=RTOL, err_msg=str((S, name, fill, tune)))
                    d2, s2 = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=fill, return_sums=True, _tune=tune)
                    assert bits_equal(N(d2), N(d)) and bits_equal(N(s2), N(s)), (S, name, fill, tune)
                    # the register kernel on the same schedule: same warped bits, same sums up to the order of the additions
                    d0, w0, s0 = ne.fused.warp_dice(G(mov, dev), G(trf, dev), G(fix, dev), fill_value=fill, return_warped=True,
                                                    return_sums=True, _tune=(tune & ~WC) | NO_WC)
                    assert bits_equal(N(w0), N(w)), (S, name, fill, tune)
                    np.testing.assert_allclose(N(s0), N(s), rtol=2e-5, atol=1e-4)


def test_mean_squared_error_prob(dev):
    """metrics.MeanSquaredErrorProb (neurite/tf/metrics.py:653-692): Keras MSE with label weights as per-element sample weights;
    value and gradients against float64 NumPy / torch"""
    import torch
    rng = np.random.default_rng(71)
    for shape in ((2, 9, 8, 7, 8), (3, 33, 5), (1, 6, 5, 4, 32)):
        L = shape[-1]
        t = rng.random(shape).astype(np.float32)
        p = (t + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        w = rng.random(L).astype(np.float32) + 0.5
        tg = torch.from_numpy(t).to(dev)
        pg = torch.from_numpy(p).to(dev).requires_grad_()
        d2 = (t.astype(np.float64) - p.astype(np.float64)) ** 2
        # no label weights: mean over the label axis, then over the rest (:692 -> keras MSE, SUM_OVER_BATCH_SIZE)
        got = ne.metrics.MeanSquaredErrorProb()(tg, pg)
        np.testing.assert_allclose(float(got.detach()), d2.mean(-1).mean(), rtol=2e-5)
        got.backward()
        np.testing.assert_allclose(pg.grad.cpu().numpy(), 2 * (p.astype(np.float64) - t) / d2.size, rtol=1e-4, atol=1e-9)
        # label weights become sample weights of the per-element losses (:676-690)
        pg.grad = None
        lw = ne.losses.MeanSquaredErrorProb(label_weights=w)
        got = lw.loss(tg, pg)
        np.testing.assert_allclose(float(got.detach()), (d2 * w).mean(), rtol=2e-5)
        got.backward()
        np.testing.assert_allclose(pg.grad.cpu().numpy(), 2 * w * (p.astype(np.float64) - t) / d2.size, rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(float(ne.metrics.MeanSquaredErrorProb(label_weights=w, reduction='sum')(tg, pg.detach())),
                                   (d2 * w).sum(), rtol=2e-5)
        np.testing.assert_allclose(float(ne.metrics.MeanSquaredErrorProb(reduction='sum')(tg, pg.detach(), sample_weight=0.5)),
                                   0.5 * d2.mean(-1).sum(), rtol=2e-5)
    with pytest.raises(ValueError, match='Label weights must be of len'):
        ne.metrics.MeanSquaredErrorProb(label_weights=[1.0, 2.0])(tg, pg)
    with pytest.raises(NotImplementedError):
        ne.metrics.MeanSquaredErrorProb()(tg, pg, sample_weight=[1.0, 2.0])
    with pytest.raises(ValueError, match='Invalid Reduction'):
        ne.metrics.MeanSquaredErrorProb(reduction='mean')


def test_mean_dice_pair_kernel(dev):
    """[sum of dice * weights, count] in one launch (csrc/dice.hip: dice_mean_pair; neurite/tf/metrics.py:499-510): the
    operand of the single all-reduce behind a batch-sharded mean_dice"""
    from neurite_amd import distributed as nd
    rng = np.random.default_rng(5)
    for B, L in ((4, 32), (1, 7), (33, 40), (300, 5)):
        d = rng.random((B, L)).astype(F)
        for w in (None, rng.random(L).astype(F), rng.random((1, L)).astype(F), rng.random((B, L)).astype(F)):
            pair = N(nd.mean_dice_pair(G(d, dev), None if w is None else G(w, dev)))
            dw = d if w is None else d * w.reshape(-1, L)                 # float32 product, as the kernel forms it
            want = dw.astype(np.float64).sum()
            assert pair[1] == B * L
            np.testing.assert_allclose(pair[0], want, rtol=1e-6)
            got = float(nd.all_reduce_mean_dice(G(d, dev), None if w is None else w))
            np.testing.assert_allclose(got, want / (B * L), rtol=1e-6)
            # a process that is alone reads the quotient the kernel formed (nrt_dice_mean_f32): the float32 division of the pair, bit for bit
            assert np.float32(got) == np.float32(pair[0]) / np.float32(pair[1])
            pend = nd.all_reduce_mean_dice(G(d, dev), None if w is None else w, async_op=True)
            assert float(pend.result()) == got
    # run-to-run bit-identical (fixed summation order)
    d = G(rng.random((64, 32)).astype(F), dev)
    a, b = N(nd.mean_dice_pair(d)), N(nd.mean_dice_pair(d))
    assert bits_equal(a, b)
    with pytest.raises(ValueError):
        nd.mean_dice_pair(d, np.ones(5, F))


def test_mse_prob_nearly_equal_maps_do_not_cancel(dev):
    """log-probability maps close to convergence (|t - p| ~ 3e-3 on values of order 1..10): the loss is a reduction of
    (t - p)^2 itself, not of sum t^2 - 2 sum t p + sum p^2 (which loses ~1 % here in float32)"""
    rng = np.random.default_rng(8)
    t = (rng.standard_normal((1, 48, 48, 48, 4)) * 3 - 4).astype(F)
    p = (t + 3e-3 * rng.standard_normal(t.shape)).astype(F)
    w = np.array([0.5, 1.0, 2.0, 4.0], F)
    want = float((w * (t.astype(np.float64) - p.astype(np.float64)) ** 2).mean())
    got = float(ne.metrics.MeanSquaredErrorProb(label_weights=w)(G(t, dev), G(p, dev)))
    assert got > 0
    np.testing.assert_allclose(got, want, rtol=1e-5)


def test_fused_bf16_storage(dev):
    """fused.warp_dice on label maps STORED as bfloat16 (csrc/fused.hip, RowT<unsigned short>): the rows are widened to float32 in
    registers and the float32 arithmetic runs, so sums / dice / min-max equal the float32 kernel on the widened maps bit for bit --
    one-hot maps (values 0 / 1) and general bfloat16-valued maps alike; schedules: x-march (default at this size) and tiles"""
    mov, fix, trf = synth.cfg2_batch(2, 48, 32, device=dev, seed0=5)
    rng = np.random.default_rng(3)
    soft = torch.from_numpy(rng.random((2, 48, 48, 48, 32)).astype(F)).to(dev).bfloat16()
    for m, f in ((mov.bfloat16(), fix.bfloat16()), (soft, fix.bfloat16()), (mov.bfloat16(), soft)):
        for tune in (0, 3 | (3 << 4) | (4 << 8)):
            for fill in (None, 0.0):
                d16, s16 = ne.fused.warp_dice(m, trf, f, fill_value=fill, return_sums=True, _tune=tune)
                d32, s32 = ne.fused.warp_dice(m.float(), trf, f.float(), fill_value=fill, return_sums=True, _tune=tune)
                assert bits_equal(N(s16), N(s32)) and bits_equal(N(d16), N(d32)), (tune, fill)
    # one-hot maps are exact in bfloat16: the same Dice as the float32 pipeline on the original maps
    assert bits_equal(N(ne.fused.warp_dice(mov.bfloat16(), trf, fix.bfloat16())), N(ne.fused.warp_dice(mov, trf, fix)))
    with pytest.raises(NotImplementedError):
        ne.fused.warp_dice(mov.bfloat16(), trf, fix)
    with pytest.raises(NotImplementedError):
        ne.fused.warp_dice(mov.bfloat16(), trf, fix.bfloat16(), return_warped=True)
    # smaller label counts take the same template
    m8, f8, t8 = synth.cfg2_batch(1, 24, 8, device=dev, seed0=9)
    assert bits_equal(N(ne.fused.warp_dice(m8.bfloat16(), t8, f8.bfloat16())), N(ne.fused.warp_dice(m8, t8, f8)))


def test_fused_kernels_recompute_under_graph_replay(dev):
    """a captured hipGraph of the fused warp + Dice (persistent wave-cache kernel with its work counters, and the register kernel) and
    of the stand-alone wave-cache warp must RECOMPUTE on every replay: the inputs are changed between replays.  (With the counters
    reset by a memset node the replays left at once and the second stage re-reduced stale partial sums -- identical inputs hid it.)"""
    mov, fix, trf = synth.cfg2_batch(2, 96, 32, device=dev, seed0=31)
    fix_a, fix_b = fix.clone(), torch.roll(fix, 5, dims=-1).contiguous()
    trf_a, trf_b = trf.clone(), (trf * 0.5).contiguous()

    def capture(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    st10 = ne.layers.SpatialTransformer()
    st10._variant = 10
    keep = ne.deferred.enabled
    ne.deferred.enabled = False
    try:
        cases = [('wave cache', lambda: ne.fused.warp_dice(mov, trf, fix, _tune=1 << 29)),
                 ('register', lambda: ne.fused.warp_dice(mov, trf, fix, _tune=1 << 30)),
                 ('wave-cache warp', lambda: st10([mov, trf]))]
        for name, fn in cases:
            g, out = capture(fn)
            for fsrc, tsrc in ((fix_a, trf_a), (fix_b, trf_a), (fix_b, trf_b), (fix_a, trf_a)):
                fix.copy_(fsrc)
                trf.copy_(tsrc)
                g.replay()
                torch.cuda.synchronize()
                got = out.clone()
                assert torch.equal(got, fn()), name
    finally:
        ne.deferred.enabled = keep
        fix.copy_(fix_a)
        trf.copy_(trf_a)
