"""
GPU parity tests of the unet layers (Conv3D on MFMA, pooling, fused upsample+concat, 1x1+softmax head) and of
the assembled models.unet against the CPU oracle.

Tolerance of the convolutions (north_star: "conv floats within 1e-5 relative"), stated ELEMENT-WISE (`close_conv`):
  * every output:  |gpu - ref| <= 8 * 2^-24 * S,  S = |b| + sum_i |x_i| |w_i|  (the float64 sum of the absolute terms of that output,
    computed by the same oracle on |x|, |w|, |b|) -- the bound of a float32 dot product whatever its length K <= 2592; measured
    2.1e-7 .. 3.4e-7 = 3.5 .. 5.7 units of 2^-24 for every kernel (MFMA implicit GEMM, persistent LDS-DMA schedule, folded decoder,
    direct), profiles/r04_lab/conv_elementwise_error.jsonl;
  * every output whose condition number S / |ref| is at most 25:  |gpu - ref| <= 1e-5 |ref|  (measured <= 4.1e-6 on the layers of
    BASELINE config 3, bench.py `unet_fwd.layers[*].max_rel_err`).  With random-sign data S / |ref| is about sqrt(K), so this
    covers 10 % .. 50 % of the outputs of these layers; for the rest the first bound IS the statement: relative error <=
    8 * 2^-24 * (S / |ref|).
  An element-wise 1e-5 for ALL outputs above 1e-3 max|ref| is not what float32 accumulation gives -- in ANY order, TensorFlow's
  included: an output that is 1000 times smaller than its terms carries their rounding (measured 1.6e-4 .. 3.9e-4 there, the same
  for the direct kernel as for the MFMA ones).
Whole networks (errors of one layer feed the next) keep the per-tensor form allclose(rtol=1e-5, atol=1e-5 * max|ref|) (`close`).
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import _lib
from neurite_amd import models as nm
import contextlib
import io
import json
import os
import warnings

from oracle import c_oracle as co
from oracle import keras_graph_oracle as kgo
from oracle import unet_oracle as uo

pytestmark = pytest.mark.gpu
F = np.float32


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def close(got, ref, tol=1e-5):
    ref = np.asarray(ref, np.float64)
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * max(1e-30, np.abs(ref).max()))


def close_conv(got, ref, absref, act=None, rel=1e-5, cond=25.0, ulps=8.0):
    """element-wise criterion of the module docstring.  ref / absref: float64 pre-activation reference and the sum of absolute
    terms; act: the layer's activation (slope <= 1: the bound on the pre-activation carries over; elu adds the 2e-6 of the
    hardware exponential the epilogue uses)"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    absref = np.asarray(absref, np.float64)
    want = uo.elu(ref) if act == 'elu' else (np.maximum(ref, 0) if act == 'relu' else ref)
    err = np.abs(got - want)
    bound = ulps * 2.0 ** -24 * absref + (3e-6 if act == 'elu' else 0.0) + 1e-30
    worst = float((err / bound).max())
    assert worst <= 1.0, 'error %.2f x the float32 dot-product bound (%g units of 2^-24 of the absolute sum)' % (worst, ulps)
    if act is None:
        well = absref <= cond * np.abs(ref)
        assert well.any()
        r = float((err[well] / np.abs(ref[well])).max())
        assert r <= rel, 'relative error %.3g on a well-conditioned output' % r
        return r
    return None


def conv_refs(x, w, b, dilation=1):
    """float64 reference of one batch entry and the sum of the absolute terms of every output"""
    ref = co.conv3d_same(x, w, b, dilation=dilation, elu=False).astype(np.float64)
    absref = co.conv3d_same(np.abs(x), np.abs(w), np.abs(b), dilation=dilation, elu=False).astype(np.float64)
    return ref, absref


def set_weights(conv, rng, scale=None):
    k = conv.kernel.shape
    fan = int(np.prod(k[:-1]))
    w = (rng.standard_normal(k) * (scale or 1.0 / np.sqrt(fan))).astype(F)
    b = (rng.standard_normal(k[-1]) * 0.1).astype(F)
    with torch.no_grad():
        conv.kernel.copy_(torch.from_numpy(w))
        conv.bias.copy_(torch.from_numpy(b))
    return w, b


@pytest.mark.parametrize('cin,cout,shape,k,dil,act', [
    (16, 16, (9, 7, 21), (3, 3, 3), 1, 'elu'), (48, 16, (8, 8, 32), (3, 3, 3), 1, 'elu'),
    (32, 64, (6, 10, 17), (3, 3, 3), 1, None), (16, 32, (12, 5, 16), (3, 3, 3), 2, 'elu'),
    (24, 5, (7, 9, 18), (3, 3, 3), 1, 'relu'), (16, 40, (5, 6, 19), (1, 3, 3), 1, 'elu'),
    (64, 16, (4, 4, 16), (1, 1, 1), 1, None), (8, 16, (10, 10, 10), (3, 3, 3), 1, 'elu'),
])
def test_conv3d_mfma_vs_oracle(dev, cin, cout, shape, k, dil, act):
    rng = np.random.default_rng(cin * 100 + cout)
    conv = nm._Conv('c', cin, cout, k, dil, 'same', act).to(dev)
    w, b = set_weights(conv, rng)
    x = rng.standard_normal((2,) + shape + (cin,)).astype(F)
    variants = [2, 1]                            # MFMA implicit GEMM, direct
    if k == (3, 3, 3) and dil == 1 and cin % 16 == 0:
        variants.append(5)                       # MFMA in the persistent LDS-DMA schedule (csrc/conv_p27.h)
    for variant in variants:
        y = N(conv(G(x, dev), variant=variant))
        for bi in range(2):
            if k == (3, 3, 3) or k == (1, 1, 1) or k == (1, 3, 3):
                ref, absref = conv_refs(x[bi], w, b, dil)
                close_conv(y[bi], ref, absref, act)


@pytest.mark.parametrize('cin,cout,shape,act', [
    (16, 16, (22, 9, 37), 'elu'),             # several tiles per persistent block, ragged on every axis, deferred stores
    (32, 32, (8, 12, 48), None),              # two chunks per tile, two N-tiles
    (48, 40, (6, 6, 20), 'relu'),             # three chunks, partial third N-tile (immediate stores)
    (16, 64, (4, 4, 16), 'elu'),              # one tile
])
def test_conv3d_persistent_schedule(dev, cin, cout, shape, act):
    """nrt_conv3d_f32 variant 5 (persistent blocks, halo by LDS-DMA, deferred stores) against the float64 oracle and the one-tile-per-block
    MFMA kernel (same accumulation order per output: bit-identical)"""
    rng = np.random.default_rng(cin + cout)
    conv = nm._Conv('c', cin, cout, (3, 3, 3), 1, 'same', act).to(dev)
    w, b = set_weights(conv, rng)
    x = rng.standard_normal((3,) + shape + (cin,)).astype(F)
    y5 = N(conv(G(x, dev), variant=5))
    y2 = N(conv(G(x, dev), variant=2))
    for bi in range(3):
        ref, absref = conv_refs(x[bi], w, b)
        close_conv(y5[bi], ref, absref, act)
    np.testing.assert_allclose(y5, y2, rtol=1e-6, atol=1e-6 * np.abs(y2).max())


def test_conv3d_fused_upsample_concat(dev):
    rng = np.random.default_rng(3)
    for (c0, c1, cout, S) in ((32, 64, 32, (8, 12, 16)), (16, 32, 16, (10, 6, 18)), (4, 12, 7, (6, 6, 6))):
        conv = nm._Conv('c', c0 + c1, cout, (3, 3, 3), 1, 'same', 'elu').to(dev)
        w, b = set_weights(conv, rng)
        skip = rng.standard_normal((2,) + S + (c0,)).astype(F)
        lo = rng.standard_normal((2,) + tuple(s // 2 for s in S) + (c1,)).astype(F)
        cat = np.concatenate([skip, lo.repeat(2, 1).repeat(2, 2).repeat(2, 3)], -1)
        mat = N(nm._upsample_concat(G(skip, dev), G(lo, dev), (2, 2, 2)))
        assert np.array_equal(mat, cat)
        for variant in (2, 1):
            y = N(conv(G(skip, dev), lo=G(lo, dev), up=(2, 2, 2), variant=variant))
            y2 = N(conv(G(cat, dev), variant=variant))
            for bi in range(2):
                close(y[bi], uo.conv(cat[bi], w, b, 'elu'))
            np.testing.assert_allclose(y, y2, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('c0,c1,cout,S,act', [
    (16, 32, 16, (8, 8, 32), 'elu'),          # the last decoder level of BASELINE config 3, whole tiles
    (32, 64, 32, (8, 12, 16), 'elu'),         # the level above (two N-tiles)
    (16, 16, 5, (10, 6, 18), None),           # ragged tiles along every axis, partial N-tile
    (32, 16, 48, (6, 14, 34), 'relu'),        # three N-tiles, skip wider than the up-sampled tensor
    (16, 48, 64, (4, 4, 16), 'elu'),          # one tile: every border at once
    (16, 32, 16, (2, 2, 2), None),            # smaller than a tile
])
def test_conv3d_folded_decoder_kernel(dev, c0, c1, cout, S, act):
    """UpSampling3D(2) + concatenate + Conv3D with the up-sampled half as 8 folded taps (nrt_conv3d_up2_f32) against the
    float64 oracle on the materialised concatenation and against the 27-tap implicit GEMM."""
    rng = np.random.default_rng(c0 + 7 * c1 + cout)
    conv = nm._Conv('c', c0 + c1, cout, (3, 3, 3), 1, 'same', act).to(dev)
    w, b = set_weights(conv, rng)
    skip = rng.standard_normal((2,) + S + (c0,)).astype(F)
    lo = rng.standard_normal((2,) + tuple(s // 2 for s in S) + (c1,)).astype(F)
    cat = np.concatenate([skip, lo.repeat(2, 1).repeat(2, 2).repeat(2, 3)], -1)
    y = N(conv(G(skip, dev), lo=G(lo, dev), up=(2, 2, 2), variant=4))
    y27 = N(conv(G(skip, dev), lo=G(lo, dev), up=(2, 2, 2), variant=2))
    for bi in range(2):
        ref, absref = conv_refs(cat[bi], w, b)
        close_conv(y[bi], ref, absref, act)              # the folded taps are pre-summed weights: same bound, other summation order
        close_conv(y27[bi], ref, absref, act)
    np.testing.assert_allclose(y, y27, rtol=1e-5, atol=1e-5 * np.abs(y27).max())
    # auto picks the folded kernel for these shapes: identical bits
    assert np.array_equal(N(conv(G(skip, dev), lo=G(lo, dev), up=(2, 2, 2))), y)


@pytest.mark.parametrize('labels,S,B', [(32, (8, 8, 32), 2), (16, (4, 12, 16), 1), (32, (12, 4, 48), 3), (16, (8, 8, 32), 2),
                                        (32, (64, 64, 64), 1), (16, (32, 64, 64), 2)])   # the last two: several tiles per block
def test_decoder_conv_with_folded_head(dev, labels, S, B):
    """nrt_conv3d_up2_head_f32 (round 5): the last decoder convolution + the 1x1x1 likelihood convolution + the channel soft-max as
    ONE kernel, against the two kernels it replaces (same float32 sums up to the order of the head's 16 products) and against the
    float64 oracle of conv -> ELU -> matmul -> softmax (models.py:1545-1555, :1596, :1601-1605).  (The first versions failed this
    test in two instructive ways, both invisible to the compiler because the instructions sit in asm statements: 16-byte stores
    whose data registers were rewritten two cycles later, and loads whose address SGPRs had just come back from a spill lane.)"""
    rng = np.random.default_rng(labels + S[2])
    c0, c1 = 16, 32
    conv = nm._Conv('c', c0 + c1, 16, (3, 3, 3), 1, 'same', 'elu').to(dev)
    w, b = set_weights(conv, rng)
    head = nm._Conv('h', 16, labels, (1, 1, 1), 1, 'same', None).to(dev)
    hw, hb = set_weights(head, rng)
    skip = rng.standard_normal((B,) + S + (c0,)).astype(F)
    lo = rng.standard_normal((B,) + tuple(s // 2 for s in S) + (c1,)).astype(F)
    assert _lib.lib().nrt_conv3d_up2_head_supported(c0, c1, 16, labels, _lib.ints(list(S))) == 1
    got = N(conv.run_with_head(G(skip, dev), G(lo, dev), head.kernel, head.bias))
    feat = conv(G(skip, dev), lo=G(lo, dev), up=(2, 2, 2), variant=4)
    two = N(nm._conv1x1_softmax(feat, head.kernel, head.bias, True, 0))
    assert got.shape == (B,) + S + (labels,)
    np.testing.assert_allclose(got, two, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(got.sum(-1), 1.0, rtol=0, atol=2e-6)
    # float64 oracle of the whole chain (small volumes; the large ones are there for the blocks that walk several tiles, where the
    # head of tile t runs inside the first chunk of tile t + 1)
    cat = np.concatenate([skip, lo.repeat(2, 1).repeat(2, 2).repeat(2, 3)], -1).astype(np.float64)
    for bi in range(B if S[0] * S[1] * S[2] <= 8192 else 0):
        ref, _ = conv_refs(cat[bi].astype(F), w, b)
        f = np.where(ref > 0, ref, np.expm1(ref))
        z = f @ hw.reshape(16, labels).astype(np.float64) + hb.astype(np.float64)
        z -= z.max(-1, keepdims=True)
        p = np.exp(z)
        p /= p.sum(-1, keepdims=True)
        np.testing.assert_allclose(got[bi], p, rtol=1e-4, atol=1e-6)
    # shapes the folded head does not take
    for args in ((16, 32, 16, 32, [8, 8, 24]), (16, 32, 16, 8, [8, 8, 32]), (32, 16, 16, 32, [8, 8, 32]), (16, 32, 32, 32, [8, 8, 32])):
        assert _lib.lib().nrt_conv3d_up2_head_supported(args[0], args[1], args[2], args[3], _lib.ints(args[4])) == 0


@pytest.mark.parametrize('labels', [16, 32])
def test_folded_head_is_run_to_run_bit_identical(dev, labels):
    """the folded head depends on hand-placed hazard padding inside asm statements (s_nop around 16-byte stores, five wait
    states in front of loads addressed by spilled SGPRs) and on hand-counted vmcnt immediates -- the failures they fixed showed up as
    results that CHANGED from run to run in a few lanes.  So: many launches on the same inputs (several tiles per persistent block, so
    that the head of tile t runs inside the first chunk of tile t + 1), every one bit-identical to the first, which itself agrees with the
    two kernels the fold replaces; other work runs on the device in between to move the timing around."""
    rng = np.random.default_rng(77 + labels)
    c0, c1, S, B = 16, 32, (32, 32, 64), 2
    conv = nm._Conv('c', c0 + c1, 16, (3, 3, 3), 1, 'same', 'elu').to(dev)
    set_weights(conv, rng)
    head = nm._Conv('h', 16, labels, (1, 1, 1), 1, 'same', None).to(dev)
    set_weights(head, rng)
    skip = G(rng.standard_normal((B,) + S + (c0,)).astype(F), dev)
    lo = G(rng.standard_normal((B,) + tuple(s // 2 for s in S) + (c1,)).astype(F), dev)
    first = conv.run_with_head(skip, lo, head.kernel, head.bias).clone()
    feat = conv(skip, lo=lo, up=(2, 2, 2), variant=4)
    two = nm._conv1x1_softmax(feat, head.kernel, head.bias, True, 0)
    np.testing.assert_allclose(N(first), N(two), rtol=2e-5, atol=2e-7)
    filler = torch.empty(1 << 22, device=dev)
    for k in range(24):
        if k % 3 == 1:
            filler.normal_()                         # (a different neighbour in the queue: the launch meets another machine state)
        again = conv.run_with_head(skip, lo, head.kernel, head.bias)
        assert torch.equal(again, first), 'launch %d differs from the first in %d elements' % (k, int((again != first).sum()))


@pytest.mark.parametrize('cout,S,B', [(16, (8, 8, 32), 2), (32, (4, 12, 16), 1), (16, (32, 32, 64), 1)])
def test_first_conv_with_folded_pooling(dev, cout, S, B):
    """nrt_conv3d_c1_pool_f32 (round 5): the single-channel first encoder convolution also emits MaxPooling3D(2) of its output
    (models.py:1378-1388, 1436-1438): both tensors bit-identical to the two kernels it replaces"""
    rng = np.random.default_rng(cout + S[0])
    conv = nm._Conv('c', 1, cout, (3, 3, 3), 1, 'same', 'elu').to(dev)
    set_weights(conv, rng)
    x = G(rng.standard_normal((B,) + S + (1,)).astype(F), dev)
    assert conv.pool_foldable(x, 0)
    y, p = conv.run_with_pool(x)
    y2 = conv(x)
    p2 = nm._maxpool(y2, (2, 2, 2), 'valid')
    assert np.array_equal(N(y), N(y2)) and np.array_equal(N(p), N(p2))
    assert not conv.pool_foldable(G(rng.standard_normal((1, 6, 8, 16, 1)).astype(F), dev), 0)          # not whole tiles
    assert not nm._Conv('d', 1, 8, (3, 3, 3), 1, 'same', 'elu').to(dev).pool_foldable(x, 0)             # 8 features


@pytest.mark.parametrize('cin,cout,S,B', [(16, 32, (8, 8, 32), 2), (16, 32, (40, 40, 48), 1), (32, 64, (12, 16, 16), 1), (16, 48, (8, 4, 16), 3)])
def test_encoder_conv_with_folded_pooling(dev, cin, cout, S, B):
    """nrt_conv3d_pool_f32 (round 6): an encoder convolution over 16 k channels also emits MaxPooling3D(2) of its output
    (models.py:1378-1388, 1436-1438) from the persistent kernel's epilogue: both tensors bit-identical to the convolution in the same
    schedule (variant 5) followed by the pooling kernel -- also with several tiles per persistent block and several batch entries"""
    rng = np.random.default_rng(cin + cout + S[0])
    conv = nm._Conv('c', cin, cout, (3, 3, 3), 1, 'same', 'elu').to(dev)
    set_weights(conv, rng)
    x = G(rng.standard_normal((B,) + S + (cin,)).astype(F), dev)
    assert conv.pool_foldable(x, 0) and conv.pool_foldable(x, 5)
    y, p = conv.run_with_pool(x)
    y2 = conv(x, variant=5)
    p2 = nm._maxpool(y2, (2, 2, 2), 'valid')
    assert np.array_equal(N(y), N(y2)) and np.array_equal(N(p), N(p2))
    assert np.array_equal(N(p), N(nm._maxpool(conv(x), (2, 2, 2), 'valid')))                       # (whatever kernel the layer takes by itself)
    for k in range(6):                                                                              # same bits every time
        y3, p3 = conv.run_with_pool(x)
        assert torch.equal(y3, y) and torch.equal(p3, p)
    assert not conv.pool_foldable(G(rng.standard_normal((1, 6, 8, 16, cin)).astype(F), dev), 0)     # not whole tiles
    assert not nm._Conv('d', cin, 16, (3, 3, 3), 1, 'same', 'elu').to(dev).pool_foldable(x, 0)      # 16 filters: the deferred-store form
    assert not conv.pool_foldable(x, 2)                                                             # another kernel was asked for


def test_unet_forward_folds_the_head(dev):
    """a unet whose last decoder convolution has 16 features and whose volume is made of whole tiles takes the folded head in
    inference (and only there); the prediction equals the layer-by-layer forward, intermediate tensors can still be requested"""
    rng = np.random.default_rng(12)
    model = ne.models.unet(16, (16, 16, 32, 1), 2, 3, 32, feat_mult=2).to(dev)
    _randomise(model, rng)
    x = G(rng.standard_normal((2, 16, 16, 32, 1)).astype(F), dev)
    assert list(model._head_of.values()) == ['unet_likelihood']
    last = list(model._head_of)[0]
    calls = []
    conv = model.layers_by_name[last]
    orig = conv.run_with_head
    conv.run_with_head = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    y = N(model(x))
    assert calls == [1]
    model.fold_head = False
    y2 = N(model(x))
    assert calls == [1]
    np.testing.assert_allclose(y, y2, rtol=2e-5, atol=2e-7)
    model.fold_head = True
    out = model(x, return_tensors=[last, 'unet_prediction'])                            # the feature tensor is wanted: no folding
    assert calls == [1] and out[last].shape[-1] == 16
    np.testing.assert_allclose(N(out['unet_prediction']), y2, rtol=1e-6, atol=1e-8)
    model.train()
    yt = model(x)                                                                       # training graph: every tensor exists
    assert calls == [1] and yt.requires_grad
    model.eval()


def test_conv3d_folded_decoder_kernel_falls_back(dev):
    """channel counts that are not multiples of 16 or other up-sampling factors stay on the 27-tap kernel"""
    rng = np.random.default_rng(5)
    conv = nm._Conv('c', 4 + 12, 7, (3, 3, 3), 1, 'same', 'elu').to(dev)
    set_weights(conv, rng)
    skip = G(rng.standard_normal((1, 6, 6, 6, 4)).astype(F), dev)
    lo = G(rng.standard_normal((1, 3, 3, 3, 12)).astype(F), dev)
    with pytest.raises(NotImplementedError):
        conv(skip, lo=lo, up=(2, 2, 2), variant=4)
    assert np.array_equal(N(conv(skip, lo=lo, up=(2, 2, 2))), N(conv(skip, lo=lo, up=(2, 2, 2), variant=2)))


def test_direct_conv_shapes(dev):
    """First layer (Cin = 1), odd channel counts, VALID padding, large dilation: the direct kernel."""
    rng = np.random.default_rng(8)
    for (cin, cout, S, k, dil, pad) in ((1, 16, (12, 11, 20), (3, 3, 3), 1, 'same'), (3, 5, (9, 8, 7), (3, 3, 3), 1, 'valid'),
                                        (2, 70, (6, 7, 8), (3, 3, 3), 1, 'same'), (4, 4, (14, 13, 12), (3, 3, 3), 4, 'same'),
                                        (5, 3, (6, 6, 9), (2, 2, 2), 1, 'same')):
        conv = nm._Conv('c', cin, cout, k, dil, pad, 'elu').to(dev)
        w, b = set_weights(conv, rng)
        x = rng.standard_normal((1,) + S + (cin,)).astype(F)
        y = N(conv(G(x, dev)))[0]
        xt = torch.from_numpy(x).permute(0, 4, 1, 2, 3).double()
        wt = torch.from_numpy(w).permute(4, 3, 0, 1, 2).double()
        if pad == 'same':
            tot = [(kk - 1) * dil for kk in k]
            padding = []
            for tpad in reversed(tot):                       # F.pad order: last dim first; TF pads floor before
                padding += [tpad // 2, tpad - tpad // 2]
            xt = torch.nn.functional.pad(xt, padding)
        ref = torch.nn.functional.conv3d(xt, wt, torch.from_numpy(b).double(), dilation=dil)[0].permute(1, 2, 3, 0).numpy()
        absref = torch.nn.functional.conv3d(xt.abs(), wt.abs(), torch.from_numpy(b).double().abs(), dilation=dil)[0].permute(1, 2, 3, 0).numpy()
        close_conv(y, ref, absref, 'elu')


def test_pool_softmax_head_elementwise(dev):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 9, 8, 7, 5)).astype(F)
    for pool in ((2, 2, 2), (3, 1, 2)):
        got = N(nm._maxpool(G(x, dev), pool, 'same'))
        for b in range(2):
            assert np.array_equal(got[b], uo.maxpool_same(x[b], pool))
        gotv = N(nm._maxpool(G(x, dev), pool, 'valid'))
        ox, oy, oz = [s // p for s, p in zip(x.shape[1:4], pool)]
        ref = x[:, :ox * pool[0], :oy * pool[1], :oz * pool[2]].reshape(2, ox, pool[0], oy, pool[1], oz, pool[2], 5).max((2, 4, 6))
        assert np.array_equal(gotv, ref)
    z = (rng.standard_normal((3, 11, 13, 33)) * 4).astype(F)
    e = np.exp(z.astype(np.float64) - z.max(-1, keepdims=True))
    close(N(nm._softmax(G(z, dev))), e / e.sum(-1, keepdims=True))
    for cin, cout in ((16, 32), (16, 4), (7, 50)):
        k = (rng.standard_normal((1, 1, 1, cin, cout)) / np.sqrt(cin)).astype(F)
        bb = rng.standard_normal(cout).astype(F)
        xin = rng.standard_normal((2, 5, 6, 7, cin)).astype(F)
        lin = xin.astype(np.float64) @ k.reshape(cin, cout).astype(np.float64) + bb
        close(N(nm._conv1x1_softmax(G(xin, dev), G(k, dev), G(bb, dev), False, 0)), lin)
        e = np.exp(lin - lin.max(-1, keepdims=True))
        close(N(nm._conv1x1_softmax(G(xin, dev), G(k, dev), G(bb, dev), True, 0)), e / e.sum(-1, keepdims=True))
    a, b2 = rng.standard_normal((2, 4, 4, 4, 6)).astype(F), rng.standard_normal((2, 4, 4, 4, 6)).astype(F)
    sc, sh = rng.standard_normal(6).astype(F), rng.standard_normal(6).astype(F)
    close(N(nm._elementwise(G(a, dev), G(b2, dev), act=1)), uo.elu(a.astype(np.float64) + b2))
    close(N(nm._elementwise(G(a, dev), scale=G(sc, dev), shift=G(sh, dev))), a.astype(np.float64) * sc + sh)


def _randomise(model, rng):
    weights, bns = {}, {}
    for name, m in model.layers_by_name.items():
        if isinstance(m, nm._Conv):
            weights[name] = set_weights(m, rng)
        else:
            C = m.gamma.shape[0]
            p = [(1 + 0.1 * rng.standard_normal(C)).astype(F), (0.1 * rng.standard_normal(C)).astype(F),
                 (0.1 * rng.standard_normal(C)).astype(F), (1 + 0.1 * rng.random(C)).astype(F)]
            with torch.no_grad():
                m.gamma.copy_(torch.from_numpy(p[0])); m.beta.copy_(torch.from_numpy(p[1]))
                m.moving_mean.copy_(torch.from_numpy(p[2])); m.moving_variance.copy_(torch.from_numpy(p[3]))
            bns[name] = p
    return weights, bns


def test_unet_small_configs_vs_oracle(dev):
    rng = np.random.default_rng(11)
    # (kwargs, input shape) -- default topology, 2 convs per level, residual + batch-norm, 2-D
    cases = [
        (dict(nb_features=8, nb_levels=3, conv_size=3, nb_labels=6, feat_mult=2), (24, 20, 28, 1), {}),
        (dict(nb_features=8, nb_levels=3, conv_size=3, nb_labels=4, feat_mult=2, nb_conv_per_level=2), (16, 16, 24, 2), {}),
        (dict(nb_features=8, nb_levels=2, conv_size=3, nb_labels=3, feat_mult=2, nb_conv_per_level=2, use_residuals=True,
              batch_norm=-1, conv_dropout=0.2), (12, 16, 20, 2), dict(use_residuals=True)),
        (dict(nb_features=16, nb_levels=3, conv_size=3, nb_labels=5, feat_mult=1, final_pred_activation='linear'),
         (18, 21, 1), {}),          # odd sizes: SAME max-pool keeps partial windows -> skip shapes must still match
    ]
    for kw, ishape, okw in cases:
        if ishape == (18, 21, 1):
            ishape = (16, 24, 1)                         # 2-D net
        model = ne.models.unet(input_shape=ishape, **kw).to(dev)
        weights, bns = _randomise(model, rng)
        x = rng.standard_normal((2,) + ishape).astype(F)
        y = N(model(G(x, dev)))
        assert y.shape == (2,) + ishape[:-1] + (kw['nb_labels'],)
        nd = len(ishape) - 1
        for b in range(2):
            xb = x[b].reshape((1,) * (3 - nd) + ishape)
            w3 = {n: (k.reshape((1,) * (3 - nd) + k.shape) if k.ndim < 5 else k, bb) for n, (k, bb) in weights.items()}
            ref = uo.unet_forward(xb, w3, kw['nb_levels'], kw.get('nb_conv_per_level', 1), pool=(1,) * (3 - nd) + (2,) * nd,
                                  bn_params=bns or None, final_pred_activation=kw.get('final_pred_activation', 'softmax'),
                                  **okw)
            close(y[b].reshape(ref.shape), ref, tol=1e-5)
        if 'final_pred_activation' not in kw:
            np.testing.assert_allclose(y.sum(-1), 1.0, rtol=1e-5)
        # MFMA and direct kernels agree
        model.conv_variant = 1
        y1 = N(model(G(x, dev)))
        model.conv_variant = 0
        np.testing.assert_allclose(y, y1, rtol=1e-4, atol=1e-5)


def test_unet_layer_names_and_errors(dev):
    m = ne.models.unet(16, (32, 32, 32, 1), 3, 3, 4, feat_mult=2)
    assert m.layer_names == ['unet_input', 'unet_conv_downarm_0_0', 'unet_maxpool_0', 'unet_conv_downarm_1_0',
                             'unet_maxpool_1', 'unet_conv_downarm_2_0', 'unet_merge_3', 'unet_conv_uparm_3_0',
                             'unet_merge_4', 'unet_conv_uparm_4_0', 'unet_likelihood', 'unet_prediction']
    assert tuple(m.get_layer('unet_conv_uparm_3_0').kernel.shape) == (3, 3, 3, 96, 32)
    assert tuple(m.get_layer('unet_conv_uparm_4_0').kernel.shape) == (3, 3, 3, 48, 16)
    with pytest.raises(ne.errors.NeuriteAmdError):
        m(torch.zeros(1, 32, 32, 32, 1))
    with pytest.raises(ValueError, match='spatial dimensions must match'):
        ne.models.unet(4, [(8, 8, 8, 1), (8, 8, 4, 1)], 2, 3, 2)
    with pytest.raises(AssertionError, match='list of lists'):
        ne.models.unet([4, 8], (8, 8, 8, 1), None, 3, 2, feat_mult=None)
    with pytest.raises(AssertionError, match='cannot do softmax'):
        ne.models.unet(4, (8, 8, 8, 1), 2, 3, 2, add_prior_layer=True, use_logp=False)
    # multi-input and intermediate tensors
    mi = ne.models.unet(4, [(8, 8, 8, 1), (8, 8, 8, 2)], 2, 3, 2).to(dev)
    out = mi([torch.randn(1, 8, 8, 8, 1, device=dev), torch.randn(1, 8, 8, 8, 2, device=dev)],
             return_tensors=['unet_merge_2', 'unet_likelihood', 'unet_prediction'])
    assert out['unet_merge_2'].shape == (1, 8, 8, 8, 8) and out['unet_likelihood'].shape == (1, 8, 8, 8, 2)


def test_unet_cfg3_full_size(dev):
    """BASELINE config 3: unet(16, (160,160,160,1), 3, 3, nb_labels=32, feat_mult=2) forward on 160^3 fp32."""
    rng = np.random.default_rng(5)
    model = ne.models.unet(16, (160, 160, 160, 1), 3, 3, 32, feat_mult=2).to(dev)
    weights, _ = _randomise(model, rng)
    x = np.random.default_rng(4).standard_normal((1, 160, 160, 160, 1)).astype(F)
    names = ['unet_conv_downarm_0_0', 'unet_conv_downarm_1_0', 'unet_conv_downarm_2_0', 'unet_conv_uparm_3_0',
             'unet_conv_uparm_4_0', 'unet_prediction']
    out = model(G(x, dev), return_tensors=names)
    ref = uo.unet_forward(x[0], weights, 3, 1, return_all=True)
    for n in names:
        close(N(out[n])[0], ref[n], tol=1e-5)


with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'unet_graph.json')) as _f:
    REF_GRAPHS = json.load(_f)


@pytest.mark.parametrize('tag', ['res_dil', 'res_dil_1conv', 'res_same_feats', 'layer_nb_feats', 'list_of_lists',
                                 'bn_dropout_res', 'dropout_plain', 'prior_logp', 'prior_p', 'multi_input', 'two_d_pool',
                                 'one_d', 'valid_enc', 'enc_default', 'enc_res_dil', 'dec_alone', 'dec_alone_res',
                                 'dilation_net'])
def test_model_vs_reference_recorded_graph(dev, tag):
    """The HIP network against oracle/keras_graph_oracle.py evaluating the graph the REFERENCE'S builder constructs
    (tests/golden/unet_graph.json, recorded from neurite/tf/models.py by tests/golden/make_golden.py): residual levels with
    dilation (the tail conv of a level has dilation 1 and no activation, models.py:1384-1388), per-conv feature lists,
    batch-norm, prior heads, several inputs, 1-D / 2-D nets with anisotropic pooling, 'valid' padding."""
    case = REF_GRAPHS[tag]
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        warnings.simplefilter('ignore')
        model = getattr(ne.models, case['builder'])(*case['args'], **case['kwargs']).to(dev)
    rng = np.random.default_rng(sum(map(ord, tag)))
    weights, bns = _randomise(model, rng)
    nd = model.ndims
    wk = {n: (k.reshape(k.shape[3 - nd:]), b) for n, (k, b) in weights.items()}          # Keras kernel shapes
    xs = [rng.standard_normal((2,) + tuple(s)).astype(F) for s in model.input_shapes]
    if tag.startswith('prior'):
        xs[1] = np.log(np.abs(xs[1]) + 0.1).astype(F) if tag == 'prior_logp' else np.abs(xs[1])
    y = N(model([G(x, dev) for x in xs] if len(xs) > 1 else G(xs[0], dev)))
    for b in range(2):
        ref = kgo.run(case['graph'], [x[b] for x in xs], wk, bns or None)
        assert y[b].shape == ref.shape
        close(y[b], ref, tol=1e-5)
