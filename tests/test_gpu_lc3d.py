"""
GPU parity tests of LocallyConnected3D (implementation 1) against the golden vectors produced by the
reference's own local_conv (tests/golden/lc3d_small.npz) and against the oracle.
Tolerances: float32 1e-5 relative (atol 1e-5 max|ref|); bfloat16 2^-8 relative per element against float64
math on the bf16-rounded inputs and weights.
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import golden_cases, load_golden
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().float().cpu().numpy()


def close(got, ref, tol):
    ref = np.asarray(ref, np.float64)
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * max(1e-30, np.abs(ref).max()))


def make_layer(dev, x, filters, ks, strides, k, b, act=None, variant=0, **kw):
    layer = ne.layers.LocallyConnected3D(filters, ks, strides=strides, activation=act, **kw)
    y0 = layer(x)                                   # builds on x's device / dtype
    with torch.no_grad():
        layer.kernel.copy_(k.to(layer.kernel.dtype))
        layer.bias.copy_(b.to(layer.bias.dtype))
    layer._variant = variant
    assert tuple(y0.shape) == layer.compute_output_shape(tuple(x.shape))
    return layer


def test_lc3d_golden(dev):
    cases = golden_cases(load_golden('lc3d_small'))
    assert len(cases) == 2
    for tag, c in cases.items():
        ks, st = tuple(int(v) for v in c['ks']), tuple(int(v) for v in c['strides'])
        x = G(c['x'], dev)
        for variant in (0, 1, 2):
            if variant == 2 and c['kernel'].shape[-1] % 4:
                continue                                  # the streaming kernel needs 16-byte weight rows
            layer = make_layer(dev, x, c['kernel'].shape[-1], ks, st, G(c['kernel'], dev), G(c['bias'], dev), variant=variant)
            assert tuple(layer.kernel.shape) == c['kernel'].shape and tuple(layer.bias.shape) == c['bias'].shape
            close(N(layer(x)), c['out'], 1e-5)


@pytest.mark.parametrize('cin,cout,S,ks,st', [(16, 16, (12, 12, 12), (3, 3, 3), (1, 1, 1)), (3, 5, (7, 8, 9), (2, 3, 2), (1, 2, 1)),
                                               (8, 32, (6, 9, 7), (3, 3, 3), (1, 1, 1)), (4, 8, (9, 9, 9), (3, 3, 3), (2, 2, 2)),
                                               (16, 64, (5, 5, 6), (3, 3, 3), (1, 1, 1)),
                                               # more than 64 filters on few input channels: 27 / 54 row groups per lane, the forms in
                                               # which two / four waves share a position and meet in LDS (rows of Cout floats)
                                               (2, 128, (4, 5, 4), (3, 3, 3), (1, 1, 1)), (4, 128, (4, 4, 5), (3, 3, 3), (1, 1, 1))])
def test_lc3d_vs_oracle(dev, cin, cout, S, ks, st):
    rng = np.random.default_rng(cin + cout)
    osh = tuple((S[d] - ks[d]) // st[d] + 1 for d in range(3))
    O, Fd = int(np.prod(osh)), int(np.prod(ks)) * cin
    x = rng.standard_normal((2,) + S + (cin,)).astype(F)
    k = (rng.standard_normal((O, Fd, cout)) / np.sqrt(Fd)).astype(F)
    b = rng.standard_normal(osh + (cout,)).astype(F)
    ref = npo.lc3d(x, k, b, ks, st)
    for act, fn in ((None, lambda v: v), ('elu', lambda v: np.where(v > 0, v, np.exp(np.minimum(v, 0)) - 1))):
        layer = make_layer(dev, G(x, dev), cout, ks, st, G(k, dev), G(b, dev), act=act)
        close(N(layer(G(x, dev))), fn(ref), 1e-5)
    # bf16: fp32 accumulation of bf16-rounded operands
    xb, kb, bb = G(x, dev).bfloat16(), G(k, dev).bfloat16(), G(b, dev).bfloat16()
    layer = make_layer(dev, xb, cout, ks, st, kb, bb)
    refb = npo.lc3d(N(xb), N(kb), N(bb), ks, st)
    y = layer(xb)
    assert y.dtype == torch.bfloat16
    close(N(y), refb, 2.0 ** -8)
    # channels_first in / out: the reference flattens the patch channel-major there (layers.py:1176-1186: inputs[:, :, slices]
    # reshaped to feature_dim => f = (cin, kr, kc, kz)) and K.bias_add RESHAPES the [or, oc, oz, filters] bias to
    # (1, filters, or, oc, oz) -- pinned by tests/golden/lc3d_impl.npz::i1_cf; here the same weights in that layout
    T = int(np.prod(ks))
    k_cf = np.ascontiguousarray(k.reshape(O, T, cin, cout).transpose(0, 2, 1, 3).reshape(O, Fd, cout))
    b_cf = np.ascontiguousarray(b.transpose(3, 0, 1, 2)).reshape(osh + (cout,))
    layer_cf = make_layer(dev, G(x, dev).permute(0, 4, 1, 2, 3).contiguous(), cout, ks, st, G(k_cf, dev), G(b_cf, dev),
                          data_format='channels_first')
    ycf = layer_cf(G(x, dev).permute(0, 4, 1, 2, 3).contiguous())
    close(N(ycf.permute(0, 2, 3, 4, 1)), ref, 1e-5)


@pytest.mark.parametrize('batch', [3, 4])
def test_lc3d_many_filters_batch(dev, batch):
    """Cout > 64 with 2 .. 4 batch entries per pass (the split forms' LDS rows are Cout wide, one per batch entry)."""
    rng = np.random.default_rng(batch)
    cin, cout, S, ks = 3, 128, (4, 4, 6), (3, 3, 3)
    osh = tuple(S[d] - ks[d] + 1 for d in range(3))
    O, Fd = int(np.prod(osh)), 27 * cin
    x = rng.standard_normal((batch,) + S + (cin,)).astype(F)
    k = (rng.standard_normal((O, Fd, cout)) / np.sqrt(Fd)).astype(F)
    b = rng.standard_normal(osh + (cout,)).astype(F)
    layer = make_layer(dev, G(x, dev), cout, ks, (1, 1, 1), G(k, dev), G(b, dev))
    close(N(layer(G(x, dev))), npo.lc3d(x, k, b, ks, (1, 1, 1)), 1e-5)
    xb, kb, bb = G(x, dev).bfloat16(), G(k, dev).bfloat16(), G(b, dev).bfloat16()
    layer = make_layer(dev, xb, cout, ks, (1, 1, 1), kb, bb)
    close(N(layer(xb)), npo.lc3d(N(xb), N(kb), N(bb), ks, (1, 1, 1)), 2.0 ** -8)


@pytest.mark.parametrize('batch', [3, 4, 5, 8, 9])
@pytest.mark.parametrize('cin,cout,ks,st,S', [(16, 16, (3, 3, 3), (1, 1, 1), (6, 7, 9)), (16, 32, (3, 3, 3), (1, 1, 1), (5, 6, 7)),
                                               (16, 16, (3, 3, 3), (2, 1, 2), (7, 6, 9)), (8, 16, (2, 2, 2), (1, 1, 1), (5, 5, 6)),
                                               (16, 8, (3, 3, 3), (1, 1, 1), (5, 5, 7)),
                                               # block-staged patches with fewer than 9 (kr, kc) runs, and with row / column strides
                                               (16, 16, (1, 3, 3), (1, 1, 1), (4, 6, 9)), (16, 16, (3, 2, 3), (1, 2, 1), (6, 7, 10)),
                                               (16, 32, (2, 1, 3), (2, 1, 1), (6, 4, 8))])
def test_lc3d_matrix_core_batches(dev, batch, cin, cout, ks, st, S, monkeypatch):
    """3 .. 8 batch entries per weight pass on v_mfma_f32_4x4x1 (csrc/lc3d.hip: lc3d_fwd_mfma; 9 entries = 8 + 1): against the oracle
    and against the vector kernel (NRT_LC_MFMA=0 is read once per process, so the comparison kernel is asked for per entry pair:
    batches of 2 never take the matrix form)"""
    rng = np.random.default_rng(batch * 100 + cin + cout)
    osh = tuple((S[d] - ks[d]) // st[d] + 1 for d in range(3))
    O, Fd = int(np.prod(osh)), int(np.prod(ks)) * cin
    x = rng.standard_normal((batch,) + S + (cin,)).astype(F)
    k = (rng.standard_normal((O, Fd, cout)) / np.sqrt(Fd)).astype(F)
    b = rng.standard_normal(osh + (cout,)).astype(F)
    for act, fn in ((None, lambda v: v), ('relu', lambda v: np.maximum(v, 0))):
        layer = make_layer(dev, G(x, dev), cout, ks, st, G(k, dev), G(b, dev), act=act)
        y = layer(G(x, dev))
        close(N(y), fn(npo.lc3d(x, k, b, ks, st)), 1e-5)
        pairs = torch.cat([layer(G(x[i:i + 2], dev)) for i in range(0, batch, 2)], 0)        # vector kernel, 2 entries per pass
        close(N(y), N(pairs).astype(np.float64), 2e-6)
    xb, kb, bb = G(x, dev).bfloat16(), G(k, dev).bfloat16(), G(b, dev).bfloat16()
    layer = make_layer(dev, xb, cout, ks, st, kb, bb)
    yb = layer(xb)
    assert yb.dtype == torch.bfloat16
    close(N(yb), npo.lc3d(N(xb), N(kb), N(bb), ks, st), 2.0 ** -8)


@pytest.mark.parametrize('cout,dtype,batch', [(16, torch.bfloat16, 8), (32, torch.bfloat16, 5), (16, torch.float32, 6), (16, torch.bfloat16, 3)])
def test_lc3d_matrix_core_many_positions(dev, cout, dtype, batch):
    """enough positions (32^3) that every block of the matrix-core kernels walks several groups: the software pipeline across positions,
    the double-buffered patches and the per-XCD ranges all turn over.  Against the vector kernel (two entries per pass) everywhere and
    against float64 arithmetic on sampled positions; also without a bias."""
    torch.manual_seed(cout + batch)
    S, cin = 34, 16
    x = torch.randn(batch, S, S, S, cin, device=dev).to(dtype)
    for use_bias in (True, False):
        layer = ne.layers.LocallyConnected3D(cout, (3, 3, 3), activation='elu', use_bias=use_bias).to(dev)
        with torch.no_grad():
            layer(x[:1].float())
            layer.to(dtype)
            if use_bias:
                layer.bias.copy_(torch.randn_like(layer.bias))
            y = layer(x)
            pairs = torch.cat([layer(x[i:i + 2]) for i in range(0, batch, 2)], 0)
        tol = 2e-6 if dtype == torch.float32 else 2.0 ** -7            # bfloat16 outputs: one rounding of sums in a different order
        np.testing.assert_allclose(N(y), N(pairs), rtol=tol, atol=tol * float(pairs.float().abs().max()))
        # float64 on sampled positions (first / last of every XCD's range included)
        O = (S - 2) ** 3
        k = layer.kernel.detach().double().cpu().numpy().reshape(O, 27 * cin, cout)
        b = layer.bias.detach().double().cpu().numpy().reshape(O, cout) if use_bias else np.zeros((O, cout))
        xs = x.double().cpu().numpy()
        per = -(-(O // 1) // 8)
        picks = sorted(set([0, O - 1] + [min(O - 1, j * (O // 8) + d) for j in range(8) for d in (0, 1, O // 8 - 1)] +
                           list(np.random.default_rng(3).integers(0, O, 40))))
        for o in picks:
            oz, oc, orr = o % (S - 2), (o // (S - 2)) % (S - 2), o // ((S - 2) ** 2)
            patch = xs[:, orr:orr + 3, oc:oc + 3, oz:oz + 3, :].reshape(batch, -1)
            want = patch @ k[o] + b[o]
            want = np.where(want > 0, want, np.exp(np.minimum(want, 0)) - 1)
            got = N(y)[:, orr, oc, oz, :]
            np.testing.assert_allclose(got, want, rtol=1e-5 if dtype == torch.float32 else 2.0 ** -7, atol=1e-5 if dtype == torch.float32 else 2.0 ** -6)
        assert per > 0


def test_lc3d_softmax_axis_follows_the_data_format(dev):
    """layers.py:1100 hands the layer output to Keras' softmax, which runs over the LAST axis of that tensor: the filters for
    channels_last, the last spatial axis for channels_first."""
    rng = np.random.default_rng(11)
    cin, cout, S, ks = 3, 8, (5, 4, 6), (2, 2, 2)
    osh = tuple(S[d] - ks[d] + 1 for d in range(3))
    O, Fd = int(np.prod(osh)), 8 * cin
    x = rng.standard_normal((2,) + S + (cin,)).astype(F)
    k = (rng.standard_normal((O, Fd, cout)) / np.sqrt(Fd)).astype(F)
    b = rng.standard_normal(osh + (cout,)).astype(F)
    pre = npo.lc3d(x, k, b, ks, (1, 1, 1)).astype(np.float64)               # [B, r, c, z, cout]

    def softmax(v, axis):
        e = np.exp(v - v.max(axis=axis, keepdims=True))
        return e / e.sum(axis=axis, keepdims=True)
    layer = make_layer(dev, G(x, dev), cout, ks, (1, 1, 1), G(k, dev), G(b, dev), act='softmax')
    close(N(layer(G(x, dev))), softmax(pre, -1), 1e-5)
    T = 8
    k_cf = np.ascontiguousarray(k.reshape(O, T, cin, cout).transpose(0, 2, 1, 3).reshape(O, Fd, cout))
    b_cf = np.ascontiguousarray(b.transpose(3, 0, 1, 2)).reshape(osh + (cout,))
    x_cf = G(x, dev).permute(0, 4, 1, 2, 3).contiguous()
    layer_cf = make_layer(dev, x_cf, cout, ks, (1, 1, 1), G(k_cf, dev), G(b_cf, dev), act='softmax', data_format='channels_first')
    y = layer_cf(x_cf)
    assert tuple(y.shape) == (2, cout) + osh
    close(N(y), softmax(pre.transpose(0, 4, 1, 2, 3), -1), 1e-5)


def test_lc3d_contract():
    with pytest.raises(ValueError, match='only "valid" is supported if implementation is 1'):
        ne.layers.LocallyConnected3D(4, 3, padding='same')                 # layers.py:934-936
    with pytest.raises(ValueError, match='kernel_size'):
        ne.layers.LocallyConnected3D(4, (3, 3))
    with pytest.raises(ValueError, match='Unrecognized implementation mode'):
        ne.layers.LocallyConnected3D(4, 3, implementation=7)
    for impl in (2, 3):
        l = ne.layers.LocallyConnected3D(4, 3, padding='same', implementation=impl)
        assert l.compute_output_shape((2, 9, 8, 7, 3)) == (2, 9, 8, 7, 4)
        assert ne.layers.LocallyConnected3D(4, 3, strides=2, padding='same', implementation=impl) \
            .compute_output_shape((2, 9, 8, 7, 3)) == (2, 5, 4, 4, 4)
    l = ne.layers.LocallyConnected3D(4, 3, strides=2, activation='elu', name='lc')
    cfg = l.get_config()
    assert cfg['filters'] == 4 and cfg['kernel_size'] == (3, 3, 3) and cfg['strides'] == (2, 2, 2)
    assert cfg['padding'] == 'valid' and cfg['implementation'] == 1 and cfg['use_bias'] is True and cfg['name'] == 'lc'
    assert l.compute_output_shape((2, 9, 9, 9, 3)) == (2, 4, 4, 4, 4)


def test_lc3d_cfg5_size_bf16_sampled(dev):
    """BASELINE config 5: [1, 96, 96, 96, 16] bf16, 3x3x3, 16 filters: 830 584 positions x 432 x 16 weights
    (11.5 GB bf16).  The CPU check covers 4096 sampled output positions exactly."""
    torch.manual_seed(7)
    x = torch.randn(1, 96, 96, 96, 16, device=dev, dtype=torch.bfloat16)
    layer = ne.layers.LocallyConnected3D(16, (3, 3, 3))
    y = layer(x)
    assert tuple(layer.kernel.shape) == (94 ** 3, 432, 16) and layer.kernel.dtype == torch.bfloat16
    with torch.no_grad():
        layer.kernel.normal_(0, 1.0 / np.sqrt(432))
        layer.bias.normal_(0, 0.1)
    y = layer(x)
    assert tuple(y.shape) == (1, 94, 94, 94, 16)
    rng = np.random.default_rng(0)
    pos = np.unique(np.concatenate([rng.integers(0, 94 ** 3, 4090), [0, 1, 93, 94 * 94, 94 ** 3 - 1, 94 ** 3 - 94]]))
    kk = layer.kernel.detach()[torch.from_numpy(pos).to(dev)].float().cpu().numpy().astype(np.float64)
    bb = layer.bias.detach().reshape(-1, 16)[torch.from_numpy(pos).to(dev)].float().cpu().numpy().astype(np.float64)
    xh = x[0].float().cpu().numpy().astype(np.float64)
    yh = N(y)[0].reshape(-1, 16)
    r, c, z = pos // (94 * 94), (pos // 94) % 94, pos % 94
    ref = np.empty((len(pos), 16))
    for i in range(len(pos)):
        patch = xh[r[i]:r[i] + 3, c[i]:c[i] + 3, z[i]:z[i] + 3].reshape(-1)
        ref[i] = patch @ kk[i] + bb[i]
    close(yh[pos], ref, 2.0 ** -8)
    # the rest of the config: softmax + label-weighted CCE on the bf16 output (from_logits form)
    lab = torch.randint(0, 16, (1, 94, 94, 94), device=dev)
    t = torch.nn.functional.one_hot(lab, 16).to(torch.bfloat16)
    w = np.linspace(0.5, 1.5, 16).astype(F)
    loss = float(ne.metrics.WeightedCategoricalCrossentropy(label_weights=w, from_logits=True)(t, y.detach()))
    from oracle import c_oracle as co
    want = co.wcce(N(t), N(y), w, from_logits=True)
    np.testing.assert_allclose(loss, want, rtol=1e-4)


@pytest.mark.parametrize('dtype,cout,ks,strides,act,S,cin', [
    ('float32', 8, (3, 3, 3), (1, 1, 1), None, (7, 6, 8), 4),
    ('float32', 4, (3, 3, 3), (1, 1, 1), 'elu', (6, 6, 6), 3),
    ('float32', 16, (2, 3, 2), (2, 1, 1), 'relu', (8, 7, 6), 5),
    ('bfloat16', 16, (3, 3, 3), (1, 1, 1), 'elu', (8, 8, 8), 16),
    ('bfloat16', 8, (3, 3, 3), (1, 1, 1), None, (6, 7, 6), 8),
])
def test_lc3d_backward(dev, dtype, cout, ks, strides, act, S, cin):
    """grad wrt kernel, bias and input vs the float64 autograd oracle (bf16: tolerance of the bf16 rounding)"""
    from oracle import grad_oracle as go
    rng = np.random.default_rng(91 + cout)
    tdt = getattr(torch, dtype)
    B = 2
    layer = ne.layers.LocallyConnected3D(cout, ks, strides=strides, activation=act)
    x = torch.from_numpy(rng.standard_normal((B,) + S + (cin,)).astype(np.float32)).to(dev).to(tdt)
    y0 = layer(x)                                                            # builds the weights
    with torch.no_grad():
        layer.kernel.copy_(torch.from_numpy((rng.standard_normal(tuple(layer.kernel.shape)) * 0.2).astype(np.float32)).to(tdt))
        layer.bias.copy_(torch.from_numpy((rng.standard_normal(tuple(layer.bias.shape)) * 0.1).astype(np.float32)).to(tdt))
    w = torch.from_numpy(rng.standard_normal(tuple(y0.shape)).astype(np.float32)).to(dev).to(tdt)
    xg = x.clone().requires_grad_()
    y = layer(xg)
    (y.float() * w.float()).sum().backward()
    xo = x.detach().cpu().double().requires_grad_()
    ko = layer.kernel.detach().cpu().double().requires_grad_()
    bo = layer.bias.detach().cpu().double().requires_grad_()
    yo = go.lc3d(xo, ko, bo, ks, strides, act)
    (yo * w.detach().cpu().double()).sum().backward()
    tol = 2e-5 if dtype == 'float32' else 2e-2

    def close(got, want, what):
        got = got.detach().cpu().double().numpy(); want = want.numpy()
        err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        assert got.shape == want.shape and err < tol, (what, err)

    close(y, yo.detach(), 'forward')
    close(layer.kernel.grad, ko.grad, 'grad_kernel')
    close(layer.bias.grad, bo.grad, 'grad_bias')
    close(xg.grad, xo.grad, 'grad_x')
    assert layer.kernel.grad.dtype == tdt and xg.grad.dtype == tdt


# ----------------------------------------------------------------------------------------------------------------------
# implementations 2 (dense-masked kernel) and 3 (sparse COO values), 'same' padding, channels_first
# (neurite/tf/layers.py:986-1028, 1260-1343): the same un-shared weights in other layouts, re-laid out once into the
# streaming layout of the HIP kernel.  Fixtures: the reference LAYER (build + call) run by tests/golden/make_golden.py.
# ----------------------------------------------------------------------------------------------------------------------

def _impl_cases():
    return golden_cases(load_golden('lc3d_impl'))


def _impl_layer(dev, c):
    ks, st = tuple(int(v) for v in c['ks']), tuple(int(v) for v in c['strides'])
    act = str(c['activation'])
    layer = ne.layers.LocallyConnected3D(int(c['filters']), ks, strides=st, padding=str(c['padding']),
                                         data_format=str(c['data_format']), activation=None if act == 'linear' else act,
                                         implementation=int(c['implementation']))
    x = G(c['x'], dev)
    y0 = layer(x)
    assert tuple(layer.kernel.shape) == c['kernel'].shape, (tuple(layer.kernel.shape), c['kernel'].shape)
    assert tuple(layer.bias.shape) == c['bias'].shape
    assert tuple(y0.shape) == c['out'].shape == tuple(layer.compute_output_shape(tuple(x.shape)))
    with torch.no_grad():
        layer.kernel.copy_(G(c['kernel'], dev))
        layer.bias.copy_(G(c['bias'], dev))
    return layer, x


def test_lc3d_implementations_golden(dev):
    cases = _impl_cases()
    assert len(cases) == 10
    for tag, c in cases.items():
        layer, x = _impl_layer(dev, c)
        close(N(layer(x)), c['out'], 1e-5)
        if int(c['implementation']) == 3:                 # the sparse values are ordered like the reference's kernel_idxs
            assert np.array_equal(np.stack(layer._plan['pairs'], 1), c['kernel_idxs']), tag
        # cached re-layout: a second call, and a call after an in-place weight update, stay correct
        close(N(layer(x)), c['out'], 1e-5)
        with torch.no_grad():
            layer.kernel.mul_(2.0)
            layer.bias.mul_(2.0)
        if str(c['activation']) == 'linear':
            close(N(layer(x)), 2.0 * c['out'].astype(np.float64), 1e-5)


@pytest.mark.parametrize('tag', ['i1_cf', 'i2_valid', 'i2_same', 'i2_cf_same', 'i3_valid', 'i3_same', 'i3_cf_same'])
def test_lc3d_implementations_backward(dev, tag):
    """gradients wrt the layer's OWN kernel layout, the bias and the input against float64 autograd of the reference's
    formulation (dense kernel x connection mask matmul for implementation 2, sparse matrix for 3; oracle/grad_oracle.py)"""
    from oracle import grad_oracle as go
    c = _impl_cases()[tag]
    layer, x = _impl_layer(dev, c)
    impl, pad, fmt = int(c['implementation']), str(c['padding']), str(c['data_format'])
    act = None if str(c['activation']) == 'linear' else str(c['activation'])
    ks, st = tuple(int(v) for v in c['ks']), tuple(int(v) for v in c['strides'])
    rng = np.random.default_rng(5)
    w = rng.standard_normal(c['out'].shape)
    xg = x.clone().requires_grad_()
    y = layer(xg)
    (y.double() * G(w, dev)).sum().backward()
    xo = torch.from_numpy(c['x']).double().requires_grad_()
    ko = torch.from_numpy(c['kernel']).double().requires_grad_()
    bo = torch.from_numpy(c['bias']).double().requires_grad_()
    cf = fmt == 'channels_first'
    ins = tuple(c['x'].shape[2:]) if cf else tuple(c['x'].shape[1:4])
    outs = tuple(c['out'].shape[2:]) if cf else tuple(c['out'].shape[1:4])
    if impl == 1:
        xcl = xo.permute(0, 2, 3, 4, 1) if cf else xo
        T = ks[0] * ks[1] * ks[2]
        cin = xcl.shape[-1]
        k1 = ko.reshape(ko.shape[0], cin, T, -1).permute(0, 2, 1, 3).reshape(ko.shape) if cf else ko     # (cin, taps) -> (taps, cin)
        b1 = bo.reshape((bo.shape[-1],) + outs).permute(1, 2, 3, 0) if cf else bo
        yo = go.lc3d(xcl, k1, b1, ks, st, act)
        yo = yo.permute(0, 4, 1, 2, 3) if cf else yo
    elif impl == 2:
        yo = go.lc3d_dense_masked(xo, ko, go.lc3d_connection_mask(ins, ks, st, pad, outs), bo, act, fmt)
    else:
        dense_shape = (int(np.prod(c['out'].shape[1:])), int(np.prod(c['x'].shape[1:])))
        yo = go.lc3d_sparse(xo, ko, c['kernel_idxs'], dense_shape, c['out'].shape[1:], bo, act, fmt)
    np.testing.assert_allclose(yo.detach().numpy(), c['out'], rtol=1e-5, atol=1e-5 * np.abs(c['out']).max())   # the oracle itself
    (yo * torch.from_numpy(w)).sum().backward()

    def chk(got, want, what):
        got = got.detach().cpu().double().numpy(); want = want.numpy()
        err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        assert got.shape == want.shape and err < 2e-5, (tag, what, err)

    chk(y, yo.detach(), 'forward')
    chk(layer.kernel.grad, ko.grad, 'grad_kernel')
    chk(layer.bias.grad, bo.grad, 'grad_bias')
    chk(xg.grad, xo.grad, 'grad_x')
