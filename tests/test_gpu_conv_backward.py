"""
GPU parity tests of the conv-stack backward (csrc/conv_bwd.hip, autograd wrappers in neurite_amd/models.py) against
torch CPU float64 autograd of the same Keras semantics (oracle/torch_unet_oracle.py).  Tolerance 2e-4 of the gradient
scale (float32 MFMA accumulation over up to ~10^4 voxels, float atomics across blocks).
"""

import contextlib
import io

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import _lib
from neurite_amd import models as M
from oracle import torch_unet_oracle as tuo

pytestmark = pytest.mark.gpu
F = np.float32
TOL = 2e-4


def G(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_() if grad else t


def N(t):
    return t.detach().cpu().numpy()


def close(got, want, what='', tol=TOL):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max()) / scale
    assert err < tol, '%s: max err / scale = %.3g' % (what, err)


@pytest.mark.parametrize('cin,cout,k,dil,act,shape', [
    (16, 16, 3, 1, 'elu', (9, 10, 17)),       # one 16-block each, ragged tiles
    (16, 16, 3, 1, 'elu', (48, 44, 12)),      # many tiles per persistent block
    (48, 16, 3, 1, 'elu', (8, 8, 16)),        # dec1-like: three cin blocks
    (96, 32, 3, 1, 'elu', (8, 4, 8)),         # dec0-like: two cin chunks of 48, two cout blocks
    (32, 64, 3, 1, 'relu', (4, 8, 8)),        # enc2-like: two cout chunks
    (1, 16, 3, 1, 'elu', (12, 9, 10)),        # first layer: cin = 1
    (8, 3, 3, 1, None, (7, 8, 9)),            # channel counts that are not multiples of 16 / 4
    (20, 24, 3, 2, 'elu', (10, 9, 12)),       # dilation 2
    (16, 5, 1, 1, None, (6, 7, 8)),           # 1x1x1 (likelihood)
    (16, 16, (1, 3, 3), 1, 'elu', (1, 12, 13)),   # a lifted 2-D conv
])
def test_conv_backward(dev, cin, cout, k, dil, act, shape):
    rng = np.random.default_rng(cin * 100 + cout)
    ks = (k,) * 3 if isinstance(k, int) else tuple(k)
    B = 2
    conv = M._Conv('c', cin, cout, ks, dilation=dil, padding='same', activation=act).to(dev)
    kern = (rng.standard_normal(ks + (cin, cout)) * 0.2).astype(F)
    bias = (rng.standard_normal(cout) * 0.1).astype(F)
    with torch.no_grad():
        conv.kernel.copy_(G(kern, dev)); conv.bias.copy_(G(bias, dev))
    x = rng.standard_normal((B,) + shape + (cin,)).astype(F)
    w = rng.standard_normal((B,) + shape + (cout,)).astype(F)
    xg = G(x, dev, True)
    y = conv(xg)
    (y * G(w, dev)).sum().backward()
    xo = torch.from_numpy(x).double().requires_grad_()
    ko = torch.from_numpy(kern).double().requires_grad_()
    bo = torch.from_numpy(bias).double().requires_grad_()
    yo = tuo.conv3d_same(xo, ko, bo, dil, act)
    (yo * torch.from_numpy(w).double()).sum().backward()
    close(N(y), yo.detach().numpy(), 'forward', 1e-4)
    close(N(conv.kernel.grad), ko.grad.numpy(), 'grad_kernel')
    close(N(conv.bias.grad), bo.grad.numpy(), 'grad_bias')
    close(N(xg.grad), xo.grad.numpy(), 'grad_x')


@pytest.mark.parametrize('cin,cout,k,dil,act,shape', [
    (16, 16, 3, 1, 'elu', (9, 10, 17)), (8, 12, 3, 2, 'relu', (11, 9, 13)), (4, 8, (1, 3, 3), 1, None, (1, 9, 12)),
    (1, 16, 3, 1, 'elu', (8, 9, 10)),
])
def test_conv_backward_valid_padding(dev, cin, cout, k, dil, act, shape):
    """padding='valid' (conv_enc / conv_dec pass `padding` to every Conv, neurite/tf/models.py:1345): forward and gradients against
    float64 autograd.  The backward embeds the output gradient in zeros at the positions a 'same' convolution would also compute
    and runs the 'same' backward kernels."""
    rng = np.random.default_rng(cin * 10 + cout + dil)
    ks = (k,) * 3 if isinstance(k, int) else tuple(k)
    B = 2
    conv = M._Conv('c', cin, cout, ks, dilation=dil, padding='valid', activation=act).to(dev)
    kern = (rng.standard_normal(ks + (cin, cout)) * 0.2).astype(F)
    bias = (rng.standard_normal(cout) * 0.1).astype(F)
    with torch.no_grad():
        conv.kernel.copy_(G(kern, dev)); conv.bias.copy_(G(bias, dev))
    x = rng.standard_normal((B,) + shape + (cin,)).astype(F)
    oshape = tuple(shape[d] - (ks[d] - 1) * dil for d in range(3))
    w = rng.standard_normal((B,) + oshape + (cout,)).astype(F)
    xg = G(x, dev, True)
    y = conv(xg)
    assert tuple(y.shape) == (B,) + oshape + (cout,)
    (y * G(w, dev)).sum().backward()
    xo = torch.from_numpy(x).double().requires_grad_()
    ko = torch.from_numpy(kern).double().requires_grad_()
    bo = torch.from_numpy(bias).double().requires_grad_()
    yo = tuo.conv3d_same(xo, ko, bo, dil, act, padding='valid')
    (yo * torch.from_numpy(w).double()).sum().backward()
    close(N(y), yo.detach().numpy(), 'forward', 1e-4)
    close(N(conv.kernel.grad), ko.grad.numpy(), 'grad_kernel')
    close(N(conv.bias.grad), bo.grad.numpy(), 'grad_bias')
    close(N(xg.grad), xo.grad.numpy(), 'grad_x')


@pytest.mark.parametrize('folded,S,c0,c1,cout', [
    (True, (8, 12, 8), 16, 32, 16),         # dec1-like: the up-sampled channels are differentiated on the low-resolution grid
    (False, (8, 12, 8), 16, 32, 16),        # the 27-tap form of the same layer
    (True, (8, 4, 16), 32, 64, 32),         # dec0-like: two cout blocks per parity group, 64 low-resolution channels
    (True, (12, 6, 20), 16, 16, 16),        # ragged tiles on the low-resolution grid
    (True, (4, 4, 4), 8, 12, 5),            # channel counts outside the folded form: falls back to the 27-tap form
])
def test_conv_backward_fused_upsample_concat_loader(dev, folded, S, c0, c1, cout):
    """decoder conv: input = concat(skip, upsample(lo)) never materialised in the forward; grads to both sources"""
    rng = np.random.default_rng(5)
    B, up = 2, (2, 2, 2)
    conv = M._Conv('c', c0 + c1, cout, (3, 3, 3), activation='elu').to(dev)
    conv.fold_backward = folded
    kern = (rng.standard_normal((3, 3, 3, c0 + c1, cout)) * 0.1).astype(F)
    with torch.no_grad():
        conv.kernel.copy_(G(kern, dev))
    skip = rng.standard_normal((B,) + S + (c0,)).astype(F)
    lo = rng.standard_normal((B,) + tuple(s // 2 for s in S) + (c1,)).astype(F)
    w = rng.standard_normal((B,) + S + (cout,)).astype(F)
    sg, lg = G(skip, dev, True), G(lo, dev, True)
    y = conv(sg, lo=lg, up=up)
    (y * G(w, dev)).sum().backward()
    so, lo_o = torch.from_numpy(skip).double().requires_grad_(), torch.from_numpy(lo).double().requires_grad_()
    ko = torch.from_numpy(kern).double().requires_grad_()
    yo = tuo.conv3d_same(torch.cat([so, tuo.upsample(lo_o, up)], -1), ko, torch.zeros(cout, dtype=torch.float64), 1, 'elu')
    (yo * torch.from_numpy(w).double()).sum().backward()
    close(N(sg.grad), so.grad.numpy(), 'grad_skip')
    close(N(lg.grad), lo_o.grad.numpy(), 'grad_lo')
    close(N(conv.kernel.grad), ko.grad.numpy(), 'grad_kernel')


def test_small_layer_backward(dev):
    rng = np.random.default_rng(6)
    # max pooling: even and odd sizes (SAME keeps partial windows)
    for S in ((8, 6, 10), (7, 5, 9)):
        x = rng.standard_normal((2,) + S + (5,)).astype(F)
        xg = G(x, dev, True)
        y = M._MaxPoolFn.apply(xg, (2, 2, 2), 'same')
        w = rng.standard_normal(tuple(y.shape)).astype(F)
        (y * G(w, dev)).sum().backward()
        xo = torch.from_numpy(x).double().requires_grad_()
        yo = tuo.maxpool_same(xo, (2, 2, 2))
        (yo * torch.from_numpy(w).double()).sum().backward()
        close(N(y), yo.detach().numpy(), 'pool fwd', 1e-6)
        close(N(xg.grad), xo.grad.numpy(), 'pool grad', 1e-6)
    # merge = concat(skip, upsample(lo)) and plain up-sampling
    skip = rng.standard_normal((2, 6, 4, 8, 3)).astype(F)
    lo = rng.standard_normal((2, 3, 2, 4, 5)).astype(F)
    sg, lg = G(skip, dev, True), G(lo, dev, True)
    y = M._MergeFn.apply(sg, lg, (2, 2, 2))
    w = rng.standard_normal(tuple(y.shape)).astype(F)
    (y * G(w, dev)).sum().backward()
    so, lo_o = torch.from_numpy(skip).double().requires_grad_(), torch.from_numpy(lo).double().requires_grad_()
    (torch.cat([so, tuo.upsample(lo_o, (2, 2, 2))], -1) * torch.from_numpy(w).double()).sum().backward()
    close(N(sg.grad), so.grad.numpy(), 'merge skip', 1e-6)
    close(N(lg.grad), lo_o.grad.numpy(), 'merge lo', 1e-5)
    lg2 = G(lo, dev, True)
    M._MergeFn.apply(None, lg2, (2, 1, 2)).square().sum().backward()
    lo2 = torch.from_numpy(lo).double().requires_grad_()
    tuo.upsample(lo2, (2, 1, 2)).square().sum().backward()
    close(N(lg2.grad), lo2.grad.numpy(), 'upsample', 1e-5)
    # softmax over channels (vector and scalar paths)
    for C in (8, 5):
        z = rng.standard_normal((2, 5, 6, 7, C)).astype(F)
        w = rng.standard_normal(z.shape).astype(F)
        zg = G(z, dev, True)
        (M._SoftmaxFn.apply(zg) * G(w, dev)).sum().backward()
        zo = torch.from_numpy(z).double().requires_grad_()
        (torch.softmax(zo, -1) * torch.from_numpy(w).double()).sum().backward()
        close(N(zg.grad), zo.grad.numpy(), 'softmax C=%d' % C, 1e-5)


@pytest.mark.parametrize('kw,ishape', [
    (dict(nb_features=8, nb_levels=3, conv_size=3, nb_labels=4, feat_mult=2), (16, 8, 16, 1)),
    (dict(nb_features=8, nb_levels=2, conv_size=3, nb_labels=3, nb_conv_per_level=2), (8, 12, 8, 2)),
    (dict(nb_features=8, nb_levels=2, conv_size=3, nb_labels=3, nb_conv_per_level=2, use_residuals=True, feat_mult=2), (8, 8, 8, 2)),
])
def test_unet_training_step_gradients(dev, kw, ishape):
    """model.train(): loss = weighted CCE(one-hot, unet(x)) - mean Dice; every parameter gradient vs the float64 oracle"""
    rng = np.random.default_rng(41)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(input_shape=ishape, **kw).to(dev)
    for m in net.layers_by_name.values():
        with torch.no_grad():
            m.kernel.copy_(G((rng.standard_normal(tuple(m.kernel.shape)) * 0.2).astype(F), dev))
            m.bias.copy_(G((rng.standard_normal(tuple(m.bias.shape)) * 0.1).astype(F), dev))
    B, L = 2, kw['nb_labels']
    x = rng.standard_normal((B,) + ishape).astype(F)
    lab = rng.integers(0, L, (B,) + ishape[:-1])
    t = np.eye(L, dtype=F)[lab]
    wl = rng.uniform(0.5, 2, L).astype(F)
    assert not net.training                       # Keras predict semantics by default
    y_eval = net(G(x, dev))
    net.train()
    y = net(G(x, dev))
    close(N(y), N(y_eval), 'train-mode forward == eval forward', 1e-5)
    loss = ne.losses.CategoricalCrossentropy(wl)(G(t, dev), y) - ne.metrics.Dice(check_input_limits=False).mean_dice(G(t, dev), y)
    loss.backward()
    # oracle
    params = {k: (m.kernel.detach().cpu().double().requires_grad_(), m.bias.detach().cpu().double().requires_grad_())
              for k, m in net.layers_by_name.items()}
    yo = tuo.forward(net, torch.from_numpy(x).double(), params)
    from oracle import grad_oracle as go
    to = torch.from_numpy(t).double()
    lo = go.cce_per_voxel(to, yo, torch.from_numpy(wl).double()).mean() - go.soft_dice(to, yo).mean()
    lo.backward()
    close(float(loss.detach()), float(lo.detach()), 'loss', 1e-4)
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), params[k][0].grad.numpy(), k + ' kernel', 5e-4)
        close(N(m.bias.grad), params[k][1].grad.numpy(), k + ' bias', 5e-4)
    # one SGD step on the device lowers the loss
    with torch.no_grad():
        for m in net.layers_by_name.values():
            m.kernel -= 2e-3 * m.kernel.grad
            m.bias -= 2e-3 * m.bias.grad
    y2 = net(G(x, dev))
    loss2 = ne.losses.CategoricalCrossentropy(wl)(G(t, dev), y2) - ne.metrics.Dice(check_input_limits=False).mean_dice(G(t, dev), y2)
    assert float(loss2.detach()) < float(loss.detach())
    net.eval()
    assert net(G(x, dev)).requires_grad is False


def test_unet_training_with_feature_dropout(dev):
    """conv_dropout > 0: Keras Dropout with noise_shape [None, 1, 1, 1, C] (models.py:1390-1399) in training mode; gradients vs
    the oracle run with the recorded masks; eval mode ignores it"""
    rng = np.random.default_rng(43)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(8, (8, 8, 16, 1), 2, 3, 3, conv_dropout=0.4).to(dev)
    for m in net.layers_by_name.values():
        with torch.no_grad():
            m.kernel.copy_(G((rng.standard_normal(tuple(m.kernel.shape)) * 0.2).astype(F), dev))
    x = rng.standard_normal((3, 8, 8, 16, 1)).astype(F)
    y_eval = net(G(x, dev))
    net.train()
    y = net(G(x, dev))
    scales = {k: v.detach().cpu().double() for k, v in net.last_dropout_scales.items()}
    assert scales and all(set(np.unique(v.numpy().round(4))) <= {0.0, round(1 / 0.6, 4)} for v in scales.values())
    assert not np.allclose(N(y), N(y_eval))
    w = rng.standard_normal(tuple(y.shape)).astype(F)
    (y * G(w, dev)).sum().backward()
    params = {k: (m.kernel.detach().cpu().double().requires_grad_(), m.bias.detach().cpu().double().requires_grad_())
              for k, m in net.layers_by_name.items()}
    yo = tuo.forward(net, torch.from_numpy(x).double(), params, dropout_scales=scales)
    (yo * torch.from_numpy(w).double()).sum().backward()
    close(N(y), yo.detach().numpy(), 'forward with dropout', 1e-4)
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), params[k][0].grad.numpy(), k + ' kernel', 5e-4)


def test_unet_training_with_batch_norm(dev):
    """batch_norm=-1: training-mode BatchNormalization (batch statistics, moving averages updated) forward and gradients"""
    rng = np.random.default_rng(47)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(8, (8, 8, 8, 2), 2, 3, 3, batch_norm=-1, nb_conv_per_level=2).to(dev)
    conv = {k: m for k, m in net.layers_by_name.items() if hasattr(m, 'kernel')}
    bns = {k: m for k, m in net.layers_by_name.items() if hasattr(m, 'gamma')}
    assert bns
    for m in conv.values():
        with torch.no_grad():
            m.kernel.copy_(G((rng.standard_normal(tuple(m.kernel.shape)) * 0.3).astype(F), dev))
            m.bias.copy_(G((rng.standard_normal(tuple(m.bias.shape)) * 0.2).astype(F), dev))
    for m in bns.values():
        with torch.no_grad():
            m.gamma.copy_(G((1 + 0.2 * rng.standard_normal(tuple(m.gamma.shape))).astype(F), dev))
            m.beta.copy_(G((0.2 * rng.standard_normal(tuple(m.beta.shape))).astype(F), dev))
    x = rng.standard_normal((3, 8, 8, 8, 2)).astype(F)
    net.train()
    y = net(G(x, dev))
    w = rng.standard_normal(tuple(y.shape)).astype(F)
    (y * G(w, dev)).sum().backward()
    params = {k: (m.kernel.detach().cpu().double().requires_grad_(), m.bias.detach().cpu().double().requires_grad_()) for k, m in conv.items()}
    params.update({k: (m.gamma.detach().cpu().double().requires_grad_(), m.beta.detach().cpu().double().requires_grad_()) for k, m in bns.items()})
    yo = tuo.forward(net, torch.from_numpy(x).double(), params, bn_train=True)
    (yo * torch.from_numpy(w).double()).sum().backward()
    close(N(y), yo.detach().numpy(), 'forward (batch statistics)', 2e-4)
    for k, m in conv.items():
        close(N(m.kernel.grad), params[k][0].grad.numpy(), k + ' kernel', 2e-3)
    for k, m in bns.items():
        close(N(m.gamma.grad), params[k][0].grad.numpy(), k + ' gamma', 2e-3)
        close(N(m.beta.grad), params[k][1].grad.numpy(), k + ' beta', 2e-3)
        assert float((m.moving_mean != 0).sum()) > 0 and float((m.moving_variance != 1).sum()) > 0      # moving statistics moved
    net.eval()
    y_eval = net(G(x, dev))
    yo_eval = tuo.forward(net, torch.from_numpy(x).double())
    close(N(y_eval), yo_eval.numpy(), 'eval uses the moving statistics', 2e-4)


def test_unet_add_prior_layer(dev):
    """unet(add_prior_layer=True) = models.add_prior (models.py:378-436, log-prior form): softmax(log prior + likelihood);
    inference and training-mode gradients (incl. wrt the prior input) vs the oracle"""
    rng = np.random.default_rng(53)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        net = ne.models.unet(8, (8, 8, 8, 1), 2, 3, 4, add_prior_layer=True).to(dev)
    assert net.name == 'unet_prior' and net.input_shapes == [(8, 8, 8, 1), (8, 8, 8, 4)]
    assert net.layer_names[-3:] == ['unet_prior-input', 'unet_prior_posterior', 'unet_prior_prediction']
    for m in net.layers_by_name.values():
        with torch.no_grad():
            m.kernel.copy_(G((rng.standard_normal(tuple(m.kernel.shape)) * 0.3).astype(F), dev))
    x = rng.standard_normal((2, 8, 8, 8, 1)).astype(F)
    prior = np.log(rng.dirichlet(np.ones(4), (2, 8, 8, 8))).astype(F)
    y = net([G(x, dev), G(prior, dev)])
    yo = tuo.forward(net, [torch.from_numpy(x).double(), torch.from_numpy(prior).double()])
    close(N(y), yo.numpy(), 'posterior', 1e-4)
    np.testing.assert_allclose(N(y).sum(-1), 1.0, rtol=1e-5)
    net.train()
    pg = G(prior, dev, True)
    yt = net([G(x, dev), pg])
    w = rng.standard_normal(tuple(yt.shape)).astype(F)
    (yt * G(w, dev)).sum().backward()
    params = {k: (m.kernel.detach().cpu().double().requires_grad_(), m.bias.detach().cpu().double().requires_grad_())
              for k, m in net.layers_by_name.items()}
    po = torch.from_numpy(prior).double().requires_grad_()
    (tuo.forward(net, [torch.from_numpy(x).double(), po], params) * torch.from_numpy(w).double()).sum().backward()
    close(N(pg.grad), po.grad.numpy(), 'grad wrt the log prior', 2e-4)
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), params[k][0].grad.numpy(), k + ' kernel', 5e-4)
    with pytest.raises(AssertionError, match='cannot do softmax'):               # models.py:423
        ne.models.unet(8, (8, 8, 8, 1), 2, 3, 4, add_prior_layer=True, use_logp=False)
    # the stand-alone builder gives the same network: models.add_prior(unet(..., final 'linear'), prior_shape)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        base = ne.models.unet(8, (8, 8, 8, 1), 2, 3, 4, final_pred_activation='linear').to(dev)
        alone = ne.models.add_prior(base, (8, 8, 8, 4), name='unet_prior')
    alone.set_weights(net.get_weights())
    net.eval()
    assert torch.equal(alone([G(x, dev), G(prior, dev)]), net([G(x, dev), G(prior, dev)]))


def test_unet_add_prior_layer_probability_form(dev):
    """add_prior with use_logp=False (models.py:408-417): prior * sigmoid(likelihood), linear prediction; inference and
    training-mode gradients vs the oracle"""
    rng = np.random.default_rng(54)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        net = ne.models.unet(8, (8, 8, 8, 1), 2, 3, 4, add_prior_layer=True, use_logp=False,
                             final_pred_activation='linear').to(dev)
    assert net.layer_names[-4:] == ['unet_prior-input', 'unet_prior_likelihood_sigmoid', 'unet_prior_posterior',
                                    'unet_prior_prediction']
    for m in net.layers_by_name.values():
        with torch.no_grad():
            m.kernel.copy_(G((rng.standard_normal(tuple(m.kernel.shape)) * 0.3).astype(F), dev))
    x = rng.standard_normal((2, 8, 8, 8, 1)).astype(F)
    prior = rng.dirichlet(np.ones(4), (2, 8, 8, 8)).astype(F)
    y = net([G(x, dev), G(prior, dev)])
    yo = tuo.forward(net, [torch.from_numpy(x).double(), torch.from_numpy(prior).double()])
    close(N(y), yo.numpy(), 'posterior', 1e-4)
    assert (N(y) >= 0).all() and (N(y) <= prior + 1e-7).all()                     # sigmoid in (0, 1) scales the prior down
    net.train()
    pg = G(prior, dev, True)
    yt = net([G(x, dev), pg])
    close(N(yt), yo.numpy(), 'training-mode forward', 1e-4)
    w = rng.standard_normal(tuple(yt.shape)).astype(F)
    (yt * G(w, dev)).sum().backward()
    params = {k: (m.kernel.detach().cpu().double().requires_grad_(), m.bias.detach().cpu().double().requires_grad_())
              for k, m in net.layers_by_name.items()}
    po = torch.from_numpy(prior).double().requires_grad_()
    (tuo.forward(net, [torch.from_numpy(x).double(), po], params) * torch.from_numpy(w).double()).sum().backward()
    close(N(pg.grad), po.grad.numpy(), 'grad wrt the prior', 2e-4)
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), params[k][0].grad.numpy(), k + ' kernel', 5e-4)
        close(N(m.bias.grad), params[k][1].grad.numpy(), k + ' bias', 5e-4)
