"""
GPU tests of the joint segmentation loss (csrc/segloss.hip; metrics.JointSegLoss behind losses.multiple_losses_decorator,
neurite/tf/losses.py:225-246): soft Dice (metrics.py:415-482) + label-weighted CCE (metrics.py:619-650) of one prediction in one pass,
and one pass back that also runs through the soft-max that made the prediction (models.py:1545-1555).

Tolerances: the Dice sums are BIT-EQUAL to the separate kernel's (same accumulation order); the CCE sum differs from the separate
kernel's only in the order block partials meet (1e-6 relative); gradients 1e-5 of their largest magnitude against float64.
"""

import contextlib
import io

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import _lib
from neurite_amd import metrics as MT
from neurite_amd import models as M
from oracle import grad_oracle as go
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def G(a, dev, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(grad)


def N(t):
    return t.detach().cpu().numpy()


def close(got, want, what, tol=1e-5):
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - want).max()) / scale
    assert err <= tol, '%s: max error %.3g of the largest magnitude (tolerance %.1g)' % (what, err, tol)


def maps(rng, shape, L, one_hot=False):
    z = rng.standard_normal(shape + (L,)).astype(F) * 2
    if one_hot:
        t = np.eye(L, dtype=F)[rng.integers(0, L, shape)]
    else:
        t = rng.random(shape + (L,)).astype(F)
        t /= t.sum(-1, keepdims=True)
    return z, t


@pytest.mark.parametrize('L,shape', [(4, (2, 9, 7, 5)), (8, (1, 33, 5, 4)), (32, (3, 12, 11, 10)), (64, (2, 6, 5, 7)), (256, (1, 5, 3, 4))])
@pytest.mark.parametrize('eps,ls', [(0., 0.), (0.1, 0.2)])
def test_forward_equals_the_separate_kernels(dev, L, shape, eps, ls):
    rng = np.random.default_rng(L + len(shape))
    z, t = maps(rng, shape, L)
    p = torch.softmax(G(z, dev), -1)
    tt = G(t, dev)
    w = rng.uniform(0.5, 2, L).astype(F)
    cce_sum, dice = MT._SegLossFn.apply(tt, p, G(w, dev), (eps, ls, True), None)
    sums, d_sep, mm = MT.dice_partial_sums(tt, p, False, eps)
    assert torch.equal(dice, d_sep)                                                  # same kernel arithmetic, same order
    sep = MT._WcceFn.apply(tt, p, G(w, dev), False, ls, False)
    np.testing.assert_allclose(N(cce_sum), N(sep), rtol=1e-6)
    # ... and the oracle (float64 accumulation)
    np.testing.assert_allclose(N(dice), npo.dice(t, N(p), laplace_smoothing=eps), rtol=1e-5, atol=1e-7)
    want = npo.cce_per_voxel(t, N(p), w, label_smoothing=ls).astype(np.float64).sum()
    np.testing.assert_allclose(float(cce_sum), want, rtol=1e-5)


def test_forward_range_check_and_unsupported(dev):
    rng = np.random.default_rng(3)
    z, t = maps(rng, (1, 6, 5, 4), 8)
    p = torch.softmax(G(z, dev), -1)
    bad = G(t, dev).clone()
    bad[0, 1, 2, 3, 4] = 1.5
    with pytest.raises(ne.errors.InvalidArgumentError):
        MT._SegLossFn.apply(bad, p, None, (0., 0., True), None)
    MT._SegLossFn.apply(bad, p, None, (0., 0., False), None)                         # check_input_limits=False: no assert
    lib = _lib.lib()
    assert [L for L in range(1, 70) if lib.nrt_seg_loss_supported(L)] == [4, 8, 16, 32, 64]
    assert lib.nrt_seg_loss_supported(256) and not lib.nrt_seg_loss_supported(512)


def _float64_loss(z64, t64, w64, eps, ls, a, gd, softmax):
    p = torch.softmax(z64, -1) if softmax else z64
    n = p.numel() // p.shape[-1]
    return a * go.cce_per_voxel(t64, p, w64, label_smoothing=ls).sum() / n + (gd * go.soft_dice(t64, p, eps)).sum()


@pytest.mark.parametrize('L,shape', [(4, (2, 7, 6, 5)), (32, (2, 9, 8, 7)), (128, (1, 4, 3, 5))])
@pytest.mark.parametrize('eps,ls', [(0., 0.), (0.1, 0.1)])
def test_backward_vs_float64(dev, L, shape, eps, ls):
    rng = np.random.default_rng(7 + L)
    z, t = maps(rng, shape, L, one_hot=(ls == 0.))
    w = rng.uniform(0.5, 2, L).astype(F)
    gd = rng.standard_normal((shape[0], L)).astype(F)
    a = 1.7
    B = shape[0]
    n = int(np.prod(shape))
    # (1) gradient wrt the probabilities (no producer stamp): y_pred is a leaf
    p0 = torch.softmax(G(z, dev), -1).detach().requires_grad_()
    cce_sum, dice = MT._SegLossFn.apply(G(t, dev), p0, G(w, dev), (eps, ls, False), None)
    (a * cce_sum[0] / n + (G(gd, dev) * dice).sum()).backward()
    p64 = torch.from_numpy(N(p0)).double().requires_grad_()
    _float64_loss(p64, torch.from_numpy(t).double(), torch.from_numpy(w).double(), eps, ls, a, torch.from_numpy(gd).double(), False).backward()
    close(N(p0.grad), p64.grad.numpy(), 'd / d y_pred')
    # (2) through the soft-max: the stamp of models._softmax_with_grad routes dz straight to the logits
    zt = G(z, dev, grad=True)
    y = M._softmax_with_grad(zt)
    src = y._nrt_softmax_src
    assert src.valid_for(y) and src.inputs[0] is zt
    cce_sum, dice = MT._SegLossFn.apply(G(t, dev), y.detach(), G(w, dev), (eps, ls, False), src, *src.inputs)
    (a * cce_sum[0] / n + (G(gd, dev) * dice).sum()).backward()
    z64 = torch.from_numpy(z).double().requires_grad_()
    _float64_loss(z64, torch.from_numpy(t).double(), torch.from_numpy(w).double(), eps, ls, a, torch.from_numpy(gd).double(), True).backward()
    close(N(zt.grad), z64.grad.numpy(), 'd / d logits')
    # (3) the separate chain (Dice bwd + CCE bwd + add + soft-max bwd) gives the same to float32 rounding
    zs = G(z, dev, grad=True)
    ys = M._SoftmaxFn.apply(zs)
    sep = a * MT._WcceFn.apply(G(t, dev), ys, G(w, dev), False, ls, False)[0] / n + \
        (G(gd, dev) * MT._SoftDiceFn.apply(G(t, dev), ys, eps, False, False)).sum()
    sep.backward()
    close(N(zt.grad), N(zs.grad).astype(np.float64), 'joint vs separate chain', 2e-6)
    assert B == dice.shape[0]


def _small_unet(dev, rng, L=4):
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(8, (16, 8, 16, 1), 2, 3, L, feat_mult=2).to(dev)
    for m in net.layers_by_name.values():
        with torch.no_grad():
            m.kernel.copy_(G((rng.standard_normal(tuple(m.kernel.shape)) * 0.2).astype(F), dev))
            m.bias.copy_(G((rng.standard_normal(tuple(m.bias.shape)) * 0.1).astype(F), dev))
    net.train()
    return net


def test_decorator_pairs_dice_and_cce_through_the_unet_head(dev):
    """losses.multiple_losses_decorator([cce.loss, dice.mean_loss]) on a unet: same loss value and the same gradient of every
    parameter as the two losses evaluated on their own, with the joint kernels doing the work"""
    rng = np.random.default_rng(5)
    L, B = 4, 2
    net = _small_unet(dev, rng, L)
    x = G(rng.standard_normal((B, 16, 8, 16, 1)).astype(F), dev)
    t = G(np.eye(L, dtype=F)[rng.integers(0, L, (B, 16, 8, 16))], dev)
    wl = rng.uniform(0.5, 2, L).astype(F)
    cce = ne.losses.CategoricalCrossentropy(wl)
    dice = ne.losses.Dice(weights=np.ones((1, L), F) * 0.5, laplace_smoothing=0.01)
    joint = ne.losses.multiple_losses_decorator([cce.loss, dice.mean_loss], [1.0, 2.0])

    y = net(x)
    separate = 1.0 * cce.loss(t, y) + 2.0 * dice.mean_loss(t, y)
    separate.backward()
    want = {k: (m.kernel.grad.clone(), m.bias.grad.clone()) for k, m in net.layers_by_name.items()}
    net.zero_grad()

    before = (MT.JointSegLoss.applications, MT.JointSegLoss.through_softmax)
    y = net(x)
    assert isinstance(getattr(y, '_nrt_softmax_src', None), M.SoftmaxSource)
    total = joint(t, y)
    assert (MT.JointSegLoss.applications, MT.JointSegLoss.through_softmax) == (before[0] + 1, before[1] + 1)
    np.testing.assert_allclose(float(total), float(separate), rtol=2e-6)
    total.backward()
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), N(want[k][0]).astype(np.float64), k + ' kernel', 2e-5)
        close(N(m.bias.grad), N(want[k][1]).astype(np.float64), k + ' bias', 2e-5)

    # a second consumer of the same prediction keeps its own gradient path (the stamp only by-passes the loss pair)
    net.zero_grad()
    y = net(x)
    extra = G(rng.standard_normal(tuple(y.shape)).astype(F), dev)
    (joint(t, y) + (y * extra).sum()).backward()
    got = {k: m.kernel.grad.clone() for k, m in net.layers_by_name.items()}
    net.zero_grad()
    y = net(x)
    (1.0 * cce.loss(t, y) + 2.0 * dice.mean_loss(t, y) + (y * extra).sum()).backward()
    for k, m in net.layers_by_name.items():
        close(N(got[k]), N(m.kernel.grad).astype(np.float64), k + ' kernel, two consumers', 2e-5)

    # evaluation without gradients takes the joint forward too
    with torch.no_grad():
        before = MT.JointSegLoss.applications
        v = joint(t, net(x))
        assert MT.JointSegLoss.applications == before + 1
    np.testing.assert_allclose(float(v), float(separate), rtol=2e-6)


def test_observers_of_the_prediction_keep_their_gradient(dev):
    """the through-soft-max shortcut replaces the prediction's own autograd node, so it must step aside when somebody
    watches the gradient AT the prediction -- a hook, retain_grad -- (they used to receive nothing) and must refuse inputs that were
    modified in place after the forward pass (the closure holds them outside save_for_backward)."""
    rng = np.random.default_rng(11)
    L, B = 4, 1
    net = _small_unet(dev, rng, L)
    x = G(rng.standard_normal((B, 16, 8, 16, 1)).astype(F), dev)
    t = G(np.eye(L, dtype=F)[rng.integers(0, L, (B, 16, 8, 16))], dev)
    cce = ne.losses.CategoricalCrossentropy(rng.uniform(0.5, 2, L).astype(F))
    dice = ne.losses.Dice(laplace_smoothing=0.01)
    joint = ne.losses.multiple_losses_decorator([cce.loss, dice.mean_loss], [1.0, 2.0])

    y = net(x)
    (1.0 * cce.loss(t, y) + 2.0 * dice.mean_loss(t, y)).backward()
    want = {k: m.kernel.grad.clone() for k, m in net.layers_by_name.items()}
    net.zero_grad()

    # a hook on the prediction: called with the gradient wrt the probabilities; the joint FORWARD kernel still runs
    seen = []
    y = net(x)
    y.register_hook(lambda g: seen.append(g.clone()))
    before = (MT.JointSegLoss.applications, MT.JointSegLoss.through_softmax)
    joint(t, y).backward()
    assert (MT.JointSegLoss.applications, MT.JointSegLoss.through_softmax) == (before[0] + 1, before[1])
    assert len(seen) == 1 and seen[0].shape == y.shape and float(seen[0].abs().max()) > 0
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), N(want[k]).astype(np.float64), k + ' kernel (hook)', 2e-5)
    net.zero_grad()

    # retain_grad
    y = net(x)
    y.retain_grad()
    joint(t, y).backward()
    np.testing.assert_allclose(N(y.grad), N(seen[0]), rtol=1e-6, atol=1e-9)
    for k, m in net.layers_by_name.items():
        close(N(m.kernel.grad), N(want[k]).astype(np.float64), k + ' kernel (retain_grad)', 2e-5)
    net.zero_grad()
    # autograd.grad(loss, y): asked AFTER the loss was formed, so the shortcut was taken and the prediction is not part of the graph --
    # that is refused loudly by autograd itself ("not have been used in the graph"), never answered with zeros
    y = net(x)
    loss = joint(t, y)
    with pytest.raises(RuntimeError):
        torch.autograd.grad(loss, y)
    net.zero_grad()

    # an input of the head modified in place between forward and backward: refused, not silently differentiated
    y = net(x)
    src = y._nrt_softmax_src
    loss = joint(t, y)
    with torch.no_grad():
        src.inputs[1].mul_(1.0)                                  # the head's kernel: same values, new version
    with pytest.raises(RuntimeError, match='modified in place'):
        loss.backward()
    net.zero_grad()


def test_decorator_leaves_other_cases_alone(dev):
    rng = np.random.default_rng(9)
    before = MT.JointSegLoss.applications
    # 5 labels: no joint kernel -> the two losses run on their own, same numbers
    z, t = maps(rng, (2, 6, 5, 4), 5, one_hot=True)
    p = torch.softmax(G(z, dev), -1)
    cce, dice = ne.losses.CategoricalCrossentropy(), ne.losses.Dice()
    f = ne.losses.multiple_losses_decorator([cce.loss, dice.mean_loss])
    assert float(f(G(t, dev), p)) == float(cce.loss(G(t, dev), p) + dice.mean_loss(G(t, dev), p))
    # hard Dice, normalised Dice, logits CCE, per-element CCE: not the pair
    z, t = maps(rng, (2, 6, 5, 4), 8, one_hot=True)
    p = torch.softmax(G(z, dev), -1)
    for c, d in ((ne.losses.CategoricalCrossentropy(), ne.losses.SoftDice(normalize=True)),
                 (ne.losses.CategoricalCrossentropy(from_logits=True), ne.losses.Dice()),
                 (ne.losses.CategoricalCrossentropy(), ne.losses.HardDice(8, input_type='prob'))):
        f = ne.losses.multiple_losses_decorator([c.loss, d.mean_loss])
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            assert float(f(G(t, dev), p)) == float(c.loss(G(t, dev), p) + d.mean_loss(G(t, dev), p))
    assert MT.JointSegLoss.applications == before
    # the [B, L] Dice loss with explicit weights, three losses in the list: the pair is still found
    c, d = ne.losses.CategoricalCrossentropy(reduction='sum'), ne.losses.Dice()
    f = ne.losses.multiple_losses_decorator([c.loss, lambda a, b: (a - b).abs().mean(), lambda a, b: d.loss(a, b).sum()], [0.5, 1.0, 1.0])
    tt = G(t, dev)
    got = f(tt, p)                       # the lambda hides d from the decorator: no pair, plain evaluation
    assert MT.JointSegLoss.applications == before
    f2 = ne.losses.multiple_losses_decorator([c.loss, lambda a, b: (a - b).abs().mean(), d.mean_loss], [0.5, 1.0, 1.0])
    got2 = f2(tt, p)
    assert MT.JointSegLoss.applications == before + 1
    np.testing.assert_allclose(float(got2), float(0.5 * c.loss(tt, p) + (tt - p).abs().mean() + d.mean_loss(tt, p)), rtol=2e-6)
    assert np.isfinite(float(got))


def test_training_step_captures_into_one_hipgraph(dev):
    """forward + joint loss + backward + SGD of a unet as ONE hipGraph launch (no host<->device traffic inside the step): replays
    walk the same parameter trajectory as eager steps"""
    rng = np.random.default_rng(11)
    L, B = 4, 2
    x = G(rng.standard_normal((B, 16, 8, 16, 1)).astype(F), dev)
    t = G(np.eye(L, dtype=F)[rng.integers(0, L, (B, 16, 8, 16))], dev)
    cce, dice = ne.losses.CategoricalCrossentropy(), ne.losses.Dice(check_input_limits=False)
    seg = ne.losses.multiple_losses_decorator([cce.loss, dice.loss])
    lr = 5e-3

    def make():
        net = _small_unet(dev, np.random.default_rng(12), L)
        return net, list(net.parameters())

    def step(net, params):
        for p in params:
            p.grad = None
        loss = seg(t, net(x)).mean()
        loss.backward()
        with torch.no_grad():
            torch._foreach_add_(params, [p.grad for p in params], alpha=-lr)
        return loss

    net_e, par_e = make()
    eager = [float(step(net_e, par_e)) for _ in range(4)]
    assert eager[-1] < eager[0]

    net_g, par_g = make()
    start = [p.detach().clone() for p in par_g]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(net_g, par_g)                                   # warm-up outside the capture: one-time uploads, workspace growth
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.no_grad():
        for p, s in zip(par_g, start):
            p.copy_(s)
    for p in par_g:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step(net_g, par_g)
    replayed = []
    for _ in range(4):
        graph.replay()
        replayed.append(float(loss))
    np.testing.assert_allclose(replayed, eager, rtol=1e-4)
    for a, b in zip(par_e, par_g):
        close(N(b), N(a).astype(np.float64), 'parameters after 4 steps', 1e-4)


def test_joint_loss_full_size_equals_the_separate_kernels(dev):
    """BASELINE config 3 output size (1 x 160^3 x 32): every block of the joint kernels walks its whole voxel range (2048 blocks x 2000
    voxels); forward and the gradient through a soft-max against the separate kernels"""
    torch.manual_seed(3)
    z = torch.randn(1, 160, 160, 160, 32, device=dev)
    t = torch.nn.functional.one_hot(torch.randint(0, 32, (1, 160, 160, 160), device=dev), 32).float()
    w = torch.rand(32, device=dev) + 0.5
    gd = torch.randn(1, 32, device=dev)
    n = 160 ** 3
    zj = z.clone().requires_grad_()
    y = M._softmax_with_grad(zj)
    src = y._nrt_softmax_src
    cce_sum, dice = MT._SegLossFn.apply(t, y.detach(), w, (0.0, 0.0, False), src, *src.inputs)
    (1.3 * cce_sum[0] / n + (gd * dice).sum()).backward()
    zs = z.clone().requires_grad_()
    ys = M._SoftmaxFn.apply(zs)
    cce_sep = MT._WcceFn.apply(t, ys, w, False, 0.0, False)
    dice_sep = MT._SoftDiceFn.apply(t, ys, 0.0, False, False)
    (1.3 * cce_sep[0] / n + (gd * dice_sep).sum()).backward()
    assert torch.equal(dice, dice_sep)
    np.testing.assert_allclose(N(cce_sum), N(cce_sep), rtol=1e-6)
    close(N(zj.grad), N(zs.grad).astype(np.float64), 'd / d logits at full size', 2e-6)
