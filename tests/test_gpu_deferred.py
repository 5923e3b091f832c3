"""
The reference-signature pipeline  warped = SpatialTransformer()([moving, trf]); d = Dice().dice(fixed, warped)
(neurite/tf/models.py:806-807 + neurite/tf/metrics.py:415-482) reaching the fused kernel through a deferred warp
(neurite_amd/deferred.py): same numbers as the eager two-kernel form, `warped` bit-identical whenever it is evaluated.
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import bits_equal
from neurite_amd import synth
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
F = np.float32


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture()
def batch(dev):
    return synth.cfg2_batch(2, 32, 8, device=dev, seed0=11)


def eager(fn):
    keep = ne.deferred.enabled
    ne.deferred.enabled = False
    try:
        return fn()
    finally:
        ne.deferred.enabled = keep


def test_deferred_pipeline_matches_eager_and_oracle(dev, batch):
    mov, fix, trf = batch
    st = ne.layers.SpatialTransformer()
    dice = ne.metrics.Dice(check_input_limits=False)
    warped = st([mov, trf])
    assert isinstance(warped, ne.deferred.DeferredWarp) and warped.pending
    assert tuple(warped.shape) == tuple(fix.shape) and warped.dtype == torch.float32 and warped.device == mov.device
    d = dice.dice(fix, warped)
    assert warped.pending                                        # Dice ran the fused kernel: the warp was never written
    w_e = eager(lambda: st([mov, trf]))
    assert type(w_e) is torch.Tensor
    d_e = dice.dice(fix, w_e)
    np.testing.assert_allclose(N(d), N(d_e), rtol=1e-6, atol=1e-7)
    # symmetric in the two maps, mean_dice and the loss wrappers go the same way
    np.testing.assert_allclose(N(dice.dice(st([mov, trf]), fix)), N(d_e), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(float(dice.mean_dice(fix, st([mov, trf]))), float(d_e.mean()), rtol=1e-6)
    np.testing.assert_allclose(float(ne.losses.Dice(check_input_limits=False).mean_loss(fix, st([mov, trf]))),
                               -float(d_e.mean()), rtol=1e-6)
    # against the C oracle
    for b in range(2):
        w = co.interpn(N(mov)[b], N(trf)[b], 'linear', None, loc_mode=1)
        sums, _ = co.dice_sums(N(fix)[b:b + 1], w[None])
        np.testing.assert_allclose(N(d)[b], co.dice_from_sums(sums)[0], rtol=1e-5)
    # any other use evaluates the warp with the stand-alone kernel: bit-identical to the eager result
    assert bits_equal(N(warped), N(w_e)) and not warped.pending
    w2 = st([mov, trf])
    assert torch.equal(w2[0, 3:5], w_e[0, 3:5]) and not w2.pending
    w3 = st([mov, trf])
    assert bits_equal(np.asarray(w3.cpu()), N(w_e)) and float((w3 - w_e).abs().max()) == 0.0
    # a materialised DeferredWarp handed to Dice takes the ordinary kernel
    np.testing.assert_allclose(N(dice.dice(fix, w3)), N(d_e), rtol=0, atol=0)


def test_default_range_assert_without_a_host_round_trip(dev, batch):
    """VERDICT r5 item 5 (neurite/tf/metrics.py:439-444): `Dice()` with the reference's default `check_input_limits=True` on the
    deferred-warp pipeline does not stop the host at every call.  The values come back as a CheckedTensor: device operations pass, host
    access raises `InvalidArgumentError('value outside range')`; a result that never reaches the host raises at a LATER deferred-assert call
    (or at `checked.flush()`), once its extrema have arrived; `checked.enabled = False` raises at the call site."""
    from neurite_amd import checked
    mov, fix, trf = batch
    st = ne.layers.SpatialTransformer()
    D = ne.metrics.Dice()
    checked.flush()
    overshoots = float(eager(lambda: st([mov, trf])).max()) > 1.0
    assert overshoots                                            # (one-hot maps, tri-linear weights: an ulp above 1)
    want = ne.metrics.Dice(check_input_limits=False).dice(fix, st([mov, trf]))
    # ---- a FAILING assert -----------------------------------------------------------------------------------------------------------
    d = D.dice(fix, st([mov, trf]))                              # no exception here, no host synchronisation
    assert isinstance(d, checked.CheckedTensor) and tuple(d.shape) == tuple(want.shape) and d.device == want.device
    m = (d * 2.0).sum()                                          # device operations see the values
    assert type(m) is torch.Tensor and float(m) == float((want * 2.0).sum())
    for touch in (lambda: d.cpu(), lambda: d[0, 0].item(), lambda: d.tolist(), lambda: d.detach().cpu().numpy(), lambda: repr(d),
                  lambda: d.numpy(), lambda: d.reshape(-1)[:3].clone().tolist(), lambda: float(d[1, 2])):
        with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):
            touch()
    assert checked.pending_count() == 0
    # ... that nobody looks at on the host is reported by a later call, or by flush()
    d1 = D.dice(fix, st([mov, trf]))
    torch.cuda.synchronize()
    with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):
        D.dice(fix, st([mov, trf]))
    checked._pending.clear()
    d2 = D.dice(fix, st([mov, trf]))
    with pytest.raises(ne.errors.InvalidArgumentError):
        checked.flush()
    assert checked.pending_count() == 0
    del d1, d2
    # mean_dice stops the host for its finite test anyway: the assert is looked at there
    with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):
        D.mean_dice(fix, st([mov, trf]))
    checked.flush()
    # ---- a PASSING assert: maps scaled into the range ---------------------------------------------------------------------------------
    movh, fixh = mov * 0.5, fix * 0.5
    wanth = ne.metrics.Dice(check_input_limits=False).dice(fixh, st([movh, trf]))
    for _ in range(3):
        dh = D.dice(fixh, st([movh, trf]))
        assert isinstance(dh, checked.CheckedTensor)
    assert torch.equal(dh.cpu(), wanth.cpu()) and bits_equal(N(dh), N(wanth))
    checked.flush()
    assert checked.pending_count() == 0
    assert float(D.mean_dice(fixh, st([movh, trf]))) == float(wanth.mean())
    # ---- the switch: eager raise at the call site ---------------------------------------------------------------------------------------
    keep = checked.enabled
    checked.enabled = False
    try:
        with pytest.raises(ne.errors.InvalidArgumentError, match='value outside range'):
            D.dice(fix, st([mov, trf]))
        assert type(D.dice(fixh, st([movh, trf]))) is torch.Tensor
    finally:
        checked.enabled = keep


def test_deferred_respects_semantics(dev, batch):
    mov, fix, trf = batch
    st = ne.layers.SpatialTransformer()
    # the reference's default range assert fires for a tri-linearly warped one-hot map (an ulp above 1) in both forms (the deferred
    # pipeline raises when the values are brought to the host: neurite_amd/checked.py, test_default_range_assert_without_a_host_round_trip)
    for form in (lambda: ne.metrics.Dice().dice(fix, st([mov, trf])).cpu(), lambda: eager(lambda: ne.metrics.Dice().dice(fix, st([mov, trf])))):
        try:
            form()
            raised = False
        except ne.errors.InvalidArgumentError:
            raised = True
        assert raised == (float(eager(lambda: st([mov, trf])).max()) > 1.0)
    # fill value, single_transform, laplace smoothing travel with the deferred warp
    stf = ne.layers.SpatialTransformer(fill_value=0.0, single_transform=True)
    d = ne.metrics.Dice(check_input_limits=False, laplace_smoothing=0.5).dice(fix, stf([mov, trf[:1]]))
    d_e = eager(lambda: ne.metrics.Dice(check_input_limits=False, laplace_smoothing=0.5).dice(fix, stf([mov, trf[:1]])))
    np.testing.assert_allclose(N(d), N(d_e), rtol=1e-6, atol=1e-7)
    # not deferred: nearest interpolation, gradients being recorded, label counts the fused kernel does not take
    assert type(ne.layers.SpatialTransformer(interp_method='nearest')([mov, trf])) is torch.Tensor
    tg = trf.clone().requires_grad_()
    out = st([mov, tg])
    assert type(out) is torch.Tensor and out.requires_grad
    with torch.no_grad():
        assert isinstance(st([mov, tg]), ne.deferred.DeferredWarp)
    assert type(st([mov[..., :3].contiguous(), trf])) is torch.Tensor
    # normalize=True and hard Dice evaluate the warp and take their own kernels
    w = st([mov, trf])
    dn = ne.metrics.Dice(check_input_limits=False, normalize=True).dice(fix, w)
    assert not w.pending
    np.testing.assert_allclose(N(dn), N(eager(lambda: ne.metrics.Dice(check_input_limits=False, normalize=True).dice(fix, st([mov, trf])))),
                               rtol=1e-6)
    # a gradient wrt the OTHER map: the deferred warp is evaluated and the ordinary backward runs
    fg = fix.clone().requires_grad_()
    ne.metrics.Dice(check_input_limits=False).mean_dice(fg, st([mov, trf])).backward()
    assert fg.grad is not None and float(fg.grad.abs().sum()) > 0
    # host accessors that bypass the dispatcher
    w = st([mov, trf])
    assert w.data_ptr() != 0 and not w.pending
    assert isinstance(st([mov, trf]).tolist()[0][0][0][0][0], float)
    assert st([mov, trf]).cpu().numpy().shape == tuple(fix.shape)


def test_spatial_transformer_under_inference_mode(dev, batch):
    """torch.inference_mode(): inference tensors carry no version counter, so an in-place change could not be detected -- the layer
    warps eagerly there (bit-identical values, no DeferredWarp, no exception)."""
    mov, fix, trf = batch
    st = ne.layers.SpatialTransformer()
    dice = ne.metrics.Dice(check_input_limits=False)
    ref = eager(lambda: st([mov, trf]))
    d_ref = eager(lambda: dice.dice(fix, ref))
    with torch.inference_mode():
        flow = trf * 1.0                                             # an inference tensor, as a network's output would be
        w = st([mov, flow])
        assert not isinstance(w, ne.deferred.DeferredWarp)
        assert bits_equal(N(w), N(ref))
        np.testing.assert_allclose(N(dice.dice(fix, w)), N(d_ref), rtol=1e-6, atol=1e-7)
    w2 = st([mov, flow])                                             # inference tensor used outside the mode: still eager
    assert not isinstance(w2, ne.deferred.DeferredWarp) and bits_equal(N(w2), N(ref))


def test_deferred_warp_is_safe_against_buffer_reuse(dev, batch):
    """warped = st([buf, trf]); buf.copy_(next); dice(fixed, warped): the eager path (and TF's immutable tensors) give the Dice of the
    ORIGINAL data.  The deferred path must give the same or refuse loudly -- never the Dice of the new data."""
    mov, fix, trf = batch
    other = torch.roll(mov, 3, dims=1).contiguous()
    st = ne.layers.SpatialTransformer()
    dice = ne.metrics.Dice(check_input_limits=False)
    d_ref = eager(lambda: dice.dice(fix, st([mov, trf])))
    # buffer reuse between the layer and its consumer
    for mutate in (lambda buf, t: buf.copy_(other), lambda buf, t: buf.mul_(0.5), lambda buf, t: t.add_(1.0), lambda buf, t: buf[:, :4].zero_()):
        buf, t = mov.clone(), trf.clone()
        warped = st([buf, t])
        assert isinstance(warped, ne.deferred.DeferredWarp)
        mutate(buf, t)
        with pytest.raises(ne.deferred.DeferredWarpError):
            dice.dice(fix, warped)                                   # the fused path checks before it launches
        with pytest.raises(ne.deferred.DeferredWarpError):
            warped.sum()                                             # and so does every other consumer
    # eager evaluation (deferred.enabled = False) is the documented remedy and gives the value of the original data
    buf = mov.clone()
    w = eager(lambda: st([buf, trf]))
    buf.copy_(other)
    np.testing.assert_allclose(N(dice.dice(fix, w)), N(d_ref), rtol=1e-6, atol=1e-7)
    # inputs that had to be converted are private copies: overwriting the caller's tensor afterwards is harmless
    buf16 = mov.to(torch.float16)
    w16 = st([buf16, trf])
    ref16 = eager(lambda: st([mov.to(torch.float16), trf]))
    buf16.zero_()
    assert bits_equal(N(w16.float()), N(ref16.float()))
    # unaligned / non-contiguous second maps fall back to the ordinary kernels instead of surfacing kernel argument errors
    fix_pad = torch.zeros((fix.shape[0],) + tuple(fix.shape[1:-1]) + (fix.shape[-1] + 1,), device=dev)[..., 1:]
    fix_pad.copy_(fix)
    np.testing.assert_allclose(N(dice.dice(fix_pad, st([mov, trf]))), N(d_ref), rtol=1e-6, atol=1e-7)


def test_full_size_reference_api_pipeline_and_few_channel_warps(dev):
    """BASELINE config 2 size (160^3 x 32 one-hot, sigma = 3 field): the reference-signature pipeline with the deferred warp against
    the C oracle's warp + Dice; the same maps stored as bfloat16 give the same Dice (bit for bit against the same kernel structure); and the few-channel kernels
    (variant 8) at 160^3: a C = 1 image and a C = 3 flow warped by the field, bit-exact against the C oracle"""
    mov, fix, trf = synth.cfg2_batch(1, 160, 32, device=dev, seed0=1)
    d = ne.metrics.Dice(check_input_limits=False).dice(fix, ne.layers.SpatialTransformer()([mov, trf]))
    w = co.interpn(N(mov)[0], N(trf)[0], 'linear', None, loc_mode=1)
    sums, _ = co.dice_sums(N(fix), w[None])
    np.testing.assert_allclose(N(d), co.dice_from_sums(sums), rtol=1e-5)
    # bfloat16 storage, float32 arithmetic: the warped rows are the same values; against the same kernel structure (register kernel,
    # tune bit 30) the Dice is the same bit for bit, against the default float32 kernel (wave-cache: another summation order) to 1e-6
    d16 = N(ne.fused.warp_dice(mov.bfloat16(), trf, fix.bfloat16()))
    assert bits_equal(d16, N(ne.fused.warp_dice(mov, trf, fix, _tune=1 << 30)))
    np.testing.assert_allclose(d16, N(ne.fused.warp_dice(mov, trf, fix)), rtol=1e-6)
    del mov, fix, w
    rng = np.random.default_rng(2)
    for C in (1, 3):
        vol = rng.standard_normal((160, 160, 160, C)).astype(F)
        got = N(ne.layers.SpatialTransformer()([torch.from_numpy(vol[None]).to(dev), trf]))[0]
        assert bits_equal(got, co.interpn(vol, N(trf)[0], 'linear', None, loc_mode=1)), C
    # Resize(2) of a half-resolution flow (models.py:804) at full size, lean row kernel vs the oracle's resize
    from oracle import np_oracle as npo
    half = rng.standard_normal((80, 80, 80, 3)).astype(F)
    got = N(ne.layers.Resize(2)(torch.from_numpy(half[None]).to(dev)))[0]
    lin = [npo.tf_linspace(0., 79., 160) for _ in range(3)]
    assert bits_equal(got, co.interpn(half, lin, 'linear', None, loc_mode=2))


def test_deferred_warp_behaves_like_its_tensor_everywhere_else(dev, batch):
    """a pending warp handed to code that knows nothing about it -- pickling, copies, .to(), saving, views, in-place operations, stacking,
    printing -- is evaluated on first use and from then on IS the eager result ('fragile under pickling and .to()')"""
    import copy
    import io
    import pickle
    mov, fix, trf = batch
    st = ne.layers.SpatialTransformer()
    want = eager(lambda: st([mov, trf]))

    def pending():
        w = st([mov, trf])
        assert isinstance(w, ne.deferred.DeferredWarp) and w.pending
        return w

    back = pickle.loads(pickle.dumps(pending()))
    assert torch.equal(torch.as_tensor(back).to(dev), want)
    assert torch.equal(copy.deepcopy(pending()), want) and torch.equal(copy.copy(pending()), want)
    buf = io.BytesIO()
    torch.save(pending(), buf)
    buf.seek(0)
    assert torch.equal(torch.load(buf, weights_only=False).to(dev), want)
    for conv in (lambda w: w.to('cpu'), lambda w: w.cpu(), lambda w: w.to(torch.float64), lambda w: w.to(dev, torch.float16), lambda w: w.double(),
                 lambda w: w.contiguous(), lambda w: w.clone(), lambda w: w.detach(), lambda w: w.to(dev)):
        got = conv(pending())
        assert torch.equal(got.to(dev, torch.float32), conv(want).to(dev, torch.float32))
    # views, reductions, stacking, arithmetic with ordinary tensors on either side
    w = pending()
    assert torch.equal(w[1, 3:9, ..., 2], want[1, 3:9, ..., 2]) and not w.pending
    assert torch.equal(pending().permute(0, 4, 1, 2, 3), want.permute(0, 4, 1, 2, 3))
    assert torch.equal(pending().reshape(2, -1, 8), want.reshape(2, -1, 8))
    assert float(pending().sum()) == float(want.sum()) and float(pending().max()) == float(want.max())
    assert torch.equal(torch.stack([pending(), pending()]), torch.stack([want, want]))
    assert torch.equal(fix - pending(), fix - want) and torch.equal(pending() * 2, want * 2)
    assert torch.equal(torch.where(pending() > 0.5, fix, mov), torch.where(want > 0.5, fix, mov))
    # in-place operations act on the evaluated values
    w = pending()
    w += 1
    assert torch.equal(w, want + 1)
    w = pending()
    w.clamp_(0.2, 0.8)
    assert torch.equal(w, want.clamp(0.2, 0.8))
    # as the OUT / source of copies, as a module input
    dst = torch.empty_like(want)
    dst.copy_(pending())
    assert torch.equal(dst, want)
    assert torch.equal(torch.nn.Identity()(pending()), want)
    assert torch.equal(torch.nn.functional.avg_pool3d(pending().permute(0, 4, 1, 2, 3), 2), torch.nn.functional.avg_pool3d(want.permute(0, 4, 1, 2, 3), 2))
    # metadata never evaluates
    w = pending()
    assert tuple(w.shape) == tuple(want.shape) and w.dtype == want.dtype and w.device == want.device and w.dim() == 5 and w.numel() == want.numel()
    assert w.is_contiguous() and not w.requires_grad and w.pending
    assert 'pending' in repr(w) and w.pending                     # printing a pending warp says so instead of computing it
    assert 'tensor' in repr(w.materialize()) and 'tensor' in repr(w)
