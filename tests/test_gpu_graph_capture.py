"""
hipGraph capture of the product's ops (torch.cuda.graph around the Python calls; the C ABI launches on torch's capture stream): a
replay must RECOMPUTE from the current contents of its input buffers.  Every case changes its inputs between replays and compares
with an eager call -- with identical inputs a replay that silently does nothing looks correct (round 4: the memset node that reset
the persistent gather's work counters did not take effect between replays).
"""

import contextlib
import io
import warnings

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import synth

pytestmark = pytest.mark.gpu


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()                                       # one-time uploads, workspace growth, lazy allocations: outside the capture
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def check(name, fn, buffers, variants):
    """buffers: tensors the op reads; variants: list of tuples of replacement contents (same shapes)"""
    g, out = capture(fn)
    outs = out if isinstance(out, (list, tuple)) else [out]
    for contents in variants + variants[:1]:
        for buf, src in zip(buffers, contents):
            buf.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        want = fn()
        want = want if isinstance(want, (list, tuple)) else [want]
        for a, b in zip(got, want):
            assert torch.equal(a, b), name


def test_metric_path_ops_under_graph_replay(dev):
    mov, fix, trf = synth.cfg2_batch(2, 64, 32, device=dev, seed0=41)
    movs = [mov.clone(), torch.roll(mov, 3, dims=-1).contiguous()]
    fixs = [fix.clone(), torch.roll(fix, 9, dims=-1).contiguous()]
    trfs = [trf.clone(), (trf * 0.3).contiguous()]
    keep = ne.deferred.enabled
    ne.deferred.enabled = False                   # eager forms; the deferred pipeline is the last case
    try:
        st = ne.layers.SpatialTransformer()
        stn = ne.layers.SpatialTransformer(interp_method='nearest', fill_value=0.0)
        check('linear warp', lambda: st([mov, trf]), [mov, trf], [(movs[0], trfs[0]), (movs[1], trfs[1])])
        check('nearest warp', lambda: stn([mov, trf]), [mov, trf], [(movs[0], trfs[0]), (movs[1], trfs[1])])
        small = mov[:, ::2, ::2, ::2].contiguous()
        smalls = [small.clone(), (small * 0.5 + 0.1).contiguous()]
        check('resize', lambda: ne.layers.Resize(2)(small), [small], [(smalls[0],), (smalls[1],)])
        flow = trf[:, ::2, ::2, ::2].contiguous()
        flows = [flow.clone(), (flow * 0.25).contiguous()]
        check('vecint', lambda: ne.layers.VecInt(method='ss', int_steps=3)(flow), [flow], [(flows[0],), (flows[1],)])
        dice = ne.metrics.Dice(check_input_limits=False)
        check('soft dice', lambda: dice.dice(fix, mov), [fix, mov], [(fixs[0], movs[0]), (fixs[1], movs[1])])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            hard = ne.metrics.HardDice(32, input_type='prob', check_input_limits=False)
            check('hard dice from probabilities', lambda: hard.dice(fix, mov), [fix, mov], [(fixs[0], movs[0]), (fixs[1], movs[1])])
            m20, f20 = mov[..., :20].contiguous(), fix[..., :20].contiguous()
            hard20 = ne.metrics.HardDice(20, input_type='prob', check_input_limits=False)
            check('hard dice, 20 labels', lambda: hard20.dice(f20, m20), [f20, m20],
                  [(f20.clone(), m20.clone()), (torch.roll(f20, 4, dims=-1).contiguous(), torch.roll(m20, 2, dims=-1).contiguous())])
        lab_t, lab_p = fix.argmax(-1).to(torch.int32), mov.argmax(-1).to(torch.int32)
        hl = ne.metrics.HardDice(32, input_type='max_label', check_input_limits=False)
        check('hard dice of label maps', lambda: hl.dice(lab_t, lab_p), [lab_t, lab_p],
              [(lab_t.clone(), lab_p.clone()), (lab_p.clone(), torch.roll(lab_t, 5, dims=1).contiguous())])
        p = torch.softmax(torch.randn(2, 64, 64, 64, 32, device=dev), -1)
        ps = [p.clone(), torch.roll(p, 1, dims=-1).contiguous()]
        w = torch.rand(32, device=dev) + 0.5
        cce = ne.losses.CategoricalCrossentropy(label_weights=w)
        check('weighted cce', lambda: cce.loss(fix, p), [fix, p], [(fixs[0], ps[0]), (fixs[1], ps[1])])
        joint = ne.losses.multiple_losses_decorator([cce.loss, ne.losses.Dice(check_input_limits=False).loss])
        check('joint dice + cce', lambda: joint(fix, p), [fix, p], [(fixs[0], ps[0]), (fixs[1], ps[1])])
        ne.deferred.enabled = True
        check('deferred warp -> fused dice', lambda: dice.dice(fix, st([mov, trf])), [fix, mov, trf],
              [(fixs[0], movs[0], trfs[0]), (fixs[1], movs[1], trfs[1])])
    finally:
        ne.deferred.enabled = keep


def test_layers_under_graph_replay(dev):
    rng = np.random.default_rng(5)
    # LocallyConnected3D: vector kernel (batch 2) and matrix-core kernel (batch 5), bfloat16
    for batch in (2, 5):
        x = torch.randn(batch, 8, 9, 10, 16, device=dev).bfloat16()
        xs = [x.clone(), (x * 0.5).contiguous()]
        layer = ne.layers.LocallyConnected3D(16, (3, 3, 3), activation='elu').to(dev)
        with torch.no_grad():
            layer(x.float())
            layer.to(torch.bfloat16)
            check('lc3d batch %d' % batch, lambda: layer(x), [x], [(xs[0],), (xs[1],)])
    # unet forward (Keras predict semantics: no_grad)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(8, (32, 32, 32, 1), 2, 3, 4, feat_mult=2).to(dev)
    v = torch.randn(1, 32, 32, 32, 1, device=dev)
    vs = [v.clone(), torch.roll(v, 3, dims=2).contiguous()]
    check('unet forward', lambda: net(v), [v], [(vs[0],), (vs[1],)])
    assert rng is not None


def test_two_streams_do_not_share_scratch(dev):
    """the same ops issued on two streams at once (different inputs): reductions keep their partial sums in a per-(device, stream)
    workspace, the persistent gathers their work counters in that workspace / in a ring of counter sets -- results equal the serial ones"""
    a = synth.cfg2_batch(2, 96, 32, device=dev, seed0=51)
    b = synth.cfg2_batch(2, 96, 32, device=dev, seed0=77)
    dice = ne.metrics.Dice(check_input_limits=False)
    w = torch.rand(32, device=dev) + 0.5
    cce = ne.losses.CategoricalCrossentropy(label_weights=w)
    st10 = ne.layers.SpatialTransformer()
    st10._variant = 10

    def ops(mov, fix, trf):
        keep = ne.deferred.enabled
        ne.deferred.enabled = False
        try:
            return [ne.fused.warp_dice(mov, trf, fix), dice.dice(fix, mov), cce.loss(fix, mov.clamp_min(1e-3)), st10([mov, trf])]
        finally:
            ne.deferred.enabled = keep
    want_a, want_b = ops(*a), ops(*b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            got_a = ops(*a)
        with torch.cuda.stream(s2):
            got_b = ops(*b)
        with torch.cuda.stream(s1):
            got_a2 = ops(*a)
        torch.cuda.synchronize()
        for got, want in ((got_a, want_a), (got_b, want_b), (got_a2, want_a)):
            for x, y in zip(got, want):
                assert torch.equal(x, y)


def test_counter_slots_belong_to_streams_and_to_captured_launches(dev):
    """the work counters of the persistent gather and the tickets of the one-launch CCE used to come from a 64-slot
    round-robin ring -- two launches 64 ring launches apart on different streams, or a replayed graph and eager work, could meet in one
    slot.  Now a stream owns its slot, every launch recorded during capture owns one, and nothing else is ever handed the same words."""
    import ctypes as C
    from neurite_amd import _lib
    lib = _lib.lib()
    _lib.init_device(dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    idx = lambda s: lib.nrt_counters_slot_index(C.c_void_p(s.cuda_stream))
    i1, i2 = idx(s1), idx(s2)
    assert i1 >= 0 and i2 >= 0 and i1 != i2
    for _ in range(200):                                       # however many launches lie in between
        assert idx(s1) == i1 and idx(s2) == i2
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = torch.cuda.current_stream()
        c1, c2 = idx(cs), idx(cs)
    assert c1 >= 0 and c2 >= 0 and len({c1, c2, i1, i2}) == 4   # one slot per recorded launch, none of them a stream's

    # a graph replayed on one stream while eager launches of the same kernels run on another, many times over, different inputs
    a = synth.cfg2_batch(2, 96, 32, device=dev, seed0=61)
    b = synth.cfg2_batch(2, 96, 32, device=dev, seed0=87)
    w = torch.rand(32, device=dev) + 0.5
    cce = ne.losses.CategoricalCrossentropy(label_weights=w)

    def ops(mov, fix, trf):
        return [ne.fused.warp_dice(mov, trf, fix), cce.loss(fix, mov.clamp_min(1e-3))]
    want_a = [t.clone() for t in ops(*a)]
    want_b = [t.clone() for t in ops(*b)]
    g, out = capture(lambda: ops(*a))
    s3 = torch.cuda.Stream()
    for k in range(40):
        with torch.cuda.stream(s3):
            got_b = ops(*b)
        g.replay()
        with torch.cuda.stream(s3):
            got_b2 = ops(*b)
        torch.cuda.synchronize()
        for x, y in zip(out, want_a):
            assert torch.equal(x, y), 'replay %d' % k
        for got in (got_b, got_b2):
            for x, y in zip(got, want_b):
                assert torch.equal(x, y), 'eager launch beside replay %d' % k
    # recovery entry: zero-fills the pool (nothing is in flight here), results unchanged afterwards
    _lib.check(lib.nrt_counters_reset(_lib.stream_ptr(dev)), 'nrt_counters_reset')
    torch.cuda.synchronize()
    for x, y in zip(ops(*a), want_a):
        assert torch.equal(x, y)


def test_views_and_misaligned_storage(dev):
    """inputs that are views with a 4-byte storage offset (not 16-byte aligned) or non-contiguous: every op gives the numbers it gives
    on aligned contiguous copies (the vector kernels need 16-byte rows; the entry points must notice, not fault)"""
    mov, fix, trf = synth.cfg2_batch(2, 48, 32, device=dev, seed0=61)

    def shifted(t):                                # same values, storage offset of one element
        flat = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
        v = flat[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 != 0 and v.is_contiguous()
        return v

    def strided(t):                                # same values, channel-last view of a channel-first buffer
        v = t.permute(0, 4, 1, 2, 3).contiguous().permute(0, 2, 3, 4, 1)
        assert not v.is_contiguous()
        return v
    keep = ne.deferred.enabled
    ne.deferred.enabled = False
    try:
        dice = ne.metrics.Dice(check_input_limits=False)
        w = torch.rand(32, device=dev) + 0.5
        cce = ne.losses.CategoricalCrossentropy(label_weights=w)
        st, stn = ne.layers.SpatialTransformer(), ne.layers.SpatialTransformer(interp_method='nearest')
        p = mov.clamp_min(1e-3)
        ref = dict(warp=st([mov, trf]), nearest=stn([mov, trf]), dice=dice.dice(fix, mov), cce=cce.loss(fix, p),
                   fused=ne.fused.warp_dice(mov, trf, fix), resize=ne.layers.Resize(2)(mov[:, ::2, ::2, ::2].contiguous()))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref['hard'] = ne.metrics.HardDice(32, input_type='prob', check_input_limits=False).dice(fix, mov)
        for name, f in (('offset', shifted), ('strided', strided)):
            m, fx, pp = f(mov), f(fix), f(p)
            t = f(trf) if name == 'offset' else trf.permute(0, 4, 1, 2, 3).contiguous().permute(0, 2, 3, 4, 1)
            assert torch.equal(st([m, t]), ref['warp']), name
            assert torch.equal(stn([m, t]), ref['nearest']), name
            assert torch.equal(dice.dice(fx, m), ref['dice']), name
            assert torch.equal(ne.fused.warp_dice(m, t, fx), ref['fused']), name
            np.testing.assert_allclose(float(cce.loss(fx, pp)), float(ref['cce']), rtol=1e-6)
            assert torch.equal(ne.layers.Resize(2)(f(mov[:, ::2, ::2, ::2].contiguous())), ref['resize']), name
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                assert torch.equal(ne.metrics.HardDice(32, input_type='prob', check_input_limits=False).dice(fx, m), ref['hard']), name
    finally:
        ne.deferred.enabled = keep
    # layers: a unet and a LocallyConnected3D on a view with a 4-byte storage offset
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(8, (16, 16, 16, 2), 2, 3, 4, feat_mult=2).to(dev)
    v = torch.randn(1, 16, 16, 16, 2, device=dev)
    assert torch.equal(net(shifted(v)), net(v)) and torch.equal(net(strided(v)), net(v))
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(3, 7, 8, 9, 16, device=dev).to(dtype)
        layer = ne.layers.LocallyConnected3D(16, (3, 3, 3)).to(dev)
        with torch.no_grad():
            layer(x.float())
            layer.to(dtype)
            want = layer(x)
            # (a misaligned volume takes the un-staged vector kernel: same products, another summation order)
            tol = 1e-5 if dtype == torch.float32 else 2e-2
            for got in (layer(shifted(x)), layer(strided(x))):
                np.testing.assert_allclose(got.float().cpu().numpy(), want.float().cpu().numpy(), rtol=tol, atol=tol)

