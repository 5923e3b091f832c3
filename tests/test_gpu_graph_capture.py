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
