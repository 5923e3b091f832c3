"""
Generate tests/golden/*.npz by running the REFERENCE'S OWN SOURCE (/root/reference/neurite)
on top of tests/golden/tf_shim.py (a NumPy stand-in for the TensorFlow leaf primitives).

    MPLBACKEND=Agg python tests/golden/make_golden.py

Runs only in the build container (it needs /root/reference); the produced fixtures are
committed and travel to the GPU box.  Nothing at test/bench run time reads /root/reference.
Each fixture stores the inputs, the call's keyword arguments and the reference's outputs.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tf_shim  # noqa: E402

tf_shim.install()
sys.path.insert(0, '/root/reference')
import neurite as ne  # noqa: E402  (the real reference package)

T = tf_shim.Tensor
F = np.float32


def A(x):
    return x.a if isinstance(x, T) else np.asarray(x)


def ijk(shape):
    return np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing='ij'), -1).astype(F)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def gen_interpn():
    rng = np.random.default_rng(1234)
    cases = {}

    def add(tag, vol, loc, method='linear', fill=None, loc_list=False):
        lv = [T(np.ascontiguousarray(loc[..., d])) for d in range(loc.shape[-1])] if loc_list else T(loc)
        out = A(ne.utils.interpn(T(vol), lv, interp_method=method, fill_value=fill))
        cases[tag + '__vol'] = vol
        cases[tag + '__loc'] = loc
        cases[tag + '__method'] = np.array(method)
        cases[tag + '__fill'] = np.array(np.nan if fill is None else fill, F)
        cases[tag + '__hasfill'] = np.array(fill is not None)
        cases[tag + '__out'] = out

    S = (9, 7, 11)
    vol3 = rng.standard_normal(S + (3,)).astype(F)
    loc3 = (ijk(S) + rng.normal(0, 3, S + (3,))).astype(F)
    for method in ('linear', 'nearest'):
        for fill in (None, 0.0, -2.5):
            add('d3c3_%s_%s' % (method, 'nofill' if fill is None else ('fill%g' % fill)), vol3, loc3, method, fill)
    # half-integer and exactly-on-grid locations (round-half-even, unit weights), far out of range
    special = np.array([-3.5, -1.0, -0.5, -0.0, 0.0, 0.5, 1.0, 1.5, 2.5, 3.0, 5.5, 6.0, 6.5, 7.0, 10.0, 10.5,
                        11.0, 1e6, -1e6], F)
    g = np.stack(np.meshgrid(special, special[:7], special, indexing='ij'), -1).astype(F)
    for method in ('linear', 'nearest'):
        add('d3c3_special_%s' % method, vol3, g, method, None)
        add('d3c3_special_%s_fill' % method, vol3, g, method, 7.0)
    # list-of-tensors loc, volume without channel axis, different output shape
    volnc = rng.standard_normal((6, 8, 5)).astype(F)
    locnc = rng.uniform(-2, 9, (4, 3, 7, 3)).astype(F)
    add('d3_nochan_list', volnc, locnc, 'linear', None, loc_list=True)
    # C = 32 (the headline channel count), C = 1, a singleton spatial dim
    vol32 = rng.standard_normal((6, 5, 7, 32)).astype(F)
    loc32 = (ijk((6, 5, 7)) + rng.normal(0, 1.5, (6, 5, 7, 3))).astype(F)
    add('d3c32_linear', vol32, loc32, 'linear', None)
    add('d3c32_nearest_fill', vol32, loc32, 'nearest', 0.0)
    vol1s = rng.standard_normal((1, 6, 4, 2)).astype(F)
    loc1s = rng.uniform(-1, 6, (3, 6, 4, 3)).astype(F)
    add('d3_singleton_dim', vol1s, loc1s, 'linear', None)
    # 2-D and 1-D
    vol2 = rng.standard_normal((13, 5, 4)).astype(F)
    loc2 = rng.uniform(-2, 15, (6, 8, 2)).astype(F)
    add('d2c4_linear', vol2, loc2, 'linear', None)
    add('d2c4_nearest_fill', vol2, loc2, 'nearest', 1.5)
    vol1 = rng.standard_normal((13, 4)).astype(F)
    loc1 = rng.uniform(-2, 15, (20, 1)).astype(F)
    add('d1c4_linear_fill', vol1, loc1, 'linear', 0.0)
    # integer locations (cast to float32, utils.py:123-125) and integer-valued volume with nearest
    loci = rng.integers(-2, 12, (5, 4, 6, 3)).astype(np.int32)
    add('d3c3_intloc', vol3, loci, 'linear', None)
    voli = rng.integers(0, 32, S + (1,)).astype(np.int32)
    add('d3_intvol_nearest', voli, loc3, 'nearest', None)
    save('interpn_small', **cases)

    # BASELINE config 1: 32^3 fp32 volume, loc = ijk + N(0, 3), seed 0
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((32, 32, 32)).astype(F)
    loc = (ijk((32, 32, 32)) + rng.normal(0, 3, (32, 32, 32, 3)).astype(F)).astype(F)
    out = A(ne.utils.interpn(T(vol), T(loc), interp_method='linear'))
    save('interpn_cfg1_32', vol=vol, loc=loc, out=out)



def gen_interpn_dtypes():
    """interpn on the volume dtypes / ranks beyond float32 1-3-D: float16 and float64 volumes (loc is cast to the volume
    dtype and the arithmetic runs in it, utils.py:123-127), 4-D and 5-D volumes (2^D corners, :159)."""
    rng = np.random.default_rng(4321)
    cases = {}

    def add(tag, vol, loc, method='linear', fill=None):
        out = A(ne.utils.interpn(T(vol), T(loc), interp_method=method, fill_value=fill))
        assert out.dtype == vol.dtype, (tag, out.dtype, vol.dtype)
        cases[tag + '__vol'] = vol
        cases[tag + '__loc'] = loc
        cases[tag + '__method'] = np.array(method)
        cases[tag + '__fill'] = np.array(np.nan if fill is None else fill, F)
        cases[tag + '__hasfill'] = np.array(fill is not None)
        cases[tag + '__out'] = out

    S = (9, 7, 11)
    base = rng.standard_normal(S + (3,))
    loc = (ijk(S) + rng.normal(0, 3, S + (3,))).astype(F)
    special = np.array([-3.5, -0.5, 0.0, 0.5, 1.5, 2.5, 6.0, 6.5, 10.0, 10.5, 1e4, -1e4], F)
    gs = np.stack(np.meshgrid(special, special[:6], special, indexing='ij'), -1).astype(F)
    for name, dt in (('f16', np.float16), ('f64', np.float64)):
        vol = base.astype(dt)
        for method in ('linear', 'nearest'):
            for fill in (None, -2.5):
                add('%s_d3c3_%s_%s' % (name, method, 'nofill' if fill is None else 'fill'), vol, loc, method, fill)
            add('%s_d3c3_special_%s' % (name, method), vol, gs, method, 7.0)
        add('%s_d3c3_locsame' % name, vol, loc.astype(dt), 'linear', None)          # loc already of the volume dtype
        v2 = rng.standard_normal((13, 5, 4)).astype(dt)
        add('%s_d2c4_linear' % name, v2, rng.uniform(-2, 15, (6, 8, 2)).astype(F), 'linear', 0.0)
        v1 = rng.standard_normal((13, 4)).astype(dt)
        add('%s_d1c4_linear' % name, v1, rng.uniform(-2, 15, (20, 1)).astype(F), 'linear', None)
    add('f64_d3c3_loc64', base.astype(np.float64), loc.astype(np.float64) + 1e-9, 'linear', None)
    # ranks 4 and 5 (float32 and one float16 / int32-nearest case)
    v4 = rng.standard_normal((4, 5, 3, 6, 2)).astype(F)
    l4 = rng.uniform(-1.5, 6.5, (3, 4, 2, 5, 4)).astype(F)
    add('f32_d4c2_linear', v4, l4, 'linear', None)
    add('f32_d4c2_linear_fill', v4, l4, 'linear', 1.25)
    add('f32_d4c2_nearest', v4, l4, 'nearest', None)
    add('f16_d4c2_linear', v4.astype(np.float16), l4, 'linear', None)
    add('i32_d4c1_nearest', rng.integers(0, 9, (4, 5, 3, 6, 1)).astype(np.int32), l4, 'nearest', None)
    v5 = rng.standard_normal((3, 2, 4, 3, 2, 2)).astype(F)
    l5 = rng.uniform(-1, 4, (2, 3, 2, 2, 3, 5)).astype(F)
    add('f32_d5c2_linear', v5, l5, 'linear', 0.0)
    save('interpn_dtypes', **cases)


def gen_resize():
    rng = np.random.default_rng(77)
    cases = {}
    vol = rng.standard_normal((5, 6, 7, 2)).astype(F)
    for tag, z, method in (('x2', 2, 'linear'), ('half', 0.5, 'linear'), ('aniso', [1.5, 2, 0.7], 'linear'),
                           ('x3_nearest', 3, 'nearest'), ('mixed1', [1, 2, 1], 'linear')):
        out = A(ne.utils.resize(T(vol), z, interp_method=method))
        cases[tag + '__vol'] = vol
        cases[tag + '__zoom'] = np.atleast_1d(np.array(z, np.float64))
        cases[tag + '__method'] = np.array(method)
        cases[tag + '__out'] = out
    # identity zoom returns the input object (utils.py:250-251)
    v = T(vol)
    assert ne.utils.zoom(v, 1) is v
    # Resize layer: batch map (layers.py:154-181); deformation-field upsample 2x as in models.py:804
    x = rng.standard_normal((2, 4, 5, 3, 3)).astype(F)
    layer = ne.layers.Resize(2, interp_method='linear')
    out = A(layer(T(x)))
    cases['layer__x'] = x
    cases['layer__zoom'] = np.array([2.0])
    cases['layer__out'] = out
    cases['layer__out_shape'] = np.array(layer.compute_output_shape(x.shape))
    x2 = rng.standard_normal((3, 6, 4, 2)).astype(F)          # 2-D, anisotropic zoom list
    layer2 = ne.layers.Zoom([0.5, 1.5])
    cases['layer2d__x'] = x2
    cases['layer2d__zoom'] = np.array([0.5, 1.5])
    cases['layer2d__out'] = A(layer2(T(x2)))
    save('resize_small', **cases)


def gen_index_helpers():
    rng = np.random.default_rng(5)
    cases = {}
    siz = (4, 5, 6)
    subs = [rng.integers(0, s, (7, 3)).astype(np.int32) for s in siz]
    cases['sub2ind__siz'] = np.array(siz)
    for d in range(3):
        cases['sub2ind__sub%d' % d] = subs[d]
    cases['sub2ind__out'] = A(ne.utils.sub2ind2d(tf_shim.TensorShape(siz), [T(s) for s in subs]))
    ws = [rng.standard_normal(11).astype(F) for _ in range(3)]
    for d in range(3):
        cases['prodn__w%d' % d] = ws[d]
    cases['prodn__out'] = A(ne.utils.prod_n([T(w) for w in ws]))
    for d, g in enumerate(ne.utils.volshape_to_ndgrid((3, 4, 2))):
        cases['ndgrid__%d' % d] = A(g)
    # NOTE: for 'xy' indexing the reference's tile-based meshgrid (utils.py:471-475) swaps the tile
    # multiples incorrectly when the first two sizes differ: it returns grids of shape (3, 3, 2) for
    # sizes (3, 4, 2) where tf.meshgrid returns (4, 3, 2).  Both are recorded; neurite_amd follows
    # tf.meshgrid (the documented intent) -- see DESIGN.md.  'ij' (all the hot path uses) is unaffected.
    for d, g in enumerate(ne.utils.volshape_to_meshgrid((3, 4, 2), indexing='xy')):
        cases['meshgrid_xy_nonsquare_refbug__%d' % d] = A(g)
    for d, g in enumerate(ne.utils.volshape_to_meshgrid((3, 3, 2), indexing='xy')):
        cases['meshgrid_xy__%d' % d] = A(g)
    for d, g in enumerate(ne.utils.volshape_to_meshgrid((3, 4, 2), indexing='ij')):
        cases['meshgrid_ij__%d' % d] = A(g)
    x = rng.standard_normal((2, 3, 4, 5, 6)).astype(F)
    cases['bcf__x'] = x
    cases['bcf__out'] = A(ne.utils.batch_channel_flatten(T(x)))
    cases['flatten_axes_12__out'] = A(ne.utils.flatten_axes(T(x), [1, 2]))
    save('index_helpers', **cases)


def gen_dice():
    import warnings
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(99)
    cases = {}
    B, S, L = 2, (6, 5, 7), 5
    lab_t = rng.integers(0, L, (B,) + S)
    lab_p = np.where(rng.random((B,) + S) < 0.7, lab_t, rng.integers(0, L, (B,) + S))
    lab_p[0][lab_p[0] == 3] = 0                     # label 3 absent from batch 0's prediction
    lab_t[1][lab_t[1] == 4] = 1
    lab_p[1][lab_p[1] == 4] = 1                     # label 4 absent from both in batch 1 (0/0 -> 0)
    oh_t = np.eye(L, dtype=F)[lab_t]
    oh_p = np.eye(L, dtype=F)[lab_p]
    pr_t = rng.random((B,) + S + (L,)).astype(F)
    pr_t /= pr_t.sum(-1, keepdims=True)
    pr_p = rng.random((B,) + S + (L,)).astype(F)
    pr_p /= pr_p.sum(-1, keepdims=True)
    pr_t = np.clip(pr_t, 0, 1).astype(F)
    pr_p = np.clip(pr_p, 0, 1).astype(F)
    w = rng.random((1, L)).astype(F)
    cases.update(lab_t=lab_t.astype(np.int32), lab_p=lab_p.astype(np.int32), oh_t=oh_t, oh_p=oh_p,
                 pr_t=pr_t, pr_p=pr_p, w=w)

    cases['soft_onehot'] = A(ne.metrics.Dice().dice(T(oh_t), T(oh_p)))
    cases['soft_prob'] = A(ne.metrics.SoftDice().dice(T(pr_t), T(pr_p)))
    cases['soft_prob_laplace'] = A(ne.metrics.SoftDice(laplace_smoothing=0.1).dice(T(pr_t), T(pr_p)))
    unn_t, unn_p = (pr_t * F(0.5)).astype(F), (pr_p * F(0.25)).astype(F)
    cases['soft_prob_normalize'] = A(ne.metrics.SoftDice(normalize=True).dice(T(unn_t), T(unn_p)))
    cases['unn_t'], cases['unn_p'] = unn_t, unn_p
    cases['mean_soft_prob'] = A(ne.metrics.Dice().mean_dice(T(pr_t), T(pr_p)))
    cases['mean_soft_prob_w'] = A(ne.metrics.Dice(weights=T(w)).mean_dice(T(pr_t), T(pr_p)))
    cases['hard_prob'] = A(ne.metrics.HardDice(L, input_type='prob').dice(T(pr_t), T(pr_p)))
    cases['hard_prob_nolabels'] = A(ne.metrics.Dice(dice_type='hard', input_type='prob').dice(T(pr_t), T(pr_p)))
    cases['hard_label'] = A(ne.metrics.HardDice(L).dice(T(lab_t.astype(np.int32)), T(lab_p.astype(np.int32))))
    cases['hard_label_laplace'] = A(ne.metrics.HardDice(L, laplace_smoothing=1.0).dice(
        T(lab_t.astype(np.int32)), T(lab_p.astype(np.int32))))
    cases['loss_soft_prob'] = A(ne.losses.Dice().loss(T(pr_t), T(pr_p)))
    cases['mean_loss_soft_prob'] = A(ne.losses.Dice().mean_loss(T(pr_t), T(pr_p)))
    cases['loss_hard_label'] = A(ne.losses.HardDice(L).loss(T(lab_t.astype(np.int32)), T(lab_p.astype(np.int32))))
    # range assert fires (metrics.py:439-444)
    bad = pr_p.copy()
    bad[0, 0, 0, 0, 0] = 1.5
    try:
        ne.metrics.Dice().dice(T(pr_t), T(bad))
        raised = False
    except tf_shim.InvalidArgumentError:
        raised = True
    assert raised
    cases['not_checked_bad'] = A(ne.metrics.Dice(check_input_limits=False).dice(T(pr_t), T(bad)))
    cases['bad'] = bad
    # float16 probability maps (metrics.py:415-482 is dtype-agnostic: products and quotients in float16, the shim's reduce_sum
    # accumulates wide and rounds once, as TensorFlow's reductions of 16-bit tensors do)
    h_t, h_p = pr_t.astype(np.float16), pr_p.astype(np.float16)
    cases['f16_t'], cases['f16_p'] = h_t, h_p
    cases['f16_soft_prob'] = A(ne.metrics.SoftDice().dice(T(h_t), T(h_p)))
    cases['f16_soft_prob_laplace'] = A(ne.metrics.SoftDice(laplace_smoothing=0.1).dice(T(h_t), T(h_p)))
    cases['f16_mean_soft_prob'] = A(ne.metrics.Dice().mean_dice(T(h_t), T(h_p)))
    cases['f16_hard_prob'] = A(ne.metrics.HardDice(L, input_type='prob').dice(T(h_t), T(h_p)))
    assert cases['f16_soft_prob'].dtype == np.float16
    save('dice_small', **cases)


def gen_cce():
    rng = np.random.default_rng(321)
    cases = {}
    B, S, C = 2, (4, 5, 3), 6
    lab = rng.integers(0, C, (B,) + S)
    t = np.eye(C, dtype=F)[lab]
    p = rng.random((B,) + S + (C,)).astype(F) + F(0.01)
    p[0, 0, 0, 0, :] = 0
    p[0, 0, 0, 0, 2] = 1.0                        # exercises the 1e-7 clip on both sides
    w = (rng.random(C) + 0.2).astype(F)
    sw = rng.random((B,) + S).astype(F)
    z = rng.standard_normal((B,) + S + (C,)).astype(F) * 3
    cases.update(t=t, p=p, w=w, sw=sw, z=z)
    cases['plain'] = A(ne.metrics.CategoricalCrossentropy()(T(t), T(p)))
    cases['weighted'] = A(ne.metrics.CategoricalCrossentropy(label_weights=w).cce(T(t), T(p)))
    cases['weighted_smooth'] = A(ne.metrics.CategoricalCrossentropy(label_weights=w, label_smoothing=0.1)(T(t), T(p)))
    cases['weighted_logits'] = A(ne.metrics.CategoricalCrossentropy(label_weights=w, from_logits=True)(T(t), T(z)))
    cases['weighted_sw'] = A(ne.metrics.CategoricalCrossentropy(label_weights=w)(T(t), T(p), sample_weight=T(sw)))
    cases['loss_weighted'] = A(ne.losses.CategoricalCrossentropy(label_weights=w).loss(T(t), T(p)))
    try:
        ne.metrics.CategoricalCrossentropy(label_weights=w[:-1])(T(t), T(p))
        raised = False
    except ValueError:
        raised = True
    assert raised
    save('cce_small', **cases)


def gen_lc3d():
    rng = np.random.default_rng(2024)
    cases = {}
    for tag, in_shape, ks, st, cout in (('k323_s112', (2, 5, 4, 6, 3), (3, 2, 3), (1, 1, 2), 4),
                                        ('k333_s111', (1, 5, 5, 5, 2), (3, 3, 3), (1, 1, 1), 3)):
        x = rng.standard_normal(in_shape).astype(F)
        osh = tuple((in_shape[1 + d] - ks[d]) // st[d] + 1 for d in range(3))
        O, Fdim = int(np.prod(osh)), int(np.prod(ks)) * in_shape[-1]
        k = (rng.standard_normal((O, Fdim, cout)) / np.sqrt(Fdim)).astype(F)
        b = rng.standard_normal(osh + (cout,)).astype(F)
        out = ne.layers.LocallyConnected3D.local_conv(T(x), T(k), ks, st, osh, 'channels_last')
        out = A(tf_shim.k_bias_add(out, T(b), data_format='channels_last'))
        cases[tag + '__x'] = x
        cases[tag + '__kernel'] = k
        cases[tag + '__bias'] = b
        cases[tag + '__ks'] = np.array(ks)
        cases[tag + '__strides'] = np.array(st)
        cases[tag + '__out'] = out
    save('lc3d_small', **cases)



def gen_lc3d_impl():
    """The reference's LocallyConnected3D LAYER (build + call, neurite/tf/layers.py:912-1102) in its three weight layouts:
    implementation 1 [O, F, Cout] (also channels_first, whose patch flattening is channel-major, :1176-1186), 2 dense-masked
    [in..., Cin, out..., Cout] (:986-1006, 1260-1308) and 3 sparse COO values in sorted (out_idx, in_idx) order (:1008-1028,
    1311-1343), 'valid' and 'same' padding, strides, bias, activation."""
    rng = np.random.default_rng(20260926)
    cases = {}
    specs = [
        ('i1_cl', dict(implementation=1), (2, 5, 4, 6, 3), 4, (3, 2, 3), (1, 1, 2)),
        ('i1_cf', dict(implementation=1, data_format='channels_first'), (2, 3, 5, 4, 6), 4, (3, 2, 3), (1, 1, 2)),
        ('i2_valid', dict(implementation=2), (2, 5, 4, 6, 3), 4, (3, 2, 3), (1, 1, 2)),
        ('i2_same', dict(implementation=2, padding='same'), (2, 5, 4, 6, 2), 4, (3, 3, 3), (1, 1, 1)),
        ('i2_same_stride', dict(implementation=2, padding='same', activation='relu'), (1, 6, 5, 7, 2), 3, (3, 2, 3), (2, 1, 3)),
        ('i2_cf_same', dict(implementation=2, padding='same', data_format='channels_first'), (2, 2, 4, 5, 3), 4, (3, 3, 2), (1, 2, 1)),
        ('i3_valid', dict(implementation=3), (2, 5, 4, 6, 3), 4, (3, 2, 3), (1, 1, 2)),
        ('i3_same', dict(implementation=3, padding='same', activation='elu'), (2, 5, 4, 6, 2), 4, (3, 3, 3), (1, 1, 1)),
        ('i3_same_stride', dict(implementation=3, padding='same'), (1, 6, 5, 7, 2), 3, (2, 3, 3), (2, 1, 3)),
        ('i3_cf_same', dict(implementation=3, padding='same', data_format='channels_first'), (2, 2, 4, 5, 3), 4, (3, 3, 2), (1, 2, 1)),
    ]
    for tag, kw, in_shape, filters, ks, st in specs:
        layer = ne.layers.LocallyConnected3D(filters, ks, strides=st, **kw)
        x = rng.standard_normal(in_shape).astype(F)
        layer(T(x))                                              # builds the weights (zeros in the shim)
        kshape = tuple(A(layer.kernel).shape)
        fan = int(np.prod(ks)) * (in_shape[1] if kw.get('data_format') == 'channels_first' else in_shape[-1])
        layer.kernel.a[...] = (rng.standard_normal(kshape) / np.sqrt(fan)).astype(F)
        layer.bias.a[...] = rng.standard_normal(tuple(A(layer.bias).shape)).astype(F)
        out = A(layer(T(x)))
        cases[tag + '__x'] = x
        cases[tag + '__kernel'] = A(layer.kernel).copy()
        cases[tag + '__bias'] = A(layer.bias).copy()
        cases[tag + '__ks'] = np.array(ks)
        cases[tag + '__strides'] = np.array(st)
        cases[tag + '__filters'] = np.array(filters)
        cases[tag + '__implementation'] = np.array(kw['implementation'])
        cases[tag + '__padding'] = np.array(kw.get('padding', 'valid'))
        cases[tag + '__data_format'] = np.array(kw.get('data_format', 'channels_last'))
        cases[tag + '__activation'] = np.array(kw.get('activation') or 'linear')
        cases[tag + '__out'] = out
        if kw['implementation'] == 3:
            cases[tag + '__kernel_idxs'] = np.asarray(layer.kernel_idxs, np.int64)
    save('lc3d_impl', **cases)


def gen_filter():
    """gaussian_kernel (utils.py:581-662), separable_conv (:665-751), minmax_norm (:953-968), layers.GaussianBlur (:251-364)."""
    rng = np.random.default_rng(11)
    cases = {}
    # gaussian_kernel: joint and separated, explicit window, 'xy' indexing
    for tag, kw in (('gk_iso3', dict(sigma=[1.5, 1.5, 1.5])), ('gk_aniso', dict(sigma=[0.7, 2.0])),
                    ('gk_win', dict(sigma=[1.0, 2.0], windowsize=[5, 4])), ('gk_xy', dict(sigma=[1.0, 2.0], indexing='xy')),
                    ('gk_tiny', dict(sigma=0))):
        k = ne.utils.gaussian_kernel(**kw)
        cases[tag + '__joint'] = A(k)
        ks = ne.utils.gaussian_kernel(separate=True, **kw)
        ks = ks if isinstance(ks, list) else [ks]
        for i, kk in enumerate(ks):
            cases[tag + '__sep%d' % i] = A(kk)
    # separable_conv
    x3 = rng.standard_normal((9, 8, 11, 3)).astype(F)
    xb = rng.standard_normal((2, 7, 10, 6, 2)).astype(F)
    x2 = rng.standard_normal((12, 9, 1)).astype(F)
    k3 = [A(k).astype(F) for k in ne.utils.gaussian_kernel([1.0, 0.6, 1.4], separate=True)]
    k5 = rng.standard_normal(5).astype(F)
    k4 = rng.standard_normal(4).astype(F)
    cases['sc_x3'] = x3; cases['sc_xb'] = xb; cases['sc_x2'] = x2
    cases['sc_k3_0'], cases['sc_k3_1'], cases['sc_k3_2'] = k3
    cases['sc_k5'] = k5; cases['sc_k4'] = k4
    cases['sc_all_axes__out'] = A(ne.utils.separable_conv(T(x3), [T(k) for k in k3]))
    cases['sc_single_kernel__out'] = A(ne.utils.separable_conv(T(x3), T(k5)))
    cases['sc_axis1__out'] = A(ne.utils.separable_conv(T(x3), T(k5), axis=1))
    cases['sc_axes02_valid__out'] = A(ne.utils.separable_conv(T(x3), [T(k5), T(k4)], axis=[0, 2], padding='VALID'))
    cases['sc_even_same__out'] = A(ne.utils.separable_conv(T(x3), T(k4)))
    cases['sc_stride2__out'] = A(ne.utils.separable_conv(T(x3), T(k5), strides=2))
    cases['sc_stride_list__out'] = A(ne.utils.separable_conv(T(x3), [T(k5), T(k4)], axis=[1, 2], strides=[2, 3]))
    cases['sc_dil2__out'] = A(ne.utils.separable_conv(T(x3), T(k5), dilations=2))
    cases['sc_batched__out'] = A(ne.utils.separable_conv(T(xb), [T(k) for k in k3], batched=True))
    cases['sc_2d__out'] = A(ne.utils.separable_conv(T(x2), T(k5)))
    # GaussianBlur layer
    cases['blur_sigma'] = np.array([1.0, 0.0, 2.0], F)
    cases['blur__out'] = A(ne.layers.GaussianBlur(sigma=[1.0, 0.0, 2.0])(T(xb)))
    cases['blur_iso__out'] = A(ne.layers.GaussianBlur(sigma=1.3)(T(xb)))
    # minmax_norm
    cases['mm_all__out'] = A(ne.utils.minmax_norm(T(xb)))
    cases['mm_per_batch__out'] = A(ne.utils.minmax_norm(T(xb), axis=(1, 2, 3, 4)))
    cases['mm_per_batch_feature__out'] = A(ne.utils.minmax_norm(T(xb), axis=(1, 2, 3)))
    cases['mm_const__out'] = A(ne.utils.minmax_norm(T(np.full((3, 4, 2), 2.5, F))))
    save('filter_small', **cases)


def gen_mi():
    """metrics.MutualInformation (metrics.py:41-336) and utils.soft_quantize (utils.py:1099-1172)."""
    import contextlib, io
    rng = np.random.default_rng(13)
    cases = {}
    x = rng.random((2, 6, 5, 7, 1)).astype(F)
    y = (0.7 * x + 0.3 * rng.random(x.shape)).astype(F)
    p3 = rng.random((2, 6, 5, 7, 3)).astype(F)
    q3 = (0.5 * p3 + 0.5 * rng.random(p3.shape)).astype(F)
    p16 = rng.random((2, 6, 5, 7, 16)).astype(F)
    p16 /= p16.sum(-1, keepdims=True)
    q16 = rng.random((2, 6, 5, 7, 16)).astype(F)
    q16 /= q16.sum(-1, keepdims=True)
    cases.update(x=x, y=y, p3=p3, q3=q3, p16=p16, q16=q16)
    with contextlib.redirect_stdout(io.StringIO()):                       # the constructor prints soft_bin_alpha
        mi16 = ne.metrics.MutualInformation()
        mi8 = ne.metrics.MutualInformation(nb_bins=8, min_clip=0.1, max_clip=0.9)
        mia = ne.metrics.MutualInformation(nb_bins=16, soft_bin_alpha=50.0)
    cases['alpha16'] = A(mi16.soft_bin_alpha).astype(F)
    cases['volumes16__out'] = A(mi16.volumes(T(x), T(y)))
    cases['volumes8clip__out'] = A(mi8.volumes(T(x), T(y)))
    cases['volumes_alpha50__out'] = A(mia.volumes(T(x), T(y)))
    cases['channelwise__out'] = A(mi16.channelwise(T(p3), T(q3)))
    cases['segs__out'] = A(mi16.segs(T(p16), T(q16)))
    cases['maps_self__out'] = A(mi16.maps(T(p16), T(p16)))
    cases['volume_seg__out'] = A(mi16.volume_seg(T(x), T(p16)))
    cases['seg_volume__out'] = A(mi16.volume_seg(T(p16), T(y)))
    small = rng.random((3, 4)).astype(F)
    cases['sq_in'] = small
    cases['sq_nb5__out'] = A(ne.utils.soft_quantize(T(small), nb_bins=5, alpha=3.0))
    cases['sq_centers__out'] = A(ne.utils.soft_quantize(T(small), bin_centers=np.array([0.1, 0.5, 0.9], F), nb_bins=None, alpha=2.0))
    cases['sq_log_clip__out'] = A(ne.utils.soft_quantize(T(small), nb_bins=4, alpha=1.5, min_clip=0.2, max_clip=0.8, return_log=True))
    save('mi_small', **cases)



def gen_augment():
    """augment.draw_crop_mask (augment.py:218-290) and utils.subsample_axis (utils.py:754-826) with scripted uniform draws:
    the reference's own control flow decides how many draws are taken and in which order."""
    rng = np.random.default_rng(21)
    cases = {}
    n = 0
    for shape, kw in (((2, 10, 6, 1), dict(crop_min=0.1, crop_max=0.5, axis=(1, 2))),
                      ((2, 10, 6, 1), dict(crop_min=0.3, crop_max=0.3, axis=1)),
                      ((1, 17, 9, 12, 2), dict(crop_min=0.0, crop_max=0.8, axis=None, bilateral=True)),
                      ((3, 32, 2), dict(crop_min=0.2, crop_max=0.6, axis=1, prob=0.5)),
                      ((3, 32, 2), dict(crop_min=0.2, crop_max=0.6, axis=(0, 1, 2), prob=0.5, bilateral=True)),
                      ((1, 64, 64, 1), dict(crop_min=0.05, crop_max=0.95, axis=(1, 2)))):
        for rep in range(4):
            draws = rng.random(6).astype(F)
            tf_shim.RANDOM_SCRIPT[:] = [float(d) for d in draws]
            mask = A(ne.utils.augment.draw_crop_mask(T(np.zeros(shape, F)), seed=1, **kw))
            used = 6 - len(tf_shim.RANDOM_SCRIPT)
            cases['crop%d__shape' % n] = np.asarray(shape)
            cases['crop%d__draws' % n] = draws[:used]
            cases['crop%d__mask' % n] = mask
            for k, v in kw.items():
                cases['crop%d__%s' % (n, k)] = np.asarray(-1 if v is None else v)
            n += 1
    cases['ncrop'] = np.asarray(n)
    n = 0
    for shape, kw in (((2, 12, 3), dict(stride_min=2, stride_max=5, axes=(1,))),
                      ((1, 33, 20, 2), dict(stride_min=1, stride_max=8, axes=(1, 2))),
                      ((1, 33, 20, 2), dict(stride_min=1, stride_max=8, axes=(1, 2), prob=0.5)),
                      ((2, 16, 16, 16, 1), dict(stride_min=3, stride_max=3, axes=(1, 2, 3), upsample=False)),
                      ((1, 160, 1), dict(stride_min=1.5, stride_max=6.5, axes=1))):
        for rep in range(4):
            draws = rng.random(4).astype(F)
            tf_shim.RANDOM_SCRIPT[:] = [float(d) for d in draws]
            x = rng.standard_normal(shape).astype(F)
            y = A(ne.utils.subsample_axis(T(x), seed=1, **kw))
            used = 4 - len(tf_shim.RANDOM_SCRIPT)
            cases['sub%d__x' % n] = x
            cases['sub%d__draws' % n] = draws[:used]
            cases['sub%d__out' % n] = y
            for k, v in kw.items():
                cases['sub%d__%s' % (n, k)] = np.asarray(v)
            n += 1
    cases['nsub'] = np.asarray(n)
    # gaussian_kernel(random=True) (utils.py:643-650): one uniform draw per axis for the SD
    n = 0
    for kw in (dict(sigma=[2.0, 3.0, 1.0], min_sigma=[0.5, 0.5, 0.5]), dict(sigma=4.0, min_sigma=1.0), dict(sigma=[1.5, 1.5], min_sigma=0)):
        for rep in range(3):
            nsig = len(kw['sigma']) if isinstance(kw['sigma'], list) else 1
            draws = rng.random(nsig).astype(F)
            tf_shim.RANDOM_SCRIPT[:] = [float(d) for d in draws]
            ks = ne.utils.gaussian_kernel(separate=True, random=True, seed=5, **kw)
            ks = ks if isinstance(ks, list) else [ks]
            cases['gk%d__draws' % n] = draws
            cases['gk%d__sigma' % n] = np.asarray(kw['sigma'], F)
            cases['gk%d__min_sigma' % n] = np.asarray(kw['min_sigma'], F)
            for i, k in enumerate(ks):
                cases['gk%d__k%d' % (n, i)] = A(k)
            n += 1
    cases['ngk'] = np.asarray(n)
    save('augment_small', **cases)


# builder name, positional arguments, keyword arguments: the cases of tests/golden/unet_graph.json
UNET_GRAPH_CASES = [
    ('cfg3', 'unet', [16, [160, 160, 160, 1], 3, 3, 32], dict(feat_mult=2)),
    ('cfg3_2conv', 'unet', [16, [160, 160, 160, 1], 3, 3, 32], dict(feat_mult=2, nb_conv_per_level=2)),
    ('res_dil', 'unet', [8, [32, 32, 32, 2], 3, 3, 4], dict(feat_mult=2, use_residuals=True, dilation_rate_mult=2,
                                                          nb_conv_per_level=2)),
    ('res_dil_1conv', 'unet', [8, [32, 32, 32, 2], 3, 3, 4], dict(feat_mult=2, use_residuals=True, dilation_rate_mult=2)),
    ('res_same_feats', 'unet', [8, [16, 16, 16, 8], 2, 3, 3], dict(use_residuals=True, nb_conv_per_level=2)),
    ('layer_nb_feats', 'unet', [4, [16, 16, 16, 1], 2, 3, 3],
     dict(nb_conv_per_level=2, layer_nb_feats=[3, 5, 7, 9, 11, 13])),
    ('list_of_lists', 'unet', [[[4, 6], [8], [10, 12, 14]], [24, 24, 24, 1], None, 3, 5], dict(feat_mult=None)),
    ('bn_dropout_res', 'unet', [6, [16, 16, 16, 3], 3, 3, 4], dict(feat_mult=2, use_residuals=True, nb_conv_per_level=2,
                                                                conv_dropout=0.25, batch_norm=-1)),
    ('dropout_plain', 'unet', [6, [16, 16, 16, 3], 2, 3, 4], dict(conv_dropout=0.5)),
    ('prior_logp', 'unet', [4, [16, 16, 16, 1], 2, 3, 5], dict(add_prior_layer=True)),
    ('prior_p', 'unet', [4, [16, 16, 16, 1], 2, 3, 5], dict(add_prior_layer=True, use_logp=False,
                                                           final_pred_activation='linear')),
    ('multi_input', 'unet', [4, [[16, 16, 16, 1], [16, 16, 16, 2]], 2, 3, 3], dict()),
    ('two_d_pool', 'unet', [4, [32, 54, 2], 3, [3, 5], 3], dict(pool_size=[2, 3], activation='relu',
                                                               final_pred_activation='linear')),
    ('one_d', 'unet', [4, [64, 1], 3, 3, 2], dict(final_pred_activation=None)),
    ('valid_enc', 'conv_enc', [4, [30, 30, 30, 1], 2, 3], dict(padding='valid', name='enc')),
    ('enc_default', 'conv_enc', [8, [16, 16, 16, 2], 3, 3], dict(name='enc', feat_mult=2)),
    ('enc_res_dil', 'conv_enc', [8, [32, 32, 32, 2], 3, 3], dict(name='enc', feat_mult=2, use_residuals=True,
                                                                dilation_rate_mult=2)),
    ('dec_alone', 'conv_dec', [8, [4, 4, 4, 16], 3, 3, 5], dict(name='dec', feat_mult=2)),
    ('dec_alone_res', 'conv_dec', [8, [4, 4, 4, 16], 3, 3, 5], dict(name='dec', feat_mult=2, use_residuals=True,
                                                                   dilation_rate_mult=3)),
    ('dilation_net', 'dilation_net', [8, [32, 32, 32, 1], 3, 3, 4], dict(dilation_rate_mult=2, feat_mult=2,
                                                                         use_residuals=True, nb_conv_per_level=2)),
]


def gen_unet_graph():
    """The layer graphs the reference's own builders construct (neurite/tf/models.py:45-246, 378-436, 1309-1617), recorded by
    tests/golden/keras_record.py: per layer its name, Keras class, constructor arguments, input producers, output shape."""
    import contextlib
    import io
    import json
    import warnings
    import keras_record
    graphs = {}
    for tag, builder, args, kwargs in UNET_GRAPH_CASES:
        keras_record.reset()
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            warnings.simplefilter('ignore')
            model = getattr(ne.models, builder)(*args, **kwargs)
        graphs[tag] = {'builder': builder, 'args': args, 'kwargs': kwargs, 'graph': model.graph()}
    path = os.path.join(HERE, 'unet_graph.json')
    with open(path, 'w') as f:
        json.dump(graphs, f, indent=1, sort_keys=True)
        f.write('\n')
    print('%-28s %7.1f KB  (%d graphs)' % ('unet_graph.json', os.path.getsize(path) / 1024, len(graphs)))


if __name__ == '__main__':
    gen_unet_graph()
    gen_mi()
    gen_filter()
    gen_interpn()
    gen_interpn_dtypes()
    gen_resize()
    gen_index_helpers()
    gen_dice()
    gen_cce()
    gen_lc3d()
    gen_lc3d_impl()
    gen_augment()
