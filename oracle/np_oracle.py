"""
NumPy restatement of the neurite hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference lines it follows (paths relative to
/root/reference).  Arithmetic is done in the dtype the TensorFlow graph would use
(float32 unless the caller passes something else), one NumPy op per TF op, in the
same order, so results of the float paths are reproducible bit-for-bit by any
IEEE-754 implementation that performs the same sequence of roundings.

Reductions (Dice sums, CCE mean) have no defined order in TF; they are accumulated
in float64 here and the tests compare with a relative tolerance of 1e-5, except for
one-hot / label inputs where all partial sums are small integers and any order is
exact.
"""

import itertools

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# index helpers
# --------------------------------------------------------------------------------------

def sub2ind2d(siz, subs):
    """neurite/tf/utils/utils.py:1068-1082 -- row-major flat index (despite the docstring)."""
    assert len(siz) == len(subs), 'found inconsistent siz and subs: %d %d' % (len(siz), len(subs))
    k = np.cumprod(siz[::-1])
    ndx = subs[-1]
    for i, v in enumerate(subs[:-1][::-1]):
        ndx = ndx + v * k[i]
    return ndx


def prod_n(lst):
    """neurite/tf/utils/utils.py:1085-1092 -- sequential left-to-right product."""
    prod = lst[0]
    for p in lst[1:]:
        prod = prod * p
    return prod


def meshgrid(*args, indexing='xy'):
    """neurite/tf/utils/utils.py:398-476 (tile-based meshgrid; 'xy' swaps the first two axes)."""
    if indexing not in ('xy', 'ij'):
        raise ValueError("indexing parameter must be either 'xy' or 'ij'")
    return [np.ascontiguousarray(g) for g in np.meshgrid(*args, indexing=indexing)]


def ndgrid(*args):
    """neurite/tf/utils/utils.py:382-395."""
    return meshgrid(*args, indexing='ij')


def volshape_to_ndgrid(volshape):
    """neurite/tf/utils/utils.py:333-353."""
    if not all(float(d).is_integer() for d in volshape):
        raise ValueError('volshape needs to be a list of integers')
    return ndgrid(*[np.arange(0, d, dtype=np.int32) for d in volshape])


def volshape_to_meshgrid(volshape, indexing='xy'):
    """neurite/tf/utils/utils.py:356-379."""
    if not all(float(d).is_integer() for d in volshape):
        raise ValueError('volshape needs to be a list of integers')
    return meshgrid(*[np.arange(0, d, dtype=np.int32) for d in volshape], indexing=indexing)


def batch_channel_flatten(x):
    """neurite/tf/utils/utils.py:1175-1226 -- [B, ..., C] -> [B, V, C] (a view)."""
    return x.reshape(x.shape[0], -1, x.shape[-1])


# --------------------------------------------------------------------------------------
# interpn  (neurite/tf/utils/utils.py:73-220)
# --------------------------------------------------------------------------------------

def interpn(vol, loc, interp_method='linear', fill_value=None):
    """
    Op-for-op restatement of neurite/tf/utils/utils.py:73-220.

    vol: [*S] or [*S, C]; loc: list of D arrays or [*S', D] array.
    """
    vol = np.asarray(vol)
    if isinstance(loc, (list, tuple)):
        loc = np.stack(loc, -1)                                            # :106-107
    loc = np.asarray(loc)
    nb_dims = loc.shape[-1]                                                # :108
    input_vol_shape = vol.shape

    if vol.ndim not in [nb_dims, nb_dims + 1]:                             # :111-113
        raise Exception("Number of loc Tensors %d does not match volume dimension %d"
                        % (nb_dims, len(vol.shape[:-1])))
    if nb_dims > vol.ndim:                                                 # :115-117
        raise Exception("Loc dimension %d does not match volume dimension %d"
                        % (nb_dims, vol.ndim))
    if vol.ndim == nb_dims:                                                # :119-120
        vol = vol[..., None]

    vol_floating = np.issubdtype(vol.dtype, np.floating)
    if not np.issubdtype(loc.dtype, np.floating):                          # :123-125
        loc = loc.astype(vol.dtype if vol_floating else np.float32)
    elif vol_floating and vol.dtype != loc.dtype:                          # :126-127
        loc = loc.astype(vol.dtype)
    ldt = loc.dtype.type

    volshape = vol.shape
    max_loc = [d - 1 for d in vol.shape]                                   # :134
    vol_reshape = vol.reshape(-1, volshape[-1])                            # :177

    if interp_method == 'linear':
        loc0 = np.floor(loc)                                               # :139
        clipped_loc = [np.clip(loc[..., d], ldt(0), ldt(max_loc[d])) for d in range(nb_dims)]   # :142
        loc0lst = [np.clip(loc0[..., d], ldt(0), ldt(max_loc[d])) for d in range(nb_dims)]      # :143
        loc1 = [np.clip(loc0lst[d] + ldt(1), ldt(0), ldt(max_loc[d])) for d in range(nb_dims)]  # :146
        locs = [[f.astype(np.int32) for f in loc0lst], [f.astype(np.int32) for f in loc1]]      # :147
        diff_loc1 = [loc1[d] - clipped_loc[d] for d in range(nb_dims)]     # :152
        diff_loc0 = [ldt(1) - d for d in diff_loc1]                        # :153
        weights_loc = [diff_loc1, diff_loc0]                               # :155

        cube_pts = list(itertools.product([0, 1], repeat=nb_dims))         # :159
        interp_vol = 0                                                     # :160
        for c in cube_pts:                                                 # :162
            subs = [locs[c[d]][d] for d in range(nb_dims)]                 # :170
            idx = sub2ind2d(vol.shape[:-1], subs)                          # :176
            vol_val = vol_reshape[idx]                                     # :178
            wts_lst = [weights_loc[c[d]][d] for d in range(nb_dims)]       # :183
            wt = prod_n(wts_lst)[..., None]                                # :187-188
            interp_vol = interp_vol + wt * vol_val                         # :191
    else:
        assert interp_method == 'nearest', \
            'method should be linear or nearest, got: %s' % interp_method  # :194-195
        roundloc = np.rint(loc).astype(np.int32)                           # :196  (tf.round = half-to-even)
        roundloc = [np.clip(roundloc[..., d], 0, max_loc[d]) for d in range(nb_dims)]   # :197
        idx = sub2ind2d(vol.shape[:-1], roundloc)                          # :203
        interp_vol = vol_reshape[idx]                                      # :204

    if fill_value is not None:                                             # :206-213
        out_type = interp_vol.dtype
        fill = np.asarray(fill_value).astype(out_type)
        below = [loc[..., d] < 0 for d in range(nb_dims)]
        above = [loc[..., d] > max_loc[d] for d in range(nb_dims)]
        oob = np.any(np.stack(below + above, axis=-1), axis=-1, keepdims=True)
        with np.errstate(invalid='ignore'):
            interp_vol = interp_vol * np.logical_not(oob).astype(out_type)
            interp_vol = interp_vol + oob.astype(out_type) * fill

    if len(input_vol_shape) == nb_dims:                                    # :216-218
        assert interp_vol.shape[-1] == 1, 'Something went wrong with interpn channels'
        interp_vol = interp_vol[..., 0]
    return interp_vol



# --------------------------------------------------------------------------------------
# interpn in a narrow float type that NumPy does not have (bfloat16)
# --------------------------------------------------------------------------------------

def round_bf16(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (the value set TF's bfloat16 tensors hold)"""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32)
    nan = (u & np.uint32(0x7fffffff)) > np.uint32(0x7f800000)
    r = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    r = np.where(nan, (u & np.uint32(0xffff0000)) | np.uint32(0x00400000), r).astype(np.uint32)
    return r.view(np.float32)


def round_f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def interpn_emulated(vol, loc, interp_method='linear', fill_value=None, rnd=round_bf16):
    """
    neurite/tf/utils/utils.py:73-220 for a volume of a narrow float type T, emulated on float32 arrays that only ever hold
    values of T: `rnd` (float32 -> nearest T, as float32) is applied to the inputs (tf.cast of loc to the volume dtype,
    :123-127) and after EVERY arithmetic op -- which is how TensorFlow evaluates half / bfloat16 element-wise ops (compute in
    float, round the result; exact for +, -, * because 24 >= 2p + 2).  Same op sequence as `interpn` above; with
    rnd = round_f16 it reproduces `interpn` on float16 arrays bit for bit and with rnd = identity the float32 path
    (tests/test_oracle.py), which is what vouches for the bfloat16 use.
    vol [*S, C] (values of T), loc [*S', D].
    """
    vol = rnd(np.asarray(vol, np.float32))
    if isinstance(loc, (list, tuple)):
        loc = np.stack(loc, -1)
    loc = rnd(np.asarray(loc, np.float32))                                 # :126-127 cast to the volume dtype
    nb_dims = loc.shape[-1]
    assert vol.ndim == nb_dims + 1
    f = np.float32
    max_loc = [rnd(f(d - 1)) for d in vol.shape[:-1]]                     # python ints become tensors of dtype T inside clip
    vol_reshape = vol.reshape(-1, vol.shape[-1])
    if interp_method == 'linear':
        loc0 = np.floor(loc)
        clipped = [np.clip(loc[..., d], f(0), max_loc[d]) for d in range(nb_dims)]
        loc0lst = [np.clip(loc0[..., d], f(0), max_loc[d]) for d in range(nb_dims)]
        loc1 = [np.clip(rnd(loc0lst[d] + f(1)), f(0), max_loc[d]) for d in range(nb_dims)]
        locs = [[np.clip(x.astype(np.int32), 0, vol.shape[d] - 1) for d, x in enumerate(loc0lst)],
                [np.clip(x.astype(np.int32), 0, vol.shape[d] - 1) for d, x in enumerate(loc1)]]
        diff_loc1 = [rnd(loc1[d] - clipped[d]) for d in range(nb_dims)]
        diff_loc0 = [rnd(f(1) - d) for d in diff_loc1]
        weights_loc = [diff_loc1, diff_loc0]
        interp_vol = f(0)
        for c in itertools.product([0, 1], repeat=nb_dims):
            subs = [locs[c[d]][d] for d in range(nb_dims)]
            idx = sub2ind2d(vol.shape[:-1], subs)
            vol_val = vol_reshape[idx]
            wt = weights_loc[c[0]][0]
            for d in range(1, nb_dims):
                wt = rnd(wt * weights_loc[c[d]][d])
            interp_vol = rnd(interp_vol + rnd(wt[..., None] * vol_val))
    else:
        assert interp_method == 'nearest'
        with np.errstate(invalid='ignore'):
            roundloc = np.rint(loc).astype(np.int32)
        roundloc = [np.clip(roundloc[..., d], 0, vol.shape[d] - 1) for d in range(nb_dims)]
        interp_vol = vol_reshape[sub2ind2d(vol.shape[:-1], roundloc)]
    if fill_value is not None:
        fill = rnd(f(fill_value))
        below = [loc[..., d] < 0 for d in range(nb_dims)]
        above = [loc[..., d] > max_loc[d] for d in range(nb_dims)]
        oob = np.any(np.stack(below + above, axis=-1), axis=-1, keepdims=True)
        with np.errstate(invalid='ignore'):
            interp_vol = rnd(interp_vol * np.logical_not(oob).astype(np.float32))
            interp_vol = rnd(interp_vol + rnd(oob.astype(np.float32) * fill))
    return interp_vol


def interpn_f64(vol, loc, fill_value=None):
    """Independent float64 'truth' for the linear path (same clamping rules, exact weights)."""
    vol = np.asarray(vol, np.float64)
    squeeze = False
    loc = np.asarray(loc, np.float64)
    D = loc.shape[-1]
    if vol.ndim == D:
        vol = vol[..., None]
        squeeze = True
    S = vol.shape[:-1]
    out = np.zeros(loc.shape[:-1] + (vol.shape[-1],))
    i0, i1, w0 = [], [], []
    for d in range(D):
        m = S[d] - 1
        cl = np.clip(loc[..., d], 0, m)
        l0 = np.clip(np.floor(loc[..., d]), 0, m)
        l1 = np.clip(l0 + 1, 0, m)
        i0.append(l0.astype(np.int64)); i1.append(l1.astype(np.int64)); w0.append(l1 - cl)
    for c in itertools.product([0, 1], repeat=D):
        idx = tuple((i1 if c[d] else i0)[d] for d in range(D))
        wt = np.ones(loc.shape[:-1])
        for d in range(D):
            wt = wt * ((1 - w0[d]) if c[d] else w0[d])
        out += wt[..., None] * vol[idx]
    if fill_value is not None:
        oob = np.zeros(loc.shape[:-1], bool)
        for d in range(D):
            oob |= (loc[..., d] < 0) | (loc[..., d] > S[d] - 1)
        out = np.where(oob[..., None], float(fill_value), out)
    return out[..., 0] if squeeze else out


# --------------------------------------------------------------------------------------
# resize / zoom  (neurite/tf/utils/utils.py:223-265)
# --------------------------------------------------------------------------------------

def tf_linspace(start, stop, num, dtype=np.float32):
    """
    tf.linspace(start, stop, num) as TF2 computes it (TF semantics, restated):
    delta = (stop-start)/(num-1) in dtype; x_i = start + delta*i for 0<i<num-1;
    first element = start, last element = stop exactly; num==1 -> [start].
    """
    t = np.dtype(dtype).type
    start, stop = t(start), t(stop)
    if num == 1:
        return np.array([start], dtype)
    delta = t((stop - start) / t(num - 1))
    i = np.arange(1, num - 1).astype(dtype)
    mid = (start + delta * i).astype(dtype)
    return np.concatenate([[start], mid, [stop]]).astype(dtype)


def resize_new_shape(vol_shape, zoom_factor):
    """neurite/tf/utils/utils.py:256-257 (python float multiply, int() truncation)."""
    return [int(vol_shape[f] * zoom_factor[f]) for f in range(len(zoom_factor))]


def resize(vol, zoom_factor, interp_method='linear'):
    """neurite/tf/utils/utils.py:223-262."""
    vol = np.asarray(vol)
    if isinstance(zoom_factor, (list, tuple)):                             # :237-242
        ndims = len(zoom_factor)
        vol_shape = vol.shape[:ndims]
        assert len(vol_shape) in (ndims, ndims + 1), \
            "zoom_factor length %d does not match ndims %d" % (len(vol_shape), ndims)
    else:                                                                  # :244-247
        vol_shape = vol.shape[:-1]
        ndims = len(vol_shape)
        zoom_factor = [zoom_factor] * ndims
    if all(z == 1 for z in zoom_factor):                                   # :250-251
        return vol
    new_shape = resize_new_shape(vol_shape, zoom_factor)                   # :256-257
    lin = [tf_linspace(0., vol_shape[d] - 1., new_shape[d]) for d in range(ndims)]   # :259
    grid = ndgrid(*lin)                                                    # :260
    return interpn(vol, grid, interp_method=interp_method)                 # :262


zoom = resize


def resize_layer(x, zoom_factor, interp_method='linear'):
    """neurite/tf/layers.py:154-181 -- Resize.call: map utils.resize over the batch axis."""
    x = np.asarray(x)
    ndims = x.ndim - 2
    if not isinstance(zoom_factor, (list, tuple)):                         # layers.py:142-147
        zoom_factor = [zoom_factor] * ndims
    else:
        assert len(zoom_factor) == ndims, \
            'zoom factor length {} does not match number of dimensions {}'.format(len(zoom_factor), ndims)
    return np.stack([resize(x[b], list(zoom_factor), interp_method) for b in range(x.shape[0])], 0)


# --------------------------------------------------------------------------------------
# SpatialTransformer / transform  (voxelmorph, not in tree; call sites
# neurite/tf/models.py:806-807, 1157-1159; behaviour per SURVEY.md A.4)
# --------------------------------------------------------------------------------------

def transform(vol, loc_shift, interp_method='linear', indexing='ij', fill_value=None):
    """vxm.utils.transform: loc = cast(meshgrid, shift.dtype) + shift; interpn(vol, loc)."""
    loc_shift = np.asarray(loc_shift)
    volshape = loc_shift.shape[:-1]
    nb_dims = len(volshape)
    mesh = volshape_to_meshgrid(volshape, indexing=indexing)
    loc = [mesh[d].astype(loc_shift.dtype) + loc_shift[..., d] for d in range(nb_dims)]
    return interpn(vol, loc, interp_method=interp_method, fill_value=fill_value)


def affine_to_dense_shift(matrix, shape, shift_center=True, indexing='ij'):
    """
    vxm.utils.affine_to_dense_shift (believed semantics): shift[q] = A [q-c;1] - (q-c).
    matrix: [D, D+1] (or [D+1, D+1], last row dropped).  float32 throughout.
    """
    matrix = np.asarray(matrix, F32)
    D = len(shape)
    if matrix.shape[-2] == D + 1:
        matrix = matrix[:D]
    mesh = volshape_to_meshgrid(shape, indexing=indexing)
    mesh = [m.astype(F32) for m in mesh]
    if shift_center:
        mesh = [mesh[d] - F32((shape[d] - 1) / 2) for d in range(D)]
    flat = [m.reshape(-1) for m in mesh]
    flat.append(np.ones(flat[0].shape, F32))
    mesh_matrix = np.stack(flat, 1).T                       # [D+1, V]
    loc_matrix = (matrix @ mesh_matrix).astype(F32)         # [D, V]
    loc = loc_matrix.T.reshape(list(shape) + [D])
    return loc - np.stack(mesh, -1)


def spatial_transformer(vol, trf, interp_method='linear', indexing='ij', single_transform=False,
                        fill_value=None, shift_center=True):
    """vxm.layers.SpatialTransformer.call on [vol [B,*S,C], trf [B,*S',D] | affine [B,D,D+1]]."""
    vol = np.asarray(vol)
    trf = np.asarray(trf)
    D = vol.ndim - 2
    outs = []
    for b in range(vol.shape[0]):
        t = trf[0] if single_transform else trf[b]
        if t.ndim == 2 and t.shape[-1] == D + 1 and t.shape[0] in (D, D + 1):    # affine (a dense 1-D flow [X, 1] is also 2-D)
            t = affine_to_dense_shift(t, vol.shape[1:-1], shift_center=shift_center, indexing=indexing)
        elif indexing == 'xy':
            # voxelmorph swaps the first two displacement components for 'xy' flows
            t = np.concatenate([t[..., 1:2], t[..., 0:1], t[..., 2:]], -1)
        outs.append(transform(vol[b], t, interp_method=interp_method, indexing='ij',
                              fill_value=fill_value))
    return np.stack(outs, 0)


# --------------------------------------------------------------------------------------
# voxelmorph companions built on transform() (call sites neurite/tf/models.py:802-804, 1131, 1149-1154).
# voxelmorph is not vendored: published semantics restated -- parity unpinned, like SpatialTransformer.
# --------------------------------------------------------------------------------------

def integrate_vec(vec, method='ss', nb_steps=7):
    """vxm.utils.integrate_vec, stationary field [*S, D] float32."""
    vec = np.asarray(vec, F32)
    if method in ('ss', 'scaling_and_squaring'):
        vec = (vec / F32(2 ** nb_steps)).astype(F32)
        for _ in range(nb_steps):
            vec = (vec + transform(vec, vec)).astype(F32)
        return vec
    assert method == 'quadrature'
    vec = (vec / F32(nb_steps)).astype(F32)
    disp = vec
    for _ in range(nb_steps - 1):
        disp = (disp + transform(vec, disp)).astype(F32)
    return disp


def rescale_dense_transform(trf, factor, interp_method='linear'):
    """vxm.utils.rescale_dense_transform for one field [*S, D]."""
    trf = np.asarray(trf, F32)
    if factor < 1:
        return (resize(trf, factor, interp_method) * F32(factor)).astype(F32)
    return resize((trf * F32(factor)).astype(F32), factor, interp_method)


def is_affine_shape(shape):
    return len(shape) == 2 and shape[-1] != 1


def make_square_affine(mat):
    mat = np.asarray(mat, F32)
    if mat.shape[-2] == mat.shape[-1]:
        return mat
    row = np.zeros(mat.shape[:-2] + (1, mat.shape[-1]), F32)
    row[..., -1] = 1
    return np.concatenate([mat, row], -2)


def compose(transforms, interp_method='linear', shift_center=True):
    """vxm.utils.compose ('ij' indexing), un-batched."""
    curr = np.asarray(transforms[-1], F32)
    for nxt in reversed(transforms[:-1]):
        nxt = np.asarray(nxt, F32)
        dense = next((t for t in (nxt, curr) if not is_affine_shape(t.shape)), None)
        if dense is not None:
            shape = dense.shape[:-1]
            if is_affine_shape(nxt.shape):
                nxt = affine_to_dense_shift(nxt, shape, shift_center=shift_center)
            if is_affine_shape(curr.shape):
                curr = affine_to_dense_shift(curr, shape, shift_center=shift_center)
            curr = (curr + transform(nxt, curr, interp_method=interp_method)).astype(F32)
        else:
            curr = (make_square_affine(nxt) @ make_square_affine(curr))[:-1].astype(F32)
    return curr


# --------------------------------------------------------------------------------------
# Dice  (neurite/tf/metrics.py:415-510, neurite/tf/losses.py:68-95)
# --------------------------------------------------------------------------------------

def divide_no_nan(x, y):
    """tf.math.divide_no_nan: 0 where y == 0."""
    x = np.asarray(x)
    y = np.asarray(y)
    out = np.zeros(np.broadcast(x, y).shape, dtype=np.result_type(x, y))
    np.divide(x, y, out=out, where=(y != 0))
    return out


def one_hot(idx, nb_labels, dtype=F32):
    """tf.one_hot: out-of-range index -> all-zero row."""
    idx = np.asarray(idx)
    out = np.zeros(idx.shape + (nb_labels,), dtype)
    ok = (idx >= 0) & (idx < nb_labels)
    np.put_along_axis(out, np.where(ok, idx, 0)[..., None].astype(np.int64), ok[..., None].astype(dtype), -1)
    return out


def dice_sums(y_true, y_pred):
    """Σ_v t*p, Σ_v t², Σ_v p² per (batch,label), accumulated in float64. metrics.py:471-477."""
    t = batch_channel_flatten(np.asarray(y_true)).astype(np.float64)
    p = batch_channel_flatten(np.asarray(y_pred)).astype(np.float64)
    return (t * p).sum(1), (t * t).sum(1), (p * p).sum(1)


def dice(y_true, y_pred, dice_type='soft', input_type='prob', nb_labels=None,
         laplace_smoothing=0., normalize=False, check_input_limits=True):
    """neurite/tf/metrics.py:415-482.  Returns float32 [B, L]."""
    y_true = np.asarray(y_true)
    y_pred = np.asarray(y_pred)
    if input_type in ['prob', 'one_hot']:
        if normalize:                                                       # :434-436
            y_true = divide_no_nan(y_true, y_true.sum(-1, keepdims=True))
            y_pred = divide_no_nan(y_pred, y_pred.sum(-1, keepdims=True))
        if check_input_limits:                                              # :439-444
            for a in (y_true, y_pred):
                if not (a.min() >= 0. and a.max() <= 1.):
                    raise ValueError('value outside range')
    if dice_type == 'hard':                                                 # :450-468
        if input_type == 'prob':
            if nb_labels is None:
                nb_labels = y_pred.shape[-1]
            y_pred = np.argmax(y_pred, -1)
            y_true = np.argmax(y_true, -1)
        y_pred = one_hot(np.asarray(y_pred).astype(np.int64), nb_labels)
        y_true = one_hot(np.asarray(y_true).astype(np.int64), nb_labels)
    stp, stt, spp = dice_sums(y_true, y_pred)                               # :471-477
    top = (2 * stp).astype(F32)
    bottom = (stt.astype(F32) + spp.astype(F32)).astype(F32)
    if laplace_smoothing > 0:                                               # :478-480
        eps = F32(laplace_smoothing)
        return ((top + eps) / (bottom + eps)).astype(F32)
    return divide_no_nan(top, bottom).astype(F32)                           # :482


def mean_dice(y_true, y_pred, weights=None, **kw):
    """neurite/tf/metrics.py:484-510."""
    d = dice(y_true, y_pred, **kw)
    if weights is not None:
        weights = np.asarray(weights)
        assert weights.ndim == 2, 'weights should be a matrix broadcastable to [batch_size, nb_labels]'
        d = d * weights.astype(F32)
    m = np.mean(d.astype(np.float64)).astype(F32)
    assert np.isfinite(m), 'metric not finite'
    return m


# --------------------------------------------------------------------------------------
# label-weighted categorical cross-entropy (neurite/tf/metrics.py:619-650 + Keras CCE)
# --------------------------------------------------------------------------------------

KERAS_EPSILON = 1e-7


def cce_per_voxel(y_true, y_pred, label_weights=None, from_logits=False, label_smoothing=0.):
    """
    Per-element loss [B, *S] in float64 from inputs first rounded to float32 (bf16 callers pass
    values already representable in bf16).  metrics.py:641-648 then Keras
    categorical_crossentropy (TF semantics restated: smoothing, p/sum(p), clip, -sum t log p).
    """
    p = np.asarray(y_pred).astype(np.float64)
    t = np.asarray(y_true).astype(np.float64)
    C = p.shape[-1]
    if label_weights is not None:
        w = np.asarray(label_weights)
        if w.shape[-1] != C:
            raise ValueError(f'Label weights must be of len {C}, but got {w.shape[-1]}.')   # :644-645
        t = w.astype(np.float64) * t                                                         # :648
    if label_smoothing:
        t = t * (1.0 - label_smoothing) + (label_smoothing / C)
    if from_logits:
        z = p - p.max(-1, keepdims=True)
        logq = z - np.log(np.exp(z).sum(-1, keepdims=True))
    else:
        q = p / p.sum(-1, keepdims=True)
        q = np.clip(q, KERAS_EPSILON, 1. - KERAS_EPSILON)
        logq = np.log(q)
    return -(t * logq).sum(-1)


def cce(y_true, y_pred, label_weights=None, sample_weight=None, **kw):
    """Scalar loss, reduction SUM_OVER_BATCH_SIZE (mean over all B*V elements)."""
    l = cce_per_voxel(y_true, y_pred, label_weights, **kw)
    if sample_weight is not None:
        sw = np.asarray(sample_weight, np.float64)
        while sw.ndim < l.ndim:
            sw = sw[..., None]
        l = l * sw
    return np.float32(l.sum() / l.size)


# --------------------------------------------------------------------------------------
# LocallyConnected3D implementation 1 (neurite/tf/layers.py:951-1047, 1072-1102, 1126-1197)
# --------------------------------------------------------------------------------------

def conv_output_length(input_length, filter_size, padding, stride, dilation=1):
    """keras conv_utils.conv_output_length (TF semantics)."""
    if input_length is None:
        return None
    dilated = filter_size + (filter_size - 1) * (dilation - 1)
    if padding in ('same', 'causal'):
        out = input_length
    elif padding == 'valid':
        out = input_length - dilated + 1
    elif padding == 'full':
        out = input_length + dilated - 1
    else:
        raise ValueError(padding)
    return (out + stride - 1) // stride


def lc3d(x, kernel, bias=None, kernel_size=(3, 3, 3), strides=(1, 1, 1), accumulate=np.float64):
    """
    x [B,R,C,Z,Cin]; kernel [O, kr*kc*kz*Cin, Cout]; bias [or,oc,oz,Cout] or None.
    Patch flatten order (kr,kc,kz,cin) row-major (layers.py:1179-1186), output positions
    row-major (:1172-1173), out[b,o,:] = patch[b,o,:] @ kernel[o] (:1189), + bias (:1098-1099).
    """
    x = np.asarray(x)
    B, R, Cc, Z, Cin = x.shape
    kr, kc, kz = kernel_size
    sr, sc, sz = strides
    orr = conv_output_length(R, kr, 'valid', sr)
    occ = conv_output_length(Cc, kc, 'valid', sc)
    ozz = conv_output_length(Z, kz, 'valid', sz)
    Cout = kernel.shape[-1]
    out = np.zeros((B, orr, occ, ozz, Cout), accumulate)
    o = 0
    for r in range(orr):
        for c in range(occ):
            for z in range(ozz):
                patch = x[:, r * sr:r * sr + kr, c * sc:c * sc + kc, z * sz:z * sz + kz, :]
                patch = patch.reshape(B, -1).astype(accumulate)
                out[:, r, c, z, :] = patch @ kernel[o].astype(accumulate)
                o += 1
    if bias is not None:
        out = out + np.asarray(bias).astype(accumulate)[None]
    return out


# --------------------------------------------------------------------------------------
# Keras layers the unet instantiates (TF semantics, restated; used by the unet oracle)
# --------------------------------------------------------------------------------------

def elu(x):
    """Keras ELU alpha=1: x>0 ? x : exp(x)-1."""
    x = np.asarray(x)
    return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1).astype(x.dtype)


def softmax_lastdim(x):
    x = np.asarray(x, np.float64)
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


# --------------------------------------------------------------------------------------
# filtering / normalisation of the synthesis front-end (neurite/tf/utils/utils.py:581-751, 953-968;
# neurite/tf/layers.py:251-364; neurite/tf/utils/augment.py:7-62)
# --------------------------------------------------------------------------------------

def gaussian_kernel(sigma, windowsize=None, indexing='ij', separate=False, dtype=F32):
    """utils.py:581-662 (non-random): exp(-x^2 / (2 s^2)) on a centred grid, normalised to sum 1."""
    if not isinstance(sigma, (list, tuple)):
        sigma = [sigma]
    sigma = [max(f, np.finfo(dtype).eps) for f in sigma]                     # :628
    if windowsize is None:
        windowsize = [np.round(f * 3) * 2 + 1 for f in sigma]                # :633
    if not isinstance(windowsize, (list, tuple)):
        windowsize = [windowsize]
    if len(sigma) != len(windowsize):
        raise ValueError(f'sigma {sigma} and width {windowsize} differ in length')
    center = [(w - 1) / 2 for w in windowsize]                               # :640
    mesh = [np.arange(w) - c for w, c in zip(windowsize, center)]
    mesh = [-0.5 * x ** 2 for x in mesh]
    if not separate:
        mesh = np.meshgrid(*mesh, indexing=indexing)
    mesh = [np.asarray(m, dtype) for m in mesh]
    exponent = [(m / dtype(s) ** 2).astype(dtype) for m, s in zip(mesh, sigma)]        # :653
    if not separate:
        exponent = [np.sum(np.stack(exponent), axis=0, dtype=dtype)]
    kernel = [np.exp(x).astype(dtype) for x in exponent]
    kernel = [(x / x.sum(dtype=dtype)).astype(dtype) for x in kernel]
    return kernel if len(kernel) > 1 else kernel[0]


def conv1d_axis(x, k, axis, padding='SAME', stride=1, dilation=1):
    """cross-correlation of x [..] with the 1-D kernel k along `axis` (TF SAME/VALID rules), float64 accumulation."""
    x = np.asarray(x)
    k = np.asarray(k, np.float64).ravel()
    n, w = x.shape[axis], k.size
    ke = (w - 1) * dilation + 1
    if padding.upper() == 'SAME':
        o = -(-n // stride)
        tot = max((o - 1) * stride + ke - n, 0)
        before = tot // 2
    else:
        o = (n - ke) // stride + 1
        before = tot = 0
    pad = [(0, 0)] * x.ndim
    pad[axis] = (before, tot - before)
    xp = np.pad(x.astype(np.float64), pad)
    out = 0
    for t in range(w):
        sl = [slice(None)] * x.ndim
        st = t * dilation
        sl[axis] = slice(st, st + (o - 1) * stride + 1, stride)
        out = out + k[t] * xp[tuple(sl)]
    return out.astype(x.dtype)


def separable_conv(x, kernels, axis=None, batched=False, padding='SAME', strides=None, dilations=None):
    """utils.py:665-751: the same 1-D filters across features, one pass per spatial axis."""
    x = np.asarray(x)
    if not batched:
        x = x[None]
    num_dim = x.ndim - 2
    if np.isscalar(axis):
        axis = [axis]
    if axis is None:
        axis = list(range(num_dim))
    assert all(ax in range(num_dim) for ax in axis), 'non-spatial axis passed'

    def conform(v):
        v = np.ravel(1 if v is None else v).tolist()
        return v * len(axis) if len(v) == 1 else v
    strides, dilations = conform(strides), conform(dilations)
    assert len(strides) == len(axis), 'number of strides and axes differ'
    assert len(dilations) == len(axis), 'number of dilations and axes differ'
    if not isinstance(kernels, (tuple, list)):
        kernels = [kernels]
    if len(kernels) == 1:
        kernels = list(kernels) * len(axis)
    assert len(kernels) == len(axis), 'number of kernels and axes differ'
    for ax, k, s, d in zip(axis, kernels, strides, dilations):
        x = conv1d_axis(x, k, ax + 1, padding, int(s), int(d))
    return x if batched else x[0]


def gaussian_blur(x, sigma):
    """layers.GaussianBlur (non-random) on [B, *S, C]."""
    nd = x.ndim - 2
    sigma = np.ravel(sigma).tolist()
    if len(sigma) == 1:
        sigma = sigma * nd
    if not any(s > 0 for s in sigma):
        return x
    return separable_conv(x, gaussian_kernel(sigma, separate=True) if nd > 1 else [gaussian_kernel(sigma, separate=True)],
                          batched=True)


def minmax_norm(x, axis=None):
    """utils.py:953-968."""
    x = np.asarray(x)
    mn = x.min(axis=axis, keepdims=True)
    mx = x.max(axis=axis, keepdims=True)
    return divide_no_nan(x - mn, mx - mn)


# --------------------------------------------------------------------------------------
# soft quantisation and mutual information (neurite/tf/utils/utils.py:1099-1172, neurite/tf/metrics.py:41-336)
# --------------------------------------------------------------------------------------

def soft_quantize(x, bin_centers=None, nb_bins=16, alpha=1, min_clip=-np.inf, max_clip=np.inf, return_log=False):
    """utils.py:1099-1172: weight exp(-alpha (v - c_b)^2) of every value for every bin, float32 like the reference."""
    x = np.asarray(x, F32)
    if bin_centers is not None:
        bin_centers = np.asarray(bin_centers, F32)
        assert nb_bins is None, 'cannot provide both bin_centers and nb_bins'
    else:
        if nb_bins is None:
            nb_bins = 16
        bin_centers = tf_linspace(x.min(), x.max(), nb_bins)                # :1152-1154
    xc = np.clip(x[..., None], F32(min_clip), F32(max_clip))                # :1157-1158
    log = (-F32(alpha) * np.square(xc - bin_centers)).astype(F32)           # :1166-1167
    return log if return_log else np.exp(log).astype(F32)


def mi_default_alpha(nb_bins=16):
    """metrics.py:111-118: alpha = 1 / (2 sigma^2), sigma = 0.5 / (nb_bins - 1), evaluated in float32 as tf.square does."""
    sigma = 0.5 / (nb_bins - 1)
    return F32(1) / (F32(2) * np.square(F32(sigma)))


def mi_maps(x, y, eps=1e-7):
    """metrics.py:228-282: MI per batch entry of two maps [bs, ..., B] (float64 accumulation)."""
    x = np.asarray(x)
    y = np.asarray(y)
    if x.shape != y.shape:
        raise ValueError('maps: shapes differ')                              # tf.debugging.assert_equal :249
    if (x < 0).any() or (y < 0).any():
        raise ValueError('maps: negative values')                            # :250-251
    bs, B = x.shape[0], x.shape[-1]
    xf = x.reshape(bs, -1, B).astype(np.float64)
    yf = y.reshape(bs, -1, B).astype(np.float64)
    pxy = np.einsum('bvi,bvj->bij', xf, yf)
    pxy = pxy / (pxy.sum((1, 2), keepdims=True) + eps)
    px = xf.sum(1, keepdims=True)
    px = px / (px.sum(2, keepdims=True) + eps)
    py = yf.sum(1, keepdims=True)
    py = py / (py.sum(2, keepdims=True) + eps)
    pxpy = np.einsum('bki,bkj->bij', px, py) + eps
    return (pxy * np.log(pxy / pxpy + eps)).sum((1, 2)).astype(F32)


def mi_channelwise(x, y, nb_bins=16, alpha=None, min_clip=-np.inf, max_clip=np.inf):
    """metrics.py:188-226: bins are placed between the extrema of the WHOLE tensor (all channels and batch entries)."""
    x = np.asarray(x, F32)
    y = np.asarray(y, F32)
    assert x.shape == y.shape, 'volume shapes do not match'
    alpha = mi_default_alpha(nb_bins) if alpha is None else alpha
    bs, C = x.shape[0], x.shape[-1]
    xq = soft_quantize(x.reshape(bs, -1, C), None, nb_bins, alpha, min_clip, max_clip)     # [bs, V, C, B]
    yq = soft_quantize(y.reshape(bs, -1, C), None, nb_bins, alpha, min_clip, max_clip)
    return np.stack([mi_maps(xq[:, :, c], yq[:, :, c]) for c in range(C)], 1)


def mi_volumes(x, y, **kw):
    """metrics.py:119-142."""
    assert np.asarray(x).shape[-1] == 1 and np.asarray(y).shape[-1] == 1, 'volume_mi requires two single-channel volumes. See channelwise().'
    return mi_channelwise(x, y, **kw).reshape(-1)


def mi_volume_seg(x, y, nb_bins=16, alpha=None, min_clip=-np.inf, max_clip=np.inf):
    """metrics.py:156-186."""
    x = np.asarray(x, F32)
    y = np.asarray(y, F32)
    alpha = mi_default_alpha(nb_bins) if alpha is None else alpha
    assert min(x.shape[-1], y.shape[-1]) == 1, 'volume_seg_mi requires one single-channel volume.'
    assert max(x.shape[-1], y.shape[-1]) > 1, 'volume_seg_mi requires one multi-channel segmentation.'
    if x.shape[-1] == 1:
        x = soft_quantize(x[..., 0], None, nb_bins, alpha, min_clip, max_clip)
    else:
        y = soft_quantize(y[..., 0], None, nb_bins, alpha, min_clip, max_clip)
    return mi_maps(x, y)


# --------------------------------------------------------------------------------------
# label-to-image synthesis, deterministic part given the random draws (neurite/tf/models.py:819-918)
# --------------------------------------------------------------------------------------

def synth_image(idx, noise, mean, std, bg_zero, blur_kernels, bias, normalize, gamma, dc):
    """idx [B,*S,1] warped label indices (float), noise [B,*S], mean/std [B,C,L], bg_zero [B,C] or None, blur_kernels list of
    1-D kernels or None, bias [B,*S,1] or None, gamma/dc [B,C] or None.  Returns the image [B,*S,C] (float32 semantics)."""
    idx = np.asarray(idx)[..., 0].astype(np.int64)
    B = idx.shape[0]
    C = mean.shape[1]
    nd = idx.ndim - 1
    img = np.zeros(idx.shape + (C,), F32)
    for b in range(B):
        for c in range(C):
            img[b, ..., c] = (noise[b] * std[b, c][idx[b]] + mean[b, c][idx[b]]).astype(F32)          # :831-839
            if bg_zero is not None and bg_zero[b, c] != 0:
                img[b, ..., c] = np.where(idx[b] == 0, F32(0), img[b, ..., c])                          # :842-849
    if blur_kernels is not None:
        img = separable_conv(img, list(blur_kernels), batched=True)                                    # :852-857
    if bias is not None:
        img = (img * np.exp(np.asarray(bias, F32))).astype(F32)                                        # :871
    img = np.clip(img, 0, 255).astype(F32)                                                             # :874
    if normalize:
        img = np.stack([minmax_norm(img[b]) for b in range(B)], 0).astype(F32)                         # :875-876
    sh = (B,) + (1,) * nd + (C,)
    if gamma is not None:
        img = np.power(img, np.exp(np.asarray(gamma, F32)).reshape(sh)).astype(F32)                    # :877-881
    if dc is not None:
        img = (img + np.asarray(dc, F32).reshape(sh)).astype(F32)                                      # :882-888
    return img
