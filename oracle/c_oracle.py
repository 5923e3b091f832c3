"""
ctypes wrapper around oracle/_build/liboracle.so (TEST INFRASTRUCTURE -- see oracle/__init__.py).
NumPy in, NumPy out.  Builds the library on first use if gcc is available.
"""

import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(_build.SRC):
            path = _build.build()
        _lib = C.CDLL(path)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def interpn(vol, loc, interp_method='linear', fill_value=None, loc_mode=0, out_shape=None, out=None):
    """
    loc_mode 0: loc [*S', D] absolute; 1: loc is a shift added to the identity grid;
    2: loc is a list of D coordinate tables (resize()).  vol [*S, C] float32 (C axis required).
    out: optional pre-allocated (pre-touched) float32 result buffer -- timing loops pass one so that they measure the
    algorithm and not the page faults of a fresh allocation.
    """
    vol = np.ascontiguousarray(vol, np.float32)
    if loc_mode == 2:
        tables = [np.ascontiguousarray(t, np.float32) for t in loc]
        D = len(tables)
        out_shape = [len(t) for t in tables]
        locbuf = np.concatenate(tables)
    else:
        locbuf = np.ascontiguousarray(loc, np.float32)
        D = locbuf.shape[-1]
        out_shape = list(locbuf.shape[:-1])
    assert vol.ndim == D + 1
    Cc = vol.shape[-1]
    if out is None:
        out = np.empty(list(out_shape) + [Cc], np.float32)
    else:
        assert out.dtype == np.float32 and out.flags.c_contiguous and list(out.shape) == list(out_shape) + [Cc]
    method = {'linear': 0, 'nearest': 1}[interp_method]
    rc = lib().orc_interpn_f32(_p(vol), C.c_int(D), _ints(vol.shape[:-1]), C.c_int(Cc), _p(locbuf),
                               C.c_int(loc_mode), _ints(out_shape), C.c_int(method),
                               C.c_int(fill_value is not None),
                               C.c_float(0.0 if fill_value is None else fill_value), _p(out))
    assert rc == 0
    return out


def dice_sums(y_true, y_pred):
    t = np.ascontiguousarray(y_true, np.float32)
    p = np.ascontiguousarray(y_pred, np.float32)
    B, L = t.shape[0], t.shape[-1]
    V = t.size // (B * L)
    sums = np.empty((B, 3, L), np.float64)
    mm = np.empty(4, np.float32)
    rc = lib().orc_dice_sums_f32(_p(t), _p(p), C.c_int(B), C.c_int64(V), C.c_int(L), _p(sums), _p(mm))
    assert rc == 0
    return sums, mm


def dice_from_sums(sums, laplace_smoothing=0.):
    top = (2 * sums[:, 0]).astype(np.float32)
    bottom = (sums[:, 1].astype(np.float32) + sums[:, 2].astype(np.float32)).astype(np.float32)
    if laplace_smoothing > 0:
        eps = np.float32(laplace_smoothing)
        return ((top + eps) / (bottom + eps)).astype(np.float32)
    out = np.zeros_like(top)
    np.divide(top, bottom, out=out, where=bottom != 0)
    return out


def dice_hard_counts_prob(y_true, y_pred):
    t = np.ascontiguousarray(y_true, np.float32)
    p = np.ascontiguousarray(y_pred, np.float32)
    B, L = t.shape[0], t.shape[-1]
    V = t.size // (B * L)
    counts = np.empty((B, 3, L), np.int64)
    rc = lib().orc_dice_hard_counts_prob_f32(_p(t), _p(p), C.c_int(B), C.c_int64(V), C.c_int(L), _p(counts))
    assert rc == 0
    return counts


def dice_hard_counts_label(y_true, y_pred, nb_labels):
    t = np.ascontiguousarray(y_true, np.int32)
    p = np.ascontiguousarray(y_pred, np.int32)
    B = t.shape[0]
    V = t.size // B
    counts = np.empty((B, 3, nb_labels), np.int64)
    rc = lib().orc_dice_hard_counts_label_i32(_p(t), _p(p), C.c_int(B), C.c_int64(V), C.c_int(nb_labels),
                                              _p(counts))
    assert rc == 0
    return counts


def wcce(y_true, y_pred, label_weights=None, from_logits=False, label_smoothing=0., per_voxel=False):
    t = np.ascontiguousarray(y_true, np.float32)
    p = np.ascontiguousarray(y_pred, np.float32)
    Cc = t.shape[-1]
    N = t.size // Cc
    w = None if label_weights is None else np.ascontiguousarray(label_weights, np.float32)
    total = C.c_double(0.0)
    pv = np.empty(N, np.float64) if per_voxel else None
    rc = lib().orc_wcce_f32(_p(t), _p(p), None if w is None else _p(w), C.c_int64(N), C.c_int(Cc),
                            C.c_int(bool(from_logits)), C.c_double(label_smoothing), C.byref(total),
                            None if pv is None else _p(pv))
    assert rc == 0
    loss = np.float32(total.value / N)
    return (loss, pv.reshape(t.shape[:-1])) if per_voxel else loss


def conv3d_same(x, kernel, bias=None, dilation=1, elu=False):
    """x [X,Y,Z,Cin], kernel [kx,ky,kz,Cin,Cout] (Keras layout) -> [X,Y,Z,Cout] float32."""
    x = np.ascontiguousarray(x, np.float32)
    k = np.ascontiguousarray(kernel, np.float32)
    X, Y, Z, Cin = x.shape
    kx, ky, kz, Cin2, Cout = k.shape
    assert Cin == Cin2
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    y = np.empty((X, Y, Z, Cout), np.float32)
    rc = lib().orc_conv3d_same_f32(_p(x), X, Y, Z, Cin, _p(k), kx, ky, kz, Cout, C.c_int(dilation),
                                   None if b is None else _p(b), C.c_int(bool(elu)), _p(y))
    assert rc == 0
    return y
