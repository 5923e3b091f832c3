"""ctypes binding of tools/lab/liblab.so (lab kernels and probes; see tools/lab/README.md).  Mirrors neurite_amd.fused.warp_dice /
the batched SpatialTransformer call for the LDS-row-cache kernel so that it can be checked against the product bit for bit."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from neurite_amd import _lib                                                   # noqa: E402

_vp, _i, _ll, _f, _sz, _ip = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.POINTER(C.c_int)
_SIG = {
    'nrt_lab_lc_workspace_bytes': (_sz, [_ip, _i, _i, _i]),
    'nrt_lab_lc_warp_dice_f32': (_i, [_vp, _vp, _vp, _vp, _ip, _ip, _i, _i, _ll, _i, _i, _f, _f, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    'nrt_lab_lc_interpn_f32': (_i, [_vp, _vp, _vp, _ip, _ip, _i, _ll, _ll, _i, _i, _f, _i, _vp]),
    'nrt_lab_membench_copy_f32': (_i, [_vp, _vp, _ll, _i, _i, _vp]),
    'nrt_lab_membench_l1_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'nrt_lab_stream_f32': (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _vp]),
}
_h = None


def lib():
    global _h
    if _h is None:
        path = os.path.join(HERE, 'liblab.so')
        if not os.path.exists(path):
            raise RuntimeError('tools/lab/liblab.so not built: python tools/lab/build.py')
        _h = C.CDLL(path)
        for name, (res, args) in _SIG.items():
            fn = getattr(_h, name)
            fn.restype, fn.argtypes = res, args
    return _h


def lc_warp_dice(moving, trf, fixed, fill_value=None, laplace_smoothing=0., return_warped=False, return_sums=False, tune=0):
    """SpatialTransformer + soft Dice through the LDS-row-cache kernel (32 float32 labels, 3-D).  Same results contract as
    neurite_amd.fused.warp_dice."""
    dev = _lib.require_device(moving, trf, fixed)
    B, L = moving.shape[0], moving.shape[-1]
    mov, fix, shift = moving.contiguous(), fixed.contiguous(), trf.to(torch.float32).contiguous()
    S, O = list(mov.shape[1:-1]), list(fix.shape[1:-1])
    sums = torch.empty((B, 3, L), dtype=torch.float32, device=dev)
    dice = torch.empty((B, L), dtype=torch.float32, device=dev)
    warped = torch.empty_like(fix) if return_warped else None
    o_shape = _lib.ints(O)
    nws = lib().nrt_lab_lc_workspace_bytes(o_shape, L, B, int(tune))
    ws = _lib.workspace(dev, nws)
    has_fill = fill_value is not None
    with torch.cuda.device(dev):
        rc = lib().nrt_lab_lc_warp_dice_f32(_lib.ptr(mov), _lib.ptr(shift), _lib.ptr(fix), _lib.ptr(warped), _lib.ints(S), o_shape, L, B,
                                            shift[0].numel(), _lib.LOC_SHIFT, int(has_fill), float(fill_value) if has_fill else 0.0,
                                            float(laplace_smoothing), _lib.ptr(sums), _lib.ptr(dice), None, int(tune), _lib.ptr(ws), nws,
                                            _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_lab_lc_warp_dice_f32')
    out = (dice,)
    if return_warped:
        out += (warped,)
    if return_sums:
        out += (sums,)
    return out[0] if len(out) == 1 else out


def lc_interpn(vol, loc, out_spatial, loc_mode, fill_value=None, tune=0):
    """Batched linear interpn of [B, *S, 32] volumes through the LDS-row-cache kernel; loc [B, *O, 3] (absolute / shift) or None
    (linspace, resize)."""
    dev = _lib.require_device(vol, loc)
    v = vol.contiguous()
    B = v.shape[0]
    out = torch.empty((B,) + tuple(out_spatial) + (32,), dtype=torch.float32, device=dev)
    lc = loc.to(torch.float32).contiguous() if loc is not None else None
    has_fill = fill_value is not None
    with torch.cuda.device(dev):
        rc = lib().nrt_lab_lc_interpn_f32(_lib.ptr(v), _lib.ptr(lc), _lib.ptr(out), _lib.ints(list(v.shape[1:-1])), _lib.ints(list(out_spatial)), B,
                                          v[0].numel(), lc[0].numel() if lc is not None else 0, int(loc_mode), int(has_fill),
                                          float(fill_value) if has_fill else 0.0, int(tune), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_lab_lc_interpn_f32')
    return out
