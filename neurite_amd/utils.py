"""
neurite_amd.utils -- the tensor utilities of neurite's hot path on MI355X.

Mirrors the names, arguments, defaults and error behaviour of neurite/tf/utils/utils.py for
interpn (:73), resize/zoom (:223,:265), volshape_to_ndgrid (:333), volshape_to_meshgrid (:356),
ndgrid (:382), meshgrid (:398), sub2ind2d (:1068), prod_n (:1085), batch_channel_flatten (:1175),
flatten_axes (:1195); plus voxelmorph's transform()/affine_to_dense_shift, which the reference
calls but does not vendor (neurite/tf/models.py:806-807, 1157-1159).

Tensors are torch tensors on a ROCm device in the reference's channels-last layout.  All sampling
work is done by the HIP kernels in csrc/interpn.hip through the C ABI; nothing here falls back
to PyTorch or the CPU.
"""

import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ['interpn', 'resize', 'zoom', 'transform', 'affine_to_dense_shift', 'integrate_vec', 'compose',
           'gaussian_kernel', 'separable_conv', 'minmax_norm', 'soft_quantize',
           'rescale_dense_transform', 'rescale_affine', 'is_affine_shape', 'validate_affine_shape', 'make_square_affine', 'volshape_to_ndgrid',
           'volshape_to_meshgrid', 'ndgrid', 'meshgrid', 'sub2ind2d', 'prod_n', 'batch_channel_flatten',
           'flatten_batch_channel', 'flatten_axes']

_METHODS = {'linear': _lib.INTERP_LINEAR, 'nearest': _lib.INTERP_NEAREST}
_SMALL_INTS = (torch.int8, torch.uint8, torch.int16, torch.bool)


class _NoBackward(torch.autograd.Function):
    """Forward-only ops: reaching backward is an error, never a silent zero gradient."""

    @staticmethod
    def forward(ctx, fn, *tensors):
        with torch.no_grad():
            return fn()

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError('neurite_amd: backward of the HIP hot-path ops is not implemented yet '
                                  '(forward-only build; see DESIGN.md "what comes next").')


def _maybe_tracked(fn, *tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return _NoBackward.apply(fn, *[t for t in tensors if isinstance(t, torch.Tensor)])
    return fn()


class _InterpnFn(torch.autograd.Function):
    """
    interpn (float32, 1-3-D) with the hand-written backward of csrc/backward.hip: gradients wrt the volume (scatter-add)
    and, for linear interpolation, wrt the sampling locations / displacement field (what TF's autodiff derives from
    utils.py:137-213); nearest interpolation passes no gradient to the locations (tf.round).
    """

    @staticmethod
    def forward(ctx, vol, loc, cfg):
        vol = vol.contiguous()
        loc = None if loc is None else loc.contiguous()
        with torch.no_grad():
            out = _launch_interpn(vol, loc, **cfg)
        ctx.save_for_backward(vol, loc)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, grad_out):
        vol, loc = ctx.saved_tensors
        need_vol = ctx.needs_input_grad[0]
        need_loc = loc is not None and ctx.needs_input_grad[1]
        gvol, gloc = _launch_interpn_bwd(vol, loc, grad_out, ctx.cfg, need_vol, need_loc)
        return gvol, gloc, None


def _launch_interpn_bwd(vol, loc, grad_out, cfg, need_vol, need_loc):
    lib = _lib.lib()
    dev = _lib.require_device(vol, loc, grad_out)
    batched, single = cfg['batched'], cfg.get('single_transform', False)
    B = vol.shape[0] if batched else 1
    S = list(vol.shape[1:-1]) if batched else list(vol.shape[:-1])
    Cc, D = vol.shape[-1], len(S)
    out_spatial = [int(s) for s in cfg['out_spatial']]
    abi_spatial = out_spatial if len(out_spatial) == D else [int(np.prod(out_spatial))] + [1] * (D - 1)     # see _launch_interpn
    g = grad_out.to(torch.float32).contiguous()
    if cfg['method'] == _lib.INTERP_NEAREST:
        # utils.py:193-204: tf.round has no gradient (TF returns None for loc; zeros here so that optimisers see a tensor),
        # the volume receives tf.gather's scatter-add
        gvol = torch.zeros_like(vol) if need_vol else None
        gloc = torch.zeros_like(loc) if (need_loc and loc is not None) else None
        if gvol is not None and g.numel():
            nvol = int(np.prod(S)) * Cc
            nloc = int(np.prod(out_spatial)) * D
            loc_bs = 0 if (single or loc is None) else nloc
            with torch.cuda.device(dev):
                rc = lib.nrt_interpn_nearest_bwd_f32(_lib.ptr(loc), _lib.ptr(g), _lib.ptr(gvol), D, _lib.ints(S),
                                                     _lib.ints(abi_spatial), Cc, B, nvol, loc_bs, cfg['loc_mode'],
                                                     int(cfg['fill_value'] is not None), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_interpn_nearest_bwd_f32')
        return gvol, gloc
    gvol = torch.zeros_like(vol) if need_vol else None
    gloc = None
    if need_loc:
        gloc = torch.empty(([B] if batched else []) + out_spatial + [D], dtype=torch.float32, device=dev)
    if g.numel() == 0 or not (need_vol or need_loc):
        if gloc is not None:
            gloc.zero_()
    else:
        nvol = int(np.prod(S)) * Cc
        nloc = int(np.prod(out_spatial)) * D
        loc_bs = 0 if (single or loc is None) else nloc
        with torch.cuda.device(dev):
            rc = lib.nrt_interpn_bwd_f32(_lib.ptr(vol), _lib.ptr(loc), _lib.ptr(g), _lib.ptr(gvol), _lib.ptr(gloc), D,
                                         _lib.ints(S), _lib.ints(abi_spatial), Cc, B, nvol, loc_bs, cfg['loc_mode'],
                                         int(cfg['fill_value'] is not None), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_interpn_bwd_f32')
    if gloc is not None and single and batched:
        gloc = gloc.sum(0, keepdim=True)       # one transform shared by the whole batch
    return gvol, gloc


def _interp_op(vol, loc, out_spatial, loc_mode, method, fill_value, batched, single_transform=False,
               variant=0, tune=0):
    """Forward launch, recorded for autograd when an input requires grad (linear float32, 1-3-D only)."""
    cfg = dict(out_spatial=[int(s) for s in out_spatial], loc_mode=loc_mode, method=method, fill_value=fill_value,
               batched=batched, single_transform=single_transform, variant=variant, tune=tune)
    needs = torch.is_grad_enabled() and (vol.requires_grad or (loc is not None and loc.requires_grad))
    if not needs:
        return _launch_interpn(vol, loc, **cfg)
    vol_rank = vol.dim() - (2 if batched else 1)                        # the backward kernels take 1- to 3-D volumes
    if vol.dtype == torch.float32 and vol_rank <= 3:
        return _InterpnFn.apply(vol, loc, cfg)
    return _NoBackward.apply(lambda: _launch_interpn(vol, loc, **cfg), vol, *([] if loc is None else [loc]))


def _launch_interpn(vol, loc, out_spatial, loc_mode, method, fill_value, batched, single_transform=False,
                    variant=0, tune=0):
    """
    vol [B?, *S, C] contiguous; loc [B?, *S', D] float32 contiguous or None (LINSPACE).
    Returns out [B?, *S', C] with vol's dtype.
    """
    lib = _lib.lib()
    dev = _lib.require_device(vol, loc)
    vol = vol.contiguous()
    if batched:
        B = vol.shape[0]
        S = list(vol.shape[1:-1])
    else:
        B = 1
        S = list(vol.shape[:-1])
    Cc = vol.shape[-1]
    D = len(S)
    if D < 1 or D > _ANY_MAXD:
        raise NotImplementedError('neurite_amd.interpn supports 1- to %d-D volumes, got %d-D (2^D corner rows per output; '
                                  'the reference, neurite/tf/utils/utils.py:159, has no rank limit)' % (_ANY_MAXD, D))
    out_spatial = [int(s) for s in out_spatial]
    out_shape = ([B] if batched else []) + out_spatial + [Cc]
    out = torch.empty(out_shape, dtype=vol.dtype, device=dev)
    if out.numel() == 0:
        return out
    nvol = int(np.prod(S)) * Cc
    nloc = int(np.prod(out_spatial)) * D
    # the C ABI takes exactly D output extents.  The reference accepts location tensors of any leading rank (e.g. [N, D] sample
    # points for a 3-D volume): flatten them to [N, 1, ...]; the output is contiguous, so the shape above already is its final one
    if len(out_spatial) != D:
        if loc_mode == _lib.LOC_LINSPACE:
            raise ValueError('resize needs one output extent per volume dimension')
        out_spatial = [int(np.prod(out_spatial))] + [1] * (D - 1)
    if loc is not None:
        loc = loc.contiguous()
    vol_bs = nvol
    loc_bs = 0 if (single_transform or loc is None) else nloc
    has_fill = fill_value is not None
    st = _lib.stream_ptr(dev)
    with torch.cuda.device(dev):
        if vol.dtype in _ANY_DTYPES and (D > 3 or vol.dtype not in (torch.float32, torch.int32)):
            # float16 / bfloat16 / float64 volumes, and ranks 4-8: the dtype- and rank-generic kernel (csrc/interpn_any.hip);
            # arithmetic in the volume dtype, one rounding per op, as TensorFlow evaluates utils.py:137-213
            if vol.dtype == torch.int32:
                assert method == _lib.INTERP_NEAREST
            loc64 = loc is not None and loc.dtype == torch.float64
            assert not loc64 or (vol.dtype == torch.float64 and loc_mode == _lib.LOC_ABSOLUTE)
            rc = lib.nrt_interpn_any(_lib.ptr(vol), _lib.ptr(loc), _lib.ptr(out), _ANY_DTYPES[vol.dtype], D, _lib.ints(S),
                                     _lib.ints(out_spatial), Cc, B, vol_bs, loc_bs, loc_mode, int(loc64), method,
                                     int(has_fill), float(fill_value) if has_fill else 0.0, st)
        elif vol.dtype == torch.float32:
            rc = lib.nrt_interpn_f32_ex(_lib.ptr(vol), _lib.ptr(loc), _lib.ptr(out), D, _lib.ints(S),
                                        _lib.ints(out_spatial), Cc, B, vol_bs, loc_bs, loc_mode, method,
                                        int(has_fill), float(fill_value) if has_fill else 0.0,
                                        int(variant), int(tune), st)
        elif vol.dtype == torch.int32:
            assert method == _lib.INTERP_NEAREST
            rc = lib.nrt_interpn_nearest_i32(_lib.ptr(vol), _lib.ptr(loc), _lib.ptr(out), D, _lib.ints(S),
                                             _lib.ints(out_spatial), Cc, B, vol_bs, loc_bs, loc_mode,
                                             int(has_fill), int(fill_value) if has_fill else 0, st)
        else:
            raise NotImplementedError('unsupported volume dtype %s' % vol.dtype)
    _lib.check(rc, 'nrt_interpn')
    return out


_ANY_MAXD = 8
_ANY_DTYPES = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16,
               torch.float64: _lib.DT_F64, torch.int32: _lib.DT_I32}


def _loc_for(vol, loc):
    """`loc` as the kernels take it.  The reference casts loc to the volume's dtype when both are floating (utils.py:123-127)
    and integer locations to float32 (:124-125).  The kernels read float32 locations and round them to the volume dtype
    themselves (float16 / bfloat16: RNE, the same rounding as tf.cast; values that already are of the narrow type survive the
    float32 detour unchanged); only a float64 volume reads float64 locations."""
    if not loc.dtype.is_floating_point:
        return loc.to(torch.float32)
    if vol.dtype == torch.float64:
        return loc.to(torch.float64)
    if vol.dtype in (torch.float16, torch.bfloat16):
        return loc.to(vol.dtype).to(torch.float32) if loc.dtype == torch.float64 else loc.to(torch.float32)
    return loc.to(torch.float32)


def _prepare_vol(vol, interp_method):
    """dtype policy of the HIP path.  Returns (vol as the kernels take it, restore_dtype)."""
    if vol.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.float64):
        return vol, None
    # integer volumes
    if interp_method == 'linear':
        # the reference multiplies float weights with the gathered values (utils.py:191): TF raises
        raise TypeError('linear interpolation of an integer volume (%s) is a dtype error in the reference; '
                        'cast the volume to float32 first' % vol.dtype)
    if vol.dtype == torch.int32:
        return vol, None
    if vol.dtype in _SMALL_INTS:
        return vol.to(torch.int32), vol.dtype
    raise NotImplementedError('nearest interpolation of %s volumes is not implemented' % vol.dtype)


def interpn(vol, loc, interp_method='linear', fill_value=None, *, _variant=0, _tune=0):
    """
    N-D gridded interpolation (neurite/tf/utils/utils.py:73-220).

    vol: [*vol_shape] or [*vol_shape, C];  loc: list of D tensors or a [*new_shape, D] tensor;
    interp_method 'linear' | 'nearest'; fill_value: value outside the domain (None = edge clamp).
    Returns a tensor shaped like the entries of loc (+ channel axis if vol had one).
    """
    if isinstance(loc, (list, tuple)):
        loc = torch.stack([torch.as_tensor(l) for l in loc], -1)             # :106-107
    nb_dims = loc.shape[-1]                                                  # :108
    input_vol_ndim = vol.dim()

    if vol.dim() not in [nb_dims, nb_dims + 1]:                              # :111-113
        raise Exception("Number of loc Tensors %d does not match volume dimension %d"
                        % (nb_dims, len(vol.shape[:-1])))
    if nb_dims > vol.dim():                                                  # :115-117
        raise Exception("Loc dimension %d does not match volume dimension %d" % (nb_dims, vol.dim()))
    if interp_method != 'linear':                                            # :193-195
        assert interp_method == 'nearest', 'method should be linear or nearest, got: %s' % interp_method

    _lib.require_device(vol, loc)
    if vol.dim() == nb_dims:                                                 # :119-120
        vol = vol.unsqueeze(-1)
    loc = _loc_for(vol, loc)                                                 # :123-127
    vol32, restore = _prepare_vol(vol, interp_method)
    if vol32.dtype == torch.int32 and fill_value is not None and float(fill_value) != int(fill_value):
        raise ValueError('fill_value %r is not representable in the integer volume dtype' % (fill_value,))

    out = _interp_op(vol32, loc, loc.shape[:-1], _lib.LOC_ABSOLUTE, _METHODS[interp_method],
                     fill_value, batched=False, variant=_variant, tune=_tune)
    if restore is not None:
        out = out.to(restore)
    if input_vol_ndim == nb_dims:                                            # :216-218
        out = out[..., 0]
    return out


def _new_shape(vol_shape, zoom_factor):
    return [int(vol_shape[f] * zoom_factor[f]) for f in range(len(zoom_factor))]     # :256-257


def resize(vol, zoom_factor, interp_method='linear'):
    """
    Align-corners resize of one (un-batched) volume (neurite/tf/utils/utils.py:223-262).
    If zoom_factor is a list it determines ndims and vol may or may not carry a channel axis; if it
    is a scalar, vol must be [*vol_shape, C].
    """
    if isinstance(zoom_factor, (list, tuple)):                               # :237-242
        ndims = len(zoom_factor)
        vol_shape = list(vol.shape[:ndims])
        assert len(vol_shape) in (ndims, ndims + 1), \
            "zoom_factor length %d does not match ndims %d" % (len(vol_shape), ndims)
    else:                                                                    # :244-247
        vol_shape = list(vol.shape[:-1])
        ndims = len(vol_shape)
        zoom_factor = [zoom_factor] * ndims
    if all(z == 1 for z in zoom_factor):                                     # :250-251
        return vol
    if interp_method != 'linear':
        assert interp_method == 'nearest', 'method should be linear or nearest, got: %s' % interp_method
    _lib.require_device(vol)
    if vol.dim() not in (ndims, ndims + 1):
        raise Exception("Number of loc Tensors %d does not match volume dimension %d"
                        % (ndims, len(vol.shape[:-1])))
    new_shape = _new_shape(vol_shape, zoom_factor)
    squeeze = vol.dim() == ndims
    v = vol.unsqueeze(-1) if squeeze else vol
    vol32, restore = _prepare_vol(v, interp_method)

    out = _interp_op(vol32, None, new_shape, _lib.LOC_LINSPACE, _METHODS[interp_method], None, batched=False)
    if restore is not None:
        out = out.to(restore)
    return out[..., 0] if squeeze else out


zoom = resize


def transform(vol, loc_shift, interp_method='linear', indexing='ij', fill_value=None):
    """
    voxelmorph.utils.transform for one (un-batched) volume: out[q] = interpn(vol, q + loc_shift[q]).
    The identity grid is never materialised (the kernel derives q from its thread index).
    vol [*S, C] (or [*S]); loc_shift [*S', D] in voxel units.
    """
    if indexing not in ('ij', 'xy'):
        raise ValueError("indexing has to be 'ij' (matrix) or 'xy' (cartesian)")
    D = loc_shift.shape[-1]
    if vol.dim() not in (D, D + 1):
        raise Exception("Number of loc Tensors %d does not match volume dimension %d"
                        % (D, len(vol.shape[:-1])))
    if interp_method != 'linear':
        assert interp_method == 'nearest', 'method should be linear or nearest, got: %s' % interp_method
    _lib.require_device(vol, loc_shift)
    squeeze = vol.dim() == D
    v = vol.unsqueeze(-1) if squeeze else vol
    shift = loc_shift.to(torch.float32)
    if indexing == 'xy' and D > 1:
        shift = torch.cat([shift[..., 1:2], shift[..., 0:1], shift[..., 2:]], -1)
    vol32, restore = _prepare_vol(v, interp_method)

    out = _interp_op(vol32, shift, shift.shape[:-1], _lib.LOC_SHIFT, _METHODS[interp_method], fill_value,
                     batched=False)
    if restore is not None:
        out = out.to(restore)
    return out[..., 0] if squeeze else out


def affine_to_dense_shift(matrix, shape, shift_center=True, indexing='ij'):
    """
    voxelmorph.utils.affine_to_dense_shift: dense displacement field [*shape, D] of an affine
    [D, D+1] (or [D+1, D+1]); a batch of affines [B, D, D+1] gives [B, *shape, D].  float32.  Device matrices: one kernel
    (csrc/vxm.hip); host matrices, 'xy' indexing and matrices that need a gradient: torch ops (tiny matmul over the grid).
    """
    D = len(shape)
    matrix = torch.as_tensor(matrix, dtype=torch.float32)
    if matrix.shape[-2] == D + 1:
        matrix = matrix[..., :D, :]
    if tuple(matrix.shape[-2:]) != (D, D + 1):
        raise ValueError('affine matrix must be [%d, %d] or [%d, %d]' % (D, D + 1, D + 1, D + 1))
    if (indexing == 'ij' and matrix.device.type == 'cuda' and D in (2, 3) and matrix.dim() in (2, 3)
            and not (torch.is_grad_enabled() and matrix.requires_grad)):
        # one kernel writes the D floats of a voxel (csrc/vxm.hip); a leading batch axis gives [B, *shape, D].  Under autograd
        # (an affine that a network predicted) the differentiable torch form below runs instead.
        lib = _lib.lib()
        m = matrix.detach().contiguous()
        batch = m.shape[0] if m.dim() == 3 else 1
        out = torch.empty(((batch,) if m.dim() == 3 else ()) + tuple(int(n) for n in shape) + (D,), dtype=torch.float32, device=m.device)
        if out.numel():
            with torch.cuda.device(m.device):
                rc = lib.nrt_affine_to_dense_shift_f32(_lib.ptr(m), int(batch), D, _lib.ints([int(n) for n in shape]),
                                                       int(bool(shift_center)), _lib.ptr(out), _lib.stream_ptr(m.device))
            _lib.check(rc, 'nrt_affine_to_dense_shift_f32')
        return out
    if matrix.dim() == 3:
        return torch.stack([affine_to_dense_shift(matrix[b], shape, shift_center=shift_center, indexing=indexing)
                            for b in range(matrix.shape[0])], 0)
    if indexing == 'ij' and matrix.device.type == 'cuda':
        # the grid is built where the matrix lives (the host grid + copy cost ~15 ms per 160^3 field)
        lin = [torch.arange(int(n), dtype=torch.float32, device=matrix.device) for n in shape]
        mesh = list(torch.meshgrid(*lin, indexing='ij'))
    else:
        mesh = volshape_to_meshgrid(shape, indexing=indexing)
        mesh = [m.to(matrix.device, torch.float32) for m in mesh]
    if shift_center:
        mesh = [mesh[d] - (shape[d] - 1) / 2 for d in range(D)]
    flat = [m.reshape(-1) for m in mesh]
    flat.append(torch.ones_like(flat[0]))
    mesh_matrix = torch.stack(flat, 0)                        # [D+1, V]
    loc = (matrix @ mesh_matrix).transpose(0, 1).reshape(list(shape) + [D])
    return loc - torch.stack(mesh, -1)


# --------------------------------------------------------------------------------------
# VoxelMorph companions of SpatialTransformer (called by the reference next to it, neurite/tf/models.py:802-804,
# 1131, 1149-1154; voxelmorph itself is not vendored, so these follow its published semantics)
# --------------------------------------------------------------------------------------

def _warp_add(src, shift, batched, interp_method='linear', fill_value=None):
    """
    shift + transform(src, shift) in 'ij' coordinates: one kernel pass (csrc/interpn.hip, nrt_interpn_add_f32).
    src [B?, *S, C], shift [B?, *S', D] float32.  Falls back to the differentiable two-op form under autograd.
    """
    lib = _lib.lib()
    dev = _lib.require_device(src, shift)
    if interp_method != 'linear':
        assert interp_method == 'nearest', 'method should be linear or nearest, got: %s' % interp_method
    if src.dtype != torch.float32 or shift.dtype != torch.float32:
        raise NotImplementedError('displacement fields must be float32')
    D = shift.shape[-1]
    if src.shape[-1] != shift.shape[-1] and src.shape[-1] != D:
        raise ValueError('cannot add a %d-channel warped field to a %d-D displacement' % (src.shape[-1], D))
    needs = torch.is_grad_enabled() and (src.requires_grad or shift.requires_grad)
    out_spatial = list(shift.shape[1:-1] if batched else shift.shape[:-1])
    if needs or interp_method != 'linear':
        return shift + _interp_op(src, shift, out_spatial, _lib.LOC_SHIFT, _METHODS[interp_method], fill_value,
                                  batched=batched)
    src = src.contiguous()
    shift = shift.contiguous()
    B = src.shape[0] if batched else 1
    S = list(src.shape[1:-1] if batched else src.shape[:-1])
    if len(S) != D or D < 1 or D > 3:
        raise Exception("Number of loc Tensors %d does not match volume dimension %d" % (D, len(S)))
    if batched and shift.shape[0] != B:
        raise ValueError('batch size of the two transforms differs')
    Cc = src.shape[-1]
    out = torch.empty_like(shift)
    if out.numel() == 0:
        return out
    has_fill = fill_value is not None
    with torch.cuda.device(dev):
        rc = lib.nrt_interpn_add_f32(_lib.ptr(src), _lib.ptr(shift), _lib.ptr(shift), _lib.ptr(out), D, _lib.ints(S),
                                     _lib.ints(out_spatial), Cc, B, int(np.prod(S)) * Cc,
                                     int(np.prod(out_spatial)) * D, int(np.prod(out_spatial)) * Cc, _lib.LOC_SHIFT,
                                     int(has_fill), float(fill_value) if has_fill else 0.0, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_interpn_add_f32')
    return out


def integrate_vec(vec, time_dep=False, method='ss', _batched=False, **kwargs):
    """
    voxelmorph.utils.integrate_vec for one stationary velocity field [*S, D] ('ij' coordinates).
    'ss' / 'scaling_and_squaring': vec /= 2**nb_steps; nb_steps x (vec += transform(vec, vec)).
    'quadrature': vec /= nb_steps; disp = vec; (nb_steps - 1) x (disp += transform(vec, disp)).
    """
    if method not in ['ss', 'scaling_and_squaring', 'ode', 'quadrature']:
        raise ValueError("method has to be 'scaling_and_squaring' or 'ode'. found: %s" % method)
    if time_dep:
        raise NotImplementedError('integrate_vec: time-dependent velocity fields are not implemented')
    if method == 'ode':
        raise NotImplementedError("integrate_vec: method='ode' (tf odeint) is not implemented")
    _lib.require_device(vec)
    vec = vec.to(torch.float32)
    nb_steps = kwargs['nb_steps']
    if method in ['ss', 'scaling_and_squaring']:
        assert nb_steps >= 0, 'nb_steps should be >= 0, found: %d' % nb_steps
        vec = vec / (2 ** nb_steps)
        for _ in range(nb_steps):
            vec = _warp_add(vec, vec, _batched)
        return vec
    assert nb_steps >= 1, 'nb_steps should be >= 1, found: %d' % nb_steps
    vec = vec / nb_steps
    disp = vec
    for _ in range(nb_steps - 1):
        disp = _warp_add(vec, disp, _batched)
    return disp


def validate_affine_shape(shape):
    """voxelmorph.utils.validate_affine_shape: (..., N, N+1) or (..., N+1, N+1) with N in (2, 3)."""
    ndim = shape[-1] - 1
    rows = shape[-2]
    if ndim not in (2, 3) or rows not in (ndim, ndim + 1):
        raise ValueError(f'Affine matrix must be of shape (2, 3) or (3, 4), got {tuple(shape[-2:])}.')


def is_affine_shape(shape):
    """voxelmorph.utils.is_affine_shape on a shape without the batch axis."""
    if len(shape) == 2 and shape[-1] != 1:
        validate_affine_shape(shape)
        return True
    return False


def make_square_affine(mat):
    """voxelmorph.utils.make_square_affine: append the homogeneous row to [..., N, N+1] matrices."""
    validate_affine_shape(mat.shape)
    if mat.shape[-2] == mat.shape[-1]:
        return mat
    row = torch.zeros(tuple(mat.shape[:-2]) + (1, mat.shape[-1]), dtype=mat.dtype, device=mat.device)
    row[..., -1] = 1
    return torch.cat([mat, row], -2)


def rescale_affine(mat, factor):
    """voxelmorph.utils.rescale_affine: scale the translation column."""
    return torch.cat([mat[..., :-1], (mat[..., -1] * factor).unsqueeze(-1)], -1)


def rescale_dense_transform(transform_field, factor, interp_method='linear', _batched=False):
    """
    voxelmorph.utils.rescale_dense_transform: resize a displacement field and scale its vectors; the resize comes
    first when shrinking (factor < 1) and last when growing, as upstream.
    """
    from . import layers as _layers

    def rs(t):
        if _batched:
            return _layers.Resize(factor, interp_method=interp_method)(t)
        return resize(t, factor, interp_method=interp_method)

    if factor < 1:
        return rs(transform_field) * factor
    return rs(transform_field * factor)


def compose(transforms, interp_method='linear', shift_center=True, indexing='ij', _batched=False):
    """
    voxelmorph.utils.compose: compose a list of affine matrices and/or dense displacement fields (un-batched) into
    one transform, T = transforms[0] o transforms[1] o ...; the result is dense if any input is.
    """
    if indexing != 'ij':
        raise ValueError('Compose transform only supports ij indexing')
    if len(transforms) < 2:
        raise ValueError('Compose transform list size must be greater than 1')

    def affine(t):
        return is_affine_shape(t.shape[1:] if _batched else t.shape)

    def densify(t, shape):
        if not affine(t):
            return t
        return affine_to_dense_shift(t, shape, shift_center=shift_center, indexing=indexing)     # [B, D, D+1] -> [B, *shape, D]

    curr = transforms[-1]
    for nxt in reversed(transforms[:-1]):
        dense = next((t for t in (nxt, curr) if not affine(t)), None)
        if dense is not None:
            shape = list(dense.shape[1:-1] if _batched else dense.shape[:-1])
            nxt_d, curr_d = densify(nxt, shape), densify(curr, shape)
            curr = _warp_add(nxt_d.to(torch.float32), curr_d.to(torch.float32), _batched, interp_method)
        else:
            curr = torch.matmul(make_square_affine(nxt), make_square_affine(curr))[..., :-1, :]
    return curr


# --------------------------------------------------------------------------------------
# filtering and normalisation (synthesis front-end; neurite/tf/utils/utils.py:581-751, 953-968)
# --------------------------------------------------------------------------------------

def gaussian_kernel(sigma, windowsize=None, indexing='ij', separate=False, random=False, min_sigma=0,
                    dtype=torch.float32, seed=None, device=None, _draws=None):
    """
    N-dimensional Gaussian kernel (utils.py:581-662), or a list of N 1-D kernels with separate=True.  Host-side: the
    kernels are a few dozen numbers.  random=True draws each SD uniformly from [min_sigma, sigma) with torch's RNG
    (the reference uses tf.random.uniform: same distribution, different stream).
    """
    if not dtype.is_floating_point:
        raise AssertionError(f'{dtype} is not a real floating-point type')
    npdt = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16}.get(dtype, np.float32)
    if not isinstance(sigma, (list, tuple)):
        sigma = [sigma]
    if not isinstance(min_sigma, (list, tuple)):
        min_sigma = [min_sigma] * len(sigma)
    sigma = [max(f, np.finfo(npdt).eps) for f in sigma]
    min_sigma = [max(f, np.finfo(npdt).eps) for f in min_sigma]
    if windowsize is None:
        windowsize = [np.round(f * 3) * 2 + 1 for f in sigma]
    if not isinstance(windowsize, (list, tuple)):
        windowsize = [windowsize]
    if len(sigma) != len(windowsize):
        raise ValueError(f'sigma {sigma} and width {windowsize} differ in length')
    center = [(w - 1) / 2 for w in windowsize]
    mesh = [np.arange(w) - c for w, c in zip(windowsize, center)]
    mesh = [-0.5 * x ** 2 for x in mesh]
    if not separate:
        mesh = np.meshgrid(*mesh, indexing=indexing)
    mesh = [torch.as_tensor(np.asarray(m), dtype=dtype, device=device) for m in mesh]
    if random:
        gen = torch.Generator(device='cpu')
        gen.manual_seed(int(np.random.default_rng(seed).integers(2 ** 31 - 1)))
        draws = None if _draws is None else list(_draws)         # tests: the uniform [0, 1) numbers to use

        def uniform():
            return np.float32(draws.pop(0)) if draws is not None else np.float32(float(torch.rand((), generator=gen)))
        # tf.random.uniform(minval=a, maxval=b) = u * (b - a) + a in float32
        sigma = [float(uniform() * np.float32(np.float32(b) - np.float32(a)) + np.float32(a)) for a, b in zip(min_sigma, sigma)]
    exponent = [m / torch.tensor(s, dtype=dtype) ** 2 for m, s in zip(mesh, sigma)]
    if not separate:
        exponent = [torch.stack(exponent).sum(0)]
    kernel = [torch.exp(x) for x in exponent]
    kernel = [x / x.sum() for x in kernel]
    return kernel if len(kernel) > 1 else kernel[0]


def _conv1d_axis(x, k, ax, padding, stride, dilation):
    """one separable pass along tensor axis `ax` of the contiguous float32 tensor x (csrc/filter.hip)."""
    lib = _lib.lib()
    dev = x.device
    n, w = x.shape[ax], k.numel()
    ke = (w - 1) * dilation + 1
    if padding.upper() == 'SAME':
        o = -(-n // stride)
        before = max((o - 1) * stride + ke - n, 0) // 2
    elif padding.upper() == 'VALID':
        o = (n - ke) // stride + 1
        before = 0
        if o < 1:
            raise ValueError('Negative dimension size caused by a VALID convolution of width %d along a length-%d axis' % (ke, n))
    else:
        raise ValueError('padding must be "SAME" or "VALID"')
    outer = int(np.prod(x.shape[:ax])) if ax else 1
    inner = int(np.prod(x.shape[ax + 1:]))
    y = torch.empty(tuple(x.shape[:ax]) + (o,) + tuple(x.shape[ax + 1:]), dtype=torch.float32, device=dev)
    if y.numel():
        with torch.cuda.device(dev):
            rc = lib.nrt_conv1d_axis_f32(_lib.ptr(x), _lib.ptr(k), _lib.ptr(y), outer, n, inner, o, w, int(stride),
                                         int(dilation), int(before), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv1d_axis_f32')
    return y


def separable_conv(x, kernels, axis=None, batched=False, padding='SAME', strides=None, dilations=None):
    """
    Apply 1-D kernels along spatial axes of a tensor with a trailing feature dimension, the same filters across
    features (utils.py:665-751).  One kernel pass per axis on the tensor as it lies in memory -- no transposes.
    """
    dev = _lib.require_device(x)
    if x.dtype != torch.float32:
        raise NotImplementedError('separable_conv: float32 tensors, got %s' % x.dtype)
    if not batched:
        x = x.unsqueeze(0)
    num_dim = x.dim() - 2
    if np.isscalar(axis):
        axis = [axis]
    axes_space = range(num_dim)
    if axis is None:
        axis = list(axes_space)
    assert all(ax in axes_space for ax in axis), 'non-spatial axis passed'

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
    x = x.contiguous()
    for ax, k, s, d in zip(axis, kernels, strides, dilations):
        k = torch.as_tensor(k, dtype=torch.float32).reshape(-1).to(dev).contiguous()
        x = _conv1d_axis(x, k, ax + 1, padding, int(s), int(d))
    return x if batched else x[0]


def minmax_norm(x, axis=None):
    """
    Min-max normalise with a safe division (utils.py:953-968).  axis: dimensions to reduce (None = all); they must
    form one contiguous run (e.g. all, all but the batch, all spatial axes of a channels-last batch).
    """
    lib = _lib.lib()
    dev = _lib.require_device(x)
    if x.dtype != torch.float32:
        raise NotImplementedError('minmax_norm: float32 tensors, got %s' % x.dtype)
    nd = x.dim()
    axes = sorted(set(range(nd))) if axis is None else sorted(set(int(a) % nd for a in np.ravel(axis)))
    if not axes or axes != list(range(axes[0], axes[-1] + 1)):
        raise NotImplementedError('minmax_norm: the reduced axes must be contiguous, got %r' % (axis,))
    x = x.contiguous()
    outer = int(np.prod(x.shape[:axes[0]])) if axes[0] else 1
    red = int(np.prod(x.shape[axes[0]:axes[-1] + 1]))
    inner = int(np.prod(x.shape[axes[-1] + 1:]))
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    nws = lib.nrt_minmax_workspace_bytes(outer, inner)
    ws = torch.empty((max(int(nws), 16),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_minmax_norm_f32(_lib.ptr(x), _lib.ptr(y), outer, red, inner, _lib.ptr(ws), nws, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_minmax_norm_f32')
    return y


def _device_minmax(x):
    """[min, max] of a float32 tensor as a 2-element device tensor (csrc/filter.hip; no host synchronisation)."""
    lib = _lib.lib()
    dev = x.device
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    nws = int(lib.nrt_minmax_workspace_bytes(1, 1))
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_minmax_f32(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_minmax_f32')
    return out


def _bin_centers(x, bin_centers, nb_bins):
    """bin centres of soft_quantize as a float32 device tensor: given, or tf.linspace(min(x), max(x), nb_bins) (:1152-1154)."""
    dev = x.device
    if bin_centers is not None:
        assert nb_bins is None, 'cannot provide both bin_centers and nb_bins'
        return torch.as_tensor(bin_centers, dtype=torch.float32).reshape(-1).to(dev).contiguous()
    if nb_bins is None:
        nb_bins = 16
    # one reduction + one tiny kernel (csrc/filter.hip): start + delta * i, the ends exact as tf.linspace's are
    lib = _lib.lib()
    x = x.contiguous()
    c = torch.empty((int(nb_bins),), dtype=torch.float32, device=dev)
    nws = int(lib.nrt_minmax_workspace_bytes(1, 1))
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_bin_centers_f32(_lib.ptr(x), x.numel(), int(nb_bins), _lib.ptr(c), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_bin_centers_f32')
    return c


def soft_quantize(x, bin_centers=None, nb_bins=16, alpha=1, min_clip=-np.inf, max_clip=np.inf, return_log=False):
    """
    (Softly) quantise the values of a tensor with RBFs (utils.py:1099-1172): value v gets weight exp(-alpha (v - c)^2) for
    every bin centre c.  Returns a tensor with one more dimension [..., B].  Specify bin_centers OR nb_bins (centres are then
    spread between the extrema of x).
    """
    lib = _lib.lib()
    dev = _lib.require_device(x)
    if x.dtype != torch.float32:
        raise NotImplementedError('soft_quantize: float32 tensors, got %s' % x.dtype)
    x = x.contiguous()
    centers = _bin_centers(x.detach(), bin_centers, nb_bins)
    cfg = (float(alpha), float(min_clip), float(max_clip), int(bool(return_log)))
    if torch.is_grad_enabled() and x.requires_grad:
        return _SoftQuantizeFn.apply(x, centers, cfg)
    return _soft_quantize_fwd(x, centers, cfg)


def _soft_quantize_fwd(x, centers, cfg):
    lib = _lib.lib()
    dev = x.device
    nb = centers.numel()
    out = torch.empty(tuple(x.shape) + (nb,), dtype=torch.float32, device=dev)
    if x.numel():
        with torch.cuda.device(dev):
            rc = lib.nrt_soft_quantize_f32(_lib.ptr(x), _lib.ptr(centers), cfg[0], cfg[1], cfg[2], cfg[3], _lib.ptr(out),
                                           x.numel(), nb, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_soft_quantize_f32')
    return out


class _SoftQuantizeFn(torch.autograd.Function):
    """soft_quantize with its backward wrt x (csrc/mi.hip: soft_quantize_bwd).  The bin centres are treated as constants:
    when they come from tf.linspace(min(x), max(x)) the reference also sends a term to the two extremal voxels -- the same
    convention as the fused MutualInformation backward."""

    @staticmethod
    def forward(ctx, x, centers, cfg):
        ctx.save_for_backward(x, centers)
        ctx.cfg = cfg
        with torch.no_grad():
            return _soft_quantize_fwd(x, centers, cfg)

    @staticmethod
    def backward(ctx, g):
        x, centers = ctx.saved_tensors
        cfg = ctx.cfg
        lib = _lib.lib()
        dev = x.device
        g = g.to(torch.float32).contiguous()
        gx = torch.empty_like(x)
        if x.numel():
            with torch.cuda.device(dev):
                rc = lib.nrt_soft_quantize_bwd_f32(_lib.ptr(x), _lib.ptr(centers), cfg[0], cfg[1], cfg[2], cfg[3], _lib.ptr(g),
                                                   _lib.ptr(gx), x.numel(), centers.numel(), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_soft_quantize_bwd_f32')
        return gx, None, None


def soft_digitize(*args, **kwargs):
    """alias of soft_quantize (utils.py:1095-1096)"""
    return soft_quantize(*args, **kwargs)


# --------------------------------------------------------------------------------------
# index / grid helpers (host-side; the kernels never need the materialised grids)
# --------------------------------------------------------------------------------------

def volshape_to_ndgrid(volshape, **kwargs):
    """neurite/tf/utils/utils.py:333-353."""
    if not all(float(d).is_integer() for d in volshape):
        raise ValueError("volshape needs to be a list of integers")
    linvec = [torch.arange(0, int(d), dtype=torch.int32) for d in volshape]
    return ndgrid(*linvec, **kwargs)


def volshape_to_meshgrid(volshape, **kwargs):
    """neurite/tf/utils/utils.py:356-379 ('xy' default like tf.meshgrid)."""
    if not all(float(d).is_integer() for d in volshape):
        raise ValueError("volshape needs to be a list of integers")
    linvec = [torch.arange(0, int(d), dtype=torch.int32) for d in volshape]
    return meshgrid(*linvec, **kwargs)


def ndgrid(*args, **kwargs):
    """neurite/tf/utils/utils.py:382-395."""
    return meshgrid(*args, indexing='ij', **kwargs)


def meshgrid(*args, **kwargs):
    """neurite/tf/utils/utils.py:398-476."""
    indexing = kwargs.pop("indexing", "xy")
    kwargs.pop("name", None)
    if kwargs:
        key = list(kwargs.keys())[0]
        raise TypeError("'{}' is an invalid keyword argument for this function".format(key))
    if indexing not in ("xy", "ij"):
        raise ValueError("indexing parameter must be either 'xy' or 'ij'")
    args = [torch.as_tensor(a) for a in args]
    return [g.contiguous() for g in torch.meshgrid(*args, indexing=indexing)]


def sub2ind2d(siz, subs, **kwargs):
    """neurite/tf/utils/utils.py:1068-1082 (row-major flat index)."""
    assert len(siz) == len(subs), 'found inconsistent siz and subs: %d %d' % (len(siz), len(subs))
    k = np.cumprod(list(siz)[::-1])
    ndx = subs[-1]
    for i, v in enumerate(list(subs[:-1])[::-1]):
        ndx = ndx + v * int(k[i])
    return ndx


def prod_n(lst):
    """neurite/tf/utils/utils.py:1085-1092."""
    prod = lst[0]
    for p in lst[1:]:
        prod = prod * p
    return prod


def batch_channel_flatten(x):
    """neurite/tf/utils/utils.py:1175-1188: [B, ..., C] -> [B, V, C] (a view, never a copy)."""
    return flatten_axes(x, range(1, x.dim() - 1))


flatten_batch_channel = batch_channel_flatten


def flatten_axes(x, axes):
    """neurite/tf/utils/utils.py:1195-1226."""
    assert isinstance(axes, (list, tuple, range)), 'axes must be list or tuple of axes to be flattened'
    assert np.all(np.diff(axes) == 1), 'axes need to be contiguous'
    if axes[0] < 0:
        assert axes[-1] < 0, 'if one axis is negative, all have to be negative'
    assert axes[-1] < x.dim(), 'axis %d outside max axis %d' % (axes[-1], x.dim() - 1)
    shp = list(x.shape)
    new = shp[:axes[0]] + [-1]
    if axes[-1] < x.dim() - 1 and not (axes[-1] == -1):
        new += shp[axes[-1] + 1:]
    return x.reshape(new)


# --------------------------------------------------------------------------------------
# affine sampling helpers of voxelmorph that neurite's labels_to_image_new instantiates (neurite/tf/models.py:1068-1112:
# vxm.layers.DrawAffineParams, ParamsToAffineMatrix, vxm.utils.draw_flip_matrix / draw_swap_matrix).  voxelmorph is not
# vendored in the reference tree; these follow its published interface.  Host-side: a handful of numbers per batch entry.
# --------------------------------------------------------------------------------------

def _host_generator(seed):
    g = torch.Generator(device='cpu')
    if seed is None:
        g.seed()
    else:
        g.manual_seed(int(seed) % (2 ** 63 - 1))
    return g


def draw_affine_params(shift=None, rot=None, scale=None, shear=None, normal_shift=False, normal_rot=False,
                       normal_scale=False, normal_shear=False, shift_scale=False, ndims=3, batch_shape=None, concat=True,
                       dtype=torch.float32, seeds={}):
    """
    Draw translation, rotation, scaling and shearing parameters: uniformly in [-bound, bound] or, with normal_*, normally
    with the bound as SD (scaling: truncated at two SDs).  Returns [*batch_shape, M] with M = 12 (3-D) or 6 (2-D), ordered
    shift, rot, scale, shear; scaling parameters are deviations from 1 unless shift_scale.
    """
    assert ndims in (2, 3), 'only 2D and 3D supported'
    splits = dict(shift=ndims, rot=3 if ndims == 3 else 1, scale=ndims, shear=3 if ndims == 3 else 1)
    bounds = dict(shift=shift, rot=rot, scale=scale, shear=shear)
    normal = dict(shift=normal_shift, rot=normal_rot, scale=normal_scale, shear=normal_shear)
    batch_shape = [] if batch_shape is None else [int(b) for b in np.ravel(batch_shape)]
    par = {}
    for key, n in splits.items():
        lim = np.ravel(0 if bounds[key] is None else bounds[key]).astype(np.float64)
        if lim.size == 1:
            lim = np.repeat(lim, n)
        assert lim.size == n, f'unexpected number of {key} bounds {lim.size}, expected 1 or {n}'
        lim = torch.as_tensor(lim, dtype=torch.float64)
        gen = _host_generator(seeds.get(key) if isinstance(seeds, dict) else None)
        shp = batch_shape + [n]
        if normal[key]:
            draw = torch.randn(shp, generator=gen, dtype=torch.float64)
            if key == 'scale':                                  # tf.random.truncated_normal: re-draw beyond two SDs
                for _ in range(64):
                    bad = draw.abs() > 2
                    if not bool(bad.any()):
                        break
                    draw = torch.where(bad, torch.randn(shp, generator=gen, dtype=torch.float64), draw)
                draw = draw.clamp(-2, 2)
            draw = draw * lim
        else:
            draw = (torch.rand(shp, generator=gen, dtype=torch.float64) * 2 - 1) * lim
        par[key] = draw.to(dtype)
    if shift_scale:
        par['scale'] = par['scale'] + 1
    return torch.cat(list(par.values()), -1) if concat else par


def angles_to_rotation_matrix(ang, deg=True, ndims=3):
    """Rotation matrices [..., N, N] from angles [..., 1] (2-D) or [..., 3] (3-D, intrinsic R = Rx @ Ry @ Rz)."""
    assert ndims in (2, 3), 'only 2D and 3D supported'
    ang = torch.as_tensor(ang)
    if ang.dim() == 0:
        ang = ang.reshape(1)
    n = 1 if ndims == 2 else 3
    if ang.shape[-1] < n:
        ang = torch.cat([ang, ang.new_zeros(ang.shape[:-1] + (n - ang.shape[-1],))], -1)
    ang = ang[..., :n]
    if deg:
        ang = ang * (np.pi / 180)
    c, s = torch.cos(ang), torch.sin(ang)
    one, zero = torch.ones_like(c[..., 0]), torch.zeros_like(c[..., 0])

    def mat(rows):
        return torch.stack([torch.stack(r, -1) for r in rows], -2)
    if ndims == 2:
        return mat([[c[..., 0], -s[..., 0]], [s[..., 0], c[..., 0]]])
    rx = mat([[one, zero, zero], [zero, c[..., 0], -s[..., 0]], [zero, s[..., 0], c[..., 0]]])
    ry = mat([[c[..., 1], zero, s[..., 1]], [zero, one, zero], [-s[..., 1], zero, c[..., 1]]])
    rz = mat([[c[..., 2], -s[..., 2], zero], [s[..., 2], c[..., 2], zero], [zero, zero, one]])
    return rx @ ry @ rz


def params_to_affine_matrix(par, deg=True, shift_scale=False, last_row=False, ndims=3):
    """
    Affine matrices [..., N, N+1] (N+1 rows with last_row) from parameter vectors ordered shift, rot, scale, shear
    (missing trailing parameters are neutral): T @ R @ diag(scale) @ shear with an upper-triangular shear matrix.
    """
    assert ndims in (2, 3), 'only 2D and 3D supported'
    par = torch.as_tensor(par)
    if not par.dtype.is_floating_point:
        par = par.to(torch.float32)
    nrot = 3 if ndims == 3 else 1
    widths = (ndims, nrot, ndims, nrot)
    total = sum(widths)
    if par.shape[-1] > total:
        raise ValueError(f'number of params exceeds value {total} expected for dimensionality')
    if par.shape[-1] < total:
        pad = par.new_zeros(par.shape[:-1] + (total - par.shape[-1],))
        if par.shape[-1] <= ndims + nrot and not shift_scale:      # scaling parameters were not given at all: neutral = 1
            pad[..., ndims + nrot - par.shape[-1]:2 * ndims + nrot - par.shape[-1]] = 1
        par = torch.cat([par, pad], -1)
    shift, rot, scale, shear = torch.split(par, widths, -1)
    if shift_scale:
        scale = scale + 1
    m_rot = angles_to_rotation_matrix(rot, deg=deg, ndims=ndims)
    m_scale = torch.diag_embed(scale)
    m_shear = torch.eye(ndims, dtype=par.dtype, device=par.device).expand(par.shape[:-1] + (ndims, ndims)).clone()
    iu = np.triu_indices(ndims, k=1)
    for k, (i, j) in enumerate(zip(*iu)):
        m_shear[..., i, j] = shear[..., k]
    out = torch.cat([m_rot @ m_scale @ m_shear, shift[..., None]], -1)
    if last_row:
        row = out.new_zeros(out.shape[:-2] + (1, ndims + 1))
        row[..., -1] = 1
        out = torch.cat([out, row], -2)
    return out


def draw_flip_matrix(grid_shape, shift_center=True, last_row=True, dtype=torch.float32, seed=None):
    """Matrix that flips each axis of an N-D grid with probability 1/2 (about the grid centre unless shift_center)."""
    ndims = len(grid_shape)
    bit = (torch.randn(ndims, generator=_host_generator(seed)) > 0).to(dtype)
    out = torch.zeros(ndims, ndims + 1, dtype=dtype)
    out[:, :ndims] = torch.diag(1 - 2 * bit)
    if not shift_center:
        out[:, -1] = (torch.as_tensor(np.asarray(grid_shape), dtype=dtype) - 1) * bit
    if last_row:
        out = torch.cat([out, torch.tensor([[0.] * ndims + [1.]], dtype=dtype)], 0)
    return out


def draw_swap_matrix(ndims, last_row=True, dtype=torch.float32, seed=None):
    """Matrix that randomly permutes the axes of N-D space."""
    perm = torch.randperm(ndims, generator=_host_generator(seed))
    out = torch.zeros(ndims, ndims + 1, dtype=dtype)
    out[torch.arange(ndims), perm] = 1
    if last_row:
        out = torch.cat([out, torch.tensor([[0.] * ndims + [1.]], dtype=dtype)], 0)
    return out


def _axis_gather(x, index, ax):
    """y = tf.gather(x, index, axis=ax) of a float32 device tensor (csrc/synth.hip)."""
    lib = _lib.lib()
    dev = _lib.require_device(x)
    x = x.contiguous()
    idx = torch.as_tensor(np.asarray(index, dtype=np.int32)).to(dev)
    outer = int(np.prod(x.shape[:ax])) if ax else 1
    inner = int(np.prod(x.shape[ax + 1:]))
    y = torch.empty(tuple(x.shape[:ax]) + (idx.numel(),) + tuple(x.shape[ax + 1:]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_synth_axis_gather_f32(_lib.ptr(x), _lib.ptr(idx), _lib.ptr(y), outer, x.shape[ax], idx.numel(), inner,
                                           _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_synth_axis_gather_f32')
    return y


def subsample_axis_indices(width, thick):
    """index lists of utils.subsample_axis (utils.py:815-824): slices kept, and the map back to `width` samples"""
    num_slice = int(np.float32(width) / np.float32(thick) + np.float32(0.5))
    # tf.linspace re-casts integer end points to the dtype of its step, float64: the indices are truncations of doubles
    down = (np.linspace(0, width - 1, num_slice, dtype=np.float64) + 0.5).astype(np.int32)
    up = (np.linspace(0, num_slice - 1, width, dtype=np.float64) + 0.5).astype(np.int32)
    return down, up


def _subsample_draws(num_axes, stride_min, stride_max, prob, seed, draws=None):
    """(index into the axis list, slice thickness) of utils.subsample_axis (utils.py:801-813), in the reference's order of
    draws; `draws` (tests): uniform [0, 1) numbers to use instead of the generator's"""
    gen = _host_generator(seed)
    draws = None if draws is None else list(draws)

    def uniform():
        return np.float32(draws.pop(0)) if draws is not None else np.float32(float(torch.rand((), generator=gen)))
    ax = int(np.floor(np.float64(uniform()) * num_axes))
    thick = uniform() * np.float32(np.float32(stride_max) - np.float32(stride_min)) + np.float32(stride_min)
    assert 0 <= prob <= 1, f'{prob} not a probability'
    if prob < 1:
        bit = np.float32(uniform() < np.float32(prob))
        thick = thick * bit + (np.float32(1) - bit)
    return ax, thick


def subsample_axis(x, stride_min=1, stride_max=8, axes=None, prob=1, upsample=True, seed=None, _draws=None):
    """
    Symmetrically subsample a tensor by a random factor (stride) along one randomly drawn axis with nearest-neighbour
    interpolation and optionally up-sample it again (utils.py:754-826).  float32 device tensors.
    """
    if x.dtype != torch.float32:
        raise NotImplementedError('subsample_axis: float32 tensors, got %s' % x.dtype)
    num_dim = x.dim()
    if axes is None:
        axes = range(num_dim)
    if np.isscalar(axes):
        axes = [axes]
    axes = list(axes)
    assert all(i in range(num_dim) for i in axes), 'invalid axis passed'
    assert 0 < stride_min and stride_min <= stride_max, 'invalid strides'
    ax, thick = _subsample_draws(len(axes), stride_min, stride_max, prob, seed, _draws)
    ax = axes[ax]
    width = x.shape[ax]
    down, up = subsample_axis_indices(width, thick)
    if upsample:
        return _axis_gather(x, down[up], ax)                 # the two gathers of the reference composed into one
    return _axis_gather(x, down, ax)
