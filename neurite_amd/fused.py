"""
neurite_amd.fused -- SpatialTransformer + soft Dice in one pass over HBM.

    d = ne.fused.warp_dice(moving, trf, fixed)                # [B, L]
is numerically the pipeline
    warped = ne.layers.SpatialTransformer()([moving, trf]);  d = ne.metrics.Dice(check_input_limits=False).dice(fixed, warped)
(neurite/tf/models.py:806-807 + neurite/tf/metrics.py:415-482) but never writes `warped`: the blended row is
consumed in registers (csrc/fused.hip).  This is the Dice-of-a-warped-segmentation loss/metric of
VoxelMorph-style training; the reference has no fused form (TensorFlow materialises every intermediate).
"""

import torch

from . import _lib
from . import utils
from .errors import InvalidArgumentError

__all__ = ['warp_dice']


class _WarpDiceFn(torch.autograd.Function):
    """dice [B, L] as a function of the (already 'ij'-ordered) displacement field; backward in csrc/backward.hip."""

    @staticmethod
    def forward(ctx, shift, run, run_backward):
        ctx.run_backward = run_backward
        return run()

    @staticmethod
    def backward(ctx, grad_dice):
        return ctx.run_backward(grad_dice), None, None


def warp_dice(moving, trf, fixed, indexing='ij', single_transform=False, fill_value=None, laplace_smoothing=0.,
              check_input_limits=False, return_warped=False, return_sums=False, _tune=0):
    """
    moving [B, X, Y, Z, L], trf [B, X', Y', Z', 3] (voxel displacements), fixed [B, X', Y', Z', L]; float32 maps (or both maps
    stored as bfloat16: float32 arithmetic on the widened values, exact for one-hot label maps, half the bytes), 3-D,
    L in {4, 8, 16, 32, 64, 128, 256}.  Linear interpolation.  Returns dice [B, L] (optionally also the warped
    volume and the partial sums [B, 3, L]).  check_input_limits defaults to False because a tri-linearly
    warped one-hot map exceeds 1.0 by an ulp (see tests); pass True for the reference's asserts (one read-back of the extrema per
    call), or 'deferred' for the same asserts without the host round trip: the values come back as a `checked.CheckedTensor`
    (neurite_amd/checked.py) that raises when they are brought to the host, or at a later call once the extrema have arrived.
    """
    lib = _lib.lib()
    dev = _lib.require_device(moving, trf, fixed)
    if moving.dim() != 5 or fixed.dim() != 5 or trf.dim() != 5 or trf.shape[-1] != 3:
        raise Exception('warp_dice expects 3-D volumes [B, X, Y, Z, L] and a displacement field [B, X, Y, Z, 3]')
    if indexing not in ('ij', 'xy'):
        raise ValueError("indexing has to be 'ij' (matrix) or 'xy' (cartesian)")
    if moving.dtype != fixed.dtype or moving.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError('warp_dice takes two float32 maps, or two bfloat16 maps (bf16 STORAGE, float32 arithmetic: '
                                  'exact for one-hot label maps, half the bytes per row); got %s and %s'
                                  % (moving.dtype, fixed.dtype))
    bf16 = moving.dtype == torch.bfloat16
    if bf16 and return_warped:
        raise NotImplementedError('warp_dice on bfloat16 maps does not return the warped volume (use layers.SpatialTransformer)')
    B, L = moving.shape[0], moving.shape[-1]
    if L % 4 or (L // 4) not in (1, 2, 4, 8, 16, 32, 64):
        raise NotImplementedError('warp_dice: nb_labels must be 4 * 2^k, got %d (use the unfused layers)' % L)
    if fixed.shape[0] != B or fixed.shape[-1] != L or tuple(fixed.shape[1:-1]) != tuple(trf.shape[1:-1]):
        raise ValueError('fixed must be [B, *trf_spatial, L]')
    if not single_transform and trf.shape[0] != B:
        raise ValueError('batch size of the transform does not match the volume')
    mov = moving.contiguous()
    fix = fixed.contiguous()
    # the kernels read whole 16-byte row pieces: a view whose storage offset is not a multiple of 16 bytes is copied once
    if mov.data_ptr() % 16:
        mov = mov.clone()
    if fix.data_ptr() % 16:
        fix = fix.clone()
    shift = trf.to(torch.float32)
    if indexing == 'xy':
        shift = torch.cat([shift[..., 1:2], shift[..., 0:1], shift[..., 2:]], -1)
    shift = (shift[:1] if single_transform else shift).contiguous()
    S = list(mov.shape[1:-1])
    O = list(fix.shape[1:-1])
    sums = torch.empty((B, 3, L), dtype=torch.float32, device=dev)
    dice = torch.empty((B, L), dtype=torch.float32, device=dev)
    minmax = torch.empty((4,), dtype=torch.float32, device=dev)
    warped = torch.empty_like(fix) if return_warped else None
    if bf16 and torch.is_grad_enabled() and trf.requires_grad:
        raise NotImplementedError('warp_dice on bfloat16 maps has no backward; pass float32 maps for registration training')
    o_shape = _lib.ints(O)
    nws = lib.nrt_warp_dice_workspace_bytes(o_shape, L, B, int(_tune))
    ws = _lib.workspace(dev, nws)
    has_fill = fill_value is not None

    loc_bs = 0 if single_transform else shift[0].numel()

    def run():
        with torch.cuda.device(dev):
            if bf16:
                rc = lib.nrt_warp_dice_soft_bf16(_lib.ptr(mov), _lib.ptr(shift), _lib.ptr(fix), _lib.ints(S), o_shape, L, B,
                                                 loc_bs, _lib.LOC_SHIFT, int(has_fill), float(fill_value) if has_fill else 0.0,
                                                 float(laplace_smoothing), _lib.ptr(sums), _lib.ptr(dice),
                                                 _lib.ptr(minmax) if check_input_limits else None,
                                                 int(_tune), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
            else:
                rc = lib.nrt_warp_dice_soft_f32(_lib.ptr(mov), _lib.ptr(shift), _lib.ptr(fix), _lib.ptr(warped),
                                                _lib.ints(S), o_shape, L, B, loc_bs, _lib.LOC_SHIFT,
                                                int(has_fill), float(fill_value) if has_fill else 0.0,
                                                float(laplace_smoothing), _lib.ptr(sums), _lib.ptr(dice),
                                                _lib.ptr(minmax) if check_input_limits else None,
                                                int(_tune), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_warp_dice_soft')
        if check_input_limits == 'deferred':
            from . import checked
            return checked.wrap(dice, minmax, 'fused warp + Dice')
        if check_input_limits:
            mn_t, mx_t, mn_p, mx_p = [float(v) for v in minmax.tolist()]
            if not (mn_t >= 0. and mn_p >= 0. and mx_t <= 1. and mx_p <= 1.):
                raise InvalidArgumentError('value outside range')
        return dice

    def run_backward(grad_dice):
        gshift = torch.empty((B,) + tuple(shift.shape[1:]), dtype=torch.float32, device=dev)
        g = grad_dice.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            rc = lib.nrt_warp_dice_bwd_f32(_lib.ptr(mov), _lib.ptr(shift), _lib.ptr(fix), _lib.ptr(sums), _lib.ptr(g),
                                           _lib.ptr(gshift), _lib.ints(S), o_shape, L, B, loc_bs, _lib.LOC_SHIFT,
                                           int(has_fill), float(laplace_smoothing), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_warp_dice_bwd_f32')
        return gshift.sum(0, keepdim=True) if single_transform else gshift

    if torch.is_grad_enabled() and (moving.requires_grad or fixed.requires_grad):
        raise NotImplementedError('warp_dice: only the transform is differentiable in the fused form; use '
                                  'layers.SpatialTransformer + metrics.Dice for gradients wrt the volumes')
    if torch.is_grad_enabled() and shift.requires_grad:
        if check_input_limits == 'deferred':
            check_input_limits = True                  # (a graph is being recorded: the assert is looked at before the node exists)
        d = _WarpDiceFn.apply(shift, run, run_backward)
    else:
        d = run()
    out = (d,)
    if return_warped:
        out += (warped,)
    if return_sums:
        out += (sums,)
    return out[0] if len(out) == 1 else out
