"""
neurite_amd.metrics -- Dice and label-weighted categorical cross-entropy on MI355X.

Dice / SoftDice / HardDice       neurite/tf/metrics.py:339-616
CategoricalCrossentropy          neurite/tf/metrics.py:619-650 (+ tf.keras.losses.CategoricalCrossentropy)
WeightedCategoricalCrossentropy  alias (the name BASELINE.json uses; not in the reference tree)

Same constructor arguments, defaults, method names, return shapes, asserts and warnings as the
reference.  All voxel-sized work runs in csrc/dice.hip and csrc/cce.hip through the C ABI;
what remains here is argument handling and arithmetic on [B, L]-sized results.
"""

import warnings

import numpy as np
import torch

from . import _lib
from . import utils
from .errors import InvalidArgumentError

__all__ = ['Dice', 'SoftDice', 'HardDice', 'CategoricalCrossentropy', 'WeightedCategoricalCrossentropy',
           'dice_partial_sums']

_INT_DTYPES = (torch.int8, torch.uint8, torch.int16, torch.int32, torch.int64, torch.bool)


def _as_f32(x, what):
    if x.dtype != torch.float32:
        if x.dtype in (torch.float16, torch.bfloat16, torch.float64):
            raise NotImplementedError('%s: the HIP Dice path takes float32 probability maps, got %s'
                                      % (what, x.dtype))
        raise TypeError('%s: expected a float32 probability / one-hot map, got %s' % (what, x.dtype))
    return x.contiguous()


def dice_partial_sums(y_true, y_pred, normalize=False, laplace_smoothing=0.):
    """
    One pass over two [B, ..., L] float32 maps on the GPU.
    Returns (sums [B, 3, L] = sum t*p, sum t^2, sum p^2;  dice [B, L];  minmax [4] = min t, max t, min p, max p).
    `sums` is the quantity to all-reduce when a batch entry is split across ranks.
    """
    lib = _lib.lib()
    dev = _lib.require_device(y_true, y_pred)
    t = _as_f32(y_true, 'y_true')
    p = _as_f32(y_pred, 'y_pred')
    if t.shape != p.shape:
        raise ValueError('y_true and y_pred must have the same shape, got %s and %s'
                         % (tuple(t.shape), tuple(p.shape)))
    B, L = t.shape[0], t.shape[-1]
    V = t.numel() // max(B * L, 1)
    sums = torch.empty((B, 3, L), dtype=torch.float32, device=dev)
    dice = torch.empty((B, L), dtype=torch.float32, device=dev)
    minmax = torch.empty((4,), dtype=torch.float32, device=dev)
    if t.numel() == 0:          # no voxels: all sums are zero; the finalize kernel still does the division
        if B * L == 0:
            return sums, dice, minmax.fill_(0)
        sums.zero_()
        with torch.cuda.device(dev):
            rc = lib.nrt_dice_from_sums_f32(_lib.ptr(sums), L, B, float(laplace_smoothing), _lib.ptr(dice),
                                            _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_dice_from_sums_f32')
        minmax.copy_(torch.tensor([float('inf'), float('-inf'), float('inf'), float('-inf')]))
        return sums, dice, minmax
    nws = lib.nrt_dice_workspace_bytes(V, L, B)
    ws = _lib.workspace(dev, nws)
    with torch.cuda.device(dev):
        rc = lib.nrt_dice_soft_f32(_lib.ptr(t), _lib.ptr(p), V, L, B, int(bool(normalize)),
                                   float(laplace_smoothing), _lib.ptr(sums), _lib.ptr(dice), _lib.ptr(minmax),
                                   _lib.ptr(ws), nws, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_dice_soft_f32')
    return sums, dice, minmax


class _SoftDiceFn(torch.autograd.Function):
    """Soft Dice [B, L] with the backward of csrc/backward.hip (gradients wrt both maps)."""

    @staticmethod
    def forward(ctx, y_true, y_pred, eps, normalize, check_limits):
        t = _as_f32(y_true, 'y_true')
        p = _as_f32(y_pred, 'y_pred')
        sums, d, mm = dice_partial_sums(t, p, normalize, eps)
        if check_limits:
            _check_limits(mm)
        ctx.save_for_backward(t, p, sums)
        ctx.eps, ctx.normalize = eps, normalize
        return d

    @staticmethod
    def backward(ctx, grad_dice):
        if ctx.normalize:
            raise NotImplementedError('neurite_amd: backward of Dice(normalize=True) is not implemented')
        t, p, sums = ctx.saved_tensors
        lib = _lib.lib()
        dev = t.device
        B, L = t.shape[0], t.shape[-1]
        V = t.numel() // max(B * L, 1)
        g = grad_dice.to(torch.float32).contiguous()
        gt = torch.empty_like(t) if ctx.needs_input_grad[0] else None
        gp = torch.empty_like(p) if ctx.needs_input_grad[1] else None
        if t.numel() and (gt is not None or gp is not None):
            with torch.cuda.device(dev):
                rc = lib.nrt_dice_soft_bwd_f32(_lib.ptr(t), _lib.ptr(p), _lib.ptr(sums), _lib.ptr(g), V, L, B,
                                               float(ctx.eps), _lib.ptr(gp), _lib.ptr(gt), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_dice_soft_bwd_f32')
        return gt, gp, None, None, None


class _WcceFn(torch.autograd.Function):
    """Weighted CCE: returns (sum of the per-voxel losses [1]) or the per-voxel losses; backward wrt y_pred."""

    @staticmethod
    def forward(ctx, t, p, w, from_logits, label_smoothing, per_voxel):
        lib = _lib.lib()
        dev = p.device
        yf = p.shape[-1]
        N = p.numel() // max(yf, 1)
        loss_sum = torch.empty((1,), dtype=torch.float32, device=dev)
        pv = torch.empty(p.shape[:-1], dtype=torch.float32, device=dev) if per_voxel else None
        nws = lib.nrt_wcce_workspace_bytes(N, yf)
        ws = _lib.workspace(dev, nws)
        dt = _lib.DT_F32 if p.dtype == torch.float32 else _lib.DT_BF16
        with torch.cuda.device(dev):
            rc = lib.nrt_wcce(_lib.ptr(t), _lib.ptr(p), dt, _lib.ptr(w), N, yf, int(from_logits),
                              float(label_smoothing), _lib.ptr(loss_sum), _lib.ptr(pv), _lib.ptr(ws), nws,
                              _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_wcce')
        ctx.save_for_backward(t, p, w)
        ctx.cfg = (from_logits, label_smoothing, per_voxel)
        return pv if per_voxel else loss_sum

    @staticmethod
    def backward(ctx, grad):
        t, p, w = ctx.saved_tensors
        from_logits, label_smoothing, per_voxel = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise NotImplementedError('neurite_amd: gradient of the CCE wrt y_true is not implemented')
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None
        if p.dtype != torch.float32:
            raise NotImplementedError('neurite_amd: CCE backward takes float32 inputs')
        lib = _lib.lib()
        dev = p.device
        yf = p.shape[-1]
        N = p.numel() // max(yf, 1)
        g = grad.to(torch.float32).contiguous()
        gp = torch.empty_like(p)
        if N:
            with torch.cuda.device(dev):
                rc = lib.nrt_wcce_bwd_f32(_lib.ptr(t), _lib.ptr(p), _lib.ptr(w), None if per_voxel else _lib.ptr(g),
                                          _lib.ptr(g) if per_voxel else None, N, yf, int(from_logits),
                                          float(label_smoothing), 1.0, _lib.ptr(gp), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_wcce_bwd_f32')
        return None, gp, None, None, None, None


def _check_limits(minmax):
    mn_t, mx_t, mn_p, mx_p = [float(v) for v in minmax.tolist()]      # one device->host sync
    msg = 'value outside range'
    if not (mn_t >= 0. and mn_p >= 0. and mx_t <= 1. and mx_p <= 1.):  # also catches NaN
        raise InvalidArgumentError(msg)


class Dice:
    """
    Dice of two Tensors; 'soft' and 'hard', weighting per label (or per batch entry).
    Arguments exactly as neurite/tf/metrics.py:352-359.
    """

    def __init__(self, dice_type='soft', input_type='prob', nb_labels=None, weights=None,
                 check_input_limits=True, laplace_smoothing=0., normalize=False):
        self.dice_type = dice_type
        self.input_type = input_type
        self.nb_labels = nb_labels
        self.weights = weights
        self.normalize = normalize
        self.check_input_limits = check_input_limits
        self.laplace_smoothing = laplace_smoothing

        assert self.input_type in ['prob', 'max_label']                                   # metrics.py:406
        if self.dice_type == 'hard' and self.input_type == 'max_label':                   # :408-409
            assert self.nb_labels is not None, 'If doing hard Dice need nb_labels'
        if self.dice_type == 'soft':                                                      # :411-413
            assert self.input_type in ['prob', 'one_hot'], \
                'if doing soft Dice, must use probabilistic (one_hot)encoding'

    # ------------------------------------------------------------------------------------------
    def dice(self, y_true, y_pred):
        """
        y_true, y_pred: [B, ..., nb_labels] (prob / one-hot) or [B, ...] (max_label).
        Returns [B, nb_labels] float32 (neurite/tf/metrics.py:415-482).
        """
        lib = _lib.lib()
        dev = _lib.require_device(y_true, y_pred)
        eps = float(self.laplace_smoothing)

        if self.dice_type != 'hard':
            if torch.is_grad_enabled() and (y_true.requires_grad or y_pred.requires_grad):
                return _SoftDiceFn.apply(y_true, y_pred, eps, bool(self.normalize), bool(self.check_input_limits))
            _, d, mm = dice_partial_sums(y_true, y_pred, self.normalize, eps)
            if self.check_input_limits:                                                   # :439-444
                _check_limits(mm)
            return d

        # ---- hard Dice (:450-468): integer counting, bit-exact -------------------------------
        if self.input_type == 'prob':
            warnings.warn('You are using ne.metrics.Dice with probabilistic inputs'
                          'and computing *hard* dice. \n For this, we use argmax to'
                          'get the optimal label at each location, which is not'
                          'differentiable. Do not use expecting gradients.')
            t = _as_f32(y_true, 'y_true')
            p = _as_f32(y_pred, 'y_pred')
            if t.shape != p.shape:
                raise ValueError('y_true and y_pred must have the same shape')
            if self.nb_labels is None:                                                    # :460-461
                self.nb_labels = p.shape[-1]
            if self.nb_labels != p.shape[-1]:
                raise NotImplementedError('hard Dice on prob maps with nb_labels != last dimension')
            if self.normalize or self.check_input_limits:
                # the arg-max is invariant to the (positive) per-voxel normalisation; the range
                # asserts of :439-444 still apply to the probabilistic inputs
                _, _, mm = dice_partial_sums(t, p, self.normalize, 0.)
                if self.check_input_limits:
                    _check_limits(mm)
                elif float(mm[0]) < 0 or float(mm[2]) < 0:
                    raise NotImplementedError('normalize=True with negative inputs on the hard path')
            B, L = t.shape[0], t.shape[-1]
            V = t.numel() // max(B * L, 1)
            counts = torch.empty((B, 3, L), dtype=torch.int64, device=dev)
            d = torch.empty((B, L), dtype=torch.float32, device=dev)
            nws = lib.nrt_dice_workspace_bytes(V, L, B)
            ws = _lib.workspace(dev, nws)
            with torch.cuda.device(dev):
                rc = lib.nrt_dice_hard_prob_f32(_lib.ptr(t), _lib.ptr(p), V, L, B, eps, _lib.ptr(counts),
                                                _lib.ptr(d), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_dice_hard_prob_f32')
            return d

        # max_label inputs: label id at every location (tf.one_hot needs integer indices)
        for name, y in (('y_true', y_true), ('y_pred', y_pred)):
            if y.dtype not in _INT_DTYPES:
                raise TypeError('%s: max_label inputs must hold integer label ids (tf.one_hot requires an '
                                'integer tensor), got %s' % (name, y.dtype))
        if y_true.shape != y_pred.shape:
            raise ValueError('y_true and y_pred must have the same shape')
        t = y_true.to(torch.int32).contiguous()
        p = y_pred.to(torch.int32).contiguous()
        B = t.shape[0]
        V = t.numel() // max(B, 1)
        L = int(self.nb_labels)
        counts = torch.empty((B, 3, L), dtype=torch.int64, device=dev)
        d = torch.empty((B, L), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.nrt_dice_hard_label_i32(_lib.ptr(t), _lib.ptr(p), V, L, B, eps, _lib.ptr(counts),
                                             _lib.ptr(d), None, 0, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_dice_hard_label_i32')
        return d

    def mean_dice(self, y_true, y_pred):
        """mean dice across all patches and labels, optionally weighted (neurite/tf/metrics.py:484-510)."""
        dice_metric = self.dice(y_true, y_pred)
        if self.weights is not None:                                                      # :502-505
            w = self.weights
            assert len(w.shape) == 2, 'weights should be a matrix broadcastable to [batch_size, nb_labels]'
            dice_metric = dice_metric * torch.as_tensor(np.asarray(w) if not isinstance(w, torch.Tensor) else w,
                                                        dtype=torch.float32).to(dice_metric.device)
        mean_dice_metric = dice_metric.mean()                                             # :508
        if not bool(torch.isfinite(mean_dice_metric)):                                    # :509
            raise InvalidArgumentError('metric not finite')
        return mean_dice_metric

    def loss(self, y_true, y_pred):
        """Deprecated in the reference (neurite/tf/metrics.py:512-519)."""
        warnings.warn('ne.metrics.*.loss functions are deprecated.'
                      'Please use the ne.losses.*.loss functions.')
        return - self.mean_dice(y_true, y_pred)


class SoftDice(Dice):
    """neurite/tf/metrics.py:522-560."""

    def __init__(self, weights=None, check_input_limits=True, laplace_smoothing=0., normalize=False):
        super().__init__(dice_type='soft', input_type='prob', weights=weights,
                         check_input_limits=check_input_limits, laplace_smoothing=laplace_smoothing,
                         normalize=normalize)


class HardDice(Dice):
    """neurite/tf/metrics.py:563-616."""

    def __init__(self, nb_labels, input_type='max_label', weights=None, check_input_limits=True,
                 laplace_smoothing=0., normalize=False):
        super().__init__(dice_type='hard', input_type=input_type, nb_labels=nb_labels, weights=weights,
                         check_input_limits=check_input_limits, laplace_smoothing=laplace_smoothing,
                         normalize=normalize)


class CategoricalCrossentropy:
    """
    tf.keras.losses.CategoricalCrossentropy with label_weights as an explicit parameter
    (neurite/tf/metrics.py:619-650).  Keras keyword arguments honoured: from_logits,
    label_smoothing, reduction ('auto' / 'sum_over_batch_size' / 'sum' / 'none'), name; axis must be -1.
    Inputs may be float32 or bfloat16 (arithmetic is float32 either way); returns a float32 scalar
    (or the per-element losses for reduction='none').
    """

    def __init__(self, label_weights=None, **kwargs):
        self.label_weights = None
        if label_weights is not None:
            self.label_weights = torch.as_tensor(np.asarray(label_weights, dtype=np.float32)
                                                 if not isinstance(label_weights, torch.Tensor)
                                                 else label_weights)
        self.from_logits = bool(kwargs.pop('from_logits', False))
        self.label_smoothing = float(kwargs.pop('label_smoothing', 0.))
        self.reduction = kwargs.pop('reduction', 'auto')
        self.name = kwargs.pop('name', 'categorical_crossentropy')
        axis = kwargs.pop('axis', -1)
        if axis != -1:
            raise NotImplementedError('CategoricalCrossentropy: only axis=-1 (channels-last) is supported')
        if kwargs:
            raise TypeError('unexpected keyword arguments: %s' % sorted(kwargs))
        if self.reduction not in ('auto', 'sum_over_batch_size', 'sum', 'none'):
            raise ValueError('Invalid Reduction Key: %s' % self.reduction)

    def __call__(self, y_true, y_pred, sample_weight=None):
        return self.cce(y_true, y_pred, sample_weight=sample_weight)

    def cce(self, y_true, y_pred, sample_weight=None):
        yf = y_pred.shape[-1]
        if self.label_weights is not None:
            lf = self.label_weights.shape[-1]
            if yf != lf:                                                                  # metrics.py:644-645
                raise ValueError(f'Label weights must be of len {yf}, but got {lf}.')
        lib = _lib.lib()
        dev = _lib.require_device(y_true, y_pred)
        if y_true.shape != y_pred.shape:
            raise ValueError('y_true and y_pred must have the same shape')
        if y_pred.dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError('CategoricalCrossentropy: float32 or bfloat16 inputs, got %s' % y_pred.dtype)
        p = y_pred.contiguous()
        t = y_true.to(p.dtype).contiguous()           # keras casts y_true to y_pred's dtype
        w = None if self.label_weights is None else self.label_weights.to(dev, torch.float32).contiguous()
        N = p.numel() // max(yf, 1)
        need_pv = sample_weight is not None or self.reduction == 'none'
        res = _WcceFn.apply(t, p, w, self.from_logits, self.label_smoothing, need_pv)
        if not need_pv:
            return res[0] if self.reduction == 'sum' else res[0] / N
        losses = res
        if sample_weight is not None:
            sw = torch.as_tensor(sample_weight, dtype=torch.float32, device=dev)
            while sw.dim() < losses.dim():
                sw = sw.unsqueeze(-1)
            losses = losses * sw
        if self.reduction == 'none':
            return losses
        if self.reduction == 'sum':
            return losses.sum()
        return losses.sum() / losses.numel()


WeightedCategoricalCrossentropy = CategoricalCrossentropy
