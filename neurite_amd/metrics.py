"""
neurite_amd.metrics -- Dice and label-weighted categorical cross-entropy on MI355X.

Dice / SoftDice / HardDice       neurite/tf/metrics.py:339-616
CategoricalCrossentropy          neurite/tf/metrics.py:619-650 (+ tf.keras.losses.CategoricalCrossentropy)
WeightedCategoricalCrossentropy  alias (the name BASELINE.json uses; not in the reference tree)

Same constructor arguments, defaults, method names, return shapes, asserts and warnings as the
reference.  All voxel-sized work runs in csrc/dice.hip and csrc/cce.hip through the C ABI;
what remains here is argument handling and arithmetic on [B, L]-sized results.
"""

import threading
import warnings

import numpy as np
import torch

from . import _lib
from . import utils
from .errors import InvalidArgumentError

__all__ = ['Dice', 'SoftDice', 'HardDice', 'CategoricalCrossentropy', 'WeightedCategoricalCrossentropy',
           'dice_partial_sums', 'MutualInformation', 'MeanSquaredErrorProb', 'JointSegLoss']

_INT_DTYPES = (torch.int8, torch.uint8, torch.int16, torch.int32, torch.int64, torch.bool)


_MAP_DTYPES = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}


def _as_f32(x, what):
    """a float32 tensor for the kernels that take only float32 (backward passes): 16-bit maps are widened, which is exact"""
    if x.dtype != torch.float32:
        if x.dtype in (torch.float16, torch.bfloat16):
            return x.to(torch.float32).contiguous()
        if x.dtype == torch.float64:
            raise NotImplementedError('%s: the HIP Dice path takes float32 / bfloat16 / float16 probability maps, got float64' % what)
        raise TypeError('%s: expected a floating-point probability / one-hot map, got %s' % (what, x.dtype))
    return x.contiguous()


def _as_maps(y_true, y_pred):
    """both maps as the Dice kernels take them: contiguous, one common storage dtype (float32, bfloat16 or float16; the arithmetic
    is float32 whatever the storage, csrc/dice.hip).  Maps of different dtypes are widened to float32 first."""
    for what, x in (('y_true', y_true), ('y_pred', y_pred)):
        if x.dtype == torch.float64:
            raise NotImplementedError('%s: the HIP Dice path takes float32 / bfloat16 / float16 probability maps, got float64' % what)
        if x.dtype not in _MAP_DTYPES:
            raise TypeError('%s: expected a floating-point probability / one-hot map, got %s' % (what, x.dtype))
    if y_true.dtype != y_pred.dtype:
        y_true, y_pred = y_true.to(torch.float32), y_pred.to(torch.float32)
    return y_true.contiguous(), y_pred.contiguous(), _MAP_DTYPES[y_true.dtype]


def dice_partial_sums(y_true, y_pred, normalize=False, laplace_smoothing=0.):
    """
    One pass over two [B, ..., L] maps on the GPU (float32, or bfloat16 / float16 storage with float32 arithmetic).
    Returns (sums [B, 3, L] = sum t*p, sum t^2, sum p^2;  dice [B, L];  minmax [4] = min t, max t, min p, max p).
    `sums` is the quantity to all-reduce when a batch entry is split across ranks.
    """
    lib = _lib.lib()
    dev = _lib.require_device(y_true, y_pred)
    t, p, dt = _as_maps(y_true, y_pred)
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
        rc = lib.nrt_dice_soft(_lib.ptr(t), _lib.ptr(p), dt, V, L, B, int(bool(normalize)),
                               float(laplace_smoothing), _lib.ptr(sums), _lib.ptr(dice), _lib.ptr(minmax),
                               _lib.ptr(ws), nws, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_dice_soft')
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
        ctx.in_dtypes = (y_true.dtype, y_pred.dtype)       # 16-bit maps: float32 arithmetic, gradients rounded to the maps' dtype
        return d

    @staticmethod
    def backward(ctx, grad_dice):
        t, p, sums = ctx.saved_tensors
        lib = _lib.lib()
        dev = t.device
        B, L = t.shape[0], t.shape[-1]
        V = t.numel() // max(B * L, 1)
        g = grad_dice.to(torch.float32).contiguous()
        gt = torch.empty_like(t) if ctx.needs_input_grad[0] else None
        gp = torch.empty_like(p) if ctx.needs_input_grad[1] else None
        if t.numel() and (gt is not None or gp is not None):
            fn = lib.nrt_dice_soft_bwd_norm_f32 if ctx.normalize else lib.nrt_dice_soft_bwd_f32
            with torch.cuda.device(dev):
                rc = fn(_lib.ptr(t), _lib.ptr(p), _lib.ptr(sums), _lib.ptr(g), V, L, B, float(ctx.eps), _lib.ptr(gp),
                        _lib.ptr(gt), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_dice_soft_bwd(_norm)_f32')
        if gt is not None:
            gt = gt.to(ctx.in_dtypes[0])
        if gp is not None:
            gp = gp.to(ctx.in_dtypes[1])
        return gt, gp, None, None, None


def _wcce_launch(t, p, w, from_logits, label_smoothing, per_voxel, divide_by=None):
    """one launch of the weighted-CCE kernel: the sum of the per-voxel losses [1] (divided by `divide_by` inside the kernel when given),
    or the per-voxel losses"""
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
        if divide_by is None:
            rc = lib.nrt_wcce(_lib.ptr(t), _lib.ptr(p), dt, _lib.ptr(w), N, yf, int(from_logits),
                              float(label_smoothing), _lib.ptr(loss_sum), _lib.ptr(pv), _lib.ptr(ws), nws,
                              _lib.stream_ptr(dev))
        else:
            rc = lib.nrt_wcce_mean(_lib.ptr(t), _lib.ptr(p), dt, _lib.ptr(w), N, yf, int(from_logits),
                                   float(label_smoothing), float(divide_by), _lib.ptr(loss_sum), _lib.ptr(pv), _lib.ptr(ws), nws,
                                   _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_wcce' if divide_by is None else 'nrt_wcce_mean')
    return pv if per_voxel else loss_sum


class _WcceFn(torch.autograd.Function):
    """Weighted CCE: returns (sum of the per-voxel losses [1], divided by `divide_by` IN the kernel when given: the same float32 sum and
    float32 division as the call that builds no graph -- ADVICE r5: `sum / N` as a torch op multiplies by 1/N, one ulp apart) or the
    per-voxel losses; backward wrt y_pred."""

    @staticmethod
    def forward(ctx, t, p, w, from_logits, label_smoothing, per_voxel, divide_by=None):
        ctx.save_for_backward(t, p, w)
        ctx.cfg = (from_logits, label_smoothing, per_voxel, divide_by)
        return _wcce_launch(t, p, w, from_logits, label_smoothing, per_voxel, divide_by=divide_by)

    @staticmethod
    def backward(ctx, grad):
        t, p, w = ctx.saved_tensors
        from_logits, label_smoothing, per_voxel, divide_by = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise NotImplementedError('neurite_amd: gradient of the CCE wrt y_true is not implemented')
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None, None
        if divide_by is not None:
            grad = grad / float(divide_by)
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
        return None, gp, None, None, None, None, None


class _SegLossFn(torch.autograd.Function):
    """
    Soft Dice [B, L] and the weighted CCE sum [1] of one pair of float32 maps from ONE pass over them (csrc/segloss.hip), and one
    pass back.  `src` (models.SoftmaxSource or None): y_pred is the untouched soft-max output of a producer whose autograd inputs
    are `src_inputs`; the backward then forms d loss / d logits itself and hands it to the producer's gradient routine, and y_pred
    enters detached -- no gradient wrt the probabilities and no separate soft-max backward pass exist.
    """

    @staticmethod
    def forward(ctx, t, p, w, cfg, src, *src_inputs):
        eps, smoothing, check_limits = cfg
        lib = _lib.lib()
        dev = p.device
        B, L = p.shape[0], p.shape[-1]
        V = p.numel() // (B * L)
        sums = torch.empty((B, 3, L), dtype=torch.float32, device=dev)
        dice = torch.empty((B, L), dtype=torch.float32, device=dev)
        minmax = torch.empty((4,), dtype=torch.float32, device=dev)
        cce_sum = torch.empty((1,), dtype=torch.float32, device=dev)
        nws = lib.nrt_seg_loss_workspace_bytes(V, L, B)
        ws = _lib.workspace(dev, nws)
        with torch.cuda.device(dev):
            rc = lib.nrt_seg_loss_f32(_lib.ptr(t), _lib.ptr(p), _lib.ptr(w), V, L, B, float(smoothing), float(eps), _lib.ptr(sums),
                                      _lib.ptr(dice), _lib.ptr(minmax), _lib.ptr(cce_sum), _lib.ptr(ws), nws, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_seg_loss_f32')
        if check_limits:
            _check_limits(minmax)
        ctx.save_for_backward(t, p, w, sums)
        ctx.cfg, ctx.src = cfg, src
        return cce_sum, dice

    @staticmethod
    def backward(ctx, g_cce, g_dice):
        t, p, w, sums = ctx.saved_tensors
        eps, smoothing, _ = ctx.cfg
        src = ctx.src
        if ctx.needs_input_grad[0]:
            raise NotImplementedError('neurite_amd: the joint Dice + CCE loss has no gradient wrt y_true')
        lib = _lib.lib()
        dev = p.device
        B, L = p.shape[0], p.shape[-1]
        V = p.numel() // (B * L)
        g_cce = g_cce.to(torch.float32).contiguous()
        g_dice = g_dice.to(torch.float32).contiguous()
        grad = torch.empty_like(p)
        with torch.cuda.device(dev):
            rc = lib.nrt_seg_loss_bwd_f32(_lib.ptr(t), _lib.ptr(p), _lib.ptr(w), _lib.ptr(sums), _lib.ptr(g_dice), _lib.ptr(g_cce), V, L, B,
                                          float(smoothing), float(eps), int(src is not None), _lib.ptr(grad), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_seg_loss_bwd_f32')
        if src is None:
            return (None, grad, None, None, None)
        return (None, None, None, None, None) + tuple(src.grads_from_dz(grad, ctx.needs_input_grad[5:]))


class JointSegLoss:
    """
    One evaluation of a soft Dice object and a CategoricalCrossentropy object on the same (y_true, y_pred) -- the pair of losses
    neurite/tf/losses.py:225-246 (multiple_losses_decorator) sums for a segmentation net.  While a JointSegLoss is open (a `with`
    block on this thread), `dice_obj.dice(y_true, y_pred)` and `cce_obj.cce(y_true, y_pred)` called with THESE tensor objects take
    their numbers from one shared _SegLossFn application; everything else those methods do (weights, means, reductions, the finite
    check) is unchanged.  `JointSegLoss.open(...)` returns None when the pair does not qualify (then nothing is intercepted).
    """
    _tls = threading.local()
    applications = 0            # _SegLossFn evaluations so far (tests check that the joint path really ran)
    through_softmax = 0         # ... of which attached to the producer of the soft-max

    def __init__(self, dice_obj, cce_obj, y_true, y_pred):
        self.dice_obj, self.cce_obj, self.y_true, self.y_pred = dice_obj, cce_obj, y_true, y_pred
        self._result = None

    @classmethod
    def open(cls, dice_obj, cce_obj, y_true, y_pred):
        from .deferred import DeferredWarp
        if not (isinstance(y_true, torch.Tensor) and isinstance(y_pred, torch.Tensor)) or \
                isinstance(y_true, DeferredWarp) or isinstance(y_pred, DeferredWarp):
            return None
        if dice_obj.dice_type != 'soft' or dice_obj.normalize or cce_obj.from_logits or cce_obj.reduction == 'none':
            return None
        if not y_pred.is_cuda or y_true.device != y_pred.device or y_pred.dtype != torch.float32 or y_true.shape != y_pred.shape:
            return None
        if y_pred.dim() < 2 or y_pred.numel() == 0 or not y_true.dtype.is_floating_point or y_true.dtype == torch.float64:
            return None
        if y_true.requires_grad and torch.is_grad_enabled():
            return None
        L = y_pred.shape[-1]
        if cce_obj.label_weights is not None and cce_obj.label_weights.shape[-1] != L:
            return None                                      # the CCE raises its own error on the ordinary path
        if not _lib.lib().nrt_seg_loss_supported(L) or y_pred.shape[0] > 65535:
            return None
        return cls(dice_obj, cce_obj, y_true, y_pred)

    def __enter__(self):
        self._prev = getattr(self._tls, 'current', None)
        self._tls.current = self
        return self

    def __exit__(self, *exc):
        self._tls.current = self._prev
        return False

    @classmethod
    def lookup(cls, owner, y_true, y_pred):
        j = getattr(cls._tls, 'current', None)
        if j is not None and (owner is j.dice_obj or owner is j.cce_obj) and y_true is j.y_true and y_pred is j.y_pred:
            return j
        return None

    def result(self):
        """(cce_sum [1], dice [B, L]); computed at the first request"""
        if self._result is None:
            p = self.y_pred.contiguous()
            t = self.y_true.detach().to(torch.float32).contiguous()
            if p.data_ptr() % 16 or t.data_ptr() % 16:
                p, t = p.clone(), t.clone()
            lw = self.cce_obj.label_weights
            w = None if lw is None else self.cce_obj._weights_on(p.device)
            cfg = (float(self.dice_obj.laplace_smoothing), float(self.cce_obj.label_smoothing), bool(self.dice_obj.check_input_limits))
            src = getattr(self.y_pred, '_nrt_softmax_src', None)
            JointSegLoss.applications += 1
            if src is not None and p is self.y_pred and src.valid_for(self.y_pred):
                JointSegLoss.through_softmax += 1
                self._result = _SegLossFn.apply(t, p.detach(), w, cfg, src, *src.inputs)
            else:
                self._result = _SegLossFn.apply(t, p, w, cfg, None)
        return self._result


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

    def _dice_of_deferred_warp(self, y_true, y_pred, eps):
        """Dice(fixed, SpatialTransformer([moving, trf])) with the warp still pending (neurite_amd/deferred.py): one fused kernel
        gathers and reduces, the warped volume is never written.  Soft Dice is symmetric in its two arguments, so either one may
        be the pending warp.  None = not applicable (the caller takes the ordinary path, which evaluates the warp)."""
        from . import deferred, fused
        a_def = isinstance(y_true, deferred.DeferredWarp) and y_true.pending
        b_def = isinstance(y_pred, deferred.DeferredWarp) and y_pred.pending
        if a_def == b_def or self.normalize:
            return None
        warp, other = (y_pred, y_true) if b_def else (y_true, y_pred)
        if isinstance(other, deferred.DeferredWarp):
            other = other.materialize()
        if other.dtype != torch.float32 or tuple(other.shape) != tuple(warp.shape) or other.device != warp.device:
            return None
        warp.check_sources()            # DeferredWarpError if an input was overwritten since SpatialTransformer returned
        src = warp._sources
        if other.data_ptr() % 16 or src['vol'].data_ptr() % 16 or not other.is_contiguous():
            return None                 # what the fused kernel cannot take goes the ordinary way (which evaluates the warp)
        from . import checked
        # the range asserts of :439-444 on this (already lazy) pipeline travel with the result instead of stopping the host at every
        # call (checked.py); `checked.enabled = False` raises at the call site as everywhere else
        limits = ('deferred' if checked.is_enabled() else True) if self.check_input_limits else False
        try:
            return fused.warp_dice(src['vol'], src['shift'], other, indexing='ij', single_transform=src['single_transform'],
                                   fill_value=src['fill_value'], laplace_smoothing=eps, check_input_limits=limits)
        except (NotImplementedError, _lib.NeuriteAmdError):
            # sizes beyond the fused kernel's 32-bit offsets and the like: the eager pipeline handles them
            return None

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
            joint = JointSegLoss.lookup(self, y_true, y_pred)
            if joint is not None:
                return joint.result()[1]
            if torch.is_grad_enabled() and (y_true.requires_grad or y_pred.requires_grad):
                from .deferred import materialize
                return _SoftDiceFn.apply(materialize(y_true), materialize(y_pred), eps, bool(self.normalize),
                                         bool(self.check_input_limits))
            fused_d = self._dice_of_deferred_warp(y_true, y_pred, eps)
            if fused_d is not None:
                return fused_d
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
            t, p, dt = _as_maps(y_true, y_pred)
            if t.shape != p.shape:
                raise ValueError('y_true and y_pred must have the same shape')
            if self.nb_labels is None:                                                    # :460-461
                self.nb_labels = p.shape[-1]
            if self.nb_labels != p.shape[-1]:
                raise NotImplementedError('hard Dice on prob maps with nb_labels != last dimension')
            B, L = t.shape[0], t.shape[-1]
            V = t.numel() // max(B * L, 1)
            # the range asserts of :439-444 apply to the probabilistic inputs: their extrema come out of the counting pass
            # itself when its vector kernel runs (label counts that are multiples of 4, up to 256), else (and for normalize=True, whose asserts look at the
            # normalised maps; the arg-max is invariant to the positive per-voxel normalisation) out of a soft pass
            fused_limits = (self.check_input_limits and not self.normalize and L % 4 == 0
                            and 4 <= L <= 256 and t.data_ptr() % 16 == 0 and p.data_ptr() % 16 == 0)
            if (self.normalize or self.check_input_limits) and not fused_limits:
                _, _, mm = dice_partial_sums(t, p, self.normalize, 0.)
                if self.check_input_limits:
                    _check_limits(mm)
                elif float(mm[0]) < 0 or float(mm[2]) < 0:
                    raise NotImplementedError('normalize=True with negative inputs on the hard path')
            counts = torch.empty((B, 3, L), dtype=torch.int64, device=dev)
            d = torch.empty((B, L), dtype=torch.float32, device=dev)
            nws = lib.nrt_dice_workspace_bytes(V, L, B)
            ws = _lib.workspace(dev, nws)
            mm = torch.empty((4,), dtype=torch.float32, device=dev) if fused_limits else None
            with torch.cuda.device(dev):
                rc = lib.nrt_dice_hard_prob(_lib.ptr(t), _lib.ptr(p), dt, V, L, B, eps, _lib.ptr(counts), _lib.ptr(d),
                                            _lib.ptr(mm) if fused_limits else None, _lib.ptr(ws), nws, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_dice_hard_prob')
            if fused_limits:
                _check_limits(mm)
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
        # block histograms reduced as rows (no global atomics) while the row buffer stays small; label sets in the thousands keep
        # the atomic form, whose workspace is nothing (the row buffer grows with 2048 * 3 * L per batch entry)
        nws = lib.nrt_dice_workspace_bytes(V, L, B) if L <= 256 else 0
        ws = _lib.workspace(dev, nws) if nws else None
        with torch.cuda.device(dev):
            rc = lib.nrt_dice_hard_label_i32(_lib.ptr(t), _lib.ptr(p), V, L, B, eps, _lib.ptr(counts),
                                             _lib.ptr(d), _lib.ptr(ws) if nws else None, nws, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_dice_hard_label_i32')
        return d

    def mean_dice(self, y_true, y_pred):
        """mean dice across all patches and labels, optionally weighted (neurite/tf/metrics.py:484-510)."""
        dice_metric = self.dice(y_true, y_pred)
        from . import checked
        if isinstance(dice_metric, checked.CheckedTensor):
            dice_metric = dice_metric.checked()     # (the finite test below stops the host anyway: the range assert is looked at here)
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
        self.label_weights = label_weights              # (property: the object keeps its OWN copy)
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

    @property
    def label_weights(self):
        return self._label_weights

    @label_weights.setter
    def label_weights(self, value):
        # a private copy (a tensor made with as_tensor from a NumPy array shares its memory -- edits of the array never showed
        # in `_version` and the device copy went stale); assigning the attribute again is how the weights are changed
        if value is None:
            self._label_weights = None
        elif isinstance(value, torch.Tensor):
            self._label_weights = value.detach().clone()
        else:
            self._label_weights = torch.tensor(np.asarray(value, dtype=np.float32))
        self._w_cache = {}

    def _weights_on(self, dev):
        """label_weights as a float32 tensor on `dev`, copied once per device (again after the attribute was assigned, or after an in-place
        edit of the tensor the attribute returns): a host-resident weight vector used to cost one pageable host-to-device copy per call,
        more than the kernel at config 5's size."""
        lw = self._label_weights
        if lw is None:
            return None
        try:
            version = lw._version
        except RuntimeError:                         # an inference tensor keeps no version counter and cannot be edited in place outside inference mode
            version = -1
        hit = self._w_cache.get(str(dev))
        if hit is None or hit[0] != version:
            hit = (version, lw.detach().to(dev, torch.float32).contiguous())
            self._w_cache[str(dev)] = hit
        return hit[1]

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
        w = self._weights_on(dev)
        N = p.numel() // max(yf, 1)
        need_pv = sample_weight is not None or self.reduction == 'none'
        joint = None if need_pv else JointSegLoss.lookup(self, y_true, y_pred)
        if joint is not None:
            res = joint.result()[0]
        else:
            # the mean's division happens in the kernel (one launch per call), with or without an autograd node: one formulation, one value
            mean = not need_pv and self.reduction != 'sum' and N > 0
            if torch.is_grad_enabled() and (p.requires_grad or t.requires_grad):
                res = _WcceFn.apply(t, p, w, self.from_logits, self.label_smoothing, need_pv, N if mean else None)
            else:
                res = _wcce_launch(t, p, w, self.from_logits, self.label_smoothing, need_pv, divide_by=N if mean else None)
            if mean:
                return res[0]
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


# --------------------------------------------------------------------------------------
# MeanSquaredErrorProb (neurite/tf/metrics.py:653-692)
# --------------------------------------------------------------------------------------

class _MseProbFn(torch.autograd.Function):
    """sum over everything of w_l (t - p)^2 in one pass over the two maps (nrt_sqdiff_sums_f32: the difference is formed in
    registers and its squares go through the Dice sums reduction; expanding to sum t^2 - 2 sum t p + sum p^2 cancels
    catastrophically for maps that nearly agree -- ADVICE r1: 0.6 % error at |t - p| ~ 3e-3 from the float32 sums alone).
    Backward dp = 2 w_l (p - t) g on the per-channel a x + b y kernel."""

    @staticmethod
    def forward(ctx, t, p, w):
        lib = _lib.lib()
        dev = p.device
        L = p.shape[-1]
        tc, pc = t.contiguous(), p.contiguous()
        B = pc.shape[0] if pc.dim() > 1 else 1
        V = pc.numel() // max(B * L, 1)
        if pc.numel() == 0:
            ctx.save_for_backward(t, p, w)
            return torch.zeros((), dtype=torch.float32, device=dev)
        sums = torch.empty((B, 3, L), dtype=torch.float32, device=dev)
        scratch = torch.empty((B, L), dtype=torch.float32, device=dev)
        nws = lib.nrt_dice_workspace_bytes(V, L, B)
        ws = _lib.workspace(dev, nws)
        with torch.cuda.device(dev):
            rc = lib.nrt_sqdiff_sums_f32(_lib.ptr(tc), _lib.ptr(pc), V, L, B, _lib.ptr(sums), _lib.ptr(scratch), _lib.ptr(ws), nws,
                                         _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_sqdiff_sums_f32')
        per_label = sums[:, 1].double().sum(0)                                                             # [L] sum d^2
        ctx.save_for_backward(t, p, w)
        return (per_label * w.double()).sum().to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        t, p, w = ctx.saved_tensors
        lib = _lib.lib()
        dev = p.device
        L = p.shape[-1]
        ca = (2.0 * w * g).to(torch.float32).contiguous()
        cb = (-ca).contiguous()
        zero = torch.zeros(L, dtype=torch.float32, device=dev)
        dp = dt = None
        pc, tc = p.contiguous(), t.contiguous()
        if ctx.needs_input_grad[1]:
            dp = torch.empty_like(pc)
            with torch.cuda.device(dev):
                rc = lib.nrt_channel_axpby_f32(_lib.ptr(pc), _lib.ptr(tc), _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(zero), _lib.ptr(dp),
                                               pc.numel(), L, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_channel_axpby_f32')
        if ctx.needs_input_grad[0]:
            dt = torch.empty_like(tc)
            with torch.cuda.device(dev):
                rc = lib.nrt_channel_axpby_f32(_lib.ptr(tc), _lib.ptr(pc), _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(zero), _lib.ptr(dt),
                                               tc.numel(), L, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_channel_axpby_f32')
        return dt, dp, None


class MeanSquaredErrorProb:
    """
    tf.keras.losses.MeanSquaredError over label (log-)probability maps with label weights along the last axis
    (neurite/tf/metrics.py:653-692): mean over all elements of w_l (y_true - y_pred)^2 (without label weights Keras averages
    over the label axis first -- the same number).  Keras keyword arguments honoured: reduction ('auto' /
    'sum_over_batch_size' / 'sum'), name.  sample_weight: None or a scalar.  float32 [B, ..., L]; differentiable.
    """

    def __init__(self, label_weights=None, **kwargs):
        self.label_weights = None
        if label_weights is not None:
            self.label_weights = torch.as_tensor(np.asarray(label_weights, dtype=np.float32)
                                                 if not isinstance(label_weights, torch.Tensor) else label_weights)
        self.reduction = kwargs.pop('reduction', 'auto')
        self.name = kwargs.pop('name', 'mean_squared_error')
        if kwargs:
            raise TypeError('unexpected keyword arguments: %s' % sorted(kwargs))
        if self.reduction not in ('auto', 'sum_over_batch_size', 'sum', 'none'):
            raise ValueError('Invalid Reduction Key: %s' % self.reduction)
        if self.reduction == 'none':
            raise NotImplementedError("MeanSquaredErrorProb: reduction='none' is not implemented")

    def __call__(self, y_true, y_pred, sample_weight=None):
        return self.mse(y_true, y_pred, sample_weight=sample_weight)

    def mse(self, y_true, y_pred, sample_weight=None):
        yf = y_pred.shape[-1]
        if self.label_weights is not None:
            lf = self.label_weights.shape[0]
            if yf != lf:                                                                  # metrics.py:679-680
                raise ValueError(f'Label weights must be of len {yf}, but got {lf}.')
        dev = _lib.require_device(y_true, y_pred)
        if y_true.shape != y_pred.shape:
            raise ValueError('y_true and y_pred must have the same shape')
        if sample_weight is not None and np.ndim(sample_weight) != 0:
            raise NotImplementedError('MeanSquaredErrorProb: scalar sample_weight only')
        w = torch.ones(yf, dtype=torch.float32, device=dev) if self.label_weights is None \
            else self.label_weights.to(dev, torch.float32).contiguous()
        total = _MseProbFn.apply(_as_f32(y_true, 'y_true'), _as_f32(y_pred, 'y_pred'), w)
        if sample_weight is not None:
            total = total * float(sample_weight)
        if self.reduction == 'sum':
            # Keras sums the per-sample losses: with label weights those are per element, without them per voxel (mean over labels)
            return total if self.label_weights is not None else total / yf
        return total / y_pred.numel()


# --------------------------------------------------------------------------------------
# MutualInformation (neurite/tf/metrics.py:41-336)
# --------------------------------------------------------------------------------------

def _mi_from_joint(joint, sx, sy, eps=1e-7):
    """metrics.py:262-281 on the [items, B, B] joint histogram and the [items, B] marginal sums.  Under autograd: torch ops on
    the tiny tensors, so that autodiff supplies d mi / d joint for the backward kernel; otherwise one kernel (csrc/mi.hip)."""
    needs_graph = torch.is_grad_enabled() and (joint.requires_grad or sx.requires_grad or sy.requires_grad)
    if not needs_graph and joint.is_cuda and joint.shape[-1] <= 64:
        lib = _lib.lib()
        joint, sx, sy = joint.contiguous(), sx.contiguous(), sy.contiguous()
        mi = torch.empty((joint.shape[0],), dtype=torch.float32, device=joint.device)
        with torch.cuda.device(joint.device):
            rc = lib.nrt_mi_from_joint_f32(_lib.ptr(joint), _lib.ptr(sx), _lib.ptr(sy), int(joint.shape[0]), int(joint.shape[-1]),
                                           float(eps), _lib.ptr(mi), _lib.stream_ptr(joint.device))
        _lib.check(rc, 'nrt_mi_from_joint_f32')
        return mi
    pxy = joint / (joint.sum((1, 2), keepdim=True) + eps)
    px = sx / (sx.sum(1, keepdim=True) + eps)
    py = sy / (sy.sum(1, keepdim=True) + eps)
    pxpy = px[:, :, None] * py[:, None, :] + eps
    return (pxy * torch.log(pxy / pxpy + eps)).sum((1, 2))


class _MiJointFn(torch.autograd.Function):
    """joint soft histogram + marginals of two images [B, V, C] (csrc/mi.hip); backward wrt both images."""

    @staticmethod
    def forward(ctx, x, y, cx, cy, alpha, lo, hi):
        lib = _lib.lib()
        dev = x.device
        B, V, C = x.shape
        nb = cx.numel()
        acc = torch.zeros((B * C * (nb * nb + 2 * nb),), dtype=torch.float32, device=dev)        # one zero fill for the three
        joint = acc[:B * C * nb * nb].view(B * C, nb, nb)
        sx = acc[B * C * nb * nb:B * C * (nb * nb + nb)].view(B * C, nb)
        sy = acc[B * C * (nb * nb + nb):].view(B * C, nb)
        with torch.cuda.device(dev):
            rc = lib.nrt_mi_joint_f32(_lib.ptr(x), _lib.ptr(y), _lib.ptr(cx), _lib.ptr(cy), float(alpha), float(lo), float(hi),
                                      B, V, C, nb, _lib.ptr(joint), _lib.ptr(sx), _lib.ptr(sy), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_mi_joint_f32')
        ctx.save_for_backward(x, y, cx, cy)
        ctx.cfg = (float(alpha), float(lo), float(hi))
        return joint, sx, sy

    @staticmethod
    def backward(ctx, gj, gsx, gsy):
        x, y, cx, cy = ctx.saved_tensors
        alpha, lo, hi = ctx.cfg
        lib = _lib.lib()
        dev = x.device
        B, V, C = x.shape
        nb = cx.numel()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None, None, None, None, None, None
        gj = torch.zeros((B * C, nb, nb), dtype=torch.float32, device=dev) if gj is None else gj.contiguous()
        gsx = torch.zeros((B * C, nb), dtype=torch.float32, device=dev) if gsx is None else gsx.contiguous()
        gsy = torch.zeros((B * C, nb), dtype=torch.float32, device=dev) if gsy is None else gsy.contiguous()
        with torch.cuda.device(dev):
            rc = lib.nrt_mi_joint_bwd_f32(_lib.ptr(x), _lib.ptr(y), _lib.ptr(cx), _lib.ptr(cy), alpha, lo, hi, B, V, C, nb,
                                          _lib.ptr(gj), _lib.ptr(gsx), _lib.ptr(gsy), _lib.ptr(gx), _lib.ptr(gy),
                                          _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_mi_joint_bwd_f32')
        return gx, gy, None, None, None, None, None


class _MiMapsFn(torch.autograd.Function):
    """joint[b] = x[b]^T y[b] and the column sums of two maps [bs, V, Bn] (metrics.py:256-269).  Forward: the 1x1x1 case of the
    conv weight-gradient contraction (MFMA over voxels; its bias output is sum_v y).  Backward: dx = y gJ^T + gsx and
    dy = x gJ + gsy, i.e. 1x1 convolutions with the [Bn, Bn] histogram gradient as weights."""

    @staticmethod
    def forward(ctx, xf, yf):
        lib = _lib.lib()
        dev = xf.device
        bs, V, Bn = xf.shape
        joint = torch.zeros((bs, Bn, Bn), dtype=torch.float32, device=dev)
        sx = torch.zeros((bs, Bn), dtype=torch.float32, device=dev)
        sy = torch.zeros((bs, Bn), dtype=torch.float32, device=dev)
        shape = [V // 32, 4, 8] if V % 32 == 0 else [V, 1, 1]
        with torch.cuda.device(dev):
            for b in range(bs):
                rc = lib.nrt_conv3d_wgrad_f32(_lib.ptr(xf[b]), _lib.ptr(yf[b]), _lib.ptr(joint[b]), _lib.ptr(sy[b]), 1,
                                              _lib.ints(shape), Bn, Bn, _lib.ints([1, 1, 1]), 1, _lib.stream_ptr(dev))
                _lib.check(rc, 'nrt_conv3d_wgrad_f32')
            rc = lib.nrt_colsum_f32(_lib.ptr(xf), bs, V, Bn, _lib.ptr(sx), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_colsum_f32')
        ctx.save_for_backward(xf, yf)
        return joint, sx, sy

    @staticmethod
    def backward(ctx, gj, gsx, gsy):
        xf, yf = ctx.saved_tensors
        lib = _lib.lib()
        dev = xf.device
        bs, V, Bn = xf.shape
        if Bn > 64:
            raise NotImplementedError('neurite_amd: backward of MutualInformation.maps with more than 64 bins')
        gj = torch.zeros((bs, Bn, Bn), dtype=torch.float32, device=dev) if gj is None else gj.contiguous()
        gsx = torch.zeros((bs, Bn), dtype=torch.float32, device=dev) if gsx is None else gsx.contiguous()
        gsy = torch.zeros((bs, Bn), dtype=torch.float32, device=dev) if gsy is None else gsy.contiguous()
        gx = torch.empty_like(xf) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(yf) if ctx.needs_input_grad[1] else None
        gjt = gj.transpose(1, 2).contiguous()
        with torch.cuda.device(dev):
            for b in range(bs):
                if gx is not None:          # dx[v, i] = sum_j y[v, j] gJ[i, j] + gsx[i]: weights [cin = j][cout = i] = gJ^T
                    rc = lib.nrt_conv1x1_softmax_f32(_lib.ptr(yf[b]), _lib.ptr(gjt[b]), _lib.ptr(gsx[b]), _lib.ptr(gx[b]), V, Bn, Bn,
                                                     0, 0, _lib.stream_ptr(dev))
                    _lib.check(rc, 'nrt_conv1x1_softmax_f32')
                if gy is not None:          # dy[v, j] = sum_i x[v, i] gJ[i, j] + gsy[j]
                    rc = lib.nrt_conv1x1_softmax_f32(_lib.ptr(xf[b]), _lib.ptr(gj[b]), _lib.ptr(gsy[b]), _lib.ptr(gy[b]), V, Bn, Bn,
                                                     0, 0, _lib.stream_ptr(dev))
                    _lib.check(rc, 'nrt_conv1x1_softmax_f32')
        return gx, gy


class MutualInformation:
    """
    Soft mutual information for intensity volumes and probabilistic volumes (neurite/tf/metrics.py:41-336):
    `volumes`, `segs`, `volume_seg`, `channelwise`, `maps`.  Same constructor arguments and defaults; like the reference
    the constructor prints soft_bin_alpha when it derives it.  The image paths (`volumes`, `channelwise`) never build the
    [V, nb_bins] soft maps: csrc/mi.hip forms the bin weights in registers and contracts them on the matrix cores; they
    are differentiable wrt both images (bin centres held constant).  nb_bins <= 32 on those paths.
    """

    def __init__(self, bin_centers=None, nb_bins=None, soft_bin_alpha=None, min_clip=None, max_clip=None):
        self.bin_centers = None
        if bin_centers is not None:
            self.bin_centers = torch.as_tensor(np.asarray(bin_centers, dtype=np.float32))
            assert nb_bins is None, 'cannot provide both bin_centers and nb_bins'
            nb_bins = self.bin_centers.shape[0]
        self.nb_bins = nb_bins
        if bin_centers is None and nb_bins is None:
            self.nb_bins = 16
        self.min_clip = -np.inf if min_clip is None else min_clip
        self.max_clip = np.inf if max_clip is None else max_clip
        self.soft_bin_alpha = soft_bin_alpha
        if self.soft_bin_alpha is None:
            sigma_ratio = 0.5
            if self.bin_centers is None:
                sigma = np.float32(sigma_ratio / (self.nb_bins - 1))
            else:
                sigma = np.float32(sigma_ratio * float(np.mean(np.diff(self.bin_centers.numpy()))))
            self.soft_bin_alpha = float(np.float32(1) / (np.float32(2) * np.square(sigma)))       # tf.square works in float32
            print(self.soft_bin_alpha)

    # the reference passes both bin_centers and nb_bins to soft_quantize, which asserts (utils.py:1143); here given centres work
    def _centers(self, t):
        return utils._bin_centers(t, self.bin_centers, None if self.bin_centers is not None else self.nb_bins)

    def _soft_sim_map(self, x):
        return utils.soft_quantize(x, bin_centers=self.bin_centers, nb_bins=None if self.bin_centers is not None else self.nb_bins,
                                   alpha=self.soft_bin_alpha, min_clip=self.min_clip, max_clip=self.max_clip, return_log=False)

    def _soft_log_sim_map(self, x):
        return utils.soft_quantize(x, bin_centers=self.bin_centers, nb_bins=None if self.bin_centers is not None else self.nb_bins,
                                   alpha=self.soft_bin_alpha, min_clip=self.min_clip, max_clip=self.max_clip, return_log=True)

    def volumes(self, x, y):
        """MI for each item of a batch of single-channel volumes [bs, ..., 1] -> [bs]."""
        msg = 'volume_mi requires two single-channel volumes. See channelwise().'
        if x.shape[-1] != 1 or y.shape[-1] != 1:
            raise InvalidArgumentError(msg)
        return self.channelwise(x, y).reshape(-1)

    def segs(self, x, y):
        """MI between two probabilistic segmentation maps [bs, ..., nb_labels] -> [bs]."""
        return self.maps(x, y)

    def volume_seg(self, x, y):
        """MI between a volume [bs, ..., 1] and a probabilistic segmentation [bs, ..., nb_labels] (either order) -> [bs]."""
        cx_, cy_ = x.shape[-1], y.shape[-1]
        if min(cx_, cy_) != 1:
            raise InvalidArgumentError('volume_seg_mi requires one single-channel volume.')
        if not max(cx_, cy_) > 1:
            raise InvalidArgumentError('volume_seg_mi requires one multi-channel segmentation.')
        if cx_ == 1:
            x = self._soft_sim_map(x[..., 0])
        else:
            y = self._soft_sim_map(y[..., 0])
        return self.maps(x, y)

    def channelwise(self, x, y):
        """MI for each channel of x and y [bs, ..., C] -> [bs, C] (bins spread between the extrema of each whole tensor)."""
        _lib.require_device(x, y)
        if tuple(x.shape) != tuple(y.shape):
            raise InvalidArgumentError('volume shapes do not match')
        if x.dtype != torch.float32 or y.dtype != torch.float32:
            raise NotImplementedError('MutualInformation: float32 tensors')
        nb = self.nb_bins
        if nb > 32:
            raise NotImplementedError('MutualInformation on images: nb_bins <= 32 on the HIP path, got %d' % nb)
        bs, C = x.shape[0], x.shape[-1]
        xf = x.reshape(bs, -1, C).contiguous()
        yf = y.reshape(bs, -1, C).contiguous()
        cx, cy = self._centers(xf.detach()), self._centers(yf.detach())
        joint, sx, sy = _MiJointFn.apply(xf, yf, cx, cy, self.soft_bin_alpha, self.min_clip, self.max_clip)
        return _mi_from_joint(joint, sx, sy).reshape(bs, C)

    def maps(self, x, y):
        """MI per batch entry of two probability / similarity maps [bs, ..., B] -> [bs]; differentiable wrt both maps."""
        _lib.require_device(x, y)
        if tuple(x.shape) != tuple(y.shape):
            raise InvalidArgumentError('')                                              # tf.debugging.assert_equal :249
        if x.dtype != torch.float32 or y.dtype != torch.float32:
            raise NotImplementedError('MutualInformation: float32 tensors')
        bs, Bn = x.shape[0], x.shape[-1]
        xf = x.reshape(bs, -1, Bn).contiguous()
        yf = y.reshape(bs, -1, Bn).contiguous()
        mm = torch.stack([utils._device_minmax(xf.detach()), utils._device_minmax(yf.detach())])
        if not bool((mm[:, 0] >= 0).all()):                                             # assert_non_negative :250-251
            raise InvalidArgumentError('')
        joint, sx, sy = _MiMapsFn.apply(xf, yf)
        return _mi_from_joint(joint, sx, sy)
