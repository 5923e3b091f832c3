"""
Deferred warps: how the reference-signature pipeline reaches the fused kernel.

    warped = SpatialTransformer()([moving, trf]);   d = Dice().dice(fixed, warped)

is the metric pipeline of neurite/tf/models.py:806-807 + neurite/tf/metrics.py:415-482.  Run eagerly it writes `warped`
(128 B per voxel at 32 labels) only for the Dice kernel to read it back; the fused kernel (csrc/fused.hip) never writes it.
TensorFlow's graph mode lets a compiler see both ops; an eager host has to defer: under the conditions below
`SpatialTransformer` returns a `DeferredWarp` -- a tensor whose metadata (shape, dtype, device) is real and whose values are
computed on first use by ANY torch operation or by any neurite_amd kernel.  `metrics.Dice.dice` recognises an unmaterialised
DeferredWarp as one of its arguments and launches the fused warp + Dice kernel on (moving, trf, other map) instead.  Every other
consumer simply triggers the stand-alone interpn kernel, exactly as an eager call would have.

Deferral happens only for: linear interpolation, float32 3-D volumes with 4 * 2^k labels, dense displacement fields, no
gradient being recorded (training graphs stay eager: autograd needs the real tensor), not under `torch.inference_mode()` (inference
tensors have no version counter, see Immutability), and `deferred.enabled` (env NRT_DEFER_WARP, default on).  The values are bit-identical to the eager path whenever they are materialised.

Immutability.  TensorFlow tensors are immutable, so in the reference the value of `warped` is fixed when SpatialTransformer
returns.  A deferred warp reads its inputs when it is first used, and PyTorch lets a program overwrite them in between
(`buf.copy_(next_batch)`, `moving.mul_(mask)`).  The DeferredWarp therefore records the version counters (and addresses) of the
volume and the transform it aliases and checks them when it is evaluated -- by `materialize()` or by the fused Dice kernel: if
either was modified in place, the result the eager path would have produced no longer exists and a `DeferredWarpError` is raised
(loudly, at the use site, naming the remedy: keep the inputs unchanged until the result is used, pass clones, or set
`neurite_amd.deferred.enabled = False`, or for one block of one thread `with neurite_amd.deferred.scope(False):`).  Inputs that had to be converted (dtype, layout) are private copies and cannot change.
"""

import contextlib
import os
import threading

import torch
from torch.utils._pytree import tree_map

enabled = os.environ.get('NRT_DEFER_WARP', '1') != '0'      # process-wide default; `scope()` overrides it for one thread and one block

_local = threading.local()


def is_enabled():
    """what SpatialTransformer consults: the innermost `scope()` of this thread, else the process-wide `enabled`"""
    stack = getattr(_local, 'stack', None)
    return stack[-1] if stack else enabled


@contextlib.contextmanager
def scope(on):
    """`with neurite_amd.deferred.scope(False): ...` -- eager warps inside the block, in this thread only, whatever the process-wide
    switch says (and the other way round); nests.  A library that calls into neurite_amd from several threads, or only wants one
    evaluation eager, does not have to touch the module attribute."""
    stack = getattr(_local, 'stack', None)
    if stack is None:
        stack = _local.stack = []
    stack.append(bool(on))
    try:
        yield
    finally:
        stack.pop()


class DeferredWarpError(RuntimeError):
    """an input of a deferred SpatialTransformer call was modified in place before the result was used"""


def _stamp(t):
    # (inference tensors have no version counter -- reading it raises; SpatialTransformer does not defer them, this is the backstop)
    return (None if t.is_inference() else t._version, t.data_ptr(), tuple(t.shape), tuple(t.stride()))


class DeferredWarp(torch.Tensor):
    @staticmethod
    def __new__(cls, shape, dtype, device, thunk, sources):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device, requires_grad=False)
        t._thunk = thunk
        t._value = None
        t._sources = sources            # dict(vol, shift, single_transform, fill_value): what the fused kernel needs
        t._stamps = tuple((k, sources[k], _stamp(sources[k])) for k in ('vol', 'shift') if sources and sources.get(k) is not None)
        return t

    # ---- evaluation ------------------------------------------------------------------------------------------------
    @property
    def pending(self):
        return self._value is None

    def check_sources(self):
        """raise DeferredWarpError if the volume or the transform changed since SpatialTransformer was called"""
        for name, t, stamp in self._stamps or ():
            if _stamp(t) != stamp:
                raise DeferredWarpError(
                    'the %s passed to SpatialTransformer was modified in place before the (deferred) warped volume was used: the '
                    'warp of the ORIGINAL data can no longer be computed.  Keep the inputs unchanged until the result has been used, '
                    'pass clones, or set neurite_amd.deferred.enabled = False (env NRT_DEFER_WARP=0) for eager evaluation'
                    % ('volume' if name == 'vol' else 'transform'))

    def materialize(self):
        if self._value is None:
            self.check_sources()
            self._value = self._thunk()
            self._thunk = None
            self._sources = None
            self._stamps = None
        return self._value

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def un(a):
            return a.materialize() if isinstance(a, DeferredWarp) else a
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))

    # ---- the tensor methods that do not go through the dispatcher -----------------------------------------------------
    def data_ptr(self):
        return self.materialize().data_ptr()

    def numpy(self, *a, **k):
        return self.materialize().numpy(*a, **k)

    def tolist(self):
        return self.materialize().tolist()

    def item(self):
        return self.materialize().item()

    def __array__(self, *a, **k):
        return self.materialize().__array__(*a, **k)

    def __dlpack__(self, *a, **k):
        return self.materialize().__dlpack__(*a, **k)

    def __reduce_ex__(self, proto):
        return self.materialize().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        # (torch.Tensor.__deepcopy__ rebuilds the subclass through new_empty(): a copy of a warp is simply a copy of its values)
        import copy
        return copy.deepcopy(self.materialize(), memo)

    def __copy__(self):
        import copy
        return copy.copy(self.materialize())

    def untyped_storage(self):
        return self.materialize().untyped_storage()

    def __repr__(self):
        return repr(self.materialize()) if self._value is not None else \
            'DeferredWarp(shape=%s, dtype=%s, device=%s, pending)' % (tuple(self.shape), self.dtype, self.device)


def materialize(t):
    """the real tensor behind `t` (identity for ordinary tensors)"""
    return t.materialize() if isinstance(t, DeferredWarp) else t
