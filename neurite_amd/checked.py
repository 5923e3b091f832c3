"""
Range asserts without a host round trip per call.

The reference's `Dice` asserts that both maps lie in [0, 1] (`check_input_limits=True`, the default: neurite/tf/metrics.py:439-444 --
`tf.debugging.assert_*` ops that TensorFlow runs as part of the graph and that abort the step when they fail).  An eager host that
wants to RAISE at the call site has to read the extrema back: one device -> host synchronisation per call, 14 % of the fused
warp + Dice step (14.4 against 16.8 Gvoxel/s).  On the deferred-warp pipeline -- which is lazy already, see deferred.py --
the assert is deferred as well:

  * the fused kernel writes the four extrema next to the Dice values (it always did);
  * `Dice.dice` returns the values as a `CheckedTensor`: device operations pass straight through to the real tensor, anything that
    brings the values to the HOST (`item`, `tolist`, `numpy`, `cpu`, `float()`, `bool()`, printing) first waits for the extrema and
    raises `InvalidArgumentError('value outside range')` if the assert fails -- the values of a failed assert never reach the host;
  * the extrema also travel to pinned host memory on the launch stream (16 bytes, asynchronous); every later deferred-assert call, and
    `flush()`, looks at the copies that have ARRIVED and raises for the earliest failed one -- a result that is only ever consumed by
    device code (a mean that is all-reduced, a loss that is back-propagated) still fails loudly, one or two steps late, like an
    asynchronous HIP error.

`neurite_amd.checked.enabled = False` (env NRT_DEFER_LIMIT_CHECKS=0) restores the eager raise at the call site everywhere;
`with neurite_amd.checked.scope(False):` does so for one block of one thread.
Every other path (materialised tensors, hard Dice, the joint loss, training graphs) raises eagerly as before.
"""

import collections
import contextlib
import os
import threading

import torch
from torch.utils._pytree import tree_map

from .errors import InvalidArgumentError

enabled = os.environ.get('NRT_DEFER_LIMIT_CHECKS', '1') != '0'      # process-wide default; `scope()` overrides it per thread and block

_local = threading.local()


def is_enabled():
    """what Dice consults: the innermost `scope()` of this thread, else the process-wide `enabled`"""
    stack = getattr(_local, 'stack', None)
    return stack[-1] if stack else enabled


@contextlib.contextmanager
def scope(on):
    """`with neurite_amd.checked.scope(False): ...` -- range asserts raise at the call site inside the block (this thread only); nests"""
    stack = getattr(_local, 'stack', None)
    if stack is None:
        stack = _local.stack = []
    stack.append(bool(on))
    try:
        yield
    finally:
        stack.pop()


_pending = collections.deque()
_pinned = []                      # recycled 4-float pinned host buffers


class PendingCheck:
    """the extrema [min t, max t, min p, max p] of one call on their way to the host"""

    __slots__ = ('host', 'event', 'what', 'done', 'failed')

    def __init__(self, minmax, what):
        self.host = _pinned.pop() if _pinned else torch.empty(4, dtype=torch.float32).pin_memory()
        self.host.copy_(minmax, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.what = what
        self.done = False
        self.failed = False

    def arrived(self):
        return self.done or self.event.query()

    def resolve(self, block=True):
        """look at the extrema (waiting for them if `block`); raises InvalidArgumentError once, at the first look that finds a failure"""
        if self.done:
            if self.failed:
                raise InvalidArgumentError('value outside range')
            return True
        if not block and not self.event.query():
            return False
        self.event.synchronize()
        mn_t, mx_t, mn_p, mx_p = self.host.tolist()
        _pinned.append(self.host)
        self.host = None
        self.done = True
        try:
            _pending.remove(self)
        except ValueError:
            pass
        if not (mn_t >= 0. and mn_p >= 0. and mx_t <= 1. and mx_p <= 1.):           # (also catches NaN)
            self.failed = True
            raise InvalidArgumentError('value outside range')
        return True


def register(minmax, what=''):
    c = PendingCheck(minmax, what)
    _pending.append(c)
    return c


def poll():
    """raise for the earliest registered assert whose extrema have arrived and failed; never waits"""
    for c in list(_pending):
        if not c.arrived():
            break                                      # (copies complete in issue order per stream: what follows is younger)
        c.resolve(block=False)


def flush():
    """wait for every outstanding assert; raises InvalidArgumentError if any failed (all are cleared either way)"""
    err = None
    while _pending:
        try:
            _pending[0].resolve(block=True)
        except InvalidArgumentError as e:
            err = err or e
    if err is not None:
        raise err


def pending_count():
    return len(_pending)


_HOST_OPS = None
_SAME_VALUES = None


def _host_ops():
    global _HOST_OPS
    if _HOST_OPS is None:
        aten = torch.ops.aten
        _HOST_OPS = {aten._local_scalar_dense.default, aten.equal.default, aten.is_nonzero.default}
    return _HOST_OPS


def _same_values():
    """operations whose result is (a piece of) the same values: it keeps the pending assert (`d.detach().cpu()`, `d[0].item()`)"""
    global _SAME_VALUES
    if _SAME_VALUES is None:
        aten = torch.ops.aten
        names = ('detach.default', 'alias.default', 'clone.default', 'view.default', '_unsafe_view.default', 'reshape.default',
                 'select.int', 'slice.Tensor', 'squeeze.default', 'squeeze.dim', 'unsqueeze.default', 't.default', 'transpose.int',
                 'permute.default', 'expand.default', 'contiguous.default', 'flatten.using_ints', 'index.Tensor', 'unbind.int',
                 'split.Tensor', 'narrow.default', 'lift_fresh.default', 'detach_.default')
        ops = set()
        for n in names:
            pkt, ov = n.split('.')
            try:
                ops.add(getattr(getattr(aten, pkt), ov))
            except AttributeError:
                pass
        _SAME_VALUES = ops
    return _SAME_VALUES


class CheckedTensor(torch.Tensor):
    """values whose range assert is still on its way: device operations see the real tensor, the host sees them only after the assert"""

    @staticmethod
    def __new__(cls, value, check):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(value.shape), dtype=value.dtype, device=value.device, requires_grad=False)
        t._value = value
        t._check = check
        return t

    def checked(self):
        """the real tensor, after the assert (waits for the extrema; raises InvalidArgumentError)"""
        self._check.resolve(block=True)
        return self._value

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        to_host = func in _host_ops()
        if not to_host and func is torch.ops.aten._to_copy.default:
            dev = kwargs.get('device')
            to_host = dev is not None and torch.device(dev).type == 'cpu'

        checks = []

        def un(a):
            if isinstance(a, CheckedTensor):
                checks.append(a._check)
                return a.checked() if to_host else a._value
            return a
        out = func(*tree_map(un, args), **tree_map(un, kwargs))
        if not to_host and func in _same_values() and len(checks) == 1 and (not checks[0].done or checks[0].failed):
            def re(o):
                return CheckedTensor(o, checks[0]) if type(o) is torch.Tensor else o
            out = tree_map(re, out)
        return out

    # ---- the tensor methods that do not go through the dispatcher ------------------------------------------------------------------
    def data_ptr(self):
        return self._value.data_ptr()

    def numpy(self, *a, **k):
        return self.checked().numpy(*a, **k)

    def tolist(self):
        return self.checked().tolist()

    def item(self):
        return self.checked().item()

    def __array__(self, *a, **k):
        return self.checked().__array__(*a, **k)

    def __dlpack__(self, *a, **k):
        return self.checked().__dlpack__(*a, **k)

    def __reduce_ex__(self, proto):
        return self.checked().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        import copy
        return copy.deepcopy(self.checked(), memo)

    def __copy__(self):
        import copy
        return copy.copy(self.checked())

    def untyped_storage(self):
        return self._value.untyped_storage()

    def __repr__(self):
        return repr(self.checked())


def wrap(value, minmax, what=''):
    """`value` with the range assert on `minmax` [4] (device) pending; earlier asserts that have arrived are looked at first"""
    poll()
    return CheckedTensor(value, register(minmax, what))


def unwrap(t):
    """the real tensor behind `t` WITHOUT looking at the assert (identity for ordinary tensors): for device-side consumers"""
    return t._value if isinstance(t, CheckedTensor) else t
