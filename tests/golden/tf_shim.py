"""
A small NumPy stand-in for the TensorFlow / Keras-backend *leaf primitives* that the reference's
hot path calls, so that the reference's OWN Python source under /root/reference can be executed
in a container that has no TensorFlow (TEST INFRASTRUCTURE; used only by make_golden.py).

What this pins: everything the reference itself decides -- control flow, op order, index
arithmetic, corner ordering, weight pairing, clamping order, reshape/flatten conventions.
What this does NOT pin: the leaf semantics of the TF ops themselves, which are restated here from
their documented behaviour (tf.round = half-to-even, float->int32 cast truncates,
tf.clip_by_value = min(max(x, lo), hi) in the tensor dtype, tf.linspace endpoints exact,
tf.math.divide_no_nan, tf.argmax ties -> lowest index, tf.one_hot out-of-range -> zero row,
Keras CCE: normalise, clip to [1e-7, 1-1e-7], -sum t log p, mean over elements).

Typing rule enforced on purpose: a binary op between two tensors of different dtypes raises
(as TensorFlow does); Python / NumPy scalars adopt the tensor's dtype.  Every op is computed in
the tensor dtype with one rounding, like an eager TF CPU kernel.
"""

import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np


# ----------------------------------------------------------------------------- dtypes / shapes

class DType:
    def __init__(self, np_dtype):
        self.np = np.dtype(np_dtype)

    @property
    def is_floating(self):
        return np.issubdtype(self.np, np.floating)

    @property
    def is_integer(self):
        return np.issubdtype(self.np, np.integer)

    @property
    def base_dtype(self):
        return self

    @property
    def name(self):
        return self.np.name

    @property
    def as_numpy_dtype(self):
        return self.np.type

    def __eq__(self, other):
        try:
            return self.np == as_np_dtype(other)
        except TypeError:
            return False

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.np)

    def __repr__(self):
        return 'tf.' + self.np.name


def as_np_dtype(d):
    if isinstance(d, DType):
        return d.np
    return np.dtype(d)


class Dimension(int):
    pass


class TensorShape(tuple):
    def as_list(self):
        return list(self)

    @property
    def rank(self):
        return len(self)

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r


class Tensor:
    __array_priority__ = 1000

    def __init__(self, a, dtype=None):
        if isinstance(a, Tensor):
            a = a.a
        self.a = np.asarray(a, dtype=None if dtype is None else as_np_dtype(dtype))

    # -- TF-ish surface
    @property
    def shape(self):
        return TensorShape(self.a.shape)

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return DType(self.a.dtype)

    def numpy(self):
        return self.a

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __len__(self):
        return self.a.shape[0]

    def __iter__(self):
        for i in range(self.a.shape[0]):
            yield Tensor(self.a[i])

    def __getitem__(self, idx):
        if isinstance(idx, list):
            idx = tuple(idx)
        if isinstance(idx, Tensor):
            idx = idx.a
        return Tensor(self.a[idx])

    def __repr__(self):
        return 'ShimTensor(%r)' % (self.a,)

    __hash__ = object.__hash__

    # -- arithmetic in the tensor dtype
    def _co(self, other):
        if isinstance(other, Tensor):
            if other.a.dtype != self.a.dtype:
                raise TypeError('dtype mismatch in binary op: %s vs %s' % (self.a.dtype, other.a.dtype))
            return other.a
        return np.asarray(other).astype(self.a.dtype)

    def _bin(self, other, fn, swap=False):
        o = self._co(other)
        with np.errstate(all='ignore'):
            r = fn(o, self.a) if swap else fn(self.a, o)
        return Tensor(np.asarray(r))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __neg__(self): return Tensor(-self.a)
    def __pow__(self, o): return self._bin(o, np.power)
    def __rpow__(self, o): return self._bin(o, np.power, True)
    def __lt__(self, o): return self._bin(o, np.less)
    def __le__(self, o): return self._bin(o, np.less_equal)
    def __gt__(self, o): return self._bin(o, np.greater)
    def __ge__(self, o): return self._bin(o, np.greater_equal)
    def __eq__(self, o): return bool(np.all(self.a == A(o)))          # eager tensors compare by value (used in `if t == 1`)
    def __ne__(self, o): return not self.__eq__(o)
    __hash__ = object.__hash__


def T(x, dtype=None):
    return x if (isinstance(x, Tensor) and dtype is None) else Tensor(x, dtype)


def A(x):
    return x.a if isinstance(x, Tensor) else np.asarray(x)


# ----------------------------------------------------------------------------- tf.* primitives

def convert_to_tensor(x, dtype=None, **kw):
    if isinstance(x, Tensor):
        return x if dtype is None else cast(x, dtype)
    a = np.asarray(x)
    if dtype is None:
        if a.dtype == np.float64 and not isinstance(x, np.ndarray):
            a = a.astype(np.float32)       # python floats -> float32, like TF
        if a.dtype == np.int64 and not isinstance(x, np.ndarray):
            a = a.astype(np.int32)         # python ints -> int32
    else:
        a = a.astype(as_np_dtype(dtype))
    return Tensor(a)


def constant(v, dtype=None, **kw):
    return convert_to_tensor(v, dtype)


def cast(x, dtype):
    a = A(x)
    nd = as_np_dtype(dtype)
    with np.errstate(all='ignore'):
        if np.issubdtype(nd, np.integer) and np.issubdtype(a.dtype, np.floating):
            return Tensor(np.trunc(a).astype(nd))          # float -> int truncates toward zero
        return Tensor(a.astype(nd))


def stack(values, axis=0, **kw):
    if isinstance(values, Tensor):
        return values
    arrs = [A(convert_to_tensor(v)) for v in values]
    if len({a.dtype for a in arrs}) != 1:
        raise TypeError('stack of mixed dtypes')
    return Tensor(np.stack(arrs, axis))


def concat(values, axis, **kw):
    if isinstance(values, Tensor):
        return values                        # tf.concat of a single tensor is the tensor
    arrs = [A(convert_to_tensor(v)) for v in values]
    return Tensor(np.concatenate(arrs, axis))


def floor(x): return Tensor(np.floor(A(x)))
def round_(x): return Tensor(np.rint(A(x)))                 # half-to-even
def exp(x): return Tensor(np.exp(A(x)))
def log(x): return Tensor(np.log(A(x)))
def square(x): return Tensor(np.square(np.float32(x) if isinstance(x, (float, int)) else A(x)))     # python scalars become float32
def less(x, y): return T(x) < y
def greater(x, y): return T(x) > y
def logical_not(x): return Tensor(np.logical_not(A(x)))


def clip_by_value(t, lo, hi, **kw):
    a = A(t)
    lo = np.asarray(A(lo)).astype(a.dtype)
    hi = np.asarray(A(hi)).astype(a.dtype)
    return Tensor(np.minimum(np.maximum(a, lo), hi))


def _shape_arg(shape):
    if isinstance(shape, Tensor):
        shape = shape.a
    return tuple(int(A(s)) for s in shape)


def reshape(x, shape, **kw):
    return Tensor(A(x).reshape(_shape_arg(shape)))


def gather(params, indices, axis=0, **kw):
    if isinstance(params, (list, tuple, range)):
        params = np.asarray(list(params))
    return Tensor(np.take(A(params), A(indices), axis=int(A(axis))))


def reduce_any(x, axis=None, keepdims=False):
    return Tensor(np.any(A(x), axis=axis, keepdims=keepdims))


def reduce_sum(x, axis=None, keepdims=False):
    a = A(x)
    axis = tuple(axis) if isinstance(axis, list) else axis
    return Tensor(np.sum(a.astype(np.float64), axis=axis, keepdims=keepdims).astype(a.dtype))


def linspace(start, stop, num, **kw):
    """tf.linspace in float32 for python-float endpoints (TF semantics restated)."""
    start, stop, num = (A(v) if isinstance(v, Tensor) else v for v in (start, stop, num))
    ints = all(isinstance(v, (int, np.integer)) or (isinstance(v, np.ndarray) and np.issubdtype(v.dtype, np.integer))
               for v in (start, stop))
    # integer end points: TF's linspace divides with truediv (float64) and re-casts start / stop to that dtype
    f = np.float64 if ints else np.float32
    start, stop, num = f(start), f(stop), int(num)
    if num == 1:
        return Tensor(np.array([start], f))
    delta = f((stop - start) / f(num - 1))
    i = np.arange(1, num - 1).astype(f)
    return Tensor(np.concatenate([[start], (start + delta * i).astype(f), [stop]]).astype(f))


def range_(start, limit=None, delta=1, dtype=None, **kw):
    if limit is None:
        start, limit = 0, start
    start, limit, delta = (A(v) if isinstance(v, Tensor) else v for v in (start, limit, delta))
    if any(isinstance(v, (float, np.floating)) or (isinstance(v, np.ndarray) and np.issubdtype(v.dtype, np.floating))
           for v in (start, limit, delta)):
        # float range: size = ceil((limit - start) / delta), element i = start + i * delta, all in float32
        f = np.float32
        start, limit, delta = f(start), f(limit), f(delta)
        n = int(np.ceil(np.abs((limit - start) / delta)))
        a = (start + np.arange(n).astype(f) * delta).astype(f)
        return Tensor(a if dtype is None else a.astype(as_np_dtype(dtype)))
    a = np.arange(start, limit, delta)
    if dtype is not None:
        a = a.astype(as_np_dtype(dtype))
    elif np.issubdtype(a.dtype, np.integer):
        a = a.astype(np.int32)
    return Tensor(a)


def size(x, **kw):
    return Tensor(np.int32(A(x).size))


def tile(x, multiples, **kw):
    return Tensor(np.tile(A(x), _shape_arg(multiples)))


def ones(shape, dtype=np.float32, **kw):
    return Tensor(np.ones(_shape_arg(shape), as_np_dtype(dtype)))


def zeros(shape, dtype=np.float32, **kw):
    return Tensor(np.zeros(_shape_arg(shape), as_np_dtype(dtype)))


def map_fn(fn, elems, **kw):
    if isinstance(elems, (list, tuple)):
        n = A(elems[0]).shape[0]
        outs = [fn([T(A(e)[i]) for e in elems]) for i in range(n)]
    else:
        outs = [fn(T(A(elems)[i])) for i in range(A(elems).shape[0])]
    return Tensor(np.stack([A(o) for o in outs], 0))


def divide_no_nan(x, y, **kw):
    x, y = A(x), A(y)
    if x.dtype != y.dtype:
        raise TypeError('divide_no_nan dtype mismatch')
    out = np.zeros(np.broadcast(x, y).shape, x.dtype)
    np.divide(x, y, out=out, where=(y != 0))
    return Tensor(out)


class InvalidArgumentError(Exception):
    pass


def _assert(cond, msg):
    if not bool(np.all(cond)):
        raise InvalidArgumentError(msg)


def assert_greater_equal(x, y, message='', **kw): _assert(A(x) >= A(y), message)
def assert_less_equal(x, y, message='', **kw): _assert(A(x) <= A(y), message)
def assert_all_finite(x, message='', **kw): _assert(np.isfinite(A(x)), message)


# ----------------------------------------------------------------------------- keras.backend

def k_expand_dims(x, axis=-1): return Tensor(np.expand_dims(A(x), axis))
def k_shape(x): return Tensor(np.array(A(x).shape, np.int32))
def k_int_shape(x): return tuple(A(x).shape)
def k_ndim(x): return A(x).ndim
def k_square(x): return Tensor(np.square(A(x)))
def k_epsilon(): return 1e-7


def k_sum(x, axis=None, keepdims=False):
    # TF's reduction order is unspecified; accumulate wide and round once
    a = A(x)
    axis = tuple(axis) if isinstance(axis, list) else axis
    return Tensor(np.sum(a.astype(np.float64), axis=axis, keepdims=keepdims).astype(a.dtype))


def k_mean(x, axis=None, keepdims=False):
    a = A(x)
    return Tensor(np.mean(a.astype(np.float64), axis=axis, keepdims=keepdims).astype(a.dtype))


def k_argmax(x, axis=-1): return Tensor(np.argmax(A(x), axis=axis).astype(np.int64))


def k_one_hot(indices, num_classes):
    idx = A(indices).astype(np.int64)
    out = np.zeros(idx.shape + (int(num_classes),), np.float32)
    ok = (idx >= 0) & (idx < num_classes)
    np.put_along_axis(out, np.where(ok, idx, 0)[..., None], ok[..., None].astype(np.float32), -1)
    return Tensor(out)


def k_concatenate(tensors, axis=-1): return Tensor(np.concatenate([A(t) for t in tensors], axis))
def k_permute_dimensions(x, pattern): return Tensor(np.transpose(A(x), tuple(pattern)))


def k_batch_dot(x, y, axes=None):
    x, y = A(x), A(y)
    assert axes is None and x.ndim == 3 and y.ndim == 3
    return Tensor(np.einsum('obf,ofc->obc', x.astype(np.float64), y.astype(np.float64)).astype(x.dtype))


def k_bias_add(x, bias, data_format=None):
    b = A(bias)
    assert data_format in (None, 'channels_last', 'channels_first')
    if data_format == 'channels_first' and b.ndim > 1:
        # keras.backend.bias_add, N-D bias, channels_first data: x + reshape(bias, (1, bias_shape[-1]) + bias_shape[:-1])
        # -- a RESHAPE of the (spatial..., C) array, not a transpose (TF semantics, restated)
        b = b.reshape((b.shape[-1],) + b.shape[:-1])
    return T(x) + T(b[None])


# ----------------------------------------------------------------------------- keras classes

class Layer:
    """Just enough of keras.layers.Layer: build on first call with the input shape, then call."""

    def __init__(self, name=None, **kwargs):
        self.name = name
        self.built = False
        self._weights = {}

    def build(self, input_shape):
        self.built = True

    def add_weight(self, shape=None, initializer=None, name=None, **kw):
        w = Tensor(np.zeros(tuple(shape), np.float32))
        self._weights[name] = w
        return w

    def __call__(self, inputs, **kw):
        if not self.built:
            if isinstance(inputs, (list, tuple)):
                shp = [tuple(A(i).shape) for i in inputs]
            else:
                shp = tuple(A(inputs).shape)
            self.build(shp)
            self.built = True
        return self.call(inputs, **kw)

    def get_config(self):
        return {'name': self.name}


class KerasCCE:
    """tf.keras.losses.CategoricalCrossentropy (TF semantics restated; float64 internally)."""

    def __init__(self, from_logits=False, label_smoothing=0., **kw):
        self.from_logits = from_logits
        self.label_smoothing = label_smoothing

    def __call__(self, y_true, y_pred, sample_weight=None):
        p = A(y_pred).astype(np.float64)
        t = A(y_true).astype(np.float64)
        C = p.shape[-1]
        if self.label_smoothing:
            t = t * (1.0 - self.label_smoothing) + self.label_smoothing / C
        if self.from_logits:
            z = p - p.max(-1, keepdims=True)
            logq = z - np.log(np.exp(z).sum(-1, keepdims=True))
        else:
            q = p / p.sum(-1, keepdims=True)
            logq = np.log(np.clip(q, 1e-7, 1 - 1e-7))
        l = -(t * logq).sum(-1)
        if sample_weight is not None:
            sw = A(sample_weight).astype(np.float64)
            while sw.ndim < l.ndim:
                sw = sw[..., None]
            l = l * sw
        return Tensor(np.float32(l.sum() / l.size))


# ----------------------------------------------------------------------------- module plumbing

class _Stub:
    """Permissive placeholder for everything the hot path never touches."""

    def __init__(self, name='stub'):
        self.__name__ = name

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]                      # behaves as an identity decorator
        return _Stub(self.__name__ + '()')

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Stub(self.__name__ + '.' + n)

    def __mro_entries__(self, bases):
        return (object,)


class _AutoModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name[:1].isupper():
            v = type(name, (object,), {'__init__': lambda self, *a, **k: None})
        else:
            v = _Stub(self.__name__ + '.' + name)
        setattr(self, name, v)
        return v


_ROOTS = ('tensorflow', 'pystrum', 'nibabel', 'h5py')


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in _ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _AutoModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


def shape_(x, **kw): return Tensor(np.array(A(x).shape, np.int32))
def transpose_(x, perm=None, **kw): return Tensor(np.transpose(A(x), None if perm is None else tuple(int(p) for p in A(perm))))


def reduce_prod(x, axis=None, keepdims=False):
    return Tensor(np.prod(A(x), axis=axis, keepdims=keepdims))


def reduce_min(x, axis=None, keepdims=False):
    return Tensor(np.min(A(x), axis=None if axis is None else tuple(np.ravel(axis)), keepdims=keepdims))


def reduce_max(x, axis=None, keepdims=False):
    return Tensor(np.max(A(x), axis=None if axis is None else tuple(np.ravel(axis)), keepdims=keepdims))


def nn_convolution(x, filters, padding='VALID', strides=None, dilations=None, **kw):
    """tf.nn.convolution for [N, *S, 1] inputs and [*k, 1, 1] filters (what separable_conv issues): cross-correlation,
    TF SAME padding (total = max((out-1)*stride + (k-1)*dil + 1 - n, 0), before = total // 2), float32 accumulation."""
    x = A(x)
    f = A(filters)
    nd = x.ndim - 2
    assert x.shape[-1] == 1 and f.shape[-2:] == (1, 1), 'shim supports single-channel convolution only'
    strides = [1] * nd if strides is None else [int(s) for s in np.ravel(strides)]
    dil = [1] * nd if dilations is None else [int(d) for d in np.ravel(dilations)]
    k = f.shape[:nd]
    xs = x[..., 0]
    outs, pads = [], []
    for d in range(nd):
        n, ke = xs.shape[1 + d], (k[d] - 1) * dil[d] + 1
        if padding.upper() == 'SAME':
            o = -(-n // strides[d])
            tot = max((o - 1) * strides[d] + ke - n, 0)
            pads.append((tot // 2, tot - tot // 2))
        else:
            o = (n - ke) // strides[d] + 1
            pads.append((0, 0))
        outs.append(o)
    xp = np.pad(xs, [(0, 0)] + pads)
    out = np.zeros((xs.shape[0],) + tuple(outs), x.dtype)
    for tap in np.ndindex(*k):
        sl = [slice(None)]
        for d in range(nd):
            st = tap[d] * dil[d]
            sl.append(slice(st, st + (outs[d] - 1) * strides[d] + 1, strides[d]))
        out = (out + f[tap + (0, 0)] * xp[tuple(sl)]).astype(x.dtype)
    return Tensor(out[..., None])



# ---- keras conv_utils (tensorflow.python.keras.utils.conv_utils), restated: shape bookkeeping of LocallyConnected3D ----
def cu_normalize_tuple(value, n, name):
    if isinstance(value, int):
        return (value,) * n
    value_tuple = tuple(value)
    if len(value_tuple) != n:
        raise ValueError('The `' + name + '` argument must be a tuple of ' + str(n) + ' integers. Received: ' + str(value))
    for v in value_tuple:
        int(v)
    return value_tuple


def cu_normalize_padding(value):
    if isinstance(value, (list, tuple)):
        return value
    padding = value.lower()
    if padding not in {'valid', 'same', 'causal'}:
        raise ValueError('The `padding` argument must be a list/tuple or one of "valid", "same" (or "causal", only for '
                         '`Conv1D). Received: ' + str(padding))
    return padding


def cu_normalize_data_format(value):
    if value is None:
        value = 'channels_last'
    data_format = value.lower()
    if data_format not in {'channels_first', 'channels_last'}:
        raise ValueError('The `data_format` argument must be one of "channels_first", "channels_last". Received: ' + str(value))
    return data_format


def cu_conv_output_length(input_length, filter_size, padding, stride, dilation=1):
    if input_length is None:
        return None
    assert padding in {'same', 'valid', 'full', 'causal'}
    dilated = filter_size + (filter_size - 1) * (dilation - 1)
    if padding in ('same', 'causal'):
        output_length = input_length
    elif padding == 'valid':
        output_length = input_length - dilated + 1
    else:
        output_length = input_length + dilated - 1
    return (output_length + stride - 1) // stride


def cu_conv_connected_inputs(input_shape, kernel_shape, output_position, strides, padding):
    ranges = []
    for d in range(len(input_shape)):
        left_shift = int(kernel_shape[d] / 2)
        right_shift = kernel_shape[d] - left_shift
        center = output_position[d] * strides[d]
        if padding == 'valid':
            center += left_shift
        ranges.append(range(max(0, center - left_shift), min(input_shape[d], center + right_shift)))
    return ranges


def cu_conv_kernel_mask(input_shape, kernel_shape, strides, padding):
    """keras conv_utils.conv_kernel_mask: boolean [*input_shape, *output_shape], True where the output position reads the
    input position"""
    import itertools
    if isinstance(kernel_shape, int):
        kernel_shape = (kernel_shape,) * len(input_shape)
    if isinstance(strides, int):
        strides = (strides,) * len(input_shape)
    output_shape = tuple(0 if input_shape[d] == 0 else cu_conv_output_length(input_shape[d], kernel_shape[d], padding, strides[d])
                         for d in range(len(input_shape)))
    mask = np.zeros(tuple(input_shape) + output_shape, bool)
    for output_position in itertools.product(*[range(d) for d in output_shape]):
        ticks = cu_conv_connected_inputs(input_shape, kernel_shape, output_position, strides, padding)
        for input_position in itertools.product(*ticks):
            mask[input_position + output_position] = True
    return mask


def shape_type_conversion(fn):
    """tf_utils.shape_type_conversion: shapes go in as tuples and come out as TensorShape"""
    import functools

    @functools.wraps(fn)
    def wrapper(instance, input_shape):
        if input_shape is not None:
            input_shape = tuple(input_shape) if not isinstance(input_shape, list) else [tuple(s) for s in input_shape]
        out = fn(instance, input_shape)
        if out is not None and not isinstance(out, list):
            out = TensorShape(tuple(out))
        return out
    return wrapper


def activations_get(identifier):
    if identifier is None or identifier == 'linear':
        return lambda x: x
    if identifier == 'relu':
        return lambda x: Tensor(np.maximum(A(x), 0).astype(A(x).dtype))
    if identifier == 'elu':
        return lambda x: Tensor(np.where(A(x) > 0, A(x), np.exp(np.minimum(A(x), 0)) - 1).astype(A(x).dtype))
    if callable(identifier):
        return identifier
    raise ValueError('shim: activation %r' % (identifier,))


def linalg_matmul(a, b, **kw):
    """tf.linalg.matmul of 2-D float tensors; float64 accumulation, result in the input dtype"""
    a, b = A(a), A(b)
    return Tensor((a.astype(np.float64) @ b.astype(np.float64)).astype(a.dtype))


def sparse_tensor_dense_mat_mul(a_indices, a_values, a_shape, b, adjoint_a=False, adjoint_b=False, **kw):
    """gen_sparse_ops.sparse_tensor_dense_mat_mul: SparseTensor(a_indices, a_values, a_shape) @ b (adjoints applied first)"""
    assert not adjoint_a
    idx = np.asarray([tuple(i) for i in a_indices], np.int64).reshape(-1, 2)
    vals = A(a_values).astype(np.float64)
    dense = np.zeros(tuple(int(v) for v in a_shape), np.float64)
    np.add.at(dense, (idx[:, 0], idx[:, 1]), vals)
    bb = A(b).astype(np.float64)
    if adjoint_b:
        bb = bb.T
    return Tensor((dense @ bb).astype(A(b).dtype))


# ---- scripted randomness: tf.random.uniform returns minval + u * (maxval - minval) for the next u of RANDOM_SCRIPT -------
# (float32 arithmetic as TensorFlow's random_uniform; integer dtypes: minval + floor(u * (maxval - minval)))
RANDOM_SCRIPT = []


def random_uniform(shape=(), minval=0, maxval=None, dtype=np.float32, seed=None, **kw):
    dt = as_np_dtype(dtype)
    shp = tuple(int(v) for v in np.ravel(A(shape))) if not isinstance(shape, (list, tuple)) or len(shape) else ()
    n = int(np.prod(shp)) if shp else 1
    if len(RANDOM_SCRIPT) < n:
        raise RuntimeError('tf_shim.RANDOM_SCRIPT exhausted')
    u = np.array([RANDOM_SCRIPT.pop(0) for _ in range(n)], np.float32).reshape(shp)
    if np.issubdtype(dt, np.integer):
        if maxval is None:
            raise ValueError('maxval is required for integer dtypes')
        lo, hi = int(A(minval)), int(A(maxval))
        return Tensor((lo + np.floor(u.astype(np.float64) * (hi - lo))).astype(dt))
    hi = np.float32(1.0 if maxval is None else A(maxval))
    lo = np.float32(A(minval))
    return Tensor((u * (hi - lo) + lo).astype(dt))


def roll_(x, shift, axis, **kw):
    if isinstance(x, (list, tuple)):         # a python sequence mixing tensors and numbers is packed like tf.stack
        x = np.array([np.asarray(A(v)).reshape(()) for v in x])
    return Tensor(np.roll(A(x), int(A(shift)), int(A(axis))))


def _populate(m):
    n = m.__name__
    if n == 'pystrum':
        m.__version__ = '0.2'
    if n == 'tensorflow':
        d = dict(
            Tensor=Tensor, TensorShape=TensorShape, convert_to_tensor=convert_to_tensor, constant=constant,
            cast=cast, stack=stack, concat=concat, floor=floor, round=round_, exp=exp, square=square,
            less=less, greater=greater, logical_not=logical_not, clip_by_value=clip_by_value,
            reshape=reshape, gather=gather, reduce_any=reduce_any, reduce_sum=reduce_sum, linspace=linspace,
            range=range_, size=size, tile=tile, ones=ones, zeros=zeros, map_fn=map_fn,
            float32=DType(np.float32), float64=DType(np.float64), int32=DType(np.int32),
            int64=DType(np.int64), bool=DType(np.bool_), float16=DType(np.float16),
            expand_dims=lambda x, axis: k_expand_dims(x, axis),
            shape=shape_, transpose=transpose_, reduce_prod=reduce_prod, reduce_min=reduce_min, reduce_max=reduce_max,
            newaxis=None, minimum=lambda a, b: Tensor(np.minimum(A(a), A(b))), maximum=lambda a, b: Tensor(np.maximum(A(a), A(b))),
            reduce_mean=lambda x, axis=None, keepdims=False: Tensor(np.mean(A(x), axis=axis, keepdims=keepdims)),
            logical_and=lambda a, b: Tensor(np.logical_and(A(a), A(b))), greater_equal=lambda a, b: T(a) >= b,
            is_tensor=lambda x: isinstance(x, Tensor), roll=roll_,
        )
        for k, v in d.items():
            setattr(m, k, v)
        m.__version__ = '2.99.shim'
    if n == 'tensorflow.random':
        m.uniform = random_uniform
    if n == 'tensorflow.math':
        m.divide_no_nan = divide_no_nan
        m.log = log
        m.exp = exp
        m.reduce_prod = reduce_prod
    if n == 'tensorflow.debugging':
        m.assert_greater_equal = assert_greater_equal
        m.assert_less_equal = assert_less_equal
        m.assert_all_finite = assert_all_finite
        m.assert_equal = lambda x, y, message='', **kw: _assert(np.all(A(x) == A(y)), message)
        m.assert_non_negative = lambda x, message='', **kw: _assert(np.all(A(x) >= 0), message)
        m.assert_greater = lambda x, y, message='', **kw: _assert(np.all(A(x) > A(y)), message)
    if n == 'tensorflow.errors':
        m.InvalidArgumentError = InvalidArgumentError
    if n == 'tensorflow.compat.v1':
        m.Dimension = Dimension
        m.div_no_nan = divide_no_nan
    if n == 'tensorflow.nn':
        m.convolution = nn_convolution
    if n == 'tensorflow.experimental.numpy':
        m.diff = lambda x, **kw: Tensor(np.diff(A(x)))
    if n == 'tensorflow.dtypes':
        m.as_dtype = lambda d: d if isinstance(d, DType) else DType(d)
    if n == 'tensorflow.keras.backend':
        d = dict(expand_dims=k_expand_dims, shape=k_shape, int_shape=k_int_shape, ndim=k_ndim, sum=k_sum,
                 mean=k_mean, square=k_square, argmax=k_argmax, one_hot=k_one_hot, reshape=reshape,
                 concatenate=k_concatenate, permute_dimensions=k_permute_dimensions, batch_dot=k_batch_dot,
                 bias_add=k_bias_add, epsilon=k_epsilon, floor=floor, exp=exp, log=log,
                 cast=cast, stack=stack, clip=clip_by_value,
                 min=lambda x, axis=None, keepdims=False: Tensor(np.min(A(x), axis=axis, keepdims=keepdims)),
                 max=lambda x, axis=None, keepdims=False: Tensor(np.max(A(x), axis=axis, keepdims=keepdims)),
                 flatten=lambda x: Tensor(A(x).reshape(-1)), transpose=lambda x: Tensor(np.transpose(A(x))))
        for k, v in d.items():
            setattr(m, k, v)
    if n == 'tensorflow.keras.layers':
        m.Layer = Layer
        # symbolic, shape-propagating recorders for the Keras layer constructors the network builders call
        import keras_record
        keras_record.populate_layers_module(m)
    if n == 'tensorflow.keras.models':
        import keras_record
        m.Model = keras_record.Model
    if n == 'tensorflow.keras':
        import keras_record
        m.Model = keras_record.Model
    if n == 'tensorflow.keras.activations':
        import keras_record
        m.softmax = keras_record.activations_softmax
        m.get = activations_get
        m.serialize = lambda a: getattr(a, '__name__', None)
    if n in ('tensorflow.keras.initializers', 'tensorflow.keras.regularizers', 'tensorflow.keras.constraints'):
        m.get = lambda identifier: identifier
        m.serialize = lambda v: v
    if n == 'tensorflow.python.keras.utils.conv_utils':
        m.normalize_tuple = cu_normalize_tuple
        m.normalize_padding = cu_normalize_padding
        m.normalize_data_format = cu_normalize_data_format
        m.conv_output_length = cu_conv_output_length
        m.conv_kernel_mask = cu_conv_kernel_mask
    if n == 'tensorflow.python.keras.utils.tf_utils':
        m.shape_type_conversion = shape_type_conversion
    if n == 'tensorflow.linalg':
        m.matmul = linalg_matmul
    if n == 'tensorflow.python.ops.gen_sparse_ops':
        m.sparse_tensor_dense_mat_mul = sparse_tensor_dense_mat_mul
    if n == 'tensorflow.keras.losses':
        m.CategoricalCrossentropy = KerasCCE


def install():
    """Install the shim; afterwards `import tensorflow` etc. resolve to it."""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return
    sys.meta_path.insert(0, _Finder())
    import importlib
    # make attribute access and `import a.b.c` agree for the sub-modules we populate
    for name in ('tensorflow', 'tensorflow.math', 'tensorflow.debugging', 'tensorflow.errors',
                 'tensorflow.compat', 'tensorflow.compat.v1', 'tensorflow.dtypes', 'tensorflow.keras',
                 'tensorflow.keras.backend', 'tensorflow.keras.layers', 'tensorflow.keras.losses',
                 'tensorflow.keras.models', 'tensorflow.keras.activations', 'tensorflow.keras.initializers',
                 'tensorflow.keras.regularizers', 'tensorflow.keras.constraints', 'tensorflow.linalg',
                 'tensorflow.python.keras.utils.conv_utils', 'tensorflow.python.keras.utils.tf_utils',
                 'tensorflow.python.ops.gen_sparse_ops', 'tensorflow.keras.utils', 'tensorflow.keras.datasets',
                 'tensorflow.python', 'tensorflow.python.keras', 'tensorflow.python.keras.utils',
                 'tensorflow.python.ops', 'tensorflow.nn', 'tensorflow.random', 'tensorflow.experimental', 'tensorflow.experimental.numpy', 'pystrum', 'pystrum.pynd', 'pystrum.pytools'):
        mod = importlib.import_module(name)
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(sys.modules[parent], child, mod)
