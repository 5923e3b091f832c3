"""
Recording stand-ins for the Keras layer constructors the reference's network builders call
(neurite/tf/models.py: unet :88-246, conv_enc :1309-1442, conv_dec :1445-1617, add_prior :378-436).

The builders are plain Python over `KL.Conv3D(...)(x)`-style calls; running them on these classes yields the layer graph
the REFERENCE builds -- name, class, constructor arguments, producers of every input, output shape -- without
TensorFlow.  `tests/golden/make_golden.py` stores those graphs in `unet_graph.json`; `tests/test_unet_graph.py`
compares the graphs `neurite_amd.models` builds against them.  Only shapes are propagated (Keras' rules for
'same' / 'valid' convolution and pooling, up-sampling, concatenation); no numerics.

Test infrastructure only: nothing under neurite_amd/ imports this file.
"""

import collections

RECORD = []            # layer records in the order the layers are CALLED (Keras' topological order for these builders)
_AUTO = collections.Counter()


def reset():
    RECORD.clear()
    _AUTO.clear()


class KShape(tuple):
    def as_list(self):
        return list(self)


class KTensor:
    """symbolic tensor: static shape (batch = None) and the layer that produced it"""

    def __init__(self, shape, layer, note=None):
        self.shape = KShape(shape)
        self._layer = layer
        self.note = note

    def get_shape(self):
        return self.shape

    @property
    def name(self):
        return self._layer.name + '/output'


def _auto_name(base):
    # Keras: snake-cased class name, then base_1, base_2, ...
    n = _AUTO[base]
    _AUTO[base] += 1
    return base if n == 0 else '%s_%d' % (base, n)


def _tup(v, nd):
    if isinstance(v, (list, tuple)):
        assert len(v) == nd, (v, nd)
        return tuple(int(a) for a in v)
    return (int(v),) * nd


class _KLayer:
    keras_class = 'Layer'
    auto_base = 'layer'

    def __init__(self, name=None, **kw):
        self.name = name if name is not None else _auto_name(self.auto_base)
        self.output = None
        self.input = None

    def config(self):
        return {}

    def out_shape(self, shapes):
        return shapes[0]

    def __call__(self, inputs, **kw):
        ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        for t in ins:
            assert isinstance(t, KTensor), 'layer %s called on %r' % (self.name, type(t))
        shape = self.out_shape([t.shape for t in ins])
        out = KTensor(shape, self)
        self.input = inputs
        self.output = out
        RECORD.append({'name': self.name, 'class': self.keras_class, 'config': self.config(),
                       'inputs': [t._layer.name for t in ins], 'output_shape': [None if s is None else int(s) for s in shape]})
        return out


class InputLayer(_KLayer):
    keras_class = 'InputLayer'
    auto_base = 'input'


def Input(shape=None, name=None, **kw):
    lay = InputLayer(name=name)
    t = KTensor((None,) + tuple(int(s) for s in shape), lay)
    lay.output = t
    RECORD.append({'name': lay.name, 'class': 'InputLayer', 'config': {}, 'inputs': [],
                   'output_shape': [None] + [int(s) for s in shape]})
    return t


def _act_name(a):
    if a is None:
        return 'linear'
    if isinstance(a, str):
        return a
    return getattr(a, '__name__', repr(a))


class _ConvND(_KLayer):
    nd = 3

    def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format=None, dilation_rate=1, activation=None,
                 use_bias=True, name=None, **kw):
        self.auto_base = 'conv%dd' % self.nd
        super().__init__(name=name)
        self.keras_class = 'Conv%dD' % self.nd
        self.filters = int(filters)
        self.kernel_size = _tup(kernel_size, self.nd)
        self.strides = _tup(strides, self.nd)
        self.padding = padding
        self.dilation_rate = _tup(dilation_rate, self.nd)
        self.activation = _act_name(activation)
        self.use_bias = bool(use_bias)
        self.extra = sorted(kw)

    def config(self):
        return {'filters': self.filters, 'kernel_size': list(self.kernel_size), 'strides': list(self.strides),
                'padding': self.padding, 'dilation_rate': list(self.dilation_rate), 'activation': self.activation,
                'use_bias': self.use_bias}

    def out_shape(self, shapes):
        s = shapes[0]
        assert len(s) == self.nd + 2, (self.name, s)
        sp = []
        for d in range(self.nd):
            n = s[1 + d]
            ke = (self.kernel_size[d] - 1) * self.dilation_rate[d] + 1
            if self.padding == 'same':
                sp.append(-(-n // self.strides[d]))
            elif self.padding == 'valid':
                sp.append((n - ke) // self.strides[d] + 1)
            else:
                raise ValueError('padding %r' % (self.padding,))
        return (s[0],) + tuple(sp) + (self.filters,)


class Conv1D(_ConvND):
    nd = 1


class Conv2D(_ConvND):
    nd = 2


class Conv3D(_ConvND):
    nd = 3


class _PoolND(_KLayer):
    nd = 3

    def __init__(self, pool_size=2, strides=None, padding='valid', name=None, **kw):
        self.auto_base = 'max_pooling%dd' % self.nd
        super().__init__(name=name)
        self.keras_class = 'MaxPooling%dD' % self.nd
        self.pool_size = _tup(pool_size, self.nd)
        self.strides = self.pool_size if strides is None else _tup(strides, self.nd)
        self.padding = padding

    def config(self):
        return {'pool_size': list(self.pool_size), 'strides': list(self.strides), 'padding': self.padding}

    def out_shape(self, shapes):
        s = shapes[0]
        sp = []
        for d in range(self.nd):
            n = s[1 + d]
            if self.padding == 'same':
                sp.append(-(-n // self.strides[d]))
            else:
                sp.append((n - self.pool_size[d]) // self.strides[d] + 1)
        return (s[0],) + tuple(sp) + (s[-1],)


class MaxPooling1D(_PoolND):
    nd = 1


class MaxPooling2D(_PoolND):
    nd = 2


class MaxPooling3D(_PoolND):
    nd = 3


class _UpND(_KLayer):
    nd = 3

    def __init__(self, size=2, name=None, **kw):
        self.auto_base = 'up_sampling%dd' % self.nd
        super().__init__(name=name)
        self.keras_class = 'UpSampling%dD' % self.nd
        self.size = _tup(size, self.nd)

    def config(self):
        return {'size': list(self.size)}

    def out_shape(self, shapes):
        s = shapes[0]
        return (s[0],) + tuple(s[1 + d] * self.size[d] for d in range(self.nd)) + (s[-1],)


class UpSampling1D(_UpND):
    nd = 1


class UpSampling2D(_UpND):
    nd = 2


class UpSampling3D(_UpND):
    nd = 3


class Dropout(_KLayer):
    keras_class = 'Dropout'
    auto_base = 'dropout'

    def __init__(self, rate, noise_shape=None, seed=None, name=None, **kw):
        super().__init__(name=name)
        self.rate = float(rate)
        self.noise_shape = None if noise_shape is None else [None if v is None else int(v) for v in noise_shape]

    def config(self):
        return {'rate': self.rate, 'noise_shape': self.noise_shape}


class BatchNormalization(_KLayer):
    keras_class = 'BatchNormalization'
    auto_base = 'batch_normalization'

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        self.axis, self.momentum, self.epsilon = int(axis), float(momentum), float(epsilon)

    def config(self):
        return {'axis': self.axis, 'momentum': self.momentum, 'epsilon': self.epsilon}


class Activation(_KLayer):
    keras_class = 'Activation'
    auto_base = 'activation'

    def __init__(self, activation, name=None, **kw):
        super().__init__(name=name)
        self.activation = _act_name(activation)

    def config(self):
        return {'activation': self.activation}


class _Probe(KTensor):
    """handed to a Lambda's function: records the tf.keras.activations call made on it"""


_LAMBDA_TRACE = []


def activations_softmax(x, axis=-1):
    _LAMBDA_TRACE.append(('softmax', int(axis)))
    return x


class Lambda(_KLayer):
    keras_class = 'Lambda'
    auto_base = 'lambda'

    def __init__(self, function, name=None, **kw):
        super().__init__(name=name)
        self.function = function
        self.trace = None

    def config(self):
        return {'function': self.trace}

    def out_shape(self, shapes):
        del _LAMBDA_TRACE[:]
        probe = _Probe(shapes[0], self)
        r = self.function(probe)
        assert r is probe, 'Lambda %s: only pass-through tf.keras.activations calls are modelled' % self.name
        self.trace = [list(t) for t in _LAMBDA_TRACE]
        return shapes[0]


class _Merge(_KLayer):
    def __init__(self, name=None, **kw):
        super().__init__(name=name)


class Concatenate(_Merge):
    keras_class = 'Concatenate'
    auto_base = 'concatenate'

    def __init__(self, axis=-1, name=None, **kw):
        super().__init__(name=name)
        self.axis = int(axis)

    def config(self):
        return {'axis': self.axis}

    def out_shape(self, shapes):
        rank = len(shapes[0])
        ax = self.axis % rank
        for s in shapes[1:]:
            for d in range(rank):
                if d != ax and s[d] != shapes[0][d]:
                    raise ValueError('A `Concatenate` layer requires inputs with matching shapes except for the concat '
                                     'axis. Got inputs shapes: %s' % (shapes,))
        out = list(shapes[0])
        out[ax] = sum(s[ax] for s in shapes)
        return tuple(out)


class Add(_Merge):
    keras_class = 'Add'
    auto_base = 'add'

    def out_shape(self, shapes):
        for s in shapes[1:]:
            if tuple(s) != tuple(shapes[0]):
                raise ValueError('Operands could not be broadcast together with shapes %s' % (shapes,))
        return shapes[0]


class Multiply(Add):
    keras_class = 'Multiply'
    auto_base = 'multiply'


def concatenate(inputs, axis=-1, name=None, **kw):
    return Concatenate(axis=axis, name=name)(inputs)


def add(inputs, name=None, **kw):
    return Add(name=name)(inputs)


def multiply(inputs, name=None, **kw):
    return Multiply(name=name)(inputs)


class Model:
    """keras.Model as the builders use it: inputs / outputs, get_layer(name).output, and the list of layers that lie
    between them (every recorded layer reachable backwards from the outputs, in call order)."""

    def __init__(self, inputs=None, outputs=None, name=None, **kw):
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        self.name = name
        by_name = {}
        for r in RECORD:
            by_name[r['name']] = r          # last definition wins (names are unique in these builders)
        seen, stack = set(), [t._layer.name for t in self.outputs]
        while stack:
            n = stack.pop()
            if n in seen:
                continue
            seen.add(n)
            stack.extend(by_name[n]['inputs'])
        self.records = [r for r in RECORD if r['name'] in seen]
        self._layers = {}
        for t in self.outputs + self.inputs:
            self._collect(t)

    def _collect(self, t):
        lay = t._layer
        if lay.name in self._layers:
            return
        self._layers[lay.name] = lay
        ins = lay.input
        if ins is None:
            return
        for u in (ins if isinstance(ins, (list, tuple)) else [ins]):
            self._collect(u)

    @property
    def input(self):
        return self.inputs[0] if len(self.inputs) == 1 else self.inputs

    @property
    def output(self):
        return self.outputs[0] if len(self.outputs) == 1 else self.outputs

    def get_layer(self, name):
        if name not in self._layers:
            raise ValueError('No such layer: %s' % name)
        return self._layers[name]

    def graph(self):
        return {'name': self.name, 'inputs': [t._layer.name for t in self.inputs],
                'outputs': [t._layer.name for t in self.outputs], 'layers': self.records}


def populate_layers_module(m):
    for k in ('Input', 'InputLayer', 'Conv1D', 'Conv2D', 'Conv3D', 'MaxPooling1D', 'MaxPooling2D', 'MaxPooling3D',
              'UpSampling1D', 'UpSampling2D', 'UpSampling3D', 'Dropout', 'BatchNormalization', 'Activation', 'Lambda',
              'Concatenate', 'Add', 'Multiply', 'concatenate', 'add', 'multiply'):
        setattr(m, k, globals()[k])
