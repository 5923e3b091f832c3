"""
CPU execution of a RECORDED Keras layer graph (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The graphs in tests/golden/unet_graph.json are what the reference's own builders construct
(neurite/tf/models.py: unet :88-246, conv_enc :1309-1442, conv_dec :1445-1617, add_prior :378-436), recorded by
tests/golden/keras_record.py.  `run` evaluates such a graph layer by layer, so the network STRUCTURE of this oracle is the
reference's, not a restatement; only the Keras layer semantics are restated (TF semantics, unpinned -- no TensorFlow in
this image):
  Conv{1,2,3}D   cross-correlation, stride 1, 'same' (odd kernels: (k-1)*dil/2 zeros each side) or 'valid', bias, then the
                 activation ('elu' = x>0 ? x : exp(x)-1, 'relu', 'sigmoid', 'linear'); float64 accumulation, float32 result
  MaxPooling     stride = pool size; 'same' = partial windows at the end, 'valid' = floor
  UpSampling     nearest repeat;  Concatenate / Add / Multiply;  Activation;  Lambda [softmax, axis]
  BatchNormalization  inference form (moving statistics);  Dropout  inference form (identity)
1-D and 2-D graphs are lifted to 3-D with leading singleton axes.  One batch entry at a time.
"""

import numpy as np

from . import c_oracle as co


def _act(y, name):
    if name in (None, 'linear'):
        return y
    if name == 'elu':
        return np.where(y > 0, y, np.exp(np.minimum(y, 0)) - 1)
    if name == 'relu':
        return np.maximum(y, 0)
    if name == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-y))
    if name == 'softmax':
        e = np.exp(y - y.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)
    raise NotImplementedError('activation %r' % (name,))


def _lift(v, nd, fill=1):
    return (fill,) * (3 - nd) + tuple(int(a) for a in v)


def _conv(x, cfg, kernel, bias, nd):
    k3 = _lift(cfg['kernel_size'], nd)
    dil = cfg['dilation_rate']
    assert len(set(dil)) == 1 and set(cfg['strides']) == {1}, 'isotropic dilation, stride 1'
    d = int(dil[0])
    kernel = np.asarray(kernel, np.float32).reshape(k3 + kernel.shape[-2:])
    assert all(k % 2 == 1 for k in k3), 'odd kernels (Keras pads even kernels asymmetrically)'
    y = co.conv3d_same(np.ascontiguousarray(x, np.float32), kernel, None if bias is None else np.asarray(bias, np.float32),
                       dilation=d, elu=False).astype(np.float64)
    if cfg['padding'] == 'valid':
        sl = []
        for ax in range(3):
            p = (k3[ax] - 1) * d // 2
            sl.append(slice(p, y.shape[ax] - p))
        y = y[tuple(sl)]
    return _act(y, cfg['activation'])


def _maxpool(x, pool3, padding):
    X, Y, Z, C = x.shape
    if padding == 'same':
        o = [-(-s // p) for s, p in zip((X, Y, Z), pool3)]
        pad = np.full((o[0] * pool3[0], o[1] * pool3[1], o[2] * pool3[2], C), -np.inf, x.dtype)
        pad[:X, :Y, :Z] = x
    else:
        o = [s // p for s, p in zip((X, Y, Z), pool3)]
        pad = x[:o[0] * pool3[0], :o[1] * pool3[1], :o[2] * pool3[2]]
    return pad.reshape(o[0], pool3[0], o[1], pool3[1], o[2], pool3[2], C).max(axis=(1, 3, 5))


def run(graph, inputs, weights, bn=None, return_all=False):
    """
    graph:   {'inputs': [...], 'outputs': [...], 'layers': [{name, class, config, inputs, output_shape}, ...]}
    inputs:  list of arrays [*spatial, C] (one batch entry each), in the order of graph['inputs']
    weights: {layer name: (kernel [k.., Cin, Cout], bias [Cout])};  bn: {layer name: (gamma, beta, mean, var)}
    Returns the output [*spatial, C] in float64 (or the dict of all tensors).
    """
    layers = graph['layers']
    nd = len(layers[0]['output_shape']) - 2
    t = {}
    for name, x in zip(graph['inputs'], inputs):
        x = np.asarray(x, np.float64)
        t[name] = x.reshape(_lift(x.shape[:-1], nd) + x.shape[-1:])
    for lay in layers:
        name, cls, cfg = lay['name'], lay['class'], lay['config']
        src = [t[i] for i in lay['inputs']]
        if cls == 'InputLayer':
            assert name in t, 'no array given for input ' + name
        elif cls.startswith('Conv'):
            k, b = weights[name]
            t[name] = _conv(src[0], cfg, k, b, nd)
        elif cls.startswith('MaxPooling'):
            assert cfg['strides'] == cfg['pool_size']
            t[name] = _maxpool(src[0], _lift(cfg['pool_size'], nd), cfg['padding'])
        elif cls.startswith('UpSampling'):
            s = _lift(cfg['size'], nd)
            t[name] = src[0].repeat(s[0], 0).repeat(s[1], 1).repeat(s[2], 2)
        elif cls == 'Concatenate':
            assert cfg['axis'] in (-1, nd + 1)
            t[name] = np.concatenate(src, -1)
        elif cls == 'Add':
            t[name] = src[0] + src[1]
        elif cls == 'Multiply':
            t[name] = src[0] * src[1]
        elif cls == 'Activation':
            t[name] = _act(src[0], cfg['activation'])
        elif cls == 'Lambda':
            (fn, axis), = cfg['function']
            assert fn == 'softmax' and axis in (-1, nd + 1)
            t[name] = _act(src[0], 'softmax')
        elif cls == 'BatchNormalization':
            gamma, beta, mean, var = [np.asarray(p, np.float64) for p in bn[name]]
            t[name] = (src[0] - mean) / np.sqrt(var + cfg['epsilon']) * gamma + beta
        elif cls == 'Dropout':
            t[name] = src[0]
        else:
            raise NotImplementedError(cls)
        want = [int(v) for v in lay['output_shape'][1:]]
        got = list(t[name].shape[3 - nd:])
        assert got == want, 'layer %s: shape %s, the recorded graph says %s' % (name, got, want)
    out = {k: v.reshape(v.shape[3 - nd:]) for k, v in t.items()}
    return out if return_all else out[graph['outputs'][0]]
