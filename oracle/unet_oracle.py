"""
CPU restatement of neurite's unet forward pass (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows neurite/tf/models.py:88-246 (unet) -> :1309-1442 (conv_enc) -> :1445-1617 (conv_dec) for the options
the tests use: nb_conv_per_level, feat_mult / explicit features (taken from the weight shapes), pool_size,
activation 'elu', residuals, inference BatchNorm, softmax / linear head.  Keras layer semantics are restated
(TF semantics, unpinned): Conv = cross-correlation, SAME padding, bias, ELU = x>0 ? x : exp(x)-1;
MaxPooling SAME = partial windows at the end; UpSampling = nearest repeat; concatenate([skip, up]).
Convolutions accumulate in float64 (oracle.c orc_conv3d_same_f32), everything else is float64 NumPy.
Weights: dict layer_name -> (kernel [k,k,k,Cin,Cout], bias [Cout]); BatchNorm: name -> (gamma, beta, mean, var).
"""

import numpy as np

from . import c_oracle as co


def elu(x):
    return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1)


def maxpool_same(x, pool):
    X, Y, Z, C = x.shape
    ox, oy, oz = [-(-s // p) for s, p in zip((X, Y, Z), pool)]
    pad = np.full((ox * pool[0], oy * pool[1], oz * pool[2], C), -np.inf, x.dtype)
    pad[:X, :Y, :Z] = x
    return pad.reshape(ox, pool[0], oy, pool[1], oz, pool[2], C).max(axis=(1, 3, 5))


def upsample(x, size):
    return x.repeat(size[0], 0).repeat(size[1], 1).repeat(size[2], 2)


def conv(x, kernel, bias, activation=None, dilation=1):
    y = co.conv3d_same(x.astype(np.float32), kernel, bias, dilation=dilation, elu=False).astype(np.float64)
    return elu(y) if activation == 'elu' else y


def bn(x, params, eps=1e-3):
    gamma, beta, mean, var = [np.asarray(p, np.float64) for p in params]
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def unet_forward(x, weights, nb_levels, nb_conv_per_level=1, pool=(2, 2, 2), prefix='unet', activation='elu',
                 use_residuals=False, bn_params=None, final_pred_activation='softmax', dilation_rate_mult=1,
                 return_all=False):
    """x [X, Y, Z, C] (one batch entry).  Returns the prediction [X, Y, Z, nb_labels] (float64)."""
    t = {}
    last = x.astype(np.float64)
    for level in range(nb_levels):                                               # conv_enc :1362-1438
        first = last
        dil = dilation_rate_mult ** level
        for c in range(nb_conv_per_level):
            name = '%s_conv_downarm_%d_%d' % (prefix, level, c)
            tail = use_residuals and c == nb_conv_per_level - 1          # :1384-1388: built without conv_kwargs
            last = conv(last, *weights[name], activation=None if tail else activation, dilation=1 if tail else dil)
            t[name] = last
        if use_residuals:
            add = first
            if first.shape[-1] > 1 and last.shape[-1] > 1 and first.shape[-1] != last.shape[-1]:
                name = '%s_expand_down_merge_%d' % (prefix, level)
                add = conv(first, *weights[name], activation=activation, dilation=dil)
            last = elu(add + last) if activation == 'elu' else add + last
        if bn_params is not None:
            last = bn(last, bn_params['%s_bn_down_%d' % (prefix, level)])
        if level < nb_levels - 1:
            last = maxpool_same(last, pool)
    for level in range(nb_levels - 1):                                           # conv_dec :1514-1592
        dil = dilation_rate_mult ** (nb_levels - 2 - level)
        last = upsample(last, pool)
        up = last
        skip = t['%s_conv_downarm_%d_%d' % (prefix, nb_levels - 2 - level, nb_conv_per_level - 1)]
        last = np.concatenate([skip, last], -1)
        for c in range(nb_conv_per_level):
            name = '%s_conv_uparm_%d_%d' % (prefix, nb_levels + level, c)
            tail = use_residuals and c == nb_conv_per_level - 1          # :1552-1555
            last = conv(last, *weights[name], activation=None if tail else activation, dilation=1 if tail else dil)
            t[name] = last
        if use_residuals:
            add = up
            if up.shape[-1] > 1 and last.shape[-1] > 1 and up.shape[-1] != last.shape[-1]:
                name = '%s_expand_up_merge_%d' % (prefix, level)
                add = conv(up, *weights[name], activation=activation, dilation=dil)
            last = elu(last + add) if activation == 'elu' else last + add
        if bn_params is not None:
            last = bn(last, bn_params['%s_bn_up_%d' % (prefix, level)])
    k, b = weights['%s_likelihood' % prefix]
    like = conv(last, k, b, activation=None)
    t['%s_likelihood' % prefix] = like
    if final_pred_activation == 'softmax':
        e = np.exp(like - like.max(-1, keepdims=True))
        pred = e / e.sum(-1, keepdims=True)
    else:
        pred = like
    t['%s_prediction' % prefix] = pred
    return t if return_all else pred
