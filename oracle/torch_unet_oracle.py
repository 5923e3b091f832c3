"""
TEST INFRASTRUCTURE ONLY -- differentiable float64 restatement of the conv_enc / conv_dec / unet graphs
(neurite/tf/models.py:88-246, 1309-1617) with torch CPU ops, used as the gradient oracle of csrc/conv_bwd.hip.

Keras semantics restated: Conv3D = cross-correlation, SAME padding, kernel [kd,kh,kw,Cin,Cout];
ELU = x > 0 ? x : exp(x) - 1; MaxPooling3D stride = pool size (SAME pads -inf at the end); UpSampling3D = nearest repeat;
softmax over channels.  Forward values are pinned against oracle/unet_oracle.py (C conv oracle) in tests/test_oracle.py;
the gradients are torch.autograd's (TF's gradient rules for these ops are the textbook ones).
Only tests/ may import this module.
"""

import torch
import torch.nn.functional as Fn

F64 = torch.float64


def _cf(x):      # [B, X, Y, Z, C] -> [B, C, X, Y, Z]
    return x.permute(0, 4, 1, 2, 3)


def _cl(x):
    return x.permute(0, 2, 3, 4, 1)


def keras_activation(y, activation):
    """tf.keras.activations (TF 2.x) restated: elu = exp(x) - 1 below 0 (alpha 1), hard_sigmoid = clip(0.2 x + 0.5, 0, 1),
    leaky_relu slope 0.2, selu with Keras' constants, softmax over the channel (last) axis.  The reference hands the activation
    string straight to Keras (neurite/tf/models.py:1346, 1429, 1507, 1588; layers.py:1101)."""
    if activation in (None, 'linear'):
        return y
    if activation == 'elu':
        return torch.where(y > 0, y, torch.exp(torch.clamp(y, max=0.0)) - 1)
    if activation == 'relu':
        return torch.relu(y)
    if activation == 'sigmoid':
        return torch.sigmoid(y)
    if activation == 'tanh':
        return torch.tanh(y)
    if activation == 'softplus':
        return Fn.softplus(y, threshold=1e9)
    if activation == 'softsign':
        return y / (1 + y.abs())
    if activation == 'selu':
        scale, alpha = 1.05070098735548049342, 1.67326324235437728481
        return scale * torch.where(y > 0, y, alpha * (torch.exp(torch.clamp(y, max=0.0)) - 1))
    if activation == 'exponential':
        return torch.exp(y)
    if activation == 'hard_sigmoid':
        return torch.clamp(0.2 * y + 0.5, 0.0, 1.0)
    if activation == 'leaky_relu':
        return torch.where(y > 0, y, 0.2 * y)
    if activation == 'softmax':
        return torch.softmax(y, -1)
    raise NotImplementedError(activation)


def conv3d_same(x, kernel, bias, dilation=1, activation=None, padding='same'):
    """x [B,X,Y,Z,Cin] channels-last, kernel [kx,ky,kz,Cin,Cout]; Keras Conv3D with 'same' (odd kernels) or 'valid' padding."""
    w = kernel.permute(4, 3, 0, 1, 2)
    pad = [dilation * (k - 1) // 2 for k in kernel.shape[:3]] if padding == 'same' else 0
    y = Fn.conv3d(_cf(x), w, bias, padding=pad, dilation=dilation)
    return keras_activation(_cl(y), activation)


def maxpool_same(x, pool):
    B, X, Y, Z, C = x.shape
    pads = []
    for n, p in zip((Z, Y, X), (pool[2], pool[1], pool[0])):      # F.pad order: last dim first
        pads += [0, (-n) % p]
    xc = Fn.pad(_cf(x), pads, value=float('-inf'))
    return _cl(Fn.max_pool3d(xc, kernel_size=tuple(pool), stride=tuple(pool)))


def upsample(x, up):
    for d, u in enumerate(up):
        x = x.repeat_interleave(u, dim=1 + d)
    return x


def forward(net, x, params=None, return_tensors=None, dropout_scales=None, bn_train=False):
    """Run a neurite_amd.models.ConvNet graph (3-D nets) in float64 on the CPU.  params: {layer: (kernel, bias)} of
    float64 tensors (requires_grad as the caller likes); default: copies of the model's weights."""
    if params is None:
        params = {k: (m.kernel.detach().cpu().double(), m.bias.detach().cpu().double())
                  for k, m in net.layers_by_name.items() if hasattr(m, 'kernel')}
        params.update({k: (m.gamma.detach().cpu().double(), m.beta.detach().cpu().double())
                       for k, m in net.layers_by_name.items() if hasattr(m, 'gamma')})
    t = {}
    for op in net.ops:
        kind, name = op['kind'], op['name']
        if kind == 'input':
            t[name] = x[op['index']] if isinstance(x, (list, tuple)) else x
        elif kind == 'conv':
            src = t[op['src']]
            if op.get('lo'):
                src = torch.cat([src, upsample(t[op['lo']], op['up'])], -1)
            m = net.layers_by_name[name]
            k, b = params[name]
            t[name] = conv3d_same(src, k, b, m.dilation, m.activation, m.padding)
        elif kind == 'dropout':
            t[name] = t[op['src']]
            if dropout_scales and name in dropout_scales:             # training mode: the recorded [B, C] keep / (1 - rate) factors
                sc = dropout_scales[name]
                t[name] = t[name] * sc.reshape(sc.shape[0], 1, 1, 1, sc.shape[1])
        elif kind == 'maxpool':
            t[name] = maxpool_same(t[op['src']], op['pool'])
        elif kind == 'upsample':
            t[name] = upsample(t[op['src']], op['up'])
        elif kind == 'merge':
            t[name] = torch.cat([t[op['skip']], upsample(t[op['lo']], op['up'])], -1)
        elif kind == 'bn':
            m = net.layers_by_name[name]
            gamma, beta = params[name]
            v = t[op['src']]
            if bn_train:                                              # Keras training mode: biased batch statistics
                mean = v.mean(dim=(0, 1, 2, 3))
                var = v.var(dim=(0, 1, 2, 3), unbiased=False)
            else:
                mean, var = m.moving_mean.detach().cpu().double(), m.moving_variance.detach().cpu().double()
            t[name] = (v - mean) / torch.sqrt(var + m.epsilon) * gamma + beta
        elif kind == 'add':
            t[name] = t[op['a']] + t[op['b']]
        elif kind == 'activation':
            t[name] = keras_activation(t[op['src']], op['activation'])
        elif kind == 'multiply':                      # KL.multiply (models.py:412-417)
            t[name] = t[op['a']] * t[op['b']]
        elif kind == 'likelihood':
            k, b = params[name]
            t[name] = conv3d_same(t[op['src']], k, b, 1, None)
        elif kind == 'prediction':
            t[name] = keras_activation(t[op['src']], op['activation'])
        else:
            raise NotImplementedError(kind)
    if return_tensors:
        return {k: t[k] for k in return_tensors}
    return t[net.output_name]
