"""
TEST INFRASTRUCTURE ONLY -- gradient oracle of the hot path (CPU, float64, torch autograd).

The reference has no hand-written backward: TensorFlow differentiates the forward graphs.  This module restates
those forward graphs with differentiable torch ops whose gradient rules coincide with TF's for every op on the
path (floor: no gradient; clip_by_value / clamp: gradient on the closed range; gather: scatter-add; divide_no_nan:
zero where the denominator is zero) and lets torch.autograd produce the gradients the HIP backward kernels
(neurite_amd/csrc/backward.hip) are checked against.

Pinning: TensorFlow is not available here, so the gradients are pinned in tests/test_oracle.py against central
finite differences of the pinned forward oracle (oracle/np_oracle.py: interpn_f64, dice, cce) away from the
kinks, and the forward values of these restatements against the same oracle.

Only tests/ may import this module.
"""

import itertools

import torch

F64 = torch.float64


def interpn(vol, loc, fill_value=None):
    """neurite/tf/utils/utils.py:137-191 (linear) and :206-213 (fill).  vol [*S, C], loc [*O, D]."""
    D = loc.shape[-1]
    S = list(vol.shape[:D])
    C = vol.shape[-1]
    flat = vol.reshape(-1, C)
    loc0 = torch.floor(loc)                                                     # :139
    mx = [float(s - 1) for s in S]
    clipped = [torch.clamp(loc[..., d], 0., mx[d]) for d in range(D)]           # :142
    l0 = [torch.clamp(loc0[..., d], 0., mx[d]) for d in range(D)]               # :143
    l1 = [torch.clamp(l0[d] + 1, 0., mx[d]) for d in range(D)]                  # :146
    locs = [[x.long() for x in l0], [x.long() for x in l1]]                     # :147
    d1 = [l1[d] - clipped[d] for d in range(D)]                                 # :152
    d0 = [1 - d1[d] for d in range(D)]                                          # :153
    w = [d1, d0]
    out = 0
    for c in itertools.product([0, 1], repeat=D):                               # :160-189
        idx = 0
        for d in range(D):
            idx = idx * S[d] + locs[c[d]][d]
        wt = 1
        for d in range(D):
            wt = wt * w[c[d]][d]
        out = out + wt[..., None] * flat[idx]
    if fill_value is not None:                                                  # :206-213
        oob = torch.zeros(loc.shape[:-1], dtype=torch.bool)
        for d in range(D):
            oob = oob | (loc[..., d] < 0) | (loc[..., d] > mx[d])
        keep = (~oob).to(out.dtype)[..., None]
        out = out * keep + (1 - keep) * fill_value
    return out


def transform(vol, shift, fill_value=None):
    """voxelmorph transform(): interpn at identity grid + shift ('ij' indexing)."""
    D = shift.shape[-1]
    grid = torch.stack(torch.meshgrid(*[torch.arange(s, dtype=shift.dtype) for s in shift.shape[:-1]], indexing='ij'), -1)
    return interpn(vol, grid + shift, fill_value)


def resize_locs(S, O, dtype=F64):
    """linspace(0, S-1, O) grid of neurite/tf/utils/utils.py:259-261."""
    lin = [torch.linspace(0., S[d] - 1., O[d], dtype=dtype) for d in range(len(S))]
    return torch.stack(torch.meshgrid(*lin, indexing='ij'), -1)


def soft_dice(y_true, y_pred, laplace_smoothing=0.):
    """neurite/tf/metrics.py:470-482.  [B, ..., L] -> [B, L]."""
    B, L = y_true.shape[0], y_true.shape[-1]
    t = y_true.reshape(B, -1, L)
    p = y_pred.reshape(B, -1, L)
    top = 2 * (t * p).sum(1) + laplace_smoothing
    bottom = (t * t).sum(1) + (p * p).sum(1) + laplace_smoothing
    zero = bottom == 0
    return torch.where(zero, torch.zeros_like(top), top / torch.where(zero, torch.ones_like(bottom), bottom))


def cce_per_voxel(y_true, y_pred, label_weights=None, from_logits=False, label_smoothing=0.):
    """neurite/tf/metrics.py:648-650 + tf.keras categorical_crossentropy (axis -1)."""
    t = y_true
    if label_weights is not None:
        t = t * label_weights
    if label_smoothing:
        t = t * (1 - label_smoothing) + label_smoothing / y_pred.shape[-1]
    if from_logits:
        return -(t * torch.log_softmax(y_pred, -1)).sum(-1)
    q = y_pred / y_pred.sum(-1, keepdim=True)
    q = torch.clamp(q, 1e-7, 1 - 1e-7)
    return -(t * torch.log(q)).sum(-1)


def lc3d(x, kernel, bias=None, kernel_size=(3, 3, 3), strides=(1, 1, 1), activation=None):
    """LocallyConnected3D implementation 1 (neurite/tf/layers.py:1126-1197): x [B,R,C,Z,Cin], kernel [O, F, Cout] with the
    patch flattened in (kr, kc, kz, cin) order, positions row-major; bias [or, oc, oz, Cout]."""
    kr, kc, kz = kernel_size
    p = x.unfold(1, kr, strides[0]).unfold(2, kc, strides[1]).unfold(3, kz, strides[2])     # [B, or, oc, oz, Cin, kr, kc, kz]
    B, orr, occ, ozz = p.shape[:4]
    p = p.permute(0, 1, 2, 3, 5, 6, 7, 4).reshape(B, orr * occ * ozz, -1)                      # f = (kr, kc, kz, cin)
    y = torch.einsum('bof,ofc->boc', p, kernel).reshape(B, orr, occ, ozz, -1)
    if bias is not None:
        y = y + bias
    if activation == 'elu':
        y = torch.where(y > 0, y, torch.exp(torch.clamp(y, max=0.0)) - 1)
    elif activation == 'relu':
        y = torch.relu(y)
    return y


def _act(y, activation):
    if activation == 'elu':
        return torch.where(y > 0, y, torch.exp(torch.clamp(y, max=0.0)) - 1)
    if activation == 'relu':
        return torch.relu(y)
    return y


def lc3d_connection_mask(ins, ksize, strides, padding, outs):
    """keras conv_utils.conv_kernel_mask (what LocallyConnected3D.get_locallyconnected_mask wraps, neurite/tf/layers.py:
    1199-1257): bool [*ins, *outs], True where the output position reads the input position; the window of position p along
    axis d is [c - k//2, c + k - k//2) clipped to the volume, c = p*s (+ k//2 for 'valid') (conv_connected_inputs :1436-1484)."""
    import itertools
    import numpy as np
    mask = np.zeros(tuple(ins) + tuple(outs), bool)
    for pos in itertools.product(*[range(n) for n in outs]):
        rng = []
        for d in range(3):
            left = int(ksize[d] / 2)
            right = ksize[d] - left
            c = pos[d] * strides[d] + (left if padding == 'valid' else 0)
            rng.append(range(max(0, c - left), min(ins[d], c + right)))
        for ip in itertools.product(*rng):
            mask[ip + pos] = True
    return mask


def lc3d_dense_masked(x, kernel, mask, bias, activation=None, data_format='channels_last'):
    """LocallyConnected3D implementation 2 as the reference computes it (neurite/tf/layers.py:1260-1308): the flattened input
    times the dense kernel multiplied by the 0/1 connection mask.  x [B, *ins, Cin] (or [B, Cin, *ins]); kernel
    [*ins, Cin, *outs, Cout] (or [Cin, *ins, Cout, *outs]); mask bool [*ins, *outs]; bias [*outs, Cout]."""
    cf = data_format == 'channels_first'
    m = torch.from_numpy(mask).to(kernel.dtype)
    m = m[None, :, :, :, None, :, :, :] if cf else m[:, :, :, None, :, :, :, None]            # :1245-1251
    k = (m * kernel)
    k2 = k.reshape(int(torch.tensor(k.shape[:4]).prod()), -1)                                  # make_2d, split at ndim // 2
    y = (x.reshape(x.shape[0], -1) @ k2).reshape((x.shape[0],) + tuple(kernel.shape[4:]))
    if bias is not None:
        b = bias.reshape((bias.shape[-1],) + tuple(bias.shape[:-1])) if cf else bias           # K.bias_add: a reshape
        y = y + b
    return _act(y, activation)


def lc3d_sparse(x, values, kernel_idxs, dense_shape, out_shape, bias, activation=None, data_format='channels_last'):
    """LocallyConnected3D implementation 3 (neurite/tf/layers.py:1311-1343): SparseTensor(kernel_idxs, values, dense_shape) @
    x_flat^T, transposed and reshaped to out_shape (without batch)."""
    cf = data_format == 'channels_first'
    idx = torch.as_tensor(kernel_idxs, dtype=torch.long)
    dense = torch.zeros(tuple(int(v) for v in dense_shape), dtype=values.dtype).index_put((idx[:, 0], idx[:, 1]), values)
    y = (dense @ x.reshape(x.shape[0], -1).T).T.reshape((x.shape[0],) + tuple(out_shape))
    if bias is not None:
        b = bias.reshape((bias.shape[-1],) + tuple(bias.shape[:-1])) if cf else bias
        y = y + b
    return _act(y, activation)


def mi_channelwise(x, y, cx, cy, alpha, min_clip=float('-inf'), max_clip=float('inf'), eps=1e-7):
    """MutualInformation.channelwise (neurite/tf/metrics.py:188-282) with GIVEN bin centres cx, cy [nb] (held constant, as the
    HIP backward does): x, y [bs, ..., C] -> [bs, C]."""
    bs, C = x.shape[0], x.shape[-1]
    xf = torch.clamp(x.reshape(bs, -1, C), min_clip, max_clip)
    yf = torch.clamp(y.reshape(bs, -1, C), min_clip, max_clip)
    wx = torch.exp(-alpha * (xf[..., None] - cx) ** 2)             # [bs, V, C, nb]
    wy = torch.exp(-alpha * (yf[..., None] - cy) ** 2)
    joint = torch.einsum('bvci,bvcj->bcij', wx, wy)
    pxy = joint / (joint.sum((2, 3), keepdim=True) + eps)
    px = wx.sum(1)
    px = px / (px.sum(-1, keepdim=True) + eps)
    py = wy.sum(1)
    py = py / (py.sum(-1, keepdim=True) + eps)
    pxpy = px[..., :, None] * py[..., None, :] + eps
    return (pxy * torch.log(pxy / pxpy + eps)).sum((2, 3))


def mi_maps(x, y, eps=1e-7):
    """MutualInformation.maps (neurite/tf/metrics.py:228-282): x, y [bs, ..., B] non-negative maps -> [bs]."""
    bs, B = x.shape[0], x.shape[-1]
    xf = x.reshape(bs, -1, B)
    yf = y.reshape(bs, -1, B)
    joint = torch.einsum('bvi,bvj->bij', xf, yf)                   # :256-259
    pxy = joint / (joint.sum((1, 2), keepdim=True) + eps)          # :262
    px = xf.sum(1)
    px = px / (px.sum(1, keepdim=True) + eps)                      # :265-266
    py = yf.sum(1)
    py = py / (py.sum(1, keepdim=True) + eps)                      # :267-268
    pxpy = px[:, :, None] * py[:, None, :] + eps                   # :271-274
    return (pxy * torch.log(pxy / pxpy + eps)).sum((1, 2))         # :277-281
