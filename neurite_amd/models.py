"""
neurite_amd.models -- neurite's U-Net builders on MI355X.

unet      neurite/tf/models.py:88-246
conv_enc  neurite/tf/models.py:1309-1442
conv_dec  neurite/tf/models.py:1445-1617
conv_block: alias of conv_enc (the name BASELINE.json uses for the conv stacks; not in the reference tree)

Same arguments, defaults, layer names ('{prefix}_conv_downarm_{level}_{conv}', '{prefix}_maxpool_{level}',
'{prefix}_up_{n}', '{prefix}_merge_{n}', '{prefix}_conv_uparm_{n}_{conv}', '{prefix}_likelihood',
'{prefix}_prediction' ...) and graph as the Keras builders; the returned object is a torch.nn.Module whose
parameters keep the Keras layouts (Conv kernel [k.., Cin, Cout], bias [Cout]) so trained Keras weights map
one to one.  Forward runs on the HIP kernels of csrc/conv.hip: Conv3D = MFMA implicit GEMM (fp32), with
UpSampling3D + concatenate fused into the following convolution's loader and the final 1x1 conv fused
with the channel softmax.  1-D / 2-D nets are lifted to 3-D with unit leading dimensions.
Inference only: Dropout is the identity, BatchNormalization uses its moving statistics.
"""

import functools
import inspect
import json
import math
import sys
import warnings

import numpy as np
import torch
from torch import nn

from . import _lib
from . import utils

__all__ = ['unet', 'conv_enc', 'conv_dec', 'conv_block', 'ConvNet', 'labels_to_image', 'labels_to_image_new', 'SynthStrip', 'add_prior', 'dilation_net', 'load', 'load_config']

# element-wise activations, codes of include/neurite_amd.h (nrt_activation); definitions follow tf.keras.activations -- the reference
# hands the string straight to Keras (neurite/tf/models.py:1346, 1429, 1507, 1588)
_ACTS = {None: 0, 'linear': 0, 'elu': 1, 'relu': 2, 'sigmoid': 3, 'tanh': 4, 'softplus': 5, 'softsign': 6, 'selu': 7,
         'exponential': 8, 'hard_sigmoid': 9, 'leaky_relu': 10}
_EW_ACTS = _ACTS                              # stand-alone Activation layers (nrt_add_act_affine_f32) take the same set
_ACT_MUL_B = 0x100
_ACT_LAST_FUSED = 2          # elu, relu are fused into the conv / LocallyConnected3D epilogues; the others run as an element-wise pass


def _act_code(activation):
    if activation not in _ACTS:
        raise NotImplementedError('activation %r is not implemented by the HIP path (%s are; softmax as a layer activation runs as '
                                  'its own kernel)' % (activation, ', '.join(sorted(str(k) for k in _ACTS))))
    return _ACTS[activation]


def _triple(v, ndims, what):
    if isinstance(v, (int, np.integer)):
        v = (int(v),) * ndims
    v = tuple(int(x) for x in v)
    if len(v) != ndims:
        raise ValueError('%s must have %d entries, got %r' % (what, ndims, v))
    return (1,) * (3 - ndims) + v


def _lift(x, ndims):
    """[B, *S, C] with len(S) = ndims -> [B, 1.., *S, C] (3 spatial dims)."""
    for _ in range(3 - ndims):
        x = x.unsqueeze(1)
    return x.contiguous()


def _unlift(x, ndims):
    for _ in range(3 - ndims):
        x = x.squeeze(1)
    return x


def _jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, tuple):
        return [_jsonable(x) for x in v]
    if isinstance(v, list):
        return [_jsonable(x) for x in v]
    return v


def _store_config(func):
    """
    The builders' counterpart of modelio.store_config_args (neurite/tf/modelio.py:8-45): every argument the network was
    built with (defaults, positionals, keywords) is kept on the returned model as `model.config`, so that
    `model.save(path)` / `models.load(path)` rebuild the same architecture without the caller restating it
    (LoadableModel, modelio.py:78-143).  Graph-valued arguments (input_model, convL, src ...) are not recorded.
    """
    sig = inspect.signature(func)

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        net = func(*args, **kwargs)
        bound = sig.bind(*args, **kwargs)
        bound.apply_defaults()
        params = {}
        loadable = True
        for k, v in bound.arguments.items():
            if k in ('convL', 'src', 'src_input', 'input_model'):
                loadable = loadable and v is None
                continue
            params[k] = _jsonable(v)
        params['metadata'] = {}
        net.config = {'builder': func.__name__, 'params': params, 'loadable': loadable}
        return net
    return wrapper


def _h5py():
    """h5py where the interpreter has it, else the package's own reader / writer of the same calls (neurite_amd/h5lite.py:
    the HDF5 subset Keras files use, checked against files the HDF5 library wrote -- tests/test_h5lite.py)"""
    try:
        import h5py
        return h5py
    except ImportError:
        from . import h5lite
        return h5lite


def _h5_attr(group, name):
    """Keras splits attributes that would not fit an object header (64512 bytes) into name0, name1, ... (hdf5_format.py:
    save_attributes_to_hdf5_group / load_attributes_from_hdf5_group): read either form"""
    if name in group.attrs:
        return list(group.attrs[name])
    out, k = [], 0
    while '%s%d' % (name, k) in group.attrs:
        out.extend(group.attrs['%s%d' % (name, k)])
        k += 1
    if not k:
        raise KeyError('no attribute %r in %s' % (name, getattr(group, 'name', group)))
    return out


def _h5_set_attr(group, name, data):
    """a list of byte strings as Keras stores it: one attribute, or name0, name1, ... when it would exceed an object header"""
    arr = np.asarray(data)
    chunks, n = [arr], 1
    while any(c.nbytes > 64512 for c in chunks):
        n += 1
        chunks = np.array_split(arr, n)
    if n == 1:
        group.attrs[name] = data
    else:
        for k, c in enumerate(chunks):
            group.attrs['%s%d' % (name, k)] = c


def _is_h5(path):
    return str(path).lower().endswith(('.h5', '.hdf5', '.keras.h5'))


def _h5_str(v):
    return v.decode('utf-8') if isinstance(v, bytes) else str(v)


class _Conv(nn.Module):
    """Keras Conv{N}D (channels-last, stride 1): kernel [k1..kN, Cin, Cout], glorot_uniform; bias zeros."""

    def __init__(self, name, cin, cout, ksize3, dilation=1, padding='same', activation=None):
        super().__init__()
        self.layer_name = name
        self.cin, self.cout = int(cin), int(cout)
        self.ksize3 = tuple(ksize3)
        self.dilation = int(dilation)
        if padding not in ('same', 'valid'):
            raise ValueError('padding must be same or valid')
        self.padding = padding
        self.activation = activation
        # the channel softmax is not element-wise: a linear epilogue, then the softmax kernel
        self.post_softmax = activation == 'softmax'
        self.act = 0 if self.post_softmax else _act_code(activation)
        k = self.ksize3
        fan = k[0] * k[1] * k[2]
        limit = math.sqrt(6.0 / (fan * self.cin + fan * self.cout))
        self.kernel = nn.Parameter((torch.rand(*k, self.cin, self.cout) * 2 - 1) * limit)
        self.bias = nn.Parameter(torch.zeros(self.cout))
        self._packed = None
        self._packed_version = None
        self._packed_up2 = None                      # folded weights of the decoder form (see _run), keyed like _packed + c0
        self._packed_up2_version = None
        self.fold_backward = True                    # decoder form: differentiate the up-sampled channels on the low-resolution grid

    def invalidate_packed(self):
        """forget the MFMA-packed copy of the kernel.  The cache key below sees in-place writes through the Parameter
        (`_version`), a new storage and a device move; writes through `.data` (EMA / weight averaging: `p.data.mul_()`)
        bump no version, so the copy is also dropped on every `train()` / `eval()` switch, `set_weights`, `load_weights` and
        `load_state_dict`, and it is never used while the model is in training mode."""
        self._packed = None
        self._packed_version = None
        self._packed_up2 = None
        self._packed_up2_version = None

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def _packed_weights(self):
        ver = (self.kernel._version, self.kernel.data_ptr(), self.kernel.device)
        if self.training:
            self._packed = None                       # training loops may write through .data between steps
        if self._packed is None or self._packed_version != ver:
            lib = _lib.lib()
            dev = self.kernel.device
            n = lib.nrt_conv3d_packed_weight_floats(_lib.ints(self.ksize3), self.cin, self.cout)
            packed = torch.empty(int(n), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nrt_conv3d_pack_weights_f32(_lib.ptr(self.kernel.detach().contiguous()), _lib.ints(self.ksize3),
                                                     self.cin, self.cout, _lib.ptr(packed), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_conv3d_pack_weights_f32')
            self._packed, self._packed_version = packed, ver
        return self._packed

    def _packed_weights_up2(self, c0):
        """the kernel folded for nrt_conv3d_up2_f32: skip channels [0, c0) keep their 27 taps, the up-sampled channels
        get 8 parity classes x 8 taps of pre-summed weights"""
        ver = (self.kernel._version, self.kernel.data_ptr(), self.kernel.device, int(c0))
        if self.training:
            self._packed_up2 = None
        if self._packed_up2 is None or self._packed_up2_version != ver:
            lib = _lib.lib()
            dev = self.kernel.device
            n = lib.nrt_conv3d_up2_packed_weight_floats(int(c0), self.cin - int(c0), self.cout)
            packed = torch.empty(int(n), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nrt_conv3d_up2_pack_weights_f32(_lib.ptr(self.kernel.detach().contiguous()), int(c0), self.cin - int(c0),
                                                         self.cout, _lib.ptr(packed), _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_conv3d_up2_pack_weights_f32')
            self._packed_up2, self._packed_up2_version = packed, ver
        return self._packed_up2

    def forward(self, x, lo=None, up=None, variant=0):
        """x [B, X, Y, Z, c0]; optional lo [B, X/up, Y/up, Z/up, c1] is nearest-upsampled and concatenated after x."""
        if torch.is_grad_enabled() and (x.requires_grad or (lo is not None and lo.requires_grad)
                                        or self.kernel.requires_grad or self.bias.requires_grad):
            y = _ConvFn.apply(x, lo, self.kernel, self.bias, self, up, variant, False)
        else:
            y = self._run(x, lo, up, variant)
        if self.post_softmax:
            return _softmax_with_grad(y) if (torch.is_grad_enabled() and y.requires_grad) else _softmax(y)
        return y

    def pool_foldable(self, x, variant):
        """a 3x3x3 'same' encoder convolution whose 2x2x2 max-pooling can come out of the same kernel: the single-channel first layer
        (csrc/conv.hip: conv3d_c1_mfma<NT, true>) and, round 6, layers over 16 k input channels with 32 .. 64 filters in the persistent
        schedule (csrc/conv_p27.h: conv3d_p27_mfma<NT, true>)"""
        if self.ksize3 != (3, 3, 3) or self.dilation != 1 or x.shape[-1] != self.cin:
            return False
        if self.padding != 'same' or self.act > _ACT_LAST_FUSED or self.post_softmax or x.dtype != torch.float32:
            return False
        if self.cin == 1:
            return variant in (0, 1) and _lib.lib().nrt_conv3d_c1_pool_supported(_lib.ints(list(x.shape[1:4])), self.cout) == 1
        return variant in (0, 5) and _lib.lib().nrt_conv3d_pool_supported(self.cin, self.cout, _lib.ints(list(x.shape[1:4])), int(x.shape[0])) == 1

    def run_with_pool(self, x):
        """(act(conv(x)), MaxPooling3D(2) of it) from one kernel (nrt_conv3d_c1_pool_f32 / nrt_conv3d_pool_f32).  Inference only."""
        lib = _lib.lib()
        dev = _lib.require_device(x, self.kernel)
        x = x.contiguous()
        B, S = x.shape[0], list(x.shape[1:4])
        out = torch.empty([B] + S + [self.cout], dtype=torch.float32, device=dev)
        pooled = torch.empty([B] + [v // 2 for v in S] + [self.cout], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if self.cin == 1:
                rc = lib.nrt_conv3d_c1_pool_f32(_lib.ptr(x), _lib.ptr(self.kernel.detach().contiguous()), _lib.ptr(self.bias.detach()),
                                                _lib.ptr(out), _lib.ptr(pooled), B, _lib.ints(S), self.cout, self.act, _lib.stream_ptr(dev))
            else:
                rc = lib.nrt_conv3d_pool_f32(_lib.ptr(x), self.cin, _lib.ptr(self._packed_weights()), _lib.ptr(self.bias.detach()),
                                             _lib.ptr(out), _lib.ptr(pooled), B, _lib.ints(S), self.cout, self.act, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv3d_c1_pool_f32' if self.cin == 1 else 'nrt_conv3d_pool_f32')
        return out, pooled

    def _packed_head(self, head_kernel, labels, dev):
        """the head's kernel in matrix-core fragment order, repacked when the parameter changes (version counter)"""
        key = (head_kernel.data_ptr(), head_kernel._version, str(dev))
        if getattr(self, '_head_pack_key', None) != key:
            lib = _lib.lib()
            src = head_kernel.detach().reshape(-1, labels).contiguous()
            packed = torch.empty(16 * labels, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.nrt_conv3d_up2_head_pack_f32(_lib.ptr(src), labels, _lib.ptr(packed), _lib.stream_ptr(dev)),
                           'nrt_conv3d_up2_head_pack_f32')
            self._head_pack, self._head_pack_key = packed, key
        return self._head_pack

    def run_with_head(self, x, lo, head_kernel, head_bias):
        """softmax(act(this decoder convolution) @ head_kernel + head_bias) in one kernel (nrt_conv3d_up2_head_f32); the caller has
        checked ConvNet._head_foldable.  Inference only."""
        lib = _lib.lib()
        dev = _lib.require_device(x, lo, self.kernel)
        x, lo = x.contiguous(), lo.contiguous()
        c0, c1 = x.shape[-1], lo.shape[-1]
        B, S = x.shape[0], list(x.shape[1:4])
        labels = head_kernel.shape[-1]
        out = torch.empty([B] + S + [labels], dtype=torch.float32, device=dev)
        hw = self._packed_head(head_kernel, labels, dev)
        with torch.cuda.device(dev):
            rc = lib.nrt_conv3d_up2_head_f32(_lib.ptr(x), c0, _lib.ptr(lo), c1, _lib.ptr(self._packed_weights_up2(c0)),
                                             _lib.ptr(self.bias.detach()), _lib.ptr(hw), _lib.ptr(head_bias.detach()), labels,
                                             _lib.ptr(out), B, _lib.ints(S), self.cout, self.act, _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv3d_up2_head_f32')
        return out

    def _run(self, x, lo=None, up=None, variant=0):
        lib = _lib.lib()
        dev = _lib.require_device(x, lo, self.kernel)
        if x.dtype != torch.float32:
            raise NotImplementedError('the HIP conv path is float32')
        x = x.contiguous()
        c0 = x.shape[-1]
        c1 = 0 if lo is None else lo.shape[-1]
        if c0 + c1 != self.cin:
            raise ValueError('%s expects %d input channels, got %d' % (self.layer_name, self.cin, c0 + c1))
        B, S = x.shape[0], list(x.shape[1:4])
        if lo is not None:
            lo = lo.contiguous()
            if [S[d] // up[d] for d in range(3)] != list(lo.shape[1:4]) or any(S[d] % up[d] for d in range(3)):
                raise ValueError('%s: skip %s and up-sampled %s x %s shapes do not match'
                                 % (self.layer_name, S, list(lo.shape[1:4]), up))
        if self.padding == 'same':
            O = S
        else:
            O = [S[d] - (self.ksize3[d] - 1) * self.dilation for d in range(3)]
        out = torch.empty([B] + O + [self.cout], dtype=torch.float32, device=dev)
        w = self.kernel.detach().contiguous()
        # decoder form (UpSampling3D(2) + concatenate + 3x3x3 SAME): the up-sampled half runs as 8 folded taps on the
        # low-resolution grid (variant 0 = auto, 4 = required; 2 keeps the plain implicit GEMM over all 27 taps)
        folded = (variant in (0, 4) and lo is not None and tuple(up) == (2, 2, 2) and self.ksize3 == (3, 3, 3)
                  and self.dilation == 1 and self.padding == 'same'
                  and lib.nrt_conv3d_up2_supported(c0, c1, self.cout, _lib.ints(S)) == 1
                  # the folded kernel forms 32-bit output offsets over the whole batch (conv_up2.h launch_up2); the shape query
                  # does not see the batch, and nrt_conv3d_f32 has no such limit
                  and B * S[0] * S[1] * S[2] * self.cout < (1 << 30))
        if variant == 4 and not folded:
            raise NotImplementedError('%s: shapes outside the folded decoder kernel' % self.layer_name)
        if folded:
            with torch.cuda.device(dev):
                rc = lib.nrt_conv3d_up2_f32(_lib.ptr(x), c0, _lib.ptr(lo), c1, _lib.ptr(self._packed_weights_up2(c0)),
                                            _lib.ptr(self.bias.detach()), _lib.ptr(out), B, _lib.ints(S), self.cout,
                                            self.act if self.act <= _ACT_LAST_FUSED else 0, _lib.stream_ptr(dev))
            if not (rc == _lib.NRT_ERR_UNSUPPORTED and variant == 0):       # auto: anything the folded form declines takes the plain GEMM
                _lib.check(rc, 'nrt_conv3d_up2_f32')
                if self.act > _ACT_LAST_FUSED:
                    out = _elementwise(out, act=self.act)
                return out
        with torch.cuda.device(dev):
            rc = lib.nrt_conv3d_f32(_lib.ptr(x), c0, _lib.ptr(lo), c1, _lib.ints(up) if lo is not None else None,
                                    _lib.ptr(w), _lib.ptr(self._packed_weights()), _lib.ptr(self.bias.detach()),
                                    _lib.ptr(out), B, _lib.ints(S), _lib.ints(self.ksize3), self.cout, self.dilation,
                                    int(self.padding == 'same'), self.act if self.act <= _ACT_LAST_FUSED else 0, int(variant),
                                    _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv3d_f32')
        if self.act > _ACT_LAST_FUSED:            # activations beyond elu / relu: an element-wise pass over the layer output
            out = _elementwise(out, act=self.act)
        return out


class _BatchNorm(nn.Module):
    """Keras BatchNormalization (axis -1): gamma, beta, moving_mean, moving_variance, epsilon = 1e-3, momentum = 0.99.
    Inference uses the moving statistics; in training mode (`_BatchNormFn`) the batch statistics, and the moving ones are
    updated as moving * momentum + batch * (1 - momentum) with the biased batch variance."""

    def __init__(self, name, channels, epsilon=1e-3, momentum=0.99):
        super().__init__()
        self.layer_name = name
        self.epsilon = epsilon
        self.momentum = momentum
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))
        self.register_buffer('moving_mean', torch.zeros(channels))
        self.register_buffer('moving_variance', torch.ones(channels))


def _elementwise(a, b=None, scale=None, shift=None, act=0, mul=False):
    """act(a + b) * scale + shift, or with mul act(a) * b"""
    if mul:
        act = int(act) | _ACT_MUL_B
    lib = _lib.lib()
    dev = _lib.require_device(a, b)
    a = a.contiguous()
    b = None if b is None else b.contiguous()
    y = torch.empty_like(a)
    with torch.cuda.device(dev):
        rc = lib.nrt_add_act_affine_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y),
                                        a.numel(), a.shape[-1], int(act), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_add_act_affine_f32')
    return y


def _maxpool(x, pool3, padding):
    lib = _lib.lib()
    dev = _lib.require_device(x)
    x = x.contiguous()
    B, S, C = x.shape[0], list(x.shape[1:4]), x.shape[-1]
    same = padding == 'same'
    O = [(S[d] + pool3[d] - 1) // pool3[d] if same else S[d] // pool3[d] for d in range(3)]
    y = torch.empty([B] + O + [C], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_maxpool3d_f32(_lib.ptr(x), _lib.ptr(y), B, _lib.ints(S), C, _lib.ints(pool3), int(same),
                                   _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_maxpool3d_f32')
    return y


def _upsample_concat(skip, lo, up3):
    lib = _lib.lib()
    dev = _lib.require_device(skip, lo)
    lo = lo.contiguous()
    B, S1, c1 = lo.shape[0], list(lo.shape[1:4]), lo.shape[-1]
    S = [S1[d] * up3[d] for d in range(3)]
    c0 = 0 if skip is None else skip.shape[-1]
    if skip is not None:
        skip = skip.contiguous()
        if list(skip.shape[1:4]) != S:
            raise ValueError('concatenate: shapes %s and %s do not match' % (list(skip.shape[1:4]), S))
    y = torch.empty([B] + S + [c0 + c1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_upsample_concat_f32(_lib.ptr(skip), c0, _lib.ptr(lo), c1, _lib.ptr(y), B, _lib.ints(S),
                                         _lib.ints(up3), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_upsample_concat_f32')
    return y


def _conv1x1_softmax(x, kernel, bias, softmax, act):
    lib = _lib.lib()
    dev = _lib.require_device(x, kernel)
    x = x.contiguous()
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    y = torch.empty(list(x.shape[:-1]) + [cout], dtype=torch.float32, device=dev)
    w = kernel.detach().reshape(cin, cout).contiguous()
    fused = int(act) if int(act) <= _ACT_LAST_FUSED else 0       # elu / relu ride in the epilogue, the others as a pass of their own
    if softmax and fused != int(act):
        raise NotImplementedError('a 1x1x1 convolution with a non-fused activation followed by the channel softmax')
    with torch.cuda.device(dev):
        rc = lib.nrt_conv1x1_softmax_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias.detach()), _lib.ptr(y),
                                         x.numel() // cin, cin, cout, int(softmax), fused, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_conv1x1_softmax_f32')
    return y if fused == int(act) else _elementwise(y, act=int(act))


def _softmax(x):
    lib = _lib.lib()
    dev = _lib.require_device(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = lib.nrt_softmax_lastdim_f32(_lib.ptr(x), _lib.ptr(y), x.numel() // x.shape[-1], x.shape[-1],
                                         _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_softmax_lastdim_f32')
    return y


# --------------------------------------------------------------------------------------
# autograd: the backward kernels of csrc/conv_bwd.hip behind torch.autograd.Function
# --------------------------------------------------------------------------------------

def _act_bwd(g, y, act):
    if act == 0:
        return g.contiguous()
    lib = _lib.lib()
    dev = g.device
    g = g.contiguous()
    d = torch.empty_like(g)
    with torch.cuda.device(dev):
        rc = lib.nrt_act_bwd_f32(_lib.ptr(g), _lib.ptr(y), int(act), _lib.ptr(d), g.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_act_bwd_f32')
    return d


def _upsample_sum(g, c_off, c, lo_shape, up3):
    """grad of nearest up-sampling: channels [c_off, c_off + c) of g [B, S, Cg] summed over up^3 blocks -> [B, S/up, c]."""
    lib = _lib.lib()
    dev = g.device
    g = g.contiguous()
    B = g.shape[0]
    d = torch.empty([B] + list(lo_shape) + [c], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_upsample_sum_f32(_lib.ptr(g), g.shape[-1], int(c_off), _lib.ptr(d), int(c), B, _lib.ints(lo_shape),
                                      _lib.ints(up3), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_upsample_sum_f32')
    return d


def _conv_dgrad(dpre, wpart, ksize3, dilation):
    """grad wrt the conv input: conv(dpre, weights flipped in space and transposed in the channel axes) on the forward kernel."""
    lib = _lib.lib()
    dev = dpre.device
    cout, cpart = wpart.shape[-1], wpart.shape[-2]
    wt = wpart.flip(0, 1, 2).transpose(3, 4).contiguous()             # [k, k, k, cout, cpart]; a few KB of glue
    if tuple(ksize3) == (1, 1, 1) and cpart % 4 == 0 and cpart <= 64:
        # 1x1x1 (the likelihood layer): a per-voxel matrix product on the streaming kernel, not on the halo-tile convolution
        dpre = dpre.contiguous()
        out = torch.empty(list(dpre.shape[:-1]) + [cpart], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.nrt_conv1x1_softmax_f32(_lib.ptr(dpre), _lib.ptr(wt), None, _lib.ptr(out), dpre.numel() // cout, cout, cpart, 0, 0,
                                             _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv1x1_softmax_f32 (dgrad)')
        return out
    n = lib.nrt_conv3d_packed_weight_floats(_lib.ints(ksize3), cout, cpart)
    packed = torch.empty(int(n), dtype=torch.float32, device=dev)
    B, S = dpre.shape[0], list(dpre.shape[1:4])
    out = torch.empty([B] + S + [cpart], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_conv3d_pack_weights_f32(_lib.ptr(wt), _lib.ints(ksize3), cout, cpart, _lib.ptr(packed), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv3d_pack_weights_f32')
        rc = lib.nrt_conv3d_f32(_lib.ptr(dpre), cout, None, 0, None, _lib.ptr(wt), _lib.ptr(packed), None, _lib.ptr(out), B,
                                _lib.ints(S), _lib.ints(ksize3), cpart, int(dilation), 1, 0, 0, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_conv3d_f32 (dgrad)')
    return out


# ---- folded backward of a decoder convolution (UpSampling3D(2) + concatenate + Conv3D 3x3x3 'same') -------------------------
# Per axis, output parity p and low-resolution tap t cover the 3x3x3 taps d in S(p, t): S(0,0) = {0}, S(0,1) = {1,2},
# S(1,0) = {0,1}, S(1,1) = {2} (csrc/conv_up2.h).  _FOLD_A[p][t][d] is that membership; _FOLD_B[p][e][d] is the same
# seen from the low-resolution grid in the backward direction (3x3x3 tap e = 2 - p - t of the space-to-depth gradient).
_FOLD_A = ((( 1, 0, 0), (0, 1, 1)), ((1, 1, 0), (0, 0, 1)))
_FOLD_B = (((0, 0, 0), (0, 1, 1), (1, 0, 0)), ((0, 0, 1), (1, 1, 0), (0, 0, 0)))


_FOLD_ON_DEVICE = {}


def _fold_table(table, like):
    """_FOLD_A / _FOLD_B as a tensor on `like`'s device, uploaded once (a host->device copy per step would also keep the training
    step out of a hipGraph: copies from pageable memory are not capturable)"""
    key = (table is _FOLD_A, like.device, like.dtype)
    m = _FOLD_ON_DEVICE.get(key)
    if m is None:
        m = _FOLD_ON_DEVICE[key] = torch.tensor(table, dtype=like.dtype, device=like.device)
    return m


def _fold_dgrad_weights(k_lo):
    """kernel rows of the up-sampled channels [3,3,3,c1,cout] -> the 3x3x3 kernel [3,3,3, 8 * cout, c1] that maps the
    space-to-depth gradient (parity group P = (px*2 + py)*2 + pz, channels P*cout + co) to the gradient of the
    low-resolution tensor; only 2 x 2 x 2 taps per parity group are non-zero."""
    Bm = _fold_table(_FOLD_B, k_lo)
    c1, cout = k_lo.shape[3], k_lo.shape[4]
    return torch.einsum('xea,yfb,zgc,abcio->efgxyzoi', Bm, Bm, Bm, k_lo).reshape(3, 3, 3, 8 * cout, c1).contiguous()


def _unfold_wgrad(dwf):
    """folded weight gradient [8 parity groups, 8 taps, c1, cout] -> [3,3,3,c1,cout]"""
    Am = _fold_table(_FOLD_A, dwf)
    c1, cout = dwf.shape[2], dwf.shape[3]
    return torch.einsum('xta,yub,zvc,xyztuvio->abcio', Am, Am, Am, dwf.reshape(2, 2, 2, 2, 2, 2, c1, cout))


def _fold_backward_ok(mod, x, lo, up):
    return (lo is not None and tuple(up) == (2, 2, 2) and mod.ksize3 == (3, 3, 3) and mod.dilation == 1 and mod.padding == 'same'
            and x.shape[-1] % 16 == 0 and lo.shape[-1] % 16 == 0 and mod.cout % 16 == 0 and mod.cout <= 32 and lo.shape[-1] <= 64
            and all(s % 2 == 0 for s in x.shape[1:4]))


def _space_to_depth2(dpre):
    lib = _lib.lib()
    dev = dpre.device
    dpre = dpre.contiguous()
    B, S, C = dpre.shape[0], list(dpre.shape[1:4]), dpre.shape[-1]
    y = torch.empty([B] + [s // 2 for s in S] + [8 * C], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_space_to_depth2_f32(_lib.ptr(dpre), _lib.ptr(y), B, _lib.ints(S), C, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_space_to_depth2_f32')
    return y


def _conv_dgrad_lo_folded(s2d, k_lo):
    """gradient of the low-resolution input: 2 x 2 x 2 taps per parity group of the space-to-depth gradient (replaces the 27-tap
    dgrad over the up-sampled channels at full resolution + the 2^3 block sum)"""
    lib = _lib.lib()
    dev = s2d.device
    c1, cout = k_lo.shape[3], k_lo.shape[4]
    wf = _fold_dgrad_weights(k_lo)
    k3 = (3, 3, 3)
    n = lib.nrt_conv3d_packed_weight_floats(_lib.ints(k3), 8 * cout, c1)
    packed = torch.empty(int(n), dtype=torch.float32, device=dev)
    B, S1 = s2d.shape[0], list(s2d.shape[1:4])
    out = torch.empty([B] + S1 + [c1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_conv3d_pack_weights_f32(_lib.ptr(wf), _lib.ints(k3), 8 * cout, c1, _lib.ptr(packed), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv3d_pack_weights_f32')
        rc = lib.nrt_conv3d_s2d_taps_f32(_lib.ptr(s2d), cout, _lib.ptr(packed), _lib.ptr(out), B, _lib.ints(S1), c1, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_conv3d_s2d_taps_f32')
    return out


def _conv_wgrad_lo_folded(lo, s2d, cout):
    lib = _lib.lib()
    dev = lo.device
    lo = lo.contiguous()
    c1 = lo.shape[-1]
    dwf = torch.zeros(8, 8, c1, cout, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_conv3d_wgrad_s2d_f32(_lib.ptr(lo), _lib.ptr(s2d), _lib.ptr(dwf), lo.shape[0], _lib.ints(list(lo.shape[1:4])), c1,
                                          cout, _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_conv3d_wgrad_s2d_f32')
    return _unfold_wgrad(dwf)


class _ConvFn(torch.autograd.Function):
    """Conv3D (+ fused up-sample/concat loader, bias, activation) with dgrad / wgrad."""

    @staticmethod
    def forward(ctx, x, lo, kernel, bias, mod, up, variant, one_by_one):
        with torch.no_grad():
            if one_by_one:
                out = _conv1x1_softmax(x, kernel, bias, False, mod.act)
            else:
                out = mod._run(x, lo, up, variant)
        ctx.mod, ctx.up = mod, up
        ctx.save_for_backward(x, lo, kernel, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, lo, kernel, out = ctx.saved_tensors
        mod, up = ctx.mod, ctx.up
        lib = _lib.lib()
        dev = g.device
        dpre = _act_bwd(g, out, mod.act)
        if mod.padding != 'same':
            # A 'valid' convolution is the 'same' convolution restricted to the outputs whose window lies inside the volume
            # (output v = same-output v + pb, pb = ((k - 1) * dilation) // 2 as TF pads).  Its backward is therefore the
            # 'same' backward of the output gradient embedded in zeros at those positions (nrt_pad3d): the positions that
            # were never computed contribute nothing, and no window reads outside the input.
            S = list(x.shape[1:4])
            pb = [((mod.ksize3[d] - 1) * mod.dilation) // 2 for d in range(3)]
            padded = torch.empty([dpre.shape[0]] + S + [dpre.shape[-1]], dtype=dpre.dtype, device=dev)
            dpre = dpre.contiguous()
            with torch.cuda.device(dev):
                rc = lib.nrt_pad3d(_lib.ptr(dpre), _lib.ptr(padded), dpre.shape[0], _lib.ints(list(dpre.shape[1:4])), _lib.ints(pb),
                                   _lib.ints(S), dpre.shape[-1] * 4, 0, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_pad3d')
            dpre = padded
        need_x, need_lo, need_w, need_b = ctx.needs_input_grad[:4]
        dx = dlo = dw = db = None
        c0 = x.shape[-1]
        if mod.fold_backward and _fold_backward_ok(mod, x, lo, up):
            # decoder form: the up-sampled channels are differentiated on the low-resolution grid (8 parity groups x 2x2x2 taps of
            # the space-to-depth gradient: 0.30 of the matrix work of the 27-tap form), the skip channels as a plain convolution
            k5 = kernel.detach()
            s2d = _space_to_depth2(dpre) if (need_lo or need_w) else None
            if need_x:
                dx = _conv_dgrad(dpre, k5[..., :c0, :], mod.ksize3, 1)
            if need_lo:
                dlo = _conv_dgrad_lo_folded(s2d, k5[..., c0:, :])
            if need_w or need_b:
                dw = torch.empty_like(kernel, dtype=torch.float32)
                dws = torch.zeros(3, 3, 3, c0, mod.cout, dtype=torch.float32, device=dev)
                db = torch.zeros(mod.cout, dtype=torch.float32, device=dev)
                xs = x.contiguous()
                with torch.cuda.device(dev):
                    rc = lib.nrt_conv3d_wgrad_f32(_lib.ptr(xs), _lib.ptr(dpre), _lib.ptr(dws), _lib.ptr(db), xs.shape[0],
                                                  _lib.ints(list(xs.shape[1:4])), c0, mod.cout, _lib.ints(mod.ksize3), 1,
                                                  _lib.stream_ptr(dev))
                _lib.check(rc, 'nrt_conv3d_wgrad_f32')
                dw[..., :c0, :] = dws
                dw[..., c0:, :] = _conv_wgrad_lo_folded(lo, s2d, mod.cout)
            return dx, dlo, dw if need_w else None, db if need_b else None, None, None, None, None
        if need_w or need_b:
            dw = torch.zeros_like(kernel, dtype=torch.float32)
            db = torch.zeros(mod.cout, dtype=torch.float32, device=dev)
            xs = x.contiguous()
            lo_c = None if lo is None else lo.contiguous()
            if lo_c is not None and c0 % 4:
                xs, lo_c = _upsample_concat(xs, lo_c, up), None          # a channel quad would straddle the two sources
            B, S = xs.shape[0], list(xs.shape[1:4])
            with torch.cuda.device(dev):
                rc = lib.nrt_conv3d_wgrad2_f32(_lib.ptr(xs), xs.shape[-1], _lib.ptr(lo_c), 0 if lo_c is None else lo_c.shape[-1],
                                               _lib.ints(up) if lo_c is not None else None, _lib.ptr(dpre), _lib.ptr(dw), _lib.ptr(db),
                                               B, _lib.ints(S), mod.cout, _lib.ints(mod.ksize3), mod.dilation, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_conv3d_wgrad2_f32')
        k5 = kernel.detach()
        if lo is not None and need_x and need_lo and mod.cin <= 64:
            # one dgrad for both sources of the fused loader (dpre is staged once, three N-tiles per A fragment)
            full = _conv_dgrad(dpre, k5, mod.ksize3, mod.dilation)
            dx = full[..., :c0]
            dlo = _upsample_sum(full, c0, mod.cin - c0, list(lo.shape[1:4]), up)
        else:
            if need_x:
                dx = _conv_dgrad(dpre, k5[..., :c0, :], mod.ksize3, mod.dilation)
            if lo is not None and need_lo:
                full = _conv_dgrad(dpre, k5[..., c0:, :], mod.ksize3, mod.dilation)
                dlo = _upsample_sum(full, 0, full.shape[-1], list(lo.shape[1:4]), up)
        return dx, dlo, dw if need_w else None, db if need_b else None, None, None, None, None


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool3, padding):
        with torch.no_grad():
            y = _maxpool(x, pool3, padding)
        ctx.save_for_backward(x)
        ctx.cfg = (tuple(pool3), padding)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        pool3, padding = ctx.cfg
        lib = _lib.lib()
        dev = g.device
        x = x.contiguous()
        g = g.contiguous()
        dx = torch.empty_like(x)
        with torch.cuda.device(dev):
            rc = lib.nrt_maxpool3d_bwd_f32(_lib.ptr(x), _lib.ptr(g), _lib.ptr(dx), x.shape[0], _lib.ints(list(x.shape[1:4])),
                                           x.shape[-1], _lib.ints(pool3), int(padding == 'same'), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_maxpool3d_bwd_f32')
        return dx, None, None


class _MergeFn(torch.autograd.Function):
    """concatenate([skip, UpSampling3D(lo)]) (skip may be None: plain up-sampling)."""

    @staticmethod
    def forward(ctx, skip, lo, up3):
        with torch.no_grad():
            y = _upsample_concat(skip, lo, up3)
        ctx.cfg = (0 if skip is None else skip.shape[-1], lo.shape[-1], list(lo.shape[1:4]), tuple(up3))
        return y

    @staticmethod
    def backward(ctx, g):
        c0, c1, lo_shape, up3 = ctx.cfg
        dskip = g[..., :c0].contiguous() if c0 and ctx.needs_input_grad[0] else None
        dlo = _upsample_sum(g, c0, c1, lo_shape, up3) if ctx.needs_input_grad[1] else None
        return dskip, dlo, None


def _head_grads(x, kernel, mod, dz, needs):
    """(dx, dw, db) of the likelihood conv from dz = d loss / d logits; needs = which of the three are wanted"""
    lib = _lib.lib()
    dev = dz.device
    dx = dw = db = None
    if needs[1] or needs[2]:
        xin = x.contiguous()
        dw = torch.zeros_like(kernel, dtype=torch.float32)
        db = torch.zeros(mod.cout, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.nrt_conv3d_wgrad_f32(_lib.ptr(xin), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(db), xin.shape[0],
                                          _lib.ints(list(xin.shape[1:4])), mod.cin, mod.cout, _lib.ints(mod.ksize3), 1,
                                          _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_conv3d_wgrad_f32')
    if needs[0]:
        dx = _conv_dgrad(dz, kernel.detach(), mod.ksize3, 1)
    return dx, dw, db


class SoftmaxSource:
    """
    Stamp (`y._nrt_softmax_src`) on a channel soft-max output that is still its producer's untouched result: `inputs` are the
    autograd inputs of the producer and `grads_from_dz(dz, needs)` turns d loss / d logits into their gradients.  A loss that can
    form dz itself (metrics._SegLossFn: Dice + CCE of the prediction) attaches to `inputs` directly, and the gradient wrt the
    probabilities -- with the separate soft-max backward pass over it -- is never materialised.
    """

    def __init__(self, y, inputs, grads_from_dz):
        self.inputs, self._grads_from_dz = tuple(inputs), grads_from_dz
        self.version = y._version
        # the closure keeps the producer's inputs alive outside save_for_backward: the in-place check autograd would do is done here
        self.input_versions = tuple(None if t is None else t._version for t in self.inputs)

    def valid_for(self, y):
        """The shortcut replaces y's own autograd node: it must not be taken when somebody observes the gradient AT y (a hook on the
        prediction, retain_grad) -- those would silently see nothing -- nor when y was modified since the producer wrote it."""
        if y._version != self.version or not torch.is_grad_enabled() or not y.requires_grad:
            return False
        if y.retains_grad or getattr(y, '_backward_hooks', None):
            return False
        return True

    def grads_from_dz(self, dz, needs):
        for t, v in zip(self.inputs, self.input_versions):
            if t is not None and t._version != v:
                raise RuntimeError('neurite_amd: an input of the soft-max head was modified in place after the forward pass; its '
                                   'gradient through the joint segmentation loss would be computed from the modified values')
        return self._grads_from_dz(dz, needs)


def _stamp_softmax(y, inputs, grads_from_dz):
    if torch.is_grad_enabled() and y.requires_grad:
        y._nrt_softmax_src = SoftmaxSource(y, inputs, grads_from_dz)
    return y


class _HeadFn(torch.autograd.Function):
    """likelihood 1x1 conv + channel softmax in one pass (no logits tensor); backward from the prediction alone."""

    @staticmethod
    def forward(ctx, x, kernel, bias, mod):
        with torch.no_grad():
            y = _conv1x1_softmax(x, kernel, bias, True, 0)
        ctx.mod = mod
        ctx.save_for_backward(x, kernel, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, kernel, y = ctx.saved_tensors
        lib = _lib.lib()
        dev = g.device
        g = g.contiguous()
        dz = torch.empty_like(y)
        with torch.cuda.device(dev):
            rc = lib.nrt_softmax_bwd_f32(_lib.ptr(y), _lib.ptr(g), _lib.ptr(dz), y.numel() // y.shape[-1], y.shape[-1],
                                         _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_softmax_bwd_f32')
        return _head_grads(x, kernel, ctx.mod, dz, ctx.needs_input_grad) + (None,)


def _head(x, kernel, bias, mod):
    y = _HeadFn.apply(x, kernel, bias, mod)
    return _stamp_softmax(y, (x, kernel, bias), lambda dz, needs: _head_grads(x, kernel, mod, dz, needs))


def _softmax_with_grad(z):
    return _stamp_softmax(_SoftmaxFn.apply(z), (z,), lambda dz, needs: (dz,))


class _ChannelScaleFn(torch.autograd.Function):
    """y[b, ..., c] = x[b, ..., c] * scale[b, c]: Keras Dropout with noise_shape [None, 1, .., 1, C] (whole feature maps are
    dropped, models.py:1390-1399) once the mask is drawn; the backward is the same scaling of the gradient."""

    @staticmethod
    def _apply(x, scale):
        x = x.contiguous()
        y = torch.empty_like(x)
        lib = _lib.lib()
        dev = x.device
        zero = torch.zeros(x.shape[-1], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            for b in range(x.shape[0]):
                rc = lib.nrt_add_act_affine_f32(_lib.ptr(x[b]), None, _lib.ptr(scale[b]), _lib.ptr(zero), _lib.ptr(y[b]), x[b].numel(),
                                                x.shape[-1], 0, _lib.stream_ptr(dev))
                _lib.check(rc, 'nrt_add_act_affine_f32')
        return y

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        with torch.no_grad():
            return _ChannelScaleFn._apply(x, scale)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return _ChannelScaleFn._apply(g, scale), None


class _AddActFn(torch.autograd.Function):
    """y = act(a [+ b]): the residual merge (`KL.add`, models.py:1426) and the stand-alone Activation layers"""

    @staticmethod
    def forward(ctx, a, b, act):
        with torch.no_grad():
            y = _elementwise(a, b, act=act)
        ctx.act = act
        ctx.has_b = b is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        d = _act_bwd(g, y, ctx.act)
        return d, (d if ctx.has_b else None), None


class _MulFn(torch.autograd.Function):
    """y = a * b (`KL.multiply` of the prior and the sigmoid likelihood, models.py:412-417)"""

    @staticmethod
    def forward(ctx, a, b):
        with torch.no_grad():
            y = _elementwise(a, b, mul=True)
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = _elementwise(g, b, mul=True) if ctx.needs_input_grad[0] else None
        db = _elementwise(g, a, mul=True) if ctx.needs_input_grad[1] else None
        return da, db


def _channel_sums(a, b=None):
    lib = _lib.lib()
    dev = a.device
    C = a.shape[-1]
    out = torch.zeros(C, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_channel_sums_f32(_lib.ptr(a), _lib.ptr(b), a.numel() // C, C, _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_channel_sums_f32')
    return out


class _BatchNormFn(torch.autograd.Function):
    """training-mode BatchNormalization over all axes but the last; [C]-sized arithmetic is glue, volume passes are kernels"""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod):
        with torch.no_grad():
            x = x.contiguous()
            C = x.shape[-1]
            n = x.numel() // C
            mean = _channel_sums(x) / n
            var = torch.clamp(_channel_sums(x, x) / n - mean * mean, min=0.0)
            inv = torch.rsqrt(var + mod.epsilon)
            scale = (gamma * inv).contiguous()
            shift = (beta - mean * scale).contiguous()
            y = _elementwise(x, scale=scale, shift=shift)
            mod.moving_mean.mul_(mod.momentum).add_(mean * (1 - mod.momentum))
            mod.moving_variance.mul_(mod.momentum).add_(var * (1 - mod.momentum))
        ctx.save_for_backward(x, gamma, mean, inv)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, inv = ctx.saved_tensors
        lib = _lib.lib()
        dev = g.device
        g = g.contiguous()
        C = x.shape[-1]
        n = x.numel() // C
        sg = _channel_sums(g)
        sgx = _channel_sums(g, x)
        sgxh = (sgx - mean * sg) * inv                         # sum g * xhat
        dgamma, dbeta = sgxh, sg
        dx = None
        if ctx.needs_input_grad[0]:
            scale = gamma * inv
            m1, m2 = sg / n, sgxh / n
            A = scale.contiguous()
            B = (-scale * inv * m2).contiguous()
            C0 = (-scale * m1 + scale * inv * m2 * mean).contiguous()
            dx = torch.empty_like(x)
            with torch.cuda.device(dev):
                rc = lib.nrt_channel_axpby_f32(_lib.ptr(g), _lib.ptr(x), _lib.ptr(A), _lib.ptr(B), _lib.ptr(C0), _lib.ptr(dx),
                                               x.numel(), C, _lib.stream_ptr(dev))
            _lib.check(rc, 'nrt_channel_axpby_f32')
        return dx, dgamma, dbeta, None


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        with torch.no_grad():
            y = _softmax(z)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        lib = _lib.lib()
        dev = g.device
        g = g.contiguous()
        dz = torch.empty_like(y)
        with torch.cuda.device(dev):
            rc = lib.nrt_softmax_bwd_f32(_lib.ptr(y), _lib.ptr(g), _lib.ptr(dz), y.numel() // y.shape[-1], y.shape[-1],
                                         _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_softmax_bwd_f32')
        return dz


class _PendingConv:
    """inputs of a decoder convolution whose only consumer is the soft-max head: both run as one kernel at the head's op"""

    def __init__(self, x, lo):
        self.x, self.lo = x, lo


class ConvNet(nn.Module):
    """
    A built conv_enc / conv_dec / unet graph: an ordered list of named Keras-equivalent layers
    (`self.layer_names`, `self.get_layer(name)`) executed on the HIP kernels.
    """

    def __init__(self, name, ndims, input_shapes, ops, output, modules):
        super().__init__()
        self.name = name
        self.ndims = ndims
        self.input_shapes = input_shapes            # list of per-input shapes (without batch)
        self.ops = ops                              # list of dicts, executed in order
        self.output_name = output
        self.layers_by_name = nn.ModuleDict()
        for k, m in modules.items():
            self.layers_by_name[k] = m
        self.layer_names = [op['name'] for op in ops]
        self.output_shape = ops[-1].get('shape') if ops else None
        self.conv_variant = 0                       # 0 auto, 1 direct, 2 MFMA (tests / tuning)
        self.fold_head = True                       # inference: last decoder convolution + likelihood + soft-max as ONE kernel where it applies
        # {conv layer: the soft-max likelihood that is its ONLY consumer}: candidates for nrt_conv3d_up2_head_f32 (models.py:1545-1605)
        uses = {}
        for op in ops:
            for key in ('src', 'lo', 'skip', 'a', 'b'):
                v = op.get(key)
                for n in (v if isinstance(v, (list, tuple)) else [v]):
                    if isinstance(n, str):
                        uses.setdefault(n, []).append(op)
        # {single-channel first convolution: the 2x2x2 max-pooling that reads it}: candidates for nrt_conv3d_c1_pool_f32 (models.py:1378-1438)
        self._pool_of = {}
        for op in ops:
            if op['kind'] == 'maxpool' and tuple(op.get('pool', ())) == (2, 2, 2) and isinstance(op.get('src'), str):
                src = next((o for o in ops if o['name'] == op['src']), None)
                if src is not None and src['kind'] == 'conv' and not src.get('lo') and op['src'] not in self._pool_of:
                    self._pool_of[op['src']] = op['name']
        self._head_of = {}
        for op in ops:
            if op['kind'] == 'conv' and op.get('lo') and op['name'] != output:
                u = uses.get(op['name'], [])
                if len(u) == 1 and u[0]['kind'] == 'likelihood' and u[0].get('fuse_softmax') and u[0].get('src') == op['name']:
                    self._head_of[op['name']] = u[0]['name']
        self.eval()                                 # Keras predict semantics; model.train() records the graph for autograd

    def _head_foldable(self, conv_name, head_name, x, lo, up):
        """run-time half of the test: shapes and settings the folded kernel takes (csrc/conv_up2.h: up2_head_ok)"""
        c, h = self.layers_by_name[conv_name], self.layers_by_name[head_name]
        if lo is None or up is None or tuple(up) != (2, 2, 2) or self.conv_variant not in (0, 4):
            return False
        if c.ksize3 != (3, 3, 3) or c.dilation != 1 or c.padding != 'same' or c.act > _ACT_LAST_FUSED or c.post_softmax:
            return False
        if tuple(h.ksize3) != (1, 1, 1) or h.cin != c.cout:
            return False
        B, S = x.shape[0], list(x.shape[1:4])
        if B * S[0] * S[1] * S[2] * h.cout >= (1 << 30):
            return False
        return _lib.lib().nrt_conv3d_up2_head_supported(x.shape[-1], lo.shape[-1], c.cout, h.cout, _lib.ints(S)) == 1

    # ---- Keras Model weight API (modelio: neurite/tf/modelio.py:111-143 saves/loads `model.get_weights()` lists) ----
    def _weight_tensors(self):
        """weights in Keras order: layers in graph order; Conv: kernel, bias; BatchNormalization: gamma, beta, mean, variance"""
        out = []
        for name in self.layer_names:
            if name in self.layers_by_name:
                m = self.layers_by_name[name]
                if isinstance(m, _Conv):
                    out += [(name + '/kernel', m.kernel, self.ndims), (name + '/bias', m.bias, None)]
                elif isinstance(m, _BatchNorm):
                    out += [(name + '/gamma', m.gamma, None), (name + '/beta', m.beta, None),
                            (name + '/moving_mean', m.moving_mean, None), (name + '/moving_variance', m.moving_variance, None)]
        return out

    def get_weights(self):
        """list of numpy arrays in Keras `model.get_weights()` order and layout (Conv{N}D kernels [k1..kN, Cin, Cout])."""
        res = []
        for _, t, nd in self._weight_tensors():
            a = t.detach().cpu().numpy()
            if nd is not None and nd < 3:
                a = a.reshape(a.shape[3 - nd:])                  # drop the lifted singleton kernel dims
            res.append(a)
        return res

    def set_weights(self, weights):
        """counterpart of `get_weights` (e.g. the arrays of a Keras-trained neurite unet exported with np.savez)."""
        slots = self._weight_tensors()
        if len(weights) != len(slots):
            raise ValueError('You called `set_weights(weights)` on model "%s" with a weight list of length %d, but the '
                             'model was expecting %d weights.' % (self.name, len(weights), len(slots)))
        with torch.no_grad():
            for (name, t, nd), w in zip(slots, weights):
                w = np.asarray(w, dtype=np.float32)
                if nd is not None and nd < 3:
                    w = w.reshape((1,) * (3 - nd) + w.shape)
                if tuple(w.shape) != tuple(t.shape):
                    raise ValueError('Layer weight shape %s not compatible with provided weight shape %s (%s)'
                                     % (tuple(t.shape), tuple(w.shape), name))
                t.copy_(torch.from_numpy(w))
        for m in self.layers_by_name.values():
            if isinstance(m, _Conv):
                m.invalidate_packed()

    def _weights_by_layer(self):
        """[(layer name, [(variable name, tensor, conv ndims or None), ...])] in Keras layer order"""
        groups = []
        for name, t, nd in self._weight_tensors():
            layer, var = name.rsplit('/', 1)
            if not groups or groups[-1][0] != layer:
                groups.append((layer, []))
            groups[-1][1].append((var, t, nd))
        return groups

    def save_weights(self, path):
        """
        `.npz` (default; np.savez archive keyed by `layer/variable`) or, for a path ending in .h5 / .hdf5, the Keras
        `save_weights` HDF5 layout (root attrs `layer_names`; per layer a group with attr `weight_names`
        and one dataset per variable, `layer/kernel:0` ...) so that the file loads into the Keras-built neurite unet.
        """
        if _is_h5(path):
            h5py = _h5py()
            with h5py.File(path, 'w') as f:
                self._write_h5_weights(f)
            return
        np.savez(path, **self._npz_dict())

    def _npz_dict(self):
        return {name: a for (name, _, _), a in zip(self._weight_tensors(), self.get_weights())}

    def _write_h5_weights(self, f):
        arrays = dict(zip([n for n, _, _ in self._weight_tensors()], self.get_weights()))
        groups = self._weights_by_layer()
        _h5_set_attr(f, 'layer_names', [l.encode('utf8') for l, _ in groups])
        # (what Keras' loader looks at: without `keras_version` it takes the file for Keras 1 and transposes convolution kernels)
        f.attrs['backend'] = b'tensorflow'
        f.attrs['keras_version'] = b'2.4.0'
        for layer, vs in groups:
            g = f.create_group(layer)
            names = ['%s/%s:0' % (layer, var) for var, _, _ in vs]
            _h5_set_attr(g, 'weight_names', [n.encode('utf8') for n in names])
            for n, (var, _, _) in zip(names, vs):
                g.create_dataset(n, data=arrays[layer + '/' + var])

    def _read_h5_weights(self, f, by_name):
        """Keras HDF5 weights (the file itself for `save_weights`, its `model_weights` group for `model.save`):
        layers are matched by name when every layer of this model is present (or by_name), else in order, as Keras does."""
        if 'layer_names' not in f.attrs and 'layer_names0' not in f.attrs and 'model_weights' in f:
            f = f['model_weights']
        stored = [_h5_str(n) for n in _h5_attr(f, 'layer_names')]
        with_w = [n for n in stored if len(_h5_attr(f[n], 'weight_names'))]
        groups = self._weights_by_layer()
        mine = [l for l, _ in groups]
        if by_name:
            pairs = [(g, g[0]) for g in groups if g[0] in with_w]
        elif all(l in with_w for l in mine):
            pairs = [(g, g[0]) for g in groups]
        else:
            if len(with_w) != len(mine):
                raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                                 % (len(with_w), len(mine)))
            pairs = list(zip(groups, with_w))
        out = {}
        for (layer, vs), src in pairs:
            g = f[src]
            wn = [_h5_str(n) for n in _h5_attr(g, 'weight_names')]
            if len(wn) != len(vs):
                raise ValueError('Layer %s expects %d weights, the file holds %d for %s' % (layer, len(vs), len(wn), src))
            for (var, _, _), n in zip(vs, wn):
                out[layer + '/' + var] = np.asarray(g[n])
        return out

    def load_weights(self, path, by_name=False):
        """counterpart of `save_weights`; also reads HDF5 files written by Keras (`model.save_weights` / `model.save`: h5py if
        installed, else neurite_amd.h5lite).  by_name=True loads only the layers found in the file (Keras semantics)."""
        if _is_h5(path):
            h5py = _h5py()
            with h5py.File(path, 'r') as f:
                found = self._read_h5_weights(f, by_name)
        else:
            with np.load(path) as z:
                found = {name: z[name] for name, _, _ in self._weight_tensors() if name in z.files}
        slots = self._weight_tensors()
        if not by_name:
            missing = [n for n, _, _ in slots if n not in found]
            if missing:
                raise ValueError('weight file %s has no entry for %s' % (path, missing))
            self.set_weights([found[n] for n, _, _ in slots])
            return
        current = dict(zip([n for n, _, _ in slots], self.get_weights()))
        current.update(found)
        self.set_weights([current[n] for n, _, _ in slots])

    # ---- LoadableModel (neurite/tf/modelio.py:78-143): architecture arguments travel with the weights ----
    def get_config(self):
        if not hasattr(self, 'config'):
            raise RuntimeError('this network was not built through unet / conv_enc / conv_dec, so it has no stored config')
        return self.config['params']

    @property
    def metadata(self):
        return self.get_config()['metadata']

    def save(self, path):
        """weights + the builder arguments (`model_config`), reloadable with `neurite_amd.models.load(path)`"""
        cfg = json.dumps({'class_name': self.config['builder'] if hasattr(self, 'config') else None,
                          'config': self.get_config()})
        if hasattr(self, 'config') and not self.config['loadable']:
            raise RuntimeError('networks assembled from an input_model are saved part by part (save the encoder and the '
                               'decoder arguments), as their config cannot name the graph they were grafted on')
        if _is_h5(path):
            h5py = _h5py()
            with h5py.File(path, 'w') as f:
                f.attrs['model_config'] = cfg
                self._write_h5_weights(f.create_group('model_weights'))
            return
        np.savez(path, __model_config__=np.array(cfg), **self._npz_dict())

    def get_layer(self, name):
        if name in self.layers_by_name:
            return self.layers_by_name[name]
        if name in self.layer_names:
            return self.ops[self.layer_names.index(name)]
        raise ValueError('No such layer: %s' % name)

    def keras_graph(self):
        """
        The network as the list of Keras layers the reference builders create (neurite/tf/models.py:1309-1617, :378-436):
        per layer `name`, Keras `class`, constructor `config`, producer names of its `inputs`, and `output_shape`
        (batch = None).  Fused ops are expanded (`up_N` + `merge_N` for the in-loader up-sampling/concatenation,
        `likelihood` + `prediction` for the fused head); layers no output depends on are left out, as in a Keras Model.
        tests/test_unet_graph.py compares this with the graphs recorded from the reference's own builders.
        """
        nd = self.ndims
        layers = []

        def shape_of(op):
            sp, c = op['shape']
            return [None] + [int(v) for v in sp[3 - nd:]] + [int(c)]

        def emit(name, cls, config, inputs, out_shape):
            layers.append({'name': name, 'class': cls, 'config': config, 'inputs': list(inputs), 'output_shape': out_shape})

        ndrop = 0
        for op in self.ops:
            kind, name = op['kind'], op['name']
            if kind == 'input':
                emit(name, 'InputLayer', {}, [], shape_of(op))
            elif kind == 'input_concat':
                emit(name, 'Concatenate', {'axis': -1}, op['src'], shape_of(op))
            elif kind in ('conv', 'likelihood'):
                m = self.layers_by_name[name]
                cfg = {'filters': m.cout, 'kernel_size': [int(k) for k in m.ksize3[3 - nd:]], 'strides': [1] * nd,
                       'padding': getattr(m, 'keras_padding', m.padding), 'dilation_rate': [m.dilation] * nd,
                       'activation': m.activation if m.activation is not None else 'linear', 'use_bias': True}
                emit(name, 'Conv%dD' % nd, cfg, [op['merge'] if op.get('lo') else op['src']], shape_of(op))
            elif kind == 'dropout':
                sp, c = op['shape']
                auto = 'dropout' if ndrop == 0 else 'dropout_%d' % ndrop
                ndrop += 1
                op_name = auto                                        # the reference passes no name: Keras numbers them
                emit(op_name, 'Dropout', {'rate': float(op['rate']), 'noise_shape': [None] + [1] * nd + [int(c)]},
                     [op['src']], shape_of(op))
                layers[-1]['alias'] = name
            elif kind == 'maxpool':
                pool = [int(p) for p in op['pool'][3 - nd:]]
                emit(name, 'MaxPooling%dD' % nd, {'pool_size': pool, 'strides': pool, 'padding': op['padding']}, [op['src']],
                     shape_of(op))
            elif kind == 'upsample':
                emit(name, 'UpSampling%dD' % nd, {'size': [int(p) for p in op['up'][3 - nd:]]}, [op['src']], shape_of(op))
            elif kind == 'merge':
                if not op.get('up_materialised', True):
                    sp, c = self._builder_state['shapes'][op['up_name']]
                    emit(op['up_name'], 'UpSampling%dD' % nd, {'size': [int(p) for p in op['up'][3 - nd:]]}, [op['lo']],
                         [None] + [int(v) for v in sp[3 - nd:]] + [int(c)])
                emit(name, 'Concatenate', {'axis': nd + 1}, [op['skip'], op.get('up_name', op['lo'])], shape_of(op))
            elif kind == 'add':
                emit(name, 'Add', {}, [op['a'], op['b']], shape_of(op))
            elif kind == 'multiply':
                emit(name, 'Multiply', {}, [op['a'], op['b']], shape_of(op))
            elif kind == 'activation':
                emit(name, 'Activation', {'activation': op['activation']}, [op['src']], shape_of(op))
            elif kind == 'bn':
                m = self.layers_by_name[name]
                emit(name, 'BatchNormalization', {'axis': int(op['axis']), 'momentum': float(m.momentum),
                                                  'epsilon': float(m.epsilon)}, [op['src']], shape_of(op))
            elif kind == 'prediction':
                if op['activation'] == 'softmax':
                    emit(name, 'Lambda', {'function': [['softmax', op.get('axis', nd + 1)]]}, [op['src']], shape_of(op))
                else:
                    emit(name, 'Activation', {'activation': op['activation']}, [op['src']], shape_of(op))
            else:
                raise RuntimeError('unknown op ' + kind)
        # dropouts are referred to by their builder names inside the op list
        alias = {l['alias']: l['name'] for l in layers if 'alias' in l}
        for l in layers:
            l['inputs'] = [alias.get(i, i) for i in l['inputs']]
            l.pop('alias', None)
        by_name = {l['name']: l for l in layers}
        out = alias.get(self.output_name, self.output_name)
        seen, stack = set(), [out]
        while stack:
            n = stack.pop()
            if n not in seen:
                seen.add(n)
                stack.extend(by_name[n]['inputs'])
        inputs = [op['name'] for op in sorted((o for o in self.ops if o['kind'] == 'input'), key=lambda o: o['index'])]
        return {'name': self.name, 'inputs': inputs, 'outputs': [out], 'layers': [l for l in layers if l['name'] in seen]}

    def _affine(self, bn):
        scale = bn.gamma.detach() / torch.sqrt(bn.moving_variance + bn.epsilon)
        shift = bn.beta.detach() - bn.moving_mean * scale
        return scale.contiguous(), shift.contiguous()

    def forward(self, inputs, return_tensors=None):
        """inputs: [B, *spatial, C] (or a list for multi-input nets).  Returns the prediction tensor
        (or a dict of the named intermediate tensors listed in return_tensors)."""
        if isinstance(inputs, (list, tuple)):
            xs = list(inputs)
        else:
            xs = [inputs]
        if len(xs) != len(self.input_shapes):
            raise ValueError('%s expects %d input(s), got %d' % (self.name, len(self.input_shapes), len(xs)))
        for x, shp in zip(xs, self.input_shapes):
            _lib.require_device(x)
            if tuple(x.shape[1:]) != tuple(shp):
                raise ValueError('input shape %s does not match the model input %s' % (tuple(x.shape[1:]), tuple(shp)))
        nd = self.ndims
        t = {}
        keep = set(return_tensors or [])
        if self.training and torch.is_grad_enabled():
            return self._forward_train(xs, keep, return_tensors)
        with torch.no_grad():
            for op in self.ops:
                kind, name = op['kind'], op['name']
                if kind == 'input':
                    t[name] = _lift(xs[op['index']].to(torch.float32), nd)
                elif kind == 'input_concat':
                    t[name] = torch.cat([t[s] for s in op['src']], -1).contiguous()      # host glue (rare)
                elif kind == 'conv':
                    lo = t[op['lo']] if op.get('lo') else None
                    head = self._head_of.get(name) if self.fold_head and name not in keep else None
                    pool = self._pool_of.get(name) if self.fold_head else None
                    if head is not None and head not in keep and self._head_foldable(name, head, t[op['src']], lo, op.get('up')):
                        t[name] = _PendingConv(t[op['src']], lo)      # runs inside the likelihood op below: one kernel, no feature tensor
                    elif pool is not None and lo is None and self.layers_by_name[name].pool_foldable(t[op['src']], self.conv_variant):
                        t[name], t[pool] = self.layers_by_name[name].run_with_pool(t[op['src']])   # the pooled tensor from the same kernel
                    else:
                        t[name] = self.layers_by_name[name](t[op['src']], lo=lo, up=op.get('up'),
                                                            variant=self.conv_variant)
                elif kind == 'dropout':
                    t[name] = t[op['src']]                                                # inference: identity
                elif kind == 'maxpool':
                    if t.get(name) is None:                                               # (else: produced by the convolution in front of it)
                        t[name] = _maxpool(t[op['src']], op['pool'], op['padding'])
                elif kind == 'upsample':
                    t[name] = _upsample_concat(None, t[op['src']], op['up'])
                elif kind == 'merge':
                    if op.get('fused') and name not in keep:
                        t[name] = None            # consumed by the next conv's loader (skip + lo), never materialised
                    else:
                        t[name] = _upsample_concat(t[op['skip']], t[op['lo']], op['up'])
                elif kind == 'add':
                    t[name] = _elementwise(t[op['a']], t[op['b']])
                elif kind == 'activation':
                    t[name] = _elementwise(t[op['src']], act=_EW_ACTS[op['activation']])
                elif kind == 'multiply':
                    t[name] = _elementwise(t[op['a']], t[op['b']], mul=True)
                elif kind == 'bn':
                    scale, shift = self._affine(self.layers_by_name[name])
                    t[name] = _elementwise(t[op['src']], scale=scale, shift=shift)
                elif kind == 'likelihood':
                    m = self.layers_by_name[name]
                    if isinstance(t[op['src']], _PendingConv):
                        pc = t[op['src']]
                        t[name] = None
                        t[op['pred_name']] = self.layers_by_name[op['src']].run_with_head(pc.x, pc.lo, m.kernel, m.bias)
                        t[op['src']] = None
                    elif op.get('fuse_softmax'):
                        t[name] = _conv1x1_softmax(t[op['src']], m.kernel, m.bias, False, 0) if name in keep else None
                        t[op['pred_name']] = _conv1x1_softmax(t[op['src']], m.kernel, m.bias, True, 0)
                    else:
                        t[name] = _conv1x1_softmax(t[op['src']], m.kernel, m.bias, False, 0) \
                            if m.cout <= 64 else m(t[op['src']])
                elif kind == 'prediction':
                    if name in t and t[name] is not None:
                        pass                                    # produced by the fused likelihood
                    elif op['activation'] == 'softmax':
                        t[name] = _softmax(t[op['src']])
                    else:
                        t[name] = _elementwise(t[op['src']], act=_act_code(op['activation']))
                else:
                    raise RuntimeError('unknown op ' + kind)
        if return_tensors:
            return {k: _unlift(t[k], nd) for k in keep}
        return _unlift(t[self.output_name], nd)


    def _forward_train(self, xs, keep, return_tensors):
        """model.train(): the same graph with every op recorded for autograd (csrc/conv_bwd.hip); the likelihood conv
        and the softmax run unfused so that the logits are available to the backward."""
        nd = self.ndims
        t = {}
        self.last_dropout_scales = {}
        for op in self.ops:
            kind, name = op['kind'], op['name']
            if kind == 'input':
                t[name] = _lift(xs[op['index']].to(torch.float32), nd)
            elif kind == 'input_concat':
                t[name] = torch.cat([t[s] for s in op['src']], -1).contiguous()
            elif kind == 'conv':
                lo = t[op['lo']] if op.get('lo') else None
                m = self.layers_by_name[name]
                t[name] = _ConvFn.apply(t[op['src']], lo, m.kernel, m.bias, m, op.get('up'), self.conv_variant, False)
            elif kind == 'dropout':
                rate = float(op.get('rate', 0))
                if rate > 0:
                    src = t[op['src']]
                    kept = (torch.rand((src.shape[0], src.shape[-1]), device=src.device) >= rate).to(torch.float32)
                    scale = (kept / (1.0 - rate)).contiguous()          # [B, C] numbers: the draw is plumbing, the scaling a kernel
                    self.last_dropout_scales[name] = scale
                    t[name] = _ChannelScaleFn.apply(src, scale)
                else:
                    t[name] = t[op['src']]
            elif kind == 'maxpool':
                t[name] = _MaxPoolFn.apply(t[op['src']], op['pool'], op['padding'])
            elif kind == 'upsample':
                t[name] = _MergeFn.apply(None, t[op['src']], op['up'])
            elif kind == 'merge':
                if op.get('fused') and name not in keep:
                    t[name] = None
                else:
                    t[name] = _MergeFn.apply(t[op['skip']], t[op['lo']], op['up'])
            elif kind == 'bn':
                m = self.layers_by_name[name]
                t[name] = _BatchNormFn.apply(t[op['src']], m.gamma, m.beta, m)
            elif kind == 'add':
                t[name] = _AddActFn.apply(t[op['a']], t[op['b']], 0)
            elif kind == 'activation':
                t[name] = _AddActFn.apply(t[op['src']], None, _EW_ACTS[op['activation']])
            elif kind == 'multiply':
                t[name] = _MulFn.apply(t[op['a']], t[op['b']])
            elif kind == 'likelihood':
                m = self.layers_by_name[name]
                if op.get('fuse_softmax') and name not in keep and m.cout <= 64 and tuple(m.ksize3) == (1, 1, 1):
                    t[name] = None
                    t[op['pred_name']] = _head(t[op['src']], m.kernel, m.bias, m)
                elif m.cout <= 64:
                    t[name] = _ConvFn.apply(t[op['src']], None, m.kernel, m.bias, m, None, 0, True)
                else:
                    t[name] = _ConvFn.apply(t[op['src']], None, m.kernel, m.bias, m, None, self.conv_variant, False)
            elif kind == 'prediction':
                if name in t and t[name] is not None:
                    pass                                    # produced by the fused head
                elif op['activation'] == 'softmax':
                    t[name] = _softmax_with_grad(t[op['src']])
                elif op['activation'] in (None, 'linear'):
                    t[name] = t[op['src']]
                else:
                    t[name] = _AddActFn.apply(t[op['src']], None, _act_code(op['activation']))
            else:
                raise NotImplementedError('neurite_amd: training through %r layers (%s) is not implemented' % (kind, name))
        if return_tensors:
            return {k: _unlift(t[k], nd) for k in keep}
        return _unlift(t[self.output_name], nd)


# --------------------------------------------------------------------------------------
# builders
# --------------------------------------------------------------------------------------

class _Builder:
    def __init__(self, ndims):
        self.ndims = ndims
        self.ops = []
        self.modules = {}
        self.shapes = {}        # name -> (spatial3 tuple, channels)

    def add(self, op, shape):
        self.ops.append(op)
        self.shapes[op['name']] = shape
        op['shape'] = shape
        return op['name']


def _encoder(bld, nb_features, input_shape, nb_levels, conv_size, prefix, feat_mult, pool_size, dilation_rate_mult,
             padding, activation, layer_nb_feats, use_residuals, nb_conv_per_level, conv_dropout, batch_norm,
             src_name):
    """neurite/tf/models.py:1360-1438.  Returns the name of the last tensor."""
    ndims = bld.ndims
    pool3 = _triple(pool_size, ndims, 'pool_size')
    k3 = _triple(conv_size, ndims, 'conv_size')
    last = src_name
    lfidx = 0
    for level in range(nb_levels):
        lvl_first = last
        if isinstance(nb_features, list):                                        # :1367-1373
            lfidx = 0
            if isinstance(nb_features[level], list):
                layer_nb_feats = nb_features[level]
                nb_conv_per_level = len(layer_nb_feats)
            else:
                nb_lvl_feats = nb_conv_per_level * [nb_features[level]]
        else:
            nb_lvl_feats = int(np.round(nb_features * feat_mult ** level))      # :1375
        dil = dilation_rate_mult ** level                                        # :1376
        for conv in range(nb_conv_per_level):
            if layer_nb_feats is not None:                                       # :1379-1381
                nb_lvl_feats = layer_nb_feats[lfidx]
                lfidx += 1
            name = '%s_conv_downarm_%d_%d' % (prefix, level, conv)
            # with residuals the last conv of a level is built WITHOUT conv_kwargs: no activation and dilation_rate 1  :1384-1388
            tail = use_residuals and conv == nb_conv_per_level - 1
            act, cdil = (None, 1) if tail else (activation, dil)
            sp, cin = bld.shapes[last]
            bld.modules[name] = _Conv(name, cin, int(nb_lvl_feats), k3, cdil, padding, act)
            last = bld.add({'kind': 'conv', 'name': name, 'src': last}, (_conv_out(sp, k3, cdil, padding), int(nb_lvl_feats)))
            if conv_dropout > 0:                                                 # :1390-1399
                name = '%s_dropout_downarm_%d_%d' % (prefix, level, conv)
                last = bld.add({'kind': 'dropout', 'name': name, 'src': last, 'rate': conv_dropout}, bld.shapes[last])
        if use_residuals:                                                        # :1401-1429
            convarm = last
            nb_in, nb_out = bld.shapes[lvl_first][1], bld.shapes[convarm][1]
            add_layer = lvl_first
            if nb_in > 1 and nb_out > 1 and nb_in != nb_out:
                name = '%s_expand_down_merge_%d' % (prefix, level)
                sp, cin = bld.shapes[lvl_first]
                bld.modules[name] = _Conv(name, cin, int(nb_lvl_feats), k3, dil, padding, activation)
                add_layer = bld.add({'kind': 'conv', 'name': name, 'src': lvl_first},
                                    (_conv_out(sp, k3, dil, padding), int(nb_lvl_feats)))
                # :1412-1423 builds a Dropout on this tensor too, but its output is overwritten by the KL.add below before
                # anything reads it (add_layer was bound before the dropout): the layer is not part of the Keras model
            name = '%s_res_down_merge_%d' % (prefix, level)
            last = bld.add({'kind': 'add', 'name': name, 'a': add_layer, 'b': convarm}, bld.shapes[convarm])
            name = '%s_res_down_merge_act_%d' % (prefix, level)
            last = bld.add({'kind': 'activation', 'name': name, 'src': last, 'activation': activation}, bld.shapes[last])
        if batch_norm is not None:                                               # :1431-1433
            name = '%s_bn_down_%d' % (prefix, level)
            bld.modules[name] = _BatchNorm(name, bld.shapes[last][1])
            last = bld.add({'kind': 'bn', 'name': name, 'src': last, 'axis': batch_norm}, bld.shapes[last])
        if level < (nb_levels - 1):                                              # :1436-1438
            name = '%s_maxpool_%d' % (prefix, level)
            sp, c = bld.shapes[last]
            osp = tuple((sp[d] + pool3[d] - 1) // pool3[d] if padding == 'same' else sp[d] // pool3[d] for d in range(3))
            last = bld.add({'kind': 'maxpool', 'name': name, 'src': last, 'pool': pool3, 'padding': padding}, (osp, c))
    return last


def _conv_out(sp, k3, dil, padding):
    if padding == 'same':
        return tuple(sp)
    return tuple(sp[d] - (k3[d] - 1) * dil for d in range(3))


def _decoder(bld, nb_features, nb_levels, conv_size, nb_labels, prefix, feat_mult, pool_size, use_skip_connections,
             padding, dilation_rate_mult, activation, use_residuals, final_pred_activation, nb_conv_per_level,
             layer_nb_feats, batch_norm, conv_dropout, last, enc_conv_per_level):
    """neurite/tf/models.py:1510-1613."""
    ndims = bld.ndims
    pool3 = _triple(pool_size, ndims, 'pool_size')
    k3 = _triple(conv_size, ndims, 'conv_size')
    lfidx = 0
    for level in range(nb_levels - 1):
        if isinstance(nb_features, list):                                        # :1517-1524
            lfidx = 0
            lindex = nb_levels - level - 2
            if isinstance(nb_features[lindex], list):
                layer_nb_feats = nb_features[lindex]
                nb_conv_per_level = len(layer_nb_feats)
            else:
                nb_lvl_feats = nb_features[lindex]
        else:
            nb_lvl_feats = int(np.round(nb_features * feat_mult ** (nb_levels - 2 - level)))   # :1526
        dil = dilation_rate_mult ** (nb_levels - 2 - level)                     # :1527
        up_name = '%s_up_%d' % (prefix, nb_levels + level)                      # :1530-1532
        lo = last
        sp_lo, c_lo = bld.shapes[lo]
        sp_up = tuple(sp_lo[d] * pool3[d] for d in range(3))
        need_up_tensor = use_residuals or not use_skip_connections
        if need_up_tensor:
            last = bld.add({'kind': 'upsample', 'name': up_name, 'src': lo, 'up': pool3}, (sp_up, c_lo))
        up_tensor = last
        fused = None
        if use_skip_connections:                                                 # :1536-1542
            ncpl = enc_conv_per_level[nb_levels - 2 - level] if isinstance(enc_conv_per_level, list) else enc_conv_per_level
            conv_name = '%s_conv_downarm_%d_%d' % (prefix, nb_levels - 2 - level, ncpl - 1)
            if conv_name not in bld.shapes:
                raise ValueError('No such layer: %s' % conv_name)
            sp_s, c_s = bld.shapes[conv_name]
            if tuple(sp_s) != tuple(sp_up):
                raise ValueError('A `Concatenate` layer requires inputs with matching shapes except for the concat axis. '
                                 'Got inputs shapes: %s, %s' % (sp_s, sp_up))
            name = '%s_merge_%d' % (prefix, nb_levels + level)
            if not need_up_tensor:
                bld.shapes[up_name] = (sp_up, c_lo)      # UpSampling3D happens inside the next conv's loader
            fused = {'skip': conv_name, 'lo': lo, 'up': pool3, 'name': name}
            last = bld.add({'kind': 'merge', 'name': name, 'skip': conv_name, 'lo': lo, 'up': pool3, 'fused': True,
                            'up_name': up_name, 'up_materialised': need_up_tensor}, (sp_up, c_s + c_lo))
        for conv in range(nb_conv_per_level):                                    # :1545-1555
            if layer_nb_feats is not None:
                nb_lvl_feats = layer_nb_feats[lfidx]
                lfidx += 1
            name = '%s_conv_uparm_%d_%d' % (prefix, nb_levels + level, conv)
            tail = use_residuals and conv == nb_conv_per_level - 1                # :1552-1555, as in the encoder
            act, cdil = (None, 1) if tail else (activation, dil)
            sp, cin = bld.shapes[last]
            bld.modules[name] = _Conv(name, cin, int(nb_lvl_feats), k3, cdil, padding, act)
            op = {'kind': 'conv', 'name': name, 'src': last}
            if conv == 0 and fused is not None:
                op.update({'src': fused['skip'], 'lo': fused['lo'], 'up': fused['up'], 'merge': fused['name']})
            last = bld.add(op, (_conv_out(sp, k3, cdil, padding), int(nb_lvl_feats)))
            if conv_dropout > 0:
                name = '%s_dropout_uparm_%d_%d' % (prefix, level, conv)
                last = bld.add({'kind': 'dropout', 'name': name, 'src': last, 'rate': conv_dropout}, bld.shapes[last])
        if use_residuals:                                                        # :1568-1588
            add_layer = up_tensor
            nb_in, nb_out = bld.shapes[add_layer][1], bld.shapes[last][1]
            if nb_in > 1 and nb_out > 1 and nb_in != nb_out:
                name = '%s_expand_up_merge_%d' % (prefix, level)
                sp, cin = bld.shapes[add_layer]
                bld.modules[name] = _Conv(name, cin, int(nb_lvl_feats), k3, dil, padding, activation)
                add_layer = bld.add({'kind': 'conv', 'name': name, 'src': add_layer},
                                    (_conv_out(sp, k3, dil, padding), int(nb_lvl_feats)))
                if conv_dropout > 0:                                             # :1579-1582: a second Dropout on the conv arm
                    name = '%s_dropout_up_merge_%d_%d' % (prefix, level, nb_conv_per_level - 1)
                    last = bld.add({'kind': 'dropout', 'name': name, 'src': last, 'rate': conv_dropout}, bld.shapes[last])
            name = '%s_res_up_merge_%d' % (prefix, level)
            last = bld.add({'kind': 'add', 'name': name, 'a': last, 'b': add_layer}, bld.shapes[add_layer])
            name = '%s_res_up_merge_act_%d' % (prefix, level)
            last = bld.add({'kind': 'activation', 'name': name, 'src': last, 'activation': activation}, bld.shapes[last])
        if batch_norm is not None:                                               # :1590-1592
            name = '%s_bn_up_%d' % (prefix, level)
            bld.modules[name] = _BatchNorm(name, bld.shapes[last][1])
            last = bld.add({'kind': 'bn', 'name': name, 'src': last, 'axis': batch_norm}, bld.shapes[last])

    # likelihood (1x1 conv, no activation) and prediction  :1594-1613
    name = '%s_likelihood' % prefix
    sp, cin = bld.shapes[last]
    bld.modules[name] = _Conv(name, cin, int(nb_labels), (1, 1, 1), 1, 'same', None)
    bld.modules[name].keras_padding = 'valid'     # convL(nb_labels, 1, activation=None): Keras' default; identical at 1x1x1
    pred = '%s_prediction' % prefix
    fuse = final_pred_activation == 'softmax' and nb_labels <= 64
    last = bld.add({'kind': 'likelihood', 'name': name, 'src': last, 'fuse_softmax': fuse, 'pred_name': pred},
                   (sp, int(nb_labels)))
    if final_pred_activation == 'softmax':
        print("using final_pred_activation %s for %s" % (final_pred_activation, prefix))
        act = 'softmax'
    else:
        act = 'linear' if final_pred_activation is None else final_pred_activation
        _act_code(act)
    last = bld.add({'kind': 'prediction', 'name': pred, 'src': name, 'activation': act}, (sp, int(nb_labels)))
    return last


def _fix_residual_adds(bld):
    # the decoder's `add` takes (conv arm, add_layer); keep explicit operands consistent
    for op in bld.ops:
        if op['kind'] == 'add' and op['a'] == op['b']:
            raise RuntimeError('degenerate residual add in ' + op['name'])


def _append_prior(bld, last, model_name, prefix, spatial, nb_labels, ndims, input_index, use_logp, final_pred_activation):
    """the layers of models.add_prior (models.py:378-436) appended to a graph under construction; returns the output name"""
    sp = (1,) * (3 - ndims) + tuple(spatial)
    prior = bld.add({'kind': 'input', 'name': '%s-input' % prefix, 'index': input_index}, (sp, nb_labels))
    if use_logp:                                                                 # :401-406
        print("Breaking change: use_logp option now requires log input!", file=sys.stderr)
        post = bld.add({'kind': 'add', 'name': '%s_posterior' % prefix, 'a': prior, 'b': last}, (sp, nb_labels))
    else:                                                                        # :408-414: sigmoid likelihood x prior
        like = bld.add({'kind': 'activation', 'name': '%s_likelihood_sigmoid' % prefix, 'src': last, 'activation': 'sigmoid'},
                       (sp, nb_labels))
        post = bld.add({'kind': 'multiply', 'name': '%s_posterior' % prefix, 'a': prior, 'b': like}, (sp, nb_labels))
    if final_pred_activation == 'softmax':
        assert use_logp, 'cannot do softmax when adding prior via P()'
        print("using final_pred_activation %s for %s" % (final_pred_activation, model_name))
        return bld.add({'kind': 'prediction', 'name': '%s_prediction' % prefix, 'src': post, 'activation': 'softmax', 'axis': -1},
                       (sp, nb_labels))
    return bld.add({'kind': 'prediction', 'name': '%s_prediction' % prefix, 'src': post, 'activation': 'linear'}, (sp, nb_labels))


def add_prior(input_model, prior_shape, name='prior_model', prefix=None, use_logp=True, final_pred_activation='softmax',
              add_prior_layer_reg=0):
    """
    Append the post-prior layers to a built network (models.py:378-436): a second input `prior` [B, *prior_shape] is added
    to (use_logp: log-prior + log-likelihood, then softmax) or multiplied with (prior x sigmoid(likelihood)) the model output.
    """
    if not isinstance(input_model, ConvNet):
        raise TypeError('add_prior expects a network built by unet / conv_enc / conv_dec')
    model_name = name
    if prefix is None:
        prefix = model_name
    prior_shape = tuple(int(s) for s in prior_shape)
    ndims = input_model.ndims
    out_sp, out_c = input_model._builder_state['shapes'][input_model.output_name]
    if tuple(out_sp[3 - ndims:]) != prior_shape[:-1] or out_c != prior_shape[-1]:
        raise ValueError('prior shape %s does not match the model output %s' % (prior_shape, tuple(out_sp[3 - ndims:]) + (out_c,)))
    bld = _Builder(ndims)
    bld.ops = list(input_model.ops)
    bld.modules = dict(input_model.layers_by_name.items())
    bld.shapes = dict(input_model._builder_state['shapes'])
    last = _append_prior(bld, input_model.output_name, model_name, prefix, prior_shape[:-1], prior_shape[-1], ndims,
                         len(input_model.input_shapes), use_logp, final_pred_activation)
    net = ConvNet(model_name, ndims, list(input_model.input_shapes) + [prior_shape], bld.ops, last, bld.modules)
    net._builder_state = dict(shapes=bld.shapes)
    return net


def dilation_net(nb_features, input_shape, nb_levels, conv_size, nb_labels, name='dilation_net', prefix=None, feat_mult=1,
                 pool_size=2, use_logp=True, padding='same', dilation_rate_mult=1, activation='elu', use_residuals=False,
                 final_pred_activation='softmax', nb_conv_per_level=1, add_prior_layer=False, add_prior_layer_reg=0,
                 layer_nb_feats=None, batch_norm=None):
    """models.py:45-85: as the reference, a unet that takes only `dilation_rate_mult` from these arguments (every other
    option is passed to unet at its default)."""
    return unet(nb_features, input_shape, nb_levels, conv_size, nb_labels, name='unet', prefix=None, feat_mult=1, pool_size=2,
                use_logp=True, padding='same', activation='elu', use_residuals=False, dilation_rate_mult=dilation_rate_mult,
                final_pred_activation='softmax', nb_conv_per_level=1, add_prior_layer=False, add_prior_layer_reg=0,
                layer_nb_feats=None, batch_norm=None)


@_store_config
def conv_enc(nb_features, input_shape, nb_levels, conv_size, name=None, prefix=None, feat_mult=1, pool_size=2,
             dilation_rate_mult=1, padding='same', activation='elu', layer_nb_feats=None, use_residuals=False,
             nb_conv_per_level=2, conv_dropout=0, batch_norm=None, convL=None, src=None, src_input=None):
    """Fully convolutional encoder (neurite/tf/models.py:1309-1442)."""
    if convL is not None or src is not None or src_input is not None:
        raise NotImplementedError('convL / src / src_input are Keras-graph arguments; pass input_shape (or a list of '
                                  'input shapes to unet) instead')
    model_name = name
    if prefix is None:
        prefix = model_name
    ndims = len(input_shape) - 1
    if ndims < 1 or ndims > 3:
        raise NotImplementedError('1-, 2- and 3-D networks are supported')
    input_shape = tuple(int(s) for s in input_shape)
    bld = _Builder(ndims)
    in_name = '%s_input' % prefix
    sp3 = (1,) * (3 - ndims) + input_shape[:-1]
    bld.add({'kind': 'input', 'name': in_name, 'index': 0}, (sp3, input_shape[-1]))
    last = _encoder(bld, nb_features, input_shape, nb_levels, conv_size, prefix, feat_mult, pool_size,
                    dilation_rate_mult, padding, activation, layer_nb_feats, use_residuals, nb_conv_per_level,
                    conv_dropout, batch_norm, in_name)
    net = ConvNet(model_name, ndims, [input_shape], bld.ops, last, bld.modules)
    net._builder_state = dict(shapes=bld.shapes, nb_conv_per_level=nb_conv_per_level, nb_levels=nb_levels,
                              nb_features=nb_features)
    return net


conv_block = conv_enc


@_store_config
def conv_dec(nb_features, input_shape, nb_levels, conv_size, nb_labels, name=None, prefix=None, feat_mult=1,
             pool_size=2, use_skip_connections=False, padding='same', dilation_rate_mult=1, activation='elu',
             use_residuals=False, final_pred_activation='softmax', nb_conv_per_level=2, layer_nb_feats=None,
             batch_norm=None, conv_dropout=0, convL=None, input_model=None):
    """Fully convolutional decoder (neurite/tf/models.py:1445-1617); with input_model = an encoder built by
    conv_enc and use_skip_connections it is the U-Net."""
    if convL is not None:
        raise NotImplementedError('convL is a Keras-graph argument')
    model_name = name
    if prefix is None:
        prefix = model_name
    if use_skip_connections:                                                     # :1479-1480
        assert input_model is not None, "is using skip connections, tensors dictionary is required"
    if input_model is None:
        ndims = len(input_shape) - 1
        input_shape = tuple(int(s) for s in input_shape)
        bld = _Builder(ndims)
        in_name = '%s_input' % prefix
        bld.add({'kind': 'input', 'name': in_name, 'index': 0}, ((1,) * (3 - ndims) + input_shape[:-1], input_shape[-1]))
        last = in_name
        input_shapes = [input_shape]
        enc_ncpl = nb_conv_per_level
    else:
        ndims = input_model.ndims
        bld = _Builder(ndims)
        bld.ops = list(input_model.ops)
        bld.modules = dict(input_model.layers_by_name.items())
        bld.shapes = dict(input_model._builder_state['shapes'])
        last = input_model.output_name
        input_shapes = input_model.input_shapes
        enc_ncpl = nb_conv_per_level
        if isinstance(nb_features, list):
            enc_ncpl = [len(f) if isinstance(f, list) else nb_conv_per_level for f in nb_features]
    last = _decoder(bld, nb_features, nb_levels, conv_size, nb_labels, prefix, feat_mult, pool_size,
                    use_skip_connections, padding, dilation_rate_mult, activation, use_residuals,
                    final_pred_activation, nb_conv_per_level, layer_nb_feats, batch_norm, conv_dropout, last, enc_ncpl)
    _fix_residual_adds(bld)
    net = ConvNet(model_name, ndims, input_shapes, bld.ops, last, bld.modules)
    net._builder_state = dict(shapes=bld.shapes)
    return net


@_store_config
def unet(nb_features, input_shape, nb_levels, conv_size, nb_labels, name='unet', prefix=None, feat_mult=1,
         pool_size=2, use_logp=True, padding='same', dilation_rate_mult=1, activation='elu', use_residuals=False,
         final_pred_activation='softmax', nb_conv_per_level=1, add_prior_layer=False, add_prior_layer_reg=0,
         layer_nb_feats=None, conv_dropout=0, batch_norm=None):
    """unet-style model with the reference's parametrisation (neurite/tf/models.py:88-246)."""
    model_name = name
    if prefix is None:
        prefix = model_name
    if add_prior_layer and not use_logp:                                         # models.py:423
        assert final_pred_activation != 'softmax', 'cannot do softmax when adding prior via P()'
    multi = isinstance(input_shape[0], (tuple, list, np.ndarray))
    if multi:                                                                    # :155-170
        shapes = [tuple(int(v) for v in s) for s in input_shape]
        for s in shapes:
            if not np.array_equal(s[:-1], shapes[0][:-1]):
                raise ValueError('spatial dimensions must match if multiple input shapes '
                                 'are provided, but got shapes '
                                 f'{shapes[0][:-1]} and {s[:-1]}')
        first = shapes[0]
    else:
        shapes = [tuple(int(v) for v in input_shape)]
        first = shapes[0]
    ndims = len(first) - 1
    if isinstance(nb_features, list):                                            # :179-190
        if nb_levels is not None:
            warnings.warn('nb_levels is not None while ' + 'nb_features list of lists specified - overriding')
        if feat_mult is not None:
            warnings.warn('feat_mult is not None while ' + 'nb_features list of lists specified - overriding')
        nb_levels = len(nb_features)
        assert isinstance(nb_features[0], list), \
            'nb_features must be a scalar or a list of lists (not a list of scalars)'

    if ndims < 1 or ndims > 3:
        raise NotImplementedError('1-, 2- and 3-D networks are supported')
    bld = _Builder(ndims)
    if multi:
        names = []
        for i, s in enumerate(shapes):
            n = f'{prefix}_input_{i}'
            bld.add({'kind': 'input', 'name': n, 'index': i}, ((1,) * (3 - ndims) + s[:-1], s[-1]))
            names.append(n)
        src = bld.add({'kind': 'input_concat', 'name': f'{prefix}_input_concat', 'src': names},
                      ((1,) * (3 - ndims) + first[:-1], sum(s[-1] for s in shapes)))
    else:
        src = bld.add({'kind': 'input', 'name': '%s_input' % prefix, 'index': 0},
                      ((1,) * (3 - ndims) + first[:-1], first[-1]))
    last = _encoder(bld, nb_features, first, nb_levels, conv_size, prefix, feat_mult, pool_size, dilation_rate_mult,
                    padding, activation, layer_nb_feats, use_residuals, nb_conv_per_level, conv_dropout, batch_norm, src)
    lnf = layer_nb_feats[(nb_levels * nb_conv_per_level):] if layer_nb_feats is not None else None      # :214
    enc_ncpl = nb_conv_per_level
    if isinstance(nb_features, list):
        enc_ncpl = [len(f) for f in nb_features]
    last = _decoder(bld, nb_features, nb_levels, conv_size, nb_labels, prefix, feat_mult, pool_size, 1, padding,
                    dilation_rate_mult, activation, use_residuals, 'linear' if add_prior_layer else final_pred_activation,
                    nb_conv_per_level, lnf, batch_norm, conv_dropout, last, enc_ncpl)
    _fix_residual_adds(bld)
    if add_prior_layer:                                                          # models.add_prior, :378-436
        pname = model_name + '_prior'
        last = _append_prior(bld, last, pname, pname, tuple(first[:-1]), int(nb_labels), ndims, len(shapes), use_logp,
                             final_pred_activation)
        shapes = shapes + [tuple(first[:-1]) + (int(nb_labels),)]
        model_name = pname
    net = ConvNet(model_name, ndims, shapes, bld.ops, last, bld.modules)
    net._builder_state = dict(shapes=bld.shapes)
    return net


def load_config(path):
    """builder name and arguments stored by `ConvNet.save` (LoadableModel.load_config, neurite/tf/modelio.py:125-143)"""
    if _is_h5(path):
        h5py = _h5py()
        with h5py.File(path, 'r') as f:
            cfg = f.attrs['model_config']
    else:
        with np.load(path) as z:
            if '__model_config__' not in z.files:
                raise ValueError('%s holds weights only (written by save_weights); build the model and call load_weights' % path)
            cfg = z['__model_config__'].item()
    cfg = json.loads(_h5_str(cfg))
    return cfg['class_name'], cfg['config']


def load(path, by_name=False, **kwargs):
    """
    LoadableModel.load (neurite/tf/modelio.py:111-123): rebuild the network from the arguments saved with it (keyword
    arguments override them), then load its weights.
    """
    builder, config = load_config(path)
    builders = {'unet': unet, 'conv_enc': conv_enc, 'conv_dec': conv_dec}
    if builder not in builders:
        raise ValueError('%s was not saved from a unet / conv_enc / conv_dec network (class_name %r)' % (path, builder))
    config.update(kwargs)
    metadata = config.pop('metadata', {})
    model = builders[builder](**config)
    model.metadata.update(metadata)
    model.load_weights(path, by_name=by_name)
    return model


def labels_to_image(*args, **kwargs):
    """neurite/tf/models.py:649-918; implemented in neurite_amd.synthesis."""
    from . import synthesis
    return synthesis.labels_to_image(*args, **kwargs)


def labels_to_image_new(*args, **kwargs):
    """neurite/tf/models.py:920-1300; implemented in neurite_amd.synthesis."""
    from . import synthesis
    return synthesis.labels_to_image_new(*args, **kwargs)


class SynthStrip(nn.Module):
    """
    SynthStrip (neurite/tf/models.py:1888-1967): a label map is turned into a synthetic image by `labels_to_image`
    (integer output labels), a unet with one linear output channel is applied to it, and the model returns the unet output
    concatenated with the warped label map, [B, *inshape, 2], for a brain / non-brain loss.  Constructor arguments are
    recorded as `modelio.store_config_args` does (`get_config`, `save`, `SynthStrip.load`).
    """

    def __init__(self, inshape, labels_in, labels_out, nb_unet_features=None, nb_unet_levels=None, unet_feat_mult=1,
                 nb_unet_conv_per_level=1, src_feats=1, gen_args={}):
        super().__init__()
        ndims = len(inshape)
        assert ndims in [1, 2, 3], 'ndims should be one of 1, 2, or 3. found: %d' % ndims
        inshape = tuple(int(s) for s in inshape)
        self.config = {'builder': 'SynthStrip', 'loadable': True,
                       'params': dict(inshape=list(inshape), labels_in=_jsonable(labels_in), labels_out=_jsonable(labels_out),
                                      nb_unet_features=_jsonable(nb_unet_features), nb_unet_levels=nb_unet_levels,
                                      unet_feat_mult=unet_feat_mult, nb_unet_conv_per_level=nb_unet_conv_per_level,
                                      src_feats=src_feats, gen_args=_jsonable(dict(gen_args)), metadata={})}
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')                       # the reference builds the deprecated generator here too
            self.gen_model = labels_to_image(inshape, labels_in, labels_out, id=0, return_def=False, one_hot=False, **gen_args)
        self.unet = unet(nb_unet_features, (*inshape, 1), nb_unet_levels, ndims * (3,), 1, feat_mult=unet_feat_mult,
                         nb_conv_per_level=nb_unet_conv_per_level, final_pred_activation='linear')
        self.name = 'synthstrip'

    def get_config(self):
        return self.config['params']

    @property
    def metadata(self):
        return self.config['params']['metadata']

    def get_strip_model(self):
        """the stripping model (just the U-Net)"""
        return self.unet

    def train(self, mode=True):
        super().train(mode)
        self.unet.train(mode)                                    # ConvNet defaults to eval(); follow the wrapper's mode
        return self

    def forward(self, labels):
        synth_image, synth_labels = self.gen_model(labels)[:2]
        self.synth_image = synth_image
        out = self.unet(synth_image)
        return torch.cat([out, synth_labels.to(torch.float32)], -1)

    def save(self, path):
        """unet weights + constructor arguments (npz), reloadable with SynthStrip.load"""
        cfg = json.dumps({'class_name': 'SynthStrip', 'config': self.get_config()})
        np.savez(path, __model_config__=np.array(cfg), **self.unet._npz_dict())

    @classmethod
    def load(cls, path, **kwargs):
        builder, config = load_config(path)
        if builder != 'SynthStrip':
            raise ValueError('%s was not saved from a SynthStrip model (class_name %r)' % (path, builder))
        config.update(kwargs)
        metadata = config.pop('metadata', {})
        for key in ('labels_in', 'labels_out'):               # JSON turned integer label keys into strings
            if isinstance(config.get(key), dict):
                config[key] = {(int(k) if isinstance(k, str) and k.lstrip('-').isdigit() else k): v for k, v in config[key].items()}
        model = cls(**config)
        model.metadata.update(metadata)
        model.unet.load_weights(path)
        return model
