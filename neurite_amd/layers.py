"""
neurite_amd.layers -- the layers of neurite's hot path as torch.nn.Modules over the HIP kernels.

Resize / Zoom           neurite/tf/layers.py:91-185
SpatialTransformer      voxelmorph.layers.SpatialTransformer as the reference calls it
                        (neurite/tf/models.py:806-807 and 1157-1159; not vendored in the reference)

Same constructor arguments, defaults, lazy build on first call, get_config() keys and error
behaviour as the Keras layers.  Tensors are channels-last [B, *spatial, C] on a ROCm device.
The batch loop the reference runs serially with tf.map_fn (layers.py:171) is one batched kernel
launch here (blockIdx.y = batch entry).
"""

import torch
from torch import nn

from . import _lib
from . import utils

__all__ = ['Resize', 'Zoom', 'SpatialTransformer']


class _Layer(nn.Module):
    """The slice of the Keras Layer protocol the reference's users touch."""

    def __init__(self, name=None, **kwargs):
        super().__init__()
        kwargs.pop('dtype', None)
        kwargs.pop('trainable', None)
        kwargs.pop('input_shape', None)
        if kwargs:
            raise TypeError('unexpected keyword arguments: %s' % sorted(kwargs))
        self._name = name or self.__class__.__name__.lower()
        self.built = False

    @property
    def name(self):
        return self._name

    def build(self, input_shape):
        self.built = True

    def get_config(self):
        return {'name': self._name}

    def _maybe_build(self, inputs):
        if not self.built:
            if isinstance(inputs, (list, tuple)):
                self.build([tuple(i.shape) for i in inputs])
            else:
                self.build(tuple(inputs.shape))
            self.built = True

    def forward(self, inputs, **kwargs):
        self._maybe_build(inputs)
        return self.call(inputs, **kwargs)


class Resize(_Layer):
    """
    N-D resize layer (neurite/tf/layers.py:91-181): align-corners linear (or nearest) zoom of
    every batch entry, e.g. the 2x deformation-field upsample at neurite/tf/models.py:804.
    """

    def __init__(self, zoom_factor, interp_method='linear', **kwargs):
        super().__init__(**kwargs)
        self.zoom_factor = zoom_factor
        self.interp_method = interp_method
        self.ndims = None
        self.inshape = None

    def get_config(self):
        config = super().get_config().copy()
        config.update({'zoom_factor': self.zoom_factor, 'interp_method': self.interp_method})
        return config

    def build(self, input_shape):
        if isinstance(input_shape[0], (list, tuple)) and len(input_shape) > 1:       # layers.py:133-134
            raise Exception('Resize must be called on a list of length 1.')
        if isinstance(input_shape[0], (list, tuple)):
            input_shape = input_shape[0]
        self.ndims = len(input_shape) - 2                                             # :140
        self.inshape = input_shape
        if not isinstance(self.zoom_factor, (list, tuple)):                           # :142-147
            self.zoom_factor = [self.zoom_factor] * self.ndims
        else:
            assert len(self.zoom_factor) == self.ndims, \
                'zoom factor length {} does not match number of dimensions {}'.format(
                    len(self.zoom_factor), self.ndims)
        self.built = True

    def compute_output_shape(self, input_shape):
        output_shape = [input_shape[0]]
        output_shape += [int(input_shape[1:-1][f] * self.zoom_factor[f]) for f in range(self.ndims)]
        output_shape += [input_shape[-1]]
        return tuple(output_shape)

    def call(self, inputs):
        if isinstance(inputs, (list, tuple)):                                         # :161-165
            assert len(inputs) == 1, "inputs has to be len 1. found: %d" % len(inputs)
            vol = inputs[0]
        else:
            vol = inputs
        _lib.require_device(vol)
        vol = vol.reshape([-1, *self.inshape[1:]])                                    # :168
        zf = list(self.zoom_factor)
        if all(z == 1 for z in zf):                                                   # utils.py:250-251
            return vol
        if self.interp_method != 'linear':
            assert self.interp_method == 'nearest', \
                'method should be linear or nearest, got: %s' % self.interp_method
        new_shape = utils._new_shape(list(vol.shape[1:-1]), zf)
        vol32, restore = utils._prepare_vol(vol, self.interp_method)

        def run():
            out = utils._launch_interpn(vol32, None, new_shape, _lib.LOC_LINSPACE,
                                        utils._METHODS[self.interp_method], None, batched=True)
            return out if restore is None else out.to(restore)

        return utils._maybe_tracked(run, vol)


Zoom = Resize


class SpatialTransformer(_Layer):
    """
    N-D spatial transformer: out[b, x, :] = interpn(vol[b], x + trf[b, x, :]) (pull-back warp with a
    dense displacement field in voxel units), or an affine [B, D, D+1] first converted to a dense
    shift.  Accepts the union of voxelmorph's historical constructor arguments.

    call([vol, trf]):  vol [B, *S, C], trf [B, *S', D]  ->  [B, *S', C]
    """

    def __init__(self, interp_method='linear', indexing='ij', single_transform=False, fill_value=None,
                 shift_center=True, add_identity=True, shape=None, **kwargs):
        super().__init__(**kwargs)
        self.interp_method = interp_method
        assert indexing in ['ij', 'xy'], "indexing has to be 'ij' (matrix) or 'xy' (cartesian)"
        self.indexing = indexing
        self.single_transform = single_transform
        self.fill_value = fill_value
        self.shift_center = shift_center
        self.add_identity = add_identity
        self.shape = shape
        self.ndims = None
        self._variant, self._tune = 0, 0      # kernel selection override (tuning / tests); 0 = auto

    def get_config(self):
        config = super().get_config().copy()
        config.update({
            'interp_method': self.interp_method,
            'indexing': self.indexing,
            'single_transform': self.single_transform,
            'fill_value': self.fill_value,
            'shift_center': self.shift_center,
            'add_identity': self.add_identity,
            'shape': self.shape,
        })
        return config

    def build(self, input_shape):
        if len(input_shape) > 2:
            raise Exception('Spatial Transformer must be called on a list of length 2.')
        self.ndims = len(input_shape[0]) - 2
        self.built = True

    def call(self, inputs):
        assert len(inputs) == 2, 'inputs has to be len 2, found: %d' % len(inputs)
        vol, trf = inputs
        _lib.require_device(vol, trf)
        D = vol.dim() - 2
        if D < 1 or D > 3:
            raise NotImplementedError('SpatialTransformer supports 1-, 2- and 3-D volumes')
        if self.interp_method != 'linear':
            assert self.interp_method == 'nearest', \
                'method should be linear or nearest, got: %s' % self.interp_method
        if not self.add_identity:
            raise NotImplementedError('add_identity=False (absolute coordinates): call utils.interpn directly')
        B = vol.shape[0]

        # affine [B, D, D+1] or [B, D+1, D+1] -> dense shift (a dense 1-D flow [B, X, 1] also has rank 3)
        if trf.dim() == 3 and trf.shape[-1] == D + 1 and trf.shape[-2] in (D, D + 1):
            out_spatial = list(self.shape) if self.shape is not None else list(vol.shape[1:-1])
            nb = 1 if self.single_transform else trf.shape[0]
            trf = torch.stack([utils.affine_to_dense_shift(trf[b], out_spatial, shift_center=self.shift_center,
                                                           indexing=self.indexing).to(vol.device)
                               for b in range(nb)], 0)
        else:
            if trf.dim() != D + 2 or trf.shape[-1] != D:
                raise Exception("Number of loc Tensors %d does not match volume dimension %d"
                                % (trf.shape[-1], D))
            trf = trf.to(torch.float32)
            if self.indexing == 'xy' and D > 1:
                # cartesian flows carry (x, y, ...) = (col, row, ...): swap the first two components
                trf = torch.cat([trf[..., 1:2], trf[..., 0:1], trf[..., 2:]], -1)
        if not self.single_transform and trf.shape[0] != B:
            raise ValueError('batch size of the transform (%d) does not match the volume (%d)'
                             % (trf.shape[0], B))
        shift = trf[:1] if self.single_transform else trf
        vol32, restore = utils._prepare_vol(vol, self.interp_method)

        def run():
            out = utils._launch_interpn(vol32, shift, shift.shape[1:-1], _lib.LOC_SHIFT,
                                        utils._METHODS[self.interp_method], self.fill_value, batched=True,
                                        single_transform=self.single_transform, variant=self._variant,
                                        tune=self._tune)
            return out if restore is None else out.to(restore)

        return utils._maybe_tracked(run, vol, trf)
