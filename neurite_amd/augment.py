"""
neurite_amd.augment -- neurite/tf/utils/augment.py:7-62 (`draw_perlin`) on the HIP resize kernel.
"""

import numpy as np
import torch

from . import _lib
from . import utils

__all__ = ['draw_perlin']


def draw_perlin(out_shape, scales, min_std=0, max_std=1, dtype=torch.float32, seed=None, device=None):
    """
    Perlin-like noise: normal noise drawn at several resolutions, up-sampled (`utils.resize`, align-corners linear) and
    summed.  out_shape: [*spatial, features]; a scale of 2 means half resolution.  The SD of every level is drawn
    uniformly from [min_std, max_std).  Random numbers come from torch's generator (the reference uses tf.random: same
    distributions, a different stream -- there is no value parity for stochastic functions).
    """
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != 'cuda':
        raise _lib.NeuriteAmdError('neurite_amd runs on a ROCm device; there is no CPU fallback')
    device = torch.device(device)
    out_shape = np.asarray(out_shape, dtype=np.int32)
    if np.isscalar(scales):
        scales = [scales]
    rand = np.random.default_rng(seed)
    gen = torch.Generator(device=device)
    out = torch.zeros(tuple(int(s) for s in out_shape), dtype=dtype, device=device)
    for scale in scales:
        sample_shape = np.ceil(out_shape[:-1] / scale)
        sample_shape = np.int32((*sample_shape, out_shape[-1]))
        gen.manual_seed(int(rand.integers(2 ** 31 - 1)))
        std = min_std + (max_std - min_std) * float(torch.rand((), generator=gen, device=device))
        gen.manual_seed(int(rand.integers(2 ** 31 - 1)))
        gauss = torch.randn(tuple(int(s) for s in sample_shape), generator=gen, device=device, dtype=dtype) * std
        if scale == 1:
            up = gauss
        else:
            zoom = [o / s for o, s in zip(out_shape, sample_shape)]
            up = utils.resize(gauss, zoom[:-1])
            if tuple(up.shape) != tuple(out.shape):
                raise ValueError('Incompatible shapes: %s vs. %s (resize of a %s sample by %s)'
                                 % (tuple(out.shape), tuple(up.shape), tuple(sample_shape), zoom[:-1]))
        out = out + up
    return out
