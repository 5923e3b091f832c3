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


# --------------------------------------------------------------------------------------
# full-resolution Perlin noise, random blur, crop masks (neurite/tf/utils/augment.py:66-330) -- the augmentation
# primitives behind layers.PerlinNoise / RandomCrop and models.labels_to_image_new
# --------------------------------------------------------------------------------------

def normalize_axes(axes, shape, allowed=None, none_means_all=False):
    """neurite/py/utils.py:124-167: sorted, de-duplicated axes in [0, N); IndexError outside `allowed`."""
    ndims = len(shape)
    if allowed is None:
        allowed = range(ndims)
    if np.isscalar(allowed):
        allowed = [allowed]
    allowed = list(allowed)
    assert all(ax in range(ndims) for ax in allowed), f'allowed axes {allowed} out of bounds'
    if axes is None:
        axes = allowed if none_means_all else []
    if np.isscalar(axes):
        axes = [axes]
    orig = list(axes)
    axes = [ax + ndims if ax < 0 else ax for ax in orig]
    for ax, inp in zip(axes, orig):
        if ax not in allowed:
            raise IndexError(f'axis {inp} outside {allowed}')
    return tuple(set(axes))


def _reduce_name(reduce):
    """the global statistic kept constant by random_blur_rescale: the reference passes tf.math.reduce_std / reduce_max"""
    if isinstance(reduce, str):
        name = reduce
    else:
        name = getattr(reduce, '__name__', str(reduce))
    name = name.lower().replace('reduce_', '')
    if name in ('std', 'stddev'):
        return 'std'
    if name in ('max', 'amax'):
        return 'max'
    raise NotImplementedError('neurite_amd: reduce must be the standard deviation or the maximum (got %r)' % (reduce,))


def _global_stat(x, name):
    """0-d device tensor: population SD or maximum over all elements (device reductions of csrc, no host sync)"""
    if name == 'max':
        return utils._device_minmax(x)[1]
    lib = _lib.lib()
    dev = x.device
    s = torch.zeros(2, dtype=torch.float32, device=dev)
    flat = x.contiguous()
    with torch.cuda.device(dev):
        rc = lib.nrt_channel_sums_f32(_lib.ptr(flat), None, flat.numel(), 1, _lib.ptr(s[:1]), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_channel_sums_f32')
        rc = lib.nrt_channel_sums_f32(_lib.ptr(flat), _lib.ptr(flat), flat.numel(), 1, _lib.ptr(s[1:]), _lib.stream_ptr(dev))
        _lib.check(rc, 'nrt_channel_sums_f32')
    n = float(flat.numel())
    mean = s[0] / n
    return torch.sqrt(torch.clamp(s[1] / n - mean * mean, min=0.0))


def random_blur_rescale(x, std_min=8 / 2.355, std_max=32 / 2.355, isotropic=False, seed=None, reduce='std', batched=False):
    """
    Smooth the spatial dimensions of a tensor (trailing feature dimension) with a random Gaussian kernel per axis and rescale
    it such that a global statistic (`reduce`: 'std' or 'max', or tf/torch functions of those names) is unchanged
    (augment.py:66-113).
    """
    _lib.require_device(x)
    n_dim = x.dim() - 1 - int(batched)
    rand = np.random.default_rng(seed)
    seeds = rand.integers(np.iinfo(int).max, size=n_dim)
    kernel = [utils.gaussian_kernel(sigma=std_max, separate=True, random=True, min_sigma=std_min, dtype=x.dtype, seed=int(s))
              for s in seeds]
    if isotropic:
        kernel = kernel[:1] * n_dim
    name = _reduce_name(reduce)
    before = _global_stat(x, name)
    y = utils.separable_conv(x, kernel, batched=batched)
    after = _global_stat(y, name)
    ratio = torch.where(after != 0, before / torch.where(after != 0, after, torch.ones_like(after)), torch.zeros_like(after))
    return y * ratio, kernel                                    # tf.math.divide_no_nan


def draw_perlin_full(shape, noise_min=0.01, noise_max=1, fwhm_min=4, fwhm_max=32, isotropic=False, batched=False,
                     featured=False, reduce='std', dtype=torch.float32, axes=None, seed=None, device=None):
    """
    Perlin noise without interpolation (augment.py:116-215): per level, normal noise with a uniformly drawn SD (one SD per
    entry of `axes`) is drawn at full resolution, blurred with random anisotropic Gaussians whose FWHM lies between the
    level's bounds, rescaled to keep `reduce` constant; the levels are averaged.
    """
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != 'cuda':
        raise _lib.NeuriteAmdError('neurite_amd runs on a ROCm device; there is no CPU fallback')
    device = torch.device(device)
    assert 0 < noise_min <= noise_max, f'invalid noise-SD bounds {(noise_min, noise_max)}'
    rand = np.random.default_rng(seed)

    def draw_seed():
        return int(rand.integers(2 ** 31 - 1))
    shape = [int(s) for s in np.ravel(shape)]
    axes = list(normalize_axes(axes, shape, none_means_all=False))
    if not batched:
        shape = [1] + shape
        axes = [ax + 1 for ax in axes]
    if not featured:
        shape = shape + [1]
    shape_sd = [shape[i] if i in axes else 1 for i in range(len(shape))]
    if not hasattr(fwhm_min, '__iter__'):
        fwhm_min = [fwhm_min]
    if not hasattr(fwhm_max, '__iter__'):
        fwhm_max = [fwhm_max]
    assert len(fwhm_min) == len(fwhm_max), 'different number of lower and upper bounds'
    gen = torch.Generator(device=device)
    out = None
    for low, upp in zip(fwhm_min, fwhm_max):
        gen.manual_seed(draw_seed())
        sd = noise_min + (noise_max - noise_min) * torch.rand(shape_sd, generator=gen, device=device, dtype=dtype)
        gen.manual_seed(draw_seed())
        noise = torch.randn(shape, generator=gen, device=device, dtype=dtype) * sd
        noise, _ = random_blur_rescale(noise, std_min=low / 2.355, std_max=upp / 2.355, batched=True, isotropic=isotropic,
                                       seed=draw_seed(), reduce=reduce)
        out = noise if out is None else out + noise
    out = out / float(len(fwhm_min))
    if not batched:
        out = out[0]
    if not featured:
        out = out[..., 0]
    return out


def draw_crop_mask(x, crop_min=0, crop_max=0.5, axis=None, prob=1, bilateral=False, seed=None, _draws=None):
    """
    Mask that multiplicatively crops the field of view of an N-D tensor along one (randomly drawn) axis
    (augment.py:218-290): float32 tensor on x's device with singleton dimensions except along that axis.
    `_draws` (tests): the uniform [0, 1) numbers to use instead of the generator's, in the reference's order of draws.
    """
    shape = tuple(x.shape)
    axis = list(normalize_axes(axis, shape, none_means_all=True))
    assert 0 <= crop_min <= crop_max <= 1, f'invalid proportions {crop_min}, {crop_max}'
    gen = utils._host_generator(seed)
    draws = None if _draws is None else list(_draws)

    def uniform():
        return np.float32(draws.pop(0)) if draws is not None else np.float32(float(torch.rand((), generator=gen)))
    prop_cut = np.float32(crop_max)
    if crop_min < crop_max:                                   # tf.random.uniform: u * (maxval - minval) + minval in float32
        prop_cut = uniform() * np.float32(np.float32(crop_max) - np.float32(crop_min)) + np.float32(crop_min)
    assert 0 <= prob <= 1, f'{prob} not a probability'
    if prob < 1:
        prop_cut = prop_cut * np.float32(uniform() < np.float32(prob))
    rand_prop = uniform()
    if not bilateral:
        rand_prop = np.float32(rand_prop < np.float32(0.5))
    prop_low = prop_cut * rand_prop
    prop_cen = np.float32(1) - prop_cut
    ax = axis[int(np.floor(np.float64(uniform()) * len(axis)))]
    width = shape[ax]
    prop = (np.arange(width, dtype=np.float32) * (np.float32(1) / np.float32(width))).astype(np.float32)   # tf.range(1, delta=1/width)
    mask = np.logical_and(prop >= prop_low, prop < prop_low + prop_cen).astype(np.float32)
    out_shape = [1] * len(shape)
    out_shape[ax] = width
    return torch.from_numpy(mask).reshape(out_shape).to(x.device)
