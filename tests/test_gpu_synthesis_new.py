"""
GPU tests of labels_to_image_new (neurite_amd.synthesis; neurite/tf/models.py:920-1300) and of the augmentation layers it
instantiates (layers.Subsample / RandomCrop / GaussianNoise / PerlinNoise).  Random numbers come from torch's generators, so
value parity with TensorFlow's streams does not exist; the deterministic functions of the recorded draws are checked against
NumPy restatements, the stochastic parts through their invariants.
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import augment, synth, utils
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def N(t):
    return None if t is None else t.detach().cpu().numpy()


def label_map(dev, B, S, labels, seed=1):
    lab = torch.stack([synth.blob_labels(seed + b, size=S, nb_labels=len(labels), coarse=4, device=dev) for b in range(B)], 0)
    lut = torch.tensor(labels, device=dev)
    return lut[lab.long()][..., None].to(torch.float32)


def test_subsample_crop_noise_layers(dev):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 12, 9, 10, 3)).astype(F)
    xg = torch.from_numpy(x).to(dev)
    # Subsample: the composed index map of the reference's two gathers
    for ax, thick in ((1, 2.6), (2, 1.0), (3, 4.2)):
        down, up = utils.subsample_axis_indices(x.shape[ax], thick)
        got = N(utils._axis_gather(xg, down[up], ax))
        assert np.array_equal(got, np.take(np.take(x, down, axis=ax), up, axis=ax))
    lay = ne.layers.Subsample(stride_min=2, stride_max=4, axes=(1, 3), seed=4)
    y = N(lay(xg))
    assert y.shape == x.shape
    def slices_from(ax):                         # every slice of y along ax is some slice of x along ax
        return all(any(np.array_equal(np.take(y, i, axis=ax), np.take(x, j, axis=ax)) for j in range(x.shape[ax]))
                   for i in range(x.shape[ax]))
    assert not np.array_equal(y, x) and (slices_from(1) or slices_from(3)) and not slices_from(2)
    assert ne.layers.Subsample(prob=0)(xg) is xg and ne.layers.Subsample(stride_max=1)(xg) is xg
    nd = ne.layers.Subsample(stride_min=3, stride_max=3, axes=2, upsample=False, seed=1)(xg)
    assert nd.shape == (2, 12, 3, 10, 3)
    # RandomCrop: multiplication with the drawn mask
    crop = ne.layers.RandomCrop(crop_min=0.25, crop_max=0.5, axis=(1, 2), seed=7)
    y = N(crop(xg))
    assert np.array_equal(y, x * N(crop.last_mask))
    assert 0 < N(crop.last_mask).sum() < max(crop.last_mask.shape)
    assert ne.layers.RandomCrop(prob=0)(xg) is xg
    # GaussianNoise: x + sd * noise with one SD per (batch entry, channel), relative to max |x|
    gn = ne.layers.GaussianNoise(noise_min=0.1, noise_max=0.2, seed=5)
    y = N(gn(xg))
    sd, noise = N(gn.last_draws['sd']), N(gn.last_draws['noise'])
    assert sd.shape == (2, 3) and (sd >= 0.1 * np.abs(x).max() - 1e-6).all() and (sd <= 0.2 * np.abs(x).max() + 1e-6).all()
    np.testing.assert_allclose(y, x + sd[:, None, None, None, :] * noise, rtol=1e-6, atol=1e-6)
    only = ne.layers.GaussianNoise(noise_min=1, noise_max=1, noise_only=True, absolute=True, axes=0, seed=6)
    z = N(only(xg))
    np.testing.assert_allclose(z, N(only.last_draws['noise']), rtol=1e-6)
    assert abs(z.std() - 1) < 0.05
    assert ne.layers.GaussianNoise(noise_min=0, noise_max=0)(xg) is xg


def test_perlin_noise_layer(dev):
    x = torch.zeros(2, 24, 20, 16, 2, device=dev)
    lay = ne.layers.PerlinNoise(noise_min=0.5, noise_max=1.0, fwhm_min=(2, 6), fwhm_max=(4, 10), reduce='max', axes=-1, seed=11)
    a = N(lay(x))
    assert a.shape == (2, 24, 20, 16, 2) and np.isfinite(a).all() and a.std() > 0
    assert not np.array_equal(a[0], a[1])                                          # a fresh draw per batch entry
    b = N(ne.layers.PerlinNoise(noise_min=0.5, noise_max=1.0, fwhm_min=(2, 6), fwhm_max=(4, 10), reduce='max', axes=-1, seed=11)(x))
    assert np.array_equal(a, b)                                                    # seeded layers reproduce
    # smooth: neighbouring voxels correlate strongly, white noise would not
    c = np.corrcoef(a[0, :-1, :, :, 0].ravel(), a[0, 1:, :, :, 0].ravel())[0, 1]
    assert c > 0.5
    v = N(ne.layers.PerlinNoise(shape=(12, 10, 8, 3), noise_min=1, noise_max=1, fwhm_min=3, fwhm_max=3.0001, seed=2)(x))
    assert v.shape == (2, 12, 10, 8, 3)
    # random_blur_rescale keeps the chosen statistic
    t = torch.randn(1, 20, 18, 16, 2, device=dev)
    for stat in ('std', 'max'):
        y, _ = augment.random_blur_rescale(t, std_min=1.0, std_max=2.0, seed=3, reduce=stat, batched=True)
        want = t.std(unbiased=False) if stat == 'std' else t.max()
        got = y.std(unbiased=False) if stat == 'std' else y.max()
        assert abs(float(got) - float(want)) < 2e-4 * abs(float(want))
    with pytest.raises(NotImplementedError):
        augment.random_blur_rescale(t, reduce='median', batched=True)
    with pytest.raises(AssertionError, match='invalid noise-SD'):
        augment.draw_perlin_full((8, 8, 8), noise_min=0, device=dev)


def test_labels_to_image_new_identity_geometry(dev):
    """no spatial augmentation, no corruption: labels pass through the look-up tables, the image is the mean look-up"""
    labels = [0, 2, 3, 41, 42]
    S, B = 16, 2
    lab = label_map(dev, B, S, labels)
    model = ne.models.labels_to_image_new({0: 0, 2: 'l', 3: 'r', 41: 'l', 42: 'r'}, labels_out={2: 1, 41: 1, 3: 2, 42: 2},
                                          in_shape=(S, S, S), num_chan=2, warp_max=0, bias_max=0, noise_min=0, noise_max=0,
                                          blur_max=0, normalize=False, gamma=0, return_mean=True, seeds=dict(mean=3))
    image, oh, mean_img = model(lab)
    d = model.last_draws
    assert np.array_equal(N(d['labels']), N(lab))                                   # identity affine, nearest sampling
    assert image.shape == (B, S, S, S, 2) and oh.shape == (B, S, S, S, 2)
    raw = N(lab)[..., 0].astype(int)
    want_oh = np.stack([np.isin(raw, [2, 41]), np.isin(raw, [3, 42])], -1).astype(F)
    assert np.array_equal(N(oh), want_oh)                                           # background dropped from the one-hot maps
    gen = model.cfg['gen_lut'].astype(int)[raw]                                     # [B, S, S, S]
    mean = N(d['mean'])                                                             # [B, C, L]
    want = np.stack([np.stack([mean[b, c][gen[b]] for c in range(2)], -1) for b in range(B)], 0)
    assert np.array_equal(N(image), want) and np.array_equal(N(mean_img), want)
    assert np.array_equal(want[raw == 2], want[raw == 2]) and mean.min() >= 0 and mean.max() <= 1
    same_lr = gen[raw == 2][0] == gen[raw == 41][0]
    assert same_lr                                                                   # left / right share a generation label
    # integer output labels
    m2 = ne.models.labels_to_image_new(labels, labels_out={41: 7, 42: 7}, in_shape=(S, S, S), warp_max=0, one_hot=False,
                                       return_im=False)
    out = N(m2(lab))
    assert out.dtype == np.int32 and np.array_equal(out[..., 0], np.where(np.isin(raw, [41, 42]), 7, 0))


def test_labels_to_image_new_full_pipeline(dev):
    """every stage on; the deterministic functions of the recorded draws are recomputed in NumPy"""
    labels = list(range(6))
    S, B = 32, 2
    lab = label_map(dev, B, S, labels, seed=5)
    model = ne.models.labels_to_image_new(labels, in_shape=(S, S, S), num_chan=2, aff_shift=3, aff_rotate=10, aff_scale=0.1,
                                          aff_shear=0.05, axes_flip=True, warp_min=0.5, warp_max=2, warp_blur_min=(4, 8),
                                          warp_blur_max=(8, 16), crop_min=0.1, crop_max=0.3, crop_prob=1, zero_background=1.0,
                                          bias_blur_min=8, bias_blur_max=16, blur_min=0.5, blur_max=1.5, gamma=0.5,
                                          return_vel=True, return_def=True, return_aff=True, return_mean=True, return_bias=True,
                                          seeds=dict(warp=3, mean=4, noise=5, bias=6, blur=7, gamma=8, shift=9, rot=10, crop=11))
    image, oh, vel, dfield, aff, mean_img, bias = model(lab)
    d = model.last_draws
    assert image.shape == (B, S, S, S, 2) and oh.shape == (B, S, S, S, 6) and vel.shape == (B, S // 2, S // 2, S // 2, 3)
    assert dfield.shape == (B, S, S, S, 3) and aff.shape == (B, 4, 4) and bias.shape == image.shape
    for t in (image, oh, vel, dfield, aff, mean_img, bias):
        assert np.isfinite(N(t)).all()
    # labels: nearest-neighbour warp of the input by the recorded dense transform, then the crop mask
    warped = npo.spatial_transformer(N(lab), N(d['trans']), 'nearest', fill_value=0)
    assert np.array_equal(N(d['labels']), warped * N(model.crop.last_mask))
    raw = N(d['labels'])[..., 0].astype(int)
    assert np.array_equal(N(oh).argmax(-1)[N(oh).sum(-1) > 0], raw[N(oh).sum(-1) > 0]) and np.all(N(oh).sum(-1) == 1)
    # the deformation is VecInt(5) of the SVF, rescaled to full resolution
    want_def = np.stack([npo.rescale_dense_transform(npo.integrate_vec(N(vel)[b], 'ss', 5), 2) for b in range(B)], 0)
    np.testing.assert_allclose(N(dfield), want_def, rtol=2e-4, atol=2e-4)
    # image: ((mean look-up * exp(bias) + sd * noise) * background mask) -> blur -> min-max -> gamma
    mean = N(d['mean'])
    img = np.stack([np.stack([mean[b, c][raw[b]] for c in range(2)], -1) for b in range(B)], 0)
    assert np.array_equal(N(mean_img), img)
    np.testing.assert_allclose(N(bias), np.exp(N(d['bias_field'])), rtol=1e-6)
    img = img * np.exp(N(d['bias_field']))
    img = img + N(d['noise_sd'])[:, None, None, None, :] * N(d['noise'])
    assert N(d['bg_zero']).tolist() == [1.0, 1.0]
    img = img * (raw != 0)[..., None]
    img = npo.separable_conv(img.astype(F), [N(k) for k in d['blur_kernels']], batched=True)
    np.testing.assert_allclose(N(d['pre_norm']), img, rtol=2e-4, atol=2e-5)
    img = np.stack([npo.minmax_norm(img[b]) for b in range(B)], 0)
    img = np.power(img, N(d['gamma'])[:, None, None, None, :])
    np.testing.assert_allclose(N(image), img, rtol=2e-4, atol=2e-5)
    assert N(image).min() >= 0 and N(image).max() <= 1 + 1e-6
    g = N(d['gamma'])
    assert (g >= 0.5).all() and (g <= 1.5).all()
    # seeded components reproduce over model instances; unseeded ones (scale, shear, flip) need not
    again = ne.models.labels_to_image_new(labels, in_shape=(S, S, S), num_chan=2, warp_min=0.5, warp_max=2, warp_blur_min=(4, 8),
                                          warp_blur_max=(8, 16), return_im=False, return_map=False, return_vel=True,
                                          seeds=dict(warp=3))
    assert np.array_equal(N(again(lab)), N(vel))
    # half resolution: outputs at half the shape, SVF at that resolution
    half = ne.models.labels_to_image_new(labels, in_shape=(S, S, S), half_res=True, return_vel=True, return_def=True)
    im_h, oh_h, vel_h, def_h = half(lab)
    assert im_h.shape == (B, S // 2, S // 2, S // 2, 1) and oh_h.shape[-1] == 6 and vel_h.shape == def_h.shape == (B, S // 2, S // 2, S // 2, 3)
