"""
GPU tests of the label-to-image synthesis model (neurite_amd.synthesis.labels_to_image; neurite/tf/models.py:649-918).
The model is stochastic (torch's RNG instead of tf.random): the deterministic part is checked against the oracle given the
recorded draws, the rest structurally and statistically.
"""

import warnings

import numpy as np
import pytest
import torch

import neurite_amd as ne
from neurite_amd import synth
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def N(t):
    return None if t is None else t.detach().cpu().numpy()


def make(*a, **k):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m = ne.models.labels_to_image(*a, **k)
        assert any('deprecated' in str(x.message) for x in w)            # models.py:756
    return m


def label_map(dev, B, S, labels, seed=1):
    lab = torch.stack([synth.blob_labels(seed + b, size=S, nb_labels=len(labels), coarse=4, device=dev) for b in range(B)], 0)
    lut = torch.tensor(labels, device=dev)
    return lut[lab.long()][..., None].to(torch.int32)


def test_deterministic_part_matches_oracle(dev):
    labels = [0, 2, 3, 41, 42]
    S, B = 32, 2
    lab = label_map(dev, B, S, labels)
    model = make((S, S, S), labels, num_chan=2, seeds=dict(warp=3, mean=4, std=5, noise=6, background=7, blur=8, bias=9, gamma=10, dc_offset=11),
                 zero_background=0.6, dc_offset=0.1, bias_res=[8, 16], return_vel=True, return_def=True)
    image, lab_out, vel, dfield = model(lab)
    assert image.shape == (B, S, S, S, 2) and lab_out.shape == (B, S, S, S, 5)
    assert vel.shape == (B, S // 2, S // 2, S // 2, 3) and dfield.shape == (B, S, S, S, 3)
    d = model.last_draws
    ref = npo.synth_image(N(d['labels_warped']), N(d['noise']), N(d['mean']), N(d['std']), N(d['bg_zero']),
                          [N(k) for k in d['blur_kernels']], N(d['bias_field']), True, N(d['gamma']), N(d['dc_offset']))
    np.testing.assert_allclose(N(image), ref, rtol=2e-5, atol=2e-5)
    # labels: the warped indices are a nearest-neighbour warp of the dense relabelling; one-hot of the output lookup
    idx = N(d['labels_warped'])[..., 0].astype(int)
    assert idx.min() >= 0 and idx.max() < len(labels)
    oh = N(lab_out)
    assert np.array_equal(oh.argmax(-1), idx) and np.all(oh.sum(-1) == 1)
    dense = np.searchsorted(np.array(labels), N(lab)[..., 0])
    want = npo.spatial_transformer(dense[..., None].astype(F), N(dfield), 'nearest', fill_value=0)
    assert np.array_equal(N(d['labels_warped']), want)
    # the deformation is VecInt(5) of the SVF, doubled and up-sampled
    v = N(vel)
    df = np.stack([npo.resize((npo.integrate_vec(v[b], 'ss', 5) * F(2)).astype(F), 2) for b in range(B)], 0)
    np.testing.assert_allclose(N(dfield), df, rtol=1e-6, atol=1e-6)
    # seeded components give a reproducible SEQUENCE that advances per call (as seeded tf.random ops do): the second call of
    # this model differs from the first, and a second model built with the same seeds repeats the sequence call by call
    image2 = model(lab)[0]
    assert not torch.equal(image, image2)
    kw = dict(num_chan=2, seeds=dict(warp=3, mean=4, std=5, noise=6, background=7, blur=8, bias=9, gamma=10, dc_offset=11),
              zero_background=0.6, dc_offset=0.1, bias_res=[8, 16], return_vel=True, return_def=True)
    twin = make((S, S, S), labels, **kw)
    assert torch.equal(twin(lab)[0], image) and torch.equal(twin(lab)[0], image2)
    assert float(image.min()) >= 0.0 and float(image.max()) <= 1.0 + 0.1 + 1e-6


def test_options_and_label_conversion(dev):
    labels = [0, 4, 7, 9]
    S, B = 24, 2
    lab = label_map(dev, B, S, labels, seed=5)
    # no warp, no blur, no bias, no gamma: image = clip(noise * std + mean) normalised; labels converted through a dictionary
    model = make((S, S, S), labels, out_label_list={4: 1, 7: 1, 9: 2}, warp_std=0, blur_std=0, bias_std=0, gamma_std=0,
                 zero_background=0, one_hot=False, seeds=dict(mean=1, std=2, noise=3))
    image, lab_out = model(lab)
    assert lab_out.dtype == torch.int32 and lab_out.shape == (B, S, S, S, 1)
    conv = {0: 0, 4: 1, 7: 1, 9: 2}
    assert np.array_equal(N(lab_out), np.vectorize(conv.get)(N(lab)))
    d = model.last_draws
    ref = npo.synth_image(N(d['labels_warped']), N(d['noise']), N(d['mean']), N(d['std']), None, None, None, True, None, None)
    np.testing.assert_allclose(N(image), ref, rtol=1e-6, atol=1e-6)
    # one-hot without the background among the output labels: background voxels are all-zero rows
    model = make((S, S, S), labels, out_label_list=[4, 9], warp_std=0, seeds=dict(mean=1))
    _, oh = model(lab)
    oh = N(oh)
    assert oh.shape[-1] == 2                                              # hot labels {4, 9}; 0 and 7 map to -1 = all-zero rows (:905-913)
    src = N(lab)[..., 0]
    assert np.all(oh[src == 7] == 0) and np.all(oh[src == 0] == 0)
    assert np.all(oh[src == 4].argmax(-1) == 0) and np.all(oh[src == 9].argmax(-1) == 1) and np.all(oh[src == 9].sum(-1) == 1)
    # statistics of the intensity model: per-label means within the requested bounds
    model = make((S, S, S), labels, warp_std=0, blur_std=0, bias_std=0, gamma_std=0, normalize=False, zero_background=0,
                 mean_min=[10, 50, 100, 200], mean_max=[11, 51, 101, 201], std_min=[0, 0, 0, 0], std_max=[1, 1, 1, 1])
    image, _ = model(lab)
    im = N(image)[..., 0]
    for l, lo in zip(labels, (10, 50, 100, 200)):
        m = im[src == l].mean()
        assert lo - 0.5 < m < lo + 1.5, (l, m)
    with pytest.raises(ValueError):
        model(lab[:, :-1])


def test_one_hot_output_vector_path(dev):
    """label counts that are multiples of 4 take the 16-byte-store form of the one-hot kernel: same result as the lookup"""
    labels = [0, 3, 4, 7, 9, 12, 20, 31]
    S, B = 16, 2
    lab = label_map(dev, B, S, labels)
    model = make((S, S, S), labels, out_label_list={3: 3, 4: 3, 7: 7, 9: 9, 12: 12, 20: 20, 31: 31, 0: 0}, seeds=dict(warp=1))
    image, oh = model(lab)
    oh = N(oh)
    d = model.last_draws
    idx = N(d['labels_warped'])[..., 0].astype(int)
    hot = sorted({0, 3, 7, 9, 12, 20, 31})                                     # merged labels 3 and 4 -> 7 output classes: scalar path
    assert oh.shape[-1] == len(hot)
    model8 = make((S, S, S), labels, seeds=dict(warp=1))
    _, oh8 = model8(lab)
    oh8 = N(oh8)
    idx8 = N(model8.last_draws['labels_warped'])[..., 0].astype(int)
    assert oh8.shape[-1] == 8 and np.array_equal(oh8.argmax(-1), idx8) and np.all(oh8.sum(-1) == 1)
    assert set(np.unique(oh8)) <= {0.0, 1.0}
    lut7 = np.array([hot.index({0: 0, 3: 3, 4: 3, 7: 7, 9: 9, 12: 12, 20: 20, 31: 31}[l]) for l in labels])
    assert np.array_equal(oh.argmax(-1), lut7[idx]) and np.all(oh.sum(-1) == 1)


def test_synthstrip_forward_and_training_step(dev):
    """SynthStrip (models.py:1888-1967): generator -> unet -> concat([logit, warped labels]); one SGD step through the unet"""
    import contextlib
    import io
    labels = [0, 1, 2, 3]
    S, B = 16, 2
    lab = label_map(dev, B, S, labels)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ne.models.SynthStrip((S, S, S), labels, {1: 1, 2: 1, 3: 1}, nb_unet_features=4, nb_unet_levels=2,
                                     gen_args=dict(seeds=dict(warp=2, mean=3)))
    model = model.to(dev)
    out = model(lab)
    assert out.shape == (B, S, S, S, 2)
    seg = N(out[..., 1])
    assert set(np.unique(seg)) <= {0.0, 1.0}                                        # brain / non-brain after the output look-up
    want = N(model.get_strip_model()(model.synth_image))
    assert np.array_equal(N(out[..., :1]), want)                                    # channel 0 is the unet applied to the synthetic image
    model.train()
    out = model(lab)
    target = out[..., 1:].detach()
    loss = ((out[..., :1] - target) ** 2).mean()
    loss.backward()
    grads = [p.grad for p in model.unet.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and any(float(g.abs().max()) > 0 for g in grads)


def test_input_model_is_chained_in_front(dev):
    """models.py:763-774 / 1074-1077: with `input_model` the generator consumes its single output and takes its inputs"""
    labels = [0, 2, 3]
    S = 16

    class Relabel(torch.nn.Module):                      # a front model: {0, 1, 2} -> the label values the generator expects
        def forward(self, x):
            return torch.tensor(labels, device=x.device, dtype=torch.int32)[x.long()]
    raw = label_map(dev, 1, S, [0, 1, 2])
    seeds = dict(warp=3, mean=4, std=5, noise=6, background=7, blur=8, bias=9, gamma=10, dc_offset=11)
    plain, chained = make((S, S, S), labels, seeds=seeds), make((S, S, S), labels, seeds=seeds, input_model=Relabel())
    want, got = plain(Relabel()(raw)), chained(raw)
    assert len(want) == len(got) == 2
    for a, b in zip(want, got):
        assert torch.equal(a, b)                          # same seeds -> the same draws -> the same image and label maps
    every = {k: i + 1 for i, k in enumerate(('shift', 'rot', 'scale', 'shear', 'flip', 'swap', 'warp', 'crop', 'mean', 'bias', 'noise',
                                             'background', 'blur', 'slice', 'gamma'))}
    new = ne.models.labels_to_image_new(labels, in_shape=(S, S, S), input_model=Relabel(), seeds=every)
    ref = ne.models.labels_to_image_new(labels, in_shape=(S, S, S), seeds=every)
    for a, b in zip(ref(Relabel()(raw)), new(raw)):
        assert torch.equal(a, b)
