"""
neurite_amd.synthesis -- the label-to-image generative model of neurite/tf/models.py:649-918 (`labels_to_image`,
SynthMorph) on the HIP kernels: random SVF -> VecInt -> Resize -> nearest SpatialTransformer of the label map, per-label
intensity sampling, Gaussian blur, multiplicative bias field, clipping, min-max normalisation, gamma and DC offset, label
conversion / one-hot encoding.  Every volume-sized step is a kernel of this package (csrc/synth.hip, interpn.hip,
filter.hip); random numbers come from torch's device generator (same distributions as the reference's tf.random calls, a
different stream -- stochastic outputs have no value parity).  `SynthModel.last_draws` keeps the draws of the last call so
that the deterministic part can be checked against the oracle.
"""

import warnings

import numpy as np
import torch
from torch import nn

from . import _lib
from . import augment
from . import layers
from . import utils

__all__ = ['labels_to_image', 'SynthModel']


class SynthModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.last_draws = {}
        self.name = 'synth_%d' % cfg['id']

    # ---- random draws -------------------------------------------------------------------------------------------
    def _gen(self, dev, key, salt=0):
        g = torch.Generator(device=dev)
        seed = self.cfg['seeds'].get(key)
        if seed is None:
            g.seed()
        else:
            g.manual_seed(int(seed) + 7919 * salt)
        return g

    def forward(self, labels):
        c = self.cfg
        lib = _lib.lib()
        dev = _lib.require_device(labels)
        num_dim = c['num_dim']
        if labels.dim() != num_dim + 2 or labels.shape[-1] != 1 or tuple(labels.shape[1:-1]) != tuple(c['in_shape']):
            raise ValueError('labels_to_image expects label maps of shape [B, %s, 1], got %s'
                             % (', '.join(str(s) for s in c['in_shape']), tuple(labels.shape)))
        B = labels.shape[0]
        st = _lib.stream_ptr(dev)
        draws = {}
        # ---- labels -> dense indices [0, N) (:778-784) -----------------------------------------------------------
        lab = labels if labels.dtype == torch.int32 else labels.to(torch.int32)            # tf.cast(labels, int32) (:773-774)
        lab = lab.contiguous()
        in_lut = torch.from_numpy(c['in_lut']).to(dev)
        idx = torch.empty(lab.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.nrt_synth_relabel_i32(_lib.ptr(lab), _lib.ptr(in_lut), in_lut.numel(), _lib.ptr(idx), lab.numel(), st)
        _lib.check(rc, 'nrt_synth_relabel_i32')
        vel_field = def_field = None
        # ---- random diffeomorphic warp of the label map (:786-806) ------------------------------------------------
        if c['warp_std'] > 0:
            vel_shape = tuple(int(s) for s in c['out_shape'] // 2) + (num_dim,)
            vel_scale = np.asarray(c['warp_res']) / 2
            seed = c['seeds'].get('warp')
            vel_field = torch.stack([
                augment.draw_perlin(vel_shape, scales=vel_scale, min_std=0 if c['warp_modulate'] else c['warp_std'],
                                    max_std=c['warp_std'], seed=None if seed is None else seed + b, device=dev)
                for b in range(B)], 0)
            def_field = layers.VecInt(int_steps=5, name='vec_int_%d' % c['id'])(vel_field)
            def_field = def_field * 2                                                    # layers.RescaleValues(2), half-resolution field
            def_field = layers.Resize(2, interp_method='linear', name='def_%d' % c['id'])(def_field)
            idx = layers.SpatialTransformer(interp_method='nearest', fill_value=0,
                                            name='trans_%d' % c['id'])([idx, def_field])
            draws['vel_field'] = vel_field
        S = tuple(idx.shape[1:-1])
        V = int(np.prod(S))
        C, L = c['num_chan'], c['num_in_labels']
        # ---- per-label intensity statistics and the synthetic image (:810-839) ------------------------------------
        m0, m1, s0, s1 = [torch.as_tensor(np.asarray(a, np.float32), device=dev) for a in (c['mean_min'], c['mean_max'], c['std_min'], c['std_max'])]
        mean = m0 + (m1 - m0) * torch.rand((B, C, L), generator=self._gen(dev, 'mean'), device=dev)
        std = s0 + (s1 - s0) * torch.rand((B, C, L), generator=self._gen(dev, 'std'), device=dev)
        noise = torch.randn((B,) + S, generator=self._gen(dev, 'noise'), device=dev)
        bgz = None
        if c['zero_background'] > 0:
            flip = torch.rand((B, C), generator=self._gen(dev, 'background'), device=dev)
            bgz = (flip < c['zero_background']).to(torch.float32).contiguous()
        draws.update(mean=mean, std=std, noise=noise, bg_zero=bgz, labels_warped=idx)
        image = torch.empty((B,) + S + (C,), dtype=torch.float32, device=dev)
        idx = idx.contiguous()
        with torch.cuda.device(dev):
            rc = lib.nrt_synth_intensity_f32(_lib.ptr(idx), _lib.ptr(noise), _lib.ptr(mean.contiguous()), _lib.ptr(std.contiguous()),
                                             _lib.ptr(bgz), _lib.ptr(image), B, V, C, L, st)
        _lib.check(rc, 'nrt_synth_intensity_f32')
        # ---- blur (:851-857) ----------------------------------------------------------------------------------------
        if c['blur_std'] > 0:
            kernels = utils.gaussian_kernel([c['blur_std']] * num_dim, separate=True, random=c['blur_modulate'],
                                            dtype=image.dtype, seed=c['seeds'].get('blur'))
            kernels = kernels if isinstance(kernels, list) else [kernels]
            draws['blur_kernels'] = kernels
            image = utils.separable_conv(image, kernels, batched=True)
        # ---- bias field, clipping (:859-874) -----------------------------------------------------------------------
        bias = None
        if c['bias_std'] > 0:
            seed = c['seeds'].get('bias')
            bias = torch.stack([
                augment.draw_perlin(tuple(int(s) for s in c['out_shape']) + (1,), scales=c['bias_res'],
                                    min_std=0 if c['bias_modulate'] else c['bias_std'], max_std=c['bias_std'],
                                    seed=None if seed is None else seed + b, device=dev)
                for b in range(B)], 0).contiguous()
            if tuple(bias.shape[1:-1]) != S:
                raise ValueError('Incompatible shapes: image %s and bias field %s' % (S, tuple(bias.shape[1:-1])))
            draws['bias_field'] = bias
        out = torch.empty_like(image)
        with torch.cuda.device(dev):
            rc = lib.nrt_synth_bias_clip_f32(_lib.ptr(image), _lib.ptr(bias), _lib.ptr(out), B * V, C, 0.0, 255.0, st)
        _lib.check(rc, 'nrt_synth_bias_clip_f32')
        image = out
        # ---- normalisation, gamma, offset (:875-888) ---------------------------------------------------------------
        if c['normalize']:
            image = utils.minmax_norm(image, axis=tuple(range(1, num_dim + 2)))            # tf.map_fn(minmax_norm) over the batch
        gamma = dc = None
        if c['gamma_std'] > 0:
            gamma = (torch.randn((B, C), generator=self._gen(dev, 'gamma'), device=dev) * c['gamma_std']).contiguous()
        if c['dc_offset'] > 0:
            dc = (torch.rand((B, C), generator=self._gen(dev, 'dc_offset'), device=dev) * c['dc_offset']).contiguous()
        draws.update(gamma=gamma, dc_offset=dc)
        if gamma is not None or dc is not None:
            out = torch.empty_like(image)
            with torch.cuda.device(dev):
                rc = lib.nrt_synth_gamma_dc_f32(_lib.ptr(image), _lib.ptr(gamma), _lib.ptr(dc), _lib.ptr(out), B, V, C, st)
            _lib.check(rc, 'nrt_synth_gamma_dc_f32')
            image = out
        # ---- output labels (:890-918) ------------------------------------------------------------------------------
        out_lut = torch.from_numpy(c['out_lut']).to(dev)
        if c['one_hot']:
            depth = c['depth']
            lab_out = torch.empty((B,) + S + (depth,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nrt_synth_labels_out(_lib.ptr(idx), _lib.ptr(out_lut), out_lut.numel(), depth, _lib.ptr(lab_out), None, B * V, st)
        else:
            lab_out = torch.empty((B,) + S + (1,), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nrt_synth_labels_out(_lib.ptr(idx), _lib.ptr(out_lut), out_lut.numel(), 1, None, _lib.ptr(lab_out), B * V, st)
        _lib.check(rc, 'nrt_synth_labels_out')
        self.last_draws = draws
        outputs = [image, lab_out]
        if c['return_vel']:
            outputs.append(vel_field)
        if c['return_def']:
            outputs.append(def_field)
        return outputs


def labels_to_image(in_shape, in_label_list, out_label_list=None, out_shape=None, num_chan=1, input_model=None,
                    mean_min=None, mean_max=None, std_min=None, std_max=None, zero_background=0.2, warp_res=[16],
                    warp_std=0.5, warp_modulate=True, bias_res=40, bias_std=0.3, bias_modulate=True, blur_std=1,
                    blur_modulate=True, normalize=True, gamma_std=0.25, dc_offset=0, one_hot=True, seeds={},
                    return_vel=False, return_def=False, id=0):
    """
    Generative model for augmenting label maps and synthesising images from them (neurite/tf/models.py:649-918); same
    parameters and defaults.  Returns a module: `image, labels[, vel][, def] = model(label_map [B, *in_shape, 1])`.
    """
    warnings.warn('model `labels_to_image` is deprecated in favor `labels_to_image_new`')
    if input_model is not None:
        raise NotImplementedError('labels_to_image: input_model chaining is not implemented; call the models in sequence')
    if out_shape is None:
        out_shape = in_shape
    in_shape, out_shape = map(np.asarray, (in_shape, out_shape))
    num_dim = len(in_shape)
    in_label_list = np.int32(np.unique(in_label_list))
    num_in_labels = len(in_label_list)
    in_lut = np.zeros(np.max(in_label_list) + 1, dtype=np.float32)
    for i, lab in enumerate(in_label_list):
        in_lut[lab] = i
    if mean_min is None:
        mean_min = [0] + [25] * (num_in_labels - 1)
    if mean_max is None:
        mean_max = [225] * num_in_labels
    if std_min is None:
        std_min = [0] + [5] * (num_in_labels - 1)
    if std_max is None:
        std_max = [25] * num_in_labels
    if out_label_list is None:
        out_label_list = in_label_list
    if isinstance(out_label_list, (tuple, list, np.ndarray)):
        out_label_list = {lab: lab for lab in out_label_list}
    out_lut = np.zeros(num_in_labels, dtype='int32')
    for i, lab in enumerate(in_label_list):
        if lab in out_label_list:
            out_lut[i] = out_label_list[lab]
    depth = 0
    if one_hot:
        hot_label_list = np.unique(list(out_label_list.values()))
        hot_lut = np.full(hot_label_list[-1] + 1, fill_value=-1, dtype='int32')
        for i, lab in enumerate(hot_label_list):
            hot_lut[lab] = i
        out_lut = hot_lut[out_lut]
        depth = len(hot_label_list)
    cfg = dict(in_shape=tuple(int(s) for s in in_shape), out_shape=out_shape.astype(np.int64), num_dim=num_dim,
               num_chan=int(num_chan), num_in_labels=num_in_labels, in_lut=in_lut, out_lut=np.ascontiguousarray(out_lut, np.int32),
               depth=depth, mean_min=mean_min, mean_max=mean_max, std_min=std_min, std_max=std_max,
               zero_background=zero_background, warp_res=warp_res, warp_std=warp_std, warp_modulate=warp_modulate,
               bias_res=bias_res, bias_std=bias_std, bias_modulate=bias_modulate, blur_std=blur_std,
               blur_modulate=blur_modulate, normalize=normalize, gamma_std=gamma_std, dc_offset=dc_offset, one_hot=one_hot,
               seeds=dict(seeds), return_vel=return_vel, return_def=return_def, id=id)
    return SynthModel(cfg)
