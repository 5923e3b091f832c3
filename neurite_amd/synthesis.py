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

__all__ = ['labels_to_image', 'SynthModel', 'labels_to_image_new', 'SynthModelNew']


class SynthModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.last_draws = {}
        self.name = 'synth_%d' % cfg['id']
        self._seq = {}

    # ---- random draws -------------------------------------------------------------------------------------------
    def _seed(self, key):
        """Per-call seed of a component.  A seeded tf.random op yields a reproducible SEQUENCE that advances with every
        call (the reference's seeds, models.py:690-694); so each key owns a NumPy generator seeded once from the user's seed
        and every forward draws a fresh sub-seed from it: two models built alike agree call by call, two consecutive
        calls of one model differ.  Unseeded keys draw from fresh entropy."""
        if key not in self._seq:
            self._seq[key] = np.random.default_rng(self.cfg['seeds'].get(key))
        return int(self._seq[key].integers(2 ** 31 - 1))

    def _gen(self, dev, key):
        g = torch.Generator(device=dev)
        g.manual_seed(self._seed(key))
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
            seed = self._seed('warp')
            vel_field = torch.stack([
                augment.draw_perlin(vel_shape, scales=vel_scale, min_std=0 if c['warp_modulate'] else c['warp_std'],
                                    max_std=c['warp_std'], seed=seed + b, device=dev)
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
                                            dtype=image.dtype, seed=self._seed('blur'))
            kernels = kernels if isinstance(kernels, list) else [kernels]
            draws['blur_kernels'] = kernels
            image = utils.separable_conv(image, kernels, batched=True)
        # ---- bias field, clipping (:859-874) -----------------------------------------------------------------------
        bias = None
        if c['bias_std'] > 0:
            seed = self._seed('bias')
            bias = torch.stack([
                augment.draw_perlin(tuple(int(s) for s in c['out_shape']) + (1,), scales=c['bias_res'],
                                    min_std=0 if c['bias_modulate'] else c['bias_std'], max_std=c['bias_std'],
                                    seed=seed + b, device=dev)
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
    in_shape = np.asarray(in_shape)
    out_shape = in_shape if out_shape is None else np.asarray(out_shape)
    num_dim = in_shape.size

    # label tables (models.py:777-783, 846-865) as vectorised look-ups.  Input labels -> ranks 0 .. N-1 (their sorted order) ...
    in_label_list = np.unique(in_label_list).astype(np.int32)
    num_in_labels = in_label_list.size
    in_lut = np.zeros(int(in_label_list[-1]) + 1, np.float32)
    in_lut[in_label_list] = np.arange(num_in_labels, dtype=np.float32)
    # ... rank -> output label (0 for labels the caller drops); a sequence of labels stands for the identity mapping on them
    if out_label_list is None:
        out_label_list = in_label_list
    mapping = dict(out_label_list) if isinstance(out_label_list, dict) else {lab: lab for lab in np.asarray(out_label_list).tolist()}
    out_lut = np.array([mapping.get(int(lab), 0) for lab in in_label_list], np.int32)
    depth = 0
    if one_hot:                                   # ... and output label -> one-hot channel (its rank among the output labels)
        hot_labels = np.unique(np.fromiter(mapping.values(), dtype=np.int64))
        hot_lut = np.full(int(hot_labels[-1]) + 1, -1, np.int32)
        hot_lut[hot_labels] = np.arange(hot_labels.size, dtype=np.int32)
        out_lut = hot_lut[out_lut]
        depth = int(hot_labels.size)

    # intensity ranges per input label (models.py:812-819): the first label (background) gets zero mean / spread at the low end
    def per_label(value, first, rest):
        return [first] + [rest] * (num_in_labels - 1) if value is None else value
    mean_min, mean_max = per_label(mean_min, 0, 25), per_label(mean_max, 225, 225)
    std_min, std_max = per_label(std_min, 0, 5), per_label(std_max, 25, 25)
    cfg = dict(in_shape=tuple(int(s) for s in in_shape), out_shape=out_shape.astype(np.int64), num_dim=num_dim,
               num_chan=int(num_chan), num_in_labels=num_in_labels, in_lut=in_lut, out_lut=np.ascontiguousarray(out_lut, np.int32),
               depth=depth, mean_min=mean_min, mean_max=mean_max, std_min=std_min, std_max=std_max,
               zero_background=zero_background, warp_res=warp_res, warp_std=warp_std, warp_modulate=warp_modulate,
               bias_res=bias_res, bias_std=bias_std, bias_modulate=bias_modulate, blur_std=blur_std,
               blur_modulate=blur_modulate, normalize=normalize, gamma_std=gamma_std, dc_offset=dc_offset, one_hot=one_hot,
               seeds=dict(seeds), return_vel=return_vel, return_def=return_def, id=id)
    model = SynthModel(cfg)
    if input_model is None:
        return model
    # models.py:763-769: the generator is appended to `input_model` (one output = the label map); the returned model takes
    # input_model's inputs
    return _Chained(input_model, model)


class _Chained(torch.nn.Module):
    """`input_model` followed by the generator: labels = input_model(*inputs) (exactly one output), then the synthesis."""

    def __init__(self, first, second):
        super().__init__()
        self.first, self.second = first, second

    def forward(self, *inputs):
        labels = self.first(*inputs)
        if isinstance(labels, (list, tuple)):
            assert len(labels) == 1, 'labels_to_image: input_model must have exactly one output'
            labels = labels[0]
        return self.second(labels)


# ======================================================================================================================
# labels_to_image_new (neurite/tf/models.py:920-1300): affine + diffeomorphic augmentation of the label map, per-label
# mean intensities, bias field, noise, background clearing, blur, thick slices, normalisation, gamma, label conversion
# ======================================================================================================================

class SynthModelNew(nn.Module):
    """The model `labels_to_image_new` returns: `outputs = model(label_map [B, *in_shape, 1])`."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.last_draws = {}
        self.name = 'synth_new_%d' % cfg['id']
        self._rand = np.random.default_rng(cfg['seeds'].get('_model'))
        self._built = False

    def _seed(self, key):
        """per-call seed of a component: reproducible sequences for seeded components, fresh entropy otherwise"""
        if key not in self._seq:
            s = self.cfg['seeds'].get(key)
            self._seq[key] = np.random.default_rng(s)
        return int(self._seq[key].integers(2 ** 31 - 1))

    def _gen(self, dev, key):
        g = torch.Generator(device=dev)
        g.manual_seed(self._seed(key))
        return g

    def _build(self):
        c = self.cfg
        self._seq = {}
        nd = c['num_dim']
        out_shape = tuple(int(s) for s in c['out_shape'])
        sd = c['seeds']
        self.perlin_warp = None
        if c['warp_max'] > 0:
            vshape = tuple(int(s) for s in (c['out_shape'] // (1 if c['half_res'] else 2))) + (nd,)
            self.perlin_warp = layers.PerlinNoise(shape=vshape, noise_min=c['warp_min'], noise_max=c['warp_max'], isotropic=False,
                                                  fwhm_min=np.asarray(c['warp_blur_min']) / 2, fwhm_max=np.asarray(c['warp_blur_max']) / 2,
                                                  reduce='max', axes=-1, seed=sd.get('warp'))
            self.vec_int = layers.VecInt(int_steps=5, name='vec_int_%d' % c['id'])
            self.rescale = layers.RescaleTransform(zoom_factor=2, name='def_%d' % c['id'])
            self.compose = layers.ComposeTransform()
        self.to_dense = layers.AffineToDenseShift(out_shape, shift_center=False)
        self.warp = layers.SpatialTransformer(interp_method='nearest', fill_value=0, name='trans_%d' % c['id'])
        self.crop = layers.RandomCrop(crop_min=c['crop_min'], crop_max=c['crop_max'], prob=c['crop_prob'], axis=c['crop_axes'],
                                      seed=sd.get('crop'))
        self.perlin_bias = None
        if c['bias_max'] > 0:
            div = 2 if c['half_res'] else 1
            self.perlin_bias = layers.PerlinNoise(noise_min=c['bias_min'], noise_max=c['bias_max'], isotropic=False,
                                                  fwhm_min=c['bias_blur_min'] / div, fwhm_max=c['bias_blur_max'] / div, reduce='max',
                                                  seed=sd.get('bias'))
        self.noise = layers.GaussianNoise(c['noise_min'], c['noise_max'], seed=sd.get('noise'))
        self.blur = layers.GaussianBlur(sigma=c['blur_max'], min_sigma=c['blur_min'], random=True, seed=None)
        div = 2 if c['half_res'] else 1
        self.slices = layers.Subsample(prob=c['slice_prob'], stride_min=max(1, c['slice_stride_min'] / div),
                                       stride_max=max(1, c['slice_stride_max'] / div), axes=c['slice_axes'], seed=sd.get('slice'))
        self._built = True

    def forward(self, labels):
        c = self.cfg
        if not self._built:
            self._build()
        lib = _lib.lib()
        dev = _lib.require_device(labels)
        nd = c['num_dim']
        if labels.dim() != nd + 2 or labels.shape[-1] != 1 or tuple(labels.shape[1:-1]) != tuple(int(s) for s in c['in_shape']):
            raise ValueError('labels_to_image_new expects label maps of shape [B, %s, 1], got %s'
                             % (', '.join(str(int(s)) for s in c['in_shape']), tuple(labels.shape)))
        B = labels.shape[0]
        st = _lib.stream_ptr(dev)
        draws = {}
        lab = labels.to(torch.float32).contiguous()                       # compute_type (:1059-1061)
        in_shape, out_shape = np.asarray(c['in_shape']), np.asarray(c['out_shape'])
        # ---- affine transform (:1068-1100) -----------------------------------------------------------------------------
        par = utils.draw_affine_params(shift=c['aff_shift'], rot=c['aff_rotate'], scale=c['aff_scale'], shear=c['aff_shear'],
                                       normal_shift=c['aff_normal_shift'], normal_rot=c['aff_normal_rotate'],
                                       normal_scale=c['aff_normal_scale'], normal_shear=c['aff_normal_shear'], ndims=nd,
                                       batch_shape=[B], seeds={k: self._seed(k) for k in ('shift', 'rot', 'scale', 'shear')})
        affine = utils.params_to_affine_matrix(par, deg=True, shift_scale=True, last_row=True, ndims=nd).to(torch.float64)
        origin = np.eye(nd + 1)
        origin[:nd, -1] = -0.5 * (in_shape - 1)
        center = np.eye(nd + 1)
        center[:nd, -1] = np.round(0.5 * (in_shape - (2 if c['half_res'] else 1) * out_shape))
        scale = np.diag((*[2 if c['half_res'] else 1] * nd, 1))
        trans = torch.from_numpy(np.linalg.inv(origin)) @ affine @ torch.from_numpy(origin @ center @ scale)
        if c['axes_flip']:
            trans = trans @ utils.draw_flip_matrix(out_shape, shift_center=False, dtype=torch.float64, seed=self._seed('flip'))
        if c['axes_swap']:
            assert all(x == out_shape[0] for x in out_shape), 'non-isotropic output shape'
            trans = trans @ utils.draw_swap_matrix(nd, dtype=torch.float64, seed=self._seed('swap'))
        draws['affine'], draws['matrix'] = affine.to(torch.float32), trans.to(torch.float32)
        trans = self.to_dense(trans[:, :nd, :].to(torch.float32).to(dev))
        # ---- diffeomorphic deformation (:1102-1127) ----------------------------------------------------------------------
        vel_field = def_field = None
        if c['warp_max'] > 0:
            vel_field = self.perlin_warp(lab)
            if c['warp_zero_mean']:
                vel_field = vel_field - vel_field.mean(dim=tuple(range(1, nd + 1)), keepdim=True)
            def_field = self.vec_int(vel_field)
            if not c['half_res']:
                def_field = self.rescale(def_field)
            trans = self.compose([trans, def_field])
        draws['trans'] = trans
        lab = self.warp([lab, trans])                                      # nearest, fill 0; output grid = out_shape
        lab = torch.trunc(lab)                                             # tf.cast(labels, int32) (:1133)
        lab = self.crop(lab)
        draws['labels'] = lab
        S = tuple(lab.shape[1:-1])
        V = int(np.prod(S))
        C, L = c['num_chan'], c['num_label']
        # ---- generation labels -> indices, mean intensities (:1145-1177) ------------------------------------------------
        gen_lut = torch.from_numpy(c['gen_lut']).to(dev)
        idx = torch.empty(lab.shape, dtype=torch.float32, device=dev)
        lab_i = lab.to(torch.int32).contiguous()
        with torch.cuda.device(dev):
            rc = lib.nrt_synth_relabel_i32(_lib.ptr(lab_i), _lib.ptr(gen_lut), gen_lut.numel(), _lib.ptr(idx), lab_i.numel(), st)
        _lib.check(rc, 'nrt_synth_relabel_i32')
        m0 = torch.as_tensor(np.asarray(c['mean_min'], np.float32), device=dev)
        m1 = torch.as_tensor(np.asarray(c['mean_max'], np.float32), device=dev)
        mean = (m0 + (m1 - m0) * torch.rand((B, C, L), generator=self._gen(dev, 'mean'), device=dev)).contiguous()
        zeros_v = torch.zeros((B,) + S, dtype=torch.float32, device=dev)
        zeros_s = torch.zeros_like(mean)
        image = torch.empty((B,) + S + (C,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.nrt_synth_intensity_f32(_lib.ptr(idx.contiguous()), _lib.ptr(zeros_v), _lib.ptr(mean), _lib.ptr(zeros_s), None,
                                             _lib.ptr(image), B, V, C, L, st)
        _lib.check(rc, 'nrt_synth_intensity_f32')
        mean_image = image
        draws['mean'] = mean
        # ---- bias field (:1179-1193) --------------------------------------------------------------------------------------
        bias_field = None
        if c['bias_max'] > 0:
            bias_field = self.perlin_bias(image).contiguous()
            draws['bias_field'] = bias_field
            out = torch.empty_like(image)
            with torch.cuda.device(dev):                                  # image * exp(bias), one bias value per element
                rc = lib.nrt_synth_bias_clip_f32(_lib.ptr(image), _lib.ptr(bias_field), _lib.ptr(out), image.numel(), 1,
                                                 float('-inf'), float('inf'), st)
            _lib.check(rc, 'nrt_synth_bias_clip_f32')
            image = out
            bias_field = torch.exp(bias_field)                              # bias_func = tf.exp: what return_bias hands out
        # ---- noise (:1196) -------------------------------------------------------------------------------------------------
        image = self.noise(image)
        if c['noise_max'] > 0:
            draws['noise'], draws['noise_sd'] = self.noise.last_draws['noise'], self.noise.last_draws['sd']
        # ---- background clearing (:1198-1208) --------------------------------------------------------------------------------
        if c['zero_background'] > 0:
            flag = (torch.rand((B,), generator=self._gen(dev, 'background'), device=dev) < c['zero_background']).to(torch.float32)
            draws['bg_zero'] = flag
            out = torch.empty_like(image)
            with torch.cuda.device(dev):
                rc = lib.nrt_synth_bg_clear_f32(_lib.ptr(image.contiguous()), _lib.ptr(lab.contiguous()), _lib.ptr(flag.contiguous()),
                                                _lib.ptr(out), B, V, C, st)
            _lib.check(rc, 'nrt_synth_bg_clear_f32')
            image = out
        # ---- blur, thick slices (:1210-1226) --------------------------------------------------------------------------------
        sig_hi = self.blur._normalize_sigma(c['blur_max'], nd)
        sig_lo = self.blur._normalize_sigma(c['blur_min'], nd)
        if any(s > 0 for s in sig_hi):                                    # layers.GaussianBlur(random=True).call
            kernels = utils.gaussian_kernel(sigma=sig_hi, random=True, min_sigma=sig_lo, separate=True, dtype=image.dtype,
                                            seed=self._seed('blur'))
            kernels = kernels if isinstance(kernels, list) else [kernels]
            draws['blur_kernels'] = kernels
            image = utils.separable_conv(image, kernels, batched=True)
        image = self.slices(image)
        draws['pre_norm'] = image
        # ---- intensity manipulations (:1228-1241) ----------------------------------------------------------------------------
        if c['normalize']:
            image = utils.minmax_norm(image, axis=tuple(range(1, nd + 2)))
        if c['gamma'] > 0:
            g = (1 - c['gamma']) + 2 * c['gamma'] * torch.rand((B, C), generator=self._gen(dev, 'gamma'), device=dev)
            draws['gamma'] = g
            out = torch.empty_like(image)
            with torch.cuda.device(dev):                                  # the kernel raises to exp(.): pass log(gamma)
                rc = lib.nrt_synth_gamma_dc_f32(_lib.ptr(image.contiguous()), _lib.ptr(torch.log(g).contiguous()), None, _lib.ptr(out),
                                                B, V, C, st)
            _lib.check(rc, 'nrt_synth_gamma_dc_f32')
            image = out
        # ---- output labels (:1243-1263) ----------------------------------------------------------------------------------------
        out_lut = torch.from_numpy(c['out_lut']).to(dev)
        lab_c = lab.contiguous()
        if c['one_hot']:
            depth = c['depth']
            lab_out = torch.empty((B,) + S + (depth,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nrt_synth_labels_out(_lib.ptr(lab_c), _lib.ptr(out_lut), out_lut.numel(), depth, _lib.ptr(lab_out), None, B * V, st)
        else:
            lab_out = torch.empty((B,) + S + (1,), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nrt_synth_labels_out(_lib.ptr(lab_c), _lib.ptr(out_lut), out_lut.numel(), 1, None, _lib.ptr(lab_out), B * V, st)
        _lib.check(rc, 'nrt_synth_labels_out')
        self.last_draws = draws
        outputs = []
        if c['return_im']:
            outputs.append(image)
        if c['return_map']:
            outputs.append(lab_out)
        if c['return_vel']:
            outputs.append(vel_field)
        if c['return_def']:
            outputs.append(def_field)
        if c['return_aff']:
            outputs.append(affine.to(torch.float32).to(dev))
        if c['return_mean']:
            outputs.append(mean_image)
        if c['return_bias']:
            outputs.append(bias_field)
        return outputs[0] if len(outputs) == 1 else outputs


def labels_to_image_new(labels_in, labels_out=None, in_shape=None, out_shape=None, input_model=None, num_chan=1, aff_shift=0,
                        aff_rotate=0, aff_scale=0, aff_shear=0, aff_normal_shift=False, aff_normal_rotate=False,
                        aff_normal_scale=False, aff_normal_shear=False, axes_flip=False, axes_swap=False, warp_min=0.01,
                        warp_max=2, warp_blur_min=(8, 8), warp_blur_max=(32, 32), warp_zero_mean=False, crop_min=0, crop_max=0.2,
                        crop_prob=0, crop_axes=None, mean_min=None, mean_max=None, noise_min=0.1, noise_max=0.2,
                        zero_background=0, blur_min=0, blur_max=1, bias_min=0.01, bias_max=0.1, bias_blur_min=32,
                        bias_blur_max=64, bias_func='exp', slice_stride_min=1, slice_stride_max=8, slice_prob=0,
                        slice_axes=None, normalize=True, gamma=0.5, one_hot=True, half_res=False, seeds={}, return_im=True,
                        return_map=True, return_vel=False, return_def=False, return_aff=False, return_mean=False,
                        return_bias=False, id=0):
    """
    Build the model that augments label maps and synthesises images from them (neurite/tf/models.py:920-1300); same
    parameters, defaults and label-lookup semantics.  `bias_func` must be the exponential (the reference default, tf.exp).
    Returns a module: `outputs = model(label_map [B, *in_shape, 1])` in the reference's order (image, labels, vel, def, aff,
    mean, bias -- those requested).
    """
    if in_shape is None:
        # models.py:1083 reads the shape off input_model's symbolic output; a torch module has none, so it is always passed
        raise ValueError('labels_to_image_new needs in_shape (the spatial shape of the input label maps)')
    if not (isinstance(bias_func, str) and bias_func == 'exp') and getattr(bias_func, '__name__', '') != 'exp':
        raise NotImplementedError('labels_to_image_new: bias_func must be the exponential')
    if isinstance(seeds, str):
        seeds = [seeds]
    if isinstance(seeds, dict):
        seeds = seeds.copy()
    if not isinstance(seeds, dict):
        seeds = {f: hash(f) for f in seeds}
    known = {'shift', 'rot', 'scale', 'shear', 'flip', 'swap', 'warp', 'crop', 'mean', 'bias', 'noise', 'background', 'blur',
             'slice', 'gamma'}
    assert not (set(seeds) - known), f'unknown seeds {dict((k, v) for k, v in seeds.items() if k not in known)}'
    in_shape = np.asarray(in_shape)
    if out_shape is None:
        out_shape = in_shape
    out_shape = np.array(out_shape) // (2 if half_res else 1)
    num_dim = len(in_shape)
    if num_dim not in (2, 3):
        raise NotImplementedError('labels_to_image_new: 2-D and 3-D label maps')
    # generation labels (:1145-1153)
    if not isinstance(labels_in, dict):
        labels_in = {i: i for i in labels_in}
    labels_gen = set(labels_in.values())
    ind = {gen: i for i, gen in enumerate(labels_gen)}
    gen_lut = np.asarray([ind.get(labels_in.get(i), 0) for i in range(max(labels_in) + 1)], dtype=np.float32)
    num_label = len(labels_gen)
    if mean_min is None:
        mean_min = [0] * num_label
    if mean_max is None:
        mean_max = [1] * num_label
    if gamma > 0:
        assert 0 < gamma < 1, f'gamma value {gamma} outside interval [0, 1)'
    # output labels (:1243-1258)
    lut = list(labels_in) if labels_out is None else labels_out
    if not isinstance(lut, dict):
        lut = {i: i for i in lut}
    labels_out_set = set(lut.values())
    if one_hot:
        oind = {out: i for i, out in enumerate(labels_out_set)}
        lut = {inp: oind[out] for inp, out in lut.items()}
    out_lut = np.asarray([lut.get(i, -1 if one_hot else 0) for i in range(max(labels_in) + 1)], dtype=np.int32)
    cfg = dict(in_shape=in_shape, out_shape=out_shape, num_dim=num_dim, num_chan=int(num_chan), aff_shift=aff_shift,
               aff_rotate=aff_rotate, aff_scale=aff_scale, aff_shear=aff_shear, aff_normal_shift=aff_normal_shift,
               aff_normal_rotate=aff_normal_rotate, aff_normal_scale=aff_normal_scale, aff_normal_shear=aff_normal_shear,
               axes_flip=axes_flip, axes_swap=axes_swap, warp_min=warp_min, warp_max=warp_max, warp_blur_min=warp_blur_min,
               warp_blur_max=warp_blur_max, warp_zero_mean=warp_zero_mean, crop_min=crop_min, crop_max=crop_max,
               crop_prob=crop_prob, crop_axes=crop_axes, mean_min=mean_min, mean_max=mean_max, noise_min=noise_min,
               noise_max=noise_max, zero_background=zero_background, blur_min=blur_min, blur_max=blur_max, bias_min=bias_min,
               bias_max=bias_max, bias_blur_min=bias_blur_min, bias_blur_max=bias_blur_max, slice_stride_min=slice_stride_min,
               slice_stride_max=slice_stride_max, slice_prob=slice_prob, slice_axes=slice_axes, normalize=normalize, gamma=gamma,
               one_hot=one_hot, half_res=half_res, seeds=seeds, return_im=return_im, return_map=return_map, return_vel=return_vel,
               return_def=return_def, return_aff=return_aff, return_mean=return_mean, return_bias=return_bias, id=id,
               gen_lut=gen_lut, num_label=num_label, out_lut=out_lut, depth=len(labels_out_set))
    model = SynthModelNew(cfg)
    # models.py:1074-1077, 1301: the generator is appended to `input_model` and the result takes input_model's inputs
    return model if input_model is None else _Chained(input_model, model)
