"""
ctypes binding of libneurite_amd.so (C ABI: include/neurite_amd.h).

PyTorch is only plumbing here: it owns device memory (caching allocator) and streams.  Every
compute call goes through the C ABI with raw device pointers; there is NO CPU or eager-PyTorch
fallback -- if the HIP library is missing or a tensor is not on a ROCm device the call raises.
"""

import ctypes as C
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NEURITE_AMD_LIB') or os.path.join(_HERE, 'lib', 'libneurite_amd.so')      # NEURITE_AMD_LIB: another build of the same ABI
HEADER_PATH = os.path.join(_HERE, '..', 'include', 'neurite_amd.h')

NRT_OK = 0
NRT_ERR_INVALID_ARG, NRT_ERR_UNSUPPORTED, NRT_ERR_LAUNCH, NRT_ERR_WORKSPACE = -1, -2, -3, -4
LOC_ABSOLUTE, LOC_SHIFT, LOC_LINSPACE = 0, 1, 2
INTERP_LINEAR, INTERP_NEAREST = 0, 1
DT_F32, DT_BF16, DT_F16, DT_F64, DT_I32 = 0, 1, 2, 3, 4

_lib = None

_vp, _i, _ll, _f, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)

_SIGNATURES = {
    'nrt_status_string': (C.c_char_p, [_i]),
    'nrt_abi_version': (_i, []),
    'nrt_target_arch': (C.c_char_p, []),
    'nrt_build_id': (C.c_char_p, []),
    'nrt_init': (_i, []),
    'nrt_counters_reset': (_i, [_vp]),
    'nrt_counters_slot_index': (_i, [_vp]),
    'nrt_interpn_f32': (_i, [_vp, _vp, _vp, _i, _ip, _ip, _i, _i, _ll, _ll, _i, _i, _i, _f, _vp]),
    'nrt_interpn_f32_ex': (_i, [_vp, _vp, _vp, _i, _ip, _ip, _i, _i, _ll, _ll, _i, _i, _i, _f, _i, _i, _vp]),
    'nrt_interpn_add_f32': (_i, [_vp, _vp, _vp, _vp, _i, _ip, _ip, _i, _i, _ll, _ll, _ll, _i, _i, _f, _vp]),
    'nrt_affine_to_dense_shift_f32': (_i, [_vp, _i, _i, _ip, _i, _vp, _vp]),
    'nrt_interpn_nearest_i32': (_i, [_vp, _vp, _vp, _i, _ip, _ip, _i, _i, _ll, _ll, _i, _i, C.c_int32, _vp]),
    'nrt_interpn_any': (_i, [_vp, _vp, _vp, _i, _i, _ip, _ip, _i, _i, _ll, _ll, _i, _i, _i, _i, C.c_double, _vp]),
    'nrt_dice_workspace_bytes': (_sz, [_ll, _i, _i]),
    'nrt_dice_soft_f32': (_i, [_vp, _vp, _ll, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nrt_dice_soft': (_i, [_vp, _vp, _i, _ll, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nrt_sqdiff_sums_f32': (_i, [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'nrt_dice_hard_prob': (_i, [_vp, _vp, _i, _ll, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nrt_dice_hard_prob_f32': (_i, [_vp, _vp, _ll, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'nrt_dice_hard_prob_minmax_f32': (_i, [_vp, _vp, _ll, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nrt_dice_hard_label_i32': (_i, [_vp, _vp, _ll, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'nrt_dice_from_sums_f32': (_i, [_vp, _i, _i, _f, _vp, _vp]),
    'nrt_dice_mean_pair_f32': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'nrt_dice_mean_f32': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'nrt_wcce_workspace_bytes': (_sz, [_ll, _i]),
    'nrt_seg_loss_supported': (_i, [_i]),
    'nrt_seg_loss_workspace_bytes': (_sz, [_ll, _i, _i]),
    'nrt_seg_loss_f32': (_i, [_vp, _vp, _vp, _ll, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nrt_seg_loss_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _f, _f, _i, _vp, _vp]),
    'nrt_warp_dice_workspace_bytes': (_sz, [_ip, _i, _i, _i]),
    'nrt_warp_dice_soft_f32': (_i, [_vp, _vp, _vp, _vp, _ip, _ip, _i, _i, _ll, _i, _i, _f, _f, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    'nrt_warp_dice_kernel_name': (C.c_char_p, [_ip, _ip, _i, _i, _i, _i, _i, _i, _i]),
    'nrt_warp_dice_soft_bf16': (_i, [_vp, _vp, _vp, _ip, _ip, _i, _i, _ll, _i, _i, _f, _f, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    'nrt_conv3d_packed_weight_floats': (_sz, [_ip, _i, _i]),
    'nrt_conv3d_pack_weights_f32': (_i, [_vp, _ip, _i, _i, _vp, _vp]),
    'nrt_conv3d_f32': (_i, [_vp, _i, _vp, _i, _ip, _vp, _vp, _vp, _vp, _i, _ip, _ip, _i, _i, _i, _i, _i, _vp]),
    'nrt_conv3d_c1_pool_supported': (_i, [_ip, _i]),
    'nrt_conv3d_c1_pool_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _ip, _i, _i, _vp]),
    'nrt_conv3d_pool_supported': (_i, [_i, _i, _ip, _i]),
    'nrt_conv3d_pool_f32': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _ip, _i, _i, _vp]),
    'nrt_conv3d_up2_supported': (_i, [_i, _i, _i, _ip]),
    'nrt_conv3d_up2_packed_weight_floats': (_sz, [_i, _i, _i]),
    'nrt_conv3d_up2_pack_weights_f32': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'nrt_conv3d_up2_f32': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _ip, _i, _i, _vp]),
    'nrt_conv3d_up2_head_supported': (_i, [_i, _i, _i, _i, _ip]),
    'nrt_conv3d_up2_head_pack_f32': (_i, [_vp, _i, _vp, _vp]),
    'nrt_conv3d_up2_head_f32': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _ip, _i, _i, _vp]),
    'nrt_space_to_depth2_f32': (_i, [_vp, _vp, _i, _ip, _i, _vp]),
    'nrt_conv3d_s2d_taps_f32': (_i, [_vp, _i, _vp, _vp, _i, _ip, _i, _vp]),
    'nrt_conv1x1_softmax_f32': (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp]),
    'nrt_softmax_lastdim_f32': (_i, [_vp, _vp, _ll, _i, _vp]),
    'nrt_maxpool3d_f32': (_i, [_vp, _vp, _i, _ip, _i, _ip, _i, _vp]),
    'nrt_upsample_concat_f32': (_i, [_vp, _i, _vp, _i, _vp, _i, _ip, _ip, _vp]),
    'nrt_add_act_affine_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    'nrt_act_bwd_f32': (_i, [_vp, _vp, _i, _vp, _ll, _vp]),
    'nrt_conv3d_wgrad_f32': (_i, [_vp, _vp, _vp, _vp, _i, _ip, _i, _i, _ip, _i, _vp]),
    'nrt_conv3d_wgrad2_f32': (_i, [_vp, _i, _vp, _i, _ip, _vp, _vp, _vp, _i, _ip, _i, _ip, _i, _vp]),
    'nrt_conv3d_wgrad_s2d_f32': (_i, [_vp, _vp, _vp, _i, _ip, _i, _i, _vp]),
    'nrt_maxpool3d_bwd_f32': (_i, [_vp, _vp, _vp, _i, _ip, _i, _ip, _i, _vp]),
    'nrt_upsample_sum_f32': (_i, [_vp, _i, _i, _vp, _i, _i, _ip, _ip, _vp]),
    'nrt_softmax_bwd_f32': (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
    'nrt_channel_sums_f32': (_i, [_vp, _vp, _ll, _i, _vp, _vp]),
    'nrt_channel_axpby_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp]),
    'nrt_lc3d_f': (_i, [_vp, _vp, _vp, _vp, _i, _i, _ip, _i, _ip, _ip, _i, _i, _i, _vp]),
    'nrt_lc3d_bwd_f': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ip, _i, _ip, _ip, _i, _i, _vp]),
    'nrt_pad3d': (_i, [_vp, _vp, _i, _ip, _ip, _ip, _i, _i, _vp]),
    'nrt_conv1d_axis_f32': (_i, [_vp, _vp, _vp, _ll, _i, _ll, _i, _i, _i, _i, _i, _vp]),
    'nrt_minmax_workspace_bytes': (_sz, [_ll, _i]),
    'nrt_minmax_norm_f32': (_i, [_vp, _vp, _ll, _ll, _i, _vp, _sz, _vp]),
    'nrt_minmax_f32': (_i, [_vp, _ll, _vp, _vp, _sz, _vp]),
    'nrt_bin_centers_f32': (_i, [_vp, _ll, _i, _vp, _vp, _sz, _vp]),
    'nrt_soft_quantize_f32': (_i, [_vp, _vp, _f, _f, _f, _i, _vp, _ll, _i, _vp]),
    'nrt_mi_joint_f32': (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _ll, _i, _i, _vp, _vp, _vp, _vp]),
    'nrt_mi_joint_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nrt_colsum_f32': (_i, [_vp, _i, _ll, _i, _vp, _vp]),
    'nrt_mi_from_joint_f32': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    'nrt_synth_relabel_i32': (_i, [_vp, _vp, _i, _vp, _ll, _vp]),
    'nrt_synth_intensity_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _vp]),
    'nrt_synth_bias_clip_f32': (_i, [_vp, _vp, _vp, _ll, _i, _f, _f, _vp]),
    'nrt_synth_gamma_dc_f32': (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp]),
    'nrt_synth_labels_out': (_i, [_vp, _vp, _i, _i, _vp, _vp, _ll, _vp]),
    'nrt_synth_axis_mask_f32': (_i, [_vp, _vp, _vp, _ll, _i, _ll, _vp]),
    'nrt_synth_axis_gather_f32': (_i, [_vp, _vp, _vp, _ll, _i, _i, _ll, _vp]),
    'nrt_synth_noise_add_f32': (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp]),
    'nrt_synth_bg_clear_f32': (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp]),
    'nrt_wcce': (_i, [_vp, _vp, _i, _vp, _ll, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'nrt_wcce_mean': (_i, [_vp, _vp, _i, _vp, _ll, _i, _i, _f, C.c_double, _vp, _vp, _vp, _sz, _vp]),
    'nrt_interpn_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _ip, _ip, _i, _i, _ll, _ll, _i, _i, _vp]),
    'nrt_interpn_nearest_bwd_f32': (_i, [_vp, _vp, _vp, _i, _ip, _ip, _i, _i, _ll, _ll, _i, _i, _vp]),
    'nrt_soft_quantize_bwd_f32': (_i, [_vp, _vp, _f, _f, _f, _i, _vp, _vp, _ll, _i, _vp]),
    'nrt_dice_soft_bwd_norm_f32': (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _f, _vp, _vp, _vp]),
    'nrt_dice_soft_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _f, _vp, _vp, _vp]),
    'nrt_warp_dice_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ip, _ip, _i, _i, _ll, _i, _i, _f, _vp]),
    'nrt_wcce_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _f, _f, _vp, _vp]),
}


class NeuriteAmdError(RuntimeError):
    pass


def declared_symbols():
    """Every function name declared in include/neurite_amd.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nrt_[a-z0-9_]+)\s*\(', text)))


def lib():
    """Load the HIP library (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NeuriteAmdError(
                'libneurite_amd.so not found at %s: build it with `python -m neurite_amd.build` '
                '(hipcc, --offload-arch=gfx950). neurite_amd has no CPU fallback.' % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != NRT_OK:
        msg = lib().nrt_status_string(rc).decode()
        raise NeuriteAmdError('%s failed: %s (status %d)' % (what or 'neurite_amd call', msg, rc))


def require_device(*tensors):
    """All tensors must live on one ROCm device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError('expected a torch.Tensor, got %s' % type(t).__name__)
        if t.device.type != 'cuda':
            raise NeuriteAmdError(
                'neurite_amd runs on MI355X (ROCm) tensors only; got a tensor on %s. '
                'There is deliberately no CPU fallback.' % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise NeuriteAmdError('tensors live on different devices: %s vs %s' % (dev, t.device))
    if dev is not None and dev.index not in _initialised:
        init_device(dev)
    return dev


_initialised = set()


def init_device(dev):
    """nrt_init() on `dev`, once: the library resolves the address of its counter pool now -- outside any stream capture, where nothing
    but launches should happen (include/neurite_amd.h)."""
    with torch.cuda.device(dev):
        check(lib().nrt_init(), 'nrt_init')
    _initialised.add(dev.index)


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    # a DeferredWarp (neurite_amd/deferred.py) evaluates itself when its address is asked for
    return C.c_void_p(t.data_ptr()) if t is not None else None


def ints(values):
    return (C.c_int * len(values))(*[int(v) for v in values])


_workspaces = {}


def workspace(device, nbytes):
    """A per-device scratch buffer owned by the torch caching allocator, grown on demand.
    Kernels on one stream are ordered, so one buffer per (device, stream) is enough."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
