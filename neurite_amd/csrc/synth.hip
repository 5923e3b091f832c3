// Elementwise steps of the label-to-image synthesis model (neurite/tf/models.py:649-918, `labels_to_image`), gfx950.
// The reference strings ~25 TF ops (gather, mul, add, less, logical_and, cast, exp, clip, pow, one_hot ...) over volume-sized
// tensors; here each stage between two spatial operators (warp, blur, min-max) is one pass:
//   relabel      labels -> dense indices through a lookup table                                   (:778-784)
//   intensity    image = noise * std[label] + mean[label], optionally zeroing the background      (:819-849)
//   bias_clip    image = clip(image * exp(bias), 0, 255)                                          (:860-874)
//   gamma_dc     image = image ^ exp(gamma[b, c]) + dc[b, c]                                       (:877-888)
//   labels_out   indices -> output labels (lookup) and one-hot encoding, -1 = dropped label         (:890-918)
// Random numbers are drawn by the caller (device RNG); these kernels are deterministic functions of their inputs.

#include "nrt_common.h"

namespace {

unsigned sblocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256ll * 16) b = 256ll * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

__global__ __launch_bounds__(256) void relabel(const int *__restrict__ labels, const float *__restrict__ lut, int lut_len,
                                               float *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int l = labels[i];
        out[i] = (l >= 0 && l < lut_len) ? lut[l] : 0.0f;          // tf.gather on the GPU returns 0 for out-of-range indices
    }
}

// labels, noise [B, V] (float indices; ONE normal draw per voxel shared by the channels, :831), out [B, V, C],
// mean / std [B, C, L], bgzero [B, C] (0 or 1)
__global__ __launch_bounds__(256) void intensity(const float *__restrict__ labels, const float *__restrict__ noise,
                                                 const float *__restrict__ mean, const float *__restrict__ stdv,
                                                 const float *__restrict__ bgzero, float *__restrict__ out, long long V, int C, int L) {
    const int b = blockIdx.y;
    const long long n = V * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long v = e / C;
        const int c = (int)(e - v * C);
        int l = (int)labels[(long long)b * V + v];
        l = min(max(l, 0), L - 1);
        const long long t = ((long long)b * C + c) * L + l;
        float val = noise[(long long)b * V + v] * stdv[t] + mean[t];
        if (bgzero && l == 0 && bgzero[b * C + c] != 0.0f) val = val * 0.0f;      // image *= 1 - mask  (keeps NaN/Inf semantics)
        out[(long long)b * n + e] = val;
    }
}

// image [N, C], bias [N] (one channel, broadcast over C) or null
__global__ __launch_bounds__(256) void bias_clip(const float *__restrict__ image, const float *__restrict__ bias, float *__restrict__ out,
                                                 long long n, int C, float lo, float hi) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n * C; e += (long long)gridDim.x * 256) {
        float v = image[e];
        if (bias) v = v * expf(bias[e / C]);
        out[e] = fminf(fmaxf(v, lo), hi);
    }
}

// image [B, V, C]; gamma, dc [B, C] or null
__global__ __launch_bounds__(256) void gamma_dc(const float *__restrict__ image, const float *__restrict__ gamma,
                                                const float *__restrict__ dc, float *__restrict__ out, long long V, int C) {
    const int b = blockIdx.y;
    const long long n = V * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        float v = image[(long long)b * n + e];
        if (gamma) v = powf(v, expf(gamma[b * C + c]));
        if (dc) v += dc[b * C + c];
        out[(long long)b * n + e] = v;
    }
}

// idx [N] float indices -> lut[idx]; one_hot: out [N, depth] float32 (all zero for lut value < 0 or >= depth); else out_i32 [N]
__global__ __launch_bounds__(256) void labels_out(const float *__restrict__ idx, const int *__restrict__ lut, int lut_len, int depth,
                                                  float *__restrict__ onehot, int *__restrict__ out_i32, long long n) {
    if (onehot && (depth & 3) == 0 && n * depth < (1ll << 32) && (((uintptr_t)onehot) & 15) == 0) {
        // one 16-byte store per thread and iteration, 32-bit index arithmetic (the scalar form below spends its time in a
        // 64-bit division per element)
        // (round 2: non-temporal stores and a shift instead of the division when depth / 4 is a power of two -- a pure write
        // stream; 4 x 160^3 x 32: 0.545 ms = 3.9 TB/s before)
        const unsigned q = (unsigned)depth >> 2, total = (unsigned)((n * depth) >> 2);
        const bool pow2 = (q & (q - 1u)) == 0u;
        const unsigned sh = (unsigned)__builtin_ctz(q);
        nrt_f4 *o4 = (nrt_f4 *)onehot;
        for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
            const unsigned v = pow2 ? e >> sh : e / q;
            const int d0 = (int)(e - v * q) * 4;
            const int l = lut[min(max((int)idx[v], 0), lut_len - 1)];
            __builtin_nontemporal_store((nrt_f4){l == d0 ? 1.0f : 0.0f, l == d0 + 1 ? 1.0f : 0.0f, l == d0 + 2 ? 1.0f : 0.0f,
                                                 l == d0 + 3 ? 1.0f : 0.0f}, o4 + e);
        }
    } else if (onehot) {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n * depth; e += (long long)gridDim.x * 256) {
            const long long v = e / depth;
            const int d = (int)(e - v * depth);
            const int l = min(max((int)idx[v], 0), lut_len - 1);
            onehot[e] = lut[l] == d ? 1.0f : 0.0f;
        }
    } else {
        for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
            const int l = min(max((int)idx[v], 0), lut_len - 1);
            out_i32[v] = lut[l];
        }
    }
}


// ---- stages of labels_to_image_new (neurite/tf/models.py:920-1300) and of the augmentation layers it instantiates ------
// x viewed as [outer, A, inner]: y = x * mask[a]   (RandomCrop, layers.py:446-519 / augment.draw_crop_mask)
__global__ __launch_bounds__(256) void axis_mask(const float *__restrict__ x, const float *__restrict__ mask, float *__restrict__ y,
                                                 long long total, int A, long long inner) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int a = (int)((e / inner) % A);
        y[e] = x[e] * mask[a];
    }
}

// y[o, j, i] = x[o, idx[j], i]   (Subsample, layers.py:367-443 / utils.subsample_axis: tf.gather along one axis)
__global__ __launch_bounds__(256) void axis_gather(const float *__restrict__ x, const int *__restrict__ idx, float *__restrict__ y,
                                                   long long total, int A, int J, long long inner) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e % inner, r = e / inner;
        const int j = (int)(r % J);
        const long long o = r / J;
        y[e] = x[(o * A + idx[j]) * inner + i];
    }
}

// y[b, v, c] = x[b, v, c] + sd[b * sdb + c * sdc] * noise[b, v, c]   (GaussianNoise, layers.py:2305-2403)
__global__ __launch_bounds__(256) void noise_add(const float *__restrict__ x, const float *__restrict__ noise, const float *__restrict__ sd,
                                                 float *__restrict__ y, long long V, int C, int sdb, int sdc) {
    const int b = blockIdx.y;
    const long long n = V * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long long g = (long long)b * n + e;
        y[g] = x[g] + sd[b * sdb + c * sdc] * noise[g];
    }
}

// y[b, v, c] = image[b, v, c] * (labels[b, v] == 0 && flag[b] ? 0 : 1)   (background clearing, models.py:1213-1223)
__global__ __launch_bounds__(256) void bg_clear(const float *__restrict__ image, const float *__restrict__ labels,
                                                const float *__restrict__ flag, float *__restrict__ y, long long V, int C) {
    const int b = blockIdx.y;
    const long long n = V * C;
    const bool on = flag[b] != 0.0f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long g = (long long)b * n + e;
        const float keep = (on && labels[(long long)b * V + e / C] == 0.0f) ? 0.0f : 1.0f;
        y[g] = image[g] * keep;
    }
}

}  // namespace

extern "C" int nrt_synth_relabel_i32(const int *labels, const float *lut, int lut_len, float *out, long long n, void *stream) {
    if (!labels || !lut || !out || lut_len < 1 || n < 0) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    hipLaunchKernelGGL(relabel, dim3(sblocks(n)), dim3(256), 0, nrt_stream(stream), labels, lut, lut_len, out, n);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_intensity_f32(const float *labels, const float *noise, const float *mean, const float *stdv,
                                       const float *bg_zero, float *out, int batch, long long nvox, int channels, int nlabels,
                                       void *stream) {
    if (!labels || !noise || !mean || !stdv || !out || batch < 1 || batch > 65535 || nvox < 0 || channels < 1 || nlabels < 1)
        return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    hipLaunchKernelGGL(intensity, dim3(sblocks(nvox * channels), batch), dim3(256), 0, nrt_stream(stream), labels, noise, mean, stdv,
                       bg_zero, out, nvox, channels, nlabels);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_bias_clip_f32(const float *image, const float *bias, float *out, long long n, int channels, float lo,
                                       float hi, void *stream) {
    if (!image || !out || n < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    hipLaunchKernelGGL(bias_clip, dim3(sblocks(n * channels)), dim3(256), 0, nrt_stream(stream), image, bias, out, n, channels, lo, hi);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_gamma_dc_f32(const float *image, const float *gamma, const float *dc, float *out, int batch, long long nvox,
                                      int channels, void *stream) {
    if (!image || !out || batch < 1 || batch > 65535 || nvox < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    hipLaunchKernelGGL(gamma_dc, dim3(sblocks(nvox * channels), batch), dim3(256), 0, nrt_stream(stream), image, gamma, dc, out, nvox,
                       channels);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_labels_out(const float *idx, const int *lut, int lut_len, int depth, float *onehot, int *out_i32, long long n,
                                    void *stream) {
    if (!idx || !lut || lut_len < 1 || n < 0 || (!onehot && !out_i32) || (onehot && depth < 1)) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    hipLaunchKernelGGL(labels_out, dim3(sblocks(onehot ? n * depth : n)), dim3(256), 0, nrt_stream(stream), idx, lut, lut_len, depth,
                       onehot, out_i32, n);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_axis_mask_f32(const float *x, const float *mask, float *y, long long outer, int axis_len, long long inner,
                                       void *stream) {
    if (!x || !mask || !y || outer < 0 || axis_len < 1 || inner < 1) return NRT_ERR_INVALID_ARG;
    const long long total = outer * axis_len * inner;
    if (total == 0) return NRT_OK;
    hipLaunchKernelGGL(axis_mask, dim3(sblocks(total)), dim3(256), 0, nrt_stream(stream), x, mask, y, total, axis_len, inner);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_axis_gather_f32(const float *x, const int *index, float *y, long long outer, int axis_len, int out_len,
                                         long long inner, void *stream) {
    if (!x || !index || !y || outer < 0 || axis_len < 1 || out_len < 0 || inner < 1) return NRT_ERR_INVALID_ARG;
    const long long total = outer * out_len * inner;
    if (total == 0) return NRT_OK;
    hipLaunchKernelGGL(axis_gather, dim3(sblocks(total)), dim3(256), 0, nrt_stream(stream), x, index, y, total, axis_len, out_len, inner);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_noise_add_f32(const float *x, const float *noise, const float *sd, float *y, int batch, long long nvox,
                                       int channels, int sd_batch_stride, int sd_channel_stride, void *stream) {
    if (!x || !noise || !sd || !y || batch < 1 || batch > 65535 || nvox < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    hipLaunchKernelGGL(noise_add, dim3(sblocks(nvox * channels), batch), dim3(256), 0, nrt_stream(stream), x, noise, sd, y, nvox, channels,
                       sd_batch_stride, sd_channel_stride);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_synth_bg_clear_f32(const float *image, const float *labels, const float *flag, float *y, int batch, long long nvox,
                                      int channels, void *stream) {
    if (!image || !labels || !flag || !y || batch < 1 || batch > 65535 || nvox < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    hipLaunchKernelGGL(bg_clear, dim3(sblocks(nvox * channels), batch), dim3(256), 0, nrt_stream(stream), image, labels, flag, y, nvox, channels);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
