// Segmentation training loss in one pass per direction: soft Dice (neurite/tf/metrics.py:415-482) and the label-weighted categorical
// cross-entropy (metrics.py:619-650) of the SAME two maps [B, ..., L], float32, L = 4 * 2^k labels.
//
//   forward   nrt_seg_loss_f32      one read of y_true / y_pred yields the per-label Dice sums (+ extrema for the range asserts) and the
//                                   CCE sum: 8 L bytes per voxel instead of 16 L for the two separate reductions (dice.hip, cce.hip)
//   backward  nrt_seg_loss_bwd_f32  d(sum_bl gD[b,l] dice[b,l] + gC cce_sum) / d y_pred, optionally carried through the channel soft-max
//                                   that produced y_pred (the unet's likelihood head, models.py:1545-1555): 12 L bytes per voxel instead
//                                   of 48 L for dice-bwd + cce-bwd + add + softmax-bwd
//
// Both kernels keep the arithmetic of the separate ones (same expressions, same order inside a voxel); only the order in which block
// partials meet differs, so results agree with the separate path to float32 rounding of the sums, not bit for bit.
// HBM-bound streaming kernels: lane-group of G = L/4 lanes per voxel, float4 per lane, a block owns one contiguous voxel range.
#include "dice_reduce.h"

namespace {

constexpr float SEG_KERAS_EPS = 1e-7f;

template <int G>
__device__ __forceinline__ float seg_group_sum(float v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, NRT_WAVE);
    return v;
}

template <int G>
__global__ __launch_bounds__(DICE_BLOCK) void seg_loss_vec(const nrt_f4 *__restrict__ yt, const nrt_f4 *__restrict__ yp,
                                                           const float *__restrict__ lw, long long nvox, float smooth,
                                                           float *__restrict__ fpart, float *__restrict__ mpart,
                                                           float *__restrict__ cpart) {
    constexpr int NG = DICE_BLOCK / G;
    constexpr int L = 4 * G;
    const int b = blockIdx.y;
    const nrt_f4 *t4 = yt + (long long)b * nvox * G, *p4 = yp + (long long)b * nvox * G;
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    float wq[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (lw) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wq[k] = lw[4 * lg + k];
    }
    const float keep = 1.0f - smooth, add = smooth / (float)L;

    nrt_f4 stp = {0, 0, 0, 0}, stt = {0, 0, 0, 0}, spp = {0, 0, 0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY, cce = 0.0f;
    long long vbeg, vend;
    nrt_block_range(nvox, NG, vbeg, vend);
#pragma unroll 4
    for (long long v = vbeg + g; v < vend; v += NG) {
        const nrt_f4 t = __builtin_nontemporal_load(t4 + v * G + lg);
        const nrt_f4 p = __builtin_nontemporal_load(p4 + v * G + lg);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                             // dice.hip: dice_soft_vec
            stp[k] += t[k] * p[k];
            stt[k] += t[k] * t[k];
            spp[k] += p[k] * p[k];
            mnt = fminf(mnt, t[k]); mxt = fmaxf(mxt, t[k]);
            mnp = fminf(mnp, p[k]); mxp = fmaxf(mxp, p[k]);
        }
        const float s = seg_group_sum<G>((p[0] + p[1]) + (p[2] + p[3]));      // cce.hip: wcce_vec, probabilities
        float l = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float q = p[k] / s;
            q = fminf(fmaxf(q, SEG_KERAS_EPS), 1.0f - SEG_KERAS_EPS);
            float tt = wq[k] * t[k];                              // metrics.py:648
            if (smooth != 0.0f) tt = tt * keep + add;
            l -= tt * logf(q);
        }
        cce += l;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        stp[k] = wave_xor_add(stp[k], G);
        stt[k] = wave_xor_add(stt[k], G);
        spp[k] = wave_xor_add(spp[k], G);
    }
    for (int off = 1; off < NRT_WAVE; off <<= 1) {
        mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
        mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
        cce += __shfl_xor(cce, off, NRT_WAVE);
    }
    __shared__ float red[DICE_BLOCK / NRT_WAVE][3 * L + 5];
    const int lane = threadIdx.x & (NRT_WAVE - 1), wv = threadIdx.x / NRT_WAVE;
    if (lane < G) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[wv][0 * L + 4 * lane + k] = stp[k];
            red[wv][1 * L + 4 * lane + k] = stt[k];
            red[wv][2 * L + 4 * lane + k] = spp[k];
        }
    }
    if (lane == 0) {
        red[wv][3 * L + 0] = mnt; red[wv][3 * L + 1] = mxt; red[wv][3 * L + 2] = mnp; red[wv][3 * L + 3] = mxp;
        red[wv][3 * L + 4] = cce;
    }
    __syncthreads();
    const long long pbase = ((long long)b * gridDim.x + blockIdx.x);
    for (int i = threadIdx.x; i < 3 * L; i += DICE_BLOCK) {
        float s = red[0][i];
#pragma unroll
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2) s += red[w2][i];
        fpart[pbase * 3 * L + i] = s;
    }
    if (threadIdx.x < 4) {
        float m = red[0][3 * L + threadIdx.x];
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2)
            m = (threadIdx.x & 1) ? fmaxf(m, red[w2][3 * L + threadIdx.x]) : fminf(m, red[w2][3 * L + threadIdx.x]);
        mpart[pbase * 4 + threadIdx.x] = m;
    }
    if (threadIdx.x == 4) {
        float s = red[0][3 * L + 4];
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2) s += red[w2][3 * L + 4];
        cpart[pbase] = s;
    }
}

// block partials of the CCE -> one float, fixed order, float64 accumulation (cce.hip: wcce_finalize)
__global__ __launch_bounds__(256) void seg_cce_finalize(const float *__restrict__ part, int n, float *__restrict__ out) {
    __shared__ double sl[256];
    double a = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) a += (double)part[k];
    sl[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += sl[i];
        out[0] = (float)s;
    }
}

// SM: the gradient is taken through y_pred = softmax(z) and written as dz
template <int G, bool SM>
__global__ __launch_bounds__(256) void seg_loss_bwd_vec(const nrt_f4 *__restrict__ yt, const nrt_f4 *__restrict__ yp,
                                                        const float *__restrict__ lw, const float *__restrict__ sums,
                                                        const float *__restrict__ gdice, const float *__restrict__ gcce,
                                                        long long nvox, float smooth, float eps, nrt_f4 *__restrict__ out) {
    constexpr int NG = 256 / G;
    constexpr int L = 4 * G;
    const int b = blockIdx.y;
    const long long base = (long long)b * nvox * G;
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    const float *s3 = sums + (long long)b * 3 * L;
    float ca[4], cb[4], wq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                 // backward.hip: dice_soft_bwd_vec
        const int l = 4 * lg + k;
        const float num = 2.0f * s3[l] + eps, den = s3[L + l] + s3[2 * L + l] + eps;
        const float gd = gdice ? gdice[(long long)b * L + l] : 0.0f;
        ca[k] = 0.0f; cb[k] = 0.0f;
        if (den != 0.0f) { ca[k] = 2.0f * gd / den; cb[k] = -2.0f * gd * num / (den * den); }
        wq[k] = lw ? lw[l] : 1.0f;
    }
    const float gc = gcce ? gcce[0] : 0.0f;
    const float keep = 1.0f - smooth, add = smooth / (float)L;
    long long vbeg, vend;
    nrt_block_range(nvox, NG, vbeg, vend);
#pragma unroll 2
    for (long long v = vbeg + g; v < vend; v += NG) {
        const nrt_f4 t = yt[base + v * G + lg], p = yp[base + v * G + lg];
        const float s = seg_group_sum<G>((p[0] + p[1]) + (p[2] + p[3]));      // backward.hip: wcce_bwd_vec, probabilities
        float tt[4], q[4], rq = 0.0f;
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tt[k] = wq[k] * t[k];
            if (smooth != 0.0f) tt[k] = tt[k] * keep + add;
            q[k] = p[k] / s;
            in[k] = q[k] >= SEG_KERAS_EPS && q[k] <= 1.0f - SEG_KERAS_EPS;
            rq += in[k] ? tt[k] : 0.0f;
        }
        rq = seg_group_sum<G>(rq);
        nrt_f4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            d[k] = (ca[k] * t[k] + cb[k] * p[k]) + (-gc * ((in[k] ? tt[k] / q[k] : 0.0f) - rq) / s);
        if (SM) {                                                 // conv_bwd.hip: softmax_bwd_vec
            const float dot = seg_group_sum<G>((d[0] * p[0] + d[1] * p[1]) + (d[2] * p[2] + d[3] * p[3]));
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = p[k] * (d[k] - dot);
        }
        out[base + v * G + lg] = d;                               // read again by the head's wgrad / dgrad: a normal store
    }
}

bool seg_labels_ok(int L) { return L >= 4 && L <= 256 && (L & (L - 1)) == 0; }

}  // namespace

extern "C" int nrt_seg_loss_supported(int nlabels) { return seg_labels_ok(nlabels) ? 1 : 0; }

extern "C" size_t nrt_seg_loss_workspace_bytes(long long nvox, int nlabels, int batch) {
    (void)nvox;
    if (nlabels < 1 || batch < 1) return 0;
    return dice_ws_bytes(nlabels, batch) + (size_t)batch * DICE_MAX_BLOCKS * sizeof(float) + 256;
}

extern "C" int nrt_seg_loss_f32(const float *y_true, const float *y_pred, const float *label_weights, long long nvox, int nlabels,
                                int batch, float label_smoothing, float laplace_smoothing, float *sums, float *dice, float *minmax,
                                float *cce_sum, void *workspace, size_t workspace_bytes, void *stream) {
    if (!y_true || !y_pred || !sums || !dice || !cce_sum) return NRT_ERR_INVALID_ARG;
    if (nvox < 1 || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (!seg_labels_ok(nlabels)) return NRT_ERR_UNSUPPORTED;
    if (((uintptr_t)y_true | (uintptr_t)y_pred) & 15) return NRT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < nrt_seg_loss_workspace_bytes(nvox, nlabels, batch)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    DiceWs w = dice_ws_carve(workspace, nlabels, batch);
    float *cpart = (float *)(((uintptr_t)workspace + dice_ws_bytes(nlabels, batch) + 15) & ~(uintptr_t)15);
    const int G = nlabels / 4;
    const unsigned nblk = dice_num_blocks(nvox, (DICE_BLOCK / G) * 4);
    dim3 grid(nblk, batch);
#define NRT_SEG(GG) hipLaunchKernelGGL((seg_loss_vec<GG>), grid, dim3(DICE_BLOCK), 0, st, (const nrt_f4 *)y_true, (const nrt_f4 *)y_pred, \
                                       label_weights, nvox, label_smoothing, w.fpart, w.mpart, cpart)
    switch (G) {
        case 1: NRT_SEG(1); break;
        case 2: NRT_SEG(2); break;
        case 4: NRT_SEG(4); break;
        case 8: NRT_SEG(8); break;
        case 16: NRT_SEG(16); break;
        case 32: NRT_SEG(32); break;
        default: NRT_SEG(64); break;
    }
#undef NRT_SEG
    NRT_CHECK_LAUNCH();
    hipLaunchKernelGGL(seg_cce_finalize, dim3(1), dim3(256), 0, st, (const float *)cpart, (int)(nblk * (unsigned)batch), cce_sum);
    NRT_CHECK_LAUNCH();
    return dice_finalize_soft(w, nblk, 1, batch, nlabels, laplace_smoothing, sums, dice, minmax, st);
}

extern "C" int nrt_seg_loss_bwd_f32(const float *y_true, const float *y_pred, const float *label_weights, const float *sums,
                                    const float *grad_dice, const float *grad_cce, long long nvox, int nlabels, int batch,
                                    float label_smoothing, float laplace_smoothing, int through_softmax, float *grad, void *stream) {
    if (!y_true || !y_pred || !sums || !grad) return NRT_ERR_INVALID_ARG;
    if (nvox < 1 || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (!seg_labels_ok(nlabels)) return NRT_ERR_UNSUPPORTED;
    if (((uintptr_t)y_true | (uintptr_t)y_pred | (uintptr_t)grad) & 15) return NRT_ERR_UNSUPPORTED;
    hipStream_t st = nrt_stream(stream);
    const int G = nlabels / 4;
    const unsigned nblk = dice_num_blocks(nvox, (256 / G) * 4);
    dim3 grid(nblk, batch);
#define NRT_SEGB(GG)                                                                                                              \
    do {                                                                                                                           \
        if (through_softmax)                                                                                                       \
            hipLaunchKernelGGL((seg_loss_bwd_vec<GG, true>), grid, dim3(256), 0, st, (const nrt_f4 *)y_true, (const nrt_f4 *)y_pred, \
                               label_weights, sums, grad_dice, grad_cce, nvox, label_smoothing, laplace_smoothing, (nrt_f4 *)grad);  \
        else                                                                                                                       \
            hipLaunchKernelGGL((seg_loss_bwd_vec<GG, false>), grid, dim3(256), 0, st, (const nrt_f4 *)y_true, (const nrt_f4 *)y_pred, \
                               label_weights, sums, grad_dice, grad_cce, nvox, label_smoothing, laplace_smoothing, (nrt_f4 *)grad);  \
    } while (0)
    switch (G) {
        case 1: NRT_SEGB(1); break;
        case 2: NRT_SEGB(2); break;
        case 4: NRT_SEGB(4); break;
        case 8: NRT_SEGB(8); break;
        case 16: NRT_SEGB(16); break;
        case 32: NRT_SEGB(32); break;
        default: NRT_SEGB(64); break;
    }
#undef NRT_SEGB
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
