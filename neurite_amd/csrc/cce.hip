// Label-weighted categorical cross-entropy for gfx950 (MI355X).
//
// Replaces neurite/tf/metrics.py:640-650 (y_true *= label_weights) followed by
// tf.keras.losses.CategoricalCrossentropy: [label smoothing], p / sum_c p, clip to [1e-7, 1-1e-7],
// -sum_c t'_c log p_c, mean over all B*V elements -- about nine TensorFlow ops with V*C-sized
// temporaries -- with one pass that reads y_true and y_pred once (2*itemsize*C bytes per voxel).
//
// Layout: y [N, C] row-major with N = B*V.  Fast path: C % 4 == 0, G = C/4 lanes own one voxel, each
// lane a fixed quad of channels (16 B loads for fp32, 8 B for bf16); the channel sum / max needed
// for the normalisation is a log2(G)-step xor-shuffle inside the lane-group.  bf16 inputs are
// widened to fp32 in registers; all arithmetic and the reduction are fp32 (final stage float64).
// Reduction: lane partials -> wave shuffles -> LDS -> per-block partial -> fixed-order second stage.

#include "nrt_common.h"

namespace {

constexpr int CCE_BLOCK = 256;
constexpr int CCE_MAX_BLOCKS = 2048;
constexpr float KERAS_EPS = 1e-7f;

typedef unsigned short nrt_us4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// exp(d), d = x - max <= 0, and log(s), s = a sum of such terms that contains exp(0) = 1, so s in [1, C]: one v_exp_f32 / v_log_f32 on
// scaled arguments instead of the ~20-instruction libm expansions -- the from-logits kernel was bound by VALU issue (4 expf + 1 logf per
// lane and voxel: 14 us of arithmetic for config 5's 10 us of streaming).  Relative error of exp <= 2^-23 + |d| 2^-24, absolute error of
// log on [1, C] <= 2^-22: per-voxel loss within 3e-7 absolute, far inside the 1e-5 of the tests (bf16 inputs: 1e-4).  The
// probabilities branch keeps logf: its argument runs up to 1 - 1e-7, where the result itself is ~1e-7.
__device__ __forceinline__ float lse_exp(float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896341f); }
__device__ __forceinline__ float lse_log(float s) { return __builtin_amdgcn_logf(s) * 0.693147180559945309f; }

template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ nrt_f4 load(const void *base, long long i) {
        return __builtin_nontemporal_load((const nrt_f4 *)base + i);
    }
    static __device__ __forceinline__ float load1(const void *base, long long i) { return ((const float *)base)[i]; }
};
template <> struct Quad<unsigned short> {
    static __device__ __forceinline__ nrt_f4 load(const void *base, long long i) {
        const nrt_us4 r = __builtin_nontemporal_load((const nrt_us4 *)base + i);
        return (nrt_f4){bf16_to_f32(r[0]), bf16_to_f32(r[1]), bf16_to_f32(r[2]), bf16_to_f32(r[3])};
    }
    static __device__ __forceinline__ float load1(const void *base, long long i) {
        return bf16_to_f32(((const unsigned short *)base)[i]);
    }
};

// The sum of the block partials, by the LAST block of the same launch (round 5; a second launch cost more than it computed at
// config 5's 53 MB).  A block publishes its partial with a device-scope store (sc1: written through, the eight XCDs have their own
// L2s), waits for the acknowledgement and takes a ticket from self-cleaning counters (api.hip: nrt_ring_slot; two levels, see below); the block that
// draws the last ticket reads the partials with device-scope loads and adds them in a FIXED order in float64 (what wcce_finalize
// did: run-to-run bit-identical), then zeroes the counter for the slot's next launch.  NOT __threadfence(): at agent scope it is
// buffer_wbl2 + buffer_inv -- every block invalidating its XCD's L2 under the other blocks' streams cost 35 us at config 5.
struct WcceFin { float *loss_sum; unsigned *counter; double divide_by; };     // loss_sum[0] = sum / divide_by (1: the sum; N: the mean)
__device__ __forceinline__ void wcce_finish(float *part, float block_sum, const WcceFin &fin) {
    __shared__ bool s_last;
    __shared__ double s_acc[CCE_BLOCK];
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + blockIdx.x, block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                         // the store is acknowledged ...
        asm volatile("" ::: "memory");
        // ... before the ticket says so.  Two levels: 2048 tickets on ONE address are 2048 serialised device-scope atomics (~20 us at
        // config 5, more than the streaming); a block draws from the counter of its residue class mod 8 (64 bytes apart), the last of a
        // class draws from the ninth counter, the last of those finishes
        const unsigned k = blockIdx.x & 7u, in_class = (gridDim.x + 7u - k) >> 3, classes = gridDim.x < 8u ? gridDim.x : 8u;
        bool last = false;
        if (__hip_atomic_fetch_add(fin.counter + 16u * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_class - 1u) {
            __hip_atomic_store(fin.counter + 16u * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(fin.counter + 16u * 8u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == classes - 1u;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    double a = 0.0;
    for (unsigned k = threadIdx.x; k < gridDim.x; k += CCE_BLOCK)
        a += (double)__hip_atomic_load(part + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_acc[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < CCE_BLOCK; ++i) t += s_acc[i];      // fixed order
        fin.loss_sum[0] = fin.divide_by == 1.0 ? (float)t : (float)t / (float)fin.divide_by;      // float32 sum, float32 division: as the two-op form
        __hip_atomic_store(fin.counter + 16u * 8u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// per-lane contribution -sum_k t'_k * logq_k for this lane's channels
template <int G, typename T, bool LOGITS, bool PERVOX>
__global__ __launch_bounds__(CCE_BLOCK) void wcce_vec(const void *__restrict__ yt, const void *__restrict__ yp,
                                                      const float *__restrict__ w, long long n, float smooth,
                                                      float *__restrict__ part, float *__restrict__ per_voxel, int Gr, WcceFin fin) {
    // Gr <= G channel quads per voxel are real (channel counts 4 Gr that are no power of two: 12, 20, 24 ...): lanes lg >= Gr of a
    // lane-group load nothing, enter the soft-max as -inf (the sums as 0) and add nothing to the loss
    constexpr int NG = CCE_BLOCK / G;
    const int C = 4 * Gr;
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    const bool real = lg < Gr;
    const int lgc = real ? lg : 0;
    float wq[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wq[k] = w[4 * lgc + k];
    }
    const float keep = 1.0f - smooth, add = smooth / (float)C;
    float acc = 0.0f;

    long long vbeg, vend;                                      // one contiguous range of voxels per block (nrt_block_range)
    nrt_block_range(n, NG, vbeg, vend);
#pragma unroll 2
    for (long long v = vbeg + g; v < vend; v += NG) {
        const nrt_f4 t = Quad<T>::load(yt, v * Gr + lgc);
        nrt_f4 p = Quad<T>::load(yp, v * Gr + lgc);
        if (!real) { const float pad = LOGITS ? -INFINITY : 0.0f; p = (nrt_f4){pad, pad, pad, pad}; }
        float lq[4];
        if (LOGITS) {
            float m = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
#pragma unroll
            for (int off = 1; off < G; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, NRT_WAVE));
            float se = (lse_exp(p[0] - m) + lse_exp(p[1] - m)) + (lse_exp(p[2] - m) + lse_exp(p[3] - m));
#pragma unroll
            for (int off = 1; off < G; off <<= 1) se += __shfl_xor(se, off, NRT_WAVE);
            const float lse = lse_log(se);
#pragma unroll
            for (int k = 0; k < 4; ++k) lq[k] = (p[k] - m) - lse;
        } else {
            float s = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off, NRT_WAVE);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float q = p[k] / s;
                q = fminf(fmaxf(q, KERAS_EPS), 1.0f - KERAS_EPS);
                lq[k] = logf(q);
            }
        }
        float l = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float tt = wq[k] * t[k];                 // metrics.py:648
            if (smooth != 0.0f) tt = tt * keep + add;
            l -= tt * lq[k];
        }
        if (!real) l = 0.0f;
        if (PERVOX) {
            float lv = l;
#pragma unroll
            for (int off = 1; off < G; off <<= 1) lv += __shfl_xor(lv, off, NRT_WAVE);
            if (lg == 0) per_voxel[v] = lv;
        }
        acc += l;
    }
    for (int off = 1; off < NRT_WAVE; off <<= 1) acc += __shfl_xor(acc, off, NRT_WAVE);
    __shared__ float red[CCE_BLOCK / NRT_WAVE];
    if ((threadIdx.x & (NRT_WAVE - 1)) == 0) red[threadIdx.x / NRT_WAVE] = acc;
    __syncthreads();
    float bsum = 0.0f;
    if (threadIdx.x == 0) {
        bsum = red[0];
        for (int i = 1; i < CCE_BLOCK / NRT_WAVE; ++i) bsum += red[i];
    }
    wcce_finish(part, bsum, fin);
}

// any C: one thread per voxel
// any channel count: a thread owns a voxel.  VP > 0: the block stages the rows of VP voxels of both tensors in LDS with coalesced
// loads and the threads read their rows from there (a lane reading its own row from memory touches one line per element:
// 26 ms for 4 x 160^3 x 33 against 0.8 ms for the 32-channel vector kernel); VP == 0: rows too wide for LDS, read from memory.
// The arithmetic per voxel is the same sequence either way.
template <typename T, bool LOGITS>
__global__ __launch_bounds__(CCE_BLOCK) void wcce_generic(const void *__restrict__ yt, const void *__restrict__ yp,
                                                          const float *__restrict__ w, long long n, int C, float smooth, int VP,
                                                          float *__restrict__ part, float *__restrict__ per_voxel, WcceFin fin) {
    extern __shared__ float cg_lds[];      // [2][VP * C]
    const float keep = 1.0f - smooth, add = smooth / (float)C;
    float acc = 0.0f;
    auto voxel = [&](auto tload, auto pload) -> float {
        float l = 0.0f;
        if (LOGITS) {
            float m = -INFINITY;
            for (int c = 0; c < C; ++c) m = fmaxf(m, pload(c));
            float se = 0.0f;
            for (int c = 0; c < C; ++c) se += expf(pload(c) - m);
            const float lse = logf(se);
            for (int c = 0; c < C; ++c) {
                float tt = (w ? w[c] : 1.0f) * tload(c);
                if (smooth != 0.0f) tt = tt * keep + add;
                l -= tt * ((pload(c) - m) - lse);
            }
        } else {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s += pload(c);
            for (int c = 0; c < C; ++c) {
                float q = pload(c) / s;
                q = fminf(fmaxf(q, KERAS_EPS), 1.0f - KERAS_EPS);
                float tt = (w ? w[c] : 1.0f) * tload(c);
                if (smooth != 0.0f) tt = tt * keep + add;
                l -= tt * logf(q);
            }
        }
        return l;
    };
    if (VP > 0) {
        float *st = cg_lds, *sp = cg_lds + (long long)VP * C;
        for (long long v0 = (long long)blockIdx.x * VP; v0 < n; v0 += (long long)gridDim.x * VP) {
            const int nv = (int)((n - v0) < VP ? (n - v0) : VP);
            const long long ne = (long long)nv * C;
            __syncthreads();
            for (long long i = threadIdx.x; i < ne; i += blockDim.x) {
                st[i] = Quad<T>::load1(yt, v0 * C + i);
                sp[i] = Quad<T>::load1(yp, v0 * C + i);
            }
            __syncthreads();
            if ((int)threadIdx.x < nv) {
                const float *rt = st + (long long)threadIdx.x * C, *rp = sp + (long long)threadIdx.x * C;
                const float l = voxel([&](int c) { return rt[c]; }, [&](int c) { return rp[c]; });
                if (per_voxel) per_voxel[v0 + threadIdx.x] = l;
                acc += l;
            }
        }
    } else {
        for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
            const float l = voxel([&](int c) { return Quad<T>::load1(yt, v * C + c); }, [&](int c) { return Quad<T>::load1(yp, v * C + c); });
            if (per_voxel) per_voxel[v] = l;
            acc += l;
        }
    }
    for (int off = 1; off < NRT_WAVE; off <<= 1) acc += __shfl_xor(acc, off, NRT_WAVE);
    __shared__ float red[CCE_BLOCK / NRT_WAVE];
    if ((threadIdx.x & (NRT_WAVE - 1)) == 0) red[threadIdx.x / NRT_WAVE] = acc;
    __syncthreads();
    float bsum = 0.0f;
    if (threadIdx.x == 0) {
        bsum = red[0];
        for (int i = 1; i < CCE_BLOCK / NRT_WAVE; ++i) bsum += red[i];
    }
    wcce_finish(part, bsum, fin);
}

bool vec_channels(int C) { return C % 4 == 0 && C >= 4 && C <= 256; }       // lane-groups of the next power of two >= C / 4 lanes

template <int G, typename T>
void launch_vec(const void *t, const void *p, const float *w, long long n, int logits, float smooth, unsigned nblk,
                float *part, float *pv, hipStream_t st, int Gr, WcceFin fin) {
    dim3 grid(nblk), blk(CCE_BLOCK);
    if (logits) {
        if (pv) hipLaunchKernelGGL((wcce_vec<G, T, true, true>), grid, blk, 0, st, t, p, w, n, smooth, part, pv, Gr, fin);
        else hipLaunchKernelGGL((wcce_vec<G, T, true, false>), grid, blk, 0, st, t, p, w, n, smooth, part, pv, Gr, fin);
    } else {
        if (pv) hipLaunchKernelGGL((wcce_vec<G, T, false, true>), grid, blk, 0, st, t, p, w, n, smooth, part, pv, Gr, fin);
        else hipLaunchKernelGGL((wcce_vec<G, T, false, false>), grid, blk, 0, st, t, p, w, n, smooth, part, pv, Gr, fin);
    }
}

template <typename T>
void launch_any(const void *t, const void *p, const float *w, long long n, int C, int logits, float smooth,
                bool aligned, unsigned &nblk, float *part, float *pv, hipStream_t st, WcceFin fin) {
    if (vec_channels(C) && aligned) {
        const int Gr = C / 4;
        int G = 1;
        while (G < Gr) G <<= 1;
        long long nb = (n + (CCE_BLOCK / G) * 4 - 1) / ((CCE_BLOCK / G) * 4);
        nblk = (unsigned)(nb < 1 ? 1 : (nb > CCE_MAX_BLOCKS ? CCE_MAX_BLOCKS : nb));
        switch (G) {
            case 1: launch_vec<1, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
            case 2: launch_vec<2, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
            case 4: launch_vec<4, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
            case 8: launch_vec<8, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
            case 16: launch_vec<16, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
            case 32: launch_vec<32, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
            default: launch_vec<64, T>(t, p, w, n, logits, smooth, nblk, part, pv, st, Gr, fin); break;
        }
    } else {
        // rows of VP voxels of both tensors in LDS (48 KB): 256 voxels up to 24 channels, fewer for wider rows, none beyond 768
        int VP = CCE_BLOCK;
        while (VP > 8 && (size_t)2 * VP * C * 4 > 48 * 1024) VP >>= 1;
        if ((size_t)2 * VP * C * 4 > 48 * 1024) VP = 0;
        const int per = VP > 0 ? VP : CCE_BLOCK;
        long long nb = (n + per - 1) / per;
        nblk = (unsigned)(nb < 1 ? 1 : (nb > CCE_MAX_BLOCKS ? CCE_MAX_BLOCKS : nb));
        const size_t shm = (size_t)2 * VP * C * 4;
        if (logits) hipLaunchKernelGGL((wcce_generic<T, true>), dim3(nblk), dim3(CCE_BLOCK), shm, st, t, p, w, n, C, smooth, VP, part, pv, fin);
        else hipLaunchKernelGGL((wcce_generic<T, false>), dim3(nblk), dim3(CCE_BLOCK), shm, st, t, p, w, n, C, smooth, VP, part, pv, fin);
    }
}

}  // namespace

extern "C" size_t nrt_wcce_workspace_bytes(long long nvox_total, int channels) {
    (void)nvox_total; (void)channels;
    return (size_t)CCE_MAX_BLOCKS * sizeof(float) + 256;
}

namespace {
int wcce_impl(const void *y_true, const void *y_pred, int dtype, const float *label_weights, long long nvox_total, int channels, int from_logits,
              float label_smoothing, double divide_by, float *loss_sum, float *per_voxel, void *workspace, size_t workspace_bytes, void *stream) {
    if (!y_true || !y_pred || !loss_sum) return NRT_ERR_INVALID_ARG;
    if (nvox_total < 0 || channels < 1 || !(divide_by > 0.0)) return NRT_ERR_INVALID_ARG;
    if (dtype != NRT_DT_F32 && dtype != NRT_DT_BF16) return NRT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < nrt_wcce_workspace_bytes(nvox_total, channels)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    float *part = (float *)workspace;
    unsigned nblk = 1;
    unsigned *slot = nrt_ring_slot(st);
    if (!slot) return NRT_ERR_WORKSPACE;                       // the counter pool is exhausted (api.hip)
    WcceFin fin = {loss_sum, slot + NRT_RING_CCE_OFF, divide_by};
    const bool aligned = (((uintptr_t)y_true | (uintptr_t)y_pred) & 15) == 0;
    if (dtype == NRT_DT_F32)
        launch_any<float>(y_true, y_pred, label_weights, nvox_total, channels, from_logits, label_smoothing, aligned,
                          nblk, part, per_voxel, st, fin);
    else
        launch_any<unsigned short>(y_true, y_pred, label_weights, nvox_total, channels, from_logits, label_smoothing,
                                   aligned, nblk, part, per_voxel, st, fin);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
}  // namespace

extern "C" int nrt_wcce(const void *y_true, const void *y_pred, int dtype, const float *label_weights,
                        long long nvox_total, int channels, int from_logits, float label_smoothing, float *loss_sum,
                        float *per_voxel, void *workspace, size_t workspace_bytes, void *stream) {
    return wcce_impl(y_true, y_pred, dtype, label_weights, nvox_total, channels, from_logits, label_smoothing, 1.0, loss_sum, per_voxel,
                     workspace, workspace_bytes, stream);
}

// the same with the reduction of Keras' default ('sum_over_batch_size', metrics.py:650 -> tf.keras.losses.CategoricalCrossentropy):
// loss_sum[0] = float32(sum) / float32(divide_by), formed by the block that finishes the sum -- no second launch for the division
extern "C" int nrt_wcce_mean(const void *y_true, const void *y_pred, int dtype, const float *label_weights,
                             long long nvox_total, int channels, int from_logits, float label_smoothing, double divide_by,
                             float *loss_mean, float *per_voxel, void *workspace, size_t workspace_bytes, void *stream) {
    return wcce_impl(y_true, y_pred, dtype, label_weights, nvox_total, channels, from_logits, label_smoothing, divide_by, loss_mean, per_voxel,
                     workspace, workspace_bytes, stream);
}
