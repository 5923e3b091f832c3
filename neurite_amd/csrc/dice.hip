// Dice kernels for gfx950 (MI355X).
//
// Replaces neurite/tf/metrics.py:415-482: TensorFlow runs 3 full reductions + 3 element-wise
// temporaries (+4 more reductions for the range asserts at :441-444); here y_true and y_pred are
// read exactly once (2*4*L bytes per voxel) and everything -- sum t*p, sum t^2, sum p^2, optional
// per-voxel renormalisation (:434-436), min/max for the range asserts, or for the hard path the
// arg-max + one-hot counting (:450-468) -- happens in registers.
//
// Reduction tree (no float atomics, so results are run-to-run identical):
//   lane accumulators -> wave64 xor-shuffles across lane-groups -> LDS across the block's 4 waves
//   -> one partial per block in the caller's workspace -> second-stage kernel adds the partials
//   in a fixed order in float64 and writes sums / dice.
//
// Layout: y [B, V, L] row-major (the reference's batch_channel_flatten view, utils.py:1175-1226).
// Fast path: L % 4 == 0, G = L/4 lanes own one voxel and each lane keeps a FIXED quad of labels
// (16-byte global_load_dwordx4 per lane; a wave64 reads 64/G consecutive voxels = 1 KiB contiguous).

#include "dice_reduce.h"

namespace {

// Storage type of the two maps: float32, or bfloat16 / float16 (ST = unsigned short / _Float16).  16-bit maps are widened to
// float32 in registers (exact) and everything below runs in float32, as for float32 maps: "16-bit storage, float32 math", the
// same convention as the weighted CCE (csrc/cce.hip) -- TensorFlow's own reductions of bfloat16 / float16 tensors accumulate in
// float32 as well (metrics.py:415-482 is dtype-agnostic).  Results are float32.
typedef unsigned nrt_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 nrt_h4 __attribute__((ext_vector_type(4)));
struct DiceBf16 { unsigned short bits; };
template <typename ST> struct DiceIn;
template <> struct DiceIn<float> {
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ nrt_f4 load4(const void *base, long long i4) { return __builtin_nontemporal_load((const nrt_f4 *)base + i4); }
    static __device__ __forceinline__ float load1(const void *base, long long i) { return ((const float *)base)[i]; }
};
template <> struct DiceIn<DiceBf16> {
    static constexpr int BYTES = 2;
    static __device__ __forceinline__ nrt_f4 load4(const void *base, long long i4) {
        const nrt_u2 t = __builtin_nontemporal_load((const nrt_u2 *)base + i4);
        return (nrt_f4){__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16), __uint_as_float(t[1] & 0xffff0000u)};
    }
    static __device__ __forceinline__ float load1(const void *base, long long i) { return __uint_as_float((unsigned)((const unsigned short *)base)[i] << 16); }
};
template <> struct DiceIn<_Float16> {
    static constexpr int BYTES = 2;
    static __device__ __forceinline__ nrt_f4 load4(const void *base, long long i4) {
        const nrt_h4 t = __builtin_nontemporal_load((const nrt_h4 *)base + i4);
        return (nrt_f4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    }
    static __device__ __forceinline__ float load1(const void *base, long long i) { return (float)((const _Float16 *)base)[i]; }
};

// ============================================================================================
// soft Dice, vectorised: G lanes per voxel
// ============================================================================================
template <int G, bool NORMALIZE, typename ST = float>
__global__ __launch_bounds__(DICE_BLOCK) void dice_soft_vec(const void *__restrict__ yt, const void *__restrict__ yp,
                                                            long long nvox, float *__restrict__ fpart,
                                                            float *__restrict__ mpart, int Gr, int diff) {
    // diff: the sums are taken of the difference map d = y_true - y_pred against itself (all three rows = sum d^2 per label: the
    // label-weighted squared error of metrics.py:653-692 in one pass over the two maps)
    // Gr <= G label quads per voxel are real (label counts 4 Gr that are no power of two: 12, 20, 24 ...): the lane-group keeps G
    // lanes, lanes lg >= Gr load nothing and contribute zeros -- rows stay coalesced (Gr x 16 bytes per voxel) and the reductions stay
    // power-of-two shuffles
    constexpr int NG = DICE_BLOCK / G;           // voxels per block pass
    constexpr int L = 4 * G;
    const int Lr = 4 * Gr;
    const int b = blockIdx.y;
    const char *t4 = (const char *)yt + (long long)b * nvox * Lr * DiceIn<ST>::BYTES;
    const char *p4 = (const char *)yp + (long long)b * nvox * Lr * DiceIn<ST>::BYTES;
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    const bool real = lg < Gr;
    const int lgc = real ? lg : 0;

    nrt_f4 stp = {0, 0, 0, 0}, stt = {0, 0, 0, 0}, spp = {0, 0, 0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;

    long long vbeg, vend;                                      // one contiguous range of voxels per block (nrt_block_range)
    nrt_block_range(nvox, NG, vbeg, vend);
    vbeg += g;
    const long long vstep = NG;
#pragma unroll 4
    for (long long v = vbeg; v < vend; v += vstep) {
        nrt_f4 t = DiceIn<ST>::load4(t4, v * Gr + lgc);
        nrt_f4 p = DiceIn<ST>::load4(p4, v * Gr + lgc);
        if (!real) { t = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f}; p = t; }
        if (diff) { t = t - p; p = t; }
        if (NORMALIZE) {
            // y / sum_l y with divide_no_nan (metrics.py:435-436); the label sum spans the lane-group
            float st = (t[0] + t[1]) + (t[2] + t[3]);
            float sp = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int off = 1; off < G; off <<= 1) {
                st += __shfl_xor(st, off, NRT_WAVE);
                sp += __shfl_xor(sp, off, NRT_WAVE);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = (st == 0.0f) ? 0.0f : t[k] / st;
                p[k] = (sp == 0.0f) ? 0.0f : p[k] / sp;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            stp[k] += t[k] * p[k];
            stt[k] += t[k] * t[k];
            spp[k] += p[k] * p[k];
            if (real) {
                mnt = fminf(mnt, t[k]); mxt = fmaxf(mxt, t[k]);
                mnp = fminf(mnp, p[k]); mxp = fmaxf(mxp, p[k]);
            }
        }
    }

    // ---- wave: combine the 64/G lane-groups (same label quad) ---------------------------------
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        stp[k] = wave_xor_add(stp[k], G);
        stt[k] = wave_xor_add(stt[k], G);
        spp[k] = wave_xor_add(spp[k], G);
    }
    for (int off = 1; off < NRT_WAVE; off <<= 1) {
        mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
        mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
    }
    // ---- block: 4 waves through LDS -----------------------------------------------------------
    __shared__ float red[DICE_BLOCK / NRT_WAVE][3 * L + 4];
    const int lane = threadIdx.x & (NRT_WAVE - 1), wv = threadIdx.x / NRT_WAVE;
    if (lane < G) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[wv][0 * L + 4 * lane + k] = stp[k];
            red[wv][1 * L + 4 * lane + k] = stt[k];
            red[wv][2 * L + 4 * lane + k] = spp[k];
        }
    }
    if (lane == 0) { red[wv][3 * L + 0] = mnt; red[wv][3 * L + 1] = mxt; red[wv][3 * L + 2] = mnp; red[wv][3 * L + 3] = mxp; }
    __syncthreads();
    const long long pbase = ((long long)b * gridDim.x + blockIdx.x);
    for (int i = threadIdx.x; i < 3 * Lr; i += DICE_BLOCK) {
        const int ii = (i / Lr) * L + i % Lr;                  // [3][L] in LDS -> [3][Lr] in the partial row
        float s = red[0][ii];
#pragma unroll
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2) s += red[w2][ii];
        fpart[pbase * 3 * Lr + i] = s;
    }
    if (threadIdx.x < 4) {
        float m = red[0][3 * L + threadIdx.x];
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2)
            m = (threadIdx.x & 1) ? fmaxf(m, red[w2][3 * L + threadIdx.x]) : fminf(m, red[w2][3 * L + threadIdx.x]);
        mpart[pbase * 4 + threadIdx.x] = m;
    }
}

// soft Dice, any L (slow path for label counts that are not 4*2^k): thread = (voxel row r, label li)
// inside a label chunk of up to 256 labels (blockIdx.z); consecutive threads read consecutive labels.
template <bool NORMALIZE, typename ST = float>
__global__ __launch_bounds__(DICE_BLOCK) void dice_soft_generic(const void *__restrict__ yt, const void *__restrict__ yp,
                                                                long long nvox, int L, float *__restrict__ fpart,
                                                                float *__restrict__ mpart, int diff) {
    __shared__ float sh[3 * DICE_BLOCK];
    __shared__ float mm[DICE_BLOCK / NRT_WAVE][4];
    const int b = blockIdx.y;
    const char *t = (const char *)yt + (long long)b * nvox * L * DiceIn<ST>::BYTES;
    const char *p = (const char *)yp + (long long)b * nvox * L * DiceIn<ST>::BYTES;
    const int Lc = L < DICE_BLOCK ? L : DICE_BLOCK;       // labels in this chunk (last chunk may be short)
    const int R = DICE_BLOCK / Lc;                        // voxel rows per pass (1 when L >= 256)
    const int r = threadIdx.x / Lc, li = threadIdx.x % Lc;
    const int l = blockIdx.z * DICE_BLOCK + li;
    const bool active = (r < R) && (l < L);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;
    if (active) {
        for (long long v = (long long)blockIdx.x * R + r; v < nvox; v += (long long)gridDim.x * R) {
            float tv = DiceIn<ST>::load1(t, v * L + l), pv = DiceIn<ST>::load1(p, v * L + l);
            if (diff) { tv = tv - pv; pv = tv; }
            if (NORMALIZE) {
                float st = 0.0f, sp = 0.0f;
                for (int k = 0; k < L; ++k) { st += DiceIn<ST>::load1(t, v * L + k); sp += DiceIn<ST>::load1(p, v * L + k); }
                tv = (st == 0.0f) ? 0.0f : tv / st;
                pv = (sp == 0.0f) ? 0.0f : pv / sp;
            }
            a0 += tv * pv; a1 += tv * tv; a2 += pv * pv;
            mnt = fminf(mnt, tv); mxt = fmaxf(mxt, tv); mnp = fminf(mnp, pv); mxp = fmaxf(mxp, pv);
        }
    }
    sh[0 * DICE_BLOCK + threadIdx.x] = a0;
    sh[1 * DICE_BLOCK + threadIdx.x] = a1;
    sh[2 * DICE_BLOCK + threadIdx.x] = a2;
    for (int off = 1; off < NRT_WAVE; off <<= 1) {
        mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
        mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
    }
    if ((threadIdx.x & (NRT_WAVE - 1)) == 0) {
        mm[threadIdx.x / NRT_WAVE][0] = mnt; mm[threadIdx.x / NRT_WAVE][1] = mxt;
        mm[threadIdx.x / NRT_WAVE][2] = mnp; mm[threadIdx.x / NRT_WAVE][3] = mxp;
    }
    __syncthreads();
    const long long pblk = (long long)b * gridDim.x + blockIdx.x;
    if (threadIdx.x < Lc && l < L) {
        for (int k = 0; k < 3; ++k) {
            float s = 0.0f;
            for (int rr = 0; rr < R; ++rr) s += sh[k * DICE_BLOCK + rr * Lc + li];     // fixed order
            fpart[pblk * 3 * L + (long long)k * L + l] = s;
        }
    }
    if (threadIdx.x < 4) {
        float m = mm[0][threadIdx.x];
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2)
            m = (threadIdx.x & 1) ? fmaxf(m, mm[w2][threadIdx.x]) : fminf(m, mm[w2][threadIdx.x]);
        mpart[(pblk * gridDim.z + blockIdx.z) * 4 + threadIdx.x] = m;
    }
}

// ============================================================================================
// hard Dice from probabilities: arg-max (ties -> lowest label) + one-hot counting, G lanes/voxel
// ============================================================================================
// MINMAX: also the extrema of both inputs (the range asserts of metrics.py:439-444 without a second pass over the maps)
template <int G, bool MINMAX, typename ST = float>
__global__ __launch_bounds__(DICE_BLOCK) void dice_hard_vec(const void *__restrict__ yt, const void *__restrict__ yp,
                                                            long long nvox, unsigned *__restrict__ ipart,
                                                            float *__restrict__ mpart, int Gr) {
    constexpr int NG = DICE_BLOCK / G;
    constexpr int L = 4 * G;
    const int Lr = 4 * Gr;                         // Gr <= G real label quads per voxel (see dice_soft_vec); padding lanes never win the arg-max
    const int b = blockIdx.y;
    const char *t4 = (const char *)yt + (long long)b * nvox * Lr * DiceIn<ST>::BYTES;
    const char *p4 = (const char *)yp + (long long)b * nvox * Lr * DiceIn<ST>::BYTES;
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    const bool real = lg < Gr;
    const int lgc = real ? lg : 0;
    unsigned ntp[4] = {0, 0, 0, 0}, nt[4] = {0, 0, 0, 0}, np_[4] = {0, 0, 0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;

    long long vbeg, vend;
    nrt_block_range(nvox, NG, vbeg, vend);
#pragma unroll 2
    for (long long v = vbeg + g; v < vend; v += NG) {
        nrt_f4 t = DiceIn<ST>::load4(t4, v * Gr + lgc);
        nrt_f4 p = DiceIn<ST>::load4(p4, v * Gr + lgc);
        if (!real) { t = (nrt_f4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}; p = t; }
        if (MINMAX && real) {
            mnt = fminf(mnt, fminf(fminf(t[0], t[1]), fminf(t[2], t[3]))); mxt = fmaxf(mxt, fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])));
            mnp = fminf(mnp, fminf(fminf(p[0], p[1]), fminf(p[2], p[3]))); mxp = fmaxf(mxp, fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])));
        }
        float bt = t[0], bp = p[0];
        int at = 4 * lg, ap = 4 * lg;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (t[k] > bt) { bt = t[k]; at = 4 * lg + k; }      // strict > keeps the lowest index on ties
            if (p[k] > bp) { bp = p[k]; ap = 4 * lg + k; }
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            const float ot = __shfl_xor(bt, off, NRT_WAVE), op = __shfl_xor(bp, off, NRT_WAVE);
            const int oat = __shfl_xor(at, off, NRT_WAVE), oap = __shfl_xor(ap, off, NRT_WAVE);
            if (ot > bt || (ot == bt && oat < at)) { bt = ot; at = oat; }
            if (op > bp || (op == bp && oap < ap)) { bp = op; ap = oap; }
        }
        // every lane of the group now knows (at, ap); it counts only its own 4 labels
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lab = 4 * lg + k;
            nt[k] += (at == lab);
            np_[k] += (ap == lab);
            ntp[k] += (at == lab) & (ap == lab);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ntp[k] = wave_xor_add(ntp[k], G);
        nt[k] = wave_xor_add(nt[k], G);
        np_[k] = wave_xor_add(np_[k], G);
    }
    __shared__ unsigned red[DICE_BLOCK / NRT_WAVE][3 * L];
    __shared__ float redm[DICE_BLOCK / NRT_WAVE][4];
    const int lane = threadIdx.x & (NRT_WAVE - 1), wv = threadIdx.x / NRT_WAVE;
    if (lane < G) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[wv][0 * L + 4 * lane + k] = ntp[k];
            red[wv][1 * L + 4 * lane + k] = nt[k];
            red[wv][2 * L + 4 * lane + k] = np_[k];
        }
    }
    if (MINMAX) {
        for (int off = 1; off < NRT_WAVE; off <<= 1) {
            mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
            mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
        }
        if (lane == 0) { redm[wv][0] = mnt; redm[wv][1] = mxt; redm[wv][2] = mnp; redm[wv][3] = mxp; }
    }
    __syncthreads();
    const long long pbase = ((long long)b * gridDim.x + blockIdx.x);
    for (int i = threadIdx.x; i < 3 * Lr; i += DICE_BLOCK) {
        const int ii = (i / Lr) * L + i % Lr;
        unsigned s = red[0][ii];
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2) s += red[w2][ii];
        ipart[pbase * 3 * Lr + i] = s;
    }
    if (MINMAX && threadIdx.x < 4) {
        float m = redm[0][threadIdx.x];
        for (int w2 = 1; w2 < DICE_BLOCK / NRT_WAVE; ++w2)
            m = (threadIdx.x & 1) ? fmaxf(m, redm[w2][threadIdx.x]) : fminf(m, redm[w2][threadIdx.x]);
        mpart[pbase * 4 + threadIdx.x] = m;
    }
}

// extrema of `rows` block quadruples (min t, max t, min p, max p) -> minmax[4]
__global__ __launch_bounds__(256) void minmax_rows(const float *__restrict__ mpart, int rows, float *__restrict__ minmax) {
    __shared__ float sm[256][4];
    float m[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int k = threadIdx.x; k < rows; k += 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (i & 1) ? fmaxf(m[i], mpart[(long long)k * 4 + i]) : fminf(m[i], mpart[(long long)k * 4 + i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[threadIdx.x][i] = m[i];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int i = threadIdx.x;
        float r = sm[0][i];
        for (int k = 1; k < 256; ++k) r = (i & 1) ? fmaxf(r, sm[k][i]) : fminf(r, sm[k][i]);
        minmax[i] = r;
    }
}

// hard Dice from int32 label maps: per-block LDS histogram (integer atomics: order-independent)
// One LDS atomic per DISTINCT label of a wave: label maps are piecewise constant, so the 64 lanes of a wave hold one to three
// labels and per-lane atomics would all queue on the same LDS address.  The first four distinct labels are counted by ballot +
// popcount and added by one lane each; whatever is left takes the per-lane atomic.
__device__ __forceinline__ void wave_hist_add(unsigned *hist, int label, bool ok) {
    const int lane = threadIdx.x & (NRT_WAVE - 1);
    unsigned long long todo = __ballot(ok);
    for (int it = 0; it < 4 && todo; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lab = __shfl(label, leader, NRT_WAVE);
        const unsigned long long same = __ballot(ok && label == lab);
        if (lane == leader) atomicAdd(&hist[lab], (unsigned)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&hist[label], 1u);
}

// hard Dice, any L, from probabilities.  A block stages the rows of VP voxels (VP * L floats, both maps in turn) in LDS with
// coalesced 4-byte loads, then a thread scans its voxel's row for the arg-max (ties -> lowest label); the counts go to a per-block
// LDS histogram (one atomic per distinct label of a wave) and to `counts` once per block.  (The first version read the rows
// straight from memory, a lane per voxel -- 80-byte strides at 20 labels -- and sent three global atomics per VOXEL.)
template <typename ST>
__global__ __launch_bounds__(DICE_BLOCK) void dice_hard_prob_generic(const void *__restrict__ yt, const void *__restrict__ yp,
                                                                     long long nvox, int L, int VP, long long *counts) {
    extern __shared__ unsigned dh_lds[];    // hist [3 * L], rows [VP * L]
    unsigned *hist = dh_lds;
    float *rows = (float *)(dh_lds + 3 * L);
    const int b = blockIdx.y;
    const char *t = (const char *)yt + (long long)b * nvox * L * DiceIn<ST>::BYTES;
    const char *p = (const char *)yp + (long long)b * nvox * L * DiceIn<ST>::BYTES;
    unsigned long long *c = (unsigned long long *)counts + (long long)b * 3 * L;
    for (int i = threadIdx.x; i < 3 * L; i += blockDim.x) hist[i] = 0u;
    auto argmax_of = [&](const char *src, long long v0, int nv) -> int {
        const long long n = (long long)nv * L;
        __syncthreads();                                     // the previous scan is over
        for (long long i = threadIdx.x; i < n; i += blockDim.x) rows[i] = DiceIn<ST>::load1(src, v0 * L + i);
        __syncthreads();
        int am = 0;
        if ((int)threadIdx.x < nv) {
            const float *r = rows + (long long)threadIdx.x * L;
            float best = r[0];
            for (int l = 1; l < L; ++l) { const float x = r[l]; if (x > best) { best = x; am = l; } }
        }
        return am;
    };
    for (long long v0 = (long long)blockIdx.x * VP; v0 < nvox; v0 += (long long)gridDim.x * VP) {
        const int nv = (int)((nvox - v0) < VP ? (nvox - v0) : VP);
        const int at = argmax_of(t, v0, nv);
        const int ap = argmax_of(p, v0, nv);
        const bool live = (int)threadIdx.x < nv;
        wave_hist_add(hist + L, at, live);
        wave_hist_add(hist + 2 * L, ap, live);
        wave_hist_add(hist, at, live && at == ap);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * L; i += blockDim.x)
        if (hist[i]) atomicAdd(&c[i], (unsigned long long)hist[i]);
}

// the same for label counts whose rows do not fit in LDS: rows read from memory, global atomics
template <typename ST>
__global__ __launch_bounds__(DICE_BLOCK) void dice_hard_prob_direct(const void *__restrict__ yt, const void *__restrict__ yp,
                                                                    long long nvox, int L, long long *counts) {
    const int b = blockIdx.y;
    const char *t = (const char *)yt + (long long)b * nvox * L * DiceIn<ST>::BYTES;
    const char *p = (const char *)yp + (long long)b * nvox * L * DiceIn<ST>::BYTES;
    unsigned long long *c = (unsigned long long *)counts + (long long)b * 3 * L;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long long)gridDim.x * blockDim.x) {
        int at = 0, ap = 0;
        float bt = DiceIn<ST>::load1(t, v * L), bp = DiceIn<ST>::load1(p, v * L);
        for (int l = 1; l < L; ++l) {
            const float tv = DiceIn<ST>::load1(t, v * L + l), pv = DiceIn<ST>::load1(p, v * L + l);
            if (tv > bt) { bt = tv; at = l; }
            if (pv > bp) { bp = pv; ap = l; }
        }
        atomicAdd(&c[L + at], 1ull);
        atomicAdd(&c[2 * L + ap], 1ull);
        if (at == ap) atomicAdd(&c[at], 1ull);
    }
}

// PARTIAL: the block's histogram is written as one row of ipart [B][nblk][3 L] (reduced by reduce_rows, no global atomics: 2000
// blocks adding to the same 96 addresses serialise in L2); otherwise it is added to counts (zero-filled by the caller)
template <bool PARTIAL>
__global__ __launch_bounds__(DICE_BLOCK) void dice_hard_label(const int *__restrict__ yt, const int *__restrict__ yp,
                                                              long long nvox, int L, int use_lds, long long *counts,
                                                              unsigned *__restrict__ ipart) {
    extern __shared__ unsigned hist[];      // [3*L] when use_lds
    const int b = blockIdx.y;
    const int *t = yt + (long long)b * nvox;
    const int *p = yp + (long long)b * nvox;
    unsigned long long *c = (unsigned long long *)counts + (long long)b * 3 * L;
    if (use_lds) {
        for (int i = threadIdx.x; i < 3 * L; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
    }
    auto count = [&](int a, int q, bool live) {
        const bool oka = live && a >= 0 && a < L, okq = live && q >= 0 && q < L;   // tf.one_hot: out of range -> zero row
        if (use_lds) {
            wave_hist_add(hist + L, a, oka);
            wave_hist_add(hist + 2 * L, q, okq);
            wave_hist_add(hist, a, oka && okq && a == q);
        } else {
            if (oka) atomicAdd(&c[L + a], 1ull);
            if (okq) atomicAdd(&c[2 * L + q], 1ull);
            if (oka && okq && a == q) atomicAdd(&c[a], 1ull);
        }
    };
    // 16-byte loads over the aligned body (every lane of a wave runs every round: the ballots need the whole wave)
    const long long n4 = ((((uintptr_t)t | (uintptr_t)p) & 15) == 0) ? nvox / 4 : 0;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long v0 = (long long)blockIdx.x * blockDim.x; v0 < n4; v0 += step) {
        const long long v = v0 + threadIdx.x;
        const bool live = v < n4;
        nrt_i4 a = {0, 0, 0, 0}, q = {0, 0, 0, 0};
        if (live) { a = ((const nrt_i4 *)t)[v]; q = ((const nrt_i4 *)p)[v]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) count(a[j], q[j], live);
    }
    for (long long v0 = n4 * 4 + (long long)blockIdx.x * blockDim.x; v0 < nvox; v0 += step) {
        const long long v = v0 + threadIdx.x;
        const bool live = v < nvox;
        count(live ? t[v] : 0, live ? p[v] : 0, live);
    }
    if (use_lds) {
        __syncthreads();
        if (PARTIAL) {
            const long long pbase = ((long long)b * gridDim.x + blockIdx.x);
            for (int i = threadIdx.x; i < 3 * L; i += blockDim.x) ipart[pbase * 3 * L + i] = hist[i];
        } else {
            for (int i = threadIdx.x; i < 3 * L; i += blockDim.x)
                if (hist[i]) atomicAdd(&c[i], (unsigned long long)hist[i]);
        }
    }
}

// Label maps, few labels (3 L counters x 256 lanes fit in LDS: L <= 40): every LANE owns a private histogram in LDS
// (counter (kind, label) of lane j at word (kind * L + label) * 256 + j: the lanes of an instruction hit different banks whatever
// their labels are), so counting is three conflict-free ds_add per voxel and no ballot / shuffle / same-address queue -- blob-like
// label maps put whole waves on one label, which made the shared-histogram form above run at 0.15 of the HBM roof
// (profiles/archive/r02_smallc/secondary_bench.jsonl).  Each lane streams 16-byte quads of both maps, four quads in flight.
__global__ __launch_bounds__(DICE_BLOCK) void dice_hard_label_private(const int *__restrict__ yt, const int *__restrict__ yp,
                                                                      long long nvox, int L, unsigned *__restrict__ ipart) {
    extern __shared__ unsigned hist[];      // [3 * L][256]
    const int b = blockIdx.y;
    const int *t = yt + (long long)b * nvox;
    const int *p = yp + (long long)b * nvox;
    for (int i = threadIdx.x; i < 3 * L * DICE_BLOCK / 4; i += DICE_BLOCK) ((nrt_i4 *)hist)[i] = (nrt_i4){0, 0, 0, 0};
    __syncthreads();
    unsigned *mine = hist + threadIdx.x;
    const unsigned uL = (unsigned)L;
    auto count = [&](int a, int q) {
        // tf.one_hot: an out-of-range id is an all-zero row.  The address is clamped, the increment is 0 or 1: no branch
        const unsigned ua = (unsigned)a, uq = (unsigned)q;
        const unsigned oka = ua < uL ? 1u : 0u, okq = uq < uL ? 1u : 0u;
        const unsigned ca = min(ua, uL - 1u), cq = min(uq, uL - 1u);
        atomicAdd(mine + (uL + ca) * DICE_BLOCK, oka);
        atomicAdd(mine + (2u * uL + cq) * DICE_BLOCK, okq);
        atomicAdd(mine + ca * DICE_BLOCK, (oka & okq & (ua == uq ? 1u : 0u)));
    };
    const long long n4 = ((((uintptr_t)t | (uintptr_t)p) & 15) == 0) ? nvox / 4 : 0;
    const long long step = (long long)gridDim.x * DICE_BLOCK;
    long long v = (long long)blockIdx.x * DICE_BLOCK + threadIdx.x;
    for (; v + 3 * step < n4; v += 4 * step) {
        nrt_i4 a[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = __builtin_nontemporal_load((const nrt_i4 *)t + v + u * step); q[u] = __builtin_nontemporal_load((const nrt_i4 *)p + v + u * step); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) count(a[u][j], q[u][j]);
        }
    }
    for (; v < n4; v += step) {
        const nrt_i4 a = ((const nrt_i4 *)t)[v], q = ((const nrt_i4 *)p)[v];
#pragma unroll
        for (int j = 0; j < 4; ++j) count(a[j], q[j]);
    }
    for (long long w = n4 * 4 + (long long)blockIdx.x * DICE_BLOCK + threadIdx.x; w < nvox; w += step) count(t[w], p[w]);
    __syncthreads();
    // fold the 256 private histograms: two threads per counter row (3 L rows of 256 words), 128 words each, the start staggered by
    // the row so that the threads of a wave read different banks
    const long long pbase = ((long long)b * gridDim.x + blockIdx.x);
    for (int r0 = 0; r0 < 3 * L; r0 += DICE_BLOCK / 2) {
        const int r = r0 + (threadIdx.x >> 1), half = threadIdx.x & 1;
        unsigned sum = 0;
        if (r < 3 * L)
            for (int k = 0; k < DICE_BLOCK / 2; ++k) sum += hist[r * DICE_BLOCK + half * (DICE_BLOCK / 2) + ((k + r) & (DICE_BLOCK / 2 - 1))];
        sum += __shfl_xor(sum, 1, NRT_WAVE);
        if (r < 3 * L && half == 0) ipart[pbase * 3 * L + r] = sum;
    }
}

__global__ void dice_counts_reduce(const long long *__restrict__ gcnt, int ngrp, int L, long long *counts) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 3 * L; i += blockDim.x) {
        long long s = 0;
        for (int k = 0; k < ngrp; ++k) s += gcnt[((long long)b * ngrp + k) * 3 * L + i];
        counts[(long long)b * 3 * L + i] = s;
    }
}

// dice from exact counts: one-hot squares are the counts themselves; float32 like the reference
__global__ void dice_from_counts(const long long *__restrict__ counts, int L, float eps, float *__restrict__ dice) {
    const int b = blockIdx.x;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const long long *c = counts + (long long)b * 3 * L;
        const float top = nrt_mul(2.0f, (float)c[l]);
        const float bottom = nrt_add((float)c[L + l], (float)c[2 * L + l]);
        float d;
        if (eps > 0.0f) d = nrt_add(top, eps) / nrt_add(bottom, eps);
        else d = (bottom == 0.0f) ? 0.0f : top / bottom;
        dice[(long long)b * L + l] = d;
    }
}

__global__ void dice_from_sums(const float *__restrict__ sums, int L, float eps, float *__restrict__ dice) {
    const int b = blockIdx.x;
    const float *s = sums + (long long)b * 3 * L;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const float top = nrt_mul(2.0f, s[l]);
        const float bottom = nrt_add(s[L + l], s[2 * L + l]);
        float d;
        if (eps > 0.0f) d = nrt_add(top, eps) / nrt_add(bottom, eps);
        else d = (bottom == 0.0f) ? 0.0f : top / bottom;
        dice[(long long)b * L + l] = d;
    }
}

bool vec_labels(int L) { return L % 4 == 0 && L >= 4 && L <= 256; }      // lane-groups of the next power of two >= L / 4 lanes
int vec_group(int L) { int g = 1; while (g < L / 4) g <<= 1; return g; }

template <int G, typename ST>
void launch_soft_vec(const void *t, const void *p, long long nvox, int batch, int normalize, unsigned nblk,
                     const DiceWs &w, hipStream_t st, int Gr) {
    dim3 grid(nblk, batch);
    if (normalize == 1) hipLaunchKernelGGL((dice_soft_vec<G, true, ST>), grid, dim3(DICE_BLOCK), 0, st, t, p, nvox, w.fpart, w.mpart, Gr, 0);
    else hipLaunchKernelGGL((dice_soft_vec<G, false, ST>), grid, dim3(DICE_BLOCK), 0, st, t, p, nvox, w.fpart, w.mpart, Gr, normalize == 2);
}

template <int G, typename ST>
void launch_hard_vec(const void *t, const void *p, long long nvox, int batch, unsigned nblk, const DiceWs &w,
                     bool minmax, hipStream_t st, int Gr) {
    if (minmax) hipLaunchKernelGGL((dice_hard_vec<G, true, ST>), dim3(nblk, batch), dim3(DICE_BLOCK), 0, st, t, p, nvox, w.ipart, w.mpart, Gr);
    else hipLaunchKernelGGL((dice_hard_vec<G, false, ST>), dim3(nblk, batch), dim3(DICE_BLOCK), 0, st, t, p, nvox, w.ipart, w.mpart, Gr);
}

// normalize: 0 plain, 1 y / sum_l y first, 2 the difference map against itself (nrt_sqdiff_sums_f32)
template <typename ST>
int dice_soft_impl(const void *y_true, const void *y_pred, long long nvox, int nlabels, int batch, int normalize, float laplace_smoothing,
                   float *sums, float *dice, float *minmax, void *workspace, size_t workspace_bytes, void *stream) {
    if (!y_true || !y_pred || !sums || !dice) return NRT_ERR_INVALID_ARG;
    if (nvox < 0 || nlabels < 1 || batch < 1 || batch > 65535 || normalize < 0 || normalize > 2) return NRT_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dice_ws_bytes(nlabels, batch)) return NRT_ERR_WORKSPACE;
    if (nlabels > 4096) return NRT_ERR_UNSUPPORTED;
    hipStream_t st = nrt_stream(stream);
    DiceWs w = dice_ws_carve(workspace, nlabels, batch);
    unsigned nblk, gz = 1;
    const bool aligned = (((uintptr_t)y_true | (uintptr_t)y_pred) & (4 * DiceIn<ST>::BYTES - 1)) == 0;      // 4 labels per lane access
    if (vec_labels(nlabels) && aligned) {
        const int G = vec_group(nlabels), Gr = nlabels / 4;
        nblk = dice_num_blocks(nvox, (DICE_BLOCK / G) * 4);
        switch (G) {
            case 1: launch_soft_vec<1, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
            case 2: launch_soft_vec<2, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
            case 4: launch_soft_vec<4, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
            case 8: launch_soft_vec<8, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
            case 16: launch_soft_vec<16, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
            case 32: launch_soft_vec<32, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
            default: launch_soft_vec<64, ST>(y_true, y_pred, nvox, batch, normalize, nblk, w, st, Gr); break;
        }
    } else {
        const int Lc = nlabels < DICE_BLOCK ? nlabels : DICE_BLOCK;
        gz = (unsigned)((nlabels + DICE_BLOCK - 1) / DICE_BLOCK);
        nblk = dice_num_blocks(nvox, DICE_BLOCK / Lc);
        if (nblk > DICE_MAX_BLOCKS / gz) nblk = DICE_MAX_BLOCKS / gz;
        dim3 grid(nblk, batch, gz);
        if (normalize == 1) hipLaunchKernelGGL((dice_soft_generic<true, ST>), grid, dim3(DICE_BLOCK), 0, st, y_true, y_pred, nvox, nlabels, w.fpart, w.mpart, 0);
        else hipLaunchKernelGGL((dice_soft_generic<false, ST>), grid, dim3(DICE_BLOCK), 0, st, y_true, y_pred, nvox, nlabels, w.fpart, w.mpart, normalize == 2);
    }
    NRT_CHECK_LAUNCH();
    return dice_finalize_soft(w, nblk, gz, batch, nlabels, laplace_smoothing, sums, dice, minmax, st);
}

template <typename ST>
int dice_hard_prob_impl(const void *y_true, const void *y_pred, long long nvox, int nlabels, int batch, float laplace_smoothing,
                        long long *counts, float *dice, float *minmax, void *workspace, size_t workspace_bytes, void *stream) {
    if (!y_true || !y_pred || !counts || !dice) return NRT_ERR_INVALID_ARG;
    if (nvox < 0 || nlabels < 1 || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dice_ws_bytes(nlabels, batch)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    DiceWs w = dice_ws_carve(workspace, nlabels, batch);
    const bool aligned = (((uintptr_t)y_true | (uintptr_t)y_pred) & (4 * DiceIn<ST>::BYTES - 1)) == 0;
    if (vec_labels(nlabels) && aligned) {
        const int G = vec_group(nlabels), Gr = nlabels / 4;
        const unsigned nblk = dice_num_blocks(nvox, (DICE_BLOCK / G) * 4);
        switch (G) {
            case 1: launch_hard_vec<1, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
            case 2: launch_hard_vec<2, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
            case 4: launch_hard_vec<4, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
            case 8: launch_hard_vec<8, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
            case 16: launch_hard_vec<16, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
            case 32: launch_hard_vec<32, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
            default: launch_hard_vec<64, ST>(y_true, y_pred, nvox, batch, nblk, w, minmax != nullptr, st, Gr); break;
        }
        NRT_CHECK_LAUNCH();
        const int ngrp = ((int)nblk + RED_ROWS - 1) / RED_ROWS;
        hipLaunchKernelGGL((reduce_rows<unsigned, long long>), dim3(batch, ngrp), dim3(256), 0, st,
                           (const unsigned *)w.ipart, (int)nblk, 3 * nlabels, (long long *)w.gsum,
                           (const float *)nullptr, (float *)nullptr, 0);
        NRT_CHECK_LAUNCH();
        hipLaunchKernelGGL(dice_counts_reduce, dim3(batch), dim3(256), 0, st, (const long long *)w.gsum, ngrp,
                           nlabels, counts);
        if (minmax) {
            NRT_CHECK_LAUNCH();
            hipLaunchKernelGGL(minmax_rows, dim3(1), dim3(256), 0, st, (const float *)w.mpart, (int)(nblk * (unsigned)batch), minmax);
        }
    } else {
        if (minmax) return NRT_ERR_UNSUPPORTED;       // the generic kernel has no extrema: the caller runs the soft pass for them
        if (nrt_zero_async(counts, sizeof(long long) * (size_t)batch * 3 * nlabels, st) != hipSuccess)
            return NRT_ERR_LAUNCH;
        // rows of VP voxels in LDS: VP = 256 while 256 * L floats + the histogram fit in 48 KB, fewer voxels for wide rows
        int VP = DICE_BLOCK;
        while (VP > 8 && ((size_t)VP * nlabels + 3 * (size_t)nlabels) * 4 > 48 * 1024) VP >>= 1;
        if (((size_t)VP * nlabels + 3 * (size_t)nlabels) * 4 <= 48 * 1024) {
            long long nb = (nvox + VP - 1) / VP;
            if (nb > 1024) nb = 1024;                             // one atomic per label and block at the end
            if (nb < 1) nb = 1;
            hipLaunchKernelGGL(dice_hard_prob_generic<ST>, dim3((unsigned)nb, batch), dim3(DICE_BLOCK),
                               ((size_t)VP * nlabels + 3 * (size_t)nlabels) * 4, st, y_true, y_pred, nvox, nlabels, VP, counts);
        } else {
            hipLaunchKernelGGL(dice_hard_prob_direct<ST>, dim3(dice_num_blocks(nvox, DICE_BLOCK), batch), dim3(DICE_BLOCK), 0, st, y_true, y_pred,
                               nvox, nlabels, counts);
        }
    }
    NRT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_from_counts, dim3(batch), dim3(256), 0, st, (const long long *)counts, nlabels,
                       laplace_smoothing, dice);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

}  // namespace

extern "C" size_t nrt_dice_workspace_bytes(long long nvox, int nlabels, int batch) {
    (void)nvox;
    if (nlabels < 1 || batch < 1) return 0;
    return dice_ws_bytes(nlabels, batch);
}

extern "C" int nrt_dice_soft_f32(const float *y_true, const float *y_pred, long long nvox, int nlabels, int batch,
                                 int normalize, float laplace_smoothing, float *sums, float *dice, float *minmax,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    return dice_soft_impl<float>(y_true, y_pred, nvox, nlabels, batch, normalize != 0, laplace_smoothing, sums, dice, minmax, workspace,
                                 workspace_bytes, stream);
}

extern "C" int nrt_sqdiff_sums_f32(const float *a, const float *b, long long nvox, int nlabels, int batch, float *sums, float *scratch,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    return dice_soft_impl<float>(a, b, nvox, nlabels, batch, 2, 0.0f, sums, scratch, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int nrt_dice_soft(const void *y_true, const void *y_pred, int dtype, long long nvox, int nlabels, int batch, int normalize,
                             float laplace_smoothing, float *sums, float *dice, float *minmax, void *workspace, size_t workspace_bytes,
                             void *stream) {
    normalize = normalize != 0;
    switch (dtype) {
        case NRT_DT_F32: return dice_soft_impl<float>(y_true, y_pred, nvox, nlabels, batch, normalize, laplace_smoothing, sums, dice, minmax,
                                                      workspace, workspace_bytes, stream);
        case NRT_DT_BF16: return dice_soft_impl<DiceBf16>(y_true, y_pred, nvox, nlabels, batch, normalize, laplace_smoothing, sums, dice,
                                                          minmax, workspace, workspace_bytes, stream);
        case NRT_DT_F16: return dice_soft_impl<_Float16>(y_true, y_pred, nvox, nlabels, batch, normalize, laplace_smoothing, sums, dice,
                                                         minmax, workspace, workspace_bytes, stream);
        default: return NRT_ERR_UNSUPPORTED;
    }
}

extern "C" int nrt_dice_hard_prob_f32(const float *y_true, const float *y_pred, long long nvox, int nlabels, int batch,
                                      float laplace_smoothing, long long *counts, float *dice, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return dice_hard_prob_impl<float>(y_true, y_pred, nvox, nlabels, batch, laplace_smoothing, counts, dice, nullptr, workspace,
                                      workspace_bytes, stream);
}

extern "C" int nrt_dice_hard_prob_minmax_f32(const float *y_true, const float *y_pred, long long nvox, int nlabels, int batch,
                                             float laplace_smoothing, long long *counts, float *dice, float *minmax,
                                             void *workspace, size_t workspace_bytes, void *stream) {
    return dice_hard_prob_impl<float>(y_true, y_pred, nvox, nlabels, batch, laplace_smoothing, counts, dice, minmax, workspace,
                                      workspace_bytes, stream);
}

extern "C" int nrt_dice_hard_prob(const void *y_true, const void *y_pred, int dtype, long long nvox, int nlabels, int batch,
                                  float laplace_smoothing, long long *counts, float *dice, float *minmax, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    switch (dtype) {
        case NRT_DT_F32: return dice_hard_prob_impl<float>(y_true, y_pred, nvox, nlabels, batch, laplace_smoothing, counts, dice, minmax,
                                                           workspace, workspace_bytes, stream);
        case NRT_DT_BF16: return dice_hard_prob_impl<DiceBf16>(y_true, y_pred, nvox, nlabels, batch, laplace_smoothing, counts, dice, minmax,
                                                               workspace, workspace_bytes, stream);
        case NRT_DT_F16: return dice_hard_prob_impl<_Float16>(y_true, y_pred, nvox, nlabels, batch, laplace_smoothing, counts, dice, minmax,
                                                              workspace, workspace_bytes, stream);
        default: return NRT_ERR_UNSUPPORTED;
    }
}

extern "C" int nrt_dice_hard_label_i32(const int32_t *y_true, const int32_t *y_pred, long long nvox, int nlabels,
                                       int batch, float laplace_smoothing, long long *counts, float *dice,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    if (!y_true || !y_pred || !counts || !dice) return NRT_ERR_INVALID_ARG;
    if (nvox < 0 || nlabels < 1 || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    hipStream_t st = nrt_stream(stream);
    const int use_lds = (size_t)3 * nlabels * sizeof(unsigned) <= 64 * 1024;
    const size_t shm = use_lds ? (size_t)3 * nlabels * sizeof(unsigned) : 0;
    unsigned nblk = dice_num_blocks(nvox, DICE_BLOCK * 8);
    // with a workspace (nrt_dice_workspace_bytes) the block histograms are reduced as rows; without one they are added with
    // global atomics (the round-1 form)
    const bool partial = use_lds && workspace && workspace_bytes >= dice_ws_bytes(nlabels, batch);
    if (partial) {
        if (nblk > 512u) nblk = 512u;
        DiceWs w = dice_ws_carve(workspace, nlabels, batch);
        const size_t shm_private = (size_t)3 * nlabels * DICE_BLOCK * sizeof(unsigned);
        if (shm_private <= 120 * 1024 && nvox >= 65536) {
            // lane-private histograms: one block per CU is resident (LDS), so size the grid to the chip
            // (function attributes are per device and this may be called from several host threads: set on every launch, as the
            // other large-LDS kernels do)
            if (hipFuncSetAttribute((const void *)dice_hard_label_private, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess)
                return NRT_ERR_LAUNCH;
            const unsigned per_batch = (unsigned)max(1, (int)(512 / batch));
            nblk = min(nblk, per_batch);
            hipLaunchKernelGGL(dice_hard_label_private, dim3(nblk, batch), dim3(DICE_BLOCK), shm_private, st, (const int *)y_true,
                               (const int *)y_pred, nvox, nlabels, w.ipart);
        } else
        hipLaunchKernelGGL((dice_hard_label<true>), dim3(nblk, batch), dim3(DICE_BLOCK), shm, st, (const int *)y_true,
                           (const int *)y_pred, nvox, nlabels, use_lds, counts, w.ipart);
        NRT_CHECK_LAUNCH();
        const int ngrp = ((int)nblk + RED_ROWS - 1) / RED_ROWS;
        hipLaunchKernelGGL((reduce_rows<unsigned, long long>), dim3(batch, ngrp), dim3(256), 0, st,
                           (const unsigned *)w.ipart, (int)nblk, 3 * nlabels, (long long *)w.gsum,
                           (const float *)nullptr, (float *)nullptr, 0);
        NRT_CHECK_LAUNCH();
        hipLaunchKernelGGL(dice_counts_reduce, dim3(batch), dim3(256), 0, st, (const long long *)w.gsum, ngrp,
                           nlabels, counts);
    } else {
        if (nrt_zero_async(counts, sizeof(long long) * (size_t)batch * 3 * nlabels, st) != hipSuccess)
            return NRT_ERR_LAUNCH;
        hipLaunchKernelGGL((dice_hard_label<false>), dim3(nblk, batch), dim3(DICE_BLOCK), shm, st, (const int *)y_true,
                           (const int *)y_pred, nvox, nlabels, use_lds, counts, (unsigned *)nullptr);
    }
    NRT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_from_counts, dim3(batch), dim3(256), 0, st, (const long long *)counts, nlabels,
                       laplace_smoothing, dice);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_dice_from_sums_f32(const float *sums, int nlabels, int batch, float laplace_smoothing, float *dice,
                                      void *stream) {
    if (!sums || !dice || nlabels < 1 || batch < 1) return NRT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(dice_from_sums, dim3(batch), dim3(256), 0, nrt_stream(stream), sums, nlabels, laplace_smoothing, dice);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// [sum over all (b, l) of dice * weights, number of entries] in one launch: what a rank contributes to the ONE all-reduce of
// mean_dice (neurite/tf/metrics.py:499-510: K.mean(dice * weights) over the [B, L] entries; weights [1, L] or [B, L]).
// One block, fixed order (thread i owns entries i, i + 256, ...; LDS tree) => bit-reproducible.
namespace {
__global__ __launch_bounds__(256) void dice_mean_pair(const float *__restrict__ dice, const float *__restrict__ weights,
                                                      int n, int wmod, float *__restrict__ out2, int with_mean) {
    __shared__ double sl[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = dice[i];
        if (weights) v = nrt_mul(v, weights[i % wmod]);
        acc += (double)v;
    }
    sl[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sl[threadIdx.x] += sl[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = (float)sl[0]; out2[1] = (float)n;
        if (with_mean) out2[2] = (float)sl[0] / (float)n;       // the float32 division a single rank would otherwise launch for
    }
}
}  // namespace

extern "C" int nrt_dice_mean_pair_f32(const float *dice, const float *weights, int nlabels, int batch, int weights_per_batch,
                                      float *out2, void *stream) {
    if (!dice || !out2 || nlabels < 1 || batch < 1) return NRT_ERR_INVALID_ARG;
    const long long n = (long long)nlabels * batch;
    if (n >= (1ll << 31)) return NRT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dice_mean_pair, dim3(1), dim3(256), 0, nrt_stream(stream), dice, weights, (int)n,
                       weights_per_batch ? (int)n : nlabels, out2, 0);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_dice_mean_f32(const float *dice, const float *weights, int nlabels, int batch, int weights_per_batch,
                                 float *out3, void *stream) {
    if (!dice || !out3 || nlabels < 1 || batch < 1) return NRT_ERR_INVALID_ARG;
    const long long n = (long long)nlabels * batch;
    if (n >= (1ll << 31)) return NRT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dice_mean_pair, dim3(1), dim3(256), 0, nrt_stream(stream), dice, weights, (int)n,
                       weights_per_batch ? (int)n : nlabels, out3, 1);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
