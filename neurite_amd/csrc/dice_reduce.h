// Shared pieces of the Dice reductions (dice.hip, fused.hip): workspace layout, wave helpers and the
// deterministic two-level second stage.
#pragma once

#include "nrt_common.h"

namespace {

constexpr int DICE_BLOCK = 256;
constexpr int DICE_MAX_BLOCKS = 2048;      // 8 blocks per CU; the rest is grid-strided

__host__ __device__ inline unsigned dice_num_blocks(long long nvox, int vox_per_pass) {
    long long nb = (nvox + vox_per_pass - 1) / vox_per_pass;
    if (nb > DICE_MAX_BLOCKS) nb = DICE_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

// workspace layout (per call):  double-free, float partials then int partials
//   fpart [B][nblk][3][L] float      soft sums
//   mpart [B][nblk][4]    float      min t, max t, min p, max p
//   ipart [B][nblk][3][L] uint32     hard counts
struct DiceWs {
    float *fpart;
    float *mpart;
    unsigned *ipart;
    double *gsum;        // [B][ngrp][3L] level-1 sums (float64; reinterpreted as int64 for the hard path)
    float *gmm;          // [B][ngrp][4]
};

inline size_t dice_ws_bytes(int L, int batch) {
    size_t per = (size_t)batch * DICE_MAX_BLOCKS;
    size_t grp = (size_t)batch * ((DICE_MAX_BLOCKS + 63) / 64);
    return per * 3 * L * sizeof(float) + per * 4 * sizeof(float) + per * 3 * L * sizeof(unsigned) +
           grp * 3 * L * sizeof(double) + grp * 4 * sizeof(float) + 256;
}

inline DiceWs dice_ws_carve(void *ws, int L, int batch) {
    DiceWs w;
    size_t per = (size_t)batch * DICE_MAX_BLOCKS;
    char *p = (char *)ws;
    w.fpart = (float *)p; p += per * 3 * L * sizeof(float);
    w.mpart = (float *)p; p += per * 4 * sizeof(float);
    w.ipart = (unsigned *)p; p += per * 3 * L * sizeof(unsigned);
    size_t grp = (size_t)batch * ((DICE_MAX_BLOCKS + 63) / 64);
    p = (char *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    w.gsum = (double *)p; p += grp * 3 * L * sizeof(double);
    w.gmm = (float *)p;
    return w;
}

template <typename T>
__device__ __forceinline__ T wave_xor_add(T v, int from) {
    // sum over lanes that agree in (lane % from): xor offsets from, 2*from, ..., 32
    for (int off = from; off < NRT_WAVE; off <<= 1) v += __shfl_xor(v, off, NRT_WAVE);
    return v;
}

// Second stage, two levels so that no thread walks a long dependent chain (a single-level version took
// 63 us for 2048 partials, profiles/archive/r01_session1): level 1 reduces groups of RED_ROWS block partials in
// float64 (one block per group), level 2 adds the <= 32 group sums in a fixed order and does the
// division.  Fixed partition + fixed order => bit-reproducible.
constexpr int RED_ROWS = 64;

// extrema quadruples (min t, max t, min p, max p): thread t holds a partial of component t & 3; 256 threads; the combined value
// comes back in threads 0..3.  (Threads 0..3 walking the rows themselves was a chain of dependent-issue loads: ~0.3 us per row.)
__device__ __forceinline__ float combine_minmax4(float m, float *smm /* [16] */) {
    const int i = threadIdx.x & 3;
#pragma unroll
    for (int off = 4; off < NRT_WAVE; off <<= 1) {
        const float o = __shfl_xor(m, off, NRT_WAVE);
        m = (i & 1) ? fmaxf(m, o) : fminf(m, o);
    }
    if ((threadIdx.x & (NRT_WAVE - 1)) < 4) smm[(threadIdx.x / NRT_WAVE) * 4 + i] = m;
    __syncthreads();
    if (threadIdx.x < 4) {
        m = smm[i];
        for (int w = 1; w < 256 / NRT_WAVE; ++w) m = (i & 1) ? fmaxf(m, smm[w * 4 + i]) : fminf(m, smm[w * 4 + i]);
    }
    return m;
}

template <typename TIN, typename TACC>
__global__ __launch_bounds__(256) void reduce_rows(const TIN *__restrict__ in, int rows, int ncol, TACC *__restrict__ out,
                                                    const float *__restrict__ mm_in, float *__restrict__ mm_out, int mm_rows) {
    __shared__ TACC sl[256];
    const int b = blockIdx.x, grp = blockIdx.y;
    const int r0 = grp * RED_ROWS, r1 = min(r0 + RED_ROWS, rows);
    for (int c0 = 0; c0 < ncol; c0 += 256) {
        const int cols = min(ncol - c0, 256);
        const int S = 256 / cols;
        const int i = threadIdx.x % cols, sidx = threadIdx.x / cols;
        TACC acc = 0;
        if (sidx < S) {
            // 8 independent loads in flight per thread, then a fixed-order add (a dependent load-add chain
            // cost ~0.6 us per row here: the partials sit in another XCD's L2)
            constexpr int U = 8;                                 // loads in flight per thread (16: no faster; same order of additions for any U)
            for (int k = r0 + sidx; k < r1; k += U * S) {
                TIN v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u * S;
                    v[u] = kk < r1 ? in[((long long)b * rows + kk) * ncol + c0 + i] : (TIN)0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc += (TACC)v[u];
            }
        }
        sl[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < cols) {
            TACC tot = 0;
            for (int ss = 0; ss < S; ++ss) tot += sl[ss * cols + threadIdx.x];
            out[((long long)b * gridDim.y + grp) * ncol + c0 + threadIdx.x] = tot;
        }
        __syncthreads();
    }
    if (mm_in) {                               // min t, max t, min p, max p of this group's partials (block-uniform branch)
        __shared__ float smm[16];
        const int i = threadIdx.x & 3;
        const int q0 = (int)((long long)r0 * mm_rows / rows), q1 = (int)((long long)r1 * mm_rows / rows);
        float m = (i & 1) ? -INFINITY : INFINITY;
        for (int k = q0 + (int)(threadIdx.x >> 2); k < q1; k += 64) {
            const float v = mm_in[((long long)b * mm_rows + k) * 4 + i];
            m = (i & 1) ? fmaxf(m, v) : fminf(m, v);
        }
        m = combine_minmax4(m, smm);
        if (threadIdx.x < 4) mm_out[((long long)b * gridDim.y + grp) * 4 + i] = m;
    }
}

__global__ __launch_bounds__(256) void dice_soft_finalize(const double *__restrict__ gsum, const float *__restrict__ gmm,
                                                          int ngrp, int L, float eps, float *__restrict__ sums,
                                                          float *__restrict__ dice, float *__restrict__ minmax) {
    extern __shared__ float fs[];   // [3*L]
    const int b = blockIdx.x;
    const int ncol = 3 * L;
    for (int i = threadIdx.x; i < ncol; i += blockDim.x) {
        double tot = 0.0;
        for (int g = 0; g < ngrp; g += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (g + u < ngrp) ? gsum[((long long)b * ngrp + g + u) * ncol + i] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += v[u];
        }
        const float f = (float)tot;
        fs[i] = f;
        sums[(long long)b * ncol + i] = f;
    }
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const float top = nrt_mul(2.0f, fs[l]);                         // metrics.py:476
        const float bottom = nrt_add(fs[L + l], fs[2 * L + l]);         // :477
        float d;
        if (eps > 0.0f) d = nrt_add(top, eps) / nrt_add(bottom, eps);   // :478-480
        else d = (bottom == 0.0f) ? 0.0f : top / bottom;                // :482 divide_no_nan
        dice[(long long)b * L + l] = d;
    }
    if (minmax && b == 0) {
        __shared__ float smm[16];
        const int i = threadIdx.x & 3;
        float m = (i & 1) ? -INFINITY : INFINITY;
        for (int k = (int)(threadIdx.x >> 2); k < (int)gridDim.x * ngrp; k += 64) {
            const float v = gmm[(long long)k * 4 + i];
            m = (i & 1) ? fmaxf(m, v) : fminf(m, v);
        }
        m = combine_minmax4(m, smm);
        if (threadIdx.x < 4) minmax[i] = m;
    }
}

// launch the two-level second stage for the soft sums (fpart/mpart -> sums, dice, minmax)
inline int dice_finalize_soft(const DiceWs &w, unsigned nblk, unsigned gz, int batch, int nlabels, float eps,
                              float *sums, float *dice, float *minmax, hipStream_t st) {
    const int rows = (int)nblk;
    const int ngrp = (rows + RED_ROWS - 1) / RED_ROWS;
    hipLaunchKernelGGL((reduce_rows<float, double>), dim3(batch, ngrp), dim3(256), 0, st, (const float *)w.fpart, rows,
                       3 * nlabels, w.gsum, (const float *)w.mpart, w.gmm, (int)(nblk * gz));
    NRT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_soft_finalize, dim3(batch), dim3(256), (size_t)3 * nlabels * sizeof(float), st,
                       (const double *)w.gsum, (const float *)w.gmm, ngrp, nlabels, eps, sums, dice, minmax);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

}  // namespace
