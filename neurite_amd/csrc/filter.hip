// Separable filtering and min-max normalisation of the synthesis front-end, gfx950.
//
// utils.separable_conv (neurite/tf/utils/utils.py:665-751) transposes the tensor to [B*C, *S, 1] and calls
// tf.nn.convolution once per axis with a [1..w..1, 1, 1] filter; layers.GaussianBlur (:251-364) is that with Gaussian
// kernels.  Here one pass is one kernel on the channels-last tensor as it lies in memory, viewed as
// [outer, A, inner] around the filtered axis: y[o, a, i] = sum_t k[t] * x[o, a * stride + t * dilation - pad, i]
// (zero outside: TF SAME puts total // 2 of the padding in front).  HBM-bound: each pass reads and writes the
// tensor once; the w taps of neighbouring outputs overlap in L1/L2.
// utils.minmax_norm (:953-968): (x - min) / (max - min) with div_no_nan over any contiguous run of axes, viewed as
// [outer, R, inner]; the extrema are order-independent, so they are reduced with integer atomics on a monotone
// encoding of the floats (deterministic), then applied in a second pass.

#include <cstdlib>
#include <type_traits>

#include "nrt_common.h"

namespace {

struct C1Args {
    const float *x;
    const float *k;
    float *y;
    long long outer;
    int A, Aout, W, stride, dil, pad;
    long long inner;        // elements (scalar kernel) or float4 groups (vector kernel)
};

template <bool VEC>
__global__ __launch_bounds__(256) void conv1d_axis(C1Args a) {
    typedef typename std::conditional<VEC, nrt_f4, float>::type T;
    const T *x = (const T *)a.x;
    T *y = (T *)a.y;
    const long long total = a.outer * a.Aout * a.inner;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e % a.inner;
        const long long r = e / a.inner;
        const int ao = (int)(r % a.Aout);
        const long long o = r / a.Aout;
        const T *xo = x + o * a.A * a.inner + i;
        const int a0 = ao * a.stride - a.pad;
        T acc = T{};
        for (int t = 0; t < a.W; ++t) {
            const int ai = a0 + t * a.dil;
            if (ai >= 0 && ai < a.A) {
                const T v = xo[(long long)ai * a.inner];
                acc = acc + a.k[t] * v;
            }
        }
        y[e] = acc;
    }
}

// innermost-axis pass of single-channel tensors (inner == 1, stride == 1, dilation == 1): a thread produces 4 consecutive
// outputs from W + 3 inputs instead of 4 W (same tap order per output => bit-identical to conv1d_axis<false>)
__global__ __launch_bounds__(256) void conv1d_axis_run4(C1Args a) {
    extern __shared__ float ks[];
    for (int t = threadIdx.x; t < a.W; t += 256) ks[t] = a.k[t];
    __syncthreads();
    const long long nrun = (a.Aout + 3) / 4;
    const long long total = a.outer * nrun;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long o = e / nrun;
        const int a0 = (int)(e - o * nrun) * 4;
        const float *xo = a.x + o * a.A;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int t = 0; t < a.W + 3; ++t) {
            const int ai = a0 - a.pad + t;
            const float v = (ai >= 0 && ai < a.A) ? xo[ai] : 0.0f;
            const bool inside = ai >= 0 && ai < a.A;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kt = t - r;
                if (inside && kt >= 0 && kt < a.W) acc[r] = acc[r] + ks[kt] * v;
            }
        }
        float *yo = a.y + o * a.Aout + a0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (a0 + r < a.Aout) yo[r] = acc[r];
    }
}

// ---- the fast passes (stride 1, dilation 1) -----------------------------------------------------------------------------------
// A separable pass is a copy with arithmetic: 8 bytes per element if every input is read from HBM once.  The plain kernels
// above issue W loads and W scalar tap loads per output.  These two keep R running sums per lane and walk the taps in
// blocks of TB: per block TB new inputs (rows kernel) or three 16-byte LDS reads (innermost axis), the block's TB taps by two
// broadcast LDS reads, and TB * R multiply-adds whose operands are all registers.  An output still receives its taps in
// ascending order, one rounding per multiply and per add, so the sums are those of conv1d_axis.  Inputs outside the tensor are
// read as zero instead of being skipped: tap * 0 added to a running sum that started at +0 changes nothing (the sum is never
// -0), as long as the tap is finite -- a block that sees a non-finite tap takes the plain per-output loop instead.
constexpr int CF_TB = 8, CF_WMAX = 256;

// stage the taps in LDS (zero-padded to a multiple of TB); true when all of them are finite
__device__ __forceinline__ bool stage_taps(const float *__restrict__ k, int W, float *ks) {
    int bad = 0;
    for (int t = threadIdx.x; t < CF_WMAX + CF_TB; t += 256) {
        const float v = t < W ? k[t] : 0.0f;
        ks[t] = v;
        bad |= !(fabsf(v) <= 3.402823466e38f);
    }
    return __syncthreads_or(bad) == 0;
}

// conv1d_axis_rows: any axis but the innermost.  A lane owns one inner position (a float4 of them when VEC) and R consecutive
// outputs along the axis; consecutive lanes are consecutive in memory, so every load / store instruction is coalesced.
template <bool VEC, int R>
__global__ __launch_bounds__(256) void conv1d_axis_rows(C1Args a, int chunks) {
    typedef typename std::conditional<VEC, nrt_f4, float>::type T;
    constexpr int TB = CF_TB;
    static_assert(R - 1 <= TB, "the register window is rotated by one tap block");
    __shared__ __attribute__((aligned(16))) float ks[CF_WMAX + CF_TB];
    const bool finite = stage_taps(a.k, a.W, ks);
    const T *x = (const T *)a.x;
    T *y = (T *)a.y;
    const long long total = a.outer * chunks * a.inner;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e % a.inner;
        const long long r2 = e / a.inner;
        const int a0 = (int)(r2 % chunks) * R;
        const long long o = r2 / chunks;
        const int first = a0 - a.pad;                       // input index of tap 0 of output a0
        const T *xo = x + o * a.A * a.inner + i;
        T acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = T{};
        if (finite) {
            auto in = [&](int m) -> T {                     // input `first + m`, zero outside the axis
                const int ai = first + m;
                return (ai >= 0 && ai < a.A) ? xo[(long long)ai * a.inner] : T{};
            };
            T xin[TB + R - 1];
#pragma unroll
            for (int j = 0; j < R - 1; ++j) xin[j] = in(j);
            for (int kb = 0; kb < a.W; kb += TB) {
#pragma unroll
                for (int j = 0; j < TB; ++j) xin[R - 1 + j] = (kb + j < a.W) ? in(kb + R - 1 + j) : T{};
                const nrt_f4 k0 = *(const nrt_f4 *)(ks + kb), k1 = *(const nrt_f4 *)(ks + kb + 4);
                const float kk[TB] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    if (kb + j < a.W) {                     // uniform: a padded tap is skipped, not multiplied (0 * Inf)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = acc[r] + kk[j] * xin[j + r];
                    }
                }
#pragma unroll
                for (int j = 0; j < R - 1; ++j) xin[j] = xin[j + TB];
            }
        } else {
            for (int r = 0; r < R; ++r) {                   // the plain loop: out-of-range taps skipped
                if (a0 + r >= a.Aout) break;
                T s = T{};
                for (int t = 0; t < a.W; ++t) {
                    const int ai = first + r + t;
                    if (ai >= 0 && ai < a.A) s = s + a.k[t] * xo[(long long)ai * a.inner];
                }
#pragma unroll
                for (int q = 0; q < R; ++q)
                    if (q == r) acc[q] = s;
            }
        }
        T *yp = y + (o * a.Aout + a0) * a.inner + i;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (a0 + r < a.Aout) __builtin_nontemporal_store(acc[r], yp + (long long)r * a.inner);
    }
}

// conv1d_inner_lds: the innermost axis of single-channel tensors (inner == 1).  A block of 256 lanes owns 256 / LPR rows and a
// segment of 8 LPR outputs of them (LPR = 4 .. 32 lanes per row, chosen on the host so that the segments tile the axis with
// the least waste); it stages the segment's inputs (8 LPR + W - 1 per row, zero outside the row) in LDS with 16-byte loads of
// consecutive lanes, then every lane forms 8 consecutive outputs, reading its 16-float window of a tap block with four 16-byte
// LDS reads.
struct CiGeom {
    int lpr_log2;        // lanes per row
    int segs;            // segments per row
    int spanq;           // 16-byte chunks of a staged row: (8 LPR + roundup8(W) + 8) / 4
    unsigned m_spanq;    // 2^32 / spanq + 1
};

__global__ __launch_bounds__(256) void conv1d_inner_lds(C1Args a, CiGeom g) {
    constexpr int TB = CF_TB, R = 8;
    extern __shared__ __attribute__((aligned(16))) float ci_lds[];      // [rows][span] inputs, then the taps
    const int rows = 256 >> g.lpr_log2, span = g.spanq * 4;
    float *ks = ci_lds + rows * span;
    const bool finite = stage_taps(a.k, a.W, ks);
    const long long rb = blockIdx.x / g.segs;
    const int seg0 = (int)(blockIdx.x - rb * g.segs) * (R << g.lpr_log2);
    const long long row0 = rb * rows;
    const int first = seg0 - a.pad;
    for (int idx = threadIdx.x; idx < rows * g.spanq; idx += 256) {
        const int rr = (int)__umulhi((unsigned)idx, g.m_spanq), c = (idx - rr * g.spanq) * 4;
        const int ai = first + c;
        const long long row = row0 + rr;
        nrt_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row < a.outer) {
            const float *xr = a.x + row * a.A;
            if (ai >= 0 && ai + 3 < a.A) {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                const f4u t = *(const f4u *)(xr + ai);              // 4-byte aligned 16-byte load
                v = (nrt_f4){t[0], t[1], t[2], t[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ai + j >= 0 && ai + j < a.A) v[j] = xr[ai + j];
            }
        }
        *(nrt_f4 *)(ci_lds + rr * span + c) = v;
    }
    __syncthreads();
    const int rr = threadIdx.x >> g.lpr_log2, c0 = (threadIdx.x & ((1 << g.lpr_log2) - 1)) * R;
    const long long row = row0 + rr;
    if (row >= a.outer || seg0 + c0 >= a.Aout) return;
    const float *sp = ci_lds + rr * span + c0;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    if (finite) {
        for (int kb = 0; kb < a.W; kb += TB) {
            float xin[TB + R];
#pragma unroll
            for (int q = 0; q < (TB + R) / 4; ++q) {
                const nrt_f4 t = *(const nrt_f4 *)(sp + kb + 4 * q);
                xin[4 * q] = t[0]; xin[4 * q + 1] = t[1]; xin[4 * q + 2] = t[2]; xin[4 * q + 3] = t[3];
            }
            const nrt_f4 k0 = *(const nrt_f4 *)(ks + kb), k1 = *(const nrt_f4 *)(ks + kb + 4);
            const float kk[TB] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                if (kb + j < a.W) {
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = acc[r] + kk[j] * xin[j + r];
                }
            }
        }
    } else {
        for (int t = 0; t < a.W + R - 1; ++t) {
            const int ai = first + c0 + t;
            if (ai < 0 || ai >= a.A) continue;
            const float v = sp[t];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kt = t - r;
                if (kt >= 0 && kt < a.W) acc[r] = acc[r] + a.k[kt] * v;
            }
        }
    }
    float *yo = a.y + row * a.Aout + seg0 + c0;
    if (seg0 + c0 + R - 1 < a.Aout && ((a.Aout & 3) == 0)) {
        __builtin_nontemporal_store((nrt_f4){acc[0], acc[1], acc[2], acc[3]}, (nrt_f4 *)yo);
        __builtin_nontemporal_store((nrt_f4){acc[4], acc[5], acc[6], acc[7]}, (nrt_f4 *)yo + 1);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (seg0 + c0 + r < a.Aout) yo[r] = acc[r];
    }
}

// geometry of conv1d_inner_lds for an axis of `out_len` outputs and `width` taps: the lanes per row that stage the fewest inputs
CiGeom ci_geometry(int out_len, int width) {
    const int wp = ((width + CF_TB - 1) / CF_TB) * CF_TB;
    CiGeom best = {};
    long long best_cost = -1;
    for (int l2 = 2; l2 <= 5; ++l2) {
        const int seg = 8 << l2, segs = (out_len + seg - 1) / seg;
        const long long cost = (long long)segs * (seg + wp + 8);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best.lpr_log2 = l2; best.segs = segs; best.spanq = (seg + wp + 8) / 4;
        }
    }
    best.m_spanq = (unsigned)(0x100000000ull / (unsigned)best.spanq) + 1u;
    return best;
}

__device__ __forceinline__ int f2key(float f) {
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void minmax_init(int *ws, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        ws[2 * i] = 0x7fffffff;            // running min (key of +max)
        ws[2 * i + 1] = (int)0x80000000;   // running max
    }
}

// x viewed as [outer, R, inner]; ws [outer, inner, 2] keys.  blockIdx.y = outer index.
__global__ __launch_bounds__(256) void minmax_reduce(const float *__restrict__ x, int *__restrict__ ws, long long R, int inner) {
    extern __shared__ int sm[];            // [inner][2], inner <= 1024
    const int o = blockIdx.y;
    for (int i = threadIdx.x; i < inner; i += 256) { sm[2 * i] = 0x7fffffff; sm[2 * i + 1] = (int)0x80000000; }
    __syncthreads();
    const float *xo = x + (long long)o * R * inner;
    const long long n = R * inner;
    if (inner == 1) {
        int mn = 0x7fffffff, mx = (int)0x80000000;
        const long long n4 = (((uintptr_t)xo & 15) == 0) ? n / 4 : 0;           // 16-byte loads over the aligned body
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
            const nrt_f4 v = ((const nrt_f4 *)xo)[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int k = f2key(v[j]); mn = min(mn, k); mx = max(mx, k); }
        }
        for (long long e = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            const int k = f2key(xo[e]);
            mn = min(mn, k); mx = max(mx, k);
        }
        for (int off = 1; off < 64; off <<= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&sm[0], mn); atomicMax(&sm[1], mx); }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            const int i = (int)(e % inner);
            const int k = f2key(xo[e]);
            atomicMin(&sm[2 * i], k); atomicMax(&sm[2 * i + 1], k);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < inner; i += 256) {
        atomicMin(&ws[((long long)o * inner + i) * 2], sm[2 * i]);
        atomicMax(&ws[((long long)o * inner + i) * 2 + 1], sm[2 * i + 1]);
    }
}

__global__ __launch_bounds__(256) void minmax_apply(const float *__restrict__ x, const int *__restrict__ ws, float *__restrict__ y,
                                                    long long R, int inner) {
    const int o = blockIdx.y;
    const long long n = R * inner;
    const float *xo = x + (long long)o * n;
    float *yo = y + (long long)o * n;
    if (inner == 1) {
        const float mn = key2f(ws[(long long)o * 2]), mx = key2f(ws[(long long)o * 2 + 1]);
        const float den = mx - mn;
        const long long n4 = ((((uintptr_t)xo | (uintptr_t)yo) & 15) == 0) ? n / 4 : 0;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
            const nrt_f4 v = ((const nrt_f4 *)xo)[e];
            nrt_f4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = den != 0.0f ? (v[j] - mn) / den : 0.0f;
            __builtin_nontemporal_store(r, (nrt_f4 *)yo + e);
        }
        for (long long e = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
            yo[e] = den != 0.0f ? (xo[e] - mn) / den : 0.0f;
        return;
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int i = inner == 1 ? 0 : (int)(e % inner);
        const float mn = key2f(ws[((long long)o * inner + i) * 2]), mx = key2f(ws[((long long)o * inner + i) * 2 + 1]);
        const float den = mx - mn;
        yo[e] = den != 0.0f ? (xo[e] - mn) / den : 0.0f;      // tf div_no_nan
    }
}

// inner == 1 (one pair of extrema per outer entry): two launches, no atomics.  Every block of the first writes its pair of keys;
// every block of the second reduces the (at most MM_NB) pairs of its entry again, then normalises its share.  (The first
// version sent 2 atomics per block to one address per entry: 2000 same-address atomics, 95 us for 4 x 160^3.)
constexpr int MM_NB = 256;

__device__ __forceinline__ void block_minmax(int &mn, int &mx, int *sm) {      // all 256 lanes get the block's extrema
    for (int off = 1; off < 64; off <<= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
    if ((threadIdx.x & 63) == 0) { sm[(threadIdx.x >> 6) * 2] = mn; sm[(threadIdx.x >> 6) * 2 + 1] = mx; }
    __syncthreads();
    mn = min(min(sm[0], sm[2]), min(sm[4], sm[6]));
    mx = max(max(sm[1], sm[3]), max(sm[5], sm[7]));
}

__global__ __launch_bounds__(256) void minmax_reduce1(const float *__restrict__ x, int *__restrict__ part, long long n) {
    __shared__ int sm[8];
    const float *xo = x + (long long)blockIdx.y * n;
    int mn = 0x7fffffff, mx = (int)0x80000000;
    const long long n4 = (((uintptr_t)xo & 15) == 0) ? n / 4 : 0;           // 16-byte loads over the aligned body
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
        const nrt_f4 v = ((const nrt_f4 *)xo)[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = f2key(v[j]); mn = min(mn, k); mx = max(mx, k); }
    }
    for (long long e = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int k = f2key(xo[e]);
        mn = min(mn, k); mx = max(mx, k);
    }
    block_minmax(mn, mx, sm);
    if (threadIdx.x == 0) {
        part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2] = mn;
        part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = mx;
    }
}

__global__ __launch_bounds__(256) void minmax_apply1(const float *__restrict__ x, const int *__restrict__ part, int nb,
                                                     float *__restrict__ y, long long n) {
    __shared__ int sm[8];
    const int o = blockIdx.y;
    int kmn = 0x7fffffff, kmx = (int)0x80000000;
    if ((int)threadIdx.x < nb) { kmn = part[((long long)o * nb + threadIdx.x) * 2]; kmx = part[((long long)o * nb + threadIdx.x) * 2 + 1]; }
    block_minmax(kmn, kmx, sm);
    const float mn = key2f(kmn), mx = key2f(kmx);
    const float den = mx - mn;
    const float *xo = x + (long long)o * n;
    float *yo = y + (long long)o * n;
    const long long n4 = ((((uintptr_t)xo | (uintptr_t)yo) & 15) == 0) ? n / 4 : 0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
        const nrt_f4 v = ((const nrt_f4 *)xo)[e];
        nrt_f4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = den != 0.0f ? (v[j] - mn) / den : 0.0f;      // tf div_no_nan
        __builtin_nontemporal_store(r, (nrt_f4 *)yo + e);
    }
    for (long long e = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
        yo[e] = den != 0.0f ? (xo[e] - mn) / den : 0.0f;
}

__global__ __launch_bounds__(256) void minmax_decode(const int *part, int nb, float *out) {
    __shared__ int sm[8];
    int mn = 0x7fffffff, mx = (int)0x80000000;
    if ((int)threadIdx.x < nb) { mn = part[threadIdx.x * 2]; mx = part[threadIdx.x * 2 + 1]; }
    block_minmax(mn, mx, sm);
    if (threadIdx.x == 0) { out[0] = key2f(mn); out[1] = key2f(mx); }
}

// tf.linspace(min(x), max(x), nb) from the block pairs of minmax_reduce1 (utils.py:1152-1154): start + delta * i with
// delta = (stop - start) / (nb - 1), one rounding per op, the ends exact
__global__ __launch_bounds__(256) void minmax_centers(const int *part, int nbp, int nb, float *out) {
    __shared__ int sm[8];
    int kmn = 0x7fffffff, kmx = (int)0x80000000;
    if ((int)threadIdx.x < nbp) { kmn = part[threadIdx.x * 2]; kmx = part[threadIdx.x * 2 + 1]; }
    block_minmax(kmn, kmx, sm);
    const float mn = key2f(kmn), mx = key2f(kmx);
    const float delta = nb > 1 ? (mx - mn) / (float)(nb - 1) : 0.0f;
    for (int i = threadIdx.x; i < nb; i += 256) {
        float c = nrt_add(mn, nrt_mul(delta, (float)i));
        if (i == 0) c = mn;
        if (i == nb - 1 && nb > 1) c = mx;
        out[i] = c;
    }
}

unsigned fblocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256ll * 16) b = 256ll * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int nrt_conv1d_axis_f32(const float *x, const float *kernel, float *y, long long outer, int axis_len, long long inner,
                                   int out_len, int width, int stride, int dilation, int pad_before, void *stream) {
    if (!x || !kernel || !y || outer < 0 || axis_len < 1 || inner < 1 || out_len < 0 || width < 1 || stride < 1 || dilation < 1)
        return NRT_ERR_INVALID_ARG;
    if (outer == 0 || out_len == 0) return NRT_OK;
    C1Args a;
    a.x = x; a.k = kernel; a.y = y; a.outer = outer; a.A = axis_len; a.Aout = out_len; a.W = width; a.stride = stride;
    a.dil = dilation; a.pad = pad_before; a.inner = inner;
    hipStream_t st = nrt_stream(stream);
    const bool vec = inner % 4 == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    const char *ge = getenv("NRT_CONV1D_GENERIC");                // tests: the plain kernels only
    const bool fast = !(ge && ge[0] == '1') && stride == 1 && dilation == 1 && width <= 256 && out_len >= 8;
    constexpr int R = 8;
    if (fast && inner == 1 && outer >= 8 && (((uintptr_t)y & 15) == 0)) {
        const CiGeom g = ci_geometry(out_len, width);
        const int rows = 256 >> g.lpr_log2;
        const long long nblk = ((outer + rows - 1) / rows) * g.segs;
        if (nblk < (1ll << 31)) {
            hipLaunchKernelGGL(conv1d_inner_lds, dim3((unsigned)nblk), dim3(256),
                               (size_t)(rows * g.spanq * 4 + CF_WMAX + CF_TB) * sizeof(float), st, a, g);
            NRT_CHECK_LAUNCH();
            return NRT_OK;
        }
    }
    if (fast && ((vec && inner >= 64) || (!vec && inner >= 32))) {
        const int chunks = (out_len + R - 1) / R;
        if (vec) {
            a.inner = inner / 4;
            hipLaunchKernelGGL((conv1d_axis_rows<true, R>), dim3(fblocks(outer * chunks * a.inner)), dim3(256), 0, st, a, chunks);
        } else {
            hipLaunchKernelGGL((conv1d_axis_rows<false, R>), dim3(fblocks(outer * chunks * inner)), dim3(256), 0, st, a, chunks);
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    if (inner == 1 && stride == 1 && dilation == 1 && out_len >= 8 && width <= 1024) {
        hipLaunchKernelGGL(conv1d_axis_run4, dim3(fblocks(outer * ((out_len + 3) / 4))), dim3(256), (size_t)width * sizeof(float), st, a);
    } else if (vec) {
        a.inner = inner / 4;
        hipLaunchKernelGGL((conv1d_axis<true>), dim3(fblocks(outer * out_len * a.inner)), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((conv1d_axis<false>), dim3(fblocks(outer * out_len * inner)), dim3(256), 0, st, a);
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" size_t nrt_minmax_workspace_bytes(long long outer, int inner) {
    return (size_t)(outer > 0 ? outer : 0) * (size_t)(inner == 1 ? MM_NB : (inner > 0 ? inner : 0)) * 2 * sizeof(int);
}

extern "C" int nrt_minmax_norm_f32(const float *x, float *y, long long outer, long long reduce_len, int inner, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (!x || !y || outer < 0 || reduce_len < 0 || inner < 1) return NRT_ERR_INVALID_ARG;
    if (inner > 1024 || outer > 65535) return NRT_ERR_UNSUPPORTED;
    if (outer == 0 || reduce_len == 0) return NRT_OK;
    if (!workspace || workspace_bytes < nrt_minmax_workspace_bytes(outer, inner)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    int *ws = (int *)workspace;
    if (inner == 1) {
        long long nb = (reduce_len + 256 * 16 - 1) / (256 * 16);
        if (nb > MM_NB) nb = MM_NB;
        hipLaunchKernelGGL(minmax_reduce1, dim3((unsigned)nb, (unsigned)outer), dim3(256), 0, st, x, ws, reduce_len);
        long long ab = ((reduce_len + 3) / 4 + 255) / 256;
        const long long cap = outer >= 16 ? 64 : 4096 / outer;
        if (ab > cap) ab = cap;
        hipLaunchKernelGGL(minmax_apply1, dim3((unsigned)ab, (unsigned)outer), dim3(256), 0, st, x, ws, (int)nb, y, reduce_len);
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    hipLaunchKernelGGL(minmax_init, dim3(fblocks(outer * inner)), dim3(256), 0, st, ws, outer * inner);
    long long bx = (reduce_len * inner + 256 * 16 - 1) / (256 * 16);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(minmax_reduce, dim3((unsigned)bx, (unsigned)outer), dim3(256), (size_t)inner * 2 * sizeof(int), st, x, ws,
                       reduce_len, inner);
    hipLaunchKernelGGL(minmax_apply, dim3(fblocks(inner == 1 ? (reduce_len + 3) / 4 : reduce_len * inner), (unsigned)outer), dim3(256), 0, st,
                       x, ws, y, reduce_len, inner);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_minmax_f32(const float *x, long long n, float *out2, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !out2 || n < 1) return NRT_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < nrt_minmax_workspace_bytes(1, 1)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    int *ws = (int *)workspace;
    long long nb = (n + 256 * 16 - 1) / (256 * 16);
    if (nb > MM_NB) nb = MM_NB;
    hipLaunchKernelGGL(minmax_reduce1, dim3((unsigned)nb, 1), dim3(256), 0, st, x, ws, n);
    hipLaunchKernelGGL(minmax_decode, dim3(1), dim3(256), 0, st, ws, (int)nb, out2);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_bin_centers_f32(const float *x, long long n, int nb_bins, float *centers, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    if (!x || !centers || n < 1 || nb_bins < 1) return NRT_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < nrt_minmax_workspace_bytes(1, 1)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    int *ws = (int *)workspace;
    long long nb = (n + 256 * 16 - 1) / (256 * 16);
    if (nb > MM_NB) nb = MM_NB;
    hipLaunchKernelGGL(minmax_reduce1, dim3((unsigned)nb, 1), dim3(256), 0, st, x, ws, n);
    hipLaunchKernelGGL(minmax_centers, dim3(1), dim3(256), 0, st, ws, (int)nb, nb_bins, centers);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
