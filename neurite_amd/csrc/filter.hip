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
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
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
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int i = inner == 1 ? 0 : (int)(e % inner);
        const float mn = key2f(ws[((long long)o * inner + i) * 2]), mx = key2f(ws[((long long)o * inner + i) * 2 + 1]);
        const float den = mx - mn;
        yo[e] = den != 0.0f ? (xo[e] - mn) / den : 0.0f;      // tf div_no_nan
    }
}

__global__ void minmax_decode(const int *ws, float *out) {
    if (threadIdx.x == 0) { out[0] = key2f(ws[0]); out[1] = key2f(ws[1]); }
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
    return (size_t)(outer > 0 ? outer : 0) * (size_t)(inner > 0 ? inner : 0) * 2 * sizeof(int);
}

extern "C" int nrt_minmax_norm_f32(const float *x, float *y, long long outer, long long reduce_len, int inner, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (!x || !y || outer < 0 || reduce_len < 0 || inner < 1) return NRT_ERR_INVALID_ARG;
    if (inner > 1024 || outer > 65535) return NRT_ERR_UNSUPPORTED;
    if (outer == 0 || reduce_len == 0) return NRT_OK;
    if (!workspace || workspace_bytes < nrt_minmax_workspace_bytes(outer, inner)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    int *ws = (int *)workspace;
    hipLaunchKernelGGL(minmax_init, dim3(fblocks(outer * inner)), dim3(256), 0, st, ws, outer * inner);
    long long bx = (reduce_len * inner + 256 * 16 - 1) / (256 * 16);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(minmax_reduce, dim3((unsigned)bx, (unsigned)outer), dim3(256), (size_t)inner * 2 * sizeof(int), st, x, ws,
                       reduce_len, inner);
    hipLaunchKernelGGL(minmax_apply, dim3(fblocks(reduce_len * inner), (unsigned)outer), dim3(256), 0, st, x, ws, y, reduce_len, inner);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_minmax_f32(const float *x, long long n, float *out2, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !out2 || n < 1) return NRT_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < 2 * sizeof(int)) return NRT_ERR_WORKSPACE;
    hipStream_t st = nrt_stream(stream);
    int *ws = (int *)workspace;
    hipLaunchKernelGGL(minmax_init, dim3(1), dim3(256), 0, st, ws, 1ll);
    long long bx = (n + 256 * 16 - 1) / (256 * 16);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(minmax_reduce, dim3((unsigned)bx, 1), dim3(256), 2 * sizeof(int), st, x, ws, n, 1);
    hipLaunchKernelGGL(minmax_decode, dim3(1), dim3(64), 0, st, ws, out2);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
