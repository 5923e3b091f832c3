// Lab: memory-system probes (not part of the product library; built into tools/lab/liblab.so).
#include "nrt_common.h"

// ---- diagnostic: streaming copy, to calibrate what a mixed read/write stream reaches on this chip ----
namespace {
template <bool NT>
__global__ __launch_bounds__(256) void membench_copy(const nrt_f4 *__restrict__ src, nrt_f4 *__restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
#pragma unroll 4
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const nrt_f4 v = NT ? __builtin_nontemporal_load(&src[i]) : src[i];
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}

// L1-resident gather: every block re-reads a private window of `rows` 128-byte rows (rows * 128 B <= L1) with the
// lane pattern of the interpn kernels (8 lanes x 16 B per row, 8 rows per wave instruction).  pattern 0: the 8 rows
// of a wave are consecutive (one contiguous KiB); 1: scattered over the window; 2: 8 loads per step that overlap like
// the 8 corners of consecutive-z voxels (rows r, r+1, r+sy, r+sy+1, ...).  Measures what the TA/L1 path delivers.
__global__ __launch_bounds__(256) void membench_l1(const nrt_f4 *__restrict__ src, float *__restrict__ sink, int rows,
                                                   int iters, int pattern) {
    const int lg = threadIdx.x & 7, g = threadIdx.x >> 3;          // 32 lane-groups per block
    const nrt_f4 *win = src + (long long)blockIdx.x * rows * 8;
    nrt_f4 acc = {0, 0, 0, 0};
    unsigned r = (unsigned)g * (pattern == 1 ? 37u : 1u);
    for (int it = 0; it < iters; ++it) {
        nrt_f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned row;
            if (pattern == 2) row = r + (unsigned)(k & 1) + (unsigned)((k >> 1) & 1) * 20u + (unsigned)(k >> 2) * 97u;
            else row = r + (unsigned)k * (pattern == 1 ? 53u : 32u);
            v[k] = win[(row % (unsigned)rows) * 8u + lg];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
        r += pattern == 1 ? 11u : 3u;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
}  // namespace

extern "C" int nrt_lab_membench_l1_f32(const float *src, float *sink, int rows, int iters, int pattern, int blocks, void *stream) {
    if (!src || !sink || rows < 1 || iters < 1 || blocks < 1) return NRT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(membench_l1, dim3(blocks), dim3(256), 0, nrt_stream(stream), (const nrt_f4 *)src, sink, rows, iters, pattern);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_lab_membench_copy_f32(const float *src, float *dst, long long n, int nontemporal, int blocks, void *stream) {
    if (!src || !dst || n < 0 || (n & 3)) return NRT_ERR_INVALID_ARG;
    if (blocks <= 0) blocks = 2048;
    if (nontemporal) hipLaunchKernelGGL((membench_copy<true>), dim3(blocks), dim3(256), 0, nrt_stream(stream), (const nrt_f4 *)src, (nrt_f4 *)dst, n / 4);
    else hipLaunchKernelGGL((membench_copy<false>), dim3(blocks), dim3(256), 0, nrt_stream(stream), (const nrt_f4 *)src, (nrt_f4 *)dst, n / 4);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}


// ---- tuned streaming copy: float4 per lane, UNROLL independent loads in flight per thread, grid sized by the caller ----
namespace {
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void lab_copy(const nrt_f4 *__restrict__ src, nrt_f4 *__restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        nrt_f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
template <int UNROLL>
__global__ __launch_bounds__(256) void lab_read(const nrt_f4 *__restrict__ src, float *__restrict__ sink, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    nrt_f4 acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        nrt_f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
}  // namespace

// kind 0: copy, 1: read only.  unroll in {1, 2, 4, 8}; nontemporal 0 / 1; blocks x 256 threads
extern "C" int nrt_lab_stream_f32(const float *src, float *dst, long long n, int kind, int unroll, int nontemporal, int blocks, void *stream) {
    if (!src || !dst || n < 0 || (n & 3) || blocks < 1) return NRT_ERR_INVALID_ARG;
    hipStream_t st = nrt_stream(stream);
    const nrt_f4 *s4 = (const nrt_f4 *)src; nrt_f4 *d4 = (nrt_f4 *)dst; const long long n4 = n / 4;
#define LAB_COPY(U) do { if (kind == 1) hipLaunchKernelGGL((lab_read<U>), dim3(blocks), dim3(256), 0, st, s4, dst, n4); \
        else if (nontemporal) hipLaunchKernelGGL((lab_copy<U, true>), dim3(blocks), dim3(256), 0, st, s4, d4, n4); \
        else hipLaunchKernelGGL((lab_copy<U, false>), dim3(blocks), dim3(256), 0, st, s4, d4, n4); } while (0)
    switch (unroll) { case 1: LAB_COPY(1); break; case 2: LAB_COPY(2); break; case 4: LAB_COPY(4); break; default: LAB_COPY(8); break; }
#undef LAB_COPY
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
