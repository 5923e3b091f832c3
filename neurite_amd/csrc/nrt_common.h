// Shared device/host helpers for the gfx950 kernels (wave64 everywhere; no CUDA compatibility).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neurite_amd.h"

#define NRT_WAVE 64
#define NRT_NXCD 8          // MI355X: 8 XCDs, block b is observed to run on XCD b % 8
#define NRT_MAXD 3

typedef float nrt_f4 __attribute__((ext_vector_type(4)));
typedef float nrt_f2 __attribute__((ext_vector_type(2)));
typedef int nrt_i4 __attribute__((ext_vector_type(4)));

#define NRT_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return NRT_ERR_LAUNCH;        \
    } while (0)

// Separately rounded float ops.  The library is built with -ffp-contract=off as well; these keep
// the reference's one-rounding-per-op sequence explicit where bit-parity depends on it.
__device__ __forceinline__ float nrt_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float nrt_add(float a, float b) { return __fadd_rn(a, b); }

// a * b + c with 24-bit operands (full-rate v_mad_u32_u24; v_mul_lo_u32 / v_mad_u64_u32 issue at a quarter of the rate).
// As inline asm because LLVM demotes a __umul24 whose result only feeds another 24-bit multiply back to a 32-bit one.
__device__ __forceinline__ unsigned nrt_mad24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// 3 * q without a multiplier (the compiler folds shift-add forms back into v_mul_lo_u32)
__device__ __forceinline__ unsigned nrt_times3(unsigned q) {
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(r) : "v"(q));
    return r;
}
__device__ __forceinline__ float nrt_sub(float a, float b) { return __fsub_rn(a, b); }

// tf.clip_by_value(v, lo, hi) = min(max(v, lo), hi).  fmaxf/fminf also squash NaN to a finite
// bound, so an index derived from the result can never leave the volume.
__device__ __forceinline__ float nrt_clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ int nrt_clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// XCD-aware logical block id: consecutive logical blocks stay on one XCD (one L2), XCD k owns the
// k-th contiguous eighth of the logical range.  Launch with gridDim.x = NRT_NXCD * ceil(n / NRT_NXCD);
// ids >= n must exit.  Placement only affects speed, never results.
__device__ __forceinline__ unsigned nrt_xcd_block(unsigned bid, unsigned grid) {
    unsigned per = grid / NRT_NXCD;
    return (bid % NRT_NXCD) * per + bid / NRT_NXCD;
}

// Contiguous share [beg, end) of n work items for this block (block.x of grid.x), in multiples of `unit` items.  A grid-strided
// loop makes every block jump by the whole grid each iteration (8 MB apart on the bench tensors): measured on soft Dice 5.2 TB/s
// against 6.4 TB/s when a block streams through ONE contiguous range (DRAM / TLB page locality), so the streaming kernels use this.
__device__ __forceinline__ void nrt_block_range(long long n, long long unit, long long &beg, long long &end) {
    const long long chunk = (long long)gridDim.x * unit;
    const long long per = ((n + chunk - 1) / chunk) * unit;
    beg = (long long)blockIdx.x * per;
    if (beg > n) beg = n;
    end = beg + per < n ? beg + per : n;
}

static inline unsigned nrt_xcd_grid(unsigned nblocks) {
    return NRT_NXCD * ((nblocks + NRT_NXCD - 1) / NRT_NXCD);
}

static inline hipStream_t nrt_stream(void *s) { return (hipStream_t)s; }

// Zero fill on a stream by a KERNEL.  hipMemsetAsync is not used on paths that may run inside a captured hipGraph: the memset node of the
// persistent gather's work counters did not take effect between replays (ROCm 7.2; tools/graph_fused_probe.py, fused_wc.h).
// `bytes` is a multiple of 4 and `p` 4-byte aligned (every caller zeroes float / int / int64 tensors).
static __global__ void nrt_zero_words(unsigned *__restrict__ p, size_t nwords) {
    const size_t n4 = (((uintptr_t)p & 15) == 0) ? nwords >> 2 : 0;            // 16-byte stores where the pointer allows
    typedef unsigned nrt_zu4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        ((nrt_zu4 *)p)[i] = (nrt_zu4){0u, 0u, 0u, 0u};
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t nrt_zero_async(void *p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    const size_t nwords = bytes >> 2;
    size_t blocks = (nwords / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nrt_zero_words, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned *)p, nwords);
    return hipGetLastError();
}

// a slot of NRT_RING_WORDS zeroed device words for the atomic counters of the launches of ONE stream (or of one captured launch); a
// kernel's last block leaves its words zeroed (api.hip).  The users keep disjoint words inside a slot.
constexpr unsigned NRT_RING_SLOTS = 8192, NRT_RING_STREAM_SLOTS = 1024, NRT_RING_WORDS = 512;     // 16 MB of device memory
constexpr unsigned NRT_RING_GATHER_OFF = 0, NRT_RING_CCE_OFF = 256;
unsigned *nrt_ring_slot(hipStream_t st);

// compute units of the current device (persistent kernels size their grid with it)
static inline int nrt_num_cus() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
