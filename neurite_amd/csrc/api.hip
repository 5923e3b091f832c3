// Library-level entry points of the C ABI (include/neurite_amd.h).
#include "nrt_common.h"

extern "C" const char *nrt_status_string(int status) {
    switch (status) {
        case NRT_OK: return "ok";
        case NRT_ERR_INVALID_ARG: return "invalid argument";
        case NRT_ERR_UNSUPPORTED: return "combination not supported by the HIP path";
        case NRT_ERR_LAUNCH: return "HIP kernel launch failed";
        case NRT_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}

extern "C" int nrt_abi_version(void) { return 1; }

extern "C" const char *nrt_target_arch(void) { return "gfx950"; }

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
}  // namespace

extern "C" int nrt_membench_copy_f32(const float *src, float *dst, long long n, int nontemporal, int blocks, void *stream) {
    if (!src || !dst || n < 0 || (n & 3)) return NRT_ERR_INVALID_ARG;
    if (blocks <= 0) blocks = 2048;
    if (nontemporal) hipLaunchKernelGGL((membench_copy<true>), dim3(blocks), dim3(256), 0, nrt_stream(stream), (const nrt_f4 *)src, (nrt_f4 *)dst, n / 4);
    else hipLaunchKernelGGL((membench_copy<false>), dim3(blocks), dim3(256), 0, nrt_stream(stream), (const nrt_f4 *)src, (nrt_f4 *)dst, n / 4);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
