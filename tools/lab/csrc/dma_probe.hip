// Does global_load_lds_dwordx4 reach LDS offsets beyond 64 KB on gfx950 (M0 width)?  And does a VMEM instruction issued with
// EXEC = 0 count in vmcnt?  Prints what comes back.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const float *src, float *out, int n_off, const unsigned *offs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)sm);
    for (int i = 0; i < n_off; ++i) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + offs[i]);
        // poison
        ((f4 *)(sm + offs[i]))[threadIdx.x] = (f4){-1, -1, -1, -1};
        __syncthreads();
        const unsigned voff = (unsigned)(i * 1024 + threadIdx.x * 16);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(src), "s"(dst) : "memory");
        __syncthreads();
        f4 v = ((f4 *)(sm + offs[i]))[threadIdx.x];
        out[i * 64 + threadIdx.x] = v[0];
    }
}
int main() {
    const int N = 6;
    unsigned h_offs[N] = {0, 32768, 65536 - 1024, 65536, 70656 - 1024, 160 * 1024 - 1024};
    float *src, *out; unsigned *offs;
    (void)hipMalloc(&src, N * 1024); (void)hipMalloc(&out, N * 64 * 4); (void)hipMalloc(&offs, sizeof(h_offs));
    float h[N * 256]; for (int i = 0; i < N * 256; ++i) h[i] = (float)i;
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice); (void)hipMemcpy(offs, h_offs, sizeof(h_offs), hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<<<1, 64, 160 * 1024>>>(src, out, N, offs);
    hipError_t e = hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(e));
    float r[N * 64]; (void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) {
        int ok = 1; for (int t = 0; t < 64; ++t) ok &= (r[i * 64 + t] == (float)(i * 256 + t * 4));
        printf("LDS offset %6u: %s (lane0 %.0f lane63 %.0f, expect %d %d)\n", h_offs[i], ok ? "OK" : "WRONG", r[i * 64], r[i * 64 + 63], i * 256, i * 256 + 252);
    }
    return 0;
}
