// VALU issue-rate probe for gfx950: cycles per wave64 instruction for v_fma_f32 / v_pk_fma_f32 / v_pk_mul+add chains,
// dependent vs independent, at 1..4 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND, int CHAINS>
__global__ void k(float *out, int iters, unsigned long long *clk) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f2){a[i], a[i] + 1.f}; }
    const float m = 1.0001f, c = 0.5f; const f2 m2 = {m, m}, c2 = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);
                else if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
                else if (KIND == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2)); }
                else if (KIND == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2)); }
                else if (KIND == 4) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); }
                else if (KIND == 5) { asm volatile("v_lshl_add_u32 %0, %0, 1, %0" : "+v"(a[i])); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int KIND, int CHAINS> void run(const char *name, int waves_per_simd) {
    float *out; unsigned long long *clk, h;
    hipMalloc(&out, 4); hipMalloc(&clk, 8);
    const int iters = 2000, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd, blocks = 256 * ((256 * waves_per_simd + threads - 1) / threads);
    k<KIND, CHAINS><<<blocks, threads>>>(out, iters, clk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<KIND, CHAINS><<<blocks, threads>>>(out, iters, clk); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * CHAINS;
    printf("%-14s chains %d waves/SIMD %d: %.2f clk per instr per wave (memtime), %.2f clk per instr per SIMD (wall @2.4GHz)\n", name, CHAINS,
           waves_per_simd, h / n, ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
}
int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 1>("v_fma_f32", w); run<0, 8>("v_fma_f32", w);
        run<1, 1>("v_pk_fma_f32", w); run<1, 8>("v_pk_fma_f32", w);
        run<2, 1>("v_pk_mul_f32", w); run<2, 8>("v_pk_mul_f32", w);
        run<3, 8>("v_pk_add_f32", w);
        run<4, 8>("v_mad_u32_u24", w); run<5, 1>("v_lshl_add_u32", w); run<5, 8>("v_lshl_add_u32", w);
    }
    return 0;
}
