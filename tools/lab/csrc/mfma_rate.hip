// What does the fp32 matrix pipe deliver when nothing else is in the way?  Waves that only issue independent
// v_mfma_f32_16x16x4_f32 (16 accumulators), 1 / 2 / 4 waves per SIMD on every CU: TFLOP/s against the 157.3 nominal
// (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz) gives the clock the chip actually holds under a matrix load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a, float b) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3];
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 16 * 2048.0;
            printf("{\"probe\": \"mfma_f32_16x16x4\", \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_157.3\": %.3f, "
                   "\"implied_GHz\": %.3f}\n", wps, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, flop / ms / 1e9 / 157.3 * 2.4);
        }
    }
    return 0;
}
