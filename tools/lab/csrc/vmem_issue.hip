// Per-wave / per-CU rate of row-gather loads (8 lanes x 16 B per 128-B row, 8 rows per wave instruction) on gfx950:
// plain global_load_dwordx4 into registers vs LDS-DMA (global_load_lds_dwordx4), K instructions in flight per wave per round,
// W waves per CU.  Rows are random inside a window that is either L2-resident (per XCD) or HBM-sized.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int K, bool DMA, int NV = 0>
__global__ __launch_bounds__(64) void k(const char *src, unsigned nrows_mask, int rounds, float *sink, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x, g = lane >> 3, lg = lane & 7;
    unsigned seed = (blockIdx.x * 64u + g) * 2654435761u + 12345u;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)sm);
    f4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        unsigned off[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { seed = seed * 1664525u + 1013904223u; off[j] = ((seed >> 8) & nrows_mask) * 128u + lg * 16u; }
        if (DMA) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                glds16(src, off[j], lds + j * 1024);
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc[q & 3]));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += ((const f4 *)sm)[lane];
        } else {
            f4 v[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                v[j] = *(const f4 *)(src + off[j]);
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc[q & 3]));
            }
#pragma unroll
            for (int j = 0; j < K; ++j) acc += v[j];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] + acc[1] == 1234.5f) sink[0] = acc[2];
    if (lane == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int K, bool DMA, int NV = 0> void run(const char *src, unsigned mask, int waves_per_cu, const char *what) {
    float *sink; unsigned long long *clk, h;
    (void)hipMalloc(&sink, 4); (void)hipMalloc(&clk, 8);
    const int rounds = 400, blocks = 256 * waves_per_cu;
    const size_t lds = DMA ? K * 1024 : 0;
    k<K, DMA, NV><<<blocks, 64, lds>>>(src, mask, 50, sink, clk);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<K, DMA, NV><<<blocks, 64, lds>>>(src, mask, rounds, sink, clk); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double lines_wave = (double)rounds * K * 8;
    printf("%-5s %-4s NV=%2d K=%2d waves/CU=%2d: %7.1f clk per instr per wave, %6.2f clk per line per CU (wall@2.4GHz), %6.2f TB/s\n", what, DMA ? "dma" : "reg", NV, K,
           waves_per_cu, h / ((double)rounds * K), ms * 1e-3 * 2.4e9 / (lines_wave * waves_per_cu), lines_wave * blocks * 128 / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = 1ull << 31;   // 2 GiB
    char *src; (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes);
    struct { unsigned mask; const char *what; } W[2] = {{(1u << 14) - 1, "L2"}, {(1u << 24) - 1, "HBM"}};   // 2 MiB / 2 GiB of rows
    for (int wpc : {1, 4, 5}) {
        run<24, true, 0>(src, W[0].mask, wpc, "L2"); run<24, true, 8>(src, W[0].mask, wpc, "L2"); run<24, true, 16>(src, W[0].mask, wpc, "L2");
        run<24, true, 32>(src, W[0].mask, wpc, "L2");
        run<24, false, 0>(src, W[0].mask, wpc, "L2"); run<24, false, 8>(src, W[0].mask, wpc, "L2"); run<24, false, 16>(src, W[0].mask, wpc, "L2");
    }
    return 0;
}
