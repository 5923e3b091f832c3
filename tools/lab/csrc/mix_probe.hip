// Does an L2-hitting row load cost as much as an HBM-missing one when both are issued by the same wave (in-order return)?
// Row gathers (8 lanes x 16 B per row).  "mixed": every wave alternates HBM-resident and L2-resident rows (ratio 1 : R).
// "split": 1 wave in (R+1) reads only HBM rows, the others only L2 rows -- same totals.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(64) void k(const char *src, unsigned hbm_mask, unsigned l2_mask, int ratio, int split, int rounds, float *sink) {
    const int lane = threadIdx.x, g = lane >> 3, lg = lane & 7;
    unsigned seed = (blockIdx.x * 64u + g) * 2654435761u + 12345u;
    const unsigned l2_base = (blockIdx.x % 8u) * (l2_mask + 1u);       // an L2-sized window per XCD
    f4 acc = {0, 0, 0, 0};
    const int role = split ? ((int)(blockIdx.x / 8u) % (ratio + 1) == 0 ? 1 : 2) : 0;   // 1: HBM only, 2: L2 only, 0: mixed
    for (int r = 0; r < rounds; ++r) {
        f4 v[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            seed = seed * 1664525u + 1013904223u;
            const bool hbm = role == 1 || (role == 0 && (j % (ratio + 1)) == 0);
            const unsigned row = hbm ? ((seed >> 8) & hbm_mask) : l2_base + ((seed >> 8) & l2_mask);
            v[j] = *(const f4 *)(src + (size_t)row * 128u + lg * 16u);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) acc += v[j];
    }
    if (acc[0] + acc[1] == 1234.5f) sink[0] = acc[2];
}
int main() {
    const size_t bytes = 1ull << 31;
    char *src; (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes);
    float *sink; (void)hipMalloc(&sink, 4);
    const unsigned hbm_mask = (1u << 24) - 1, l2_mask = (1u << 14) - 1;     // 2 GiB / 2 MiB of rows
    for (int ratio : {1, 2, 3}) for (int wpc : {6, 12}) for (int split = 0; split < 2; ++split) {
        const int K = 12, rounds = 300, blocks = 256 * wpc;
        k<12><<<blocks, 64>>>(src, hbm_mask, l2_mask, ratio, split, 30, sink);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); k<12><<<blocks, 64>>>(src, hbm_mask, l2_mask, ratio, split, rounds, sink); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double lines = (double)rounds * K * 8 * blocks;
        printf("HBM:L2 = 1:%d waves/CU %2d %-5s: %.3f ms, %.2f clk per line per CU, HBM lines at %.2f TB/s\n", ratio, wpc, split ? "split" : "mixed", ms,
               ms * 1e-3 * 2.4e9 * 256 / lines, lines / (ratio + 1) * 128 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
