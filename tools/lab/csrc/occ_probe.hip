// Occupancy facts for LDS-heavy workgroups on gfx950: how many 128-thread workgroups with N KB of LDS are resident per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BYTES> __global__ __launch_bounds__(128) void k(float *out, unsigned long long *clk) {
    __shared__ unsigned char sm[BYTES];
    sm[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 2000000ull) {}    // ~20 ms at 100 MHz
    if (threadIdx.x == 0) { out[blockIdx.x] = sm[5]; }
}
template <int BYTES> void run() {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<BYTES>, 128, 0);
    float *out; unsigned long long *clk;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&clk, 8);
    // timing: 256 CUs x m blocks: if all resident the launch takes one spin period
    for (int m : {1, 2, 3, 4, 5}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); k<BYTES><<<256 * m, 128>>>(out, clk); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("LDS %6d B: occupancy API %d blocks/CU; %d blocks per CU launched -> %.1f ms\n", BYTES, nb, m, ms);
    }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    run<39264>(); run<32768>(); run<40960>(); run<20480>();
    return 0;
}
