// Throughput (clk per wave64 instruction per SIMD, at 2 waves/SIMD, 8 independent chains) of the VALU ops the gather uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OPS(X) \
  X(0, "v_add_u32 (VOP2)", "v_add_u32_e32 %0, %1, %0") \
  X(1, "v_and_b32", "v_and_b32_e32 %0, %1, %0") \
  X(2, "v_lshlrev_b32", "v_lshlrev_b32_e32 %0, 1, %0") \
  X(3, "v_cndmask_b32", "v_cndmask_b32_e32 %0, %1, %0, vcc") \
  X(4, "v_mul_f32", "v_mul_f32_e32 %0, %1, %0") \
  X(5, "v_add_f32", "v_add_f32_e32 %0, %1, %0") \
  X(6, "v_min_f32", "v_min_f32_e32 %0, %1, %0") \
  X(7, "v_floor_f32", "v_floor_f32_e32 %0, %0") \
  X(8, "v_cvt_i32_f32", "v_cvt_i32_f32_e32 %0, %0") \
  X(9, "v_mul_u32_u24 (VOP2)", "v_mul_u32_u24_e32 %0, %1, %0") \
  X(10, "v_mul_lo_u32", "v_mul_lo_u32 %0, %1, %0") \
  X(11, "v_mad_u32_u24", "v_mad_u32_u24 %0, %1, %0, %0") \
  X(12, "v_perm_b32", "v_perm_b32 %0, %1, %0, %1") \
  X(13, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 16") \
  X(14, "v_min3_f32", "v_min3_f32 %0, %1, %0, %0") \
  X(15, "v_mov_b32_dpp quad", "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(16, "v_pk_min_u16", "v_pk_min_u16 %0, %1, %0") \
  X(17, "v_add3_u32", "v_add3_u32 %0, %1, %0, %0") \
  X(18, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1") \
  X(19, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1") \
  X(20, "v_fma_f32", "v_fma_f32 %0, %1, %0, %0") \
  X(21, "v_fmac_f32 (VOP2)", "v_fmac_f32_e32 %0, %1, %1") \
  X(22, "v_sub_f32", "v_sub_f32_e32 %0, %1, %0") \
  X(23, "v_mov_b32", "v_mov_b32_e32 %0, %1") \
  X(24, "v_max_f32", "v_max_f32_e32 %0, %1, %0") \
  X(25, "v_xor_b32", "v_xor_b32_e32 %0, %1, %0") \
  X(26, "v_mul_f32 VOP3 (e64)", "v_mul_f32_e64 %0, %1, %0") \
  X(27, "v_add_f32 e64 clamp-less", "v_add_f32_e64 %0, %1, %0") \
  X(28, "v_cmp_lt_u32+nothing", "v_cmp_lt_u32_e32 vcc, %1, %0") \
  X(29, "v_add_u32_sdwa", "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
template <int KIND>
__global__ void k(unsigned *out, int iters, unsigned long long *clk) {
    unsigned a[8]; unsigned m = threadIdx.x | 0x3f800000u;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 3 + i + 0x3f800000u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(N, NAME, ASM) if (KIND == N) asm volatile(ASM : "+v"(a[i]) : "v"(m) : "vcc");
                OPS(X)
#undef X
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345u) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int KIND> void run(const char *name) {
    unsigned *out; unsigned long long *clk, h;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&clk, 8);
    const int iters = 1000, wps = 2, threads = 512, blocks = 256;
    k<KIND><<<blocks, threads>>>(out, iters, clk);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<KIND><<<blocks, threads>>>(out, iters, clk); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * 8;
    printf("%-28s %.2f clk/instr/wave   %.2f clk/instr/SIMD (wall @2.4GHz)\n", name, h / n, ms * 1e-3 * 2.4e9 / (n * wps));
}
int main() {
#define X(N, NAME, ASM) run<N>(NAME);
    OPS(X)
#undef X
    return 0;
}
