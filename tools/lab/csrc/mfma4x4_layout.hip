// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per wave instruction), printed as a table.
//   hipcc --offload-arch=gfx950 -O2 mfma4x4_layout.hip -o mfma4x4_layout && ./mfma4x4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out) {
    const int l = threadIdx.x;
    // A = lane + 1, B = 1000 + lane: a product identifies (lane of A, lane of B) uniquely
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1000 + l), d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const double v = h[l * 4 + r];
            const int lb = (int)(v / 1000.0 / 1.0 + 0.5);          
            int fa = -1, fb = -1;
            for (int a = 0; a < 64 && fa < 0; ++a)
                for (int b = 0; b < 64; ++b)
                    if ((double)(a + 1) * (1000.0 + b) == v && a / 4 == l / 4 && b / 4 == l / 4) { fa = a; fb = b; break; }
            (void)lb;
            const int want_a = (l / 4) * 4 + r, want_b = l;         // hypothesis: D[reg r] of lane l = A(lane 4*blk + r) * B(lane l)
            if (fa != want_a || fb != want_b) ok = 0;
            if (l < 8) printf("lane %2d reg %d = %10.0f  -> A lane %2d, B lane %2d\n", l, r, v, fa, fb);
        }
    printf("hypothesis D[r] of lane l = A[lane 4*(l/4)+r] * B[lane l]: %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
