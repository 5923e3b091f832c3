/* Calling the C ABI from plain C (C99): what a cgo / JNI / N-API binding does underneath.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_example.c -Lneurite_amd/lib -lneurite_amd -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/neurite_amd/lib -Wl,-rpath,/opt/rocm/lib -o c_abi_example
 *
 * Without arguments it only queries the library (runs anywhere); with `run` it warps a 32^3 x 4 volume by a constant
 * half-voxel shift on the first GPU and checks one interior value against the closed form.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "neurite_amd.h"

/* the handful of HIP runtime entry points the example needs (declared here so that it stays plain C) */
extern int hipMalloc(void **ptr, size_t size);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind);
extern int hipDeviceSynchronize(void);

int main(int argc, char **argv) {
    printf("neurite_amd C ABI %d for %s; status 0 = \"%s\"\n", nrt_abi_version(), nrt_target_arch(), nrt_status_string(0));
    if (argc < 2 || strcmp(argv[1], "run") != 0) return 0;

    enum { S = 32, C = 4 };
    const size_t nvox = (size_t)S * S * S;
    float *h_vol = (float *)malloc(nvox * C * sizeof(float));
    float *h_trf = (float *)malloc(nvox * 3 * sizeof(float));
    float *h_out = (float *)malloc(nvox * C * sizeof(float));
    for (size_t v = 0; v < nvox; ++v) {
        const int z = (int)(v % S);
        for (int c = 0; c < C; ++c) h_vol[v * C + c] = (float)(z + c);      /* linear in z: interpolation is exact */
        h_trf[v * 3 + 0] = 0.0f; h_trf[v * 3 + 1] = 0.0f; h_trf[v * 3 + 2] = 0.5f;
    }
    void *d_vol, *d_trf, *d_out;
    if (hipMalloc(&d_vol, nvox * C * sizeof(float)) || hipMalloc(&d_trf, nvox * 3 * sizeof(float)) ||
        hipMalloc(&d_out, nvox * C * sizeof(float))) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    hipMemcpy(d_vol, h_vol, nvox * C * sizeof(float), 1 /* hipMemcpyHostToDevice */);
    hipMemcpy(d_trf, h_trf, nvox * 3 * sizeof(float), 1);
    const int shape[3] = {S, S, S};
    const int rc = nrt_interpn_f32((const float *)d_vol, (const float *)d_trf, (float *)d_out, 3, shape, shape, C, 1,
                                   (long long)(nvox * C), (long long)(nvox * 3), NRT_LOC_SHIFT, NRT_INTERP_LINEAR,
                                   0, 0.0f, NULL);
    if (rc != NRT_OK) { fprintf(stderr, "nrt_interpn_f32: %s\n", nrt_status_string(rc)); return 3; }
    hipDeviceSynchronize();
    hipMemcpy(h_out, d_out, nvox * C * sizeof(float), 2 /* hipMemcpyDeviceToHost */);
    const size_t probe = ((size_t)5 * S + 6) * S + 7;                         /* voxel (5, 6, 7), channel 1: 7.5 + 1 */
    const float got = h_out[probe * C + 1];
    printf("out[5,6,7,1] = %g (expected 8.5)\n", got);
    hipFree(d_vol); hipFree(d_trf); hipFree(d_out);
    free(h_vol); free(h_trf); free(h_out);
    return got == 8.5f ? 0 : 1;
}
