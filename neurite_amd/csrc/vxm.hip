// voxelmorph companions that are pure index arithmetic, gfx950.
//
// affine_to_dense_shift (voxelmorph.utils, called by SpatialTransformer on affine inputs, by AffineToDenseShift and by the
// synthesis models, neurite/tf/models.py:1131-1154): shift[q] = A [q - c; 1] - (q - c) with c = (shape - 1) / 2 when
// shift_center.  The reference builds the mesh grid, stacks it to a [D + 1, V] matrix, multiplies, transposes and subtracts the
// stacked mesh -- ten passes over V-sized tensors; here one kernel writes the D floats of a voxel (12 bytes at D = 3).
// The D + 1 products of a row are added left to right, one rounding per multiply and per add (tf.matmul leaves the order
// open; the tests compare at 1e-5).

#include "nrt_common.h"

namespace {

struct AffArgs {
    const float *m;          // [batch][D][D + 1]
    float *out;              // [batch][nvox][D]
    int S[3];
    float c[3];              // centre offsets (0 without shift_center)
    long long nvox;
};

template <int D>
__global__ __launch_bounds__(256) void affine_shift(AffArgs a) {
    const int b = blockIdx.y;
    const float *m = a.m + (long long)b * D * (D + 1);
    float M[D][D + 1];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= D; ++j) M[i][j] = m[i * (D + 1) + j];
    float *out = a.out + (long long)b * a.nvox * D;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < a.nvox; q += (long long)gridDim.x * 256) {
        int idx[3];
        long long r = q;
#pragma unroll
        for (int d = D - 1; d >= 0; --d) { idx[d] = (int)(r % a.S[d]); r /= a.S[d]; }
        float p[D];
#pragma unroll
        for (int d = 0; d < D; ++d) p[d] = nrt_sub((float)idx[d], a.c[d]);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float s = nrt_mul(M[i][0], p[0]);
#pragma unroll
            for (int j = 1; j < D; ++j) s = nrt_add(s, nrt_mul(M[i][j], p[j]));
            s = nrt_add(s, M[i][D]);                         // the homogeneous coordinate is 1
            __builtin_nontemporal_store(nrt_sub(s, p[i]), out + q * D + i);
        }
    }
}

}  // namespace

extern "C" int nrt_affine_to_dense_shift_f32(const float *matrix, int batch, int ndim, const int *shape, int shift_center, float *out,
                                             void *stream) {
    if (!matrix || !shape || !out || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (ndim != 2 && ndim != 3) return NRT_ERR_UNSUPPORTED;
    AffArgs a;
    a.m = matrix; a.out = out; a.nvox = 1;
    for (int d = 0; d < 3; ++d) {
        a.S[d] = d < ndim ? shape[d] : 1;
        if (a.S[d] < 1) return NRT_ERR_INVALID_ARG;
        a.c[d] = (shift_center && d < ndim) ? (float)(a.S[d] - 1) / 2.0f : 0.0f;
        a.nvox *= a.S[d];
    }
    long long blocks = (a.nvox + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks, (unsigned)batch);
    if (ndim == 2) hipLaunchKernelGGL((affine_shift<2>), grid, dim3(256), 0, nrt_stream(stream), a);
    else hipLaunchKernelGGL((affine_shift<3>), grid, dim3(256), 0, nrt_stream(stream), a);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
