// Soft quantisation and mutual information (neurite/tf/utils/utils.py:1099-1172,
// neurite/tf/metrics.py:41-336), gfx950.
//
// MutualInformation.volumes / channelwise soft-quantise both images into nb bins (w_b(v) = exp(-alpha (x_v - c_b)^2)),
// build the joint histogram J[i][j] = sum_v wx_i(v) wy_j(v) with a batched matmul over the [V, nb] maps, and reduce it to
// one number per (batch, channel).  The [V, nb] maps are 16x the images and exist only to be contracted: here they never
// do.  A wave computes the bin weights in registers in the layout of the fp32 MFMA operands -- lane l holds
// wx[bin = l & 15][voxel = l >> 4] and wy[voxel = l >> 4][bin = l & 15] -- so one v_mfma_f32_16x16x4_f32 adds four voxels
// to a 16 x 16 tile of J; the marginals are the lanes' running sums.  The kernel reads 8 bytes per voxel and writes nb^2
// floats per item.  The [items, nb, nb]-sized arithmetic that follows (normalisation, log, sum) is host-side glue.
// The backward recomputes the weights: dL/dx_v = sum_i (sum_j G_ij wy_j(v) + gx_i) * wx_i(v) * (-2 alpha (x_v - c_i)).

#include <cstdlib>

#include "nrt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MiArgs {
    const float *x, *y;      // item (b, c) voxel v at [(b * V + v) * C + c]
    const float *cx, *cy;    // bin centres [nb] (device)
    float alpha, lo, hi;     // clip range
    long long V;
    int C, nb, items;        // items = B * C
    float *joint, *sx, *sy;  // [items, nb, nb], [items, nb], [items, nb]; accumulated with atomics (zero-filled by the caller)
};

__device__ __forceinline__ float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// NBT = ceil(nb / 16) tiles per side (1 or 2); grid (chunks, items)
template <int NBT>
__global__ __launch_bounds__(256) void mi_joint(MiArgs a) {
    const int item = blockIdx.y, b = item / a.C, c = item % a.C;
    const float *xb = a.x + ((long long)b * a.V) * a.C + c;
    const float *yb = a.y + ((long long)b * a.V) * a.C + c;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float cxv[NBT], cyv[NBT];
    bool live[NBT];
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
        live[t] = 16 * t + l15 < a.nb;
        cxv[t] = live[t] ? a.cx[16 * t + l15] : 0.0f;
        cyv[t] = live[t] ? a.cy[16 * t + l15] : 0.0f;
    }
    f32x4 acc[NBT][NBT];
    float sx[NBT], sy[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
        sx[i] = 0.0f; sy[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NBT; ++j) acc[i][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    // a wave takes 64 voxels at a time (one coalesced load), 16 MFMA steps of 4 voxels
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long v0 = ((long long)blockIdx.x * 4 + wv) * 64; v0 < a.V; v0 += nwaves * 64) {
        const long long v = v0 + lane;
        const bool in = v < a.V;
        const float xl = in ? clipf(xb[v * a.C], a.lo, a.hi) : 0.0f;
        const float yl = in ? clipf(yb[v * a.C], a.lo, a.hi) : 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int src = 4 * s + l4;
            const float xv = __shfl(xl, src, 64), yv = __shfl(yl, src, 64);
            const bool vin = v0 + src < a.V;
            float wa[NBT], wb[NBT];
#pragma unroll
            for (int t = 0; t < NBT; ++t) {
                const float dx = xv - cxv[t], dy = yv - cyv[t];
                wa[t] = (live[t] && vin) ? __expf(-a.alpha * (dx * dx)) : 0.0f;
                wb[t] = (live[t] && vin) ? __expf(-a.alpha * (dy * dy)) : 0.0f;
                sx[t] += wa[t]; sy[t] += wb[t];
            }
#pragma unroll
            for (int i = 0; i < NBT; ++i)
#pragma unroll
                for (int j = 0; j < NBT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], wb[j], acc[i][j], 0, 0, 0);
        }
    }
    // marginals: sum the four voxel slots of a bin (lanes l, l^16, l^32, l^48)
    // The four waves of the block are summed in LDS first and a block sends ONE atomic per histogram entry: every wave of
    // every block adding to the same nb^2 + 2 nb addresses of an item (round 1: 4 x 1000 same-address atomics per entry and item)
    // serialised in L2 and cost more than the arithmetic.
    constexpr int NJ = NBT * NBT * 256, NM = NBT * 16;
    __shared__ float red[4][NJ + 2 * NM];
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
        sx[t] += __shfl_xor(sx[t], 16, 64); sx[t] += __shfl_xor(sx[t], 32, 64);
        sy[t] += __shfl_xor(sy[t], 16, 64); sy[t] += __shfl_xor(sy[t], 32, 64);
        if (l4 == 0) { red[wv][NJ + 16 * t + l15] = sx[t]; red[wv][NJ + NM + 16 * t + l15] = sy[t]; }
    }
    // joint: D row = 4 (l >> 4) + r (x bin), col = l & 15 (y bin); tile (i, j) holds entry [row][col] at (i NBT + j) 256 + row 16 + col
#pragma unroll
    for (int i = 0; i < NBT; ++i)
#pragma unroll
        for (int j = 0; j < NBT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv][(i * NBT + j) * 256 + (4 * l4 + r) * 16 + l15] = acc[i][j][r];
    __syncthreads();
    for (int e = threadIdx.x; e < NJ + 2 * NM; e += 256) {
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (e < NJ) {
            const int tile = e >> 8, row = (e >> 4) & 15, col = e & 15;
            const int bi = 16 * (tile / NBT) + row, bj = 16 * (tile % NBT) + col;
            if (bi < a.nb && bj < a.nb) unsafeAtomicAdd(&a.joint[((long long)item * a.nb + bi) * a.nb + bj], v);
        } else {
            const int m = e - NJ, which = m / NM, bin = m % NM;
            if (bin < a.nb) unsafeAtomicAdd(&(which ? a.sy : a.sx)[(long long)item * a.nb + bin], v);
        }
    }
}

// backward: one thread per voxel of an item; G [items, nb, nb], gsx / gsy [items, nb]
__global__ __launch_bounds__(256) void mi_joint_bwd(MiArgs a, const float *__restrict__ G, const float *__restrict__ gsx,
                                                    const float *__restrict__ gsy, float *__restrict__ gx, float *__restrict__ gy) {
    extern __shared__ float sm[];          // G [nb][nb], gsx [nb], gsy [nb], cx [nb], cy [nb]
    const int item = blockIdx.y, b = item / a.C, c = item % a.C;
    const int nb = a.nb;
    float *sG = sm, *sgx = sm + nb * nb, *sgy = sgx + nb, *scx = sgy + nb, *scy = scx + nb;
    for (int i = threadIdx.x; i < nb * nb; i += 256) sG[i] = G[(long long)item * nb * nb + i];
    for (int i = threadIdx.x; i < nb; i += 256) {
        sgx[i] = gsx[(long long)item * nb + i]; sgy[i] = gsy[(long long)item * nb + i];
        scx[i] = a.cx[i]; scy[i] = a.cy[i];
    }
    __syncthreads();
    const long long base = ((long long)b * a.V) * a.C + c;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < a.V; v += (long long)gridDim.x * 256) {
        const float xr = a.x[base + v * a.C], yr = a.y[base + v * a.C];
        const float xv = clipf(xr, a.lo, a.hi), yv = clipf(yr, a.lo, a.hi);
        const bool xin = xr >= a.lo && xr <= a.hi, yin = yr >= a.lo && yr <= a.hi;      // clip_by_value passes the gradient inside
        float wy[32], dgx = 0.0f, dgy = 0.0f;
        for (int j = 0; j < nb; ++j) { const float d = yv - scy[j]; wy[j] = __expf(-a.alpha * d * d); }
        float ty[32];
        for (int j = 0; j < nb; ++j) ty[j] = sgy[j];
        for (int i = 0; i < nb; ++i) {
            const float d = xv - scx[i];
            const float wx = __expf(-a.alpha * d * d);
            float t = sgx[i];
            for (int j = 0; j < nb; ++j) { t += sG[i * nb + j] * wy[j]; ty[j] += sG[i * nb + j] * wx; }
            dgx += t * wx * (-2.0f * a.alpha * d);
        }
        for (int j = 0; j < nb; ++j) dgy += ty[j] * wy[j] * (-2.0f * a.alpha * (yv - scy[j]));
        if (gx) gx[base + v * a.C] = xin ? dgx : 0.0f;
        if (gy) gy[base + v * a.C] = yin ? dgy : 0.0f;
    }
}

// backward on the matrix cores.  For 16 voxels at a time a wave forms
//   T[i][v] = sum_j G[i][j] wy_j(v)      (d L / d wx_i(v) = T + gsx[i])        A = G,   B = wy
//   U[j][v] = sum_i G[i][j] wx_i(v)      (d L / d wy_j(v) = U + gsy[j])        A = G^T, B = wx
// with v_mfma_f32_16x16x4_f32 (operand layout as in mi_joint: lane l holds A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15],
// D[row = 4 (l >> 4) + r][col = l & 15]).  The k index of step t in lane group g = l >> 4 is bin 4 g + t, so the four bin weights a
// lane computes as B operands are exactly the four it needs for its rows of D in the epilogue: 32 exponentials per voxel, as in
// the forward.  d L / d x_v = sum_i (T[i][v] + gsx[i]) wx_i(v) (-2 alpha (x_v - c_i)) is summed over the lane's four rows and then
// over the four lane groups (xor 16, 32); every lane keeps the result of the sub-step its own voxel belongs to, so the gradient
// of 64 voxels leaves as one coalesced store.  (The scalar kernel above keeps two nb-sized arrays per thread, which live in
// scratch memory: 1.25 ms for one gradient of 4 x 160^3 against 0.22 ms for the forward.)
template <int NBT>
__global__ __launch_bounds__(256) void mi_joint_bwd_mfma(MiArgs a, const float *__restrict__ G, const float *__restrict__ gsx,
                                                         const float *__restrict__ gsy, float *__restrict__ gx, float *__restrict__ gy) {
    const int item = blockIdx.y, b = item / a.C, c = item % a.C;
    const int nb = a.nb;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const float *Gi = G + (long long)item * nb * nb;
    // loop invariants: the lane's elements of G (both operand orders), its bins' centres and marginal gradients
    float gT[NBT][NBT][4], gU[NBT][NBT][4];                 // gT[it][jt][t] = G[16 it + l15][16 jt + 4 g4 + t]; gU[jt][it][t] = G[16 it + 4 g4 + t][16 jt + l15]
    float cxb[NBT][4], cyb[NBT][4], gsxb[NBT][4], gsyb[NBT][4];
    bool liveb[NBT][4];
#pragma unroll
    for (int it = 0; it < NBT; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int bin = 16 * it + 4 * g4 + t;
            liveb[it][t] = bin < nb;
            cxb[it][t] = liveb[it][t] ? a.cx[bin] : 0.0f;
            cyb[it][t] = liveb[it][t] ? a.cy[bin] : 0.0f;
            gsxb[it][t] = liveb[it][t] ? gsx[(long long)item * nb + bin] : 0.0f;
            gsyb[it][t] = liveb[it][t] ? gsy[(long long)item * nb + bin] : 0.0f;
        }
#pragma unroll
    for (int it = 0; it < NBT; ++it)
#pragma unroll
        for (int jt = 0; jt < NBT; ++jt)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int iT = 16 * it + l15, jT = 16 * jt + 4 * g4 + t;
                gT[it][jt][t] = (iT < nb && jT < nb) ? Gi[iT * nb + jT] : 0.0f;
                const int iU = 16 * it + 4 * g4 + t, jU = 16 * jt + l15;
                gU[jt][it][t] = (iU < nb && jU < nb) ? Gi[iU * nb + jU] : 0.0f;
            }
    const long long base = ((long long)b * a.V) * a.C + c;
    const float m2a = -2.0f * a.alpha;
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long v0 = ((long long)blockIdx.x * 4 + wv) * 64; v0 < a.V; v0 += nwaves * 64) {
        const long long v = v0 + lane;
        const bool in = v < a.V;
        const float xr = in ? a.x[base + v * a.C] : 0.0f, yr = in ? a.y[base + v * a.C] : 0.0f;
        const float xl = clipf(xr, a.lo, a.hi), yl = clipf(yr, a.lo, a.hi);
        float minex = 0.0f, miney = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int src = 16 * s + l15;                    // this sub-step's voxel of the lane's column
            const float xv = __shfl(xl, src, 64), yv = __shfl(yl, src, 64);
            float wx[NBT][4], wy[NBT][4], dx[NBT][4], dy[NBT][4];
#pragma unroll
            for (int it = 0; it < NBT; ++it)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dx[it][t] = xv - cxb[it][t]; dy[it][t] = yv - cyb[it][t];
                    wx[it][t] = liveb[it][t] ? __expf(-a.alpha * dx[it][t] * dx[it][t]) : 0.0f;
                    wy[it][t] = liveb[it][t] ? __expf(-a.alpha * dy[it][t] * dy[it][t]) : 0.0f;
                }
            if (gx) {
                float part = 0.0f;
#pragma unroll
                for (int it = 0; it < NBT; ++it) {
                    f32x4 T = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < NBT; ++jt)
#pragma unroll
                        for (int t = 0; t < 4; ++t) T = __builtin_amdgcn_mfma_f32_16x16x4f32(gT[it][jt][t], wy[jt][t], T, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) part += (T[r] + gsxb[it][r]) * wx[it][r] * (m2a * dx[it][r]);
                }
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                if (s == g4) minex = part;
            }
            if (gy) {
                float part = 0.0f;
#pragma unroll
                for (int jt = 0; jt < NBT; ++jt) {
                    f32x4 U = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int it = 0; it < NBT; ++it)
#pragma unroll
                        for (int t = 0; t < 4; ++t) U = __builtin_amdgcn_mfma_f32_16x16x4f32(gU[jt][it][t], wx[it][t], U, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) part += (U[r] + gsyb[jt][r]) * wy[jt][r] * (m2a * dy[jt][r]);
                }
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                if (s == g4) miney = part;
            }
        }
        if (in) {
            const bool xin = xr >= a.lo && xr <= a.hi, yin = yr >= a.lo && yr <= a.hi;      // clip_by_value passes the gradient inside
            if (gx) gx[base + v * a.C] = xin ? minex : 0.0f;
            if (gy) gy[base + v * a.C] = yin ? miney : 0.0f;
        }
    }
}

__global__ __launch_bounds__(256) void soft_quantize(const float *__restrict__ x, const float *__restrict__ centers, float alpha,
                                                     float lo, float hi, int ret_log, float *__restrict__ out, long long n, int nb) {
    const long long total = n * nb;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long v = e / nb;
        const int bidx = (int)(e - v * nb);
        const float d = clipf(x[v], lo, hi) - centers[bidx];
        const float lg = -alpha * (d * d);
        out[e] = ret_log ? lg : expf(lg);
    }
}

// backward of soft_quantize (neurite/tf/utils/utils.py:1099-1172) wrt x with the bin centres held constant:
//   out_b = exp(-alpha (xc - c_b)^2)  (or -alpha (xc - c_b)^2 with return_log),  xc = clip(x, lo, hi)
//   d out_b / d x = [lo <= x <= hi] * (-2 alpha (xc - c_b)) * (out_b | 1)
// One thread per voxel sums over the bins (g is [n, nb], read row-wise).
__global__ __launch_bounds__(256) void soft_quantize_bwd(const float *__restrict__ x, const float *__restrict__ centers, float alpha,
                                                         float lo, float hi, int ret_log, const float *__restrict__ g,
                                                         float *__restrict__ gx, long long n, int nb, int staged) {
    extern __shared__ float sq_lds[];       // staged: the gradient rows of the block's 256 voxels, read with coalesced loads
    for (long long v0 = (long long)blockIdx.x * 256; v0 < n; v0 += (long long)gridDim.x * 256) {
        const long long v = v0 + threadIdx.x;
        const int nv = (int)((n - v0) < 256 ? (n - v0) : 256);
        const float *grow = g + v * nb;
        if (staged) {
            __syncthreads();
            for (long long i = threadIdx.x; i < (long long)nv * nb; i += 256) sq_lds[i] = g[v0 * nb + i];
            __syncthreads();
            grow = sq_lds + (long long)threadIdx.x * nb;
        }
        if (v < n) {
            const float xv = x[v];
            const float xc = clipf(xv, lo, hi);
            const float pass = (xv >= lo && xv <= hi) ? 1.0f : 0.0f;          // tf.clip_by_value passes the gradient on the closed range
            float acc = 0.0f;
            for (int b = 0; b < nb; ++b) {
                const float d = xc - centers[b];
                const float dd = -2.0f * alpha * d;
                acc += grow[b] * (ret_log ? dd : dd * expf(-alpha * (d * d)));
            }
            gx[v] = pass * acc;
        }
    }
}

// column sums of a [n, C] matrix per batch item: out[item][c] += sum_v x[item][v][c]  (atomics, zero-filled by the caller)
__global__ __launch_bounds__(256) void colsum(const float *__restrict__ x, long long n, int C, float *__restrict__ out) {
    extern __shared__ float sm[];
    const int item = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += 256) sm[i] = 0.0f;
    __syncthreads();
    const float *xb = x + (long long)item * n * C;
    const long long total = n * C;
    // a thread keeps one column when 256 % C == 0 (the stride preserves e % C); otherwise LDS atomics per element
    if (256 % C == 0) {
        float s = 0.0f;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) s += xb[e];
        atomicAdd(&sm[threadIdx.x % C], s);
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
            atomicAdd(&sm[(int)(e % C)], xb[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) unsafeAtomicAdd(&out[(long long)item * C + i], sm[i]);
}

// metrics.py:262-281 on one item's joint histogram and marginal sums: pxy = J / (sum J + eps), px = sx / (sum sx + eps),
// py likewise, mi = sum_ij pxy log(pxy / (px_i py_j + eps) + eps).  One block per item (the inference path; under autograd the
// same arithmetic runs as torch ops so that d mi / d J comes from autodiff).
__device__ __forceinline__ float block_sum(float v, float *sm) {          // every lane gets the block's sum
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                                                       // sm may still be read from a previous call
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void mi_from_joint(const float *__restrict__ J, const float *__restrict__ sx,
                                                     const float *__restrict__ sy, int nb, float eps, float *__restrict__ out) {
    __shared__ float sm[4], px[64], py[64];
    const int item = blockIdx.x;
    const float *Ji = J + (long long)item * nb * nb;
    float a = 0.0f;
    for (int e = threadIdx.x; e < nb * nb; e += 256) a += Ji[e];
    const float sj = block_sum(a, sm);
    const float tsx = block_sum((int)threadIdx.x < nb ? sx[(long long)item * nb + threadIdx.x] : 0.0f, sm);
    const float tsy = block_sum((int)threadIdx.x < nb ? sy[(long long)item * nb + threadIdx.x] : 0.0f, sm);
    if ((int)threadIdx.x < nb) {
        px[threadIdx.x] = sx[(long long)item * nb + threadIdx.x] / (tsx + eps);
        py[threadIdx.x] = sy[(long long)item * nb + threadIdx.x] / (tsy + eps);
    }
    __syncthreads();
    float t = 0.0f;
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const float pxy = Ji[e] / (sj + eps);
        const float pp = px[e / nb] * py[e % nb] + eps;
        t += pxy * logf(pxy / pp + eps);
    }
    const float mi = block_sum(t, sm);
    if (threadIdx.x == 0) out[item] = mi;
}

unsigned mblocks(long long n, int per) {
    long long b = (n + per - 1) / per;
    if (b > 256ll * 8) b = 256ll * 8;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int nrt_mi_joint_f32(const float *x, const float *y, const float *centers_x, const float *centers_y, float alpha,
                                float min_clip, float max_clip, int batch, long long nvox, int channels, int nb_bins, float *joint,
                                float *sum_x, float *sum_y, void *stream) {
    if (!x || !y || !centers_x || !centers_y || !joint || !sum_x || !sum_y) return NRT_ERR_INVALID_ARG;
    if (batch < 1 || nvox < 0 || channels < 1 || nb_bins < 1) return NRT_ERR_INVALID_ARG;
    if (nb_bins > 32 || (long long)batch * channels > 65535) return NRT_ERR_UNSUPPORTED;
    if (nvox == 0) return NRT_OK;
    MiArgs a;
    a.x = x; a.y = y; a.cx = centers_x; a.cy = centers_y; a.alpha = alpha; a.lo = min_clip; a.hi = max_clip;
    a.V = nvox; a.C = channels; a.nb = nb_bins; a.items = batch * channels; a.joint = joint; a.sx = sum_x; a.sy = sum_y;
    unsigned chunks = mblocks(nvox, 256 * 16);
    if (chunks > 256u) chunks = 256u;                       // <= 256 atomics per histogram entry and item
    dim3 grid(chunks, (unsigned)a.items);
    if (nb_bins <= 16) hipLaunchKernelGGL((mi_joint<1>), grid, dim3(256), 0, nrt_stream(stream), a);
    else hipLaunchKernelGGL((mi_joint<2>), grid, dim3(256), 0, nrt_stream(stream), a);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_mi_joint_bwd_f32(const float *x, const float *y, const float *centers_x, const float *centers_y, float alpha,
                                    float min_clip, float max_clip, int batch, long long nvox, int channels, int nb_bins,
                                    const float *grad_joint, const float *grad_sum_x, const float *grad_sum_y, float *grad_x,
                                    float *grad_y, void *stream) {
    if (!x || !y || !centers_x || !centers_y || !grad_joint || !grad_sum_x || !grad_sum_y || (!grad_x && !grad_y)) return NRT_ERR_INVALID_ARG;
    if (batch < 1 || nvox < 0 || channels < 1 || nb_bins < 1) return NRT_ERR_INVALID_ARG;
    if (nb_bins > 32 || (long long)batch * channels > 65535) return NRT_ERR_UNSUPPORTED;
    if (nvox == 0) return NRT_OK;
    MiArgs a;
    a.x = x; a.y = y; a.cx = centers_x; a.cy = centers_y; a.alpha = alpha; a.lo = min_clip; a.hi = max_clip;
    a.V = nvox; a.C = channels; a.nb = nb_bins; a.items = batch * channels; a.joint = nullptr; a.sx = nullptr; a.sy = nullptr;
    const char *sc = getenv("NRT_MI_BWD_SCALAR");                 // tests: the one-thread-per-voxel kernel
    if (sc && sc[0] == '1') {
        const size_t shm = (size_t)(nb_bins * nb_bins + 4 * nb_bins) * sizeof(float);
        hipLaunchKernelGGL(mi_joint_bwd, dim3(mblocks(nvox, 256), (unsigned)a.items), dim3(256), shm, nrt_stream(stream), a, grad_joint,
                           grad_sum_x, grad_sum_y, grad_x, grad_y);
    } else {
        dim3 grid(mblocks(nvox, 256 * 4), (unsigned)a.items);
        if (nb_bins <= 16) hipLaunchKernelGGL((mi_joint_bwd_mfma<1>), grid, dim3(256), 0, nrt_stream(stream), a, grad_joint, grad_sum_x,
                                              grad_sum_y, grad_x, grad_y);
        else hipLaunchKernelGGL((mi_joint_bwd_mfma<2>), grid, dim3(256), 0, nrt_stream(stream), a, grad_joint, grad_sum_x, grad_sum_y,
                                grad_x, grad_y);
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_soft_quantize_f32(const float *x, const float *centers, float alpha, float min_clip, float max_clip,
                                     int return_log, float *out, long long n, int nb_bins, void *stream) {
    if (!x || !centers || !out || n < 0 || nb_bins < 1) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    hipLaunchKernelGGL(soft_quantize, dim3(mblocks(n * nb_bins, 256)), dim3(256), 0, nrt_stream(stream), x, centers, alpha, min_clip,
                       max_clip, return_log, out, n, nb_bins);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_soft_quantize_bwd_f32(const float *x, const float *centers, float alpha, float min_clip, float max_clip,
                                         int return_log, const float *grad_out, float *grad_x, long long n, int nb_bins, void *stream) {
    if (!x || !centers || !grad_out || !grad_x || n < 0 || nb_bins < 1) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    const int staged = (size_t)256 * nb_bins * sizeof(float) <= 48 * 1024;      // a lane reading its own row touches a line per bin
    hipLaunchKernelGGL(soft_quantize_bwd, dim3(mblocks(n, 256)), dim3(256), staged ? (size_t)256 * nb_bins * sizeof(float) : 0,
                       nrt_stream(stream), x, centers, alpha, min_clip, max_clip, return_log, grad_out, grad_x, n, nb_bins, staged);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_colsum_f32(const float *x, int items, long long rows, int cols, float *out, void *stream) {
    if (!x || !out || items < 1 || items > 65535 || rows < 0 || cols < 1 || cols > 4096) return NRT_ERR_INVALID_ARG;
    if (rows == 0) return NRT_OK;
    unsigned chunks = mblocks(rows * cols, 256 * 32);
    if (chunks > 256u) chunks = 256u;                       // one atomic per column and block: keep the same-address chain short
    hipLaunchKernelGGL(colsum, dim3(chunks, (unsigned)items), dim3(256), (size_t)cols * sizeof(float),
                       nrt_stream(stream), x, rows, cols, out);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_mi_from_joint_f32(const float *joint, const float *sum_x, const float *sum_y, int items, int nb_bins, float eps,
                                     float *mi, void *stream) {
    if (!joint || !sum_x || !sum_y || !mi || items < 1 || nb_bins < 1) return NRT_ERR_INVALID_ARG;
    if (nb_bins > 64) return NRT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mi_from_joint, dim3((unsigned)items), dim3(256), 0, nrt_stream(stream), joint, sum_x, sum_y, nb_bins, eps, mi);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
