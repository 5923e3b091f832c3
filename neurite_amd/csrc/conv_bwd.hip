// Backward kernels of the unet layer set (conv3d dgrad / wgrad and the small Keras layers), gfx950.
//
// Keras Conv3D (neurite/tf/models.py:1345-1347, 1506-1508), stride 1, SAME padding, y = act(conv(x, W) + b):
//   dpre = g * act'(y)                         nrt_act_bwd_f32   (ELU: y > 0 ? 1 : y + 1; ReLU: y > 0; sigmoid: y (1 - y))
//   dX   = conv(dpre, flip(W)^T)               the forward MFMA kernel with transformed weights (host side)
//   dW[t][ci][co] = sum_v x[v + off(t)][ci] * dpre[v][co],  db[co] = sum_v dpre[v][co]       nrt_conv3d_wgrad_f32
// MaxPooling3D (:1438), UpSampling3D (:1531, nearest) and the channel softmax (:1604) have the elementwise backward
// kernels below.
//
// wgrad as MFMA implicit GEMM (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate): for every tap the contraction runs
// over voxels, D[ci][co] += A[ci][k = voxel] * B[k = voxel][co], and both operands are rows of channels-last tiles in
// LDS -- lane l of a wave reads x[voxel(l >> 4)][16 cib + (l & 15)] and dpre[voxel(l >> 4)][16 cob + (l & 15)]: 16
// consecutive floats per voxel, conflict-free when the row stride is 16 mod 32.  A block keeps a 4 x 4 x 8 voxel
// tile of dpre and its halo tile of x in LDS, the four waves split the taps (wave w: taps w, w + 4, ...), and the
// D tiles stay in registers over all the voxel tiles a (persistent) block visits; one float atomic per weight and
// block at the end.  x is read once per (cin chunk, cout chunk) pair, weights never.

#include <type_traits>

#include "nrt_common.h"
#include "activations.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));


// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd(const nrt_f4 *__restrict__ g, const nrt_f4 *__restrict__ y, int act,
                                               nrt_f4 *__restrict__ d, long long n4) {
    long long ibeg, iend;                                      // one contiguous range per block (nrt_block_range)
    nrt_block_range(n4, 256, ibeg, iend);
    for (long long i = ibeg + threadIdx.x; i < iend; i += 256) {
        const nrt_f4 gv = g[i], yv = y[i];
        nrt_f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = gv[k] * nrt_activate_slope(yv[k], act);
        }
        d[i] = o;
    }
}

__global__ __launch_bounds__(256) void act_bwd_tail(const float *__restrict__ g, const float *__restrict__ y, int act,
                                                    float *__restrict__ d, long long from, long long n) {
    const long long i = from + threadIdx.x;
    if (i < n) {
        d[i] = g[i] * nrt_activate_slope(y[i], act);
    }
}

// ---------------------------------------------------------------------------------------------
// max pooling, stride == pool: every input voxel belongs to at most one window; the window's gradient goes to its
// first maximum (scan order x, y, z), zeros to the rest
__global__ __launch_bounds__(256) void maxpool_bwd(const float *__restrict__ x, const float *__restrict__ g,
                                                   float *__restrict__ dx, int X, int Y, int Z, int C, int OX, int OY, int OZ,
                                                   int px, int py, int pz) {
    const int b = blockIdx.y;
    const float *xb = x + (long long)b * X * Y * Z * C;
    const float *gb = g + (long long)b * OX * OY * OZ * C;
    float *db = dx + (long long)b * X * Y * Z * C;
    const long long total = (long long)OX * OY * OZ * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long long q = e / C;
        const int oz = (int)(q % OZ), oy = (int)((q / OZ) % OY), ox = (int)(q / ((long long)OZ * OY));
        float best = -INFINITY;
        long long arg = -1;
        for (int i = 0; i < px; ++i) for (int j = 0; j < py; ++j) for (int k = 0; k < pz; ++k) {
            const int xx = ox * px + i, yy = oy * py + j, zz = oz * pz + k;
            if (xx >= X || yy >= Y || zz >= Z) continue;
            const long long idx = (((long long)xx * Y + yy) * Z + zz) * C + c;
            const float v = xb[idx];
            if (arg < 0 || v > best) { best = v; arg = idx; }
        }
        const float gv = gb[e];
        for (int i = 0; i < px; ++i) for (int j = 0; j < py; ++j) for (int k = 0; k < pz; ++k) {
            const int xx = ox * px + i, yy = oy * py + j, zz = oz * pz + k;
            if (xx >= X || yy >= Y || zz >= Z) continue;
            const long long idx = (((long long)xx * Y + yy) * Z + zz) * C + c;
            db[idx] = idx == arg ? gv : 0.0f;
        }
    }
}

// nearest up-sampling backward: dlo[v] = sum of the up^3 fine voxels
__global__ __launch_bounds__(256) void upsample_sum(const float *__restrict__ g, int gstride, int goff, float *__restrict__ dlo,
                                                    int X1, int Y1, int Z1, int C, int ux, int uy, int uz) {
    const int b = blockIdx.y;
    const int Y = Y1 * uy, Z = Z1 * uz;
    const float *gb = g + (long long)b * X1 * ux * Y * Z * gstride + goff;
    float *db = dlo + (long long)b * X1 * Y1 * Z1 * C;
    const long long total = (long long)X1 * Y1 * Z1 * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long long q = e / C;
        const int z = (int)(q % Z1), y = (int)((q / Z1) % Y1), x = (int)(q / ((long long)Z1 * Y1));
        float s = 0.0f;
        for (int i = 0; i < ux; ++i) for (int j = 0; j < uy; ++j) for (int k = 0; k < uz; ++k)
            s += gb[((((long long)(x * ux + i)) * Y + (y * uy + j)) * Z + (z * uz + k)) * gstride + c];
        db[e] = s;
    }
}

// softmax backward over the last axis: dz = y * (g - sum_c g_c y_c); one thread per voxel
__global__ __launch_bounds__(256) void softmax_bwd(const float *__restrict__ y, const float *__restrict__ g,
                                                   float *__restrict__ dz, long long nvox, int C) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long long)gridDim.x * 256) {
        const float *yv = y + v * C, *gv = g + v * C;
        float dot = 0.0f;
        for (int c = 0; c < C; ++c) dot += gv[c] * yv[c];
        for (int c = 0; c < C; ++c) dz[v * C + c] = yv[c] * (gv[c] - dot);
    }
}

template <int G>
__global__ __launch_bounds__(256) void softmax_bwd_vec(const nrt_f4 *__restrict__ y, const nrt_f4 *__restrict__ g,
                                                       nrt_f4 *__restrict__ dz, long long nvox) {
    constexpr int NG = 256 / G;
    const int lg = threadIdx.x % G;
    const long long ngroups = (long long)gridDim.x * NG;
    const long long niter = (nvox + ngroups - 1) / ngroups;
    for (long long it = 0; it < niter; ++it) {
        const long long vv = ((long long)blockIdx.x * niter + it) * NG + threadIdx.x / G;      // a block streams one contiguous range (nrt_block_range)
        const bool live = vv < nvox;
        const long long v = live ? vv : nvox - 1;
        const nrt_f4 yv = y[v * G + lg], gv = g[v * G + lg];
        float dot = (gv[0] * yv[0] + gv[1] * yv[1]) + (gv[2] * yv[2] + gv[3] * yv[3]);
#pragma unroll
        for (int off = 1; off < G; off <<= 1) dot += __shfl_xor(dot, off, 64);
        nrt_f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = yv[k] * (gv[k] - dot);
        if (live) dz[v * G + lg] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad
// ---------------------------------------------------------------------------------------------
struct WgArgs {
    const float *x;          // [B, X, Y, Z, Cin]
    const float *dp;         // [B, X, Y, Z, Cout]
    float *dw;               // [kx, ky, kz, Cin, Cout], accumulated with atomics
    float *db;               // [Cout] or null, accumulated with atomics
    int B, X, Y, Z, Cin, Cout;
    int kx, ky, kz, dil;
    int ntx, nty, ntz;       // voxel tiles per volume
    int ncic, ncoc;          // channel chunks
    const float *x1;         // optional second source [B, X/ux, Y/uy, Z/uz, c1], nearest up-sampled and concatenated after the c0
    int c0, c1, ux, uy, uz;  // channels of x (the layer saw concat(x, upsample(x1)), Cin = c0 + c1); c0 % 4 == 0 when x1 is given
    int im2col;              // single-channel input, 3x3x3 kernel: the 27 taps are staged as 27 "channels" of a 1x1x1 conv
                             // (Cin = 27, kx = ky = kz = 1 above; dW[tap][0][co] and dW[0][tap][co] are the same memory)
    int dps;                 // channels per voxel row of dp (Cout, or 8 * Cout in the folded form)
    int fold;                // folded decoder backward: dp is the space-to-depth tensor of the fine gradient (8 parity groups of Cout
                             // channels), blockIdx.z = parity group P, its 8 taps are e = p + t per axis (t in {0,1}), and
                             // dw is [8 groups][8 taps][Cin][Cout]
};

constexpr int WT_X = 4, WT_Y = 4, WT_Z = 8;     // voxel tile: 128 voxels = 32 k-steps of 4
// WG_MAXT (template): taps per wave = ceil(27 / 4) = 7, or 2 for the 8 taps of the folded form

// NA, NB: 16-channel blocks of the cin / cout chunk handled by one block
// K3: 3x3x3 kernel, dilation 1 -- the halo geometry is a compile-time constant, which turns the staging loop's row
// decode (two runtime divisions per 16-byte load) into multiply-shifts
// KM = 1: 1x1x1 kernel (also the im2col form of the first layer) -- same staging; the four waves split the 32 k-steps
// instead of the taps and each adds its partial sums at the end
template <int NA, int NB, int KM, int WG_MAXT = 7>
__global__ __launch_bounds__(256) void conv3d_wgrad(WgArgs a) {
    constexpr bool K3 = KM == 3, K1 = KM == 1, KF = K3 || K1;
    constexpr int HALO = K3 ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CC = 16 * NA, CO = 16 * NB;
    constexpr int RSA = (CC % 32 == 0) ? CC + 16 : CC;          // row strides: 16 mod 32 floats
    constexpr int RSB = (CO % 32 == 0) ? CO + 16 : CO;
    const int hx = KF ? HALO : (a.kx > 1 ? a.dil : 0), hy = KF ? HALO : (a.ky > 1 ? a.dil : 0), hz = KF ? HALO : (a.kz > 1 ? a.dil : 0);
    const int HX = WT_X + 2 * hx, HY = WT_Y + 2 * hy, HZ = WT_Z + 2 * hz;
    const int nrowsA = HX * HY * HZ;
    float *la = lds;                                            // [nrowsA][RSA]
    float *lb = lds + nrowsA * RSA;                             // [128][RSB]
    const bool FOLD = WG_MAXT == 2;
    const int ntap = FOLD ? 8 : K3 ? 27 : K1 ? 1 : a.kx * a.ky * a.kz;
    const int P = FOLD ? blockIdx.z : 0;                        // parity group: its channels of dp and its taps
    const int cic = blockIdx.y % a.ncic, coc = blockIdx.y / a.ncic;
    const int ci0 = cic * CC, co0 = coc * CO;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    f32x4 acc[WG_MAXT][NA][NB];
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
        for (int na = 0; na < NA; ++na)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[i][na][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;

    // tap offsets (in LDS rows) of this wave's taps
    int toff[WG_MAXT];
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
        const int t = min(wv + 4 * i, ntap - 1);
        const int kz = K3 ? 3 : K1 ? 1 : a.kz, ky = K3 ? 3 : K1 ? 1 : a.ky, dil = KF ? 1 : a.dil;
        int dz = t % kz, dy = (t / kz) % ky, dx = t / (kz * ky);
        if (FOLD) { dx = ((P >> 2) & 1) + ((t >> 2) & 1); dy = ((P >> 1) & 1) + ((t >> 1) & 1); dz = (P & 1) + (t & 1); }
        toff[i] = ((dx * dil) * HY + dy * dil) * HZ + dz * dil;
    }

    const long long tiles_per_vol = (long long)a.ntx * a.nty * a.ntz;
    const long long ntiles = tiles_per_vol * a.B;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_vol);
        const long long tv = tile % tiles_per_vol;
        const int x0 = (int)(tv / ((long long)a.nty * a.ntz)) * WT_X, y0 = (int)((tv / a.ntz) % a.nty) * WT_Y,
                  z0 = (int)(tv % a.ntz) * WT_Z;
        const float *xb = a.x + (long long)b * a.X * a.Y * a.Z * (a.x1 ? a.c0 : a.Cin);
        const float *pb = a.dp + (long long)b * a.X * a.Y * a.Z * a.dps + P * a.Cout;
        __syncthreads();                                        // previous tile fully consumed
        // ---- stage the x halo tile (zero outside the volume = SAME padding, zero beyond Cin) ---------------
        if (KF) {
            // channel counts are multiples of 4 here (launch_wgrad): every element is one unconditional 16-byte load from
            // a clamped address, six (two for dpre) in flight per thread, zeroed by select afterwards
            constexpr int QA = CC / 4, TOTA = (WT_X + 2 * HALO) * (WT_Y + 2 * HALO) * (WT_Z + 2 * HALO) * QA, UB = 6;
            const int cs = a.x1 ? a.c0 : a.Cin;
            auto stage_x = [&](auto has_x1) {
                constexpr bool X1SRC = decltype(has_x1)::value;
                // up-sampling factors are powers of two on this path (launch_wgrad)
                const int shx = X1SRC ? __ffs(a.ux) - 1 : 0, shy = X1SRC ? __ffs(a.uy) - 1 : 0, shz = X1SRC ? __ffs(a.uz) - 1 : 0;
                // element offsets fit 32 bits and the strides 24 bits on this path (launch_wgrad): full-rate multiplies
                const unsigned sZ = (unsigned)cs, sY = (unsigned)a.Z * sZ, sX = (unsigned)a.Y * sY;
                const unsigned Y1 = X1SRC ? a.Y >> shy : 1, Z1 = X1SRC ? a.Z >> shz : 1;
                const unsigned tZ = X1SRC ? (unsigned)a.c1 : 0, tY = Z1 * tZ, tX = Y1 * tY;
                const float *xlo = X1SRC ? a.x1 + (long long)b * (a.X >> shx) * tX : nullptr;
                for (int e0 = threadIdx.x; e0 < TOTA; e0 += 256 * UB) {
                    nrt_f4 v[UB];
                    bool ok[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int e = e0 + 256 * u, ee = e < TOTA ? e : 0;
                        const int r = ee / QA, c4 = (ee % QA) * 4;
                        const int rz = r % HZ, ry = (r / HZ) % HY, rx = r / (HZ * HY);
                        const int gx = x0 + rx - HALO, gy = y0 + ry - HALO, gz = z0 + rz - HALO, ch = ci0 + c4;
                        ok[u] = (e < TOTA) & (gx >= 0) & (gx < a.X) & (gy >= 0) & (gy < a.Y) & (gz >= 0) & (gz < a.Z) & (ch < a.Cin);
                        unsigned off = __umul24((unsigned)gx, sX) + __umul24((unsigned)gy, sY) + __umul24((unsigned)gz, sZ) + (unsigned)ch;
                        const float *base = xb;
                        if (X1SRC) {
                            const unsigned off1 = __umul24((unsigned)(gx >> shx), tX) + __umul24((unsigned)(gy >> shy), tY) +
                                                  __umul24((unsigned)(gz >> shz), tZ) + (unsigned)(ch - a.c0);
                            const bool lo = ch >= a.c0;
                            off = lo ? off1 : off;
                            base = lo ? xlo : xb;
                        }
                        off = ok[u] ? off : 0u;
                        v[u] = *(const nrt_f4 *)(base + off);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int e = e0 + 256 * u;
                        if (e < TOTA) {
                            const int r = e / QA, c4 = (e % QA) * 4;
                            *(nrt_f4 *)(la + r * RSA + c4) = ok[u] ? v[u] : (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
                        }
                    }
                }
            };
            if (K1 && a.im2col) {
                // first layer: the 27 taps of the single-channel input are the "channels" of this tile's rows; eight
                // clamped scalar loads in flight per thread
                constexpr int TOTI = 128 * CC, UI = 8;
                const float *x1 = a.x + (long long)b * a.X * a.Y * a.Z;
                const unsigned sY = (unsigned)a.Z, sX = (unsigned)a.Y * sY;
                for (int e0 = threadIdx.x; e0 < TOTI; e0 += 256 * UI) {
                    float v[UI];
                    bool ok[UI];
#pragma unroll
                    for (int u = 0; u < UI; ++u) {
                        const int e = e0 + 256 * u, ee = e < TOTI ? e : 0;
                        const int r = ee / CC, t = ci0 + ee % CC;
                        const int rz = r % WT_Z, ry = (r / WT_Z) % WT_Y, rx = r / (WT_Z * WT_Y);
                        const int gx = x0 + rx + (t / 9 - 1) * a.dil, gy = y0 + ry + ((t / 3) % 3 - 1) * a.dil, gz = z0 + rz + (t % 3 - 1) * a.dil;
                        ok[u] = (e < TOTI) & (t < 27) & (gx >= 0) & (gx < a.X) & (gy >= 0) & (gy < a.Y) & (gz >= 0) & (gz < a.Z);
                        const unsigned off = __umul24((unsigned)gx, sX) + __umul24((unsigned)gy, sY) + (unsigned)gz;
                        v[u] = x1[ok[u] ? off : 0u];
                    }
#pragma unroll
                    for (int u = 0; u < UI; ++u) {
                        const int e = e0 + 256 * u;
                        if (e < TOTI) la[(e / CC) * RSA + e % CC] = ok[u] ? v[u] : 0.0f;
                    }
                }
            } else if (a.x1) stage_x(std::true_type{});
            else stage_x(std::false_type{});
            constexpr int QB = CO / 4, TOTB = 128 * QB, UBB = (TOTB + 255) / 256;
            nrt_f4 vb[UBB];
            bool okb[UBB];
#pragma unroll
            for (int u = 0; u < UBB; ++u) {
                const int e = threadIdx.x + 256 * u, ee = e < TOTB ? e : 0;
                const int r = ee / QB, c4 = (ee % QB) * 4;
                const int rz = r % WT_Z, ry = (r / WT_Z) % WT_Y, rx = r / (WT_Z * WT_Y);
                const int gx = x0 + rx, gy = y0 + ry, gz = z0 + rz;
                okb[u] = e < TOTB && gx < a.X && gy < a.Y && gz < a.Z && co0 + c4 < a.Cout;
                const float *src = okb[u] ? pb + (((long long)gx * a.Y + gy) * a.Z + gz) * a.dps + co0 + c4 : a.dp;
                vb[u] = *(const nrt_f4 *)src;
            }
#pragma unroll
            for (int u = 0; u < UBB; ++u) {
                const int e = threadIdx.x + 256 * u;
                if (e < TOTB) *(nrt_f4 *)(lb + (e / QB) * RSB + (e % QB) * 4) = okb[u] ? vb[u] : (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
            }
        } else {
        if (a.im2col) {
            const float *x1 = a.x + (long long)b * a.X * a.Y * a.Z;
            for (int e = threadIdx.x; e < 128 * CC; e += 256) {
                const int r = e / CC, tl = e % CC, t = ci0 + tl;
                const int rz = r % WT_Z, ry = (r / WT_Z) % WT_Y, rx = r / (WT_Z * WT_Y);
                const int gx = x0 + rx + (t / 9 - 1) * a.dil, gy = y0 + ry + ((t / 3) % 3 - 1) * a.dil, gz = z0 + rz + (t % 3 - 1) * a.dil;
                float v = 0.0f;
                if (t < 27 && gx >= 0 && gx < a.X && gy >= 0 && gy < a.Y && gz >= 0 && gz < a.Z)
                    v = x1[((long long)gx * a.Y + gy) * a.Z + gz];
                la[r * RSA + tl] = v;
            }
        } else
        for (int e = threadIdx.x; e < nrowsA * (CC / 4); e += 256) {
            const int r = e / (CC / 4), c4 = (e % (CC / 4)) * 4;
            const int rz = r % HZ, ry = (r / HZ) % HY, rx = r / (HZ * HY);
            const int gx = x0 + rx - hx, gy = y0 + ry - hy, gz = z0 + rz - hz;
            nrt_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (gx >= 0 && gx < a.X && gy >= 0 && gy < a.Y && gz >= 0 && gz < a.Z) {
                const int ch = ci0 + c4;
                if (a.x1 && ch >= a.c0) {
                    // fused UpSampling3D + concatenate loader: the channel quad comes from the low-resolution tensor
                    const int X1 = a.X / a.ux, Y1 = a.Y / a.uy, Z1 = a.Z / a.uz;
                    const float *src = a.x1 + ((((long long)b * X1 + gx / a.ux) * Y1 + gy / a.uy) * Z1 + gz / a.uz) * a.c1 + (ch - a.c0);
                    if (ch + 3 < a.Cin && (a.c1 & 3) == 0) v = *(const nrt_f4 *)src;
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (ch + k < a.Cin) v[k] = src[k];
                    }
                } else {
                    const int cs = a.x1 ? a.c0 : a.Cin;                // channel stride of the first source
                    const float *src = xb + (((long long)gx * a.Y + gy) * a.Z + gz) * cs + ch;
                    if (ch + 3 < cs && (cs & 3) == 0) v = *(const nrt_f4 *)src;
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (ch + k < cs) v[k] = src[k];
                    }
                }
            }
            *(nrt_f4 *)(la + r * RSA + c4) = v;
        }
        // ---- stage the dpre tile (zero outside the volume and beyond Cout) ------------------------------------
        for (int e = threadIdx.x; e < 128 * (CO / 4); e += 256) {
            const int r = e / (CO / 4), c4 = (e % (CO / 4)) * 4;
            const int rz = r % WT_Z, ry = (r / WT_Z) % WT_Y, rx = r / (WT_Z * WT_Y);
            const int gx = x0 + rx, gy = y0 + ry, gz = z0 + rz;
            nrt_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (gx < a.X && gy < a.Y && gz < a.Z) {
                const float *src = pb + (((long long)gx * a.Y + gy) * a.Z + gz) * a.dps + co0 + c4;
                if (co0 + c4 + 3 < a.Cout && (a.Cout & 3) == 0) v = *(const nrt_f4 *)src;
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (co0 + c4 + k < a.Cout) v[k] = src[k];
                }
            }
            *(nrt_f4 *)(lb + r * RSB + c4) = v;
        }
        }
        __syncthreads();
        // ---- bias gradient: column sums of the dpre tile (cin chunk 0 only) --------------------------------
        if (a.db && cic == 0 && threadIdx.x < CO) {
            float s = 0.0f;
            for (int r = 0; r < 128; ++r) s += lb[r * RSB + threadIdx.x];
            bsum += s;
        }
        // ---- 32 k-steps of 4 consecutive-z voxels --------------------------------------------------------------
        if (K1) {
            // one tap: wave w takes k-steps w, w + 4, ... (tile row 4 ks + lane / 16), all reads first, then the MFMAs
            const float *pa0 = la + (4 * wv + l4) * RSA + l15, *pb0 = lb + (4 * wv + l4) * RSB + l15;
            float af[8][NA], bf[8][NB];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int na = 0; na < NA; ++na) af[j][na] = pa0[j * 16 * RSA + na * 16];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bf[j][nb] = pb0[j * 16 * RSB + nb * 16];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int na = 0; na < NA; ++na)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[0][na][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][na], bf[j][nb], acc[0][na][nb], 0, 0, 0);
        } else if (K3) {
            // fully unrolled and double-buffered: every LDS address is a per-tap base register plus an immediate, the
            // operands of step ks + 1 are read while the matrix pipe works on step ks, no tap conditionals (wave 3's
            // seventh tap repeats tap 26 and is dropped at the end)
            const float *pbl = lb + l4 * RSB + l15;
            const float *pal[WG_MAXT];
#pragma unroll
            for (int i = 0; i < WG_MAXT; ++i) pal[i] = la + (toff[i] + l4) * RSA + l15;
            auto frag = [&](int ks, float (&af)[WG_MAXT][NA], float (&bf)[NB]) {
                const int zh = ks & 1, yy = (ks >> 1) & 3, xx = ks >> 3;
                const int vrow = (xx * WT_Y + yy) * WT_Z + zh * 4, arow = (xx * HY + yy) * HZ + zh * 4;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bf[nb] = pbl[vrow * RSB + nb * 16];
#pragma unroll
                for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
                    for (int na = 0; na < NA; ++na) af[i][na] = pal[i][arow * RSA + na * 16];
            };
            auto mma = [&](const float (&af)[WG_MAXT][NA], const float (&bf)[NB]) {
#pragma unroll
                for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
                    for (int na = 0; na < NA; ++na)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[i][na][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][na], bf[nb], acc[i][na][nb], 0, 0, 0);
            };
            float afA[WG_MAXT][NA], afB[WG_MAXT][NA], bfA[NB], bfB[NB];
            frag(0, afA, bfA);
#pragma unroll
            for (int ks = 0; ks < 32; ks += 2) {
                __builtin_amdgcn_sched_barrier(0);          // keep the reads of the next step ahead of this step's MFMAs
                frag(ks + 1, afB, bfB);
                __builtin_amdgcn_sched_barrier(0);
                mma(afA, bfA);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 32) frag(ks + 2, afA, bfA);
                __builtin_amdgcn_sched_barrier(0);
                mma(afB, bfB);
            }
        } else
        for (int ks = 0; ks < 32; ++ks) {
            const int zh = ks & 1, yy = (ks >> 1) & 3, xx = ks >> 3;
            const int vrow = (xx * WT_Y + yy) * WT_Z + zh * 4 + l4;                       // row in the dpre tile
            const int arow = (xx * HY + yy) * HZ + zh * 4 + l4;                           // row in the halo tile, tap (0,0,0)
            float bf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = lb[vrow * RSB + nb * 16 + l15];
#pragma unroll
            for (int i = 0; i < WG_MAXT; ++i) {
                if (wv + 4 * i < ntap) {
#pragma unroll
                    for (int na = 0; na < NA; ++na) {
                        const float af = la[(arow + toff[i]) * RSA + na * 16 + l15];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[i][na][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[nb], acc[i][na][nb], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- accumulate into global memory: D row = 4 (lane >> 4) + r (ci), col = lane & 15 (co) -----------------------
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
        const int t = K1 ? 0 : wv + 4 * i;
        if (K1 ? i == 0 : t < ntap) {
#pragma unroll
            for (int na = 0; na < NA; ++na)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co = co0 + nb * 16 + l15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = ci0 + na * 16 + l4 * 4 + r;
                        if (ci < a.Cin && co < a.Cout)
                            unsafeAtomicAdd(&a.dw[((long long)(P * 8 + t) * a.Cin + ci) * a.Cout + co], acc[i][na][nb][r]);
                    }
                }
        }
    }
    if (a.db && cic == 0 && threadIdx.x < CO && co0 + (int)threadIdx.x < a.Cout) unsafeAtomicAdd(&a.db[co0 + threadIdx.x], bsum);
}

// Folded weight gradient of a decoder convolution, all 8 parity groups per block (see nrt_conv3d_wgrad_s2d_f32):
//   dWf[P][t][ci][co] = sum_q lo[q + p + t - 1][ci] * dY'[q][P * Cout + co]
// x = lo [B, X, Y, Z, Cin] on the low-resolution grid, dp = the space-to-depth gradient [B, X, Y, Z, 8 * Cout].  One block stages the
// halo tile of lo ONCE and the dY' tile of all 8 groups (128 rows x 8 * 16 channels), wave w owns the taps t = (i, w >> 1, w & 1),
// i in {0, 1}, of every group: 16 accumulator sets of NA x 1 tiles.  (The grid.z-per-group form of the first version staged the halo
// tile 8 times per voxel tile and was bound by that traffic.)
template <int NA>
__global__ __launch_bounds__(256) void conv3d_wgrad_fold(WgArgs a) {
    constexpr int CC = 16 * NA, CO = 16, COB = 8 * CO;
    constexpr int RSA = (CC % 32 == 0) ? CC + 16 : CC, RSB = COB + 16;       // row strides: 16 mod 32 floats
    constexpr int HX = WT_X + 2, HY = WT_Y + 2, HZ = WT_Z + 2, NROWA = HX * HY * HZ;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *la = lds;                                            // [NROWA][RSA]
    float *lb = lds + NROWA * RSA;                              // [128][RSB]: column block P = parity group
    const int cic = blockIdx.y % a.ncic, coc = blockIdx.y / a.ncic;
    const int ci0 = cic * CC, co0 = coc * CO;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    f32x4 acc[8][2][NA];
#pragma unroll
    for (int P = 0; P < 8; ++P)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int na = 0; na < NA; ++na) acc[P][i][na] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // tap (tx, ty, tz) = (i, wv >> 1, wv & 1) of group P = (px, py, pz) reads halo row offset (px + tx, py + ty, pz + tz)
    const float *pal[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pal[i] = la + (((i * HY + (wv >> 1)) * HZ + (wv & 1)) + l4) * RSA + l15;
    const float *pbl = lb + l4 * RSB + l15;

    const long long tiles_per_vol = (long long)a.ntx * a.nty * a.ntz;
    const long long ntiles = tiles_per_vol * a.B;
    const unsigned sZ = (unsigned)a.Cin, sY = (unsigned)a.Z * sZ, sX = (unsigned)a.Y * sY;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_vol);
        const long long tv = tile % tiles_per_vol;
        const int x0 = (int)(tv / ((long long)a.nty * a.ntz)) * WT_X, y0 = (int)((tv / a.ntz) % a.nty) * WT_Y, z0 = (int)(tv % a.ntz) * WT_Z;
        const float *xb = a.x + (long long)b * a.X * a.Y * a.Z * a.Cin;
        const float *pb = a.dp + (long long)b * a.X * a.Y * a.Z * a.dps;
        __syncthreads();                                        // previous tile fully consumed
        {   // halo tile of lo: unconditional 16-byte loads from clamped addresses, six in flight per thread, zeroed by select
            constexpr int QA = CC / 4, TOTA = NROWA * QA, UB = 6;
            for (int e0 = threadIdx.x; e0 < TOTA; e0 += 256 * UB) {
                nrt_f4 v[UB];
                bool ok[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = e0 + 256 * u, ee = e < TOTA ? e : 0;
                    const int r = ee / QA, c4 = (ee % QA) * 4;
                    const int rz = r % HZ, ry = (r / HZ) % HY, rx = r / (HZ * HY);
                    const int gx = x0 + rx - 1, gy = y0 + ry - 1, gz = z0 + rz - 1, ch = ci0 + c4;
                    ok[u] = (e < TOTA) & (gx >= 0) & (gx < a.X) & (gy >= 0) & (gy < a.Y) & (gz >= 0) & (gz < a.Z) & (ch < a.Cin);
                    const unsigned off = __umul24((unsigned)gx, sX) + __umul24((unsigned)gy, sY) + __umul24((unsigned)gz, sZ) + (unsigned)ch;
                    v[u] = *(const nrt_f4 *)(xb + (ok[u] ? off : 0u));
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = e0 + 256 * u;
                    if (e < TOTA) *(nrt_f4 *)(la + (e / QA) * RSA + (e % QA) * 4) = ok[u] ? v[u] : (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
        {   // dY' tile: 128 voxels x 8 groups x 16 channels (this block's cout chunk of every group)
            constexpr int QB = COB / 4, TOTB = 128 * QB, UBB = 8;
            for (int e0 = threadIdx.x; e0 < TOTB; e0 += 256 * UBB) {
                nrt_f4 vb[UBB];
                bool okb[UBB];
#pragma unroll
                for (int u = 0; u < UBB; ++u) {
                    const int e = e0 + 256 * u;
                    const int r = e / QB, q = e % QB, P = q / (CO / 4), c4 = (q % (CO / 4)) * 4;
                    const int rz = r % WT_Z, ry = (r / WT_Z) % WT_Y, rx = r / (WT_Z * WT_Y);
                    const int gx = x0 + rx, gy = y0 + ry, gz = z0 + rz;
                    okb[u] = gx < a.X && gy < a.Y && gz < a.Z && co0 + c4 < a.Cout;
                    const float *src = okb[u] ? pb + (((long long)gx * a.Y + gy) * a.Z + gz) * a.dps + P * a.Cout + co0 + c4 : a.dp;
                    vb[u] = *(const nrt_f4 *)src;
                }
#pragma unroll
                for (int u = 0; u < UBB; ++u) {
                    const int e = e0 + 256 * u;
                    *(nrt_f4 *)(lb + (e / QB) * RSB + (e % QB) * 4) = okb[u] ? vb[u] : (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
        __syncthreads();
        // ---- 32 k-steps of 4 consecutive-z voxels; the operands of step ks + 1 are read while the matrix pipe works on ks ----------
        auto frag = [&](int ks, float (&af)[8][2][NA], float (&bf)[8]) __attribute__((always_inline)) {
            const int zh = ks & 1, yy = (ks >> 1) & 3, xx = ks >> 3;
            const int vrow = (xx * WT_Y + yy) * WT_Z + zh * 4, arow = (xx * HY + yy) * HZ + zh * 4;
#pragma unroll
            for (int P = 0; P < 8; ++P) {
                bf[P] = pbl[vrow * RSB + P * CO];
                const int poff = (((P >> 2) & 1) * HY + ((P >> 1) & 1)) * HZ + (P & 1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int na = 0; na < NA; ++na) af[P][i][na] = pal[i][(arow + poff) * RSA + na * 16];
            }
        };
        auto mma = [&](const float (&af)[8][2][NA], const float (&bf)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int P = 0; P < 8; ++P)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int na = 0; na < NA; ++na)
                        acc[P][i][na] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[P][i][na], bf[P], acc[P][i][na], 0, 0, 0);
        };
        float afA[8][2][NA], afB[8][2][NA], bfA[8], bfB[8];
        frag(0, afA, bfA);
#pragma unroll
        for (int ks = 0; ks < 32; ks += 2) {
            __builtin_amdgcn_sched_barrier(0);
            frag(ks + 1, afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
            mma(afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < 32) frag(ks + 2, afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            mma(afB, bfB);
        }
    }
    // ---- accumulate: D row = 4 (lane >> 4) + r (ci), col = lane & 15 (co); tap index t = 4 i + wv ----------------------------------
#pragma unroll
    for (int P = 0; P < 8; ++P)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int na = 0; na < NA; ++na) {
                const int co = co0 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + na * 16 + l4 * 4 + r;
                    if (ci < a.Cin && co < a.Cout)
                        unsafeAtomicAdd(&a.dw[((long long)(P * 8 + 4 * i + wv) * a.Cin + ci) * a.Cout + co], acc[P][i][na][r]);
                }
            }
}

template <int NA>
int launch_wgrad_fold(WgArgs &a, hipStream_t st) {
    constexpr int CC = 16 * NA, RSA = (CC % 32 == 0) ? CC + 16 : CC, RSB = 8 * 16 + 16;
    const size_t lds = ((size_t)(WT_X + 2) * (WT_Y + 2) * (WT_Z + 2) * RSA + 128 * RSB) * sizeof(float);
    a.ncic = (a.Cin + CC - 1) / CC;
    a.ncoc = (a.Cout + 15) / 16;
    const long long ntiles = (long long)a.ntx * a.nty * a.ntz * a.B;
    long long bx = 256ll / (a.ncic * a.ncoc);                   // one block per CU (its LDS), about one resident wave of blocks
    if (bx < 32) bx = 32;
    if (bx > ntiles) bx = ntiles;
    if (hipFuncSetAttribute((const void *)conv3d_wgrad_fold<NA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return NRT_ERR_LAUNCH;
    hipLaunchKernelGGL((conv3d_wgrad_fold<NA>), dim3((unsigned)bx, a.ncic * a.ncoc), dim3(256), lds, st, a);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// 1x1x1 weight gradient with 16 input channels (the likelihood layer of the unets): D[ci][co] = sum_v x[v][ci] dz[v][co],
// db[co] = sum_v dz[v][co].  No LDS tiles: the MFMA operand layouts ARE rows of the channels-last tensors -- A[m = ci][k = voxel]
// is x_flat[16 (v0 + k) + m] (64 consecutive floats per wave instruction), B[k][n] = dz[(v0 + k) Cout + 16 nb + n] -- so a wave
// streams its own contiguous voxel range with UNR k-steps of loads in flight; the bias gradient is one more MFMA per N-tile with an
// all-ones A operand.  The four waves of a block meet in LDS, one float atomic per weight and block.  (The tile kernel above stages
// 24 KB per 128 voxels for 64 MFMAs behind two barriers: 0.40 ms for 786 MB.)
template <int NB>
__global__ __launch_bounds__(256) void conv1x1_wgrad16(const float *__restrict__ x, const float *__restrict__ dz, float *__restrict__ dw,
                                                       float *__restrict__ db, long long nvox, int Cout) {
    constexpr int UNR = 8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, k = lane >> 4;
    f32x4 acc[NB], accb[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { acc[nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; accb[nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
    long long vbeg, vend;                                       // this block's contiguous voxel range, a quarter per wave
    nrt_block_range(nvox, 4 * 4 * UNR, vbeg, vend);
    const long long per = (vend - vbeg + 3) / 4, span = ((per + 4 * UNR - 1) / (4 * UNR)) * (4 * UNR);
    const long long wbeg = vbeg + wv * span, wend = wbeg + span < vend ? wbeg + span : vend;
    for (long long v0 = wbeg; v0 < wend; v0 += 4 * UNR) {
        float av[UNR], bvv[UNR][NB];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long v = v0 + 4 * u + k;
            const bool ok = v < wend;
            const long long vv = ok ? v : vbeg;
            av[u] = x[vv * 16 + m];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bvv[u][nb] = dz[vv * Cout + nb * 16 + m];
            if (!ok) {
                av[u] = 0.0f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bvv[u][nb] = 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bvv[u][nb], acc[nb], 0, 0, 0);
                if (db) accb[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bvv[u][nb], accb[nb], 0, 0, 0);
            }
    }
    // ---- block: the four waves' partial D tiles (row = 4 (lane >> 4) + r = ci, column = lane & 15 = co) -------------------------
    __shared__ float red[4][NB][5][64];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][nb][r][lane] = acc[nb][r];
        red[wv][nb][4][lane] = accb[nb][0];                     // row 0 of the ones-product (lanes 0..15) = column sums
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NB * 5 * 64; e += 256) {
        const int nb = e / 320, r = (e / 64) % 5, l = e & 63;
        const float s = (red[0][nb][r][l] + red[1][nb][r][l]) + (red[2][nb][r][l] + red[3][nb][r][l]);
        const int co = nb * 16 + (l & 15);
        if (co >= Cout) continue;
        if (r < 4) unsafeAtomicAdd(&dw[(long long)(4 * (l >> 4) + r) * Cout + co], s);
        else if (db && l < 16) unsafeAtomicAdd(&db[co], s);
    }
}

// Weight gradient of the single-channel first layer (3x3x3, dilation 1, SAME): D[tap][co] = sum_v x[v + off(tap)] dz[v][co],
// db[co] = sum_v dz[v][co].  Same idea as conv1x1_wgrad16: no LDS tiles -- A[m = tap][k = voxel] is a gather from the 16 MB
// single-channel volume (its 9 neighbouring z-rows sit in L1 / L2), B[k][n] = dz rows; a wave walks whole z-rows of its contiguous
// range of (b, x, y) rows, 4 voxels per k-step, UNR k-steps of loads in flight.  (The im2col tile kernel took 0.41 ms for 278 MB.)
template <int NB>
__global__ __launch_bounds__(256) void conv3d_c1_wgrad(const float *__restrict__ x, const float *__restrict__ dz, float *__restrict__ dw,
                                                       float *__restrict__ db, int B, int X, int Y, int Z, int Cout) {
    constexpr int UNR = 8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, k = lane >> 4;
    f32x4 acc[2][NB], accb[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        acc[0][nb] = acc[1][nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        accb[nb] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    // this lane's two taps (M-tile 0: taps 0..15, M-tile 1: taps 16..26 and 5 dead rows)
    int tdx[2], tdy[2], tdz[2];
    bool tlive[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int t = mt * 16 + m;
        tlive[mt] = t < 27;
        const int tt = tlive[mt] ? t : 13;
        tdx[mt] = tt / 9 - 1; tdy[mt] = (tt / 3) % 3 - 1; tdz[mt] = tt % 3 - 1;
    }
    long long rbeg, rend;                                       // contiguous range of (b, x, y) rows per block, a quarter per wave
    const long long nrows = (long long)B * X * Y;
    nrt_block_range(nrows, 4, rbeg, rend);
    const long long per = (rend - rbeg + 3) / 4;
    const long long wbeg = rbeg + wv * per, wend = wbeg + per < rend ? wbeg + per : rend;
    for (long long row = wbeg; row < wend; ++row) {
        const unsigned r32 = (unsigned)row, q32 = r32 / (unsigned)Y;       // rows fit 32 bits (launcher)
        const int y = (int)(r32 - q32 * (unsigned)Y), xx = (int)(q32 % (unsigned)X);
        const long long bX = (long long)(q32 / (unsigned)X) * X;  // b * X
        const float *ap[2];
        bool aok[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int gx = xx + tdx[mt], gy = y + tdy[mt];
            aok[mt] = tlive[mt] && gx >= 0 && gx < X && gy >= 0 && gy < Y;
            ap[mt] = x + ((bX + (aok[mt] ? gx : xx)) * Y + (aok[mt] ? gy : y)) * Z + tdz[mt];
        }
        const float *bp = dz + row * Z * Cout + m;
        for (int z0 = 0; z0 < Z; z0 += 4 * UNR) {
            float av[UNR][2], bvv[UNR][NB];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int z = z0 + 4 * u + k;
                const bool zin = z < Z;
                const int zc = zin ? z : 0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int zz = zc + tdz[mt];
                    const bool ok = aok[mt] && zin && zz >= 0 && zz < Z;
                    const float v = ap[mt][ok ? zc : -tdz[mt]];     // (clamped to the row start when masked)
                    av[u][mt] = ok ? v : 0.0f;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v = bp[(long long)zc * Cout + nb * 16];
                    bvv[u][nb] = zin ? v : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bvv[u][nb], acc[0][nb], 0, 0, 0);
                    acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bvv[u][nb], acc[1][nb], 0, 0, 0);
                    if (db) accb[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bvv[u][nb], accb[nb], 0, 0, 0);
                }
        }
    }
    __shared__ float red[4][NB][9][64];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[wv][nb][r][lane] = acc[0][nb][r]; red[wv][nb][4 + r][lane] = acc[1][nb][r]; }
        red[wv][nb][8][lane] = accb[nb][0];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NB * 9 * 64; e += 256) {
        const int nb = e / 576, r = (e / 64) % 9, l = e & 63;
        const float s = (red[0][nb][r][l] + red[1][nb][r][l]) + (red[2][nb][r][l] + red[3][nb][r][l]);
        const int co = nb * 16 + (l & 15);
        if (r < 8) {
            const int t = (r >> 2) * 16 + 4 * (l >> 4) + (r & 3);
            if (t < 27) unsafeAtomicAdd(&dw[(long long)t * Cout + co], s);
        } else if (db && l < 16) unsafeAtomicAdd(&db[co], s);
    }
}

template <int NA, int NB>
int launch_wgrad(WgArgs &a, hipStream_t st) {
    constexpr int CC = 16 * NA, CO = 16 * NB;
    constexpr int RSA = (CC % 32 == 0) ? CC + 16 : CC, RSB = (CO % 32 == 0) ? CO + 16 : CO;
    const int hx = a.kx > 1 ? a.dil : 0, hy = a.ky > 1 ? a.dil : 0, hz = a.kz > 1 ? a.dil : 0;
    const size_t lds = ((size_t)(WT_X + 2 * hx) * (WT_Y + 2 * hy) * (WT_Z + 2 * hz) * RSA + 128 * RSB) * sizeof(float);
    if (lds > 160 * 1024) return NRT_ERR_UNSUPPORTED;
    a.ncic = (a.Cin + CC - 1) / CC;
    a.ncoc = (a.Cout + CO - 1) / CO;
    const long long ntiles = (long long)a.ntx * a.nty * a.ntz * a.B;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    long long bx = 256ll * per_cu / (a.ncic * a.ncoc);           // about one resident wave of blocks
    if (bx < 64) bx = 64;
    if (bx > ntiles) bx = ntiles;
    const bool quads = (a.Cout & 3) == 0 && (a.im2col || ((a.Cin & 3) == 0 &&
                       (!a.x1 || ((a.c0 & 3) == 0 && (a.c1 & 3) == 0 && (a.ux & (a.ux - 1)) == 0 && (a.uy & (a.uy - 1)) == 0 &&
                                  (a.uz & (a.uz - 1)) == 0))));
    const bool small = (long long)a.X * a.Y * a.Z * (a.im2col ? 1 : (a.x1 ? a.c0 : a.Cin)) < (1ll << 31) &&
                       (long long)a.Y * a.Z * (a.im2col ? 1 : a.Cin) < (1ll << 24);
    const int km = !(quads && small) ? 0 : (a.kx == 3 && a.ky == 3 && a.kz == 3 && a.dil == 1 && !a.im2col) ? 3
                   : (a.kx == 1 && a.ky == 1 && a.kz == 1) ? 1 : 0;
#define NRT_WG_LAUNCH(KM)                                                                                                   \
    do {                                                                                                                    \
        if (lds > 48 * 1024)                                                                                                \
            (void)hipFuncSetAttribute((const void *)conv3d_wgrad<NA, NB, KM>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      (int)lds);                                                                            \
        hipLaunchKernelGGL((conv3d_wgrad<NA, NB, KM>), dim3((unsigned)bx, a.ncic * a.ncoc), dim3(256), lds, st, a);         \
    } while (0)
    if (a.fold) {
        if (km != 3) return NRT_ERR_UNSUPPORTED;
        bx = 256ll * per_cu / (a.ncic * a.ncoc * 8);
        if (bx < 16) bx = 16;
        if (bx > ntiles) bx = ntiles;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)conv3d_wgrad<NA, NB, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((conv3d_wgrad<NA, NB, 3, 2>), dim3((unsigned)bx, a.ncic * a.ncoc, 8), dim3(256), lds, st, a);
    } else if (km == 3) NRT_WG_LAUNCH(3);
    else if (km == 1) NRT_WG_LAUNCH(1);
    else NRT_WG_LAUNCH(0);
#undef NRT_WG_LAUNCH
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNormalization in training mode (Keras, axis = -1): per-channel statistics over [rows, C]
//   channel_sums:   out[c] += sum_r a[r][c] * (b ? b[r][c] : 1)      (sum x, sum x^2 = a = b = x, sum g, sum g x)
//   channel_axpby:  y[r][c] = A[c] * a[r][c] + B[c] * b[r][c] + C0[c]    (the backward: dx = A g + B x + C0)
// float atomics across blocks; the caller zero-fills `out`
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sums(const float *__restrict__ a, const float *__restrict__ b, long long rows, int C,
                                                    float *__restrict__ out) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < C; i += 256) sm[i] = 0.0f;
    __syncthreads();
    const long long total = rows * C;
    if (256 % C == 0) {                    // the grid stride keeps e % C: a thread owns one channel
        float s = 0.0f;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
            s += b ? a[e] * b[e] : a[e];
        atomicAdd(&sm[threadIdx.x % C], s);
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
            atomicAdd(&sm[(int)(e % C)], b ? a[e] * b[e] : a[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) unsafeAtomicAdd(&out[i], sm[i]);
}

__global__ __launch_bounds__(256) void channel_axpby(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ A,
                                                     const float *__restrict__ B, const float *__restrict__ C0, float *__restrict__ y,
                                                     long long n, int C) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        y[e] = A[c] * a[e] + B[c] * b[e] + C0[c];
    }
}

unsigned ew_blocks(long long n, int per) {
    long long b = (n + per - 1) / per;
    if (b > 256ll * 16) b = 256ll * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int nrt_act_bwd_f32(const float *grad_out, const float *y, int activation, float *grad_pre, long long n,
                               void *stream) {
    if (!grad_out || !y || !grad_pre || n < 0) return NRT_ERR_INVALID_ARG;
    if (activation < ACT_NONE || activation > ACT_LAST) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    const bool al = ((((uintptr_t)grad_out | (uintptr_t)y | (uintptr_t)grad_pre) & 15) == 0);
    const long long n4 = al ? n / 4 : 0;
    if (n4) hipLaunchKernelGGL(act_bwd, dim3(ew_blocks(n4, 256)), dim3(256), 0, st, (const nrt_f4 *)grad_out, (const nrt_f4 *)y,
                               activation, (nrt_f4 *)grad_pre, n4);
    for (long long from = n4 * 4; from < n; from += 256)
        hipLaunchKernelGGL(act_bwd_tail, dim3(1), dim3(256), 0, st, grad_out, y, activation, grad_pre, from, n);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_maxpool3d_bwd_f32(const float *x, const float *grad_out, float *grad_x, int batch, const int *shape,
                                     int channels, const int *pool, int padding_same, void *stream) {
    if (!x || !grad_out || !grad_x || !shape || !pool || batch < 1 || batch > 65535 || channels < 1) return NRT_ERR_INVALID_ARG;
    const int X = shape[0], Y = shape[1], Z = shape[2];
    if (pool[0] < 1 || pool[1] < 1 || pool[2] < 1) return NRT_ERR_INVALID_ARG;
    const int OX = padding_same ? (X + pool[0] - 1) / pool[0] : X / pool[0];
    const int OY = padding_same ? (Y + pool[1] - 1) / pool[1] : Y / pool[1];
    const int OZ = padding_same ? (Z + pool[2] - 1) / pool[2] : Z / pool[2];
    hipStream_t st = nrt_stream(stream);
    if (!padding_same && (X % pool[0] || Y % pool[1] || Z % pool[2]))       // voxels outside every window get zero
        if (nrt_zero_async(grad_x, (size_t)batch * X * Y * Z * channels * sizeof(float), st) != hipSuccess) return NRT_ERR_LAUNCH;
    const long long total = (long long)OX * OY * OZ * channels;
    if (total == 0) return NRT_OK;
    hipLaunchKernelGGL(maxpool_bwd, dim3(ew_blocks(total, 256), batch), dim3(256), 0, st, x, grad_out, grad_x, X, Y, Z, channels,
                       OX, OY, OZ, pool[0], pool[1], pool[2]);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_upsample_sum_f32(const float *grad_up, int grad_channels, int channel_offset, float *grad_lo, int channels,
                                    int batch, const int *lo_shape, const int *up, void *stream) {
    if (!grad_up || !grad_lo || !lo_shape || !up || batch < 1 || batch > 65535 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (channel_offset < 0 || channel_offset + channels > grad_channels) return NRT_ERR_INVALID_ARG;
    if (up[0] < 1 || up[1] < 1 || up[2] < 1) return NRT_ERR_INVALID_ARG;
    const long long total = (long long)lo_shape[0] * lo_shape[1] * lo_shape[2] * channels;
    if (total == 0) return NRT_OK;
    hipLaunchKernelGGL(upsample_sum, dim3(ew_blocks(total, 256), batch), dim3(256), 0, nrt_stream(stream), grad_up, grad_channels,
                       channel_offset, grad_lo, lo_shape[0], lo_shape[1], lo_shape[2], channels, up[0], up[1], up[2]);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_softmax_bwd_f32(const float *y, const float *grad_out, float *grad_in, long long nvox, int channels,
                                   void *stream) {
    if (!y || !grad_out || !grad_in || nvox < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    const int G = channels / 4;
    const bool al = ((((uintptr_t)y | (uintptr_t)grad_out | (uintptr_t)grad_in) & 15) == 0);
    if (al && channels % 4 == 0 && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16)) {
        const unsigned vb = ew_blocks(nvox, 256 / G);
#define NRT_SMB(GG) hipLaunchKernelGGL((softmax_bwd_vec<GG>), dim3(vb), dim3(256), 0, st, (const nrt_f4 *)y, (const nrt_f4 *)grad_out, (nrt_f4 *)grad_in, nvox)
        switch (G) {
            case 1: NRT_SMB(1); break;
            case 2: NRT_SMB(2); break;
            case 4: NRT_SMB(4); break;
            case 8: NRT_SMB(8); break;
            default: NRT_SMB(16); break;
        }
#undef NRT_SMB
    } else {
        hipLaunchKernelGGL(softmax_bwd, dim3(ew_blocks(nvox, 256)), dim3(256), 0, st, y, grad_out, grad_in, nvox, channels);
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_conv3d_wgrad2_f32(const float *x, int c0, const float *x_lo, int c1, const int *up, const float *grad_pre,
                                     float *grad_weights, float *grad_bias, int batch, const int *shape, int cout,
                                     const int *ksize, int dilation, void *stream) {
    if (!x || !grad_pre || !grad_weights || !shape || !ksize) return NRT_ERR_INVALID_ARG;
    if (batch < 1 || c0 < 1 || c1 < 0 || cout < 1 || dilation < 1) return NRT_ERR_INVALID_ARG;
    if (c1 > 0 && (!x_lo || !up)) return NRT_ERR_INVALID_ARG;
    for (int d = 0; d < 3; ++d) {
        if (shape[d] < 1 || ksize[d] < 1) return NRT_ERR_INVALID_ARG;
        if (ksize[d] != 1 && ksize[d] != 3) return NRT_ERR_UNSUPPORTED;           // the kernel sizes of the conv stacks
        if (c1 > 0 && (up[d] < 1 || shape[d] % up[d])) return NRT_ERR_INVALID_ARG;
    }
    if (dilation > 2) return NRT_ERR_UNSUPPORTED;
    if (c1 > 0 && (c0 % 4)) return NRT_ERR_UNSUPPORTED;                           // a staged channel quad must not straddle the sources
    int cin = c0 + c1;
    WgArgs a;
    a.x = x; a.dp = grad_pre; a.dw = grad_weights; a.db = grad_bias;
    a.x1 = c1 > 0 ? x_lo : nullptr; a.c0 = c0; a.c1 = c1;
    a.ux = c1 > 0 ? up[0] : 1; a.uy = c1 > 0 ? up[1] : 1; a.uz = c1 > 0 ? up[2] : 1;
    a.B = batch; a.X = shape[0]; a.Y = shape[1]; a.Z = shape[2]; a.Cin = cin; a.Cout = cout;
    a.kx = ksize[0]; a.ky = ksize[1]; a.kz = ksize[2]; a.dil = dilation;
    a.ntx = (a.X + WT_X - 1) / WT_X; a.nty = (a.Y + WT_Y - 1) / WT_Y; a.ntz = (a.Z + WT_Z - 1) / WT_Z;
    a.im2col = 0;
    a.dps = cout; a.fold = 0;
    if (cin == 1 && ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3 && dilation == 1 && cout % 16 == 0 && cout <= 64 &&
        (long long)shape[0] * shape[1] * shape[2] < (1ll << 31)) {
        // the single-channel first layer: gathered straight from the volume (conv3d_c1_wgrad)
        const long long nrows = (long long)batch * shape[0] * shape[1];
        const unsigned blocks = nrows / 4 < 512 ? (unsigned)(nrows / 4 > 0 ? nrows / 4 : 1) : 512u;
        hipStream_t st1 = nrt_stream(stream);
        switch (cout / 16) {
            case 1: hipLaunchKernelGGL((conv3d_c1_wgrad<1>), dim3(blocks), dim3(256), 0, st1, x, grad_pre, grad_weights, grad_bias, batch, shape[0], shape[1], shape[2], cout); break;
            case 2: hipLaunchKernelGGL((conv3d_c1_wgrad<2>), dim3(blocks), dim3(256), 0, st1, x, grad_pre, grad_weights, grad_bias, batch, shape[0], shape[1], shape[2], cout); break;
            case 3: hipLaunchKernelGGL((conv3d_c1_wgrad<3>), dim3(blocks), dim3(256), 0, st1, x, grad_pre, grad_weights, grad_bias, batch, shape[0], shape[1], shape[2], cout); break;
            default: hipLaunchKernelGGL((conv3d_c1_wgrad<4>), dim3(blocks), dim3(256), 0, st1, x, grad_pre, grad_weights, grad_bias, batch, shape[0], shape[1], shape[2], cout); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    if (cin == 1 && ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3) {
        a.im2col = 1; a.Cin = 27; a.kx = a.ky = a.kz = 1;
        cin = 27;
    }
    hipStream_t st = nrt_stream(stream);
    if (ksize[0] == 1 && ksize[1] == 1 && ksize[2] == 1 && c1 == 0 && c0 == 16 && cout % 16 == 0 && cout <= 64 && !a.im2col) {
        // the likelihood layer: streamed straight from the channels-last rows (conv1x1_wgrad16)
        const long long nvox = (long long)batch * shape[0] * shape[1] * shape[2];
        const unsigned blocks = nvox / 128 < 512 ? (unsigned)(nvox / 128 > 0 ? nvox / 128 : 1) : 512u;
        switch ((cout + 15) / 16) {
            case 1: hipLaunchKernelGGL((conv1x1_wgrad16<1>), dim3(blocks), dim3(256), 0, st, x, grad_pre, grad_weights, grad_bias, nvox, cout); break;
            case 2: hipLaunchKernelGGL((conv1x1_wgrad16<2>), dim3(blocks), dim3(256), 0, st, x, grad_pre, grad_weights, grad_bias, nvox, cout); break;
            case 3: hipLaunchKernelGGL((conv1x1_wgrad16<3>), dim3(blocks), dim3(256), 0, st, x, grad_pre, grad_weights, grad_bias, nvox, cout); break;
            default: hipLaunchKernelGGL((conv1x1_wgrad16<4>), dim3(blocks), dim3(256), 0, st, x, grad_pre, grad_weights, grad_bias, nvox, cout); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    int na = cin <= 16 ? 1 : (cin <= 32 ? 2 : 3);
    const int nb = cout <= 16 ? 1 : 2;
    // dilated kernels have larger halo tiles: shrink the cin chunk until the tiles fit the LDS
    for (; na >= 1; --na) {
        int rc;
        if (na == 1) rc = nb == 1 ? launch_wgrad<1, 1>(a, st) : launch_wgrad<1, 2>(a, st);
        else if (na == 2) rc = nb == 1 ? launch_wgrad<2, 1>(a, st) : launch_wgrad<2, 2>(a, st);
        else rc = nb == 1 ? launch_wgrad<3, 1>(a, st) : launch_wgrad<3, 2>(a, st);
        if (rc != NRT_ERR_UNSUPPORTED) return rc;
    }
    return NRT_ERR_UNSUPPORTED;
}

extern "C" int nrt_conv3d_wgrad_f32(const float *x, const float *grad_pre, float *grad_weights, float *grad_bias, int batch,
                                    const int *shape, int cin, int cout, const int *ksize, int dilation, void *stream) {
    return nrt_conv3d_wgrad2_f32(x, cin, nullptr, 0, nullptr, grad_pre, grad_weights, grad_bias, batch, shape, cout, ksize, dilation,
                                 stream);
}

extern "C" int nrt_conv3d_wgrad_s2d_f32(const float *x_lo, const float *grad_pre_s2d, float *grad_folded, int batch, const int *shape,
                                        int cin, int group, void *stream) {
    if (!x_lo || !grad_pre_s2d || !grad_folded || !shape || batch < 1 || cin < 4 || cin % 4 || group < 4 || group % 4)
        return NRT_ERR_INVALID_ARG;
    for (int d = 0; d < 3; ++d)
        if (shape[d] < 1) return NRT_ERR_INVALID_ARG;
    if (group > 32) return NRT_ERR_UNSUPPORTED;                                   // one cout chunk per parity group
    WgArgs a;
    a.x = x_lo; a.dp = grad_pre_s2d; a.dw = grad_folded; a.db = nullptr;
    a.x1 = nullptr; a.c0 = cin; a.c1 = 0; a.ux = a.uy = a.uz = 1;
    a.B = batch; a.X = shape[0]; a.Y = shape[1]; a.Z = shape[2]; a.Cin = cin; a.Cout = group;
    a.kx = a.ky = a.kz = 3; a.dil = 1;
    a.ntx = (a.X + WT_X - 1) / WT_X; a.nty = (a.Y + WT_Y - 1) / WT_Y; a.ntz = (a.Z + WT_Z - 1) / WT_Z;
    a.im2col = 0;
    a.dps = 8 * group; a.fold = 1;
    hipStream_t st = nrt_stream(stream);
    // all 8 parity groups per block when the staging's assumptions hold (quad-aligned channels, 32-bit offsets, 24-bit strides)
    const bool foldall = cin % 4 == 0 && group % 4 == 0 && (long long)a.X * a.Y * a.Z * cin < (1ll << 31) &&
                         (long long)a.Y * a.Z * cin < (1ll << 24) && ((((uintptr_t)x_lo) | ((uintptr_t)grad_pre_s2d)) & 15) == 0;
    static int kfold = -1;
    if (kfold < 0) { const char *e = getenv("NRT_WGRAD_FOLDALL"); kfold = e ? atoi(e) : 1; }
    if (foldall && kfold) return cin <= 16 ? launch_wgrad_fold<1>(a, st) : launch_wgrad_fold<2>(a, st);
    const int na = cin <= 16 ? 1 : (cin <= 32 ? 2 : 3), nb = group <= 16 ? 1 : 2;
    if (na == 1) return nb == 1 ? launch_wgrad<1, 1>(a, st) : launch_wgrad<1, 2>(a, st);
    if (na == 2) return nb == 1 ? launch_wgrad<2, 1>(a, st) : launch_wgrad<2, 2>(a, st);
    return nb == 1 ? launch_wgrad<3, 1>(a, st) : launch_wgrad<3, 2>(a, st);
}

extern "C" int nrt_channel_sums_f32(const float *a, const float *b, long long rows, int channels, float *out, void *stream) {
    if (!a || !out || rows < 0 || channels < 1 || channels > 4096) return NRT_ERR_INVALID_ARG;
    if (rows == 0) return NRT_OK;
    long long bx = (rows * channels + 256 * 32 - 1) / (256 * 32);
    if (bx > 512) bx = 512;                 // every block ends with one atomic per channel: same-address atomics serialise in L2
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(channel_sums, dim3((unsigned)bx), dim3(256), (size_t)channels * sizeof(float), nrt_stream(stream), a, b, rows,
                       channels, out);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_channel_axpby_f32(const float *a, const float *b, const float *coef_a, const float *coef_b, const float *coef_c,
                                     float *y, long long n, int channels, void *stream) {
    if (!a || !b || !coef_a || !coef_b || !coef_c || !y || n < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    hipLaunchKernelGGL(channel_axpby, dim3(ew_blocks(n, 256)), dim3(256), 0, nrt_stream(stream), a, b, coef_a, coef_b, coef_c, y, n,
                       channels);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
