// Conv3D stack of neurite's unet for gfx950 (MI355X): the Keras layers models.py:1378-1388 (encoder
// convs), :1436-1438 (MaxPooling3D), :1531-1542 (UpSampling3D + concatenate), :1545-1555 (decoder
// convs), :1596 (1x1 "likelihood" conv) and :1601-1605 (channel softmax).
//
// Layout: channels-last fp32 exactly as Keras: x [B, X, Y, Z, Cin], kernel [kx, ky, kz, Cin, Cout],
// y [B, X, Y, Z, Cout]; cross-correlation, stride 1, SAME padding (floor((k-1)*dil/2) before).
//
// conv3d_mfma_f32  -- implicit GEMM on the matrix cores, M = output voxels, N = Cout, K = taps * Cin,
//   with v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD = the 157 TFLOP/s fp32 peak).
//   Block = 4 waves = a 4(x) x 4(y) x 16(z) output tile; wave w owns the x-slab w: 4 M-tiles (one per
//   y) of 16 consecutive-z voxels, all N-tiles.  K is walked in chunks of 16 input channels: the halo
//   tile [4+2p][4+2p][16+2p] x 16 ch of the chunk is staged in LDS (row stride 20 floats: the 16-lane
//   groups of ds_read_b128 then fall on distinct banks), each lane reads ONE float4 (4 consecutive
//   channels of one voxel) per tap and M-tile and feeds 4 MFMAs with it; weights are pre-packed in
//   fragment order so that a lane's 4 B-operands are one 16-byte load from L2.
//   UpSampling3D + concatenate are fused into the halo loader: channels >= c0 are read from a second
//   tensor at (x/ux, y/uy, z/uz), so the 786 MB concat tensor of the last decoder level never exists.
//   Epilogue: + bias, ELU as exp(x)-1 (TF semantics), 64-byte row segments per voxel.
// conv3d_direct_f32 -- any shape (Cin = 1 first layer, odd channel counts, VALID padding, big dilation):
//   one thread per output voxel, 16 output channels at a time in registers.  HBM-bound cases only.
// conv1x1_softmax_f32 -- the likelihood conv with the channel softmax fused (one pass, no logits tensor).
// maxpool3d_f32, upsample_concat_f32, softmax_lastdim_f32 -- the remaining Keras layers.

#include <type_traits>
#include <utility>

#include "nrt_common.h"
#include "activations.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float activate(float v, int act) { return nrt_activate_fused(v, act); }
// exp(d) for the channel softmax (d = x - max <= 0): one v_exp_f32 on the argument scaled by log2(e) instead of the ~17-instruction
// libm expansion -- the 1x1 + softmax head was bound by VALU issue (185 instructions per 8 voxels), not by its 786 MB.  Relative error
// <= 2^-23 + |d| 2^-24 (|d| < 40 for anything that survives the sum): below 2e-6, inside the 1e-5 of the layer tests
__device__ __forceinline__ float softmax_exp(float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896341f); }
__device__ __forceinline__ float softmax_rcp(float s) { return __builtin_amdgcn_rcpf(s); }          // s in [1, C]: 1 ulp
__device__ __forceinline__ float activate_ew(float v, int act) { return nrt_activate(v, act); }

struct ConvArgs {
    const float *src0;       // [B, X, Y, Z, c0]
    const float *src1;       // [B, X/ux, Y/uy, Z/uz, c1] or null
    const float *bias;       // [Cout] or null
    float *out;              // [B, OX, OY, OZ, Cout]
    int X, Y, Z;             // input (= src0) spatial shape
    int OX, OY, OZ;          // output spatial shape
    int c0, c1, Cout;
    int ux, uy, uz;          // up-sampling factors of src1
    int X1, Y1, Z1;          // src1 spatial shape
    int kx, ky, kz;
    int dil;
    int px, py, pz;          // padding before
    int act;
    int fold;                // > 0: input channels come in 8 parity groups of `fold` (space-to-depth of a 2x finer tensor) and a group
                             // only has the 2 x 2 x 2 taps (e = (p ? 1 : 2) - t per axis) of the folded decoder backward
};

// ============================================================================================
// MFMA implicit GEMM
// ============================================================================================
constexpr int CT_X = 4, CT_Y = 4, CT_Z = 16;   // output tile
constexpr int LDS_ROW = 20;                     // floats per staged voxel row (16 channels + 4 pad)

// FAST: 3x3x3 kernel, dilation 1 -- the tap loop is fully unrolled, every LDS address is base + immediate
template <int NT, bool FAST, bool FOLD = false>
__global__ __launch_bounds__(256) void conv3d_mfma(ConvArgs a, const float *__restrict__ wpacked, unsigned nblk,
                                                   unsigned nbx, unsigned nby, unsigned nbz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lb = nrt_xcd_block(blockIdx.x, gridDim.x);
    if (lb >= nblk) return;
    const int b = blockIdx.y;
    const int bz = lb % nbz, by = (lb / nbz) % nby, bx = lb / (nbz * nby);
    const int x0 = bx * CT_X, y0 = by * CT_Y, z0 = bz * CT_Z;
    const int hx = FAST ? 1 : (a.kx > 1 ? a.dil : 0), hy = FAST ? 1 : (a.ky > 1 ? a.dil : 0),
              hz = FAST ? 1 : (a.kz > 1 ? a.dil : 0);                                               // halo per side
    const int HX = CT_X + 2 * hx, HY = CT_Y + 2 * hy, HZ = CT_Z + 2 * hz;
    const int nrows = HX * HY * HZ;
    const int Cin = a.c0 + a.c1;
    const int nchunk = (Cin + 15) / 16;
    const int ntap = FAST ? 27 : a.kx * a.ky * a.kz;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // N split (small grids): blockIdx.z takes NT of the ntt 16-channel output blocks, so that 300 tiles of a 40^3 layer become 600 or
    // 1200 blocks with the same tile shape
    const int ntt = (a.Cout + 15) >> 4, nt0 = blockIdx.z * NT;

    const float *s0 = a.src0 + (long long)b * a.X * a.Y * a.Z * a.c0;
    const float *s1 = a.src1 ? a.src1 + (long long)b * a.X1 * a.Y1 * a.Z1 * a.c1 : nullptr;

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // halo-tile loader: row r (voxel of the halo tile), float4 q4 of its 16-channel chunk -> registers
    const int q4 = threadIdx.x & 3;
    auto load_row = [&](int r, int cbase) -> f32x4 {
        const int rz = r % HZ, ry = (r / HZ) % HY, rx = r / (HZ * HY);
        const int x = x0 - hx + rx, y = y0 - hy + ry, z = z0 - hz + rz;
        f32x4 v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (r < nrows && x >= 0 && x < a.X && y >= 0 && y < a.Y && z >= 0 && z < a.Z) {
            const int c = cbase + 4 * q4;
            if (c < a.c0) {
                const float *p = s0 + (((long long)x * a.Y + y) * a.Z + z) * a.c0 + c;
                if (c + 3 < a.c0 && (a.c0 & 3) == 0) v = *(const f32x4 *)p;
                else { for (int e = 0; e < 4; ++e) if (c + e < a.c0) v[e] = p[e]; }
            } else if (c < Cin) {
                const int c1 = c - a.c0;
                const float *p = s1 + (((long long)(x / a.ux) * a.Y1 + (y / a.uy)) * a.Z1 + (z / a.uz)) * a.c1 + c1;
                if (c1 + 3 < a.c1 && (a.c1 & 3) == 0 && (a.c0 & 3) == 0) v = *(const f32x4 *)p;
                else { for (int e = 0; e < 4; ++e) if (c1 + e < a.c1) v[e] = p[e]; }
            }
        }
        return v;
    };
    // The next chunk's halo tile is fetched into registers while the current chunk is on the matrix cores
    // (PF rows per thread), and only written to LDS after the barrier that retires the current chunk.
    constexpr int PF = 11;                                    // 64 * 11 = 704 >= 648 rows (3x3x3, dilation 1)
    const bool prefetch = nrows <= 64 * PF;
    f32x4 stage[PF];
    unsigned okbits = 0;
    // FAST (launch_mfma guarantees quad-aligned channel counts, power-of-two up-sampling, 32-bit offsets, 24-bit
    // strides): a row is ONE unconditional 16-byte load from a clamped address, zeroed by select when it is written to
    // LDS -- the eleven loads of a chunk leave back to back and are only waited for after the chunk's MFMAs
    const unsigned sZ0 = (unsigned)a.c0, sY0 = (unsigned)a.Z * sZ0, sX0 = (unsigned)a.Y * sY0;
    const int shx = FAST && a.c1 ? __ffs(a.ux) - 1 : 0, shy = FAST && a.c1 ? __ffs(a.uy) - 1 : 0, shz = FAST && a.c1 ? __ffs(a.uz) - 1 : 0;
    const unsigned tZ = (unsigned)a.c1, tY = (unsigned)a.Z1 * tZ, tX = (unsigned)a.Y1 * tY;
    auto fetch_fast = [&](int cbase, auto two_sources) {
        constexpr bool TWO = decltype(two_sources)::value;
        okbits = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int r = (threadIdx.x >> 2) + 64 * i;
            const int rz = r % HZ, ry = (r / HZ) % HY, rx = r / (HZ * HY);
            const int x = x0 - 1 + rx, y = y0 - 1 + ry, z = z0 - 1 + rz, c = cbase + 4 * q4;
            const bool ok = (r < nrows) & (x >= 0) & (x < a.X) & (y >= 0) & (y < a.Y) & (z >= 0) & (z < a.Z) & (c < Cin);
            unsigned off = __umul24((unsigned)x, sX0) + __umul24((unsigned)y, sY0) + __umul24((unsigned)z, sZ0) + (unsigned)c;
            const float *base = s0;
            if (TWO) {
                const unsigned off1 = __umul24((unsigned)(x >> shx), tX) + __umul24((unsigned)(y >> shy), tY) +
                                      __umul24((unsigned)(z >> shz), tZ) + (unsigned)(c - a.c0);
                const bool lo = c >= a.c0;
                off = lo ? off1 : off;
                base = lo ? s1 : s0;
            }
            off = ok ? off : 0u;
            okbits |= (ok ? 1u : 0u) << i;
            stage[i] = *(const f32x4 *)(base + off);
        }
    };
    if (FAST) {
        if (a.c1) fetch_fast(0, std::true_type{});
        else fetch_fast(0, std::false_type{});
    } else if (prefetch) {
#pragma unroll
        for (int i = 0; i < PF; ++i) stage[i] = load_row((threadIdx.x >> 2) + 64 * i, 0);
    }
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();                                   // previous chunk fully consumed
        const int cbase = ch * 16;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int r = (threadIdx.x >> 2) + 64 * i;
                if (r < nrows)
                    *(f32x4 *)&lds[r * LDS_ROW + 4 * q4] = ((okbits >> i) & 1u) ? stage[i] : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
        } else if (prefetch) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int r = (threadIdx.x >> 2) + 64 * i;
                if (r < nrows) *(f32x4 *)&lds[r * LDS_ROW + 4 * q4] = stage[i];
            }
        } else {
            for (int r = threadIdx.x >> 2; r < nrows; r += 64) *(f32x4 *)&lds[r * LDS_ROW + 4 * q4] = load_row(r, cbase);
        }
        __syncthreads();
        const f32x4 *wp = (const f32x4 *)wpacked + (((long long)ch * ntap) * ntt + nt0) * 64 + lane;
        // vmcnt retires in order: a weight load issued after the halo prefetch can only be waited for together with it.
        // So the weights of the first WPRE taps are requested first, then the next chunk's halo rows, and the first
        // weight load behind them is not needed before WPRE taps (WPRE * 16 NT MFMAs, ~3-4 k cycles) have run
        constexpr int WPRE = FAST ? (NT == 1 ? 6 : NT == 2 ? 4 : NT == 3 ? 3 : 2) : 1;
        f32x4 bpre[WPRE][NT];
#pragma unroll
        for (int t = 0; t < WPRE; ++t) {
            int tt = t;
            if (FOLD) {                                          // t-th tap of this chunk's parity group
                const int P = cbase / a.fold;
                tt = ((((P & 4) ? 1 : 2) - ((t >> 2) & 1)) * 3 + (((P & 2) ? 1 : 2) - ((t >> 1) & 1))) * 3 + (((P & 1) ? 1 : 2) - (t & 1));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bpre[t][nt] = wp[(tt * ntt + nt) * 64];
        }
        if (FAST) {
            if (ch + 1 < nchunk) {
                if (a.c1) fetch_fast(cbase + 16, std::true_type{});
                else fetch_fast(cbase + 16, std::false_type{});
            }
        } else if (prefetch && ch + 1 < nchunk) {
#pragma unroll
            for (int i = 0; i < PF; ++i) stage[i] = load_row((threadIdx.x >> 2) + 64 * i, cbase + 16);
        }
        // ---- taps ------------------------------------------------------------------------------
        f32x4 bfrag[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bfrag[nt] = bpre[0][nt];
        if (FAST && FOLD) {
            // this chunk belongs to one parity group: its 8 taps, 3x3x3 tap index e = (p ? 1 : 2) - t per axis (scalar
            // arithmetic; one VALU add per tap puts the tap's row offset on the lane's LDS base)
            constexpr int FHY = CT_Y + 2, FHZ = CT_Z + 2;
            const int P = cbase / a.fold, ex0 = (P & 4) ? 1 : 2, ey0 = (P & 2) ? 1 : 2, ez0 = (P & 1) ? 1 : 2;
            const float *abase = &lds[((w * FHY) * FHZ + li) * LDS_ROW + 4 * kq];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ex = ex0 - ((k >> 2) & 1), ey = ey0 - ((k >> 1) & 1), ez = ez0 - (k & 1);
                f32x4 bnext[NT];
                const int kn = k + 1 < 8 ? k + 1 : k;
                const int tn = ((ex0 - ((kn >> 2) & 1)) * 3 + (ey0 - ((kn >> 1) & 1))) * 3 + (ez0 - (kn & 1));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bnext[nt] = (kn < WPRE) ? bpre[kn][nt] : wp[(tn * ntt + nt) * 64];
                const float *at = abase + ((ex * FHY + ey) * FHZ + ez) * LDS_ROW;
                f32x4 av[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) av[mt] = *(const f32x4 *)(at + (mt * FHZ) * LDS_ROW);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][m], bfrag[nt][m], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bfrag[nt] = bnext[nt];
            }
        } else if (FAST) {
            constexpr int FHY = CT_Y + 2, FHZ = CT_Z + 2;
            const float *abase = &lds[((w * FHY) * FHZ + li) * LDS_ROW + 4 * kq];
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                f32x4 bnext[NT];
                const int tn = (t + 1 < 27) ? t + 1 : t;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bnext[nt] = (tn < WPRE) ? bpre[tn][nt] : wp[(tn * ntt + nt) * 64];
                const int dz = t % 3, dy = (t / 3) % 3, dx = t / 9;
                f32x4 av[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    av[mt] = *(const f32x4 *)(abase + ((dx * FHY + (mt + dy)) * FHZ + dz) * LDS_ROW);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][m], bfrag[nt][m], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bfrag[nt] = bnext[nt];
            }
        } else {
        for (int t = 0; t < ntap; ++t) {
            f32x4 bnext[NT];
            const int tn = (t + 1 < ntap) ? t + 1 : t;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bnext[nt] = wp[((long long)tn * ntt + nt) * 64];
            const int dz = t % a.kz, dy = (t / a.kz) % a.ky, dx = t / (a.kz * a.ky);
            const int rx = w + dx * a.dil, rz = li + dz * a.dil;
            f32x4 av[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ry = mt + dy * a.dil;
                av[mt] = *(const f32x4 *)&lds[((rx * HY + ry) * HZ + rz) * LDS_ROW + 4 * kq];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][m], bfrag[nt][m], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfrag[nt] = bnext[nt];
        }
        }
    }
    // ---- epilogue: D[row = (lane>>4)*4 + r][col = lane&15] ---------------------------------------
    float *ob = a.out + (long long)b * a.OX * a.OY * a.OZ * a.Cout;
    const int x = x0 + w;
    if (x < a.OX) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int y = y0 + mt;
            if (y >= a.OY) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = (nt0 + nt) * 16 + li;
                if (co >= a.Cout) continue;
                const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int z = z0 + kq * 4 + r;
                    if (z < a.OZ)
                        ob[(((long long)x * a.OY + y) * a.OZ + z) * a.Cout + co] = activate(acc[mt][nt][r] + bv, a.act);
                }
            }
        }
    }
}

// pack Keras-layout weights [ntap, Cin, Cout] into fragment order [chunk][tap][nt][lane][m]:
// value = W[tap][16 chunk + 4 (lane>>4) + m][16 nt + (lane & 15)], zero outside
__global__ void conv3d_pack_weights(const float *__restrict__ w, int ntap, int Cin, int Cout, int NT, int nchunk,
                                    float *__restrict__ packed) {
    const long long total = (long long)nchunk * ntap * NT * 64 * 4;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int m = e & 3, lane = (e >> 2) & 63;
        long long r = e >> 8;
        const int nt = r % NT; r /= NT;
        const int t = r % ntap; const int ch = r / ntap;
        const int ci = ch * 16 + 4 * (lane >> 4) + m, co = nt * 16 + (lane & 15);
        packed[e] = (ci < Cin && co < Cout) ? w[((long long)t * Cin + ci) * Cout + co] : 0.0f;
    }
}

// ============================================================================================
// direct convolution: one thread per output voxel, 16 output channels per pass
// ============================================================================================
__global__ __launch_bounds__(256) void conv3d_direct(ConvArgs a, const float *__restrict__ w) {
    const int b = blockIdx.y;
    const int Cin = a.c0 + a.c1;
    const float *s0 = a.src0 + (long long)b * a.X * a.Y * a.Z * a.c0;
    const float *s1 = a.src1 ? a.src1 + (long long)b * a.X1 * a.Y1 * a.Z1 * a.c1 : nullptr;
    float *ob = a.out + (long long)b * a.OX * a.OY * a.OZ * a.Cout;
    const long long nvox = (long long)a.OX * a.OY * a.OZ;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nvox; q += (long long)gridDim.x * blockDim.x) {
        const int oz = q % a.OZ, oy = (q / a.OZ) % a.OY, ox = q / ((long long)a.OZ * a.OY);
        for (int cb = 0; cb < a.Cout; cb += 16) {
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
            for (int dx = 0; dx < a.kx; ++dx) {
                const int x = ox + dx * a.dil - a.px;
                if (x < 0 || x >= a.X) continue;
                for (int dy = 0; dy < a.ky; ++dy) {
                    const int y = oy + dy * a.dil - a.py;
                    if (y < 0 || y >= a.Y) continue;
                    for (int dz = 0; dz < a.kz; ++dz) {
                        const int z = oz + dz * a.dil - a.pz;
                        if (z < 0 || z >= a.Z) continue;
                        const float *wt = w + ((long long)((dx * a.ky + dy) * a.kz + dz) * Cin) * a.Cout + cb;
                        const float *p0 = s0 + (((long long)x * a.Y + y) * a.Z + z) * a.c0;
                        for (int ci = 0; ci < a.c0; ++ci) {
                            const float xv = p0[ci];
#pragma unroll
                            for (int j = 0; j < 16; ++j)
                                if (cb + j < a.Cout) acc[j] = fmaf(xv, wt[(long long)ci * a.Cout + j], acc[j]);
                        }
                        if (s1) {
                            const float *p1 = s1 + (((long long)(x / a.ux) * a.Y1 + (y / a.uy)) * a.Z1 + (z / a.uz)) * a.c1;
                            for (int ci = 0; ci < a.c1; ++ci) {
                                const float xv = p1[ci];
#pragma unroll
                                for (int j = 0; j < 16; ++j)
                                    if (cb + j < a.Cout) acc[j] = fmaf(xv, wt[(long long)(a.c0 + ci) * a.Cout + j], acc[j]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (cb + j < a.Cout) ob[q * a.Cout + cb + j] = activate(acc[j] + (a.bias ? a.bias[cb + j] : 0.0f), a.act);
        }
    }
}

// ============================================================================================
// 1x1 conv (+ optional channel softmax) : the likelihood / prediction head.  Cout <= 64 in registers.
// ============================================================================================
template <int CO_MAX>
__global__ __launch_bounds__(256) void conv1x1_softmax(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias, float *__restrict__ y,
                                                       long long nvox, int Cin, int Cout, int softmax, int act) {
    extern __shared__ float wl[];          // [Cin][Cout] + [Cout]
    for (int i = threadIdx.x; i < Cin * Cout; i += blockDim.x) wl[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) wl[Cin * Cout + i] = bias ? bias[i] : 0.0f;
    __syncthreads();
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nvox; q += (long long)gridDim.x * blockDim.x) {
        float acc[CO_MAX];
#pragma unroll
        for (int j = 0; j < CO_MAX; ++j) acc[j] = j < Cout ? wl[Cin * Cout + j] : 0.0f;
        const float *xp = x + q * Cin;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xp[ci];
#pragma unroll
            for (int j = 0; j < CO_MAX; ++j)
                if (j < Cout) acc[j] = fmaf(xv, wl[ci * Cout + j], acc[j]);
        }
        if (softmax) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < CO_MAX; ++j) if (j < Cout) m = fmaxf(m, acc[j]);
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < CO_MAX; ++j) if (j < Cout) { acc[j] = expf(acc[j] - m); s += acc[j]; }
            const float inv = 1.0f / s;
#pragma unroll
            for (int j = 0; j < CO_MAX; ++j) if (j < Cout) acc[j] *= inv;
        } else {
#pragma unroll
            for (int j = 0; j < CO_MAX; ++j) if (j < Cout) acc[j] = activate(acc[j], act);
        }
        float *yp = y + q * Cout;
#pragma unroll
        for (int j = 0; j < CO_MAX; ++j) if (j < Cout) yp[j] = acc[j];
    }
}

// ---- vectorised forms of the two HBM-bound layers: G = Cout/4 lanes per voxel, each lane owns 4 output
// channels, so a voxel's output row is written as one contiguous Cout*4-byte segment (the one-thread-per-voxel
// forms above write 64-128 B per thread at a 64-128 B stride: 64 cache lines per store instruction).
// conv1x1_vec: the likelihood head (+ softmax across the lane-group with xor-shuffles).
// CIN > 0: the input-channel count as a template parameter (multiple of 4, 16-byte aligned rows): the voxel's row is requested
// as CIN / 4 float4 loads up front (the G lanes of a voxel read the same addresses: one L1 access) and the contraction is fully
// unrolled; CIN = 0 keeps the run-time loop with one scalar load per input channel (a chain of Cin dependent L1 round trips per voxel)
template <int G, int CIN = 0>
__global__ __launch_bounds__(256) void conv1x1_vec(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ y, long long nvox,
                                                   int Cin, int softmax, int act) {
    constexpr int Cout = 4 * G;
    constexpr int NG = 256 / G;
    extern __shared__ float wl[];          // [Cin][Cout] + [Cout]
    for (int i = threadIdx.x; i < Cin * Cout; i += blockDim.x) wl[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) wl[Cin * Cout + i] = bias ? bias[i] : 0.0f;
    __syncthreads();
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    // U voxel groups per wave and iteration when the rows are loaded as float4: their U * CIN / 4 loads are in flight together (one
    // group per iteration kept 512 bytes per wave in flight -- the kernel ran at the latency of its loads, 4 TB/s)
    constexpr int U = CIN > 0 ? 4 : 1;
    const long long stride = (long long)gridDim.x * NG;
    for (long long q0 = (long long)blockIdx.x * NG + g; q0 < nvox; q0 += U * stride) {
        f32x4 xr[U][CIN > 0 ? CIN / 4 : 1];
        if (CIN > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long qq = q0 + u * stride < nvox ? q0 + u * stride : nvox - 1;
#pragma unroll
                for (int k = 0; k < CIN / 4; ++k) xr[u][k] = *(const f32x4 *)(x + qq * CIN + 4 * k);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = q0 + u * stride;
            if (q >= nvox) break;
            f32x4 acc = *(const f32x4 *)&wl[Cin * Cout + 4 * lg];
            if (CIN > 0) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float xv = xr[u][ci >> 2][ci & 3];
                    const f32x4 wv = *(const f32x4 *)&wl[ci * Cout + 4 * lg];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv, wv[e], acc[e]);
                }
            } else {
                const float *xp = x + q * Cin;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float xv = xp[ci];                       // same address for the G lanes of the voxel (broadcast)
                    const f32x4 wv = *(const f32x4 *)&wl[ci * Cout + 4 * lg];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv, wv[e], acc[e]);
                }
            }
            if (softmax) {
                float m = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
#pragma unroll
                for (int off = 1; off < G; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                float sum = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] = softmax_exp(acc[e] - m); sum += acc[e]; }
#pragma unroll
                for (int off = 1; off < G; off <<= 1) sum += __shfl_xor(sum, off, 64);
                const float inv = softmax_rcp(sum);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] *= inv;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = activate(acc[e], act);
            }
            __builtin_nontemporal_store(acc, (f32x4 *)(y + q * Cout) + lg);
        }
    }
}

// 1x1x1 convolution (+ channel softmax) with CIN in {16, 32} input channels and 4 G outputs -- the likelihood layer of the unets and
// its input gradient: every lane loads a DISTINCT float4 of the input (64 lanes = 1 KB contiguous per instruction, U x CIN / 16
// instructions in flight per wave), the rows of 16 voxels cross to the G lanes of their voxel through a wave-private LDS tile.
// (conv1x1_vec lets the G lanes of a voxel load the same row: 512 distinct bytes in flight per wave, and the kernel ran at the
// latency of its loads -- 4 TB/s on 786 MB.)
template <int G, int CIN>
__global__ __launch_bounds__(256) void conv1x1_rows(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ bias, float *__restrict__ y, long long nvox,
                                                    int softmax, int act) {
    constexpr int Cout = 4 * G, U = CIN == 16 ? 4 : 2, LPT = CIN / 16, VPP = 64 / G, NPASS = 16 / VPP;   // loads per 16-voxel tile; voxels per pass
    static_assert(G == 4 || G == 8 || G == 16, "16 voxels per tile = a whole number of passes");
    __shared__ __attribute__((aligned(16))) float xs[4][16 * CIN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = lane % G, gv = lane / G;
    f32x4 wr[CIN];                                                                       // this lane's 4 output channels of every input channel
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) wr[ci] = *(const f32x4 *)(w + ci * Cout + 4 * lg);
    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias) bv = *(const f32x4 *)(bias + 4 * lg);
    float *tile = xs[wave];
    const long long nwaves = (long long)gridDim.x * 4;
    const f32x4 *x4 = (const f32x4 *)x;
    const long long n4 = nvox * (CIN / 4);
    for (long long base = ((long long)blockIdx.x * 4 + wave) * (16 * U); base < nvox; base += nwaves * (16 * U)) {
        f32x4 xq[U][LPT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                const long long e = (base + u * 16) * (CIN / 4) + k * 64 + lane;        // float4 index: the tile is 16 * CIN / 4 contiguous float4
                xq[u][k] = x4[e < n4 ? e : n4 - 1];
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < LPT; ++k) *(f32x4 *)(tile + (k * 64 + lane) * 4) = xq[u][k];   // [voxel][CIN floats] as loaded
            __builtin_amdgcn_wave_barrier();                                              // same wave: LDS operations execute in order
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const long long q = base + u * 16 + p * VPP + gv;
                const float *row = tile + (p * VPP + gv) * CIN;
                f32x4 acc = bv;
#pragma unroll
                for (int k = 0; k < CIN / 4; ++k) {
                    const f32x4 xv = *(const f32x4 *)(row + 4 * k);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv[j], wr[4 * k + j][e], acc[e]);
                }
                if (softmax) {
                    float m = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
#pragma unroll
                    for (int off = 1; off < G; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                    float sum = 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[e] = softmax_exp(acc[e] - m); sum += acc[e]; }
#pragma unroll
                    for (int off = 1; off < G; off <<= 1) sum += __shfl_xor(sum, off, 64);
                    const float inv = softmax_rcp(sum);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] *= inv;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = activate(acc[e], act);
                }
                if (q < nvox) __builtin_nontemporal_store(acc, (f32x4 *)(y + q * Cout) + lg);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// conv3d_c1_vec: first encoder layer, Cin == 1 (27 taps of a scalar image -> Cout channels), SAME padding.
template <int G>
__global__ __launch_bounds__(256) void conv3d_c1_vec(ConvArgs a, const float *__restrict__ w) {
    constexpr int Cout = 4 * G;
    constexpr int NG = 256 / G;
    extern __shared__ float wl[];          // [ntap][Cout] + [Cout]
    const int ntap = a.kx * a.ky * a.kz;
    for (int i = threadIdx.x; i < ntap * Cout; i += blockDim.x) wl[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) wl[ntap * Cout + i] = a.bias ? a.bias[i] : 0.0f;
    __syncthreads();
    const int b = blockIdx.y;
    const float *s0 = a.src0 + (long long)b * a.X * a.Y * a.Z;
    float *ob = a.out + (long long)b * a.OX * a.OY * a.OZ * Cout;
    const long long nvox = (long long)a.OX * a.OY * a.OZ;
    const int lg = threadIdx.x % G;
    const long long g = threadIdx.x / G;
    for (long long q = (long long)blockIdx.x * NG + g; q < nvox; q += (long long)gridDim.x * NG) {
        const int oz = (int)(q % a.OZ), oy = (int)((q / a.OZ) % a.OY), ox = (int)(q / ((long long)a.OZ * a.OY));
        f32x4 acc = *(const f32x4 *)&wl[ntap * Cout + 4 * lg];
        int t = 0;
        for (int dx = 0; dx < a.kx; ++dx) {
            const int x = ox + dx * a.dil - a.px;
            for (int dy = 0; dy < a.ky; ++dy) {
                const int y = oy + dy * a.dil - a.py;
                for (int dz = 0; dz < a.kz; ++dz, ++t) {
                    const int z = oz + dz * a.dil - a.pz;
                    const bool in = x >= 0 && x < a.X && y >= 0 && y < a.Y && z >= 0 && z < a.Z;
                    const float xv = in ? s0[((long long)x * a.Y + y) * a.Z + z] : 0.0f;
                    const f32x4 wv = *(const f32x4 *)&wl[t * Cout + 4 * lg];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv, wv[e], acc[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = activate(acc[e], a.act);
        *((f32x4 *)(ob + q * Cout) + lg) = acc;
    }
}

// conv3d_c1_mfma: first encoder layer (Cin == 1, 3x3x3, SAME, dilation 1) as a [voxels x 27] x [27 x Cout] GEMM on the
// matrix cores.  The im2col matrix is never built: a block stages the 6 x 6 x 18 halo of its 4 x 4 x 16 output tile in
// LDS (2.6 KB) and lane l of a wave reads A[i = voxel z = l & 15][k = tap 4 ks + (l >> 4)] = halo[voxel + offset(tap)]
// for 7 k-steps (taps 27 is padded with a zero weight).  The scalar-FMA version (conv3d_c1_vec) spends ~800 lane
// instructions per voxel on 27 x Cout FMAs; here it is 7 MFMAs per 16 voxels and the layer becomes a 262 MB write.
// POOL (round 5): the 2 x 2 x 2 MaxPooling3D that follows the first encoder convolution (models.py:1436-1438) computed from the tile while
// it is still in LDS -- the full-resolution output is written as before (the decoder's skip connection reads it), the pooled tensor
// in addition: 33 MB more to write at 160^3 x 16 instead of a second kernel that reads 262 MB back.  z pairs and y pairs are a wave's
// own (rows of its LDS tile, consecutive y iterations), x pairs meet in LDS after one block barrier per tile.
template <int NT, bool POOL = false>
__global__ __launch_bounds__(256) void conv3d_c1_mfma(ConvArgs a, const float *__restrict__ w, unsigned nby, unsigned nbz,
                                                      unsigned nblk, float *__restrict__ pool_out) {
    // persistent blocks (round 3): the weights, bias and tap offsets are loaded once per block instead of once per 4x4x16 tile, and the
    // halo of tile i + 1 is fetched into registers while tile i is on the matrix cores (one tile per block paid ~150 instructions of
    // prologue and two exposed memory latencies for 28 MFMAs per wave)
    constexpr int HX = 6, HY = 6, HZ = 18, NH = HX * HY * HZ, PH = (NH + 255) / 256;
    __shared__ float halo[NH];
    __shared__ __attribute__((aligned(16))) float otile[4][16 * 16 * NT];
    __shared__ __attribute__((aligned(16))) float ptile[POOL ? 4 : 1][2][8 * 16 * NT];     // per wave (x): [y pair][z pair][channel]
    const int b = blockIdx.y;
    const float *s0 = a.src0 + (long long)b * a.X * a.Y * a.Z;
    float *ob = a.out + (long long)b * a.OX * a.OY * a.OZ * a.Cout;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // B fragments (weights) and the lane's tap offsets, 7 k-steps
    float bf[NT][7];
    int toff[7];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int tap = 4 * ks + l4;
        const int tt = tap < 27 ? tap : 0;
        toff[ks] = ((tt / 9) * HY + (tt / 3) % 3) * HZ + tt % 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt][ks] = tap < 27 ? w[tap * a.Cout + nt * 16 + l15] : 0.0f;
    }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias ? a.bias[nt * 16 + l15] : 0.0f;
    // the halo elements this thread stages (fixed positions inside the tile)
    int hrx[PH], hry[PH], hrz[PH];
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        const int e = threadIdx.x + 256 * i, ee = e < NH ? e : 0;
        hrz[i] = ee % HZ; hry[i] = (ee / HZ) % HY; hrx[i] = ee / (HZ * HY);
    }
    float hv[PH];
    auto fetch = [&](unsigned lb) __attribute__((always_inline)) {
        const int bz = lb % nbz, by = (lb / nbz) % nby, bx = lb / (nbz * nby);
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int gx = bx * 4 + hrx[i] - 1, gy = by * 4 + hry[i] - 1, gz = bz * 16 + hrz[i] - 1;
            const bool ok = gx >= 0 && gx < a.X && gy >= 0 && gy < a.Y && gz >= 0 && gz < a.Z;
            const float v = s0[ok ? ((long long)gx * a.Y + gy) * a.Z + gz : 0];
            hv[i] = ok ? v : 0.0f;
        }
    };
    // XCD-aware persistent walk: block (xcd, slot) takes tiles slot, slot + J, ... of its XCD's contiguous eighth
    const unsigned xcd = blockIdx.x % NRT_NXCD, J = gridDim.x / NRT_NXCD;
    const unsigned T8 = (nblk + NRT_NXCD - 1) / NRT_NXCD;
    const unsigned tend = (xcd + 1) * T8 < nblk ? (xcd + 1) * T8 : nblk;
    unsigned lb = xcd * T8 + blockIdx.x / NRT_NXCD;
    if (lb >= tend) return;
    fetch(lb);
    for (; lb < tend; lb += J) {
        __syncthreads();                                      // the previous tile has been read
#pragma unroll
        for (int i = 0; i < PH; ++i)
            if (threadIdx.x + 256 * i < NH) halo[threadIdx.x + 256 * i] = hv[i];
        __syncthreads();
        const int bz = lb % nbz, by = (lb / nbz) % nby, bx = lb / (nbz * nby);
        const int x0 = bx * 4, y0 = by * 4, z0 = bz * 16;
        if (lb + J < tend) fetch(lb + J);
        // wave wv owns x = x0 + wv; its 4 groups are the y rows; a group = 16 consecutive z
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int base = (wv * HY + g) * HZ + l15;            // halo index of (x, y, z) at tap (0,0,0)
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const float af = halo[base + toff[ks]];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[nt][ks], acc[nt], 0, 0, 0);
            }
            const int x = x0 + wv, y = y0 + g;
            // the wave's 16 voxels x Cout outputs are ONE contiguous run of the output tensor: transposed through a wave-private LDS
            // tile so that every lane stores 16 bytes (1 KB contiguous per store instruction instead of four 64-byte segments)
            float *ot = otile[wv];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ot[(l4 * 4 + r) * (16 * NT) + nt * 16 + l15] = activate(acc[nt][r] + bias[nt], a.act);
            __builtin_amdgcn_wave_barrier();
            if (x < a.OX && y < a.OY) {
                f32x4 *po = (f32x4 *)(ob + (((long long)x * a.OY + y) * a.OZ + z0) * a.Cout);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int j = lane + 64 * i;                          // float4 index inside the run: voxel = 4 j / Cout
                    if (z0 + (4 * j) / (16 * NT) < a.OZ) __builtin_nontemporal_store(*(const f32x4 *)(ot + 4 * j), po + j);
                }
            }
            if (POOL) {
                // z pairs of this y row: 8 x 16 NT values, lane = (z pair, 4 channels) -- 2 NT float4 per lane
                constexpr int C4 = 4 * NT;                                // float4 per voxel row
                float *pt = ptile[wv][g >> 1];
#pragma unroll
                for (int i = 0; i < (8 * C4 + 63) / 64; ++i) {
                    const int j = lane + 64 * i;
                    if (j < 8 * C4) {
                        const int zp = j / C4, c4 = j % C4;
                        const f32x4 v0 = *(const f32x4 *)(ot + (2 * zp) * (16 * NT) + 4 * c4), v1 = *(const f32x4 *)(ot + (2 * zp + 1) * (16 * NT) + 4 * c4);
                        f32x4 m = (f32x4){fmaxf(v0[0], v1[0]), fmaxf(v0[1], v1[1]), fmaxf(v0[2], v1[2]), fmaxf(v0[3], v1[3])};
                        if (g & 1) {                                      // the second row of a y pair: fold the first one in
                            const f32x4 q = *(const f32x4 *)(pt + 4 * j);
                            m = (f32x4){fmaxf(m[0], q[0]), fmaxf(m[1], q[1]), fmaxf(m[2], q[2]), fmaxf(m[3], q[3])};
                        }
                        *(f32x4 *)(pt + 4 * j) = m;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (POOL) {
            __syncthreads();                                          // the four x planes of the tile are pooled over y and z
            // 2 x pairs x 2 y pairs x 8 z pairs x 4 NT float4 = 128 NT float4: one (or NT / 2 ...) per thread
            constexpr int C4 = 4 * NT, PER = 2 * 2 * 8 * C4;
            float *pb = pool_out + (long long)b * (a.OX / 2) * (a.OY / 2) * (a.OZ / 2) * a.Cout;
            for (int j = threadIdx.x; j < PER; j += 256) {
                const int c4 = j % C4, zp = (j / C4) % 8, yp = (j / (8 * C4)) % 2, xp = j / (16 * C4);
                const f32x4 u = *(const f32x4 *)&ptile[2 * xp][yp][(zp * C4 + c4) * 4], v = *(const f32x4 *)&ptile[2 * xp + 1][yp][(zp * C4 + c4) * 4];
                const f32x4 m = (f32x4){fmaxf(u[0], v[0]), fmaxf(u[1], v[1]), fmaxf(u[2], v[2]), fmaxf(u[3], v[3])};
                const int px_ = x0 / 2 + xp, py_ = y0 / 2 + yp, pz_ = z0 / 2 + zp;
                __builtin_nontemporal_store(m, (f32x4 *)(pb + (((long long)px_ * (a.OY / 2) + py_) * (a.OZ / 2) + pz_) * a.Cout) + c4);
            }
        }
    }
}

// G = C/4 lanes per voxel: coalesced 16-byte loads / stores, channel max and sum by xor-shuffles inside the lane-group
template <int G>
__global__ __launch_bounds__(256) void softmax_lastdim_vec(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, long long n) {
    constexpr int NG = 256 / G;
    const int lg = threadIdx.x % G;
    const long long ngroups = (long long)gridDim.x * NG;
    const long long niter = (n + ngroups - 1) / ngroups;
    for (long long it = 0; it < niter; ++it) {
        const long long vv = (long long)blockIdx.x * NG + threadIdx.x / G + it * ngroups;
        const bool live = vv < n;
        const long long v = live ? vv : n - 1;
        const f32x4 xv = x[v * G + lg];
        float m = fmaxf(fmaxf(xv[0], xv[1]), fmaxf(xv[2], xv[3]));
#pragma unroll
        for (int off = 1; off < G; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        f32x4 e;
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = softmax_exp(xv[k] - m);
        float s = (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
        for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off, 64);
        const float inv = softmax_rcp(s);
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] *= inv;
        if (live) y[v * G + lg] = e;
    }
}

__global__ __launch_bounds__(256) void softmax_lastdim(const float *__restrict__ x, float *__restrict__ y, long long n, int C) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        const float *xp = x + q * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, xp[c]);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s += expf(xp[c] - m);
        const float inv = 1.0f / s;
        for (int c = 0; c < C; ++c) y[q * C + c] = expf(xp[c] - m) * inv;
    }
}

// MaxPooling3D, stride = pool size, SAME (partial windows at the end) or VALID; one thread per (voxel, channel)
__global__ __launch_bounds__(256) void maxpool3d(const float *__restrict__ x, float *__restrict__ y, int X, int Y, int Z,
                                                 int C, int OX, int OY, int OZ, int px, int py, int pz) {
    const int b = blockIdx.y;
    const float *xb = x + (long long)b * X * Y * Z * C;
    float *yb = y + (long long)b * OX * OY * OZ * C;
    const long long total = (long long)OX * OY * OZ * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = e % C;
        long long q = e / C;
        const int oz = q % OZ, oy = (q / OZ) % OY, ox = q / ((long long)OZ * OY);
        float m = -INFINITY;
        for (int dx = 0; dx < px; ++dx) {
            const int xx = ox * px + dx; if (xx >= X) break;
            for (int dy = 0; dy < py; ++dy) {
                const int yy = oy * py + dy; if (yy >= Y) break;
                for (int dz = 0; dz < pz; ++dz) {
                    const int zz = oz * pz + dz; if (zz >= Z) break;
                    m = fmaxf(m, xb[(((long long)xx * Y + yy) * Z + zz) * C + c]);
                }
            }
        }
        yb[e] = m;
    }
}

// UpSampling3D (nearest repeat) of `lo` + concatenate([skip, up], channel axis) -- stand-alone form
__global__ __launch_bounds__(256) void upsample_concat(const float *__restrict__ skip, int c0, const float *__restrict__ lo,
                                                       int c1, float *__restrict__ y, int X, int Y, int Z, int ux, int uy,
                                                       int uz) {
    const int b = blockIdx.y;
    const int C = c0 + c1, X1 = X / ux, Y1 = Y / uy, Z1 = Z / uz;
    const float *sb = skip ? skip + (long long)b * X * Y * Z * c0 : nullptr;
    const float *lb = lo + (long long)b * X1 * Y1 * Z1 * c1;
    float *yb = y + (long long)b * X * Y * Z * C;
    const long long total = (long long)X * Y * Z * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = e % C;
        long long q = e / C;
        const int z = q % Z, yy = (q / Z) % Y, x = q / ((long long)Z * Y);
        yb[e] = (c < c0) ? sb[q * c0 + c]
                         : lb[(((long long)(x / ux) * Y1 + (yy / uy)) * Z1 + (z / uz)) * c1 + (c - c0)];
    }
}

// y = act(a + b) [+ per-channel affine]: residual merges (models.py:1423-1429) and inference BatchNorm;
// with ACT_MUL_B in the activation word y = act(a) * b: the likelihood x prior merge of models.add_prior (models.py:408-414)
__global__ __launch_bounds__(256) void add_act_affine(const float *__restrict__ a, const float *__restrict__ bsrc,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      float *__restrict__ y, long long n, int C, int act) {
    const bool mul = (act & ACT_MUL_B) != 0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        float v = a[e];
        if (mul) v = activate_ew(v, act & 0xff) * bsrc[e];
        else {
            if (bsrc) v += bsrc[e];
            v = activate_ew(v, act);
        }
        if (scale) v = v * scale[e % C] + shift[e % C];
        y[e] = v;
    }
}

// the same on float4 (n, C multiples of 4, 16-byte aligned tensors): the channel of a thread's float4 advances by a fixed step per
// grid stride, so the per-element 64-bit modulo of the scalar form becomes one per thread
__global__ __launch_bounds__(256) void add_act_affine_v4(const f32x4 *__restrict__ a, const f32x4 *__restrict__ bsrc,
                                                         const float *__restrict__ scale, const float *__restrict__ shift,
                                                         f32x4 *__restrict__ y, long long n4, int C4, int act) {
    const bool mul = (act & ACT_MUL_B) != 0;
    long long ebeg, eend;                                      // one contiguous range per block (nrt_block_range)
    nrt_block_range(n4, 256, ebeg, eend);
    long long e = ebeg + threadIdx.x;
    unsigned c = (unsigned)(e % C4);
    const unsigned cstep = (unsigned)(256 % C4);
    for (; e < eend; e += 256) {
        f32x4 v = a[e];
        if (mul) {
            const f32x4 b = bsrc[e];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = activate_ew(v[k], act & 0xff) * b[k];
        } else {
            if (bsrc) v += bsrc[e];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = activate_ew(v[k], act);
        }
        if (scale) {
            const f32x4 sc = *(const f32x4 *)(scale + 4 * c), sh = *(const f32x4 *)(shift + 4 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] * sc[k] + sh[k];
        }
        y[e] = v;
        c += cstep;
        if (c >= (unsigned)C4) c -= (unsigned)C4;
    }
}

int conv_args(ConvArgs &a, const float *src0, int c0, const float *src1, int c1, const int *up, const float *bias,
              float *out, const int *shape, const int *ksize, int cout, int dilation, int padding_same, int act) {
    if (!src0 || !out || !shape || !ksize) return NRT_ERR_INVALID_ARG;
    if (c0 < 1 || c1 < 0 || cout < 1 || dilation < 1) return NRT_ERR_INVALID_ARG;
    if (c1 > 0 && (!src1 || !up)) return NRT_ERR_INVALID_ARG;
    a.src0 = src0; a.src1 = c1 > 0 ? src1 : nullptr; a.bias = bias; a.out = out;
    a.X = shape[0]; a.Y = shape[1]; a.Z = shape[2];
    a.c0 = c0; a.c1 = c1; a.Cout = cout;
    a.ux = c1 > 0 ? up[0] : 1; a.uy = c1 > 0 ? up[1] : 1; a.uz = c1 > 0 ? up[2] : 1;
    if (a.ux < 1 || a.uy < 1 || a.uz < 1) return NRT_ERR_INVALID_ARG;
    if (c1 > 0 && (a.X % a.ux || a.Y % a.uy || a.Z % a.uz)) return NRT_ERR_INVALID_ARG;
    a.X1 = a.X / a.ux; a.Y1 = a.Y / a.uy; a.Z1 = a.Z / a.uz;
    a.kx = ksize[0]; a.ky = ksize[1]; a.kz = ksize[2]; a.dil = dilation;
    if (a.kx < 1 || a.ky < 1 || a.kz < 1) return NRT_ERR_INVALID_ARG;
    if (padding_same) {
        a.px = ((a.kx - 1) * dilation) / 2; a.py = ((a.ky - 1) * dilation) / 2; a.pz = ((a.kz - 1) * dilation) / 2;
        a.OX = a.X; a.OY = a.Y; a.OZ = a.Z;
    } else {
        a.px = a.py = a.pz = 0;
        a.OX = a.X - (a.kx - 1) * dilation; a.OY = a.Y - (a.ky - 1) * dilation; a.OZ = a.Z - (a.kz - 1) * dilation;
        if (a.OX < 1 || a.OY < 1 || a.OZ < 1) return NRT_ERR_INVALID_ARG;
    }
    a.act = act;
    a.fold = 0;
    return NRT_OK;
}

bool mfma_ok(const ConvArgs &a, int padding_same) {
    if (!padding_same) return false;
    auto okk = [](int k) { return k == 1 || k == 3; };
    if (!okk(a.kx) || !okk(a.ky) || !okk(a.kz)) return false;
    if (a.dil > 2) return false;
    if (a.Cout > 64) return false;
    if ((a.c0 + a.c1) < 8) return false;                         // K too thin: the direct kernel is HBM-bound anyway
    if (a.c1 > 0 && (a.c0 % 4)) return false;
    return true;
}

size_t mfma_lds_bytes(const ConvArgs &a) {
    const int hx = a.kx > 1 ? a.dil : 0, hy = a.ky > 1 ? a.dil : 0, hz = a.kz > 1 ? a.dil : 0;
    return (size_t)(CT_X + 2 * hx) * (CT_Y + 2 * hy) * (CT_Z + 2 * hz) * LDS_ROW * sizeof(float);
}

template <int NT>
int launch_mfma(const ConvArgs &a, const float *wpacked, int batch, hipStream_t st, int nsplit = 1) {
    const unsigned nbx = (a.OX + CT_X - 1) / CT_X, nby = (a.OY + CT_Y - 1) / CT_Y, nbz = (a.OZ + CT_Z - 1) / CT_Z;
    const unsigned nblk = nbx * nby * nbz;
    const size_t shm = mfma_lds_bytes(a);
    const bool pow2 = a.c1 == 0 || ((a.ux & (a.ux - 1)) == 0 && (a.uy & (a.uy - 1)) == 0 && (a.uz & (a.uz - 1)) == 0);
    const bool fast = a.kx == 3 && a.ky == 3 && a.kz == 3 && a.dil == 1 && (a.c0 & 3) == 0 && (a.c1 & 3) == 0 && pow2 &&
                      (long long)a.X * a.Y * a.Z * a.c0 < (1ll << 31) && (long long)a.Y * a.Z * (a.c0 > a.c1 ? a.c0 : a.c1) < (1ll << 24);
    if (shm > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)conv3d_mfma<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return NRT_ERR_LAUNCH;
    }
    dim3 grid(nrt_xcd_grid(nblk), batch, nsplit);
    if (a.fold) {
        if (!fast || a.c1 || a.fold % 16 || a.c0 != 8 * a.fold) return NRT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv3d_mfma<NT, true, true>), grid, dim3(256), shm, st, a, wpacked, nblk, nbx, nby, nbz);
    } else if (fast) hipLaunchKernelGGL((conv3d_mfma<NT, true>), grid, dim3(256), shm, st, a, wpacked, nblk, nbx, nby, nbz);
    else hipLaunchKernelGGL((conv3d_mfma<NT, false>), grid, dim3(256), shm, st, a, wpacked, nblk, nbx, nby, nbz);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// all N-tiles in one block, or -- when the grid has fewer than two tiles per CU (40^3 layers: 300 tiles on 256 CUs) -- the 16-channel
// output blocks split over blockIdx.z so that every CU gets several smaller blocks
int dispatch_mfma(const ConvArgs &a, const float *wpacked, int batch, hipStream_t st) {
    const int ntt = (a.Cout + 15) / 16;
    const long long tiles = (long long)batch * ((a.OX + CT_X - 1) / CT_X) * ((a.OY + CT_Y - 1) / CT_Y) * ((a.OZ + CT_Z - 1) / CT_Z);
    const long long want = 2ll * nrt_num_cus();
    int split = 1;
    if (ntt > 1 && tiles < want) {
        if (ntt == 4) split = tiles * 2 >= want ? 2 : 4;
        else split = ntt;                                        // 2 or 3 output blocks: one per block
    }
    switch (ntt / split) {
        case 1: return launch_mfma<1>(a, wpacked, batch, st, split);
        case 2: return launch_mfma<2>(a, wpacked, batch, st, split);
        case 3: return launch_mfma<3>(a, wpacked, batch, st, split);
        default: return launch_mfma<4>(a, wpacked, batch, st, split);
    }
}

#include "conv_up2.h"
#include "conv_p27.h"

// y[b][q][P * C + c] = x[b][2 q + p][c], P = (px * 2 + py) * 2 + pz: the 8 parity sub-lattices of x as channel groups
__global__ __launch_bounds__(256) void space_to_depth2(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, int X1, int Y1, int Z1, int C4) {
    const int b = blockIdx.y;
    const long long total = (long long)X1 * Y1 * Z1 * 8 * C4;
    const f32x4 *xb = x + (long long)b * total;
    f32x4 *yb = y + (long long)b * total;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long long r = e / C4;
        const int P = (int)(r & 7); r >>= 3;
        const int z = (int)(r % Z1), yy = (int)((r / Z1) % Y1), xx = (int)(r / ((long long)Z1 * Y1));
        const long long src = (((long long)(2 * xx + (P >> 2)) * (2 * Y1) + (2 * yy + ((P >> 1) & 1))) * (2 * Z1) + (2 * z + (P & 1))) * C4 + c;
        yb[e] = xb[src];
    }
}

}  // namespace

extern "C" size_t nrt_conv3d_packed_weight_floats(const int *ksize, int cin, int cout) {
    if (!ksize || cin < 1 || cout < 1) return 0;
    const size_t ntap = (size_t)ksize[0] * ksize[1] * ksize[2];
    return (size_t)((cin + 15) / 16) * ntap * ((cout + 15) / 16) * 256;
}

extern "C" int nrt_conv3d_pack_weights_f32(const float *weights, const int *ksize, int cin, int cout, float *packed,
                                           void *stream) {
    if (!weights || !packed || !ksize || cin < 1 || cout < 1) return NRT_ERR_INVALID_ARG;
    const int ntap = ksize[0] * ksize[1] * ksize[2], NT = (cout + 15) / 16, nchunk = (cin + 15) / 16;
    hipLaunchKernelGGL(conv3d_pack_weights, dim3(256), dim3(256), 0, nrt_stream(stream), weights, ntap, cin, cout, NT, nchunk,
                       packed);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" size_t nrt_conv3d_up2_packed_weight_floats(int c0, int c1, int cout) {
    if (c0 < 16 || c1 < 16 || c0 % 16 || c1 % 16 || cout < 1) return 0;
    return up2_weight_floats(c0, c1, cout) + U2_ZERO_FLOATS;
}

extern "C" int nrt_conv3d_up2_pack_weights_f32(const float *weights, int c0, int c1, int cout, float *packed, void *stream) {
    if (!weights || !packed || nrt_conv3d_up2_packed_weight_floats(c0, c1, cout) == 0) return NRT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(conv3d_pack_weights_up2, dim3(256), dim3(256), 0, nrt_stream(stream), weights, c0, c1, cout, (cout + 15) / 16,
                       packed);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_conv3d_up2_supported(int c0, int c1, int cout, const int *shape) {
    if (!shape) return 0;
    ConvArgs a;
    const int k3[3] = {3, 3, 3}, up[3] = {2, 2, 2};
    if (c0 < 1 || conv_args(a, (const float *)16, c0, (const float *)16, c1, up, nullptr, (float *)16, shape, k3, cout, 1, 1, 0) != NRT_OK)
        return 0;
    return up2_ok(a, 1) ? 1 : 0;
}

extern "C" int nrt_conv3d_up2_f32(const float *skip, int c0, const float *lo, int c1, const float *packed_weights, const float *bias,
                                  float *out, int batch, const int *shape, int cout, int activation, void *stream) {
    ConvArgs a;
    const int k3[3] = {3, 3, 3}, up[3] = {2, 2, 2};
    int rc = conv_args(a, skip, c0, lo, c1, up, bias, out, shape, k3, cout, 1, 1, activation);
    if (rc != NRT_OK) return rc;
    if (!packed_weights || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;
    if (!up2_ok(a, 1)) return NRT_ERR_UNSUPPORTED;
    hipStream_t st = nrt_stream(stream);
    switch ((cout + 15) / 16) {
        case 1: return launch_up2<1>(a, packed_weights, batch, st);
        case 2: return launch_up2<2>(a, packed_weights, batch, st);
        case 3: return launch_up2<3>(a, packed_weights, batch, st);
        default: return launch_up2<4>(a, packed_weights, batch, st);
    }
}

extern "C" int nrt_conv3d_up2_head_supported(int c0, int c1, int cout, int labels, const int *shape) {
    if (!shape) return 0;
    ConvArgs a;
    const int k3[3] = {3, 3, 3}, up[3] = {2, 2, 2};
    float dummy;
    if (conv_args(a, &dummy, c0, &dummy, c1, up, nullptr, &dummy, shape, k3, cout, 1, 1, ACT_NONE) != NRT_OK) return 0;
    return up2_head_ok(a, labels) ? 1 : 0;
}

extern "C" int nrt_conv3d_up2_head_pack_f32(const float *head_weights, int labels, float *packed, void *stream) {
    if (!head_weights || !packed || (labels != 16 && labels != 32)) return NRT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(conv3d_pack_head_up2, dim3(2), dim3(256), 0, nrt_stream(stream), head_weights, labels, packed);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_conv3d_up2_head_f32(const float *skip, int c0, const float *lo, int c1, const float *packed_weights, const float *bias,
                                       const float *head_weights, const float *head_bias, int labels, float *out, int batch,
                                       const int *shape, int cout, int activation, void *stream) {
    ConvArgs a;
    const int k3[3] = {3, 3, 3}, up[3] = {2, 2, 2};
    int rc = conv_args(a, skip, c0, lo, c1, up, bias, out, shape, k3, cout, 1, 1, activation);
    if (rc != NRT_OK) return rc;
    if (!packed_weights || !head_weights || !head_bias || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if ((((uintptr_t)head_weights) | ((uintptr_t)head_bias)) & 15) return NRT_ERR_INVALID_ARG;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;
    if (!up2_head_ok(a, labels)) return NRT_ERR_UNSUPPORTED;
    const U2Head head = {head_weights, head_bias, labels};
    hipStream_t st = nrt_stream(stream);
    return labels == 16 ? launch_up2<1, 1>(a, packed_weights, batch, st, head) : launch_up2<1, 2>(a, packed_weights, batch, st, head);
}

extern "C" int nrt_space_to_depth2_f32(const float *x, float *y, int batch, const int *shape, int channels, void *stream) {
    if (!x || !y || !shape || batch < 1 || batch > 65535 || channels < 4 || channels % 4) return NRT_ERR_INVALID_ARG;
    if (shape[0] < 2 || shape[1] < 2 || shape[2] < 2 || shape[0] % 2 || shape[1] % 2 || shape[2] % 2) return NRT_ERR_INVALID_ARG;
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return NRT_ERR_UNSUPPORTED;
    const long long total = (long long)shape[0] * shape[1] * shape[2] * (channels / 4);
    long long nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(space_to_depth2, dim3((unsigned)nb, batch), dim3(256), 0, nrt_stream(stream), (const f32x4 *)x, (f32x4 *)y,
                       shape[0] / 2, shape[1] / 2, shape[2] / 2, channels / 4);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_conv3d_s2d_taps_f32(const float *x, int group, const float *packed_weights, float *out, int batch,
                                       const int *shape, int cout, void *stream) {
    ConvArgs a;
    const int k3[3] = {3, 3, 3};
    if (group < 16 || group % 16 || !packed_weights) return NRT_ERR_INVALID_ARG;
    int rc = conv_args(a, x, 8 * group, nullptr, 0, nullptr, nullptr, out, shape, k3, cout, 1, 1, ACT_NONE);
    if (rc != NRT_OK) return rc;
    if (batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (!mfma_ok(a, 1)) return NRT_ERR_UNSUPPORTED;
    a.fold = group;
    hipStream_t st = nrt_stream(stream);
    return dispatch_mfma(a, packed_weights, batch, st);
}

extern "C" int nrt_conv3d_f32(const float *src0, int c0, const float *src1, int c1, const int *up, const float *weights,
                              const float *packed_weights, const float *bias, float *out, int batch, const int *shape,
                              const int *ksize, int cout, int dilation, int padding_same, int activation, int variant,
                              void *stream) {
    ConvArgs a;
    int rc = conv_args(a, src0, c0, src1, c1, up, bias, out, shape, ksize, cout, dilation, padding_same, activation);
    if (rc != NRT_OK) return rc;
    if (batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;
    hipStream_t st = nrt_stream(stream);
    const bool can_mfma = mfma_ok(a, padding_same) && packed_weights != nullptr;
    // variant 5: the persistent LDS-DMA schedule (conv_p27.h); auto takes it when there is more than one tile per CU
    const bool can_p27 = can_mfma && p27_ok(a, padding_same, batch);
    if (variant == 0 && can_p27 &&
        (long long)batch * ((a.OX + CT_X - 1) / CT_X) * ((a.OY + CT_Y - 1) / CT_Y) * ((a.OZ + CT_Z - 1) / CT_Z) >= 2ll * nrt_num_cus())
        variant = 5;
    if (variant == 5) {
        if (!can_p27) return NRT_ERR_UNSUPPORTED;
        switch ((cout + 15) / 16) {
            case 1: return launch_p27<1>(a, packed_weights, batch, st);
            case 2: return launch_p27<2>(a, packed_weights, batch, st);
            case 3: return launch_p27<3>(a, packed_weights, batch, st);
            default: return launch_p27<4>(a, packed_weights, batch, st);
        }
    }
    if (variant == 0) variant = can_mfma ? 2 : 1;
    if (variant == 2) {
        if (!can_mfma) return NRT_ERR_UNSUPPORTED;
        return dispatch_mfma(a, packed_weights, batch, st);
    }
    if ((variant != 1 && variant != 3) || !weights) return NRT_ERR_INVALID_ARG;
    const long long nvox = (long long)a.OX * a.OY * a.OZ;
    const int Gc = cout / 4;
    const bool c1_ok = (c0 == 1 && c1 == 0 && padding_same && cout % 4 == 0 && (Gc == 1 || Gc == 2 || Gc == 4 || Gc == 8 || Gc == 16) &&
                        (((uintptr_t)out) & 15) == 0);
    if (variant == 3 && !c1_ok) return NRT_ERR_UNSUPPORTED;
    const bool c1_mfma = c0 == 1 && c1 == 0 && padding_same && a.kx == 3 && a.ky == 3 && a.kz == 3 && a.dil == 1 &&
                         cout % 16 == 0 && cout <= 64 && (((uintptr_t)out) & 15) == 0;
    if (variant == 1 && c1_mfma) {                         // matrix-core form of the single-channel first layer
        const unsigned nbx = (a.OX + 3) / 4, nby = (a.OY + 3) / 4, nbz = (a.OZ + 15) / 16;
        const unsigned nblk = nbx * nby * nbz;
        const unsigned T8c = (nblk + NRT_NXCD - 1) / NRT_NXCD, perx = 8u * (unsigned)nrt_num_cus() / NRT_NXCD;     // 8 persistent blocks per CU
        dim3 grid(NRT_NXCD * (T8c < perx ? T8c : perx), batch);
        switch (cout / 16) {
            case 1: hipLaunchKernelGGL((conv3d_c1_mfma<1>), grid, dim3(256), 0, st, a, weights, nby, nbz, nblk, (float *)nullptr); break;
            case 2: hipLaunchKernelGGL((conv3d_c1_mfma<2>), grid, dim3(256), 0, st, a, weights, nby, nbz, nblk, (float *)nullptr); break;
            case 3: hipLaunchKernelGGL((conv3d_c1_mfma<3>), grid, dim3(256), 0, st, a, weights, nby, nbz, nblk, (float *)nullptr); break;
            default: hipLaunchKernelGGL((conv3d_c1_mfma<4>), grid, dim3(256), 0, st, a, weights, nby, nbz, nblk, (float *)nullptr); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    if (c1_ok && (variant == 3 || variant == 1)) {        // variant 1 with Cin == 1 takes the vectorised form too
        const size_t shm = (size_t)(a.kx * a.ky * a.kz + 1) * cout * sizeof(float);
        const unsigned ng = 256 / Gc;
        unsigned blocks = (unsigned)((nvox + ng - 1) / ng);
        if (blocks > 256u * 32u) blocks = 256u * 32u;
        dim3 grid(blocks, batch);
        switch (Gc) {
            case 1: hipLaunchKernelGGL((conv3d_c1_vec<1>), grid, dim3(256), shm, st, a, weights); break;
            case 2: hipLaunchKernelGGL((conv3d_c1_vec<2>), grid, dim3(256), shm, st, a, weights); break;
            case 4: hipLaunchKernelGGL((conv3d_c1_vec<4>), grid, dim3(256), shm, st, a, weights); break;
            case 8: hipLaunchKernelGGL((conv3d_c1_vec<8>), grid, dim3(256), shm, st, a, weights); break;
            default: hipLaunchKernelGGL((conv3d_c1_vec<16>), grid, dim3(256), shm, st, a, weights); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    unsigned blocks = (unsigned)((nvox + 255) / 256);
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(conv3d_direct, dim3(blocks, batch), dim3(256), 0, st, a, weights);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// the single-channel first layer + the 2x2x2 max-pooling behind it (models.py:1378-1388, 1436-1438) in one kernel: `out` as nrt_conv3d_f32
// writes it, `pool_out` [batch, shape / 2, cout] in addition.  shape a multiple of (4, 4, 16), cout 16 or 32, 3x3x3 SAME.
extern "C" int nrt_conv3d_c1_pool_supported(const int *shape, int cout) {
    return shape && shape[0] > 0 && shape[1] > 0 && shape[2] > 0 && shape[0] % 4 == 0 && shape[1] % 4 == 0 && shape[2] % 16 == 0 &&
           (cout == 16 || cout == 32) ? 1 : 0;
}
extern "C" int nrt_conv3d_c1_pool_f32(const float *src, const float *weights, const float *bias, float *out, float *pool_out, int batch,
                                      const int *shape, int cout, int activation, void *stream) {
    if (!pool_out || !weights || !nrt_conv3d_c1_pool_supported(shape, cout)) return NRT_ERR_UNSUPPORTED;
    ConvArgs a;
    const int k3[3] = {3, 3, 3};
    int rc = conv_args(a, src, 1, nullptr, 0, nullptr, bias, out, shape, k3, cout, 1, 1, activation);
    if (rc != NRT_OK) return rc;
    if (batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;
    if ((((uintptr_t)out) | ((uintptr_t)pool_out)) & 15) return NRT_ERR_INVALID_ARG;
    hipStream_t st = nrt_stream(stream);
    const unsigned nbx = (a.OX + 3) / 4, nby = (a.OY + 3) / 4, nbz = (a.OZ + 15) / 16;
    const unsigned nblk = nbx * nby * nbz;
    const unsigned T8c = (nblk + NRT_NXCD - 1) / NRT_NXCD, perx = 8u * (unsigned)nrt_num_cus() / NRT_NXCD;
    dim3 grid(NRT_NXCD * (T8c < perx ? T8c : perx), batch);
    if (cout == 16) hipLaunchKernelGGL((conv3d_c1_mfma<1, true>), grid, dim3(256), 0, st, a, weights, nby, nbz, nblk, pool_out);
    else hipLaunchKernelGGL((conv3d_c1_mfma<2, true>), grid, dim3(256), 0, st, a, weights, nby, nbz, nblk, pool_out);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// A 3x3x3 'same' convolution over 16 k input channels AND the MaxPooling3D(2) of its activated output from one kernel (conv_p27.h, POOL):
// the encoder pattern of models.py:1378-1388 + 1436-1438 below the first level (whose single-channel form is nrt_conv3d_c1_pool_f32)
extern "C" int nrt_conv3d_pool_supported(int c0, int cout, const int *shape, int batch) {
    if (!shape || batch < 1 || batch > 65535) return 0;
    ConvArgs a;
    const int k3[3] = {3, 3, 3};
    float dummy;
    if (conv_args(a, &dummy, c0, nullptr, 0, nullptr, nullptr, &dummy, shape, k3, cout, 1, 1, ACT_NONE) != NRT_OK) return 0;
    return mfma_ok(a, 1) && p27_pool_ok(a, batch) ? 1 : 0;
}
extern "C" int nrt_conv3d_pool_f32(const float *src, int c0, const float *packed_weights, const float *bias, float *out, float *pool_out,
                                   int batch, const int *shape, int cout, int activation, void *stream) {
    if (!pool_out || !packed_weights) return NRT_ERR_INVALID_ARG;
    ConvArgs a;
    const int k3[3] = {3, 3, 3};
    int rc = conv_args(a, src, c0, nullptr, 0, nullptr, bias, out, shape, k3, cout, 1, 1, activation);
    if (rc != NRT_OK) return rc;
    if (batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;
    if ((((uintptr_t)out) | ((uintptr_t)pool_out)) & 15) return NRT_ERR_INVALID_ARG;
    if (!mfma_ok(a, 1) || !p27_pool_ok(a, batch)) return NRT_ERR_UNSUPPORTED;
    hipStream_t st = nrt_stream(stream);
    switch (cout / 16) {
        case 2: return launch_p27<2, true>(a, packed_weights, batch, st, pool_out);
        case 3: return launch_p27<3, true>(a, packed_weights, batch, st, pool_out);
        default: return launch_p27<4, true>(a, packed_weights, batch, st, pool_out);
    }
}

extern "C" int nrt_conv1x1_softmax_f32(const float *x, const float *weights, const float *bias, float *y,
                                       long long nvox, int cin, int cout, int softmax, int activation, void *stream) {
    if (!x || !weights || !y || nvox < 0 || cin < 1 || cout < 1) return NRT_ERR_INVALID_ARG;
    if (cout > 64 || (size_t)(cin + 1) * cout * sizeof(float) > 64 * 1024) return NRT_ERR_UNSUPPORTED;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;   // the kernels fuse none / elu / relu only
    if (nvox == 0) return NRT_OK;
    unsigned blocks = (unsigned)((nvox + 255) / 256);
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    const size_t shm = (size_t)(cin + 1) * cout * sizeof(float);
    hipStream_t st = nrt_stream(stream);
    const int G = cout / 4;
    if (cout % 4 == 0 && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16) && (((uintptr_t)y) & 15) == 0) {
        const unsigned ng = 256 / G;
        unsigned vb = (unsigned)((nvox + ng - 1) / ng);
        if (vb > 256u * 32u) vb = 256u * 32u;
        const bool rows16 = (((uintptr_t)x) & 15) == 0;
        if ((cin == 16 || cin == 32) && rows16 && (G == 4 || G == 8 || G == 16) && ((((uintptr_t)weights) | ((uintptr_t)bias)) & 15) == 0) {
            // the likelihood layer of the unets (16 features -> 16 / 32 / 64 labels) and its input gradient (32 -> 16)
            unsigned rb = (unsigned)((nvox + 255) / 256);
            if (rb > 256u * 5u) rb = 256u * 5u;
#define NRT_ROWS(GG, CC) hipLaunchKernelGGL((conv1x1_rows<GG, CC>), dim3(rb), dim3(256), 0, st, x, weights, bias, y, nvox, softmax, activation)
            if (cin == 16) { if (G == 4) NRT_ROWS(4, 16); else if (G == 8) NRT_ROWS(8, 16); else NRT_ROWS(16, 16); }
            else { if (G == 4) NRT_ROWS(4, 32); else if (G == 8) NRT_ROWS(8, 32); else NRT_ROWS(16, 32); }
#undef NRT_ROWS
            NRT_CHECK_LAUNCH();
            return NRT_OK;
        }
        switch (G) {
            case 1: hipLaunchKernelGGL((conv1x1_vec<1>), dim3(vb), dim3(256), shm, st, x, weights, bias, y, nvox, cin, softmax, activation); break;
            case 2: hipLaunchKernelGGL((conv1x1_vec<2>), dim3(vb), dim3(256), shm, st, x, weights, bias, y, nvox, cin, softmax, activation); break;
            case 4: hipLaunchKernelGGL((conv1x1_vec<4>), dim3(vb), dim3(256), shm, st, x, weights, bias, y, nvox, cin, softmax, activation); break;
            case 8: hipLaunchKernelGGL((conv1x1_vec<8>), dim3(vb), dim3(256), shm, st, x, weights, bias, y, nvox, cin, softmax, activation); break;
            default: hipLaunchKernelGGL((conv1x1_vec<16>), dim3(vb), dim3(256), shm, st, x, weights, bias, y, nvox, cin, softmax, activation); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    if (cout <= 16) hipLaunchKernelGGL((conv1x1_softmax<16>), dim3(blocks), dim3(256), shm, st, x, weights, bias, y, nvox, cin, cout, softmax, activation);
    else if (cout <= 32) hipLaunchKernelGGL((conv1x1_softmax<32>), dim3(blocks), dim3(256), shm, st, x, weights, bias, y, nvox, cin, cout, softmax, activation);
    else hipLaunchKernelGGL((conv1x1_softmax<64>), dim3(blocks), dim3(256), shm, st, x, weights, bias, y, nvox, cin, cout, softmax, activation);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_softmax_lastdim_f32(const float *x, float *y, long long n, int channels, void *stream) {
    if (!x || !y || n < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    const int G = channels / 4;
    if (channels % 4 == 0 && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0)) {
        const long long per = 256 / G;
        unsigned vb = (unsigned)((n + per - 1) / per);
        if (vb > 256u * 16u) vb = 256u * 16u;
        hipStream_t st = nrt_stream(stream);
        switch (G) {
            case 1: hipLaunchKernelGGL((softmax_lastdim_vec<1>), dim3(vb), dim3(256), 0, st, (const f32x4 *)x, (f32x4 *)y, n); break;
            case 2: hipLaunchKernelGGL((softmax_lastdim_vec<2>), dim3(vb), dim3(256), 0, st, (const f32x4 *)x, (f32x4 *)y, n); break;
            case 4: hipLaunchKernelGGL((softmax_lastdim_vec<4>), dim3(vb), dim3(256), 0, st, (const f32x4 *)x, (f32x4 *)y, n); break;
            case 8: hipLaunchKernelGGL((softmax_lastdim_vec<8>), dim3(vb), dim3(256), 0, st, (const f32x4 *)x, (f32x4 *)y, n); break;
            default: hipLaunchKernelGGL((softmax_lastdim_vec<16>), dim3(vb), dim3(256), 0, st, (const f32x4 *)x, (f32x4 *)y, n); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    hipLaunchKernelGGL(softmax_lastdim, dim3(blocks), dim3(256), 0, nrt_stream(stream), x, y, n, channels);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_maxpool3d_f32(const float *x, float *y, int batch, const int *shape, int channels, const int *pool,
                                 int padding_same, void *stream) {
    if (!x || !y || !shape || !pool || batch < 1 || batch > 65535 || channels < 1) return NRT_ERR_INVALID_ARG;
    for (int d = 0; d < 3; ++d) if (pool[d] < 1 || shape[d] < 1) return NRT_ERR_INVALID_ARG;
    int o[3];
    for (int d = 0; d < 3; ++d) o[d] = padding_same ? (shape[d] + pool[d] - 1) / pool[d] : shape[d] / pool[d];
    const long long total = (long long)o[0] * o[1] * o[2] * channels;
    if (total == 0) return NRT_OK;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(maxpool3d, dim3(blocks, batch), dim3(256), 0, nrt_stream(stream), x, y, shape[0], shape[1], shape[2],
                       channels, o[0], o[1], o[2], pool[0], pool[1], pool[2]);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_upsample_concat_f32(const float *skip, int c0, const float *lo, int c1, float *y, int batch,
                                       const int *shape, const int *up, void *stream) {
    if (!lo || !y || !shape || !up || batch < 1 || batch > 65535 || c0 < 0 || c1 < 1) return NRT_ERR_INVALID_ARG;
    if (c0 > 0 && !skip) return NRT_ERR_INVALID_ARG;
    for (int d = 0; d < 3; ++d) if (up[d] < 1 || shape[d] < 1 || shape[d] % up[d]) return NRT_ERR_INVALID_ARG;
    const long long total = (long long)shape[0] * shape[1] * shape[2] * (c0 + c1);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(upsample_concat, dim3(blocks, batch), dim3(256), 0, nrt_stream(stream), c0 > 0 ? skip : nullptr, c0, lo,
                       c1, y, shape[0], shape[1], shape[2], up[0], up[1], up[2]);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_add_act_affine_f32(const float *a, const float *b, const float *scale, const float *shift, float *y,
                                      long long n, int channels, int activation, void *stream) {
    if (!a || !y || n < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return NRT_ERR_INVALID_ARG;
    if ((activation & 0xff) > ACT_LAST || (activation & ~(0xff | ACT_MUL_B)) || activation < 0) return NRT_ERR_INVALID_ARG;
    if ((activation & ACT_MUL_B) && !b) return NRT_ERR_INVALID_ARG;
    if (n == 0) return NRT_OK;
    const bool v4 = n % 4 == 0 && (!scale || channels % 4 == 0) && channels >= 1 &&
                    ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0;
    if (v4) {
        const long long n4 = n / 4;
        unsigned blocks4 = (unsigned)((n4 + 255) / 256 < 256 * 16 ? (n4 + 255) / 256 : 256 * 16);
        hipLaunchKernelGGL(add_act_affine_v4, dim3(blocks4), dim3(256), 0, nrt_stream(stream), (const f32x4 *)a, (const f32x4 *)b, scale,
                           shift, (f32x4 *)y, n4, scale ? channels / 4 : 1, activation);
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(add_act_affine, dim3(blocks), dim3(256), 0, nrt_stream(stream), a, b, scale, shift, y, n, channels,
                       activation);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
