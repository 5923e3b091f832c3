// interpn, linear, 3-D, 1..4 channels, per-voxel locations: the wave-autonomous LDS-staged kernel (variant 9).
//
// What limits the few-channel warps is the number of LANE accesses the texture-address unit has to serve (DESIGN.md 4.1.3:
// ~1 clk per lane of a gather whose lanes are not consecutive, ~0.25 clk per lane of a consecutive 16-byte access).  The lean
// tile kernel (variant 8) spends 4 (C <= 2) or 8 scattered lane accesses per voxel.  Here EVERY global access is a 16-byte
// access of consecutive lanes, and no wave ever waits for another one:
//   * a WAVE owns a 2 x 2 x 16 sub-tile (x, y, z; lane = (xx * 2 + yy) * 16 + zz) and walks a list of tiles (persistent waves;
//     tiles are dealt so that the waves of one XCD work on one contiguous run of the volume);
//   * the tile's locations (4 rows of 192 bytes) arrive as one 16-byte buffer load per lane (lanes 0..47), already requested
//     while the previous tile was computed, and are redistributed through the wave's private LDS;
//   * every lane does its voxel's corner arithmetic; the wave stages a FIXED 5 x 5 x 24 box of the source volume whose origin
//     is derived from lane 0 (corner index minus 1 / 1 / 3): 150 * C consecutive-lane 16-byte buffer loads whose per-lane
//     offsets are loop invariants (the tile's origin is a scalar offset), reads past the end of the volume return zero and are
//     never used;
//   * a lane whose 8 corners lie inside the box (every lane, for a deformation whose displacement changes by less than about
//     1 / 1 / 3 voxels across the tile) reads them from LDS; any other lane gathers them from global memory as variant 8 does;
//   * blend in the reference's op order (interpn_generic's, one rounding per op), results -> LDS -> 16-byte buffer stores.
// There is no __syncthreads: a wave's LDS operations execute in order.  Bit-identical to the generic kernel and the oracle
// (tests/test_gpu_interpn.py::test_lds2_kernel).

#include "interpn_core.h"
#include "lean.h"
#include "lean_core.h"

namespace {

typedef unsigned l2_u32x4 __attribute__((ext_vector_type(4)));

template <int C> struct Lds2Cfg {
    static constexpr int EX = 5, EY = 5, EZ = 24;          // voxels of the staged box
    static constexpr int MX = 1, MY = 1, MZ = 3;           // box origin = lane 0's lower corner minus these
    static constexpr int RS = EZ * C;                       // floats of a box row (a multiple of 4)
    static constexpr int QPR = RS / 4;                      // 16-byte chunks per box row
    static constexpr int NQ = EX * EY * QPR;                // chunks of the box
    static constexpr int NLD = (NQ + 63) / 64;              // chunk loads per lane
    static constexpr int IOF = 64 * (C > 3 ? C : 3);        // floats of the I/O area (locations, then results)
    static constexpr int WF = IOF + EX * EY * RS;           // floats of LDS per wave
};

struct Lds2Geom {
    unsigned nTy, nTz, ntiles;       // tiles along y / z, in total (per batch element)
    unsigned m_ty, m_tz;             // ceil(2^32 / n) for the two divisions (0: n == 1)
    unsigned vol_bytes, loc_bytes, out_bytes;
};

__device__ __forceinline__ unsigned l2_div(unsigned n, unsigned d, unsigned m) { return d == 1u ? n : __umulhi(n, m); }

template <int C, int MODE>
__global__ __launch_bounds__(256) void interpn_lds2(InterpArgs a, Lds2Geom g) {
    typedef Lds2Cfg<C> K;
    extern __shared__ __attribute__((aligned(16))) float l2_lds[];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wib = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *io = l2_lds + wib * (unsigned)K::WF;
    float *box = io + K::IOF;
    const int b = blockIdx.y;
    const char *vol = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const int O0 = a.O[0], O1 = a.O[1], O2 = a.O[2];
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, (int)g.vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t lr =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.loc + (long long)b * a.loc_bs), 0, (int)g.loc_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orr =
        __builtin_amdgcn_make_buffer_rsrc((void *)((float *)a.out + (long long)b * a.out_bs), 0, (int)g.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.addend ? a.addend + (long long)b * a.addend_bs : a.loc), 0, a.addend ? (int)g.out_bytes : 0, 0x00020000);

    // ---- loop invariants of this lane ---------------------------------------------------------------------------------------
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];
    const unsigned line_b = SZ * (unsigned)(C * 4), plane_b = SY * line_b;
    const unsigned orow_q = (unsigned)O2, oplane_q = (unsigned)O1 * (unsigned)O2;      // voxels per output line / plane
    // locations: lane l < 48 reads chunk l % 12 of row l / 12 (row = xx * 2 + yy)
    const unsigned lrow = lane / 12u, lcol = lane - lrow * 12u;
    const unsigned loc_vo = ((lrow >> 1) * oplane_q + (lrow & 1u) * orow_q) * 12u + lcol * 16u;
    // results: lane l < 16 C writes chunk l % (4 C) of row l / (4 C)
    const unsigned srow = lane / (4u * C), scol = lane - srow * (4u * C);
    const unsigned out_vo = ((srow >> 1) * oplane_q + (srow & 1u) * orow_q) * (unsigned)(C * 4) + scol * 16u;
    // box chunks lane + 64 k
    unsigned box_vo[K::NLD], box_ld[K::NLD];
#pragma unroll
    for (int k = 0; k < K::NLD; ++k) {
        const unsigned s = min(lane + 64u * (unsigned)k, (unsigned)(K::NQ - 1));
        const unsigned row = s / (unsigned)K::QPR, col = s - row * (unsigned)K::QPR;
        const unsigned bx = row / (unsigned)K::EY, by = row - bx * (unsigned)K::EY;
        box_vo[k] = bx * plane_b + by * line_b + col * 16u;
        box_ld[k] = row * (unsigned)K::RS + col * 4u;
    }
    // this lane's voxel inside the tile
    const unsigned vx = lane >> 5, vy = (lane >> 4) & 1u, vz = lane & 15u;
    const float mxx = (float)(a.S[0] - 1), mxy = (float)(a.S[1] - 1), mxz = (float)(a.S[2] - 1);

    // ---- the tiles of this wave: in every round the waves of one XCD take one contiguous run of tiles -----------------------
    const unsigned per_xcd = gridDim.x >> 3;                                  // gridDim.x is a multiple of 8
    const unsigned slot = ((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * 4u + wib;
    const unsigned nw = gridDim.x * 4u;
    unsigned tile = slot;
    if (tile >= g.ntiles) return;

    auto origin = [&](unsigned tl, int &x0, int &y0, int &z0) {
        const unsigned t2 = l2_div(tl, g.nTz, g.m_tz);
        const unsigned tx = l2_div(t2, g.nTy, g.m_ty);
        z0 = (int)((tl - t2 * g.nTz) << 4);
        y0 = (int)((t2 - tx * g.nTy) << 1);
        x0 = (int)(tx << 1);
    };
    auto load_loc = [&](unsigned tl) -> l2_u32x4 {
        int x0, y0, z0;
        origin(tl, x0, y0, z0);
        const unsigned so = (((unsigned)x0 * (unsigned)O1 + (unsigned)y0) * (unsigned)O2 + (unsigned)z0) * 12u;
        return __builtin_amdgcn_raw_buffer_load_b128(lr, loc_vo, so, 0);      // rows past the last plane: out of range -> 0
    };

    l2_u32x4 lq = load_loc(tile);
    for (;;) {
        int x0, y0, z0;
        origin(tile, x0, y0, z0);
        const unsigned next = tile + nw;
        l2_u32x4 lq_next = lq;
        if (next < g.ntiles) lq_next = load_loc(next);
        // ---- 1. locations through LDS: chunk l of the tile's 48 lands at floats [4 l, 4 l + 4) ------------------------------
        if (lane < 48u) *(l2_u32x4 *)(io + lane * 4u) = lq;
        __builtin_amdgcn_wave_barrier();
        const bool valid = x0 + (int)vx < O0 && y0 + (int)vy < O1;           // z is always inside: the z extent is a multiple of 16
        // lanes of rows outside the volume take the tile's first voxel (always inside): they stay inside the box
        const float *lp = io + (valid ? (lane >> 4) * 48u + vz * 3u : 0u);
        float p[3] = {lp[0], lp[1], lp[2]};
        if (MODE == NRT_LOC_SHIFT) {
            p[0] = nrt_add((float)(valid ? x0 + (int)vx : x0), p[0]);
            p[1] = nrt_add((float)(valid ? y0 + (int)vy : y0), p[1]);
            p[2] = nrt_add((float)(valid ? z0 + (int)vz : z0), p[2]);
        }
        int ix, iy, iz, ux, uy, uz;
        float w0x, w1x, w0y, w1y, w0z, w1z;
        lean_corner(p[0], mxx, a.S[0] - 1, ix, ux, w0x, w1x);
        lean_corner(p[1], mxy, a.S[1] - 1, iy, uy, w0y, w1y);
        lean_corner(p[2], mxz, a.S[2] - 1, iz, uz, w0z, w1z);
        // ---- 2. the box: origin from lane 0 (scalar), 16-byte chunks of consecutive lanes -------------------------------------
        const int lox = max(__builtin_amdgcn_readfirstlane(ix) - K::MX, 0);
        const int loy = max(__builtin_amdgcn_readfirstlane(iy) - K::MY, 0);
        const int loz = max(__builtin_amdgcn_readfirstlane(iz) - K::MZ, 0);
        const unsigned sbase = (((unsigned)lox * SY + (unsigned)loy) * SZ + (unsigned)loz) * (unsigned)(C * 4);
        l2_u32x4 q[K::NLD];
#pragma unroll
        for (int k = 0; k < K::NLD; ++k) q[k] = __builtin_amdgcn_raw_buffer_load_b128(vr, box_vo[k], sbase, 0);
#pragma unroll
        for (int k = 0; k < K::NLD; ++k)
            if ((k + 1) * 64 <= K::NQ || lane + 64u * (unsigned)k < (unsigned)K::NQ) *(l2_u32x4 *)(box + box_ld[k]) = q[k];
        __builtin_amdgcn_wave_barrier();
        // ---- 3. corners and blend (the rounding sequence of interpn_generic) ------------------------------------------------
        const unsigned dx = (unsigned)(ix - lox), dy = (unsigned)(iy - loy), dz = (unsigned)(iz - loz);
        const bool inbox = dx <= (unsigned)(K::EX - 2) && dy <= (unsigned)(K::EY - 2) && dz <= (unsigned)(K::EZ - 2);
        const float wxy[4] = {nrt_mul(w0x, w0y), nrt_mul(w0x, w1y), nrt_mul(w1x, w0y), nrt_mul(w1x, w1y)};
        float v[8][C];
        if (inbox) {
            const float *pb = box + (dx * (unsigned)K::EY + dy) * (unsigned)K::RS + dz * (unsigned)C;
            const unsigned sx = ux ? (unsigned)(K::EY * K::RS) : 0u, sy = uy ? (unsigned)K::RS : 0u;
#pragma unroll
            for (int xy = 0; xy < 4; ++xy) {
                const float *pv = pb + ((xy & 2) ? sx : 0u) + ((xy & 1) ? sy : 0u);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float lo = pv[c], hi = pv[C + c];           // the z neighbour is inside the box (dz <= EZ - 2)
                    v[xy * 2][c] = lo;
                    v[xy * 2 + 1][c] = uz ? hi : lo;
                }
            }
        } else {
            const unsigned base = nrt_mad24(nrt_mad24((unsigned)ix, SY, (unsigned)iy), SZ, (unsigned)iz) * (unsigned)(C * 4);
            const unsigned sx = ux ? plane_b : 0u, sy = uy ? line_b : 0u, sz = uz ? (unsigned)(C * 4) : 0u;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner)
                load_c<C>(vol, base + ((corner & 4) ? sx : 0u) + ((corner & 2) ? sy : 0u) + ((corner & 1) ? sz : 0u), v[corner]);
        }
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0f;                           // :160
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float wt = nrt_mul(wxy[corner >> 1], (corner & 1) ? w1z : w0z);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = nrt_add(acc[c], nrt_mul(wt, v[corner][c]));         // :191
        }
        if (a.has_fill) {
            const bool oob = (p[0] < 0.0f) || (p[0] > mxx) || (p[1] < 0.0f) || (p[1] > mxy) || (p[2] < 0.0f) || (p[2] > mxz);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = apply_fill(acc[c], oob, a.fill_f);
        }
        // ---- 4. results -> LDS -> 16-byte stores: row r of the tile is floats [16 C r, 16 C (r + 1)) --------------------------
#pragma unroll
        for (int c = 0; c < C; ++c) io[lane * (unsigned)C + (unsigned)c] = acc[c];
        __builtin_amdgcn_wave_barrier();
        if (lane < 16u * C && x0 + (int)(srow >> 1) < O0 && y0 + (int)(srow & 1u) < O1) {
            const unsigned so = (((unsigned)x0 * (unsigned)O1 + (unsigned)y0) * (unsigned)O2 + (unsigned)z0) * (unsigned)(C * 4);
            nrt_f4 o = *(const nrt_f4 *)(io + lane * 4u);
            if (a.addend) {
                const nrt_f4 ad = __builtin_bit_cast(nrt_f4, __builtin_amdgcn_raw_buffer_load_b128(ar, out_vo, so, 0));
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = nrt_add(ad[k], o[k]);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(l2_u32x4, o), orr, out_vo, so, 2);      // nt
        }
        __builtin_amdgcn_wave_barrier();
        if (next >= g.ntiles) break;
        tile = next;
        lq = lq_next;
    }
}

template <int C>
void launch_lds2(const InterpArgs &a, int batch, int mode, hipStream_t st) {
    typedef Lds2Cfg<C> K;
    Lds2Geom g;
    const unsigned nTx = (a.O[0] + 1) >> 1;
    g.nTy = (a.O[1] + 1) >> 1;
    g.nTz = (unsigned)a.O[2] >> 4;
    g.ntiles = nTx * g.nTy * g.nTz;
    g.m_ty = g.nTy == 1 ? 0u : (unsigned)(0x100000000ull / g.nTy) + 1u;
    g.m_tz = g.nTz == 1 ? 0u : (unsigned)(0x100000000ull / g.nTz) + 1u;
    g.vol_bytes = (unsigned)((unsigned long long)a.S[0] * a.S[1] * a.S[2] * C * 4ull);
    g.loc_bytes = (unsigned)((unsigned long long)a.nout * 12ull);
    g.out_bytes = (unsigned)((unsigned long long)a.nout * C * 4ull);
    const size_t dyn = (size_t)4 * K::WF * sizeof(float);
    // resident blocks: 32 waves or 160 KB of LDS per CU, 256 CUs, shared by the batch
    static const int tune_bpc = getenv("NRT_LDS2_BPC") ? atoi(getenv("NRT_LDS2_BPC")) : 0;
    unsigned bpc = (unsigned)((160u * 1024u) / dyn);
    if (bpc > 8u) bpc = 8u;
    if (tune_bpc > 0 && (unsigned)tune_bpc < bpc) bpc = (unsigned)tune_bpc;
    unsigned gx = (256u * bpc + (unsigned)batch - 1u) / (unsigned)batch;
    const unsigned need = (g.ntiles + 3u) / 4u;
    if (gx > need) gx = need;
    gx = (gx + 7u) & ~7u;
    dim3 grid(gx, batch), blk(256);
    if (mode == NRT_LOC_ABSOLUTE) hipLaunchKernelGGL((interpn_lds2<C, NRT_LOC_ABSOLUTE>), grid, blk, dyn, st, a, g);
    else hipLaunchKernelGGL((interpn_lds2<C, NRT_LOC_SHIFT>), grid, blk, dyn, st, a, g);
}

}  // namespace

// per-voxel locations only; z extent a multiple of 16 (whole 16-byte chunks per tile row), 16-byte aligned tensors and batch
// strides, sizes inside the 31-bit buffer ranges and the exact range of the multiply-high divisions
bool nrt_lds2_supported(const int *vol_shape, const int *out_shape, int channels, int ndim, const void *vol, const void *loc,
                        const void *out, const void *addend, long long vol_bs, long long loc_bs, long long addend_bs, int loc_mode) {
    if (ndim != 3 || channels < 1 || channels > 4 || loc_mode == NRT_LOC_LINSPACE || !loc) return false;
    unsigned long long vbytes = 4ull * channels, nout = 1;
    for (int d = 0; d < 3; ++d) {
        if (vol_shape[d] < 1 || out_shape[d] < 1 || vol_shape[d] >= (1 << 12) || out_shape[d] >= (1 << 12)) return false;
        vbytes *= (unsigned long long)vol_shape[d];
        nout *= (unsigned long long)out_shape[d];
    }
    if (vbytes >= (1ull << 31) || nout * 16ull >= (1ull << 31)) return false;
    if (out_shape[2] % 16 != 0) return false;
    // umulhi(n, 2^32 / d + 1) == n / d needs n * d < 2^32
    const unsigned long long nTy = (out_shape[1] + 1) / 2, nTz = out_shape[2] / 16, nT = (unsigned long long)((out_shape[0] + 1) / 2) * nTy * nTz;
    if (nT * (nTy > nTz ? nTy : nTz) >= (1ull << 32)) return false;
    if ((((uintptr_t)vol | (uintptr_t)out | (uintptr_t)loc | (uintptr_t)addend) & 15) != 0) return false;
    if ((vol_bs * 4) % 16 != 0 || (loc_bs * 4) % 16 != 0 || (addend_bs * 4) % 16 != 0) return false;
    return true;
}

int nrt_lds2_launch(const void *args, int batch, int mode, void *stream) {
    const InterpArgs &a = *(const InterpArgs *)args;
    hipStream_t st = nrt_stream(stream);
    switch (a.C) {
        case 1: launch_lds2<1>(a, batch, mode, st); break;
        case 2: launch_lds2<2>(a, batch, mode, st); break;
        case 3: launch_lds2<3>(a, batch, mode, st); break;
        default: launch_lds2<4>(a, batch, mode, st); break;
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
