// interpn, linear, 3-D, 1..4 channels (images, displacement fields): the instruction-lean kernel (variant 8).
//
// With 4..16 bytes per voxel the warp of an image or of a flow field (SpatialTransformer on C = 1 images, Resize(2) of the
// C = 3 deformation at neurite/tf/models.py:804, VecInt / compose) is not bound by HBM but by instruction issue: the
// LDS-staged kernel (variant 6) spends ~450 issue slots per voxel on bounding boxes, the box copy with its divisions and
// 64-bit addresses (0.23-0.25 of the HBM roof at 160^3, profiles/archive/r01_session50).  Here nothing is staged: the source of a
// few-channel volume is small (16 MB at 160^3 x 1) and neighbouring lanes read neighbouring addresses, so the 8 corner
// reads are L1 / L2 hits; what is left is the reference's arithmetic, strength-reduced:
//   * a lane owns VPL consecutive-z output voxels (4 at C = 1, 2 at C = 2): x, y and their corner arithmetic are
//     computed once per lane, the location is read with 16-byte loads, the result written with one 16-byte store;
//   * blocks are laid out over (x-plane, chunk of the plane): the only division left is one multiply-high per lane;
//   * tf.clip_by_value is one v_med3_f32, every corner address a 32-bit byte offset from a uniform base (one add per
//     corner), channels of a corner one 4/8/12/16-byte load.
// Same float32 op sequence as interpn_generic (one rounding per op, corners in itertools.product order) => bit-identical
// results (tests/test_gpu_interpn.py); optional fill and addend epilogue (VecInt / compose).

#include <stdlib.h>

#include "interpn_core.h"
#include "lean.h"
#include "lean_core.h"

namespace {

template <int C, int VPL, int MODE>
__global__ __launch_bounds__(256) void interpn_lean(InterpArgs a, unsigned lpr, unsigned m_lpr, unsigned cpp, unsigned nblk) {
    const unsigned logical = nrt_xcd_block(blockIdx.x, gridDim.x);
    if (logical >= nblk) return;
    const unsigned x = logical / cpp, chunk = logical - x * cpp;          // uniform: scalar ALU
    // XPOSE (more than 16 contiguous bytes per lane, e.g. 3 channels x 4 voxels = 48): the results leave through LDS so that every store
    // instruction writes 16 consecutive bytes per lane -- stored straight from the lanes, the 16-byte pieces are 48 bytes apart,
    // every line is written by three instructions and Resize(2) of 4 x 80^3 x 3 ran at 0.089 ms instead of 0.072
    constexpr int NF = VPL * C;                                           // floats a lane produces
    constexpr bool XPOSE = NF > 4 && NF % 4 == 0;
    __shared__ __attribute__((aligned(16))) float s_out[XPOSE ? 256 * NF : 4];
    const unsigned glim = (unsigned)a.O[1] * lpr;
    unsigned g = chunk * 256u + threadIdx.x;                              // group of VPL voxels inside the x-plane
    const bool lane_valid = g < glim;
    if (!XPOSE && !lane_valid) return;
    if (!lane_valid) g = glim - 1u;                                       // XPOSE: the lane computes a repeat and stores nothing
    const unsigned y = lpr == 1u ? g : __umulhi(g, m_lpr);                // g / lpr, exact for g * lpr < 2^32 (checked on the host)
    const unsigned z0 = (g - y * lpr) * (unsigned)VPL;
    const int b = blockIdx.y;
    const char *vol = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    float *out = (float *)a.out + (long long)b * a.out_bs;
    const unsigned q0 = nrt_mad24(nrt_mad24(x, (unsigned)a.O[1], y), (unsigned)a.O[2], z0);

    // ---- locations of the VPL voxels -----------------------------------------------------------------------------
    float raw[VPL * 3];
    if (MODE != NRT_LOC_LINSPACE) {
        const float *lp = locb + (size_t)q0 * 3u;
        if constexpr (VPL == 4) {
            const nrt_f4 t0 = ((const nrt_f4 *)lp)[0], t1 = ((const nrt_f4 *)lp)[1], t2 = ((const nrt_f4 *)lp)[2];
            raw[0] = t0[0]; raw[1] = t0[1]; raw[2] = t0[2]; raw[3] = t0[3];
            raw[4] = t1[0]; raw[5] = t1[1]; raw[6] = t1[2]; raw[7] = t1[3];
            raw[8] = t2[0]; raw[9] = t2[1]; raw[10] = t2[2]; raw[11] = t2[3];
        } else if constexpr (VPL == 2) {
            const nrt_f2 t0 = ((const nrt_f2 *)lp)[0], t1 = ((const nrt_f2 *)lp)[1], t2 = ((const nrt_f2 *)lp)[2];
            raw[0] = t0[0]; raw[1] = t0[1]; raw[2] = t1[0]; raw[3] = t1[1]; raw[4] = t2[0]; raw[5] = t2[1];
        } else {
            raw[0] = lp[0]; raw[1] = lp[1]; raw[2] = lp[2];
        }
    }
    // ---- x and y: shared by the lane's voxels unless the location is per voxel ---------------------------------------
    const float mxx = (float)(a.S[0] - 1), mxy = (float)(a.S[1] - 1), mxz = (float)(a.S[2] - 1);
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];
    const unsigned str_x = SY * SZ * (unsigned)(C * 4), str_y = SZ * (unsigned)(C * 4), str_z = (unsigned)(C * 4);
    float res[VPL][C];
    auto linspace = [&](int d, int qv) -> float {        // tf.linspace(0, S - 1, O): the ends exact, delta * i between them
        return (qv == 0) ? 0.0f : ((qv == a.O[d] - 1) ? (float)(a.S[d] - 1) : nrt_mul(a.delta[d], (float)qv));
    };
    // regular grids: the lane's voxels share x and y -- their corner arithmetic, x*y weight products and row base are formed once
    // (written out: the compiler does not merge the copies of the unrolled loop)
    float spx = 0.0f, spy = 0.0f, sw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned sbase = 0u, ssx = 0u, ssy = 0u;
    if (MODE == NRT_LOC_LINSPACE) {
        spx = linspace(0, (int)x);
        spy = linspace(1, (int)y);
        int ix, iy, ux, uy;
        float w0x, w1x, w0y, w1y;
        lean_corner(spx, mxx, a.S[0] - 1, ix, ux, w0x, w1x);
        lean_corner(spy, mxy, a.S[1] - 1, iy, uy, w0y, w1y);
        sw[0] = nrt_mul(w0x, w0y); sw[1] = nrt_mul(w0x, w1y); sw[2] = nrt_mul(w1x, w0y); sw[3] = nrt_mul(w1x, w1y);
        sbase = nrt_mad24(nrt_mad24((unsigned)ix, SY, (unsigned)iy), SZ, 0u) * (unsigned)(C * 4);
        ssx = ux ? str_x : 0u;
        ssy = uy ? str_y : 0u;
    }
    // regular grids, several voxels per lane: the z locations of a lane's voxels rise monotonically, so consecutive voxels sit in the
    // same source cell or in the next one; the (x, y) rows' values at z = c_lo and z = c_hi stay in registers and only what changed
    // is loaded (Resize(2): 12 corner loads per 4 voxels instead of 32 -- the texture-address unit is what these kernels wait for)
    constexpr bool ZCACHE = MODE == NRT_LOC_LINSPACE && VPL > 1;
    float c_lo[4][C], c_hi[4][C];
    int z_lo = -1, z_hi = -1, zp_held = -1;
    float c_pair[4][2 * C];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int qd[3] = {(int)x, (int)y, (int)z0 + k};
        float p[3];
        int iz, uz;
        float w0z, w1z, wxy[4];
        unsigned base, sx, sy;
        if (MODE == NRT_LOC_LINSPACE) {
            p[0] = spx; p[1] = spy; p[2] = linspace(2, qd[2]);
            lean_corner(p[2], mxz, a.S[2] - 1, iz, uz, w0z, w1z);
            base = sbase + (unsigned)iz * (unsigned)(C * 4);
            sx = ssx; sy = ssy;
#pragma unroll
            for (int j = 0; j < 4; ++j) wxy[j] = sw[j];
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) p[d] = MODE == NRT_LOC_ABSOLUTE ? raw[3 * k + d] : nrt_add((float)qd[d], raw[3 * k + d]);
            int ix, iy, ux, uy;
            float w0x, w1x, w0y, w1y;
            lean_corner(p[0], mxx, a.S[0] - 1, ix, ux, w0x, w1x);
            lean_corner(p[1], mxy, a.S[1] - 1, iy, uy, w0y, w1y);
            lean_corner(p[2], mxz, a.S[2] - 1, iz, uz, w0z, w1z);
            base = nrt_mad24(nrt_mad24((unsigned)ix, SY, (unsigned)iy), SZ, (unsigned)iz) * (unsigned)(C * 4);
            sx = ux ? str_x : 0u; sy = uy ? str_y : 0u;
            // (w_x * w_y) * w_z: the x*y products are shared by the two z corners (the rounding sequence of prod_n)
            wxy[0] = nrt_mul(w0x, w0y); wxy[1] = nrt_mul(w0x, w1y); wxy[2] = nrt_mul(w1x, w0y); wxy[3] = nrt_mul(w1x, w1y);
        }
        const unsigned sz = uz ? str_z : 0u;
        float v[8][C];
        bool paired = false;
        if constexpr (C <= 2) paired = a.S[2] >= 2;                      // uniform; a 1-voxel z extent has no pair to load
        if (paired) {
            // the two z corners of an (x, y) row by ONE load of 2 C floats (as the tile form below does): half the lane accesses
            // the texture-address unit has to serve.  At the upper border the pair starts one voxel earlier.
            const unsigned izp = (unsigned)min(iz, a.S[2] - 2);
            const bool second = (unsigned)iz != izp;
            const unsigned pbase = base - ((unsigned)iz - izp) * (unsigned)(C * 4);
            if (!ZCACHE || (int)izp != zp_held) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    load_c<(C <= 2 ? 2 * C : C)>(vol, pbase + ((r & 2) ? sx : 0u) + ((r & 1) ? sy : 0u),
                                                 (float (&)[(C <= 2 ? 2 * C : C)])c_pair[r]);
                zp_held = (int)izp;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float lo_v = second ? c_pair[r][C + c] : c_pair[r][c];
                    v[2 * r][c] = lo_v;
                    v[2 * r + 1][c] = uz ? c_pair[r][C + c] : lo_v;
                }
        } else if (ZCACHE) {
            const int zu = iz + uz;                                      // the upper corner's plane
            if (iz != z_lo) {
                if (iz == z_hi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < C; ++c) c_lo[r][c] = c_hi[r][c];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) load_c<C>(vol, base + ((r & 2) ? sx : 0u) + ((r & 1) ? sy : 0u), c_lo[r]);
                }
                z_lo = iz;
            }
            if (zu != z_hi) {
                if (zu == z_lo) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < C; ++c) c_hi[r][c] = c_lo[r][c];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) load_c<C>(vol, base + ((r & 2) ? sx : 0u) + ((r & 1) ? sy : 0u) + sz, c_hi[r]);
                }
                z_hi = zu;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c) { v[2 * r][c] = c_lo[r][c]; v[2 * r + 1][c] = c_hi[r][c]; }
        } else {
#pragma unroll
            for (int corner = 0; corner < 8; ++corner)
                load_c<C>(vol, base + ((corner & 4) ? sx : 0u) + ((corner & 2) ? sy : 0u) + ((corner & 1) ? sz : 0u), v[corner]);
        }
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0f;                       // :160
        // (channel pairs as 2-vectors -- v_pk_mul_f32 / v_pk_add_f32 -- were measured: 10-14 % fewer instructions, no gain at C = 2 / 3
        // and 68 -> 76 us at C = 4)
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float wt = nrt_mul(wxy[corner >> 1], (corner & 1) ? w1z : w0z);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = nrt_add(acc[c], nrt_mul(wt, v[corner][c]));     // :191
        }
        if (a.has_fill) {
            const bool oob = (p[0] < 0.0f) || (p[0] > mxx) || (p[1] < 0.0f) || (p[1] > mxy) || (p[2] < 0.0f) || (p[2] > mxz);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = apply_fill(acc[c], oob, a.fill_f);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) res[k][c] = acc[c];
    }
    // ---- epilogue: optional addend, one vector store per lane ---------------------------------------------------------
    float *po = out + (size_t)q0 * (unsigned)C;
    if (a.addend) {
        const float *pa = a.addend + (long long)b * a.addend_bs + (size_t)q0 * (unsigned)C;
#pragma unroll
        for (int k = 0; k < VPL; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) res[k][c] = nrt_add(pa[k * C + c], res[k][c]);
    }
    if constexpr (XPOSE) {
        // the block's groups are consecutive in memory: lane t's NF floats are floats [NF t, NF (t + 1)) of the block's run
        const float *r = &res[0][0];
#pragma unroll
        for (int j = 0; j < NF / 4; ++j)
            ((nrt_f4 *)s_out)[threadIdx.x * (unsigned)(NF / 4) + j] = (nrt_f4){r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
        __syncthreads();
        const unsigned first = chunk * 256u;                              // first group of the block
        const unsigned ngrp = min(256u, glim - first);
        const unsigned y0b = lpr == 1u ? first : __umulhi(first, m_lpr);
        const unsigned qb = nrt_mad24(nrt_mad24(x, (unsigned)a.O[1], y0b), (unsigned)a.O[2], (first - y0b * lpr) * (unsigned)VPL);
        nrt_f4 *ob = (nrt_f4 *)(out + (size_t)qb * (unsigned)C);
#pragma unroll
        for (int j = 0; j < NF / 4; ++j) {
            const unsigned e = threadIdx.x + 256u * (unsigned)j;          // 16-byte chunk of the run
            if (e < ngrp * (unsigned)(NF / 4)) __builtin_nontemporal_store(((const nrt_f4 *)s_out)[e], ob + e);
        }
    } else if constexpr (VPL * C == 4) {
        const float *r = &res[0][0];
        *(nrt_f4 *)po = (nrt_f4){r[0], r[1], r[2], r[3]};
    } else if constexpr (VPL * C == 2) {
        *(nrt_f2 *)po = (nrt_f2){res[0][0], res[VPL - 1][C - 1]};
    } else if constexpr ((VPL * C) % 4 == 0) {
        const float *r = &res[0][0];                                     // e.g. 4 voxels x 3 channels: 48 contiguous bytes
#pragma unroll
        for (int j = 0; j < VPL * C / 4; ++j) ((nrt_f4 *)po)[j] = (nrt_f4){r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
    } else {
#pragma unroll
        for (int k = 0; k < VPL; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) po[k * C + c] = res[k][c];
    }
}

// ---- tile form (per-voxel locations: ABSOLUTE / SHIFT) ---------------------------------------------------------------------
// With a displacement field every voxel has its own location, so several voxels per lane share nothing, and lanes that are
// 4 voxels apart in z make every corner read touch 2.5x the cache lines (measured: 0.29 ms against 0.18 ms of the LDS kernel
// for 4 x 160^3 x 1).  Here a lane owns ONE voxel and a block a compact 2 x 4 x 32 (x, y, z) tile: a wave reads two 32-voxel
// z-runs per corner (3-4 lines), and the tile's source box is ~110 lines that stay in L1 / L2 for the tile's 8 corner reads.
// METHOD: 0 linear; 1 nearest on float32 data; 2 nearest on int32 data (utils.py:193-204: one gathered element per channel; the
// fill arithmetic runs in the volume's dtype)
template <int C, int MODE, int METHOD = 0>
__global__ __launch_bounds__(256) void interpn_lean_tile(InterpArgs a, int ltz, int lty, unsigned nTy, unsigned nTz, unsigned ntiles,
                                                         unsigned tpb, unsigned nblk) {
    // a block walks `tpb` consecutive tiles (z fastest): the scalar set-up is paid once, the tile index advances without
    // divisions, and the location of the next tile's voxel is requested before the current tile's corners are blended
    const unsigned blk = nrt_xcd_block(blockIdx.x, gridDim.x);
    if (blk >= nblk) return;
    const int ltx = 8 - ltz - lty;
    unsigned tile = blk * tpb;
    const unsigned tend = min(tile + tpb, ntiles);
    unsigned tzi = tile % nTz, t2 = tile / nTz;                              // uniform: scalar ALU, once per block
    unsigned tyi = t2 % nTy, txi = t2 / nTy;
    const unsigned l = threadIdx.x;
    const int lx = (int)(l >> (ltz + lty)), ly = (int)((l >> ltz) & ((1u << lty) - 1u)), lz = (int)(l & ((1u << ltz) - 1u));
    const int b = blockIdx.y;
    const char *vol = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const char *locb = (const char *)(a.loc + (long long)b * a.loc_bs);
    float *out = (float *)a.out + (long long)b * a.out_bs;
    const float *addb = a.addend ? a.addend + (long long)b * a.addend_bs : nullptr;
    const float mxx = (float)(a.S[0] - 1), mxy = (float)(a.S[1] - 1), mxz = (float)(a.S[2] - 1);
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];
    const unsigned str_x = SY * SZ * (unsigned)(C * 4), str_y = SZ * (unsigned)(C * 4), str_z = (unsigned)(C * 4);
    const int O0 = a.O[0], O1 = a.O[1], O2 = a.O[2];

    auto coords = [&](unsigned cx, unsigned cy, unsigned cz, int (&qd)[3], bool &valid) {
        qd[0] = (int)(cx << ltx) + lx; qd[1] = (int)(cy << lty) + ly; qd[2] = (int)(cz << ltz) + lz;
        valid = qd[0] < O0 && qd[1] < O1 && qd[2] < O2;
        qd[0] = min(qd[0], O0 - 1); qd[1] = min(qd[1], O1 - 1); qd[2] = min(qd[2], O2 - 1);            // loads stay unconditional
    };
    auto flat = [&](const int (&qd)[3]) {
        return nrt_mad24(nrt_mad24((unsigned)qd[0], (unsigned)O1, (unsigned)qd[1]), (unsigned)O2, (unsigned)qd[2]);
    };
    int qd[3];
    bool valid;
    coords(txi, tyi, tzi, qd, valid);
    unsigned q = flat(qd);
    float pn[3];
    {
        const float *lp = (const float *)(locb + (size_t)(nrt_times3(q) << 2));
        pn[0] = lp[0]; pn[1] = lp[1]; pn[2] = lp[2];
    }
    for (; tile < tend; ++tile) {
        float p[3] = {pn[0], pn[1], pn[2]};
        const int cq[3] = {qd[0], qd[1], qd[2]};
        const unsigned cqf = q;
        const bool cvalid = valid;
        // next tile (clamped to the last one of this block: its location load is then a repeat, never out of range)
        if (tile + 1 < tend) {
            if (++tzi == nTz) { tzi = 0; if (++tyi == nTy) { tyi = 0; ++txi; } }
            coords(txi, tyi, tzi, qd, valid);
            q = flat(qd);
        }
        {
            const float *lp = (const float *)(locb + (size_t)(nrt_times3(q) << 2));
            pn[0] = lp[0]; pn[1] = lp[1]; pn[2] = lp[2];
        }
        if (MODE == NRT_LOC_SHIFT) {
#pragma unroll
            for (int d = 0; d < 3; ++d) p[d] = nrt_add((float)cq[d], p[d]);
        }
        if constexpr (METHOD != 0) {
            const unsigned ri = nrt_mad24(nrt_mad24((unsigned)nearest_1d(p[0], a.S[0]), SY, (unsigned)nearest_1d(p[1], a.S[1])), SZ,
                                          (unsigned)nearest_1d(p[2], a.S[2])) * (unsigned)(C * 4);
            float v[C];
            load_c<C>(vol, ri, v);
            if (a.has_fill) {
                const bool oob = (p[0] < 0.0f) || (p[0] > mxx) || (p[1] < 0.0f) || (p[1] > mxy) || (p[2] < 0.0f) || (p[2] > mxz);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (METHOD == 1) v[c] = apply_fill(v[c], oob, a.fill_f);
                    else v[c] = __int_as_float(__float_as_int(v[c]) * (oob ? 0 : 1) + (oob ? 1 : 0) * a.fill_i);
                }
            }
            if (cvalid) {
                float *po = out + (size_t)cqf * (unsigned)C;
                if constexpr (C == 4) __builtin_nontemporal_store((nrt_f4){v[0], v[1], v[2], v[3]}, (nrt_f4 *)po);
                else if constexpr (C == 2) __builtin_nontemporal_store((nrt_f2){v[0], v[1]}, (nrt_f2 *)po);
                else {
#pragma unroll
                    for (int c = 0; c < C; ++c) __builtin_nontemporal_store(v[c], po + c);
                }
            }
            continue;
        }
        int ix, iy, iz, ux, uy, uz;
        float w0x, w1x, w0y, w1y, w0z, w1z;
        lean_corner(p[0], mxx, a.S[0] - 1, ix, ux, w0x, w1x);
        lean_corner(p[1], mxy, a.S[1] - 1, iy, uy, w0y, w1y);
        lean_corner(p[2], mxz, a.S[2] - 1, iz, uz, w0z, w1z);
        const float wxy[4] = {nrt_mul(w0x, w0y), nrt_mul(w0x, w1y), nrt_mul(w1x, w0y), nrt_mul(w1x, w1y)};
        const unsigned sx = ux ? str_x : 0u, sy = uy ? str_y : 0u, sz = uz ? str_z : 0u;
        float v[8][C];
        bool paired = false;
        if constexpr (C <= 2) paired = a.S[2] >= 2;                      // uniform; a 1-voxel z extent has no pair to load
        if (paired) {
            // The texture-address unit spends ~1 cycle per LANE of a gather whose lanes are not consecutive (measured: 8.07 cache
            // accesses per voxel, TA busy 88 %, profiles/archive/r02_smallc): the two z corners of an (x, y) row are neighbours in
            // memory, so ONE load of 2 C floats fetches both -- 4 lane accesses per voxel instead of 8.  At the upper border
            // (z1 == z0 == SZ - 1) the pair starts one voxel earlier and both corners take its second half.
            const unsigned izp = (unsigned)min(iz, a.S[2] - 2);
            const bool second = (unsigned)iz != izp;                     // the lower corner is the pair's upper half
            const unsigned base = nrt_mad24(nrt_mad24((unsigned)ix, SY, (unsigned)iy), SZ, izp) * (unsigned)(C * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t[2 * C];
                load_c<(C <= 2 ? 2 * C : C)>(vol, base + ((r & 2) ? sx : 0u) + ((r & 1) ? sy : 0u), (float (&)[(C <= 2 ? 2 * C : C)])t);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float lo_v = second ? t[C + c] : t[c];
                    v[2 * r][c] = lo_v;
                    v[2 * r + 1][c] = uz ? t[C + c] : lo_v;
                }
            }
        } else {
            const unsigned base = nrt_mad24(nrt_mad24((unsigned)ix, SY, (unsigned)iy), SZ, (unsigned)iz) * (unsigned)(C * 4);
#pragma unroll
            for (int corner = 0; corner < 8; ++corner)
                load_c<C>(vol, base + ((corner & 4) ? sx : 0u) + ((corner & 2) ? sy : 0u) + ((corner & 1) ? sz : 0u), v[corner]);
        }
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0f;                       // :160
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float wt = nrt_mul(wxy[corner >> 1], (corner & 1) ? w1z : w0z);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = nrt_add(acc[c], nrt_mul(wt, v[corner][c]));     // :191
        }
        if (a.has_fill) {
            const bool oob = (p[0] < 0.0f) || (p[0] > mxx) || (p[1] < 0.0f) || (p[1] > mxy) || (p[2] < 0.0f) || (p[2] > mxz);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = apply_fill(acc[c], oob, a.fill_f);
        }
        if (addb) {
            const float *pa = addb + (size_t)cqf * (unsigned)C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = nrt_add(pa[c], acc[c]);
        }
        if (cvalid) {
            float *po = out + (size_t)cqf * (unsigned)C;
            if constexpr (C == 4) __builtin_nontemporal_store((nrt_f4){acc[0], acc[1], acc[2], acc[3]}, (nrt_f4 *)po);
            else if constexpr (C == 2) __builtin_nontemporal_store((nrt_f2){acc[0], acc[1]}, (nrt_f2 *)po);
            else {
                // C = 3: three 4-byte stores measured faster than one 12-byte store (0.331 vs 0.347 ms, 4 x 160^3 x 3)
#pragma unroll
                for (int c = 0; c < C; ++c) __builtin_nontemporal_store(acc[c], po + c);
            }
        }
    }
}

// ---- box form (per-voxel locations, linear; round 6; variant 11, NOT the default: measured slower, see nrt_lean_launch) ---------------
// The tile form above is bound by the texture-address unit: with a displacement field the lanes of a corner load are not consecutive
// dwords and TA serves them about one LANE per clock -- 4 lane accesses per voxel at C = 1, 8 at C = 3: 0.117 / 0.264 ms per 4 x 160^3
// against a roof of 0.041 / 0.082 (profiles/r04_smallc).  This form stages what north_star names: a block owns a
// TX x 8 x 32 (x, y, z) tile of OUTPUT voxels, finds the bounding box of the source voxels its corners touch (a block-wide min / max of the
// clipped floors), copies the box from HBM into LDS with 16-byte loads that are consecutive along z (one line request per ~8 lanes
// instead of one per lane), and every lane gathers its 8 corners from LDS (`ds_read`: 64 independent addresses per instruction).
// Rounds 1-2 had built this idea twice (variants 6 and 9: 450 issue slots per voxel for boxes found per wave, divisions and 64-bit
// addresses in the copy -- slower than the plain gather); here the box is per BLOCK, the copy is a row loop with one multiply-shift
// per 16-byte piece, the location of a voxel is read once and kept in registers across both phases.  A box that does not fit the LDS
// budget (steep or incoherent fields: the tile's source footprint grows with the displacement gradient) takes the direct gather of
// the tile form for that tile -- decided per tile, inside the kernel, from the same min / max.
// Same float32 operations in the same order as the tile form (lean_corner, (wx wy) wz, corners in itertools.product order): bit-identical.
constexpr int BOX_TY = 8, BOX_TZ = 32;

template <int C, int MODE, int TX>
__global__ __launch_bounds__(256) void interpn_lean_box(InterpArgs a, unsigned nTy, unsigned nTz, unsigned ntiles, unsigned box_bytes) {
    extern __shared__ __attribute__((aligned(16))) char box[];           // box_bytes + 32 (slack for the hi-corner over-read) + 96 (bbox exchange)
    const unsigned tile = nrt_xcd_block(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const unsigned tzi = tile % nTz, t2 = tile / nTz, tyi = t2 % nTy, txi = t2 / nTy;     // uniform: scalar ALU
    const unsigned t = threadIdx.x;
    const int lz = (int)(t & 31u), ly = (int)(t >> 5);
    const int b = blockIdx.y;
    const char *locb = (const char *)(a.loc + (long long)b * a.loc_bs);
    float *out = (float *)a.out + (long long)b * a.out_bs;
    const float *addb = a.addend ? a.addend + (long long)b * a.addend_bs : nullptr;
    const float mxx = (float)(a.S[0] - 1), mxy = (float)(a.S[1] - 1), mxz = (float)(a.S[2] - 1);
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];
    constexpr unsigned C4 = (unsigned)(C * 4);
    const int O0 = a.O[0], O1 = a.O[1], O2 = a.O[2];
    const int y = (int)tyi * BOX_TY + ly, z = (int)tzi * BOX_TZ + lz;
    const bool yzv = y < O1 && z < O2;
    const int yc = min(y, O1 - 1), zc = min(z, O2 - 1);

    // ---- phase 1: the locations of this lane's TX voxels (one per x-plane of the tile), the box of their lower corners -------------
    float p[TX][3];
    unsigned qf[TX];
    int mn0 = 0x7fffffff, mn1 = 0x7fffffff, mn2 = 0x7fffffff, mx0 = 0, mx1 = 0, mx2 = 0;
#pragma unroll
    for (int k = 0; k < TX; ++k) {
        const int xc = min((int)txi * TX + k, O0 - 1);                   // (voxels past the volume repeat its last plane: loads stay unconditional)
        qf[k] = nrt_mad24(nrt_mad24((unsigned)xc, (unsigned)O1, (unsigned)yc), (unsigned)O2, (unsigned)zc);
        const float *lp = (const float *)(locb + (size_t)(nrt_times3(qf[k]) << 2));
        const float r0 = lp[0], r1 = lp[1], r2 = lp[2];
        if (MODE == NRT_LOC_SHIFT) { p[k][0] = nrt_add((float)xc, r0); p[k][1] = nrt_add((float)yc, r1); p[k][2] = nrt_add((float)zc, r2); }
        else { p[k][0] = r0; p[k][1] = r1; p[k][2] = r2; }
        // lower corner as lean_corner forms it (floor, clip, integer clamp for NaN)
        const int i0 = min(max((int)__builtin_amdgcn_fmed3f(floorf(p[k][0]), 0.0f, mxx), 0), a.S[0] - 1);
        const int i1 = min(max((int)__builtin_amdgcn_fmed3f(floorf(p[k][1]), 0.0f, mxy), 0), a.S[1] - 1);
        const int i2 = min(max((int)__builtin_amdgcn_fmed3f(floorf(p[k][2]), 0.0f, mxz), 0), a.S[2] - 1);
        mn0 = min(mn0, i0); mx0 = max(mx0, i0); mn1 = min(mn1, i1); mx1 = max(mx1, i1); mn2 = min(mn2, i2); mx2 = max(mx2, i2);
    }
#pragma unroll
    for (int off = 1; off < NRT_WAVE; off <<= 1) {
        mn0 = min(mn0, __shfl_xor(mn0, off, NRT_WAVE)); mx0 = max(mx0, __shfl_xor(mx0, off, NRT_WAVE));
        mn1 = min(mn1, __shfl_xor(mn1, off, NRT_WAVE)); mx1 = max(mx1, __shfl_xor(mx1, off, NRT_WAVE));
        mn2 = min(mn2, __shfl_xor(mn2, off, NRT_WAVE)); mx2 = max(mx2, __shfl_xor(mx2, off, NRT_WAVE));
    }
    int *bb = (int *)(box + box_bytes + 32u);                            // [4 waves][6]
    if ((t & 63u) == 0u) {
        int *w = bb + (t >> 6) * 6u;
        w[0] = mn0; w[1] = mx0; w[2] = mn1; w[3] = mx1; w[4] = mn2; w[5] = mx2;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        mn0 = min(mn0, bb[w * 6 + 0]); mx0 = max(mx0, bb[w * 6 + 1]); mn1 = min(mn1, bb[w * 6 + 2]);
        mx1 = max(mx1, bb[w * 6 + 3]); mn2 = min(mn2, bb[w * 6 + 4]); mx2 = max(mx2, bb[w * 6 + 5]);
    }
    // (uniform for the compiler too: the copy loop's bounds and the row stride live in scalar registers)
    const unsigned x0b = (unsigned)__builtin_amdgcn_readfirstlane(mn0), y0b = (unsigned)__builtin_amdgcn_readfirstlane(mn1);
    const unsigned z0v = (unsigned)__builtin_amdgcn_readfirstlane(mn2);
    const unsigned nx = (unsigned)min(__builtin_amdgcn_readfirstlane(mx0) + 1, a.S[0] - 1) - x0b + 1u;       // upper corners: one further, inside the volume
    const unsigned ny = (unsigned)min(__builtin_amdgcn_readfirstlane(mx1) + 1, a.S[1] - 1) - y0b + 1u;
    const unsigned z1v = (unsigned)min(__builtin_amdgcn_readfirstlane(mx2) + 1, a.S[2] - 1);
    const unsigned zb0 = (z0v * C4) & ~15u, cpr = ((((z1v + 1u) * C4 + 15u) & ~15u) - zb0) >> 4;           // 16-byte pieces of a z-run
    // LDS rows are 16 / 32 / 64 pieces long (the lanes that walk a row): a wave instruction then fills 4 / 2 / 1 CONSECUTIVE rows -- the
    // form LDS-DMA needs (destination = M0 + 16 lane) -- and a sweep of the block 4 KB
    const unsigned lsh = cpr <= 16u ? 4u : (cpr <= 32u ? 5u : 6u);
    const unsigned rowb = 16u << lsh, rps = 256u >> lsh, nrows = nx * ny;
    const unsigned nsweep = (nrows + rps - 1u) / rps;
    const bool staged = nsweep * 4096u <= box_bytes && nrows * ny < 65536u && cpr <= 64u;

    // ---- phase 2: the box, HBM -> LDS by LDS-DMA (no registers in between, every piece requested before the first one is waited for) --
    if (staged) {
        const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((const float *)a.vol + (long long)b * a.vol_bs), 0, (int)((unsigned)a.S[0] * SY * SZ * C4), 0x00020000);
        const unsigned k = t & ((1u << lsh) - 1u), r0 = t >> lsh;
        const unsigned m_ny = (65536u + ny - 1u) / ny;                                                    // r / ny = (r m) >> 16 for r ny < 2^16
        const unsigned wave_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(t >> 6)) * 1024u;          // this wave's KB of a sweep
        typedef __attribute__((address_space(3))) void bx_lds;
        for (unsigned sw = 0; sw < nsweep; ++sw) {
            const unsigned r = sw * rps + r0;
            const bool act = r < nrows && k < cpr;
            const unsigned rx = (r * m_ny) >> 16, ry = r - rx * ny;
            const unsigned goff = nrt_mad24(nrt_mad24(x0b + rx, SY, y0b + ry), SZ, 0u) * C4 + zb0 + (k << 4);
            // (a piece that reaches past the volume's end comes back zero-padded; inactive lanes read nothing and store zeros)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (bx_lds *)(box + sw * 4096u + wave_dst), 16, act ? goff : 0xffffffffu, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                                  // vmcnt(0): this wave's pieces have landed
    }
    __syncthreads();

    // ---- phase 3: corners from LDS (or, for a box that did not fit, from memory as the tile form reads them), blend, store -----------
    const char *vol = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const unsigned str_x = SY * SZ * C4, str_y = SZ * C4;
#pragma unroll
    for (int k = 0; k < TX; ++k) {
        int ix, iy, iz, ux, uy, uz;
        float w0x, w1x, w0y, w1y, w0z, w1z;
        lean_corner(p[k][0], mxx, a.S[0] - 1, ix, ux, w0x, w1x);
        lean_corner(p[k][1], mxy, a.S[1] - 1, iy, uy, w0y, w1y);
        lean_corner(p[k][2], mxz, a.S[2] - 1, iz, uz, w0z, w1z);
        const float wxy[4] = {nrt_mul(w0x, w0y), nrt_mul(w0x, w1y), nrt_mul(w1x, w0y), nrt_mul(w1x, w1y)};
        float v[8][C];
        if (staged) {
            const unsigned base = nrt_mad24(nrt_mad24((unsigned)ix - x0b, ny, (unsigned)iy - y0b), rowb, (unsigned)iz * C4 - zb0);
            const unsigned sx = ux ? ny * rowb : 0u, sy = uy ? rowb : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *lp = (const float *)(box + base + ((r & 2) ? sx : 0u) + ((r & 1) ? sy : 0u));
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float lo_v = lp[c], hi_v = lp[C + c];          // (the z-neighbour is read whether or not it is a corner: one ds_read2 / wider read)
                    v[2 * r][c] = lo_v;
                    v[2 * r + 1][c] = uz ? hi_v : lo_v;
                }
            }
        } else {
            const unsigned base = nrt_mad24(nrt_mad24((unsigned)ix, SY, (unsigned)iy), SZ, (unsigned)iz) * C4;
            const unsigned sx = ux ? str_x : 0u, sy = uy ? str_y : 0u, sz = uz ? C4 : 0u;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner)
                load_c<C>(vol, base + ((corner & 4) ? sx : 0u) + ((corner & 2) ? sy : 0u) + ((corner & 1) ? sz : 0u), v[corner]);
        }
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0f;                       // :160
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float wt = nrt_mul(wxy[corner >> 1], (corner & 1) ? w1z : w0z);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = nrt_add(acc[c], nrt_mul(wt, v[corner][c]));     // :191
        }
        if (a.has_fill) {
            const bool oob = (p[k][0] < 0.0f) || (p[k][0] > mxx) || (p[k][1] < 0.0f) || (p[k][1] > mxy) || (p[k][2] < 0.0f) || (p[k][2] > mxz);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = apply_fill(acc[c], oob, a.fill_f);
        }
        if (addb) {
            const float *pa = addb + (size_t)qf[k] * (unsigned)C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = nrt_add(pa[c], acc[c]);
        }
        if (yzv && (int)txi * TX + k < O0) {
            float *po = out + (size_t)qf[k] * (unsigned)C;
            if constexpr (C == 4) __builtin_nontemporal_store((nrt_f4){acc[0], acc[1], acc[2], acc[3]}, (nrt_f4 *)po);
            else if constexpr (C == 2) __builtin_nontemporal_store((nrt_f2){acc[0], acc[1]}, (nrt_f2 *)po);
            else {
#pragma unroll
                for (int c = 0; c < C; ++c) __builtin_nontemporal_store(acc[c], po + c);
            }
        }
    }
}

// box budget per block: the C = 1 box of an 8 x 8 x 32 tile whose footprint stretches by g is (8 g + 2)^2 rows of 256 bytes (z extents up
// to 64 floats) -- 31 KB at g = 1.1, 49 KB at g = 1.5: 48 KB = three blocks per CU; C >= 2 tiles are 4 planes thick and take 64 KB
template <int C>
int launch_lean_box(const InterpArgs &a, int batch, int mode, hipStream_t st) {
    constexpr int TX = C == 1 ? 8 : 4;
    const unsigned box_bytes = C == 1 ? 48u * 1024u : 64u * 1024u;
    const unsigned nTx = (a.O[0] + TX - 1) / TX, nTy = (a.O[1] + BOX_TY - 1) / BOX_TY, nTz = (a.O[2] + BOX_TZ - 1) / BOX_TZ;
    const unsigned ntiles = nTx * nTy * nTz;
    const unsigned shm = box_bytes + 32u + 96u;
    dim3 grid(nrt_xcd_grid(ntiles), batch), blk(256);
    if (mode == NRT_LOC_ABSOLUTE) {
        if (hipFuncSetAttribute((const void *)interpn_lean_box<C, NRT_LOC_ABSOLUTE, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return NRT_ERR_LAUNCH;
        hipLaunchKernelGGL((interpn_lean_box<C, NRT_LOC_ABSOLUTE, TX>), grid, blk, shm, st, a, nTy, nTz, ntiles, box_bytes);
    } else {
        if (hipFuncSetAttribute((const void *)interpn_lean_box<C, NRT_LOC_SHIFT, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return NRT_ERR_LAUNCH;
        hipLaunchKernelGGL((interpn_lean_box<C, NRT_LOC_SHIFT, TX>), grid, blk, shm, st, a, nTy, nTz, ntiles, box_bytes);
    }
    return NRT_OK;
}

template <int C>
void launch_lean_tile(const InterpArgs &a, int batch, int mode, int method_kind, hipStream_t st) {
    // 2 x 4 x 32 tiles (a wave: two 32-voxel z-runs); linear interpolation of 2..4 channels: 8 x 2 x 16 (a wave: 2 x 2 x 16), whose
    // corner loads touch fewer distinct lines per instruction (tools/lean_geo_sweep.sh, 4 x 160^3: C = 2 0.206 -> 0.186 ms,
    // C = 3 0.299 -> 0.263, C = 4 0.344 -> 0.297; C = 1 0.117 -> 0.136).  Short z extents trade z for y / x
    int ltz = 5, lty = 2;
    if (C >= 2 && method_kind == 0) { ltz = 4; lty = 1; }
    while (ltz > 0 && (1 << (ltz - 1)) >= a.O[2]) { --ltz; ++lty; }
    while (lty > 0 && (1 << (lty - 1)) >= a.O[1]) --lty;
    static int geo_env = -1;                  // experiments: NRT_LEAN_GEO = ltz * 16 + lty (tools/lean_geo_sweep.sh)
    if (geo_env < 0) { const char *e = getenv("NRT_LEAN_GEO"); geo_env = e ? atoi(e) : 0; }
    if (geo_env > 0 && (geo_env >> 4) + (geo_env & 15) <= 8) { ltz = geo_env >> 4; lty = geo_env & 15; }
    const int ltx = 8 - ltz - lty;
    const unsigned nTx = (a.O[0] + (1 << ltx) - 1) >> ltx, nTy = (a.O[1] + (1 << lty) - 1) >> lty, nTz = (a.O[2] + (1 << ltz) - 1) >> ltz;
    const unsigned ntiles = nTx * nTy * nTz;
    static int tpb_env = -1;
    if (tpb_env < 0) { const char *e = getenv("NRT_LEAN_TPB"); tpb_env = e ? atoi(e) : 0; }
    // tiles per block, measured at 4 x 160^3 (tools/lean_sweep.sh): C = 1 0.176 / 0.169 / 0.173 ms at 1 / 4 / 8; C = 3 0.298 / 0.323 / 0.331
    unsigned tpb = tpb_env > 0 ? (unsigned)tpb_env : (C == 1 ? 4u : 1u);
    while (tpb > 1 && (ntiles / tpb) * (unsigned)batch < 2048u) tpb >>= 1;     // keep the chip full on small volumes
    const unsigned nblk = (ntiles + tpb - 1) / tpb;
    dim3 grid(nrt_xcd_grid(nblk), batch), blk(256);
#define NRT_LEAN_T(MODE)                                                                                                         \
    switch (method_kind) {                                                                                                       \
        case 1: hipLaunchKernelGGL((interpn_lean_tile<C, MODE, 1>), grid, blk, 0, st, a, ltz, lty, nTy, nTz, ntiles, tpb, nblk); break; \
        case 2: hipLaunchKernelGGL((interpn_lean_tile<C, MODE, 2>), grid, blk, 0, st, a, ltz, lty, nTy, nTz, ntiles, tpb, nblk); break; \
        default: hipLaunchKernelGGL((interpn_lean_tile<C, MODE, 0>), grid, blk, 0, st, a, ltz, lty, nTy, nTz, ntiles, tpb, nblk); break; \
    }
    if (mode == NRT_LOC_ABSOLUTE) { NRT_LEAN_T(NRT_LOC_ABSOLUTE) }
    else { NRT_LEAN_T(NRT_LOC_SHIFT) }
#undef NRT_LEAN_T
}

template <int C, int VPL>
void launch_lean_cv(const InterpArgs &a, int batch, int mode, hipStream_t st) {
    const unsigned lpr = (unsigned)a.O[2] / VPL;
    const unsigned m_lpr = lpr == 1 ? 0u : (unsigned)(0x100000000ull / lpr) + 1u;
    const unsigned groups = (unsigned)a.O[1] * lpr;
    const unsigned cpp = (groups + 255u) / 256u;
    const unsigned nblk = cpp * (unsigned)a.O[0];
    dim3 grid(nrt_xcd_grid(nblk), batch), blk(256);
    if (lpr == 1) {
        // g / 1: the multiply-high constant would overflow; lpr == 1 means one group per row (y = g)
        switch (mode) {
            case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((interpn_lean<C, VPL, NRT_LOC_ABSOLUTE>), grid, blk, 0, st, a, 1u, 0u, cpp, nblk); break;
            case NRT_LOC_SHIFT: hipLaunchKernelGGL((interpn_lean<C, VPL, NRT_LOC_SHIFT>), grid, blk, 0, st, a, 1u, 0u, cpp, nblk); break;
            default: hipLaunchKernelGGL((interpn_lean<C, VPL, NRT_LOC_LINSPACE>), grid, blk, 0, st, a, 1u, 0u, cpp, nblk); break;
        }
        return;
    }
    switch (mode) {
        case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((interpn_lean<C, VPL, NRT_LOC_ABSOLUTE>), grid, blk, 0, st, a, lpr, m_lpr, cpp, nblk); break;
        case NRT_LOC_SHIFT: hipLaunchKernelGGL((interpn_lean<C, VPL, NRT_LOC_SHIFT>), grid, blk, 0, st, a, lpr, m_lpr, cpp, nblk); break;
        default: hipLaunchKernelGGL((interpn_lean<C, VPL, NRT_LOC_LINSPACE>), grid, blk, 0, st, a, lpr, m_lpr, cpp, nblk); break;
    }
}

}  // namespace

// the kernel addresses the source with 32-bit byte offsets and 24-bit multiplies, and divides the in-plane group index by a
// multiply-high that is exact while groups * lanes-per-row < 2^32
bool nrt_lean_supported(const int *vol_shape, const int *out_shape, int channels, int ndim, const void *vol, const void *loc,
                        const void *out, long long vol_bs, long long loc_bs) {
    if (ndim != 3 || channels < 1 || channels > 4) return false;
    unsigned long long vbytes = 4ull * channels;
    for (int d = 0; d < 3; ++d) {
        if (vol_shape[d] < 1 || out_shape[d] < 1 || vol_shape[d] >= (1 << 12) || out_shape[d] >= (1 << 12)) return false;
        vbytes *= (unsigned long long)vol_shape[d];
    }
    if (vbytes >= (1ull << 32)) return false;
    // 32-bit byte offsets into the location field (12 B per output voxel) and into the output / addend rows (4 C B per voxel)
    const unsigned long long nout = (unsigned long long)out_shape[0] * out_shape[1] * out_shape[2];
    if (nout * 12ull >= (1ull << 32) || nout * 4ull * (unsigned long long)channels >= (1ull << 32)) return false;
    // vector accesses: C floats per source voxel, VPL * C per output group, VPL * 3 per location group.  Bases must be
    // 16-byte aligned; the batch strides keep the alignment the channel count needs (the per-lane vector widths over z are
    // chosen at launch from the z extent and the location stride)
    if ((((uintptr_t)vol | (uintptr_t)out | (uintptr_t)loc) & 15) != 0) return false;
    const int valign = channels == 4 ? 16 : (channels == 2 ? 8 : 4);
    if ((vol_bs * 4) % valign != 0) return false;
    (void)loc_bs;
    return true;
}

int nrt_lean_launch(const void *args, int batch, int mode, int method_kind, void *stream, int form) {
    const InterpArgs &a = *(const InterpArgs *)args;
    hipStream_t st = nrt_stream(stream);
    // per-voxel locations, linear: the tile form unless the box form (source bounding box of a tile staged in LDS, variant 11) is asked
    // for by name -- measured on MI355X, 4 x 160^3 (tools/box_ab.py, profiles/r06_lab/box_ab.jsonl): the box form is bit-identical and
    // SLOWER on every field and channel count (bench field C = 1 0.137 - 0.159 ms against 0.114 - 0.127, C = 3 0.271 against 0.257; gentle
    // field C = 1 0.081 against 0.072): what it saves the texture unit it spends on the box search, the barriers between its three
    // phases and boxes that outgrow the LDS where the field is steep.  Kept as the A/B partner; needs a z extent of at least a tile's 32.
    if (mode != NRT_LOC_LINSPACE && method_kind == 0 && form == NRT_LEAN_FORM_BOX) {
        if (a.O[2] < BOX_TZ || batch > 65535) return NRT_ERR_UNSUPPORTED;
        int rc;
        switch (a.C) {
            case 1: rc = launch_lean_box<1>(a, batch, mode, st); break;
            case 2: rc = launch_lean_box<2>(a, batch, mode, st); break;
            case 3: rc = launch_lean_box<3>(a, batch, mode, st); break;
            default: rc = launch_lean_box<4>(a, batch, mode, st); break;
        }
        if (rc != NRT_OK) return rc;
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    if (form == NRT_LEAN_FORM_BOX) return NRT_ERR_UNSUPPORTED;
    if (mode != NRT_LOC_LINSPACE) {           // per-voxel locations: one voxel per lane, compact tiles
        switch (a.C) {
            case 1: launch_lean_tile<1>(a, batch, mode, method_kind, st); break;
            case 2: launch_lean_tile<2>(a, batch, mode, method_kind, st); break;
            case 3: launch_lean_tile<3>(a, batch, mode, method_kind, st); break;
            default: launch_lean_tile<4>(a, batch, mode, method_kind, st); break;
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    if (method_kind != 0) return NRT_ERR_UNSUPPORTED;      // the row form (regular grids) is linear only
    switch (a.C) {
        case 1:
            // 4 voxels per lane: 16-byte location loads and stores need z % 4 == 0 (then nout % 4 == 0 and the contiguous batch
            // strides of loc / out / addend are multiples of 16 bytes too; a foreign loc stride is checked)
            if (a.O[2] % 4 == 0 && (a.loc_bs * 4) % 16 == 0 && (a.addend_bs * 4) % 16 == 0) launch_lean_cv<1, 4>(a, batch, mode, st);
            else launch_lean_cv<1, 1>(a, batch, mode, st);
            break;
        case 2:
            if (mode == NRT_LOC_LINSPACE && a.O[2] % 4 == 0 && (a.out_bs * 4) % 16 == 0) launch_lean_cv<2, 4>(a, batch, mode, st);
            else if (a.O[2] % 2 == 0 && (a.loc_bs * 4) % 8 == 0 && (a.addend_bs * 4) % 16 == 0) launch_lean_cv<2, 2>(a, batch, mode, st);
            else launch_lean_cv<2, 1>(a, batch, mode, st);
            break;
        // C = 2 .. 4 on regular grids: four voxels per lane (the x / y corner arithmetic is shared), results transposed through LDS
        case 3:
            if (mode == NRT_LOC_LINSPACE && a.O[2] % 4 == 0 && (a.out_bs * 4) % 16 == 0) launch_lean_cv<3, 4>(a, batch, mode, st);
            else launch_lean_cv<3, 1>(a, batch, mode, st);
            break;
        default:
            if (mode == NRT_LOC_LINSPACE && a.O[2] % 4 == 0) launch_lean_cv<4, 4>(a, batch, mode, st);
            else launch_lean_cv<4, 1>(a, batch, mode, st);
            break;
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
