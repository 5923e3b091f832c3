// interpn / SpatialTransformer / Resize kernels for gfx950 (MI355X).
//
// Replaces the ~100 TensorFlow ops that neurite/tf/utils/utils.py:73-220 issues per call (8 gathers +
// ~90 element-wise ops, each materialising a V- or V*C-sized temporary) with ONE pass: every source
// row is read once from HBM, every output row written once, the identity grid (utils.py:333-476)
// and the linspace grid of resize() (utils.py:259-260) are computed in registers.
//
// Layout in HBM (the reference's): vol [B, X, Y, Z, C], loc/shift [B, X', Y', Z', D],
// out [B, X', Y', Z', C]; row-major, channel fastest.  A voxel's C channels are one contiguous
// "row" (128 B at C = 32 fp32 = exactly one cache line / HBM burst).
//
// Three kernels:
//   interpn_generic   one thread per output element, any C, D in {1,2,3}, float32 (linear/nearest)
//                     or int32 (nearest).  Coalesced over channels.
//   interpn_rows      C % 4 == 0: G = C/4 lanes own one voxel, each lane moves 16 B of every corner
//                     row (global_load_dwordx4); a wave64 covers 64/G consecutive-z voxels, so the
//                     corner rows of one wave instruction are (for smooth fields) one contiguous run.
//   interpn_zrun_c32  C == 32, D == 3, linear: an 8-lane group walks a run of consecutive-z outputs
//                     and keeps the four upper-z corner rows in registers as the next voxel's lower-z
//                     rows (4 row loads per voxel instead of 8); loc/shift is fetched 8 voxels at a
//                     time (96 B, coalesced) and broadcast inside the group with wave shuffles; the
//                     next voxel's rows are in flight while the current one is blended.
//
// Arithmetic follows the reference op-for-op in float32 with one rounding per TF op (no FMA
// contraction), corners accumulated in itertools.product order (utils.py:159-191), so the linear
// path is bit-identical to the CPU restatement, not merely within 1e-5.

#include "interpn_core.h"
#include "lean.h"
#include "wc.h"

namespace {

// ============================================================================================
// generic: one thread per output element
// ============================================================================================
template <int D, int MODE, int METHOD, typename T>
__global__ __launch_bounds__(256) void interpn_generic(InterpArgs a) {
    const int b = blockIdx.y;
    const T *vol = (const T *)a.vol + (long long)b * a.vol_bs;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    T *out = (T *)a.out + (long long)b * a.out_bs;
    const unsigned long long total = (unsigned long long)a.nout * (unsigned)a.C;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const unsigned q = (unsigned)(e / (unsigned)a.C);
        const int c = (int)(e - (unsigned long long)q * (unsigned)a.C);
        int qd[NRT_MAXD];
        float p[NRT_MAXD];
        decode<D>(a, q, qd);
        load_loc<D, MODE>(a, locb, q, qd, p);
        const bool oob = a.has_fill ? out_of_bounds<D>(a, p) : false;
        if (METHOD == NRT_INTERP_NEAREST) {
            long long idx = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) idx = idx * a.S[d] + nearest_1d(p[d], a.S[d]);
            T v = vol[idx * a.C + c];
            if (a.has_fill) {
                if (sizeof(T) == 4 && __is_same(T, float)) {
                    float fv = apply_fill(*(float *)&v, oob, a.fill_f);
                    v = *(T *)&fv;
                } else {
                    int iv = *(int *)&v;
                    iv = iv * (oob ? 0 : 1) + (oob ? 1 : 0) * a.fill_i;
                    v = *(T *)&iv;
                }
            }
            out[e] = v;
        } else {
            int i0[NRT_MAXD], i1[NRT_MAXD];
            float w0[NRT_MAXD], w1[NRT_MAXD];
#pragma unroll
            for (int d = 0; d < D; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
            float acc = 0.0f;                                    // :160
#pragma unroll
            for (int corner = 0; corner < (1 << D); ++corner) {  // product([0,1], repeat=D) order
                long long idx = 0;
                float wt = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int bit = (corner >> (D - 1 - d)) & 1;
                    idx = idx * a.S[d] + (bit ? i1[d] : i0[d]);              // sub2ind2d :1068-1082
                    const float w = bit ? w1[d] : w0[d];
                    wt = (d == 0) ? w : nrt_mul(wt, w);                      // prod_n :1085-1092
                }
                const float v = ((const float *)vol)[idx * a.C + c];
                acc = nrt_add(acc, nrt_mul(wt, v));                          // :191
            }
            if (a.has_fill) acc = apply_fill(acc, oob, a.fill_f);
            if (a.addend) acc = nrt_add(a.addend[(long long)b * a.addend_bs + (long long)e], acc);
            ((float *)out)[e] = acc;
        }
    }
}

// ============================================================================================
// rows: G = C/4 lanes per voxel, float4 per lane, D == 3 (D < 3 is padded with size-1 dims by the host)
// ============================================================================================
template <int G, int MODE, int METHOD>
__global__ __launch_bounds__(256) void interpn_rows(InterpArgs a, unsigned nblk, unsigned vox_per_block) {
    constexpr int D = 3;
    constexpr int NG = 256 / G;      // voxels per block pass
    const unsigned lb = nrt_xcd_block(blockIdx.x, gridDim.x);
    if (lb >= nblk) return;
    const int b = blockIdx.y;
    const nrt_f4 *vol = (const nrt_f4 *)((const float *)a.vol + (long long)b * a.vol_bs);
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    nrt_f4 *out = (nrt_f4 *)((float *)a.out + (long long)b * a.out_bs);
    const int lg = threadIdx.x % G;
    const unsigned g = threadIdx.x / G;
    const unsigned q0 = lb * vox_per_block;
    const unsigned q1 = min(q0 + vox_per_block, a.nout);
    const int Y = a.S[1], Z = a.S[2];

#pragma unroll 2
    for (unsigned q = q0 + g; q < q1; q += NG) {
        int qd[NRT_MAXD];
        float p[NRT_MAXD];
        decode<D>(a, q, qd);
        load_loc<D, MODE>(a, locb, q, qd, p);
        const bool oob = a.has_fill ? out_of_bounds<D>(a, p) : false;
        nrt_f4 acc;
        if (METHOD == NRT_INTERP_NEAREST) {
            const long long idx = ((long long)nearest_1d(p[0], a.S[0]) * Y + nearest_1d(p[1], Y)) * Z
                                  + nearest_1d(p[2], Z);
            acc = vol[idx * G + lg];
        } else {
            int i0[3], i1[3];
            float w0[3], w1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
            nrt_f4 v[8];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int ix = (corner & 4) ? i1[0] : i0[0];
                const int iy = (corner & 2) ? i1[1] : i0[1];
                const int iz = (corner & 1) ? i1[2] : i0[2];
                v[corner] = vol[(((long long)ix * Y + iy) * Z + iz) * G + lg];
            }
            acc = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const float wt = nrt_mul(nrt_mul((corner & 4) ? w1[0] : w0[0], (corner & 2) ? w1[1] : w0[1]),
                                         (corner & 1) ? w1[2] : w0[2]);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = nrt_add(acc[k], nrt_mul(wt, v[corner][k]));
            }
        }
        if (a.has_fill) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = apply_fill(acc[k], oob, a.fill_f);
        }
        __builtin_nontemporal_store(acc, &out[(long long)q * G + lg]);
    }
}

// ============================================================================================
// z-run: C == 32 (8 lanes x float4 per row), D == 3, linear.
// Block = 256 threads = 32 lane-groups = a 4(x) x 8(y) patch of output columns; each group walks
// z in [zc*LZ, zc*LZ+LZ).  Logical blocks are XCD-contiguous in x so neighbouring patches share L2.
// ============================================================================================
struct ZIdx {
    int ix0, ix1, iy0, iy1, iz0, iz1;
    float wx0, wx1, wy0, wy1, wz0, wz1;
    bool oob;
};

template <int MODE>
__device__ __forceinline__ void zrun_index(const InterpArgs &a, int x, int y, int z, float sx, float sy, float sz,
                                           ZIdx &o) {
    float p[NRT_MAXD];
    if (MODE == NRT_LOC_ABSOLUTE) {
        p[0] = sx; p[1] = sy; p[2] = sz;
    } else if (MODE == NRT_LOC_SHIFT) {
        p[0] = nrt_add((float)x, sx); p[1] = nrt_add((float)y, sy); p[2] = nrt_add((float)z, sz);
    } else {
        const int qd[3] = {x, y, z};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            p[d] = (qd[d] == 0) ? 0.0f
                 : ((qd[d] == a.O[d] - 1) ? (float)(a.S[d] - 1) : nrt_mul(a.delta[d], (float)qd[d]));
    }
    corner_1d(p[0], a.S[0], o.ix0, o.ix1, o.wx0, o.wx1);
    corner_1d(p[1], a.S[1], o.iy0, o.iy1, o.wy0, o.wy1);
    corner_1d(p[2], a.S[2], o.iz0, o.iz1, o.wz0, o.wz1);
    o.oob = a.has_fill ? out_of_bounds<3>(a, p) : false;
}

__device__ __forceinline__ unsigned zrun_col(const InterpArgs &a, const ZIdx &i, int c) {
    // column base (row index at z = 0) of xy-corner c = cx*2 + cy; the host guarantees that a whole
    // volume is < 4 GiB so that rows are addressed as uniform base (SGPR pair) + 32-bit byte offset
    const int ix = (c & 2) ? i.ix1 : i.ix0;
    const int iy = (c & 1) ? i.iy1 : i.iy0;
    return ((unsigned)ix * (unsigned)a.S[1] + (unsigned)iy) * (unsigned)a.S[2];
}

__device__ __forceinline__ nrt_f4 zrun_row(const char *__restrict__ volb, unsigned row, int lg) {
    return *(const nrt_f4 *)(volb + (size_t)((row * 8u + (unsigned)lg) * 16u));
}

// One z-step.  Planes A (lower-z rows of the current voxel), B (its upper-z rows) are complete or in
// flight from earlier steps; C is free.  Order matters for latency hiding:
//   1. index voxel z+1 and issue its upper-z rows into C          (4 x 16 B per lane in flight)
//   2. blend voxel z from A and B, store                          (waits only for A/B: vmcnt(4))
//   3. if the run is not coherent (rare), reload B in place with voxel z+1's lower-z rows
// The caller rotates roles (A,B,C) -> (B,C,A) -> (C,A,B) by unrolling x3, so no register is ever
// copied while its load is outstanding.
struct ZShift {
    float s0, s1, s2;   // current batch of 8 voxels (24 floats): lane lg holds flat elements lg, lg+8, lg+16
    float n0, n1, n2;   // next batch, prefetched one batch ahead
};

template <int MODE>
__device__ __forceinline__ bool zrun_step(const InterpArgs &a, const char *__restrict__ vol,
                                          const float *__restrict__ locq, nrt_f4 *__restrict__ outq,
                                          int x, int y, int z, int zbeg, int zend, int lg, int grp_base,
                                          ZIdx &cur, ZShift &sh, nrt_f4 (&A)[4], nrt_f4 (&B)[4], nrt_f4 (&C)[4]) {
    const bool has_next = (z + 1 < zend);
    ZIdx nxt = cur;
    bool reuse = true;
    if (has_next) {
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        if (MODE != NRT_LOC_LINSPACE) {
            const int kb = (z + 1 - zbeg) & 7;
            if (kb == 0) {                                   // uniform: entering the prefetched batch
                sh.s0 = sh.n0; sh.s1 = sh.n1; sh.s2 = sh.n2;
                const int zn = z + 1 + 8;
                if (zn < zend) {
                    const int nrem = (zend - zn) * 3;
                    const float *sp = locq + (long long)zn * 3;
                    sh.n0 = (lg < nrem) ? __builtin_nontemporal_load(sp + lg) : 0.0f;
                    sh.n1 = (lg + 8 < nrem) ? __builtin_nontemporal_load(sp + lg + 8) : 0.0f;
                    sh.n2 = (lg + 16 < nrem) ? __builtin_nontemporal_load(sp + lg + 16) : 0.0f;
                }
            }
            const int e = 3 * kb;
            // element e+d lives in register (e+d)>>3 of lane (e+d)&7 of this lane-group
            const int e1 = e + 1, e2 = e + 2;
            const float r0 = (e < 8) ? sh.s0 : ((e < 16) ? sh.s1 : sh.s2);
            const float r1 = (e1 < 8) ? sh.s0 : ((e1 < 16) ? sh.s1 : sh.s2);
            const float r2 = (e2 < 8) ? sh.s0 : ((e2 < 16) ? sh.s1 : sh.s2);
            sx = __shfl(r0, grp_base | (e & 7), 64);
            sy = __shfl(r1, grp_base | (e1 & 7), 64);
            sz = __shfl(r2, grp_base | (e2 & 7), 64);
        }
        zrun_index<MODE>(a, x, y, z + 1, sx, sy, sz, nxt);
        reuse = (nxt.ix0 == cur.ix0) && (nxt.ix1 == cur.ix1) && (nxt.iy0 == cur.iy0) &&
                (nxt.iy1 == cur.iy1) && (nxt.iz0 == cur.iz1);
#pragma unroll
        for (int c = 0; c < 4; ++c) C[c] = zrun_row(vol, zrun_col(a, nxt, c) + (unsigned)nxt.iz1, lg);
    }
    // blend the current voxel (utils.py:159-191 corner order: x, y, z with z fastest)
    nrt_f4 acc = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float wt = nrt_mul(nrt_mul((corner & 4) ? cur.wx1 : cur.wx0, (corner & 2) ? cur.wy1 : cur.wy0),
                                 (corner & 1) ? cur.wz1 : cur.wz0);
        const nrt_f4 v = (corner & 1) ? B[corner >> 1] : A[corner >> 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = nrt_add(acc[k], nrt_mul(wt, v[k]));
    }
    if (a.has_fill) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = apply_fill(acc[k], cur.oob, a.fill_f);
    }
    __builtin_nontemporal_store(acc, &outq[(long long)z * 8 + lg]);
    if (!reuse) {
#pragma unroll
        for (int c = 0; c < 4; ++c) B[c] = zrun_row(vol, zrun_col(a, nxt, c) + (unsigned)nxt.iz0, lg);
    }
    cur = nxt;
    return !has_next;
}

// Block = WX x WY waves, each wave a 2(x) x 4(y) patch of lane-groups: output patch (2 WX) x (4 WY) columns.
// Bigger patches shrink the halo that every block re-fetches beyond L2 (measured, profiles/: with the
// 4x8 patch FETCH_SIZE was 1.6x the algorithmic bytes = exactly the (5*9)/(4*8) halo of unshared blocks).
// zc_outer: order of the blocks inside one XCD's slab -- 0: z-chunks of one patch are consecutive;
// 1: neighbouring patches of one z-chunk are consecutive, so blocks resident together are x/y
// neighbours marching through z in step and meet their shared halo rows in L2.
template <int MODE, int WX, int WY>
__global__ __launch_bounds__(WX * WY * 64) void interpn_zrun_c32(InterpArgs a, int LZ, unsigned nTy, unsigned nZc,
                                                                 unsigned nblk, int zc_outer, int lreg, int nbatch) {
    const unsigned per = gridDim.x / NRT_NXCD;
    const unsigned kx = blockIdx.x % NRT_NXCD, jx = blockIdx.x / NRT_NXCD;
    const unsigned nT2 = nblk / nZc;                        // (x,y) patches
    const unsigned per2 = per / nZc;                        // patches owned by one XCD
    unsigned t, zc;
    int b = blockIdx.y;
    if (lreg > 0) {
        // region order: the blocks an XCD runs together are the 2^lreg x 2^lreg patches of one compact (x,y) window
        // of one volume, all marching through z; their shared halo rows meet in that XCD's L2 (grid.y == 1)
        const unsigned R = 1u << lreg, nTx = nT2 / nTy;
        const unsigned nRx = (nTx + R - 1) / R, nRy = (nTy + R - 1) / R;
        const unsigned nT2p = nRx * nRy * R * R;
        const unsigned u = kx * per + jx, U = (unsigned)nbatch * nZc * nT2p;
        if (u >= U) return;
        const unsigned tp = u % nT2p, rest = u / nT2p;
        zc = rest % nZc;
        b = (int)(rest / nZc);
        const unsigned reg = tp / (R * R), w = tp % (R * R);
        const unsigned tx = (reg / nRy) * R + w / R, ty = (reg % nRy) * R + w % R;
        if (tx >= nTx || ty >= nTy) return;
        t = tx * nTy + ty;
    } else {
        if (zc_outer) { zc = jx / per2; t = kx * per2 + jx % per2; }
        else { zc = jx % nZc; t = kx * per2 + jx / nZc; }
        if (t >= nT2) return;
    }
    const char *vol = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    nrt_f4 *out = (nrt_f4 *)((float *)a.out + (long long)b * a.out_bs);

    const unsigned ty = t % nTy, tx = t / nTy;
    const int lane = threadIdx.x & 63;
    const int lg = lane & 7;                 // 16-byte slice of the 128-byte row
    const int j = lane >> 3;                 // group within the wave: 2(x) x 4(y)
    const int w = threadIdx.x >> 6;          // wave within the block: WX(x) x WY(y)
    const int x = (int)tx * (2 * WX) + (w / WY) * 2 + (j >> 2);
    const int y = (int)ty * (4 * WY) + (w % WY) * 4 + (j & 3);
    if (x >= a.O[0] || y >= a.O[1]) return;  // whole lane-groups leave together; no block barrier below
    const int zbeg = (int)zc * LZ;
    const int zend = min(zbeg + LZ, a.O[2]);
    const long long colq = ((long long)x * a.O[1] + y) * a.O[2];   // output voxel index at z = 0
    const int grp_base = lane & 56;
    const float *locq = locb ? locb + colq * 3 : nullptr;
    nrt_f4 *outq = out + colq * 8;

    nrt_f4 P0[4], P1[4], P2[4];
    ZIdx cur;
    ZShift sh = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    if (MODE != NRT_LOC_LINSPACE) {
        {
            const int nrem = (zend - zbeg) * 3;
            const float *sp = locq + (long long)zbeg * 3;
            sh.s0 = (lg < nrem) ? __builtin_nontemporal_load(sp + lg) : 0.0f;
            sh.s1 = (lg + 8 < nrem) ? __builtin_nontemporal_load(sp + lg + 8) : 0.0f;
            sh.s2 = (lg + 16 < nrem) ? __builtin_nontemporal_load(sp + lg + 16) : 0.0f;
        }
        if (zbeg + 8 < zend) {
            const int nrem = (zend - zbeg - 8) * 3;
            const float *sp = locq + (long long)(zbeg + 8) * 3;
            sh.n0 = (lg < nrem) ? __builtin_nontemporal_load(sp + lg) : 0.0f;
            sh.n1 = (lg + 8 < nrem) ? __builtin_nontemporal_load(sp + lg + 8) : 0.0f;
            sh.n2 = (lg + 16 < nrem) ? __builtin_nontemporal_load(sp + lg + 16) : 0.0f;
        }
        sx = __shfl(sh.s0, grp_base | 0, 64);
        sy = __shfl(sh.s0, grp_base | 1, 64);
        sz = __shfl(sh.s0, grp_base | 2, 64);
    }
    zrun_index<MODE>(a, x, y, zbeg, sx, sy, sz, cur);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned col = zrun_col(a, cur, c);
        P0[c] = zrun_row(vol, col + (unsigned)cur.iz0, lg);
        P1[c] = zrun_row(vol, col + (unsigned)cur.iz1, lg);
    }
    int z = zbeg;
    while (true) {
        if (zrun_step<MODE>(a, vol, locq, outq, x, y, z, zbeg, zend, lg, grp_base, cur, sh, P0, P1, P2)) break;
        ++z;
        if (zrun_step<MODE>(a, vol, locq, outq, x, y, z, zbeg, zend, lg, grp_base, cur, sh, P1, P2, P0)) break;
        ++z;
        if (zrun_step<MODE>(a, vol, locq, outq, x, y, z, zbeg, zend, lg, grp_base, cur, sh, P2, P0, P1)) break;
        ++z;
    }
}

// ============================================================================================
// tile: G = C/4 lanes per voxel, 3-D output tiles, two voxels per lane-group in flight.
//
// Measured on MI355X (profiles/archive/r01_session1): the one-voxel-at-a-time kernels above run at ~3.3 TB/s
// although the chip moves 9 TB/s of rows in the incoherent worst case -- they are bound by the
// dependent chain  loc load -> address -> 8 row loads -> blend  (two HBM latencies per voxel), and a
// z-line traversal re-fetches every row for its x-neighbour line (FETCH_SIZE 1.9x algorithmic).  So:
//  * a block owns a TX x TY x TZ output tile (tz fastest: a wave64 still reads 64/G consecutive-z rows
//    = one contiguous run) and sweeps it in passes, so x/y/z neighbours re-use rows through L1/L2;
//  * tiles are dealt to XCDs in contiguous x-slabs, and inside a slab either z-fastest or z-outermost
//    so that the blocks resident on one XCD form a compact brick;
//  * software pipeline, depth 2: loc of pass p+2 and the 8 rows of pass p+1 are in flight while pass p
//    is blended; every load is unconditional (edge voxels clamp their address, only the store is
//    predicated) so the compiler's vmcnt counts stay exact and nothing drains the queue.
// ============================================================================================
template <int G, int MODE>
__global__ __launch_bounds__(256) void interpn_tile(InterpArgs a, TileGeom tg) {
    constexpr int NG = 256 / G;
    const unsigned per = tg.per2 * tg.nTz;                     // blocks per XCD
    const unsigned k = blockIdx.x % NRT_NXCD, j = blockIdx.x / NRT_NXCD;
    if (j >= per) return;
    unsigned t2l, tzi;
    if (tg.z_outer) { tzi = j / tg.per2; t2l = j % tg.per2; }
    else { tzi = j % tg.nTz; t2l = j / tg.nTz; }
    const unsigned t2 = k * tg.per2 + t2l;
    if (t2 >= tg.nT2) return;
    const int x0 = (int)(t2 / tg.nTy) << tg.ltx, y0 = (int)(t2 % tg.nTy) << tg.lty, z0 = (int)tzi * tg.tz;

    const int b = blockIdx.y;
    const char *volb = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    nrt_f4 *out = (nrt_f4 *)((float *)a.out + (long long)b * a.out_bs);
    const int lg = threadIdx.x % G;
    const int g = threadIdx.x / G;
    const int npass = tg.plane_major ? tg.tz : (1 << (tg.ltx + tg.lty + tg.ltz)) / NG;
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];

    // output voxel of (pass, this lane-group); coordinates clamped into the volume for addressing
    auto voxel = [&](int pass, int (&qd)[NRT_MAXD], bool &valid) {
        int x, y, z;
        tile_voxel(tg, NG, pass, g, x0, y0, z0, x, y, z);
        valid = (x < a.O[0]) && (y < a.O[1]) && (z < a.O[2]);
        qd[0] = min(x, a.O[0] - 1); qd[1] = min(y, a.O[1] - 1); qd[2] = min(z, a.O[2] - 1);
    };
    auto fetch_loc = [&](int pass, float (&p)[NRT_MAXD]) {
        int qd[NRT_MAXD]; bool valid;
        voxel(pass, qd, valid);
        const unsigned q = ((unsigned)qd[0] * (unsigned)a.O[1] + (unsigned)qd[1]) * (unsigned)a.O[2] + (unsigned)qd[2];
        if (MODE != NRT_LOC_LINSPACE) {
            const float *lp = locb + (long long)q * 3;
            p[0] = lp[0]; p[1] = lp[1]; p[2] = lp[2];
        }
    };
    // finish the location of `pass` from the prefetched loc; compute corners, weights and row offsets
    auto prepare = [&](int pass, const float (&praw)[NRT_MAXD], TileMeta &m, unsigned (&off)[8]) {
        int qd[NRT_MAXD];
        voxel(pass, qd, m.valid);
        m.q = ((unsigned)qd[0] * (unsigned)a.O[1] + (unsigned)qd[1]) * (unsigned)a.O[2] + (unsigned)qd[2];
        float p[NRT_MAXD];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (MODE == NRT_LOC_ABSOLUTE) p[d] = praw[d];
            else if (MODE == NRT_LOC_SHIFT) p[d] = nrt_add((float)qd[d], praw[d]);
            else p[d] = (qd[d] == 0) ? 0.0f
                      : ((qd[d] == a.O[d] - 1) ? (float)(a.S[d] - 1) : nrt_mul(a.delta[d], (float)qd[d]));
        }
        int i0[3], i1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], m.w0[d], m.w1[d]);
        m.oob = a.has_fill ? out_of_bounds<3>(a, p) : false;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const unsigned ix = (corner & 4) ? i1[0] : i0[0];
            const unsigned iy = (corner & 2) ? i1[1] : i0[1];
            const unsigned iz = (corner & 1) ? i1[2] : i0[2];
            off[corner] = (((ix * SY + iy) * SZ + iz) * (unsigned)G + (unsigned)lg) * 16u;
        }
    };
    auto load_rows = [&](const unsigned (&off)[8], nrt_f4 (&R)[8]) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) R[corner] = *(const nrt_f4 *)(volb + (size_t)off[corner]);
    };
    auto finish = [&](const TileMeta &m, const nrt_f4 (&R)[8]) {
        nrt_f4 acc = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float wt = nrt_mul(nrt_mul((corner & 4) ? m.w1[0] : m.w0[0], (corner & 2) ? m.w1[1] : m.w0[1]),
                                     (corner & 1) ? m.w1[2] : m.w0[2]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = nrt_add(acc[c], nrt_mul(wt, R[corner][c]));
        }
        if (a.has_fill) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = apply_fill(acc[c], m.oob, a.fill_f);
        }
        if (m.valid) __builtin_nontemporal_store(acc, &out[(long long)m.q * G + lg]);
    };

    // Issue order in steady state:  L(p+1) R(p) | L(p+2) R(p+1) | L(p+3) R(p+2) ...   (L = loc, R = 8 rows)
    // so that waiting for L(p+1) leaves R(p) in flight (vmcnt 8) and blending pass p leaves L(p+2),
    // R(p+1) in flight (vmcnt 11): nothing ever drains the queue.
    nrt_f4 Ra[8], Rb[8];
    TileMeta Ma, Mb;
    unsigned off[8];
    float pn[NRT_MAXD] = {0.0f, 0.0f, 0.0f};
    const int last = npass - 1;
    // No load below sits under a condition: passes beyond the tile re-address its last pass (cache hits)
    // and only their store is suppressed.
    // sched_barrier(0) pins the issue order: without it hipcc sinks the row loads below the previous
    // pass's blend (to save registers) and the pipeline degenerates to one pass in flight.
    fetch_loc(0, pn);
    prepare(0, pn, Ma, off);
    fetch_loc(min(1, last), pn);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(off, Ra);
    __builtin_amdgcn_sched_barrier(0);
    for (int pass = 0; pass < npass; pass += 2) {
        prepare(min(pass + 1, last), pn, Mb, off);
        Mb.valid = Mb.valid && (pass + 1 < npass);
        __builtin_amdgcn_sched_barrier(0);
        fetch_loc(min(pass + 2, last), pn);
        load_rows(off, Rb);
        __builtin_amdgcn_sched_barrier(0);
        finish(Ma, Ra);
        __builtin_amdgcn_sched_barrier(0);
        prepare(min(pass + 2, last), pn, Ma, off);
        Ma.valid = Ma.valid && (pass + 2 < npass);
        __builtin_amdgcn_sched_barrier(0);
        fetch_loc(min(pass + 3, last), pn);
        load_rows(off, Ra);
        __builtin_amdgcn_sched_barrier(0);
        finish(Mb, Rb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int MODE, int METHOD, typename T>
void launch_generic_d(const InterpArgs &a, int ndim, int batch, hipStream_t st) {
    const unsigned long long total = (unsigned long long)a.nout * (unsigned)a.C;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 256u * 16u) blocks = 256u * 16u;     // grid-stride the rest
    dim3 grid(blocks, batch);
    switch (ndim) {
        case 1: hipLaunchKernelGGL((interpn_generic<1, MODE, METHOD, T>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((interpn_generic<2, MODE, METHOD, T>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((interpn_generic<3, MODE, METHOD, T>), grid, dim3(256), 0, st, a); break;
    }
}

template <int METHOD, typename T>
void launch_generic(const InterpArgs &a, int ndim, int batch, int mode, hipStream_t st) {
    switch (mode) {
        case NRT_LOC_ABSOLUTE: launch_generic_d<NRT_LOC_ABSOLUTE, METHOD, T>(a, ndim, batch, st); break;
        case NRT_LOC_SHIFT: launch_generic_d<NRT_LOC_SHIFT, METHOD, T>(a, ndim, batch, st); break;
        default: launch_generic_d<NRT_LOC_LINSPACE, METHOD, T>(a, ndim, batch, st); break;
    }
}

template <int G, int MODE>
void launch_rows_m(const InterpArgs &a, int batch, int method, int tune, hipStream_t st) {
    constexpr unsigned NG = 256 / G;
    unsigned passes = tune > 0 ? (unsigned)tune : 4u;
    const unsigned vpb = NG * passes;
    const unsigned nblk = (a.nout + vpb - 1) / vpb;
    dim3 grid(nrt_xcd_grid(nblk), batch);
    if (method == NRT_INTERP_NEAREST)
        hipLaunchKernelGGL((interpn_rows<G, MODE, NRT_INTERP_NEAREST>), grid, dim3(256), 0, st, a, nblk, vpb);
    else
        hipLaunchKernelGGL((interpn_rows<G, MODE, NRT_INTERP_LINEAR>), grid, dim3(256), 0, st, a, nblk, vpb);
}

template <int G>
void launch_rows(const InterpArgs &a, int batch, int mode, int method, int tune, hipStream_t st) {
    switch (mode) {
        case NRT_LOC_ABSOLUTE: launch_rows_m<G, NRT_LOC_ABSOLUTE>(a, batch, method, tune, st); break;
        case NRT_LOC_SHIFT: launch_rows_m<G, NRT_LOC_SHIFT>(a, batch, method, tune, st); break;
        default: launch_rows_m<G, NRT_LOC_LINSPACE>(a, batch, method, tune, st); break;
    }
}

// ============================================================================================
// lds: few channels (C <= 4, images and displacement fields), 3-D, linear.
//
// With 4..16 bytes per source voxel the 8 corner gathers of the generic kernel are 8 scattered dword loads per
// output element (measured 0.13-0.19 of the HBM roof at 160^3).  Here a block owns an 8 x 8 x 16 output tile, finds
// the bounding box of the tile's corner indices (block min/max), copies that box of the source into LDS with
// coalesced loads (every source voxel is fetched once per tile: (7*1.5+2)^2 * (15*1.5+2) * 4 B = 17 KB at C = 1), and
// gathers the corners from LDS.  A tile whose box does not fit (rough fields) gathers from global memory instead.
// Same float32 op sequence as interpn_generic => bit-identical results; optional addend epilogue (compose / VecInt).
// ============================================================================================
// tile 8 x 8 x 16 (C <= 2) or 4 x 8 x 16 (C = 3, 4); LDS sized for the box of a field with gradient ~0.5 so that
// 4..8 blocks share a CU (the phases of a tile -- locations, box copy, gather -- are latency-bound on their own)
template <int C> struct LdsCfg {
    static constexpr int LTX = C <= 2 ? 3 : 2;
    static constexpr int CL = C == 3 ? 4 : C;                  // channel stride in LDS: one ds_read_b32/b64/b128 per corner
    static constexpr int CAP = C == 1 ? 5120 : 10240;          // floats (20 / 40 KB)
};

bool rows_supported(int channels) {
    if (channels % 4) return false;
    const int g = channels / 4;
    return g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64;
}

void launch_rows_any(const InterpArgs &a, int batch, int mode, int method, int tune, hipStream_t st) {
    switch (a.C / 4) {
        case 1: launch_rows<1>(a, batch, mode, method, tune, st); break;
        case 2: launch_rows<2>(a, batch, mode, method, tune, st); break;
        case 4: launch_rows<4>(a, batch, mode, method, tune, st); break;
        case 8: launch_rows<8>(a, batch, mode, method, tune, st); break;
        case 16: launch_rows<16>(a, batch, mode, method, tune, st); break;
        case 32: launch_rows<32>(a, batch, mode, method, tune, st); break;
        default: launch_rows<64>(a, batch, mode, method, tune, st); break;
    }
}

template <int WX, int WY>
void launch_zrun_w(const InterpArgs &a, int batch, int mode, int LZ, int zc_outer, int lreg, unsigned dyn, hipStream_t st) {
    const unsigned nTx = (a.O[0] + 2 * WX - 1) / (2 * WX), nTy = (a.O[1] + 4 * WY - 1) / (4 * WY);
    const unsigned nZc = (a.O[2] + LZ - 1) / LZ;
    const unsigned per2 = (nTx * nTy + NRT_NXCD - 1) / NRT_NXCD;       // patches per XCD
    const unsigned nblk = nTx * nTy * nZc;
    dim3 grid(NRT_NXCD * per2 * nZc, batch), blk(WX * WY * 64);
    if (lreg > 0) {
        const unsigned R = 1u << lreg;
        const unsigned nT2p = ((nTx + R - 1) / R) * ((nTy + R - 1) / R) * R * R;
        grid = dim3(nrt_xcd_grid(nT2p * nZc * (unsigned)batch), 1);
    }
#define NRT_ZRUN(MODE)                                                                                              \
    do {                                                                                                            \
        if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)interpn_zrun_c32<MODE, WX, WY>,               \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);       \
        hipLaunchKernelGGL((interpn_zrun_c32<MODE, WX, WY>), grid, blk, dyn, st, a, LZ, nTy, nZc, nblk, zc_outer,   \
                           lreg, batch);                                                                            \
    } while (0)
    switch (mode) {
        case NRT_LOC_ABSOLUTE: NRT_ZRUN(NRT_LOC_ABSOLUTE); break;
        case NRT_LOC_SHIFT: NRT_ZRUN(NRT_LOC_SHIFT); break;
        default: NRT_ZRUN(NRT_LOC_LINSPACE); break;
    }
#undef NRT_ZRUN
}

// tune = LZ | zc_outer << 12 | patch << 16 | lreg << 20 | lds_kb << 24
//   patch 0: 4x8, 1: 8x8, 2: 8x16, 3: 4x16 columns per block; lreg > 0: region order (2^lreg x 2^lreg patches per
//   region); lds_kb: unused dynamic LDS per block, caps the blocks per CU so that an XCD's L2 holds one region's rows
void launch_zrun(const InterpArgs &a, int batch, int mode, int variant, int tune, hipStream_t st) {
    (void)variant;
    int LZ = tune & 0xfff;
    const int zc_outer = (tune >> 12) & 1, patch = (tune >> 16) & 3, lreg = (tune >> 20) & 7;
    const unsigned dyn = (unsigned)((tune >> 24) & 0x7f) * 1024u;
    if (LZ <= 0 || LZ > a.O[2]) LZ = a.O[2];
    switch (patch) {
        case 1: launch_zrun_w<4, 2>(a, batch, mode, LZ, zc_outer, lreg, dyn, st); break;
        case 2: launch_zrun_w<4, 4>(a, batch, mode, LZ, zc_outer, lreg, dyn, st); break;
        case 3: launch_zrun_w<2, 4>(a, batch, mode, LZ, zc_outer, lreg, dyn, st); break;
        default: launch_zrun_w<2, 2>(a, batch, mode, LZ, zc_outer, lreg, dyn, st); break;
    }
}

template <int G>
void launch_tile(const InterpArgs &a, int batch, int mode, int tune, hipStream_t st) {
    TileGeom tg;
    unsigned ntiles;
    tile_geometry(a.O, G, tune, 2 | (2 << 4) | (4 << 8) | (1 << 12), tg, ntiles);
    dim3 grid(ntiles, batch);
    switch (mode) {
        case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((interpn_tile<G, NRT_LOC_ABSOLUTE>), grid, dim3(256), 0, st, a, tg); break;
        case NRT_LOC_SHIFT: hipLaunchKernelGGL((interpn_tile<G, NRT_LOC_SHIFT>), grid, dim3(256), 0, st, a, tg); break;
        default: hipLaunchKernelGGL((interpn_tile<G, NRT_LOC_LINSPACE>), grid, dim3(256), 0, st, a, tg); break;
    }
}

void launch_tile_any(const InterpArgs &a, int batch, int mode, int tune, hipStream_t st) {
    switch (a.C / 4) {
        case 1: launch_tile<1>(a, batch, mode, tune, st); break;
        case 2: launch_tile<2>(a, batch, mode, tune, st); break;
        case 4: launch_tile<4>(a, batch, mode, tune, st); break;
        case 8: launch_tile<8>(a, batch, mode, tune, st); break;
        case 16: launch_tile<16>(a, batch, mode, tune, st); break;
        case 32: launch_tile<32>(a, batch, mode, tune, st); break;
        default: launch_tile<64>(a, batch, mode, tune, st); break;
    }
}

}  // namespace

// Default kernel choice, set from measurements on MI355X (profiles/): see DESIGN.md.
static int g_auto_c32_variant = 3;
static int g_auto_c32_tune = 20 | (1 << 16);     // z-run, 8x8 patch, z-chunks of 20 (profiles/r01: sweeps)

extern "C" int nrt_interpn_f32_ex(const float *vol, const float *loc, float *out, int ndim, const int *vol_shape,
                                  const int *out_shape, int channels, int batch, long long vol_batch_stride,
                                  long long loc_batch_stride, int loc_mode, int method, int has_fill,
                                  float fill_value, int variant, int tune, void *stream) {
    InterpArgs a;
    int rc = fill_args(a, vol, loc, out, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    if (method != NRT_INTERP_LINEAR && method != NRT_INTERP_NEAREST) return NRT_ERR_INVALID_ARG;
    a.fill_f = fill_value;
    if (a.nout == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    const bool aligned = (((uintptr_t)vol | (uintptr_t)out) & 15) == 0 &&
                         ((vol_batch_stride * 4) % 16 == 0);
    const bool can_rows = rows_supported(channels) && aligned && ndim == 3;
    unsigned long long vol_bytes = 4ull * channels;
    for (int d = 0; d < ndim; ++d) vol_bytes *= (unsigned long long)vol_shape[d];
    const bool can_zrun = can_rows && channels == 32 && ndim == 3 && method == NRT_INTERP_LINEAR &&
                          vol_bytes < (1ull << 32);
    const bool can_lean = (method == NRT_INTERP_LINEAR ? (loc_mode == NRT_LOC_LINSPACE || loc) : (loc_mode != NRT_LOC_LINSPACE && loc)) &&
                          nrt_lean_supported(a.S, a.O, channels, ndim, vol, loc, out, vol_batch_stride, loc_batch_stride);
    const bool variant_was_auto = variant == 0;
    if (variant == 0) {
        if (can_zrun) {
            // Displacement fields and absolute locations: the wave-cache kernel (variant 10, fused_wc.h) since its round-5 schedule --
            // per 4 x 160^3 x 32 on the bench field 0.96 ms against 1.25 for the z-run register kernel, 0.89 / 0.89 on an identity
            // field, 1.87 / 2.15 on an incoherent one (tools/wc_bench.py, profiles/r05_lab/wc_schedule_ab.jsonl).  Regular grids
            // (Resize) keep the z-run kernel: consecutive outputs share their corner planes there, which is what it is built for.
            if (loc_mode != NRT_LOC_LINSPACE && tune == 0 && nrt_wc_interpn_supported(&a, batch)) variant = 10;
            else { variant = g_auto_c32_variant; if (tune == 0) tune = (variant >= 3) ? g_auto_c32_tune : 0; }
        }
        else if (can_lean) variant = 8;
        else if (can_rows && method == NRT_INTERP_LINEAR && vol_bytes < (1ull << 32) && a.nout >= 4096) {
            // 8 / 16 / 64 ... channels (feature maps): the pipelined 3-D tiles beat the row kernel at 4 x 160^3 -- C = 8 0.490 vs
            // 0.636 ms, C = 16 0.787 vs 0.907, C = 64 2.70 vs 3.09 (tools/midc_sweep.py, profiles/archive/r02_smallc/midc_sweep.jsonl)
            variant = 5;
            if (tune == 0) tune = channels <= 16 ? (2 | (3 << 4) | (4 << 8) | (1 << 12)) : 0;
        }
        else if (can_rows) variant = 2;
        else if (channels % 4 == 0 && aligned) {
            // 12, 20, 24 ... channels: the rank-templated kernel of interpn_any.hip with 4 channels per thread (16-byte accesses,
            // corner arithmetic once per group); the same operations in the same order as interpn_generic
            return nrt_interpn_any(vol, loc, out, NRT_DT_F32, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                                   loc_batch_stride, loc_mode, 0, method, has_fill, (double)fill_value, stream);
        }
        else variant = 1;
    }
    if ((variant == 3 || variant == 4) && !can_zrun) return NRT_ERR_UNSUPPORTED;
    const bool can_tile = can_rows && method == NRT_INTERP_LINEAR && vol_bytes < (1ull << 32);
    if (variant == 5 && !can_tile) return NRT_ERR_UNSUPPORTED;
    if (variant == 2 && !can_rows) return NRT_ERR_UNSUPPORTED;
    switch (variant) {
        case 1:
            if (method == NRT_INTERP_LINEAR) launch_generic<NRT_INTERP_LINEAR, float>(a, ndim, batch, loc_mode, st);
            else launch_generic<NRT_INTERP_NEAREST, float>(a, ndim, batch, loc_mode, st);
            break;
        case 2: launch_rows_any(a, batch, loc_mode, method, tune, st); break;
        case 3:
        case 4: launch_zrun(a, batch, loc_mode, variant, tune, st); break;
        case 5: launch_tile_any(a, batch, loc_mode, tune, st); break;
        case 8:                             // few channels: tile form (one voxel per lane, corners through the texture unit)
            if (!can_lean) return NRT_ERR_UNSUPPORTED;
            return nrt_lean_launch(&a, batch, loc_mode, method == NRT_INTERP_NEAREST ? 1 : 0, st, variant_was_auto ? NRT_LEAN_FORM_AUTO : NRT_LEAN_FORM_TILE);
        case 11:                            // few channels, per-voxel locations, linear: box form (source box of a tile staged in LDS)
            if (!can_lean || method != NRT_INTERP_LINEAR || loc_mode == NRT_LOC_LINSPACE) return NRT_ERR_UNSUPPORTED;
            return nrt_lean_launch(&a, batch, loc_mode, 0, st, NRT_LEAN_FORM_BOX);
        case 10:                            // wave-private LDS row cache (fused_wc.h) without the Dice half
            if (!can_zrun || !nrt_wc_interpn_supported(&a, batch)) return NRT_ERR_UNSUPPORTED;
            return nrt_wc_interpn_launch(&a, batch, loc_mode, st);
        default: return NRT_ERR_INVALID_ARG;
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_interpn_add_f32(const float *vol, const float *loc, const float *addend, float *out, int ndim,
                                   const int *vol_shape, const int *out_shape, int channels, int batch,
                                   long long vol_batch_stride, long long loc_batch_stride,
                                   long long addend_batch_stride, int loc_mode, int has_fill, float fill_value,
                                   void *stream) {
    if (!addend) return NRT_ERR_INVALID_ARG;
    InterpArgs a;
    int rc = fill_args(a, vol, loc, out, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    a.fill_f = fill_value;
    a.addend = addend; a.addend_bs = addend_batch_stride;
    if (a.nout == 0) return NRT_OK;
    if ((loc_mode == NRT_LOC_LINSPACE || loc) && (((uintptr_t)addend) & 15) == 0 && (addend_batch_stride * 4) % 16 == 0 &&
        nrt_lean_supported(a.S, a.O, channels, ndim, vol, loc, out, vol_batch_stride, loc_batch_stride))
        return nrt_lean_launch(&a, batch, loc_mode, 0, stream);
    launch_generic<NRT_INTERP_LINEAR, float>(a, ndim, batch, loc_mode, nrt_stream(stream));
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_interpn_f32(const float *vol, const float *loc, float *out, int ndim, const int *vol_shape,
                               const int *out_shape, int channels, int batch, long long vol_batch_stride,
                               long long loc_batch_stride, int loc_mode, int method, int has_fill,
                               float fill_value, void *stream) {
    return nrt_interpn_f32_ex(vol, loc, out, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                              loc_batch_stride, loc_mode, method, has_fill, fill_value, 0, 0, stream);
}

extern "C" int nrt_interpn_nearest_i32(const int32_t *vol, const float *loc, int32_t *out, int ndim,
                                       const int *vol_shape, const int *out_shape, int channels, int batch,
                                       long long vol_batch_stride, long long loc_batch_stride, int loc_mode,
                                       int has_fill, int32_t fill_value, void *stream) {
    InterpArgs a;
    int rc = fill_args(a, vol, loc, out, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    a.fill_i = fill_value;
    if (a.nout == 0) return NRT_OK;
    if (loc_mode != NRT_LOC_LINSPACE && loc &&
        nrt_lean_supported(a.S, a.O, channels, ndim, vol, loc, out, vol_batch_stride, loc_batch_stride))
        return nrt_lean_launch(&a, batch, loc_mode, 2, stream);        // label maps (C <= 4): the lean tile kernel
    launch_generic<NRT_INTERP_NEAREST, int32_t>(a, ndim, batch, loc_mode, nrt_stream(stream));
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
