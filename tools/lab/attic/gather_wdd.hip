// Wave-window de-duplicating gather for 32-channel fp32 volumes on gfx950 (MI355X): interpn / SpatialTransformer / Resize
// (neurite/tf/utils/utils.py:137-191, linear) with or without the fused soft-Dice sums (neurite/tf/metrics.py:476-477).
//
// Why: the tri-linear blend of one output voxel reads 8 corner rows of 128 B.  Neighbouring outputs share most of them, but
// the per-CU vector L1 (32 KB) only catches the z-neighbour overlap of back-to-back instructions, so the row kernels
// push ~6.5 L1-miss lines per voxel through the TA/L1 pipe at ~5 clk per line and sit on that roof, not on HBM
// (profiles/r01_session30).  Here the de-duplication is explicit and exact, per wave, with no block barrier
// and no atomics (returning LDS atomics retire ~1 lane per clock per CU: a ds_cmpswap hash cost 0.9 ms per launch,
// profiles/r02):
//
//   a work-group is ONE wave.  It owns a (y,z) patch and marches along x in windows of 64 output voxels
//   (WX x PY x PZ).  Per window:
//   1. lane-per-voxel: location, floor/clip/weights (utils.py:139-153) and the 8 corner row indices -- once per voxel, not
//      once per lane of an 8-lane group as in the row kernels;
//   2. the bounding box of the window's corner rows in the source volume (packed 16-bit min/max over the wave, DPP) is
//      laid over a byte map in LDS (<= 2048 entries); every lane marks its 8 corners with a plain byte store (all
//      writers store the same 1: idempotent, no atomic); every lane then scans 32 bytes of the map, a wave prefix sum
//      turns marks into positions in the row buffer, written back in place; a second byte load per corner is the
//      lookup.  ~170 of the 512 requests are distinct on the benchmark field.  A window whose box or whose distinct rows
//      exceed the buffers is halved (32, then 16 voxels; a 16-voxel piece whose box is still too large is fetched
//      without de-duplication);
//   3. every distinct row is fetched ONCE, straight into LDS (global_load_lds_dwordx4: 8 lanes x 16 B per row, 8 rows
//      per instruction, no VGPRs, all of the window's rows in flight together);
//   4. blend: an 8-lane group per voxel reads its 8 corner rows from LDS (ds_read_b128) and blends them with the
//      reference's op sequence (one rounding per op, corners in itertools.product order => bit-identical to the
//      other kernels and to the oracle); the row is stored (16 B per lane, 1 KB per wave instruction) and/or
//      multiplied into the Dice sums.
//
// Traffic through the TA per voxel: ~2.7 corner rows + 1 output/fixed row + 0.1 loc, instead of 8-9.
#include <stdlib.h>

#include "interpn_core.h"
#include "wdd.h"

namespace {

constexpr int WD_CAP = 2048;          // byte-map entries (bounding box of a window's corner rows)

struct WdGeom {
    unsigned npy, npz;                // patches along y and z
    unsigned ncol;                    // patches per volume, padded to whole regions
    unsigned nseg, seglen;            // x segments per patch, planes per segment
    unsigned nbatch;
    int lry, lrz;                     // log2 region extent in patches
};

struct WdArgs {
    WddCall c;
    WdGeom g;
};

template <int NR>
struct WdLds {
    static constexpr int MAP = NR * 128;          // unsigned char[WD_CAP]: mark, then position in the row buffer
    static constexpr int LIST = MAP + WD_CAP;     // unsigned[NR]: position -> row index
    static constexpr int META = LIST + NR * 4;    // per voxel 64 B: 8 x u32 LDS byte offsets (+ flags), 8 corner weights
    static constexpr int TOTAL = (META + 64 * 64 + 255) & ~255;
};

// one LDS-DMA instruction: every active lane copies 16 B from base + voff to lds_dst + 16 * lane (lds_dst wave-uniform).
// M0 is not used by anything else in this kernel (checked in the ISA), so it is not saved.
__device__ __forceinline__ void wd_glds16(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

typedef unsigned short wd_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned wd_pkmin(unsigned a, unsigned b) {
    const wd_us2 m = __builtin_elementwise_min(__builtin_bit_cast(wd_us2, a), __builtin_bit_cast(wd_us2, b));
    return __builtin_bit_cast(unsigned, m);
}

// minimum of both 16-bit halves over the wave -> two scalars
__device__ __forceinline__ void wd_wave_pkmin(unsigned v, unsigned &lo, unsigned &hi) {
    v = wd_pkmin(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = wd_pkmin(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = wd_pkmin(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = wd_pkmin(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));    // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16),
                   c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    lo = min(min(a & 0xffffu, b & 0xffffu), min(c & 0xffffu, d & 0xffffu));
    hi = min(min(a >> 16, b >> 16), min(c >> 16, d >> 16));
}

// inclusive prefix sum over the wave (row_shr 1/2/4/8, then row_bcast 15 / 31)
__device__ __forceinline__ unsigned wd_wave_scan(unsigned x) {
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

constexpr unsigned WD_F_VALID = 1u << 30, WD_F_OOB = 1u << 31;

template <int MODE, int LWX, int LPY, int LPZ, int NR, bool STORE, bool DICE, bool MINMAX, int WPB>
__global__ __launch_bounds__(64 * WPB) void gather_wdd(WdArgs A) {
    constexpr int SIZE0 = NR >= 160 ? 64 : 32;        // voxels per fetch + blend round (the row buffer holds NR rows)
    static_assert(LWX + LPY + LPZ == 6, "a window is 64 voxels");
    static_assert(NR >= 96 && NR <= 256 && NR % 8 == 0, "row buffer");
    constexpr int WX = 1 << LWX, PY = 1 << LPY, PZ = 1 << LPZ;
    constexpr int L = 32;
    using LD = WdLds<NR>;
    // WPB independent waves per work-group (no barrier anywhere): a work-group's waves are dealt to different SIMDs
    extern __shared__ __attribute__((aligned(16))) unsigned char wd_smem_all[];
    const unsigned wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *wd_smem = wd_smem_all + wave_in_block * (unsigned)LD::TOTAL;
    unsigned char *map = wd_smem + LD::MAP;
    unsigned *rowlist = (unsigned *)(wd_smem + LD::LIST);
    nrt_i4 *meta = (nrt_i4 *)(wd_smem + LD::META);
    const unsigned lds_rows = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)wd_smem);

    const WddCall &c = A.c;
    const WdGeom &gm = A.g;
    const int lane = threadIdx.x & 63, lg = lane & 7, g = lane >> 3;

    // ---- which (batch, x segment, patch) does this wave own?  XCD k owns a contiguous range of units; consecutive units
    //      fill one region of 2^lry x 2^lrz patches, so the waves resident on an XCD cover a compact (y,z) window that
    //      marches through x and the rows they share stay in that XCD's L2
    const unsigned kx = blockIdx.x % NRT_NXCD, jb = (blockIdx.x / NRT_NXCD) * WPB + wave_in_block;
    const unsigned per_batch = gm.ncol * gm.nseg, U = per_batch * gm.nbatch, perU = (gridDim.x / NRT_NXCD) * WPB;
    const unsigned u = kx * perU + jb;
    if (u >= U) return;
    const int b = (int)(u / per_batch);
    const unsigned prow = u % per_batch;
    const unsigned useg = prow / gm.ncol, ucol = prow % gm.ncol;
    const unsigned RY = 1u << gm.lry, RZ = 1u << gm.lrz, nRz = (gm.npz + RZ - 1) / RZ;
    const unsigned reg = ucol / (RY * RZ), wi = ucol % (RY * RZ);
    const unsigned cy = (reg / nRz) * RY + wi / RZ, cz = (reg % nRz) * RZ + wi % RZ;
    const int x0 = (int)(useg * gm.seglen), y0 = (int)cy << LPY, z0 = (int)cz << LPZ;
    int xlen = min((int)gm.seglen, c.O[0] - x0);
    if (cy >= gm.npy || cz >= gm.npz) xlen = 0;
    const int xend = x0 + max(xlen, 0);

    const char *volb = (const char *)(c.vol + (long long)b * c.vol_bs);
    const float *locb = c.loc ? c.loc + (long long)b * c.loc_bs : nullptr;
    nrt_f4 *outb = STORE ? (nrt_f4 *)(c.out + (long long)b * c.out_bs) : nullptr;
    const nrt_f4 *fixb = DICE ? (const nrt_f4 *)(c.fixed + (long long)b * c.out_bs) : nullptr;
    const unsigned SY = (unsigned)c.S[1], SZ = (unsigned)c.S[2], SYZ = SY * SZ;
    const unsigned O1 = (unsigned)c.O[1], O2 = (unsigned)c.O[2];

    // lane-per-voxel coordinates inside a window
    const int wz = lane & (PZ - 1), wy = (lane >> LPZ) & (PY - 1), wxl = lane >> (LPZ + LPY);
    const int yv = y0 + wy, zv = z0 + wz;
    const bool yz_ok = yv < c.O[1] && zv < c.O[2];
    const int yc = min(yv, c.O[1] - 1), zc = min(zv, c.O[2] - 1);
    const unsigned lg16 = (unsigned)lg * 16u;

    nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;

    nrt_i4 *mapq = (nrt_i4 *)map + 2 * lane;                  // this lane's 32 bytes of the map
    auto clear_map = [&]() {
        const nrt_i4 z = {0, 0, 0, 0};
        mapq[0] = z;
        mapq[1] = z;
    };
    float pn[3] = {0.0f, 0.0f, 0.0f};
    auto fetch_loc = [&](int xw) {
        if (MODE != NRT_LOC_LINSPACE) {
            const unsigned xq = (unsigned)min(xw + wxl, c.O[0] - 1);
            const unsigned q = nrt_mad24(nrt_mad24(xq, O1, (unsigned)yc), O2, (unsigned)zc);
            const float *lp = locb + (size_t)nrt_times3(q);
            pn[0] = lp[0]; pn[1] = lp[1]; pn[2] = lp[2];
        }
    };

    const int dbg = c.tune >> 24;                     // diagnostics only (bit 0: no de-duplication, 1: no row fetch, 2: no blend,
                                                      // 3: the Dice sums are replaced by per-phase clock counts)
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if (dbg & 8) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) ph[k] += t - tprev;
            tprev = t;
        }
    };
    clear_map();
    for (int i = lane; i < NR; i += 64) rowlist[i] = 0u;                   // unused slots of the last fetch instruction read row 0

    // ---- state of the window being indexed / de-duplicated (lane-per-voxel) ---------------------------------------------
    int i0x = 0, i1x = 0, i0y = 0, i1y = 0, i0z = 0, i1z = 0;
    bool ux = false, uy = false, uz = false, valid = false, oob = false;
    unsigned r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    nrt_f2 wt01 = {0, 0}, wt23 = {0, 0}, wt45 = {0, 0}, wt67 = {0, 0};
    // ---- 1. lane-per-voxel index arithmetic of the window at xw -----------------------------------------------------------
    auto index_window = [&](int xw) {
        const int x = xw + wxl;
        valid = yz_ok && x < xend;
        const int qd[3] = {min(x, c.O[0] - 1), yc, zc};
        float p[NRT_MAXD];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (MODE == NRT_LOC_ABSOLUTE) p[d] = pn[d];
            else if (MODE == NRT_LOC_SHIFT) p[d] = nrt_add((float)qd[d], pn[d]);
            else p[d] = (qd[d] == 0) ? 0.0f : ((qd[d] == c.O[d] - 1) ? (float)(c.S[d] - 1) : nrt_mul(c.delta[d], (float)qd[d]));
        }
        if (xw + WX < xend) fetch_loc(xw + WX);                            // next window's locations: in flight until then
        float w0x, w0y, w0z, w1x, w1y, w1z;
        corner_1d(p[0], c.S[0], i0x, i1x, w0x, w1x);
        corner_1d(p[1], c.S[1], i0y, i1y, w0y, w1y);
        corner_1d(p[2], c.S[2], i0z, i1z, w0z, w1z);
        oob = false;
        if (c.has_fill) {
#pragma unroll
            for (int d = 0; d < 3; ++d) oob = oob || (p[d] < 0.0f) || (p[d] > (float)(c.S[d] - 1));
        }
        // the upper corner of a dimension is the lower one or its successor (clipped at the border): row index and byte-map
        // index of corner k are those of corner 0 plus per-dimension increments
        ux = i1x != i0x; uy = i1y != i0y; uz = i1z != i0z;
        const unsigned r0 = nrt_mad24(nrt_mad24((unsigned)i0x, SY, (unsigned)i0y), SZ, (unsigned)i0z);
        const unsigned rdx = ux ? SYZ : 0u, rdy = uy ? SZ : 0u, rdz = uz ? 1u : 0u;
        r[0] = r0; r[1] = r0 + rdz; r[2] = r0 + rdy; r[3] = r[2] + rdz;
        r[4] = r0 + rdx; r[5] = r[4] + rdz; r[6] = r[4] + rdy; r[7] = r[6] + rdz;
        // corner weights (wx * wy) * wz in the reference's order (utils.py:1085-1092 prod_n), two per packed multiply
        const nrt_f2 wy2 = {w0y, w1y}, wz2 = {w0z, w1z};
        const nrt_f2 wxy0 = (nrt_f2){w0x, w0x} * wy2, wxy1 = (nrt_f2){w1x, w1x} * wy2;
        wt01 = (nrt_f2){wxy0[0], wxy0[0]} * wz2; wt23 = (nrt_f2){wxy0[1], wxy0[1]} * wz2;
        wt45 = (nrt_f2){wxy1[0], wxy1[0]} * wz2; wt67 = (nrt_f2){wxy1[1], wxy1[1]} * wz2;
    };

    // ---- 2. de-duplicate the corner rows of voxels [lo, lo + size) of the indexed window.  `size` shrinks (64 -> 32 -> 16)
    //         until the bounding box fits the byte map and the distinct rows fit the row buffer.  Leaves the row list in LDS
    //         and the positions in pos[]; returns the number of rows to fetch. -------------------------------------------------
    auto dedup = [&](int lo, int &size, unsigned (&pos)[8]) -> int {
        for (;;) {
            bool in_range = lane >= lo && lane < lo + size;
            bool act = valid && in_range;
            // 2a. bounding box of the corner rows (wave-uniform)
            unsigned bx, by, bz, nx1, ny1, nz1;
            wd_wave_pkmin(act ? ((unsigned)i0x | ((unsigned)i0y << 16)) : 0xffffffffu, bx, by);
            wd_wave_pkmin(act ? ((unsigned)i0z | ((0xffffu - (unsigned)i1x) << 16)) : 0xffffffffu, bz, nx1);
            wd_wave_pkmin(act ? ((0xffffu - (unsigned)i1y) | ((0xffffu - (unsigned)i1z) << 16)) : 0xffffffffu, ny1, nz1);
            const bool any_act = bx != 0xffffu;
            const unsigned ex = any_act ? (0xffffu - nx1) - bx + 1u : 0u, ey = (0xffffu - ny1) - by + 1u, ez = (0xffffu - nz1) - bz + 1u;
            const unsigned nbox = any_act ? ex * ey * ez : 0u;
            if ((nbox > (unsigned)WD_CAP || (dbg & 1)) && size > 16) { size >>= 1; continue; }
            int total;
            if (nbox > (unsigned)WD_CAP || (dbg & 1)) {                     // no de-duplication: 8 rows per voxel, 16 voxels
#pragma unroll
                for (int k = 0; k < 8; ++k) pos[k] = act ? (unsigned)((lane - lo) * 8 + k) : 0u;
                total = size * 8;
            } else {
                // 2b. mark, scan, look up
                const unsigned eyz = ey * ez, rdz = uz ? 1u : 0u;
                const unsigned idx0 = nrt_mad24(nrt_mad24((unsigned)i0x - bx, ey, (unsigned)i0y - by), ez, (unsigned)i0z - bz);
                const unsigned ddx = ux ? eyz : 0u, ddy = uy ? ez : 0u;
                unsigned idx[8];
                idx[0] = idx0; idx[1] = idx0 + rdz; idx[2] = idx0 + ddy; idx[3] = idx[2] + rdz;
                idx[4] = idx0 + ddx; idx[5] = idx[4] + rdz; idx[6] = idx[4] + ddy; idx[7] = idx[6] + rdz;
                unsigned w[8], inc[8], offs[8], off, incl;
                for (;;) {
                    if (act) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) map[idx[k]] = (unsigned char)1;
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                    const nrt_i4 q0 = mapq[0], q1 = mapq[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { w[j] = (unsigned)q0[j]; w[4 + j] = (unsigned)q1[j]; }
                    off = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        unsigned t = w[j] + (w[j] << 8);
                        t += t << 16;                                       // byte k = number of marks in bytes 0..k of this dword
                        inc[j] = t; offs[j] = off; off += t >> 24;
                    }
                    incl = wd_wave_scan(off);
                    total = (int)__builtin_amdgcn_readlane(incl, 63);
                    if (total <= NR || size <= 16) break;                   // wave-uniform
                    // more distinct rows than the buffer holds: same box, half the voxels
                    clear_map();
                    size >>= 1;
                    in_range = lane >= lo && lane < lo + size;
                    act = valid && in_range;
                }
                const unsigned base = incl - off;
                nrt_i4 p0, p1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned bo = base + offs[j];
                    const unsigned pb = inc[j] - w[j] + __builtin_amdgcn_perm(bo, bo, 0u);   // exclusive count, mod 256 per byte
                    if (j < 4) p0[j] = (int)pb; else p1[j - 4] = (int)pb;
                }
                mapq[0] = p0;
                mapq[1] = p1;
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) pos[k] = act ? (unsigned)map[idx[k]] : 0u;
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                clear_map();
            }
            if (act) {
#pragma unroll
                for (int k = 0; k < 8; ++k) rowlist[pos[k]] = r[k];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            return total;
        }
    };

    // ---- the pipeline.  A job is a piece [lo, lo + size) of a window.  Per iteration: start the row fetch of the current
    //      job (rows straight into LDS, fixed rows into registers), index / de-duplicate the NEXT job while those loads
    //      are in flight, then wait and blend the current job. -----------------------------------------------------------------
    if (xlen > 0) {
        fetch_loc(x0);
        stamp(-1);
        index_window(x0);
        stamp(0);
        int cxw = x0, clo = 0, csize = SIZE0, ctotal;
        unsigned cpos[8];
        // the per-voxel record of the current job, produced by the de-duplication, written to LDS when its turn comes
        nrt_i4 co0, co1;
        nrt_f4 cwa, cwb;
        auto make_record = [&](const unsigned (&pos)[8]) {
            co0[0] = (int)((pos[0] << 7) | (valid ? WD_F_VALID : 0u) | (oob ? WD_F_OOB : 0u));
            co0[1] = (int)(pos[1] << 7); co0[2] = (int)(pos[2] << 7); co0[3] = (int)(pos[3] << 7);
            co1[0] = (int)(pos[4] << 7); co1[1] = (int)(pos[5] << 7); co1[2] = (int)(pos[6] << 7); co1[3] = (int)(pos[7] << 7);
            cwa = (nrt_f4){wt01[0], wt01[1], wt23[0], wt23[1]};
            cwb = (nrt_f4){wt45[0], wt45[1], wt67[0], wt67[1]};
        };
        ctotal = dedup(clo, csize, cpos);
        make_record(cpos);
        stamp(1);
        for (;;) {
            // ---- 3. every distinct row of the current job once, straight into LDS ---------------------------------------
            {
                unsigned rr[NR / 8];
#pragma unroll
                for (int j = 0; j < NR / 8; ++j) rr[j] = rowlist[8 * j + g];
                if (!(dbg & 2)) {
#pragma unroll
                    for (int j = 0; j < NR / 8; ++j) {
                        if (8 * j < ctotal)                                  // wave-uniform; slots past the total re-fetch stale (valid) rows
                            wd_glds16(volb, (rr[j] << 7) + lg16, lds_rows + (unsigned)(8 * j) * 128u);
                    }
                }
            }
            if (lane >= clo && lane < clo + csize) {
                meta[4 * lane] = co0;
                meta[4 * lane + 1] = co1;
                ((nrt_f4 *)meta)[4 * lane + 2] = cwa;
                ((nrt_f4 *)meta)[4 * lane + 3] = cwb;
            }
            const int ilo = clo >> 3, ihi = (clo + csize) >> 3;
            nrt_f4 T[8];
            if (DICE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    T[i] = (nrt_f4){0, 0, 0, 0};
                    if (i >= ilo && i < ihi) {                               // wave-uniform
                        const int v = 8 * i + g;
                        const int vx = min(cxw + (v >> (LPZ + LPY)), c.O[0] - 1);
                        const int vy = min(y0 + ((v >> LPZ) & (PY - 1)), c.O[1] - 1), vz = min(z0 + (v & (PZ - 1)), c.O[2] - 1);
                        const unsigned q = nrt_mad24(nrt_mad24((unsigned)vx, O1, (unsigned)vy), O2, (unsigned)vz);
                        T[i] = __builtin_nontemporal_load(fixb + ((size_t)q * 8u + (unsigned)lg));
                    }
                }
            }
            stamp(2);
            // ---- next job: the rest of this window, or the next window ----------------------------------------------------
            int nxw = cxw, nlo = clo + csize, nsize = 0, ntotal = 0;
            unsigned npos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool have_next = true;
            if (nlo >= 64) {
                nxw = cxw + WX; nlo = 0;
                have_next = nxw < xend;
                if (have_next) { stamp(-1); index_window(nxw); stamp(0); }
            }
            if (have_next) {
                nsize = nlo ? min(SIZE0, nlo & -nlo) : SIZE0;              // a piece stays aligned to its size
                ntotal = dedup(nlo, nsize, npos);
                stamp(1);
            }
            // ---- 4. blend the current job from LDS -------------------------------------------------------------------------
            stamp(-1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            stamp(3);
            struct Rec { nrt_i4 o0, o1; nrt_f4 wa, wb; };
            auto ld_rec = [&](int i, Rec &m) {
                const int v = 8 * i + g;
                m.o0 = meta[4 * v]; m.o1 = meta[4 * v + 1];
                m.wa = ((const nrt_f4 *)meta)[4 * v + 2]; m.wb = ((const nrt_f4 *)meta)[4 * v + 3];
            };
            auto ld_rows = [&](const Rec &m, nrt_f4 (&R)[8]) {
                const unsigned off[8] = {(unsigned)m.o0[0] & 0xffffu, (unsigned)m.o0[1], (unsigned)m.o0[2], (unsigned)m.o0[3],
                                         (unsigned)m.o1[0], (unsigned)m.o1[1], (unsigned)m.o1[2], (unsigned)m.o1[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k) R[k] = *(const nrt_f4 *)(wd_smem + (off[k] + lg16));
            };
            auto blend_math = [&](int i, const Rec &m, const nrt_f4 (&R)[8]) {
                const int v = 8 * i + g;
                const bool vvalid = (unsigned)m.o0[0] & WD_F_VALID, voob = (unsigned)m.o0[0] & WD_F_OOB;
                const float wt[8] = {m.wa[0], m.wa[1], m.wa[2], m.wa[3], m.wb[0], m.wb[1], m.wb[2], m.wb[3]};
                nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
                for (int corner = 0; corner < 8; ++corner) {
                    const nrt_f2 w2 = {wt[corner], wt[corner]};
                    al = al + w2 * (nrt_f2){R[corner][0], R[corner][1]};
                    ah = ah + w2 * (nrt_f2){R[corner][2], R[corner][3]};
                }
                nrt_f4 acc = {al[0], al[1], ah[0], ah[1]};
                if (c.has_fill) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = apply_fill(acc[k], voob, c.fill);
                }
                if (vvalid) {
                    if (STORE) {
                        const unsigned vx = (unsigned)(cxw + (v >> (LPZ + LPY)));
                        const unsigned vy = (unsigned)(y0 + ((v >> LPZ) & (PY - 1))), vz = (unsigned)(z0 + (v & (PZ - 1)));
                        const unsigned q = nrt_mad24(nrt_mad24(vx, O1, vy), O2, vz);
                        __builtin_nontemporal_store(acc, outb + ((size_t)q * 8u + (unsigned)lg));
                    }
                    if (DICE) {
                        const nrt_f2 pl = {acc[0], acc[1]}, ph2 = {acc[2], acc[3]}, tl = {T[i][0], T[i][1]}, th = {T[i][2], T[i][3]};
                        stp_l = stp_l + tl * pl; stp_h = stp_h + th * ph2;
                        stt_l = stt_l + tl * tl; stt_h = stt_h + th * th;
                        spp_l = spp_l + pl * pl; spp_h = spp_h + ph2 * ph2;
                        if (MINMAX) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                mnt = fminf(mnt, T[i][k]); mxt = fmaxf(mxt, T[i][k]);
                                mnp = fminf(mnp, acc[k]); mxp = fmaxf(mxp, acc[k]);
                            }
                        }
                    }
                }
            };
            if (!(dbg & 4)) {
                if (csize == 64) {
                    // software pipeline: records two iterations ahead, rows one iteration ahead (LDS returns in order)
                    Rec M[3];
                    nrt_f4 R[2][8];
                    ld_rec(0, M[0]);
                    ld_rec(1, M[1]);
                    ld_rows(M[0], R[0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i + 2 < 8) ld_rec(i + 2, M[(i + 2) % 3]);
                        if (i + 1 < 8) ld_rows(M[(i + 1) % 3], R[(i + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        blend_math(i, M[i % 3], R[i & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int ip = 0; ip < 4; ++ip) {
                        if (2 * ip >= ilo && 2 * ip < ihi) {                // wave-uniform
                            Rec M[2];
                            nrt_f4 R[2][8];
                            ld_rec(2 * ip, M[0]);
                            ld_rec(2 * ip + 1, M[1]);
                            ld_rows(M[0], R[0]);
                            ld_rows(M[1], R[1]);
                            __builtin_amdgcn_sched_barrier(0);
                            blend_math(2 * ip, M[0], R[0]);
                            blend_math(2 * ip + 1, M[1], R[1]);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            stamp(4);
            ph[5] += 1;
            if (!have_next) break;
            cxw = nxw; clo = nlo; csize = nsize; ctotal = ntotal;
            make_record(npos);
        }
    }


    if (DICE) {
        // one partial row per wave: [3][L] sums + min/max (second stage: dice_reduce.h)
        nrt_f4 stp = {stp_l[0], stp_l[1], stp_h[0], stp_h[1]}, stt = {stt_l[0], stt_l[1], stt_h[0], stt_h[1]},
               spp = {spp_l[0], spp_l[1], spp_h[0], spp_h[1]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (int off = 8; off < NRT_WAVE; off <<= 1) {
                stp[k] += __shfl_xor(stp[k], off, NRT_WAVE);
                stt[k] += __shfl_xor(stt[k], off, NRT_WAVE);
                spp[k] += __shfl_xor(spp[k], off, NRT_WAVE);
            }
        }
        if (MINMAX) {
            for (int off = 1; off < NRT_WAVE; off <<= 1) {
                mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
                mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
            }
        }
        const long long pbase = (long long)b * per_batch + prow;
        if (dbg & 8) {
            const bool l0 = lane == 0;
            stp = (nrt_f4){l0 ? (float)ph[0] : 0.f, l0 ? (float)ph[1] : 0.f, l0 ? (float)ph[2] : 0.f, l0 ? (float)ph[3] : 0.f};
            stt = (nrt_f4){l0 ? (float)ph[4] : 0.f, l0 ? (float)ph[5] : 0.f, l0 ? 1.f : 0.f, 0.f};
        }
        if (lane < 8) {
            nrt_f4 *fp = (nrt_f4 *)(c.fpart + pbase * 3 * L);
            fp[0 * 8 + lane] = stp;
            fp[1 * 8 + lane] = stt;
            fp[2 * 8 + lane] = spp;
        }
        if (lane == 0) {
            nrt_f4 mm = {mnt, mxt, mnp, mxp};
            *(nrt_f4 *)(c.mpart + pbase * 4) = mm;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gather_lpv: the same lane-per-voxel index arithmetic and group-per-voxel blend, WITHOUT the de-duplication: the 8 corner
// rows of a voxel come straight from memory into registers (through the L1).  What it removes relative to the row kernels of
// interpn.hip / fused.hip is VALU work: there every lane of an 8-lane group repeats the location / floor / clip / weight /
// address arithmetic of its voxel (~100 of ~180 VALU instructions per 8 voxels; a wave64 VALU instruction costs ~4.2 clk
// and a VMEM instruction blocks its wave for ~49 clk, tools/lab/: those kernels are bound by per-wave issue, not by the
// TA or HBM).  Here the arithmetic runs once per voxel (one lane), the per-voxel record (8 row offsets, 8 corner weights,
// flags: 64 B) goes through a wave-private 4 KB LDS buffer, and the blend loop is ~60 VALU + 9 VMEM per 8 voxels.  No row
// buffer => ~16 waves per CU.
// ---------------------------------------------------------------------------------------------------------------------
// SHARE: the WPB = 4 waves of a work-group own the SAME patch and each blends two of a window's eight iterations, so the
// rows a window's voxels share are requested by one CU within a short time (L1 hits / merged misses instead of L2 requests).
template <int MODE, int LWX, int LPY, int LPZ, bool STORE, bool DICE, bool MINMAX, int WPB, bool SHARE>
__global__ __launch_bounds__(64 * WPB) void gather_lpv(WdArgs A) {
    static_assert(!SHARE || WPB == 4, "shared windows: four waves x two iterations");
    static_assert(LWX + LPY + LPZ == 6, "a window is 64 voxels");
    constexpr int WX = 1 << LWX, PY = 1 << LPY, PZ = 1 << LPZ;
    constexpr int L = 32;
    __shared__ __attribute__((aligned(16))) nrt_i4 lp_meta_all[WPB][64 * 4];
    const unsigned wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nrt_i4 *meta = lp_meta_all[wave_in_block];

    const WddCall &c = A.c;
    const WdGeom &gm = A.g;
    const int lane = threadIdx.x & 63, lg = lane & 7, g = lane >> 3;
    const unsigned kx = blockIdx.x % NRT_NXCD, jb = SHARE ? blockIdx.x / NRT_NXCD : (blockIdx.x / NRT_NXCD) * WPB + wave_in_block;
    const unsigned per_batch = gm.ncol * gm.nseg, U = per_batch * gm.nbatch, perU = (gridDim.x / NRT_NXCD) * (SHARE ? 1 : WPB);
    const unsigned u = kx * perU + jb;
    if (u >= U) return;
    const int b = (int)(u / per_batch);
    const unsigned prow = u % per_batch;
    const unsigned useg = prow / gm.ncol, ucol = prow % gm.ncol;
    const unsigned RY = 1u << gm.lry, RZ = 1u << gm.lrz, nRz = (gm.npz + RZ - 1) / RZ;
    const unsigned reg = ucol / (RY * RZ), wi = ucol % (RY * RZ);
    const unsigned cy = (reg / nRz) * RY + wi / RZ, cz = (reg % nRz) * RZ + wi % RZ;
    const int x0 = (int)(useg * gm.seglen), y0 = (int)cy << LPY, z0 = (int)cz << LPZ;
    int xlen = min((int)gm.seglen, c.O[0] - x0);
    if (cy >= gm.npy || cz >= gm.npz) xlen = 0;
    const int xend = x0 + max(xlen, 0);

    const char *volb = (const char *)(c.vol + (long long)b * c.vol_bs);
    const float *locb = c.loc ? c.loc + (long long)b * c.loc_bs : nullptr;
    nrt_f4 *outb = STORE ? (nrt_f4 *)(c.out + (long long)b * c.out_bs) : nullptr;
    const nrt_f4 *fixb = DICE ? (const nrt_f4 *)(c.fixed + (long long)b * c.out_bs) : nullptr;
    const unsigned SY = (unsigned)c.S[1], SZ = (unsigned)c.S[2], SYZ = SY * SZ;
    const unsigned O1 = (unsigned)c.O[1], O2 = (unsigned)c.O[2];
    const int wz = lane & (PZ - 1), wy = (lane >> LPZ) & (PY - 1), wxl = lane >> (LPZ + LPY);
    const int yv = y0 + wy, zv = z0 + wz;
    const bool yz_ok = yv < c.O[1] && zv < c.O[2];
    const int yc = min(yv, c.O[1] - 1), zc = min(zv, c.O[2] - 1);
    const unsigned lg16 = (unsigned)lg * 16u;

    nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;
    float pn[3] = {0.0f, 0.0f, 0.0f};
    auto fetch_loc = [&](int xw) {
        if (MODE != NRT_LOC_LINSPACE) {
            const unsigned xq = (unsigned)min(xw + wxl, c.O[0] - 1);
            const unsigned q = nrt_mad24(nrt_mad24(xq, O1, (unsigned)yc), O2, (unsigned)zc);
            const float *lp = locb + (size_t)nrt_times3(q);
            pn[0] = lp[0]; pn[1] = lp[1]; pn[2] = lp[2];
        }
    };
    struct Rec { nrt_i4 o0, o1; nrt_f4 wa, wb; };
    // ---- lane-per-voxel: location, corner indices and weights (utils.py:139-153), once per voxel: the record of voxel `lane`
    auto index_window = [&](int xw, Rec &rec) {
        const int x = xw + wxl;
        const bool valid = yz_ok && x < xend;
        const int qd[3] = {min(x, c.O[0] - 1), yc, zc};
        float p[NRT_MAXD];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (MODE == NRT_LOC_ABSOLUTE) p[d] = pn[d];
            else if (MODE == NRT_LOC_SHIFT) p[d] = nrt_add((float)qd[d], pn[d]);
            else p[d] = (qd[d] == 0) ? 0.0f : ((qd[d] == c.O[d] - 1) ? (float)(c.S[d] - 1) : nrt_mul(c.delta[d], (float)qd[d]));
        }
        if (xw + WX < xend) fetch_loc(xw + WX);
        int i0x, i1x, i0y, i1y, i0z, i1z;
        float w0x, w0y, w0z, w1x, w1y, w1z;
        corner_1d(p[0], c.S[0], i0x, i1x, w0x, w1x);
        corner_1d(p[1], c.S[1], i0y, i1y, w0y, w1y);
        corner_1d(p[2], c.S[2], i0z, i1z, w0z, w1z);
        bool oob = false;
        if (c.has_fill) {
#pragma unroll
            for (int d = 0; d < 3; ++d) oob = oob || (p[d] < 0.0f) || (p[d] > (float)(c.S[d] - 1));
        }
        const unsigned r0 = nrt_mad24(nrt_mad24((unsigned)i0x, SY, (unsigned)i0y), SZ, (unsigned)i0z) << 7;   // byte offset of the row
        const unsigned rdx = i1x != i0x ? SYZ << 7 : 0u, rdy = i1y != i0y ? SZ << 7 : 0u, rdz = i1z != i0z ? 128u : 0u;
        const nrt_f2 wy2 = {w0y, w1y}, wz2 = {w0z, w1z};
        const nrt_f2 wxy0 = (nrt_f2){w0x, w0x} * wy2, wxy1 = (nrt_f2){w1x, w1x} * wy2;     // (wx * wy) * wz: utils.py:1085-1092
        const nrt_f2 wt01 = (nrt_f2){wxy0[0], wxy0[0]} * wz2, wt23 = (nrt_f2){wxy0[1], wxy0[1]} * wz2;
        const nrt_f2 wt45 = (nrt_f2){wxy1[0], wxy1[0]} * wz2, wt67 = (nrt_f2){wxy1[1], wxy1[1]} * wz2;
        rec.o0[0] = (int)r0; rec.o0[1] = (int)(r0 + rdz); rec.o0[2] = (int)(r0 + rdy); rec.o0[3] = (int)(r0 + rdy + rdz);
        rec.o1[0] = (int)(r0 + rdx); rec.o1[1] = (int)(r0 + rdx + rdz); rec.o1[2] = (int)(r0 + rdx + rdy);
        // the flags ride in the spare low bits of the last offset (rows are 128-byte aligned)
        rec.o1[3] = (int)((r0 + rdx + rdy + rdz) | (valid ? 1u : 0u) | (oob ? 2u : 0u));
        rec.wa = (nrt_f4){wt01[0], wt01[1], wt23[0], wt23[1]};
        rec.wb = (nrt_f4){wt45[0], wt45[1], wt67[0], wt67[1]};
    };
    auto put_rec = [&](const Rec &rec) {
        __builtin_amdgcn_wave_barrier();                                    // the previous window's records have been read
        asm volatile("" ::: "memory");
        meta[4 * lane] = rec.o0;
        meta[4 * lane + 1] = rec.o1;
        ((nrt_f4 *)meta)[4 * lane + 2] = rec.wa;
        ((nrt_f4 *)meta)[4 * lane + 3] = rec.wb;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };
    auto ld_rec = [&](int i, Rec &m) {
        const int v = 8 * i + g;
        m.o0 = meta[4 * v]; m.o1 = meta[4 * v + 1];
        m.wa = ((const nrt_f4 *)meta)[4 * v + 2]; m.wb = ((const nrt_f4 *)meta)[4 * v + 3];
    };
    int xw = x0;                                                             // window being blended
    auto voxel_q = [&](int i, bool clampit) {
        const int v = 8 * i + g;
        int vx = xw + (v >> (LPZ + LPY)), vy = y0 + ((v >> LPZ) & (PY - 1)), vz = z0 + (v & (PZ - 1));
        if (clampit) { vx = min(vx, c.O[0] - 1); vy = min(vy, c.O[1] - 1); vz = min(vz, c.O[2] - 1); }
        return nrt_mad24(nrt_mad24((unsigned)vx, O1, (unsigned)vy), O2, (unsigned)vz);
    };
    auto ld_rows = [&](int i, const Rec &m, nrt_f4 (&R)[8], nrt_f4 &T) {
        const unsigned off[8] = {(unsigned)m.o0[0], (unsigned)m.o0[1], (unsigned)m.o0[2], (unsigned)m.o0[3],
                                 (unsigned)m.o1[0], (unsigned)m.o1[1], (unsigned)m.o1[2], (unsigned)m.o1[3] & ~127u};
#pragma unroll
        for (int k = 0; k < 8; ++k) R[k] = *(const nrt_f4 *)(volb + (size_t)(off[k] + lg16));
        if (DICE) T = __builtin_nontemporal_load(fixb + ((size_t)voxel_q(i, true) * 8u + (unsigned)lg));
    };
    auto blend_math = [&](int i, const Rec &m, const nrt_f4 (&R)[8], const nrt_f4 &T) {
        const bool vvalid = (unsigned)m.o1[3] & 1u, voob = (unsigned)m.o1[3] & 2u;
        const float wt[8] = {m.wa[0], m.wa[1], m.wa[2], m.wa[3], m.wb[0], m.wb[1], m.wb[2], m.wb[3]};
        nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const nrt_f2 w2 = {wt[corner], wt[corner]};
            al = al + w2 * (nrt_f2){R[corner][0], R[corner][1]};
            ah = ah + w2 * (nrt_f2){R[corner][2], R[corner][3]};
        }
        nrt_f4 acc = {al[0], al[1], ah[0], ah[1]};
        if (c.has_fill) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = apply_fill(acc[k], voob, c.fill);
        }
        if (vvalid) {
            if (STORE) __builtin_nontemporal_store(acc, outb + ((size_t)voxel_q(i, false) * 8u + (unsigned)lg));
            if (DICE) {
                const nrt_f2 pl = {acc[0], acc[1]}, ph2 = {acc[2], acc[3]}, tl = {T[0], T[1]}, th = {T[2], T[3]};
                stp_l = stp_l + tl * pl; stp_h = stp_h + th * ph2;
                stt_l = stt_l + tl * tl; stt_h = stt_h + th * th;
                spp_l = spp_l + pl * pl; spp_h = spp_h + ph2 * ph2;
                if (MINMAX) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        mnt = fminf(mnt, T[k]); mxt = fmaxf(mxt, T[k]);
                        mnp = fminf(mnp, acc[k]); mxp = fmaxf(mxp, acc[k]);
                    }
                }
            }
        }
    };
    if (xlen > 0) {
        Rec mine;
        fetch_loc(x0);
        index_window(x0, mine);
        put_rec(mine);
        for (; xw < xend; xw += WX) {
            const bool more = xw + WX < xend;
            if (SHARE) {
                // this wave: iterations 2 w and 2 w + 1; their rows are in flight while the next window is indexed
                const int it0 = 2 * (int)wave_in_block, it1 = it0 + 1;
                Rec M[2];
                nrt_f4 R[2][8], T[2];
                ld_rec(it0, M[0]);
                ld_rec(it1, M[1]);
                ld_rows(it0, M[0], R[0], T[0]);
                ld_rows(it1, M[1], R[1], T[1]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) index_window(xw + WX, mine);
                __builtin_amdgcn_sched_barrier(0);
                blend_math(it0, M[0], R[0], T[0]);
                blend_math(it1, M[1], R[1], T[1]);
                if (more) put_rec(mine);
            } else {
                // 8 iterations of 8 voxels; the rows of iteration i + 1 are in flight while i is blended
                Rec M[2];
                nrt_f4 R[2][8], T[2];
                ld_rec(0, M[0]);
                ld_rows(0, M[0], R[0], T[0]);
                ld_rec(1, M[1]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i + 1 < 8) ld_rows(i + 1, M[(i + 1) & 1], R[(i + 1) & 1], T[(i + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    blend_math(i, M[i & 1], R[i & 1], T[i & 1]);
                    if (i + 2 < 8) ld_rec(i + 2, M[i & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more) { index_window(xw + WX, mine); put_rec(mine); }
            }
        }
    }

    if (DICE) {
        nrt_f4 stp = {stp_l[0], stp_l[1], stp_h[0], stp_h[1]}, stt = {stt_l[0], stt_l[1], stt_h[0], stt_h[1]},
               spp = {spp_l[0], spp_l[1], spp_h[0], spp_h[1]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (int off = 8; off < NRT_WAVE; off <<= 1) {
                stp[k] += __shfl_xor(stp[k], off, NRT_WAVE);
                stt[k] += __shfl_xor(stt[k], off, NRT_WAVE);
                spp[k] += __shfl_xor(spp[k], off, NRT_WAVE);
            }
        }
        if (MINMAX) {
            for (int off = 1; off < NRT_WAVE; off <<= 1) {
                mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
                mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
            }
        }
        const long long pbase = SHARE ? ((long long)b * per_batch + prow) * WPB + wave_in_block : (long long)b * per_batch + prow;
        if (lane < 8) {
            nrt_f4 *fp = (nrt_f4 *)(c.fpart + pbase * 3 * L);
            fp[0 * 8 + lane] = stp;
            fp[1 * 8 + lane] = stt;
            fp[2 * 8 + lane] = spp;
        }
        if (lane == 0) {
            nrt_f4 mm = {mnt, mxt, mnp, mxp};
            *(nrt_f4 *)(c.mpart + pbase * 4) = mm;
        }
    }
}

// tune: bits 0-1 window shape (0: 4x4x4, 1: 4x2x8, 2: 2x4x8 as x,y,z), bit 2 row buffer (0: 192 rows = 5 waves per CU,
//       1: 256 rows = 4 waves per CU), bits 8-15 x segments (0 = auto), bits 16-18 lry, 19-21 lrz (0 = default region)
void wd_shape(int tune, int &lwx, int &lpy, int &lpz) {
    switch (tune & 3) {
        case 1: lwx = 2; lpy = 1; lpz = 3; break;
        case 2: lwx = 1; lpy = 2; lpz = 3; break;
        case 3: lwx = 3; lpy = 0; lpz = 3; break;
        default: lwx = 2; lpy = 2; lpz = 2; break;
    }
}

void wd_geometry(const int *O, int batch, int tune, WdGeom &g) {
    if (tune < 0) tune = 0;
    int lwx, lpy, lpz;
    wd_shape(tune, lwx, lpy, lpz);
    g.npy = ((unsigned)O[1] + (1u << lpy) - 1) >> lpy;
    g.npz = ((unsigned)O[2] + (1u << lpz) - 1) >> lpz;
    g.lry = (tune >> 16) & 7; g.lrz = (tune >> 19) & 7;
    if (g.lry == 0 && g.lrz == 0) { g.lry = 5 - lpy; g.lrz = 6 - lpz; }          // a region covers 32 x 64 voxels in (y,z)
    const unsigned RY = 1u << g.lry, RZ = 1u << g.lrz;
    g.ncol = ((g.npy + RY - 1) / RY) * ((g.npz + RZ - 1) / RZ) * RY * RZ;
    g.nbatch = (unsigned)batch;
    unsigned nseg = (unsigned)(tune >> 8) & 0xffu;
    if (nseg == 0) {                                                             // auto: >= 4 generations of the resident waves
        const unsigned wpc = ((tune >> 6) & 1) ? 16u : 5u;                       // gather_lpv / gather_wdd waves per CU
        const unsigned want = 4u * wpc * 256u, have = g.ncol * (unsigned)batch;
        nseg = (want + have - 1) / have;
        const unsigned cap = (unsigned)O[0] / 16u > 0 ? (unsigned)O[0] / 16u : 1u;
        if (nseg > cap) nseg = cap;
        if (nseg < 1) nseg = 1;
    }
    if (nseg > (unsigned)O[0]) nseg = (unsigned)O[0];
    const unsigned wx = 1u << lwx;
    g.seglen = ((((unsigned)O[0] + nseg - 1) / nseg) + wx - 1) / wx * wx;       // whole windows
    g.nseg = ((unsigned)O[0] + g.seglen - 1) / g.seglen;
}

template <int MODE, int LWX, int LPY, int LPZ, int NR, int WPB>
void wd_launch_nr(const WdArgs &A, unsigned units, hipStream_t st) {
    const bool store = A.c.out != nullptr, dice = A.c.fixed != nullptr, mm = A.c.minmax != 0;
    const size_t lds = (size_t)WdLds<NR>::TOTAL * WPB;
    const dim3 grid(nrt_xcd_grid((units + WPB - 1) / WPB)), blk(64 * WPB);
#define WD_GO(S_, D_, M_)                                                                                                   \
    do {                                                                                                                    \
        static bool attr = false;                                                                                           \
        if (!attr && lds > 48 * 1024) {                                                                                     \
            (void)hipFuncSetAttribute((const void *)gather_wdd<MODE, LWX, LPY, LPZ, NR, S_, D_, M_, WPB>,                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
            attr = true;                                                                                                    \
        }                                                                                                                   \
        hipLaunchKernelGGL((gather_wdd<MODE, LWX, LPY, LPZ, NR, S_, D_, M_, WPB>), grid, blk, lds, st, A);                  \
    } while (0)
    if (store && dice) WD_GO(true, true, true);
    else if (dice && mm) WD_GO(false, true, true);
    else if (dice) WD_GO(false, true, false);
    else WD_GO(true, false, false);
#undef WD_GO
}

template <int MODE, int LWX, int LPY, int LPZ, bool SHARE>
void lp_launch(const WdArgs &A, unsigned units, hipStream_t st) {
    constexpr int WPB = 4;
    const bool store = A.c.out != nullptr, dice = A.c.fixed != nullptr, mm = A.c.minmax != 0;
    const dim3 grid(nrt_xcd_grid(SHARE ? units : (units + WPB - 1) / WPB)), blk(64 * WPB);
    if (store && dice) hipLaunchKernelGGL((gather_lpv<MODE, LWX, LPY, LPZ, true, true, true, WPB, SHARE>), grid, blk, 0, st, A);
    else if (dice && mm) hipLaunchKernelGGL((gather_lpv<MODE, LWX, LPY, LPZ, false, true, true, WPB, SHARE>), grid, blk, 0, st, A);
    else if (dice) hipLaunchKernelGGL((gather_lpv<MODE, LWX, LPY, LPZ, false, true, false, WPB, SHARE>), grid, blk, 0, st, A);
    else hipLaunchKernelGGL((gather_lpv<MODE, LWX, LPY, LPZ, true, false, false, WPB, SHARE>), grid, blk, 0, st, A);
}

template <int MODE>
void wd_launch_mode(const WdArgs &A, unsigned units, int tune, hipStream_t st) {
    const bool big = (tune >> 2) & 1;
    if ((tune >> 6) & 1) {                                                       // lane-per-voxel gather without the row buffer
        if ((tune >> 7) & 1) {                                                   // four waves share a window
            switch (tune & 3) {
                case 1: lp_launch<MODE, 2, 1, 3, true>(A, units, st); break;
                default: lp_launch<MODE, 2, 2, 2, true>(A, units, st); break;
            }
            return;
        }
        switch (tune & 3) {
            case 1: lp_launch<MODE, 2, 1, 3, false>(A, units, st); break;
            case 2: lp_launch<MODE, 1, 2, 3, false>(A, units, st); break;
            case 3: lp_launch<MODE, 3, 0, 3, false>(A, units, st); break;
            default: lp_launch<MODE, 2, 2, 2, false>(A, units, st); break;
        }
        return;
    }
    if ((tune >> 3) & 1) { wd_launch_nr<MODE, 2, 2, 2, 120, 1>(A, units, st); return; }
    if ((tune >> 4) & 1) { wd_launch_nr<MODE, 2, 2, 2, 192, 4>(A, units, st); return; }
    if ((tune >> 5) & 1) { wd_launch_nr<MODE, 2, 2, 2, 192, 2>(A, units, st); return; }
    switch (tune & 3) {
        case 1:
            if (big) wd_launch_nr<MODE, 2, 1, 3, 256, 1>(A, units, st); else wd_launch_nr<MODE, 2, 1, 3, 192, 1>(A, units, st);
            break;
        case 2:
            if (big) wd_launch_nr<MODE, 1, 2, 3, 256, 1>(A, units, st); else wd_launch_nr<MODE, 1, 2, 3, 192, 1>(A, units, st);
            break;
        default:
            if (big) wd_launch_nr<MODE, 2, 2, 2, 256, 1>(A, units, st); else wd_launch_nr<MODE, 2, 2, 2, 192, 1>(A, units, st);
            break;
    }
}

}  // namespace

bool nrt_wdd_supported(const int *S, const int *O, int channels) {
    if (channels != 32 || !S || !O) return false;
    const unsigned long long nrows = (unsigned long long)S[0] * S[1] * S[2];
    if (nrows >= (1ull << 24)) return false;                                      // row index: 24-bit multiplies, u32 byte offsets
    if ((long long)S[0] * S[1] >= (1 << 24) || S[2] >= (1 << 24)) return false;
    if (S[0] >= 65535 || S[1] >= 65535 || S[2] >= 65535) return false;          // bounding box: packed 16-bit coordinates
    if ((long long)O[0] * O[1] >= (1 << 24) || O[2] >= (1 << 24)) return false;
    if ((unsigned long long)O[0] * O[1] * O[2] * 128ull >= (1ull << 36)) return false;
    return O[0] > 0 && O[1] > 0 && O[2] > 0;
}

unsigned nrt_wdd_rows(const int *O, int batch, int tune) {
    WdGeom g;
    wd_geometry(O, batch, tune, g);
    const bool share = ((tune >> 6) & 1) && ((tune >> 7) & 1);                   // gather_lpv, four waves per window: a row per wave
    return g.ncol * g.nseg * (share ? 4u : 1u);
}

int nrt_wdd_launch(const WddCall &c, hipStream_t st) {
    WdArgs A;
    A.c = c;
    const int tune = c.tune < 0 ? 0 : c.tune;
    wd_geometry(c.O, c.batch, tune, A.g);
    const unsigned units = A.g.ncol * A.g.nseg * (unsigned)c.batch;
    switch (c.mode) {
        case NRT_LOC_ABSOLUTE: wd_launch_mode<NRT_LOC_ABSOLUTE>(A, units, tune, st); break;
        case NRT_LOC_SHIFT: wd_launch_mode<NRT_LOC_SHIFT>(A, units, tune, st); break;
        default: wd_launch_mode<NRT_LOC_LINSPACE>(A, units, tune, st); break;
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
