// Shared device/host pieces of the interpn kernels (interpn.hip, fused.hip): argument block, sampling
// location, the per-dimension corner arithmetic of neurite/tf/utils/utils.py:139-153, fill, tile geometry.
#pragma once

#include "nrt_common.h"

namespace {

struct InterpArgs {
    const void *vol;
    const float *loc;
    void *out;
    int S[NRT_MAXD];       // source spatial shape (unused dims = 1)
    int O[NRT_MAXD];       // output spatial shape
    int C;
    long long vol_bs, loc_bs, out_bs;   // batch strides in elements
    float delta[NRT_MAXD]; // linspace step per dim: fl((S-1)/(O-1))
    unsigned nout;         // prod(O)
    int has_fill;
    float fill_f;
    int fill_i;
    const float *addend;   // generic linear kernel only: out = addend + interp (vxm compose / integrate), or null
    long long addend_bs;
};

// ---- sampling location of output voxel q (coordinates qd) -------------------------------------
template <int D, int MODE>
__device__ __forceinline__ void load_loc(const InterpArgs &a, const float *locb, unsigned q,
                                         const int (&qd)[NRT_MAXD], float (&p)[NRT_MAXD]) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (MODE == NRT_LOC_ABSOLUTE) {
            p[d] = locb[(long long)q * D + d];
        } else if (MODE == NRT_LOC_SHIFT) {
            // vxm transform(): cast(mesh, float32) + shift     (one rounding)
            p[d] = nrt_add((float)qd[d], locb[(long long)q * D + d]);
        } else {
            // tf.linspace(0., S-1., O): first = 0, last = S-1 exactly, middle = 0 + delta*i
            p[d] = (qd[d] == 0) ? 0.0f
                 : ((qd[d] == a.O[d] - 1) ? (float)(a.S[d] - 1) : nrt_mul(a.delta[d], (float)qd[d]));
        }
    }
}

template <int D>
__device__ __forceinline__ void decode(const InterpArgs &a, unsigned q, int (&qd)[NRT_MAXD]) {
    unsigned r = q;
#pragma unroll
    for (int d = D - 1; d > 0; --d) { qd[d] = (int)(r % (unsigned)a.O[d]); r /= (unsigned)a.O[d]; }
    qd[0] = (int)r;
}

// utils.py:139-153 for one dimension
__device__ __forceinline__ void corner_1d(float p, int size, int &i0, int &i1, float &w0, float &w1) {
    const float mx = (float)(size - 1);
    const float f = floorf(p);                       // :139
    const float cl = nrt_clip(p, 0.0f, mx);          // :142
    const float l0 = nrt_clip(f, 0.0f, mx);          // :143
    const float l1 = nrt_clip(nrt_add(l0, 1.0f), 0.0f, mx);   // :146
    i0 = (int)l0; i1 = (int)l1;                      // :147
    w0 = nrt_sub(l1, cl);                            // :152  weight of the lower corner
    w1 = nrt_sub(1.0f, w0);                          // :153  weight of the upper corner
}

__device__ __forceinline__ int nearest_1d(float p, int size) {
    // :196-197  int32(round_half_even(p)) clipped to [0, size-1]; v_cvt_i32_f32 saturates, NaN -> 0
    return nrt_clampi((int)rintf(p), 0, size - 1);
}

template <int D>
__device__ __forceinline__ bool out_of_bounds(const InterpArgs &a, const float (&p)[NRT_MAXD]) {
    bool oob = false;                                // :209-211 (unclipped location)
#pragma unroll
    for (int d = 0; d < D; ++d) oob = oob || (p[d] < 0.0f) || (p[d] > (float)(a.S[d] - 1));
    return oob;
}

__device__ __forceinline__ float apply_fill(float v, bool oob, float fill) {
    // :212-213   v * float(!oob) + float(oob) * fill   (NaN/Inf propagate exactly as in the reference)
    return nrt_add(nrt_mul(v, oob ? 0.0f : 1.0f), nrt_mul(oob ? 1.0f : 0.0f, fill));
}

struct TileGeom {
    int ltx, lty, ltz;          // log2 of the tile extent
    unsigned nTy, nTz;          // tiles along y and z
    unsigned nT2;               // tiles in the (x,y) plane = nTx * nTy
    unsigned per2;              // (x,y) tiles owned by one XCD = ceil(nT2 / 8)
    int z_outer;                // order inside an XCD's slab: 0 = z fastest, 1 = z outermost
    int plane_major;            // C == 32 only: a pass is one z-plane of a 4(x) x 8(y) patch, tz = z-chunk length
    int tz;                     // tile extent along z (= 1 << ltz unless plane_major)
    // x-march (fused kernel): a block owns one (y,z) patch of 2^lty x 2^ltz voxels and walks x over one of nseg
    // segments; every block is resident at once and the blocks of an XCD advance through the source x-planes
    // together, so plane i+1 of a step is the L2-resident plane i of the next one
    int x_march;
    unsigned ncol, nseg, seglen;   // (y,z) patches per volume, x segments, x planes per segment
    unsigned nbatch;
    int lry, lrz;                  // log2 of the region extent in patches: consecutive blocks fill a (2^lry x 2^lrz) patch region
    int depth_sync;                // backward x-march kernels: passes between block barriers (power of two), 0 = none
    // mixed block lengths (fused forward kernels, xmarch_setup_mixed): of the het_cpx columns an XCD owns, the first het_full are
    // marched whole by one block each and the others in nseg pieces of seglen planes, the pieces LAST in launch order: the last round
    // of blocks is short, so the chip drains in a fraction of a march.  het_cpx = 0: every column in nseg equal pieces (xmarch_setup).
    unsigned het_cpx, het_full;
    unsigned items_x;              // work items (blocks of the non-persistent launch) per XCD
    // partial rows per batch entry.  Equal pieces: ncol x nseg.  Mixed lengths: ONE row per column (whole columns, first pieces) followed
    // by nseg - 1 rows per SPLIT column of the entry, padded to the entry with the most split columns -- with a row per (column, piece)
    // four fifths of the rows of a 4-volume launch were zeros written by the gather and read back by the second stage (15 us of 915)
    unsigned prows;
};

// tune word of the tiled kernels: ltx | lty << 4 | ltz << 8 | z_outer << 12 | plane_major << 13 | x_march << 14 | LZ << 16
// (x_march, fused kernel only: bits 16.. = number of x segments, 0 = auto)
inline void tile_geometry(const int *out_shape, int G, int tune, int default_tune, TileGeom &tg, unsigned &ntiles) {
    const int NG = 256 / G, WZ = 64 / G;
    if (tune <= 0) tune = default_tune;
    tg.ltx = tune & 15; tg.lty = (tune >> 4) & 15; tg.ltz = (tune >> 8) & 15; tg.z_outer = (tune >> 12) & 1;
    tg.plane_major = ((tune >> 13) & 1) && G == 8;
    tg.x_march = 0; tg.ncol = 0; tg.nseg = 1; tg.seglen = 0; tg.nbatch = 1; tg.lry = 0; tg.lrz = 0; tg.depth_sync = 0;
    tg.het_cpx = 0; tg.het_full = 0; tg.items_x = 0; tg.prows = 0;
    if (tg.plane_major) {
        tg.ltx = 2; tg.lty = 3; tg.ltz = 0;
        tg.tz = (tune >> 16) & 0xfff;
        if (tg.tz <= 0 || tg.tz > out_shape[2]) tg.tz = out_shape[2];
    } else {
        while ((1 << tg.ltz) < WZ) ++tg.ltz;                       // a wave must stay inside one z-run
        while ((1 << (tg.ltx + tg.lty + tg.ltz)) < NG) ++tg.lty;  // at least one pass
        tg.tz = 1 << tg.ltz;
    }
    const unsigned nTx = (out_shape[0] + (1 << tg.ltx) - 1) >> tg.ltx;
    tg.nTy = (out_shape[1] + (1 << tg.lty) - 1) >> tg.lty;
    tg.nTz = (out_shape[2] + tg.tz - 1) / tg.tz;
    tg.nT2 = nTx * tg.nTy;
    tg.per2 = (tg.nT2 + NRT_NXCD - 1) / NRT_NXCD;
    ntiles = NRT_NXCD * tg.per2 * tg.nTz;
}

// output voxel handled by lane-group g in pass `pass` of the tile at (x0, y0, z0)
__device__ __forceinline__ void tile_voxel(const TileGeom &tg, int NG, int pass, int g, int x0, int y0, int z0,
                                           int &x, int &y, int &z) {
    if (tg.plane_major) {
        const int w = g >> 3, j = g & 7;            // wave 2x2 in (x,y), lane-groups 2x4 inside the wave
        x = x0 + (w >> 1) * 2 + (j >> 2);
        y = y0 + (w & 1) * 4 + (j & 3);
        z = z0 + pass;
    } else {
        const int s = pass * NG + g;
        x = x0 + (s >> (tg.ltz + tg.lty));
        y = y0 + ((s >> tg.ltz) & ((1 << tg.lty) - 1));
        z = z0 + (s & ((1 << tg.ltz) - 1));
    }
}


// ---- x-march schedule shared by the fused forward and the backward gathers ---------------------------------------
// host: fill the x-march fields of tg for `tune` (lty/ltz = patch, bits 16-23 segments, 24-26 lry, 27-29 lrz); returns the
// number of blocks per batch entry
inline unsigned xmarch_setup(const int *out_shape, int batch, int t, TileGeom &tg) {
    tg.x_march = 1;
    tg.nbatch = (unsigned)batch;
    tg.lry = (t >> 24) & 7; tg.lrz = (t >> 27) & 7;
    const unsigned RY = 1u << tg.lry, RZ = 1u << tg.lrz;             // regions are padded; out-of-range patches are empty blocks
    tg.ncol = ((tg.nTy + RY - 1) / RY) * ((tg.nTz + RZ - 1) / RZ) * RY * RZ;
    unsigned nseg = (unsigned)(t >> 16) & 0xffu;
    if (nseg == 0) {
        // auto: at least twelve blocks per CU over the launch, and a last round that is nearly full.  Two blocks are resident per CU,
        // so a launch runs in rounds of 2 x CUs blocks of (about) equal length: the 3200 full-length marches of 4 x 160^3 are 6.25
        // rounds -- the seventh is a quarter full.  Measured (tools/nseg_probe.py, profiles/r04_lab/nseg_probe.jsonl), register
        // kernel / wave-cache kernel: batch 4 1.133 / 1.110 ms with 1 segment, 1.109 / 1.074 with 2, 1.144 / 1.122 with 4 (short
        // marches re-fetch their first planes); batch 1 0.347 / 0.339 with 1, 0.312 / 0.313 with 4 or 5.
        const unsigned cols = tg.ncol * (unsigned)batch, slots = 2u * (unsigned)nrt_num_cus();
        nseg = (3072u + cols - 1) / cols;
        if (nseg < 1) nseg = 1;
        for (; nseg < 8; ++nseg) {
            const unsigned blocks = cols * nseg, rounds = (blocks + slots - 1) / slots;
            if ((unsigned long long)(rounds * slots - blocks) * 20ull <= blocks) break;             // idle slots of the last round <= 5 %
        }
    }
    if (nseg > (unsigned)out_shape[0]) nseg = (unsigned)out_shape[0];
    tg.seglen = ((unsigned)out_shape[0] + nseg - 1) / nseg;
    tg.nseg = ((unsigned)out_shape[0] + tg.seglen - 1) / tg.seglen;
    tg.items_x = (tg.ncol * tg.nseg * (unsigned)batch + NRT_NXCD - 1) / NRT_NXCD;
    tg.prows = tg.ncol * tg.nseg;
    return tg.prows;
}

// columns c' < c (c = XCD * het_cpx + position in the XCD's list) that are marched in pieces
__host__ __device__ inline unsigned xmarch_split_before(const TileGeom &tg, unsigned c) {
    const unsigned cl = c % tg.het_cpx;
    return (c / tg.het_cpx) * (tg.het_cpx - tg.het_full) + (cl > tg.het_full ? cl - tg.het_full : 0u);
}

// host: mixed block lengths for the fused forward kernels when the tune word leaves the segments to us.  Blocks of (about) equal
// length run in rounds of `slots` = 2 x CUs; with C columns per launch the last round is C mod slots full and costs a whole march:
// 4 x 160^3 = 3200 columns = 6.25 rounds, and 32 x 160^3 (100 rounds) runs at 0.248 ms per volume where 4 volumes take 0.277.
// Equal pieces (xmarch_setup) shorten the last round but every piece re-fetches its first planes: measured +7 % / +15 % / +22 % for
// 2 / 4 / 8 pieces per column (tools/nseg_probe.py).  So: whole columns for the full rounds, pieces only for the remainder, launched
// last.  The model below picks the number of whole columns per XCD (a multiple of its slots) and the pieces per remaining column;
// returns the partial rows per batch entry (ncol x pieces) and the grid size in `grid`.
inline unsigned xmarch_setup_mixed(const int *out_shape, int batch, int t, TileGeom &tg, unsigned &grid) {
    tg.x_march = 1;
    tg.nbatch = (unsigned)batch;
    tg.lry = (t >> 24) & 7; tg.lrz = (t >> 27) & 7;
    const unsigned RY = 1u << tg.lry, RZ = 1u << tg.lrz;
    tg.ncol = ((tg.nTy + RY - 1) / RY) * ((tg.nTz + RZ - 1) / RZ) * RY * RZ;
    const unsigned cols = tg.ncol * (unsigned)batch, cpx = (cols + NRT_NXCD - 1) / NRT_NXCD;
    const unsigned slots = (unsigned)max(1, 2 * nrt_num_cus() / NRT_NXCD);      // (a partition reporting fewer than 4 CUs: never 0)
    static const int PS[7] = {1, 2, 3, 4, 5, 6, 8};
    static const float OVH[7] = {0.f, 0.07f, 0.11f, 0.15f, 0.17f, 0.19f, 0.22f};
    float best = 1e30f;
    unsigned bF = 0, bP = 1;
    for (unsigned F = 0; F <= cpx; F += slots) {
        const unsigned rem = cpx - F;
        for (int i = 0; i < 7; ++i) {
            const unsigned P = (unsigned)PS[i];
            if (P > (unsigned)out_shape[0]) break;
            const float cost = (float)(F / slots) + (rem ? (float)((rem * P + slots - 1) / slots) / (float)P * (1.0f + OVH[i]) : 0.0f);
            if (cost < best - 1e-6f) { best = cost; bF = F; bP = P; }
            if (!rem) break;
        }
    }
    tg.het_cpx = cpx; tg.het_full = bF;
    tg.nseg = bP;
    tg.seglen = ((unsigned)out_shape[0] + bP - 1) / bP;
    tg.items_x = bF + (cpx - bF) * bP;
    grid = NRT_NXCD * tg.items_x;
    unsigned maxsplit = 0;
    for (unsigned b = 0; b < (unsigned)batch; ++b) {
        const unsigned c1 = (b + 1) * tg.ncol < cols ? (b + 1) * tg.ncol : cols;
        const unsigned sp = xmarch_split_before(tg, c1) - xmarch_split_before(tg, b * tg.ncol);
        if (sp > maxsplit) maxsplit = sp;
    }
    tg.prows = tg.ncol + maxsplit * (bP - 1);
    return tg.prows;
}

// device: the work of this block under either schedule.  false: nothing (the block exits; it owns no partial row)
struct XmWork { int b; unsigned ucol, prow; int x0, xlen; bool whole; };
__device__ __forceinline__ bool xmarch_work_at(const TileGeom &tg, int O0, unsigned k, unsigned jb, XmWork &w);
__device__ __forceinline__ bool xmarch_work(const TileGeom &tg, int O0, XmWork &w) {
    return xmarch_work_at(tg, O0, blockIdx.x % NRT_NXCD, blockIdx.x / NRT_NXCD, w);
}
// item jb of XCD k's list (a persistent block may take items of another XCD's list when its own is empty)
__device__ __forceinline__ bool xmarch_work_at(const TileGeom &tg, int O0, unsigned k, unsigned jb, XmWork &w) {
    if (tg.het_cpx) {
        unsigned cl, piece = 0;
        w.whole = jb < tg.het_full;
        if (w.whole) cl = jb;
        else { const unsigned r = jb - tg.het_full; cl = tg.het_full + r / tg.nseg; piece = r % tg.nseg; }
        const unsigned c = k * tg.het_cpx + cl;
        if (cl >= tg.het_cpx || c >= tg.ncol * tg.nbatch) return false;
        w.b = (int)(c / tg.ncol);
        w.ucol = c % tg.ncol;
        // one row per column, then nseg - 1 rows per split column of the batch entry (TileGeom::prows)
        w.prow = piece == 0 ? w.ucol
                            : tg.ncol + (xmarch_split_before(tg, c) - xmarch_split_before(tg, (unsigned)w.b * tg.ncol)) * (tg.nseg - 1) + (piece - 1);
        w.x0 = w.whole ? 0 : (int)(piece * tg.seglen);
        w.xlen = w.whole ? O0 : min((int)tg.seglen, O0 - w.x0);
        return true;
    }
    // one piece per block: XCD k owns the contiguous range [k * perU, (k + 1) * perU) of (batch, segment, patch)
    const unsigned per_batch = tg.ncol * tg.nseg, U = per_batch * tg.nbatch;
    const unsigned perU = tg.items_x ? tg.items_x : gridDim.x / NRT_NXCD;
    const unsigned u = k * perU + jb;
    if (jb >= perU || u >= U) return false;
    w.b = (int)(u / per_batch);
    w.prow = u % per_batch;
    w.ucol = w.prow % tg.ncol;
    w.x0 = (int)((w.prow / tg.ncol) * tg.seglen);
    w.xlen = min((int)tg.seglen, O0 - w.x0);
    w.whole = false;
    return true;
}
// Mixed lengths: the rows of a batch entry are padded to the entry with the most split columns; the block that marches the entry's first
// column (from x = 0) fills the entry's padding rows with zeros (and neutral extrema).  Every other row is written by its own item.
__device__ __forceinline__ void xmarch_zero_rows(const TileGeom &tg, const XmWork &w, int L3, float *fpart, float *mpart) {
    if (!tg.het_cpx || w.ucol != 0 || w.x0 != 0) return;
    const unsigned c0 = (unsigned)w.b * tg.ncol, cols = tg.ncol * tg.nbatch;
    const unsigned c1 = c0 + tg.ncol < cols ? c0 + tg.ncol : cols;
    const unsigned used = tg.ncol + (xmarch_split_before(tg, c1) - xmarch_split_before(tg, c0)) * (tg.nseg - 1);
    for (unsigned r = used; r < tg.prows; ++r) {
        const long long row = (long long)w.b * tg.prows + r;
        for (int i = threadIdx.x; i < L3; i += blockDim.x) fpart[row * L3 + i] = 0.0f;
        if (threadIdx.x < 4) mpart[row * 4 + threadIdx.x] = (threadIdx.x & 1) ? -INFINITY : INFINITY;
    }
}

// d (sum_c g_c warped_c) / d loc for one lane's four channels, from its pieces of the eight corner rows (corner = 4 cx + 2 cy + cz):
// the inner product of g with every corner row, then per axis the weighted differences of corner pairs
//   d/dx = m_x sum_{y,z} (wy wz) (dot[1,y,z] - dot[0,y,z])   and likewise for y and z   (m = 0 outside the volume: `clip` has no slope)
// TF's autodiff of utils.py:137-191 fixes no order for this sum, so (unlike the forward blend) multiply-adds are fused and the pairs
// of z-neighbours ride the packed instructions: 46 VALU instructions against 76 for the scalar form of rounds 2-4.
// The caller sums the three components over the lanes of the voxel.
__device__ __forceinline__ void loc_grad_rows(const nrt_f4 (&v)[8], nrt_f2 gl2, nrt_f2 gh2, float W0x, float W1x, float W0y, float W1y,
                                              float W0z, float W1z, float Mx, float My, float Mz, float (&gacc)[3]) {
    nrt_f2 D[4];                                                  // D[2 cx + cy] = {dot of corner cz = 0, of corner cz = 1}
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const nrt_f2 s0 = __builtin_elementwise_fma(gh2, (nrt_f2){v[2 * j][2], v[2 * j][3]}, gl2 * (nrt_f2){v[2 * j][0], v[2 * j][1]});
        const nrt_f2 s1 = __builtin_elementwise_fma(gh2, (nrt_f2){v[2 * j + 1][2], v[2 * j + 1][3]}, gl2 * (nrt_f2){v[2 * j + 1][0], v[2 * j + 1][1]});
        float d0 = s0[0] + s0[1], d1 = s1[0] + s1[1];
        asm("" : "+v"(d0), "+v"(d1));                             // (two adds: the compiler's packed form of them costs three moves)
        D[j] = (nrt_f2){d0, d1};
    }
    const nrt_f2 wz2 = {W0z, W1z};
    const nrt_f2 wyz0 = (nrt_f2){W0y, W0y} * wz2, wyz1 = (nrt_f2){W1y, W1y} * wz2;     // wy wz: [y0z0, y0z1], [y1z0, y1z1]
    const nrt_f2 wxz0 = (nrt_f2){W0x, W0x} * wz2, wxz1 = (nrt_f2){W1x, W1x} * wz2;     // wx wz
    const nrt_f2 ax = __builtin_elementwise_fma(D[3] - D[1], wyz1, (D[2] - D[0]) * wyz0);
    const nrt_f2 ay = __builtin_elementwise_fma(D[3] - D[2], wxz1, (D[1] - D[0]) * wxz0);
    // z: sum_j (wx wy)[j] (D[j][1] - D[j][0]) = the difference of the halves of sum_j (wx wy)[j] D[j]
    const float w00 = W0x * W0y, w01 = W0x * W1y, w10 = W1x * W0y, w11 = W1x * W1y;
    nrt_f2 az = (nrt_f2){w00, w00} * D[0];
    az = __builtin_elementwise_fma((nrt_f2){w01, w01}, D[1], az);
    az = __builtin_elementwise_fma((nrt_f2){w10, w10}, D[2], az);
    az = __builtin_elementwise_fma((nrt_f2){w11, w11}, D[3], az);
    gacc[0] = Mx * (ax[0] + ax[1]);
    gacc[1] = My * (ay[0] + ay[1]);
    gacc[2] = Mz * (az[1] - az[0]);
}


// default x-march tune for 32-channel volumes: 4 x 8 (y,z) patches, regions of 8 x 2 patches = 32 x 16 voxels.  (8 x 4 until round 5:
// with the round-5 gather the region shape is worth < 0.5 % at 4 volumes -- 16 x 2, 8 x 2, 4 x 4 within noise of each other -- but at
// ONE volume an XCD owns 100 columns, and regions of 16 patches lose fewer neighbours at the XCD boundaries than regions of 32:
// 0.275 -> 0.251 ms; tools/region_sweep.py, profiles/r05_lab/region_sweep.jsonl)
inline int xmarch_default_tune() { return 3 | (2 << 4) | (3 << 8) | (1 << 14) | (3 << 24) | (1 << 27); }
inline bool xmarch_applies(const int *out_shape, int batch) {
    const unsigned cols = ((unsigned)(out_shape[1] + 3) / 4) * ((unsigned)(out_shape[2] + 7) / 8);
    return out_shape[0] >= 16 && cols * (unsigned)batch >= 512;
}

// device: which (batch, patch, x segment) does this block own?  false = nothing (padding block)
__device__ __forceinline__ bool xmarch_block(const TileGeom &tg, int O0, int &b, unsigned &prow, int &x0, int &y0, int &z0, int &xlen) {
    const unsigned k = blockIdx.x % NRT_NXCD, jb = blockIdx.x / NRT_NXCD;
    const unsigned per_batch = tg.ncol * tg.nseg, U = per_batch * tg.nbatch;
    const unsigned perU = gridDim.x / NRT_NXCD;
    const unsigned u = k * perU + jb;
    if (jb >= perU || u >= U) return false;
    b = (int)(u / per_batch);
    prow = u % per_batch;
    const unsigned useg = prow / tg.ncol, ucol = prow % tg.ncol;
    const unsigned RY = 1u << tg.lry, RZ = 1u << tg.lrz;
    const unsigned nRz = (tg.nTz + RZ - 1) / RZ;
    const unsigned reg = ucol / (RY * RZ), w = ucol % (RY * RZ);
    const unsigned cy = (reg / nRz) * RY + w / RZ, cz = (reg % nRz) * RZ + w % RZ;
    x0 = (int)(useg * tg.seglen);
    y0 = (int)cy << tg.lty;
    z0 = (int)cz << tg.ltz;
    xlen = min((int)tg.seglen, O0 - x0);
    return cy < tg.nTy && cz < tg.nTz && xlen > 0;
}

struct TileMeta {
    float w0[3], w1[3];
    unsigned q;
    bool oob, valid;
};

inline int fill_args(InterpArgs &a, const void *vol, const float *loc, void *out, int ndim, const int *vol_shape,
              const int *out_shape, int channels, int batch, long long vol_bs, long long loc_bs, int loc_mode,
              int has_fill) {
    if (!vol || !out || !vol_shape || !out_shape) return NRT_ERR_INVALID_ARG;
    if (ndim < 1 || ndim > NRT_MAXD || channels < 1 || batch < 1) return NRT_ERR_INVALID_ARG;
    if (loc_mode < 0 || loc_mode > 2) return NRT_ERR_INVALID_ARG;
    if (loc_mode != NRT_LOC_LINSPACE && !loc) return NRT_ERR_INVALID_ARG;
    if (batch > 65535) return NRT_ERR_UNSUPPORTED;
    a.vol = vol; a.loc = loc; a.out = out; a.C = channels;
    unsigned long long nin = 1, nout = 1;
    for (int d = 0; d < NRT_MAXD; ++d) {
        a.S[d] = d < ndim ? vol_shape[d] : 1;
        a.O[d] = d < ndim ? out_shape[d] : 1;
        if (a.S[d] < 1 || a.O[d] < 0) return NRT_ERR_INVALID_ARG;
        nin *= (unsigned long long)a.S[d];
        nout *= (unsigned long long)a.O[d];
        // tf.linspace: delta = (stop - start) / (num - 1) in float32
        a.delta[d] = a.O[d] > 1 ? (float)(a.S[d] - 1) / (float)(a.O[d] - 1) : 0.0f;
    }
    if (nin * (unsigned long long)channels >= (1ull << 40) || nout >= (1ull << 31)) return NRT_ERR_UNSUPPORTED;
    a.nout = (unsigned)nout;
    a.vol_bs = vol_bs; a.loc_bs = loc_bs; a.out_bs = (long long)nout * channels;
    a.has_fill = has_fill ? 1 : 0;
    a.fill_f = 0.0f; a.fill_i = 0;
    a.addend = nullptr; a.addend_bs = 0;
    return NRT_OK;
}

}  // namespace
