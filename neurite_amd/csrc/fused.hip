// Fused SpatialTransformer + soft Dice for gfx950 (MI355X).
//
// The metric pipeline of BASELINE config 2/4 is  warped = SpatialTransformer(moving, trf);
// dice = Dice(fixed, warped).  Run as two kernels it moves 268 + 256 = 524 B per voxel:
// `warped` (128 B/voxel at L = 32) is written by one kernel and immediately re-read by the other.
// Here the blended row never leaves the registers: the lane that holds labels 4*lg..4*lg+3 of the warped
// voxel loads the same 16 bytes of `fixed` and accumulates sum t*p, sum t^2, sum p^2 (+ min/max) on the
// spot -- 268 + 128 - 128 = 268 B/voxel of algorithmic traffic, no `warped` tensor unless asked for.
//
// The arithmetic of the warp is bit-identical to interpn.hip (same op sequence), the Dice reduction is
// the same deterministic tree as dice.hip (lane registers -> wave xor-shuffles -> LDS -> per-block
// partial -> two-level fixed-order second stage), so fused and unfused results agree to the last bit
// in `warped` and to float32 reduction-order noise (<= 1e-6 relative) in the sums.
//
// Structure = interpn_tile (3-D output tiles, XCD-contiguous slabs, depth-2 software pipeline with
// unconditional loads); the `fixed` row rides along as a ninth load of every pass.

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "dice_reduce.h"
#include "interpn_core.h"
#include "wc.h"
#include "fused_wc.h"

namespace {

// Storage type of the two label maps.  float: 16 bytes per lane and row.  bfloat16 (unsigned short): 8 bytes per lane -- a
// 32-label row is 64 bytes, two z-neighbours share a cache line -- widened to float32 in registers (exact), after which the
// arithmetic is the float32 kernel's: for maps whose values are bfloat16 numbers (one-hot label maps are) the result is
// bit-identical to the float32 kernel on the widened maps.  "bf16 storage, fp32 math": an extension of this package, not a
// reference behaviour (TensorFlow would run the blend in bfloat16; interpn_any.hip does that for `interpn` on bf16 volumes).
typedef unsigned nrt_u2 __attribute__((ext_vector_type(2)));
template <typename ST> struct RowT;
template <> struct RowT<float> {
    typedef nrt_f4 T;
    static constexpr unsigned BYTES = 16;
    static __device__ __forceinline__ nrt_f4 widen(const nrt_f4 &t) { return t; }
};
template <> struct RowT<unsigned short> {
    typedef nrt_u2 T;
    static constexpr unsigned BYTES = 8;
    static __device__ __forceinline__ nrt_f4 widen(const nrt_u2 &t) {
        return (nrt_f4){__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16),
                        __uint_as_float(t[1] & 0xffff0000u)};
    }
};


// Passes between block barriers (power of two).  The four waves of a block are y-neighbours: half of a wave's corner rows are its
// neighbour's, and they merge in L1 only while the waves request them at about the same time.  Left alone the waves drift apart; a
// barrier every 8 passes keeps them together at no measurable cost: 1.174 -> 1.157 ms on one box, 1.148 -> 1.109 on another (every 2
// passes: slower; 16: the same; 32: less), profiles/archive/r03_lab/fused_occupancy_depth.jsonl
constexpr int FUSED_SYNC = 8;
// Waves per SIMD the x-march instance is compiled for (register budget 512 / MINW).  Two blocks of four waves run per CU (the LDS
// padding below), so 4 only squeezed the kernel into 128 registers with three of them spilled; at 3 it takes 130, nothing spills:
// 1.158 -> 1.140 ms (two alternating repeats, profiles/archive/r03_lab/fused_occupancy_depth.jsonl)
constexpr int FUSED_MINW = 3;

template <int G, int MODE, bool STORE, int MINW, typename ST = float>
__global__ __launch_bounds__(256, MINW) void warp_dice_tile(InterpArgs a, TileGeom tg, const void *__restrict__ fixed,
                                                      float *__restrict__ fpart, float *__restrict__ mpart) {
    typedef typename RowT<ST>::T Row;
    constexpr unsigned RB = RowT<ST>::BYTES;
    constexpr int NG = 256 / G;
    constexpr int L = 4 * G;
    // persistent blocks: block (k = XCD, jb) walks the tiles jb, jb + nb, jb + 2 nb ... of XCD k's slab and
    // writes ONE partial at the end (gridDim.x <= 2048 keeps the second stage short)
    const unsigned k = blockIdx.x % NRT_NXCD, jb = blockIdx.x / NRT_NXCD;
    unsigned per = tg.per2 * tg.nTz;                           // tiles per XCD
    unsigned nb = gridDim.x / NRT_NXCD;
    int b = blockIdx.y;
    unsigned ucol = 0, prow = blockIdx.x;                      // x-march: patch, partial row inside the batch
    XmWork xw = {};
    if (tg.x_march) {
        if (!xmarch_work(tg, a.O[0], xw)) return;              // one column (or piece of one) per block
        b = xw.b; prow = xw.prow; ucol = xw.ucol;
        per = jb + 1; nb = 1;                                  // the tile loop below runs exactly once
    }

    const char *volb = (const char *)a.vol + (long long)b * a.vol_bs * (long long)sizeof(ST);
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    nrt_f4 *out = (nrt_f4 *)((float *)a.out + (long long)b * a.out_bs);
    const char *fix = (const char *)fixed + (long long)b * a.out_bs * (long long)sizeof(ST);
    const int lg = threadIdx.x % G;
    const int g = threadIdx.x / G;
    int npass = tg.plane_major ? tg.tz : (1 << (tg.ltx + tg.lty + tg.ltz)) / NG;
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];

    // sums as float2 halves: the blend and the Dice accumulation run on v_pk_mul_f32 / v_pk_add_f32 (two IEEE fp32 operations
    // per issue slot, no fusion -> bit-identical to the scalar sequence); the kernel is bound by VALU issue, not by memory
    nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;

    for (unsigned j = jb; j < per; j += nb) {
        unsigned t2l, tzi;
        if (tg.z_outer) { tzi = j / tg.per2; t2l = j % tg.per2; }
        else { tzi = j % tg.nTz; t2l = j / tg.nTz; }
        const unsigned t2 = k * tg.per2 + t2l;
        if (!tg.x_march && t2 >= tg.nT2) continue;
        int x0 = (int)(t2 / tg.nTy) << tg.ltx, y0 = (int)(t2 % tg.nTy) << tg.lty, z0 = (int)tzi * tg.tz;
        if (tg.x_march) {
            x0 = xw.x0;
            // patches are enumerated region by region (2^lry x 2^lrz patches, row-major inside and across regions) so
            // that the blocks resident on an XCD at one time cover a compact (y,z) window
            const unsigned RY = 1u << tg.lry, RZ = 1u << tg.lrz;
            const unsigned nRz = (tg.nTz + RZ - 1) / RZ;
            const unsigned reg = ucol / (RY * RZ), w = ucol % (RY * RZ);
            const unsigned cy = (reg / nRz) * RY + w / RZ, cz = (reg % nRz) * RZ + w % RZ;
            y0 = (int)cy << tg.lty;
            z0 = (int)cz << tg.ltz;
            if (cy >= tg.nTy || cz >= tg.nTz) npass = 0;
            const int xlen = xw.xlen;
            if (npass) npass = (xlen << (tg.lty + tg.ltz)) / NG;
            if (npass <= 0) continue;
        }
        auto voxel = [&](int pass, int (&qd)[NRT_MAXD], bool &valid) {
            int x, y, z;
            tile_voxel(tg, NG, pass, g, x0, y0, z0, x, y, z);
            valid = (x < a.O[0]) && (y < a.O[1]) && (z < a.O[2]);
            qd[0] = min(x, a.O[0] - 1); qd[1] = min(y, a.O[1] - 1); qd[2] = min(z, a.O[2] - 1);
        };
        auto fetch_loc = [&](int pass, float (&p)[NRT_MAXD]) {
            int qd[NRT_MAXD]; bool valid;
            voxel(pass, qd, valid);
            // operands below 2^24 (checked by the C entry): full-rate 24-bit multiplies, 32-bit byte offsets from a uniform base
            const unsigned q = nrt_mad24(nrt_mad24((unsigned)qd[0], (unsigned)a.O[1], (unsigned)qd[1]), (unsigned)a.O[2], (unsigned)qd[2]);
            if (MODE != NRT_LOC_LINSPACE) {
                const float *lp = (const float *)((const char *)locb + (size_t)(nrt_times3(q) << 2));
                p[0] = lp[0]; p[1] = lp[1]; p[2] = lp[2];
            }
        };
        // the per-voxel state that travels from prepare() to finish() is passed as separate scalars: as members of one struct
        // the three lower weights were re-loaded pairwise, which kept the struct in memory (= in LDS, 24 B per thread)
        struct FM { float w0x, w0y, w0z, w1x, w1y, w1z; unsigned q; bool oob, valid; };
        auto prepare = [&](int pass, const float (&praw)[NRT_MAXD], float &W0x, float &W0y, float &W0z, unsigned &Q, bool &VALID,
                           bool &OOB, unsigned (&off)[8]) {
            FM m;
            int qd[NRT_MAXD];
            voxel(pass, qd, m.valid);
            m.q = nrt_mad24(nrt_mad24((unsigned)qd[0], (unsigned)a.O[1], (unsigned)qd[1]), (unsigned)a.O[2], (unsigned)qd[2]);
            float p[NRT_MAXD];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (MODE == NRT_LOC_ABSOLUTE) p[d] = praw[d];
                else if (MODE == NRT_LOC_SHIFT) p[d] = nrt_add((float)qd[d], praw[d]);
                else p[d] = (qd[d] == 0) ? 0.0f
                          : ((qd[d] == a.O[d] - 1) ? (float)(a.S[d] - 1) : nrt_mul(a.delta[d], (float)qd[d]));
            }
            int i0x, i1x, i0y, i1y, i0z, i1z;              // scalars, not arrays: an array here ends up in LDS
            corner_1d(p[0], a.S[0], i0x, i1x, m.w0x, m.w1x);
            corner_1d(p[1], a.S[1], i0y, i1y, m.w0y, m.w1y);
            corner_1d(p[2], a.S[2], i0z, i1z, m.w0z, m.w1z);
            m.oob = a.has_fill ? out_of_bounds<3>(a, p) : false;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const unsigned ix = (corner & 4) ? i1x : i0x;
                const unsigned iy = (corner & 2) ? i1y : i0y;
                const unsigned iz = (corner & 1) ? i1z : i0z;
                off[corner] = (nrt_mad24(nrt_mad24(ix, SY, iy), SZ, iz) * (unsigned)G + (unsigned)lg) * RB;
            }
            W0x = m.w0x; W0y = m.w0y; W0z = m.w0z; Q = m.q; VALID = m.valid; OOB = m.oob;
        };
        auto load_rows = [&](const unsigned (&off)[8], unsigned q, Row (&R)[8], Row &T) {
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) R[corner] = *(const Row *)(volb + (size_t)off[corner]);
            T = __builtin_nontemporal_load((const Row *)(fix + (size_t)((q * (unsigned)G + (unsigned)lg) * RB)));
        };
        auto finish = [&](float W0x, float W0y, float W0z, unsigned Q, bool VALID, bool OOB, const Row (&Rraw)[8], const Row &Traw) {
            nrt_f4 R[8];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) R[corner] = RowT<ST>::widen(Rraw[corner]);
            const nrt_f4 T = RowT<ST>::widen(Traw);
            FM m;
            m.w0x = W0x; m.w0y = W0y; m.w0z = W0z; m.q = Q; m.valid = VALID; m.oob = OOB;
            m.w1x = nrt_sub(1.0f, W0x); m.w1y = nrt_sub(1.0f, W0y); m.w1z = nrt_sub(1.0f, W0z);      // corner_1d's w1
            // corner weights (wx * wy) * wz in the reference's order, two corners per packed multiply
            const nrt_f2 wy2 = {m.w0y, m.w1y}, wz2 = {m.w0z, m.w1z};
            const nrt_f2 wxy0 = (nrt_f2){m.w0x, m.w0x} * wy2, wxy1 = (nrt_f2){m.w1x, m.w1x} * wy2;
            nrt_f2 wt2[4];
            wt2[0] = (nrt_f2){wxy0[0], wxy0[0]} * wz2;          // corners 0, 1
            wt2[1] = (nrt_f2){wxy0[1], wxy0[1]} * wz2;          // corners 2, 3
            wt2[2] = (nrt_f2){wxy1[0], wxy1[0]} * wz2;          // corners 4, 5
            wt2[3] = (nrt_f2){wxy1[1], wxy1[1]} * wz2;          // corners 6, 7
            nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const float wt = wt2[corner >> 1][corner & 1];
                const nrt_f2 w2 = {wt, wt};
                al = al + w2 * (nrt_f2){R[corner][0], R[corner][1]};
                ah = ah + w2 * (nrt_f2){R[corner][2], R[corner][3]};
            }
            nrt_f4 acc = {al[0], al[1], ah[0], ah[1]};
            if (a.has_fill) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = apply_fill(acc[c], m.oob, a.fill_f);
            }
            if (m.valid) {
                if (STORE) __builtin_nontemporal_store(acc, (nrt_f4 *)((char *)out + (size_t)((m.q * (unsigned)G + (unsigned)lg) * 16u)));
                const nrt_f2 pl = {acc[0], acc[1]}, ph = {acc[2], acc[3]}, tl = {T[0], T[1]}, th = {T[2], T[3]};
                stp_l = stp_l + tl * pl; stp_h = stp_h + th * ph;
                stt_l = stt_l + tl * tl; stt_h = stt_h + th * th;
                spp_l = spp_l + pl * pl; spp_h = spp_h + ph * ph;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    mnt = fminf(mnt, T[c]); mxt = fmaxf(mxt, T[c]);
                    mnp = fminf(mnp, acc[c]); mxp = fmaxf(mxp, acc[c]);
                }
            }
        };

        Row Ra[8], Rb[8], Ta, Tb;
        float Ax, Ay, Az, Bx, By, Bz;
        unsigned Aq, Bq;
        bool Av, Ao, Bv, Bo;
        unsigned off[8];
        float pn[NRT_MAXD] = {0.0f, 0.0f, 0.0f};
        const int last = npass - 1;
        fetch_loc(0, pn);
        prepare(0, pn, Ax, Ay, Az, Aq, Av, Ao, off);
        fetch_loc(min(1, last), pn);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(off, Aq, Ra, Ta);
        __builtin_amdgcn_sched_barrier(0);
        for (int pass = 0; pass < npass; pass += 2) {
            if ((pass & (FUSED_SYNC - 1)) == 0) __builtin_amdgcn_s_barrier();     // the block's four waves kept loosely in step
            prepare(min(pass + 1, last), pn, Bx, By, Bz, Bq, Bv, Bo, off);
            Bv = Bv && (pass + 1 < npass);
            __builtin_amdgcn_sched_barrier(0);
            fetch_loc(min(pass + 2, last), pn);
            load_rows(off, Bq, Rb, Tb);
            __builtin_amdgcn_sched_barrier(0);
            finish(Ax, Ay, Az, Aq, Av, Ao, Ra, Ta);
            __builtin_amdgcn_sched_barrier(0);
            prepare(min(pass + 2, last), pn, Ax, Ay, Az, Aq, Av, Ao, off);
            Av = Av && (pass + 2 < npass);
            __builtin_amdgcn_sched_barrier(0);
            fetch_loc(min(pass + 3, last), pn);
            load_rows(off, Aq, Ra, Ta);
            __builtin_amdgcn_sched_barrier(0);
            finish(Bx, By, Bz, Bq, Bv, Bo, Rb, Tb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    nrt_f4 stp = {stp_l[0], stp_l[1], stp_h[0], stp_h[1]}, stt = {stt_l[0], stt_l[1], stt_h[0], stt_h[1]},
           spp = {spp_l[0], spp_l[1], spp_h[0], spp_h[1]};
    // ---- block reduction (identical tree to dice_soft_vec) -------------------------------------
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        stp[c] = wave_xor_add(stp[c], G);
        stt[c] = wave_xor_add(stt[c], G);
        spp[c] = wave_xor_add(spp[c], G);
    }
    for (int off = 1; off < NRT_WAVE; off <<= 1) {
        mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
        mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
    }
    __shared__ float red[4][3 * L + 4];
    const int lane = threadIdx.x & (NRT_WAVE - 1), wv = threadIdx.x / NRT_WAVE;
    if (lane < G) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[wv][0 * L + 4 * lane + c] = stp[c];
            red[wv][1 * L + 4 * lane + c] = stt[c];
            red[wv][2 * L + 4 * lane + c] = spp[c];
        }
    }
    if (lane == 0) { red[wv][3 * L + 0] = mnt; red[wv][3 * L + 1] = mxt; red[wv][3 * L + 2] = mnp; red[wv][3 * L + 3] = mxp; }
    __syncthreads();
    const long long pbase = tg.x_march ? ((long long)b * tg.prows + prow) : ((long long)b * gridDim.x + blockIdx.x);
    for (int i = threadIdx.x; i < 3 * L; i += 256) {
        float s = red[0][i];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) s += red[w2][i];
        fpart[pbase * 3 * L + i] = s;
    }
    if (threadIdx.x < 4) {
        float m = red[0][3 * L + threadIdx.x];
        for (int w2 = 1; w2 < 4; ++w2)
            m = (threadIdx.x & 1) ? fmaxf(m, red[w2][3 * L + threadIdx.x]) : fminf(m, red[w2][3 * L + threadIdx.x]);
        mpart[pbase * 4 + threadIdx.x] = m;
    }
    if (tg.x_march) xmarch_zero_rows(tg, xw, 3 * L, fpart, mpart);
}

// the fused kernel writes one partial per block: size the workspace for its grid
inline unsigned fused_xmarch_grid(const TileGeom &tg, unsigned nblocks, int batch) {
    return tg.het_cpx ? NRT_NXCD * (tg.het_full + (tg.het_cpx - tg.het_full) * tg.nseg) : nrt_xcd_grid(nblocks * (unsigned)batch);
}

size_t fused_ws_bytes(unsigned nblocks, int L, int batch) {
    const size_t rows = (size_t)batch * nblocks;
    const size_t grp = (size_t)batch * ((nblocks + RED_ROWS - 1) / RED_ROWS);
    return rows * 3 * L * sizeof(float) + rows * 4 * sizeof(float) + 16 + grp * 3 * L * sizeof(double) +
           grp * 4 * sizeof(float) + 256;
}

// nblocks = partial rows per batch entry
void fused_geom(const int *out_shape, int G, int batch, int tune, TileGeom &tg, unsigned &nblocks) {
    tile_geometry(out_shape, G, tune, 3 | (3 << 4) | (4 << 8), tg, nblocks);   // default 8 x 8 x 16 tiles (profiles/r01)
    if (nblocks > (unsigned)DICE_MAX_BLOCKS) nblocks = DICE_MAX_BLOCKS;        // multiple of 8; blocks loop over tiles
    int t = tune <= 0 ? 0 : tune;
    if (t == 0 && G == 8 && xmarch_applies(out_shape, batch)) {
        // default for 32 labels: x-march over 4 x 8 (y,z) patches, blocks dealt to the XCDs region by region
        // (8 x 4 patches = 32 x 32 voxels).  Same speed as the 8 x 8 x 16 tiles (both sit on the L1-miss path, see
        // DESIGN.md 4.3) but 0.69x their L2-miss traffic: 1.11x instead of 1.61x the algorithmic bytes (profiles/).
        t = xmarch_default_tune();
        tile_geometry(out_shape, G, t, t, tg, nblocks);
    }
    if (((t >> 14) & 1) && !tg.plane_major && out_shape[0] > 0) {
        // segments left to us (bits 16-23 zero): whole columns + pieces for the last round; an explicit count: equal pieces
        unsigned grid;
        if (((t >> 16) & 0xff) == 0) nblocks = xmarch_setup_mixed(out_shape, batch, t, tg, grid);
        else nblocks = xmarch_setup(out_shape, batch, t, tg);
    }
}

// tune bit 30: keep the register kernel where the wave-cache kernel (fused_wc.h, the default for 32 float32 labels) would apply;
// bit 29 (historical: ask for the wave-cache kernel) is accepted and ignored
constexpr int FUSED_TUNE_WC = 1 << 29, FUSED_TUNE_NO_WC = 1 << 30;

// Which kernel a call launches -- ONE decision, used by the launch and by nrt_warp_dice_kernel_name (bench.py keys its counter
// evidence on that name)
struct FusedChoice { bool wc, persist; };
inline FusedChoice fused_choose(const InterpArgs &a, const TileGeom &tg, int G, bool f32_storage, bool allow_wc) {
    FusedChoice c = {false, false};
    if (G == 8 && f32_storage && allow_wc && wc_applies(tg, G, a)) {
        c.wc = true;
        c.persist = wc_persistent(tg);
    }
    return c;
}

template <int G, typename ST>
int launch_fused(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, int mode, bool store,
                 const void *fixed, float *fpart, float *mpart, hipStream_t st, bool allow_wc, bool want_minmax) {
    if constexpr (G == 8 && std::is_same<ST, float>::value) {
        if (fused_choose(a, tg, G, true, allow_wc).wc)
            return launch_wc(a, tg, nblocks, batch, mode, store, want_minmax, (const float *)fixed, fpart, mpart, st);
    }
    dim3 grid(nblocks, batch), blk(256);
    // x-march: 75 KB of unused dynamic LDS per block = two blocks per CU, so that the blocks an XCD runs together are one region whose
    // rows stay in its L2
    const unsigned dyn = tg.x_march ? 75u * 1024u : 0u;
    if (tg.x_march) {
        grid = dim3(fused_xmarch_grid(tg, nblocks, batch), 1);
#define NRT_FUSED_X(MODE)                                                                                           \
    if (store) hipLaunchKernelGGL((warp_dice_tile<G, MODE, true, FUSED_MINW, ST>), grid, blk, 0, st, a, tg, fixed, fpart, mpart); \
    else {                                                                                                          \
        if (hipFuncSetAttribute((const void *)warp_dice_tile<G, MODE, false, FUSED_MINW, ST>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return NRT_ERR_LAUNCH; \
        hipLaunchKernelGGL((warp_dice_tile<G, MODE, false, FUSED_MINW, ST>), grid, blk, dyn, st, a, tg, fixed, fpart, mpart); \
    }
        switch (mode) {
            case NRT_LOC_ABSOLUTE: NRT_FUSED_X(NRT_LOC_ABSOLUTE); break;
            case NRT_LOC_SHIFT: NRT_FUSED_X(NRT_LOC_SHIFT); break;
            default: NRT_FUSED_X(NRT_LOC_LINSPACE); break;
        }
#undef NRT_FUSED_X
        return NRT_OK;
    }
#define NRT_FUSED(MODE)                                                                                          \
    if (store) hipLaunchKernelGGL((warp_dice_tile<G, MODE, true, 1, ST>), grid, blk, 0, st, a, tg, fixed, fpart, mpart); \
    else hipLaunchKernelGGL((warp_dice_tile<G, MODE, false, 1, ST>), grid, blk, 0, st, a, tg, fixed, fpart, mpart);
    switch (mode) {
        case NRT_LOC_ABSOLUTE: NRT_FUSED(NRT_LOC_ABSOLUTE); break;
        case NRT_LOC_SHIFT: NRT_FUSED(NRT_LOC_SHIFT); break;
        default: NRT_FUSED(NRT_LOC_LINSPACE); break;
    }
#undef NRT_FUSED
    return NRT_OK;
}

}  // namespace

extern "C" size_t nrt_warp_dice_workspace_bytes(const int *out_shape, int nlabels, int batch, int tune) {
    if (!out_shape || nlabels < 4 || nlabels % 4 || batch < 1) return 0;
    TileGeom tg;
    unsigned nblocks;
    fused_geom(out_shape, nlabels / 4, batch, tune > 0 ? (tune & ~(FUSED_TUNE_NO_WC | FUSED_TUNE_WC)) : tune, tg, nblocks);
    return fused_ws_bytes(nblocks, nlabels, batch);
}

namespace {
template <typename ST>
int warp_dice_soft_impl(const void *moving, const float *loc, const void *fixed, float *warped, const int *vol_shape,
                        const int *out_shape, int nlabels, int batch, long long loc_batch_stride, int loc_mode, int has_fill,
                        float fill_value, float laplace_smoothing, float *sums, float *dice, float *minmax, int tune,
                        void *workspace, size_t workspace_bytes, void *stream) {
    if (!fixed || !sums || !dice) return NRT_ERR_INVALID_ARG;
    if (nlabels % 4) return NRT_ERR_UNSUPPORTED;
    const int G = nlabels / 4;
    if (!(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64)) return NRT_ERR_UNSUPPORTED;
    InterpArgs a;
    long long vol_bs = (long long)nlabels;
    if (vol_shape) for (int d = 0; d < 3; ++d) vol_bs *= vol_shape[d];
    float dummy;
    int rc = fill_args(a, moving, loc, warped ? (void *)warped : (void *)&dummy, 3, vol_shape, out_shape, nlabels, batch,
                       vol_bs, loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    if (!warped) a.out = nullptr;
    a.fill_f = fill_value;
    if ((unsigned long long)vol_bs * sizeof(ST) >= (1ull << 32)) return NRT_ERR_UNSUPPORTED;
    // the kernel addresses `fixed` / `warped` / `loc` rows with 32-bit byte offsets and forms row indices with 24-bit multiplies
    if ((unsigned long long)a.nout * (unsigned long long)nlabels * 4ull >= (1ull << 32)) return NRT_ERR_UNSUPPORTED;
    if (!std::is_same<ST, float>::value && warped) return NRT_ERR_UNSUPPORTED;      // the warped volume is a float32 output
    if ((long long)vol_shape[0] * vol_shape[1] >= (1 << 24) || vol_shape[2] >= (1 << 24) ||
        (long long)out_shape[0] * out_shape[1] >= (1 << 24) || out_shape[2] >= (1 << 24)) return NRT_ERR_UNSUPPORTED;
    if ((((uintptr_t)moving | (uintptr_t)fixed | (uintptr_t)warped) & 15) != 0) return NRT_ERR_INVALID_ARG;
    if (a.nout == 0) return NRT_ERR_INVALID_ARG;
    TileGeom tg;
    unsigned nblocks;
    const bool allow_wc = !(tune > 0 && (tune & FUSED_TUNE_NO_WC));
    if (tune > 0) tune &= ~(FUSED_TUNE_NO_WC | FUSED_TUNE_WC);
    fused_geom(out_shape, G, batch, tune, tg, nblocks);
    if (!workspace || workspace_bytes < fused_ws_bytes(nblocks, nlabels, batch)) return NRT_ERR_WORKSPACE;
    // carve: fpart, mpart, gsum, gmm
    DiceWs w;
    const size_t rows = (size_t)batch * nblocks;
    char *p = (char *)workspace;
    w.fpart = (float *)p; p += rows * 3 * nlabels * sizeof(float);
    w.mpart = (float *)p; p += rows * 4 * sizeof(float);
    p = (char *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    w.gsum = (double *)p; p += (size_t)batch * ((nblocks + RED_ROWS - 1) / RED_ROWS) * 3 * nlabels * sizeof(double);
    w.gmm = (float *)p; p += (size_t)batch * ((nblocks + RED_ROWS - 1) / RED_ROWS) * 4 * sizeof(float);
    w.ipart = nullptr;
    hipStream_t st = nrt_stream(stream);
    const bool store = warped != nullptr;
    switch (G) {
        case 1: rc = launch_fused<1, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
        case 2: rc = launch_fused<2, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
        case 4: rc = launch_fused<4, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
        case 8: rc = launch_fused<8, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
        case 16: rc = launch_fused<16, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
        case 32: rc = launch_fused<32, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
        default: rc = launch_fused<64, ST>(a, tg, nblocks, batch, loc_mode, store, fixed, w.fpart, w.mpart, st, allow_wc, minmax != nullptr); break;
    }
    if (rc != NRT_OK) return rc;              // nothing (or not everything) was launched: do not reduce stale partial sums
    NRT_CHECK_LAUNCH();
    return dice_finalize_soft(w, nblocks, 1, batch, nlabels, laplace_smoothing, sums, dice, minmax, st);
}
}  // namespace

// The kernel (template instantiation, spelled as rocprofv3 prints it without its namespace) that nrt_warp_dice_soft_f32 launches for
// these arguments: bench.py joins its timing with the counter passes under profiles/ by this name and refuses to quote HBM traffic
// that was measured on another instantiation.
extern "C" const char *nrt_warp_dice_kernel_name(const int *out_shape, const int *vol_shape, int nlabels, int batch, int loc_mode,
                                                 int has_fill, int store, int want_minmax, int tune) {
    static thread_local char name[96];
    name[0] = 0;
    // (the float32 entry point; "" for arguments it would reject)
    if (!out_shape || !vol_shape || nlabels < 4 || nlabels % 4 || batch < 1 || loc_mode < 0 || loc_mode > 2) return name;
    const int G = nlabels / 4;
    if (!(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64)) return name;
    unsigned long long nout = 1, nvol = 1;
    for (int d = 0; d < 3; ++d) {
        if (out_shape[d] < 1 || vol_shape[d] < 1) return name;
        nout *= (unsigned long long)out_shape[d]; nvol *= (unsigned long long)vol_shape[d];
    }
    if (nvol * nlabels * 4ull >= (1ull << 32) || nout * nlabels * 4ull >= (1ull << 32)) return name;
    if ((long long)vol_shape[0] * vol_shape[1] >= (1 << 24) || vol_shape[2] >= (1 << 24) ||
        (long long)out_shape[0] * out_shape[1] >= (1 << 24) || out_shape[2] >= (1 << 24)) return name;
    const bool allow_wc = !(tune > 0 && (tune & FUSED_TUNE_NO_WC));
    if (tune > 0) tune &= ~(FUSED_TUNE_NO_WC | FUSED_TUNE_WC);
    TileGeom tg;
    unsigned nblocks;
    fused_geom(out_shape, G, batch, tune, tg, nblocks);
    InterpArgs a = {};
    for (int d = 0; d < 3; ++d) { a.S[d] = vol_shape[d]; a.O[d] = out_shape[d]; }
    const char *tf[2] = {"false", "true"};
    const FusedChoice c = fused_choose(a, tg, G, true, allow_wc);
    if (c.wc)
        snprintf(name, sizeof(name), "warp_dice_wc<%d, %s, %s, %s, true, %s>", loc_mode, tf[store != 0], tf[want_minmax != 0], tf[has_fill != 0],
                 tf[c.persist]);
    else
        snprintf(name, sizeof(name), "warp_dice_tile<%d, %d, %s, %d, float>", G, loc_mode, tf[store != 0], tg.x_march ? FUSED_MINW : 1);
    return name;
}

extern "C" int nrt_warp_dice_soft_f32(const float *moving, const float *loc, const float *fixed, float *warped,
                                      const int *vol_shape, const int *out_shape, int nlabels, int batch,
                                      long long loc_batch_stride, int loc_mode, int has_fill, float fill_value,
                                      float laplace_smoothing, float *sums, float *dice, float *minmax,
                                      int tune, void *workspace, size_t workspace_bytes, void *stream) {
    return warp_dice_soft_impl<float>(moving, loc, fixed, warped, vol_shape, out_shape, nlabels, batch, loc_batch_stride, loc_mode,
                                      has_fill, fill_value, laplace_smoothing, sums, dice, minmax, tune, workspace, workspace_bytes,
                                      stream);
}

extern "C" int nrt_warp_dice_soft_bf16(const void *moving, const float *loc, const void *fixed, const int *vol_shape,
                                       const int *out_shape, int nlabels, int batch, long long loc_batch_stride, int loc_mode,
                                       int has_fill, float fill_value, float laplace_smoothing, float *sums, float *dice,
                                       float *minmax, int tune, void *workspace, size_t workspace_bytes, void *stream) {
    return warp_dice_soft_impl<unsigned short>(moving, loc, fixed, nullptr, vol_shape, out_shape, nlabels, batch, loc_batch_stride,
                                               loc_mode, has_fill, fill_value, laplace_smoothing, sums, dice, minmax, tune, workspace,
                                               workspace_bytes, stream);
}
