// Fused SpatialTransformer + soft Dice for 32 float32 labels with a WAVE-PRIVATE LDS ROW CACHE (gfx950).
//
// Why (profiles/r04_lab/l1_access_curve.jsonl): the register kernel (warp_dice_tile) requests all 8 corner rows
// of every voxel from L1.  Dropping corner requests from 8 to 4 / 2 / 1 per voxel with everything else unchanged takes the
// launch from 1.085 ms to 0.843 / 0.788 / 0.755: the cost is the NUMBER OF ROW REQUESTS that pass the texture-address / L1
// pipe (hit or miss), not HBM.  Reading the 8 corners from LDS instead (ds_read_b128: 256 B/clk/CU, its own pipe) with 2 rows
// per voxel through L1 measures 0.88 ms (EXP=20 probe).  So: fetch every DISTINCT row once per wave and pass, keep it across
// passes, blend from LDS.
//
// Structure.  The x-march schedule of warp_dice_tile is kept (a block of four waves owns a 4 x 8 (y,z) patch and walks x, one
// x-plane per pass; blocks dealt to the XCDs region by region).  A wave owns a 2 x 4 sub-patch: 8 voxels per pass = one voxel
// per 8-lane group, lane p of a group holds labels 4p..4p+3 (16 bytes of every row).  Each wave has its own direct-mapped cache
// of 128 rows (16 KB) + 16 overflow rows in LDS; NOTHING is shared between waves, so there is no barrier, no claim protocol and
// no atomic anywhere: a wave's LDS operations execute in program order.  Per pass:
//   1. every lane computes the corner geometry of its group's voxel (as the register kernel does) and takes ONE of the 64
//      (voxel, corner) references of the pass: row id `rid`, slot = (ix & 3, iy & 3, iz & 7);
//   2. probe: read tag[slot]; references that miss write {rid, lane}; everybody reads the tag back.  tag.rid == rid: the row is
//      (or will be) in the slot -- `served`; the missing lane whose lane id came back is its `loader`; a reference whose slot now
//      names another row of this same pass is an `orphan` (about 2 % of the references on the bench field): it gets an overflow
//      row for this pass only;
//   3. loaders + orphans are compacted (ballot / mbcnt) into a fetch list in LDS; the list is read back so that 8-lane groups
//      fetch one 128-byte row each: ceil(n / 8) load instructions per pass instead of 8 (n = 21 on the bench field), into
//      registers;
//   4. the fetched rows are stored to their slots at the top of the NEXT loop iteration (the loads of pass p + 1 fly while pass p is
//      blended: the registers are the double buffer, the cache needs none);
//   5. blend: 8 ds_read_b128 per lane from the slots recorded in step 2 (broadcast through 64 bytes of LDS), then exactly the
//      arithmetic of warp_dice_tile / interpn.hip (same op order: bit-identical warped rows).
// A pass with more than 16 orphans (incoherent fields) invalidates the cache and takes the register path for that pass.
//
// Model of the request stream (tools/wc_sim.py, bench field): 4.8 distinct rows per voxel inside a pass, 2.7 fetched per voxel with
// the cache (8 through L1 in the register kernel), 0.16 orphan references per voxel.

#pragma once

#include "dice_reduce.h"
#include "interpn_core.h"

namespace {

constexpr int WC_SLOTS = 128;                                   // direct-mapped rows per wave
constexpr int WC_OVF = 16;                                      // overflow rows (orphans of the current pass)
constexpr int WC_TRASH_ROW = WC_SLOTS + WC_OVF;                 // where the lanes without a row of their own store (branch-free masking)
constexpr int WC_ROWS = WC_TRASH_ROW + 1;
constexpr unsigned WC_TAGS_OFF = WC_ROWS * 128;                 // {rid, lane} per slot, + one trash entry
constexpr unsigned WC_LISTR_OFF = WC_TAGS_OFF + (WC_SLOTS + 2) * 8;   // fetch list: row ids, transposed (entry e at (e & 7) * 8 + (e >> 3)), + trash
constexpr unsigned WC_LISTD_OFF = WC_LISTR_OFF + 68 * 4;        // fetch list: destination rows (bytes), same order, + trash
constexpr unsigned WC_BC_OFF = WC_LISTD_OFF + 80;               // byte offset of the source row of every (voxel, corner) reference (words)
constexpr unsigned WC_WAVE_BYTES = WC_BC_OFF + 256;             // 20208
constexpr unsigned WC_BLOCK_BYTES = 4 * WC_WAVE_BYTES;          // 80832: two blocks per CU
static_assert(WC_LISTR_OFF % 16 == 0 && WC_LISTD_OFF % 16 == 0 && WC_BC_OFF % 16 == 0 && WC_WAVE_BYTES % 16 == 0, "LDS layout");
static_assert(2 * WC_BLOCK_BYTES <= 160 * 1024, "two blocks per CU");

#ifndef NRT_FUSED_WCLOADS
#define NRT_FUSED_WCLOADS 0
#endif
// fused form (the sums stay in registers): row loads only as far as the fetch list goes.  Stand-alone warp (the blended row is STORED):
// always eight loads -- with a conditional number the compiler's in-order vmcnt count must assume none was issued, every wait then
// also covered the store of the previous pass, and the write acknowledgement sat in the wave's critical path every pass
// (stand-alone warp 1.305 -> 1.222 ms; the fused kernel is slower with fixed loads: 1.068 -> 1.143, profiles/r04_lab/wc_fixed8.txt)
#define WC_FIXED_LOADS (DICE ? NRT_FUSED_WCLOADS : 8)
#ifndef NRT_FUSED_WCSYNC
#define NRT_FUSED_WCSYNC 0
#endif
#ifndef NRT_WC_SYNC
#define NRT_WC_SYNC NRT_FUSED_WCSYNC          // passes between block barriers; 0 = none (measured: 1.093 none, 1.15 / 1.12 / 1.10 for 2 / 8 / 32)
#endif

// LDS pointers carry their address space explicitly: a volatile access through a generic pointer compiles to flat_load / flat_store
// with a full vmcnt(0) wait around it
typedef __attribute__((address_space(3))) char wc_lds_char;
typedef __attribute__((address_space(3))) nrt_f4 wc_lds_f4;
typedef __attribute__((address_space(3))) volatile nrt_i4 wc_lds_vi4;
typedef __attribute__((address_space(3))) volatile unsigned long long wc_lds_vu64;
typedef __attribute__((address_space(3))) volatile unsigned wc_lds_vu32;
typedef __attribute__((address_space(3))) volatile unsigned char wc_lds_vu8;
typedef unsigned wc_u4 __attribute__((ext_vector_type(4)));
typedef float wc_f3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ unsigned wc_byte(unsigned lo, unsigned hi, int i) {
    return ((i < 4 ? lo : hi) >> ((i & 3) * 8)) & 0xffu;
}

// utils.py:139-153 for one dimension with three operations less than corner_1d and the same bits: l0 = floor(clip(p)) is
// clip(floor(p)) for every float (the bounds are integers; NaN clips to 0 either way), and l0 + 1 >= 1 needs no lower clip
__device__ __forceinline__ void wc_corner(float pv, float mx, int &i0, int &i1, float &w0) {
    const float cl = fminf(fmaxf(pv, 0.0f), mx);
    const float l0 = floorf(cl);
    const float l1 = fminf(nrt_add(l0, 1.0f), mx);
    i0 = (int)l0; i1 = (int)l1;
    w0 = nrt_sub(l1, cl);
}

// one work item (a column of a 4 x 8 patch, or a piece of one) by one block.
// DICE = false: the warp alone (nrt_interpn_f32 variant 10): no fixed map, no sums, no partials -- the same gather and blend
template <int MODE, bool STORE, bool MM, bool FILL, bool DICE>
__device__ __forceinline__ void wc_item(const InterpArgs &a, const TileGeom &tg, const float *__restrict__ fixed, float *__restrict__ fpart,
                                        float *__restrict__ mpart, const XmWork &xw, char *wc_smem) {
    constexpr int G = 8, L = 32;
    const int b = xw.b;
    const unsigned prow = xw.prow, ucol = xw.ucol;
    const unsigned RY = 1u << tg.lry, RZ = 1u << tg.lrz;
    const unsigned nRz = (tg.nTz + RZ - 1) / RZ;
    const unsigned reg = ucol / (RY * RZ), wr = ucol % (RY * RZ);
    const unsigned cy = (reg / nRz) * RY + wr / RZ, cz = (reg % nRz) * RZ + wr % RZ;
    const int x0 = xw.x0, y0 = (int)cy << 2, z0 = (int)cz << 3;
    int npass = xw.xlen;                                         // one x-plane of the patch per pass
    if (cy >= tg.nTy || cz >= tg.nTz || npass < 0) npass = 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, p = lane & 7;                       // group = voxel of the pass, p = 16-byte piece of a row / corner of the reference
    // voxel of group g inside the wave's 2 x 4 sub-patch.  ds_read_b128 serves lanes in sets that pair the groups (0,3), (1,2),
    // (4,7), (5,6): a pair reads without bank conflict when its two rows sit in different halves of the banks = slots of different
    // parity = (mostly) z-neighbours, so those pairs are z-neighbours: z offsets 0, 2, 3, 1 for g & 3 = 0, 1, 2, 3
    const int dzt = (0x1320 >> ((g & 3) * 4)) & 15;
    const int yy = y0 + 2 * (wave >> 1) + (g >> 2), zz = z0 + 4 * (wave & 1) + dzt;
    const bool yzvalid = (yy < a.O[1]) && (zz < a.O[2]);
    const int yc = min(yy, a.O[1] - 1), zc = min(zz, a.O[2] - 1);
    const unsigned qyz = (unsigned)yc * (unsigned)a.O[2] + (unsigned)zc, qstep = (unsigned)a.O[1] * (unsigned)a.O[2];
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];
    const float mxx = (float)(a.S[0] - 1), mxy = (float)(a.S[1] - 1), mxz = (float)(a.S[2] - 1);
    const float fyc = (float)yc, fzc = (float)zc;
    // linspace locations along y and z do not change along the march
    const float ly = (yc == 0) ? 0.0f : ((yc == a.O[1] - 1) ? mxy : nrt_mul(a.delta[1], fyc));
    const float lz = (zc == 0) ? 0.0f : ((zc == a.O[2] - 1) ? mxz : nrt_mul(a.delta[2], fzc));

    // wave-uniform bases advance with the pass, the lane contributes constant 32-bit offsets
    const char *volb = (const char *)a.vol + (long long)b * a.vol_bs * 4ll;
    const char *locb = (const char *)(a.loc ? a.loc + (long long)b * a.loc_bs : nullptr);
    char *outb = (char *)((float *)a.out + (long long)b * a.out_bs);
    const char *fixb = (const char *)fixed + (long long)b * a.out_bs * 4ll;
    const unsigned loc_lane = qyz * 12u, row_lane = (qyz * 8u + (unsigned)p) * 16u;
    const unsigned long long volbytes = (unsigned long long)a.S[0] * SY * SZ * 128ull;       // < 2^32 (checked by the C entry)
    // (the descriptor has to be wave-uniform for the compiler too: built from readfirstlane'd halves, or every load becomes a waterfall loop)
    const unsigned long long vb64 = (unsigned long long)volb;
    const void *volb_u = (const void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(vb64 >> 32)) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)vb64));   // (it returns int)
    // offsets at or past num_records read zeros without touching memory: that is how lanes (and list entries) without a row are masked
    const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc((void *)volb_u, 0, __builtin_amdgcn_readfirstlane((int)(unsigned)volbytes),
                                                                           0x00020000);

    wc_lds_char *wl = (wc_lds_char *)wc_smem + wave * WC_WAVE_BYTES;
    wc_lds_vu64 *tags = (wc_lds_vu64 *)(wl + WC_TAGS_OFF);
    wc_lds_vu32 *listR = (wc_lds_vu32 *)(wl + WC_LISTR_OFF);
    wc_lds_vu8 *listD = (wc_lds_vu8 *)(wl + WC_LISTD_OFF);
    wc_lds_vu32 *bc = (wc_lds_vu32 *)(wl + WC_BC_OFF);
    wc_lds_char *lrow = wl + p * 16;                              // this lane's piece of row 0
    tags[lane] = ~0ull;                                          // no row id is 0xffffffff
    tags[lane + 64] = ~0ull;
    const unsigned long long lane_hi = (unsigned long long)lane << 32;
    const bool cx1 = (p & 4) != 0, cy1 = (p & 2) != 0, cz1 = (p & 1) != 0;

    nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;

    // ---- per-pass state.  Two passes are in flight: the loop is unrolled by two and the passes alternate between the states A and B
    // (deliver(X) -> blend(X) -> manage(X <- pass + 2): the rows of pass p + 1 and p + 2 fly while pass p is blended)
    struct Pass {
        float w0x, w0y, w0z;         // lower-corner weights
        unsigned sl[8];              // byte offset (row * 128) of the source row of the 8 corners in this wave's LDS rows
        unsigned fd_lo, fd_hi;       // destination row of fetch-list entries g, 8 + g, ...
        int n;                       // wave-uniform: entries of the fetch list; -1: register path (F = the voxel's 8 corner rows)
        bool oob;
        nrt_f4 T;                    // the fixed row
        nrt_f4 F[8];                 // rows in flight
        float pn[3];                 // location of the pass that will use this state next
    };
    Pass A, B;
    auto reset = [&](Pass &s) {
        s.w0x = s.w0y = s.w0z = 0.f; s.fd_lo = s.fd_hi = 0; s.n = 0; s.oob = false;
        s.T = (nrt_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.F[i] = (nrt_f4){0.f, 0.f, 0.f, 0.f}; s.sl[i] = 0; }
        s.pn[0] = s.pn[1] = s.pn[2] = 0.f;
    };
    reset(A); reset(B);

    auto fetch_loc = [&](int pass, Pass &s) {
        if (MODE != NRT_LOC_LINSPACE) {
            const float *lp = (const float *)(locb + (size_t)((unsigned)(x0 + pass) * qstep) * 12u + loc_lane);
            s.pn[0] = lp[0]; s.pn[1] = lp[1]; s.pn[2] = lp[2];
        }
    };
    auto fetch_row = [&](unsigned rid) -> nrt_f4 {
        return __builtin_bit_cast(nrt_f4, __builtin_amdgcn_raw_buffer_load_b128(vres, (rid * 8u + (unsigned)p) * 16u, 0, 0));
    };

    // Incoherent fields: a pass with more colliding rows than the overflow holds takes the register path (its 8 corner rows straight
    // from memory).  NRT_WC_SKIP > 0 would also skip the probing of the next passes; measured it only hurts: on the bench field 0.85 %
    // of the passes overflow and each skip leaves the cache cold (1.26 ms with 15 against 1.09), on U(-80, 80) displacements nothing
    // is gained (1.886 / 1.874 ms, register kernel 1.85): profiles/r04_lab/wc_skip_variants.jsonl.
#ifndef NRT_FUSED_WCSKIP
#define NRT_FUSED_WCSKIP 0
#endif
#define NRT_WC_SKIP NRT_FUSED_WCSKIP
#ifndef NRT_FUSED_WCBLENDF
#define NRT_FUSED_WCBLENDF 0      // 1: a register-path pass is blended straight from its registers (a second copy of the blend in the loop)
#endif
    int skip = 0;                                                // wave-uniform
    // geometry + cache management + fetch of pass `pass` (its location is in s.pn); leaves the pass in s
    auto manage = [&](int pass, Pass &s) {
        const int xq = x0 + pass;
        float px, py, pz;
        if (MODE == NRT_LOC_ABSOLUTE) { px = s.pn[0]; py = s.pn[1]; pz = s.pn[2]; }
        else if (MODE == NRT_LOC_SHIFT) { px = nrt_add((float)xq, s.pn[0]); py = nrt_add(fyc, s.pn[1]); pz = nrt_add(fzc, s.pn[2]); }
        else { px = (xq == 0) ? 0.0f : ((xq == a.O[0] - 1) ? mxx : nrt_mul(a.delta[0], (float)xq)); py = ly; pz = lz; }
        int i0x, i1x, i0y, i1y, i0z, i1z;
        wc_corner(px, mxx, i0x, i1x, s.w0x);
        wc_corner(py, mxy, i0y, i1y, s.w0y);
        wc_corner(pz, mxz, i0z, i1z, s.w0z);
        if (FILL) s.oob = (px < 0.0f) || (px > mxx) || (py < 0.0f) || (py > mxy) || (pz < 0.0f) || (pz > mxz);
        auto direct = [&]() {                                     // this voxel's 8 corner rows straight into F (corner order)
            s.n = -1;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const unsigned jx = (corner & 4) ? i1x : i0x, jy = (corner & 2) ? i1y : i0y, jz = (corner & 1) ? i1z : i0z;
                s.F[corner] = fetch_row(nrt_mad24(nrt_mad24(jx, SY, jy), SZ, jz));
            }
        };
        if (__builtin_expect(skip > 0, 0)) {
            --skip;
            direct();
            if (DICE) s.T = __builtin_nontemporal_load((const nrt_f4 *)(fixb + (size_t)((unsigned)xq * qstep) * 128u + row_lane));
            return;
        }
        // this lane's reference: corner p of the group's voxel
        const unsigned ix = cx1 ? i1x : i0x, iy = cy1 ? i1y : i0y, iz = cz1 ? i1z : i0z;
        const unsigned rid = nrt_mad24(nrt_mad24(ix, SY, iy), SZ, iz);
        const unsigned slot = ((ix & 3u) << 5) | ((iy & 3u) << 3) | (iz & 7u);
        const unsigned long long t1 = tags[slot];
        const bool miss = (unsigned)t1 != rid;
        tags[miss ? slot : (unsigned)WC_SLOTS] = (unsigned long long)rid | lane_hi;       // hits store to the trash entry
        const unsigned long long t2 = tags[slot];
        const bool served = (unsigned)t2 == rid;
        const bool loader = miss && served && (unsigned)(t2 >> 32) == (unsigned)lane;
        const bool orphan = !served;
        const unsigned long long Ml = __builtin_amdgcn_ballot_w64(loader), Mo = __builtin_amdgcn_ballot_w64(orphan);
        const int nl = __builtin_popcountll(Ml), no = __builtin_popcountll(Mo);
        if (__builtin_expect(no > WC_OVF, 0)) {
            // incoherent field: more rows collide than the overflow holds.  Forget the cache (the tags written above name rows that
            // will not be fetched) and fetch this voxel's 8 corner rows directly
            tags[lane] = ~0ull;                                   // (all of it: deliver() parks the pass's 64 corner rows in rows 0 .. 63)
            tags[lane + 64] = ~0ull;
            skip = NRT_WC_SKIP;
            direct();
        } else {
            // EVERY lane writes one entry of the fetch list: loaders and orphans their row and its destination at positions 0 .. n - 1,
            // the others a row id past the volume (reads as zeros, touches no memory) with the trash row as destination at n .. 63 --
            // so fetching and delivering need no per-lane condition, only the wave-uniform count of load instructions
            const unsigned long long Mf = Ml | Mo;
            const unsigned rl = __builtin_amdgcn_mbcnt_hi((unsigned)(Ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Ml, 0u));
            const unsigned ro = __builtin_amdgcn_mbcnt_hi((unsigned)(Mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Mo, 0u));
            const unsigned rf = __builtin_amdgcn_mbcnt_hi((unsigned)(Mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Mf, 0u));
            s.n = nl + no;
            const unsigned src = orphan ? (unsigned)WC_SLOTS + ro : slot;
            const unsigned e = loader ? rl : (orphan ? (unsigned)nl + ro : (unsigned)s.n + ((unsigned)lane - rf));
            const bool fetcher = loader || orphan;
            const unsigned tp = ((e & 7u) << 3) | (e >> 3);
            listR[tp] = fetcher ? rid : 0x01ffffffu;
            listD[tp] = (unsigned char)(fetcher ? src : (unsigned)WC_TRASH_ROW);
            bc[lane] = src * 128u;
            // group g fetches entries g, 8 + g, 16 + g ...: 8 consecutive words / bytes of the transposed list
            const nrt_i4 ra = *(wc_lds_vi4 *)(wl + WC_LISTR_OFF + g * 32), rb = *(wc_lds_vi4 *)(wl + WC_LISTR_OFF + g * 32 + 16);
            const unsigned long long dd = *(wc_lds_vu64 *)(wl + WC_LISTD_OFF + g * 8);
            const nrt_i4 sa = *(wc_lds_vi4 *)(wl + WC_BC_OFF + g * 32), sb = *(wc_lds_vi4 *)(wl + WC_BC_OFF + g * 32 + 16);
            s.fd_lo = (unsigned)dd; s.fd_hi = (unsigned)(dd >> 32);
#pragma unroll
            for (int c = 0; c < 4; ++c) { s.sl[c] = (unsigned)sa[c]; s.sl[4 + c] = (unsigned)sb[c]; }
            // The first WC_FIXED_LOADS row loads are issued whatever n is (entries past n address bytes past the volume: no memory
            // access): the compiler's vmcnt bookkeeping takes the path with the FEWEST loads as what may be in flight, and with every
            // load under a condition that path has none -- it then waited for all but the last two operations at every use, i.e. for
            // the other state's pass as well (v3: 1.37 ms).
#pragma unroll
            for (int i = 0; i < WC_FIXED_LOADS; ++i) s.F[i] = fetch_row((unsigned)(i < 4 ? ra[i & 3] : rb[i & 3]));
#pragma unroll
            for (int i = WC_FIXED_LOADS; i < 8; ++i) {
                if (8 * i < s.n) s.F[i] = fetch_row((unsigned)(i < 4 ? ra[i & 3] : rb[i & 3]));      // wave-uniform condition
            }
        }
        if (DICE) s.T = __builtin_nontemporal_load((const nrt_f4 *)(fixb + (size_t)((unsigned)xq * qstep) * 128u + row_lane));
    };

    // rows fetched for the pass -> their cache / overflow rows
    auto deliver = [&](Pass &s) {
#if !NRT_FUSED_WCBLENDF
        if (__builtin_expect(s.n < 0, 0)) {                        // register path: the voxel's 8 corner rows go to rows 8 g .. 8 g + 7
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                *(wc_lds_f4 *)(lrow + (g * 8 + corner) * 128) = s.F[corner];
                s.sl[corner] = (unsigned)(g * 8 + corner) * 128u;
            }
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (8 * i < s.n) *(wc_lds_f4 *)(lrow + wc_byte(s.fd_lo, s.fd_hi, i) * 128u) = s.F[i];     // wave-uniform condition
        }
    };

    auto blend = [&](int pass, Pass &s, bool live, const nrt_f4 (&R)[8]) {
        const float w1x = nrt_sub(1.0f, s.w0x), w1y = nrt_sub(1.0f, s.w0y), w1z = nrt_sub(1.0f, s.w0z);      // corner_1d's w1
        const nrt_f2 wy2 = {s.w0y, w1y}, wz2 = {s.w0z, w1z};
        const nrt_f2 wxy0 = (nrt_f2){s.w0x, s.w0x} * wy2, wxy1 = (nrt_f2){w1x, w1x} * wy2;
        nrt_f2 wt2[4];
        wt2[0] = (nrt_f2){wxy0[0], wxy0[0]} * wz2;
        wt2[1] = (nrt_f2){wxy0[1], wxy0[1]} * wz2;
        wt2[2] = (nrt_f2){wxy1[0], wxy1[0]} * wz2;
        wt2[3] = (nrt_f2){wxy1[1], wxy1[1]} * wz2;
        nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float wt = wt2[corner >> 1][corner & 1];
            const nrt_f2 w2 = {wt, wt};
            al = al + w2 * (nrt_f2){R[corner][0], R[corner][1]};
            ah = ah + w2 * (nrt_f2){R[corner][2], R[corner][3]};
        }
        nrt_f4 acc = {al[0], al[1], ah[0], ah[1]};
        if (FILL) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = apply_fill(acc[c], s.oob, a.fill_f);
        }
        if (yzvalid && live) {
            if (STORE) __builtin_nontemporal_store(acc, (nrt_f4 *)(outb + (size_t)((unsigned)(x0 + pass) * qstep) * 128u + row_lane));
            const nrt_f2 pl = {acc[0], acc[1]}, ph = {acc[2], acc[3]}, tl = {s.T[0], s.T[1]}, th = {s.T[2], s.T[3]};
            if (DICE) {
                stp_l = stp_l + tl * pl; stp_h = stp_h + th * ph;
                stt_l = stt_l + tl * tl; stt_h = stt_h + th * th;
                spp_l = spp_l + pl * pl; spp_h = spp_h + ph * ph;
            }
            if (DICE && MM) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    mnt = fminf(mnt, s.T[c]); mxt = fmaxf(mxt, s.T[c]);
                    mnp = fminf(mnp, acc[c]); mxp = fmaxf(mxp, acc[c]);
                }
            }
        }
    };
    // one pass: store its rows, blend it, then re-use its state and row registers for the pass two ahead.  The loop body is the
    // same straight sequence for every pair of passes (past the end of the march the last pass is managed and blended again with its
    // sums masked): with a conditional half-step the compiler has to assume that the other state's loads may not have been issued and
    // waits for everything in flight.
    const int last = npass - 1;
    auto step = [&](int pass, Pass &s) {
#if NRT_FUSED_WCBLENDF
        if (__builtin_expect(s.n < 0, 0)) {
            blend(pass, s, pass <= last, s.F);                     // register path: F holds the voxel's 8 corner rows
        } else
#endif
        {
            deliver(s);
            nrt_f4 R[8];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) R[corner] = *(const wc_lds_f4 *)(lrow + s.sl[corner]);
            __builtin_amdgcn_sched_barrier(0);                     // all eight reads in flight before the first product: one LDS round trip
            blend(pass, s, pass <= last, R);
        }
        manage(min(pass + 2, last), s);
        fetch_loc(min(pass + 4, last), s);
    };

    if (npass > 0) {
        fetch_loc(0, A);
        fetch_loc(min(1, last), B);
        manage(0, A);
        fetch_loc(min(2, last), A);
        manage(min(1, last), B);
        fetch_loc(min(3, last), B);
        for (int pass = 0; pass < npass; pass += 2) {
#if NRT_WC_SYNC > 0
            if ((pass & (NRT_WC_SYNC - 1)) == 0) __builtin_amdgcn_s_barrier();
#endif
            step(pass, A);
            step(pass + 1, B);
        }
    }

    if (!DICE) return;
    nrt_f4 stp = {stp_l[0], stp_l[1], stp_h[0], stp_h[1]}, stt = {stt_l[0], stt_l[1], stt_h[0], stt_h[1]},
           spp = {spp_l[0], spp_l[1], spp_h[0], spp_h[1]};
    // ---- block reduction (identical tree to warp_dice_tile / dice_soft_vec) ---------------------------------------------------
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
    __syncthreads();                                              // every wave is done with its cache: the rows double as scratch
    float (*red)[3 * L + 4] = (float (*)[3 * L + 4])wc_smem;
    if (lane < G) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[wave][0 * L + 4 * lane + c] = stp[c];
            red[wave][1 * L + 4 * lane + c] = stt[c];
            red[wave][2 * L + 4 * lane + c] = spp[c];
        }
    }
    if (lane == 0) { red[wave][3 * L + 0] = mnt; red[wave][3 * L + 1] = mxt; red[wave][3 * L + 2] = mnp; red[wave][3 * L + 3] = mxp; }
    __syncthreads();
    const long long pbase = (long long)b * (tg.ncol * tg.nseg) + prow;
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
    xmarch_zero_rows(tg, xw, 3 * L, fpart, mpart);
}

// The work counters of a persistent launch are zeroed by a KERNEL on the launch stream (nrt_zero_async), not by hipMemsetAsync: inside
// a captured hipGraph the memset node did not take effect between replays (ROCm 7.2) -- every replay after the first found the lists
// exhausted, its blocks left at once and the second stage re-reduced the previous replay's partial sums: stale results in 0.03 ms
// (tools/graph_fused_probe.py; tests/test_gpu_dice_cce.py::test_fused_kernels_recompute_under_graph_replay).

// PERSIST: 2 blocks per CU stay resident and take items from per-XCD lists (atomic counters in `queue`, zeroed before the launch):
// first their own XCD's (its L2 holds the neighbouring columns), then the others'.  With one block per item the XCDs finish up to
// 5 % apart (tools/block_trace.py: last block of an XCD at 1044 .. 1097 us) and the slots of a finished XCD idle; the items are the
// same either way, every item writes its own partial row, so the sums do not depend on who computed what.
template <int MODE, bool STORE, bool MM, bool FILL, bool DICE = true, bool PERSIST = false>
__global__ __launch_bounds__(256, 2) void warp_dice_wc(InterpArgs a, TileGeom tg, const float *__restrict__ fixed,
                                                        float *__restrict__ fpart, float *__restrict__ mpart, unsigned *__restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char wc_smem[];
    if (!PERSIST) {
        XmWork xw;
        if (!xmarch_work(tg, a.O[0], xw)) return;
        wc_item<MODE, STORE, MM, FILL, DICE>(a, tg, fixed, fpart, mpart, xw, wc_smem);
        return;
    }
    __shared__ unsigned s_next[2];
    const unsigned home = blockIdx.x % NRT_NXCD;
    for (;;) {
        __syncthreads();                                          // the previous item's LDS (cache rows, reduction scratch) is done with
        if (threadIdx.x == 0) {
            unsigned kk = home, jb = 0xffffffffu;
            for (unsigned t = 0; t < NRT_NXCD; ++t) {
                const unsigned q = (home + t) % NRT_NXCD;
                const unsigned v = atomicAdd(&queue[q * 16u], 1u);                 // (the counters sit 64 bytes apart)
                if (v < tg.items_x) { kk = q; jb = v; break; }
            }
            s_next[0] = kk; s_next[1] = jb;
        }
        __syncthreads();
        const unsigned kk = s_next[0], jb = s_next[1];
        if (jb == 0xffffffffu) break;
        XmWork xw;
        if (!xmarch_work_at(tg, a.O[0], kk, jb, xw)) continue;
        wc_item<MODE, STORE, MM, FILL, DICE>(a, tg, fixed, fpart, mpart, xw, wc_smem);
    }
}

#ifndef NRT_FUSED_WCPERSIST
#define NRT_FUSED_WCPERSIST 1
#endif
template <int MODE, bool STORE, bool MM, bool FILL>
int launch_wc_inst(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, const float *fixed, float *fpart, float *mpart,
                   unsigned *queue, hipStream_t st) {
    const unsigned items = NRT_NXCD * tg.items_x, slots = 2u * (unsigned)nrt_num_cus();
    if (NRT_FUSED_WCPERSIST && queue && items > slots) {
        if (hipFuncSetAttribute((const void *)warp_dice_wc<MODE, STORE, MM, FILL, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WC_BLOCK_BYTES) != hipSuccess)
            return NRT_ERR_LAUNCH;
        if (nrt_zero_async(queue, NRT_NXCD * 64, st) != hipSuccess) return NRT_ERR_LAUNCH;
        hipLaunchKernelGGL((warp_dice_wc<MODE, STORE, MM, FILL, true, true>), dim3(nrt_xcd_grid(slots)), dim3(256), WC_BLOCK_BYTES, st, a, tg, fixed,
                           fpart, mpart, queue);
        return NRT_OK;
    }
    if (hipFuncSetAttribute((const void *)warp_dice_wc<MODE, STORE, MM, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WC_BLOCK_BYTES) != hipSuccess)
        return NRT_ERR_LAUNCH;
    hipLaunchKernelGGL((warp_dice_wc<MODE, STORE, MM, FILL>), dim3(items), dim3(256), WC_BLOCK_BYTES, st, a, tg, fixed, fpart, mpart,
                       (unsigned *)nullptr);
    (void)nblocks; (void)batch;
    return NRT_OK;
}

template <int MODE, bool FILL>
int launch_wc_fill(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, bool store, bool minmax, const float *fixed,
                   float *fpart, float *mpart, unsigned *queue, hipStream_t st) {
    if (store) return minmax ? launch_wc_inst<MODE, true, true, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, queue, st)
                             : launch_wc_inst<MODE, true, false, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, queue, st);
    return minmax ? launch_wc_inst<MODE, false, true, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, queue, st)
                  : launch_wc_inst<MODE, false, false, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, queue, st);
}

template <int MODE>
int launch_wc_mode(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, bool store, bool minmax, const float *fixed,
                   float *fpart, float *mpart, unsigned *queue, hipStream_t st) {
    return a.has_fill ? launch_wc_fill<MODE, true>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, queue, st)
                      : launch_wc_fill<MODE, false>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, queue, st);
}

// the wave-cache form applies to: 32 float32 labels, x-march geometry with 4 x 8 patches
// (and volumes below 0xffffff00 bytes per batch entry: the masked list entries address the bytes past that)
inline bool wc_applies(const TileGeom &tg, int G, const InterpArgs &a) {
    return tg.x_march && G == 8 && tg.lty == 2 && tg.ltz == 3 && !tg.plane_major &&
           (unsigned long long)a.S[0] * a.S[1] * a.S[2] * 128ull < 0xffffff00ull;
}

// minmax: the caller wants the value range of both maps (check_input_limits); without it the kernel does not track it (the partial
// rows then carry +-inf, which nothing reads)
inline int launch_wc(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, int mode, bool store, bool minmax,
                     const float *fixed, float *fpart, float *mpart, unsigned *queue, hipStream_t st) {
    switch (mode) {
        case NRT_LOC_ABSOLUTE: return launch_wc_mode<NRT_LOC_ABSOLUTE>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, queue, st);
        case NRT_LOC_SHIFT: return launch_wc_mode<NRT_LOC_SHIFT>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, queue, st);
        default: return launch_wc_mode<NRT_LOC_LINSPACE>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, queue, st);
    }
}

// the warp alone through the same kernel (interpn.hip, variant 10): mixed block lengths, persistent blocks.  The work counters come
// from a per-device ring of 64 sets (a launch takes the next one: launches that overlap on different streams do not share a set)
inline unsigned *wc_queue_slot() {
    static unsigned *ring[64];
    static unsigned next[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ring[dev] && hipMalloc((void **)&ring[dev], 64 * NRT_NXCD * 64) != hipSuccess) { ring[dev] = nullptr; return nullptr; }
    return ring[dev] + (size_t)(__atomic_fetch_add(&next[dev], 1u, __ATOMIC_RELAXED) % 64u) * (NRT_NXCD * 16);
}

template <int MODE, bool FILL>
int launch_wc_interpn_inst(const InterpArgs &a, const TileGeom &tg, hipStream_t st) {
    const unsigned items = NRT_NXCD * tg.items_x, slots = 2u * (unsigned)nrt_num_cus();
    unsigned *queue = (NRT_FUSED_WCPERSIST && items > slots) ? wc_queue_slot() : nullptr;
    if (queue) {
        if (hipFuncSetAttribute((const void *)warp_dice_wc<MODE, true, false, FILL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WC_BLOCK_BYTES) != hipSuccess)
            return NRT_ERR_LAUNCH;
        if (nrt_zero_async(queue, NRT_NXCD * 64, st) != hipSuccess) return NRT_ERR_LAUNCH;
        hipLaunchKernelGGL((warp_dice_wc<MODE, true, false, FILL, false, true>), dim3(nrt_xcd_grid(slots)), dim3(256), WC_BLOCK_BYTES, st, a, tg,
                           (const float *)nullptr, (float *)nullptr, (float *)nullptr, queue);
        return NRT_OK;
    }
    if (hipFuncSetAttribute((const void *)warp_dice_wc<MODE, true, false, FILL, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)WC_BLOCK_BYTES) != hipSuccess)
        return NRT_ERR_LAUNCH;
    hipLaunchKernelGGL((warp_dice_wc<MODE, true, false, FILL, false>), dim3(items), dim3(256), WC_BLOCK_BYTES, st, a, tg, (const float *)nullptr,
                       (float *)nullptr, (float *)nullptr, (unsigned *)nullptr);
    return NRT_OK;
}

}  // namespace

bool nrt_wc_interpn_supported(const void *args, int batch) {
    const InterpArgs &a = *(const InterpArgs *)args;
    if (a.C != 32 || !xmarch_applies(a.O, batch)) return false;
    if ((unsigned long long)a.nout * 128ull >= (1ull << 32)) return false;                 // 32-bit output offsets
    if ((long long)a.S[0] * a.S[1] >= (1 << 24) || a.S[2] >= (1 << 24) || (long long)a.O[0] * a.O[1] >= (1 << 24) || a.O[2] >= (1 << 24)) return false;
    TileGeom tg;
    unsigned nblocks, grid;
    const int t = xmarch_default_tune();
    tile_geometry(a.O, 8, t, t, tg, nblocks);
    xmarch_setup_mixed(a.O, batch, t, tg, grid);
    return wc_applies(tg, 8, a);
}

int nrt_wc_interpn_launch(const void *args, int batch, int mode, void *stream) {
    const InterpArgs &a = *(const InterpArgs *)args;
    TileGeom tg;
    unsigned nblocks, grid;
    const int t = xmarch_default_tune();
    tile_geometry(a.O, 8, t, t, tg, nblocks);
    xmarch_setup_mixed(a.O, batch, t, tg, grid);
    hipStream_t st = nrt_stream(stream);
    int rc;
    switch (mode) {
        case NRT_LOC_ABSOLUTE: rc = a.has_fill ? launch_wc_interpn_inst<NRT_LOC_ABSOLUTE, true>(a, tg, st) : launch_wc_interpn_inst<NRT_LOC_ABSOLUTE, false>(a, tg, st); break;
        case NRT_LOC_SHIFT: rc = a.has_fill ? launch_wc_interpn_inst<NRT_LOC_SHIFT, true>(a, tg, st) : launch_wc_interpn_inst<NRT_LOC_SHIFT, false>(a, tg, st); break;
        default: rc = a.has_fill ? launch_wc_interpn_inst<NRT_LOC_LINSPACE, true>(a, tg, st) : launch_wc_interpn_inst<NRT_LOC_LINSPACE, false>(a, tg, st); break;
    }
    if (rc != NRT_OK) return rc;
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
