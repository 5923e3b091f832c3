// Fused SpatialTransformer + soft Dice for 32 float32 labels with a WAVE-PRIVATE LDS ROW CACHE (gfx950).
//
// Why (profiles/r04_lab/l1_access_curve.jsonl): the register kernel (warp_dice_tile) requests all 8 corner rows
// of every voxel from L1.  Dropping corner requests from 8 to 4 / 2 / 1 per voxel with everything else unchanged takes the
// launch from 1.085 ms to 0.843 / 0.788 / 0.755: the cost is the NUMBER OF ROW REQUESTS that pass the texture-address / L1
// pipe (hit or miss), not HBM.  So: fetch every DISTINCT row once per wave and pass, keep it across passes, blend from LDS.
//
// Structure.  The x-march schedule of warp_dice_tile is kept (a block of four waves owns a 4 x 8 (y,z) patch and walks x, one
// x-plane per pass; blocks dealt to the XCDs region by region).  A wave owns a 2 x 4 sub-patch: 8 voxels per pass = one voxel
// per 8-lane group, lane p of a group holds labels 4p..4p+3 (16 bytes of every row).  Each wave has its own direct-mapped cache
// of 128 rows (16 KB) + 16 overflow rows in LDS; NOTHING is shared between waves, so there is no barrier, no claim protocol and
// no atomic anywhere: a wave's LDS operations execute in program order.  The work of a pass:
//   M1a  every lane computes the corner geometry of its group's voxel and takes ONE of the 64 (voxel, corner) references of the
//        pass: row id `rid`, slot = (ix & 3, iy & 3, iz & 7); reads tag[slot];
//   M1b  references that miss write {rid, lane}; everybody reads the tag back;
//   M1c  tag.rid == rid: the row is (or will be) in the slot -- `served`; the missing lane whose own tag came back is its `loader`;
//        a reference whose slot now names another row of this same pass is an `orphan` (about 2 % of the references on the bench
//        field): it gets an overflow row for this pass only.  Loaders + orphans are compacted (ballot / mbcnt) into a fetch list
//        in LDS (two words per entry: byte offset of the row in the volume, of its LDS row), transposed so that 8-lane groups read their entries back
//        with two ds_read_b128 and fetch one 128-byte row each: ceil(n / 8) load instructions per pass instead of 8 (n = 23 on
//        the bench field); the source row of every reference is broadcast to its group through 128 bytes
//        of LDS (16-bit offsets, one ds_read_b128, unpacked by the SDWA operands of the address additions);
//   M2   the row loads (four always -- entries past n address bytes past the volume: no memory access -- and four more under a
//        wave-uniform branch when n > 32: 19 % of the passes), the fixed row;
//   D    the fetched rows are stored to their slots;
//   B    blend: 8 ds_read_b128 per lane from the slots recorded in M1c, then exactly the arithmetic of warp_dice_tile /
//        interpn.hip (same op order: bit-identical warped rows).
// A pass with more than 16 orphans (incoherent fields) invalidates the cache and fetches its 64 references as they are.
//
// Round 5: the SCHEDULE.  Counters of the round-4 form (profiles/r04_lab/pmc_wc_vs_reg.json): 581 wave quad-cycles per pass of which
// 218 wait on counters and 60 on issue, VALU 62 % busy at 2 waves per SIMD, 20 branch instructions per pass -- the management chain
// (tag read -> write -> read -> ballots -> list write -> list read -> loads) was four serial LDS round trips per pass, cut into ~20
// basic blocks the compiler cannot schedule across.  Now a step is (nearly) one basic block and the chain of pass p + 2 is threaded
// through the blend of pass p:  M2(p + 1) | D(p), B-reads(p) | M1a(p + 2) | blend corners 0..3 | M1b | corners 4..7 | M1c | Dice sums.
// Every LDS round trip of the management has a stretch of the blend's arithmetic to hide behind; the row loads of pass p + 1 go out
// first thing in step p, BEFORE the wait for pass p's own rows (a late row does not hold back the next requests), one step before they
// are stored -- the lead matters: requested half a step later the launch takes 12 % longer (profiles/r05_lab/wc_probes.jsonl).  The fixed row, the locations and the stored row go through buffer descriptors with the
// pass offset in an SGPR (no per-pass address arithmetic); tags are 32-bit (row id << 6 | lane).
//
// Model of the request stream (tools/wc_sim.py, bench field): 4.8 distinct rows per voxel inside a pass, 2.7 fetched per voxel with
// the cache (8 through L1 in the register kernel), 0.16 orphan references per voxel.

#pragma once

#include <type_traits>

#include "dice_reduce.h"
#include "interpn_core.h"

namespace {

#ifndef WC_NST_WARP
#define WC_NST_WARP 3                                           // pass states of the stand-alone warp (2: requests one step ahead)
#endif
#ifndef WC_NST_FUSED
#define WC_NST_FUSED 2                                          // pass states of the fused warp + Dice (forward)
#endif
#ifndef WC_NST_BWD
#define WC_NST_BWD 3                                            // pass states of the backward forms (they store the location gradient every pass)
#endif
constexpr int WC_SLOTS = 128;                                   // direct-mapped rows per wave
constexpr int WC_OVF = 16;                                      // overflow rows (orphans of the current pass)
constexpr int WC_TRASH_ROW = WC_SLOTS + WC_OVF;                 // where the list entries without a row store (branch-free masking)
constexpr int WC_ROWS = WC_TRASH_ROW + 1;
constexpr unsigned WC_NOROW = 0x01ffffffu;                      // row id of "no row": its bytes lie past every volume this kernel takes (wc_applies)
constexpr unsigned WC_TAGS_OFF = WC_ROWS * 128;                 // u32 tag per slot (row id << 6 | lane), + one trash entry
constexpr unsigned WC_LIST_OFF = WC_TAGS_OFF + 528;             // fetch list: 64 x {row offset in the volume, LDS row offset}, transposed
                                                                // (entry e at (e & 7) * 8 + (e >> 3))
constexpr unsigned WC_BC_OFF = WC_LIST_OFF + 512;               // u16 byte offset of the source row of every (voxel, corner) reference
constexpr unsigned WC_WAVE_BYTES = WC_BC_OFF + 128;             // 19728
constexpr unsigned WC_BLOCK_BYTES = 4 * WC_WAVE_BYTES;          // 78912: two blocks per CU
static_assert(WC_LIST_OFF % 16 == 0 && WC_BC_OFF % 16 == 0 && WC_WAVE_BYTES % 16 == 0, "LDS layout");
static_assert((WC_SLOTS + 1) * 4 <= 528 && WC_ROWS <= 256, "LDS layout");
static_assert(2 * WC_BLOCK_BYTES <= 160 * 1024, "two blocks per CU");

// LDS pointers carry their address space explicitly: a volatile access through a generic pointer compiles to flat_load / flat_store
// with a full vmcnt(0) wait around it
typedef __attribute__((address_space(3))) char wc_lds_char;
typedef __attribute__((address_space(3))) nrt_f4 wc_lds_f4;
typedef __attribute__((address_space(3))) volatile nrt_i4 wc_lds_vi4;
typedef __attribute__((address_space(3))) volatile unsigned wc_lds_vu32;
typedef unsigned wc_u3 __attribute__((ext_vector_type(3)));
typedef unsigned wc_u4 __attribute__((ext_vector_type(4)));
typedef float wc_f3 __attribute__((ext_vector_type(3)));

// utils.py:139-153 for one dimension with three operations less than corner_1d and the same bits: l0 = floor(clip(p)) is
// clip(floor(p)) for every float (the bounds are integers; NaN clips to 0 either way), and l0 + 1 >= 1 needs no lower clip
__device__ __forceinline__ void wc_corner(float pv, float mx, int &i0, int &i1, float &w0) {
    const float cl = fminf(fmaxf(pv, 0.0f), mx);
    const float l0 = floorf(cl);
    const float l1 = fminf(nrt_add(l0, 1.0f), mx);
    i0 = (int)l0; i1 = (int)l1;
    w0 = nrt_sub(l1, cl);
}

// a buffer descriptor has to be wave-uniform FOR THE COMPILER too (else every access becomes a waterfall loop): the base is rebuilt
// from readfirstlane'd halves (readfirstlane returns int: widen through unsigned).  Offsets at or past `bytes` read zeros without
// touching memory.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wc_rsrc(const void *base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    const void *u = (const void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
                                   (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// what the backward form needs besides the forward's arguments
struct WcBwd {
    const float *sums, *gdice;   // BWD == 1: the forward's [B, 3, L] sums and d loss / d dice [B, L]
    float eps;
    float *gloc;                 // [B, nout, 3]
};

// one work item (a column of a 4 x 8 patch, or a piece of one) by one block.
// DICE = false: the warp alone (nrt_interpn_f32 variant 10): no fixed map, no sums, no partials -- the same gather and blend.
// BWD (round 5): d loss / d loc on the same gather (what TF autodiff yields for utils.py:137-191 -- `floor` no gradient, `clip` on the
// closed range -- composed with the soft Dice of metrics.py:476-482).  1: `fixed` is the fixed map and the gradient of the Dice wrt the
// warped row is formed from the forward's sums; 2: `fixed` IS the incoming gradient row (the plain warp).  Same arithmetic, same order
// as warp_dice_bwd_xm (backward.hip), which it replaces at 32 float32 channels.
template <int MODE, bool STORE, bool MM, bool FILL, bool DICE, int BWD = 0>
__device__ __forceinline__ void wc_item(const InterpArgs &a, const TileGeom &tg, const float *__restrict__ fixed, float *__restrict__ fpart,
                                        float *__restrict__ mpart, const XmWork &xw, char *wc_smem, const WcBwd &bw = WcBwd{nullptr, nullptr, 0.0f, nullptr}) {
    constexpr int G = 8, L = 32;
    static_assert(BWD == 0 || (DICE && !STORE && !MM), "the backward form reads a second row per voxel and stores the location gradient only");
    // the work item came through LDS (persistent blocks): tell the compiler that it is wave-uniform, the pass offsets below are SGPRs
    const int b = __builtin_amdgcn_readfirstlane(xw.b);
    const unsigned prow = (unsigned)__builtin_amdgcn_readfirstlane((int)xw.prow), ucol = (unsigned)__builtin_amdgcn_readfirstlane((int)xw.ucol);
    const unsigned RY = 1u << tg.lry, RZ = 1u << tg.lrz;
    const unsigned nRz = (tg.nTz + RZ - 1) / RZ;
    const unsigned reg = ucol / (RY * RZ), wr = ucol % (RY * RZ);
    const unsigned cy = (reg / nRz) * RY + wr / RZ, cz = (reg % nRz) * RZ + wr % RZ;
    const int x0 = __builtin_amdgcn_readfirstlane(xw.x0), y0 = (int)cy << 2, z0 = (int)cz << 3;
    int npass = __builtin_amdgcn_readfirstlane(xw.xlen);         // one x-plane of the patch per pass
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

    // wave-uniform descriptors; a pass contributes an SGPR offset, the lane a constant 32-bit one
    const char *volb = (const char *)a.vol + (long long)b * a.vol_bs * 4ll;
    const unsigned long long volbytes = (unsigned long long)a.S[0] * SY * SZ * 128ull;       // < 0xffffff00 (wc_applies)
    const unsigned outbytes = a.nout * 128u;                                                 // < 2^32 (checked by the C entries)
    const __amdgpu_buffer_rsrc_t vres = wc_rsrc(volb, (unsigned)volbytes);
    const __amdgpu_buffer_rsrc_t lres = wc_rsrc(MODE != NRT_LOC_LINSPACE ? (const char *)(a.loc + (long long)b * a.loc_bs) : volb, a.nout * 12u);
    const __amdgpu_buffer_rsrc_t fres = wc_rsrc(DICE ? (const char *)fixed + (long long)b * a.out_bs * 4ll : volb, outbytes);
    const __amdgpu_buffer_rsrc_t ores = wc_rsrc(STORE ? (const char *)a.out + (long long)b * a.out_bs * 4ll : volb, outbytes);
    const __amdgpu_buffer_rsrc_t gres = wc_rsrc(BWD ? (const char *)(bw.gloc + (long long)b * a.nout * 3) : volb, a.nout * 12u);
    const unsigned gl_lane = (p == 0 && yzvalid) ? qyz * 12u : 0xffffffffu;     // one lane of a voxel's eight stores its three components
    // d dice / d warped = ca t + cb p for this lane's four labels (metrics.py:476-482 differentiated; divide_no_nan: zero)
    float ca[4] = {0.0f, 0.0f, 0.0f, 0.0f}, cb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (BWD == 1) {
        const float *sm = bw.sums + (long long)b * 3 * L;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = p * 4 + k;
            const float num = 2.0f * sm[l] + bw.eps, den = sm[L + l] + sm[2 * L + l] + bw.eps;
            const float gd = bw.gdice[(long long)b * L + l];
            if (den != 0.0f) { ca[k] = 2.0f * gd / den; cb[k] = -2.0f * gd * num / (den * den); }
        }
    }
    const unsigned loc_lane = qyz * 12u, row_lane = (qyz * 8u + (unsigned)p) * 16u;
    const unsigned out_lane = yzvalid ? row_lane : 0xffffffffu;   // a store past the descriptor's size is dropped: edge patches need no branch
    const unsigned loc_step = qstep * 12u, row_step = qstep * 128u;
    const unsigned p16 = (unsigned)p * 16u;

    wc_lds_char *wl = (wc_lds_char *)wc_smem + wave * WC_WAVE_BYTES;
    wc_lds_vu32 *tags = (wc_lds_vu32 *)(wl + WC_TAGS_OFF);
    wc_lds_char *lrow = wl + p * 16;                              // this lane's piece of row 0
    tags[lane] = ~0u;                                            // no row id is 0x3ffffff
    tags[lane + 64] = ~0u;
    const bool cx1 = (p & 4) != 0, cy1 = (p & 2) != 0, cz1 = (p & 1) != 0;

    nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
    float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;

    // ---- per-pass state.  Two passes are in flight: the loop is unrolled by two and the passes alternate between the states A and B
    struct Pass {
        float w0x, w0y, w0z;         // lower-corner weights
        unsigned sl[8];              // byte offset (row * 128) of the source row of the 8 corners in this wave's LDS rows
        unsigned lr[4], ld[4];       // fetch-list entries g, 8 + g, 16 + g, 24 + g: byte offset of the row in the volume, of its LDS row
        unsigned xr[4], xd[4];       // entries 32 + g ... (n > 32 only)
        int n;                       // wave-uniform: entries of the fetch list
        int xq;                      // wave-uniform: output x-plane of the pass
        float mk[3];                 // BWD: 1 where the location lies inside the closed range of the axis, else 0 (the slope of `clip`)
        bool oob;
        nrt_f4 T;                    // the fixed row
        nrt_f4 F[4], Fx[4];          // rows in flight
        float pn[3];                 // location of the pass that will use this state next
    };
    // a pass under management: from the tag read (m1a) to the fetch list (m1c)
    struct Mg { float w0x, w0y, w0z; bool oob, miss; unsigned rid, slot, t1, t2, mytag; float mk[3]; };
    // NST states: a pass is managed NST steps before it is blended and its rows are requested NST - 1 steps before.  Memory operations of a
    // wave complete IN ORDER, stores included: a row requested one step ahead waits for the acknowledgement of the store of the pass in front
    // of it; requested two steps ahead it does not.  Measured, alternating libraries on one box (profiles/r06_lab/e6_three_states_standalone.jsonl,
    // e7_three_states_fused.jsonl): the stand-alone warp (a 128-byte store per voxel) 6 - 8 % faster with three states at every batch size, the
    // backward forms (a 12-byte store per voxel) 4 - 5 %, the fused forward (no store in its stream) within 1 % -- it keeps two (its third state
    // only fits at 251 - 255 registers).
    constexpr int NST = ((BWD ? WC_NST_BWD : (DICE ? WC_NST_FUSED : WC_NST_WARP)) > 2) ? 3 : 2;
    // XSH: the rows 32 .. 63 of a long fetch list (19 % of the passes) travel through ONE shared set of registers, requested at the end of the
    // step before their use, instead of a set per state requested with the others -- what lets a third state fit next to the Dice / gradient
    // registers (backward forms: 217 - 243 registers; the stand-alone warp keeps a set per state: 198 - 221)
    constexpr bool XSH = NST == 3 && DICE;
    Pass A, B, C;
    nrt_f4 FxS[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto reset = [&](Pass &s) {
        s.w0x = s.w0y = s.w0z = 0.f; s.n = 0; s.xq = x0; s.oob = false; s.mk[0] = s.mk[1] = s.mk[2] = 0.f;
        s.T = (nrt_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) { s.F[i] = s.Fx[i] = (nrt_f4){0.f, 0.f, 0.f, 0.f}; s.lr[i] = s.xr[i] = WC_NOROW << 7; s.ld[i] = s.xd[i] = WC_TRASH_ROW * 128u; }
#pragma unroll
        for (int i = 0; i < 8; ++i) s.sl[i] = 0;
        s.pn[0] = s.pn[1] = s.pn[2] = 0.f;
    };
    reset(A); reset(B);
    if (NST == 3) reset(C);

    auto fetch_loc = [&](int pass, Pass &s) {
        if (MODE != NRT_LOC_LINSPACE) {
            const wc_u3 v = __builtin_bit_cast(wc_u3, __builtin_amdgcn_raw_buffer_load_b96(lres, loc_lane, (unsigned)(x0 + pass) * loc_step, 0));
            s.pn[0] = __uint_as_float(v[0]); s.pn[1] = __uint_as_float(v[1]); s.pn[2] = __uint_as_float(v[2]);
        }
    };
    auto fetch_row = [&](unsigned rowoff) -> nrt_f4 {
        return __builtin_bit_cast(nrt_f4, __builtin_amdgcn_raw_buffer_load_b128(vres, rowoff | p16, 0, 0));
    };

    // M1a: geometry of pass `pass` (its location is in s.pn), this lane's reference, first tag read
    auto m1a = [&](int pass, const Pass &s, Mg &m) {
        const int xq = x0 + pass;
        float px, py, pz;
        if (MODE == NRT_LOC_ABSOLUTE) { px = s.pn[0]; py = s.pn[1]; pz = s.pn[2]; }
        else if (MODE == NRT_LOC_SHIFT) { px = nrt_add((float)xq, s.pn[0]); py = nrt_add(fyc, s.pn[1]); pz = nrt_add(fzc, s.pn[2]); }
        else { px = (xq == 0) ? 0.0f : ((xq == a.O[0] - 1) ? mxx : nrt_mul(a.delta[0], (float)xq)); py = ly; pz = lz; }
        int i0x, i1x, i0y, i1y, i0z, i1z;
        wc_corner(px, mxx, i0x, i1x, m.w0x);
        wc_corner(py, mxy, i0y, i1y, m.w0y);
        wc_corner(pz, mxz, i0z, i1z, m.w0z);
        m.oob = FILL ? ((px < 0.0f) || (px > mxx) || (py < 0.0f) || (py > mxy) || (pz < 0.0f) || (pz > mxz)) : false;
        if (BWD) {               // inside the closed range <=> clipping leaves the value alone (NaN: outside, as the comparisons have it)
            m.mk[0] = (fminf(fmaxf(px, 0.0f), mxx) == px) ? 1.0f : 0.0f;
            m.mk[1] = (fminf(fmaxf(py, 0.0f), mxy) == py) ? 1.0f : 0.0f;
            m.mk[2] = (fminf(fmaxf(pz, 0.0f), mxz) == pz) ? 1.0f : 0.0f;
        }
        const unsigned ix = cx1 ? i1x : i0x, iy = cy1 ? i1y : i0y, iz = cz1 ? i1z : i0z;      // corner p of the group's voxel
        m.rid = nrt_mad24(nrt_mad24(ix, SY, iy), SZ, iz);
        m.slot = ((ix & 3u) << 5) | ((iy & 3u) << 3) | (iz & 7u);
        m.mytag = (m.rid << 6) | (unsigned)lane;
        m.t1 = tags[m.slot];
    };
    // M1b: references that miss claim their slot (hits store to the trash entry), everybody reads the tag back
    auto m1b = [&](Mg &m) {
        m.miss = (m.t1 ^ m.mytag) >= 64u;
        tags[m.miss ? m.slot : (unsigned)WC_SLOTS] = m.mytag;
        m.t2 = tags[m.slot];
    };
    // M1c: roles, fetch list, source rows of the group's 8 corners; leaves the pass in s (its row loads are issue(s))
    auto m1c = [&](int pass, const Mg &m, Pass &s) {
        const unsigned x2 = m.t2 ^ m.mytag;
        const bool served = x2 < 64u;
        const bool loader = m.miss && x2 == 0u;
        const bool orphan = !served;
        const unsigned long long Ml = __builtin_amdgcn_ballot_w64(loader), Mo = __builtin_amdgcn_ballot_w64(orphan);
        const unsigned long long Mf = Ml | Mo;
        const int no = __builtin_popcountll(Mo);
        int n = __builtin_popcountll(Mf);
        const unsigned ro = __builtin_amdgcn_mbcnt_hi((unsigned)(Mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Mo, 0u));
        const unsigned rf = __builtin_amdgcn_mbcnt_hi((unsigned)(Mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Mf, 0u));
        bool fetcher = loader || orphan;
        unsigned src = orphan ? (unsigned)WC_SLOTS + ro : m.slot;
        // EVERY lane writes one entry of the fetch list: loaders and orphans their row and its destination at positions 0 .. n - 1,
        // the others a row past the volume (reads as zeros, touches no memory) with the trash row as destination at n .. 63 -- so
        // fetching and delivering need no per-lane condition
        unsigned e = fetcher ? rf : (unsigned)n + ((unsigned)lane - rf);
        if (__builtin_expect(no > WC_OVF, 0)) {
            // incoherent field: more rows collide than the overflow holds.  Forget the cache (the tags written above name rows that
            // will not be fetched) and fetch the 64 references as they are, into rows 0 .. 63
            tags[lane] = ~0u;
            tags[lane + 64] = ~0u;
            e = (unsigned)lane; fetcher = true; src = (unsigned)lane; n = 64;
        }
        const unsigned src128 = src * 128u;
        const unsigned tp = ((e & 7u) << 3) | (e >> 3);
        typedef unsigned wc_u2 __attribute__((ext_vector_type(2)));
        *(__attribute__((address_space(3))) volatile wc_u2 *)(wl + WC_LIST_OFF + tp * 8u) =
            (wc_u2){fetcher ? (m.rid << 7) : (WC_NOROW << 7), fetcher ? src128 : (unsigned)WC_TRASH_ROW * 128u};
        ((__attribute__((address_space(3))) volatile unsigned short *)(wl + WC_BC_OFF))[lane] = (unsigned short)src128;
        // group g fetches entries g, 8 + g, 16 + g ...: consecutive entries of the transposed list
        const nrt_i4 la = *(wc_lds_vi4 *)(wl + WC_LIST_OFF + g * 64), lb = *(wc_lds_vi4 *)(wl + WC_LIST_OFF + g * 64 + 16);
        const nrt_i4 sh = *(wc_lds_vi4 *)(wl + WC_BC_OFF + g * 16);
        s.lr[0] = (unsigned)la[0]; s.ld[0] = (unsigned)la[1]; s.lr[1] = (unsigned)la[2]; s.ld[1] = (unsigned)la[3];
        s.lr[2] = (unsigned)lb[0]; s.ld[2] = (unsigned)lb[1]; s.lr[3] = (unsigned)lb[2]; s.ld[3] = (unsigned)lb[3];
        if (__builtin_expect(n > 32, 0)) {
            const nrt_i4 lc = *(wc_lds_vi4 *)(wl + WC_LIST_OFF + g * 64 + 32), le = *(wc_lds_vi4 *)(wl + WC_LIST_OFF + g * 64 + 48);
            s.xr[0] = (unsigned)lc[0]; s.xd[0] = (unsigned)lc[1]; s.xr[1] = (unsigned)lc[2]; s.xd[1] = (unsigned)lc[3];
            s.xr[2] = (unsigned)le[0]; s.xd[2] = (unsigned)le[1]; s.xr[3] = (unsigned)le[2]; s.xd[3] = (unsigned)le[3];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { s.sl[2 * c] = (unsigned)sh[c] & 0xffffu; s.sl[2 * c + 1] = (unsigned)sh[c] >> 16; }
        s.w0x = m.w0x; s.w0y = m.w0y; s.w0z = m.w0z; s.oob = m.oob; s.n = n;
        if (BWD) { s.mk[0] = m.mk[0]; s.mk[1] = m.mk[1]; s.mk[2] = m.mk[2]; }
        s.xq = x0 + pass;
    };
    // M2: the row loads of the pass in s.  Four whatever n is (entries past n address bytes past the volume: no memory access): with
    // every load under a condition the compiler's in-order vmcnt count assumes none was issued and each wait covers the other state's
    // pass as well (round 4, v3: 1.37 ms); four more when the list is longer than 32 (19 % of the passes on the bench field)
    auto issue = [&](Pass &s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.F[i] = fetch_row(s.lr[i]);
        if (!XSH && __builtin_expect(s.n > 32, 0)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s.Fx[i] = fetch_row(s.xr[i]);
        }
        if (DICE) s.T = __builtin_bit_cast(nrt_f4, __builtin_amdgcn_raw_buffer_load_b128(fres, row_lane, (unsigned)s.xq * row_step, 2));
    };
    // (XSH) the rows 32 .. of the pass in s, into the shared registers: called at the end of the step BEFORE the one that blends the pass,
    // behind the delivery of the previous pass's
    auto issue_x = [&](Pass &s) {
        if (XSH && __builtin_expect(s.n > 32, 0)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) FxS[i] = fetch_row(s.xr[i]);
        }
    };
    // D: rows fetched for the pass -> their cache / overflow rows
    auto deliver = [&](Pass &s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(wc_lds_f4 *)(lrow + s.ld[i]) = s.F[i];
        if (__builtin_expect(s.n > 32, 0)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(wc_lds_f4 *)(lrow + s.xd[i]) = XSH ? FxS[i] : s.Fx[i];
        }
    };

    // One pass.  `s` holds pass `pass` (managed NST steps ago, rows in flight since NST - 1 steps), `o` pass + NST - 1 (managed in the last
    // step, its loads go out first thing here); the management of pass + NST is threaded through the blend so that each of its LDS round
    // trips has arithmetic to hide behind.  MASKED: the wave has voxels outside the volume (edge patches) whose sums must not count.
    // (Tried and measured slower, profiles/r05_lab/wc_probes.jsonl: the management BEFORE the row stores with the requests of pass + 2
    // leaving in the same step, 1.10 ms against 0.95; three management states with the requests two steps ahead: 256 registers, spills.)
    const int last = npass - 1;
    auto step = [&](int pass, Pass &s, Pass &o, Pass &nx, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        issue(o);                    // (before the wait for this pass's rows: a late row must not hold back the next pass's requests)
        __builtin_amdgcn_sched_barrier(0);
        deliver(s);
        nrt_f4 R[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) R[corner] = *(const wc_lds_f4 *)(lrow + s.sl[corner]);
        __builtin_amdgcn_sched_barrier(0);
        Mg m;
        m1a(min(pass + NST, last), s, m);
        __builtin_amdgcn_sched_barrier(0);
        // corner weights (wx * wy) * wz in the reference's order, two corners per packed multiply
        const float w1x = nrt_sub(1.0f, s.w0x), w1y = nrt_sub(1.0f, s.w0y), w1z = nrt_sub(1.0f, s.w0z);      // corner_1d's w1
        const nrt_f2 wy2 = {s.w0y, w1y}, wz2 = {s.w0z, w1z};
        const nrt_f2 wxy0 = (nrt_f2){s.w0x, s.w0x} * wy2, wxy1 = (nrt_f2){w1x, w1x} * wy2;
        nrt_f2 wt2[4];
        wt2[0] = (nrt_f2){wxy0[0], wxy0[0]} * wz2;
        wt2[1] = (nrt_f2){wxy0[1], wxy0[1]} * wz2;
        wt2[2] = (nrt_f2){wxy1[0], wxy1[0]} * wz2;
        wt2[3] = (nrt_f2){wxy1[1], wxy1[1]} * wz2;
        nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
        constexpr bool BLEND = BWD != 2;                          // (the gradient of the plain warp does not need the warped row)
#pragma unroll
        for (int corner = 0; corner < (BLEND ? 4 : 0); ++corner) {
            const float wt = wt2[corner >> 1][corner & 1];
            const nrt_f2 w2 = {wt, wt};
            al = al + w2 * (nrt_f2){R[corner][0], R[corner][1]};
            ah = ah + w2 * (nrt_f2){R[corner][2], R[corner][3]};
        }
        // (pure arithmetic is not ordered against the barriers by itself -- the compiler had sunk the whole blend below M1c: the
        // empty asm statements pin it, they take part in the chain of side effects the barriers are on)
        asm volatile("" : "+v"(al), "+v"(ah));
        __builtin_amdgcn_sched_barrier(0);
        m1b(m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int corner = 4; corner < (BLEND ? 8 : 4); ++corner) {
            const float wt = wt2[corner >> 1][corner & 1];
            const nrt_f2 w2 = {wt, wt};
            al = al + w2 * (nrt_f2){R[corner][0], R[corner][1]};
            ah = ah + w2 * (nrt_f2){R[corner][2], R[corner][3]};
        }
        nrt_f4 acc = {al[0], al[1], ah[0], ah[1]};
        const unsigned xqcur = (unsigned)s.xq;
        // (M1c below replaces the pass in s: what the gradient needs of this one)
        const float cw0x = s.w0x, cw0y = s.w0y, cw0z = s.w0z, cmx = s.mk[0], cmy = s.mk[1], cmz = s.mk[2];
        const bool coob = s.oob;
        if (!BWD) {
            if (FILL) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = apply_fill(acc[c], s.oob, a.fill_f);
            }
            if (STORE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wc_u4, acc), ores, out_lane, (unsigned)s.xq * row_step, 2);
        }
        // the Dice sums / the gradient wait behind M1c: they cover the round trip of its list, and the fixed row (a stream from HBM, the
        // slowest load of the pass, requested at the start of the last step) is first touched here
        const nrt_f4 Tcur = s.T;
        asm volatile("" : "+v"(acc));
        __builtin_amdgcn_sched_barrier(0);
        m1c(min(pass + NST, last), m, s);
        __builtin_amdgcn_sched_barrier(0);
        if (BWD) {
            // d loss / d warped for this lane's four labels, then its slope along the three axes (interpn_core.h: loc_grad_rows)
            nrt_f2 gl2 = {Tcur[0], Tcur[1]}, gh2 = {Tcur[2], Tcur[3]};       // BWD == 2: the incoming gradient row itself
            if (BWD == 1) {
                gl2 = __builtin_elementwise_fma((nrt_f2){cb[0], cb[1]}, (nrt_f2){acc[0], acc[1]}, (nrt_f2){ca[0], ca[1]} * gl2);
                gh2 = __builtin_elementwise_fma((nrt_f2){cb[2], cb[3]}, (nrt_f2){acc[2], acc[3]}, (nrt_f2){ca[2], ca[3]} * gh2);
            }
            if (FILL && coob) { gl2 = (nrt_f2){0.0f, 0.0f}; gh2 = gl2; }
            float gacc[3];
            loc_grad_rows(R, gl2, gh2, cw0x, w1x, cw0y, w1y, cw0z, w1z, cmx, cmy, cmz, gacc);
            // sum over the voxel's 8 lanes on the DPP network (quad xor 1, quad xor 2, then the mirrored half: after the two quad steps a
            // quad's lanes hold the same value, so i <-> 7 - i adds the other quad exactly as xor 4 would)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float r = gacc[d];
                r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0xB1, 0xF, 0xF, true));
                r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x4E, 0xF, 0xF, true));
                r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x141, 0xF, 0xF, true));
                gacc[d] = r;
            }
            // (one store per 8 passes with all lanes active, lane p keeping the sums of pass 8 k + p, measured the same 1.19 ms)
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(wc_u3, (wc_f3){gacc[0], gacc[1], gacc[2]}), gres, gl_lane, xqcur * loc_step, 0);
        } else if (DICE) {
            nrt_f4 T = Tcur;
            if (MASKED) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { T[c] = yzvalid ? T[c] : 0.0f; acc[c] = yzvalid ? acc[c] : 0.0f; }
            }
            const nrt_f2 pl = {acc[0], acc[1]}, ph = {acc[2], acc[3]}, tl = {T[0], T[1]}, th = {T[2], T[3]};
            stp_l = stp_l + tl * pl; stp_h = stp_h + th * ph;
            stt_l = stt_l + tl * tl; stt_h = stt_h + th * th;
            spp_l = spp_l + pl * pl; spp_h = spp_h + ph * ph;
            if (MM) {
                if (!MASKED || yzvalid) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        mnt = fminf(mnt, Tcur[c]); mxt = fmaxf(mxt, Tcur[c]);
                        mnp = fminf(mnp, acc[c]); mxp = fmaxf(mxp, acc[c]);
                    }
                }
            }
        }
        if (DICE && !BWD) asm volatile("" : "+v"(stp_l), "+v"(stp_h), "+v"(stt_l), "+v"(stt_h), "+v"(spp_l), "+v"(spp_h));
        else asm volatile("" :: "v"(acc));
        __builtin_amdgcn_sched_barrier(0);
        fetch_loc(min(pass + 2 * NST, last), s);
        issue_x(nx);                 // (XSH: nx = the pass of the next step)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto march = [&](auto masked) {
        Mg m;
        if constexpr (NST == 3) {
            // pass p lives in state p % 3; step p requests the rows of pass p + 2 (managed in step p - 1) and manages pass p + 3
            fetch_loc(0, A);
            fetch_loc(min(1, last), B);
            fetch_loc(min(2, last), C);
            m1a(0, A, m); m1b(m); m1c(0, m, A);
            issue(A);
            fetch_loc(min(3, last), A);
            m1a(min(1, last), B, m); m1b(m); m1c(min(1, last), m, B);
            issue(B);
            fetch_loc(min(4, last), B);
            m1a(min(2, last), C, m); m1b(m); m1c(min(2, last), m, C);
            fetch_loc(min(5, last), C);
            issue_x(A);
            __builtin_amdgcn_sched_barrier(0);
            int pass = 0;
            for (; pass + 2 < npass; pass += 3) {
                step(pass, A, C, B, masked);
                step(pass + 1, B, A, C, masked);
                step(pass + 2, C, B, A, masked);
            }
            if (pass < npass) step(pass, A, C, B, masked);        // (what the last steps prepare beyond the end is not used)
            if (pass + 1 < npass) step(pass + 1, B, A, C, masked);
            return;
        }
        fetch_loc(0, A);
        fetch_loc(min(1, last), B);
        m1a(0, A, m); m1b(m); m1c(0, m, A);
        issue(A);
        fetch_loc(min(2, last), A);
        m1a(min(1, last), B, m); m1b(m); m1c(min(1, last), m, B);
        fetch_loc(min(3, last), B);
        __builtin_amdgcn_sched_barrier(0);
        int pass = 0;
        for (; pass + 1 < npass; pass += 2) {
            step(pass, A, B, B, masked);
            step(pass + 1, B, A, A, masked);
        }
        if (pass < npass) step(pass, A, B, B, masked);            // odd march: one more pass (what it prepares beyond the end is not used)
    };
    if (npass > 0) {
        // (wave-uniform) edge patches mask the sums of the lanes whose voxel lies outside the volume
        if (DICE && !BWD && __builtin_amdgcn_ballot_w64(!yzvalid) != 0ull) march(std::true_type());
        else march(std::false_type());
    }

    if (!DICE || BWD) return;
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
    const long long pbase = (long long)b * tg.prows + prow;
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

// The work counters of a persistent launch live in a self-cleaning slot of the library's counter ring (api.hip: nrt_ring_slot): zero when
// the launch starts, zeroed again by its last block.  History: hipMemsetAsync in front of the launch did not take effect between the
// replays of a captured hipGraph (ROCm 7.2) -- every replay after the first found the lists exhausted, its blocks left at once and the
// second stage re-reduced the previous replay's partial sums, stale results in 0.03 ms (tools/graph_fused_probe.py;
// tests/test_gpu_dice_cce.py::test_fused_kernels_recompute_under_graph_replay); a zeroing KERNEL in front (round 4) cost a launch.

// PERSIST: 2 blocks per CU stay resident and take items from per-XCD lists (atomic counters in `queue`, zero at the start):
// first their own XCD's (its L2 holds the neighbouring columns), then the others'.  With one block per item the XCDs finish up to
// 5 % apart (tools/block_trace.py: last block of an XCD at 1044 .. 1097 us) and the slots of a finished XCD idle; the items are the
// same either way, every item writes its own partial row, so the sums do not depend on who computed what.
// the loop of a persistent block: ITEM(xw) computes one work item
template <typename ITEM>
__device__ __forceinline__ void wc_persistent_loop(const TileGeom &tg, int O0, unsigned *__restrict__ queue, ITEM item) {
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
        if (!xmarch_work_at(tg, O0, kk, jb, xw)) continue;
        item(xw);
    }
    // a block comes here once every list is exhausted and never touches the counters again: the last one to arrive zeroes them for the
    // slot's next launch (nrt_ring_slot)
    if (threadIdx.x == 0 && atomicAdd(&queue[NRT_NXCD * 16u], 1u) == gridDim.x - 1) {
        for (unsigned q = 0; q <= NRT_NXCD; ++q) queue[q * 16u] = 0u;
    }
}

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
    wc_persistent_loop(tg, a.O[0], queue, [&](const XmWork &xw) { wc_item<MODE, STORE, MM, FILL, DICE>(a, tg, fixed, fpart, mpart, xw, wc_smem); });
}

// d loss / d loc on the same gather (wc_item's BWD): `rows` = the fixed map (BWD 1) or the gradient arriving at the warped map (BWD 2)
template <int MODE, bool FILL, int BWD, bool PERSIST>
__global__ __launch_bounds__(256, 2) void warp_dice_wc_bwd(InterpArgs a, TileGeom tg, const float *__restrict__ rows, WcBwd bw,
                                                            unsigned *__restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char wc_smem[];
    if (!PERSIST) {
        XmWork xw;
        if (!xmarch_work(tg, a.O[0], xw)) return;
        wc_item<MODE, false, false, FILL, true, BWD>(a, tg, rows, nullptr, nullptr, xw, wc_smem, bw);
        return;
    }
    wc_persistent_loop(tg, a.O[0], queue, [&](const XmWork &xw) { wc_item<MODE, false, false, FILL, true, BWD>(a, tg, rows, nullptr, nullptr, xw, wc_smem, bw); });
}

// persistent blocks pay off when there are more work items than resident blocks
inline bool wc_persistent(const TileGeom &tg) { return NRT_NXCD * tg.items_x > 2u * (unsigned)nrt_num_cus(); }
static_assert(NRT_RING_GATHER_OFF == 0 && (NRT_NXCD + 1) * 16 <= NRT_RING_CCE_OFF, "the work counters of a launch stay in the gather's words of a slot");

template <int MODE, bool STORE, bool MM, bool FILL>
int launch_wc_inst(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, const float *fixed, float *fpart, float *mpart,
                   hipStream_t st) {
    const unsigned items = NRT_NXCD * tg.items_x, slots = 2u * (unsigned)nrt_num_cus();
    unsigned *queue = wc_persistent(tg) ? nrt_ring_slot(st) : nullptr;
    if (queue) {
        if (hipFuncSetAttribute((const void *)warp_dice_wc<MODE, STORE, MM, FILL, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WC_BLOCK_BYTES) != hipSuccess)
            return NRT_ERR_LAUNCH;
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
                   float *fpart, float *mpart, hipStream_t st) {
    if (store) return minmax ? launch_wc_inst<MODE, true, true, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, st)
                             : launch_wc_inst<MODE, true, false, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, st);
    return minmax ? launch_wc_inst<MODE, false, true, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, st)
                  : launch_wc_inst<MODE, false, false, FILL>(a, tg, nblocks, batch, fixed, fpart, mpart, st);
}

template <int MODE>
int launch_wc_mode(const InterpArgs &a, const TileGeom &tg, unsigned nblocks, int batch, bool store, bool minmax, const float *fixed,
                   float *fpart, float *mpart, hipStream_t st) {
    return a.has_fill ? launch_wc_fill<MODE, true>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, st)
                      : launch_wc_fill<MODE, false>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, st);
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
                     const float *fixed, float *fpart, float *mpart, hipStream_t st) {
    switch (mode) {
        case NRT_LOC_ABSOLUTE: return launch_wc_mode<NRT_LOC_ABSOLUTE>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, st);
        case NRT_LOC_SHIFT: return launch_wc_mode<NRT_LOC_SHIFT>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, st);
        default: return launch_wc_mode<NRT_LOC_LINSPACE>(a, tg, nblocks, batch, store, minmax, fixed, fpart, mpart, st);
    }
}

// the warp alone through the same kernel (interpn.hip, variant 10): mixed block lengths, persistent blocks
template <int MODE, bool FILL>
int launch_wc_interpn_inst(const InterpArgs &a, const TileGeom &tg, hipStream_t st) {
    const unsigned items = NRT_NXCD * tg.items_x, slots = 2u * (unsigned)nrt_num_cus();
    unsigned *queue = (items > slots) ? nrt_ring_slot(st) : nullptr;
    if (queue) {
        if (hipFuncSetAttribute((const void *)warp_dice_wc<MODE, true, false, FILL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WC_BLOCK_BYTES) != hipSuccess)
            return NRT_ERR_LAUNCH;
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

template <int MODE, bool FILL, int BWD>
int launch_wc_bwd_inst(const InterpArgs &a, const TileGeom &tg, const float *rows, const WcBwd &bw, hipStream_t st) {
    const unsigned items = NRT_NXCD * tg.items_x, slots = 2u * (unsigned)nrt_num_cus();
    unsigned *queue = (items > slots) ? nrt_ring_slot(st) : nullptr;
    if (queue) {
        if (hipFuncSetAttribute((const void *)warp_dice_wc_bwd<MODE, FILL, BWD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WC_BLOCK_BYTES) != hipSuccess)
            return NRT_ERR_LAUNCH;
        hipLaunchKernelGGL((warp_dice_wc_bwd<MODE, FILL, BWD, true>), dim3(nrt_xcd_grid(slots)), dim3(256), WC_BLOCK_BYTES, st, a, tg, rows, bw, queue);
        return NRT_OK;
    }
    if (hipFuncSetAttribute((const void *)warp_dice_wc_bwd<MODE, FILL, BWD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WC_BLOCK_BYTES) != hipSuccess)
        return NRT_ERR_LAUNCH;
    hipLaunchKernelGGL((warp_dice_wc_bwd<MODE, FILL, BWD, false>), dim3(items), dim3(256), WC_BLOCK_BYTES, st, a, tg, rows, bw, (unsigned *)nullptr);
    return NRT_OK;
}

template <int MODE, bool FILL>
int launch_wc_bwd_mode(const InterpArgs &a, const TileGeom &tg, const float *rows, const WcBwd &bw, hipStream_t st) {
    return bw.sums ? launch_wc_bwd_inst<MODE, FILL, 1>(a, tg, rows, bw, st) : launch_wc_bwd_inst<MODE, FILL, 2>(a, tg, rows, bw, st);
}

}  // namespace

int nrt_wc_bwd_launch(const void *args, int batch, int mode, const float *rows, const float *sums, const float *grad_dice, float eps,
                      float *grad_loc, void *stream) {
    const InterpArgs &a = *(const InterpArgs *)args;
    if (mode != NRT_LOC_ABSOLUTE && mode != NRT_LOC_SHIFT) return NRT_ERR_INVALID_ARG;
    TileGeom tg;
    unsigned nblocks, grid;
    const int t = xmarch_default_tune();
    tile_geometry(a.O, 8, t, t, tg, nblocks);
    xmarch_setup_mixed(a.O, batch, t, tg, grid);
    hipStream_t st = nrt_stream(stream);
    const WcBwd bw = {sums, grad_dice, eps, grad_loc};
    int rc;
    if (mode == NRT_LOC_ABSOLUTE) rc = a.has_fill ? launch_wc_bwd_mode<NRT_LOC_ABSOLUTE, true>(a, tg, rows, bw, st) : launch_wc_bwd_mode<NRT_LOC_ABSOLUTE, false>(a, tg, rows, bw, st);
    else rc = a.has_fill ? launch_wc_bwd_mode<NRT_LOC_SHIFT, true>(a, tg, rows, bw, st) : launch_wc_bwd_mode<NRT_LOC_SHIFT, false>(a, tg, rows, bw, st);
    if (rc != NRT_OK) return rc;
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

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
