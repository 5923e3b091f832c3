// LDS-row-cache gather for 32 float channels on gfx950 (MI355X): linear interpn / SpatialTransformer, optionally fused with
// the soft-Dice sums.  Reference op: neurite/tf/utils/utils.py:137-191 (+ metrics.py:415-482 for the Dice form).
//
// Why: the register kernels (interpn.hip, fused.hip) pull 8 corner rows + 1 fixed row = 1152 B per voxel through the
// texture path (TA -> L1 -> VGPR, 64 B/clk/CU); of those only ~2.3 rows are HBM misses, the rest are L1 / L2 hits that still
// occupy the path (profiles/r02_lab: hits and misses issued by one CU cost additively).  On the SURVEY 8d field a voxel shares
// its corner rows with its (y,z) neighbours and with the next x plane: an x-marching 4 x 4 patch needs only ~2.1 NEW rows per
// voxel.  So every wave keeps a private software cache of source rows in LDS and the texture path carries each row once:
//
//   * one wave = one workgroup = one 4 x 4 (y,z) patch marching along x; a step is one x plane (16 voxels);
//   * LDS per wave (38.1 KB, four waves per CU): 224-row FIFO ring + 2 x 16 scratch rows (128 B each), a 1024-entry
//     direct-mapped tag table (hash = low bits of the row's x, y, z), a fetch list and three hand-off buffers;
//   * MGMT(t+2): lane = (voxel, x-corner, y-corner) looks its two z-corner rows up in the tag table; misses are claimed through
//     the table (write token, read back: one owner per distinct row), owners take consecutive ring slots (ballot + mbcnt) and
//     append the row to the fetch list; the slots of all 8 corners and the weights go to the hand-off buffer;
//   * ISSUE(t+1): the list is fetched by global_load_lds_dwordx4 (8 rows per instruction, straight into the ring);
//   * BLEND(t): 8 lanes per voxel read the 8 corner rows from LDS (ds_read_b128) and run the reference's op sequence
//     (bit-identical to interpn.hip), the fixed row comes from registers loaded one step ahead.
//   A ring row may be overwritten only when no step that still has to blend needs it: a step may allocate at most
//   224 - (age of the oldest row the previous and the current step hit) rows; the overflow goes to the scratch rows of the
//   step and, beyond those (1-2 % of the steps on the 8d field), to direct loads in the blend ("slow" corners).
//   All vector-memory loads are inline asm: the compiler's s_waitcnt model would otherwise serialise against the LDS-DMA.
#include <stdlib.h>

#include "dice_reduce.h"
#include "interpn_core.h"
#include "lc.h"

namespace {

#ifdef LC_PHASE_MARKS
#define LC_PH(x) asm volatile("; ##PH " x ::: "memory")
#else
#define LC_PH(x)
#endif

constexpr int LC_NR = 224;            // ring rows (multiple of 8)
constexpr int LC_SCR = 16;            // scratch rows per step, two buffers
constexpr int LC_WIN = 112;           // a cached row counts as a hit while it is younger than this many allocations
constexpr unsigned LC_TAB = 32768;    // byte offsets inside the wave's LDS
constexpr unsigned LC_HAND = LC_TAB + 4096;
constexpr unsigned LC_LIST = LC_HAND + 3 * 512;
constexpr unsigned LC_LIST_SCR = 128; // list index of the first scratch entry
constexpr unsigned LC_DUMMY = LC_LIST + (LC_LIST_SCR + LC_SCR + 8) * 4;   // 64 dwords
constexpr unsigned LC_LDS = LC_DUMMY + 256;
static_assert(LC_LDS <= 40960, "four waves per CU");

struct LcK {
    const char *vol, *loc, *fixed;
    char *out;
    float *fpart, *mpart;
    int S0, S1, S2, O0, O1, O2;
    float d0, d1, d2;
    unsigned long long vol_bs, loc_bs, out_bs;   // bytes per batch entry
    int has_fill;
    float fill;
    unsigned nTy, nTz, ncol, nseg, seglen, nbatch;
    int lry, lrz;
    unsigned nyh, nzh, ntask;
    int minmax;
};

typedef unsigned nrt_u4 __attribute__((ext_vector_type(4)));
typedef float nrt_f3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ void lc_dma16(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// the same with an explicit lane mask (rows of the list that do not exist): no branch around the instruction
__device__ __forceinline__ void lc_dma16m(const void *base, unsigned voff, unsigned lds_dst, unsigned long long mask) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(base), "s"(lds_dst), "s"(mask) : "memory");
}
__device__ __forceinline__ nrt_f4 lc_ld16(const void *base, unsigned voff) {
    nrt_f4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(voff), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ nrt_f4 lc_ld16c(const void *base, unsigned voff) {     // cached (source rows)
    nrt_f4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ nrt_f3 lc_ld12(const void *base, unsigned voff) {
    nrt_f3 r;
    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ unsigned lc_mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ unsigned lc_uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ unsigned lc_sel(bool c, unsigned x, unsigned y) { return c ? x : y; }
// workgroup barrier that also publishes this wave's LDS writes / retires its LDS reads (no vmcnt: the waves time their own loads)
__device__ __forceinline__ void lc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }

// MODE = location mode; DICE = accumulate the soft-Dice sums against `fixed`; STORE = write the warped rows; FILL = fill_value
// given.  DIAG: 0 = product; 1 = no tag protocol (every corner "hits" slot id % 224, 32 arbitrary rows fetched per step): the cost
// of the data path alone (profiles/r03_lc).
// A workgroup is two waves that share one row cache: wave 0 manages it (tags, allocation, fetch list, LDS-DMA), wave 1 blends.
// One s_barrier per step: A(t) = "rows and hand-off of step t are in LDS, blend of step t - 1 is done".
template <int MODE, bool DICE, bool STORE, bool FILL, int DIAG>
__global__ __launch_bounds__(128) void gather_lc(LcK a) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[LC_LDS];
    const unsigned l = threadIdx.x & 63u;
    const bool is_mgmt = lc_uni(threadIdx.x >> 6) == 0u;
    const unsigned lds0 = lc_uni((unsigned)(size_t)sm);
    unsigned *const tab = (unsigned *)(sm + LC_TAB);
    unsigned *const list = (unsigned *)(sm + LC_LIST);
    unsigned *const dummy = (unsigned *)(sm + LC_DUMMY) + l;       // where the writes of lanes that have nothing to write go (no branch)

    const unsigned kx = blockIdx.x % NRT_NXCD, jw = blockIdx.x / NRT_NXCD, J = gridDim.x / NRT_NXCD;
    const unsigned perU = (a.ntask + NRT_NXCD - 1) / NRT_NXCD;
    const unsigned uend = min((kx + 1) * perU, a.ntask);
    const unsigned per_batch = a.ncol * a.nseg;

    for (unsigned u = kx * perU + jw; u < uend; u += J) {
        const unsigned b = u / per_batch, prow = u % per_batch;
        const unsigned useg = prow / a.ncol, ucol = prow % a.ncol;
        const unsigned RY = 1u << a.lry, RZ = 1u << a.lrz;
        const unsigned nRz = (a.nTz + RZ - 1) / RZ;
        const unsigned reg = ucol / (RY * RZ), w = ucol % (RY * RZ);
        const unsigned cy = (reg / nRz) * RY + w / RZ, cz = (reg % nRz) * RZ + w % RZ;
        const int x0 = (int)(useg * a.seglen), y0 = (int)cy * 4, z0 = (int)cz * 4;
        int len = min((int)a.seglen, a.O0 - x0);
        if (cy >= a.nTy || cz >= a.nTz) len = 0;
        const char *volb = a.vol + (unsigned long long)b * a.vol_bs;

        if (is_mgmt) {
            // ================================ wave 0: cache management ================================
            if (len > 0) {
                const char *locb = a.loc ? a.loc + (unsigned long long)b * a.loc_bs : a.vol;
                // lane = (voxel mv of the plane, x corner mxc, y corner myc), both z corners
                const unsigned mv = l & 15, myy = mv >> 2, mzz = mv & 3, mcp = l >> 4, mxc = mcp >> 1, myc = mcp & 1;
                const unsigned lrow = l >> 3, lg = l & 7;                  // DMA role: row lrow of an 8-row instruction, 16 bytes lg
                unsigned H = 0, Hpos = 0, Aprev = 0, Hprev = 0;           // allocation counter, ring position, oldest row of the previous step
                unsigned inr = 0, ins = 0, ihp = 0;                       // what the next ISSUE fetches: ring rows, scratch rows, ring position
                if (DIAG == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ((nrt_u4 *)tab)[i * 64 + l] = (nrt_u4){~0u, ~0u, ~0u, ~0u};
                }
                // lanes past the volume's edge hold a copy of the edge voxel (clamped coordinates)
                const int myq = min(y0 + (int)myy, a.O1 - 1), mzq = min(z0 + (int)mzz, a.O2 - 1);
                const bool mvalid = (y0 + (int)myy < a.O1) && (z0 + (int)mzz < a.O2);
                auto load_shift = [&](int t) -> nrt_f3 {
                    if (MODE == NRT_LOC_LINSPACE) return (nrt_f3){0, 0, 0};
                    const unsigned q = nrt_mad24(nrt_mad24((unsigned)min(x0 + t, a.O0 - 1), (unsigned)a.O1, (unsigned)myq), (unsigned)a.O2, (unsigned)mzq);
                    return lc_ld12(locb, nrt_times3(q) << 2);
                };

                // ---- MGMT(t): tag lookups, claims, slot allocation, fetch list, hand-off ----
                auto mgmt = [&](int t, bool live, const nrt_f3 &sh) {
                    const int qd[3] = {min(x0 + t, a.O0 - 1), myq, mzq};
                    const int Sd[3] = {a.S0, a.S1, a.S2}, Od[3] = {a.O0, a.O1, a.O2};
                    const float dd[3] = {a.d0, a.d1, a.d2};
                    float p[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        if (MODE == NRT_LOC_ABSOLUTE) p[d] = sh[d];
                        else if (MODE == NRT_LOC_SHIFT) p[d] = nrt_add((float)qd[d], sh[d]);
                        else p[d] = (qd[d] == 0) ? 0.0f : ((qd[d] == Od[d] - 1) ? (float)(Sd[d] - 1) : nrt_mul(dd[d], (float)qd[d]));
                    }
                    int i0x, i1x, i0y, i1y, i0z, i1z;
                    float w0x, w0y, w0z, w1x, w1y, w1z;
                    corner_1d(p[0], a.S0, i0x, i1x, w0x, w1x);
                    corner_1d(p[1], a.S1, i0y, i1y, w0y, w1y);
                    corner_1d(p[2], a.S2, i0z, i1z, w0z, w1z);
                    bool oob = false;
                    if (FILL) oob = (p[0] < 0.0f) || (p[0] > (float)(a.S0 - 1)) || (p[1] < 0.0f) || (p[1] > (float)(a.S1 - 1)) ||
                                    (p[2] < 0.0f) || (p[2] > (float)(a.S2 - 1));
                    const unsigned sx = lc_sel(mxc != 0, (unsigned)i1x, (unsigned)i0x), sy = lc_sel(myc != 0, (unsigned)i1y, (unsigned)i0y);
                    const unsigned rowxy = nrt_mad24(sx, (unsigned)a.S1, sy);
                    const unsigned hxy = ((sx & 3) << 8) | ((sy & 15) << 4), hixy = nrt_mad24(sx >> 2, a.nyh, sy >> 4);
                    unsigned id[2], idhi[2], slot[2], slow[2], tok[2];
                    unsigned *tp[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned sz = j ? (unsigned)i1z : (unsigned)i0z;
                        id[j] = nrt_mad24(rowxy, (unsigned)a.S2, sz);
                        tp[j] = tab + (hxy | (sz & 15));
                        idhi[j] = nrt_mad24(hixy, a.nzh, sz >> 4) << 17;          // kept in place: bits 17..31 of a tag
                        tok[j] = idhi[j] | 0x10000u | (l << 1) | (unsigned)j;     // the claim this lane would write
                    }
                    const unsigned id0 = nrt_mad24(nrt_mad24((unsigned)i0x, (unsigned)a.S1, (unsigned)i0y), (unsigned)a.S2, (unsigned)i0z);
                    const unsigned flags = ((unsigned)(i1x != i0x) << 26) | ((unsigned)(i1y != i0y) << 27) | ((unsigned)(i1z != i0z) << 28) |
                                           ((unsigned)mvalid << 29) | ((unsigned)oob << 30);
                    H = (H + 7u) & ~7u;                                   // ring allocations of a step start on a multiple of 8 rows
                    unsigned nr, ns;
                    if (DIAG == 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) { slot[j] = id[j] % (unsigned)LC_NR; slow[j] = 0; }
                        *(l < 32 ? &list[l] : dummy) = id[0];
                        nr = live ? 32u : 0u; ns = 0;
                    } else {
                        // 1. hits
                        unsigned e[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) e[j] = *tp[j];
                        bool miss[2];
                        unsigned age = 0;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const unsigned ag = (H - e[j]) & 0xffffu;
                            const bool hit = ((e[j] ^ idhi[j]) >> 16) == 0u && ag < (unsigned)LC_WIN;     // same row, not a claim / scratch tag, young
                            miss[j] = !hit;
                            age = max(age, lc_sel(hit, ag, 0u));
                            const unsigned pos = Hpos - ag, posw = pos + (unsigned)LC_NR;
                            slot[j] = lc_sel(hit, lc_sel((int)pos < 0, posw, pos), 0u);
                        }
                        // 2. the oldest row that this step, or the previous one (not blended yet), still reads bounds the allocation
                        age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0xB1, 0xF, 0xF, false));    // lane ^ 1
                        age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0x4E, 0xF, 0xF, false));    // lane ^ 2
                        age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0x141, 0xF, 0xF, false));   // row_half_mirror
                        age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0x140, 0xF, 0xF, false));   // row_mirror
                        const unsigned amax = max(max((unsigned)__builtin_amdgcn_readlane((int)age, 0), (unsigned)__builtin_amdgcn_readlane((int)age, 16)),
                                                  max((unsigned)__builtin_amdgcn_readlane((int)age, 32), (unsigned)__builtin_amdgcn_readlane((int)age, 48)));
                        const unsigned keep = max(amax, Aprev + (H - Hprev));
                        const unsigned room = keep >= (unsigned)LC_NR ? 0u : (unsigned)LC_NR - keep;
                        const unsigned limit = min(room, LC_LIST_SCR - 8u) & ~7u;   // whole 8-row instructions (their padding rows are written
                                                                                       // too); the list holds 128 ring entries incl. padding
                        Aprev = amax; Hprev = H;
                        // 3. claim: one owner per distinct missing row
#pragma unroll
                        for (int j = 0; j < 2; ++j) *(miss[j] ? tp[j] : dummy) = tok[j];
                        unsigned e2[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) e2[j] = *tp[j];
                        bool owner[2], follower[2], fetcher[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const bool same = ((e2[j] ^ idhi[j]) >> 17) == 0u;
                            owner[j] = miss[j] && e2[j] == tok[j];
                            follower[j] = miss[j] && same && !owner[j];
                            fetcher[j] = miss[j] && !follower[j];         // owners, and rows that lost their table entry to another row
                        }
                        // 4. consecutive slots for everything that is fetched
                        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(fetcher[0]), b1 = __builtin_amdgcn_ballot_w64(fetcher[1]);
                        const unsigned c0 = (unsigned)__builtin_popcountll(b0), ntot = c0 + (unsigned)__builtin_popcountll(b1);
                        const unsigned pos[2] = {lc_mbcnt(b0), c0 + lc_mbcnt(b1)};
                        nr = live ? min(ntot, limit) : 0u;
                        ns = live ? min(ntot - nr, (unsigned)LC_SCR) : 0u;
                        // pad both lists to whole instructions with a row of this step (overwritten below where a real entry exists)
                        unsigned *const padp = l < 8 ? &list[(nr & ~7u) + l] : &list[LC_LIST_SCR + (ns & ~7u) + (l & 7)];
                        *(l < 16 ? padp : dummy) = id[0];
                        const unsigned sbase = (unsigned)LC_NR + ((unsigned)t & 1u) * (unsigned)LC_SCR;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const bool ring = fetcher[j] && pos[j] < nr, scr = fetcher[j] && !ring && pos[j] < nr + ns;
                            const unsigned rp0 = Hpos + pos[j], rp = lc_sel(rp0 >= (unsigned)LC_NR, rp0 - (unsigned)LC_NR, rp0);
                            const unsigned si = pos[j] - nr, sslot = sbase + si;
                            slot[j] = lc_sel(ring, rp, lc_sel(scr, sslot, slot[j]));
                            slow[j] = (fetcher[j] && !ring && !scr) ? 1u : 0u;
                            unsigned *const lp_ring = &list[pos[j]], *const lp_scr = &list[LC_LIST_SCR + si];
                            unsigned *const lp = ring ? lp_ring : (scr ? lp_scr : dummy);
                            *lp = id[j];
                            const unsigned tag_ring = idhi[j] | ((H + pos[j]) & 0xffffu), tag_scr = idhi[j] | 0x18000u | sslot;
                            const unsigned tagv = lc_sel(ring, tag_ring, lc_sel(scr, tag_scr, ~0u));
                            *(owner[j] ? tp[j] : dummy) = tagv;
                        }
                        // 5. rows another lane fetches
                        unsigned e3[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) e3[j] = *tp[j];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const bool same = ((e3[j] ^ idhi[j]) >> 17) == 0u;
                            const unsigned rp0 = Hpos + ((e3[j] - H) & 0xffffu), rp = lc_sel(rp0 >= (unsigned)LC_NR, rp0 - (unsigned)LC_NR, rp0);
                            const unsigned fs = lc_sel((e3[j] & 0x10000u) != 0u, e3[j] & 0xffu, rp);      // scratch row of this step : ring row
                            slot[j] = lc_sel(follower[j], lc_sel(same, fs, 0u), slot[j]);
                            slow[j] = lc_sel(follower[j] && !same, 1u, slow[j]);
                        }
                    }
                    // hand-off: dword cp = LDS byte offsets of the lane's two rows (bit 0 = slow); dwords 4..7 (same value from the four
                    // lanes of the voxel) = weights, row of corner 0, flags
                    unsigned char *hb = sm + LC_HAND + ((unsigned)t % 3u) * 512u + mv * 32u;
                    ((unsigned *)hb)[mcp] = ((slot[0] << 7) | slow[0]) | (((slot[1] << 7) | slow[1]) << 16);
                    *(nrt_u4 *)(hb + 16) = (nrt_u4){__float_as_uint(w0x), __float_as_uint(w0y), __float_as_uint(w0z), id0 | flags};
                    inr = nr; ins = ns; ihp = Hpos;
                    H += nr;
                    const unsigned hp1 = Hpos + ((nr + 7u) & ~7u);
                    Hpos = lc_sel(hp1 >= (unsigned)LC_NR, hp1 - (unsigned)LC_NR, hp1);
                };

                // ---- ISSUE(t): fetch the list of step t into the ring / scratch rows.  Whole instructions only: the list is padded to
                // a multiple of 8 rows with the address of a row that is fetched anyway, and the rows of the padding land in the ring
                // slots that the 8-row alignment of a step leaves unused ----
                auto issue_rows = [&](int t) {
                    const unsigned nr = inr, ns = ins, hp = ihp;
                    unsigned v[8], vs[2];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = list[8 * i + lrow];
#pragma unroll
                    for (int i = 0; i < 2; ++i) vs[i] = list[LC_LIST_SCR + 8 * i + lrow];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (8u * i < nr) {
                            const unsigned rp0 = hp + 8u * i, rp = rp0 >= (unsigned)LC_NR ? rp0 - (unsigned)LC_NR : rp0;
                            lc_dma16(volb, (v[i] << 7) + lg * 16u, lds0 + rp * 128u);
                        }
                    }
                    if (nr > 64) {
                        for (unsigned i = 8; 8u * i < nr; ++i) {
                            const unsigned rp0 = hp + 8u * i, rp = rp0 >= (unsigned)LC_NR ? rp0 - (unsigned)LC_NR : rp0;
                            const unsigned idr = list[8 * i + lrow];
                            lc_dma16(volb, (idr << 7) + lg * 16u, lds0 + rp * 128u);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (8u * i < ns)
                            lc_dma16(volb, (vs[i] << 7) + lg * 16u, lds0 + ((unsigned)LC_NR + ((unsigned)t & 1u) * (unsigned)LC_SCR + 8u * i) * 128u);
                };

                // iteration t: rows of step t have landed -> A(t) -> ISSUE(t + 1), MGMT(t + 2)
                nrt_f3 S0r = load_shift(0), S1r = load_shift(min(1, len - 1));
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0r), "+v"(S1r) : : "memory");
                mgmt(0, true, S0r);
                issue_rows(0);
                S0r = load_shift(min(2, len - 1));
                mgmt(min(1, len - 1), 1 < len, S1r);
                S1r = load_shift(min(3, len - 1));
                for (int t = 0; t < len; t += 2) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0r), "+v"(S1r) : : "memory");
                    lc_barrier();                                          // A(t)
                    issue_rows(t + 1);
                    mgmt(min(t + 2, len - 1), t + 2 < len, S0r);
                    S0r = load_shift(min(t + 4, len - 1));
                    if (t + 1 >= len) break;
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0r), "+v"(S1r) : : "memory");
                    lc_barrier();                                          // A(t + 1)
                    issue_rows(t + 2);
                    mgmt(min(t + 3, len - 1), t + 3 < len, S1r);
                    S1r = load_shift(min(t + 5, len - 1));
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0r), "+v"(S1r) : : "memory");
            }
            lc_barrier();                                                  // A(len): the blend of the last step is done
        } else {
            // ================================ wave 1: blend ================================
            const char *fixb = DICE ? a.fixed + (unsigned long long)b * a.out_bs : a.vol;
            char *outb = STORE ? a.out + (unsigned long long)b * a.out_bs : nullptr;
            const unsigned bvv = l >> 3, lg = l & 7;        // lane group bvv (voxel 8 s + bvv of the plane), channels 4 lg .. 4 lg + 3
            const unsigned SYZ = (unsigned)a.S1 * (unsigned)a.S2;
            nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
            float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;
            if (len > 0) {
                unsigned qyz[2];                                             // (y, z) part of the voxel index of the two sub-passes
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned v = 8u * s + bvv;
                    qyz[s] = nrt_mad24((unsigned)min(y0 + (int)(v >> 2), a.O1 - 1), (unsigned)a.O2, (unsigned)min(z0 + (int)(v & 3), a.O2 - 1));
                }
                const unsigned OYZ = (unsigned)a.O1 * (unsigned)a.O2;
                auto row_off = [&](int t, int s) -> unsigned { return ((nrt_mad24((unsigned)min(x0 + t, a.O0 - 1), OYZ, qyz[s])) * 8u + lg) * 16u; };
                auto load_fixed = [&](int t, nrt_f4 (&F)[2]) {
                    if (!DICE) return;
#pragma unroll
                    for (int s = 0; s < 2; ++s) F[s] = lc_ld16(fixb, row_off(t, s));
                };
                // BLEND(t): 8 lanes per voxel read the 8 corner rows from LDS and run the reference's op sequence
                auto blend = [&](int t, const nrt_f4 (&F)[2]) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned v = 8u * s + bvv;
                        const unsigned char *hb = sm + LC_HAND + ((unsigned)t % 3u) * 512u + v * 32u;
                        const nrt_u4 A = *(const nrt_u4 *)hb, Bw = *(const nrt_u4 *)(hb + 16);
                        nrt_f4 R[8];
                        unsigned slowm = 0;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned f = (c & 1) ? (A[c >> 1] >> 16) : (A[c >> 1] & 0xffffu);
                            R[c] = *(const nrt_f4 *)(sm + (f & 0xfffeu) + lg * 16u);
                            slowm |= (f & 1u) << c;
                        }
                        const unsigned pack = Bw[3];
                        if (__builtin_amdgcn_ballot_w64(slowm != 0)) {        // rare: a corner row that found no place in LDS
                            const unsigned id0 = pack & 0x3ffffffu;
                            const unsigned dx = ((pack >> 26) & 1u) ? SYZ : 0u, dy = ((pack >> 27) & 1u) ? (unsigned)a.S2 : 0u, dz = (pack >> 28) & 1u;
                            nrt_f4 G[8];
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const unsigned idc = id0 + ((c & 4) ? dx : 0u) + ((c & 2) ? dy : 0u) + ((c & 1) ? dz : 0u);
                                G[c] = lc_ld16c(volb, (idc << 7) + lg * 16u);
                            }
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3]), "+v"(G[4]), "+v"(G[5]), "+v"(G[6]), "+v"(G[7]) : : "memory");
#pragma unroll
                            for (int c = 0; c < 8; ++c) R[c] = ((slowm >> c) & 1u) ? G[c] : R[c];
                        }
                        const bool valid = (pack >> 29) & 1u, oob = (pack >> 30) & 1u;
                        const float w0x = __uint_as_float(Bw[0]), w0y = __uint_as_float(Bw[1]), w0z = __uint_as_float(Bw[2]);
                        const float w1x = nrt_sub(1.0f, w0x), w1y = nrt_sub(1.0f, w0y), w1z = nrt_sub(1.0f, w0z);
                        const nrt_f2 wy2 = {w0y, w1y}, wz2 = {w0z, w1z};
                        const nrt_f2 wxy0 = (nrt_f2){w0x, w0x} * wy2, wxy1 = (nrt_f2){w1x, w1x} * wy2;
                        nrt_f2 wt2[4];
                        wt2[0] = (nrt_f2){wxy0[0], wxy0[0]} * wz2;
                        wt2[1] = (nrt_f2){wxy0[1], wxy0[1]} * wz2;
                        wt2[2] = (nrt_f2){wxy1[0], wxy1[0]} * wz2;
                        wt2[3] = (nrt_f2){wxy1[1], wxy1[1]} * wz2;
                        nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float wt = wt2[c >> 1][c & 1];
                            const nrt_f2 w2 = {wt, wt};
                            al = al + w2 * (nrt_f2){R[c][0], R[c][1]};
                            ah = ah + w2 * (nrt_f2){R[c][2], R[c][3]};
                        }
                        nrt_f4 acc = {al[0], al[1], ah[0], ah[1]};
                        if (FILL) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = apply_fill(acc[c], oob, a.fill);
                        }
                        // a lane group past the volume's edge holds a copy of the edge voxel (same rows, same bits): its store rewrites
                        // that voxel with the same value (the number of stores per step stays constant for the s_waitcnt below); only
                        // the Dice sums must not count it twice
                        if (STORE) __builtin_nontemporal_store(acc, (nrt_f4 *)(outb + (size_t)row_off(t, s)));
                        if (DICE) {
                            const nrt_f4 T = F[s];
                            nrt_f2 pl = {acc[0], acc[1]}, ph = {acc[2], acc[3]}, tl = {T[0], T[1]}, th = {T[2], T[3]};
                            if (a.minmax) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    mnt = fminf(mnt, T[c]); mxt = fmaxf(mxt, T[c]);
                                    mnp = fminf(mnp, acc[c]); mxp = fmaxf(mxp, acc[c]);
                                }
                            }
                            // a copy's t and p are zeroed: x + 0 * 0 = x exactly, no branch
                            const nrt_f2 z2 = {0.0f, 0.0f};
                            pl = valid ? pl : z2; ph = valid ? ph : z2; tl = valid ? tl : z2; th = valid ? th : z2;
                            stp_l = stp_l + tl * pl; stp_h = stp_h + th * ph;
                            stt_l = stt_l + tl * tl; stt_h = stt_h + th * th;
                            spp_l = spp_l + pl * pl; spp_h = spp_h + ph * ph;
                        }
                    }
                };
                nrt_f4 F0[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, F1[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                load_fixed(0, F0);
                for (int t = 0; t < len; t += 2) {
                    lc_barrier();                                          // A(t)
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(F0[0]), "+v"(F0[1]) : "n"(STORE ? 2 : 0) : "memory");
                    load_fixed(min(t + 1, len - 1), F1);
                    blend(t, F0);
                    if (t + 1 >= len) break;
                    lc_barrier();                                          // A(t + 1)
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(F1[0]), "+v"(F1[1]) : "n"(STORE ? 2 : 0) : "memory");
                    load_fixed(min(t + 2, len - 1), F0);
                    blend(t + 1, F1);
                }
                asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            }
            lc_barrier();                                                  // A(len)
            if (DICE) {
                nrt_f4 stp = {stp_l[0], stp_l[1], stp_h[0], stp_h[1]}, stt = {stt_l[0], stt_l[1], stt_h[0], stt_h[1]},
                       spp = {spp_l[0], spp_l[1], spp_h[0], spp_h[1]};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    stp[c] = wave_xor_add(stp[c], 8);
                    stt[c] = wave_xor_add(stt[c], 8);
                    spp[c] = wave_xor_add(spp[c], 8);
                }
                const unsigned long long prow_g = (unsigned long long)b * per_batch + prow;
                if (l < 8) {
                    float *fp = a.fpart + prow_g * 96ull;
                    *(nrt_f4 *)(fp + 4 * l) = stp;
                    *(nrt_f4 *)(fp + 32 + 4 * l) = stt;
                    *(nrt_f4 *)(fp + 64 + 4 * l) = spp;
                }
                if (a.minmax) {
                    for (int off = 1; off < NRT_WAVE; off <<= 1) {
                        mnt = fminf(mnt, __shfl_xor(mnt, off, NRT_WAVE)); mxt = fmaxf(mxt, __shfl_xor(mxt, off, NRT_WAVE));
                        mnp = fminf(mnp, __shfl_xor(mnp, off, NRT_WAVE)); mxp = fmaxf(mxp, __shfl_xor(mxp, off, NRT_WAVE));
                    }
                }
                if (l == 0) *(nrt_f4 *)(a.mpart + prow_g * 4ull) = (nrt_f4){mnt, mxt, mnp, mxp};
            }
        }
    }
}

struct LcGeom {
    unsigned nTy, nTz, ncol, nseg, seglen;
    int lry, lrz;
};

void lc_geom(const int *O, int batch, int tune, LcGeom &g) {
    g.nTy = ((unsigned)O[1] + 3) / 4;
    g.nTz = ((unsigned)O[2] + 3) / 4;
    // the 128 waves an XCD runs together cover one region of 8 x 16 patches = 32 x 64 voxels
    g.lry = 3; g.lrz = 4;
    const unsigned RY = 1u << g.lry, RZ = 1u << g.lrz;
    g.ncol = ((g.nTy + RY - 1) / RY) * ((g.nTz + RZ - 1) / RZ) * RY * RZ;
    unsigned nseg = (unsigned)tune & 0xffu;
    if (nseg == 0) {
        // auto: every wave slot of the chip (1024) gets at least ~12 tasks, and a task is at most 64 planes
        nseg = (12288u + g.ncol * (unsigned)batch - 1) / (g.ncol * (unsigned)batch);
        const unsigned cap = ((unsigned)O[0] + 63) / 64;
        if (nseg < cap) nseg = cap;
    }
    if (nseg < ((unsigned)O[0] + 255) / 256) nseg = ((unsigned)O[0] + 255) / 256;   // the 16-bit allocation counter of a task
    if (nseg > (unsigned)O[0]) nseg = (unsigned)O[0];
    if (nseg < 1) nseg = 1;
    g.seglen = ((unsigned)O[0] + nseg - 1) / nseg;
    g.nseg = ((unsigned)O[0] + g.seglen - 1) / g.seglen;
}

template <int MODE, bool FILL, int DIAG>
void lc_launch_mode(const LcK &k, unsigned grid, bool dice, bool store, hipStream_t st) {
    if (dice && store) hipLaunchKernelGGL((gather_lc<MODE, true, true, FILL, DIAG>), dim3(grid), dim3(128), 0, st, k);
    else if (dice) hipLaunchKernelGGL((gather_lc<MODE, true, false, FILL, DIAG>), dim3(grid), dim3(128), 0, st, k);
    else hipLaunchKernelGGL((gather_lc<MODE, false, true, FILL, DIAG>), dim3(grid), dim3(128), 0, st, k);
}

template <bool FILL>
void lc_launch_fill(const LcK &k, unsigned grid, int mode, bool dice, bool store, hipStream_t st) {
    switch (mode) {
        case NRT_LOC_ABSOLUTE: lc_launch_mode<NRT_LOC_ABSOLUTE, FILL, 0>(k, grid, dice, store, st); break;
        case NRT_LOC_SHIFT: lc_launch_mode<NRT_LOC_SHIFT, FILL, 0>(k, grid, dice, store, st); break;
        default: lc_launch_mode<NRT_LOC_LINSPACE, FILL, 0>(k, grid, dice, store, st); break;
    }
}

}  // namespace

bool nrt_lc_supported(const int *S, const int *O, int channels) {
    if (channels != 32) return false;
    for (int d = 0; d < 3; ++d) if (S[d] < 1 || O[d] < 1) return false;
    const unsigned long long nin = (unsigned long long)S[0] * S[1] * S[2], nout = (unsigned long long)O[0] * O[1] * O[2];
    if (nin >= (1ull << 25) || nout >= (1ull << 25)) return false;                 // row byte offsets (id << 7) are 32-bit; id in 26 bits
    if ((long long)S[0] * S[1] >= (1 << 24) || S[2] >= (1 << 24) || (long long)O[0] * O[1] >= (1 << 24) || O[2] >= (1 << 24)) return false;
    const unsigned long long hi = (unsigned long long)((S[0] + 3) / 4) * ((S[1] + 15) / 16) * ((S[2] + 15) / 16);
    return hi < 0x7fffull;                                                          // 15-bit tag; 0x7fff = empty
}

unsigned nrt_lc_rows(const int *O, int batch, int tune) {
    LcGeom g;
    lc_geom(O, batch, tune, g);
    return g.ncol * g.nseg;
}

int nrt_lc_launch(const LcCall &c, hipStream_t st) {
    if (!nrt_lc_supported(c.S, c.O, 32)) return NRT_ERR_UNSUPPORTED;
    const bool dice = c.fixed != nullptr, store = c.out != nullptr;
    if (!dice && !store) return NRT_ERR_INVALID_ARG;
    LcGeom g;
    lc_geom(c.O, c.batch, c.tune, g);
    LcK k;
    k.vol = (const char *)c.vol; k.loc = (const char *)c.loc; k.fixed = (const char *)c.fixed; k.out = (char *)c.out;
    k.fpart = c.fpart; k.mpart = c.mpart;
    k.S0 = c.S[0]; k.S1 = c.S[1]; k.S2 = c.S[2]; k.O0 = c.O[0]; k.O1 = c.O[1]; k.O2 = c.O[2];
    k.d0 = c.delta[0]; k.d1 = c.delta[1]; k.d2 = c.delta[2];
    k.vol_bs = (unsigned long long)c.vol_bs * 4ull; k.loc_bs = (unsigned long long)c.loc_bs * 4ull; k.out_bs = (unsigned long long)c.out_bs * 4ull;
    k.has_fill = c.has_fill; k.fill = c.fill;
    k.nTy = g.nTy; k.nTz = g.nTz; k.ncol = g.ncol; k.nseg = g.nseg; k.seglen = g.seglen; k.nbatch = (unsigned)c.batch;
    k.lry = g.lry; k.lrz = g.lrz;
    k.nyh = ((unsigned)c.S[1] + 15) / 16; k.nzh = ((unsigned)c.S[2] + 15) / 16;
    k.ntask = g.ncol * g.nseg * (unsigned)c.batch;
    k.minmax = c.minmax;
    // persistent waves: 4 per CU (LDS), 256 CUs; fewer when there are fewer tasks
    unsigned grid = 1024;
    if (k.ntask < grid) grid = NRT_NXCD * ((k.ntask + NRT_NXCD - 1) / NRT_NXCD);
    const int diag = (c.tune >> 8) & 3;
    if (diag == 1) {                    // diagnostic build of the data path (no tag protocol; results are NOT the warp)
        if (c.mode != NRT_LOC_SHIFT || c.has_fill) return NRT_ERR_UNSUPPORTED;
        lc_launch_mode<NRT_LOC_SHIFT, false, 1>(k, grid, dice, store, st);
    } else if (c.has_fill) lc_launch_fill<true>(k, grid, c.mode, dice, store, st);
    else lc_launch_fill<false>(k, grid, c.mode, dice, store, st);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
