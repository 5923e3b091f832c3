// LDS-row-cache gather for 32 float channels on gfx950 (MI355X): linear interpn / SpatialTransformer, optionally fused with
// the soft-Dice sums.  Reference op: neurite/tf/utils/utils.py:137-191 (+ metrics.py:415-482 for the Dice form).
//
// Why: the register kernels (interpn.hip, fused.hip) pull 8 corner rows + 1 fixed row = 1152 B per voxel through the
// texture path (TA -> L1 -> VGPR, 64 B/clk/CU); of those only ~2.3 rows are HBM misses, the rest are L1 / L2 hits that still
// occupy the path (profiles/archive/r02_lab: hits and misses issued by one CU cost additively).  On the SURVEY 8d field a voxel shares
// its corner rows with its (y,z) neighbours and with the next x plane: an x-marching 4 x 8 patch needs only ~2.0 NEW rows per
// voxel.  So a workgroup keeps a software cache of source rows in LDS and the texture path carries each row once:
//
//   * one workgroup (6 waves, 80 KB of LDS, two per CU) = one 4 x 8 (y,z) patch marching along x; a step is one x plane
//     (32 voxels);
//   * waves 0 and 1 each manage the source rows of one x parity (a voxel's two x corners have different parities): a 256-row
//     FIFO ring (128 B rows), a 512-entry direct-mapped tag table (hash = low bits of the row's x / 2, y, z) and a fetch list.
//     MGMT(t+3): lane = (voxel, y corner) looks its two z-corner rows up in the tag table; misses are claimed through the table
//     (write token, read back: one owner per distinct row), owners take consecutive ring slots (ballot + mbcnt) and append the
//     row to the fetch list; the slots and the weights go to one of four hand-off buffers.  These waves touch LDS only, and the
//     location arithmetic of step t+4 is interleaved with the LDS round trips of the protocol of step t+3;
//   * waves 2-5, ISSUE(t+2): each fetches a quarter of both lists by global_load_lds_dwordx4 (8 rows per instruction, straight
//     into the rings), and waits for its own loads of step t before the barrier of step t;
//     BLEND(t): 8 lanes per voxel read the 8 corner rows from LDS (ds_read_b128) and run the reference's op sequence
//     (bit-identical to interpn.hip); the fixed row comes from registers loaded two steps ahead;
//   * one s_barrier per step: A(t) = "rows and hand-off of step t are in LDS, the blend of step t - 1 is done".
//   A ring row may be overwritten only when no step that still has to blend needs it: a step may allocate at most
//   256 - (age of the oldest row that it or the two steps before it hit) rows; what does not fit (a few % of the steps on the
//   8d field) is loaded directly by the blend ("slow" corners).
//   All vector-memory loads are inline asm: the compiler's s_waitcnt model would otherwise serialise against the LDS-DMA.
#include <stdio.h>
#include <stdlib.h>

#include "dice_reduce.h"
#include "interpn_core.h"
#include "lc.h"

namespace {

constexpr int LC_NR = 256;            // ring rows per management wave (a power of two: ring slot = allocation counter & 255)
constexpr int LC_WIN = 96;            // a cached row counts as a hit while it is younger than this many allocations
constexpr int LC_CAP = 120;           // ring rows a wave may allocate per step (its list holds 128 entries incl. padding)
constexpr unsigned LC_TAB = 2 * LC_NR * 128;               // byte offsets inside the workgroup's LDS: two tag tables of 512 entries
constexpr unsigned LC_HAND = LC_TAB + 2 * 2048;            // 4 buffers x 32 voxels x 32 B
constexpr unsigned LC_LIST = LC_HAND + 4096;               // fetch lists [2 buffers][2 parities][128 entries]
constexpr unsigned LC_CTL = LC_LIST + 2048;                // [2 buffers][2 parities] {rows, first ring slot}
constexpr unsigned LC_SHIFT = LC_CTL + 64;                 // 4 buffers x 96 floats
constexpr unsigned LC_DUMMY = LC_SHIFT + 4 * 384;          // 128 dwords
constexpr unsigned LC_LDS = LC_DUMMY + 512;
static_assert(LC_LDS <= 81920, "two workgroups per CU");

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
    int strict;      // diagnostic: every wait of the fetching waves drains all of their loads
};

typedef unsigned nrt_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lc_dma16(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lc_dma4(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ nrt_f4 lc_ld16(const void *base, unsigned voff) {          // streamed once (fixed rows)
    nrt_f4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(voff), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ nrt_f4 lc_ld16c(const void *base, unsigned voff) {         // cached (source rows)
    nrt_f4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ unsigned lc_mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ unsigned lc_uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned lc_sel(bool c, unsigned x, unsigned y) { return c ? x : y; }
// workgroup barrier that also publishes this wave's LDS writes / retires its LDS reads (no vmcnt: the waves time their own loads)
__device__ __forceinline__ void lc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }
// wait until at most k of this wave's vector-memory instructions are outstanding (they retire in order)
__device__ __forceinline__ void lc_wait_vm(unsigned k) {
#define LC_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" : : : "memory"); break;
    switch (k) {
        LC_W(0) LC_W(1) LC_W(2) LC_W(3) LC_W(4) LC_W(5) LC_W(6) LC_W(7) LC_W(8) LC_W(9) LC_W(10) LC_W(11) LC_W(12) LC_W(13) LC_W(14) LC_W(15)
        LC_W(16) LC_W(17) LC_W(18) LC_W(19) LC_W(20) LC_W(21) LC_W(22) LC_W(23) LC_W(24) LC_W(25) LC_W(26) LC_W(27) LC_W(28) LC_W(29) LC_W(30)
        LC_W(31) LC_W(32) LC_W(33) LC_W(34) LC_W(35) LC_W(36) LC_W(37) LC_W(38) LC_W(39) LC_W(40)
        default: asm volatile("s_waitcnt vmcnt(40)" : : : "memory"); break;      // fewer outstanding than allowed is only stricter
    }
#undef LC_W
}

// DIAG == 2: per-phase shader clocks summed over all workgroups (read back and printed by the launcher; tools/lc_check.py phases)
__device__ unsigned long long lc_dbg[16];
#define LC_T(var) do { if (DIAG == 2) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg_[var] += now_ - last_; last_ = now_; } } while (0)

// what MGMT's location arithmetic hands to its protocol (one step ahead, in registers)
struct LcPrep {
    unsigned id[2], tpo[2], idhi[2], tok[2];     // row, byte offset of its tag, tag bits 17..31, the claim this lane would write
    unsigned id0flags, hoff0, hoff1;             // row of corner 0 | flags; byte offsets of the hand-off dwords (or the dummy)
    float w0x, w0y, w0z;
    bool mine;
};

// MODE = location mode; DICE = accumulate the soft-Dice sums against `fixed`; STORE = write the warped rows; FILL = fill_value
// given.  DIAG: 0 = product; 1 = no tag protocol (every corner "hits" slot id % 256, 32 arbitrary rows fetched per wave and step):
// the cost of the data path alone; 2 = product + phase clocks (profiles/archive/r03_lc).
template <int MODE, bool DICE, bool STORE, bool FILL, int DIAG>
__global__ __launch_bounds__(384, 3) void gather_lc(LcK a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const unsigned l = threadIdx.x & 63u;
    const unsigned wv = lc_uni(threadIdx.x >> 6);
    const unsigned lds0 = lc_uni((unsigned)(size_t)sm);

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
        const int x0 = (int)(useg * a.seglen), y0 = (int)cy * 4, z0 = (int)cz * 8;
        int len = min((int)a.seglen, a.O0 - x0);
        if (cy >= a.nTy || cz >= a.nTz) len = 0;
        const char *volb = a.vol + (unsigned long long)b * a.vol_bs;

        // barriers of a task (every wave): P1 (locations of steps 0..3 staged), P2 (lists of steps 0, 1 written), P3 (... and read),
        // A(0) .. A(len - 1), A(len)
        if (wv < 2) {
            // ================================ waves 0, 1: cache management of the rows with x parity P (LDS only) ================================
            const unsigned P = wv;
            unsigned *const tab = (unsigned *)(sm + LC_TAB + P * 2048u);
            const unsigned dummy_o = LC_DUMMY + (P * 64u + l) * 4u;                   // where lanes with nothing to write write (no branch)
            unsigned *const dummy = (unsigned *)(sm + dummy_o);
            const unsigned rbase = P * (unsigned)LC_NR;               // first ring row of this wave
            // lane = (voxel mv of the plane, y corner myc): the two z corners of the x corner whose source plane has parity P
            const unsigned mv = l & 31, myy = mv >> 3, mzz = mv & 7, myc = l >> 5;
            unsigned H = 0;                                            // allocation counter (16 bits used); ring slot = H & 255
            unsigned A1 = 0, H1 = 0, A2 = 0, H2 = 0;                   // oldest hit of the previous two steps (age, counter then)
            // lanes past the volume's edge hold a copy of the edge voxel (clamped coordinates)
            const int myq = min(y0 + (int)myy, a.O1 - 1), mzq = min(z0 + (int)mzz, a.O2 - 1);
            const bool mvalid = (y0 + (int)myy < a.O1) && (z0 + (int)mzz < a.O2);

            // ---- PREP(t): location -> corners, weights, row ids, tag addresses (no protocol state involved) ----
            auto prep = [&](int t, LcPrep &q) {
                const int qd[3] = {min(x0 + t, a.O0 - 1), myq, mzq};       // (a step past the end reads whatever its location buffer holds)
                const int Sd[3] = {a.S0, a.S1, a.S2}, Od[3] = {a.O0, a.O1, a.O2};
                const float dd[3] = {a.d0, a.d1, a.d2};
                const float *shp = (const float *)(sm + LC_SHIFT + ((unsigned)t & 3u) * 384u) + 3u * mv;
                float p[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (MODE == NRT_LOC_ABSOLUTE) p[d] = shp[d];
                    else if (MODE == NRT_LOC_SHIFT) p[d] = nrt_add((float)qd[d], shp[d]);
                    else p[d] = (qd[d] == 0) ? 0.0f : ((qd[d] == Od[d] - 1) ? (float)(Sd[d] - 1) : nrt_mul(dd[d], (float)qd[d]));
                }
                int i0x, i1x, i0y, i1y, i0z, i1z;
                float w1x, w1y, w1z;
                corner_1d(p[0], a.S0, i0x, i1x, q.w0x, w1x);
                corner_1d(p[1], a.S1, i0y, i1y, q.w0y, w1y);
                corner_1d(p[2], a.S2, i0z, i1z, q.w0z, w1z);
                bool oob = false;
                if (FILL) oob = (p[0] < 0.0f) || (p[0] > (float)(a.S0 - 1)) || (p[1] < 0.0f) || (p[1] > (float)(a.S1 - 1)) ||
                                (p[2] < 0.0f) || (p[2] > (float)(a.S2 - 1));
                // the x corner of parity P: corner 0 if i0x has it, else corner 1; at the clamped border (i1x == i0x) one wave owns
                // both x corners and the other none
                const bool c0mine = (((unsigned)i0x ^ P) & 1u) == 0u, both = i1x == i0x;
                q.mine = c0mine || !both;
                const unsigned mxc = c0mine ? 0u : 1u;
                const unsigned sx = c0mine ? (unsigned)i0x : (unsigned)i1x;
                const unsigned sy = myc ? (unsigned)i1y : (unsigned)i0y;
                const unsigned rowxy = nrt_mad24(sx, (unsigned)a.S1, sy);
                const unsigned hxy = ((sx & 2) << 7) | ((sy & 15) << 4), hixy = ((sx >> 2) * a.nyh + (sy >> 4)) * a.nzh;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned sz = j ? (unsigned)i1z : (unsigned)i0z;
                    q.id[j] = nrt_mad24(rowxy, (unsigned)a.S2, sz);
                    q.tpo[j] = (hxy | (sz & 15)) << 2;
                    q.idhi[j] = (hixy + (sz >> 4)) << 17;                       // kept in place: bits 17..31 of a tag
                    q.tok[j] = q.idhi[j] | 0x10000u | (l << 1) | (unsigned)j;
                }
                const unsigned id0 = nrt_mad24(nrt_mad24((unsigned)i0x, (unsigned)a.S1, (unsigned)i0y), (unsigned)a.S2, (unsigned)i0z);
                q.id0flags = id0 | ((unsigned)(i1x != i0x) << 26) | ((unsigned)(i1y != i0y) << 27) | ((unsigned)(i1z != i0z) << 28) |
                             ((unsigned)mvalid << 29) | ((unsigned)oob << 30);
                const unsigned hb = LC_HAND + ((unsigned)t & 3u) * 1024u + mv * 32u;
                const unsigned o0 = hb + (2 * mxc + myc) * 4u, o1 = hb + (2 * (1 - mxc) + myc) * 4u;
                q.hoff0 = q.mine ? o0 : dummy_o;
                q.hoff1 = (q.mine && both) ? o1 : dummy_o;
            };

            // ---- PROTO(t): tag lookups, claims, slot allocation, fetch list, hand-off ----
            auto proto = [&](int t, bool live, const LcPrep &q) {
                unsigned *const list = (unsigned *)(sm + LC_LIST + (((unsigned)t & 1u) * 2u + P) * 512u);
                unsigned *tp[2] = {(unsigned *)((unsigned char *)tab + q.tpo[0]), (unsigned *)((unsigned char *)tab + q.tpo[1])};
                unsigned slot[2], slow[2];
                H = (H + 7u) & ~7u;                                   // ring allocations of a step start on a multiple of 8 rows
                unsigned nr;
                if (DIAG == 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) { slot[j] = q.id[j] & 255u; slow[j] = 0; }
                    *(l < 32 ? &list[l] : dummy) = q.id[0];
                    nr = live ? 32u : 0u;
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
                        const bool hit = ((e[j] ^ q.idhi[j]) >> 16) == 0u && ag < (unsigned)LC_WIN;     // same row, a ring tag, young
                        miss[j] = q.mine && !hit;
                        age = max(age, lc_sel(hit && q.mine, ag, 0u));
                        slot[j] = e[j] & 255u;
                    }
                    // 2. the oldest row that this step, or the two before it (not blended yet), still reads bounds the allocation
                    age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0xB1, 0xF, 0xF, false));    // lane ^ 1
                    age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0x4E, 0xF, 0xF, false));    // lane ^ 2
                    age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0x141, 0xF, 0xF, false));   // row_half_mirror
                    age = max(age, (unsigned)__builtin_amdgcn_update_dpp((int)age, (int)age, 0x140, 0xF, 0xF, false));   // row_mirror
                    const unsigned amax = max(max((unsigned)__builtin_amdgcn_readlane((int)age, 0), (unsigned)__builtin_amdgcn_readlane((int)age, 16)),
                                              max((unsigned)__builtin_amdgcn_readlane((int)age, 32), (unsigned)__builtin_amdgcn_readlane((int)age, 48)));
                    const unsigned keep = max(amax, max(A1 + (H - H1), A2 + (H - H2)));
                    const unsigned room = keep >= (unsigned)LC_NR ? 0u : (unsigned)LC_NR - keep;
                    const unsigned limit = min(room, (unsigned)LC_CAP) & ~7u;    // whole 8-row instructions (their padding rows are written too)
                    A2 = A1; H2 = H1; A1 = amax; H1 = H;
                    // 3. claim: one owner per distinct missing row
#pragma unroll
                    for (int j = 0; j < 2; ++j) *(miss[j] ? tp[j] : dummy) = q.tok[j];
                    unsigned e2[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) e2[j] = *tp[j];
                    bool owner[2], follower[2], fetcher[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool same = ((e2[j] ^ q.idhi[j]) >> 17) == 0u;
                        owner[j] = miss[j] && e2[j] == q.tok[j];
                        follower[j] = miss[j] && same && !owner[j];
                        fetcher[j] = miss[j] && !follower[j];         // owners, and rows that lost their table entry to another row
                    }
                    // 4. consecutive slots for everything that is fetched
                    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(fetcher[0]), b1 = __builtin_amdgcn_ballot_w64(fetcher[1]);
                    const unsigned c0 = (unsigned)__builtin_popcountll(b0), ntot = c0 + (unsigned)__builtin_popcountll(b1);
                    const unsigned pos[2] = {lc_mbcnt(b0), c0 + lc_mbcnt(b1)};
                    nr = live ? min(ntot, limit) : 0u;
                    // pad the list to whole instructions with a row of this step (overwritten below where a real entry exists)
                    *(l < 8 ? &list[(nr & ~7u) + l] : dummy) = q.id[0];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool ring = fetcher[j] && pos[j] < nr;
                        const unsigned seq = H + pos[j];
                        slot[j] = lc_sel(ring, seq & 255u, slot[j]);
                        slow[j] = (fetcher[j] && !ring) ? 1u : 0u;
                        *(ring ? &list[pos[j]] : dummy) = q.id[j];
                        *(owner[j] ? tp[j] : dummy) = lc_sel(ring, q.idhi[j] | (seq & 0xffffu), ~0u);
                    }
                    // 5. rows another lane fetches
                    unsigned e3[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) e3[j] = *tp[j];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool same = ((e3[j] ^ q.idhi[j]) >> 16) == 0u;        // the owner got a ring slot
                        slot[j] = lc_sel(follower[j], e3[j] & 255u, slot[j]);
                        slow[j] = lc_sel(follower[j] && !same, 1u, slow[j]);
                    }
                }
                // hand-off: dword 2 xc + yc = the lane's two rows as (row << 1 | slow) 16-bit fields; dwords 4..7 (wave 0) = weights,
                // row of corner 0, flags
                const unsigned hv = (((rbase + slot[0]) << 1) | slow[0]) | ((((rbase + slot[1]) << 1) | slow[1]) << 16);
                *(unsigned *)(sm + q.hoff0) = hv;
                *(unsigned *)(sm + q.hoff1) = hv;
                if (P == 0)
                    *(nrt_u4 *)(sm + LC_HAND + ((unsigned)t & 3u) * 1024u + mv * 32u + 16u) =
                        (nrt_u4){__float_as_uint(q.w0x), __float_as_uint(q.w0y), __float_as_uint(q.w0z), q.id0flags};
                if (l == 0) *(uint2 *)(sm + LC_CTL + (((unsigned)t & 1u) * 2u + P) * 8u) = make_uint2(nr, H & 255u);
                H += nr;
            };

            unsigned long long dbg_[4] = {0, 0, 0, 0}, last_ = (DIAG == 2) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (DIAG != 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) ((nrt_u4 *)tab)[i * 64 + l] = (nrt_u4){~0u, ~0u, ~0u, ~0u};
            }
            lc_barrier();                                            // P1
            LcPrep qa, qb;
            prep(0, qa); proto(0, len > 0, qa);
            prep(1, qa); proto(1, 1 < len, qa);
            lc_barrier();                                            // P2
            prep(2, qa);
            lc_barrier();                                            // P3
            proto(2, 2 < len, qa);
            prep(3, qa);                                             // (the locations of step 3 were staged before P1)
            LC_T(3);
            for (int t = 0; t < len; t += 2) {
                lc_barrier();                                        // A(t)
                LC_T(1);
                proto(t + 3, t + 3 < len, qa);
                prep(t + 4, qb);
                LC_T(3);
                if (t + 1 >= len) break;
                lc_barrier();                                        // A(t + 1)
                LC_T(1);
                proto(t + 4, t + 4 < len, qb);
                prep(t + 5, qa);
                LC_T(3);
            }
            lc_barrier();                                            // A(len): the blend of the last step is done
            if (DIAG == 2 && l == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(&lc_dbg[P * 4 + i], dbg_[i]);
            }
        } else {
            // ================================ waves 2..5: fetches, and the blend of the plane's row y0 + (wv - 2) ================================
            const char *fixb = DICE ? a.fixed + (unsigned long long)b * a.out_bs : a.vol;
            const char *locb = a.loc ? a.loc + (unsigned long long)b * a.loc_bs : a.vol;
            char *outb = STORE ? a.out + (unsigned long long)b * a.out_bs : nullptr;
            const unsigned bw = wv - 2u;
            const unsigned bvv = l >> 3, lg = l & 7;        // lane group bvv = voxel z0 + bvv; channels 4 lg .. 4 lg + 3
            const unsigned SYZ = (unsigned)a.S1 * (unsigned)a.S2;
            nrt_f2 stp_l = {0, 0}, stp_h = {0, 0}, stt_l = {0, 0}, stt_h = {0, 0}, spp_l = {0, 0}, spp_h = {0, 0};
            float mnt = INFINITY, mxt = -INFINITY, mnp = INFINITY, mxp = -INFINITY;
            const unsigned yy = bw, vox = yy * 8u + bvv;
            const unsigned qyz = nrt_mad24((unsigned)min(y0 + (int)yy, a.O1 - 1), (unsigned)a.O2, (unsigned)min(z0 + (int)bvv, a.O2 - 1));
            const unsigned OYZ = (unsigned)a.O1 * (unsigned)a.O2;
            auto row_off = [&](int t) -> unsigned { return ((nrt_mad24((unsigned)min(x0 + t, a.O0 - 1), OYZ, qyz)) * 8u + lg) * 16u; };
            auto load_fixed = [&](int t, nrt_f4 &F) -> unsigned { if (DICE) { F = lc_ld16(fixb, row_off(t)); return 1u; } return 0u; };
            // locations of step t -> LDS (96 floats, voxel-major): two dword DMAs (wave 2)
            auto issue_shift = [&](int t) -> unsigned {
                if (MODE == NRT_LOC_LINSPACE || bw != 0) return 0u;
                const unsigned xq = (unsigned)min(x0 + t, a.O0 - 1);
                const unsigned dst = lds0 + LC_SHIFT + ((unsigned)t & 3u) * 384u;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned e = min(64u * i + l, 95u), ev = e / 3u, ec = e - 3u * ev;      // element -> voxel, component
                    const int yq = min(y0 + (int)(ev >> 3), a.O1 - 1), zq = min(z0 + (int)(ev & 7), a.O2 - 1);
                    const unsigned q = nrt_mad24(nrt_mad24(xq, (unsigned)a.O1, (unsigned)yq), (unsigned)a.O2, (unsigned)zq);
                    if (i == 0 || l < 32) lc_dma4(locb, (nrt_times3(q) + ec) << 2, dst + 256u * i);
                }
                return 2u;
            };
            // ---- ISSUE(t): this wave's quarter of both fetch lists of step t into the rings.  Whole instructions only: the lists are
            // padded to a multiple of 8 rows with the address of a row that is fetched anyway, and the rows of the padding land in the
            // ring slots that the 8-row alignment of a step leaves unused.  Returns the number of instructions issued ----
            auto issue_rows = [&](int t) -> unsigned {
                unsigned n = 0;
#pragma unroll
                for (unsigned P = 0; P < 2; ++P) {
                    const unsigned lb = ((unsigned)t & 1u) * 2u + P;
                    const uint2 ctl = *(const uint2 *)(sm + LC_CTL + lb * 8u);
                    const unsigned nr = lc_uni(ctl.x), hp = lc_uni(ctl.y);
                    const unsigned *list = (const unsigned *)(sm + LC_LIST + lb * 512u);
                    unsigned v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = list[8 * (4 * i + bw) + bvv];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned first = 8u * (4u * i + bw);
                        if (first < nr) {
                            lc_dma16(volb, (v[i] << 7) + lg * 16u, lds0 + (P * (unsigned)LC_NR + ((hp + first) & 255u)) * 128u);
                            ++n;
                        }
                    }
                }
                return n;
            };
            // BLEND(t): 8 lanes per voxel read the 8 corner rows from LDS and run the reference's op sequence
            auto blend = [&](int t, const nrt_f4 &T) {
                const unsigned char *hb = sm + LC_HAND + ((unsigned)t & 3u) * 1024u + vox * 32u;
                const nrt_u4 A = *(const nrt_u4 *)hb, Bw = *(const nrt_u4 *)(hb + 16);
                nrt_f4 R[8];
                unsigned slowm = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // corner c = 4 xc + 2 yc + zc: dword 2 xc + yc, half zc
                    const unsigned f = (c & 1) ? (A[c >> 1] >> 16) : (A[c >> 1] & 0xffffu);
                    R[c] = *(const nrt_f4 *)(sm + ((f & 0xfffeu) << 6) + lg * 16u);
                    slowm |= (f & 1u) << c;
                }
                const unsigned pack = Bw[3];
                if (__builtin_amdgcn_ballot_w64(slowm != 0)) {        // rare: a corner row that found no place in LDS
                    if (DIAG == 2 && l == 0) atomicAdd(&lc_dbg[12], 1ull);
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
                if (STORE) __builtin_nontemporal_store(acc, (nrt_f4 *)(outb + (size_t)row_off(t)));
                if (DICE) {
                    nrt_f2 pl = {acc[0], acc[1]}, ph = {acc[2], acc[3]}, tl = {T[0], T[1]}, th = {T[2], T[3]};
                    if (a.strict & 2) { tl = (nrt_f2){1.0f, 1.0f}; th = tl; }      // diagnostic: count voxels
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
            };

            unsigned long long dbg_[4] = {0, 0, 0, 0}, last_ = (DIAG == 2) ? __builtin_amdgcn_s_memtime() : 0ull;
            // start-up: locations of steps 0..3 (-> P1), then the fetches of steps 0 and 1 (after P2), locations of steps 4, 5
            for (int s = 0; s < 4; ++s) if (s < len) (void)issue_shift(s);
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            lc_barrier();                                            // P1
            lc_barrier();                                            // P2: the lists of steps 0 and 1 are written
            // the fixed row is loaded two steps ahead (three registers in rotation).  This wave's loads retire in order: before A(t) it
            // waits for everything but what it issued in iteration t - 1 (plus, with STORE, the row it stored in iteration t - 1)
            nrt_f4 F0 = {0, 0, 0, 0}, F1 = {0, 0, 0, 0}, F2 = {0, 0, 0, 0};
            unsigned kprev = 0;
            if (len > 0) {
                (void)load_fixed(0, F0);
                (void)issue_rows(0);
                if (4 < len) (void)issue_shift(4);
                kprev = load_fixed(min(1, len - 1), F1);
                kprev += issue_rows(1);
                if (5 < len) kprev += issue_shift(5);
            }
            lc_barrier();                                            // P3: ... and read
            for (int t = 0; t < len; t += 3) {
                // ---- step t ----
                lc_wait_vm((a.strict & 1) ? 0u : kprev + (STORE && t >= 1 ? 1u : 0u));
                LC_T(1);
                lc_barrier();                                          // A(t)
                LC_T(0);
                asm volatile("" : "+v"(F0) : : "memory");
                kprev = (t + 6 < len) ? issue_shift(t + 6) : 0u;
                kprev += load_fixed(min(t + 2, len - 1), F2);
                kprev += issue_rows(t + 2);
                LC_T(3);
                blend(t, F0);
                LC_T(2);
                if (t + 1 >= len) break;
                // ---- step t + 1 ----
                lc_wait_vm((a.strict & 1) ? 0u : kprev + (STORE ? 1u : 0u));
                LC_T(1);
                lc_barrier();                                          // A(t + 1)
                LC_T(0);
                asm volatile("" : "+v"(F1) : : "memory");
                kprev = (t + 7 < len) ? issue_shift(t + 7) : 0u;
                kprev += load_fixed(min(t + 3, len - 1), F0);
                kprev += issue_rows(t + 3);
                LC_T(3);
                blend(t + 1, F1);
                LC_T(2);
                if (t + 2 >= len) break;
                // ---- step t + 2 ----
                lc_wait_vm((a.strict & 1) ? 0u : kprev + (STORE ? 1u : 0u));
                LC_T(1);
                lc_barrier();                                          // A(t + 2)
                LC_T(0);
                asm volatile("" : "+v"(F2) : : "memory");
                kprev = (t + 8 < len) ? issue_shift(t + 8) : 0u;
                kprev += load_fixed(min(t + 4, len - 1), F1);
                kprev += issue_rows(t + 4);
                LC_T(3);
                blend(t + 2, F2);
                LC_T(2);
            }
            // drain: the registers of loads that nobody consumes any more (the look-ahead of the last steps) must stay reserved until
            // the data has landed, or a late load would overwrite whatever the compiler put there
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(F0), "+v"(F1), "+v"(F2) : : "memory");
            lc_barrier();                                                  // A(len)
            if (DIAG == 2 && l == 0 && wv == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(&lc_dbg[8 + i], dbg_[i]);
            }
            if (DICE) {
                nrt_f4 stp = {stp_l[0], stp_l[1], stp_h[0], stp_h[1]}, stt = {stt_l[0], stt_l[1], stt_h[0], stt_h[1]},
                       spp = {spp_l[0], spp_l[1], spp_h[0], spp_h[1]};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    stp[c] = wave_xor_add(stp[c], 8);
                    stt[c] = wave_xor_add(stt[c], 8);
                    spp[c] = wave_xor_add(spp[c], 8);
                }
                const unsigned long long prow_g = ((unsigned long long)b * per_batch + prow) * 4ull + bw;    // four partial rows per task
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
    g.nTz = ((unsigned)O[2] + 7) / 8;
    // the 64 workgroups an XCD runs together cover one or two regions of patches: the region shape with the least padding
    // (padding patches are empty tasks that unbalance the persistent workgroups), larger regions first
    static const int cand[][2] = {{3, 3}, {4, 2}, {2, 4}, {3, 2}, {2, 3}, {2, 2}, {3, 1}, {1, 3}, {2, 1}, {1, 2}, {1, 1}, {0, 0}};
    unsigned best = ~0u;
    for (const auto &c : cand) {
        const unsigned RY = 1u << c[0], RZ = 1u << c[1];
        const unsigned n = ((g.nTy + RY - 1) / RY) * ((g.nTz + RZ - 1) / RZ) * RY * RZ;
        if (n < best) { best = n; g.lry = c[0]; g.lrz = c[1]; g.ncol = n; }
    }
    unsigned nseg = (unsigned)tune & 0xffu;
    if (nseg == 0) {
        // auto: every workgroup slot of the chip (512) gets at least ~12 tasks, and a task is at most 96 planes
        nseg = (6144u + g.ncol * (unsigned)batch - 1) / (g.ncol * (unsigned)batch);
        const unsigned cap = ((unsigned)O[0] + 95) / 96;
        if (nseg < cap) nseg = cap;
    }
    if (nseg < ((unsigned)O[0] + 383) / 384) nseg = ((unsigned)O[0] + 383) / 384;   // the 16-bit allocation counter of a task (<= 128 per step)
    if (nseg > (unsigned)O[0]) nseg = (unsigned)O[0];
    if (nseg < 1) nseg = 1;
    g.seglen = ((unsigned)O[0] + nseg - 1) / nseg;
    g.nseg = ((unsigned)O[0] + g.seglen - 1) / g.seglen;
}

template <int MODE, bool DICE, bool STORE, bool FILL, int DIAG>
void lc_launch_one(const LcK &k, unsigned grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)gather_lc<MODE, DICE, STORE, FILL, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC_LDS);
        attr = true;
    }
    hipLaunchKernelGGL((gather_lc<MODE, DICE, STORE, FILL, DIAG>), dim3(grid), dim3(384), LC_LDS, st, k);
}

template <int MODE, bool FILL, int DIAG>
void lc_launch_mode(const LcK &k, unsigned grid, bool dice, bool store, hipStream_t st) {
    if (dice && store) lc_launch_one<MODE, true, true, FILL, DIAG>(k, grid, st);
    else if (dice) lc_launch_one<MODE, true, false, FILL, DIAG>(k, grid, st);
    else lc_launch_one<MODE, false, true, FILL, DIAG>(k, grid, st);
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
    return g.ncol * g.nseg * 4u;             // four blend waves per task, one partial row each
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
    k.strict = (c.tune >> 10) & 3;
    // persistent workgroups: 2 per CU (LDS), 256 CUs; fewer when there are fewer tasks
    unsigned grid = 512;
    if (k.ntask < grid) grid = NRT_NXCD * ((k.ntask + NRT_NXCD - 1) / NRT_NXCD);
    const int diag = (c.tune >> 8) & 3;
    if (diag == 1) {                    // diagnostic build of the data path (no tag protocol; results are NOT the warp)
        if (c.mode != NRT_LOC_SHIFT || c.has_fill) return NRT_ERR_UNSUPPORTED;
        lc_launch_mode<NRT_LOC_SHIFT, false, 1>(k, grid, dice, store, st);
    } else if (diag == 2) {             // product kernel + phase clocks (synchronous; prints to stderr)
        if (c.mode != NRT_LOC_SHIFT || c.has_fill) return NRT_ERR_UNSUPPORTED;
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lc_dbg), z, sizeof(z));
        lc_launch_mode<NRT_LOC_SHIFT, false, 2>(k, grid, dice, store, st);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(lc_dbg), sizeof(z));
        const double steps = (double)g.seglen * g.nseg * g.nTy * g.nTz * c.batch;      // workgroup steps in the launch
        fprintf(stderr, "lc phases (s_memtime ticks per step): mgmt0 barrier %.0f mgmt %.0f | mgmt1 barrier %.0f mgmt %.0f | "
                        "wave 2: barrier %.0f wait %.0f issue %.0f blend %.0f | slow sub-passes per step %.3f\n",
                z[1] / steps, z[3] / steps, z[5] / steps, z[7] / steps, z[8] / steps, z[9] / steps, z[11] / steps, z[10] / steps, z[12] / steps);
    } else if (c.has_fill) lc_launch_fill<true>(k, grid, c.mode, dice, store, st);
    else lc_launch_fill<false>(k, grid, c.mode, dice, store, st);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
