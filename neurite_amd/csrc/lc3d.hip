// LocallyConnected3D (implementation 1) for gfx950 (MI355X).
//
// Replaces neurite/tf/layers.py:1126-1197: the reference builds O = prod(out_dims) Python-level slice
// ops (830 584 at 96^3 / k=3), concatenates them into a [O, B, F] tensor (27 copies of the input) and
// runs one K.batch_dot([O,B,F],[O,F,Cout]).  Here: out[b,o,:] = patch[b,o,:] @ kernel[o] + bias[o] with
//   * the un-shared weights kernel[O, F, Cout] streamed from HBM exactly once (non-temporal 16-byte
//     loads; at BASELINE config 5 they are 11.5 GB of bf16 -- the whole cost, AI ~ 1 flop/byte);
//   * the input patch gathered straight from the (L2-resident, 28 MB) input volume -- never unfolded;
//   * fp32 accumulation; bias + activation fused; output in the input dtype.
// One wave64 per output position: lane l owns the 16-byte slice (l % LPR) of weight rows
// f = l / LPR, l / LPR + 64/LPR, ...  (LPR = lanes per weight row = Cout * itemsize / 16), so each wave
// instruction reads 1 KiB of contiguous weights; the per-lane partial sums are combined with wave
// xor-shuffles.  All weight loads of a position are issued before the first use (deep MLP), through SGPR buffer
// resources: one VGPR of offsets for the 14 weight loads and one per patch element instead of a 64-bit address each
// took the kernel from 171 to 98 VGPRs = 5 waves per SIMD, and 0.51 -> 0.72 of the HBM roof at BASELINE config 5.
// F order (kr, kc, kz, cin) row-major (layers.py:1179-1186), positions row-major (:1172-1173).

#include <stdlib.h>

#include "nrt_common.h"
#include "activations.h"

namespace {

struct LcArgs {
    const void *x;        // [B, R, C, Z, Cin]
    const void *k;        // [O, F, Cout]
    const void *bias;     // [O, Cout] or null
    void *y;              // [B, or, oc, oz, Cout]
    int B, R, C, Z, Cin;
    int kr, kc, kz, sr, sc, sz;
    int orr, occ, ozz, Cout;
    int act;
    int stage_chunks;     // > 0: the patch of a position is kr * kc runs of kz * Cin contiguous elements whose byte length is a multiple
                          // of 16: it is staged in LDS with this many 16-byte loads per batch entry (<= 256) instead of one load per
                          // element and lane (forward kernel)
};

__device__ __forceinline__ float to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                   // round to nearest even
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void store_out(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_out(unsigned short *p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float lc_act(float v, int act) { return nrt_activate_fused(v, act); }

template <typename T> __device__ __forceinline__ T buf_load_elem(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ unsigned short buf_load_elem<unsigned short>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
template <> __device__ __forceinline__ float buf_load_elem<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// T = float or unsigned short (bf16 bits); VEC = elements per 16-byte load; NB = batch entries per pass
// SPLIT = 2: layers with more than 16 row groups per lane (32 filters in bfloat16: 27) -- two waves share a position, each streams
// half of the rows with the 16-group registers budget and the halves meet in LDS (one wave with all 32 groups in flight spilled
// 200 registers to scratch)
template <typename T, int NB, int MAXIT, bool STAGED, int SPLIT = 1>
__global__ __launch_bounds__(256, 2) void lc3d_fwd(LcArgs a, int b0, int nb) {
    constexpr bool NT = true;                      // weights are streamed once: non-temporal loads
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int LPR = a.Cout / VEC;                  // lanes per weight row (power of two, <= 64)
    const int RPW = 64 / LPR;                      // weight rows per wave iteration
    const int F = a.kr * a.kc * a.kz * a.Cin;
    const int nit = (F + RPW - 1) / RPW;           // <= MAXIT * SPLIT
    const long long O = (long long)a.orr * a.occ * a.ozz;
    const int lane = threadIdx.x & 63;
    const int sl = lane % LPR, row0 = lane / LPR;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform
    const int part = wave % SPLIT, itb = part * MAXIT;                        // this wave's row groups: itb .. itb + MAXIT - 1
    const long long nwaves = (long long)gridDim.x * ((blockDim.x >> 6) / SPLIT);
    const T *xb = (const T *)a.x;
    // staged patches: [wave][NB][stage_chunks * 16 bytes], private to the wave (its LDS operations execute in order: no barrier)
    extern __shared__ __attribute__((aligned(16))) char lc_patch[];
    typedef unsigned u32x4z __attribute__((ext_vector_type(4)));
    constexpr int NCH = 2;                                                    // 16-byte loads per lane and batch entry (<= 128 chunks)
    constexpr bool staged = STAGED;
    const unsigned pbytes = (unsigned)a.stage_chunks * 16u + 16u;             // + a zero slot: what the dead row groups of a lane read
    char *mypatch = lc_patch + (size_t)(threadIdx.x >> 6) * NB * pbytes;
    // SPLIT > 1: [positions of a block x other parts <= 3][4][Cout] partial sums behind the four waves' patches (dynamic LDS sized by
    // the launcher: Cout reaches 256 in float32 and 512 in bfloat16 -- a fixed 64-column row overlapped the batch entries)
    float *red = (float *)(lc_patch + (size_t)4 * NB * pbytes);
    const int rstride = a.Cout;
    unsigned loff[MAXIT];                                                     // byte offset of this lane's element of row group it in the staged patch
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int f = (itb + it) * RPW + row0;
        loff[it] = ((itb + it < nit) && (f < F)) ? (unsigned)f * (unsigned)sizeof(T) : pbytes - 16u;
    }
    if (STAGED && lane == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) *(u32x4z *)(mypatch + b * pbytes + pbytes - 16u) = (u32x4z){0u, 0u, 0u, 0u};
    }
    unsigned choff[NCH];                                                      // byte offset of this lane's chunk from the patch origin
    {
        const unsigned cpr = (unsigned)(a.kz * a.Cin) * (unsigned)sizeof(T) / 16u;   // chunks per run
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const unsigned c = (unsigned)k * 64u + (unsigned)lane;
            const unsigned cc = staged && c < (unsigned)a.stage_chunks ? c : 0u;
            const unsigned run = staged ? cc / cpr : 0u, j = staged ? cc % cpr : 0u;
            const unsigned dr = run / (unsigned)a.kc, dc = run % (unsigned)a.kc;
            choff[k] = ((dr * (unsigned)a.C + dc) * (unsigned)a.Z) * (unsigned)a.Cin * (unsigned)sizeof(T) + j * 16u;
        }
    }
    // a lane touches the same patch elements f = it * RPW + row0 at every position: their offsets
    // relative to the patch origin are computed once (the divisions are not in the streaming loop)
    int xoff[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int f = (itb + it) * RPW + row0;
        const int ff = ((itb + it < nit) && (f < F)) ? f : F - 1;   // dead slots re-read a valid element (no branch around loads)
        const int ci = ff % a.Cin, tap = ff / a.Cin;
        const int dz = tap % a.kz, dc = (tap / a.kz) % a.kc, dr = tap / (a.kz * a.kc);
        xoff[it] = ((dr * a.C + dc) * a.Z + dz) * a.Cin + ci;
    }
    // Buffer addressing keeps the address registers out of the way of the 14 weight slices in flight: the position's
    // weight block and the input volume are described by SGPR resources (wave-uniform), a lane contributes one 32-bit
    // byte offset to all weight loads (row advance = 1 KiB per iteration in the scalar offset) and one per patch
    // element; slices past the last row read zeros (num_records = the position's exact byte count).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned w0 = (unsigned)lane * 16u;
    const unsigned wbytes = (unsigned)F * (unsigned)a.Cout * (unsigned)sizeof(T);
    const long long xbs = (long long)a.R * a.C * a.Z * a.Cin;
    unsigned xvoff[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) xvoff[it] = (unsigned)xoff[it] * (unsigned)sizeof(T);
    // (every wave of a block runs the same number of rounds: the SPLIT form meets at block barriers)
    for (long long ob = (long long)blockIdx.x * ((blockDim.x >> 6) / SPLIT); ob < O; ob += nwaves) {
        const long long oreal = ob + wave / SPLIT;
        const bool oact = oreal < O;
        const long long o = oact ? oreal : O - 1;
        const unsigned o32 = (unsigned)o, q32 = o32 / (unsigned)a.ozz;              // positions fit 32 bits (launchers): no 64-bit division here
        const int oz = (int)(o32 - q32 * (unsigned)a.ozz), oc = (int)(q32 % (unsigned)a.occ), orr = (int)(q32 / (unsigned)a.occ);
        const unsigned xbase = (unsigned)((((long long)(orr * a.sr) * a.C + oc * a.sc) * a.Z + oz * a.sz) * a.Cin * (long long)sizeof(T));
        const char *kp = (const char *)((const T *)a.k + o * (long long)F * a.Cout);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)kp, 0, (int)wbytes, 0x00020000);
        // ---- issue the loads of (up to) 16 weight rows and their patch elements before the first use; layers with more rows per
        // lane (32 filters in bfloat16: 27) take two such chunks -- all 32 at once needs more registers than a wave has and
        // spilled to scratch ---------------------------------------------------------------------------------
        constexpr int CH = MAXIT;                 // (one chunk: every instantiation has at most 16 row groups per wave)
        float acc[NB][VEC];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[b][e] = 0.0f;
#pragma unroll
        for (int c0 = 0; c0 < MAXIT; c0 += CH) {
            vec_t w[CH];
            T xr[NB][CH];
            if constexpr (STAGED) {
                // the patch as 16-byte pieces (requested BEFORE the weights: vmcnt retires in order and the pieces are wanted first),
                // through LDS to the lanes: one or two loads per batch entry instead of one per element
                const int nk = (a.stage_chunks + 63) >> 6;
                u32x4 pc[NB][NCH];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(xb + (long long)(b0 + (b < nb ? b : 0)) * xbs), 0, (int)(xbs * (long long)sizeof(T)), 0x00020000);
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if (k < nk) pc[b][k] = __builtin_amdgcn_raw_buffer_load_b128(xres, choff[k], xbase, 0);
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(wr, w0, (itb + c0 + i) * 1024, NT ? 2 : 0);
                    w[i] = __builtin_bit_cast(vec_t, raw);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if (k < nk && k * 64 + lane < a.stage_chunks) *(u32x4 *)(mypatch + b * pbytes + (k * 64 + lane) * 16) = pc[b][k];
                __builtin_amdgcn_wave_barrier();                  // the pieces of the other lanes: same wave, LDS operations in order
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(wr, w0, (itb + c0 + i) * 1024, NT ? 2 : 0);
                    w[i] = __builtin_bit_cast(vec_t, raw);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(xb + (long long)(b0 + (b < nb ? b : 0)) * xbs), 0, (int)(xbs * (long long)sizeof(T)), 0x00020000);
#pragma unroll
                    for (int i = 0; i < CH; ++i) xr[b][i] = buf_load_elem<T>(xres, xvoff[c0 + i], xbase);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int it = itb + c0 + i;
                const bool live = (it < nit) && (it * RPW + row0 < F);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    T xe;
                    if constexpr (STAGED) xe = *(const T *)(mypatch + b * pbytes + loff[c0 + i]);
                    else xe = xr[b][i];
                    const float xv = (STAGED || live) ? to_f32(xe) : 0.0f;      // staged: dead row groups read the zero slot
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[b][e] = fmaf(xv, to_f32(w[i][e]), acc[b][e]);
                }
                if (STAGED && NB > 1 && sizeof(T) == 2 && (i & 1)) __builtin_amdgcn_sched_barrier(0);   // keeps the widened weights of two row groups live, not sixteen
            }
            if (c0 + CH < MAXIT) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- combine the 64 / LPR row slices ----------------------------------------------------
        // (level by level with the NB x VEC exchanges of a level in flight together: the other nesting -- a run-time loop over the
        // levels per value -- was a chain of NB x VEC x log2(64 / LPR) dependent LDS round trips per position)
        for (int off = LPR; off < 64; off <<= 1) {
            float other[NB][VEC];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < VEC; ++e) other[b][e] = __shfl_xor(acc[b][e], off, 64);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[b][e] += other[b][e];
        }
        if (SPLIT > 1) {                                 // the other waves' shares of the sum go through LDS
            float *rp = red + (wave / SPLIT) * ((SPLIT - 1) * 4 * rstride);
            if (part > 0 && lane < LPR) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) rp[((part - 1) * 4 + b) * rstride + sl * VEC + e] = acc[b][e];
            }
            __syncthreads();
            if (part == 0 && lane < LPR) {
#pragma unroll
                for (int q = 0; q < SPLIT - 1; ++q)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[b][e] += rp[(q * 4 + b) * rstride + sl * VEC + e];
            }
            __syncthreads();
        }
        if (lane < LPR && part == 0 && oact) {
            // bias and output rows are 16-byte vectors per lane (one load, one store per batch entry; element-wise loads with a wait
            // each were a chain of NB x VEC memory latencies per position)
            vec_t bv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) bv[e] = (T)0;
            if (a.bias) bv = *(const vec_t *)((const T *)a.bias + o * a.Cout + sl * VEC);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < nb) {
                    vec_t ov;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float v = acc[b][e];
                        if (a.bias) v += to_f32(bv[e]);
                        T q;
                        store_out(&q, lc_act(v, a.act));
                        ov[e] = q;
                    }
                    *(vec_t *)((T *)a.y + ((long long)(b0 + b) * O + o) * a.Cout + sl * VEC) = ov;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Batches of 3 .. 8 entries: the products on the matrix cores (round 4).
// The streaming kernel above multiplies every weight by NB patch values on the vector ALU: at NB = 4 it is issue-bound (0.51-0.72 of
// the weight stream), and 8 entries were two passes over the weights.  Here a wave instruction is v_mfma_f32_4x4x1_16b_f32: 16
// independent 4 x 4 outer products, block blk = lane / 4:  D[blk][r][n] += A[lane 4 blk + r] * B[lane 4 blk + n]  (layout checked on
// hardware by tools/lab/csrc/mfma4x4_layout.hip).  Block blk is weight row f = 16 c + blk of chunk c; its four lanes n hold the
// row's Cout = 4 CPL filters as CPL consecutive elements each (so the 64 lanes of a chunk load still read one contiguous piece of 512 B
// or 1 KiB), A = the patch value of batch entry 4 s + (lane % 4) for that f.  One MFMA = 256 multiply-adds for one issue slot;
// register jj of the weight slice against batch set s accumulates acc[s][jj] (rows r = batch entries 4 s + r, column n = filter
// n CPL + jj).  The weights are still read exactly once per launch of <= 8 entries; the sum over the 16 row blocks of a chunk and
// over the chunks ends in a cross-lane reduction (2 DPP rotations inside a 16-lane row, then two xor-shuffles over the rows).
// Same arithmetic as the vector kernel up to the order of the float32 sums.
// ---------------------------------------------------------------------------------------------
// (probe builds of round 4 -- one weight load per group, no MFMA, one patch load per position -- showed that the patch gathers, not the
// matrix work, are what batch 8 pays for: profiles/r04_lab/lc3d_mfma_parts_off.txt; the switches are gone from this file, see e0b416e)
constexpr int LC_WAUX = 2;            // cache policy bits of the weight loads in the matrix-core kernels: nt (others made no difference)
typedef float lc_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lc_row_ror(float v, const int ctrl_is_8) {
    // v of the lane (i + n) % 16 inside each row of 16 lanes (DPP row_ror:n)
    return ctrl_is_8 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false))
                     : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
}

// CPL = Cout / 4 filters per lane (CPL * sizeof(T) = 8 or 16 bytes), S = batch sets of 4 entries, NCMAX = most chunks of 16 weight rows,
// NPC = 16-byte patch pieces per lane and batch entry (stage_chunks <= 64 NPC).
// Software pipeline ACROSS positions: the weight chunks travel in groups of GC chunks through a ring of two register buffers; when
// group g of a position has been multiplied, group g + 2 is requested into its buffer -- past the end of the position that is the
// next position's group 0 / 1, preceded by the next position's patch pieces.  A wave therefore always has two groups (14 KB) in flight,
// also while it reduces and stores a position: without this the kernel alternated between a memory phase and a compute phase
// (batch 8: 7900 clk per position and SIMD = 4250 of stream + 3700 of issue, measured; profiles/r04_lab/lc3d_batch_*.jsonl).
// Every iteration issues the same loads in the same order (the last position re-requests itself), so the in-order vmcnt counts are fixed.
template <typename T, int CPL, int S, int NCMAX, int NPC>
__global__ __launch_bounds__(256, 2) void lc3d_fwd_mfma(LcArgs a, int b0, int nb) {
    constexpr int BPL = CPL * (int)sizeof(T);                  // bytes per lane and chunk
    constexpr int WPL = BPL / 4;                               // dwords per lane and chunk
    static_assert(BPL == 8 || BPL == 16, "lane slices of 8 or 16 bytes");
    constexpr int NB = 4 * S;
    // ring of RD register buffers of GC chunks each: 8-byte lane slices keep a whole position (3 x 9 chunks, 54 registers) in flight --
    // a buffer is re-requested for the next position the moment it has been multiplied --, 16-byte slices two groups of 7 (56 registers)
    constexpr int RD = WPL == 2 ? 3 : 2;
    constexpr int GC = WPL == 2 ? 9 : 7;
    constexpr int NG = (NCMAX + GC - 1) / GC;                  // groups per position
    static_assert(NG % RD == 0 && NG * GC == NCMAX, "the ring must close on a position");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int F = a.kr * a.kc * a.kz * a.Cin;
    const long long O = (long long)a.orr * a.occ * a.ozz;
    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, n = lane & 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Positions by XCD (blocks are dealt to the 8 XCDs round-robin; the grid is a multiple of 8): XCD x owns the contiguous eighth
    // [olo, ohi) of the positions and its waves walk it side by side, so the positions an XCD works on at one time are ~11 whole
    // z-columns of neighbouring y -- the patches of z- and y-neighbours overlap by 2/3 each and now meet in ONE 4 MB L2.  With positions
    // dealt round-robin over the whole chip the patch pieces missed L2 half of the time and were a fifth of the kernel's fabric
    // traffic at batch 8 (FETCH_SIZE 1.71 GB for 1.35 GB of weights; profiles/r04_lab/lc3d_fetch_size.txt).
    const int xcd = (int)(blockIdx.x % NRT_NXCD);
    const long long nwaves = (long long)(gridDim.x / NRT_NXCD) * 4;                  // waves of this XCD
    const long long oper = (O + NRT_NXCD - 1) / NRT_NXCD;
    const long long olo = (long long)xcd * oper, ohi = olo + oper < O ? olo + oper : O;
    extern __shared__ __attribute__((aligned(16))) char lc_patch[];
    // a batch entry's patch in LDS: NCMAX chunks of 16 elements; what lies behind the layer's F elements is zeroed once and never
    // written again, so the chunks c >= F / 16 (whose weights, loaded past num_records, are zeros too) need no branch and every A read
    // has a compile-time offset
    // (+ one chunk of padding: the four batch rows a ds_read touches then fall on different banks -- with a stride of 896 bytes
    // all four hit the same one: SQ_LDS_BANK_CONFLICT was 58 % of the LDS cycles, profiles/r04_lab/pmc_lc3d_mfma_b8_v1.json)
    constexpr unsigned pbytes = (unsigned)(NCMAX + 1) * 16u * (unsigned)sizeof(T);
    char *mypatch = lc_patch + (size_t)wave * NB * pbytes;
    for (unsigned i = (unsigned)a.stage_chunks * 16u + (unsigned)lane * 16u; i < pbytes; i += 64u * 16u)
#pragma unroll
        for (int b = 0; b < NB; ++b) *(u32x4 *)(mypatch + b * pbytes + i) = (u32x4){0u, 0u, 0u, 0u};
    // staging: this lane's 16-byte pieces of the patch
    unsigned choff[NPC];
    {
        const unsigned cpr = (unsigned)(a.kz * a.Cin) * (unsigned)sizeof(T) / 16u;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
            const unsigned c = (unsigned)k * 64u + (unsigned)lane;
            const unsigned cc = c < (unsigned)a.stage_chunks ? c : 0u;
            const unsigned run = cc / cpr, j = cc % cpr;
            const unsigned dr = run / (unsigned)a.kc, dc = run % (unsigned)a.kc;
            choff[k] = ((dr * (unsigned)a.C + dc) * (unsigned)a.Z) * (unsigned)a.Cin * (unsigned)sizeof(T) + j * 16u;
        }
    }
    // A operand: patch value f = 16 c + blk of batch entry 4 s + n  (the lane index inside the block is the batch row)
    unsigned aoff[S];
#pragma unroll
    for (int s = 0; s < S; ++s) aoff[s] = (unsigned)(4 * s + n) * pbytes + (unsigned)blk * (unsigned)sizeof(T);
    const unsigned w0 = (unsigned)lane * (unsigned)BPL;
    const unsigned wbytes = (unsigned)F * (unsigned)a.Cout * (unsigned)sizeof(T);
    const long long xbs = (long long)a.R * a.C * a.Z * a.Cin;
    // ONE descriptor for the nb input volumes of this launch (nb * volume bytes < 2^31: launcher); the batch entry goes into the offset
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const T *)a.x + (long long)b0 * xbs), 0, (int)((long long)nb * xbs * (long long)sizeof(T)), 0x00020000);
    const unsigned xbs_bytes = (unsigned)(xbs * (long long)sizeof(T));

    struct Pos { long long o; unsigned xbase; bool live; };
    auto decode = [&](long long oreal) {
        Pos p;
        p.live = oreal < ohi;
        p.o = p.live ? oreal : ohi - 1;
        const unsigned o32 = (unsigned)p.o, q32 = o32 / (unsigned)a.ozz;
        const int oz = (int)(o32 - q32 * (unsigned)a.ozz), oc = (int)(q32 % (unsigned)a.occ), orr = (int)(q32 / (unsigned)a.occ);
        p.xbase = (unsigned)((((long long)(orr * a.sr) * a.C + oc * a.sc) * a.Z + oz * a.sz) * a.Cin * (long long)sizeof(T));
        return p;
    };
    auto weights_of = [&](long long o) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)((const T *)a.k + o * (long long)F * a.Cout)), 0, (int)wbytes, 0x00020000);
    };
    u32x4 pc[NB][NPC];
    unsigned w[RD][GC][WPL];
    auto issue_patch = [&](const Pos &p) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int k = 0; k < NPC; ++k)
                pc[b][k] = __builtin_amdgcn_raw_buffer_load_b128(xres, choff[k], p.xbase + (unsigned)(b < nb ? b : 0) * xbs_bytes, 0);
    };
    auto issue_group = [&](const __amdgpu_buffer_rsrc_t wr, const int buf, const int g) {
#pragma unroll
        for (int i = 0; i < GC; ++i) {
            const int c = g * GC + i;                          // chunks past the last one lie past num_records: zeros
            if constexpr (WPL == 2) {
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(wr, w0, c * 64 * BPL, LC_WAUX);
                w[buf][i][0] = raw[0]; w[buf][i][1] = raw[1];
            } else {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(wr, w0, c * 64 * BPL, LC_WAUX);
#pragma unroll
                for (int k = 0; k < 4; ++k) w[buf][i][k] = raw[k];
            }
        }
    };

    // the bias slice of a position travels with its patch pieces (requested before the weights: vmcnt retires in order, a bias load
    // issued in the epilogue could only be waited for by draining the next position's 28 weight loads)
    typedef T vec_t __attribute__((ext_vector_type(CPL)));
    auto bias_of = [&](long long o) {
        vec_t z;
#pragma unroll
        for (int e = 0; e < CPL; ++e) z[e] = (T)0;
        return a.bias ? *(const vec_t *)((const T *)a.bias + o * a.Cout + n * CPL) : z;
    };
    const long long ofirst = olo + (long long)(blockIdx.x / NRT_NXCD) * 4 + wave;
    if (ofirst >= ohi) return;                                 // (no block barrier in this kernel: a wave may leave alone)
    Pos cur = decode(ofirst);
    __amdgpu_buffer_rsrc_t wcur = weights_of(cur.o);
    issue_patch(cur);
    vec_t bcur = bias_of(cur.o), bnext = bcur;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < RD; ++g) {
        issue_group(wcur, g, g);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (;;) {
        const long long onext = cur.o + nwaves;
        const Pos nxt = decode(onext);                         // past the end: the last position again (loads only, never stored)
        const __amdgpu_buffer_rsrc_t wnext = weights_of(nxt.o);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int k = 0; k < NPC; ++k)
                if (k * 64 + lane < a.stage_chunks) *(u32x4 *)(mypatch + b * pbytes + (k * 64 + lane) * 16) = pc[b][k];
        __builtin_amdgcn_wave_barrier();

        lc_f4 acc[S][CPL];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) acc[s][jj] = (lc_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int i = 0; i < GC; ++i) {
                const int c = g * GC + i;
                if (c < NCMAX) {                               // compile-time
                    const unsigned coff = (unsigned)c * 16u * (unsigned)sizeof(T);
                    float av[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        av[s] = to_f32(*(const T *)(mypatch + aoff[s] + coff));
                    }
#pragma unroll
                    for (int jj = 0; jj < CPL; ++jj) {
                        float bw;
                        if constexpr (sizeof(T) == 2) {
                            const unsigned d = w[g % RD][i][jj >> 1];
                            bw = __uint_as_float((jj & 1) ? (d & 0xffff0000u) : (d << 16));
                        } else {
                            bw = __uint_as_float(w[g % RD][i][jj]);
                        }
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            acc[s][jj] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[s], bw, acc[s][jj], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // this buffer is free: the group RD ahead, which past the end of the position belongs to the next one
            if (g == NG - RD) { issue_patch(nxt); bnext = bias_of(nxt.o); __builtin_amdgcn_sched_barrier(0); }   // AHEAD of the weights in the queue
            if (g + RD < NG) issue_group(wcur, g % RD, g + RD);
            else issue_group(wnext, g % RD, g + RD - NG);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sum over the 16 row blocks: inside a row of 16 lanes (4 blocks) two rotations; lane (row R, block q of the row, n)
        // then keeps batch row r = q, and two xor-shuffles add the four rows of lanes ------------------------------------------
        const int q = blk & 3, R = lane >> 4;
        float val[S][CPL];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) {
                lc_f4 v = acc[s][jj];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] += lc_row_ror(v[r], 1);
                    v[r] += lc_row_ror(v[r], 0);
                }
                const float lo = (q & 1) ? v[1] : v[0], hi = (q & 1) ? v[3] : v[2];
                val[s][jj] = (q & 2) ? hi : lo;
            }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            float other[S][CPL];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) other[s][jj] = __shfl_xor(val[s][jj], off, 64);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) val[s][jj] += other[s][jj];
        }
        // lanes of row R < S write batch entry 4 R + q: CPL consecutive filters per lane, the four lanes n of a block one output row
        const int b = 4 * R + q;
        if (R < S && b < nb) {
            const vec_t bv = bcur;
            vec_t ov;
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                float v = val[0][e];
#pragma unroll
                for (int s = 1; s < S; ++s) v = (R == s) ? val[s][e] : v;
                if (a.bias) v += to_f32(bv[e]);
                T qv;
                store_out(&qv, lc_act(v, a.act));
                ov[e] = qv;
            }
            *(vec_t *)((T *)a.y + ((long long)(b0 + b) * O + cur.o) * a.Cout + n * CPL) = ov;
        }
        if (!nxt.live) break;
        cur = nxt;
        wcur = wnext;
        bcur = bnext;
    }
}

// ---------------------------------------------------------------------------------------------
// The same kernel with the patches staged ONCE PER BLOCK (the form the launcher prefers: 3 x 3 x 3 kernels over 16 input channels,
// unit z stride).  lc3d_fwd_mfma lets every wave gather its own position's patches: 8 gathers of 54 x 16 bytes per position at batch
// 8, as many L1 -> L2 requests as the position's weights, and they -- not the matrix work -- were what batches of 6-8 paid for
// (profiles/r04_lab/lc3d_mfma_parts_off.txt).  Here the four waves of a block take four z-CONSECUTIVE positions of one (row, column)
// and the block stages the union of their patches: per batch entry kr * kc runs of (kz + 3) voxels x 16 channels (192 contiguous
// bytes in bfloat16) instead of four times kr * kc runs of 96 bytes -- half the bytes, a third of the cache lines, half the load
// instructions.  The patch lives in LDS twice (the next group's pieces are written while slower waves still read this group's);
// one block barrier per group.  Wave w reads its window at a z offset of w voxels; everything else is lc3d_fwd_mfma.
// ---------------------------------------------------------------------------------------------
template <typename T, int CPL, int S, int NCMAX, int NPT>
__global__ __launch_bounds__(256, 2) void lc3d_fwd_mfma_blk(LcArgs a, int b0, int nb) {
    constexpr int BPL = CPL * (int)sizeof(T);
    constexpr int WPL = BPL / 4;
    static_assert(BPL == 8 || BPL == 16, "lane slices of 8 or 16 bytes");
    constexpr int NB = 4 * S;
    constexpr int RD = WPL == 2 ? 3 : 2;
    constexpr int GC = WPL == 2 ? 9 : 7;
    constexpr int NG = (NCMAX + GC - 1) / GC;
    static_assert(NG % RD == 0 && NG * GC == NCMAX, "the ring must close on a position");
    constexpr int KZ = 3;                                      // taps along z (launcher): chunk c = tap (c / KZ, c % KZ), 16 channels
    constexpr unsigned CINB = 16u * (unsigned)sizeof(T);       // bytes of one input voxel
    constexpr unsigned RUNB = (KZ + 3) * CINB;                 // one (kr, kc) run of the union patch
    constexpr unsigned PB = 9u * RUNB + 5u * CINB;             // patch of one batch entry: <= 9 runs, + zeros that the padded chunks and the
                                                               // last wave's window read (also staggers the four batch rows over the banks)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef T vec_t __attribute__((ext_vector_type(CPL)));
    const int F = a.kr * a.kc * a.kz * a.Cin;
    const long long O = (long long)a.orr * a.occ * a.ozz;
    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, n = lane & 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // groups of four z-consecutive positions; XCD x owns a contiguous eighth of the groups (see lc3d_fwd_mfma)
    const unsigned ZG = ((unsigned)a.ozz + 3u) >> 2;
    const long long ngrp = (long long)a.orr * a.occ * ZG;
    const int xcd = (int)(blockIdx.x % NRT_NXCD);
    const long long gstep = (long long)(gridDim.x / NRT_NXCD);
    const long long gper = (ngrp + NRT_NXCD - 1) / NRT_NXCD;
    const long long glo = (long long)xcd * gper, ghi = glo + gper < ngrp ? glo + gper : ngrp;
    extern __shared__ __attribute__((aligned(16))) char lc_patch[];
    for (unsigned i = threadIdx.x * 16u; i < 2u * NB * PB; i += 256u * 16u) *(u32x4 *)(lc_patch + i) = (u32x4){0u, 0u, 0u, 0u};
    // this thread's pieces of the union patch: (entry, run, 16-byte piece of the run)
    const unsigned ppr = RUNB / 16u, runs = (unsigned)(a.kr * a.kc), ppe = runs * ppr;
    unsigned goff[NPT], loff[NPT];
    bool pval[NPT];
    const long long xbs = (long long)a.R * a.C * a.Z * a.Cin;
    const unsigned xbs_bytes = (unsigned)(xbs * (long long)sizeof(T));
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const unsigned p = threadIdx.x + 256u * (unsigned)k;
        pval[k] = p < (unsigned)NB * ppe;
        const unsigned pp = pval[k] ? p : 0u;
        const unsigned e = pp / ppe, q = pp % ppe, run = q / ppr, j = q % ppr;
        const unsigned dr = run / (unsigned)a.kc, dc = run % (unsigned)a.kc;
        goff[k] = (e < (unsigned)nb ? e : 0u) * xbs_bytes + ((dr * (unsigned)a.C + dc) * (unsigned)a.Z) * (unsigned)a.Cin * (unsigned)sizeof(T) + j * 16u;
        loff[k] = e * PB + run * RUNB + j * 16u;
    }
    unsigned aoff[S];
#pragma unroll
    for (int s = 0; s < S; ++s) aoff[s] = (unsigned)(4 * s + n) * PB + (unsigned)wave * CINB + (unsigned)blk * (unsigned)sizeof(T);
    const unsigned w0 = (unsigned)lane * (unsigned)BPL;
    const unsigned wbytes = (unsigned)F * (unsigned)a.Cout * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const T *)a.x + (long long)b0 * xbs), 0, (int)((long long)nb * xbs * (long long)sizeof(T)), 0x00020000);

    struct Grp { long long o; unsigned xbase; bool live, act; };     // live: the group exists; act: this wave's position exists
    auto decode = [&](long long greal) {
        Grp p;
        p.live = greal < ghi;
        const unsigned g32 = (unsigned)(p.live ? greal : ghi - 1);
        const unsigned col = g32 / ZG, zg = g32 - col * ZG;
        const unsigned orr = col / (unsigned)a.occ, oc = col - orr * (unsigned)a.occ;
        const unsigned oz = 4u * zg + (unsigned)wave;
        p.act = p.live && oz < (unsigned)a.ozz;
        p.o = (long long)col * a.ozz + (oz < (unsigned)a.ozz ? oz : (unsigned)a.ozz - 1u);
        p.xbase = (unsigned)((((long long)(orr * (unsigned)a.sr) * a.C + oc * (unsigned)a.sc) * a.Z + 4u * zg) * a.Cin * (long long)sizeof(T));
        return p;
    };
    auto weights_of = [&](const Grp &p) {                      // a wave without a position streams nothing: zero records
        return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)((const T *)a.k + p.o * (long long)F * a.Cout)), 0,
                                                 p.act ? (int)wbytes : 0, 0x00020000);
    };
    auto bias_of = [&](long long o) {
        vec_t z;
#pragma unroll
        for (int e = 0; e < CPL; ++e) z[e] = (T)0;
        return a.bias ? *(const vec_t *)((const T *)a.bias + o * a.Cout + n * CPL) : z;
    };
    u32x4 pc[NPT];
    unsigned w[RD][GC][WPL];
    auto issue_patch = [&](const Grp &p) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) pc[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, goff[k], p.xbase, 0);
    };
    auto issue_group = [&](const __amdgpu_buffer_rsrc_t wr, const int buf, const int g) {
#pragma unroll
        for (int i = 0; i < GC; ++i) {
            const int c = g * GC + i;
            if constexpr (WPL == 2) {
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(wr, w0, c * 64 * BPL, LC_WAUX);
                w[buf][i][0] = raw[0]; w[buf][i][1] = raw[1];
            } else {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(wr, w0, c * 64 * BPL, LC_WAUX);
#pragma unroll
                for (int k = 0; k < 4; ++k) w[buf][i][k] = raw[k];
            }
        }
    };

    const long long gfirst = glo + (long long)(blockIdx.x / NRT_NXCD);
    if (gfirst >= ghi) return;                                 // the whole block
    Grp cur = decode(gfirst);
    __amdgpu_buffer_rsrc_t wcur = weights_of(cur);
    issue_patch(cur);
    vec_t bcur = bias_of(cur.o), bnext = bcur;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < RD; ++g) {
        issue_group(wcur, g, g);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                           // the zero fill of both patch buffers
    unsigned par = 0;                                          // which patch buffer this group uses
    for (long long gi = gfirst;; gi += gstep) {
        const Grp nxt = decode(gi + gstep);
        const __amdgpu_buffer_rsrc_t wnext = weights_of(nxt);
        char *patch = lc_patch + (size_t)par * NB * PB;
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            if (pval[k]) *(u32x4 *)(patch + loff[k]) = pc[k];
        __syncthreads();                                       // also: every wave is done with the OTHER buffer's previous contents

        lc_f4 acc[S][CPL];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) acc[s][jj] = (lc_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int i = 0; i < GC; ++i) {
                const int c = g * GC + i;
                if (c < NCMAX) {
                    const unsigned coff = (unsigned)(c / KZ) * RUNB + (unsigned)(c % KZ) * CINB;       // chunks past the layer's: zeros
                    float av[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) av[s] = to_f32(*(const T *)(patch + aoff[s] + coff));
#pragma unroll
                    for (int jj = 0; jj < CPL; ++jj) {
                        float bw;
                        if constexpr (sizeof(T) == 2) {
                            const unsigned d = w[g % RD][i][jj >> 1];
                            bw = __uint_as_float((jj & 1) ? (d & 0xffff0000u) : (d << 16));
                        } else {
                            bw = __uint_as_float(w[g % RD][i][jj]);
                        }
#pragma unroll
                        for (int s = 0; s < S; ++s) acc[s][jj] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[s], bw, acc[s][jj], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g == NG - RD) { issue_patch(nxt); bnext = bias_of(nxt.o); __builtin_amdgcn_sched_barrier(0); }
            if (g + RD < NG) issue_group(wcur, g % RD, g + RD);
            else issue_group(wnext, g % RD, g + RD - NG);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int q = blk & 3, R = lane >> 4;
        float val[S][CPL];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) {
                lc_f4 v = acc[s][jj];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] += lc_row_ror(v[r], 1);
                    v[r] += lc_row_ror(v[r], 0);
                }
                const float lo = (q & 1) ? v[1] : v[0], hi = (q & 1) ? v[3] : v[2];
                val[s][jj] = (q & 2) ? hi : lo;
            }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            float other[S][CPL];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) other[s][jj] = __shfl_xor(val[s][jj], off, 64);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) val[s][jj] += other[s][jj];
        }
        const int b = 4 * R + q;
        if (R < S && b < nb && cur.act) {
            const vec_t bv = bcur;
            vec_t ov;
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                float v = val[0][e];
#pragma unroll
                for (int s = 1; s < S; ++s) v = (R == s) ? val[s][e] : v;
                if (a.bias) v += to_f32(bv[e]);
                T qv;
                store_out(&qv, lc_act(v, a.act));
                ov[e] = qv;
            }
            *(vec_t *)((T *)a.y + ((long long)(b0 + b) * O + cur.o) * a.Cout + n * CPL) = ov;
        }
        if (!nxt.live) break;
        cur = nxt;
        wcur = wnext;
        bcur = bnext;
        par ^= 1u;
    }
}

template <typename T, int CPL>
bool launch_mfma(const LcArgs &a, hipStream_t st) {
    // experiment knob: NRT_LC_MFMA = 0 keeps the vector kernel for every batch size
    static int on = -1;
    if (on < 0) { const char *e = getenv("NRT_LC_MFMA"); on = e ? atoi(e) : 1; }
    constexpr int NCMAX = CPL * (int)sizeof(T) == 8 ? 27 : 28;          // chunks of 16 weight rows a position may have (ring geometry of the kernel)
    const int F = a.kr * a.kc * a.kz * a.Cin;
    if (!on || a.B < 3 || a.Cout != 4 * CPL || F % 16 || F / 16 > NCMAX || a.stage_chunks < 1 || a.stage_chunks > 128) return false;
    if ((size_t)a.stage_chunks * 16 != (size_t)F * sizeof(T)) return false;      // the staged pieces are exactly the patch
    if ((((uintptr_t)a.k) & 15) || (((uintptr_t)a.y) & 15) || (a.bias && (((uintptr_t)a.bias) & 15))) return false;
    if ((long long)a.R * a.C * a.Z * a.Cin * (long long)sizeof(T) * 8 >= (1ll << 31)) return false;      // 8 volumes behind one descriptor
    static int kblocks = -1;
    if (kblocks < 0) { const char *e = getenv("NRT_LC_BLOCKS"); kblocks = e ? atoi(e) : 0; }
    const long long O = (long long)a.orr * a.occ * a.ozz;
    const size_t pb = (size_t)(NCMAX + 1) * 16 * sizeof(T);
    const bool two = a.stage_chunks > 64;
    // the grid is exactly the resident blocks (a wave's load pipeline runs across its positions; a second round of blocks would run
    // on a nearly empty machine: 1024 blocks on 768 slots cost 25 % at batch 8)
    auto run = [&](auto kernel, int sets, int b0, int nb) {
        const size_t shm = (size_t)4 * 4 * sets * pb;
        static int per_cu_dev[64];                             // per instantiation (the LDS size is a function of it) and device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        int &per_cu = per_cu_dev[dev];
        if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, shm) != hipSuccess || per_cu < 1)) per_cu = 2;
        unsigned blocks = nrt_xcd_grid((unsigned)((O + 3) / 4));
        const unsigned cap = nrt_xcd_grid(kblocks > 0 ? (unsigned)kblocks : (unsigned)per_cu * (unsigned)nrt_num_cus());
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), shm, st, a, b0, nb);
    };
    // the block-staged form: 3 taps along z over exactly 16 input channels, unit z stride, <= 9 (kr, kc) runs
    static int kblk = -1;
    if (kblk < 0) { const char *e = getenv("NRT_LC_BLK"); kblk = e ? atoi(e) : 1; }
    const bool blk_ok = kblk && a.kz == 3 && a.Cin == 16 && a.sz == 1 && a.kr * a.kc <= 9 && (((uintptr_t)a.x) & 15) == 0;
    constexpr size_t CINB = 16 * sizeof(T), PBB = 9 * 6 * CINB + 5 * CINB;
    auto run_blk = [&](auto kernel, int sets, int b0, int nb) {
        const size_t shm = (size_t)2 * 4 * sets * PBB;
        static int per_cu_dev[64];                             // per instantiation and device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        int &per_cu = per_cu_dev[dev];
        // (function attributes are per device: set at every launch that needs more than the default 64 KB -- none does today)
        if (shm > 64 * 1024 && hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return false;
        if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, shm) != hipSuccess || per_cu < 1)) per_cu = 2;
        const long long ngrp = (long long)a.orr * a.occ * ((a.ozz + 3) / 4);
        unsigned blocks = nrt_xcd_grid((unsigned)ngrp);
        const unsigned cap = nrt_xcd_grid(kblocks > 0 ? (unsigned)kblocks : (unsigned)per_cu * (unsigned)nrt_num_cus());
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), shm, st, a, b0, nb);
        return true;
    };
    constexpr int NPT1 = (int)((4 * 9 * 6 * CINB / 16 + 255) / 256), NPT2 = (int)((8 * 9 * 6 * CINB / 16 + 255) / 256);
    for (int b0 = 0; b0 < a.B; b0 += 8) {
        const int nb = a.B - b0 < 8 ? a.B - b0 : 8;
        if (blk_ok) {
            // (false: nothing was launched for this chunk -- the caller's vector kernels then compute the whole batch)
            if (!(nb <= 4 ? run_blk(lc3d_fwd_mfma_blk<T, CPL, 1, NCMAX, NPT1>, 1, b0, nb) : run_blk(lc3d_fwd_mfma_blk<T, CPL, 2, NCMAX, NPT2>, 2, b0, nb)))
                return false;
            continue;
        }
        if (nb <= 4) {
            if (two) run(lc3d_fwd_mfma<T, CPL, 1, NCMAX, 2>, 1, b0, nb);
            else run(lc3d_fwd_mfma<T, CPL, 1, NCMAX, 1>, 1, b0, nb);
        } else {
            if (two) run(lc3d_fwd_mfma<T, CPL, 2, NCMAX, 2>, 2, b0, nb);
            else run(lc3d_fwd_mfma<T, CPL, 2, NCMAX, 1>, 2, b0, nb);
        }
    }
    return true;
}

// any Cout / F: one thread per (position, cout)
template <typename T>
__global__ __launch_bounds__(256) void lc3d_generic(LcArgs a) {
    const int F = a.kr * a.kc * a.kz * a.Cin;
    const long long O = (long long)a.orr * a.occ * a.ozz;
    const long long total = O * a.Cout;
    const T *xb = (const T *)a.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(e % a.Cout);
        const long long o = e / a.Cout;
        const int oz = (int)(o % a.ozz), oc = (int)((o / a.ozz) % a.occ), orr = (int)(o / ((long long)a.ozz * a.occ));
        const T *kp = (const T *)a.k + o * (long long)F * a.Cout + co;
        for (int b = 0; b < a.B; ++b) {
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) {
                const int ci = f % a.Cin, tap = f / a.Cin;
                const int dz = tap % a.kz, dc = (tap / a.kz) % a.kc, dr = tap / (a.kz * a.kc);
                const long long xi = ((((long long)b * a.R + (orr * a.sr + dr)) * a.C + (oc * a.sc + dc)) * a.Z + (oz * a.sz + dz)) * a.Cin + ci;
                acc = fmaf(to_f32(xb[xi]), to_f32(kp[(long long)f * a.Cout]), acc);
            }
            if (a.bias) acc += to_f32(((const T *)a.bias)[o * a.Cout + co]);
            store_out((T *)a.y + ((long long)b * O + o) * a.Cout + co, lc_act(acc, a.act));
        }
    }
}

template <typename T, int MAXIT, bool STAGED, int SPLIT = 1>
void launch_vec_st(const LcArgs &a, unsigned blocks, hipStream_t st) {
    for (int b0 = 0; b0 < a.B; b0 += 4) {
        const int nb = a.B - b0 < 4 ? a.B - b0 : 4;
        const size_t ps = ((size_t)a.stage_chunks * 16 + 16) * 4;               // per batch entry: 4 waves, + the zero slot
        const size_t rs = SPLIT > 1 ? (size_t)3 * 4 * a.Cout * sizeof(float) : 0;  // the SPLIT partial sums (<= 24 KB; total < 64 KB)
        if (nb == 1) hipLaunchKernelGGL((lc3d_fwd<T, 1, MAXIT, STAGED, SPLIT>), dim3(blocks), dim3(256), ps + rs, st, a, b0, nb);
        else if (nb == 2) hipLaunchKernelGGL((lc3d_fwd<T, 2, MAXIT, STAGED, SPLIT>), dim3(blocks), dim3(256), 2 * ps + rs, st, a, b0, nb);
        else hipLaunchKernelGGL((lc3d_fwd<T, 4, MAXIT, STAGED, SPLIT>), dim3(blocks), dim3(256), 4 * ps + rs, st, a, b0, nb);
    }
}

template <typename T, int MAXIT, int SPLIT = 1>
void launch_vec(const LcArgs &a, hipStream_t st) {
    // experiment knob: NRT_LC_BLOCKS (grid size)
    static int kblocks = -1;
    if (kblocks < 0) { const char *e = getenv("NRT_LC_BLOCKS"); kblocks = e ? atoi(e) : 0; }
    const long long O = (long long)a.orr * a.occ * a.ozz;
    unsigned blocks = (unsigned)((O + 4 / SPLIT - 1) / (4 / SPLIT));
    const unsigned cap = kblocks > 0 ? (unsigned)kblocks : 256u * 20u;     // 5 resident blocks per CU x 4 rounds (profiles/)
    if (blocks > cap) blocks = cap;
    if (a.stage_chunks > 0) launch_vec_st<T, MAXIT, true, SPLIT>(a, blocks, st);
    else launch_vec_st<T, MAXIT, false, SPLIT>(a, blocks, st);
}

template <typename T>
int launch_any(const LcArgs &a_in, int variant, hipStream_t st) {
    const LcArgs &a = a_in;
    constexpr int VEC = 16 / (int)sizeof(T);
    const int F = a.kr * a.kc * a.kz * a.Cin;
    bool vec_ok = (a.Cout % VEC) == 0;
    int LPR = vec_ok ? a.Cout / VEC : 0;
    vec_ok = vec_ok && LPR >= 1 && LPR <= 64 && (LPR & (LPR - 1)) == 0 && (((uintptr_t)a.k) & 15) == 0;
    int nit = vec_ok ? (F + (64 / LPR) - 1) / (64 / LPR) : 0;
    vec_ok = vec_ok && nit <= 64;
    vec_ok = vec_ok && (long long)a.R * a.C * a.Z * a.Cin * (long long)sizeof(T) < (1ll << 31) &&
             (long long)F * a.Cout * (long long)sizeof(T) < (1ll << 31);          // 32-bit buffer offsets
    if (variant == 0) variant = vec_ok ? 2 : 1;
    if (variant == 2) {
        if (!vec_ok) return NRT_ERR_UNSUPPORTED;
        LcArgs a = a_in;
        // stage the patch through LDS when its kr * kc runs (kz * Cin contiguous elements) are whole 16-byte pieces
        const long long runb = (long long)a.kz * a.Cin * (long long)sizeof(T), chunks = runb / 16 * a.kr * a.kc;
        static int kstage = -1;
        if (kstage < 0) { const char *e = getenv("NRT_LC_STAGE"); kstage = e ? atoi(e) : 1; }
        a.stage_chunks = (kstage && runb % 16 == 0 && ((long long)a.Cin * (long long)sizeof(T)) % 16 == 0 && chunks <= 128 &&
                          (((uintptr_t)a.x) & 15) == 0 && chunks * 16 * 4 * 4 <= 48 * 1024) ? (int)chunks : 0;
        bool done = false;
        if constexpr (sizeof(T) == 2) done = a.Cout == 16 ? launch_mfma<T, 4>(a, st) : (a.Cout == 32 ? launch_mfma<T, 8>(a, st) : false);
        else done = a.Cout == 16 ? launch_mfma<T, 4>(a, st) : (a.Cout == 8 ? launch_mfma<T, 2>(a, st) : false);
        if (done) { NRT_CHECK_LAUNCH(); return NRT_OK; }
        if (nit <= 8) launch_vec<T, 8>(a, st);
        else if (nit <= 14) launch_vec<T, 14>(a, st);
        else if (nit <= 16) launch_vec<T, 16>(a, st);
        else if (nit <= 32) launch_vec<T, 16, 2>(a, st);  // up to 32 row groups: two waves per position
        else launch_vec<T, 16, 4>(a, st);                 // up to 64 (32 filters in float32: 54): four
    } else {
        const long long total = (long long)a.orr * a.occ * a.ozz * a.Cout;
        unsigned blocks = (unsigned)((total + 255) / 256);
        if (blocks > 256u * 16u) blocks = 256u * 16u;
        hipLaunchKernelGGL((lc3d_generic<T>), dim3(blocks), dim3(256), 0, st, a);
    }
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// ---------------------------------------------------------------------------------------------
// backward (what TF's autodiff gives for the batch_dot of layers.py:1189): per output position o
//   dK[o][f][co] = sum_b patch[b][o][f] * dpre[b][o][co]           written once, 16-byte slices (HBM-bound: |K| bytes)
//   dbias[o][co] = sum_b dpre[b][o][co]
//   dx[b][patch element f of o] += sum_co K[o][f][co] * dpre[b][o][co]   (float atomics into a float32 buffer)
// Same wave-per-position lane mapping as the forward: lane = (weight row f mod RPW, 16-byte cout slice).
// dpre = grad_out * act'(y) is formed on the fly from the layer output y.
// ---------------------------------------------------------------------------------------------
struct LcBwdArgs {
    LcArgs f;                // x, k, (bias unused), y = layer output
    const void *g;           // grad_out [B, O, Cout]
    void *dk;                // [O, F, Cout] or null
    void *dbias;             // [O, Cout] or null
    float *dx;               // float32 [B, R, C, Z, Cin], zero-filled by the caller, or null
};

template <typename T>
__device__ __forceinline__ float lc_dpre(float g, float y, int act) { return g * nrt_activate_slope(y, act); }

// NB batch entries per pass (the first pass writes dK, later passes add to it: a lane owns its 16-byte slices exclusively);
// HAS_DX: also stream the weights and scatter the input gradient.  Buffer addressing as in the forward kernel.
template <typename T, int MAXIT, int NB, bool HAS_DX, int SPLIT = 1>
__global__ __launch_bounds__(256, 2) void lc3d_bwd(LcBwdArgs ba, int b0, int nb) {
    const LcArgs &a = ba.f;
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int LPR = a.Cout / VEC, RPW = 64 / LPR;
    const int F = a.kr * a.kc * a.kz * a.Cin;
    const int nit = (F + RPW - 1) / RPW;
    const long long O = (long long)a.orr * a.occ * a.ozz;
    const int lane = threadIdx.x & 63;
    const int sl = lane % LPR, row0 = lane / LPR;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = wave % SPLIT, itb = part * MAXIT;           // SPLIT = 2: two waves share a position, each owns half of the row groups
    const long long nwaves = (long long)gridDim.x * ((blockDim.x >> 6) / SPLIT);
    const T *xb = (const T *)a.x;
    unsigned xvoff[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int f = (itb + it) * RPW + row0;
        const int ff = ((itb + it < nit) && (f < F)) ? f : F - 1;
        const int ci = ff % a.Cin, tap = ff / a.Cin;
        const int dz = tap % a.kz, dc = (tap / a.kz) % a.kc, dr = tap / (a.kz * a.kc);
        xvoff[it] = (unsigned)((((dr * a.C + dc) * a.Z + dz) * a.Cin + ci) * (int)sizeof(T));
    }
    const unsigned w0 = (unsigned)lane * 16u;
    const unsigned wbytes = (unsigned)F * (unsigned)a.Cout * (unsigned)sizeof(T);
    const long long xbs = (long long)a.R * a.C * a.Z * a.Cin;
    for (long long o = (long long)blockIdx.x * ((blockDim.x >> 6) / SPLIT) + wave / SPLIT; o < O; o += nwaves) {
        const unsigned o32 = (unsigned)o, q32 = o32 / (unsigned)a.ozz;              // positions fit 32 bits (launchers): no 64-bit division here
        const int oz = (int)(o32 - q32 * (unsigned)a.ozz), oc = (int)(q32 % (unsigned)a.occ), orr = (int)(q32 / (unsigned)a.occ);
        const long long xbase_e = (((long long)(orr * a.sr) * a.C + oc * a.sc) * a.Z + oz * a.sz) * a.Cin;
        const unsigned xbase = (unsigned)(xbase_e * (long long)sizeof(T));
        // ---- this lane's cout slice of dpre[b][o] -------------------------------------------------------------------------
        float dp[NB][VEC], db[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) db[e] = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const long long go = ((long long)(b0 + (b < nb ? b : 0)) * O + o) * a.Cout + sl * VEC;
            const vec_t gv = *(const vec_t *)((const T *)ba.g + go);
            if (a.act != 0) {
                const vec_t yv = *(const vec_t *)((const T *)a.y + go);
#pragma unroll
                for (int e = 0; e < VEC; ++e) dp[b][e] = b < nb ? lc_dpre<T>(to_f32(gv[e]), to_f32(yv[e]), a.act) : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) dp[b][e] = b < nb ? to_f32(gv[e]) : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) db[e] += dp[b][e];
        }
        char *dkp = ba.dk ? (char *)ba.dk + o * (long long)wbytes : nullptr;
        // the patch elements (and, for dx, the weight slices) of a chunk of row groups are in flight at a time: 8 where 16 of them with
        // four batch entries and the weights would not fit in the registers
        constexpr int CH = (MAXIT >= 16 && HAS_DX && NB == 4) ? 8 : MAXIT;
#pragma unroll
        for (int c0 = 0; c0 < MAXIT; c0 += CH) {
        T xr[NB][CH];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(xb + (long long)(b0 + (b < nb ? b : 0)) * xbs), 0, (int)(xbs * (long long)sizeof(T)), 0x00020000);
#pragma unroll
            for (int i = 0; i < CH; ++i) xr[b][i] = buf_load_elem<T>(xres, xvoff[c0 + i], xbase);
        }
        vec_t w[HAS_DX ? CH : 1];
        if (HAS_DX) {
            const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
                (void *)((const char *)a.k + o * (long long)wbytes), 0, (int)wbytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < CH; ++i)
                w[HAS_DX ? i : 0] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(wr, w0, (itb + c0 + i) * 1024, 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int it = itb + c0 + i;
            const bool live = (it < nit) && (it * RPW + row0 < F);
            const unsigned off = w0 + (unsigned)it * 1024u;
            if (dkp && off < wbytes) {
                float acc[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float xv = live ? to_f32(xr[b][i]) : 0.0f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf(xv, dp[b][e], acc[e]);
                }
                vec_t *dst = (vec_t *)(dkp + off);
                if (b0 > 0) {                                   // later batch chunk: add to what the first chunk wrote
                    const vec_t old = *dst;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += to_f32(old[e]);
                }
                vec_t ov;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { T tmp; store_out(&tmp, acc[e]); ov[e] = tmp; }
                __builtin_nontemporal_store(ov, dst);
            }
        }
        if (HAS_DX) {
            // dx[b][patch element f] += sum over cout of w[f][:] * dpre[b][:]: the partial sums of all row groups and batch entries of
            // the chunk cross the cout slices of a row level by level (one level = CH x NB independent exchanges in flight; a run-time
            // loop per value was a chain of CH x NB x log2(LPR) dependent LDS round trips)
            float t[CH][NB];
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float v = 0.0f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v = fmaf(to_f32(w[HAS_DX ? i : 0][e]), dp[b][e], v);
                    t[i][b] = v;
                }
            for (int off2 = 1; off2 < LPR; off2 <<= 1) {
                float u[CH][NB];
#pragma unroll
                for (int i = 0; i < CH; ++i)
#pragma unroll
                    for (int b = 0; b < NB; ++b) u[i][b] = __shfl_xor(t[i][b], off2, 64);
#pragma unroll
                for (int i = 0; i < CH; ++i)
#pragma unroll
                    for (int b = 0; b < NB; ++b) t[i][b] += u[i][b];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int it = itb + c0 + i;
                const bool live = (it < nit) && (it * RPW + row0 < F);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (live && sl == 0 && b < nb)
                        unsafeAtomicAdd(ba.dx + (long long)(b0 + b) * xbs + xbase_e + (long long)(xvoff[c0 + i] / (unsigned)sizeof(T)), t[i][b]);
            }
        }
        }
        if (ba.dbias && lane < LPR && part == 0) {
            vec_t *dbp = (vec_t *)((T *)ba.dbias + o * a.Cout + sl * VEC);            // one 16-byte row slice per lane
            vec_t oldv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) oldv[e] = (T)0;
            if (b0 > 0) oldv = *dbp;
            vec_t nv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T q;
                store_out(&q, b0 > 0 ? db[e] + to_f32(oldv[e]) : db[e]);
                nv[e] = q;
            }
            *dbp = nv;
        }
    }
}

template <typename T, int MAXIT, int SPLIT = 1>
void launch_bwd_it(const LcBwdArgs &ba, unsigned blocks, hipStream_t st) {
    const int B = ba.f.B;
    for (int b0 = 0; b0 < B; b0 += 4) {
        const int nb = B - b0 < 4 ? B - b0 : 4;
#define NRT_LCB(NBV)                                                                                               \
        if (ba.dx) hipLaunchKernelGGL((lc3d_bwd<T, MAXIT, NBV, true, SPLIT>), dim3(blocks), dim3(256), 0, st, ba, b0, nb); \
        else hipLaunchKernelGGL((lc3d_bwd<T, MAXIT, NBV, false, SPLIT>), dim3(blocks), dim3(256), 0, st, ba, b0, nb)
        if (nb == 1) { NRT_LCB(1); } else if (nb == 2) { NRT_LCB(2); } else { NRT_LCB(4); }
#undef NRT_LCB
    }
}

template <typename T>
int launch_bwd(const LcBwdArgs &ba, hipStream_t st) {
    const LcArgs &a = ba.f;
    constexpr int VEC = 16 / (int)sizeof(T);
    const int F = a.kr * a.kc * a.kz * a.Cin;
    if (a.Cout % VEC) return NRT_ERR_UNSUPPORTED;
    const int LPR = a.Cout / VEC;
    if (LPR < 1 || LPR > 64 || (LPR & (LPR - 1))) return NRT_ERR_UNSUPPORTED;
    const int nit = (F + (64 / LPR) - 1) / (64 / LPR);
    if (nit > 64 || (long long)F * a.Cout * (long long)sizeof(T) >= (1ll << 31)) return NRT_ERR_UNSUPPORTED;
    if ((((uintptr_t)a.k | (uintptr_t)ba.g | (uintptr_t)a.y | (uintptr_t)ba.dk) & 15) != 0) return NRT_ERR_UNSUPPORTED;
    const long long O = (long long)a.orr * a.occ * a.ozz;
    unsigned blocks = (unsigned)((O + 3) / 4);
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    if ((long long)a.R * a.C * a.Z * a.Cin * (long long)sizeof(T) >= (1ll << 31)) return NRT_ERR_UNSUPPORTED;       // 32-bit buffer offsets
    if (nit <= 8) launch_bwd_it<T, 8>(ba, blocks, st);
    else if (nit <= 14) launch_bwd_it<T, 14>(ba, blocks, st);
    else if (nit <= 16) launch_bwd_it<T, 16>(ba, blocks, st);
    else if (nit <= 32) launch_bwd_it<T, 16, 2>(ba, (unsigned)min((long long)256 * 16, (O + 1) / 2), st);      // two waves per position
    else launch_bwd_it<T, 16, 4>(ba, (unsigned)min((long long)256 * 16, O), st);                             // four
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

}  // namespace

extern "C" int nrt_lc3d_bwd_f(const void *x, const void *kernel, const void *y, const void *grad_out, void *grad_kernel,
                              void *grad_bias, float *grad_x, int dtype, int batch, const int *in_shape, int cin,
                              const int *ksize, const int *strides, int cout, int activation, void *stream) {
    if (!x || !kernel || !grad_out || !in_shape || !ksize || !strides) return NRT_ERR_INVALID_ARG;
    if (!grad_kernel && !grad_bias && !grad_x) return NRT_ERR_INVALID_ARG;
    if (activation != 0 && !y) return NRT_ERR_INVALID_ARG;
    if (batch < 1 || cin < 1 || cout < 1) return NRT_ERR_INVALID_ARG;
    if (dtype != NRT_DT_F32 && dtype != NRT_DT_BF16) return NRT_ERR_UNSUPPORTED;
    LcBwdArgs ba;
    LcArgs &a = ba.f;
    a.x = x; a.k = kernel; a.bias = nullptr; a.y = (void *)y; a.B = batch;
    a.R = in_shape[0]; a.C = in_shape[1]; a.Z = in_shape[2]; a.Cin = cin;
    a.kr = ksize[0]; a.kc = ksize[1]; a.kz = ksize[2]; a.sr = strides[0]; a.sc = strides[1]; a.sz = strides[2];
    if (a.kr < 1 || a.kc < 1 || a.kz < 1 || a.sr < 1 || a.sc < 1 || a.sz < 1) return NRT_ERR_INVALID_ARG;
    if (a.R < a.kr || a.C < a.kc || a.Z < a.kz) return NRT_ERR_INVALID_ARG;
    a.orr = (a.R - a.kr) / a.sr + 1; a.occ = (a.C - a.kc) / a.sc + 1; a.ozz = (a.Z - a.kz) / a.sz + 1;
    a.Cout = cout; a.act = activation; a.stage_chunks = 0;
    ba.g = grad_out; ba.dk = grad_kernel; ba.dbias = grad_bias; ba.dx = grad_x;
    hipStream_t st = nrt_stream(stream);
    if (dtype == NRT_DT_F32) return launch_bwd<float>(ba, st);
    return launch_bwd<unsigned short>(ba, st);
}

namespace {
}  // namespace

extern "C" int nrt_lc3d_f(const void *x, const void *kernel, const void *bias, void *y, int dtype, int batch,
                          const int *in_shape, int cin, const int *ksize, const int *strides, int cout, int activation,
                          int variant, void *stream) {
    if (!x || !kernel || !y || !in_shape || !ksize || !strides) return NRT_ERR_INVALID_ARG;
    if (batch < 1 || cin < 1 || cout < 1) return NRT_ERR_INVALID_ARG;
    if (dtype != NRT_DT_F32 && dtype != NRT_DT_BF16) return NRT_ERR_UNSUPPORTED;
    if (activation < ACT_NONE || activation > ACT_LAST_FUSED) return NRT_ERR_INVALID_ARG;   // the epilogue fuses none / elu / relu only
    LcArgs a;
    a.x = x; a.k = kernel; a.bias = bias; a.y = y; a.B = batch;
    a.R = in_shape[0]; a.C = in_shape[1]; a.Z = in_shape[2]; a.Cin = cin;
    a.kr = ksize[0]; a.kc = ksize[1]; a.kz = ksize[2]; a.sr = strides[0]; a.sc = strides[1]; a.sz = strides[2];
    if (a.kr < 1 || a.kc < 1 || a.kz < 1 || a.sr < 1 || a.sc < 1 || a.sz < 1) return NRT_ERR_INVALID_ARG;
    if (a.R < a.kr || a.C < a.kc || a.Z < a.kz) return NRT_ERR_INVALID_ARG;
    a.orr = (a.R - a.kr) / a.sr + 1; a.occ = (a.C - a.kc) / a.sc + 1; a.ozz = (a.Z - a.kz) / a.sz + 1;   // 'valid'
    a.Cout = cout; a.act = activation; a.stage_chunks = 0;
    hipStream_t st = nrt_stream(stream);
    if (dtype == NRT_DT_F32) return launch_any<float>(a, variant, st);
    return launch_any<unsigned short>(a, variant, st);
}

// ---- zero padding / cropping of a channels-last volume ('same' padding of implementations 2 / 3) --------------------
// LocallyConnected3D with padding='same' (neurite/tf/layers.py:1474-1482: conv_connected_inputs clips the window at the
// volume border) = the 'valid' layer on the input zero-padded by k//2 voxels in front: the taps that fall into the padding
// multiply zeros.  One thread per 16-bit or 32-bit unit of a voxel row; rows outside the source box are written as zero.
// crop != 0 runs the copy the other way round (gradient of the padding: the interior of the padded gradient).
namespace {
template <typename U>
__global__ __launch_bounds__(256) void pad3d_rows(const U *__restrict__ in, U *__restrict__ out, int S0, int S1, int S2,
                                                   int P0, int P1, int P2, int O0, int O1, int O2, int units, int crop,
                                                   unsigned long long total) {
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x; e < total;
         e += (unsigned long long)gridDim.x * 256u) {
        unsigned long long r = e;
        const int u = (int)(r % (unsigned)units); r /= (unsigned)units;
        // the index space is the destination: the padded volume, or (crop) the un-padded one
        const int D0 = crop ? S0 : O0, D1 = crop ? S1 : O1, D2 = crop ? S2 : O2;
        const int z = (int)(r % (unsigned)D2); r /= (unsigned)D2;
        const int y = (int)(r % (unsigned)D1); r /= (unsigned)D1;
        const int x = (int)(r % (unsigned)D0); r /= (unsigned)D0;
        const long long b = (long long)r;
        if (crop) {
            const long long src = (((b * O0 + (x + P0)) * O1 + (y + P1)) * O2 + (z + P2)) * units + u;
            out[e] = in[src];
        } else {
            const int sx = x - P0, sy = y - P1, sz = z - P2;
            const bool inside = sx >= 0 && sx < S0 && sy >= 0 && sy < S1 && sz >= 0 && sz < S2;
            out[e] = inside ? in[(((b * S0 + sx) * S1 + sy) * S2 + sz) * units + u] : (U)0;
        }
    }
}
}  // namespace

extern "C" int nrt_pad3d(const void *in, void *out, int batch, const int *in_shape, const int *pad_before, const int *out_shape,
                         int row_bytes, int crop, void *stream) {
    if (!in || !out || !in_shape || !pad_before || !out_shape || batch < 1 || row_bytes < 2 || (row_bytes & 1))
        return NRT_ERR_INVALID_ARG;
    for (int d = 0; d < 3; ++d)
        if (in_shape[d] < 1 || pad_before[d] < 0 || out_shape[d] < in_shape[d] + pad_before[d]) return NRT_ERR_INVALID_ARG;
    const int *D = crop ? in_shape : out_shape;
    const bool wide = (row_bytes % 4) == 0 && (((uintptr_t)in | (uintptr_t)out) & 3) == 0;
    const int units = wide ? row_bytes / 4 : row_bytes / 2;
    const unsigned long long total = (unsigned long long)batch * D[0] * D[1] * D[2] * units;
    unsigned long long nb = (total + 255) / 256;
    if (nb > (1ull << 20)) nb = 1ull << 20;
    if (nb == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    if (wide)
        hipLaunchKernelGGL((pad3d_rows<unsigned>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned *)in, (unsigned *)out,
                           in_shape[0], in_shape[1], in_shape[2], pad_before[0], pad_before[1], pad_before[2], out_shape[0],
                           out_shape[1], out_shape[2], units, crop, total);
    else
        hipLaunchKernelGGL((pad3d_rows<unsigned short>), dim3((unsigned)nb), dim3(256), 0, st, (const unsigned short *)in,
                           (unsigned short *)out, in_shape[0], in_shape[1], in_shape[2], pad_before[0], pad_before[1],
                           pad_before[2], out_shape[0], out_shape[1], out_shape[2], units, crop, total);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
