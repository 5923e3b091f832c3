// Decoder convolution of models.unet with the up-sampled half folded -- included by conv.hip inside its anonymous
// namespace (uses ConvArgs, f32x4, CT_X/Y/Z, LDS_ROW, activate).
//
//   y = act(conv3x3x3_SAME(concat(skip [.., c0], UpSampling3D(2)(lo [.., c1]))) + bias)      neurite/tf/models.py:1531-1555
//
// Folding.  An output voxel o = 2q + p (p = its parity along one axis) sees the nearest-up-sampled tensor at o-1, o, o+1,
// i.e. the low-resolution voxels (q-1, q, q) for p = 0 and (q, q, q+1) for p = 1: per axis the three taps collapse to TWO
// taps on the low-resolution grid with weights (w0, w1+w2) resp. (w0+w1, w2); the SAME zero padding of the up-sampled grid
// is zero padding of the low-resolution grid.  In 3-D the 27 taps over the c1 up-sampled channels become 8 taps with one
// of 8 pre-summed weight sets (by output parity): K = 27 c0 + 8 c1 instead of 27 (c0 + c1) -- 0.53 of the matrix work at
// neurite's decoder shapes (c1 = 2 c0).  The sums of up to 8 weights are formed once per layer in float32.
//
// Tiling.  An MFMA M-tile must hold voxels of ONE parity class: tile = 4 x 4 x 16 outputs, wave w = the (x, y) parity
// (w & 1, w >> 1), its 4 M-tiles = (x in {px, px+2}) x (z parity), rows = (y in {py, py+2}) x (8 z of that parity).
// K is walked in chunks of 16 channels, "A" chunks (up-sampled channels, halo [4][4][10] rows of the low-resolution
// tensor) and "B" chunks (skip channels, halo [6][6][18] rows stored with z de-interleaved by parity), interleaved
// A A B A A B .. so that a B chunk never follows a B chunk.
//
// Schedule.  Blocks are persistent (2 per CU, 80 KB of LDS each) and walk tiles of one XCD's contiguous range.  Halo
// tiles go global -> LDS by DMA (global_load_lds_dwordx4: no staging registers, no ds_write), out-of-volume rows read a
// 64-byte zero block behind the packed weights.  LDS holds one B buffer and two A buffers, so the DMA of chunk k+1 --
// also across the tile boundary -- is issued when chunk k starts and has the whole chunk to land: one barrier per chunk.
// The outputs of a tile are kept in registers and stored behind the first DMA of the NEXT tile; a block never drains.
// All vector-memory instructions are inline asm in a fixed order with hand-counted s_waitcnt vmcnt(N) immediates (vmcnt
// retires in order; the compiler would have to wait for the DMA whenever it waits for a weight fragment).
//
// HNT > 0 (round 5): the LAST decoder convolution with the network's head folded in -- models.py:1596-1605, the 1 x 1 x 1 "likelihood"
// convolution to 16 HNT labels and the channel soft-max.  The tile's 16 activated feature channels never leave the block: while the
// next tile's first chunk is landing, each wave transposes its four 16 x 16 accumulator tiles through (its own 5 KB of) the idle
// skip-halo buffer into A-operand order, multiplies them by the head's [16, 16 HNT] matrix on the matrix cores (4 HNT MFMAs per
// tile of 16 voxels), takes the soft-max across the 16 lanes x HNT registers that hold a voxel's labels (DPP row rotations) and
// stores the probabilities in the deferred-store slot.  Saved at 160^3 x 16 -> 32: the 262 MB feature tensor written and read back,
// and the head kernel itself (0.167 ms of a 1.78 ms forward).  Inference only: training needs the feature tensor for the backward.

constexpr int U2_SY = 2 * 9 * LDS_ROW + 8, U2_SX = 6 * U2_SY;        // skip halo: [6][6][parity][9] rows, y-stride padded (floats)
constexpr int U2_SYA = 10 * LDS_ROW + 24, U2_SXA = 4 * U2_SYA;       // low-resolution halo: [4][4][10] rows
constexpr int U2_B_FLOATS = 6 * U2_SX, U2_A_FLOATS = 4 * U2_SXA;
constexpr int U2_NDB = (U2_B_FLOATS / 4 + 255) / 256;                // DMA instructions per wave and B chunk (13)
constexpr int U2_NDA = (U2_A_FLOATS / 4 + 255) / 256;                // .. and A chunk (4; the last one is half empty)
constexpr int U2_GAP_FLOATS = U2_NDB * 1024 - U2_B_FLOATS;           // the last B instruction overruns the buffer by 16 lanes
constexpr int U2_OFF_A = U2_B_FLOATS + U2_GAP_FLOATS;
constexpr int U2_LDS_FLOATS = U2_OFF_A + 2 * U2_A_FLOATS;
constexpr int U2_ZERO_FLOATS = 64;                                   // zero block appended to the packed weights
static_assert((2 * U2_SY) % 64 == 32 && U2_SYA % 64 == 32, "row groups must keep the 20-float bank walk of 16 consecutive rows");
static_assert(U2_LDS_FLOATS * 4 <= 81920, "two blocks per CU");
static_assert(U2_A_FLOATS / 4 == 3 * 256 + 128, "the A halo is 3.5 DMA rounds: waves 2, 3 repeat round 2 in round 3");

#ifndef U2_DEFER_MAXNT
#define U2_DEFER_MAXNT 1                                              // deferred stores cost 16 NT registers + their address math
#endif
#ifndef U2_WDIST
#define U2_WDIST 2                                                   // weight fragments are requested this many taps ahead
#endif

__device__ __forceinline__ void u2_dma16(const float *g, unsigned lds_bytes) {     // LDS[lds_bytes + 16 lane ..] = 16 bytes at g
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_bytes) : "memory");
}
__device__ __forceinline__ f32x4 u2_ldw(const void *sbase, unsigned voff) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase));
    return r;
}
// The same load with five wait states in front: an SGPR written by a VALU instruction (v_readlane: that is how the compiler brings back a
// spilled scalar) may not be the address of a vector-memory instruction for five cycles, and the compiler cannot see that the asm
// statement is one.  The head's pointers are used twice per tile, long after they were loaded: in the 16-label instantiation they came
// back from their spill lanes right in front of the load and it faulted on a garbage address.  (The weight-fragment loads of the
// chunks take SALU-computed bases; twice per tile the nops cost nothing.)
__device__ __forceinline__ f32x4 u2_ldw_s(const void *sbase, unsigned voff) {
    f32x4 r;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase));
    return r;
}
// (s_nop on both sides: a vector-memory store of more than 64 bits reads its data registers over several cycles -- a VALU instruction
// may neither have written them in the two cycles before nor overwrite them in the two cycles after.  The compiler's hazard recogniser
// keeps those distances for its own stores and cannot see into an asm statement: without the nops the first component of the
// quadruple reached memory corrupted in a few lanes, differently from run to run.)
__device__ __forceinline__ void u2_store4(const void *sbase, unsigned voff, f32x4 v) {
    asm volatile("s_nop 1\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void u2_store(const void *sbase, unsigned voff, float v) {
    asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// the fragment registers are tied into the wait so that no consumer can be scheduled above it
template <int N>
__device__ __forceinline__ void u2_wait(f32x4 &r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N < 63 ? N : 63)); }
template <int... I, class Fn>
__device__ __forceinline__ void u2_static_for(std::integer_sequence<int, I...>, Fn &&f) { (f(std::integral_constant<int, I>{}), ...); }
__device__ __forceinline__ void u2_tie(f32x4 &r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void u2_chunk_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : : : "memory");
}

// vector-memory instructions issued behind the fragments of step s when step s is waited for:
// order = P steps up front, D DMA, S stores, then step q >= P at step max(0, q - U2_WDIST); F fragments per step, NS steps
__host__ __device__ constexpr int u2_newer(int s, int P, int F, int D, int S, int NS) {
    const int hi = s + U2_WDIST < NS - 1 ? s + U2_WDIST : NS - 1;            // last step already requested
    if (s < P) return (P - 1 - s) * F + D + S + (hi >= P ? hi - P + 1 : 0) * F;
    return (hi - s) * F;
}

__device__ __forceinline__ float u2_row_ror(float v, const int n) {      // v of lane (i + n) % 16 inside each row of 16 lanes
    switch (n) {
        case 8: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
        case 4: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
        case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
        default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
    }
}

// order of a tile's chunks: bit k set = chunk k is a B (skip) chunk, the nB of them spread evenly among the nA + nB (Bresenham)
__host__ __device__ inline unsigned u2_chunk_seq(int nA, int nB) {
    const int n = nA + nB;
    unsigned seq = 0;
    for (int k = 0; k < n; ++k) seq |= (((k + 1) * nB) / n > (k * nB) / n ? 1u : 0u) << k;
    return seq;
}
// What the folded head needs of that order (it was implied by c0 < c1, now it is checked where the kernel is chosen): its
// scratch aliases the skip-halo buffer and its stores run in chunk 0 behind that chunk's `issue_next` -- so chunk 0 must be an A chunk
// (the deferred stores live in bodyA) AND chunk 1 too (else chunk 0 requests the next skip halo into the scratch before the head has
// read it), and the tile must end on a B chunk (the window "last B chunk read .. next B chunk requested" is what the scratch lives in).
inline bool u2_head_order_ok(int nA, int nB) {
    const int n = nA + nB;
    if (nA < 2 || nB < 1 || n > 32) return false;
    const unsigned seq = u2_chunk_seq(nA, nB);
    return (seq & 3u) == 0u && ((seq >> (n - 1)) & 1u) == 1u;
}

struct U2Head { const float *w, *bias; int labels; };     // the 1x1x1 head: kernel in fragment order [labels / 16][64 lanes][4], bias [labels]

template <int NT, int HNT = 0>
__global__ __launch_bounds__(256, NT <= 2 ? 2 : 1) void conv3d_up2_mfma(ConvArgs a, const float *__restrict__ wpacked, const float *__restrict__ zeros,
                                                       unsigned ntiles, unsigned nbx, unsigned nby, unsigned nbz, U2Head head) {
    static_assert(HNT == 0 || NT == 1, "the folded head takes the 16 feature channels of one N-tile");
    constexpr int OC = HNT ? 16 * HNT : 0;                              // output channels per voxel when the head is folded in
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int px = w & 1, py = w >> 1, iy = li >> 3, iz = li & 7;
    const int nA = a.c1 >> 4, nB = a.c0 >> 4, n = nA + nB;
    const unsigned seq = u2_chunk_seq(nA, nB);                          // bit k: chunk k of a tile is a B chunk (Bresenham)

    // ---- tiles of this persistent block: XCD x owns the x-th contiguous eighth, its blocks interleave --------------
    const unsigned xcd = blockIdx.x % NRT_NXCD, J = gridDim.x / NRT_NXCD;
    const unsigned T8 = (ntiles + NRT_NXCD - 1) / NRT_NXCD;
    const unsigned tend = (xcd + 1) * T8 < ntiles ? (xcd + 1) * T8 : ntiles;
    unsigned tile = xcd * T8 + blockIdx.x / NRT_NXCD;
    if (tile >= tend) return;

    // ---- what this thread moves per chunk: 16-byte pieces g = 256 i + tid of the LDS image ---------------------------
    const int sZ0 = a.c0, sY0 = a.Z * sZ0, sX0 = a.Y * sY0;
    const int tZ = a.c1, tY = a.Z1 * tZ, tX = a.Y1 * tY;
    int relB[U2_NDB], relA[U2_NDA];
    unsigned validB = 0, validA = 0;
    auto pieceB = [&](int i, int &lx, int &ly, int &lz, int &c) __attribute__((always_inline)) {      // halo row and float4 of piece i
        const int g = i * 256 + threadIdx.x;
        const int yg = g / 92, rem = g % 92, rr = rem / 5;
        c = rem % 5; lx = yg / 6; ly = yg % 6; lz = 2 * (rr % 9) + rr / 9;
        return g < U2_B_FLOATS / 4 && rem < 90 && c < 4;
    };
    auto pieceA = [&](int i, int &lx, int &ly, int &lz, int &c) __attribute__((always_inline)) {
        const int g = (i == 3 && w >= 2 ? 2 : i) * 256 + threadIdx.x;
        const int grp = g / 56, rem = g % 56;
        lz = rem / 5; c = rem % 5; lx = grp / 4; ly = grp % 4;
        return rem < 50 && c < 4;
    };
#pragma unroll
    for (int i = 0; i < U2_NDB; ++i) {
        int lx, ly, lz, c;
        validB |= (pieceB(i, lx, ly, lz, c) ? 1u : 0u) << i;
        relB[i] = (lx - 1) * sX0 + (ly - 1) * sY0 + (lz - 1) * sZ0 + 4 * c;
    }
#pragma unroll
    for (int i = 0; i < U2_NDA; ++i) {
        int lx, ly, lz, c;
        validA |= (pieceA(i, lx, ly, lz, c) ? 1u : 0u) << i;
        relA[i] = (lx - 1) * tX + (ly - 1) * tY + (lz - 1) * tZ + 4 * c;
    }
    struct Tile {
        const float *pB, *pA;        // first channel of the tile origin in skip / lo
        unsigned okB, okA;           // per DMA piece: inside the volume
        unsigned out;                // byte offset of output voxel (x0, y0, z0), channel 0
        int x0, y0, z0;
        unsigned full;               // whole tile inside the volume: its stores need no guards and can be deferred
    };
    auto decode = [&](unsigned t) __attribute__((always_inline)) {
        Tile T;
        const int bz = t % nbz, by = (t / nbz) % nby, bx = (t / (nbz * nby)) % nbx, b = t / (nbz * nby * nbx);
        T.x0 = bx * CT_X; T.y0 = by * CT_Y; T.z0 = bz * CT_Z;
        T.pB = a.src0 + ((long long)b * a.X * a.Y * a.Z + ((long long)T.x0 * a.Y + T.y0) * a.Z + T.z0) * a.c0;
        T.pA = a.src1 + ((long long)b * a.X1 * a.Y1 * a.Z1 + ((long long)(T.x0 >> 1) * a.Y1 + (T.y0 >> 1)) * a.Z1 + (T.z0 >> 1)) * a.c1;
        T.out = (unsigned)((((long long)b * a.OX + T.x0) * a.OY + T.y0) * a.OZ + T.z0) * (unsigned)(HNT ? OC : a.Cout) * 4u;
        T.full = HNT ? 1u : (T.x0 + CT_X <= a.OX && T.y0 + CT_Y <= a.OY && T.z0 + CT_Z <= a.OZ && (a.Cout & 15) == 0 &&
                 NT <= U2_DEFER_MAXNT);                                 // 16 NT deferred stores must fit the 6-bit vmcnt with the loads around them
                                                                        // (head folded in: launch_up2_head only takes volumes of whole tiles)
        const bool inner = T.x0 >= 1 && T.y0 >= 1 && T.z0 >= 1 && T.x0 + CT_X + 1 <= a.X && T.y0 + CT_Y + 1 <= a.Y && T.z0 + CT_Z + 1 <= a.Z;
        T.okB = validB; T.okA = validA;
        if (!inner) {
            T.okB = T.okA = 0;
#pragma unroll
            for (int i = 0; i < U2_NDB; ++i) {
                int lx, ly, lz, c;
                pieceB(i, lx, ly, lz, c);
                const unsigned x = T.x0 - 1 + lx, y = T.y0 - 1 + ly, z = T.z0 - 1 + lz;
                T.okB |= (x < (unsigned)a.X && y < (unsigned)a.Y && z < (unsigned)a.Z ? 1u : 0u) << i;
            }
#pragma unroll
            for (int i = 0; i < U2_NDA; ++i) {
                int lx, ly, lz, c;
                pieceA(i, lx, ly, lz, c);
                const unsigned x = (T.x0 >> 1) - 1 + lx, y = (T.y0 >> 1) - 1 + ly, z = (T.z0 >> 1) - 1 + lz;
                T.okA |= (x < (unsigned)a.X1 && y < (unsigned)a.Y1 && z < (unsigned)a.Z1 ? 1u : 0u) << i;
            }
            T.okB &= validB; T.okA &= validA;
        }
        return T;
    };
    auto issueB = [&](const float *p, unsigned ok) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < U2_NDB; ++i) u2_dma16(((ok >> i) & 1u) ? p + relB[i] : zeros, lds0 + (i * 256 + w * 64) * 16);
    };
    auto issueA = [&](const float *p, unsigned ok, unsigned buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < U2_NDA; ++i)
            u2_dma16(((ok >> i) & 1u) ? p + relA[i] : zeros,
                     lds0 + (U2_OFF_A + buf * U2_A_FLOATS) * 4 + ((i == 3 && w >= 2 ? 2 : i) * 256 + w * 64) * 16);
    };

    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (a.bias && nt * 16 + li < a.Cout) ? a.bias[nt * 16 + li] : 0.0f;
    // the head's fragments: step s of the 16-channel contraction takes channels 4 kq + s (so that a lane's four feature values are ONE
    // ds_read_b128), rows = labels 16 j + li.  They are (re)loaded at the top of the chunk that runs the head -- asm loads issued
    // BEFORE that chunk's weight fragments, so the hand-counted waits behind them are unchanged -- and are dead everywhere else: held
    // across the skip chunk (108 fragment registers) they pushed the 32-label kernel over 256 registers
    constexpr int HN = HNT ? HNT : 1;
    f32x4 hw[HN], hb[HN];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[nt]));      // the only compiler-visible load: settled here
    auto head_request = [&]() __attribute__((always_inline)) {
        if constexpr (HNT != 0) {
#pragma unroll
            for (int j = 0; j < HN; ++j) {
                // (whole 16-byte registers straight from the asm loads: a value assembled from four scalar loads is COPIED by the
                // compiler right behind the load instructions, before the data is there -- the first version of this did that)
                hw[j] = u2_ldw_s(head.w, (unsigned)((j * 64 + lane) * 16));                              // fragment order (nrt_conv3d_up2_head_pack_f32)
                hb[j] = u2_ldw_s(head.bias, (unsigned)((16 * j + 4 * kq) * 4));                          // labels 16 j + 4 kq .. + 3: this lane's rows
            }
        }
    };
    f32x4 acc[4][NT];
    float outv[4][NT][4];                                               // the previous tile's outputs until they are stored
    unsigned outBase = 0;
    bool pending = false;
    // output offsets of this lane inside a tile (bytes): rows = (y pair member, z of the parity), columns = channels
    const unsigned ocb = (unsigned)(HNT ? OC : a.Cout) * 4u;            // bytes of an output voxel's row
    const unsigned oY = (unsigned)a.OZ * ocb, oX = (unsigned)a.OY * oY;
    const unsigned outLane = px * oX + (py + 2 * (kq >> 1)) * oY + (2 * (kq & 1) * 4) * ocb + li * 4u;
    // the head's scratch: 4 x [16 voxels][20 floats] per wave at the start of the skip-halo buffer, which nobody reads between the
    // last chunk of a tile (a B chunk) and the DMA of the next tile's first B chunk -- the head runs inside that window
    float *hscr = &lds[w * 4 * 16 * LDS_ROW];
    static_assert(4 * 4 * 16 * LDS_ROW <= U2_B_FLOATS, "head scratch fits the skip-halo buffer");
    auto stores = [&]() __attribute__((always_inline)) {
        if constexpr (HNT == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        u2_store(a.out, outBase + outLane + 2 * (mt & 1) * oX + (2 * r + (mt >> 1)) * ocb + nt * 64u, outv[mt][nt][r]);
        } else {
            // features [voxel 4 kq + r][channel li] -> LDS rows of 20 floats -> fragments [voxel li][channels 4 kq .. 4 kq + 3]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) hscr[(mt * 16 + 4 * kq + r) * LDS_ROW + li] = outv[mt][0][r];
            // logits TRANSPOSED: D[label 16 j + 4 kq + r][voxel li] = sum_c W[c][label] Y[voxel][c] -- the head's matrix is the A operand,
            // the features the B operand.  A voxel's labels then sit in 4 HN registers of the 4 lanes li, li + 16, li + 32, li + 48:
            // the soft-max is 4 HN - 1 in-lane operations and two cross-row steps instead of a 16-lane reduction per row (a third of
            // the VALU work), and a lane's four labels are consecutive in memory: one 16-byte store
            const unsigned outLaneT = px * oX + (py + 2 * (li >> 3)) * oY + 2 * (li & 7) * ocb + 16u * kq;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                // (one tile at a time: the stores below are asm with a memory clobber, so the read of tile mt + 1 stays behind them
                // and the four fragments are never live together -- the kernel sits at the 256-register limit of two blocks per CU)
                const f32x4 ya = *(const f32x4 *)&hscr[(mt * 16 + li) * LDS_ROW + 4 * kq];
                f32x4 d[HN];
#pragma unroll
                for (int j = 0; j < HN; ++j) d[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                    for (int j = 0; j < HN; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(hw[j][sidx], ya[sidx], d[j], 0, 0, 0);
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < HN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { d[j][r] += hb[j][r]; m = fmaxf(m, d[j][r]); }
                m = fmaxf(m, __shfl_xor(m, 16, NRT_WAVE));
                m = fmaxf(m, __shfl_xor(m, 32, NRT_WAVE));
                float se = 0.0f;
#pragma unroll
                for (int j = 0; j < HN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { d[j][r] = softmax_exp(d[j][r] - m); se += d[j][r]; }
                se += __shfl_xor(se, 16, NRT_WAVE);
                se += __shfl_xor(se, 32, NRT_WAVE);
                const float inv = softmax_rcp(se);
#pragma unroll
                for (int j = 0; j < HN; ++j)
                    u2_store4(a.out, outBase + outLaneT + 2 * (mt & 1) * oX + (mt >> 1) * ocb + j * 64u,
                              (f32x4){d[j][0] * inv, d[j][1] * inv, d[j][2] * inv, d[j][3] * inv});
            }
        }
    };

    const unsigned wlane = lane * 16u;
    const char *wA0 = (const char *)wpacked + (size_t)w * 16 * NT * 1024;              // [chunk][wave][step][z parity][nt] KB
    const char *wB0 = (const char *)wpacked + (size_t)nA * 4 * 16 * NT * 1024;         // [chunk][tap][nt] KB

    // ---- one A chunk: 8 folded taps on the low-resolution halo ----------------------------------------------------------
    auto bodyA = [&](int ia, unsigned buf, auto Dc, auto Sc, auto &&issue_next) __attribute__((always_inline)) {
        constexpr int D = decltype(Dc)::value, S = decltype(Sc)::value;
        constexpr int P = NT == 1 ? 4 : 1, F = 2 * NT;
        const char *wp = wA0 + (size_t)ia * 4 * 16 * NT * 1024;
        f32x4 bq[8][2][NT];
        auto request = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[q][pz][nt] = u2_ldw(wp + ((q * 2 + pz) * NT + nt) * 1024, wlane);
        };
        if (S) head_request();
#pragma unroll
        for (int q = 0; q < P; ++q) request(q);
        issue_next();
        if (S) {
            if constexpr (HNT != 0) {                                    // the head's fragments: everything issued behind them may stay in flight
                u2_wait<P * F + D>(hw[0]);
#pragma unroll
                for (int j = 0; j < HN; ++j) { u2_tie(hw[j]); u2_tie(hb[j]); }
            }
            stores();
        }
        const float *abase = &lds[U2_OFF_A + buf * U2_A_FLOATS + px * U2_SXA + (py + iy) * U2_SYA + iz * LDS_ROW + 4 * kq];
        f32x4 avq[2][4];
        auto read = [&](int s) __attribute__((always_inline)) {
            const int tz = s & 1, ty = (s >> 1) & 1, tx = s >> 2;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                avq[s & 1][mt] = *(const f32x4 *)(abase + ((mt & 1) + tx) * U2_SXA + ty * U2_SYA + ((mt >> 1) + tz) * LDS_ROW);
        };
        read(0);
        u2_static_for(std::make_integer_sequence<int, 8>{}, [&](auto Sx) __attribute__((always_inline)) {
            constexpr int s = decltype(Sx)::value;
            if constexpr (s == 0)
                u2_static_for(std::make_integer_sequence<int, (U2_WDIST > P ? U2_WDIST - P : 0)>{},
                              [&](auto Q) __attribute__((always_inline)) { request(P + decltype(Q)::value); });
            if constexpr (s + U2_WDIST >= P && s + U2_WDIST < 8) request(s + U2_WDIST);
            if constexpr (s + 1 < 8) read(s + 1);                        // LDS fragments one step ahead of their MFMAs
            f32x4 (&av)[4] = avq[s & 1];
            u2_wait<u2_newer(s, P, F, D, S, 8)>(bq[s][0][0]);
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (pz + nt) u2_tie(bq[s][pz][nt]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][m], bq[s][mt >> 1][nt][m], acc[mt][nt], 0, 0, 0);
        });
    };
    // ---- one B chunk: the 27 taps on the full-resolution halo ---------------------------------------------------------------
    auto bodyB = [&](int ib, auto Dc, auto &&issue_next) __attribute__((always_inline)) {
        constexpr int D = decltype(Dc)::value;
        constexpr int P = NT == 1 ? 6 : NT == 2 ? 3 : 2, F = NT;
        const char *wp = wB0 + (size_t)ib * 27 * NT * 1024;
        f32x4 bq[27][NT];
        auto request = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[q][nt] = u2_ldw(wp + (q * NT + nt) * 1024, wlane);
        };
#pragma unroll
        for (int q = 0; q < P; ++q) request(q);
        issue_next();
        const float *abase = &lds[px * U2_SX + (py + 2 * iy) * U2_SY + iz * LDS_ROW + 4 * kq];
        f32x4 avq[2][4];
        auto read = [&](int t) __attribute__((always_inline)) {
            const int dz = t % 3, dy = (t / 3) % 3, dx = t / 9;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ix = mt & 1, pz = mt >> 1;
                avq[t & 1][mt] = *(const f32x4 *)(abase + (2 * ix + dx) * U2_SX + dy * U2_SY + ((pz + dz) & 1) * (9 * LDS_ROW) +
                                                  ((pz + dz) >> 1) * LDS_ROW);
            }
        };
        read(0);
        u2_static_for(std::make_integer_sequence<int, 27>{}, [&](auto Tx) __attribute__((always_inline)) {
            constexpr int t = decltype(Tx)::value;
            if constexpr (t == 0)
                u2_static_for(std::make_integer_sequence<int, (U2_WDIST > P ? U2_WDIST - P : 0)>{},
                              [&](auto Q) __attribute__((always_inline)) { request(P + decltype(Q)::value); });
            if constexpr (t + U2_WDIST >= P && t + U2_WDIST < 27) request(t + U2_WDIST);
            if constexpr (t + 1 < 27) read(t + 1);
            f32x4 (&av)[4] = avq[t & 1];
            u2_wait<u2_newer(t, P, F, D, 0, 27)>(bq[t][0]);
#pragma unroll
            for (int nt = 1; nt < NT; ++nt) u2_tie(bq[t][nt]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][m], bq[t][nt][m], acc[mt][nt], 0, 0, 0);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using IA = std::integral_constant<int, U2_NDA>;
    using IB = std::integral_constant<int, U2_NDB>;
    using IS = std::integral_constant<int, HNT ? 4 * HNT : 16 * NT>;

    Tile cur = decode(tile);
    unsigned aIss = 0, aUse = 0;                                        // A chunks requested / consumed: buffer = count & 1
    issueA(cur.pA, cur.okA, aIss++ & 1u);                               // a tile starts with an A chunk (nA >= 1, nB >= 1)
    for (;;) {
        const unsigned ntile = tile + J;
        const bool has_next = ntile < tend;
        Tile nxt = cur;
        if (has_next) nxt = decode(ntile);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        int ia = 0, ib = 0;
        for (int k = 0; k < n; ++k) {
            const bool last = k + 1 == n;
            const bool curB = (seq >> k) & 1u, nextB = !last && ((seq >> (k + 1)) & 1u);
            u2_chunk_barrier();                                         // chunk k has landed; chunk k-1 has been read by all waves
            if (!curB) {                                                // an A chunk is never the last of a tile
                const unsigned buf = aUse++ & 1u;
                auto nextA = [&]() __attribute__((always_inline)) { issueA(cur.pA + 16 * (ia + 1), cur.okA, aIss++ & 1u); };
                auto nextBf = [&]() __attribute__((always_inline)) { issueB(cur.pB + 16 * ib, cur.okB); };
                if (k == 0 && pending) {
                    if (nextB) bodyA(ia, buf, IB{}, IS{}, nextBf);
                    else bodyA(ia, buf, IA{}, IS{}, nextA);
                    pending = false;
                } else {
                    if (nextB) bodyA(ia, buf, IB{}, I0{}, nextBf);
                    else bodyA(ia, buf, IA{}, I0{}, nextA);
                }
                ++ia;
            } else {
                const bool late = nextB;                                // B after B shares the buffer: requested after this chunk
                if (!late && (!last || has_next)) {
                    const Tile &T = last ? nxt : cur;
                    const int ian = last ? 0 : ia;
                    bodyB(ib, IA{}, [&]() __attribute__((always_inline)) { issueA(T.pA + 16 * ian, T.okA, aIss++ & 1u); });
                } else {
                    bodyB(ib, I0{}, [&]() {});
                }
                ++ib;
                if (late) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
                    issueB(cur.pB + 16 * ib, cur.okB);
                }
            }
        }
        // ---- the tile's outputs: kept for the deferred stores (whole tiles) or stored now (ragged tiles) ----------------
        if (cur.full) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) outv[mt][nt][r] = activate(acc[mt][nt][r] + bv[nt], a.act);
            outBase = cur.out;
            pending = true;
        } else {
            const int y = cur.y0 + py + 2 * (kq >> 1);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int x = cur.x0 + px + 2 * (mt & 1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int z = cur.z0 + 2 * ((kq & 1) * 4 + r) + (mt >> 1);
                        if (x < a.OX && y < a.OY && z < a.OZ && nt * 16 + li < a.Cout)
                            *(float *)((char *)a.out + cur.out + outLane + 2 * (mt & 1) * oX + (2 * r + (mt >> 1)) * (unsigned)a.Cout * 4u + nt * 64u) =
                                activate(acc[mt][nt][r] + bv[nt], a.act);
                    }
            }
        }
        if (!has_next) break;
        tile = ntile;
        cur = nxt;
    }
    if (pending) {
        if constexpr (HNT != 0) {
            head_request();
            u2_chunk_barrier();                                         // (vmcnt(0):) the fragments are here, and the last tile's skip halo has been
#pragma unroll
            for (int j = 0; j < HN; ++j) { u2_tie(hw[j]); u2_tie(hb[j]); }       // read by every wave: it becomes the scratch
        }
        stores();
    }
}

// folded + fragment-ordered weights of conv3d_up2_mfma:
//   [c1/16 chunks][wave parity (px + 2 py)][step (tx, ty, tz)][z parity][nt][lane][m]  then  [c0/16 chunks][27 taps][nt][lane][m]
//   then U2_ZERO_FLOATS zeros (what out-of-volume halo rows read)
__global__ void conv3d_pack_weights_up2(const float *__restrict__ w, int c0, int c1, int Cout, int NT, float *__restrict__ packed) {
    const int Cin = c0 + c1, nA = c1 / 16, nB = c0 / 16;
    const long long totA = (long long)nA * 4 * 16 * NT * 256, totB = (long long)nB * 27 * NT * 256;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < totA + totB + U2_ZERO_FLOATS; e += (long long)gridDim.x * blockDim.x) {
        if (e >= totA + totB) { packed[e] = 0.0f; continue; }
        const int m = e & 3, lane = (e >> 2) & 63;
        long long r = (e < totA ? e : e - totA) >> 8;
        const int nt = r % NT; r /= NT;
        const int co = nt * 16 + (lane & 15);
        float v = 0.0f;
        if (e < totA) {
            const int pz = r & 1, s = (r >> 1) & 7, wv = (r >> 4) & 3, ch = r >> 6;
            const int tz = s & 1, ty = (s >> 1) & 1, tx = s >> 2, px = wv & 1, py = wv >> 1;
            const int ci = c0 + ch * 16 + 4 * (lane >> 4) + m;
            // taps of one axis that land on low-resolution tap t for output parity p: p=0: {0}, {1,2};  p=1: {0,1}, {2}
            auto lo = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2); };
            auto hi = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2); };
            if (co < Cout)
                for (int dx = lo(px, tx); dx <= hi(px, tx); ++dx)
                    for (int dy = lo(py, ty); dy <= hi(py, ty); ++dy)
                        for (int dz = lo(pz, tz); dz <= hi(pz, tz); ++dz)
                            v += w[((long long)((dx * 3 + dy) * 3 + dz) * Cin + ci) * Cout + co];
        } else {
            const int t = r % 27, ch = r / 27;
            const int ci = ch * 16 + 4 * (lane >> 4) + m;
            if (co < Cout) v = w[((long long)t * Cin + ci) * Cout + co];
        }
        packed[e] = v;
    }
}

// head kernel [16][labels] (Keras layout) -> fragment order [labels / 16][lane][s] = W[4 (lane / 16) + s][16 j + lane % 16]
__global__ void conv3d_pack_head_up2(const float *__restrict__ w, int labels, float *__restrict__ packed) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 16 * labels; e += gridDim.x * blockDim.x) {
        const int sidx = e & 3, lane = (e >> 2) & 63, j = e >> 8;
        packed[e] = w[(4 * (lane >> 4) + sidx) * labels + 16 * j + (lane & 15)];
    }
}

size_t up2_weight_floats(int c0, int c1, int cout) {
    const size_t NT = (size_t)(cout + 15) / 16;
    return ((size_t)(c1 / 16) * 4 * 16 + (size_t)(c0 / 16) * 27) * NT * 256;
}

bool up2_ok(const ConvArgs &a, int padding_same) {
    return padding_same && a.kx == 3 && a.ky == 3 && a.kz == 3 && a.dil == 1 && a.c1 >= 16 && a.c0 >= 16 && a.ux == 2 && a.uy == 2 &&
           a.uz == 2 && a.c0 % 16 == 0 && a.c1 % 16 == 0 && (a.c0 + a.c1) / 16 <= 32 && a.Cout <= 64 &&
           (long long)a.X * a.Y * a.Z * a.c0 < (1ll << 30) && (long long)a.X * a.Y * a.Z * a.Cout < (1ll << 30);
}

template <int NT, int HNT = 0>
int launch_up2(const ConvArgs &a, const float *wpacked, int batch, hipStream_t st, U2Head head = U2Head{nullptr, nullptr, 0}) {
    const long long oc = HNT ? 16 * HNT : a.Cout;
    if ((long long)batch * a.X * a.Y * a.Z * oc >= (1ll << 30)) return NRT_ERR_UNSUPPORTED;     // 32-bit output offsets
    const unsigned nbx = (a.OX + CT_X - 1) / CT_X, nby = (a.OY + CT_Y - 1) / CT_Y, nbz = (a.OZ + CT_Z - 1) / CT_Z;
    const unsigned ntiles = nbx * nby * nbz * (unsigned)batch;
    if (hipFuncSetAttribute((const void *)conv3d_up2_mfma<NT, HNT>, hipFuncAttributeMaxDynamicSharedMemorySize, U2_LDS_FLOATS * 4) != hipSuccess)
        return NRT_ERR_LAUNCH;
    const unsigned T8 = (ntiles + NRT_NXCD - 1) / NRT_NXCD, per_xcd = 2u * (unsigned)nrt_num_cus() / NRT_NXCD;
    const unsigned J = T8 < per_xcd ? T8 : per_xcd;
    const float *zeros = wpacked + up2_weight_floats(a.c0, a.c1, a.Cout);
    hipLaunchKernelGGL((conv3d_up2_mfma<NT, HNT>), dim3(NRT_NXCD * J), dim3(256), U2_LDS_FLOATS * 4, st, a, wpacked, zeros, ntiles, nbx, nby, nbz, head);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// the head can be folded in when: 16 feature channels, 16 or 32 labels, volumes of whole 4 x 4 x 16 tiles, and more up-sampled than skip
// channels (the chunk order of a tile is then A A .. B: the skip-halo buffer is idle while the head uses it as scratch)
bool up2_head_ok(const ConvArgs &a, int labels) {
    return up2_ok(a, 1) && a.Cout == 16 && (labels == 16 || labels == 32) && a.OX % CT_X == 0 && a.OY % CT_Y == 0 && a.OZ % CT_Z == 0 &&
           a.c0 < a.c1 && u2_head_order_ok(a.c1 >> 4, a.c0 >> 4);
}
