// Plain 3x3x3 'same' convolution (dilation 1, input channels a multiple of 16) in the schedule of conv_up2.h -- included by conv.hip
// behind conv_up2.h, whose helpers (u2_dma16, u2_ldw, u2_store, u2_wait, u2_newer, the skip-halo LDS layout) it shares.
//
// What differs from conv3d_mfma<NT, true>: persistent blocks (one per CU, tiles of one XCD's contiguous range), halo tiles by
// LDS-DMA into TWO buffers so that the DMA of chunk k + 1 -- also across the tile boundary -- is issued when chunk k starts (one
// barrier per chunk, no exposed fetch after the first tile, no staging registers, no ds_write), LDS fragments read one tap ahead,
// weights two taps ahead with hand-counted vmcnt, outputs of a tile stored behind the first DMA of the next tile.  The M-tiles use
// the parity row mapping of conv_up2.h (rows = (y pair member) x (8 z of one parity)): any assignment of voxels to MFMA rows is
// valid for a plain convolution and this one re-uses the bank-conflict-free de-interleaved halo layout.
// Weights: the standard packed layout of conv3d_pack_weights ([chunk][tap][nt][lane][m]).


constexpr int P27_BUF_FLOATS = U2_B_FLOATS + U2_GAP_FLOATS;          // one halo buffer (+ the overrun of its last DMA instruction)
constexpr int P27_LDS_FLOATS = 2 * P27_BUF_FLOATS;
// POOL (round 6): the MaxPooling3D(2) that follows an encoder convolution (models.py:1436-1438) out of the same kernel -- the
// full-resolution tensor is written as before (the decoder's skip connection reads it), the pooled one in addition, instead of a kernel
// that reads the full-resolution tensor back.  In the parity row mapping the z pairs of a pooling window are two M-tiles of ONE lane
// (mt >> 1); the x and y pairs are the four waves of the block (px, py), same lane: they meet in 4 x 8 NT x 64 floats of LDS behind the
// halo buffers, one block barrier per tile, and every wave stores a quarter of the tile's 2 x 2 x 8 pooled voxels.  Whole tiles only.
constexpr int p27_pool_floats(int NT) { return 4 * 8 * NT * 64; }

template <int NT, bool POOL = false>
__global__ __launch_bounds__(256, 1) void conv3d_p27_mfma(ConvArgs a, const float *__restrict__ wpacked, const float *__restrict__ zeros,
                                                          unsigned ntiles, unsigned nbx, unsigned nby, unsigned nbz, float *__restrict__ pool_out) {
    static_assert(!POOL || NT >= 2, "the pooled form rides the immediate stores of the NT >= 2 instantiations");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool DEFER = NT <= 1 && !POOL;                            // deferred stores as in conv_up2.h
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int px = w & 1, py = w >> 1, iy = li >> 3, iz = li & 7;
    const int nB = a.c0 >> 4;

    const unsigned xcd = blockIdx.x % NRT_NXCD, J = gridDim.x / NRT_NXCD;
    const unsigned T8 = (ntiles + NRT_NXCD - 1) / NRT_NXCD;
    const unsigned tend = (xcd + 1) * T8 < ntiles ? (xcd + 1) * T8 : ntiles;
    unsigned tile = xcd * T8 + blockIdx.x / NRT_NXCD;
    if (tile >= tend) return;

    // ---- what this thread moves per chunk: 16-byte pieces g = 256 i + tid of the LDS image (layout of conv_up2.h) ----------------
    const int sZ0 = a.c0, sY0 = a.Z * sZ0, sX0 = a.Y * sY0;
    int relB[U2_NDB];
    unsigned validB = 0;
    auto pieceB = [&](int i, int &lx, int &ly, int &lz, int &c) __attribute__((always_inline)) {
        const int g = i * 256 + threadIdx.x;
        const int yg = g / 92, rem = g % 92, rr = rem / 5;
        c = rem % 5; lx = yg / 6; ly = yg % 6; lz = 2 * (rr % 9) + rr / 9;
        return g < U2_B_FLOATS / 4 && rem < 90 && c < 4;
    };
#pragma unroll
    for (int i = 0; i < U2_NDB; ++i) {
        int lx, ly, lz, c;
        validB |= (pieceB(i, lx, ly, lz, c) ? 1u : 0u) << i;
        relB[i] = (lx - 1) * sX0 + (ly - 1) * sY0 + (lz - 1) * sZ0 + 4 * c;
    }
    struct Tile {
        const float *pB;             // first channel of the tile origin
        unsigned okB;                // per DMA piece: inside the volume
        unsigned out;                // byte offset of output voxel (x0, y0, z0), channel 0
        unsigned pool;               // POOL: float offset of pooled voxel (x0 / 2, y0 / 2, z0 / 2), channel 0
        int x0, y0, z0;
        unsigned full;
    };
    auto decode = [&](unsigned t) __attribute__((always_inline)) {
        Tile T;
        const int bz = t % nbz, by = (t / nbz) % nby, bx = (t / (nbz * nby)) % nbx, b = t / (nbz * nby * nbx);
        T.x0 = bx * CT_X; T.y0 = by * CT_Y; T.z0 = bz * CT_Z;
        T.pB = a.src0 + ((long long)b * a.X * a.Y * a.Z + ((long long)T.x0 * a.Y + T.y0) * a.Z + T.z0) * a.c0;
        T.out = (unsigned)((((long long)b * a.OX + T.x0) * a.OY + T.y0) * a.OZ + T.z0) * (unsigned)a.Cout * 4u;
        T.pool = POOL ? (unsigned)((((long long)b * (a.OX / 2) + T.x0 / 2) * (a.OY / 2) + T.y0 / 2) * (a.OZ / 2) + T.z0 / 2) * (unsigned)a.Cout : 0u;
        T.full = T.x0 + CT_X <= a.OX && T.y0 + CT_Y <= a.OY && T.z0 + CT_Z <= a.OZ && (a.Cout & 15) == 0 && DEFER;
        const bool inner = T.x0 >= 1 && T.y0 >= 1 && T.z0 >= 1 && T.x0 + CT_X + 1 <= a.X && T.y0 + CT_Y + 1 <= a.Y && T.z0 + CT_Z + 1 <= a.Z;
        T.okB = validB;
        if (!inner) {
            T.okB = 0;
#pragma unroll
            for (int i = 0; i < U2_NDB; ++i) {
                int lx, ly, lz, c;
                pieceB(i, lx, ly, lz, c);
                const unsigned x = T.x0 - 1 + lx, y = T.y0 - 1 + ly, z = T.z0 - 1 + lz;
                T.okB |= (x < (unsigned)a.X && y < (unsigned)a.Y && z < (unsigned)a.Z ? 1u : 0u) << i;
            }
            T.okB &= validB;
        }
        return T;
    };
    auto issueB = [&](const float *p, unsigned ok, unsigned buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < U2_NDB; ++i)
            u2_dma16(((ok >> i) & 1u) ? p + relB[i] : zeros, lds0 + buf * (P27_BUF_FLOATS * 4) + (i * 256 + w * 64) * 16);
    };

    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (a.bias && nt * 16 + li < a.Cout) ? a.bias[nt * 16 + li] : 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[nt]));      // the only compiler-visible load: settled here

    f32x4 acc[4][NT];
    float outv[4][NT][4];
    unsigned outBase = 0;
    bool pending = false;
    const unsigned oY = (unsigned)a.OZ * a.Cout * 4u, oX = (unsigned)a.OY * oY;
    const unsigned outLane = px * oX + (py + 2 * (kq >> 1)) * oY + (2 * (kq & 1) * 4) * (unsigned)a.Cout * 4u + li * 4u;
    auto stores = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    u2_store(a.out, outBase + outLane + 2 * (mt & 1) * oX + (2 * r + (mt >> 1)) * (unsigned)a.Cout * 4u + nt * 64u,
                             outv[mt][nt][r]);
    };
    const unsigned wlane = lane * 16u;

    // ---- one chunk of 16 input channels: the 27 taps on the halo in buffer `buf` -----------------------------------------------------
    auto body = [&](int ch, unsigned buf, auto Dc, auto Sc, auto &&issue_next) __attribute__((always_inline)) {
        constexpr int D = decltype(Dc)::value, S = decltype(Sc)::value;
        constexpr int P = NT == 1 ? 6 : NT == 2 ? 4 : 2, F = NT;
        const char *wp = (const char *)wpacked + (size_t)ch * 27 * NT * 1024;
        f32x4 bq[27][NT];
        auto request = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[q][nt] = u2_ldw(wp + (q * NT + nt) * 1024, wlane);
        };
#pragma unroll
        for (int q = 0; q < P; ++q) request(q);
        issue_next();
        if (S) stores();
        const float *abase = &lds[buf * P27_BUF_FLOATS + px * U2_SX + (py + 2 * iy) * U2_SY + iz * LDS_ROW + 4 * kq];
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
            u2_wait<u2_newer(t, P, F, D, S, 27)>(bq[t][0]);
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
    using IB = std::integral_constant<int, U2_NDB>;
    using IS = std::integral_constant<int, 16 * NT>;

    Tile cur = decode(tile);
    unsigned cIss = 0, cUse = 0;                                        // chunks requested / consumed: buffer = count & 1
    issueB(cur.pB, cur.okB, cIss++ & 1u);
    for (;;) {
        const unsigned ntile = tile + J;
        const bool has_next = ntile < tend;
        Tile nxt = cur;
        if (has_next) nxt = decode(ntile);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        for (int ch = 0; ch < nB; ++ch) {
            const bool last = ch + 1 == nB;
            u2_chunk_barrier();                                         // chunk ch has landed; chunk ch - 1 has been read by all waves
            const unsigned buf = cUse++ & 1u;
            if (!last || has_next) {
                const Tile &T = last ? nxt : cur;
                const int chn = last ? 0 : ch + 1;
                auto next = [&]() __attribute__((always_inline)) { issueB(T.pB + 16 * chn, T.okB, cIss++ & 1u); };
                if (ch == 0 && pending) { body(ch, buf, IB{}, IS{}, next); pending = false; }
                else body(ch, buf, IB{}, I0{}, next);
            } else {
                if (ch == 0 && pending) { body(ch, buf, I0{}, IS{}, [&]() {}); pending = false; }
                else body(ch, buf, I0{}, I0{}, [&]() {});
            }
        }
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
            if constexpr (POOL) {
                // entry e = (ix * 4 + r) * NT + nt of a lane: max over the z pair (M-tiles ix and ix + 2); [wave][entry][lane] in LDS
                float *pl = lds + P27_LDS_FLOATS;
#pragma unroll
                for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            pl[(w * 8 * NT + (ix * 4 + r) * NT + nt) * 64 + lane] =
                                fmaxf(activate(acc[ix][nt][r] + bv[nt], a.act), activate(acc[ix + 2][nt][r] + bv[nt], a.act));
                __syncthreads();
                // wave w finishes entries 2 NT w .. 2 NT w + 2 NT - 1: max over the four waves = the x and y pairs
                float *pb = pool_out + cur.pool;
                const unsigned pZ = (unsigned)a.Cout, pY = (unsigned)(a.OZ / 2) * pZ, pX = (unsigned)(a.OY / 2) * pY;
#pragma unroll
                for (int j = 0; j < 2 * NT; ++j) {
                    const int e = w * 2 * NT + j;
                    float m = pl[e * 64 + lane];
#pragma unroll
                    for (int ww = 1; ww < 4; ++ww) m = fmaxf(m, pl[(ww * 8 * NT + e) * 64 + lane]);
                    const int nt = e % NT, r = (e / NT) % 4, ix = e / (4 * NT);
                    pb[ix * pX + (kq >> 1) * pY + (4 * (kq & 1) + r) * pZ + nt * 16 + li] = m;
                }
                // (the next write to `pl` lies a whole tile ahead, behind chunk barriers every wave has to reach first)
            }
        }
        if (!has_next) break;
        tile = ntile;
        cur = nxt;
    }
    if (pending) stores();
}

// 256 zero bytes per device (what out-of-volume halo pieces read), allocated on first use and kept for the life of the process
const float *p27_zero_block(hipStream_t st) {
    static float *blocks[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!blocks[dev]) {
        float *p = nullptr;
        if (hipMalloc((void **)&p, 256) != hipSuccess) return nullptr;
        // zeroed synchronously (once per device): a memset queued on the first caller's stream could still be pending when another
        // stream's first launch reads the block
        if (hipMemset(p, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
        blocks[dev] = p;
    }
    return blocks[dev];
}

bool p27_ok(const ConvArgs &a, int padding_same, int batch) {
    return padding_same && a.kx == 3 && a.ky == 3 && a.kz == 3 && a.dil == 1 && a.c1 == 0 && a.c0 >= 16 && a.c0 % 16 == 0 && a.Cout <= 64 &&
           a.fold == 0 && (long long)a.X * a.Y * a.Z * a.c0 < (1ll << 30) && (long long)batch * a.X * a.Y * a.Z * a.Cout < (1ll << 30);
}

template <int NT, bool POOL = false>
int launch_p27(const ConvArgs &a, const float *wpacked, int batch, hipStream_t st, float *pool_out = nullptr) {
    const unsigned nbx = (a.OX + CT_X - 1) / CT_X, nby = (a.OY + CT_Y - 1) / CT_Y, nbz = (a.OZ + CT_Z - 1) / CT_Z;
    const unsigned ntiles = nbx * nby * nbz * (unsigned)batch;
    const int shm = (P27_LDS_FLOATS + (POOL ? p27_pool_floats(NT) : 0)) * 4;
    if (hipFuncSetAttribute((const void *)conv3d_p27_mfma<NT, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, shm) != hipSuccess)
        return NRT_ERR_LAUNCH;
    const unsigned T8 = (ntiles + NRT_NXCD - 1) / NRT_NXCD, per_xcd = (unsigned)nrt_num_cus() / NRT_NXCD;
    const unsigned J = T8 < per_xcd ? T8 : per_xcd;
    const float *zeros = p27_zero_block(st);
    if (!zeros) return NRT_ERR_LAUNCH;
    hipLaunchKernelGGL((conv3d_p27_mfma<NT, POOL>), dim3(NRT_NXCD * J), dim3(256), shm, st, a, wpacked, zeros, ntiles, nbx, nby, nbz, pool_out);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

// the pooled form takes: what p27_ok takes, 32 .. 64 output channels in whole groups of 16 (the immediate-store instantiations),
// volumes made of whole 4 x 4 x 16 tiles (every pooling window then lies inside one tile), 32-bit pooled offsets
bool p27_pool_ok(const ConvArgs &a, int batch) {
    return p27_ok(a, 1, batch) && a.Cout >= 32 && a.Cout % 16 == 0 && a.OX % CT_X == 0 && a.OY % CT_Y == 0 && a.OZ % CT_Z == 0;
}
