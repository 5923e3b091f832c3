// Backward kernels of the hot path: what TensorFlow's autodiff derives from the
// reference graphs, written out by hand so that the ops are usable as layers / losses in training.
//
// interpn (neurite/tf/utils/utils.py:137-191, 206-213), linear:
//   out[q,c] = sum_corner (prod_d w_{corner_d,d}) * vol[idx(corner), c]          [* (1-oob) + oob*fill]
//   d out / d vol   : scatter-add of wt * g[q,c] into the gathered rows            (tf.gather gradient)
//   d out / d loc_d : only through clipped_loc (tf.floor has no gradient): d w0_d = -m_d, d w1_d = +m_d with
//                     m_d = [0 <= loc_d <= max_d]  (tf.clip_by_value passes the gradient on the closed range)
//   fill: the gradient is masked by (1 - oob).  SHIFT mode: d/d shift = d/d loc.
// soft Dice (metrics.py:476-482): dice = (2 Stp + eps)/(Stt + Spp + eps)  [divide_no_nan when eps = 0]
//   d dice/d p_v = (2 t_v den - 2 p_v num) / den^2 , symmetric in t.
// weighted CCE (metrics.py:648-650 + Keras): q = p / sum p, qc = clip(q, 1e-7, 1-1e-7), l = -sum t'_c log qc_c
//   d l / d p_j = -(1/s) ( r_j - sum_c r_c q_c ),  r_c = t'_c [1e-7 <= q_c <= 1-1e-7] / qc_c
//   from logits: d l / d z_j = (sum_c t'_c) softmax_j - t'_j.
// Float atomics (global_atomic_add_f32) make grad_vol's summation order non-deterministic, like TF's own
// scatter-add on GPU; everything else is deterministic.

#include <stdlib.h>

#include "interpn_core.h"
#include "wc.h"

namespace {

struct InterpBwdArgs {
    InterpArgs f;                 // forward geometry (vol, loc; out unused)
    const float *gout;            // [B, nout, C]
    float *gvol;                  // [B, nin, C]  zero-initialised by the caller, or null
    float *gloc;                  // [B, nout, D] or null
    TileGeom tg;                  // x-march schedule (G == 8 kernels) when tg.x_march
};

__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

// one thread per output voxel, loops over channels: any C, D in {1,2,3}
template <int D, int MODE>
__global__ __launch_bounds__(256) void interpn_bwd_generic(InterpBwdArgs ba) {
    const InterpArgs &a = ba.f;
    const int b = blockIdx.y;
    const float *vol = (const float *)a.vol + (long long)b * a.vol_bs;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const float *go = ba.gout + (long long)b * a.out_bs;
    float *gv = ba.gvol ? ba.gvol + (long long)b * a.vol_bs : nullptr;
    float *gl = ba.gloc ? ba.gloc + (long long)b * a.nout * D : nullptr;
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < a.nout; q += gridDim.x * blockDim.x) {
        int qd[NRT_MAXD];
        float p[NRT_MAXD];
        decode<D>(a, q, qd);
        load_loc<D, MODE>(a, locb, q, qd, p);
        const bool oob = a.has_fill ? out_of_bounds<D>(a, p) : false;
        int i0[NRT_MAXD], i1[NRT_MAXD];
        float w0[NRT_MAXD], w1[NRT_MAXD], m[NRT_MAXD];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
            m[d] = (p[d] >= 0.0f && p[d] <= (float)(a.S[d] - 1)) ? 1.0f : 0.0f;
        }
        float gacc[NRT_MAXD] = {0.0f, 0.0f, 0.0f};
        if (!oob) {
            // per corner: row index, weight, and the weight products with one dimension's factor replaced by its derivative
            // (formed once per voxel, not once per channel)
            long long cidx[1 << D];
            float cwt[1 << D], cwexc[1 << D][D];
#pragma unroll
            for (int corner = 0; corner < (1 << D); ++corner) {
                long long idx = 0;
                float wt = 1.0f;
                float wexc[D];
#pragma unroll
                for (int d = 0; d < D; ++d) wexc[d] = 1.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int bit = (corner >> (D - 1 - d)) & 1;
                    idx = idx * a.S[d] + (bit ? i1[d] : i0[d]);
                    const float w = bit ? w1[d] : w0[d];
                    wt *= w;
#pragma unroll
                    for (int e = 0; e < D; ++e) wexc[e] *= (e == d) ? (bit ? m[d] : -m[d]) : w;
                }
                cidx[corner] = idx * a.C;
                cwt[corner] = wt;
#pragma unroll
                for (int d = 0; d < D; ++d) cwexc[corner][d] = wexc[d];
            }
            // channel counts that are multiples of 4 (12, 20, 24 ... -- the powers of two have their own kernel): 16-byte loads of
            // the gradient row and of the corner rows, the same sums channel by channel
            const bool quads = (a.C & 3) == 0 && (a.vol_bs & 3) == 0 && (a.out_bs & 3) == 0 &&
                               ((((uintptr_t)a.vol | (uintptr_t)ba.gout) & 15) == 0);
            if (quads) {
                for (int c = 0; c < a.C; c += 4) {
                    const nrt_f4 g4 = *(const nrt_f4 *)(go + (long long)q * a.C + c);
#pragma unroll
                    for (int corner = 0; corner < (1 << D); ++corner) {
                        if (gv) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) atomic_add_f32(&gv[cidx[corner] + c + k], cwt[corner] * g4[k]);
                        }
                        if (gl) {
                            const nrt_f4 v4 = *(const nrt_f4 *)(vol + cidx[corner] + c);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int d = 0; d < D; ++d) gacc[d] += g4[k] * cwexc[corner][d] * v4[k];
                        }
                    }
                }
            } else {
                for (int c = 0; c < a.C; ++c) {
                    const float g = go[(long long)q * a.C + c];
#pragma unroll
                    for (int corner = 0; corner < (1 << D); ++corner) {
                        if (gv) atomic_add_f32(&gv[cidx[corner] + c], cwt[corner] * g);
                        if (gl) {
                            const float v = vol[cidx[corner] + c];
#pragma unroll
                            for (int d = 0; d < D; ++d) gacc[d] += g * cwexc[corner][d] * v;
                        }
                    }
                }
            }
        }
        if (gl) {
#pragma unroll
            for (int d = 0; d < D; ++d) gl[(long long)q * D + d] = gacc[d];
        }
    }
}

// d out / d vol for any channel count: one thread per output ELEMENT (voxel, channel), channel fastest.  The lanes of a wave then
// cover whole rows of consecutive voxels, so an atomic instruction reaches 64 / C rows instead of 64 (the L2 atomic units are
// bound by requests, not by dwords: the per-voxel kernel above took 28 / 132 ms for 4 x 160^3 x 5 / 20).  The corner arithmetic is
// repeated per channel -- cheap next to the atomics.
template <int D, int MODE>
__global__ __launch_bounds__(256) void interpn_bwd_vol_elems(InterpBwdArgs ba) {
    const InterpArgs &a = ba.f;
    const int b = blockIdx.y;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const float *go = ba.gout + (long long)b * a.out_bs;
    float *gv = ba.gvol + (long long)b * a.vol_bs;
    const unsigned long long total = (unsigned long long)a.nout * (unsigned)a.C;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * 256u) {
        const unsigned q = (unsigned)(e / (unsigned)a.C);
        const int c = (int)(e - (unsigned long long)q * (unsigned)a.C);
        int qd[NRT_MAXD];
        float p[NRT_MAXD];
        decode<D>(a, q, qd);
        load_loc<D, MODE>(a, locb, q, qd, p);
        if (a.has_fill && out_of_bounds<D>(a, p)) continue;
        int i0[NRT_MAXD], i1[NRT_MAXD];
        float w0[NRT_MAXD], w1[NRT_MAXD];
#pragma unroll
        for (int d = 0; d < D; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
        const float g = go[e];
#pragma unroll
        for (int corner = 0; corner < (1 << D); ++corner) {
            long long idx = 0;
            float wt = 1.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int bit = (corner >> (D - 1 - d)) & 1;
                idx = idx * a.S[d] + (bit ? i1[d] : i0[d]);
                wt *= bit ? w1[d] : w0[d];
            }
            atomic_add_f32(&gv[idx * a.C + c], wt * g);
        }
    }
}

// G = C/4 lanes per voxel, 3-D
template <int G, int MODE>
__global__ __launch_bounds__(256) void interpn_bwd_rows(InterpBwdArgs ba) {
    constexpr int D = 3;
    constexpr int NG = 256 / G;
    const InterpArgs &a = ba.f;
    // x-march schedule (G == 8): the block owns a 4 x 8 (y,z) patch of one batch entry and walks x, see interpn_core.h
    const bool xm = G == 8 && ba.tg.x_march;
    int b = blockIdx.y, xm_x0 = 0, xm_y0 = 0, xm_z0 = 0, xm_len = 0;
    if (xm) {
        unsigned prow;
        if (!xmarch_block(ba.tg, a.O[0], b, prow, xm_x0, xm_y0, xm_z0, xm_len)) return;
    }
    const nrt_f4 *vol = (const nrt_f4 *)((const float *)a.vol + (long long)b * a.vol_bs);
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const nrt_f4 *go = (const nrt_f4 *)(ba.gout + (long long)b * a.out_bs);
    float *gv = ba.gvol ? ba.gvol + (long long)b * a.vol_bs : nullptr;
    float *gl = ba.gloc ? ba.gloc + (long long)b * a.nout * D : nullptr;
    const int lg = threadIdx.x % G;
    const unsigned g = threadIdx.x / G;
    const int Y = a.S[1], Z = a.S[2];
    // two voxels per lane-group in flight (all row loads of an iteration issued before the first use); every lane-group
    // runs the same number of iterations so that the shuffles below are convergent
    constexpr int U = 2;
    // C == 32: the block's U * NG voxels of an iteration hand (row, weight) pairs and their grad_out rows to LDS, and the
    // scatter runs row-major over them -- a wave's atomic instruction covers two whole 128-byte rows instead of 8 dwords
    // of 8 rows (the L2 atomic units are request-bound: 14.3 ms -> 7.2 ms per volume came from 4 -> 8 dwords per request)
    constexpr bool LDS_SCATTER = (G == 8);
    constexpr int NSLOT = LDS_SCATTER ? NG * U : 1;
    __shared__ float s_g[NSLOT * 4 * G];
    __shared__ unsigned s_idx[NSLOT * 8];
    __shared__ float s_wt[NSLOT * 8];
    const unsigned ngroups = gridDim.x * NG;
    unsigned niter = (a.nout + ngroups * U - 1) / (ngroups * U);
    if (xm) niter = ((unsigned)xm_len + U - 1) / U;
    for (unsigned it = 0; it < niter; ++it) {
        unsigned q[U];
        bool live[U], oob[U];
        int i0[U][3], i1[U][3];
        float w0[U][3], w1[U][3], m[U][3];
        nrt_f4 gq[U], v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned qq;
            if (xm) {
                const int x = xm_x0 + (int)(it * U + u), y = xm_y0 + (int)(g >> ba.tg.ltz), z = xm_z0 + (int)(g & ((1u << ba.tg.ltz) - 1u));
                const bool in = x < xm_x0 + xm_len && y < a.O[1] && z < a.O[2];
                qq = in ? ((unsigned)x * (unsigned)a.O[1] + (unsigned)y) * (unsigned)a.O[2] + (unsigned)z : 0xffffffffu;
            } else {
                qq = blockIdx.x * NG + g + (it * U + u) * ngroups;
            }
            live[u] = qq < a.nout;
            q[u] = live[u] ? qq : a.nout - 1;
            int qd[NRT_MAXD];
            float p[NRT_MAXD];
            decode<D>(a, q[u], qd);
            load_loc<D, MODE>(a, locb, q[u], qd, p);
            oob[u] = a.has_fill ? out_of_bounds<D>(a, p) : false;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                corner_1d(p[d], a.S[d], i0[u][d], i1[u][d], w0[u][d], w1[u][d]);
                m[u][d] = (p[d] >= 0.0f && p[d] <= (float)(a.S[d] - 1)) ? 1.0f : 0.0f;
            }
            gq[u] = go[(long long)q[u] * G + lg];
            if (gl) {
#pragma unroll
                for (int corner = 0; corner < 8; ++corner) {
                    const int bx = (corner >> 2) & 1, by = (corner >> 1) & 1, bz = corner & 1;
                    const long long idx = ((long long)(bx ? i1[u][0] : i0[u][0]) * Y + (by ? i1[u][1] : i0[u][1])) * Z +
                                          (bz ? i1[u][2] : i0[u][2]);
                    v[u][corner] = vol[idx * G + lg];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (oob[u] || !live[u]) gq[u] = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
            float gacc[3] = {0.0f, 0.0f, 0.0f};
            // scatter layout: atomic e of lane lg adds channel G * e + lg, so the G lanes of a voxel hit G consecutive
            // dwords per instruction (the row-load layout, channel 4 lg + e, would spread them 16 B apart); the value
            // lives in component (G e + lg) & 3 of lane (G e + lg) >> 2 of the group
            float gs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (LDS_SCATTER && gv) {
                const unsigned slot = (unsigned)u * NG + g;
                ((nrt_f4 *)s_g)[slot * G + lg] = gq[u];
                const int bx = (lg >> 2) & 1, by = (lg >> 1) & 1, bz = lg & 1;       // lane lg files corner lg
                const unsigned ix = bx ? i1[u][0] : i0[u][0], iy = by ? i1[u][1] : i0[u][1], iz = bz ? i1[u][2] : i0[u][2];
                s_idx[slot * 8 + lg] = (ix * (unsigned)Y + iy) * (unsigned)Z + iz;
                s_wt[slot * 8 + lg] = (bx ? w1[u][0] : w0[u][0]) * (by ? w1[u][1] : w0[u][1]) * (bz ? w1[u][2] : w0[u][2]);
            } else if (gv) {
                const int base = (int)(threadIdx.x & 63u) - lg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ch = G * e + lg, src = base + (ch >> 2);
                    const float c0 = __shfl(gq[u][0], src, 64), c1 = __shfl(gq[u][1], src, 64);
                    const float c2 = __shfl(gq[u][2], src, 64), c3 = __shfl(gq[u][3], src, 64);
                    gs[e] = (ch & 2) ? ((ch & 1) ? c3 : c2) : ((ch & 1) ? c1 : c0);
                }
            }
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = (corner >> 2) & 1, by = (corner >> 1) & 1, bz = corner & 1;
                const float wx = bx ? w1[u][0] : w0[u][0], wy = by ? w1[u][1] : w0[u][1], wz = bz ? w1[u][2] : w0[u][2];
                if (!LDS_SCATTER && gv && live[u] && !oob[u]) {
                    const long long idx = ((long long)(bx ? i1[u][0] : i0[u][0]) * Y + (by ? i1[u][1] : i0[u][1])) * Z +
                                          (bz ? i1[u][2] : i0[u][2]);
                    const float wt = wx * wy * wz;
                    float *dst = gv + idx * (4 * G) + lg;
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomic_add_f32(dst + G * e, wt * gs[e]);
                }
                if (gl) {
                    const nrt_f4 c = v[u][corner];
                    const float dot = gq[u][0] * c[0] + gq[u][1] * c[1] + gq[u][2] * c[2] + gq[u][3] * c[3];
                    gacc[0] += dot * (bx ? m[u][0] : -m[u][0]) * wy * wz;
                    gacc[1] += dot * wx * (by ? m[u][1] : -m[u][1]) * wz;
                    gacc[2] += dot * wx * wy * (bz ? m[u][2] : -m[u][2]);
                }
            }
            if (gl) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int off = 1; off < G; off <<= 1) gacc[d] += __shfl_xor(gacc[d], off, 64);
                if (live[u] && lg == 0) {
                    float *dst = gl + (long long)q[u] * 3;
                    dst[0] = gacc[0]; dst[1] = gacc[1]; dst[2] = gacc[2];
                }
            }
        }
        if (LDS_SCATTER && gv) {
            constexpr int C = 4 * G;
            __syncthreads();
            for (unsigned i = threadIdx.x; i < (unsigned)(NSLOT * 8 * C); i += 256) {
                const unsigned r = i / C, ch = i % C;
                const float val = s_wt[r] * s_g[(r >> 3) * C + ch];      // dead / out-of-bounds voxels filed zero rows
                if (val != 0.0f) atomic_add_f32(gv + (size_t)s_idx[r] * C + ch, val);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// soft Dice backward: elementwise over [B, V, L]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_soft_bwd(const float *__restrict__ t, const float *__restrict__ p,
                                                     const float *__restrict__ sums, const float *__restrict__ gdice,
                                                     long long nvox, int L, float eps, float *__restrict__ gp,
                                                     float *__restrict__ gt) {
    const int b = blockIdx.y;
    const long long n = nvox * L;
    const float *tb = t + (long long)b * n, *pb = p + (long long)b * n;
    const float *s = sums + (long long)b * 3 * L;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(e % L);
        const float num = 2.0f * s[l] + eps, den = s[L + l] + s[2 * L + l] + eps;
        const float g = gdice[(long long)b * L + l];
        float ca = 0.0f, cb = 0.0f;                       // d dice = ca * other + cb * self
        if (den != 0.0f) { ca = 2.0f * g / den; cb = -2.0f * g * num / (den * den); }
        const float tv = tb[e], pv = pb[e];
        if (gp) gp[(long long)b * n + e] = ca * tv + cb * pv;
        if (gt) gt[(long long)b * n + e] = ca * pv + cb * tv;
    }
}

// ---------------------------------------------------------------------------------------------
// weighted CCE backward wrt y_pred: one thread per voxel (C in a loop), float32
// ---------------------------------------------------------------------------------------------
// VP > 0: the rows of VP voxels of t and p are staged in LDS with coalesced loads, the gradient rows leave the same way (a lane
// walking its own row in memory touches one line per element: 17 ms for 4 x 160^3 x 20); VP == 0: rows too wide, straight access
__global__ __launch_bounds__(256) void wcce_bwd(const float *__restrict__ t, const float *__restrict__ p,
                                                const float *__restrict__ w, const float *__restrict__ gscalar,
                                                const float *__restrict__ gper_voxel, long long n, int C, int logits,
                                                float smooth, float scale, int VP, float *__restrict__ gp) {
    extern __shared__ float wb_lds[];       // [2][VP * C]: t rows (then the gradient rows), p rows
    const float keep = 1.0f - smooth, add = smooth / (float)C;
    auto voxel = [&](const float *tv, const float *pv, float *gv, float g) {     // gv may alias tv (element c is read before written)
        if (logits) {
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, pv[c]);
            float se = 0.0f, st = 0.0f;
            for (int c = 0; c < C; ++c) {
                se += expf(pv[c] - mx);
                float tt = (w ? w[c] : 1.0f) * tv[c];
                if (smooth != 0.0f) tt = tt * keep + add;
                st += tt;
            }
            for (int c = 0; c < C; ++c) {
                float tt = (w ? w[c] : 1.0f) * tv[c];
                if (smooth != 0.0f) tt = tt * keep + add;
                gv[c] = g * (st * expf(pv[c] - mx) / se - tt);
            }
        } else {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s += pv[c];
            float rq = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float q = pv[c] / s;
                const bool in = q >= 1e-7f && q <= 1.0f - 1e-7f;
                float tt = (w ? w[c] : 1.0f) * tv[c];
                if (smooth != 0.0f) tt = tt * keep + add;
                rq += in ? tt : 0.0f;                     // r_c q_c = t'_c when the clip is inactive
            }
            for (int c = 0; c < C; ++c) {
                const float q = pv[c] / s;
                const bool in = q >= 1e-7f && q <= 1.0f - 1e-7f;
                float tt = (w ? w[c] : 1.0f) * tv[c];
                if (smooth != 0.0f) tt = tt * keep + add;
                const float r = in ? tt / q : 0.0f;
                gv[c] = -g * (r - rq) / s;
            }
        }
    };
    if (VP > 0) {
        float *st = wb_lds, *sp = wb_lds + (long long)VP * C;
        for (long long v0 = (long long)blockIdx.x * VP; v0 < n; v0 += (long long)gridDim.x * VP) {
            const int nv = (int)((n - v0) < VP ? (n - v0) : VP);
            const long long ne = (long long)nv * C;
            __syncthreads();
            for (long long i = threadIdx.x; i < ne; i += blockDim.x) { st[i] = t[v0 * C + i]; sp[i] = p[v0 * C + i]; }
            __syncthreads();
            if ((int)threadIdx.x < nv) {
                const long long v = v0 + threadIdx.x;
                const float g = (gper_voxel ? gper_voxel[v] : gscalar[0]) * scale;
                voxel(st + (long long)threadIdx.x * C, sp + (long long)threadIdx.x * C, st + (long long)threadIdx.x * C, g);
            }
            __syncthreads();
            for (long long i = threadIdx.x; i < ne; i += blockDim.x) gp[v0 * C + i] = st[i];
        }
        return;
    }
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
        const float g = (gper_voxel ? gper_voxel[v] : gscalar[0]) * scale;
        voxel(t + v * C, p + v * C, gp + v * C, g);
    }
}

// soft Dice with normalize=True (metrics.py:434-436: t <- divide_no_nan(t, sum_l t), p likewise, per voxel): the elementwise
// gradient wrt the NORMALISED maps (coefficients from the sums, which the forward took over the normalised values) is pulled
// back through the normalisation:  d / d p_k = (g_k - sum_l g_l pn_l) / Sp   (0 where Sp == 0: divide_no_nan's gradient).
// One thread per voxel, three passes over its labels (the row stays in L1); coefficient table in LDS.
__global__ __launch_bounds__(256) void dice_soft_bwd_norm(const float *__restrict__ t, const float *__restrict__ p,
                                                          const float *__restrict__ sums, const float *__restrict__ gdice,
                                                          long long nvox, int L, float eps, float *__restrict__ gp,
                                                          float *__restrict__ gt) {
    extern __shared__ float coef[];          // ca[L], cb[L]
    const int b = blockIdx.y;
    const float *s = sums + (long long)b * 3 * L;
    for (int l = threadIdx.x; l < L; l += 256) {
        const float num = 2.0f * s[l] + eps, den = s[L + l] + s[2 * L + l] + eps;
        const float g = gdice[(long long)b * L + l];
        coef[l] = den != 0.0f ? 2.0f * g / den : 0.0f;
        coef[L + l] = den != 0.0f ? -2.0f * g * num / (den * den) : 0.0f;
    }
    __syncthreads();
    const float *tb = t + (long long)b * nvox * L, *pb = p + (long long)b * nvox * L;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long long)gridDim.x * 256) {
        const float *tv = tb + v * L, *pv = pb + v * L;
        float St = 0.0f, Sp = 0.0f;
        for (int l = 0; l < L; ++l) { St += tv[l]; Sp += pv[l]; }
        const float it = St != 0.0f ? 1.0f / St : 0.0f, ip = Sp != 0.0f ? 1.0f / Sp : 0.0f;
        float dt = 0.0f, dp = 0.0f;              // sum_l g_l * normalised value
        for (int l = 0; l < L; ++l) {
            const float tn = tv[l] * it, pn = pv[l] * ip;
            dp += (coef[l] * tn + coef[L + l] * pn) * pn;
            dt += (coef[l] * pn + coef[L + l] * tn) * tn;
        }
        for (int l = 0; l < L; ++l) {
            const float tn = tv[l] * it, pn = pv[l] * ip;
            if (gp) gp[((long long)b * nvox + v) * L + l] = ((coef[l] * tn + coef[L + l] * pn) - dp) * ip;
            if (gt) gt[((long long)b * nvox + v) * L + l] = ((coef[l] * pn + coef[L + l] * tn) - dt) * it;
        }
    }
}

// float4 version: L % 4 == 0 and 256 % (L/4) == 0, so that a thread keeps its 4 labels over the whole grid-stride loop
__global__ __launch_bounds__(256) void dice_soft_bwd_vec(const nrt_f4 *__restrict__ t, const nrt_f4 *__restrict__ p,
                                                         const float *__restrict__ sums, const float *__restrict__ gdice,
                                                         long long nvox, int L, float eps, nrt_f4 *__restrict__ gp,
                                                         nrt_f4 *__restrict__ gt) {
    const int b = blockIdx.y;
    const int G4 = L >> 2;
    const long long n4 = nvox * G4;
    const nrt_f4 *tb = t + (long long)b * n4, *pb = p + (long long)b * n4;
    const float *s = sums + (long long)b * 3 * L;
    const int l0 = (threadIdx.x % G4) * 4;
    float ca[4], cb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = l0 + k;
        const float num = 2.0f * s[l] + eps, den = s[L + l] + s[2 * L + l] + eps;
        const float g = gdice[(long long)b * L + l];
        ca[k] = 0.0f; cb[k] = 0.0f;
        if (den != 0.0f) { ca[k] = 2.0f * g / den; cb[k] = -2.0f * g * num / (den * den); }
    }
    long long ebeg, eend;                                      // one contiguous range per block (nrt_block_range; G4 | 256 keeps l0 valid)
    nrt_block_range(n4, 256, ebeg, eend);
    for (long long e = ebeg + threadIdx.x; e < eend; e += 256) {
        const nrt_f4 tv = tb[e], pv = pb[e];
        if (gp) {
            nrt_f4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ca[k] * tv[k] + cb[k] * pv[k];
            __builtin_nontemporal_store(o, &gp[(long long)b * n4 + e]);
        }
        if (gt) {
            nrt_f4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ca[k] * pv[k] + cb[k] * tv[k];
            __builtin_nontemporal_store(o, &gt[(long long)b * n4 + e]);
        }
    }
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// G = C/4 lanes per voxel
template <int G, int LOGITS>
__global__ __launch_bounds__(256) void wcce_bwd_vec(const nrt_f4 *__restrict__ t, const nrt_f4 *__restrict__ p,
                                                    const float *__restrict__ w, const float *__restrict__ gscalar,
                                                    const float *__restrict__ gper_voxel, long long n, float smooth,
                                                    float scale, nrt_f4 *__restrict__ gp) {
    constexpr int C = G * 4;
    constexpr int NG = 256 / G;
    const int lg = threadIdx.x % G;
    const float keep = 1.0f - smooth, add = smooth / (float)C;
    float wl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[k] = w ? w[lg * 4 + k] : 1.0f;
    const float gs = gscalar ? gscalar[0] * scale : 0.0f;
    const long long ngroups = (long long)gridDim.x * NG;
    const long long niter = (n + ngroups - 1) / ngroups;
    for (long long it = 0; it < niter; ++it) {
        const long long vv = ((long long)blockIdx.x * niter + it) * NG + threadIdx.x / G;      // a block streams one contiguous range (nrt_block_range)
        const bool live = vv < n;
        const long long v = live ? vv : n - 1;
        const nrt_f4 tv = t[v * G + lg], pv = p[v * G + lg];
        const float g = gper_voxel ? gper_voxel[v] * scale : gs;
        float tt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tt[k] = wl[k] * tv[k];
            if (smooth != 0.0f) tt[k] = tt[k] * keep + add;
        }
        nrt_f4 o;
        if (LOGITS) {
            const float mx = group_max<G>(fmaxf(fmaxf(pv[0], pv[1]), fmaxf(pv[2], pv[3])));
            float ex[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ex[k] = expf(pv[k] - mx);
            const float se = group_sum<G>((ex[0] + ex[1]) + (ex[2] + ex[3]));
            const float st = group_sum<G>((tt[0] + tt[1]) + (tt[2] + tt[3]));
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = g * (st * ex[k] / se - tt[k]);
        } else {
            const float s = group_sum<G>((pv[0] + pv[1]) + (pv[2] + pv[3]));
            float q[4], rq = 0.0f;
            bool in[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k] = pv[k] / s;
                in[k] = q[k] >= 1e-7f && q[k] <= 1.0f - 1e-7f;
                rq += in[k] ? tt[k] : 0.0f;
            }
            rq = group_sum<G>(rq);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = -g * ((in[k] ? tt[k] / q[k] : 0.0f) - rq) / s;
        }
        if (live) __builtin_nontemporal_store(o, &gp[v * G + lg]);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused backward of warp + soft Dice wrt the displacement field: the warped row is rebuilt in registers,
// d dice / d warped is formed on the fly from the saved sums, and only grad_loc (12 B/voxel) is written.
// G = L/4 lanes per voxel, 3-D.
// ---------------------------------------------------------------------------------------------
template <int G, int MODE>
__global__ __launch_bounds__(256) void warp_dice_bwd_rows(InterpBwdArgs ba, const float *__restrict__ fixed,
                                                          const float *__restrict__ sums,
                                                          const float *__restrict__ gdice, float eps) {
    constexpr int D = 3;
    constexpr int NG = 256 / G;
    constexpr int L = G * 4;
    const InterpArgs &a = ba.f;
    // x-march schedule (G == 8): the block owns a 4 x 8 (y,z) patch of one batch entry and walks x, see interpn_core.h
    const bool xm = G == 8 && ba.tg.x_march;
    int b = blockIdx.y, xm_x0 = 0, xm_y0 = 0, xm_z0 = 0, xm_len = 0;
    if (xm) {
        unsigned prow;
        if (!xmarch_block(ba.tg, a.O[0], b, prow, xm_x0, xm_y0, xm_z0, xm_len)) return;
    }
    const nrt_f4 *vol = (const nrt_f4 *)((const float *)a.vol + (long long)b * a.vol_bs);
    const float *locb = a.loc + (long long)b * a.loc_bs;
    const nrt_f4 *fix = (const nrt_f4 *)(fixed + (long long)b * a.out_bs);
    float *gl = ba.gloc + (long long)b * a.nout * D;
    const int lg = threadIdx.x % G;
    const unsigned g = threadIdx.x / G;
    const int Y = a.S[1], Z = a.S[2];
    float ca[4], cb[4];
    {
        const float *s = sums + (long long)b * 3 * L;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = lg * 4 + k;
            const float num = 2.0f * s[l] + eps, den = s[L + l] + s[2 * L + l] + eps;
            const float gd = gdice[(long long)b * L + l];
            ca[k] = 0.0f; cb[k] = 0.0f;
            if (den != 0.0f) { ca[k] = 2.0f * gd / den; cb[k] = -2.0f * gd * num / (den * den); }
        }
    }
    // two voxels per lane-group in flight: all 18 row loads of an iteration are issued before the first use
    constexpr int U = 2;
    const unsigned ngroups = gridDim.x * NG;
    unsigned niter = (a.nout + ngroups * U - 1) / (ngroups * U);
    if (xm) niter = ((unsigned)xm_len + U - 1) / U;
    for (unsigned it = 0; it < niter; ++it) {
        unsigned q[U];
        bool live[U], oob[U];
        float w0[U][3], w1[U][3], m[U][3];
        nrt_f4 t[U], v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned qq;
            if (xm) {
                const int x = xm_x0 + (int)(it * U + u), y = xm_y0 + (int)(g >> ba.tg.ltz), z = xm_z0 + (int)(g & ((1u << ba.tg.ltz) - 1u));
                const bool in = x < xm_x0 + xm_len && y < a.O[1] && z < a.O[2];
                qq = in ? ((unsigned)x * (unsigned)a.O[1] + (unsigned)y) * (unsigned)a.O[2] + (unsigned)z : 0xffffffffu;
            } else {
                qq = blockIdx.x * NG + g + (it * U + u) * ngroups;
            }
            live[u] = qq < a.nout;
            q[u] = live[u] ? qq : a.nout - 1;
            int qd[NRT_MAXD];
            float p[NRT_MAXD];
            decode<D>(a, q[u], qd);
            load_loc<D, MODE>(a, locb, q[u], qd, p);
            oob[u] = a.has_fill ? out_of_bounds<D>(a, p) : false;
            int i0[3], i1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                corner_1d(p[d], a.S[d], i0[d], i1[d], w0[u][d], w1[u][d]);
                m[u][d] = (p[d] >= 0.0f && p[d] <= (float)(a.S[d] - 1)) ? 1.0f : 0.0f;
            }
            t[u] = fix[(long long)q[u] * G + lg];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = (corner >> 2) & 1, by = (corner >> 1) & 1, bz = corner & 1;
                const long long idx = ((long long)(bx ? i1[0] : i0[0]) * Y + (by ? i1[1] : i0[1])) * Z + (bz ? i1[2] : i0[2]);
                v[u][corner] = vol[idx * G + lg];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            nrt_f4 wp = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = (corner >> 2) & 1, by = (corner >> 1) & 1, bz = corner & 1;
                const float wt = (bx ? w1[u][0] : w0[u][0]) * (by ? w1[u][1] : w0[u][1]) * (bz ? w1[u][2] : w0[u][2]);
#pragma unroll
                for (int k = 0; k < 4; ++k) wp[k] += wt * v[u][corner][k];
            }
            nrt_f4 gq;
#pragma unroll
            for (int k = 0; k < 4; ++k) gq[k] = (oob[u] || !live[u]) ? 0.0f : ca[k] * t[u][k] + cb[k] * wp[k];
            float gacc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = (corner >> 2) & 1, by = (corner >> 1) & 1, bz = corner & 1;
                const float wx = bx ? w1[u][0] : w0[u][0], wy = by ? w1[u][1] : w0[u][1], wz = bz ? w1[u][2] : w0[u][2];
                const nrt_f4 c = v[u][corner];
                const float dot = gq[0] * c[0] + gq[1] * c[1] + gq[2] * c[2] + gq[3] * c[3];
                gacc[0] += dot * (bx ? m[u][0] : -m[u][0]) * wy * wz;
                gacc[1] += dot * wx * (by ? m[u][1] : -m[u][1]) * wz;
                gacc[2] += dot * wx * wy * (bz ? m[u][2] : -m[u][2]);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int off = 1; off < G; off <<= 1) gacc[d] += __shfl_xor(gacc[d], off, 64);
            if (live[u] && lg == 0) {
                float *dst = gl + (long long)q[u] * 3;
                dst[0] = gacc[0]; dst[1] = gacc[1]; dst[2] = gacc[2];
            }
        }
    }
}

// The same backward for 32 labels on the x-march schedule, software-pipelined like the fused forward (fused.hip): the rows of
// x-plane p + 1 are requested before the gradient of plane p is formed, the location of plane p + 2 before those rows, the
// voxel's position comes from the block's patch (no division), so the queue of the memory pipeline never drains.  The
// gradient is the same sum as warp_dice_bwd_rows', associated differently: the eight inner products first, then per axis four
// weighted differences of corner pairs (the un-pipelined kernel spends 350 VALU instructions per voxel step, twice the forward).
// DICE = false: the plain d out / d loc of the warp (interpn_bwd_rows' grad_loc at 32 channels): `fixed` is then grad_out, whose row
// takes the place of d dice / d warped, and the warped row is not needed.
template <int MODE, bool DICE = true>
__global__ __launch_bounds__(256, 3) void warp_dice_bwd_xm(InterpBwdArgs ba, const float *__restrict__ fixed,
                                                           const float *__restrict__ sums,
                                                           const float *__restrict__ gdice, float eps) {
    constexpr int G = 8, L = 32;
    const InterpArgs &a = ba.f;
    int b, x0, y0, z0, xlen;
    unsigned prow;
    if (!xmarch_block(ba.tg, a.O[0], b, prow, x0, y0, z0, xlen)) return;
    const char *volb = (const char *)((const float *)a.vol + (long long)b * a.vol_bs);
    const char *locb = (const char *)(a.loc + (long long)b * a.loc_bs);
    const char *fix = (const char *)(fixed + (long long)b * a.out_bs);
    float *gl = ba.gloc + (long long)b * a.nout * 3;
    const int lg = threadIdx.x % G;
    const unsigned g = threadIdx.x / G;
    const unsigned SY = (unsigned)a.S[1], SZ = (unsigned)a.S[2];
    float ca[4] = {0.0f, 0.0f, 0.0f, 0.0f}, cb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (DICE) {
        const float *sm = sums + (long long)b * 3 * L;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = lg * 4 + k;
            const float num = 2.0f * sm[l] + eps, den = sm[L + l] + sm[2 * L + l] + eps;
            const float gd = gdice[(long long)b * L + l];
            ca[k] = 0.0f; cb[k] = 0.0f;
            if (den != 0.0f) { ca[k] = 2.0f * gd / den; cb[k] = -2.0f * gd * num / (den * den); }
        }
    }
    const int y = y0 + (int)(g >> ba.tg.ltz), z = z0 + (int)(g & ((1u << ba.tg.ltz) - 1u));
    const bool inyz = y < a.O[1] && z < a.O[2];
    const int yc = min(y, a.O[1] - 1), zc = min(z, a.O[2] - 1);
    const unsigned plane = (unsigned)(a.O[1] * a.O[2]), qyz = (unsigned)(yc * a.O[2] + zc);
    const int last = xlen - 1;

    auto fetch_loc = [&](int pass, float (&pr)[3]) {
        const unsigned q = (unsigned)(x0 + min(pass, last)) * plane + qyz;
        const float *lp = (const float *)(locb + (size_t)(q * 12u));
        pr[0] = lp[0]; pr[1] = lp[1]; pr[2] = lp[2];
    };
    auto prepare = [&](int pass, const float (&pr)[3], float &W0x, float &W0y, float &W0z, float &Mx, float &My, float &Mz,
                       unsigned &Q, bool &LIVE, bool &DEAD, unsigned (&off)[8]) {
        const int x = x0 + min(pass, last);
        Q = (unsigned)x * plane + qyz;
        LIVE = inyz && pass < xlen;
        float p[3];
        const int qd[3] = {x, yc, zc};
#pragma unroll
        for (int d = 0; d < 3; ++d) p[d] = (MODE == NRT_LOC_ABSOLUTE) ? pr[d] : nrt_add((float)qd[d], pr[d]);
        bool oob = false;
        if (a.has_fill) {
#pragma unroll
            for (int d = 0; d < 3; ++d) oob = oob || (p[d] < 0.0f) || (p[d] > (float)(a.S[d] - 1));
        }
        DEAD = oob || !LIVE;
        int i0x, i1x, i0y, i1y, i0z, i1z;
        float w1;
        corner_1d(p[0], a.S[0], i0x, i1x, W0x, w1);
        corner_1d(p[1], a.S[1], i0y, i1y, W0y, w1);
        corner_1d(p[2], a.S[2], i0z, i1z, W0z, w1);
        Mx = (p[0] >= 0.0f && p[0] <= (float)(a.S[0] - 1)) ? 1.0f : 0.0f;
        My = (p[1] >= 0.0f && p[1] <= (float)(a.S[1] - 1)) ? 1.0f : 0.0f;
        Mz = (p[2] >= 0.0f && p[2] <= (float)(a.S[2] - 1)) ? 1.0f : 0.0f;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const unsigned ix = (corner & 4) ? i1x : i0x, iy = (corner & 2) ? i1y : i0y, iz = (corner & 1) ? i1z : i0z;
            off[corner] = (((ix * SY + iy) * SZ + iz) * (unsigned)G + (unsigned)lg) * 16u;
        }
    };
    auto load_rows = [&](const unsigned (&off)[8], unsigned q, nrt_f4 (&R)[8], nrt_f4 &T) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) R[corner] = *(const nrt_f4 *)(volb + (size_t)off[corner]);
        T = __builtin_nontemporal_load((const nrt_f4 *)(fix + (size_t)((q * (unsigned)G + (unsigned)lg) * 16u)));
    };
    auto finish = [&](float W0x, float W0y, float W0z, float Mx, float My, float Mz, unsigned Q, bool LIVE, bool DEAD,
                      const nrt_f4 (&v)[8], const nrt_f4 &t) {
        const float W1x = nrt_sub(1.0f, W0x), W1y = nrt_sub(1.0f, W0y), W1z = nrt_sub(1.0f, W0z);       // corner_1d's w1
        // the warped row, blended as the forward blends it ((wx wy) wz per corner, corners in order), two channels per packed op
        const nrt_f2 wy2 = {W0y, W1y}, wz2 = {W0z, W1z};
        const nrt_f2 wxy0 = (nrt_f2){W0x, W0x} * wy2, wxy1 = (nrt_f2){W1x, W1x} * wy2;     // wx wy: [x0y0, x0y1], [x1y0, x1y1]
        nrt_f2 wt2[4];
        wt2[0] = (nrt_f2){wxy0[0], wxy0[0]} * wz2;
        wt2[1] = (nrt_f2){wxy0[1], wxy0[1]} * wz2;
        wt2[2] = (nrt_f2){wxy1[0], wxy1[0]} * wz2;
        wt2[3] = (nrt_f2){wxy1[1], wxy1[1]} * wz2;
        nrt_f2 gl2 = {t[0], t[1]}, gh2 = {t[2], t[3]};                 // DICE = false: the incoming gradient row itself
        if (DICE) {
            nrt_f2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const float wt = wt2[corner >> 1][corner & 1];
                const nrt_f2 w2 = {wt, wt};
                al = al + w2 * (nrt_f2){v[corner][0], v[corner][1]};
                ah = ah + w2 * (nrt_f2){v[corner][2], v[corner][3]};
            }
            // d dice / d warped for this lane's four labels
            gl2 = __builtin_elementwise_fma((nrt_f2){cb[0], cb[1]}, al, (nrt_f2){ca[0], ca[1]} * gl2);
            gh2 = __builtin_elementwise_fma((nrt_f2){cb[2], cb[3]}, ah, (nrt_f2){ca[2], ca[3]} * gh2);
        }
        if (DEAD) { gl2 = (nrt_f2){0.0f, 0.0f}; gh2 = gl2; }
        float gacc[3];
        loc_grad_rows(v, gl2, gh2, W0x, W1x, W0y, W1y, W0z, W1z, Mx, My, Mz, gacc);
        // sum over the voxel's 8 lanes on the DPP network (quad xor 1, quad xor 2, then the mirrored half: after the two quad steps a
        // quad's lanes hold the same value, so i <-> 7 - i adds the other quad exactly as xor 4 would); no LDS round trips
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float r = gacc[d];
            r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0xB1, 0xF, 0xF, true));
            r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x4E, 0xF, 0xF, true));
            r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x141, 0xF, 0xF, true));
            gacc[d] = r;
        }
        if (LIVE && lg == 0) {
            float *dst = gl + (size_t)Q * 3;
            dst[0] = gacc[0]; dst[1] = gacc[1]; dst[2] = gacc[2];
        }
    };

    nrt_f4 Ra[8], Rb[8], Ta, Tb;
    float Ax, Ay, Az, Amx, Amy, Amz, Bx, By, Bz, Bmx, Bmy, Bmz;
    unsigned Aq, Bq;
    bool Al, Ad, Bl, Bd;
    unsigned off[8];
    float pn[3];
    fetch_loc(0, pn);
    prepare(0, pn, Ax, Ay, Az, Amx, Amy, Amz, Aq, Al, Ad, off);
    fetch_loc(1, pn);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(off, Aq, Ra, Ta);
    __builtin_amdgcn_sched_barrier(0);
    for (int pass = 0; pass < xlen; pass += 2) {
        if (ba.tg.depth_sync > 0 && (pass & (ba.tg.depth_sync - 1)) == 0) __builtin_amdgcn_s_barrier();     // see NRT_FUSED_SYNC (fused.hip)
        prepare(pass + 1, pn, Bx, By, Bz, Bmx, Bmy, Bmz, Bq, Bl, Bd, off);
        __builtin_amdgcn_sched_barrier(0);
        fetch_loc(pass + 2, pn);
        load_rows(off, Bq, Rb, Tb);
        __builtin_amdgcn_sched_barrier(0);
        finish(Ax, Ay, Az, Amx, Amy, Amz, Aq, Al, Ad, Ra, Ta);
        __builtin_amdgcn_sched_barrier(0);
        prepare(pass + 2, pn, Ax, Ay, Az, Amx, Amy, Amz, Aq, Al, Ad, off);
        __builtin_amdgcn_sched_barrier(0);
        fetch_loc(pass + 3, pn);
        load_rows(off, Aq, Ra, Ta);
        __builtin_amdgcn_sched_barrier(0);
        finish(Bx, By, Bz, Bmx, Bmy, Bmz, Bq, Bl, Bd, Rb, Tb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// d out / d vol at C = 32 with the duplicate rows merged on chip before they reach L2.
//
// The scatter of interpn_bwd_rows sends 8 row-atomics (32 float atomics each) per voxel to L2, whose atomic units are the
// bottleneck: 3.6 ms per 160^3 x 32 volume (~67 clk of CU time per row).  Neighbouring voxels hit the same rows again and
// again -- 64 voxels of two x-planes of a 4 x 8 patch touch ~196 distinct rows with their 512 corner references, and the
// next planes re-touch most of them.  Here a block keeps an LDS table of 1024 row accumulators (128 KB; the x-march
// schedule runs one block per CU anyway): a (row, weight) pair claims or finds its row's slot with one `ds_cmpswap` probe
// sequence, its 32 weighted gradient values are added with LDS float atomics, and only when the table fills up (or the
// block ends) are the live rows flushed to global memory, one row-atomic each.  Pairs that find no slot in 8 probes go to
// global memory directly.  Summation order differs from the plain scatter (float atomics already made it unordered).
// G = 8 (C = 32), x-march schedule; computes d vol only (d loc has its own pass in interpn_bwd_rows).
//
// MEASURED (round 2, one 160^3 x 32 volume, tools/bwd_vol_bench.py with phases switched off one at a time): the merge works --
// the flushes' global atomics cost 0.06 ms in total -- but the LDS float atomics that feed the table take 5.8 ms, the slot
// search 1.0 ms, everything else 0.75 ms: 7.6 ms against 3.6 ms for the plain scatter.  ds_add_f32 retires about one lane
// every 3.4 clk per CU (0.3 lane-atomics per clk), SLOWER than the L2 atomic units serve the same CU (0.47 per clk at 3.6 ms).
// An on-chip merge therefore has to accumulate without LDS atomics (owner-computes over per-row chains, or a sort); this kernel
// is kept as the correct, selectable experiment (env NRT_BWD_VOL_DEDUP=1); interpn_bwd_vol_sort below is the merge that pays.
constexpr int BV_SLOTS = 1024;           // power of two
constexpr int BV_NG = 32;                // lane-groups (voxels) per x-plane of the block's 4 x 8 patch
constexpr int BV_U = 2;                  // x-planes per iteration
constexpr unsigned BV_EMPTY = 0xffffffffu;

template <int MODE>
__global__ __launch_bounds__(256) void interpn_bwd_vol_dedup(InterpBwdArgs ba) {
    constexpr int G = 8, C = 32, NPAIR = BV_NG * BV_U * 8;
    const InterpArgs &a = ba.f;
    int b = 0, x0 = 0, y0 = 0, z0 = 0, xlen = 0;
    unsigned prow;
    if (!xmarch_block(ba.tg, a.O[0], b, prow, x0, y0, z0, xlen)) return;
    extern __shared__ __attribute__((aligned(16))) float bv_lds[];
    float *acc = bv_lds;                                   // [BV_SLOTS][C]
    unsigned *tag = (unsigned *)(acc + BV_SLOTS * C);      // [BV_SLOTS]
    float *s_g = (float *)(tag + BV_SLOTS);                // [BV_NG * BV_U][C]
    unsigned *s_idx = (unsigned *)(s_g + BV_NG * BV_U * C);   // [NPAIR]
    float *s_wt = (float *)(s_idx + NPAIR);                // [NPAIR]
    unsigned *s_slot = (unsigned *)(s_wt + NPAIR);         // [NPAIR]
    __shared__ unsigned live_rows;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const nrt_f4 *go = (const nrt_f4 *)(ba.gout + (long long)b * a.out_bs);
    float *gv = ba.gvol + (long long)b * a.vol_bs;
    const int lg = threadIdx.x % G;
    const unsigned g = threadIdx.x / G;
    const unsigned Y = (unsigned)a.S[1], Z = (unsigned)a.S[2];
    for (int i = threadIdx.x; i < BV_SLOTS * C; i += 256) acc[i] = 0.0f;
    for (int i = threadIdx.x; i < BV_SLOTS; i += 256) tag[i] = BV_EMPTY;
    if (threadIdx.x == 0) live_rows = 0;
    __syncthreads();

    auto flush = [&]() {
        // every live row goes out as one 128-byte row of float atomics (a wave covers two rows per instruction)
        for (unsigned i = threadIdx.x; i < (unsigned)(BV_SLOTS * C); i += 256) {
            const unsigned slot = i / C, ch = i % C;
            const unsigned row = tag[slot];
            if (row != BV_EMPTY) {
                const float v = acc[i];
                if (v != 0.0f) atomic_add_f32(gv + (size_t)row * C + ch, v);
                acc[i] = 0.0f;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < BV_SLOTS; i += 256) tag[i] = BV_EMPTY;
        if (threadIdx.x == 0) live_rows = 0;
        __syncthreads();
    };

    const unsigned niter = ((unsigned)xlen + BV_U - 1) / BV_U;
    // the location and the gradient row of the next iteration's voxels are requested before this iteration's table work
    float pre_p[BV_U][3];
    nrt_f4 pre_g[BV_U];
    bool pre_in[BV_U];
    auto prefetch = [&](unsigned it) {
#pragma unroll
        for (int u = 0; u < BV_U; ++u) {
            const int x = x0 + (int)(it * BV_U + u), y = y0 + (int)(g >> ba.tg.ltz), z = z0 + (int)(g & ((1u << ba.tg.ltz) - 1u));
            pre_in[u] = x < x0 + xlen && y < a.O[1] && z < a.O[2];
            const unsigned q = pre_in[u] ? ((unsigned)x * (unsigned)a.O[1] + (unsigned)y) * (unsigned)a.O[2] + (unsigned)z : a.nout - 1;
            int qd[NRT_MAXD];
            float p[NRT_MAXD];
            decode<3>(a, q, qd);
            load_loc<3, MODE>(a, locb, q, qd, p);
            pre_p[u][0] = p[0]; pre_p[u][1] = p[1]; pre_p[u][2] = p[2];
            pre_g[u] = go[(long long)q * G + lg];
        }
    };
    prefetch(0);
    for (unsigned it = 0; it < niter; ++it) {
        // ---- file the iteration's (row, weight) pairs and gradient rows ------------------------------------------------
#pragma unroll
        for (int u = 0; u < BV_U; ++u) {
            float p[NRT_MAXD] = {pre_p[u][0], pre_p[u][1], pre_p[u][2]};
            const bool oob = a.has_fill ? out_of_bounds<3>(a, p) : false;
            int i0[3], i1[3];
            float w0[3], w1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
            nrt_f4 gq = pre_g[u];
            const bool dead = !pre_in[u] || oob;
            if (dead) gq = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
            const unsigned slot = (unsigned)u * BV_NG + g;
            ((nrt_f4 *)s_g)[slot * G + lg] = gq;
            const int bx = (lg >> 2) & 1, by = (lg >> 1) & 1, bz = lg & 1;       // lane lg files corner lg
            const unsigned ix = bx ? i1[0] : i0[0], iy = by ? i1[1] : i0[1], iz = bz ? i1[2] : i0[2];
            s_idx[slot * 8 + lg] = (ix * Y + iy) * Z + iz;
            s_wt[slot * 8 + lg] = dead ? 0.0f : (bx ? w1[0] : w0[0]) * (by ? w1[1] : w0[1]) * (bz ? w1[2] : w0[2]);
        }
        __syncthreads();
        if (it + 1 < niter) prefetch(it + 1);
        // ---- every pair finds (or claims) the accumulator of its row --------------------------------------------------
        for (unsigned pi = threadIdx.x; pi < (unsigned)NPAIR; pi += 256) {
            unsigned found = BV_EMPTY;                          // BV_EMPTY = no slot: straight to global memory
            if (s_wt[pi] != 0.0f) {
                const unsigned row = s_idx[pi];
                unsigned h = (row * 2654435761u) >> 22;         // multiplicative hash, top 10 bits
#pragma unroll 1
                for (int probe = 0; probe < 8; ++probe) {
                    const unsigned old = atomicCAS(&tag[h], BV_EMPTY, row);
                    if (old == BV_EMPTY) { atomicAdd(&live_rows, 1u); found = h; break; }
                    if (old == row) { found = h; break; }
                    h = (h + 1u) & (BV_SLOTS - 1u);
                }
            }
            s_slot[pi] = found;
        }
        __syncthreads();
        // ---- accumulate: a lane-group takes one pair, a lane 4 channels of its gradient row (one 16-byte LDS read, four LDS
        // float atomics); the 16 passes are independent and issue back to back ------------------------------------------
#pragma unroll
        for (int pass = 0; pass < NPAIR / 32; ++pass) {
            const unsigned r = (unsigned)pass * 32u + g;
            const float wt = s_wt[r];
            const unsigned slot = s_slot[r];
            const nrt_f4 gr = ((const nrt_f4 *)s_g)[(r >> 3) * G + lg];
            if (wt != 0.0f) {
                if (slot != BV_EMPTY) {
                    float *dst = acc + slot * C + 4 * lg;
#pragma unroll
                    for (int c = 0; c < 4; ++c) atomicAdd(dst + c, wt * gr[c]);
                } else {
                    float *dst = gv + (size_t)s_idx[r] * C + 4 * lg;
#pragma unroll
                    for (int c = 0; c < 4; ++c) atomic_add_f32(dst + c, wt * gr[c]);
                }
            }
        }
        __syncthreads();
        if (live_rows > (unsigned)(BV_SLOTS * 5 / 8)) flush();   // uniform: live_rows is read after the barrier by everyone
    }
    flush();
}

// ---------------------------------------------------------------------------------------------------------------------
// d out / d vol at C = 32, duplicate rows merged by a counting sort (no LDS float atomics).
//
// Per iteration a block files the (row, weight) pairs of BS_U x-planes of its 4 x 8 patch (128 voxels, 1024 pairs: on the bench
// field they name ~310 distinct rows, tools/ measurements in DESIGN 4.7).  Then, with integer LDS atomics only:
//   1. every pair finds or claims the slot of its row in an open-addressing table (ds_cmpswap) and takes a rank inside the slot
//      (returning ds_add_u32);
//   2. a block scan over the slot counts gives every used slot a segment of the sorted pair list and an index in the list of
//      distinct rows;
//   3. every pair writes itself to segment offset + rank;
//   4. a half-wave (32 lanes = the 32 channels) per distinct row walks the row's segment, sums weight * gradient in a register
//      and sends ONE row of float atomics to memory, then clears the slot.
// The L2 atomic units, the bottleneck of the plain scatter (interpn_bwd_rows), see 1024 / 310 = 3.3x fewer rows.
// MEASURED (round 2, one 160^3 x 32 volume, tools/bwd_vol_bench.py, time including the 524 MB zero fill): plain scatter 3.61 ms,
// this kernel 1.68 ms with 256 threads and 1.33 ms with 512 (two x-planes filed at a time, 16 half-waves merging); worst-case
// field U(-80, 80): 2.87 -> 2.58 ms.  Default for d vol at C = 32 (NRT_BWD_VOL_DEDUP=0 selects the plain scatter).
constexpr int BS_SLOTS = 2048;           // power of two, >= 2 x pairs per iteration
constexpr int BS_U = 4;                  // x-planes per iteration
constexpr int BS_NV = BV_NG * BS_U;      // voxels per iteration
constexpr int BS_NPAIR = BS_NV * 8;
constexpr int BS_NT = 512;               // threads: two x-planes are filed at a time, 16 half-waves merge rows

template <int MODE>
__global__ __launch_bounds__(BS_NT) void interpn_bwd_vol_sort(InterpBwdArgs ba) {
    constexpr int G = 8, C = 32, NT = BS_NT, NW = NT / 64, SPT = BS_SLOTS / NT, PL = NT / (BV_NG * G);   // PL: planes filed per pass
    static_assert(BS_U % PL == 0 && BS_SLOTS % NT == 0, "geometry");
    const InterpArgs &a = ba.f;
    int b = 0, x0 = 0, y0 = 0, z0 = 0, xlen = 0;
    unsigned prow;
    if (!xmarch_block(ba.tg, a.O[0], b, prow, x0, y0, z0, xlen)) return;
    __shared__ __attribute__((aligned(16))) float s_g[BS_NV * C];     // gradient rows of the iteration's voxels
    __shared__ unsigned s_idx[BS_NPAIR];                              // row of a pair; then its slot | rank << 16
    __shared__ float s_wt[BS_NPAIR];                                  // its weight
    __shared__ unsigned short s_sorted[BS_NPAIR];                     // pairs ordered by slot
    __shared__ unsigned tag[BS_SLOTS];                                // row held by a slot
    __shared__ unsigned cnt[BS_SLOTS];                                // pairs of the slot, then offset | count << 16
    __shared__ unsigned short ulist[BS_NPAIR];                        // used slots
    __shared__ unsigned wsum[NW];
    __shared__ unsigned nuniq_s;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const nrt_f4 *go = (const nrt_f4 *)(ba.gout + (long long)b * a.out_bs);
    float *gv = ba.gvol + (long long)b * a.vol_bs;
    const int lg = threadIdx.x % G;
    const unsigned g = (threadIdx.x / G) % BV_NG, pl = threadIdx.x / (G * BV_NG);     // patch voxel, plane of the pass
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned Y = (unsigned)a.S[1], Z = (unsigned)a.S[2];
    for (int i = threadIdx.x; i < BS_SLOTS; i += NT) { tag[i] = BV_EMPTY; cnt[i] = 0u; }
    __syncthreads();

    const unsigned niter = ((unsigned)xlen + BS_U - 1) / BS_U;
    for (unsigned it = 0; it < niter; ++it) {
        // ---- file the iteration's (row, weight) pairs and gradient rows ------------------------------------------------
#pragma unroll
        for (int up = 0; up < BS_U / PL; ++up) {
            const unsigned u = (unsigned)up * PL + pl;
            const int x = x0 + (int)(it * BS_U + u), y = y0 + (int)(g >> ba.tg.ltz), z = z0 + (int)(g & ((1u << ba.tg.ltz) - 1u));
            const bool in = x < x0 + xlen && y < a.O[1] && z < a.O[2];
            const unsigned q = in ? ((unsigned)x * (unsigned)a.O[1] + (unsigned)y) * (unsigned)a.O[2] + (unsigned)z : a.nout - 1;
            int qd[NRT_MAXD];
            float p[NRT_MAXD];
            decode<3>(a, q, qd);
            load_loc<3, MODE>(a, locb, q, qd, p);
            nrt_f4 gq = go[(long long)q * G + lg];
            const bool oob = a.has_fill ? out_of_bounds<3>(a, p) : false;
            int i0[3], i1[3];
            float w0[3], w1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
            const bool dead = !in || oob;
            if (dead) gq = (nrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
            const unsigned vslot = u * BV_NG + g;
            ((nrt_f4 *)s_g)[vslot * G + lg] = gq;
            const int bx = (lg >> 2) & 1, by = (lg >> 1) & 1, bz = lg & 1;       // lane lg files corner lg
            const unsigned ix = bx ? i1[0] : i0[0], iy = by ? i1[1] : i0[1], iz = bz ? i1[2] : i0[2];
            s_idx[vslot * 8 + lg] = dead ? BV_EMPTY : (ix * Y + iy) * Z + iz;       // a dead pair names no row
            s_wt[vslot * 8 + lg] = dead ? 0.0f : (bx ? w1[0] : w0[0]) * (by ? w1[1] : w0[1]) * (bz ? w1[2] : w0[2]);
        }
        __syncthreads();
        // ---- 1. slot and rank of every pair ----------------------------------------------------------------------------
        for (unsigned pi = threadIdx.x; pi < (unsigned)BS_NPAIR; pi += NT) {
            unsigned found = BV_EMPTY;
            const unsigned row = s_idx[pi];
            if (row != BV_EMPTY) {
                unsigned h = (row * 2654435761u) >> 21;         // multiplicative hash, top 11 bits
#pragma unroll 1
                for (;;) {                                      // <= 1024 distinct rows in 2048 slots: always ends
                    const unsigned old = atomicCAS(&tag[h], BV_EMPTY, row);
                    if (old == BV_EMPTY || old == row) break;
                    h = (h + 1u) & (BS_SLOTS - 1u);
                }
                found = h | (atomicAdd(&cnt[h], 1u) << 16);
            }
            s_idx[pi] = found;                                  // the row lives in tag[slot] from here on
        }
        __syncthreads();
        // ---- 2. segments: exclusive scan of the slot counts (pairs | used slots << 16), SPT slots per thread --------------
        {
            unsigned c[SPT], tot = 0u;
#pragma unroll
            for (int j = 0; j < SPT; ++j) { c[j] = cnt[threadIdx.x * SPT + j]; tot += c[j] | (c[j] ? 0x10000u : 0u); }
            unsigned incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = (unsigned)__shfl_up((int)incl, off, 64);
                if (lane >= (unsigned)off) incl += o;
            }
            if (lane == 63u) wsum[wv] = incl;
            __syncthreads();
            unsigned base = incl - tot;
            for (unsigned w2 = 0; w2 < wv; ++w2) base += wsum[w2];
            if (threadIdx.x == NT - 1) nuniq_s = (base + tot) >> 16;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (c[j]) {
                    cnt[threadIdx.x * SPT + j] = (base & 0xffffu) | (c[j] << 16);
                    ulist[base >> 16] = (unsigned short)(threadIdx.x * SPT + j);
                    base += c[j] | 0x10000u;
                }
            }
        }
        __syncthreads();
        // ---- 3. pairs into their segments ---------------------------------------------------------------------------------
        for (unsigned pi = threadIdx.x; pi < (unsigned)BS_NPAIR; pi += NT) {
            const unsigned sl = s_idx[pi];
            if (sl != BV_EMPTY) s_sorted[(cnt[sl & 0xffffu] & 0xffffu) + (sl >> 16)] = (unsigned short)pi;
        }
        __syncthreads();
        // ---- 4. one half-wave per distinct row: sum its pairs, one row of atomics, clear the slot ---------------------------
        {
            const unsigned grp = threadIdx.x >> 5, ch = threadIdx.x & 31u;
            const unsigned nuniq = nuniq_s;
            for (unsigned u = grp; u < nuniq; u += NT / 32) {
                const unsigned slot = ulist[u];
                const unsigned sg = cnt[slot], off = sg & 0xffffu, n = sg >> 16;
                const unsigned row = tag[slot];
                float acc = 0.0f;
                for (unsigned j = 0; j < n; ++j) {
                    const unsigned e = s_sorted[off + j];
                    acc += s_wt[e] * s_g[(e >> 3) * C + ch];
                }
                if (acc != 0.0f) atomic_add_f32(gv + (size_t)row * C + ch, acc);
                if (ch == 0u) { tag[slot] = BV_EMPTY; cnt[slot] = 0u; }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The counting-sort merge of interpn_bwd_vol_sort for few channels (C <= 8: flow fields -- the backward of VecInt / compose --
// and warped few-channel network outputs), 3-D.  A block takes 4 x 4 x 8 tiles of output voxels (128 voxels, 1024 pairs; the
// distinct rows are about a third of the pairs on the bench field); the rows are C floats, so the merge phase runs over
// (distinct row, channel) elements.  Same steps, integer LDS atomics only.
constexpr int BG_NV = 128, BG_NPAIR = BG_NV * 8, BG_SLOTS = 2048, BG_CMAX = 8;

template <int MODE>
__global__ __launch_bounds__(256) void interpn_bwd_vol_sort_any(InterpBwdArgs ba, unsigned nTy, unsigned nTz, unsigned ntiles) {
    constexpr int NT = 256, SPT = BG_SLOTS / NT;
    const InterpArgs &a = ba.f;
    const int C = a.C, b = blockIdx.y;
    __shared__ float s_g[BG_NV * BG_CMAX];
    __shared__ unsigned s_idx[BG_NPAIR];
    __shared__ float s_wt[BG_NPAIR];
    __shared__ unsigned short s_sorted[BG_NPAIR];
    __shared__ unsigned tag[BG_SLOTS];
    __shared__ unsigned cnt[BG_SLOTS];
    __shared__ unsigned short ulist[BG_NPAIR];
    __shared__ unsigned wsum[NT / 64];
    __shared__ unsigned nuniq_s;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const float *go = ba.gout + (long long)b * a.out_bs;
    float *gv = ba.gvol + (long long)b * a.vol_bs;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned Y = (unsigned)a.S[1], Z = (unsigned)a.S[2];
    for (int i = threadIdx.x; i < BG_SLOTS; i += NT) { tag[i] = BV_EMPTY; cnt[i] = 0u; }
    __syncthreads();
    const unsigned vi = threadIdx.x & 127u, half = threadIdx.x >> 7;            // voxel of the tile, which four corners it files
    const int lx = (int)(vi >> 5), ly = (int)((vi >> 3) & 3u), lz = (int)(vi & 7u);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned tz = tile % nTz, t2 = tile / nTz, ty = t2 % nTy, tx = t2 / nTy;
        // ---- file the tile's (row, weight) pairs and gradient rows -------------------------------------------------------
        {
            const int x = (int)(tx << 2) + lx, y = (int)(ty << 2) + ly, z = (int)(tz << 3) + lz;
            const bool in = x < a.O[0] && y < a.O[1] && z < a.O[2];
            const unsigned q = in ? ((unsigned)x * (unsigned)a.O[1] + (unsigned)y) * (unsigned)a.O[2] + (unsigned)z : a.nout - 1;
            int qd[NRT_MAXD];
            float p[NRT_MAXD];
            decode<3>(a, q, qd);
            load_loc<3, MODE>(a, locb, q, qd, p);
            const bool oob = a.has_fill ? out_of_bounds<3>(a, p) : false;
            int i0[3], i1[3];
            float w0[3], w1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) corner_1d(p[d], a.S[d], i0[d], i1[d], w0[d], w1[d]);
            const bool dead = !in || oob;
            if (half == 0u) {
                for (int c = 0; c < C; ++c) s_g[vi * (unsigned)C + c] = dead ? 0.0f : go[(long long)q * C + c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int corner = (int)half * 4 + k;
                const int bx = (corner >> 2) & 1, by = (corner >> 1) & 1, bz = corner & 1;
                const unsigned ix = bx ? i1[0] : i0[0], iy = by ? i1[1] : i0[1], iz = bz ? i1[2] : i0[2];
                s_idx[vi * 8u + corner] = dead ? BV_EMPTY : (ix * Y + iy) * Z + iz;
                s_wt[vi * 8u + corner] = dead ? 0.0f : (bx ? w1[0] : w0[0]) * (by ? w1[1] : w0[1]) * (bz ? w1[2] : w0[2]);
            }
        }
        __syncthreads();
        // ---- 1. slot and rank of every pair ----------------------------------------------------------------------------
        for (unsigned pi = threadIdx.x; pi < (unsigned)BG_NPAIR; pi += NT) {
            unsigned found = BV_EMPTY;
            const unsigned row = s_idx[pi];
            if (row != BV_EMPTY) {
                unsigned h = (row * 2654435761u) >> 21;
#pragma unroll 1
                for (;;) {
                    const unsigned old = atomicCAS(&tag[h], BV_EMPTY, row);
                    if (old == BV_EMPTY || old == row) break;
                    h = (h + 1u) & (BG_SLOTS - 1u);
                }
                found = h | (atomicAdd(&cnt[h], 1u) << 16);
            }
            s_idx[pi] = found;
        }
        __syncthreads();
        // ---- 2. segments ---------------------------------------------------------------------------------------------------
        {
            unsigned c[SPT], tot = 0u;
#pragma unroll
            for (int j = 0; j < SPT; ++j) { c[j] = cnt[threadIdx.x * SPT + j]; tot += c[j] | (c[j] ? 0x10000u : 0u); }
            unsigned incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = (unsigned)__shfl_up((int)incl, off, 64);
                if (lane >= (unsigned)off) incl += o;
            }
            if (lane == 63u) wsum[wv] = incl;
            __syncthreads();
            unsigned base = incl - tot;
            for (unsigned w2 = 0; w2 < wv; ++w2) base += wsum[w2];
            if (threadIdx.x == NT - 1) nuniq_s = (base + tot) >> 16;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (c[j]) {
                    cnt[threadIdx.x * SPT + j] = (base & 0xffffu) | (c[j] << 16);
                    ulist[base >> 16] = (unsigned short)(threadIdx.x * SPT + j);
                    base += c[j] | 0x10000u;
                }
            }
        }
        __syncthreads();
        // ---- 3. pairs into their segments ---------------------------------------------------------------------------------
        for (unsigned pi = threadIdx.x; pi < (unsigned)BG_NPAIR; pi += NT) {
            const unsigned sl = s_idx[pi];
            if (sl != BV_EMPTY) s_sorted[(cnt[sl & 0xffffu] & 0xffffu) + (sl >> 16)] = (unsigned short)pi;
        }
        __syncthreads();
        // ---- 4. one thread per (distinct row, channel): sum the row's pairs, one atomic --------------------------------------
        const unsigned nuniq = nuniq_s;
        for (unsigned e = threadIdx.x; e < nuniq * (unsigned)C; e += NT) {
            const unsigned u = e / (unsigned)C, ch = e - u * (unsigned)C;
            const unsigned slot = ulist[u];
            const unsigned sg = cnt[slot], off = sg & 0xffffu, n = sg >> 16;
            float acc = 0.0f;
            for (unsigned j = 0; j < n; ++j) {
                const unsigned pe = s_sorted[off + j];
                acc += s_wt[pe] * s_g[(pe >> 3) * (unsigned)C + ch];
            }
            if (acc != 0.0f) atomic_add_f32(gv + (size_t)tag[slot] * C + ch, acc);
        }
        __syncthreads();
        for (unsigned u = threadIdx.x; u < nuniq; u += NT) { const unsigned slot = ulist[u]; tag[slot] = BV_EMPTY; cnt[slot] = 0u; }
        __syncthreads();
    }
}

// nearest interpolation (utils.py:193-204): out[q, c] = vol[idx(round(loc_q)), c]  [* (1 - oob) + oob * fill].
// tf.round has no gradient (d / d loc = 0); d / d vol is tf.gather's scatter-add of g[q, c] into the gathered element,
// masked by (1 - oob) when a fill value is set.  One thread per output element, float atomics.
template <int D, int MODE>
__global__ __launch_bounds__(256) void interpn_nearest_bwd(InterpBwdArgs ba) {
    const InterpArgs &a = ba.f;
    const int b = blockIdx.y;
    const float *locb = a.loc ? a.loc + (long long)b * a.loc_bs : nullptr;
    const float *go = ba.gout + (long long)b * a.out_bs;
    float *gv = ba.gvol + (long long)b * a.vol_bs;
    const unsigned long long total = (unsigned long long)a.nout * (unsigned)a.C;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x; e < total;
         e += (unsigned long long)gridDim.x * 256u) {
        const unsigned q = (unsigned)(e / (unsigned)a.C);
        const int c = (int)(e - (unsigned long long)q * (unsigned)a.C);
        int qd[NRT_MAXD];
        float p[NRT_MAXD];
        decode<D>(a, q, qd);
        load_loc<D, MODE>(a, locb, q, qd, p);
        if (a.has_fill && out_of_bounds<D>(a, p)) continue;
        long long idx = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) idx = idx * a.S[d] + nearest_1d(p[d], a.S[d]);
        atomic_add_f32(&gv[idx * a.C + c], go[e]);
    }
}

}  // namespace

// NRT_BWD_WC=0: d loc by the register-pipelined kernel (warp_dice_bwd_xm) instead of the wave-cache gather (A/B runs, tests)
static bool nrt_bwd_wc() {
    const char *e = getenv("NRT_BWD_WC");         // read per call: the tests run both kernels in one process
    return !(e && e[0] == '0');
}

extern "C" int nrt_interpn_bwd_f32(const float *vol, const float *loc, const float *grad_out, float *grad_vol,
                                   float *grad_loc, int ndim, const int *vol_shape, const int *out_shape, int channels,
                                   int batch, long long vol_batch_stride, long long loc_batch_stride, int loc_mode,
                                   int has_fill, void *stream) {
    if (!grad_out || (!grad_vol && !grad_loc)) return NRT_ERR_INVALID_ARG;
    if (loc_mode == NRT_LOC_LINSPACE && grad_loc) return NRT_ERR_INVALID_ARG;
    InterpBwdArgs ba;
    float dummy;
    int rc = fill_args(ba.f, vol, loc, &dummy, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    ba.f.out = nullptr;
    ba.gout = grad_out; ba.gvol = grad_vol; ba.gloc = grad_loc;
    ba.tg.x_march = 0;
    if (ba.f.nout == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    const int G = channels / 4;
    const bool vec = ndim == 3 && channels % 4 == 0 && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64) &&
                     ((((uintptr_t)vol | (uintptr_t)grad_out) & 15) == 0) && (vol_batch_stride % 4 == 0);
#define NRT_BWD_MODE(KERNEL, ...)                                                                              \
    switch (loc_mode) {                                                                                          \
        case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, NRT_LOC_ABSOLUTE>), grid, dim3(256), 0, st, ba); break; \
        case NRT_LOC_SHIFT: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, NRT_LOC_SHIFT>), grid, dim3(256), 0, st, ba); break;       \
        default: hipLaunchKernelGGL((KERNEL<__VA_ARGS__, NRT_LOC_LINSPACE>), grid, dim3(256), 0, st, ba); break;               \
    }
    if (vec) {
        const unsigned ng = 256 / G;
        unsigned blocks = (ba.f.nout + ng - 1) / ng;
        if (blocks > 256u * 16u) blocks = 256u * 16u;
        dim3 grid(blocks, batch);
        if (G == 8 && xmarch_applies(out_shape, batch)) {         // same block schedule as the fused forward (interpn_core.h)
            unsigned nt;
            const int t = xmarch_default_tune();
            tile_geometry(out_shape, G, t, t, ba.tg, nt);
            const unsigned per_batch = xmarch_setup(out_shape, batch, t, ba.tg);
            grid = dim3(nrt_xcd_grid(per_batch * (unsigned)batch), 1);
            { const char *e = getenv("NRT_BWD_SYNC"); ba.tg.depth_sync = e ? atoi(e) : 8; }
            const char *dedup_env = getenv("NRT_BWD_VOL_DEDUP");          // read per call: tests switch it
            const int use_dedup = dedup_env ? atoi(dedup_env) : 2;       // 0 plain scatter, 1 LDS accumulator table, 2 counting-sort merge
            unsigned long long rows = 1;
            for (int d = 0; d < 3; ++d) rows *= (unsigned long long)vol_shape[d];
            if (grad_vol && use_dedup == 2 && rows < 0xffffffffull) {
                // d vol by the sort-merge kernel (duplicate rows merged before L2), d loc by the rows kernel
                InterpBwdArgs bv = ba;
                bv.gloc = nullptr;
                switch (loc_mode) {
                    case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((interpn_bwd_vol_sort<NRT_LOC_ABSOLUTE>), grid, dim3(BS_NT), 0, st, bv); break;
                    case NRT_LOC_SHIFT: hipLaunchKernelGGL((interpn_bwd_vol_sort<NRT_LOC_SHIFT>), grid, dim3(BS_NT), 0, st, bv); break;
                    default: hipLaunchKernelGGL((interpn_bwd_vol_sort<NRT_LOC_LINSPACE>), grid, dim3(BS_NT), 0, st, bv); break;
                }
                NRT_CHECK_LAUNCH();
                if (!grad_loc) return NRT_OK;
                ba.gvol = nullptr;
            } else if (grad_vol && use_dedup && rows < 0xffffffffull) {
                // d vol through the LDS row-accumulator table (duplicate rows merged before L2), d loc by the rows kernel
                const size_t dyn = (size_t)BV_SLOTS * 32 * 4 + BV_SLOTS * 4 + (size_t)BV_NG * BV_U * 32 * 4 + 3 * (size_t)BV_NG * BV_U * 8 * 4;
                InterpBwdArgs bv = ba;
                bv.gloc = nullptr;
#define NRT_BV(MODE)                                                                                                         \
    do {                                                                                                                     \
        (void)hipFuncSetAttribute((const void *)interpn_bwd_vol_dedup<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
        hipLaunchKernelGGL((interpn_bwd_vol_dedup<MODE>), grid, dim3(256), dyn, st, bv);                                     \
    } while (0)
                switch (loc_mode) {
                    case NRT_LOC_ABSOLUTE: NRT_BV(NRT_LOC_ABSOLUTE); break;
                    case NRT_LOC_SHIFT: NRT_BV(NRT_LOC_SHIFT); break;
                    default: NRT_BV(NRT_LOC_LINSPACE); break;
                }
#undef NRT_BV
                NRT_CHECK_LAUNCH();
                if (!grad_loc) return NRT_OK;
                ba.gvol = nullptr;
            }
        }
        if (G == 8 && ba.tg.x_march && !ba.gvol && ba.gloc && loc_mode != NRT_LOC_LINSPACE &&
            (unsigned long long)vol_batch_stride * 4ull < (1ull << 32) && (unsigned long long)ba.f.nout * channels * 4ull < (1ull << 32)) {
            // d loc alone at 32 channels: the wave-cache gather (fused_wc.h, round 5) where it applies,
            if (nrt_bwd_wc() && (((uintptr_t)vol | (uintptr_t)grad_out) & 15) == 0 && nrt_wc_interpn_supported(&ba.f, batch))
                return nrt_wc_bwd_launch(&ba.f, batch, loc_mode, grad_out, nullptr, nullptr, 0.0f, ba.gloc, stream);
            // else the software-pipelined x-march kernel (warp_dice_bwd_xm with grad_out in place of d dice / d warped)
            if (loc_mode == NRT_LOC_SHIFT)
                hipLaunchKernelGGL((warp_dice_bwd_xm<NRT_LOC_SHIFT, false>), grid, dim3(256), 0, st, ba, grad_out, (const float *)nullptr, (const float *)nullptr, 0.0f);
            else
                hipLaunchKernelGGL((warp_dice_bwd_xm<NRT_LOC_ABSOLUTE, false>), grid, dim3(256), 0, st, ba, grad_out, (const float *)nullptr, (const float *)nullptr, 0.0f);
            NRT_CHECK_LAUNCH();
            return NRT_OK;
        }
        switch (G) {
            case 1: NRT_BWD_MODE(interpn_bwd_rows, 1) break;
            case 2: NRT_BWD_MODE(interpn_bwd_rows, 2) break;
            case 4: NRT_BWD_MODE(interpn_bwd_rows, 4) break;
            case 8: NRT_BWD_MODE(interpn_bwd_rows, 8) break;
            case 16: NRT_BWD_MODE(interpn_bwd_rows, 16) break;
            case 32: NRT_BWD_MODE(interpn_bwd_rows, 32) break;
            default: NRT_BWD_MODE(interpn_bwd_rows, 64) break;
        }
    } else {
        unsigned long long grows = 1;
        for (int d = 0; d < ndim; ++d) grows *= (unsigned long long)vol_shape[d];
        const char *sa = getenv("NRT_BWD_VOL_SORT_ANY");               // 0: the per-element scatter below for every channel count
        if (grad_vol && ndim == 3 && channels <= BG_CMAX && grows < 0xffffffffull && !(sa && sa[0] == '0')) {
            // few channels, 3-D: duplicate rows merged on chip (counting sort), d loc by the per-voxel kernel
            const unsigned nTx = (unsigned)(out_shape[0] + 3) / 4, nTy = (unsigned)(out_shape[1] + 3) / 4, nTz = (unsigned)(out_shape[2] + 7) / 8;
            const unsigned ntiles = nTx * nTy * nTz;
            unsigned gx = ntiles < 256u * 8u ? ntiles : 256u * 8u;
            if (gx < 1u) gx = 1u;
            dim3 sgrid(gx, batch);
            InterpBwdArgs bv = ba;
            bv.gloc = nullptr;
            switch (loc_mode) {
                case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((interpn_bwd_vol_sort_any<NRT_LOC_ABSOLUTE>), sgrid, dim3(256), 0, st, bv, nTy, nTz, ntiles); break;
                case NRT_LOC_SHIFT: hipLaunchKernelGGL((interpn_bwd_vol_sort_any<NRT_LOC_SHIFT>), sgrid, dim3(256), 0, st, bv, nTy, nTz, ntiles); break;
                default: hipLaunchKernelGGL((interpn_bwd_vol_sort_any<NRT_LOC_LINSPACE>), sgrid, dim3(256), 0, st, bv, nTy, nTz, ntiles); break;
            }
            NRT_CHECK_LAUNCH();
            if (!grad_loc) return NRT_OK;
            ba.gvol = nullptr;
        } else if (grad_vol && channels > 1) {
            // d vol per element (coalesced atomic requests), d loc per voxel
            unsigned long long nb = ((unsigned long long)ba.f.nout * (unsigned)channels + 255) / 256;
            if (nb > 65536ull * 4) nb = 65536ull * 4;
            dim3 grid((unsigned)nb, batch);
            switch (ndim) {
                case 1: NRT_BWD_MODE(interpn_bwd_vol_elems, 1) break;
                case 2: NRT_BWD_MODE(interpn_bwd_vol_elems, 2) break;
                default: NRT_BWD_MODE(interpn_bwd_vol_elems, 3) break;
            }
            NRT_CHECK_LAUNCH();
            if (!grad_loc) return NRT_OK;
            ba.gvol = nullptr;
        }
        unsigned blocks = (ba.f.nout + 255) / 256;
        if (blocks > 256u * 16u) blocks = 256u * 16u;
        dim3 grid(blocks, batch);
        switch (ndim) {
            case 1: NRT_BWD_MODE(interpn_bwd_generic, 1) break;
            case 2: NRT_BWD_MODE(interpn_bwd_generic, 2) break;
            default: NRT_BWD_MODE(interpn_bwd_generic, 3) break;
        }
    }
#undef NRT_BWD_MODE
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_interpn_nearest_bwd_f32(const float *loc, const float *grad_out, float *grad_vol, int ndim, const int *vol_shape,
                                           const int *out_shape, int channels, int batch, long long vol_batch_stride,
                                           long long loc_batch_stride, int loc_mode, int has_fill, void *stream) {
    if (!grad_out || !grad_vol) return NRT_ERR_INVALID_ARG;
    InterpBwdArgs ba;
    float dummy;
    int rc = fill_args(ba.f, grad_vol, loc, &dummy, ndim, vol_shape, out_shape, channels, batch, vol_batch_stride,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    ba.f.out = nullptr;
    ba.gout = grad_out; ba.gvol = grad_vol; ba.gloc = nullptr;
    ba.tg.x_march = 0;
    if (ba.f.nout == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    unsigned long long nb = ((unsigned long long)ba.f.nout * (unsigned)channels + 255) / 256;
    if (nb > 256ull * 64) nb = 256ull * 64;
    dim3 grid((unsigned)nb, batch);
#define NRT_NB(DD)                                                                                                              \
    switch (loc_mode) {                                                                                                         \
        case NRT_LOC_ABSOLUTE: hipLaunchKernelGGL((interpn_nearest_bwd<DD, NRT_LOC_ABSOLUTE>), grid, dim3(256), 0, st, ba); break; \
        case NRT_LOC_SHIFT: hipLaunchKernelGGL((interpn_nearest_bwd<DD, NRT_LOC_SHIFT>), grid, dim3(256), 0, st, ba); break;       \
        default: hipLaunchKernelGGL((interpn_nearest_bwd<DD, NRT_LOC_LINSPACE>), grid, dim3(256), 0, st, ba); break;               \
    }
    switch (ndim) {
        case 1: NRT_NB(1) break;
        case 2: NRT_NB(2) break;
        default: NRT_NB(3) break;
    }
#undef NRT_NB
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_dice_soft_bwd_f32(const float *y_true, const float *y_pred, const float *sums, const float *grad_dice,
                                     long long nvox, int nlabels, int batch, float laplace_smoothing, float *grad_pred,
                                     float *grad_true, void *stream) {
    if (!y_true || !y_pred || !sums || !grad_dice || (!grad_pred && !grad_true)) return NRT_ERR_INVALID_ARG;
    if (nvox < 0 || nlabels < 1 || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    const long long n = nvox * nlabels;
    const int G4 = nlabels / 4;
    const uintptr_t al = (uintptr_t)y_true | (uintptr_t)y_pred | (uintptr_t)grad_pred | (uintptr_t)grad_true;
    if (nlabels % 4 == 0 && 256 % G4 == 0 && (al & 15) == 0) {
        unsigned blocks = (unsigned)((n / 4 + 255) / 256);
        if (blocks > 256u * 8u) blocks = 256u * 8u;
        hipLaunchKernelGGL(dice_soft_bwd_vec, dim3(blocks, batch), dim3(256), 0, nrt_stream(stream), (const nrt_f4 *)y_true,
                           (const nrt_f4 *)y_pred, sums, grad_dice, nvox, nlabels, laplace_smoothing, (nrt_f4 *)grad_pred,
                           (nrt_f4 *)grad_true);
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    hipLaunchKernelGGL(dice_soft_bwd, dim3(blocks, batch), dim3(256), 0, nrt_stream(stream), y_true, y_pred, sums, grad_dice,
                       nvox, nlabels, laplace_smoothing, grad_pred, grad_true);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_dice_soft_bwd_norm_f32(const float *y_true, const float *y_pred, const float *sums, const float *grad_dice,
                                          long long nvox, int nlabels, int batch, float laplace_smoothing, float *grad_pred,
                                          float *grad_true, void *stream) {
    if (!y_true || !y_pred || !sums || !grad_dice || (!grad_pred && !grad_true)) return NRT_ERR_INVALID_ARG;
    if (nvox < 0 || nlabels < 1 || nlabels > 4096 || batch < 1 || batch > 65535) return NRT_ERR_INVALID_ARG;
    if (nvox == 0) return NRT_OK;
    unsigned blocks = (unsigned)((nvox + 255) / 256);
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    hipLaunchKernelGGL(dice_soft_bwd_norm, dim3(blocks, batch), dim3(256), (size_t)2 * nlabels * sizeof(float), nrt_stream(stream),
                       y_true, y_pred, sums, grad_dice, nvox, nlabels, laplace_smoothing, grad_pred, grad_true);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_wcce_bwd_f32(const float *y_true, const float *y_pred, const float *label_weights,
                                const float *grad_scalar, const float *grad_per_voxel, long long nvox_total, int channels,
                                int from_logits, float label_smoothing, float scale, float *grad_pred, void *stream) {
    if (!y_true || !y_pred || !grad_pred || (!grad_scalar && !grad_per_voxel)) return NRT_ERR_INVALID_ARG;
    if (nvox_total < 0 || channels < 1) return NRT_ERR_INVALID_ARG;
    if (nvox_total == 0) return NRT_OK;
    const int G = channels / 4;
    const uintptr_t al = (uintptr_t)y_true | (uintptr_t)y_pred | (uintptr_t)grad_pred;
    if (channels % 4 == 0 && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16) && (al & 15) == 0) {
        const long long per_block = 256 / G;
        unsigned vb = (unsigned)((nvox_total + per_block - 1) / per_block);
        if (vb > 256u * 16u) vb = 256u * 16u;
        hipStream_t st = nrt_stream(stream);
#define NRT_CCE_BWD(GG)                                                                                              \
    if (from_logits)                                                                                                 \
        hipLaunchKernelGGL((wcce_bwd_vec<GG, 1>), dim3(vb), dim3(256), 0, st, (const nrt_f4 *)y_true,                \
                           (const nrt_f4 *)y_pred, label_weights, grad_scalar, grad_per_voxel, nvox_total,           \
                           label_smoothing, scale, (nrt_f4 *)grad_pred);                                             \
    else                                                                                                             \
        hipLaunchKernelGGL((wcce_bwd_vec<GG, 0>), dim3(vb), dim3(256), 0, st, (const nrt_f4 *)y_true,                \
                           (const nrt_f4 *)y_pred, label_weights, grad_scalar, grad_per_voxel, nvox_total,           \
                           label_smoothing, scale, (nrt_f4 *)grad_pred);
        switch (G) {
            case 1: NRT_CCE_BWD(1) break;
            case 2: NRT_CCE_BWD(2) break;
            case 4: NRT_CCE_BWD(4) break;
            case 8: NRT_CCE_BWD(8) break;
            default: NRT_CCE_BWD(16) break;
        }
#undef NRT_CCE_BWD
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    int VP = 256;                                                 // rows of VP voxels of both tensors in 48 KB of LDS
    while (VP > 8 && (size_t)2 * VP * channels * 4 > 48 * 1024) VP >>= 1;
    if ((size_t)2 * VP * channels * 4 > 48 * 1024) VP = 0;
    const int per = VP > 0 ? VP : 256;
    long long nb = (nvox_total + per - 1) / per;
    if (nb > 256ll * 16) nb = 256ll * 16;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(wcce_bwd, dim3((unsigned)nb), dim3(256), (size_t)2 * VP * channels * 4, nrt_stream(stream), y_true, y_pred,
                       label_weights, grad_scalar, grad_per_voxel, nvox_total, channels, from_logits, label_smoothing, scale, VP, grad_pred);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

extern "C" int nrt_warp_dice_bwd_f32(const float *moving, const float *loc, const float *fixed, const float *sums,
                                     const float *grad_dice, float *grad_loc, const int *vol_shape, const int *out_shape,
                                     int nlabels, int batch, long long loc_batch_stride, int loc_mode, int has_fill,
                                     float laplace_smoothing, void *stream) {
    if (!moving || !loc || !fixed || !sums || !grad_dice || !grad_loc) return NRT_ERR_INVALID_ARG;
    if (loc_mode != NRT_LOC_ABSOLUTE && loc_mode != NRT_LOC_SHIFT) return NRT_ERR_INVALID_ARG;
    const int G = nlabels / 4;
    if (nlabels % 4 || !(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64)) return NRT_ERR_UNSUPPORTED;
    if ((((uintptr_t)moving | (uintptr_t)fixed) & 15) != 0) return NRT_ERR_UNSUPPORTED;
    InterpBwdArgs ba;
    float dummy;
    long long nin = 1;
    for (int d = 0; d < 3; ++d) nin *= vol_shape[d];
    int rc = fill_args(ba.f, moving, loc, &dummy, 3, vol_shape, out_shape, nlabels, batch, nin * nlabels,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    ba.f.out = nullptr;
    ba.gout = nullptr; ba.gvol = nullptr; ba.gloc = grad_loc;
    ba.tg.x_march = 0;
    if (ba.f.nout == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    const unsigned ng = 256 / G;
    unsigned blocks = (ba.f.nout + ng - 1) / ng;
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    dim3 grid(blocks, batch);
    if (G == 8 && xmarch_applies(out_shape, batch)) {             // same block schedule as the fused forward (interpn_core.h)
        unsigned nt;
        const int t = xmarch_default_tune();
        tile_geometry(out_shape, G, t, t, ba.tg, nt);
        const unsigned per_batch = xmarch_setup(out_shape, batch, t, ba.tg);
        grid = dim3(nrt_xcd_grid(per_batch * (unsigned)batch), 1);
        { const char *e = getenv("NRT_BWD_SYNC"); ba.tg.depth_sync = e ? atoi(e) : 8; }
    }
#define NRT_WDB(GG)                                                                                              \
    if (loc_mode == NRT_LOC_SHIFT)                                                                               \
        hipLaunchKernelGGL((warp_dice_bwd_rows<GG, NRT_LOC_SHIFT>), grid, dim3(256), 0, st, ba, fixed, sums,     \
                           grad_dice, laplace_smoothing);                                                        \
    else                                                                                                         \
        hipLaunchKernelGGL((warp_dice_bwd_rows<GG, NRT_LOC_ABSOLUTE>), grid, dim3(256), 0, st, ba, fixed, sums,  \
                           grad_dice, laplace_smoothing);
    if (G == 8 && ba.tg.x_march && nrt_bwd_wc() && nrt_wc_interpn_supported(&ba.f, batch))
        return nrt_wc_bwd_launch(&ba.f, batch, loc_mode, fixed, sums, grad_dice, laplace_smoothing, grad_loc, stream);
    static int xm_pipe = -1;                       // NRT_BWD_XM=0: the un-pipelined kernel on the same schedule (A/B runs)
    if (xm_pipe < 0) { const char *e = getenv("NRT_BWD_XM"); xm_pipe = e ? atoi(e) : 1; }
    // the pipelined kernel forms 32-bit byte offsets of rows and locations
    const bool xm_fits = (unsigned long long)nin * nlabels * 4ull < (1ull << 32) &&
                         (unsigned long long)ba.f.nout * nlabels * 4ull < (1ull << 32);
    if (G == 8 && ba.tg.x_march && xm_pipe && xm_fits) {
        static int lds_kb = -1;                    // NRT_BWD_LDS_KB (experiments): unused dynamic LDS per block caps the blocks per CU
        if (lds_kb < 0) { const char *e = getenv("NRT_BWD_LDS_KB"); lds_kb = e ? atoi(e) : 0; }
        const unsigned dyn = (unsigned)lds_kb * 1024u;
        if (loc_mode == NRT_LOC_SHIFT) {
            if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)warp_dice_bwd_xm<NRT_LOC_SHIFT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            hipLaunchKernelGGL((warp_dice_bwd_xm<NRT_LOC_SHIFT>), grid, dim3(256), dyn, st, ba, fixed, sums, grad_dice, laplace_smoothing);
        } else {
            if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)warp_dice_bwd_xm<NRT_LOC_ABSOLUTE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            hipLaunchKernelGGL((warp_dice_bwd_xm<NRT_LOC_ABSOLUTE>), grid, dim3(256), dyn, st, ba, fixed, sums, grad_dice, laplace_smoothing);
        }
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    switch (G) {
        case 1: NRT_WDB(1) break;
        case 2: NRT_WDB(2) break;
        case 4: NRT_WDB(4) break;
        case 8: NRT_WDB(8) break;
        case 16: NRT_WDB(16) break;
        case 32: NRT_WDB(32) break;
        default: NRT_WDB(64) break;
    }
#undef NRT_WDB
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}
