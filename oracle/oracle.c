/*
 * oracle.c -- plain-C restatement of the neurite hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Used by tests/ as a fast
 * checker at full BASELINE sizes and by bench.py's cpu_baseline leg.  It is never
 * linked into or called from the product (neurite_amd/).
 *
 * Must be compiled with -ffp-contract=off (no FMA contraction): the float path
 * follows the reference's op sequence with one IEEE rounding per TF op, so that it
 * agrees bit-for-bit with oracle/np_oracle.py (checked in tests/test_oracle.py).
 *
 * Citations are to /root/reference/neurite/tf/... file:line.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXD 3

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static inline float clipf(float v, float lo, float hi) {
    /* tf.clip_by_value = min(max(v, lo), hi) */
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

/*
 * interpn, utils/utils.py:73-220.
 *   vol      [S0..S(D-1), C] row-major, channel fastest
 *   loc_mode 0: loc[q, d] absolute locations              (utils.py:106-108)
 *            1: loc[q, d] is a displacement; location = (float)q_d + loc[q, d]
 *               (voxelmorph transform(): identity 'ij' grid + shift)
 *            2: separable tables: loc = concat_d table_d[out_shape[d]]
 *               (resize(): ndgrid of tf.linspace vectors, utils.py:259-260)
 *   method   0 linear (:137-191), 1 nearest (:193-204)
 *   fill     applied when has_fill (:206-213)
 */
int orc_interpn_f32(const float *vol, int D, const int *vol_shape, int C,
                    const float *loc, int loc_mode, const int *out_shape,
                    int method, int has_fill, float fill, float *out) {
    if (D < 1 || D > ORC_MAXD) return -1;
    int64_t nout = 1;
    for (int d = 0; d < D; ++d) nout *= out_shape[d];
    int64_t tab_off[ORC_MAXD];
    {
        int64_t o = 0;
        for (int d = 0; d < D; ++d) { tab_off[d] = o; o += out_shape[d]; }
    }
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nout; ++q) {
        int qd[ORC_MAXD];
        {
            int64_t r = q;
            for (int d = D - 1; d >= 0; --d) { qd[d] = (int)(r % out_shape[d]); r /= out_shape[d]; }
        }
        float p[ORC_MAXD];
        for (int d = 0; d < D; ++d) {
            if (loc_mode == 0) p[d] = loc[q * D + d];
            else if (loc_mode == 1) p[d] = (float)qd[d] + loc[q * D + d];
            else p[d] = loc[tab_off[d] + qd[d]];
        }
        float *o = out + q * C;
        int oob = 0;
        if (has_fill) {
            for (int d = 0; d < D; ++d) {
                float mx = (float)(vol_shape[d] - 1);
                if (p[d] < 0.0f || p[d] > mx) oob = 1;          /* :209-211 */
            }
        }
        if (method == 0) {
            int i0[ORC_MAXD], i1[ORC_MAXD];
            float w[2][ORC_MAXD];
            for (int d = 0; d < D; ++d) {
                float mx = (float)(vol_shape[d] - 1);
                float f = floorf(p[d]);                          /* :139 */
                float cl = clipf(p[d], 0.0f, mx);                /* :142 */
                float l0 = clipf(f, 0.0f, mx);                   /* :143 */
                float l1 = clipf(l0 + 1.0f, 0.0f, mx);           /* :146 */
                i0[d] = (int)l0; i1[d] = (int)l1;                /* :147 */
                w[0][d] = l1 - cl;                               /* :152 */
                w[1][d] = 1.0f - w[0][d];                        /* :153 */
            }
            for (int c = 0; c < C; ++c) o[c] = 0.0f;             /* :160 */
            for (int corner = 0; corner < (1 << D); ++corner) {  /* itertools.product order :159 */
                int64_t idx = 0;
                float wt = 0.0f;
                for (int d = 0; d < D; ++d) {
                    int bit = (corner >> (D - 1 - d)) & 1;       /* first dim = most significant */
                    idx = idx * vol_shape[d] + (bit ? i1[d] : i0[d]);   /* :176, :1068-1082 */
                    wt = (d == 0) ? w[bit][d] : wt * w[bit][d];  /* :187, :1085-1092 */
                }
                const float *v = vol + idx * C;
                for (int c = 0; c < C; ++c) {
                    float prod = wt * v[c];
                    o[c] = o[c] + prod;                          /* :191 */
                }
            }
        } else {
            int64_t idx = 0;
            for (int d = 0; d < D; ++d) {
                int r = (int)rintf(p[d]);                        /* :196 round-half-even, trunc cast */
                int mx = vol_shape[d] - 1;
                r = r < 0 ? 0 : r; r = r > mx ? mx : r;          /* :197 */
                idx = idx * vol_shape[d] + r;
            }
            const float *v = vol + idx * C;
            for (int c = 0; c < C; ++c) o[c] = v[c];             /* :204 */
        }
        if (has_fill) {                                          /* :212-213 */
            float keep = oob ? 0.0f : 1.0f, take = oob ? 1.0f : 0.0f;
            for (int c = 0; c < C; ++c) {
                float a = o[c] * keep;
                float b = take * fill;
                o[c] = a + b;
            }
        }
    }
    return 0;
}

/*
 * Dice partial sums, metrics.py:471-477: for each (b, l)
 *   sums[b,0,l] = sum_v t*p ; sums[b,1,l] = sum_v t*t ; sums[b,2,l] = sum_v p*p    (float64)
 *   minmax = {min t, max t, min p, max p} over everything (for the :439-444 range asserts)
 */
int orc_dice_sums_f32(const float *t, const float *p, int B, int64_t V, int L,
                      double *sums, float *minmax) {
    float mn_t = INFINITY, mx_t = -INFINITY, mn_p = INFINITY, mx_p = -INFINITY;
    for (int b = 0; b < B; ++b) {
        double *s = sums + (int64_t)b * 3 * L;
        for (int i = 0; i < 3 * L; ++i) s[i] = 0.0;
        const float *tb = t + (int64_t)b * V * L, *pb = p + (int64_t)b * V * L;
        int nthr = orc_num_threads();
        double *part = (double *)calloc((size_t)nthr * 3 * L, sizeof(double));
        float *mm = (float *)malloc((size_t)nthr * 4 * sizeof(float));
        for (int i = 0; i < nthr; ++i) { mm[4*i] = INFINITY; mm[4*i+1] = -INFINITY; mm[4*i+2] = INFINITY; mm[4*i+3] = -INFINITY; }
#pragma omp parallel
        {
#ifdef _OPENMP
            int tid = omp_get_thread_num();
#else
            int tid = 0;
#endif
            double *ps = part + (size_t)tid * 3 * L;
            float *m = mm + 4 * tid;
#pragma omp for schedule(static)
            for (int64_t v = 0; v < V; ++v) {
                const float *tv = tb + v * L, *pv = pb + v * L;
                for (int l = 0; l < L; ++l) {
                    double a = tv[l], c = pv[l];
                    ps[l] += a * c; ps[L + l] += a * a; ps[2 * L + l] += c * c;
                    if (tv[l] < m[0]) m[0] = tv[l];
                    if (tv[l] > m[1]) m[1] = tv[l];
                    if (pv[l] < m[2]) m[2] = pv[l];
                    if (pv[l] > m[3]) m[3] = pv[l];
                }
            }
        }
        for (int i = 0; i < nthr; ++i) {
            for (int j = 0; j < 3 * L; ++j) s[j] += part[(size_t)i * 3 * L + j];
            if (mm[4*i] < mn_t) mn_t = mm[4*i];
            if (mm[4*i+1] > mx_t) mx_t = mm[4*i+1];
            if (mm[4*i+2] < mn_p) mn_p = mm[4*i+2];
            if (mm[4*i+3] > mx_p) mx_p = mm[4*i+3];
        }
        free(part); free(mm);
    }
    if (minmax) { minmax[0] = mn_t; minmax[1] = mx_t; minmax[2] = mn_p; minmax[3] = mx_p; }
    return 0;
}

/*
 * Hard Dice on probabilistic inputs, metrics.py:463-468: argmax (ties -> lowest index) then
 * one-hot; counts[b,0,l] = #(argmax t == l && argmax p == l), counts[b,1,l] = #(argmax t == l),
 * counts[b,2,l] = #(argmax p == l).
 */
int orc_dice_hard_counts_prob_f32(const float *t, const float *p, int B, int64_t V, int L,
                                  int64_t *counts) {
    memset(counts, 0, sizeof(int64_t) * (size_t)B * 3 * L);
    for (int b = 0; b < B; ++b) {
        int64_t *c = counts + (int64_t)b * 3 * L;
        const float *tb = t + (int64_t)b * V * L, *pb = p + (int64_t)b * V * L;
        for (int64_t v = 0; v < V; ++v) {
            const float *tv = tb + v * L, *pv = pb + v * L;
            int at = 0, ap = 0;
            for (int l = 1; l < L; ++l) { if (tv[l] > tv[at]) at = l; if (pv[l] > pv[ap]) ap = l; }
            c[L + at]++; c[2 * L + ap]++;
            if (at == ap) c[at]++;
        }
    }
    return 0;
}

/* Hard Dice on label maps (input_type='max_label'); out-of-range labels give an all-zero one-hot row. */
int orc_dice_hard_counts_label_i32(const int32_t *t, const int32_t *p, int B, int64_t V, int L,
                                   int64_t *counts) {
    memset(counts, 0, sizeof(int64_t) * (size_t)B * 3 * L);
    for (int b = 0; b < B; ++b) {
        int64_t *c = counts + (int64_t)b * 3 * L;
        for (int64_t v = 0; v < V; ++v) {
            int a = t[(int64_t)b * V + v], q = p[(int64_t)b * V + v];
            int oka = a >= 0 && a < L, okq = q >= 0 && q < L;
            if (oka) c[L + a]++;
            if (okq) c[2 * L + q]++;
            if (oka && okq && a == q) c[a]++;
        }
    }
    return 0;
}

/*
 * Label-weighted categorical cross-entropy, metrics.py:641-650 + Keras CCE (TF semantics):
 * t' = w*t; [smoothing]; q = p/sum(p); q = clip(q, 1e-7, 1-1e-7); l = -sum t' log q.
 * Returns sum over all N voxels of l in *loss_sum (float64); per_voxel (optional) gets l.
 */
int orc_wcce_f32(const float *t, const float *p, const float *w, int64_t N, int C,
                 int from_logits, double label_smoothing, double *loss_sum, double *per_voxel) {
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+:total)
    for (int64_t v = 0; v < N; ++v) {
        const float *tv = t + v * C, *pv = p + v * C;
        double l = 0.0;
        if (from_logits) {
            double mx = pv[0];
            for (int c = 1; c < C; ++c) if (pv[c] > mx) mx = pv[c];
            double se = 0.0;
            for (int c = 0; c < C; ++c) se += exp((double)pv[c] - mx);
            double lse = log(se);
            for (int c = 0; c < C; ++c) {
                double tt = (w ? (double)w[c] : 1.0) * (double)tv[c];
                if (label_smoothing != 0.0) tt = tt * (1.0 - label_smoothing) + label_smoothing / C;
                l -= tt * (((double)pv[c] - mx) - lse);
            }
        } else {
            double s = 0.0;
            for (int c = 0; c < C; ++c) s += pv[c];
            for (int c = 0; c < C; ++c) {
                double q = (double)pv[c] / s;
                q = q < 1e-7 ? 1e-7 : q; q = q > 1.0 - 1e-7 ? 1.0 - 1e-7 : q;
                double tt = (w ? (double)w[c] : 1.0) * (double)tv[c];
                if (label_smoothing != 0.0) tt = tt * (1.0 - label_smoothing) + label_smoothing / C;
                l -= tt * log(q);
            }
        }
        if (per_voxel) per_voxel[v] = l;
        total += l;
    }
    *loss_sum = total;
    return 0;
}

/*
 * Keras Conv3D (TF semantics, restated): cross-correlation, channels-last input [X,Y,Z,Cin],
 * kernel [kx,ky,kz,Cin,Cout], SAME padding with pad_before = floor(((k-1)*dil)/2), stride 1,
 * bias, optional ELU (x>0 ? x : exp(x)-1).  Accumulates in float64 ("truth"), output float32.
 * models.py:1378-1388, 1545-1555, 1596.
 */
int orc_conv3d_same_f32(const float *x, int X, int Y, int Z, int Cin,
                        const float *k, int kx, int ky, int kz, int Cout, int dil,
                        const float *bias, int act_elu, float *y) {
    int px = ((kx - 1) * dil) / 2, py = ((ky - 1) * dil) / 2, pz = ((kz - 1) * dil) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ox = 0; ox < X; ++ox)
        for (int oy = 0; oy < Y; ++oy) {
            double *acc = (double *)malloc(sizeof(double) * Cout);
            for (int oz = 0; oz < Z; ++oz) {
                for (int co = 0; co < Cout; ++co) acc[co] = bias ? (double)bias[co] : 0.0;
                for (int a = 0; a < kx; ++a) {
                    int ix = ox + a * dil - px; if (ix < 0 || ix >= X) continue;
                    for (int b = 0; b < ky; ++b) {
                        int iy = oy + b * dil - py; if (iy < 0 || iy >= Y) continue;
                        for (int c = 0; c < kz; ++c) {
                            int iz = oz + c * dil - pz; if (iz < 0 || iz >= Z) continue;
                            const float *xv = x + (((int64_t)ix * Y + iy) * Z + iz) * Cin;
                            const float *kv = k + ((((int64_t)a * ky + b) * kz + c) * Cin) * Cout;
                            for (int ci = 0; ci < Cin; ++ci) {
                                double xs = xv[ci];
                                const float *kk = kv + (int64_t)ci * Cout;
                                for (int co = 0; co < Cout; ++co) acc[co] += xs * (double)kk[co];
                            }
                        }
                    }
                }
                float *yo = y + (((int64_t)ox * Y + oy) * Z + oz) * Cout;
                for (int co = 0; co < Cout; ++co) {
                    double v = acc[co];
                    if (act_elu) v = v > 0.0 ? v : exp(v) - 1.0;
                    yo[co] = (float)v;
                }
            }
            free(acc);
        }
    return 0;
}
