// interpn for the volume dtypes and ranks the float32 1-3-D kernels of interpn.hip do not take:
// float16 / bfloat16 / float64 volumes in 1..8 dimensions, and float32 / int32 volumes in 4..8 dimensions.
//
// neurite/tf/utils/utils.py:73-220 is dtype- and rank-generic: `loc` is cast to the volume's float dtype (:123-127) and
// every operation of the linear branch (:137-191) -- floor, the three clips, the weight differences, prod_n, the
// weighted accumulation over the 2^D corners in itertools.product order -- then runs in THAT dtype, one rounding per
// operation.  TensorFlow evaluates half and bfloat16 element-wise ops by computing in float and rounding the result to
// the storage type (Eigen::half / Eigen::bfloat16); for +, -, * that equals correctly rounded arithmetic of the narrow type
// (24 >= 2 p + 2 bits).  Num<T> below reproduces exactly that: operands and results of every op are values of T.
//
// Where the location comes from (the float32 conventions of interpn_core.h): ABSOLUTE reads it (float32; float64 for float64
// volumes), SHIFT forms float32(index) + shift in float32 (vxm transform()), LINSPACE is tf.linspace in float32; the result
// is then cast to T, as interpn does with whatever it is handed.
//
// 4..8 dimensions: one thread per output ELEMENT (voxel, channel), channel fastest (interpn_any); 1..3 dimensions: the rank is a
// template parameter and a thread owns a group of channels (interpn_any_nd).  This is the coverage path -- the bandwidth-tuned
// kernels are the float32 ones.

#include "nrt_common.h"

namespace {

constexpr int ANY_MAXD = 8;          // (TensorFlow itself stops at rank-8 tensors: 7 spatial dimensions + channels is what the reference can be fed)

struct AnyArgs {
    const void *vol;
    const void *loc;
    void *out;
    int D, C;
    int S[ANY_MAXD], O[ANY_MAXD];
    long long vol_bs, loc_bs, out_bs;      // batch strides in elements
    float delta[ANY_MAXD];
    unsigned long long nelem;              // prod(O) * C
    int mode, has_fill, loc_f64;
    double fill;
};

// ---- arithmetic of the storage type ------------------------------------------------------------------------------
template <typename T> struct Num;

template <> struct Num<float> {
    typedef float S;                       // storage
    static __device__ __forceinline__ float from_f(float v) { return v; }
    static __device__ __forceinline__ float from_d(double v) { return (float)v; }
    static __device__ __forceinline__ float from_i(int v) { return (float)v; }
    static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
    static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float floor_(float a) { return floorf(a); }
    static __device__ __forceinline__ float rint_(float a) { return rintf(a); }
    static __device__ __forceinline__ float clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
    static __device__ __forceinline__ int to_i(float a) { return (int)a; }
    static __device__ __forceinline__ bool lt(float a, float b) { return a < b; }
    static __device__ __forceinline__ bool gt(float a, float b) { return a > b; }
};

template <> struct Num<double> {
    typedef double S;
    static __device__ __forceinline__ double from_f(float v) { return (double)v; }
    static __device__ __forceinline__ double from_d(double v) { return v; }
    static __device__ __forceinline__ double from_i(int v) { return (double)v; }
    static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double sub(double a, double b) { return __dsub_rn(a, b); }
    static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double floor_(double a) { return floor(a); }
    static __device__ __forceinline__ double rint_(double a) { return rint(a); }
    static __device__ __forceinline__ double clip(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
    static __device__ __forceinline__ int to_i(double a) { return (int)a; }
    static __device__ __forceinline__ bool lt(double a, double b) { return a < b; }
    static __device__ __forceinline__ bool gt(double a, double b) { return a > b; }
};

// IEEE binary16: the value lives in a float that is always exactly representable in half; every op rounds its float result
// to half (round-to-nearest-even, v_cvt_f16_f32) and widens again.
struct HalfTag {};
template <> struct Num<HalfTag> {
    typedef _Float16 S;
    static __device__ __forceinline__ float r(float v) { return (float)(_Float16)v; }
    static __device__ __forceinline__ float from_f(float v) { return r(v); }
    static __device__ __forceinline__ float from_d(double v) { return (float)(_Float16)v; }
    static __device__ __forceinline__ float from_i(int v) { return r((float)v); }      // exact int -> float below 2^24, then RNE
    static __device__ __forceinline__ float add(float a, float b) { return r(__fadd_rn(a, b)); }
    static __device__ __forceinline__ float sub(float a, float b) { return r(__fsub_rn(a, b)); }
    static __device__ __forceinline__ float mul(float a, float b) { return r(__fmul_rn(a, b)); }
    static __device__ __forceinline__ float floor_(float a) { return floorf(a); }
    static __device__ __forceinline__ float rint_(float a) { return rintf(a); }
    static __device__ __forceinline__ float clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
    static __device__ __forceinline__ int to_i(float a) { return (int)a; }
    static __device__ __forceinline__ bool lt(float a, float b) { return a < b; }
    static __device__ __forceinline__ bool gt(float a, float b) { return a > b; }
};

// bfloat16 = the upper 16 bits of a float32, round-to-nearest-even on the dropped half (NaN stays NaN)
struct Bf16Tag {};
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN -> quiet NaN); the five-instruction integer
// sequence it replaces made a bfloat16 warp 2.5x slower than a float16 one (every operation of the path rounds)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
template <> struct Num<Bf16Tag> {
    typedef unsigned short S;
    static __device__ __forceinline__ float r(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
    static __device__ __forceinline__ float from_f(float v) { return r(v); }
    static __device__ __forceinline__ float from_d(double v) { return r((float)v); }
    static __device__ __forceinline__ float from_i(int v) { return r((float)v); }
    static __device__ __forceinline__ float add(float a, float b) { return r(__fadd_rn(a, b)); }
    static __device__ __forceinline__ float sub(float a, float b) { return r(__fsub_rn(a, b)); }
    static __device__ __forceinline__ float mul(float a, float b) { return r(__fmul_rn(a, b)); }
    static __device__ __forceinline__ float floor_(float a) { return floorf(a); }
    static __device__ __forceinline__ float rint_(float a) { return rintf(a); }
    static __device__ __forceinline__ float clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
    static __device__ __forceinline__ int to_i(float a) { return (int)a; }
    static __device__ __forceinline__ bool lt(float a, float b) { return a < b; }
    static __device__ __forceinline__ bool gt(float a, float b) { return a > b; }
};

// working ("register") type and storage conversions
template <typename T> struct Work { typedef float W; };
template <> struct Work<double> { typedef double W; };

template <typename T> __device__ __forceinline__ typename Work<T>::W load_v(const void *p, long long i);
template <> __device__ __forceinline__ float load_v<float>(const void *p, long long i) { return ((const float *)p)[i]; }
template <> __device__ __forceinline__ double load_v<double>(const void *p, long long i) { return ((const double *)p)[i]; }
template <> __device__ __forceinline__ float load_v<HalfTag>(const void *p, long long i) { return (float)((const _Float16 *)p)[i]; }
template <> __device__ __forceinline__ float load_v<Bf16Tag>(const void *p, long long i) {
    return bf16_bits_to_f32(((const unsigned short *)p)[i]);
}
template <typename T> __device__ __forceinline__ void store_v(void *p, long long i, typename Work<T>::W v);
template <> __device__ __forceinline__ void store_v<float>(void *p, long long i, float v) { ((float *)p)[i] = v; }
template <> __device__ __forceinline__ void store_v<double>(void *p, long long i, double v) { ((double *)p)[i] = v; }
template <> __device__ __forceinline__ void store_v<HalfTag>(void *p, long long i, float v) { ((_Float16 *)p)[i] = (_Float16)v; }
template <> __device__ __forceinline__ void store_v<Bf16Tag>(void *p, long long i, float v) {
    ((unsigned short *)p)[i] = f32_to_bf16_bits(v);
}

// ---- sampling location of output voxel (coordinates qd), already cast to T ----------------------------------------
template <typename T>
__device__ __forceinline__ void any_loc(const AnyArgs &a, int b, unsigned long long q, const int *qd,
                                        typename Work<T>::W *p) {
    typedef Num<T> N;
    for (int d = 0; d < a.D; ++d) {
        if (a.mode == NRT_LOC_ABSOLUTE) {
            if (a.loc_f64) p[d] = N::from_d(((const double *)a.loc)[(long long)b * a.loc_bs + (long long)q * a.D + d]);
            else p[d] = N::from_f(((const float *)a.loc)[(long long)b * a.loc_bs + (long long)q * a.D + d]);
        } else if (a.mode == NRT_LOC_SHIFT) {
            p[d] = N::from_f(__fadd_rn((float)qd[d], ((const float *)a.loc)[(long long)b * a.loc_bs + (long long)q * a.D + d]));
        } else {
            const float v = (qd[d] == 0) ? 0.0f
                          : ((qd[d] == a.O[d] - 1) ? (float)(a.S[d] - 1) : __fmul_rn(a.delta[d], (float)qd[d]));
            p[d] = N::from_f(v);
        }
    }
}

template <typename T, bool NEAREST>
__global__ __launch_bounds__(256) void interpn_any(AnyArgs a) {
    typedef Num<T> N;
    typedef typename Work<T>::W W;
    const int b = blockIdx.y;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x; e < a.nelem;
         e += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long q = e / (unsigned)a.C;
        const int c = (int)(e % (unsigned)a.C);
        int qd[ANY_MAXD];
        {
            unsigned long long r = q;
            for (int d = a.D - 1; d > 0; --d) { qd[d] = (int)(r % (unsigned)a.O[d]); r /= (unsigned)a.O[d]; }
            qd[0] = (int)r;
        }
        W p[ANY_MAXD];
        any_loc<T>(a, b, q, qd, p);
        const void *volb = a.vol;
        const long long vbase = (long long)b * a.vol_bs;
        W res;
        if (NEAREST) {
            long long idx = 0;                                             // :196-203
            for (int d = 0; d < a.D; ++d) {
                const int r = nrt_clampi(N::to_i(N::rint_(p[d])), 0, a.S[d] - 1);
                idx = idx * a.S[d] + r;
            }
            res = load_v<T>(volb, vbase + idx * a.C + c);
        } else {
            int i0[ANY_MAXD], i1[ANY_MAXD];
            W w0[ANY_MAXD], w1[ANY_MAXD];
            for (int d = 0; d < a.D; ++d) {
                const W mx = N::from_i(a.S[d] - 1), zero = N::from_i(0), one = N::from_i(1);
                const W f = N::floor_(p[d]);                               // :139
                const W cl = N::clip(p[d], zero, mx);                      // :142
                const W l0 = N::clip(f, zero, mx);                         // :143
                const W l1 = N::clip(N::add(l0, one), zero, mx);           // :146
                i0[d] = nrt_clampi(N::to_i(l0), 0, a.S[d] - 1);            // :147 (the clamp only matters for NaN / rounding of mx)
                i1[d] = nrt_clampi(N::to_i(l1), 0, a.S[d] - 1);
                w0[d] = N::sub(l1, cl);                                    // :152 weight of the lower corner
                w1[d] = N::sub(one, w0[d]);                                // :153
            }
            W acc = N::from_i(0);                                          // :160
            const int ncorner = 1 << a.D;
            for (int k = 0; k < ncorner; ++k) {                            // itertools.product([0, 1], repeat=D): dim 0 slowest
                long long idx = 0;
                W wt = N::from_i(0);
                for (int d = 0; d < a.D; ++d) {
                    const int hi = (k >> (a.D - 1 - d)) & 1;
                    idx = idx * a.S[d] + (hi ? i1[d] : i0[d]);             // sub2ind2d, row-major
                    const W w = hi ? w1[d] : w0[d];
                    wt = d == 0 ? w : N::mul(wt, w);                       // prod_n, left to right
                }
                acc = N::add(acc, N::mul(wt, load_v<T>(volb, vbase + idx * a.C + c)));   // :191
            }
            res = acc;
        }
        if (a.has_fill) {                                                  // :206-213 on the un-clipped location
            bool oob = false;
            for (int d = 0; d < a.D; ++d) oob = oob || N::lt(p[d], N::from_i(0)) || N::gt(p[d], N::from_i(a.S[d] - 1));
            const W fv = N::from_d(a.fill);
            res = N::add(N::mul(res, N::from_i(oob ? 0 : 1)), N::mul(N::from_i(oob ? 1 : 0), fv));
        }
        store_v<T>(a.out, (long long)b * a.out_bs + (long long)e, res);
    }
}

// ---- 1..3 dimensions: the rank is a template parameter (no per-thread arrays in scratch memory, corner loop unrolled) and a thread
// owns CG consecutive channels of a voxel, so the location, the corner indices and the weight products are formed once per group
// instead of once per element; CG = 4 moves the group with one 8 / 16 / 32-byte access.  The arithmetic of an element is the
// sequence of interpn_any above, operation for operation.  (A bfloat16 warp of 4 x 160^3 x 32 took 15.3 ms on the per-element
// kernel against 1.3 ms for float32.)
template <typename T, int CG> struct Grp;                      // CG storage elements <-> CG working values
template <typename T> struct Grp<T, 1> {
    typedef typename Work<T>::W W;
    static __device__ __forceinline__ void load(const void *p, long long i, W (&v)[1]) { v[0] = load_v<T>(p, i); }
    static __device__ __forceinline__ void store(void *p, long long i, const W (&v)[1]) { store_v<T>(p, i, v[0]); }
};
template <> struct Grp<float, 4> {
    static __device__ __forceinline__ void load(const void *p, long long i, float (&v)[4]) {
        const nrt_f4 t = *(const nrt_f4 *)((const float *)p + i);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void store(void *p, long long i, const float (&v)[4]) {
        *(nrt_f4 *)((float *)p + i) = (nrt_f4){v[0], v[1], v[2], v[3]};
    }
};
template <> struct Grp<double, 4> {
    typedef double d2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void load(const void *p, long long i, double (&v)[4]) {
        const d2 t0 = *(const d2 *)((const double *)p + i), t1 = *(const d2 *)((const double *)p + i + 2);
        v[0] = t0[0]; v[1] = t0[1]; v[2] = t1[0]; v[3] = t1[1];
    }
    static __device__ __forceinline__ void store(void *p, long long i, const double (&v)[4]) {
        *(d2 *)((double *)p + i) = (d2){v[0], v[1]};
        *(d2 *)((double *)p + i + 2) = (d2){v[2], v[3]};
    }
};
typedef unsigned short any_u16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 any_h4 __attribute__((ext_vector_type(4)));
template <> struct Grp<HalfTag, 4> {
    static __device__ __forceinline__ void load(const void *p, long long i, float (&v)[4]) {
        const any_h4 t = *(const any_h4 *)((const _Float16 *)p + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)t[k];
    }
    static __device__ __forceinline__ void store(void *p, long long i, const float (&v)[4]) {
        *(any_h4 *)((_Float16 *)p + i) = (any_h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
};
template <> struct Grp<Bf16Tag, 4> {
    static __device__ __forceinline__ void load(const void *p, long long i, float (&v)[4]) {
        const any_u16x4 t = *(const any_u16x4 *)((const unsigned short *)p + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = bf16_bits_to_f32(t[k]);
    }
    static __device__ __forceinline__ void store(void *p, long long i, const float (&v)[4]) {
        *(any_u16x4 *)((unsigned short *)p + i) =
            (any_u16x4){f32_to_bf16_bits(v[0]), f32_to_bf16_bits(v[1]), f32_to_bf16_bits(v[2]), f32_to_bf16_bits(v[3])};
    }
};

template <typename T, bool NEAREST, int DT, int CG>
__global__ __launch_bounds__(256) void interpn_any_nd(AnyArgs a) {
    typedef Num<T> N;
    typedef typename Work<T>::W W;
    const int b = blockIdx.y;
    const unsigned ngrp = (unsigned)a.C / CG;
    const unsigned long long total = (a.nelem / (unsigned)a.C) * ngrp;
    const long long vbase = (long long)b * a.vol_bs;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long q = e / ngrp;
        const int c0 = (int)(e - q * ngrp) * CG;
        int qd[DT];
        {
            unsigned long long r = q;
#pragma unroll
            for (int d = DT - 1; d > 0; --d) { qd[d] = (int)(r % (unsigned)a.O[d]); r /= (unsigned)a.O[d]; }
            qd[0] = (int)r;
        }
        W p[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) {                                     // any_loc, rank known
            if (a.mode == NRT_LOC_ABSOLUTE) {
                if (a.loc_f64) p[d] = N::from_d(((const double *)a.loc)[(long long)b * a.loc_bs + (long long)q * DT + d]);
                else p[d] = N::from_f(((const float *)a.loc)[(long long)b * a.loc_bs + (long long)q * DT + d]);
            } else if (a.mode == NRT_LOC_SHIFT) {
                p[d] = N::from_f(__fadd_rn((float)qd[d], ((const float *)a.loc)[(long long)b * a.loc_bs + (long long)q * DT + d]));
            } else {
                const float v = (qd[d] == 0) ? 0.0f : ((qd[d] == a.O[d] - 1) ? (float)(a.S[d] - 1) : __fmul_rn(a.delta[d], (float)qd[d]));
                p[d] = N::from_f(v);
            }
        }
        W res[CG];
        if (NEAREST) {
            long long idx = 0;                                             // :196-203
#pragma unroll
            for (int d = 0; d < DT; ++d) idx = idx * a.S[d] + nrt_clampi(N::to_i(N::rint_(p[d])), 0, a.S[d] - 1);
            Grp<T, CG>::load(a.vol, vbase + idx * a.C + c0, res);
        } else {
            int i0[DT], i1[DT];
            W w0[DT], w1[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const W mx = N::from_i(a.S[d] - 1), zero = N::from_i(0), one = N::from_i(1);
                const W f = N::floor_(p[d]);                               // :139
                const W cl = N::clip(p[d], zero, mx);                      // :142
                const W l0 = N::clip(f, zero, mx);                         // :143
                const W l1 = N::clip(N::add(l0, one), zero, mx);           // :146
                i0[d] = nrt_clampi(N::to_i(l0), 0, a.S[d] - 1);            // :147
                i1[d] = nrt_clampi(N::to_i(l1), 0, a.S[d] - 1);
                w0[d] = N::sub(l1, cl);                                    // :152
                w1[d] = N::sub(one, w0[d]);                                // :153
            }
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) res[cc] = N::from_i(0);        // :160
#pragma unroll
            for (int k = 0; k < (1 << DT); ++k) {                          // itertools.product([0, 1], repeat=D): dim 0 slowest
                long long idx = 0;
                W wt = N::from_i(0);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int hi = (k >> (DT - 1 - d)) & 1;
                    idx = idx * a.S[d] + (hi ? i1[d] : i0[d]);             // sub2ind2d, row-major
                    const W w = hi ? w1[d] : w0[d];
                    wt = d == 0 ? w : N::mul(wt, w);                       // prod_n, left to right
                }
                W v[CG];
                Grp<T, CG>::load(a.vol, vbase + idx * a.C + c0, v);
#pragma unroll
                for (int cc = 0; cc < CG; ++cc) res[cc] = N::add(res[cc], N::mul(wt, v[cc]));       // :191
            }
        }
        if (a.has_fill) {                                                  // :206-213 on the un-clipped location
            bool oob = false;
#pragma unroll
            for (int d = 0; d < DT; ++d) oob = oob || N::lt(p[d], N::from_i(0)) || N::gt(p[d], N::from_i(a.S[d] - 1));
            const W fv = N::from_d(a.fill);
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) res[cc] = N::add(N::mul(res[cc], N::from_i(oob ? 0 : 1)), N::mul(N::from_i(oob ? 1 : 0), fv));
        }
        Grp<T, CG>::store(a.out, (long long)b * a.out_bs + (long long)q * a.C + c0, res);
    }
}

// int32 volumes, nearest only (4..8-D; the 1-3-D case lives in interpn.hip)
__global__ __launch_bounds__(256) void interpn_any_nearest_i32(AnyArgs a, int fill_i) {
    const int b = blockIdx.y;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x; e < a.nelem;
         e += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long q = e / (unsigned)a.C;
        const int c = (int)(e % (unsigned)a.C);
        int qd[ANY_MAXD];
        unsigned long long r = q;
        for (int d = a.D - 1; d > 0; --d) { qd[d] = (int)(r % (unsigned)a.O[d]); r /= (unsigned)a.O[d]; }
        qd[0] = (int)r;
        float p[ANY_MAXD];
        any_loc<float>(a, b, q, qd, p);
        long long idx = 0;
        bool oob = false;
        for (int d = 0; d < a.D; ++d) {
            idx = idx * a.S[d] + nrt_clampi((int)rintf(p[d]), 0, a.S[d] - 1);
            oob = oob || (p[d] < 0.0f) || (p[d] > (float)(a.S[d] - 1));
        }
        int v = ((const int *)a.vol)[(long long)b * a.vol_bs + idx * a.C + c];
        if (a.has_fill) v = v * (oob ? 0 : 1) + (oob ? 1 : 0) * fill_i;
        ((int *)a.out)[(long long)b * a.out_bs + (long long)e] = v;
    }
}

template <typename T, int DT, int CG>
void launch_any_nd(const AnyArgs &a, int batch, int method, hipStream_t st) {
    unsigned long long nb = (a.nelem / (unsigned)CG + 255) / 256;
    if (nb > 65536ull * 16) nb = 65536ull * 16;
    if (nb < 1) nb = 1;
    dim3 grid((unsigned)nb, (unsigned)batch), blk(256);
    if (method == NRT_INTERP_NEAREST) hipLaunchKernelGGL((interpn_any_nd<T, true, DT, CG>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((interpn_any_nd<T, false, DT, CG>), grid, blk, 0, st, a);
}

template <typename T>
int launch_any(const AnyArgs &a, int batch, int method, hipStream_t st) {
    if (a.D <= 3) {
        // groups of 4 channels need 4-element alignment of every row start (16 bytes for float32, 8 for the 16-bit types)
        const bool g4 = a.C % 4 == 0 && a.vol_bs % 4 == 0 && ((((uintptr_t)a.vol | (uintptr_t)a.out) & 31) == 0);
#define NRT_ANY_ND(DT)                                                           \
    do {                                                                         \
        if (g4) launch_any_nd<T, DT, 4>(a, batch, method, st);                   \
        else launch_any_nd<T, DT, 1>(a, batch, method, st);                      \
    } while (0)
        if (a.D == 1) NRT_ANY_ND(1);
        else if (a.D == 2) NRT_ANY_ND(2);
        else NRT_ANY_ND(3);
#undef NRT_ANY_ND
        NRT_CHECK_LAUNCH();
        return NRT_OK;
    }
    unsigned long long nb = (a.nelem + 255) / 256;
    if (nb > 65536ull * 16) nb = 65536ull * 16;
    dim3 grid((unsigned)nb, (unsigned)batch), blk(256);
    if (method == NRT_INTERP_NEAREST) hipLaunchKernelGGL((interpn_any<T, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((interpn_any<T, false>), grid, blk, 0, st, a);
    NRT_CHECK_LAUNCH();
    return NRT_OK;
}

}  // namespace

extern "C" int nrt_interpn_any(const void *vol, const void *loc, void *out, int dtype, int ndim, const int *vol_shape,
                               const int *out_shape, int channels, int batch, long long vol_batch_stride,
                               long long loc_batch_stride, int loc_mode, int loc_is_f64, int method, int has_fill,
                               double fill_value, void *stream) {
    if (!vol || !out || !vol_shape || !out_shape) return NRT_ERR_INVALID_ARG;
    if (ndim < 1 || ndim > ANY_MAXD || channels < 1 || batch < 1) return NRT_ERR_INVALID_ARG;
    if (loc_mode < 0 || loc_mode > 2 || (loc_mode != NRT_LOC_LINSPACE && !loc)) return NRT_ERR_INVALID_ARG;
    if (method != NRT_INTERP_LINEAR && method != NRT_INTERP_NEAREST) return NRT_ERR_INVALID_ARG;
    if (batch > 65535) return NRT_ERR_UNSUPPORTED;
    if (loc_is_f64 && !(dtype == NRT_DT_F64 && loc_mode == NRT_LOC_ABSOLUTE)) return NRT_ERR_INVALID_ARG;
    AnyArgs a;
    a.vol = vol; a.loc = loc; a.out = out; a.D = ndim; a.C = channels;
    unsigned long long nin = 1, nout = 1;
    for (int d = 0; d < ANY_MAXD; ++d) {
        a.S[d] = d < ndim ? vol_shape[d] : 1;
        a.O[d] = d < ndim ? out_shape[d] : 1;
        if (a.S[d] < 1 || a.O[d] < 0) return NRT_ERR_INVALID_ARG;
        nin *= (unsigned long long)a.S[d];
        nout *= (unsigned long long)a.O[d];
        a.delta[d] = a.O[d] > 1 ? (float)(a.S[d] - 1) / (float)(a.O[d] - 1) : 0.0f;     // tf.linspace step, float32
    }
    if (nin * (unsigned long long)channels >= (1ull << 40) || nout * (unsigned long long)channels >= (1ull << 40))
        return NRT_ERR_UNSUPPORTED;
    a.nelem = nout * (unsigned long long)channels;
    a.vol_bs = vol_batch_stride; a.loc_bs = loc_batch_stride; a.out_bs = (long long)a.nelem;
    a.mode = loc_mode; a.has_fill = has_fill ? 1 : 0; a.loc_f64 = loc_is_f64 ? 1 : 0; a.fill = fill_value;
    if (a.nelem == 0) return NRT_OK;
    hipStream_t st = nrt_stream(stream);
    switch (dtype) {
        case NRT_DT_F32: return launch_any<float>(a, batch, method, st);
        case NRT_DT_F64: return launch_any<double>(a, batch, method, st);
        case NRT_DT_F16: return launch_any<HalfTag>(a, batch, method, st);
        case NRT_DT_BF16: return launch_any<Bf16Tag>(a, batch, method, st);
        case NRT_DT_I32: {
            if (method != NRT_INTERP_NEAREST) return NRT_ERR_UNSUPPORTED;
            unsigned long long nb = (a.nelem + 255) / 256;
            if (nb > 65536ull * 16) nb = 65536ull * 16;
            hipLaunchKernelGGL(interpn_any_nearest_i32, dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, st, a, (int)fill_value);
            NRT_CHECK_LAUNCH();
            return NRT_OK;
        }
        default: return NRT_ERR_UNSUPPORTED;
    }
}
