// Device helpers shared by the few-channel interpn kernels (interpn_lean.hip, interpn_lds2.hip).
#pragma once

#include "interpn_core.h"

namespace {

template <int C> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef nrt_f2 T; };
template <> struct Vec<3> { struct __attribute__((packed, aligned(4))) T { float v[3]; }; };
template <> struct Vec<4> { typedef nrt_f4 T; };

template <int C>
__device__ __forceinline__ void load_c(const char *base, unsigned off, float (&v)[C]) {
    if constexpr (C == 1) v[0] = *(const float *)(base + off);
    else if constexpr (C == 2) { const nrt_f2 t = *(const nrt_f2 *)(base + off); v[0] = t[0]; v[1] = t[1]; }
    else if constexpr (C == 3) {
        const typename Vec<3>::T t = *(const typename Vec<3>::T *)(base + off);
        v[0] = t.v[0]; v[1] = t.v[1]; v[2] = t.v[2];
    } else { const nrt_f4 t = *(const nrt_f4 *)(base + off); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
}

// utils.py:139-153 for one dimension; `up` = 1 when the upper corner is a different voxel (l1 != l0)
__device__ __forceinline__ void lean_corner(float p, float mx, int imax, int &i0, int &up, float &w0, float &w1) {
    const float f = floorf(p);                                           // :139
    const float cl = __builtin_amdgcn_fmed3f(p, 0.0f, mx);               // :142  clip = median(p, 0, max)
    const float l0 = __builtin_amdgcn_fmed3f(f, 0.0f, mx);               // :143
    const float l1 = fminf(nrt_add(l0, 1.0f), mx);                       // :146  (l0 + 1 >= 1: the lower clip never binds)
    i0 = min(max((int)l0, 0), imax);                                     // :147; the integer clamp only acts on NaN locations
    up = (l1 > l0) ? 1 : 0;
    w0 = nrt_sub(l1, cl);                                                // :152
    w1 = nrt_sub(1.0f, w0);                                              // :153
}

}  // namespace
