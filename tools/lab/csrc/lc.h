// Internal interface of the LDS-row-cache gather (gather_lc.hip), used by interpn.hip (drop-in interpn /
// SpatialTransformer / Resize, 32 float channels) and fused.hip (SpatialTransformer + Dice).  Not part of the C ABI.
#pragma once

#include "nrt_common.h"

struct LcCall {
    const float *vol;            // [batch][S0][S1][S2][32]
    const float *loc;            // absolute locations / shifts [batch][O0][O1][O2][3], null for linspace
    float *out;                  // [batch][O0][O1][O2][32] or null (fused Dice without the warped volume)
    const float *fixed;          // Dice: [batch][O0][O1][O2][32], else null
    float *fpart, *mpart;        // Dice: one partial row per task, [batch][rows][96] and [batch][rows][4]
    int minmax;                  // Dice: also track min/max of fixed and warped (range asserts of metrics.py:439-444)
    int S[3], O[3];
    float delta[3];
    int batch;
    long long vol_bs, loc_bs, out_bs;   // batch strides in elements
    int mode, has_fill;
    float fill;
    int tune;                    // bits 0-7: x segments (0 = auto); bits 8-9: diagnostic mode (0 = product)
};

// 32 float channels, 3-D, sizes the tag encoding and the 32-bit offsets can address
__attribute__((visibility("hidden"))) bool nrt_lc_supported(const int *S, const int *O, int channels);
// Dice partial rows per batch entry (= tasks per batch entry) for this geometry
__attribute__((visibility("hidden"))) unsigned nrt_lc_rows(const int *O, int batch, int tune);
__attribute__((visibility("hidden"))) int nrt_lc_launch(const LcCall &c, hipStream_t st);
