// Lab: C entry points of the LDS-row-cache gather (gather_lc.hip) -- the same argument lists as nrt_warp_dice_soft_f32 /
// nrt_interpn_f32_ex of the product ABI, so that tools/lab/lab.py and tools/lc_check.py can compare the two bit for bit.
// Not part of libneurite_amd.so.
#include "dice_reduce.h"
#include "interpn_core.h"
#include "lc.h"

namespace {
size_t lab_ws_bytes(unsigned nblocks, int L, int batch) {
    const size_t rows = (size_t)batch * nblocks;
    const size_t grp = (size_t)batch * ((nblocks + RED_ROWS - 1) / RED_ROWS);
    return rows * 3 * L * sizeof(float) + rows * 4 * sizeof(float) + 16 + grp * 3 * L * sizeof(double) + grp * 4 * sizeof(float) + 256;
}
}  // namespace

extern "C" size_t nrt_lab_lc_workspace_bytes(const int *out_shape, int nlabels, int batch, int tune) {
    if (!out_shape || nlabels != 32 || batch < 1) return 0;
    return lab_ws_bytes(nrt_lc_rows(out_shape, batch, tune), nlabels, batch);
}

extern "C" int nrt_lab_lc_warp_dice_f32(const float *moving, const float *loc, const float *fixed, float *warped, const int *vol_shape,
                                        const int *out_shape, int nlabels, int batch, long long loc_batch_stride, int loc_mode, int has_fill,
                                        float fill_value, float laplace_smoothing, float *sums, float *dice, float *minmax, int tune,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    if (!fixed || !sums || !dice || nlabels != 32) return NRT_ERR_INVALID_ARG;
    InterpArgs a;
    long long vol_bs = nlabels;
    for (int d = 0; d < 3; ++d) vol_bs *= vol_shape[d];
    float dummy;
    int rc = fill_args(a, moving, loc, warped ? (void *)warped : (void *)&dummy, 3, vol_shape, out_shape, nlabels, batch, vol_bs,
                       loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    if ((unsigned long long)vol_bs * 4ull >= (1ull << 32) || (unsigned long long)a.nout * 128ull >= (1ull << 32)) return NRT_ERR_UNSUPPORTED;
    if ((((uintptr_t)moving | (uintptr_t)fixed | (uintptr_t)warped) & 15) != 0 || !nrt_lc_supported(vol_shape, out_shape, nlabels)) return NRT_ERR_UNSUPPORTED;
    const unsigned nrows = nrt_lc_rows(out_shape, batch, tune);
    if (!workspace || workspace_bytes < lab_ws_bytes(nrows, nlabels, batch)) return NRT_ERR_WORKSPACE;
    DiceWs w;
    const size_t rows = (size_t)batch * nrows;
    char *p = (char *)workspace;
    w.fpart = (float *)p; p += rows * 3 * nlabels * sizeof(float);
    w.mpart = (float *)p; p += rows * 4 * sizeof(float);
    p = (char *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    w.gsum = (double *)p; p += (size_t)batch * ((nrows + RED_ROWS - 1) / RED_ROWS) * 3 * nlabels * sizeof(double);
    w.gmm = (float *)p;
    w.ipart = nullptr;
    LcCall c;
    c.vol = moving; c.loc = loc; c.out = warped; c.fixed = fixed; c.fpart = w.fpart; c.mpart = w.mpart; c.minmax = minmax != nullptr;
    for (int d = 0; d < 3; ++d) { c.S[d] = a.S[d]; c.O[d] = a.O[d]; c.delta[d] = a.delta[d]; }
    c.batch = batch; c.vol_bs = a.vol_bs; c.loc_bs = a.loc_bs; c.out_bs = a.out_bs;
    c.mode = loc_mode; c.has_fill = a.has_fill; c.fill = fill_value; c.tune = tune;
    hipStream_t st = nrt_stream(stream);
    rc = nrt_lc_launch(c, st);
    if (rc != NRT_OK) return rc;
    return dice_finalize_soft(w, nrows, 1, batch, nlabels, laplace_smoothing, sums, dice, minmax, st);
}

extern "C" int nrt_lab_lc_interpn_f32(const float *vol, const float *loc, float *out, const int *vol_shape, const int *out_shape, int batch,
                                      long long vol_batch_stride, long long loc_batch_stride, int loc_mode, int has_fill, float fill_value,
                                      int tune, void *stream) {
    InterpArgs a;
    int rc = fill_args(a, vol, loc, out, 3, vol_shape, out_shape, 32, batch, vol_batch_stride, loc_batch_stride, loc_mode, has_fill);
    if (rc != NRT_OK) return rc;
    if (a.nout == 0) return NRT_OK;
    unsigned long long vol_bytes = 128ull;
    for (int d = 0; d < 3; ++d) vol_bytes *= (unsigned long long)vol_shape[d];
    if (vol_bytes >= (1ull << 32) || (unsigned long long)a.nout * 128ull >= (1ull << 32) || !nrt_lc_supported(a.S, a.O, 32) ||
        (((uintptr_t)vol | (uintptr_t)out) & 15) != 0 || ((uintptr_t)loc & 3) != 0 || (vol_batch_stride * 4) % 16 != 0)
        return NRT_ERR_UNSUPPORTED;
    LcCall w;
    w.vol = vol; w.loc = loc; w.out = out; w.fixed = nullptr; w.fpart = nullptr; w.mpart = nullptr; w.minmax = 0;
    for (int d = 0; d < 3; ++d) { w.S[d] = a.S[d]; w.O[d] = a.O[d]; w.delta[d] = a.delta[d]; }
    w.batch = batch; w.vol_bs = a.vol_bs; w.loc_bs = a.loc_bs; w.out_bs = a.out_bs;
    w.mode = loc_mode; w.has_fill = a.has_fill; w.fill = fill_value; w.tune = tune;
    return nrt_lc_launch(w, nrt_stream(stream));
}
