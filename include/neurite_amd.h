/*
 * neurite_amd.h -- C ABI of libneurite_amd.so, the MI355X (gfx950 / CDNA4) implementation of
 * neurite's 3-D volume hot path.
 *
 * The reference (adalca/neurite) is pure Python on TensorFlow and has no FFI of its own
 * (SURVEY.md section 8b); its boundary is a set of Python call signatures.  Each entry point below
 * replaces the TensorFlow op sequence that one reference function issues, and is what a
 * `neurite/torch/` backend (the dormant switch at neurite/__init__.py:33-42) would bind.
 * INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *  - All tensor pointers are DEVICE pointers (HIP), row-major, channels-last, exactly the
 *    reference layout [B, *spatial, C]; int* shape arguments are HOST pointers.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream), allocates nothing and never synchronises.  Scratch memory is supplied by the
 *    caller: ask nrt_*_workspace_bytes() and pass a device buffer of at least that size.
 *  - Return value: NRT_OK or a negative nrt_status; nothing throws across the ABI.
 *    nrt_status_string() describes a code.
 *  - Inputs are never written.  Pointers must be 16-byte aligned (torch allocations are).
 */
#ifndef NEURITE_AMD_H
#define NEURITE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    NRT_OK = 0,
    NRT_ERR_INVALID_ARG = -1,   /* NULL pointer, bad rank, non-positive size ...            */
    NRT_ERR_UNSUPPORTED = -2,   /* combination not implemented by the HIP path              */
    NRT_ERR_LAUNCH = -3,        /* hipLaunchKernel / hipGetLastError reported a failure     */
    NRT_ERR_WORKSPACE = -4      /* workspace missing or too small                           */
} nrt_status;

const char *nrt_status_string(int status);
/* ABI version, bumped when a signature changes. */
int nrt_abi_version(void);
/* Identity of the sources the library was built from (16 hex digits of a sha256 over flags, headers and kernels; "unknown" for a
 * build outside neurite_amd/build.py).  The reference is pure Python and has no counterpart; __graft_entry__.build() uses it to
 * refuse a shipped binary that does not match the tree. */
const char *nrt_build_id(void);
/* Name of the GPU architecture the library was compiled for ("gfx950"). */
const char *nrt_target_arch(void);
/* Library state on the CURRENT device (no reference counterpart: TensorFlow owns its runtime state, neurite/tf has none).
 * Kernels that coordinate their blocks through atomic counters (the persistent gather's work lists, the one-launch weighted
 * cross-entropy) keep them in a device-resident pool of self-cleaning slots: one slot per stream for eager launches, a slot of its own
 * for every launch recorded during stream capture (a captured graph may be replayed beside any eager work).
 *   nrt_init            resolves the pool's address; optional (the first launch does it), but call it once per device BEFORE the first
 *                       stream capture so that nothing but launches happens inside the capture.
 *   nrt_counters_reset  zero-fills the pool on `stream`: recovery after a kernel was aborted mid-flight (nothing else leaves a slot
 *                       dirty); not needed in normal operation, must not run beside launches of this library.
 *   nrt_counters_slot_index   index of the slot a launch on `stream` would use right now (-1: more streams than the pool has stream
 *                       slots); diagnostic / tests. */
int nrt_init(void);
int nrt_counters_reset(void *stream);
int nrt_counters_slot_index(void *stream);

/* ------------------------------------------------------------------------------------------
 * interpn / SpatialTransformer / Resize
 * replaces: neurite/tf/utils/utils.py:73-220 (interpn), :223-265 (resize -> interpn on a
 * linspace grid), :333-476 (identity grids), and voxelmorph's transform()/SpatialTransformer
 * (call sites neurite/tf/models.py:806-807, 1157-1159).
 * ------------------------------------------------------------------------------------------ */
typedef enum {
    NRT_LOC_ABSOLUTE = 0,  /* loc[b, q, 0:D] are sampling locations (interpn's own contract)   */
    NRT_LOC_SHIFT = 1,     /* loc[b, q, 0:D] are displacements, location = float(q_d) + loc    */
                           /* (SpatialTransformer: identity 'ij' grid never materialised)      */
    NRT_LOC_LINSPACE = 2   /* loc == NULL; location_d = tf.linspace(0, S_d-1, out_d)[q_d]      */
                           /* (resize()/Resize: align-corners grid computed in registers)      */
} nrt_loc_mode;

typedef enum { NRT_INTERP_LINEAR = 0, NRT_INTERP_NEAREST = 1 } nrt_interp_method;

/*
 * out[b, q, c] = interpn(vol[b], location(b, q))     q over out_shape, c < channels
 *   vol  [batch, vol_shape[0..ndim-1], channels]  (vol_batch_stride elements between volumes;
 *                                                  0 re-uses one volume for every b)
 *   loc  [batch, out_shape[0..ndim-1], ndim]      (loc_batch_stride elements between fields;
 *                                                  0 = single_transform)
 *   out  [batch, out_shape..., channels]           dense
 * ndim in {1,2,3}.  Linear: 2^ndim corners blended in itertools.product order with separately
 * rounded multiplies/adds (bit-identical to the reference's float32 op sequence); nearest:
 * round-half-even, pure data movement; fill: out*(!oob) + oob*fill on the UNCLIPPED location.
 */
int nrt_interpn_f32(const float *vol, const float *loc, float *out,
                    int ndim, const int *vol_shape, const int *out_shape, int channels,
                    int batch, long long vol_batch_stride, long long loc_batch_stride,
                    int loc_mode, int method, int has_fill, float fill_value, void *stream);

/* Same, selecting a specific kernel (for tuning / benchmarking).  variant:
 *   0 auto | 1 generic element-per-thread | 2 row-per-lane-group (C%4==0)
 *   3 z-run with register reuse of the shared corner rows (ndim 3, C==32; the auto choice on regular grids: NRT_LOC_LINSPACE)
 *   5 3-D tiles with a depth-2 software pipeline (C%4==0, linear)
 *   8 few-channel kernel: one voxel per lane, z corners of a row by one load (ndim 3, C<=4; the auto choice there)
 *   10 wave-private LDS row cache on the x-march schedule (ndim 3, C==32, linear; the auto choice for displacement fields and
 *      absolute locations since round 5: at least as fast as variant 3 on every field measured)
 * tune: variant-specific knob (variant 3: z-chunk length | order | patch | region bits; variant 5: tile geometry). */
int nrt_interpn_f32_ex(const float *vol, const float *loc, float *out,
                       int ndim, const int *vol_shape, const int *out_shape, int channels,
                       int batch, long long vol_batch_stride, long long loc_batch_stride,
                       int loc_mode, int method, int has_fill, float fill_value,
                       int variant, int tune, void *stream);

/* Nearest-neighbour lookup on int32 volumes (label maps); fill arithmetic done in int32. */
int nrt_interpn_nearest_i32(const int32_t *vol, const float *loc, int32_t *out,
                            int ndim, const int *vol_shape, const int *out_shape, int channels,
                            int batch, long long vol_batch_stride, long long loc_batch_stride,
                            int loc_mode, int has_fill, int32_t fill_value, void *stream);

/* interpn for every other dtype / rank the reference accepts (neurite/tf/utils/utils.py:106-127, 137-213 are dtype- and
 * rank-generic): float16, bfloat16 and float64 volumes in 1..8 dimensions, float32 and (nearest only) int32 volumes in
 * 4..8 dimensions (TensorFlow itself stops at rank-8 tensors).  `dtype` is an nrt_dtype value (below).  loc is cast to the volume dtype and the arithmetic runs in that
 * dtype, one rounding per operation, as TensorFlow evaluates it.  loc: float32 [batch, out_shape, ndim] (float64 when
 * loc_is_f64, float64 volumes with NRT_LOC_ABSOLUTE only), NULL for NRT_LOC_LINSPACE.  vol / out are `dtype`. */
int nrt_interpn_any(const void *vol, const void *loc, void *out, int dtype, int ndim, const int *vol_shape,
                    const int *out_shape, int channels, int batch, long long vol_batch_stride,
                    long long loc_batch_stride, int loc_mode, int loc_is_f64, int method, int has_fill,
                    double fill_value, void *stream);

/* out = addend + interpn_linear(vol, loc): the update of voxelmorph's compose() (curr + transform(nxt, curr)) and of
 * integrate_vec() (vec += transform(vec, vec); scaling and squaring) in one pass -- call sites
 * neurite/tf/models.py:802-804, 1131, 1149-1154 (VecInt / ComposeTransform next to SpatialTransformer).
 * addend [batch, out_shape, channels]; same rounding as the two separate TF ops (interp, then one add). */
int nrt_interpn_add_f32(const float *vol, const float *loc, const float *addend, float *out, int ndim,
                        const int *vol_shape, const int *out_shape, int channels, int batch,
                        long long vol_batch_stride, long long loc_batch_stride, long long addend_batch_stride,
                        int loc_mode, int has_fill, float fill_value, void *stream);

/* voxelmorph.utils.affine_to_dense_shift ('ij' indexing; called on affine inputs of SpatialTransformer / AffineToDenseShift and by
 * the synthesis models, neurite/tf/models.py:1131-1154): out[b, q, :] = A_b [q - c; 1] - (q - c), c = (shape - 1) / 2 when
 * shift_center else 0.  matrix [batch, ndim, ndim + 1], out [batch, *shape, ndim]; ndim 2 or 3. */
int nrt_affine_to_dense_shift_f32(const float *matrix, int batch, int ndim, const int *shape, int shift_center, float *out,
                                  void *stream);

/* ------------------------------------------------------------------------------------------
 * Dice
 * replaces: neurite/tf/metrics.py:415-482 (Dice.dice) incl. the optional renormalisation
 * (:434-436), the range asserts (:439-444, returned as min/max) and the hard path (:450-468).
 * ------------------------------------------------------------------------------------------ */
size_t nrt_dice_workspace_bytes(long long nvox, int nlabels, int batch);

/*
 * Soft Dice.  y_true, y_pred [batch, nvox, nlabels] float32.
 *   sums   [batch, 3, nlabels] float32 : sum_v t*p, sum_v t*t, sum_v p*p
 *   dice   [batch, nlabels]    float32 : laplace>0 ? (2*tp+eps)/(tt+pp+eps) : divide_no_nan(2*tp, tt+pp)
 *   minmax [4] float32 (may be NULL)   : min t, max t, min p, max p over the whole call
 *                                        (after normalisation if normalize != 0)
 * Deterministic: wavefront shuffles -> LDS -> per-block partials -> fixed-order second stage.
 */
int nrt_dice_soft_f32(const float *y_true, const float *y_pred, long long nvox, int nlabels, int batch,
                      int normalize, float laplace_smoothing,
                      float *sums, float *dice, float *minmax,
                      void *workspace, size_t workspace_bytes, void *stream);
/* The same for maps STORED as float32, bfloat16 or float16 (dtype: NRT_DT_F32 / NRT_DT_BF16 / NRT_DT_F16, both maps alike): the
 * values are widened to float32 in registers and every sum is a float32 sum, as TensorFlow's reductions of 16-bit tensors
 * accumulate; sums / dice / minmax are float32.  (neurite/tf/metrics.py:415-482 is dtype-agnostic.) */
int nrt_dice_soft(const void *y_true, const void *y_pred, int dtype, long long nvox, int nlabels, int batch, int normalize,
                  float laplace_smoothing, float *sums, float *dice, float *minmax, void *workspace, size_t workspace_bytes,
                  void *stream);
/* Sums of the squared difference of two label maps, per batch entry and label, in ONE pass over the two maps: what
 * MeanSquaredErrorProb (neurite/tf/metrics.py:653-692: mean of w_l (y_true - y_pred)^2) reduces.  The difference is formed in
 * registers (exact, no t^2 - 2 t p + p^2 expansion), squared and summed by the nrt_dice_soft_f32 kernels in their order.
 *   a, b [batch, nvox, nlabels]; sums [batch, 3, nlabels] float32, every one of the three rows = sum_v (a - b)^2;
 *   scratch [batch, nlabels] float32 (overwritten); workspace as nrt_dice_soft_f32. */
int nrt_sqdiff_sums_f32(const float *a, const float *b, long long nvox, int nlabels, int batch, float *sums, float *scratch,
                        void *workspace, size_t workspace_bytes, void *stream);


/*
 * Hard Dice from probabilistic maps: argmax over labels (ties -> lowest index) then one-hot.
 *   counts [batch, 3, nlabels] int64 : #(a_t==l && a_p==l), #(a_t==l), #(a_p==l)   (exact)
 *   dice   [batch, nlabels] float32
 */
int nrt_dice_hard_prob_f32(const float *y_true, const float *y_pred, long long nvox, int nlabels, int batch,
                           float laplace_smoothing, long long *counts, float *dice,
                           void *workspace, size_t workspace_bytes, void *stream);
/* the same with minmax [4] = min t, max t, min p, max p of the inputs from the same pass (the range asserts of
 * neurite/tf/metrics.py:439-444); nlabels a multiple of 4 up to 256 and 16-byte aligned maps, else NRT_ERR_UNSUPPORTED */
int nrt_dice_hard_prob_minmax_f32(const float *y_true, const float *y_pred, long long nvox, int nlabels, int batch,
                                  float laplace_smoothing, long long *counts, float *dice, float *minmax,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* Hard Dice of probability maps stored as float32 / bfloat16 / float16 (arg-max of the stored values: exact in any of them);
 * minmax [4] may be NULL. */
int nrt_dice_hard_prob(const void *y_true, const void *y_pred, int dtype, long long nvox, int nlabels, int batch,
                       float laplace_smoothing, long long *counts, float *dice, float *minmax, void *workspace,
                       size_t workspace_bytes, void *stream);


/* Hard Dice from label maps [batch, nvox] int32; labels outside [0, nlabels) match nothing.  workspace: NULL, or
 * nrt_dice_workspace_bytes(nvox, nlabels, batch) bytes (then the block histograms are reduced without global atomics). */
int nrt_dice_hard_label_i32(const int32_t *y_true, const int32_t *y_pred, long long nvox, int nlabels,
                            int batch, float laplace_smoothing, long long *counts, float *dice,
                            void *workspace, size_t workspace_bytes, void *stream);

/* dice[b,l] from externally reduced sums (e.g. after an RCCL all-reduce of `sums`). */
int nrt_dice_from_sums_f32(const float *sums, int nlabels, int batch, float laplace_smoothing,
                           float *dice, void *stream);

/* out2[0] = sum over the [batch, nlabels] entries of dice * weights, out2[1] = batch * nlabels: the pair a rank
 * contributes to the single all-reduce behind Dice.mean_dice over a batch sharded across GPUs
 * (neurite/tf/metrics.py:499-510, K.mean(dice * weights)).  weights: NULL, [nlabels], or with
 * weights_per_batch != 0 [batch, nlabels].  One launch, fixed summation order. */
int nrt_dice_mean_pair_f32(const float *dice, const float *weights, int nlabels, int batch,
                           int weights_per_batch, float *out2, void *stream);

/* The same launch writing three floats: [sum, count, sum / count] -- a process that is alone (no collective to wait for) reads the mean
 * K.mean(dice * weights) of neurite/tf/metrics.py:499-510 from out3[2] without launching the division. */
int nrt_dice_mean_f32(const float *dice, const float *weights, int nlabels, int batch, int weights_per_batch, float *out3,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused SpatialTransformer + soft Dice  (the BASELINE "interpn+Dice" pipeline in one pass)
 * replaces: SpatialTransformer (see nrt_interpn_f32, NRT_LOC_SHIFT) immediately followed by
 * Dice.dice(fixed, warped) (neurite/tf/metrics.py:415-482) without materialising `warped`.
 *   moving [batch, vol_shape, nlabels], loc [batch, out_shape, 3] (or NULL for NRT_LOC_LINSPACE),
 *   fixed  [batch, out_shape, nlabels];  warped [batch, out_shape, nlabels] or NULL (not written).
 *   sums / dice / minmax as nrt_dice_soft_f32 with y_true = fixed, y_pred = warped.
 * 3-D only, nlabels in {4, 8, 16, 32, 64, 128, 256}.  tune: tile shape knob (0 = default).
 * Size limits (NRT_ERR_UNSUPPORTED beyond them; use nrt_interpn_f32 + nrt_dice_soft_f32): one batch entry of moving /
 * fixed / warped below 4 GiB, shape[0] * shape[1] and shape[2] below 2^24 for both shapes (32-bit row offsets, 24-bit
 * index multiplies).
 * ------------------------------------------------------------------------------------------ */
size_t nrt_warp_dice_workspace_bytes(const int *out_shape, int nlabels, int batch, int tune);
int nrt_warp_dice_soft_f32(const float *moving, const float *loc, const float *fixed, float *warped,
                           const int *vol_shape, const int *out_shape, int nlabels, int batch,
                           long long loc_batch_stride, int loc_mode, int has_fill, float fill_value,
                           float laplace_smoothing, float *sums, float *dice, float *minmax,
                           int tune, void *workspace, size_t workspace_bytes, void *stream);

/* The same with the two label maps STORED as bfloat16 (moving [batch, vol_shape, nlabels], fixed [batch, out_shape, nlabels]):
 * half the bytes per row (a 32-label row is 64 B, two z-neighbours share a cache line).  The rows are widened to float32 in
 * registers and the arithmetic is the float32 kernel's, so for maps whose values are bfloat16 numbers (one-hot label maps)
 * sums / dice are bit-identical to nrt_warp_dice_soft_f32 on the widened maps.  An extension of this package (the reference
 * has no fused form; TensorFlow would blend bfloat16 tensors in bfloat16 -- nrt_interpn_any does that for interpn).
 * No `warped` output. */
int nrt_warp_dice_soft_bf16(const void *moving, const float *loc, const void *fixed, const int *vol_shape,
                            const int *out_shape, int nlabels, int batch, long long loc_batch_stride, int loc_mode,
                            int has_fill, float fill_value, float laplace_smoothing, float *sums, float *dice,
                            float *minmax, int tune, void *workspace, size_t workspace_bytes, void *stream);

/* Name of the kernel instantiation nrt_warp_dice_soft_f32 launches for these arguments, as a profiler prints it (e.g.
 * "warp_dice_tile<8, 1, false, 3, float>"); "" for arguments the entry point rejects.  The returned buffer is thread-local.
 * Measurement plumbing (bench.py joins its timing with the counter passes under profiles/ by this name); no reference counterpart. */
const char *nrt_warp_dice_kernel_name(const int *out_shape, const int *vol_shape, int nlabels, int batch, int loc_mode, int has_fill,
                                      int store, int want_minmax, int tune);

/* ------------------------------------------------------------------------------------------
 * Label-weighted categorical cross-entropy
 * replaces: neurite/tf/metrics.py:640-650 + tf.keras.losses.CategoricalCrossentropy
 * ------------------------------------------------------------------------------------------ */
typedef enum { NRT_DT_F32 = 0, NRT_DT_BF16 = 1, NRT_DT_F16 = 2, NRT_DT_F64 = 3, NRT_DT_I32 = 4 } nrt_dtype;

size_t nrt_wcce_workspace_bytes(long long nvox_total, int channels);

/*
 * y_true, y_pred [nvox_total, channels] of `dtype`; label_weights [channels] float32 or NULL.
 *   t' = w*t; [t' = t'*(1-s) + s/C]; from_logits ? -sum t' log_softmax(z)
 *                                               : q = clip(p/sum p, 1e-7, 1-1e-7), -sum t' log q
 *   loss_sum  [1] float32 : sum over all voxels (the caller divides by nvox_total)
 *   per_voxel [nvox_total] float32 or NULL
 * Arithmetic in float32 regardless of the input dtype.
 */
int nrt_wcce(const void *y_true, const void *y_pred, int dtype, const float *label_weights,
             long long nvox_total, int channels, int from_logits, float label_smoothing,
             float *loss_sum, float *per_voxel,
             void *workspace, size_t workspace_bytes, void *stream);
/* The same with the division of the default reduction ('sum_over_batch_size': metrics.py:650 -> Keras CategoricalCrossentropy) done by
 * the block that finishes the sum: loss_mean[0] = float32(sum) / float32(divide_by); one launch, nothing for the caller to divide. */
int nrt_wcce_mean(const void *y_true, const void *y_pred, int dtype, const float *label_weights,
                  long long nvox_total, int channels, int from_logits, float label_smoothing, double divide_by,
                  float *loss_mean, float *per_voxel,
                  void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * unet layers
 * replaces the Keras layers that neurite/tf/models.py instantiates: Conv3D (:1378-1388, :1545-1555,
 * :1596), MaxPooling3D (:1436-1438), UpSampling3D + concatenate (:1531-1542), the channel softmax
 * (:1601-1605) and the residual add / activation / BatchNormalization (:1401-1433).
 * All tensors channels-last float32 [batch, X, Y, Z, C]; `shape`, `ksize`, `pool`, `up` are host int[3].
 * ------------------------------------------------------------------------------------------ */
/* element-wise activations as tf.keras.activations defines them (elu alpha 1, hard_sigmoid = clip(0.2 x + 0.5, 0, 1), leaky_relu slope
 * 0.2).  The convolution / LocallyConnected3D entry points (nrt_conv3d_f32, nrt_conv3d_up2_f32, nrt_conv1x1_softmax_f32,
 * nrt_lc3d_f) fuse NONE / ELU / RELU into their epilogues and return NRT_ERR_INVALID_ARG for the other codes: run those as an
 * element-wise pass over the layer output (nrt_add_act_affine_f32, which accepts all of them, as do nrt_act_bwd_f32 and the
 * backward entry points).  The channel softmax is a kernel of its own (nrt_softmax_lastdim_f32). */
typedef enum { NRT_ACT_NONE = 0, NRT_ACT_ELU = 1, NRT_ACT_RELU = 2, NRT_ACT_SIGMOID = 3, NRT_ACT_TANH = 4, NRT_ACT_SOFTPLUS = 5,
               NRT_ACT_SOFTSIGN = 6, NRT_ACT_SELU = 7, NRT_ACT_EXPONENTIAL = 8, NRT_ACT_HARD_SIGMOID = 9, NRT_ACT_LEAKY_RELU = 10 } nrt_activation;
/* nrt_add_act_affine_f32 only: or-ed into `activation`, y = act(a) * b instead of act(a + b) (models.add_prior, use_logp=False) */
#define NRT_ACT_MUL_B 0x100

/* Weights re-ordered for the MFMA kernel ("packed"): query the size, pack once per layer. */
size_t nrt_conv3d_packed_weight_floats(const int *ksize, int cin, int cout);
int nrt_conv3d_pack_weights_f32(const float *weights /* Keras [kx,ky,kz,cin,cout] */, const int *ksize,
                                int cin, int cout, float *packed, void *stream);

/*
 * y = act(conv3d(concat(src0, upsample_nearest(src1, up)), W) + bias), cross-correlation, stride 1,
 * dilation `dilation`, SAME (output = shape) or VALID padding.
 *   src0 [batch, shape, c0]; src1 [batch, shape/up, c1] or NULL with c1 = 0 (plain convolution);
 *   weights Keras layout [kx,ky,kz,c0+c1,cout] (direct kernel), packed_weights from
 *   nrt_conv3d_pack_weights_f32 (MFMA kernel; may be NULL => direct kernel); bias [cout] or NULL.
 * variant 0 = auto (MFMA implicit GEMM when k in {1,3}^3, SAME, dilation <= 2, cout <= 64, cin >= 8; its persistent LDS-DMA
 * schedule for 3x3x3, dilation 1, cin % 16 == 0 when there is more than one tile per CU), 1 = direct, 2 = MFMA (one tile per
 * block), 5 = MFMA in the persistent schedule.
 */
int nrt_conv3d_f32(const float *src0, int c0, const float *src1, int c1, const int *up,
                   const float *weights, const float *packed_weights, const float *bias, float *out,
                   int batch, const int *shape, const int *ksize, int cout, int dilation,
                   int padding_same, int activation, int variant, void *stream);

/*
 * The single-channel first encoder convolution with the 2x2x2 MaxPooling3D behind it (neurite/tf/models.py:1378-1388 and
 * :1436-1438) in one kernel: `out` [batch, shape, cout] exactly as nrt_conv3d_f32 writes it (the decoder's skip connection reads
 * it), and `pool_out` [batch, shape / 2, cout] = MaxPooling3D(2)(out) in addition -- the pooled values are taken from the tile
 * while it is in LDS instead of reading the full-resolution tensor back.  3x3x3 SAME, weights in Keras layout [3,3,3,1,cout],
 * cout 16 or 32, shape a multiple of (4, 4, 16); bit-identical to nrt_conv3d_f32 followed by nrt_maxpool3d_f32.
 */
int nrt_conv3d_c1_pool_supported(const int *shape, int cout);
int nrt_conv3d_c1_pool_f32(const float *src /* [batch, shape, 1] */, const float *weights, const float *bias, float *out,
                           float *pool_out, int batch, const int *shape, int cout, int activation, void *stream);
/*
 * The same pattern below the first level (round 6): a 3x3x3 SAME convolution over c0 = 16 k input channels (packed weights of
 * nrt_conv3d_pack_weights_f32) and the MaxPooling3D(2) of its activated output from one kernel (models.py:1378-1388 + 1436-1438) --
 * `out` [batch, shape, cout] as nrt_conv3d_f32 (variant 5) writes it, `pool_out` [batch, shape / 2, cout] in addition.  cout 32, 48 or
 * 64, shape a multiple of (4, 4, 16); bit-identical to nrt_conv3d_f32 followed by nrt_maxpool3d_f32.
 */
int nrt_conv3d_pool_supported(int c0, int cout, const int *shape, int batch);
int nrt_conv3d_pool_f32(const float *src /* [batch, shape, c0] */, int c0, const float *packed_weights, const float *bias, float *out,
                        float *pool_out, int batch, const int *shape, int cout, int activation, void *stream);

/*
 * The decoder convolution of models.unet (neurite/tf/models.py:1531-1555: UpSampling3D(2) -> concatenate([skip, up]) ->
 * Conv3D 3x3x3 SAME) with the up-sampled half FOLDED: on a nearest-up-sampled tensor the 27 taps collapse to 8 taps
 * on the low-resolution grid, with one of 8 pre-summed weight sets chosen by the parity of the output voxel
 * (K = 27 c0 + 8 c1 instead of 27 (c0 + c1)).  Same result as nrt_conv3d_f32(skip, c0, lo, c1, up = {2,2,2}, ...)
 * up to float32 rounding of the weight sums.  c0 % 16 == 0, c1 % 16 == 0, cout <= 64, activation none / elu / relu.
 *   nrt_conv3d_up2_supported            1 when the shapes qualify (otherwise use nrt_conv3d_f32)
 *   nrt_conv3d_up2_packed_weight_floats size of the folded, fragment-ordered weights
 *   nrt_conv3d_up2_pack_weights_f32     weights Keras layout [3,3,3,c0+c1,cout] -> packed
 */
int nrt_conv3d_up2_supported(int c0, int c1, int cout, const int *shape);
size_t nrt_conv3d_up2_packed_weight_floats(int c0, int c1, int cout);
int nrt_conv3d_up2_pack_weights_f32(const float *weights, int c0, int c1, int cout, float *packed, void *stream);
int nrt_conv3d_up2_f32(const float *skip /* [batch, shape, c0] */, int c0, const float *lo /* [batch, shape/2, c1] */, int c1,
                       const float *packed_weights, const float *bias, float *out, int batch, const int *shape, int cout,
                       int activation, void *stream);
/*
 * The LAST decoder convolution of models.unet with the network's head folded in (neurite/tf/models.py:1545-1555 the convolution,
 * :1596 the 1x1x1 "likelihood" convolution, :1601-1605 the channel soft-max): out = softmax(act(conv_up2(skip, lo)) @ head_weights
 * + head_bias) [batch, shape, labels]; the 16-channel feature tensor between the two convolutions is never written.  Inference
 * only (the backward of a training step needs that tensor).  cout == 16, labels 16 or 32, shape a multiple of (4, 4, 16), c0 < c1;
 * the head's kernel packed by nrt_conv3d_up2_head_pack_f32 (Keras layout [16, labels] -> 16 * labels floats in matrix-core
 * fragment order), both head arrays 16-byte aligned.  Same values as nrt_conv3d_up2_f32 followed by nrt_conv1x1_softmax_f32 up
 * to the float32 summation order of the 16-term head products.
 */
int nrt_conv3d_up2_head_supported(int c0, int c1, int cout, int labels, const int *shape);
int nrt_conv3d_up2_head_pack_f32(const float *head_weights /* [16, labels] */, int labels, float *packed /* [16 * labels] */, void *stream);
int nrt_conv3d_up2_head_f32(const float *skip, int c0, const float *lo, int c1, const float *packed_weights, const float *bias,
                            const float *packed_head_weights, const float *head_bias /* [labels] */, int labels,
                            float *out /* [batch, shape, labels] */, int batch, const int *shape, int cout, int activation,
                            void *stream);

/*
 * Backward of the folded decoder convolution with respect to its low-resolution input (what tf.GradientTape derives through
 * UpSampling3D + concatenate + Conv3D, neurite/tf/models.py:1531-1555): the gradient of the up-sampled tensor summed over
 * the 2^3 blocks is a 4x4x4 stride-2 convolution of dpre = 8 parity sub-lattices x 2x2x2 taps on the low-resolution grid.
 *   nrt_space_to_depth2_f32   y[b][q][P * channels + c] = x[b][2 q + p][c], P = (px * 2 + py) * 2 + pz (shape = x's, even)
 *   nrt_conv3d_s2d_taps_f32   3x3x3 SAME convolution of x [batch, shape, 8 * group] in which the channels of parity group P only
 *                             use the taps e = (p ? 1 : 2) - t, t in {0, 1}, per axis (all other weights are taken as zero);
 *                             packed_weights = nrt_conv3d_pack_weights_f32 of [3,3,3, 8 * group, cout]; group % 16 == 0
 */
int nrt_space_to_depth2_f32(const float *x, float *y, int batch, const int *shape, int channels, void *stream);
int nrt_conv3d_s2d_taps_f32(const float *x, int group, const float *packed_weights, float *out, int batch, const int *shape,
                            int cout, void *stream);

/* 1x1 convolution with the channel softmax (or an activation) fused; cout <= 64. x [nvox, cin]. */
int nrt_conv1x1_softmax_f32(const float *x, const float *weights /* [cin, cout] */, const float *bias,
                            float *y, long long nvox, int cin, int cout, int softmax, int activation,
                            void *stream);
int nrt_softmax_lastdim_f32(const float *x, float *y, long long n, int channels, void *stream);
/* MaxPooling3D, stride = pool; SAME keeps partial windows (output ceil(n/p)), VALID drops them. */
int nrt_maxpool3d_f32(const float *x, float *y, int batch, const int *shape, int channels,
                      const int *pool, int padding_same, void *stream);
/* y = concat(skip [batch, shape, c0] (may be NULL with c0 = 0), upsample_nearest(lo [batch, shape/up, c1])) */
int nrt_upsample_concat_f32(const float *skip, int c0, const float *lo, int c1, float *y, int batch,
                            const int *shape, const int *up, void *stream);
/* y = act(a + b) * scale[c] + shift[c]   (b, scale/shift optional): residual merge / inference BatchNorm */
int nrt_add_act_affine_f32(const float *a, const float *b, const float *scale, const float *shift,
                           float *y, long long n, int channels, int activation, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the conv stack (what tf.GradientTape derives for neurite/tf/models.py:1345-1347, 1385, 1438,
 * 1506-1508, 1531, 1604; float32, channels-last, stride 1)
 *   nrt_act_bwd_f32        grad_pre = grad_out * act'(y), from the layer OUTPUT y (ELU: y > 0 ? 1 : y + 1)
 *   nrt_conv3d_wgrad_f32   grad_weights [kx,ky,kz,cin,cout] += sum_v x[v + off(tap)] (x) grad_pre[v] (SAME padding) and
 *                          grad_bias [cout] += sum_v grad_pre[v]; both must be ZERO-FILLED by the caller (float
 *                          atomics, one per weight and block); x is the conv input as the layer saw it
 *                          (the concatenation, if any, materialised); ksize entries 1 or 3, dilation <= 2
 *   (grad_x = nrt_conv3d_f32(grad_pre, weights flipped in space and transposed in the channel axes))
 *   nrt_maxpool3d_bwd_f32  stride == pool: the window's gradient goes to its first maximum
 *   nrt_upsample_sum_f32   nearest up-sampling: grad_lo[v] = sum over the up^3 fine voxels of channels
 *                          [channel_offset, channel_offset + channels) of grad_up [.., grad_channels]
 *   nrt_softmax_bwd_f32    grad_in = y * (grad_out - sum_c grad_out_c y_c)
 * ------------------------------------------------------------------------------------------ */
int nrt_act_bwd_f32(const float *grad_out, const float *y, int activation, float *grad_pre, long long n, void *stream);
int nrt_conv3d_wgrad_f32(const float *x, const float *grad_pre, float *grad_weights, float *grad_bias, int batch,
                         const int *shape, int cin, int cout, const int *ksize, int dilation, void *stream);
/* the same with the forward kernel's fused loader: the layer input was concat(x [.., c0], UpSampling3D(x_lo [.., c1])), which
 * is read from its two sources and never materialised (c0 % 4 == 0; x_lo may be NULL with c1 = 0) */
int nrt_conv3d_wgrad2_f32(const float *x, int c0, const float *x_lo, int c1, const int *up, const float *grad_pre,
                          float *grad_weights, float *grad_bias, int batch, const int *shape, int cout, const int *ksize,
                          int dilation, void *stream);
/* folded form for the up-sampled channels of a decoder convolution (see nrt_conv3d_up2_f32): x_lo [batch, shape, cin] is the
 * low-resolution tensor, grad_pre_s2d [batch, shape, 8 * group] = nrt_space_to_depth2_f32 of the layer's grad_pre (group = cout);
 * grad_folded [8 parity groups][8 taps (tx, ty, tz)][cin][group] += sum_q x_lo[q + p + t - 1] (x) grad_pre_s2d[q][P], ZERO-FILLED by
 * the caller; the 27-tap gradient is dW[d] = sum over the (p, t) with d in S(p, t) per axis (S(0,0) = {0}, S(0,1) = {1,2},
 * S(1,0) = {0,1}, S(1,1) = {2}) -- a 27 x 64 matrix product over cin * group values (host side) */
int nrt_conv3d_wgrad_s2d_f32(const float *x_lo, const float *grad_pre_s2d, float *grad_folded, int batch, const int *shape, int cin,
                             int group, void *stream);
int nrt_maxpool3d_bwd_f32(const float *x, const float *grad_out, float *grad_x, int batch, const int *shape,
                          int channels, const int *pool, int padding_same, void *stream);
int nrt_upsample_sum_f32(const float *grad_up, int grad_channels, int channel_offset, float *grad_lo, int channels,
                         int batch, const int *lo_shape, const int *up, void *stream);
int nrt_softmax_bwd_f32(const float *y, const float *grad_out, float *grad_in, long long nvox, int channels, void *stream);
/* BatchNormalization in training mode (Keras axis -1; models.py:1431-1434, 1585-1588): per-channel sums over [rows, channels],
 * out[c] += sum_r a[r][c] * (b ? b[r][c] : 1)  (ZERO-FILLED by the caller), and y = coef_a[c] a + coef_b[c] b + coef_c[c] */
int nrt_channel_sums_f32(const float *a, const float *b, long long rows, int channels, float *out, void *stream);
int nrt_channel_axpby_f32(const float *a, const float *b, const float *coef_a, const float *coef_b, const float *coef_c, float *y,
                          long long n, int channels, void *stream);

/* ------------------------------------------------------------------------------------------
 * LocallyConnected3D, implementation 1 ('valid' padding, channels-last)
 * replaces: neurite/tf/layers.py:1126-1197 (local_conv: O slice ops + concat + K.batch_dot) and the
 * bias / activation of :1098-1101.
 *   x [batch, in_shape, cin]; kernel [O, kr*kc*kz*cin, cout] (feature order kr,kc,kz,cin; O = output
 *   positions row-major); bias [O, cout] or NULL; y [batch, out_shape, cout]; all of `dtype`
 *   (NRT_DT_F32 or NRT_DT_BF16), accumulation in float32.  Weights are read exactly once per 1-2 batch entries (vector kernel) or per
 *   <= 8 entries (batches of 3 and more: matrix-core kernel, v_mfma_f32_4x4x1; 16-channel 3x3x3 layers stage their patches per block).
 * variant 0 auto | 1 generic | 2 weight-streaming wave-per-position kernels.
 * ------------------------------------------------------------------------------------------ */
int nrt_lc3d_f(const void *x, const void *kernel, const void *bias, void *y, int dtype, int batch,
               const int *in_shape, int cin, const int *ksize, const int *strides, int cout,
               int activation, int variant, void *stream);
/* Backward of the same layer (autodiff of the batch_dot of layers.py:1189 + bias/activation :1098-1101):
 *   grad_kernel [O, F, cout] and grad_bias [O, cout] (`dtype`, written once, may be NULL), grad_x float32
 *   [batch, in_shape, cin] accumulated with float atomics (ZERO-FILLED by the caller, may be NULL);
 *   y = the layer output (needed when activation != 0), grad_out [batch, out_shape, cout].
 *   cout * itemsize must be a multiple of 16 (the weight-streaming lane layout). */
int nrt_lc3d_bwd_f(const void *x, const void *kernel, const void *y, const void *grad_out, void *grad_kernel,
                   void *grad_bias, float *grad_x, int dtype, int batch, const int *in_shape, int cin,
                   const int *ksize, const int *strides, int cout, int activation, void *stream);

/* Zero-pad (crop == 0) a channels-last volume [batch, in_shape, row] into [batch, out_shape, row] with pad_before voxels
 * in front of every axis, or (crop != 0) copy the interior of a padded volume back out -- `in` is then the padded
 * [batch, out_shape, row] tensor and `out` the [batch, in_shape, row] one.  row_bytes = channels * itemsize (even).
 * padding='same' of LocallyConnected3D implementations 2 / 3 (neurite/tf/layers.py:934-936, 1474-1482: the window is
 * clipped at the border) is the 'valid' layer on the input padded by kernel_size // 2; the crop is its gradient. */
int nrt_pad3d(const void *in, void *out, int batch, const int *in_shape, const int *pad_before, const int *out_shape,
              int row_bytes, int crop, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward passes (what tf.GradientTape derives from the reference graphs; float32)
 *
 * nrt_interpn_bwd_f32: gradients of linear interpn / transform (neurite/tf/utils/utils.py:137-191, :206-213).
 *   grad_out [batch, out_shape, C]; grad_vol [batch, vol_shape, C] must be ZERO-FILLED by the caller and is
 *   accumulated with float atomics (tf.gather's scatter-add gradient), or NULL; grad_loc [batch, out_shape, D]
 *   is overwritten (gradient wrt loc == gradient wrt the displacement in NRT_LOC_SHIFT mode), or NULL.
 *   The location gradient flows only through clipped_loc (:142; tf.floor has none) on the closed range
 *   [0, size-1]; with has_fill the gradient is zero at out-of-bounds voxels (:209-213).
 * nrt_dice_soft_bwd_f32: d dice[b,l] / d y_pred and / d y_true from the saved sums [batch, 3, L]
 *   (neurite/tf/metrics.py:476-482; divide_no_nan => zero gradient where the denominator is 0).
 * nrt_wcce_bwd_f32: d loss / d y_pred of the weighted CCE (metrics.py:648-650 + Keras normalise/clip/log);
 *   upstream gradient either one device scalar (grad_scalar) or per voxel (grad_per_voxel), times `scale`.
 * ------------------------------------------------------------------------------------------ */
int nrt_interpn_bwd_f32(const float *vol, const float *loc, const float *grad_out, float *grad_vol,
                        float *grad_loc, int ndim, const int *vol_shape, const int *out_shape, int channels,
                        int batch, long long vol_batch_stride, long long loc_batch_stride, int loc_mode,
                        int has_fill, void *stream);

/* Backward of nearest-neighbour interpn (neurite/tf/utils/utils.py:193-204) wrt the volume: tf.gather's scatter-add of
 * grad_out [batch, out_shape, channels] into grad_vol [batch, vol_shape, channels] (ZERO-FILLED by the caller; float atomics),
 * masked where a fill value applies.  The location has no gradient (tf.round). */
int nrt_interpn_nearest_bwd_f32(const float *loc, const float *grad_out, float *grad_vol, int ndim, const int *vol_shape,
                                const int *out_shape, int channels, int batch, long long vol_batch_stride,
                                long long loc_batch_stride, int loc_mode, int has_fill, void *stream);
int nrt_dice_soft_bwd_f32(const float *y_true, const float *y_pred, const float *sums, const float *grad_dice,
                          long long nvox, int nlabels, int batch, float laplace_smoothing, float *grad_pred,
                          float *grad_true, void *stream);

/* The same for Dice(normalize=True) (neurite/tf/metrics.py:434-436): `sums` are those of the per-voxel normalised maps, the
 * gradient is pulled back through t <- divide_no_nan(t, sum_l t) (and p likewise) to the RAW maps y_true / y_pred. */
int nrt_dice_soft_bwd_norm_f32(const float *y_true, const float *y_pred, const float *sums, const float *grad_dice,
                               long long nvox, int nlabels, int batch, float laplace_smoothing, float *grad_pred,
                               float *grad_true, void *stream);
/* Fused backward of nrt_warp_dice_soft_f32 wrt the displacement / location field: rebuilds the warped row in
 * registers, forms d dice / d warped from `sums` (as returned by the forward) and grad_dice [batch, L], and writes
 * grad_loc [batch, out_shape, 3] only.  `warped` and its gradient never touch HBM.  At 32 float32 labels on volumes that take the
 * forward's x-march schedule it runs on the forward's wave-cache gather (csrc/fused_wc.h, BWD; environment NRT_BWD_WC=0 selects the
 * register-pipelined kernel of rounds 2-4: same bits); the grad_loc of nrt_interpn_bwd_f32 at 32 channels likewise. */
int nrt_warp_dice_bwd_f32(const float *moving, const float *loc, const float *fixed, const float *sums,
                          const float *grad_dice, float *grad_loc, const int *vol_shape, const int *out_shape,
                          int nlabels, int batch, long long loc_batch_stride, int loc_mode, int has_fill,
                          float laplace_smoothing, void *stream);
int nrt_wcce_bwd_f32(const float *y_true, const float *y_pred, const float *label_weights,
                     const float *grad_scalar, const float *grad_per_voxel, long long nvox_total, int channels,
                     int from_logits, float label_smoothing, float scale, float *grad_pred, void *stream);

/* ------------------------------------------------------------------------------------------
 * Segmentation training loss: soft Dice (neurite/tf/metrics.py:415-482) AND the label-weighted categorical cross-entropy
 * (metrics.py:619-650, probabilities, not logits) of the same pair of maps [batch, nvox, nlabels] float32 in one pass per direction --
 * what neurite/tf/losses.py:225-246 (multiple_losses_decorator) evaluates as two losses over the unet's soft-max output
 * (models.py:1545-1555).  nlabels = 4, 8, ..., 256 (a power of two; nrt_seg_loss_supported), 16-byte aligned maps.
 *   nrt_seg_loss_f32       sums [batch, 3, L], dice [batch, L], minmax [4] or NULL exactly as nrt_dice_soft_f32 (normalize = 0);
 *                          cce_sum [1] exactly as nrt_wcce (from_logits = 0, no per-voxel output).
 *   nrt_seg_loss_bwd_f32   grad = d( sum_bl grad_dice[b,l] dice[b,l] + grad_cce[0] cce_sum ) / d y_pred (either upstream pointer may
 *                          be NULL = 0); through_softmax != 0: y_pred = softmax(z) over the labels and grad = d / d z (what the
 *                          head's nrt_softmax_bwd_f32 would make of it).
 * ------------------------------------------------------------------------------------------ */
int nrt_seg_loss_supported(int nlabels);
size_t nrt_seg_loss_workspace_bytes(long long nvox, int nlabels, int batch);
int nrt_seg_loss_f32(const float *y_true, const float *y_pred, const float *label_weights, long long nvox, int nlabels, int batch,
                     float label_smoothing, float laplace_smoothing, float *sums, float *dice, float *minmax, float *cce_sum,
                     void *workspace, size_t workspace_bytes, void *stream);
int nrt_seg_loss_bwd_f32(const float *y_true, const float *y_pred, const float *label_weights, const float *sums,
                         const float *grad_dice, const float *grad_cce, long long nvox, int nlabels, int batch,
                         float label_smoothing, float laplace_smoothing, int through_softmax, float *grad, void *stream);

/* ------------------------------------------------------------------------------------------
 * Synthesis front-end: separable filtering and min-max normalisation
 *   nrt_conv1d_axis_f32   one pass of utils.separable_conv (neurite/tf/utils/utils.py:665-751) / layers.GaussianBlur
 *                         (layers.py:251-364): the tensor viewed as [outer, axis_len, inner] around the filtered axis,
 *                         y[o, a, i] = sum_t kernel[t] * x[o, a*stride + t*dilation - pad_before, i], zero outside
 *                         (cross-correlation, as tf.nn.convolution); y is [outer, out_len, inner]
 *   nrt_minmax_norm_f32   utils.minmax_norm (utils.py:953-968) over a contiguous run of axes: x viewed as
 *                         [outer, reduce_len, inner]; y = div_no_nan(x - min, max - min) per (outer, inner)
 * ------------------------------------------------------------------------------------------ */
int nrt_conv1d_axis_f32(const float *x, const float *kernel, float *y, long long outer, int axis_len, long long inner,
                        int out_len, int width, int stride, int dilation, int pad_before, void *stream);
size_t nrt_minmax_workspace_bytes(long long outer, int inner);
int nrt_minmax_norm_f32(const float *x, float *y, long long outer, long long reduce_len, int inner, void *workspace,
                        size_t workspace_bytes, void *stream);
/* out2 = {min, max} of x[0..n) on the device (no host synchronisation); workspace >= nrt_minmax_workspace_bytes(1, 1) */
int nrt_minmax_f32(const float *x, long long n, float *out2, void *workspace, size_t workspace_bytes, void *stream);
/* centers[0..nb_bins) = tf.linspace(min(x), max(x), nb_bins), the bin centres of utils.soft_quantize / MutualInformation
 * (neurite/tf/utils/utils.py:1152-1154), on the device; workspace as nrt_minmax_f32 */
int nrt_bin_centers_f32(const float *x, long long n, int nb_bins, float *centers, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * Soft quantisation and mutual information (neurite/tf/utils/utils.py:1099-1172, neurite/tf/metrics.py:41-336)
 *   nrt_soft_quantize_f32   out[v, b] = exp(-alpha (clip(x_v) - centers_b)^2)  (or its log)
 *   nrt_mi_joint_f32        x, y [batch, nvox, channels]; per item (b, c): joint[i][j] += sum_v wx_i(v) wy_j(v),
 *                           sum_x[i] += sum_v wx_i(v), sum_y likewise -- the contraction of MutualInformation.channelwise
 *                           without the [V, nb] maps (fp32 MFMA, weights formed in registers); nb_bins <= 32;
 *                           outputs are accumulated with float atomics and must be ZERO-FILLED by the caller
 *   nrt_mi_joint_bwd_f32    gradient wrt x / y given d loss / d joint, d sum_x, d sum_y (bin centres held constant)
 *   nrt_colsum_f32          out[item, c] += sum_r x[item, r, c]  (marginals of probability maps; zero-filled by the caller)
 * ------------------------------------------------------------------------------------------ */
int nrt_soft_quantize_f32(const float *x, const float *centers, float alpha, float min_clip, float max_clip, int return_log,
                          float *out, long long n, int nb_bins, void *stream);

/* Backward of soft_quantize wrt x (bin centres held constant): grad_x [n] from grad_out [n, nb_bins]. */
int nrt_soft_quantize_bwd_f32(const float *x, const float *centers, float alpha, float min_clip, float max_clip,
                              int return_log, const float *grad_out, float *grad_x, long long n, int nb_bins, void *stream);
int nrt_mi_joint_f32(const float *x, const float *y, const float *centers_x, const float *centers_y, float alpha, float min_clip,
                     float max_clip, int batch, long long nvox, int channels, int nb_bins, float *joint, float *sum_x,
                     float *sum_y, void *stream);
int nrt_mi_joint_bwd_f32(const float *x, const float *y, const float *centers_x, const float *centers_y, float alpha,
                         float min_clip, float max_clip, int batch, long long nvox, int channels, int nb_bins,
                         const float *grad_joint, const float *grad_sum_x, const float *grad_sum_y, float *grad_x, float *grad_y,
                         void *stream);
int nrt_colsum_f32(const float *x, int items, long long rows, int cols, float *out, void *stream);
/* mi[item] from joint [items, nb, nb] and the marginal sums [items, nb] (neurite/tf/metrics.py:262-281); nb_bins <= 64 */
int nrt_mi_from_joint_f32(const float *joint, const float *sum_x, const float *sum_y, int items, int nb_bins, float eps, float *mi,
                          void *stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise stages of the label-to-image synthesis model (neurite/tf/models.py:649-918); random numbers are inputs
 *   nrt_synth_relabel_i32     out[i] = lut[labels[i]] (labels -> dense indices, :778-784)
 *   nrt_synth_intensity_f32   labels, noise [B, V] (indices; one draw per voxel), out [B, V, C], mean/std [B, C, L], bg_zero [B, C] or NULL:
 *                             out = noise * std[label] + mean[label], zeroed where label == 0 and bg_zero (:819-849)
 *   nrt_synth_bias_clip_f32   out = clip(image * exp(bias), lo, hi); image [n, C], bias [n] or NULL (:860-874)
 *   nrt_synth_gamma_dc_f32    out = image ^ exp(gamma[b, c]) + dc[b, c]; gamma / dc [B, C] or NULL (:877-888)
 *   nrt_synth_labels_out      indices -> lut (int32 out) or its one-hot encoding [n, depth] (-1 = dropped label) (:890-918)
 * ------------------------------------------------------------------------------------------ */
int nrt_synth_relabel_i32(const int *labels, const float *lut, int lut_len, float *out, long long n, void *stream);
int nrt_synth_intensity_f32(const float *labels, const float *noise, const float *mean, const float *stdv, const float *bg_zero,
                            float *out, int batch, long long nvox, int channels, int nlabels, void *stream);
int nrt_synth_bias_clip_f32(const float *image, const float *bias, float *out, long long n, int channels, float lo, float hi,
                            void *stream);
int nrt_synth_gamma_dc_f32(const float *image, const float *gamma, const float *dc, float *out, int batch, long long nvox,
                           int channels, void *stream);
int nrt_synth_labels_out(const float *idx, const int *lut, int lut_len, int depth, float *onehot, int *out_i32, long long n,
                         void *stream);
/* Stages of labels_to_image_new (neurite/tf/models.py:920-1300) and of the augmentation layers it instantiates:
 *   nrt_synth_axis_mask_f32    x viewed [outer, axis_len, inner]: y = x * mask[a]   (layers.RandomCrop, layers.py:446-519)
 *   nrt_synth_axis_gather_f32  y[o, j, i] = x[o, index[j], i], j < out_len        (layers.Subsample, layers.py:367-443)
 *   nrt_synth_noise_add_f32    y[b, v, c] = x + sd[b * sd_batch_stride + c * sd_channel_stride] * noise   (layers.GaussianNoise, :2305-2403)
 *   nrt_synth_bg_clear_f32     y[b, v, c] = image * (labels[b, v] == 0 && flag[b] ? 0 : 1)   (models.py:1213-1223) */
int nrt_synth_axis_mask_f32(const float *x, const float *mask, float *y, long long outer, int axis_len, long long inner,
                            void *stream);
int nrt_synth_axis_gather_f32(const float *x, const int *index, float *y, long long outer, int axis_len, int out_len,
                              long long inner, void *stream);
int nrt_synth_noise_add_f32(const float *x, const float *noise, const float *sd, float *y, int batch, long long nvox,
                            int channels, int sd_batch_stride, int sd_channel_stride, void *stream);
int nrt_synth_bg_clear_f32(const float *image, const float *labels, const float *flag, float *y, int batch, long long nvox,
                           int channels, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NEURITE_AMD_H */
