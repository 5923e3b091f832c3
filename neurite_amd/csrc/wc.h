// fused.hip (fused_wc.h) <-> interpn.hip: the wave-private LDS row cache kernel as a stand-alone warp (nrt_interpn_f32 variant 10)
#pragma once

// args: the InterpArgs of the call (interpn_core.h).  32 float32 channels, 3-D, linear, volumes large enough for the x-march schedule
bool nrt_wc_interpn_supported(const void *args, int batch);
int nrt_wc_interpn_launch(const void *args, int batch, int mode, void *stream);

// d loss / d loc on the same gather, where nrt_wc_interpn_supported holds and the locations are given (absolute or shift).
// sums != nullptr: `rows` is the fixed map, the gradient of the soft Dice wrt the warped map is formed from the forward's sums and
// grad_dice (nrt_warp_dice_bwd_f32); sums == nullptr: `rows` is the gradient arriving at the warped map (nrt_interpn_bwd_f32's grad_loc)
int nrt_wc_bwd_launch(const void *args, int batch, int mode, const float *rows, const float *sums, const float *grad_dice, float eps,
                      float *grad_loc, void *stream);
