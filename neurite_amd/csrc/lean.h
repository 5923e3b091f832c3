// interpn_lean.hip <-> interpn.hip: the instruction-lean few-channel linear kernel (variant 8)
#pragma once

// true when interpn_lean can run this call (3-D, 1..4 channels, 16-byte aligned tensors, sizes below the 32-bit / 24-bit
// limits of its address arithmetic)
bool nrt_lean_supported(const int *vol_shape, const int *out_shape, int channels, int ndim, const void *vol, const void *loc,
                        const void *out, long long vol_bs, long long loc_bs);
// args: the InterpArgs of the call (interpn_core.h); method_kind 0 linear, 1 nearest (float32 data), 2 nearest (int32 data; per-voxel
// locations only); form: which kernel takes per-voxel locations with linear interpolation -- 0 the library's choice (the tile form), or
// one of the two by name (interpn variants 8 / 11: the A/B partners, bit-identical)
enum { NRT_LEAN_FORM_AUTO = 0, NRT_LEAN_FORM_TILE = 1, NRT_LEAN_FORM_BOX = 2 };
int nrt_lean_launch(const void *args, int batch, int mode, int method_kind, void *stream, int form = NRT_LEAN_FORM_AUTO);

