// fused.hip (fused_wc.h) <-> interpn.hip: the wave-private LDS row cache kernel as a stand-alone warp (nrt_interpn_f32 variant 10)
#pragma once

// args: the InterpArgs of the call (interpn_core.h).  32 float32 channels, 3-D, linear, volumes large enough for the x-march schedule
bool nrt_wc_interpn_supported(const void *args, int batch);
int nrt_wc_interpn_launch(const void *args, int batch, int mode, void *stream);
