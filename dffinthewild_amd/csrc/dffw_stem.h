// stem_pipe (dffw_stem.hip): the pixel-pair stem as a persistent, software-pipelined kernel; host declarations.
#pragma once
#include "dffw_conv_tile.h"

namespace dffw {

// the launch is one stem_pipe serves: split-bf16 storage, the 32 x 32 pair-form tile configuration, fp32 stack source, whole tiles,
// epilogue out = [relu](acc)
bool stem_pipe_ok(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t);
hipError_t launch_stem_pipe(const ConvArgs &a, const TileArgs &t, int wgs, hipStream_t s);   // wgs: persistent grid size (0: two per CU)
void stem_pipe_kernel_name(const ConvArgs &a, char *buf, int n);

}  // namespace dffw
