// conv_pp (ping-pong LDS-tiled conv, dffw_conv_pp.hip): host declarations.  Uses conv_tile's TileCfg / TileArgs / packing.
#pragma once
#include "dffw_conv_tile.h"

namespace dffw {

bool conv_pp_has(const TileCfg *cfg);   // a ping-pong instantiation of this conv_tile configuration exists
int conv_pp_groups(const TileCfg *cfg); // ... and how many wave groups (units in flight) one of its workgroups holds
// t.grid = workgroups to launch (a multiple of 8, at most one per CU: the kernel is persistent); t.nsplit = output-channel
// slabs per tile (cfg->nt * t.nsplit == t.nt_total); t.ksplit / t.pass_split are not used
hipError_t launch_conv_pp(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s);
void conv_pp_kernel_name(int prec, const TileCfg *cfg, char *buf, int n);

}  // namespace dffw
