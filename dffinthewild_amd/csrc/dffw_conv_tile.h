// LDS-tiled implicit-GEMM convolution (conv_tile): shared declarations for host and device.
#pragma once
#include "dffw_internal.h"

namespace dffw {

// geometry classes of DFF_net's convs that conv_tile covers
enum Geo {
    G3S1 = 0,  // 3x3x3, stride 1, pad 1                      (31 layers, 59 % of FLOPs)
    G3S2 = 1,  // 3x3x3, stride (1,2,2), pad 1                (10 layers)
    G3T = 2,   // transposed 3x3x3 s(1,2,2) p1 op(0,1,1): 4 sub-pixel passes over one input tile (11 layers)
    G2S1 = 3,  // 1x3x3 per-slice conv, pad (0,1,1)           (6 layers)
    G2D = 4,   // the stem: 1x9x9, dilation (1,2,2), pad (0,8,8), on the paired-pixel (W+2)-wide input (see stack_in)
    G2S2 = 5,  // 1x3x3 per-slice conv, stride (1,2,2), pad (0,1,1): the down-sampling blocks of the alignment network
    G2P = 6,   // the stem in pixel-pair form: result rows 0-7 = the 8 channels of pixel x, rows 8-15 = of pixel x+2 (both see the
               // same five paired-pixel records per filter row); reads the fp32 / raw focal stack only (no record volume)
    GEO_COUNT = 6
};

struct GeoInfo {
    int minz, maxz, miny, maxy, minx, maxx;  // tap offset range along slices / rows / cols
    int s;                         // input stride over rows/cols
    int os;                        // output stride (2 for the transposed conv)
    int npass;
};

inline GeoInfo geo_info(int geo) {
    switch (geo) {
        case G3S1: return GeoInfo{-1, 1, -1, 1, -1, 1, 1, 1, 1};
        case G3S2: return GeoInfo{-1, 1, -1, 1, -1, 1, 2, 1, 1};
        case G3T: return GeoInfo{-1, 1, 0, 1, 0, 1, 1, 2, 4};
        case G2D: return GeoInfo{0, 0, -8, 8, -6, 10, 1, 1, 1};
        case G2S2: return GeoInfo{0, 0, -1, 1, -1, 1, 2, 1, 1};
        case G2P: return GeoInfo{0, 0, -8, 8, -6, 10, 1, 1, 1};
        default: return GeoInfo{0, 0, -1, 1, -1, 1, 1, 1, 1};
    }
}

// one instantiated kernel configuration
struct TileCfg {
    int id;
    int geo, nt, cg;     // selection key: geometry class, 16-channel output tiles, channels staged per LDS fill
    int tz, ty, tx;      // output tile (grid points) per workgroup
    // derived LDS image constants (host needs them to precompute tap offsets)
    int fz, fy, fx, fxl;
    int pipe;            // 1: software-pipelined MFMA loop (compute-bound layers), 0: lean loop (bandwidth-bound)
    int nw;              // waves per workgroup (4, or 8 for the wide tiles) -- per TEAM for the team configurations
    int kt = 1;          // teams per workgroup (> 1: the contraction depth split inside the workgroup, see conv_tile's header; nw * kt waves)
};

struct TileArgs {
    int npass, nstage;
    int KC[4];                // 32-deep contraction chunks per stage, per pass
    const int *tab[4];        // [KC*4] per 8-channel group: LDS byte offset of its tap (+ channel octet * 16)
    const uint16_t *wpk[4];   // [stage][KC][NT][part][64][8]
    int ooy[4], oox[4];       // output sub-pixel phase of each pass
    int tiles_z, tiles_y, tiles_x;
    int total_tiles;
    int nt_total;             // 16-channel output tiles of the layer (weights are packed for all of them)
    int nsplit;               // grid.y: output-channel split, each workgroup produces nt_total/nsplit tiles
    int grid;                 // workgroups to launch along x = 8 * ceil(total_tiles / 8) (one tile each)
    int pass_split;           // transposed conv on few tiles: grid.z = 4, each workgroup runs ONE sub-pixel pass (disjoint outputs)
    int ksplit;               // grid.z: split of the contraction depth (channel-group stages) over workgroups, 1 = none
    float *partial;           // ksplit > 1: fp32 partial sums [ksplit][output pixel][nt_total*16], finished by splitk_finish
    int64_t partial_stride;   // elements per split = output pixels * nt_total*16
    int warm;                 // few-tile launches on cold weights: every workgroup first touches the weight lines of its whole contraction walk
};

// returns nullptr when no instantiation covers (geo, nt, cg)
// wide: prefer an 8-wave 640-point instantiation when one exists
const TileCfg *tile_cfg_find(int geo, int nt, int cg, bool wide = false);
// the 4-wave configuration with exactly this block shape, or nullptr
const TileCfg *tile_cfg_find_shape(int geo, int nt, int cg, int tz, int ty, int tx);
// configuration with the same geometry, channel group and TILE SHAPE as `base` but `nt` output tiles (for splits)
const TileCfg *tile_cfg_find_like(const TileCfg *base, int nt);
// the team configuration that replaces a split-K launch of `base` (same pack) over `nstage` stages -- one stage per team -- with `nt` output tiles per
// workgroup, or nullptr
const TileCfg *tile_cfg_find_team(const TileCfg *base, int nstage, int nt);
bool tile_cfg_has_splitk(const TileCfg *c);   // a split-K instantiation of this configuration exists
bool tile_cfg_has_sums(const TileCfg *c);     // a row-sums instantiation (DFFW_ARGS_SUMS: per-row sums through ConvArgs::outf, nothing stored) exists
int tile_cfg_count();
const TileCfg *tile_cfg_at(int i);
hipError_t launch_conv_tile(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s);
bool tile_lean(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t);   // the launch runs the LEAN instantiation (straight-line epilogue)
void conv_tile_kernel_name(int prec, const TileCfg *cfg, bool splitk, bool lean, char *buf, int n);

}  // namespace dffw
