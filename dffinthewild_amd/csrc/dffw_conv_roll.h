// conv_roll (rolling-window 3x3x3 conv along the slice axis, dffw_conv_roll.hip): host/device declarations.
#pragma once
#include "dffw_internal.h"

namespace dffw {

struct RollArgs {
    const uint16_t *wroll;   // filter in fragment order [dz*5 + k5][part][64 lanes][8] (pack_conv)
    int tiles_y, tiles_x;    // columns per sample
    int zsplit;              // a sample's slices are walked by zsplit workgroups (contiguous ranges)
    int total_tiles;         // B * zsplit * tiles_y * tiles_x
    int wgs;                 // workgroups to launch (0: two per CU)
    int pair;                // filter packed for the pixel-pair kernel (<= 8 output channels)
    const uint16_t *wroll2;  // conv_roll_efd: the pooled branch's filter (null: plain strided conv)
    const float *bias2;      // ... and its BatchNorm shift (added to a.bias in the accumulator init)
};

constexpr int ROLL_CHUNKS = 15;        // 3 slices x 5 chunks of (2 in-slice taps x 16 channels)
constexpr int ROLL_CHUNKS_PAIR = 18;   // pixel-pair form: 3 slices x 3 filter rows x 2 halves of (2 input columns x 16 channels)

void roll_tile(int *ty, int *tx);   // column footprint of the instantiated kernel
hipError_t launch_conv_roll(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s);
// the launch's epilogue is one the straight-line routine covers (epilogue_lean, dffw_device.h) -> the LEAN instantiation runs; `res_variant`:
// the kernel family has an instantiation that prefetches a residual
bool roll_lean(int prec, const ConvArgs &a, bool res_variant);
void conv_roll_kernel_name(int prec, const ConvArgs &a, bool pair, char *buf, int n);   // the instantiation launch_conv_roll picks for `a`
// conv_rollx (dffw_conv_rollx.hip): the software-pipelined step (epilogue of step n-1 and fill of slice n+5 inside the contraction of step n,
// buffer-addressed fills) for the pair-form layers in split-bf16 storage; launch_conv_roll / conv_roll_kernel_name route to it when it applies
bool rollx_pair_ok(int prec, const ConvArgs &a, bool pair);
hipError_t launch_conv_rollx_pair(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_rollx_pair_kernel_name(const ConvArgs &a, char *buf, int n);
// ... and for 32 -> 16 channels with the contraction split over the two 16-channel input halves (8 waves, columns of 8 x 16; filter packed as
// [half][ROLL_CHUNKS chunks][part][64 lanes][8], chunk order as conv_roll's plain form)
bool rollx_k2_ok(int prec, const ConvArgs &a);
hipError_t launch_conv_rollx_k2(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_rollx_k2_kernel_name(const ConvArgs &a, char *buf, int n);
// conv_rollk (dffw_conv_rollk.hip, round 5): 3x3x3 stride 1 over 32 / 64 input channels, 32 output channels per launch (RollArgs::pair = first
// 16-channel output tile of the launch), the contraction split over the workgroup's waves: wave w = (16-channel group w >> 1, tap half w & 1)
// holds ROLLK_CHUNKS chunks of (2 taps x 16 channels) for two output tiles.  Filter packed as [32-channel output pair][wave][chunk][output
// tile][part][64 lanes][8]; columns of 8 x 8 output pixels
#define DFFW_ROLLK_TY 8
#define DFFW_ROLLK_TX 8
constexpr int ROLLK_CHUNKS = 7;
void rollk_tile(int *ty, int *tx);
int rollk_waves(int prec, const ConvArgs &a);   // waves per workgroup (= input channels / 8) when the kernel covers the launch, else 0
hipError_t launch_conv_rollk(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_rollk_kernel_name(const ConvArgs &a, char *buf, int n);
// conv_slice32 (dffw_conv_slice.hip, round 5): per-slice 1x3x3, 32 -> 32 channels, every wave holds the whole filter: SLICE32_CHUNKS chunks of (one tap x
// 32 channels) for two output tiles, packed as [chunk][output tile][part][64 lanes][8]; columns of 8 x 16 output pixels, all slices of a sample
#define DFFW_SLICE_TY 8
#define DFFW_SLICE_TX 16
constexpr int SLICE32_CHUNKS = 9;
void slice32_tile(int *ty, int *tx);
bool slice32_ok(int prec, const ConvArgs &a);
hipError_t launch_conv_slice32(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_slice32_kernel_name(const ConvArgs &a, char *buf, int n);
// conv_slice64: the same for 64 -> 64 channels, a wave holds ONE 16-channel output tile's filter: SLICE64_CHUNKS chunks of (one tap x 32 channels); weights
// [output tile 4][chunk = tap * 2 + channel half][part][64 lanes][8]
constexpr int SLICE64_CHUNKS = 18;
// ... and its HEAD variant (cin = 32 features + 2 flow channels in 40-channel records, slice-broadcast residual): 9 feature chunks + 3 chunks over the fifth
// channel octet (K octet g of chunk 9 + k = tap 4k + g)
constexpr int SLICE64_HEAD_CHUNKS = 12;
// ... and its CAT variant (conv_slice32_cat: 32 -> 32 over t + the block's 1x1x1 shortcut over x as a tenth chunk; weights [output tile 2][chunk][part][64 lanes][8])
constexpr int SLICE32_CAT_CHUNKS = 10;
bool slice64_ok(int prec, const ConvArgs &a);
hipError_t launch_conv_slice64(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_slice64_kernel_name(const ConvArgs &a, char *buf, int n);
// transposed 3x3x3 s(1,2,2), 16 -> 8 channels (tiles are columns of the INPUT grid; filter packed as ROLL_CHUNKS_T chunks)
constexpr int ROLL_CHUNKS_T = 9;
// transposed 3x3x3 s(1,2,2), 32 -> 16 channels, one launch per output row phase py (filter packed per phase: 9 / 18 chunks of one
// tap x 32 channels, phase 1 after phase 0 in one buffer)
constexpr int ROLL_CHUNKS_T32_0 = 9, ROLL_CHUNKS_T32_1 = 18;
void roll_t32_tile(int py, int *ty, int *tx);   // input-grid column of sweep py
hipError_t launch_conv_roll_t32(int prec, int py, const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_roll_t32_kernel_name(int prec, int py, const ConvArgs &a, char *buf, int n);
// fused EFD block / strided 3x3x3 conv, 8 -> 16 channels (tiles are 4 x 16 columns of the OUTPUT grid); both filters packed as
// ROLL_CHUNKS_8 chunks [dz][3 chunks of 4 taps x 8 channels]
constexpr int ROLL_CHUNKS_8 = 9;
void efd_roll_tile(int *ty, int *tx);
hipError_t launch_conv_roll_efd(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_roll_efd_kernel_name(int prec, const ConvArgs &a, bool dual, char *buf, int n);
// strided 3x3x3 over 16 (kh = 1) or 32 (kh = 2) input channels: 16 -> 16 (nt = 1: columns of 4 x 16 output pixels), 16 / 32 -> 32
// (nt = 2: 4 x 8; 64 output channels = two launches with RollArgs::pair = first output tile).  Filter packed per (16-channel output
// tile, 16-channel input half) as ROLL_CHUNKS chunks [dz][5 chunks of 2 in-slice taps x 16 channels]: [output tile][half][chunk]
void s2_roll_tile(int nt, int *ty, int *tx);
hipError_t launch_conv_roll_s2(int prec, int nt, int kh, const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_roll_s2_kernel_name(int prec, int nt, int kh, const ConvArgs &a, char *buf, int n);
hipError_t launch_conv_roll_t(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_roll_t_kernel_name(int prec, const ConvArgs &a, char *buf, int n);

}  // namespace dffw
