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
// conv_rollt (dffw_conv_rollt.hip, round 6): transposed 3x3x3 s(1,2,2) p1 op(0,1,1) over 32 / 64 input channels, 32 output channels per workgroup row
// (grid.y = 32-channel output half), rolling window over 8 x 8 columns of the INPUT grid.  The filter is resident and split over the workgroup's waves
// by OUTPUT PHASE (py, px) -- 3 / 6 / 6 / 12 of the 27 taps -- and 16-channel output tile; only phase (1,1) at 64 input channels is also split over K
// (between the wave pair A / B, which exchange one partial tile per operand tile through LDS).  A wave's program is a list of operand fragment
// sets (window slice d, row tap dy, column tap dx, 32-channel chunk) and the accumulator slots each set feeds; host packing and kernel share it.
#define DFFW_ROLLT_TY 8
#define DFFW_ROLLT_TX 8
namespace rollt {
enum Role { R_A = 0, R_B = 1, R_C = 2, R_D = 3, R_A32 = 4, R_C32 = 5 };
constexpr int MAXU = 15;   // weight units (one tap x 32 channels x 16 outputs, hi + lo) per wave: the filter buffer's wave stride
template <int ROLE>
struct Prog {
    static constexpr bool AB = ROLE == R_A || ROLE == R_B || ROLE == R_A32;   // phases (1,1) [slot 0] + (0,0) [slot 1]
    static constexpr bool XCH = ROLE == R_A || ROLE == R_B;                   // K-split pair: A sends slot 0, B sends slot 1
    static constexpr int SEND = ROLE == R_A ? 0 : ROLE == R_B ? 1 : -1;
    static constexpr int NS = ROLE == R_C32 ? 9 : 12;                         // operand fragment sets per operand-tile pair
    static constexpr int NACC = (ROLE == R_C || ROLE == R_D) ? 1 : 2;         // accumulator slots
    static constexpr int NOWN = XCH ? 1 : NACC;                               // ... whose result this wave finishes
    static constexpr int own(int k) { return XCH ? 1 - SEND : k; }            // k-th own slot
    static constexpr int d(int i) { return ROLE == R_C32 ? i / 3 : i / 4; }   // window slice 0..2 = input slice z - 1 + d = filter slice 2 - d
    static constexpr int dy(int i) { return AB ? (i % 4) / 2 : ROLE == R_C ? 0 : ROLE == R_D ? i % 2 : (i % 3 == 2 ? 1 : 0); }
    static constexpr int dx(int i) { return AB ? i % 2 : ROLE == R_C ? i % 2 : ROLE == R_D ? 0 : (i % 3 == 1 ? 1 : 0); }
    static constexpr int chunk(int i) { return ROLE == R_B ? 1 : (ROLE == R_C || ROLE == R_D) ? (i % 4) / 2 : 0; }
    static constexpr int feeds(int i) { return AB ? (i % 4 == 0 ? 3 : 1) : ROLE == R_C32 ? (i % 3 == 0 ? 3 : i % 3 == 1 ? 1 : 2) : 1; }   // bit mask of slots
    static constexpr int phase(int slot) { return AB ? (slot == 0 ? 3 : 0) : ROLE == R_C ? 1 : ROLE == R_D ? 2 : (slot == 0 ? 1 : 2); }   // py * 2 + px
    static constexpr int nfeed(int i) { return (feeds(i) & 1) + (feeds(i) >> 1); }
    static constexpr int ubase(int i) { return i == 0 ? 0 : ubase(i - 1) + nfeed(i - 1); }   // first weight unit of set i (one per fed slot, slot order)
    static constexpr int NU = ubase(NS);
    static_assert(NU <= MAXU, "filter share");
};
// filter tap (ky or kx) of output phase bit p at input offset dd (p = 0: offset 0 only, tap 1; p = 1: offset 0 -> tap 2, offset 1 -> tap 0)
constexpr int tap_of(int p, int dd) { return p == 0 ? 1 : (dd ? 0 : 2); }
}   // namespace rollt
// wave w of an 8-wave workgroup (64 input channels): output tile (w >> 1) & 1, role (w >> 2) * 2 + (w & 1) = A, B | C, D; of a 4-wave one (32): tile w >> 1, role A32 / C32
// (32 -> 16 channels, the wide form: eight waves = roles A32 / C32 x four pixel sub-blocks; the filter buffer holds the two roles' shares once)
inline int rollt_role(int cin, int wave) { return cin == 64 ? (wave >> 2) * 2 + (wave & 1) : (wave & 1 ? rollt::R_C32 : rollt::R_A32); }
bool rollt_ok(int prec, const ConvArgs &a);   // the kernel covers the launch (a.Ng/Hg/Wg = input grid)
void rollt_tile(int cout, int *ty, int *tx);  // input-grid column of the instantiation that serves `cout` output channels (16: the wide form, 8 x 16)
hipError_t launch_conv_rollt(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_rollt_kernel_name(const ConvArgs &a, char *buf, int n);
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
// conv_efd16 (dffw_conv_efd16.hip, round 6): the fused EFD block of the 16-channel stage, 16 -> 32 channels on 8 x 8 columns of the OUTPUT grid: a.in0 = x, a.in1 = its
// (1,2,2) max-pool; t.wroll / t.wroll2 = the strided / pooled branch's filter in conv_roll_s2's order, a.bias / t.bias2 their BatchNorm shifts
void efd16_tile(int *ty, int *tx);
bool efd16_ok(int prec, const ConvArgs &a, const RollArgs &t);
hipError_t launch_conv_efd16(const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_efd16_kernel_name(char *buf, int n);
// strided 3x3x3 over 16 (kh = 1) or 32 (kh = 2) input channels: 16 -> 16 (nt = 1: columns of 4 x 16 output pixels), 16 / 32 -> 32
// (nt = 2: 4 x 8; 64 output channels = two launches with RollArgs::pair = first output tile).  Filter packed per (16-channel output
// tile, 16-channel input half) as ROLL_CHUNKS chunks [dz][5 chunks of 2 in-slice taps x 16 channels]: [output tile][half][chunk]
void s2_roll_tile(int nt, int *ty, int *tx);
hipError_t launch_conv_roll_s2(int prec, int nt, int kh, const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_roll_s2_kernel_name(int prec, int nt, int kh, const ConvArgs &a, char *buf, int n);
hipError_t launch_conv_roll_t(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s);
void conv_roll_t_kernel_name(int prec, const ConvArgs &a, char *buf, int n);

}  // namespace dffw
