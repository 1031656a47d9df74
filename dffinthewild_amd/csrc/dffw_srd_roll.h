// srd_roll (fused SRD block of the 8-channel full-resolution stage, dffw_srd_roll.hip): host/device declarations.
#pragma once
#include "dffw_internal.h"

namespace dffw {

struct SrdArgs {
    const uint16_t *x;        // block input (B,N,H,W,8) in storage format
    uint16_t *out;            // block output, same shape
    uint16_t *pooled;         // (B,N,H/2,W/2,8) max-pool (1,2,2) of out, or null
    const uint16_t *w0, *w2;  // conv.0 / conv.2 filters as MFMA A-fragments [3 chunks][part][64 lanes][8] in pixel-pair form (chunk = filter row, pack_conv)
    const float *b0, *b2;     // their BatchNorm shifts (>= 16 floats, zero padded)
    const float *w3, *w1;     // attention weights fp32 [kz][ci][co] and [ci][co] (unused by the MFMA form, kept for reference)
    const uint16_t *w3f, *w1f;   // the same as MFMA A-fragments: conv3x1x1 [2 chunks][part][64][8], conv1x1x1 [part][64][8] (pack_conv)
    const uint16_t *zero;     // >= 16 zero bytes (out-of-image LDS-DMA lanes)
    int B, N, H, W;
    int tiles_y, tiles_x, total_tiles;   // 8 x 16 columns per sample, B * tiles_y * tiles_x
    int wgs;                  // workgroups to launch (0: three per CU)
#ifdef DFFW_TRACE_BUILD
    unsigned long long *trace;   // debug step timeline (make TRACE=1, DFFW_TRACE_LAYER); the field exists in trace builds only (a larger
                                 // argument block changes the kernels' register allocation)
#endif
};

constexpr int SRD_CHUNKS = 3;
void srd_roll_tile(int *ty, int *tx);
hipError_t launch_srd_roll(int prec, const SrdArgs &a, hipStream_t s);
void srd_roll_kernel_name(int prec, bool pool, char *buf, int n);
// the 16-channel block (columns of 4 x 16 pixels; 1x3x3 filters as 5 chunks of 2 taps x 16 channels)
constexpr int SRD16_CHUNKS = 5;
void srd_roll16_tile(int *ty, int *tx);
hipError_t launch_srd_roll16(int prec, const SrdArgs &a, hipStream_t s);
void srd_roll16_kernel_name(int prec, bool pool, char *buf, int n);
// srd_pipe16 (round 6): the same block as a software pipeline over the slice stream (stage A of position p, B of p - 1, C of p - 2 in one step, one barrier)
hipError_t launch_srd_pipe16(int prec, const SrdArgs &a, hipStream_t s);
void srd_pipe16_kernel_name(int prec, bool pool, char *buf, int n);

// a stride-1 residual block of the alignment network (8 or 16 -> 16 channels, columns of 8 x 16 pixels): a.w0 = conv.0 as 3 (8
// input channels: 4 taps per chunk) or 5 chunks, a.w2 = conv.2 as 5 chunks + 1 shortcut chunk (pack_conv); a.b0 / a.b2 their shifts
constexpr int OF_CHUNKS_B = 6;
// sums: the block's output is not stored, a.out receives per (slice, column) 18 fp32 16-channel vectors (per wave: sum / first column / last column of its
// two rows; first / last row; four corners) for head_tail_finish_tiles (dffw_kernels.hip); 16 input channels only
hipError_t launch_of_roll(int prec, bool cin8, const SrdArgs &a, hipStream_t s, bool sums = false);
void of_roll_kernel_name(int prec, bool cin8, char *buf, int n, bool sums = false);
// of_first: of_roll8's block with its input taken from the planar fp32 focal stack (B,3,N,H,W) = a.w3 (a.x unused); filters as for of_roll8
hipError_t launch_of_first(int prec, const SrdArgs &a, hipStream_t s);
void of_first_kernel_name(int prec, char *buf, int n);
// of_s2: the down-sampling residual block 8 -> 16 channels of the alignment network as one kernel: a.x = block input (B,N,2H,2W,8),
// a.out (B,N,H,W,16) (a.H, a.W = OUTPUT size, columns of 8 x 16 output pixels); a.w0 = conv.0 (1x3x3 stride 2, 8 -> 16) as 3 chunks
// (K octet g of chunk k = tap 4k + g), a.w2 = conv.2 (16 -> 16) in srd_roll16's order, a.w3f = the 1x1x1 shortcut as one chunk (K octet
// 0 = its 8 input channels), a.b0 / a.b2 the BatchNorm shifts (pack_conv)
hipError_t launch_of_s2(int prec, const SrdArgs &a, hipStream_t s);
void of_s2_kernel_name(int prec, char *buf, int n);
// head_warp: first conv of the level-1 / level-2 alignment head over the FOV-warped CF-channel features (+ flow), CF = 8 / 16, the warp
// done while staging
struct HeadWarpArgs {
    const uint16_t *fe;       // level features (B,N,H,W,CF) in storage format
    const uint16_t *ref;      // reference part (B,1,H,W,2 CF): conv#ref of the warped reference slice (BatchNorm scale, no shift)
    uint16_t *out;            // (B,N,H,W,2 CF)
    const uint16_t *w;        // the [cur (CF) | flow (2)] filter as head_warp_chunks(CF) chunks [output tile][part][64 lanes][8]: K octet g of
                              // chunk k = o = 4k + g -> (tap o / OCT, channel octet o % OCT), OCT = CF / 8 + 1 (pack_conv; for CF = 8 this
                              // is srd_roll16's order)
    const float *bias;        // BatchNorm shift (2 CF floats)
    const float *alpha, *fov; // warp parameters so far (B,3,N), fields of view (B,N)
    int B, N, H, W;
    int tiles_y, tiles_x, total_tiles;   // 8 x 16 columns
    int wgs;                  // workgroups to launch (0: default)
};
constexpr int head_warp_chunks(int cf) { return (9 * (cf / 8 + 1) + 3) / 4; }
constexpr int head_warp_max_planes() { return 320; }   // B N: the kernel keeps every plane's three warp parameters in LDS (launch_head_warp rejects more)
hipError_t launch_head_warp(int prec, int cf, const HeadWarpArgs &a, hipStream_t s);
void head_warp_kernel_name(int prec, int cf, char *buf, int n);
// ... and the 8 -> 8 channel blocks (pixel-pair form): a.w0 = conv.0 as 3 pair-form chunks, a.w2 = conv.2 as 3 chunks + 1 shortcut chunk
hipError_t launch_of_roll8(int prec, const SrdArgs &a, hipStream_t s);
void of_roll8_kernel_name(int prec, char *buf, int n);
// the attention tail of the 32-channel block on the matrix cores (no LDS; W % 16 == 0).  w3f: [3 slices][2 output tiles][part][64][8],
// w1f: [2 channel chunks][fragment][2 output tiles][64][8] (pack_conv)
hipError_t launch_srd_attention_mfma(int prec, const uint16_t *feat, uint16_t *out, const uint16_t *w3f, const uint16_t *w1f, int B, int N,
                                     int H, int W, hipStream_t s);

}  // namespace dffw
