// Internal declarations shared by the graph executor (dffw_engine.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// thread-local error message of the C ABI (dffw_last_error); returns `code`
int dffw_fail(int code, const char *fmt, ...);

namespace dffw {

// ---- activation storage ---------------------------------------------------------------------
// Every intermediate volume lives in HBM channels-last, one "pixel" (b, slice, y, x) at a time:
//     act[pixel][part][channel]   16-bit elements, part in {hi, lo} for split-bf16, {hi} otherwise
// so the C channels that form the contraction axis of the next conv are contiguous (16-byte
// MFMA operand loads) and one pixel of an 8-channel split-bf16 volume is a single 32-byte line.
struct Act {
    uint16_t *p = nullptr;
    int B = 0, N = 0, H = 0, W = 0, C = 0;
    int64_t pixels() const { return (int64_t)B * N * H * W; }
};

enum Prec { P_BF16X3 = 0, P_FP16 = 1, P_BF16 = 2 };
inline int prec_parts(int prec) { return prec == P_BF16X3 ? 2 : 1; }

// One entry per group of 8 contraction indices (8 consecutive input channels of one filter tap).
struct TapEntry {
    int dz, dy, dx;  // input offset of the tap relative to the (strided) output coordinate
    int coff;        // first channel of the group inside the (virtually concatenated) input
};
static_assert(sizeof(TapEntry) == 16, "TapEntry must be 16 bytes");

#define DFFW_TAP_INVALID (1 << 20)

// A raw focal stack as the reference's loaders hold it before `FS/127.5 - 1.0` (dffw_forward_raw): uint8 or fp32
// 0..255, any layout, described by element strides; rows >= h / cols >= w of the (padded) stack read as -1.
struct RawStack {
    const void *p;                  // null: not used
    int dtype;                      // DFFW_RAW_U8 / DFFW_RAW_F32, | DFFW_RAW_NORM_F64 (float64 normalisation of the FS6 loader)
    int64_t sb, sn, sy, sx, sc;     // element strides: sample, slice, row, col, colour channel
    int h, w;                       // rows / cols present in the source
};

#define DFFW_RAW_NORM_F64_BIT 16   // == DFFW_RAW_NORM_F64 of include/dffw.h (device code does not include that header)
#define DFFW_ARGS_RAW 8
#define DFFW_ARGS_SUMS 128
// kernel-path switches of the forward's snapshot (Switches, dffw_engine.cpp) that the launchers consult: carried in ConvArgs::dbg so that the
// kernel name reported and the instantiation launched cannot disagree (no getenv at launch time)
#define DFFW_ARGS_NO_LEAN_TILE 16   // DFFW_NO_LEAN_TILE: conv_tile's generic epilogue instead of its LEAN instantiations
#define DFFW_ARGS_NO_LEAN_ROLL 64   // DFFW_NO_LEAN_ROLL: the same for the rolling-window kernels
#define DFFW_ARGS_NO_ROLLX 256   // DFFW_NO_ROLLX: conv_roll's serial step instead of conv_rollx's pipelined one
#define DFFW_ARGS_NO_SLICE32 1024 // DFFW_NO_SLICE32: conv_tile instead of the streaming per-slice kernel conv_slice32 (1x3x3, 32 -> 32 channels)
#define DFFW_ARGS_NO_ROLLT 2048  // DFFW_NO_ROLLT: conv_tile instead of the streaming transposed-conv kernel conv_rollt (32 / 64 -> 32 / 64 channels)
#define DFFW_ARGS_NO_ROLLK 512   // DFFW_NO_ROLLK: conv_tile instead of the K-split rolling window conv_rollk (32 / 64 -> 32 / 64 channels)
// (ConvArgs must not grow: the register allocation of the lean transposed-conv kernels is sensitive to its size, a
// 56-byte larger argument block cost them 38 %)
struct ConvArgs {
    const uint16_t *in0, *in1;  // in1: second half of a virtual channel concat (C1 = 0: none)
    int C0, C1;
    int B, Ni, Hi, Wi;          // input volume
    int Ng, Hg, Wg;             // logical grid whose points are the GEMM columns (B*Ng*Hg*Wg of them)
    int sy, sx;                 // input row/col = grid row/col * s + tap offset
    int No, Ho, Wo;             // output volume
    int osy, osx, ooy, oox;     // output row/col = grid row/col * os + oo (sub-pixel phases of convT)
    int Cout, KC;               // KC = number of 32-deep contraction chunks
    const TapEntry *tab;        // [KC*4]
    const uint16_t *wpk;        // weights in MFMA fragment order [KC][NT][part][64 lanes][8]
    const float *bias;          // [NT*16] folded BatchNorm shift (zero padded)
    const uint16_t *res0, *res1;// residual volumes in the output's geometry, or null
    int res_bcast;              // res0 holds ONE slice per sample (B,1,Ho,Wo,Cout) and is added to every output slice
    uint16_t *out;              // output volume, or null
    uint16_t *out_pre;          // optional second output: value before the residual add
    float *outf;                // fp32 planar output (B,outf_ch,No,Ho,Wo) instead of an activation volume, or null
    int outf_ch;                // 1: score volume of a Cout == 1 layer; 3: warp parameters of the alignment heads
    int64_t outf_plane;         // No*Ho*Wo (used when outf_ch > 1)
    const float *cls_w;         // fused 1x1x1 classifier (DEN.py:51-55): Cout fp32 weights applied to the final value, or null
    float *cls_out;             // its fp32 score volume (B,No,Ho,Wo)
    int relu;                   // 0: none   1: relu(acc+res)   2: relu(acc)+res
    int dbg;                    // ablation switches for profiling (0 in production): 1 no fill, 2 no MFMA loop, 4 no stores;
                                // bit 3 (DFFW_ARGS_RAW): fs32 points to a RawStack in device memory instead of the fp32 stack
    const uint16_t *zero;       // >= 16 zero bytes in device memory (source of out-of-volume LDS-DMA lanes)
    const float *fs32;          // stem only: the fp32 planar focal stack (B,3,N,H,Wi-2); when set the kernel builds its paired-pixel
                                // records on the fly instead of reading a materialised volume through in0
    unsigned long long *trace;  // debug (DFFW_TRACE_LAYER): 8 x u64 per tile = s_memtime at phase boundaries + HW_ID, or null
    int64_t M;                  // B*Ng*Hg*Wg
};

// launchers (dffw_kernels.hip)
int conv_nt_for(int cout);  // 16-channel output tiles the conv kernel picked for `cout` iterates over
hipError_t launch_conv(int prec, const ConvArgs &a, hipStream_t s);
void conv_kernel_name(int prec, int cout, char *buf, int n);  // name of the instantiation launch_conv picks
void conv_kernel_name_for(int prec, const ConvArgs &a, char *buf, int n);   // name of the kernel launch_conv picks for `a` (conv_small for small grids)
hipError_t launch_set_raw(const RawStack &rs, RawStack *dst, hipStream_t s);   // writes the descriptor into device memory (enqueue-only)
hipError_t launch_stack_in(int prec, const float *FS, uint16_t *out, int B, int N, int H, int W, hipStream_t s);
hipError_t launch_from_ncdhw(int prec, const float *x, uint16_t *out, int B, int C, int N, int H, int W, hipStream_t s);
hipError_t launch_to_ncdhw(int prec, const uint16_t *x, float *out, int B, int C, int N, int H, int W, hipStream_t s);
hipError_t launch_pool(int prec, int mode, int k, const uint16_t *x, uint16_t *out, int B, int N, int H, int W, int C, hipStream_t s);
// the three average pools (1,2,2), (1,4,4), (1,8,8) of one volume in one pass, each bit-identical to launch_pool's (H, W multiples of 8)
hipError_t launch_pool3(int prec, const uint16_t *x, uint16_t *o2, uint16_t *o4, uint16_t *o8, int B, int N, int H, int W, int C, hipStream_t s);
bool srd_attention_supported(int C);
// pooled: optional (B,N,H/2,W/2,C) volume receiving the (1,2,2) max-pool of the result, or null
hipError_t launch_srd_attention(int prec, const uint16_t *feat, uint16_t *out, const float *w3, const float *w1, int B, int N,
                                int H, int W, int C, uint16_t *pooled, hipStream_t s);
hipError_t launch_splitk_finish(int prec, const float *partial, int ksplit, int64_t stride, int64_t M, int cpad, int Cout,
                                const float *bias, const uint16_t *res0, int relu, uint16_t *out, hipStream_t s);
hipError_t launch_from_ncdhw_pad(int prec, const float *x, uint16_t *out, int B, int Cs, int C, int N, int H, int W, hipStream_t s);
// mode 0: [warp(fe)[last slice] | warp(fe)[slice n] | flow | pad] (2C+8 channels, N slices); mode 1: [warp(fe)[slice n] | flow | pad]
// (C+8 channels, N slices); mode 2: warp(fe)[last slice] alone (C channels, ONE slice per sample)
hipError_t launch_flow_volume(int prec, const uint16_t *fe, uint16_t *out, const float *alpha, const float *fov, int B, int N,
                              int H, int W, int C, int mode, hipStream_t s);
hipError_t launch_alpha_mean(const float *head, float *alpha, float *raw, int B, int N, int64_t hw, hipStream_t s);
// alpha head tail (last conv + plane mean as plane sums of the head's last activation volume v (B,N,H,W,C)); partial: device scratch of
// B*N*(head_tail_chunks()+4)*C doubles; w: PackedConv::whead
int head_tail_chunks(int B, int N, int64_t hw);
// ... from the per-tile vectors of of_roll_kernel<.., SUMS> (tsum: B*N*tiles_y*tiles_x*18*C floats)
// ... from the per-row-segment vectors of conv_tile's row-sums variant (rows: B*N*H*tiles_x*3*C floats)
hipError_t launch_head_tail_rows(const float *rows, double *seg, int tiles_x, const float *w, float *alpha, float *raw, int B, int N, int H, int W,
                                 int C, hipStream_t s);
// seg: device scratch of head_tail_tiles_scratch_bytes()
int64_t head_tail_tiles_scratch_bytes(int B, int N, int C);
hipError_t launch_head_tail_tiles(const float *tsum, double *seg, int tiles_y, int tiles_x, const float *w, float *alpha, float *raw, int B,
                                  int N, int H, int W, int C, hipStream_t s);
hipError_t launch_head_tail(int prec, const uint16_t *v, double *partial, const float *w, float *alpha, float *raw, int B, int N, int H, int W,
                            int C, hipStream_t s);
hipError_t launch_fov_warp(const float *x, const float *alpha, const float *fov, float *out, float *flow, int B, int C, int N,
                           int H, int W, int alpha_from_sample0, hipStream_t s);
// regression heads (DEN.py:83-90, 110-126) that share the output size and the focus distances: score volumes (B,N,h_k,w_k) -> depth maps (B,H,W)
struct RegressHeads {
    const float *score[4];
    float *depth[4];
    int h[4], w[4];
    int n;
    int nofuse;   // DFFW_NO_REGRESS_FUSED (the forward's switch snapshot): one workgroup row per head (regress_kernel) instead of regress_fused_kernel
};
hipError_t launch_regress_heads(const RegressHeads &hd, int B, int N, int H, int W, const float *fd, int64_t fsb, int64_t fsn, int64_t fsh,
                                int64_t fsw, hipStream_t s);
hipError_t launch_regress(const float *score, int B, int N, int h, int w, int H, int W, const float *fd,
                          int64_t fsb, int64_t fsn, int64_t fsh, int64_t fsw, float *depth, hipStream_t s);

}  // namespace dffw
