// gfx950 (MI355X, CDNA4) kernels of the depth-from-focus engine.  wave = 64 lanes everywhere.
//
//  conv_igemm    implicit-GEMM convolution on the matrix cores (v_mfma_f32_16x16x32_{bf16,f16}):
//                D[cout][pixel] += W[cout][k] * X[k][pixel], k = (filter tap, input channel).
//                Covers every conv of DFF_net (1x9x9 dilated, 1x3x3, 3x1x1, 1x1x1, 3x3x3 at stride
//                1 and (1,2,2), and the 4 sub-pixel phases of the transposed 3x3x3) through a tap
//                table; BatchNorm shift, up to two residual adds and ReLU are fused in the epilogue.
//  pool          (1,k,k) max / average pooling on channels-last volumes.
//  stack_in      focal stack (B,3,N,H,W) fp32 planar -> channels-last 8-channel volume.
//  regress       bilinear resize + softplus normalisation + focus-distance expectation.
#include <cstdio>
#include <cstdlib>

#include "dffw_device.h"
#include "dffw_internal.h"

namespace dffw {

// ---- implicit-GEMM convolution -----------------------------------------------------------------
// Workgroup = 4 waves; each wave owns MT*16 consecutive grid points (GEMM columns) and all
// NT*16 output channels (GEMM rows).  Operand fragments for v_mfma_f32_16x16x32: lane l supplies
// row/column (l & 15) and the 8 contraction indices of group (l >> 4); the result registers hold
// D[cout = (l>>4)*4 + i][pixel = l & 15], i.e. 4 consecutive channels of one pixel per lane, which
// is exactly an 8-byte channels-last store.
template <int PREC, int NT, int MT>
__global__ __launch_bounds__(256) void conv_igemm(const ConvArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int64_t pos0 = ((int64_t)blockIdx.x * 4 + wave) * (MT * 16);
    if (pos0 >= a.M) return;

    const int ps0 = PARTS * a.C0, ps1 = PARTS * a.C1;  // pixel stride in elements
    const int64_t samp0 = (int64_t)a.Ni * a.Hi * a.Wi * ps0;
    const int64_t samp1 = (int64_t)a.Ni * a.Hi * a.Wi * ps1;

    int pn[MT], py[MT], px[MT], pbase[MT];
    bool pv[MT];
    int64_t opix[MT];
    const uint16_t *s0[MT];
    const uint16_t *s1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t p = pos0 + mt * 16 + r;
        pv[mt] = p < a.M;
        const int64_t pp = pv[mt] ? p : 0;
        const int x = (int)(pp % a.Wg);
        int64_t t = pp / a.Wg;
        const int y = (int)(t % a.Hg);
        t /= a.Hg;
        const int n = (int)(t % a.Ng);
        const int b = (int)(t / a.Ng);
        pn[mt] = n;
        py[mt] = y * a.sy;
        px[mt] = x * a.sx;
        pbase[mt] = (n * a.Hi + py[mt]) * a.Wi + px[mt];
        s0[mt] = a.in0 + b * samp0;
        s1[mt] = a.in1 + b * samp1;
        opix[mt] = (((int64_t)b * a.No + n) * a.Ho + (y * a.osy + a.ooy)) * a.Wo + (x * a.osx + a.oox);
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const short8 *wbase = reinterpret_cast<const short8 *>(a.wpk) + lane;
    const short8 zero8 = short8{0, 0, 0, 0, 0, 0, 0, 0};

    for (int kc = 0; kc < a.KC; ++kc) {
        const TapEntry te = a.tab[kc * 4 + g];
        const bool tvalid = te.coff >= 0;
        const bool second = te.coff >= a.C0;
        const int cc = tvalid ? (second ? te.coff - a.C0 : te.coff) : 0;
        const int ps = second ? ps1 : ps0;
        const int csrc = second ? a.C1 : a.C0;
        const int delta = (te.dz * a.Hi + te.dy) * a.Wi + te.dx;

        short8 wf[NT][PARTS];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt)
                wf[nt][pt] = wbase[((int64_t)kc * NT * PARTS + nt * PARTS + pt) * 64];

#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int iz = pn[mt] + te.dz, iy = py[mt] + te.dy, ix = px[mt] + te.dx;
            const bool ok = pv[mt] && tvalid && (unsigned)iz < (unsigned)a.Ni && (unsigned)iy < (unsigned)a.Hi &&
                            (unsigned)ix < (unsigned)a.Wi;
            const uint16_t *src = (second ? s1[mt] : s0[mt]) + ((int64_t)(pbase[mt] + delta) * ps + cc);
            short8 xh = zero8, xl = zero8;
            if (ok) {
                xh = *reinterpret_cast<const short8 *>(src);
                if constexpr (PARTS == 2) xl = *reinterpret_cast<const short8 *>(src + csrc);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (PARTS == 2) {
                    acc[nt][mt] = mma<F16>(wf[nt][1], xh, acc[nt][mt]);  // w_lo * x_hi
                    acc[nt][mt] = mma<F16>(wf[nt][0], xl, acc[nt][mt]);  // w_hi * x_lo
                }
                acc[nt][mt] = mma<F16>(wf[nt][0], xh, acc[nt][mt]);      // w_hi * x_hi
            }
        }
    }

    // ---- epilogue (shared with conv_tile, see dffw_device.h) ------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float cls = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) epilogue_quad<PREC, false>(a, acc[nt][mt], nt, g, opix[mt], pv[mt], cls, uint4{}, uint4{});
        epilogue_cls(a, cls, g, opix[mt], pv[mt]);
    }
}

// ---- conv_small: the same implicit GEMM for SMALL grids (the 1/16 .. 1/32-resolution pyramid at batch 1: a few hundred to a
// few thousand grid points, 64 .. 128 output channels, contraction depth up to 27 x 192) ------------------------------------------
// conv_igemm gives a wave 64 grid points, ALL output channels and the whole contraction: a 5 x 7 x 7 layer is ONE workgroup walking
// 54 chunks with a dependent gather in each (86 us for 54 MFLOP).  Here a workgroup owns ONE operand tile (16 grid points) and ONE
// 16-channel output tile; its KS waves split the contraction depth between them, each wave requests ALL operand and weight
// fragments of its share up front (tap entries first, then every gather in flight at once: one L2 round trip instead of one per
// chunk), contracts, and the partial tiles are summed through LDS by wave 0, which runs the shared epilogue.  Grid = operand
// tiles x output tiles, so even a 245-point layer spreads over 64 workgroups x KS waves.  No LDS staging of the input: at these
// sizes the whole volume lives in L2.
template <int PREC, int KS>
__global__ __launch_bounds__(KS * 64) void conv_small(const ConvArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int BATCH = 8;                       // chunks whose fragments are in flight together (16 VGPRs each in split-bf16)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int nt = blockIdx.y;
    const int NT = gridDim.y;
    __shared__ f32x4 red[KS][64];

    const int ps0 = PARTS * a.C0, ps1 = PARTS * a.C1;
    const int64_t p = (int64_t)blockIdx.x * 16 + r;
    const bool pv = p < a.M;
    const int64_t pp = pv ? p : 0;
    const int x = (int)(pp % a.Wg);
    int64_t tq = pp / a.Wg;
    const int y = (int)(tq % a.Hg);
    tq /= a.Hg;
    const int n = (int)(tq % a.Ng);
    const int b = (int)(tq / a.Ng);
    const int py = y * a.sy, px = x * a.sx;
    const int pbase = (n * a.Hi + py) * a.Wi + px;
    const uint16_t *s0 = a.in0 + (int64_t)b * a.Ni * a.Hi * a.Wi * ps0;
    const uint16_t *s1 = a.in1 + (int64_t)b * a.Ni * a.Hi * a.Wi * ps1;
    const int64_t opix = (((int64_t)b * a.No + n) * a.Ho + (y * a.osy + a.ooy)) * a.Wo + (x * a.osx + a.oox);

    // this wave's share of the contraction: chunks [k0, k1)
    const int k0 = (int)((int64_t)a.KC * wave / KS), k1 = (int)((int64_t)a.KC * (wave + 1) / KS);
    const short8 *wbase = reinterpret_cast<const short8 *>(a.wpk) + lane;
    const short8 zero8 = short8{0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    for (int kb = k0; kb < k1; kb += BATCH) {
        TapEntry te[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) te[i] = a.tab[(kb + i < k1 ? kb + i : k1 - 1) * 4 + g];
        short8 xh[BATCH], xl[BATCH], wf[BATCH][PARTS];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int kc = kb + i < k1 ? kb + i : k1 - 1;      // (past the end: the last chunk again, its products are skipped below)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) wf[i][pt] = wbase[((int64_t)kc * NT * PARTS + nt * PARTS + pt) * 64];
            const bool tvalid = te[i].coff >= 0;
            const bool second = te[i].coff >= a.C0;
            const int cc = tvalid ? (second ? te[i].coff - a.C0 : te[i].coff) : 0;
            const int iz = n + te[i].dz, iy = py + te[i].dy, ix = px + te[i].dx;
            const bool ok = pv && tvalid && (unsigned)iz < (unsigned)a.Ni && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const int delta = (te[i].dz * a.Hi + te[i].dy) * a.Wi + te[i].dx;
            const uint16_t *src = (second ? s1 : s0) + ((int64_t)(pbase + delta) * (second ? ps1 : ps0) + cc);
            xh[i] = zero8;
            xl[i] = zero8;
            if (ok) {
                xh[i] = *reinterpret_cast<const short8 *>(src);
                if constexpr (PARTS == 2) xl[i] = *reinterpret_cast<const short8 *>(src + (second ? a.C1 : a.C0));
            }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (kb + i < k1) {   // two accumulators alternately: a lone dependent chain issues an MFMA every ~36 cycles instead of 16
                f32x4 &pa = (i & 1) ? acc2 : acc, &qa = (i & 1) ? acc : acc2;
                if constexpr (PARTS == 2) {
                    pa = mma<F16>(wf[i][1], xh[i], pa);
                    qa = mma<F16>(wf[i][0], xl[i], qa);
                }
                pa = mma<F16>(wf[i][0], xh[i], pa);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += acc2[i];
    if constexpr (KS > 1) {
        red[wave][lane] = acc;
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < KS; ++w) {   // fixed order: deterministic
            const f32x4 o = red[w][lane];
            acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
        }
    }
    float cls = 0.f;
    epilogue_quad<PREC, false>(a, acc, nt, g, opix, pv, cls, uint4{}, uint4{});
}

// true when conv_small serves this launch better than conv_igemm: few operand tiles (conv_igemm would start fewer than ~64
// workgroups) and no fused classifier (its dot product spans all output tiles of a pixel, which conv_small deals to workgroups)
static bool conv_small_wanted(const ConvArgs &a) {
    const int64_t tiles = (a.M + 15) / 16;
    return !a.cls_w && tiles <= 4096 && a.KC >= 1;
}

template <int PREC>
static hipError_t launch_conv_small(const ConvArgs &a, hipStream_t s) {
    const int nt = conv_nt_for(a.Cout);           // weights are packed for this many 16-channel output tiles
    const int nt_live = (a.Cout + 15) / 16;       // ... of which these carry channels
    const dim3 grid((unsigned)((a.M + 15) / 16), (unsigned)nt);
    (void)nt_live;
    if (a.KC >= 8) hipLaunchKernelGGL((conv_small<PREC, 8>), grid, dim3(512), 0, s, a);
    else if (a.KC >= 4) hipLaunchKernelGGL((conv_small<PREC, 4>), grid, dim3(256), 0, s, a);
    else if (a.KC >= 2) hipLaunchKernelGGL((conv_small<PREC, 2>), grid, dim3(128), 0, s, a);
    else hipLaunchKernelGGL((conv_small<PREC, 1>), grid, dim3(64), 0, s, a);
    return hipGetLastError();
}

template <int PREC, int NT, int MT>
static hipError_t launch_conv_t(const ConvArgs &a, hipStream_t s) {
    const int64_t per_wg = 4 * MT * 16;
    const int64_t grid = (a.M + per_wg - 1) / per_wg;
    hipLaunchKernelGGL((conv_igemm<PREC, NT, MT>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int PREC>
static hipError_t launch_conv_p(const ConvArgs &a, hipStream_t s) {
    if (conv_small_wanted(a) && !(a.dbg & 32)) return launch_conv_small<PREC>(a, s);
    const int nt = (a.Cout + 15) / 16;
    if (nt <= 1) return launch_conv_t<PREC, 1, 4>(a, s);
    if (nt <= 2) return launch_conv_t<PREC, 2, 4>(a, s);
    if (nt <= 4) return launch_conv_t<PREC, 4, 4>(a, s);
    if (nt <= 8) return launch_conv_t<PREC, 8, 2>(a, s);
    return hipErrorInvalidValue;
}

// number of 16-channel output tiles the kernel chosen for `cout` iterates over (weights are packed for it)
int conv_nt_for(int cout) {
    const int nt = (cout + 15) / 16;
    return nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : 8;
}

void conv_kernel_name(int prec, int cout, char *buf, int n) {
    const int nt = conv_nt_for(cout);
    snprintf(buf, n, "dffw::conv_igemm<%d, %d, %d>", prec, nt, nt == 8 ? 2 : 4);
}

// name of the kernel launch_conv picks for these arguments (conv_small for small grids, else conv_igemm)
void conv_kernel_name_for(int prec, const ConvArgs &a, char *buf, int n) {
    if (conv_small_wanted(a) && !(a.dbg & 32)) snprintf(buf, n, "dffw::conv_small<%d, %d>", prec, a.KC >= 8 ? 8 : a.KC >= 4 ? 4 : a.KC >= 2 ? 2 : 1);
    else conv_kernel_name(prec, a.Cout, buf, n);
}

hipError_t launch_conv(int prec, const ConvArgs &a, hipStream_t s) {
    switch (prec) {
        case P_BF16X3: return launch_conv_p<P_BF16X3>(a, s);
        case P_FP16: return launch_conv_p<P_FP16>(a, s);
        case P_BF16: return launch_conv_p<P_BF16>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---- layout conversion -------------------------------------------------------------------------
// focal stack (B,3,N,H,W) fp32 -> paired-pixel volume (B,N,H,W+2) of 8-channel records: record q of a
// row holds RGB of pixel q-2 in channels 0..2 and RGB of pixel q in channels 4..6 (zero outside
// [0,W); channels 3 and 7 zero).  The stem conv is dilated by 2, so its x-taps j and j+1 read pixels
// x+2j-8 and x+2j-6: with this pairing both sit in ONE 16-byte record (q = x+2j-6) and the 9x9x3
// stencil becomes 9x5 taps of a full 8-channel contraction group (360 instead of 648 deep).
template <int PREC>
__global__ __launch_bounds__(256) void stack_in_kernel(const float *__restrict__ FS, uint16_t *__restrict__ out, int B,
                                                       int N, int H, int W) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int W2 = W + 2;
    const int64_t plane = (int64_t)N * H * W;
    const int64_t total = (int64_t)B * N * H * W2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % W2);
        const int64_t row = i / W2;              // (b*N + n)*H + y
        const int64_t b = row / ((int64_t)N * H);
        const int64_t nh = row - b * (int64_t)N * H;
        const float *src = FS + b * 3 * plane + nh * W;
        short8 h = short8{0, 0, 0, 0, 0, 0, 0, 0}, l = h;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint16_t hi, lo;
            Fmt<PREC>::split(q >= 2 ? src[c * plane + q - 2] : 0.f, hi, lo);
            h[c] = (short)hi;
            l[c] = (short)lo;
            Fmt<PREC>::split(q < W ? src[c * plane + q] : 0.f, hi, lo);
            h[4 + c] = (short)hi;
            l[4 + c] = (short)lo;
        }
        short8 *dst = reinterpret_cast<short8 *>(out + i * (PARTS * 8));
        dst[0] = h;
        if constexpr (PARTS == 2) dst[1] = l;
    }
}

template <int PREC>
__global__ __launch_bounds__(256) void from_ncdhw_kernel(const float *__restrict__ x, uint16_t *__restrict__ out, int B,
                                                         int C, int64_t plane) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int64_t total = (int64_t)B * plane * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int64_t b = pix / plane, q = pix - b * plane;
        Fmt<PREC>::store(out + pix * (PARTS * C), C, c, x[(b * C + c) * plane + q]);
    }
}

// planar fp32 (B,Cs,plane) -> channels-last C-channel volume, channels Cs..C-1 zero (the 3-channel focal stack
// as the 8-channel input of the alignment network's first block)
template <int PREC>
__global__ __launch_bounds__(256) void from_ncdhw_pad_kernel(const float *__restrict__ x, uint16_t *__restrict__ out, int B,
                                                             int Cs, int C, int64_t plane) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    if (C == 8) {
        // the focal stack as the alignment network's 8-channel input: one thread per pixel, plane-coalesced loads, the whole
        // record as one 16-byte store per part (one thread per element meant 2-byte stores)
        const int64_t npix = (int64_t)B * plane;
        for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
            const int64_t b = pix / plane, q = pix - b * plane;
            short8 h = short8{0, 0, 0, 0, 0, 0, 0, 0}, l = h;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < Cs) {
                    uint16_t hi, lo;
                    Fmt<PREC>::split(x[(b * Cs + c) * plane + q], hi, lo);
                    h[c] = (short)hi;
                    l[c] = (short)lo;
                }
            }
            uint16_t *o = out + pix * (PARTS * 8);
            *reinterpret_cast<short8 *>(o) = h;
            if constexpr (PARTS == 2) *reinterpret_cast<short8 *>(o + 8) = l;
        }
        return;
    }
    const int64_t total = (int64_t)B * plane * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int64_t b = pix / plane, q = pix - b * plane;
        Fmt<PREC>::store(out + pix * (PARTS * C), C, c, c < Cs ? x[(b * Cs + c) * plane + q] : 0.f);
    }
}

template <int PREC>
__global__ __launch_bounds__(256) void to_ncdhw_kernel(const uint16_t *__restrict__ x, float *__restrict__ out, int B,
                                                       int C, int64_t plane) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int64_t total = (int64_t)B * plane * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i % plane;
        const int64_t t = i / plane;
        const int c = (int)(t % C);
        const int64_t b = t / C;
        out[i] = Fmt<PREC>::load(x + (b * plane + q) * (PARTS * C), C, c);
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

#define DFFW_PREC_SWITCH(prec, CALL)                    \
    switch (prec) {                                     \
        case P_BF16X3: { constexpr int PR = P_BF16X3; CALL; break; } \
        case P_FP16: { constexpr int PR = P_FP16; CALL; break; }     \
        case P_BF16: { constexpr int PR = P_BF16; CALL; break; }     \
        default: return hipErrorInvalidValue;           \
    }

hipError_t launch_stack_in(int prec, const float *FS, uint16_t *out, int B, int N, int H, int W, hipStream_t s) {
    const int64_t total = (int64_t)B * N * H * (W + 2);
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((stack_in_kernel<PR>), dim3(grid_for(total)), dim3(256), 0, s, FS, out, B, N, H, W));
    return hipGetLastError();
}

hipError_t launch_from_ncdhw(int prec, const float *x, uint16_t *out, int B, int C, int N, int H, int W, hipStream_t s) {
    const int64_t plane = (int64_t)N * H * W;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((from_ncdhw_kernel<PR>), dim3(grid_for(B * plane * C)), dim3(256), 0, s, x, out, B, C, plane));
    return hipGetLastError();
}

hipError_t launch_from_ncdhw_pad(int prec, const float *x, uint16_t *out, int B, int Cs, int C, int N, int H, int W, hipStream_t s) {
    const int64_t plane = (int64_t)N * H * W;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((from_ncdhw_pad_kernel<PR>), dim3(grid_for(B * plane * C)), dim3(256), 0, s, x, out, B, Cs, C, plane));
    return hipGetLastError();
}

hipError_t launch_to_ncdhw(int prec, const uint16_t *x, float *out, int B, int C, int N, int H, int W, hipStream_t s) {
    const int64_t plane = (int64_t)N * H * W;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((to_ncdhw_kernel<PR>), dim3(grid_for(B * plane * C)), dim3(256), 0, s, x, out, B, C, plane));
    return hipGetLastError();
}

// ---- split-K finish ----------------------------------------------------------------------------------------
// Sum of the ksplit fp32 partial volumes of a split contraction (fixed order: deterministic), then the conv
// epilogue: BatchNorm shift, residual, ReLU, storage format.  One thread per (output pixel, 4 channels).
template <int PREC>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float *__restrict__ partial, int ksplit, int64_t stride,
                                                            int64_t M, int cpad, int Cout, const float *__restrict__ bias,
                                                            const uint16_t *__restrict__ res0, int relu, uint16_t *__restrict__ out) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int q4 = Cout / 4;
    const int64_t total = M * q4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % q4) * 4;
        const int64_t pix = i / q4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(partial + pix * cpad + c0);
        for (int z = 1; z < ksplit; ++z) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(partial + z * stride + pix * cpad + c0);
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = v[k] + bias[c0 + k];
            if (relu == 2) x = fmaxf(x, 0.f);
            if (res0) x += Fmt<PREC>::load(res0 + pix * (PARTS * Cout), Cout, c0 + k);
            if (relu == 1) x = fmaxf(x, 0.f);
            Fmt<PREC>::store(out + pix * (PARTS * Cout), Cout, c0 + k, x);
        }
    }
}

hipError_t launch_splitk_finish(int prec, const float *partial, int ksplit, int64_t stride, int64_t M, int cpad, int Cout,
                                const float *bias, const uint16_t *res0, int relu, uint16_t *out, hipStream_t s) {
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((splitk_finish_kernel<PR>), dim3(grid_for(M * (Cout / 4))), dim3(256), 0, s, partial,
                                                ksplit, stride, M, cpad, Cout, bias, res0, relu, out));
    return hipGetLastError();
}

// ---- pooling -----------------------------------------------------------------------------------
// One thread per (output pixel, 8-channel group): 16-byte loads per part, fp32 reduce, re-split.
template <int PREC>
__global__ __launch_bounds__(256) void pool_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ out, int B, int N,
                                                   int H, int W, int C, int k, int mode) {
    // k adjacent lanes per (output pixel, channel octet): lane dy takes row dy of the k x k window (its k loads are independent
    // and requested together), the rows are combined by a xor-shuffle tree -- the same order at every batch size, and a
    // 64-load serial chain per thread (27 us for the (1,8,8) pool of one stack) becomes 8 loads + 3 shuffles.  k = 2, 4, 8.
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int Ho = H / k, Wo = W / k, C8 = C / 8;
    const int64_t total = (int64_t)B * N * Ho * Wo * C8 * k;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int dy = (int)(i0 & (k - 1));
        const int64_t i = i0 / k;
        const int c8 = (int)(i % C8);
        int64_t t = i / C8;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int64_t bn = t / Ho;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mode == 0 ? -INFINITY : 0.f;
        const uint16_t *row = x + ((bn * H + (oy * k + dy)) * W + ox * k) * (int64_t)(PARTS * C) + c8 * 8;
        for (int dx = 0; dx < k; ++dx) {
            const uint16_t *p = row + dx * (PARTS * C);
            const short8 h = *reinterpret_cast<const short8 *>(p);
            short8 l = short8{0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (PARTS == 2) l = *reinterpret_cast<const short8 *>(p + C);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = Fmt<PREC>::join((uint16_t)h[j], (uint16_t)l[j]);
                v[j] = mode == 0 ? fmaxf(v[j], f) : v[j] + f;
            }
        }
        for (int off = 1; off < k; off <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float o = __shfl_xor(v[j], off);
                v[j] = mode == 0 ? fmaxf(v[j], o) : v[j] + o;
            }
        }
        if (dy != 0) continue;
        short8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint16_t hi, lo;
            Fmt<PREC>::split(mode == 0 ? v[j] : v[j] * inv, hi, lo);
            h[j] = (short)hi;
            l[j] = (short)lo;
        }
        uint16_t *q = out + ((bn * Ho + oy) * Wo + ox) * (int64_t)(PARTS * C) + c8 * 8;
        *reinterpret_cast<short8 *>(q) = h;
        if constexpr (PARTS == 2) *reinterpret_cast<short8 *>(q + C) = l;
    }
}

// pool3_kernel: the pyramid's three average pools (1,2,2), (1,4,4), (1,8,8) of the same volume (hourglassup.forward, DEN.py:212-216) in ONE pass: as three
// launches the 32-channel volume was read three times (0.17 GB each at batch 32).  Work item = (8 x 8 block, channel octet); its 8 lanes hold one block row
// each (8 pixels x 8 channels, joined to fp32 in registers) and every level is summed from those registers in pool_kernel's own order -- the window row
// left to right, then the xor-shuffle tree over the window's rows -- so the three outputs are bit-identical to the three launches (tested).
template <int PREC>
__global__ __launch_bounds__(256) void pool3_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ o2, uint16_t *__restrict__ o4, uint16_t *__restrict__ o8,
                                                    int B, int N, int H, int W, int C) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int H8 = H / 8, W8 = W / 8, C8 = C / 8;
    const int64_t total = (int64_t)B * N * H8 * W8 * C8 * 8;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int dy = (int)(i0 & 7);
        const int64_t i = i0 >> 3;
        const int c8 = (int)(i % C8);
        int64_t t = i / C8;
        const int bx = (int)(t % W8);
        t /= W8;
        const int by = (int)(t % H8);
        const int64_t bn = t / H8;
        float v[8][8];   // [pixel of the row][channel]
        const uint16_t *row = x + ((bn * H + (by * 8 + dy)) * W + bx * 8) * (int64_t)(PARTS * C) + c8 * 8;
#pragma unroll
        for (int dx = 0; dx < 8; ++dx) {
            const uint16_t *p = row + dx * (PARTS * C);
            const short8 h = *reinterpret_cast<const short8 *>(p);
            short8 l = short8{0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (PARTS == 2) l = *reinterpret_cast<const short8 *>(p + C);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[dx][j] = Fmt<PREC>::join((uint16_t)h[j], (uint16_t)l[j]);
        }
        auto emit = [&](uint16_t *out, int k, int Ho, int Wo, int oy, int ox, const float (&sum)[8]) {
            const float inv = 1.0f / (float)(k * k);
            short8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint16_t hi, lo;
                Fmt<PREC>::split(sum[j] * inv, hi, lo);
                h[j] = (short)hi;
                l[j] = (short)lo;
            }
            uint16_t *q = out + ((bn * Ho + oy) * Wo + ox) * (int64_t)(PARTS * C) + c8 * 8;
            *reinterpret_cast<short8 *>(q) = h;
            if constexpr (PARTS == 2) *reinterpret_cast<short8 *>(q + C) = l;
        };
        // level k: windows of k columns inside the row (summed left to right from zero, as pool_kernel does), then the tree over k rows
        auto level = [&](auto K, uint16_t *out) {
            constexpr int k = decltype(K)::value;
#pragma unroll
            for (int wx = 0; wx < 8 / k; ++wx) {
                float sum[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int dx = 0; dx < k; ++dx) a = a + v[wx * k + dx][j];
                    sum[j] = a;
                }
#pragma unroll
                for (int off = 1; off < k; off <<= 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum[j] = sum[j] + __shfl_xor(sum[j], off);
                }
                if ((dy & (k - 1)) == 0) emit(out, k, H / k, W / k, (by * 8 + dy) / k, bx * (8 / k) + wx, sum);
            }
        };
        level(std::integral_constant<int, 2>{}, o2);
        level(std::integral_constant<int, 4>{}, o4);
        level(std::integral_constant<int, 8>{}, o8);
    }
}

hipError_t launch_pool3(int prec, const uint16_t *x, uint16_t *o2, uint16_t *o4, uint16_t *o8, int B, int N, int H, int W, int C, hipStream_t s) {
    if (C % 8 || H % 8 || W % 8) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * N * (H / 8) * (W / 8) * (C / 8) * 8;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((pool3_kernel<PR>), dim3(grid_for(total)), dim3(256), 0, s, x, o2, o4, o8, B, N, H, W, C));
    return hipGetLastError();
}

hipError_t launch_pool(int prec, int mode, int k, const uint16_t *x, uint16_t *out, int B, int N, int H, int W, int C,
                       hipStream_t s) {
    if (C % 8 || H % k || W % k || (k != 2 && k != 4 && k != 8)) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * N * (H / k) * (W / k) * (C / 8) * k;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((pool_kernel<PR>), dim3(grid_for(total)), dim3(256), 0, s, x, out, B, N, H, W, C, k, mode));
    return hipGetLastError();
}

// ---- fused cross-slice attention of the SRD block ---------------------------------------------------
// out = feat + relu(W1 . relu(W3 . [feat(n-1); feat(n); feat(n+1)]))      (DEN.py:320-330, no BN, no bias)
// Both convs are pointwise in (y,x) and only couple neighbouring slices, so the whole chain is one
// streaming pass: one thread per (pixel, slice) reads the three neighbouring feature vectors (the
// two re-reads hit in L2: HBM sees every vector once) and writes the result once — the unfused form
// moves 5x the bytes.  The 4*C*C weights are wave-uniform: they arrive through scalar loads and feed
// the FMAs as SGPR operands, in exact fp32 (no loop around them, so they are never spilled).
// Threads are laid out so that 4 consecutive lanes hold a 2x2 pixel block (a wave = 2 rows x 32 columns, still
// 1 KiB contiguous per row): when `pooled` is given, the (1,2,2) max-pool of the result that the following EFD
// block needs (DEN.py:310) is reduced across the quad with two DPP shuffles per channel and written by the
// quad's first lane, which replaces a separate pooling pass over the full-resolution volume.  H, W even.
template <int PREC, int C, bool POOL>
__global__ __launch_bounds__(256) void srd_attention_kernel(const uint16_t *__restrict__ feat, uint16_t *__restrict__ out,
                                                            const float *__restrict__ w3,  // [kz][ci][co]
                                                            const float *__restrict__ w1,  // [ci][co]
                                                            int B, int N, int H, int W, uint16_t *__restrict__ pooled) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr int REC = PARTS * C;  // 16-bit elements per pixel record
    const int HW = H * W;
    const int64_t total = (int64_t)B * N * HW;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;          // total is a multiple of 4: whole quads leave together
    const int64_t q = t >> 2;
    const int W2 = W >> 1, H2 = H >> 1;
    const int qx = (int)(q % W2);
    const int64_t qr = q / W2;
    const int qy = (int)(qr % H2);
    const int64_t bn = qr / H2;
    const int n = (int)(bn % N);
    const int64_t p = (bn * H + 2 * qy + (int)((t >> 1) & 1)) * W + 2 * qx + (int)(t & 1);
    const uint16_t *src = feat + p * REC;
    const int64_t nstride = (int64_t)HW * REC;

    auto load = [&](const uint16_t *r, bool valid, float (&f)[C]) {
#pragma unroll
        for (int c8 = 0; c8 < C / 8; ++c8) {
            short8 h = short8{0, 0, 0, 0, 0, 0, 0, 0}, l = h;
            if (valid) {
                h = *reinterpret_cast<const short8 *>(r + c8 * 8);
                if constexpr (PARTS == 2) l = *reinterpret_cast<const short8 *>(r + C + c8 * 8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) f[c8 * 8 + j] = Fmt<PREC>::join((uint16_t)h[j], (uint16_t)l[j]);
        }
    };

    float fp[C], fc[C], fn[C];
    load(src - nstride, n > 0, fp);
    load(src, true, fc);
    load(src + nstride, n + 1 < N, fn);

    float a[C];
#pragma unroll
    for (int co = 0; co < C; ++co) a[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < C; ++ci)
#pragma unroll
        for (int co = 0; co < C; ++co) {
            a[co] = fmaf(w3[(0 * C + ci) * C + co], fp[ci], a[co]);
            a[co] = fmaf(w3[(1 * C + ci) * C + co], fc[ci], a[co]);
            a[co] = fmaf(w3[(2 * C + ci) * C + co], fn[ci], a[co]);
        }
    float o[C];
#pragma unroll
    for (int co = 0; co < C; ++co) o[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
        const float r = fmaxf(a[ci], 0.f);
#pragma unroll
        for (int co = 0; co < C; ++co) o[co] = fmaf(w1[ci * C + co], r, o[co]);
    }
    uint16_t *w = out + p * REC;
    short8 ph[C / 8], pl[C / 8];
#pragma unroll
    for (int c8 = 0; c8 < C / 8; ++c8) {
        short8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint16_t hi, lo;
            Fmt<PREC>::split(fc[c8 * 8 + j] + fmaxf(o[c8 * 8 + j], 0.f), hi, lo);
            h[j] = (short)hi;
            l[j] = (short)lo;
            if constexpr (POOL) {   // max over the quad of the value as stored (what a separate pooling pass would read)
                float m = Fmt<PREC>::join(hi, lo);
                m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
                m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
                Fmt<PREC>::split(m, hi, lo);
                ph[c8][j] = (short)hi;
                pl[c8][j] = (short)lo;
            }
        }
        *reinterpret_cast<short8 *>(w + c8 * 8) = h;
        if constexpr (PARTS == 2) *reinterpret_cast<short8 *>(w + C + c8 * 8) = l;
    }
    if constexpr (POOL) {
        if ((t & 3) == 0) {
            uint16_t *pw = pooled + ((bn * H2 + qy) * W2 + qx) * REC;
#pragma unroll
            for (int c8 = 0; c8 < C / 8; ++c8) {
                *reinterpret_cast<short8 *>(pw + c8 * 8) = ph[c8];
                if constexpr (PARTS == 2) *reinterpret_cast<short8 *>(pw + C + c8 * 8) = pl[c8];
            }
        }
    }
}

bool srd_attention_supported(int C) { return C == 8 || C == 16; }

hipError_t launch_srd_attention(int prec, const uint16_t *feat, uint16_t *out, const float *w3, const float *w1, int B, int N,
                                int H, int W, int C, uint16_t *pooled, hipStream_t s) {
    if ((H | W) & 1) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * N * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (C == 8 && pooled) {
        DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((srd_attention_kernel<PR, 8, true>), dim3(grid), dim3(256), 0, s, feat, out, w3, w1, B, N, H, W, pooled));
    } else if (C == 8) {
        DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((srd_attention_kernel<PR, 8, false>), dim3(grid), dim3(256), 0, s, feat, out, w3, w1, B, N, H, W, pooled));
    } else if (C == 16 && pooled) {
        DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((srd_attention_kernel<PR, 16, true>), dim3(grid), dim3(256), 0, s, feat, out, w3, w1, B, N, H, W, pooled));
    } else if (C == 16) {
        DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((srd_attention_kernel<PR, 16, false>), dim3(grid), dim3(256), 0, s, feat, out, w3, w1, B, N, H, W, pooled));
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- FOV warp of the End_to_End alignment path (reference End_to_End/End_to_End.py:106-134) ----------------
// out[b,c,n,y,x] = trilinear sample (zeros padding, align_corners=True) of x[b,c] at
//     (x - flow_x, y - flow_y, n),   flow_x = (W//2)*(FOV_n + a0_n - 1)*lin_x + a1_n,  flow_y likewise with
// (H//2), lin_y and a2_n.  The reference builds five (B,.,N,H,W) coordinate grids on the CPU and copies them
// to the device on every call; here the grid is analytic, one thread per (b,n,y,x) computes the four
// bilinear corners once and reuses them for every channel.  The fp32 operation order of the reference
// (normalise to [-1,1], then grid_sample's un-normalise) is kept so the result matches to rounding.
// (WarpPoint / warp_point / warp_octet: dffw_device.h, shared with conv_tile's warp-fill variant)
// OFF: the type of a corner's byte offset inside one (sample, channel) volume -- unsigned whenever the volume is below 4 GiB (any real stack): the loads and the
// store then address as uniform base + 32-bit lane offset (round 5: the 64-bit per-lane address arithmetic was 100 of the kernel's 470 vector instructions, and the
// kernel is bound by them: valu_issue 0.80); int64_t otherwise.  Sample arithmetic and summation order are the same in both.
template <typename OFF>
__global__ __launch_bounds__(256) void fov_warp_kernel(const float *__restrict__ x, const float *__restrict__ alpha,
                                                       const float *__restrict__ fov, float *__restrict__ out,
                                                       float *__restrict__ flow, int B, int C, int N, int H, int W,
                                                       int alpha_from_sample0) {
    // grid = (pixel blocks of a slice, B * N): 32-bit index arithmetic (as one flat 64-bit index the four divisions per pixel were the
    // kernel's longest dependency chain)
    const int64_t plane = (int64_t)H * W;
    const int n = blockIdx.y % N, b = blockIdx.y / N;
    // the reference's batch>1 quirk (End_to_End.py:112): `alpha[:,0,:,:] + FOVs` broadcasts to (B,B,N,1,1) and `[:,0]` keeps
    // alpha[0,n] + FOVs[b,n] -- sample 0's scale offset, but every sample's OWN field of view
    const int ab = alpha_from_sample0 ? 0 : b;
    const float a0 = alpha[(ab * 3 + 0) * N + n], a1 = alpha[(b * 3 + 1) * N + n], a2 = alpha[(b * 3 + 2) * N + n];
    const float f = a0 + fov[b * N + n];
    const float gz = 2.0f * (float)n / (float)(N > 1 ? N - 1 : 1) - 1.0f;
    const float sz = ((gz + 1.0f) * 0.5f) * (float)(N - 1);
    const float z0f = floorf(sz);
    const int z0 = (int)z0f;
    const float wz1 = sz - z0f;
    const float wz[2] = {1.0f - wz1, wz1};
    const bool blend_z = wz1 != 0.f;   // (wave-uniform: the slice coordinate depends on n alone)
    // 32-bit offsets go through buffer instructions (scalar descriptor of the wave-uniform base + the lane's byte offset: no 64-bit address per lane)
    auto ld = [](const char *base, OFF off) -> float {
        if constexpr (sizeof(OFF) == 4) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)0xFFFFFFFF, 0x00020000);
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
        } else {
            return *reinterpret_cast<const float *>(base + off);
        }
    };
    auto st = [](char *base, OFF off, float v) {
        if constexpr (sizeof(OFF) == 4) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)0xFFFFFFFF, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 0);
        } else {
            *reinterpret_cast<float *>(base + off) = v;
        }
    };
    for (unsigned ip = blockIdx.x * 256u + threadIdx.x; ip < (unsigned)plane; ip += gridDim.x * 256u) {
        const int yy = (int)(ip / (unsigned)W), xx = (int)(ip - (unsigned)yy * (unsigned)W);
        const WarpPoint wp = warp_point(xx, yy, H, W, f, a1, a2);
        if (flow) {
            flow[((int64_t)(b * 2 + 0) * N + n) * plane + ip] = wp.fx;
            flow[((int64_t)(b * 2 + 1) * N + n) * plane + ip] = wp.fy;
        }
        const float sx = wp.sx, sy = wp.sy;
        const float x0f = floorf(sx), y0f = floorf(sy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float wx1 = sx - x0f, wy1 = sy - y0f;
        const float wx[2] = {1.0f - wx1, wx1}, wy[2] = {1.0f - wy1, wy1};
        // the (up to) 8 corners once per pixel: byte offset, weight, validity; then per channel all corner loads are requested
        // together (as nested loops with early-outs every corner waited for its own load).  Corners outside the volume, or on a
        // slice with zero weight, are skipped exactly as before, and the sum runs in the same (dz, dy, dx) order.
        OFF coff[8];
        float cw[8];
        bool cok[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
            const int zz = z0 + dz, yc = y0 + dy, xc = x0 + dx;
            cok[k] = zz >= 0 && zz < N && wz[dz] != 0.f && yc >= 0 && yc < H && xc >= 0 && xc < W;
            coff[k] = cok[k] ? ((OFF)zz * (OFF)plane + (OFF)(yc * W + xc)) * (OFF)4 : (OFF)0;
            cw[k] = wx[dx] * wy[dy] * wz[dz];
        }
        const OFF ooff = (OFF)ip * (OFF)4;
        // the slice coordinate is the slice index itself for most slices (exactly: for N = 10 all but slices 1 and 2, where it falls
        // one ulp short): then only the four in-plane corners are requested
        if (!blend_z) {
            for (int c = 0; c < C; ++c) {
                const char *src = reinterpret_cast<const char *>(x + ((int64_t)b * C + c) * N * plane);   // (wave-uniform bases)
                char *dst = reinterpret_cast<char *>(out + (((int64_t)b * C + c) * N + n) * plane);
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ld(src, coff[k]);
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (cok[k]) acc += v[k] * cw[k];
                st(dst, ooff, acc);
            }
            continue;
        }
        for (int c = 0; c < C; ++c) {
            const char *src = reinterpret_cast<const char *>(x + ((int64_t)b * C + c) * N * plane);
            char *dst = reinterpret_cast<char *>(out + (((int64_t)b * C + c) * N + n) * plane);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = ld(src, coff[k]);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (cok[k]) acc += v[k] * cw[k];
            st(dst, ooff, acc);
        }
    }
}

hipError_t launch_fov_warp(const float *x, const float *alpha, const float *fov, float *out, float *flow, int B, int C, int N,
                           int H, int W, int alpha_from_sample0, hipStream_t s) {
    const int64_t plane = (int64_t)H * W;
    if (plane >= (1ll << 31) || (int64_t)B * N > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)std::min<int64_t>((plane + 255) / 256, 65535), B * N);
    if ((int64_t)N * plane * 4 < (1ll << 32)) hipLaunchKernelGGL(fov_warp_kernel<unsigned>, grid, dim3(256), 0, s, x, alpha, fov, out, flow, B, C, N, H, W, alpha_from_sample0);
    else hipLaunchKernelGGL(fov_warp_kernel<int64_t>, grid, dim3(256), 0, s, x, alpha, fov, out, flow, B, C, N, H, W, alpha_from_sample0);
    return hipGetLastError();
}

// ---- alignment-network glue (reference End_to_End/End_to_End.py:71-105) -------------------------------------
// flow_volume: the input of one alpha head, built in one pass from the level's feature volume `fe`
// (channels-last, C channels): per pixel of slice n
//     [ warp(fe)[last slice] (C) | warp(fe)[slice n] (C) | flow_x, flow_y of slice n | 6 zero channels ]
// = torch.cat of End_to_End.py:81-84 padded to a multiple of 8 channels.  The warp is FOV_warp with the
// warp parameters accumulated so far (bilinear in-plane; the slice coordinate of the reference's 3-D grid is
// the slice index itself up to 1 ulp, so no blend across slices is done).  One thread per (pixel, 8-channel
// group): 4 corner records of 16 bytes per part in, one record out.
template <int PREC>
__global__ __launch_bounds__(256) void flow_volume_kernel(const uint16_t *__restrict__ fe, uint16_t *__restrict__ out,
                                                          const float *__restrict__ alpha, const float *__restrict__ fov, int B,
                                                          int N, int H, int W, int C, int mode) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    // channel groups of 8 per output pixel: mode 0 [ref | cur | flow], mode 1 [cur | flow], mode 2 [ref] (one slice per sample)
    const int CG = C / 8;
    const int G = mode == 0 ? 2 * CG + 1 : (mode == 1 ? CG + 1 : CG), Cout = G * 8;
    const int No = mode == 2 ? 1 : N;
    const int64_t total = (int64_t)B * No * H * W * G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        const int64_t pix = i / G;
        const int xx = (int)(pix % W);
        int64_t t = pix / W;
        const int yy = (int)(t % H);
        t /= H;
        const int n = mode == 2 ? N - 1 : (int)(t % No);
        const int b = (int)(t / No);
        const bool is_flow = mode != 2 && g == G - 1;
        const bool is_ref = mode == 2 || (mode == 0 && g < CG);
        const int src = is_ref ? N - 1 : n;   // reference slice for the first C channels, this slice otherwise
        const float f = alpha[(b * 3 + 0) * N + src] + fov[b * N + src];
        const WarpPoint wp = warp_point(xx, yy, H, W, f, alpha[(b * 3 + 1) * N + src], alpha[(b * 3 + 2) * N + src]);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (is_flow) {
            v[0] = wp.fx;
            v[1] = wp.fy;
        } else {
            const int cg = (mode == 0 && g >= CG) ? g - CG : g;
            warp_octet<PREC>(fe + ((int64_t)(b * N + src) * H * W) * (PARTS * C) + cg * 8, C, H, W, wp, v);
        }
        uint4 h, l;
        Fmt<PREC>::split2(v[0], v[1], h.x, l.x);
        Fmt<PREC>::split2(v[2], v[3], h.y, l.y);
        Fmt<PREC>::split2(v[4], v[5], h.z, l.z);
        Fmt<PREC>::split2(v[6], v[7], h.w, l.w);
        uint16_t *dst = out + pix * (PARTS * Cout) + g * 8;
        *reinterpret_cast<uint4 *>(dst) = h;
        if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(dst + Cout) = l;
    }
}

hipError_t launch_flow_volume(int prec, const uint16_t *fe, uint16_t *out, const float *alpha, const float *fov, int B, int N,
                              int H, int W, int C, int mode, hipStream_t s) {
    const int G = mode == 0 ? 2 * (C / 8) + 1 : (mode == 1 ? C / 8 + 1 : C / 8);
    const int64_t total = (int64_t)B * (mode == 2 ? 1 : N) * H * W * G;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((flow_volume_kernel<PR>), dim3(grid_for(total)), dim3(256), 0, s, fe, out, alpha, fov, B, N, H, W, C, mode));
    return hipGetLastError();
}

// alpha_mean: the AdaptiveAvgPool3d((10,1,1)) that ends every alpha head (End_to_End.py:46,57,68) fused with the
// update of End_to_End.py:86-87,94-95,102-103.  head: fp32 (B,3,N,h,w) from the head's last conv; one workgroup
// per (b, parameter, slice) averages its plane (fixed reduction order: deterministic), writes the raw mean to
// raw[(b*3+c)*N+n] and adds it — the scale term c == 0 damped by 0.001 — to the accumulated alpha.
__global__ __launch_bounds__(256) void alpha_mean_kernel(const float *__restrict__ head, float *__restrict__ alpha,
                                                         float *__restrict__ raw, int N, int64_t hw) {
    __shared__ float part[4];
    const int idx = blockIdx.x;
    const float *src = head + (int64_t)idx * hw;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t i = threadIdx.x;
    for (; i + 768 < hw; i += 1024) {
        s0 += src[i];
        s1 += src[i + 256];
        s2 += src[i + 512];
        s3 += src[i + 768];
    }
    for (; i < hw; i += 256) s0 += src[i];
    float sum = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = ((part[0] + part[1]) + (part[2] + part[3])) / (float)hw;
        const int c = (idx / N) % 3;
        if (raw) raw[idx] = m;
        alpha[idx] += (c == 0) ? 0.001f * m : m;
    }
}

hipError_t launch_alpha_mean(const float *head, float *alpha, float *raw, int B, int N, int64_t hw, hipStream_t s) {
    hipLaunchKernelGGL(alpha_mean_kernel, dim3(B * 3 * N), dim3(256), 0, s, head, alpha, raw, N, hw);
    return hipGetLastError();
}

// ---- alpha head tail: Conv3d(C, 3, (1,3,3), padding (0,1,1)) + AdaptiveAvgPool3d((10,1,1)) as plane sums -----------------
// (End_to_End.py:44-46,55-57,66-68.)  The head's last conv is linear and its result is only ever averaged over the whole
// plane of a slice, so the pair collapses exactly (in real arithmetic):
//     mean_{y,x} conv(v)[c] = bias[c] + 1/(H*W) * sum_{ci,dy,dx} w[c][ci][dy][dx] * S[dy][dx][ci],
//     S[dy][dx][ci] = sum of v[ci] over the H x W window the tap (dy,dx) sees = total - (one border row) - (one border column)
//                     + (the corner both removed): tap row dy = 0 reads rows -1..H-2 (row H-1 never), dy = 2 rows 1..H (row 0 never).
// So instead of a 16/32/64 -> 3 channel conv over the full volume (13 dead rows of every MFMA result tile, 3 fp32 planes written and
// read back by the mean) the volume is read ONCE by a streaming sum:
//   plane_sums_kernel: grid (chunks, B*N); a workgroup adds up a contiguous range of pixel records of one slice, every thread a
//     fixed 16-byte piece position of the record (8 channels of one part), float partial per thread, combined in a fixed order
//     -> partial[(plane * (chunks + 4) + chunk) * C + ci] (double); four more workgroups per plane sum the two border rows / columns.
//   head_tail_finish_kernel: one wave per (b, slice): partials -> totals (fixed order), the four corners read from the volume
//     itself, the 27 x C multiply-adds in double, then the update of alpha_mean_kernel (raw mean out, scale term damped by 0.001,
//     accumulated).  Deterministic and independent of the batch position.
// sums[k][ci]: 0 plane total, 1 row 0, 2 row H-1, 3 column 0, 4 column W-1, 5..8 corners (0,0) (0,W-1) (H-1,0) (H-1,W-1);
// w: [3][C][9] then [3] bias.  Thread c (0..2) computes the mean of output channel c and applies alpha_mean_kernel's update.
__device__ __forceinline__ void head_tail_apply(const double (*sums)[64], const float *__restrict__ w, float *__restrict__ alpha,
                                                float *__restrict__ raw, int b, int n, int N, int64_t hw, int C, int c) {
    const float *wc = w + (int64_t)c * C * 9;
    double acc = 0.0;
    for (int k = 0; k < C; ++k) {
        const double T = sums[0][k], R0 = sums[1][k], RL = sums[2][k], C0 = sums[3][k], CL = sums[4][k];
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                double S = T - (dy == 0 ? RL : dy == 2 ? R0 : 0.0) - (dx == 0 ? CL : dx == 2 ? C0 : 0.0);
                if (dy == 0 && dx == 0) S += sums[8][k];
                if (dy == 0 && dx == 2) S += sums[7][k];
                if (dy == 2 && dx == 0) S += sums[6][k];
                if (dy == 2 && dx == 2) S += sums[5][k];
                acc += (double)wc[k * 9 + dy * 3 + dx] * S;
            }
    }
    const float m = (float)((double)w[3 * C * 9 + c] + acc / (double)hw);
    const int idx = (b * 3 + c) * N + n;
    if (raw) raw[idx] = m;
    alpha[idx] += (c == 0) ? 0.001f * m : m;
}

template <int PREC>
__global__ __launch_bounds__(256) void plane_sums_kernel(const uint16_t *__restrict__ v, double *__restrict__ partial, int C, int H, int W,
                                                         int64_t px_per_chunk, int nchunk) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int PR = PARTS * C / 8;                       // 16-byte pieces per pixel record; 256 % PR == 0
    const int64_t plane = blockIdx.y, hw = (int64_t)H * W;
    double *dst = partial + (plane * (nchunk + 4) + blockIdx.x) * C;
    __shared__ float red[256][9];
    if ((int)blockIdx.x >= nchunk) {
        // the last four workgroups of a plane: one border line each (row 0, row H-1, column 0, column W-1); thread = (position
        // phase, channel), a channel's 256 / C partial sums are combined in thread order
        const int line = blockIdx.x - nchunk, len = line < 2 ? W : H;
        const int64_t first = line == 1 ? (int64_t)(H - 1) * W : line == 3 ? W - 1 : 0, step = line < 2 ? 1 : W;
        const uint16_t *vp = v + (plane * hw + first) * (int64_t)(PARTS * C);
        const int ci = threadIdx.x % C, ph = threadIdx.x / C, nph = 256 / C;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int p = ph;
        for (; p + 3 * nph < len; p += 4 * nph) {
            const float a0 = Fmt<PREC>::load(vp + (int64_t)p * step * (PARTS * C), C, ci);
            const float a1 = Fmt<PREC>::load(vp + (int64_t)(p + nph) * step * (PARTS * C), C, ci);
            const float a2 = Fmt<PREC>::load(vp + (int64_t)(p + 2 * nph) * step * (PARTS * C), C, ci);
            const float a3 = Fmt<PREC>::load(vp + (int64_t)(p + 3 * nph) * step * (PARTS * C), C, ci);
            t0 += a0; t1 += a1; t2 += a2; t3 += a3;
        }
        for (; p < len; p += nph) t0 += Fmt<PREC>::load(vp + (int64_t)p * step * (PARTS * C), C, ci);
        red[threadIdx.x][0] = (t0 + t1) + (t2 + t3);
        __syncthreads();
        if ((int)threadIdx.x < C) {
            double a = 0.0;
            for (int k = 0; k < nph; ++k) a += (double)red[k * C + threadIdx.x][0];
            dst[threadIdx.x] = a;
        }
        return;
    }
    const int64_t p0 = (int64_t)blockIdx.x * px_per_chunk, p1 = p0 + px_per_chunk < hw ? p0 + px_per_chunk : hw;
    const uint4 *src = reinterpret_cast<const uint4 *>(v + (plane * hw + p0) * (int64_t)(PARTS * C));
    const int64_t n = (p1 - p0) * PR;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add = [&](const uint4 q) {
        float a, b;
        Fmt<PREC>::join2(q.x, 0u, a, b); s[0] += a; s[1] += b;
        Fmt<PREC>::join2(q.y, 0u, a, b); s[2] += a; s[3] += b;
        Fmt<PREC>::join2(q.z, 0u, a, b); s[4] += a; s[5] += b;
        Fmt<PREC>::join2(q.w, 0u, a, b); s[6] += a; s[7] += b;
    };
    int64_t i = threadIdx.x;
    for (; i + 768 < n; i += 1024) {                     // four loads in flight per thread
        const uint4 q0 = src[i], q1 = src[i + 256], q2 = src[i + 512], q3 = src[i + 768];
        add(q0); add(q1); add(q2); add(q3);
    }
    for (; i < n; i += 256) add(src[i]);
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    if ((int)threadIdx.x < C) {
        // channel ci lives in the pieces (part * C/8 + ci/8) of a record, i.e. in the threads t with t % PR == that; hi and lo add up
        const int ci = threadIdx.x, o = ci >> 3, e = ci & 7, C8 = C >> 3;
        double acc = 0.0;
        for (int t = 0; t < 256; ++t)
            if ((t % PR) % C8 == o) acc += (double)red[t][e];
        dst[ci] = acc;
    }
}

template <int PREC>
__global__ __launch_bounds__(64) void head_tail_finish_kernel(const uint16_t *__restrict__ v, const double *__restrict__ partial, int nchunk,
                                                              const float *__restrict__ w, float *__restrict__ alpha, float *__restrict__ raw,
                                                              int N, int H, int W, int C) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    // sums[k][ci]: 0 total, 1 row 0, 2 row H-1, 3 column 0, 4 column W-1, 5..8 corners (0,0) (0,W-1) (H-1,0) (H-1,W-1)
    __shared__ double sums[9][64];
    const int plane = blockIdx.x, b = plane / N, n = plane % N;
    const int tid = threadIdx.x;
    const int64_t hw = (int64_t)H * W;
    const uint16_t *vp = v + (int64_t)plane * hw * (PARTS * C);
    auto val = [&](int y, int x, int ci) -> double { return (double)Fmt<PREC>::load(vp + ((int64_t)y * W + x) * (PARTS * C), C, ci); };
    if (tid < C) {
        const double *pp = partial + (int64_t)plane * (nchunk + 4) * C + tid;
        double t = 0.0;
        for (int k = 0; k < nchunk; ++k) t += pp[(int64_t)k * C];
        sums[0][tid] = t;
        for (int k = 0; k < 4; ++k) sums[1 + k][tid] = pp[(int64_t)(nchunk + k) * C];
        sums[5][tid] = val(0, 0, tid);
        sums[6][tid] = val(0, W - 1, tid);
        sums[7][tid] = val(H - 1, 0, tid);
        sums[8][tid] = val(H - 1, W - 1, tid);
    }
    __syncthreads();
    if (tid < 3) head_tail_apply(sums, w, alpha, raw, b, n, N, hw, C, tid);
}

// the same finish for of_roll_kernel<.., SUMS>'s per-tile vectors: tsum[(plane * tiles + tile) * 18 * C + k * C + c], k = 3w + {0,1,2} sum
// over wave w's two rows of the tile / their first pixels / their last pixels, 12 / 13 first / last row of the tile, 14..17 corners (TL,
// TR, BL, BR); tiles in row-major order.  Two steps, both in a fixed order: head_tail_tiles_reduce_kernel (grid (HT_SEG, planes)) adds up
// a contiguous range of tiles into seg[(plane * HT_SEG + segment) * 9 * C + k * C + c] (doubles, the nine sums of head_tail_apply);
// head_tail_tiles_finish_kernel adds the segments and applies the weights.
constexpr int HT_SEG = 16;
__global__ __launch_bounds__(256) void head_tail_tiles_reduce_kernel(const float *__restrict__ tsum, double *__restrict__ seg, int tiles_y,
                                                                     int tiles_x, int C) {
    __shared__ double red[9][256];
    const int plane = blockIdx.y;
    const int tid = threadIdx.x, c = tid % C, j0 = tid / C, nj = 256 / C;
    const int ntiles = tiles_y * tiles_x;
    const int per = (ntiles + HT_SEG - 1) / HT_SEG, t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    const float *tp = tsum + (int64_t)plane * ntiles * 18 * C + c;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = t0 + j0; t < t1; t += nj) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const float *q = tp + (int64_t)t * 18 * C;
        const bool top = ty == 0, bot = ty == tiles_y - 1, lef = tx == 0, rig = tx == tiles_x - 1;
        acc[0] += ((double)q[0] + (double)q[3 * C]) + ((double)q[6 * C] + (double)q[9 * C]);
        if (top) acc[1] += (double)q[12 * C];
        if (bot) acc[2] += (double)q[13 * C];
        if (lef) acc[3] += ((double)q[1 * C] + (double)q[4 * C]) + ((double)q[7 * C] + (double)q[10 * C]);
        if (rig) acc[4] += ((double)q[2 * C] + (double)q[5 * C]) + ((double)q[8 * C] + (double)q[11 * C]);
        if (top && lef) acc[5] += (double)q[14 * C];
        if (top && rig) acc[6] += (double)q[15 * C];
        if (bot && lef) acc[7] += (double)q[16 * C];
        if (bot && rig) acc[8] += (double)q[17 * C];
    }
    for (int k = 0; k < 9; ++k) red[k][tid] = acc[k];
    __syncthreads();
    if (tid < C)
        for (int k = 0; k < 9; ++k) {
            double a = 0.0;
            for (int j = 0; j < nj; ++j) a += red[k][j * C + tid];
            seg[(((int64_t)plane * HT_SEG + blockIdx.x) * 9 + k) * C + tid] = a;
        }
}

__global__ __launch_bounds__(64) void head_tail_tiles_finish_kernel(const double *__restrict__ seg, const float *__restrict__ w,
                                                                    float *__restrict__ alpha, float *__restrict__ raw, int N, int H, int W, int C) {
    __shared__ double sums[9][64];
    const int plane = blockIdx.x, b = plane / N, n = plane % N, tid = threadIdx.x;
    if (tid < C)
        for (int k = 0; k < 9; ++k) {
            double a = 0.0;
            for (int sgi = 0; sgi < HT_SEG; ++sgi) a += seg[(((int64_t)plane * HT_SEG + sgi) * 9 + k) * C + tid];
            sums[k][tid] = a;
        }
    __syncthreads();
    if (tid < 3) head_tail_apply(sums, w, alpha, raw, b, n, N, (int64_t)H * W, C, tid);
}

// ... and for conv_tile's row-sums variant: rows[((plane * H + y) * tiles_x + tx) * 3 * C + k * C + c], k = 0 sum over the row segment of 16
// pixels, 1 / 2 its first / last pixel.  Same two steps (segments = contiguous ranges of rows).
__global__ __launch_bounds__(256) void head_tail_rows_reduce_kernel(const float *__restrict__ rows, double *__restrict__ seg, int H, int tiles_x, int C) {
    __shared__ double red[9][256];
    const int plane = blockIdx.y;
    const int tid = threadIdx.x, c = tid % C, j0 = tid / C, nj = 256 / C;
    const int per = (H + HT_SEG - 1) / HT_SEG, y0 = blockIdx.x * per, y1 = y0 + per < H ? y0 + per : H;
    const int n0 = y0 * tiles_x, n1 = y1 * tiles_x;
    const float *tp = rows + (int64_t)plane * H * tiles_x * 3 * C + c;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = n0 + j0; t < n1; t += nj) {
        const int y = t / tiles_x, tx = t - y * tiles_x;
        const float *q = tp + (int64_t)t * 3 * C;
        const bool top = y == 0, bot = y == H - 1, lef = tx == 0, rig = tx == tiles_x - 1;
        const double rs = (double)q[0];
        acc[0] += rs;
        if (top) acc[1] += rs;
        if (bot) acc[2] += rs;
        if (lef) acc[3] += (double)q[C];
        if (rig) acc[4] += (double)q[2 * C];
        if (top && lef) acc[5] += (double)q[C];
        if (top && rig) acc[6] += (double)q[2 * C];
        if (bot && lef) acc[7] += (double)q[C];
        if (bot && rig) acc[8] += (double)q[2 * C];
    }
    for (int k = 0; k < 9; ++k) red[k][tid] = acc[k];
    __syncthreads();
    if (tid < C)
        for (int k = 0; k < 9; ++k) {
            double a = 0.0;
            for (int j = 0; j < nj; ++j) a += red[k][j * C + tid];
            seg[(((int64_t)plane * HT_SEG + blockIdx.x) * 9 + k) * C + tid] = a;
        }
}

int64_t head_tail_tiles_scratch_bytes(int B, int N, int C) { return (int64_t)B * N * HT_SEG * 9 * C * (int64_t)sizeof(double); }

hipError_t launch_head_tail_tiles(const float *tsum, double *seg, int tiles_y, int tiles_x, const float *w, float *alpha, float *raw, int B,
                                  int N, int H, int W, int C, hipStream_t s) {
    if (C < 8 || C > 64 || 256 % C) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_tail_tiles_reduce_kernel, dim3(HT_SEG, B * N), dim3(256), 0, s, tsum, seg, tiles_y, tiles_x, C);
    hipError_t h = hipGetLastError();
    if (h != hipSuccess) return h;
    hipLaunchKernelGGL(head_tail_tiles_finish_kernel, dim3(B * N), dim3(64), 0, s, seg, w, alpha, raw, N, H, W, C);
    return hipGetLastError();
}

hipError_t launch_head_tail_rows(const float *rows, double *seg, int tiles_x, const float *w, float *alpha, float *raw, int B, int N, int H, int W,
                                 int C, hipStream_t s) {
    if (C < 8 || C > 64 || 256 % C) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_tail_rows_reduce_kernel, dim3(HT_SEG, B * N), dim3(256), 0, s, rows, seg, H, tiles_x, C);
    hipError_t h = hipGetLastError();
    if (h != hipSuccess) return h;
    hipLaunchKernelGGL(head_tail_tiles_finish_kernel, dim3(B * N), dim3(64), 0, s, seg, w, alpha, raw, N, H, W, C);
    return hipGetLastError();
}

int head_tail_chunks(int B, int N, int64_t hw) {
    int nchunk = (int)((2560 + (int64_t)B * N - 1) / ((int64_t)B * N));   // ~10 workgroups per CU over the launch
    if (nchunk < 1) nchunk = 1;
    if ((int64_t)nchunk * 256 > hw) nchunk = (int)((hw + 255) / 256);
    return nchunk;
}

hipError_t launch_head_tail(int prec, const uint16_t *v, double *partial, const float *w, float *alpha, float *raw, int B, int N, int H, int W,
                            int C, hipStream_t s) {
    const int64_t hw = (int64_t)H * W;
    const int nchunk0 = head_tail_chunks(B, N, hw);
    const int64_t ppc = (hw + nchunk0 - 1) / nchunk0;
    const int nchunk = (int)((hw + ppc - 1) / ppc);      // no empty chunk; <= head_tail_chunks()
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((plane_sums_kernel<PR>), dim3(nchunk + 4, B * N), dim3(256), 0, s, v, partial, C, H, W, ppc, nchunk));
    hipError_t h = hipGetLastError();
    if (h != hipSuccess) return h;
    DFFW_PREC_SWITCH(prec, hipLaunchKernelGGL((head_tail_finish_kernel<PR>), dim3(B * N), dim3(64), 0, s, v, partial, nchunk, w, alpha, raw, N, H, W, C));
    return hipGetLastError();
}

// ---- depth regression --------------------------------------------------------------------------
// One thread per output pixel; the N <= ~15 per-slice scores of a pixel are consumed in a register
// loop (for fixed slice n consecutive lanes read consecutive x: coalesced), so the soft-argmin
// over focus slices needs no cross-lane traffic at all.
//   s_n   = bilinear(score[b,n], y, x)             (align_corners=False, PyTorch's index rule)
//   p_n   = softplus(s_n) + 1e-6                   (beta 1, threshold 20)
//   depth = sum_n fd_n p_n / sum_n p_n             (DEN.py:88-90)
// softplus (beta 1, threshold 20) = v > 20 ? v : log1p(exp(v)) on the hardware transcendentals instead of libm's
// expf/log1pf (the regression kernels were bound by those two calls):
//   e = exp(v) = exp2(v*log2e), the product carried in two floats so that the result keeps ~1 ulp for |v| up to 88;
//   log1p(e)  = log(w) * e / (w - 1) with w = fl(1 + e) (the classic correction for the rounding of 1 + e: exact-rounded it is within 3.3e-7
//               of log1p over v in [-30, 20]), e itself where w == 1; v_log_f32 and v_rcp_f32, no branch.  (Rounds 1-3 used a 9-term series below
//               e = 0.1 and an IEEE division above it: ~55 vector instructions per value with both sides of the divergent branch executed, and
//               the fused head kernel -- 40 values per pixel -- ran with its vector issue port 1.01 busy, profiles/r04_pmc_conv_kernels.txt.)
// Measured against torch.nn.functional.softplus in tests/test_gpu_ops.py::test_regression_head (2e-6 rel-L2).
__device__ __forceinline__ float softplus_fast(float v) {
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.925963033500414e-08f, LN2 = 0.693147182464599609375f;
    const float hi = v * L2E;
    const float lo = __builtin_fmaf(v, L2E, -hi) + v * L2E_LO;
    float e = __builtin_amdgcn_exp2f(hi);
    e = __builtin_fmaf(e, lo * LN2, e);
    const float w = 1.f + e, d = w - 1.f;
    const float r = (__builtin_amdgcn_logf(w) * LN2) * (e * __builtin_amdgcn_rcpf(d));
    const float sp = d == 0.f ? e : r;
    return v > 20.f ? v : sp;
}

// Up to four regression heads in ONE launch (blockIdx.y = head): the heads at 1/8, 1/4 and 1/2 resolution are 30-us launches of a
// few thousand workgroups each; side by side with the full-resolution head they fill the gaps of each other's load chains.
__global__ __launch_bounds__(256) void regress_kernel(const RegressHeads hd, int B, int N, int H, int W, const float *__restrict__ fd, int64_t fsb,
                                                      int64_t fsn, int64_t fsh, int64_t fsw) {
    const int head = blockIdx.y;
    const float *__restrict__ score = hd.score[head];
    float *__restrict__ depth = hd.depth[head];
    const int h = hd.h[head], w = hd.w[head];
    const int64_t total = (int64_t)B * H * W;
    const float sch = (float)h / (float)H, scw = (float)w / (float)W;
    const bool small = total < (1ll << 31);      // 32-bit index arithmetic whenever it fits (two 64-bit divisions per pixel otherwise)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int X, Y;
        int64_t b;
        if (small) {
            const unsigned u = (unsigned)i, t = u / (unsigned)W;
            X = (int)(u - t * (unsigned)W);
            const unsigned bb = t / (unsigned)H;
            Y = (int)(t - bb * (unsigned)H);
            b = bb;
        } else {
            X = (int)(i % W);
            const int64_t t = i / W;
            Y = (int)(t % H);
            b = t / H;
        }
        float sy = ((float)Y + 0.5f) * sch - 0.5f;
        float sx = ((float)X + 0.5f) * scw - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float *sp = score + b * N * (int64_t)h * w;
        const float *fp = fd + b * fsb + Y * fsh + X * fsw;
        float num = 0.f, den = 0.f;
        // five slices at a time: their 25 loads are all requested before the first softplus (with a run-time slice loop every
        // iteration waited for its own five loads: the kernel ran at a quarter of what its traffic and transcendentals need);
        // the sums still run over the slices in order
        const int o00 = y0 * w + x0, o01 = y0 * w + x1, o10 = y1 * w + x0, o11 = y1 * w + x1;
        for (int n0 = 0; n0 < N; n0 += 5) {
            float s00[5], s01[5], s10[5], s11[5], f[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int n = n0 + k < N ? n0 + k : N - 1;
                const float *pl = sp + (int64_t)n * h * w;
                s00[k] = pl[o00];
                s01[k] = pl[o01];
                s10[k] = pl[o10];
                s11[k] = pl[o11];
                f[k] = fp[n * fsn];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (n0 + k < N) {
                    const float v = hy * (hx * s00[k] + lx * s01[k]) + ly * (hx * s10[k] + lx * s11[k]);
                    const float p = softplus_fast(v) + 1e-6f;
                    den += p;
                    num += f[k] * p;
                }
            }
        }
        depth[i] = num / den;
    }
}

// regress_fused_kernel: all heads of a pixel in one thread (N <= 16 slices).  As blockIdx.y = head every head re-read the pixel's N focus distances
// (a dense (B,N,H,W) map: 84 MB per head at batch 32) and the full-resolution head (pred3, h == H) fetched its one score value per slice
// four times through the bilinear taps.  Here the focus distances are loaded once into registers, the heads run one after the other over them,
// and a head at output resolution takes the single-load path (hy = hx = 1, ly = lx = 0: the value is the same, bit for bit).  Per head the
// arithmetic and its order are regress_kernel's.
template <int NMAX>
__global__ __launch_bounds__(256) void regress_fused_kernel(const RegressHeads hd, int B, int N, int H, int W, const float *__restrict__ fd, int64_t fsb,
                                                            int64_t fsn, int64_t fsh, int64_t fsw) {
    const unsigned total = (unsigned)B * H * W;
    for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < total; u += gridDim.x * blockDim.x) {
        const unsigned t = u / (unsigned)W;
        const int X = (int)(u - t * (unsigned)W);
        const unsigned bb = t / (unsigned)H;
        const int Y = (int)(t - bb * (unsigned)H);
        const float *fp = fd + bb * fsb + Y * fsh + X * fsw;
        float f[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) f[n] = fp[(n < N ? n : N - 1) * fsn];
        for (int head = 0; head < hd.n; ++head) {
            const float *__restrict__ score = hd.score[head];
            const int h = hd.h[head], w = hd.w[head];
            const float *sp = score + (int64_t)bb * N * h * w;
            float num = 0.f, den = 0.f;
            if (h == H && w == W) {
                const int o = Y * w + X;
                float sv[NMAX];
#pragma unroll
                for (int n = 0; n < NMAX; ++n) sv[n] = sp[(int64_t)(n < N ? n : N - 1) * h * w + o];
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    if (n < N) {
                        // (1*(1*s + 0*s) + 0*(1*s + 0*s) of the general path is s itself, so is hy*(hx*s00 + lx*s01) + ly*(...) in fp32 for finite s)
                        const float p = softplus_fast(sv[n]) + 1e-6f;
                        den += p;
                        num += f[n] * p;
                    }
                }
            } else {
                const float sch = (float)h / (float)H, scw = (float)w / (float)W;
                float sy = ((float)Y + 0.5f) * sch - 0.5f;
                float sx = ((float)X + 0.5f) * scw - 0.5f;
                sy = sy < 0.f ? 0.f : sy;
                sx = sx < 0.f ? 0.f : sx;
                const int y0 = (int)sy, x0 = (int)sx;
                const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float ly = sy - (float)y0, lx = sx - (float)x0;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const int o00 = y0 * w + x0, o01 = y0 * w + x1, o10 = y1 * w + x0, o11 = y1 * w + x1;
#pragma unroll   // blocks of 5 slices, fully unrolled: f[n0 + k] is a register, not a select chain over the NMAX candidates
                for (int n0 = 0; n0 < NMAX; n0 += 5) {
                    if (n0 < N) {
                        float s00[5], s01[5], s10[5], s11[5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const int n = n0 + k < N ? n0 + k : N - 1;
                            const float *pl = sp + (int64_t)n * h * w;
                            s00[k] = pl[o00];
                            s01[k] = pl[o01];
                            s10[k] = pl[o10];
                            s11[k] = pl[o11];
                        }
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            if (n0 + k < N) {
                                const float v = hy * (hx * s00[k] + lx * s01[k]) + ly * (hx * s10[k] + lx * s11[k]);
                                const float p = softplus_fast(v) + 1e-6f;
                                den += p;
                                num += f[n0 + k < NMAX ? n0 + k : NMAX - 1] * p;
                            }
                        }
                    }
                }
            }
            hd.depth[head][u] = num / den;
        }
    }
}

hipError_t launch_regress_heads(const RegressHeads &hd, int B, int N, int H, int W, const float *fd, int64_t fsb, int64_t fsn, int64_t fsh,
                                int64_t fsw, hipStream_t s) {
    if (hd.n < 1 || hd.n > 4) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * H * W;
    // the fused form (one thread = all heads of a pixel) for the shapes the network produces; DFFW_NO_REGRESS_MERGE callers pass one head at a time
    // and keep regress_kernel
    if (hd.n > 1 && N <= 16 && total < (1ll << 31) && !hd.nofuse) {
        if (N <= 10) hipLaunchKernelGGL((regress_fused_kernel<10>), dim3(grid_for(total)), dim3(256), 0, s, hd, B, N, H, W, fd, fsb, fsn, fsh, fsw);
        else hipLaunchKernelGGL((regress_fused_kernel<16>), dim3(grid_for(total)), dim3(256), 0, s, hd, B, N, H, W, fd, fsb, fsn, fsh, fsw);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(regress_kernel, dim3(grid_for(total), (unsigned)hd.n), dim3(256), 0, s, hd, B, N, H, W, fd, fsb, fsn, fsh, fsw);
    return hipGetLastError();
}

hipError_t launch_regress(const float *score, int B, int N, int h, int w, int H, int W, const float *fd, int64_t fsb,
                          int64_t fsn, int64_t fsh, int64_t fsw, float *depth, hipStream_t s) {
    RegressHeads hd{};
    hd.n = 1;
    hd.score[0] = score;
    hd.depth[0] = depth;
    hd.h[0] = h;
    hd.w[0] = w;
    return launch_regress_heads(hd, B, N, H, W, fd, fsb, fsn, fsh, fsw, s);
}

__global__ void set_raw_kernel(RawStack rs, RawStack *dst) { *dst = rs; }
hipError_t launch_set_raw(const RawStack &rs, RawStack *dst, hipStream_t s) {
    hipLaunchKernelGGL(set_raw_kernel, dim3(1), dim3(1), 0, s, rs, dst);
    return hipGetLastError();
}

}  // namespace dffw
