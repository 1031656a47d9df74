// Input and output side of the forward: the tensor assembly the reference's
// data loaders do in NumPy before `model(FS, focus_dists)` and the crop / normalise / colour-map / metrics pass its
// scripts do on the depth map afterwards, as HIP kernels on the caller's stream so that neither side of the forward
// goes through host memory.  All three are bandwidth-bound byte/float streaming passes (HBM roofline).
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <stdint.h>

#include "../../include/dffw.h"
#include "dffw_internal.h"

namespace dffw {

// ---- pack_stack -----------------------------------------------------------------------------------------------
// out[b][c][n][y][x] = raw[b,n,y,x,c] / 127.5 - 1 for y < h, x < w, else -1 (the loaders' constant pad value),
// float32 arithmetic in the loaders' order (divide, then subtract: test_Dataloader.py:84,122,213; Test_dataloader.py:58).
// NORM64: the FS6 loader (test_Dataloader.py:31-39) accumulates its images into np.zeros((256,256,3,0)) -- a float64
// array -- so its `mats_input/127.5 - 1.0` runs in float64 and torch.Tensor() rounds once to float32 at the end.
// One thread per output row segment of 4 pixels: the writes (the larger side: 12 B/pixel vs 3 B/pixel of uint8
// input) are 16-byte coalesced; the strided source reads go through L2.
template <typename T, bool NORM64>
__global__ __launch_bounds__(256) void pack_stack_kernel(const T *raw, int64_t sb, int64_t sn, int64_t sy, int64_t sx, int64_t sc, int B, int N,
                                                          int h, int w, int Hp, int Wp, float *out) {
    const int64_t W4 = Wp / 4;
    const int64_t total = (int64_t)B * 3 * N * Hp * W4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x0 = (int)(i % W4) * 4;
        int64_t t = i / W4;
        const int y = (int)(t % Hp);
        t /= Hp;
        const int n = (int)(t % N);
        t /= N;
        const int c = (int)(t % 3);
        const int b = (int)(t / 3);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k;
            v[k] = -1.0f;
            if (y < h && x < w) {
                const T u = raw[b * sb + n * sn + y * sy + x * sx + c * sc];
                v[k] = NORM64 ? (float)__dsub_rn(__ddiv_rn((double)u, 127.5), 1.0) : __fsub_rn(__fdiv_rn((float)u, 127.5f), 1.0f);
            }
        }
        *reinterpret_cast<float4 *>(out + ((((int64_t)b * 3 + c) * N + n) * Hp + y) * Wp + x0) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- colorize -------------------------------------------------------------------------------------------------
// ordered-int trick: monotone map float -> int so that integer atomicMin/Max order floats
__device__ __forceinline__ int f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void minmax_init_kernel(int *mm, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        mm[2 * i] = 0x7FFFFFFF;       // +max ordered int
        mm[2 * i + 1] = (int)0x80000000;
    }
}

// min / max of every (H,W) map of the batch (np.min / np.max over the whole, still padded, map: TRS.py:40)
__global__ __launch_bounds__(256) void minmax_kernel(const float *d, int64_t hw, int *mm) {
    const int b = blockIdx.y;
    const float *p = d + (int64_t)b * hw;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = p[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm + 2 * b, f2ord(lo));
        atomicMax(mm + 2 * b + 1, f2ord(hi));
    }
}

__global__ void range_set_kernel(float *range, int B, float lo, float hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        range[2 * i] = lo;
        range[2 * i + 1] = hi;
    }
}

__global__ void minmax_finish_kernel(int *mm, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * B) reinterpret_cast<float *>(mm)[i] = ord2f(mm[i]);
}

// warped stack -> uint8 slice images (test_real_scenes.py:42-47).  One thread per output pixel: three planar reads (coalesced over
// x), one 3-byte interleaved store.
__global__ __launch_bounds__(256) void unpack_stack_kernel(const float *__restrict__ warp, int B, int N, int H, int W, int h, int w,
                                                          uint8_t *__restrict__ img) {
    const int64_t total = (int64_t)B * N * h * w, plane = (int64_t)N * H * W;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const int64_t t = i / w;
        const int y = (int)(t % h);
        const int64_t bn = t / h;
        const int n = (int)(bn % N), b = (int)(bn / N);
        const float *src = warp + (int64_t)b * 3 * plane + ((int64_t)n * H + y) * W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = __fmul_rn(127.5f, __fadd_rn(src[c * plane], 1.0f));
            // x86 NumPy: cvttss2si to int32 (NaN / out of range -> 0x80000000), low byte kept
            const int q = (v > -2147483648.0f && v < 2147483648.0f) ? (int)v : 0;
            img[i * 3 + c] = (uint8_t)(q & 255);
        }
    }
}

// rgb[b][y][x][:] = lut[index((depth - lo) / (hi - lo))], y < h, x < w: the crop of test.py:124-126 / TRS.py:52, the
// normalisation of test.py:132 (fixed range) or TRS.py:40 (the map's own range), and matplotlib's colour-map lookup
// (float32 x*256 truncated; x == 1 -> 255; below 0 / above 1 clamp to the end colours; NaN -> black).
__global__ __launch_bounds__(256) void colorize_kernel(const float *depth, int H, int W, int h, int w, const float *range, int range_stride,
                                                        const uint8_t *lut, uint8_t *rgb, int B) {
    const int64_t total = (int64_t)B * h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int64_t t = i / w;
        const int y = (int)(t % h);
        const int b = (int)(t / h);
        const float lo = range[b * range_stride], hi = range[b * range_stride + 1];
        const float v = depth[((int64_t)b * H + y) * W + x];
        const float u = __fdiv_rn(__fsub_rn(v, lo), __fsub_rn(hi, lo));
        float s = __fmul_rn(u, 256.0f);
        if (s == 256.0f) s = 255.0f;
        int idx = s < 0.f ? 0 : (s >= 256.0f ? 255 : (int)s);
        uint8_t r = 0, g = 0, bl = 0;
        if (!(s != s)) {
            r = lut[idx * 3];
            g = lut[idx * 3 + 1];
            bl = lut[idx * 3 + 2];
        }
        uint8_t *o = rgb + i * 3;
        o[0] = r;
        o[1] = g;
        o[2] = bl;
    }
}

// ---- metrics --------------------------------------------------------------------------------------------------
// Masked error sums of metrics.py:90-127 for every sample: per-pixel terms in float32 like the NumPy expressions,
// accumulation in float64, fixed reduction order (block partials summed by one thread) so that the result does not
// depend on scheduling.
constexpr int NSUM = 12;   // n, abs_rel, sq_rel, sq, abs, sqlog, acc1, acc2, acc3, conf, conf*sq, conf*abs
__global__ __launch_bounds__(256) void metrics_partial_kernel(const float *est, int H, int W, const float *gt, const uint8_t *mask, const float *conf,
                                                               int h, int w, double *partial) {
    const int b = blockIdx.y;
    double s[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) s[k] = 0.0;
    const int64_t hw = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        if (!mask[(int64_t)b * hw + i]) continue;
        const float e = est[((int64_t)b * H + y) * W + x], g = gt[(int64_t)b * hw + i];
        const float d = __fsub_rn(g, e);
        const float sq = __fmul_rn(d, d), ab = fabsf(d);
        s[0] += 1.0;
        s[1] += (double)__fdiv_rn(ab, g);
        s[2] += (double)__fdiv_rn(sq, g);
        s[3] += (double)sq;
        s[4] += (double)ab;
        const float dl = __fsub_rn(logf(g), logf(e));
        s[5] += (double)__fmul_rn(dl, dl);
        const float th = fmaxf(__fdiv_rn(e, g), __fdiv_rn(g, e));
        s[6] += th < 1.25f ? 1.0 : 0.0;
        s[7] += th < 1.5625f ? 1.0 : 0.0;
        s[8] += th < 1.953125f ? 1.0 : 0.0;
        if (conf) {
            const float c = conf[(int64_t)b * hw + i];
            s[9] += (double)c;
            s[10] += (double)__fmul_rn(c, sq);
            s[11] += (double)__fmul_rn(c, ab);
        }
    }
    __shared__ double sh[4][NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        double v = s[k];
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NSUM) {
        const int k = threadIdx.x;
        partial[((int64_t)b * gridDim.x + blockIdx.x) * NSUM + k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    }
}

__global__ void metrics_finish_kernel(const double *partial, int nblk, double *out) {
    const int b = blockIdx.x;
    __shared__ double s[NSUM];
    if (threadIdx.x < NSUM) {
        double v = 0.0;
        for (int i = 0; i < nblk; ++i) v += partial[((int64_t)b * nblk + i) * NSUM + threadIdx.x];
        s[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = out + (int64_t)b * DFFW_N_METRICS;
        const double n = s[0];
        o[0] = n;
        o[1] = s[1] / n;                 // mask_abs_rel    metrics.py:90-91
        o[2] = s[2] / n;                 // mask_sq_rel     metrics.py:93-94
        o[3] = s[3] / n;                 // mask_mse        metrics.py:96-97
        o[4] = s[4] / n;                 // mask_mae        metrics.py:99-100
        o[5] = sqrt(s[3] / n);           // mask_rmse       metrics.py:102-103
        o[6] = sqrt(s[5] / n);           // mask_rmse_log   metrics.py:105-109
        o[7] = s[6] / n;                 // mask_accuracy_k metrics.py:112-121, k = 1
        o[8] = s[7] / n;                 //                 k = 2
        o[9] = s[8] / n;                 //                 k = 3
        o[10] = s[10] / s[9];            // mask_mse_w_conf metrics.py:123-124
        o[11] = s[11] / s[9];            // mask_mae_w_conf metrics.py:126-127
    }
}

// matplotlib's 'jet' (the colour map of test.py:129, TRS.py:46): LinearSegmentedColormap, 256 entries, each the
// piecewise-linear segment data evaluated at i/255; as uint8 after the scripts' `255 * rgb` + astype(uint8) truncation
static void jet_lut_u8(uint8_t lut[768]) {
    static const double R[][2] = {{0.0, 0.0}, {0.35, 0.0}, {0.66, 1.0}, {0.89, 1.0}, {1.0, 0.5}};
    static const double G[][2] = {{0.0, 0.0}, {0.125, 0.0}, {0.375, 1.0}, {0.64, 1.0}, {0.91, 0.0}, {1.0, 0.0}};
    static const double Bl[][2] = {{0.0, 0.5}, {0.11, 1.0}, {0.34, 1.0}, {0.65, 0.0}, {1.0, 0.0}};
    auto ev = [](const double (*seg)[2], int n, double x) {
#pragma clang fp contract(off)   // NumPy multiplies, then adds: no fused multiply-add
        // matplotlib.colors._create_lookup_table: break points and sample positions are both scaled by N-1 = 255 first
        for (int i = 1; i < n; ++i)
            if (x <= seg[i][0] * 255.0) {
                const double f = (x - seg[i - 1][0] * 255.0) / (seg[i][0] * 255.0 - seg[i - 1][0] * 255.0);
                return f * (seg[i][1] - seg[i - 1][1]) + seg[i - 1][1];
            }
        return seg[n - 1][1];
    };
    for (int i = 0; i < 256; ++i) {
        const double x = 255.0 * (i == 255 ? 1.0 : (double)i * (1.0 / 255.0));   // (N-1) * np.linspace(0, 1, N)[i]
        double c[3] = {ev(R, 5, x), ev(G, 6, x), ev(Bl, 5, x)};
        for (int k = 0; k < 3; ++k) {
            double v = c[k] < 0 ? 0 : (c[k] > 1 ? 1 : c[k]);
            lut[i * 3 + k] = (uint8_t)(255.0 * v);
        }
    }
}

}  // namespace dffw

using namespace dffw;

#define IO_HIPCHK(expr)                                                                                \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return dffw_fail(DFFW_EHIP, "%s -> %s", #expr, hipGetErrorString(_e));   \
    } while (0)

extern "C" {

int dffw_jet_lut(uint8_t *lut768) {
    if (!lut768) return dffw_fail(DFFW_EINVAL, "null argument");
    jet_lut_u8(lut768);
    return DFFW_OK;
}

int dffw_pack_stack(int device, const void *raw, int dtype, const int64_t strides[5], int B, int N, int h, int w, int Hp, int Wp,
                    float *FS, void *hip_stream) {
    if (!raw || !strides || !FS) return dffw_fail(DFFW_EINVAL, "null argument");
    const bool norm64 = (dtype & DFFW_RAW_NORM_F64) != 0;
    dtype &= ~DFFW_RAW_NORM_F64;
    if (dtype != DFFW_RAW_U8 && dtype != DFFW_RAW_F32) return dffw_fail(DFFW_EINVAL, "unknown raw dtype %d", dtype);
    if (B < 1 || N < 1 || h < 1 || w < 1) return dffw_fail(DFFW_EINVAL, "empty stack (B=%d N=%d h=%d w=%d)", B, N, h, w);
    if (Hp < h || Wp < w || Hp % 32 || Wp % 32)
        return dffw_fail(DFFW_EINVAL, "padded size %dx%d must cover %dx%d and be a multiple of 32 (DEN.py down-samples 5 times)", Hp, Wp, h, w);
    IO_HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    const int64_t total = (int64_t)B * 3 * N * Hp * (Wp / 4);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
#define DFFW_PACK(T, N64)                                                                                                        \
    hipLaunchKernelGGL((pack_stack_kernel<T, N64>), dim3(grid), dim3(256), 0, s, (const T *)raw, strides[0], strides[1], strides[2], strides[3], \
                       strides[4], B, N, h, w, Hp, Wp, FS)
    if (dtype == DFFW_RAW_U8) {
        if (norm64) DFFW_PACK(uint8_t, true);
        else DFFW_PACK(uint8_t, false);
    } else {
        if (norm64) DFFW_PACK(float, true);
        else DFFW_PACK(float, false);
    }
#undef DFFW_PACK
    IO_HIPCHK(hipGetLastError());
    return DFFW_OK;
}

int dffw_colorize(int device, const float *depth, int B, int H, int W, int h, int w, int mode, float lo, float hi, float *range,
                  uint8_t *rgb, void *hip_stream) {
    if (!depth || !rgb || !range) return dffw_fail(DFFW_EINVAL, "null argument");
    if (B < 1 || h < 1 || w < 1 || h > H || w > W) return dffw_fail(DFFW_EINVAL, "crop %dx%d does not fit the %dx%d map", h, w, H, W);
    if (mode != DFFW_RANGE_FIXED && mode != DFFW_RANGE_MINMAX) return dffw_fail(DFFW_EINVAL, "unknown range mode %d", mode);
    IO_HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    // the 768-byte colour table lives in device memory once per device (never freed: process lifetime)
    static uint8_t *lut_dev[64] = {};
    if (device < 0 || device >= 64) return dffw_fail(DFFW_EINVAL, "device %d", device);
    if (!lut_dev[device]) {
        uint8_t lut[768];
        jet_lut_u8(lut);
        uint8_t *p = nullptr;
        IO_HIPCHK(hipMalloc((void **)&p, 768));
        IO_HIPCHK(hipMemcpy(p, lut, 768, hipMemcpyHostToDevice));
        lut_dev[device] = p;
    }
    if (mode == DFFW_RANGE_MINMAX) {
        hipLaunchKernelGGL(minmax_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (int *)range, B);
        const int64_t hw = (int64_t)H * W;
        hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)std::min<int64_t>((hw + 255) / 256, 128), B), dim3(256), 0, s, depth, hw, (int *)range);
        hipLaunchKernelGGL(minmax_finish_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, s, (int *)range, B);
    } else {
        hipLaunchKernelGGL(range_set_kernel, dim3((B + 63) / 64), dim3(64), 0, s, range, B, lo, hi);
    }
    const int64_t total = (int64_t)B * h * w;
    hipLaunchKernelGGL(colorize_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 256 * 16)), dim3(256), 0, s, depth, H, W, h, w, range, 2,
                       lut_dev[device], rgb, B);
    IO_HIPCHK(hipGetLastError());
    return DFFW_OK;
}

int dffw_unpack_stack(int device, const float *warp, int B, int N, int H, int W, int h, int w, uint8_t *images, void *hip_stream) {
    if (!warp || !images) return dffw_fail(DFFW_EINVAL, "null argument");
    if (B < 1 || N < 1 || h < 1 || w < 1 || h > H || w > W) return dffw_fail(DFFW_EINVAL, "crop %dx%d does not fit the %dx%d stack", h, w, H, W);
    IO_HIPCHK(hipSetDevice(device));
    const int64_t total = (int64_t)B * N * h * w;
    hipLaunchKernelGGL(unpack_stack_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32)), dim3(256), 0, (hipStream_t)hip_stream, warp,
                       B, N, H, W, h, w, images);
    IO_HIPCHK(hipGetLastError());
    return DFFW_OK;
}

int64_t dffw_metrics_scratch_bytes(int B) { return (int64_t)B * 64 * NSUM * (int64_t)sizeof(double); }

int dffw_metrics(int device, const float *est, int B, int H, int W, const float *gt, const uint8_t *mask, const float *conf, int h, int w,
                 double *out, void *scratch, int64_t scratch_bytes, void *hip_stream) {
    if (!est || !gt || !mask || !out || !scratch) return dffw_fail(DFFW_EINVAL, "null argument");
    if (B < 1 || h < 1 || w < 1 || h > H || w > W) return dffw_fail(DFFW_EINVAL, "crop %dx%d does not fit the %dx%d map", h, w, H, W);
    if (scratch_bytes < dffw_metrics_scratch_bytes(B)) return dffw_fail(DFFW_ENOMEM, "metrics scratch too small");
    IO_HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    const int nblk = 64;
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(nblk, B), dim3(256), 0, s, est, H, W, gt, mask, conf, h, w, (double *)scratch);
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(B), dim3(64), 0, s, (const double *)scratch, nblk, out);
    IO_HIPCHK(hipGetLastError());
    return DFFW_OK;
}

}  // extern "C"

// ---- dffw_probe_peaks: what this very GPU sustains, measured the same way bench.py measures the kernels (HIP events) -----------
// The roofline fractions are quoted against the datasheet (2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E).  A chip clocks to its
// power budget, so the sustained ceilings are lower and differ between boxes; this probe reports them for the box the bench runs
// on: (a) v_mfma_f32_16x16x32_bf16 issued back to back out of registers (10 independent accumulators, 3 waves per SIMD, no
// memory traffic at all), (b) float4 streaming: a copy of 1 GiB (read + write counted) and a pure read of 2 GiB, the better of the two.
namespace dffw {
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(4))) float probe_f32x4;

__global__ __launch_bounds__(256) void probe_mfma_kernel(float *out, int iters, unsigned seed) {
    probe_bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const unsigned h = (threadIdx.x * 8 + i) * 2654435761u + seed;     // random-looking operands: zeros would clock higher
        a[i] = (__bf16)((float)((h >> 8) & 255) * 0.01f - 1.f);
        b[i] = (__bf16)((float)((h >> 16) & 255) * 0.01f - 1.f);
    }
    probe_f32x4 acc[10];
    for (int k = 0; k < 10; ++k) acc[k] = probe_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 10; ++k) s += acc[k][0] + acc[k][3];
    if (s == 12345.678f) out[threadIdx.x] = s;    // never true: keeps the loop alive
}

// MODE 0: copy (read + write), 1: read only (sum kept alive), four 16-byte accesses per lane in flight per iteration
template <int MODE>
__global__ __launch_bounds__(256) void probe_stream_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n) {
    // a workgroup walks contiguous 16 KiB blocks (4 x 256 lanes x 16 B), blocks dealt round-robin over the grid
    const int64_t nblk = n / 1024;
    float acc = 0.f;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t i = blk * 1024 + threadIdx.x;
        const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        if constexpr (MODE == 0) {
            dst[i] = a;
            dst[i + 256] = b;
            dst[i + 512] = c;
            dst[i + 768] = d;
        } else {
            acc += a.x + b.y + c.z + d.w;
        }
    }
    if (MODE == 1 && acc == 12345.678f) dst[0] = make_float4(acc, 0.f, 0.f, 0.f);
}
}  // namespace dffw

extern "C" int dffw_probe_peaks(int device, float *mfma_tflops, float *hbm_gbs, void *hip_stream) {
    if (!mfma_tflops || !hbm_gbs) return dffw_fail(DFFW_EINVAL, "null argument");
    IO_HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    // everything acquired here is released on every exit path
    struct Guard {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        char *buf = nullptr;
        ~Guard() {
            if (buf) (void)hipFree(buf);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } gd;
    IO_HIPCHK(hipEventCreate(&gd.e0));
    IO_HIPCHK(hipEventCreate(&gd.e1));
    const int64_t bytes = (int64_t)1 << 30;
    IO_HIPCHK(hipMalloc((void **)&gd.buf, 2 * bytes));
    char *buf = gd.buf;
    IO_HIPCHK(hipMemsetAsync(buf, 1, 2 * bytes, s));
    float best_m = 1e30f, best_c = 1e30f, best_r = 1e30f, ms = 0.f;
    const int iters = 4000, blocks = 256 * 3;
    for (int rep = 0; rep < 4; ++rep) {   // first repetition = warm-up
        IO_HIPCHK(hipEventRecord(gd.e0, s));
        hipLaunchKernelGGL(dffw::probe_mfma_kernel, dim3(blocks), dim3(256), 0, s, (float *)buf, iters, 7u + rep);
        IO_HIPCHK(hipGetLastError());
        IO_HIPCHK(hipEventRecord(gd.e1, s));
        IO_HIPCHK(hipEventSynchronize(gd.e1));
        IO_HIPCHK(hipEventElapsedTime(&ms, gd.e0, gd.e1));
        if (rep && ms < best_m) best_m = ms;
        IO_HIPCHK(hipEventRecord(gd.e0, s));
        hipLaunchKernelGGL(dffw::probe_stream_kernel<0>, dim3(256 * 8), dim3(256), 0, s, (const float4 *)buf, (float4 *)(buf + bytes), bytes / 16);
        IO_HIPCHK(hipGetLastError());
        IO_HIPCHK(hipEventRecord(gd.e1, s));
        IO_HIPCHK(hipEventSynchronize(gd.e1));
        IO_HIPCHK(hipEventElapsedTime(&ms, gd.e0, gd.e1));
        if (rep && ms < best_c) best_c = ms;
        IO_HIPCHK(hipEventRecord(gd.e0, s));
        hipLaunchKernelGGL(dffw::probe_stream_kernel<1>, dim3(256 * 8), dim3(256), 0, s, (const float4 *)buf, (float4 *)(buf + bytes), 2 * bytes / 16);
        IO_HIPCHK(hipGetLastError());
        IO_HIPCHK(hipEventRecord(gd.e1, s));
        IO_HIPCHK(hipEventSynchronize(gd.e1));
        IO_HIPCHK(hipEventElapsedTime(&ms, gd.e0, gd.e1));
        if (rep && ms < best_r) best_r = ms;
    }
    if (!(best_m < 1e29f && best_c < 1e29f && best_r < 1e29f) || best_m <= 0.f || best_c <= 0.f || best_r <= 0.f)
        return dffw_fail(DFFW_EHIP, "probe kernels were not timed");
    *mfma_tflops = (float)((double)blocks * 4 * iters * 10 * (2.0 * 16 * 16 * 32) / (best_m * 1e-3) / 1e12);
    // the better of the two streaming forms: copy of 1 GiB (read + write counted) and a pure read of 2 GiB
    *hbm_gbs = (float)std::max(2.0 * (double)bytes / (best_c * 1e-3) / 1e9, 2.0 * (double)bytes / (best_r * 1e-3) / 1e9);
    return DFFW_OK;
}
