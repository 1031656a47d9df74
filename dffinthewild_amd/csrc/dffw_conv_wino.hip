// conv_wino32: a 3x3x3 stride-1 padding-1 conv over 32 input channels with the in-plane 3x3 taps in Winograd F(2x2, 3x3) form
// (the reference's nn.Conv3d + folded BatchNorm3d of the 32-channel hourglass / SPP layers, Depth_Estimation_Network/DEN.py:5-11, 96-120
// and submodule.py:117-160).  16 MFMA multiplies per 2x2 outputs and slice tap instead of 36:
//     U = G g G^T        per (cout, cin, dz): done once at weight-pack time in float64, split into bf16 hi + lo (pack_conv)
//     V = B^T d B        per 4x4 input patch: fp32 from the stored hi + lo activations, re-split into hi + lo         (phase T)
//     M[pos] += U[pos][dz] . V[pos][z + dz - 1]    three MFMA products each, fp32 accumulation, 16 positions          (phase M)
//     y = A^T M A        fp32, then the usual epilogue (BatchNorm shift, residual, ReLU, hi + lo split)               (phase O)
// The arithmetic is not the direct kernels' (sums are re-associated by the transforms), so results differ from conv_tile's in the
// last bits; tools/winograd_emulation.py measured pred3 on the nine goldens at 0.97e-5 .. 4.2e-5 rel-L2 (direct: 0.65e-5 .. 3.7e-5).
//
// One workgroup of 8 waves walks one column of 4 x 16 output pixels (16 blocks of 2 x 2) through the slices, slice-stationary: input
// slice z is transformed once and contributes to the three output slices z+1, z, z-1 (accumulator sets A0, A1, A2, rotated every
// step), so LDS holds V of ONE slice (16 positions x 16 blocks x 32 channels x hi + lo = 48 KB at a 96-byte pitch, conflict-free
// ds_read_b128) plus the 32 KB hand-over of finished M values.  Wave w owns positions 2w and 2w+1 for all 16 blocks: its filter
// fragments U[2][3 dz][2 nt][hi, lo] = 96 VGPRs stay in registers for the whole column, and a step is 36 MFMAs per wave for 2 LDS
// fragment reads.  Phases of a step, two workgroup barriers:
//     VALU phase: all 512 threads (16 blocks x 4 transform rows x 8 channel quads) turn the prefetched slice into V, then issue the next
//                 slice's 16 loads (in flight across the MFMA phase); threads 0-255 (16 blocks x 2 output rows x 8 output-channel quads)
//                 then run the output transform + epilogue of the slice finished one step earlier
//     MFMA phase: all 8 waves contract, write the finished accumulator set to the hand-over buffer, rotate.
#include "dffw_conv_wino.h"
#include "dffw_device.h"

#include <cstdio>

namespace dffw {
namespace {

constexpr int WB = 16;                          // blocks per column step (2 rows x 8)
constexpr int W_PITCH = 96;                     // bytes per (position, block) in one V plane: 32 channels x 2 B, padded from 64
constexpr int W_VPLANE = 16 * WB * W_PITCH;     // one part (hi or lo) of V
constexpr int W_MFLOATS = 16 * WB * 32;         // hand-over buffer: [position][block][32 output channels] fp32
constexpr int W_FX = 18, W_FPIX = 6 * W_FX;       // a slice's input footprint: (4 + 2) x (16 + 2) pixels
constexpr int W_RAWSLOT = 16384;                // ... as 128-byte records, rounded up to 16 wave-instructions of 1 KB; two slots
constexpr int W_LDS = 2 * W_VPLANE + W_MFLOATS * 4 + 2 * W_RAWSLOT;

__device__ __forceinline__ void wino_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

template <int PREC>
__global__ __launch_bounds__(512) void conv_wino32(const ConvArgs a, const WinoArgs t) {
    static_assert(PREC == P_BF16X3, "the Winograd path exists for the split-bf16 storage only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *V = smem;
    float *Mb = reinterpret_cast<float *>(smem + 2 * W_VPLANE);
    unsigned char *raw = smem + 2 * W_VPLANE + W_MFLOATS * 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int slab = blockIdx.y;
    int u = blockIdx.x;
    const int tx = u % t.tiles_x;
    u /= t.tiles_x;
    const int ty = u % t.tiles_y;
    const int bs = u / t.tiles_y;
    const int y0 = ty * WINO_TY, x0 = tx * WINO_TX;
    const int N = a.Ni, H = a.Hi, W = a.Wi;

    // ---- the wave's filter fragments and accumulators -----------------------------------------------------------------
    short8 U[2][3][2][2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    U[pp][dz][nt][part] = *reinterpret_cast<const short8 *>(
                        t.u + ((((((size_t)slab * 16 + (2 * wave + pp)) * 3 + dz) * 2 + nt) * 2 + part) * 512) + lane * 8);
    f32x4 acc[2][3][2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[pp][k][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // MFMA-phase LDS addresses: the wave's V fragments (positions 2w, 2w+1; block r, channel octet g) and its hand-over rows
    const unsigned vrd = lds0 + ((2 * wave) * WB + r) * W_PITCH + g * 16;
    const unsigned mwr = lds0 + 2 * W_VPLANE + (((2 * wave) * WB + r) * 32 + g * 4) * 4;

    // ---- phase T role (all 512 threads): (block, transform row xi, channel quad) ---------------------------------------
    const int cq = tid & 7, xi = (tid >> 3) & 3, tb = tid >> 5;
    const int tby = tb >> 3, tbx = tb & 7;
    // row xi of B^T has two non-zeros: rows (ia, ib) of the patch with signs (sa, sb)
    const int ia = xi == 0 ? 0 : 1, ib = xi == 3 ? 3 : 2;
    const float sa = xi == 2 ? -1.f : 1.f, sb = (xi == 0 || xi == 3) ? -1.f : 1.f;
    // the slice's 6 x 18 pixel footprint sits in LDS as whole 128-byte records [hi 32 | lo 32] x 2 B (out-of-volume pixels = zeros)
    const int ra = ((2 * tby + ia) * W_FX + 2 * tbx) * 128 + cq * 8, rb = ((2 * tby + ib) * W_FX + 2 * tbx) * 128 + cq * 8;

    // ---- slice fill by LDS-DMA: 108 pixels x 8 pieces of 16 B, two rounds of 512 lanes; one wave instruction = 1 KB of the slot ------
    uint32_t foff[2];
    bool fok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = k * 512 + tid, pix = p >> 3, fy = pix / W_FX, fx = pix - fy * W_FX;
        const int iy = y0 - 1 + fy, ix = x0 - 1 + fx;
        fok[k] = pix < W_FPIX && iy >= 0 && iy < H && ix >= 0 && ix < W;
        foff[k] = (((uint32_t)(bs * N) * H + iy) * W + ix) * 128u + (p & 7) * 16u;   // (the launcher checks the volume is < 4 GB)
    }
    const uint32_t slice_b = (uint32_t)H * W * 128u;
    const unsigned char *inb = reinterpret_cast<const unsigned char *>(a.in0);
    auto fill = [&](int z) {
        unsigned char *slot = raw + (z & 1) * W_RAWSLOT;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const void *src = fok[k] ? (const void *)(inb + (foff[k] + (uint32_t)z * slice_b)) : (const void *)a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + (k * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    if (N > 0) fill(0);
    // (the builtin, not asm: hipcc's wait-count pass must see that the filter fragments have arrived, or it waits vmcnt(0) -- i.e. for the
    // slice in flight -- in front of every step's first MFMA)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    asm volatile("s_barrier" ::: "memory");

    // ---- phase O role (threads 0-255): (block, output row, output-channel quad) ----------------------------------------
    const int oq = tid & 7, orow = (tid >> 3) & 1, ob = (tid >> 4) & 15;
    const int oby = ob >> 3, obx = ob & 7;

    StepTrace tr((blockIdx.x < 512 && blockIdx.y == 0) ? a.trace : nullptr, wave, lane, 8);
    tr.no_skip();
    for (int s = 0; s <= N + 1; ++s) {
        tr.stamp(0);
        if (s < N) {
            float dp[4][4];
            const unsigned char *rs = raw + (s & 1) * W_RAWSLOT;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 ah = *reinterpret_cast<const uint2 *>(rs + ra + j * 128), al = *reinterpret_cast<const uint2 *>(rs + ra + j * 128 + 64);
                const uint2 bh = *reinterpret_cast<const uint2 *>(rs + rb + j * 128), bl = *reinterpret_cast<const uint2 *>(rs + rb + j * 128 + 64);
                float fa[4], fb[4];
                Fmt<PREC>::join2(ah.x, al.x, fa[0], fa[1]);
                Fmt<PREC>::join2(ah.y, al.y, fa[2], fa[3]);
                Fmt<PREC>::join2(bh.x, bl.x, fb[0], fb[1]);
                Fmt<PREC>::join2(bh.y, bl.y, fb[2], fb[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) dp[j][c] = sa * fa[c] + sb * fb[c];
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[c] = nu == 0 ? dp[0][c] - dp[2][c] : nu == 1 ? dp[1][c] + dp[2][c] : nu == 2 ? dp[2][c] - dp[1][c] : dp[1][c] - dp[3][c];
                uint2 vh, vl;
                Fmt<PREC>::split2(v[0], v[1], vh.x, vl.x);
                Fmt<PREC>::split2(v[2], v[3], vh.y, vl.y);
                unsigned char *dst = V + ((xi * 4 + nu) * WB + tb) * W_PITCH + cq * 8;
                *reinterpret_cast<uint2 *>(dst) = vh;
                *reinterpret_cast<uint2 *>(dst + W_VPLANE) = vl;
            }
        }
        tr.stamp(1);
        if (tid < 256 && s >= 2) {
            const int zo = s - 2;
            const int oy = y0 + 2 * oby + orow, ox = x0 + 2 * obx;
            const int64_t pix = (((int64_t)bs * N + zo) * H + oy) * W + ox;
            const int C = a.Cout, co = slab * 32 + oq * 4;
            f32x4 yv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int x = orow + e;   // row of M this output row uses: A^T row 0 = (1, 1, 1, 0), row 1 = (0, 1, -1, -1)
                f32x4 m[4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) m[nu] = *reinterpret_cast<const f32x4 *>(Mb + ((x * 4 + nu) * WB + ob) * 32 + oq * 4);
                const float sg = (orow == 1 && e > 0) ? -1.f : 1.f;
                yv[0] += sg * (m[0] + m[1] + m[2]);
                yv[1] += sg * (m[1] - m[2] - m[3]);
            }
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + co);
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                f32x4 v = yv[px] + bv;
                const int64_t off = (pix + px) * 2 * C + co;
                if (a.res0) {
                    const uint2 rh = *reinterpret_cast<const uint2 *>(a.res0 + off), rl = *reinterpret_cast<const uint2 *>(a.res0 + off + C);
                    float r0, r1;
                    Fmt<PREC>::join2(rh.x, rl.x, r0, r1); v[0] += r0; v[1] += r1;
                    Fmt<PREC>::join2(rh.y, rl.y, r0, r1); v[2] += r0; v[3] += r1;
                }
                if (a.relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                }
                uint2 vh, vl;
                Fmt<PREC>::split2(v[0], v[1], vh.x, vl.x);
                Fmt<PREC>::split2(v[2], v[3], vh.y, vl.y);
                *reinterpret_cast<uint2 *>(a.out + off) = vh;
                *reinterpret_cast<uint2 *>(a.out + off + C) = vl;
            }
        }
        tr.stamp(2);
        if (s + 1 < N) fill(s + 1);   // in flight across the MFMA phase
        wino_barrier();
        tr.stamp(3);
        if (s <= N) {
            if (s < N) {
                // (inline asm: behind an outstanding LDS-DMA hipcc puts vmcnt(0) in front of every LDS access, which would wait for the slice
                // just requested instead of letting it land during the MFMAs)
                short8 xh[2], xl[2];
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(xh[0]), "=&v"(xl[0]), "=&v"(xh[1]), "=&v"(xl[1])
                             : "v"(vrd), "n"(W_VPLANE), "n"(WB * W_PITCH), "n"(WB * W_PITCH + W_VPLANE));
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            f32x4 c = acc[pp][dz][nt];
                            c = mma<false>(U[pp][dz][nt][1], xh[pp], c);
                            c = mma<false>(U[pp][dz][nt][0], xl[pp], c);
                            c = mma<false>(U[pp][dz][nt][0], xh[pp], c);
                            acc[pp][dz][nt] = c;
                        }
            }
            // set 2 (filter slice dz = 2 applied to input slice s) completes output slice s - 1
            if (s >= 1) {
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:64\n\tds_write_b128 %0, %3 offset:%5\n\tds_write_b128 %0, %4 offset:%6"
                             ::"v"(mwr), "v"(acc[0][2][0]), "v"(acc[0][2][1]), "v"(acc[1][2][0]), "v"(acc[1][2][1]), "n"(WB * 32 * 4), "n"(WB * 32 * 4 + 64));
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[pp][2][nt] = acc[pp][1][nt];
                    acc[pp][1][nt] = acc[pp][0][nt];
                    acc[pp][0][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
        tr.stamp(4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the next slice has landed, for every wave
        tr.stamp(5);
        tr.next();
    }
}

hipError_t launch_conv_wino32(int prec, const ConvArgs &a, const WinoArgs &t, hipStream_t s) {
    if ((int64_t)a.B * a.Ni * a.Hi * a.Wi * 128 >= (int64_t)1 << 32) return hipErrorInvalidValue;
    if (prec != P_BF16X3 || a.C0 != 32 || a.C1 != 0 || a.Cout % 32 || a.Hi % WINO_TY || a.Wi % WINO_TX || !a.out) return hipErrorInvalidValue;
    auto k = conv_wino32<P_BF16X3>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const dim3 grid((unsigned)(a.B * t.tiles_y * t.tiles_x), (unsigned)(a.Cout / 32));
    hipLaunchKernelGGL(k, grid, dim3(512), W_LDS, s, a, t);
    return hipGetLastError();
}

void conv_wino32_kernel_name(int prec, const ConvArgs &, char *buf, int n) { snprintf(buf, n, "dffw::conv_wino32<%d>", prec); }

}  // namespace dffw
