// stem_pipe: the stem in pixel-pair form (conv 1x9x9 dil (1,2,2) pad (0,8,8), 3 -> 8 channels + BN + ReLU, DEN.py:131-143) as a PERSISTENT,
// software-pipelined kernel (round 4), gfx950 / MI355X.
//
// conv_tile<.., G2P, 1, 1, 32, 32, 8, 0, 8, ..> runs one 32 x 32 tile per workgroup: load the 48 x 48 x 3 fp32 footprint, split it to
// paired-pixel records in LDS, barrier, contract (weight fragments from L2, one chunk ahead), store, exit.  Its phase ablation
// (tools/ablate_layers.sh, r04): fill alone 0.14 ms + contraction 0.20 + epilogue 0.08 = the 0.41 ms measured -- additive, although the
// contraction by itself sits at the matrix pipe's sustained rate.  Here a workgroup walks many tiles:
//   * the next tile's footprint is requested into REGISTERS (one float4 per colour plane and thread) before the current tile's
//     contraction and converted / written to LDS after it: its L2 / HBM latency is covered by the workgroup's own MFMAs;
//   * the filter (12 chunks x hi/lo = 24 KiB of MFMA A-fragments) is staged in LDS once per workgroup: no weight stream through the
//     vector-memory queue, so the only loads in the loop are the prefetch (vmcnt retires in order: a weight-fragment wait would drain it);
//   * no LDS-DMA in the kernel: every wait is hipcc's own.
// Arithmetic, operation order and record packing are conv_tile's (bit-identical results; tested): per chunk and operand tile
// acc = w_lo x_hi, + w_hi x_lo, + w_hi x_hi into one accumulator initialised with the BatchNorm shift.
// LDS: one image of 48 x 24 records (hi + lo planes, 36 KiB) + the filter (24 KiB) = 60 KiB, two workgroups per CU.
#include <algorithm>
#include <cstdio>

#include "dffw_conv_geom.h"
#include "dffw_device.h"
#include "dffw_stem.h"

namespace dffw {

namespace stemp {
constexpr int TY = 32, TX = 32, NW = 8, KC = 12;
using T = TileT<G2P, 1, TY, TX, 8>;
using G = GeoT<G2P>;
constexpr int PIXB = 16;
constexpr int PLANEB = (T::FPIX * PIXB + 1023) / 1024 * 1024;
constexpr int IMGB = 2 * PLANEB;
constexpr int WB = KC * 2 * 1024;             // filter fragments: [chunk][part][64 lanes][16 bytes]
constexpr int MTW = T::MT / NW;               // operand tiles (16 pixel pairs) per wave
constexpr int QR = T::FXL / 2, NQ = T::FY * QR;   // quads of 4 consecutive pixels = two records each
constexpr int NITQ = (NQ + NW * 64 - 1) / (NW * 64);
static_assert(T::FY == 48 && T::FXL == 24 && MTW == 4 && NITQ == 2, "the 32 x 32 pair tile");
}   // namespace stemp

template <bool RELU>
__global__ __launch_bounds__(512) void stem_pipe(const ConvArgs a, const TileArgs t) {
    using namespace stemp;
    constexpr int PREC = P_BF16X3;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[IMGB + WB];
    unsigned char *const wlds = smem + IMGB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;

    // ---- this workgroup's tiles: XCD x (= blockIdx % 8) owns a contiguous range of the tile sequence (x fastest, then y, slice, sample)
    // and its workgroups take them round-robin ----
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int tfirst, tend;
    {
        const int q = t.total_tiles >> 3, rem = t.total_tiles & 7;
        const int xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        tend = xs + q + (xcd < rem ? 1 : 0);
        tfirst = xs + widx;
    }
    if (tfirst >= tend) return;
    struct Coord {
        int b, gz0, gy0, gx0;
    };
    auto decode = [&](int tile) {
        Coord c;
        const int txi = tile % t.tiles_x;
        int tt = tile / t.tiles_x;
        const int tyi = tt % t.tiles_y;
        tt /= t.tiles_y;
        c.gz0 = tt % t.tiles_z;
        c.b = tt / t.tiles_z;
        c.gy0 = tyi * TY;
        c.gx0 = txi * TX;
        return c;
    };

    // ---- per-lane constants of the wave's operand tiles (conv_tile's pair form): pair k of a row = pixels (x, x+2) with
    // x = (k & 1) + 4 * (k >> 1); lane rows 0-1 end up with pixel x, rows 2-3 with pixel x+2, each as "row g & 1" of its own record ----
    int pofs[MTW], voff[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (wave * MTW + j) * 16 + r;
        constexpr int HP = TX / 2;
        const int k = p % HP, ty = p / HP;
        const int tx = (k & 1) + 4 * (k >> 1) + 2 * (g >> 1);
        pofs[j] = (ty * T::FXL + k) * PIXB;
        voff[j] = (ty * a.Wo + tx) * 16 + (g & 1) * 8;
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + g * 4);
    // tap offsets of this lane group, all twelve chunks (the table is tiny and the same for every tile)
    int toff[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) toff[kc] = t.tab[0][kc * 4 + g];
    // the filter into LDS, once: 24 KiB = 3 x 16 bytes per thread
    {
        const uint4 *wsrc = reinterpret_cast<const uint4 *>(t.wpk[0]);
#pragma unroll
        for (int i = 0; i < WB / 16 / (NW * 64); ++i) reinterpret_cast<uint4 *>(wlds)[tid + i * NW * 64] = wsrc[tid + i * NW * 64];
    }

    // ---- footprint of a tile: quad p2 = 4 consecutive, 16-byte aligned pixels of one row = the records at packed columns 2m, 2m+1 ----
    const int W = a.Wi - 2;
    const int64_t plane = (int64_t)a.Ni * a.Hi * W;
    f32x4 q0[NITQ], q1[NITQ], q2[NITQ];
    bool qin[NITQ];
    auto load_tile = [&](const Coord &c) {
        const float *src = a.fs32 + (int64_t)c.b * 3 * plane + (int64_t)c.gz0 * a.Hi * W;
        const int iy0 = c.gy0 + G::MINY;
#pragma unroll
        for (int it = 0; it < NITQ; ++it) {
            const int p2 = tid + it * NW * 64;
            const int fy = p2 / QR, m = p2 - fy * QR;
            const int iy = iy0 + fy, x = c.gx0 - 8 + 4 * m;
            qin[it] = p2 < NQ && (unsigned)iy < (unsigned)a.Hi && (unsigned)x < (unsigned)W;
            q0[it] = q1[it] = q2[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qin[it]) {
                const float *sp = src + (int64_t)iy * W + x;
                q0[it] = *reinterpret_cast<const f32x4 *>(sp);
                q1[it] = *reinterpret_cast<const f32x4 *>(sp + plane);
                q2[it] = *reinterpret_cast<const f32x4 *>(sp + 2 * plane);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < NITQ; ++it) {
            const int p2 = tid + it * NW * 64;
            if (p2 >= NQ) break;
            const int fy = p2 / QR, m = p2 - fy * QR;
            short8 ha = short8{0, 0, 0, 0, 0, 0, 0, 0}, la = ha, hb = ha, lb = ha;
            if (qin[it]) {
                // record = [c0 c1 c2 0] of its first pixel | [c0 c1 c2 0] of its second (conv_tile's fill_from_stack, same packing and rounding)
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                const f32x4 c0 = q0[it], c1 = q1[it], c2 = q2[it];
                uint32_t h[4], l[4];
                Fmt<PREC>::split2(c0[0], c1[0], h[0], l[0]);
                Fmt<PREC>::split2(c2[0], 0.f, h[1], l[1]);
                Fmt<PREC>::split2(c0[2], c1[2], h[2], l[2]);
                Fmt<PREC>::split2(c2[2], 0.f, h[3], l[3]);
                ha = __builtin_bit_cast(short8, (u32x4v){h[0], h[1], h[2], h[3]});
                la = __builtin_bit_cast(short8, (u32x4v){l[0], l[1], l[2], l[3]});
                Fmt<PREC>::split2(c0[1], c1[1], h[0], l[0]);
                Fmt<PREC>::split2(c2[1], 0.f, h[1], l[1]);
                Fmt<PREC>::split2(c0[3], c1[3], h[2], l[2]);
                Fmt<PREC>::split2(c2[3], 0.f, h[3], l[3]);
                hb = __builtin_bit_cast(short8, (u32x4v){h[0], h[1], h[2], h[3]});
                lb = __builtin_bit_cast(short8, (u32x4v){l[0], l[1], l[2], l[3]});
            }
            unsigned char *dst = smem + (fy * T::FXL + 2 * m) * PIXB;
            *reinterpret_cast<short8 *>(dst) = ha;
            *reinterpret_cast<short8 *>(dst + PIXB) = hb;
            *reinterpret_cast<short8 *>(dst + PLANEB) = la;
            *reinterpret_cast<short8 *>(dst + PLANEB + PIXB) = lb;
        }
    };

    Coord cur = decode(tfirst);
    load_tile(cur);
    store_tile();
    __syncthreads();
    for (int tile = tfirst; tile < tend; tile += wgs_per_xcd) {
        const int nxt = tile + wgs_per_xcd;
        const bool more = nxt < tend;
        Coord cn = cur;
        if (more) {
            cn = decode(nxt);
            load_tile(cn);   // in flight under this tile's contraction
        }
        // ---- contraction: 12 chunks x 4 operand tiles, filter fragments from LDS ----
        // Chunks 0-8 are the filter rows ky (K octet g = pair column 2g), chunks 9-11 the last pair column (the engine's pair-form pack).  The wave's operand
        // tiles are the rows ty .. ty + 3 of one 16-pair column block, and the dilation is 2: the fragment of (row ty + 2, chunk ky) IS the fragment of
        // (row ty, chunk ky + 1), so rows j and j + 2 share their loads -- 10 instead of 18 per row pair, 88 instead of 120 ds_read_b128 per wave and tile on a
        // kernel whose busiest unit is the LDS port (0.65; matrix pipe 0.59).  Every accumulator still takes its chunks in the order 0 ... 11 and the three
        // products of a chunk in conv_tile's order: bit-identical to the table walk.
        f32x4 acc[MTW];
#pragma unroll
        for (int j = 0; j < MTW; ++j) acc[j] = bias4;
        static_assert(MTW == 4 && KC == 12, "row pairs (j, j + 2), nine filter rows + three chunks of the last pair column");
        {
            short8 wph = short8{0, 0, 0, 0, 0, 0, 0, 0}, wpl = wph;
#pragma unroll
            for (int ky = 0; ky <= 9; ++ky) {
                short8 whi = wph, wlo = wpl;
                if (ky < 9) {
                    whi = *reinterpret_cast<const short8 *>(wlds + (ky * 2 + 0) * 1024 + lane * 16);
                    wlo = *reinterpret_cast<const short8 *>(wlds + (ky * 2 + 1) * 1024 + lane * 16);
                }
                const int tofs = ky < 9 ? toff[ky] : toff[8] + 2 * T::FXL * PIXB;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const unsigned char *lp = smem + pofs[jj] + tofs;
                    const short8 xh = *reinterpret_cast<const short8 *>(lp);
                    const short8 xl = *reinterpret_cast<const short8 *>(lp + PLANEB);
                    if (ky < 9) {
                        acc[jj] = mma<false>(wlo, xh, acc[jj]);
                        acc[jj] = mma<false>(whi, xl, acc[jj]);
                        acc[jj] = mma<false>(whi, xh, acc[jj]);
                    }
                    if (ky > 0) {
                        acc[jj + 2] = mma<false>(wpl, xh, acc[jj + 2]);
                        acc[jj + 2] = mma<false>(wph, xl, acc[jj + 2]);
                        acc[jj + 2] = mma<false>(wph, xh, acc[jj + 2]);
                    }
                }
                wph = whi;
                wpl = wlo;
            }
        }
#pragma unroll
        for (int kc = 9; kc < KC; ++kc) {
            const short8 whi = *reinterpret_cast<const short8 *>(wlds + (kc * 2 + 0) * 1024 + lane * 16);
            const short8 wlo = *reinterpret_cast<const short8 *>(wlds + (kc * 2 + 1) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                const unsigned char *lp = smem + pofs[j] + toff[kc];
                const short8 xh = *reinterpret_cast<const short8 *>(lp);
                const short8 xl = *reinterpret_cast<const short8 *>(lp + PLANEB);
                acc[j] = mma<false>(wlo, xh, acc[j]);
                acc[j] = mma<false>(whi, xl, acc[j]);
                acc[j] = mma<false>(whi, xh, acc[j]);
            }
        }
        // ---- epilogue: conv_tile's packed 8-channel form (tiles are whole: the host requires H, W multiples of the tile) ----
        {
            const int64_t obase = (((int64_t)cur.b * a.No + cur.gz0) * a.Ho + cur.gy0) * a.Wo + cur.gx0;
            uint16_t *ob = a.out + obase * 16;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                float cls = 0.f;
                epilogue_lean_t<PREC>(ob, nullptr, voff[j], acc[j][0], acc[j][1], acc[j][2], acc[j][3], false, uint4{}, RELU, false, zero4, cls, true);
            }
        }
        if (more) {
            __syncthreads();   // every wave is done reading this tile's image
            store_tile();
            __syncthreads();
        }
        cur = cn;
    }
}

bool stem_pipe_ok(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t) {
    return prec == P_BF16X3 && cfg && cfg->geo == G2P && cfg->ty == stemp::TY && cfg->tx == stemp::TX && cfg->nw == stemp::NW && t.KC[0] == stemp::KC &&
           a.fs32 && !(a.dbg & (DFFW_ARGS_RAW | DFFW_ARGS_SUMS | 7)) && a.out && !a.out_pre && !a.outf && !a.res0 && !a.res1 && !a.cls_w && a.relu != 2 &&
           a.Cout == 8 && a.Hg % stemp::TY == 0 && a.Wg % stemp::TX == 0 && t.nsplit == 1 && t.ksplit <= 1 && !a.trace && (a.Wi - 2) % 4 == 0;
}

hipError_t launch_stem_pipe(const ConvArgs &a, const TileArgs &t, int wgs, hipStream_t s) {
    const int want = wgs > 0 ? wgs : 512;   // two resident workgroups per CU
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(stemp::NW * 64);
    if (a.relu == 1) hipLaunchKernelGGL((stem_pipe<true>), grid, block, 0, s, a, t);
    else hipLaunchKernelGGL((stem_pipe<false>), grid, block, 0, s, a, t);
    return hipGetLastError();
}

void stem_pipe_kernel_name(const ConvArgs &a, char *buf, int n) { snprintf(buf, n, "dffw::stem_pipe<%s>", a.relu == 1 ? "true" : "false"); }

}  // namespace dffw
