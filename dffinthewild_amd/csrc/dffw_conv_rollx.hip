// conv_rollx: the rolling-window 3x3x3 convolution of dffw_conv_roll.hip as a SOFTWARE-PIPELINED step (round 4), gfx950 / MI355X.
//
// conv_roll's step is a serial chain per wave -- queue the next slice's LDS-DMA, contract, wait + barrier, epilogue -- and the step
// timeline (profiles/r03_step_timeline.txt) shows the contraction taking 1.4 k of its 3.7 k cycles: the other 2.3 k are ~170 scalar /
// vector instructions that issue one by one while the matrix pipe idles, and inside the contraction the three products of a chunk
// form two dependent accumulator chains (a dependent 16x16x32 MFMA issues every ~36 cycles, an independent one every 16).  Here
//   * the epilogue of step n-1 (split to hi + lo, lane-row exchange, store) and the fill of slice n+5 are dealt out over the
//     chunk gaps of step n's contraction: they issue in the shadow of the MFMAs instead of in front of / behind them;
//   * three accumulators, one per product (w_hi x_hi, w_lo x_hi, w_hi x_lo), summed once per step: no dependent MFMA pair inside a chunk;
//   * the fill uses buffer addressing (`buffer_load_dwordx4 ... lds`): wave-uniform descriptor (per-unit base) + per-lane 32-bit byte
//     offset that only changes with the unit + SGPR slice offset; out-of-image lanes carry an out-of-range offset and the
//     hardware range check writes their zeros -- no 64-bit pointer per piece, no zero page, no per-piece select, one M0 write per piece;
//   * the step body is instantiated per (live(n), live(n-1), live(n-2)) so the interleaved code is straight-line (a counted
//     s_waitcnt tied to fragment registers must not sit behind a branch) and the vmcnt window is exact;
//   * LDS slice image = [tensor / channel octet][part][pixel] planes of 16-byte entries (the two 8-channel inputs of a virtual concat
//     are different tensors, and a DMA piece has ONE descriptor), rows stored even columns first; a wave's operand tile is rows w and
//     w + 4 of the 8-row column: their 72-entry distance is 8 mod 16, so the 16 lanes of a ds_read_b128 row group cover 16 distinct
//     16-byte bank groups without padding.
// Pair form only (<= 8 output channels: `dres4.conv0`, DEN.py:240-284; result rows 0-7 / 8-15 = the 8 channels of the even / odd pixel
// of a horizontally adjacent pair), split-bf16 storage, epilogue out = [relu](acc).  Everything else stays on conv_roll; the filter
// packing (pack_conv, ROLL_CHUNKS_PAIR) is conv_roll's.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "dffw_conv_roll.h"
#include "dffw_device.h"

namespace dffw {

namespace rollx {
constexpr int TY = 8, TX = 16, FY = TY + 2, FX = TX + 2, HALF = FX / 2, NW = 4, RING = 6;
constexpr int ROWE = 2 * FX;          // 16-byte entries per footprint row of a tensor plane: [hi: 18 pixels][lo: 18 pixels]
constexpr int TPL = 6144;             // a tensor's plane (10 rows of 576 bytes), padded to 6 DMA pieces of 1 KiB
constexpr int SLOTB = 2 * TPL;        // ring slot = one input slice of the column's footprint
constexpr int PPW = 3;                // DMA pieces per wave and slice (12 pieces, 4 waves)
constexpr int NCH = 18;               // contraction chunks: 3 slices x 3 filter rows x 2 halves of the pair's 4 input columns
static_assert(FY * ROWE * 16 <= TPL && TPL % 1024 == 0 && 2 * (TPL / 1024) == NW * PPW, "piece layout");
static_assert((2 * ROWE) % 16 == 8, "a wave's two rows must be 8 bank groups apart");
}   // namespace rollx

// RELU: out = relu(acc) (every pair-form layer of the network) or out = acc
// ABL (development only, DFFW_ROLLX_ABL): timing ablations -- 1 no operand reads, 2 no MFMAs, 4 no fill, 8 no stores, 16 no barrier
template <bool RELU, int ABL = 0, bool NTS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rollx_pair(const ConvArgs a, const RollArgs t) {
    using namespace rollx;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOTB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;

    // ---- this workgroup's units (columns of one sample / slice range), as conv_roll: XCD x owns a contiguous range -------------
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = t.total_tiles >> 3, rem = t.total_tiles & 7;
        const int xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        uend = xs + q + (xcd < rem ? 1 : 0);
        ufirst = xs + widx;
    }
    if (ufirst >= uend) return;
    struct Unit {
        int b, zbeg, nz, gy0, gx0;
    };
    auto decode = [&](int u) {
        Unit c;
        const int txi = u % t.tiles_x;
        int tt = u / t.tiles_x;
        const int tyi = tt % t.tiles_y;
        tt /= t.tiles_y;
        const int zp = tt % t.zsplit;
        c.b = tt / t.zsplit;
        c.gy0 = tyi * TY;
        c.gx0 = txi * TX;
        c.zbeg = zp * a.No / t.zsplit;
        c.nz = (zp + 1) * a.No / t.zsplit - c.zbeg;
        return c;
    };

    // ---- fill: this wave's three pieces of a slice belong to ONE source (waves 0-1: in0 / channel octet 0, waves 2-3: in1 / octet 1) ----
    const bool two = a.C1 != 0;                        // virtual concat of two 8-channel tensors (else one 16-channel tensor)
    const int tsel = wave >> 1;
    const int recb = two ? 32 : 64;                    // bytes per pixel record of the source tensor ([hi C][lo C])
    const char *tbase = reinterpret_cast<const char *>((two && tsel) ? a.in1 : a.in0) + (two ? 0 : tsel * 16);
    const int slice_bytes = a.Hi * a.Wi * recb;
    int fyx[PPW], foff[PPW];                           // per lane and piece: footprint row | column << 8; byte offset from the footprint origin
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int e = ((wave & 1) * PPW + k) * 64 + lane;          // 16-byte entry inside the tensor plane
        const int fy = e / ROWE, j = e - fy * ROWE;                // [row][part][18 pixels]
        const int part = j >= FX ? 1 : 0, sx = j - part * FX;
        const int fx = sx < HALF ? 2 * sx : 2 * (sx - HALF) + 1;   // rows are stored even columns first
        fyx[k] = fy | (fx << 8) | (e < FY * ROWE ? 0 : 1 << 16);   // bit 16: padding entry of the plane, never in the image
        foff[k] = (fy * a.Wi + fx) * recb + part * (recb / 2);
    }
    int fvo[PPW];                                      // ... with the out-of-image lanes of the current unit pushed out of range
    const char *fbase = tbase;                         // descriptor base of the unit being fetched: its footprint origin in slice 0
    int fu = ufirst, fq = 0, fslices = 0, fz = 0;      // fill cursor: unit, slice inside it, its slice count, input slice index
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fslices = c.nz + 2;
        fz = c.zbeg - 1;
        fbase = tbase + ((int64_t)c.b * a.Ni * a.Hi * a.Wi + (int64_t)(c.gy0 - 1) * a.Wi + (c.gx0 - 1)) * recb;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int iy = c.gy0 - 1 + (fyx[k] & 0xFF), ix = c.gx0 - 1 + ((fyx[k] >> 8) & 0xFF);
            fvo[k] = (!(fyx[k] >> 16) && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) ? foff[k] : (int)0x80000000;
        }
    };
    setup_fill();
    int fslotb = 0;                                    // byte offset of the ring slot the next slice goes to
    // the three pieces of the stream's next slice; `K` = which one (compile time), so that they can be dealt over three chunk gaps
    auto issue_piece = [&](auto K) {
        constexpr int k = decltype(K)::value;
        const bool zin = (unsigned)fz < (unsigned)a.Ni && fu < uend;           // slices above / below the volume and past the stream: zeros
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fbase), 0, zin ? (int)0x80000000 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + fslotb + (wave * PPW + k) * 1024), 16, fvo[k],
                                                 zin ? fz * slice_bytes : 0, 0, 0);
    };
    auto advance_fill = [&]() {
        fslotb = (fslotb + SLOTB == RING * SLOTB) ? 0 : fslotb + SLOTB;
        ++fz;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) {
        issue_piece(std::integral_constant<int, 0>{});
        issue_piece(std::integral_constant<int, 1>{});
        issue_piece(std::integral_constant<int, 2>{});
        advance_fill();
    }

    // ---- per-lane operand / output addressing.  Column r of the wave's tile = pair pp of row w + 4*rr; lane group g reads input
    // column 2*pp + 2*half + (g >> 1) (even ones in the first half of the LDS row), channel octet g & 1, and ends up with
    // channels (g & 1)*4.. of pixel 2*pp + (g >> 1)
    const int rr = r >> 3, pp = r & 7;
    const int prow = (wave & 1) + 4 * (wave >> 1) + 2 * rr, pcol = 2 * pp + (g >> 1);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned lbase = lds0 + (g & 1) * TPL + (prow * ROWE + (g >> 1) * HALF + pp) * 16;
    const int vob = ((prow * a.Wo + pcol) * 16 + (g & 1) * 8) * 2;   // byte offset of the lane's 16-byte output piece inside the slice's column

    // ---- the filter: 18 A-fragments per part, resident for the whole walk ----
    short8 w[NCH][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[c][0] = wp[(c * 2 + 0) * 64];
            w[c][1] = wp[(c * 2 + 1) * 64];
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + (g & 1) * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue slices, filter, bias (compiler-visible, so that no later wait is invented)
    asm volatile("s_barrier" ::: "memory");

    int sidxb = 0;                        // byte offset of the ring slot of the window's first slice
    f32x4 acc0 = bias4, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    char *pptr = nullptr;                 // where the pending step's output slice starts (wave-uniform)
    StepTrace trc(a.trace, wave, lane, NW);

    // One step of the stream.  LIVE: this window produces an output slice; PEND: the previous one did (its epilogue runs now);
    // PEND2: the one before that did (its store is still inside the vmcnt window).
    auto step = [&](auto LIVE_, auto PEND_, auto PEND2_, char *optr) {
        constexpr bool LIVE = decltype(LIVE_)::value, PEND = decltype(PEND_)::value, PEND2 = decltype(PEND2_)::value;
        trc.stamp(0);
        // the pending step's value: its three accumulators are complete (a barrier ago)
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        uint32_t h01 = 0, h23 = 0, l01 = 0, l23 = 0;
        if constexpr (PEND) {
            v0 = acc0[0] + (acc1[0] + acc2[0]);
            v1 = acc0[1] + (acc1[1] + acc2[1]);
            v2 = acc0[2] + (acc1[2] + acc2[2]);
            v3 = acc0[3] + (acc1[3] + acc2[3]);
            if constexpr (RELU) {
                v0 = relu_bits(v0);
                v1 = relu_bits(v1);
                v2 = relu_bits(v2);
                v3 = relu_bits(v3);
            }
        }
        // the side work, dealt over the chunk gaps (dead steps run it back to back): epilogue of the pending step first -- its store must
        // precede this step's DMA pieces in the (in-order) vmcnt queue -- then the fill
        auto side = [&](auto S) {
            constexpr int s = decltype(S)::value;
            typedef __attribute__((ext_vector_type(2))) float f2;
            typedef __attribute__((ext_vector_type(2))) __bf16 b2;
            if constexpr (PEND && s == 0) {
                h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v0, v1}, b2));
                h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v2, v3}, b2));
            }
            if constexpr (PEND && s == 1) {
                v0 -= __uint_as_float(h01 << 16);
                v1 -= __uint_as_float(h01 & 0xFFFF0000u);
                v2 -= __uint_as_float(h23 << 16);
                v3 -= __uint_as_float(h23 & 0xFFFF0000u);
            }
            if constexpr (PEND && s == 2) {
                l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v0, v1}, b2));
                l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v2, v3}, b2));
            }
            if constexpr (PEND && s == 3) swap16(h01, l01);
            if constexpr (PEND && s == 4) swap16(h23, l23);
            if constexpr (PEND && s == 5 && !(ABL & 8)) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 *dst = reinterpret_cast<u32x4 *>(pptr + (uint32_t)vob);
                if constexpr (NTS) __builtin_nontemporal_store(u32x4{h01, h23, l01, l23}, dst);
                else *dst = u32x4{h01, h23, l01, l23};
            }
            if constexpr (PEND && s == 5 && (ABL & 8)) asm volatile("" ::"v"(h01), "v"(h23), "v"(l01), "v"(l23));
            if constexpr (s == 6 && !(ABL & 4)) issue_piece(std::integral_constant<int, 0>{});
            if constexpr (s == 7 && !(ABL & 4)) issue_piece(std::integral_constant<int, 1>{});
            if constexpr (s == 8 && !(ABL & 4)) issue_piece(std::integral_constant<int, 2>{});
        };
        if constexpr (LIVE) {
            // ring slots of the window's three slices
            unsigned ad[3];
            {
                int sb = sidxb;
                ad[0] = lbase + sb;
                sb = (sb + SLOTB == RING * SLOTB) ? 0 : sb + SLOTB;
                ad[1] = lbase + sb;
                sb = (sb + SLOTB == RING * SLOTB) ? 0 : sb + SLOTB;
                ad[2] = lbase + sb;
            }
            // operand fragments two chunks ahead of the MFMAs; reads and counted waits are inline asm (beside LDS-DMA hipcc degrades every
            // lgkmcnt wait to 0).  Chunk c = (slice c/6, filter row (c%6)/2, half c%2): an immediate offset from the slice's lane base
            trc.stamp(1);
            constexpr int DEPTH = 2, NB = DEPTH + 1;
            short8 x[NB][2];
            auto fetch = [](auto C, short8 (&xx)[NB][2], const unsigned (&adr)[3]) {
                constexpr int c = decltype(C)::value;
                constexpr int imm = (((c % 6) / 2) * ROWE + (c % 2)) * 16;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[c % NB][0]) : "v"(adr[c / 6]), "n"(imm));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[c % NB][1]) : "v"(adr[c / 6]), "n"(imm + FX * 16));
            };
            if constexpr (!(ABL & 1)) static_for<DEPTH>([&](auto C) { fetch(C, x, ad); });
            else static_for<NB>([&](auto C) { x[decltype(C)::value][0] = w[0][0]; x[decltype(C)::value][1] = w[1][1]; });
            f32x4 n0 = bias4, n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
            static_for<NCH>([&](auto C) {
                constexpr int c = decltype(C)::value;
                if constexpr (c + DEPTH < NCH && !(ABL & 1)) fetch(std::integral_constant<int, c + DEPTH>{}, x, ad);
                constexpr int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * 2;
                if constexpr (!(ABL & 1)) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x[c % NB][0]), "+v"(x[c % NB][1]) : "n"(ahead));
                else asm volatile("" : "+v"(x[c % NB][0]), "+v"(x[c % NB][1]));
                if constexpr (!(ABL & 2)) {
                    n0 = mma<false>(w[c][0], x[c % NB][0], n0);
                    n1 = mma<false>(w[c][1], x[c % NB][0], n1);
                    n2 = mma<false>(w[c][0], x[c % NB][1], n2);
                } else {
                    asm volatile("" ::"v"(x[c % NB][0]), "v"(x[c % NB][1]), "v"(w[c][0]), "v"(w[c][1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c >= 1 && c <= 9) {
                    side(std::integral_constant<int, c - 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            acc0 = n0;
            acc1 = n1;
            acc2 = n2;
        } else {
            static_for<9>([&](auto S) { side(S); });
        }
        trc.stamp(2);
        // Queue order per step: [store of step n-1] [3 pieces of slice n+5].  Slice n+3 (queued two steps ago) must have landed: everything
        // younger -- the pieces of this and the previous step and the stores between them -- may stay in flight across the barrier.
        constexpr int INFLIGHT = 2 * PPW + (PEND ? 1 : 0) + (PEND2 ? 1 : 0);
        if constexpr (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        trc.stamp(3);
        sidxb = (sidxb + SLOTB == RING * SLOTB) ? 0 : sidxb + SLOTB;
        advance_fill();
        pptr = optr;
        trc.stamp(4);
        trc.next();
    };

    int hist = 0;   // bit 0: the previous step was live, bit 1: the one before
    auto dispatch = [&](bool live, char *optr) {
        using T = std::true_type;
        using F = std::false_type;
        switch ((live ? 4 : 0) | hist) {
            case 0: step(F{}, F{}, F{}, optr); break;
            case 1: step(F{}, T{}, F{}, optr); break;
            case 2: step(F{}, F{}, T{}, optr); break;
            case 3: step(F{}, T{}, T{}, optr); break;
            case 4: step(T{}, F{}, F{}, optr); break;
            case 5: step(T{}, T{}, F{}, optr); break;
            case 6: step(T{}, F{}, T{}, optr); break;
            default: step(T{}, T{}, T{}, optr); break;
        }
        hist = ((hist << 1) & 2) | (live ? 1 : 0);
    };

    const int64_t ostride = (int64_t)a.Ho * a.Wo * 32;   // bytes per output slice
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        char *optr = reinterpret_cast<char *>(a.out) + ((((int64_t)U.b * a.No + U.zbeg) * a.Ho + U.gy0) * a.Wo + U.gx0) * 32;
        for (int st = 0; st < U.nz + 2; ++st) {
            dispatch(st < U.nz, optr);   // windows starting on the unit's last two slices straddle two units: no output
            optr += ostride;
        }
    }
    // the last live step's epilogue is still pending (a unit ends with two dead steps, which have run it) -- and the slices queued
    // past the end of the stream are still in flight: a wave must not retire before its LDS-DMA has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- conv_rollx_k2: the same pipelined step for 32 input -> 16 output channels (`dres3.conv0`, DEN.py:240-284: the full-resolution conv
// of the middle hourglass over the concat of two 16-channel volumes) ------------------------------------------------------------------
// 16 output channels fill the MFMA result rows (no pixel pairs) and the filter of 32 input channels (27 chunks, 216 VGPRs) does not fit one
// wave, so the contraction is split over the two 16-channel halves of the input: waves 0-3 hold the filter of half 0 (15 chunks [dz][5
// chunks of 2 in-slice taps x 16 channels], tap 9 = zeros: conv_roll's order), waves 4-7 that of half 1; wave w and wave w + 4 contract the
// same two 16-pixel rows of the 8 x 16 column.  After its contraction a wave hands ONE of its two partial tiles to its partner through LDS
// (double-buffered: the partner may be a step ahead) and keeps the other: in the next step -- inside that step's contraction, as in
// conv_rollx_pair -- it adds the partner's partial to its own and runs the epilogue of that one tile.  One workgroup of 8 waves per CU
// (ring of 5 slices x 2 halves x 12 KiB + 16 KiB exchange = 136 KiB), i.e. the same two waves per SIMD as the 4-wave kernels.
// LDS slice image per half: [row][part][18 pixels][channel octet] in 16-byte entries (rows of 1152 bytes): a DMA piece covers whole 64-byte
// records of ~0.9 footprint rows; the 16 lanes of an operand read are 16 consecutive pixels at a 32-byte pitch and lane rows g, g + 1 take
// the two channel octets of the same tap: conflict-free as in conv_roll.
namespace rollk2 {
constexpr int TY = 8, TX = 16, FY = TY + 2, FX = TX + 2, NW = 8, RING = 5;
constexpr int ROWE = 4 * FX;          // 16-byte entries per footprint row of a half: [hi: 18 pixels x 2 octets][lo: the same]
constexpr int ROWB = ROWE * 16;       // 1152
constexpr int PARTB = 2 * FX * 16;    // 576: offset of a row's lo part
constexpr int HPL = 12 * 1024;        // a half's plane: 10 rows of 1152 bytes, padded to 12 DMA pieces
constexpr int SLOTB = 2 * HPL;
constexpr int PPW = 3;                // DMA pieces per wave and slice (24 pieces, 8 waves)
constexpr int NCH = 15;
constexpr int XCH_OFF = RING * SLOTB, XCHB = 8 * 1024;   // two exchange buffers of 8 tiles x 1 KiB behind the ring
static_assert(FY * ROWB <= HPL && 2 * (HPL / 1024) == NW * PPW, "piece layout");
}   // namespace rollk2

template <bool RELU>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rollx_k2(const ConvArgs a, const RollArgs t) {
    using namespace rollk2;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XCH_OFF + 2 * XCHB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave >> 2, wq = wave & 3;          // K half (= source tensor of a concat) and the pair of rows this wave contracts
    const int g = lane >> 4, r = lane & 15;

    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = t.total_tiles >> 3, rem = t.total_tiles & 7;
        const int xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        uend = xs + q + (xcd < rem ? 1 : 0);
        ufirst = xs + widx;
    }
    if (ufirst >= uend) return;
    struct Unit {
        int b, zbeg, nz, gy0, gx0;
    };
    auto decode = [&](int u) {
        Unit c;
        const int txi = u % t.tiles_x;
        int tt = u / t.tiles_x;
        const int tyi = tt % t.tiles_y;
        tt /= t.tiles_y;
        const int zp = tt % t.zsplit;
        c.b = tt / t.zsplit;
        c.gy0 = tyi * TY;
        c.gx0 = txi * TX;
        c.zbeg = zp * a.No / t.zsplit;
        c.nz = (zp + 1) * a.No / t.zsplit - c.zbeg;
        return c;
    };

    // ---- fill: a wave's three pieces belong to its own half (waves 0-3: in0 / channels 0-15, waves 4-7: in1 / channels 16-31) ----
    const bool two = a.C1 != 0;
    const int recb = two ? 64 : 128;                    // bytes per pixel record of the source tensor ([hi C][lo C])
    const char *tbase = reinterpret_cast<const char *>((two && half) ? a.in1 : a.in0) + (two ? 0 : half * 32);
    const int slice_bytes = a.Hi * a.Wi * recb;
    int fyx[PPW], foff[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int e = (wq * PPW + k) * 64 + lane;                  // 16-byte entry inside the half's plane: [row][part][pixel][octet]
        const int fy = e / ROWE, j = e - fy * ROWE;
        const int part = j >= 2 * FX ? 1 : 0, q = j - part * 2 * FX;
        const int fx = q >> 1, oct = q & 1;
        fyx[k] = fy | (fx << 8) | (e < FY * ROWE ? 0 : 1 << 16);
        foff[k] = (fy * a.Wi + fx) * recb + part * (recb / 2) + oct * 16;
    }
    int fvo[PPW];
    const char *fbase = tbase;
    int fu = ufirst, fq = 0, fslices = 0, fz = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fslices = c.nz + 2;
        fz = c.zbeg - 1;
        fbase = tbase + ((int64_t)c.b * a.Ni * a.Hi * a.Wi + (int64_t)(c.gy0 - 1) * a.Wi + (c.gx0 - 1)) * recb;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int iy = c.gy0 - 1 + (fyx[k] & 0xFF), ix = c.gx0 - 1 + ((fyx[k] >> 8) & 0xFF);
            fvo[k] = (!(fyx[k] >> 16) && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) ? foff[k] : (int)0x80000000;
        }
    };
    setup_fill();
    int fslotb = 0;
    auto issue_piece = [&](auto K) {
        constexpr int k = decltype(K)::value;
        const bool zin = (unsigned)fz < (unsigned)a.Ni && fu < uend;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fbase), 0, zin ? (int)0x80000000 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + fslotb + (wave * PPW + k) * 1024), 16, fvo[k],
                                                 zin ? fz * slice_bytes : 0, 0, 0);
    };
    auto advance_fill = [&]() {
        fslotb = (fslotb + SLOTB == RING * SLOTB) ? 0 : fslotb + SLOTB;
        ++fz;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) {
        issue_piece(std::integral_constant<int, 0>{});
        issue_piece(std::integral_constant<int, 1>{});
        issue_piece(std::integral_constant<int, 2>{});
        advance_fill();
    }

    // ---- operand addressing: tile j of the wave = row 2*wq + j, column r; K octet g of chunk k5 = (tap 2*k5 + (g >> 1), channel octet g & 1)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned aoff[5];
#pragma unroll
    for (int k5 = 0; k5 < 5; ++k5) {
        const int tap9 = 2 * k5 + (g >> 1);
        const int dy = tap9 < 9 ? tap9 / 3 : 0, dx = tap9 < 9 ? tap9 % 3 : 0;
        aoff[k5] = lds0 + half * HPL + (2 * wq + dy) * ROWB + ((r + dx) * 2 + (g & 1)) * 16;
    }
    // the tile this wave finishes: half 0 keeps its row 2*wq, half 1 its row 2*wq + 1; the other one goes to the partner
    const int myrow = 2 * wq + half;
    const int vob = ((myrow * a.Wo + r) * 32 + (g & 1) * 16 + (g >> 1) * 8) * 2;   // byte offset of the lane's 16-byte piece ([hi 16][lo 16] records)
    const unsigned xch_wr = lds0 + XCH_OFF + ((2 * wq + (half ^ 1)) * 64 + lane) * 16;   // partial of the row the PARTNER finishes
    const unsigned xch_rd = lds0 + XCH_OFF + (myrow * 64 + lane) * 16;                   // the partner's partial of my row

    short8 w[NCH][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + (size_t)half * NCH * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[c][0] = wp[(c * 2 + 0) * 64];
            w[c][1] = wp[(c * 2 + 1) * 64];
        }
    }
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + g * 4);
    if (half) bias4 = f32x4{0.f, 0.f, 0.f, 0.f};      // the BatchNorm shift enters once, through half 0's accumulators
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    int sidxb = 0, xpar = 0;              // ring slot of the window's first slice; exchange buffer of the step being contracted
    f32x4 mine = {0.f, 0.f, 0.f, 0.f};   // this wave's own partial of the tile it finishes (pending step)
    char *pptr = nullptr;

    auto step = [&](auto LIVE_, auto PEND_, auto PEND2_, char *optr) {
        constexpr bool LIVE = decltype(LIVE_)::value, PEND = decltype(PEND_)::value, PEND2 = decltype(PEND2_)::value;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        uint32_t h01 = 0, h23 = 0, l01 = 0, l23 = 0;
        f32x4 theirs = {0.f, 0.f, 0.f, 0.f};
        // the partner's partial of the pending step: requested first, so that the counted lgkmcnt waits of the contraction cover it (DS
        // operations retire in order); a dead step waits for it directly
        if constexpr (PEND) {
            const unsigned ad = xch_rd + (xpar ^ 1) * XCHB;
            asm volatile("ds_read_b128 %0, %1" : "=v"(theirs) : "v"(ad));
            if constexpr (!LIVE) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(theirs));
        }
        auto side = [&](auto S) {
            constexpr int s = decltype(S)::value;
            typedef __attribute__((ext_vector_type(2))) float f2;
            typedef __attribute__((ext_vector_type(2))) __bf16 b2;
            if constexpr (PEND && s == 0) {
                v0 = mine[0] + theirs[0];
                v1 = mine[1] + theirs[1];
                v2 = mine[2] + theirs[2];
                v3 = mine[3] + theirs[3];
                if constexpr (RELU) {
                    v0 = relu_bits(v0);
                    v1 = relu_bits(v1);
                    v2 = relu_bits(v2);
                    v3 = relu_bits(v3);
                }
            }
            if constexpr (PEND && s == 1) {
                h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v0, v1}, b2));
                h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v2, v3}, b2));
            }
            if constexpr (PEND && s == 2) {
                v0 -= __uint_as_float(h01 << 16);
                v1 -= __uint_as_float(h01 & 0xFFFF0000u);
                v2 -= __uint_as_float(h23 << 16);
                v3 -= __uint_as_float(h23 & 0xFFFF0000u);
            }
            if constexpr (PEND && s == 3) {
                l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v0, v1}, b2));
                l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v2, v3}, b2));
            }
            if constexpr (PEND && s == 4) swap16(h01, l01);
            if constexpr (PEND && s == 5) swap16(h23, l23);
            if constexpr (PEND && s == 6) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<u32x4 *>(pptr + (uint32_t)vob) = u32x4{h01, h23, l01, l23};
            }
            if constexpr (s == 7) issue_piece(std::integral_constant<int, 0>{});
            if constexpr (s == 8) issue_piece(std::integral_constant<int, 1>{});
            if constexpr (s == 9) issue_piece(std::integral_constant<int, 2>{});
        };
        if constexpr (LIVE) {
            int sb[3];
            sb[0] = sidxb;
            sb[1] = (sb[0] + SLOTB == RING * SLOTB) ? 0 : sb[0] + SLOTB;
            sb[2] = (sb[1] + SLOTB == RING * SLOTB) ? 0 : sb[1] + SLOTB;
            constexpr int DEPTH = 2, NB = DEPTH + 1;
            short8 x[NB][2][2];   // [buffer][tile][part]
            auto fetch = [](auto C, short8 (&xx)[NB][2][2], const unsigned (&ao)[5], const int (&sbb)[3]) {
                constexpr int c = decltype(C)::value;
                const unsigned ad = ao[c % 5] + sbb[c / 5];
                asm volatile("ds_read_b128 %0, %1" : "=v"(xx[c % NB][0][0]) : "v"(ad));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[c % NB][0][1]) : "v"(ad), "n"(PARTB));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[c % NB][1][0]) : "v"(ad), "n"(ROWB));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[c % NB][1][1]) : "v"(ad), "n"(ROWB + PARTB));
            };
            static_for<DEPTH>([&](auto C) { fetch(C, x, aoff, sb); });
            f32x4 n[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                n[j][0] = bias4;
                n[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                n[j][2] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            static_for<NCH>([&](auto C) {
                constexpr int c = decltype(C)::value;
                if constexpr (c + DEPTH < NCH) fetch(std::integral_constant<int, c + DEPTH>{}, x, aoff, sb);
                constexpr int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * 4;
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[c % NB][0][0]), "+v"(x[c % NB][0][1]), "+v"(x[c % NB][1][0]), "+v"(x[c % NB][1][1]) : "n"(ahead));
                if constexpr (PEND && c == 0) asm volatile("" : "+v"(theirs));   // older than chunk 0's reads: landed
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    n[j][0] = mma<false>(w[c][0], x[c % NB][j][0], n[j][0]);
                    n[j][1] = mma<false>(w[c][1], x[c % NB][j][0], n[j][1]);
                    n[j][2] = mma<false>(w[c][0], x[c % NB][j][1], n[j][2]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c >= 1 && c <= 10) {
                    side(std::integral_constant<int, c - 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // hand the partner its row's partial, keep mine
            f32x4 give;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s0 = n[0][0][i] + (n[0][1][i] + n[0][2][i]), s1 = n[1][0][i] + (n[1][1][i] + n[1][2][i]);
                mine[i] = half ? s1 : s0;
                give[i] = half ? s0 : s1;
            }
            const unsigned wad = xch_wr + xpar * XCHB;
            asm volatile("ds_write_b128 %0, %1" ::"v"(wad), "v"(give) : "memory");
        } else {
            static_for<10>([&](auto S) { side(S); });
        }
        constexpr int INFLIGHT = PPW + (PEND ? 1 : 0);   // ring of 5: the slice queued in the PREVIOUS step is the next window's last one
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
        (void)PEND2;
        sidxb = (sidxb + SLOTB == RING * SLOTB) ? 0 : sidxb + SLOTB;
        xpar ^= 1;
        advance_fill();
        pptr = optr;
    };

    int hist = 0;
    auto dispatch = [&](bool live, char *optr) {
        using T = std::true_type;
        using F = std::false_type;
        switch ((live ? 2 : 0) | (hist & 1)) {
            case 0: step(F{}, F{}, F{}, optr); break;
            case 1: step(F{}, T{}, F{}, optr); break;
            case 2: step(T{}, F{}, F{}, optr); break;
            default: step(T{}, T{}, F{}, optr); break;
        }
        hist = live ? 1 : 0;
    };

    const int64_t ostride = (int64_t)a.Ho * a.Wo * 64;   // bytes per output slice (16 channels, hi + lo)
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        char *optr = reinterpret_cast<char *>(a.out) + ((((int64_t)U.b * a.No + U.zbeg) * a.Ho + U.gy0) * a.Wo + U.gx0) * 64;
        for (int st = 0; st < U.nz + 2; ++st) {
            dispatch(st < U.nz, optr);
            optr += ostride;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool rollx_k2_ok(int prec, const ConvArgs &a) {
    if (prec != P_BF16X3 || (a.dbg & DFFW_ARGS_NO_ROLLX)) return false;
    if (!a.out || a.out_pre || a.outf || a.res0 || a.res1 || a.res_bcast || a.cls_w || a.relu == 2 || a.Cout != 16) return false;
    if (!((a.C0 == 32 && a.C1 == 0) || (a.C0 == 16 && a.C1 == 16))) return false;
    const int64_t recb = a.C1 ? 64 : 128;
    return (int64_t)(a.Ni + 1) * a.Hi * a.Wi * recb < (1ll << 31);
}

hipError_t launch_conv_rollx_k2(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 256;   // one 8-wave workgroup per CU
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(rollk2::NW * 64);
    if (a.relu == 1) hipLaunchKernelGGL((conv_rollx_k2<true>), grid, block, 0, s, a, t);
    else hipLaunchKernelGGL((conv_rollx_k2<false>), grid, block, 0, s, a, t);
    return hipGetLastError();
}

void conv_rollx_k2_kernel_name(const ConvArgs &a, char *buf, int n) { snprintf(buf, n, "dffw::conv_rollx_k2<%s>", a.relu == 1 ? "true" : "false"); }

bool rollx_pair_ok(int prec, const ConvArgs &a, bool pair) {
    if (prec != P_BF16X3 || !pair || (a.dbg & DFFW_ARGS_NO_ROLLX)) return false;
    if (!a.out || a.out_pre || a.outf || a.res0 || a.res1 || a.res_bcast || a.cls_w || a.relu == 2 || a.Cout != 8) return false;
    if (!((a.C0 == 16 && a.C1 == 0) || (a.C0 == 8 && a.C1 == 8))) return false;
    // 32-bit buffer offsets: a sample's input volume (+ one footprint) stays below 2^31 bytes
    const int64_t recb = a.C1 ? 32 : 64;
    return (int64_t)(a.Ni + 1) * a.Hi * a.Wi * recb < (1ll << 31);
}

hipError_t launch_conv_rollx_pair(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 512;
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(rollx::NW * 64);
    if (a.relu != 1) hipLaunchKernelGGL((conv_rollx_pair<false>), grid, block, 0, s, a, t);
#ifdef DFFW_ABL_BUILD   // development (make ABL=1): timing ablations (DFFW_ROLLX_ABL; results are wrong with any bit set), non-temporal stores (DFFW_ROLLX_NTS)
    else if (getenv("DFFW_ROLLX_ABL") && atoi(getenv("DFFW_ROLLX_ABL")) == 3) hipLaunchKernelGGL((conv_rollx_pair<true, 3>), grid, block, 0, s, a, t);
    else if (getenv("DFFW_ROLLX_ABL") && atoi(getenv("DFFW_ROLLX_ABL")) == 12) hipLaunchKernelGGL((conv_rollx_pair<true, 12>), grid, block, 0, s, a, t);
    else if (getenv("DFFW_ROLLX_NTS")) hipLaunchKernelGGL((conv_rollx_pair<true, 0, true>), grid, block, 0, s, a, t);
#endif
    else hipLaunchKernelGGL((conv_rollx_pair<true>), grid, block, 0, s, a, t);
    return hipGetLastError();
}

void conv_rollx_pair_kernel_name(const ConvArgs &a, char *buf, int n) { snprintf(buf, n, "dffw::conv_rollx_pair<%s>", a.relu == 1 ? "true" : "false"); }

}  // namespace dffw
