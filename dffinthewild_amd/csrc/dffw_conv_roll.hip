// conv_roll: 3x3x3 stride-1 convolution as a 2.5-D rolling window along the focus-slice axis, for gfx950 (MI355X).
//
// conv_tile stages a TZ x TY x TX block with its full halo, waits for it, contracts, stores, exits: every workgroup
// walks fill -> barrier -> MFMA -> stores in sequence and only other resident workgroups cover the gaps.  For the
// 16-channel layers of the full-resolution hourglass (DEN.py:240-284, `dres4.conv0/conv2`: the heaviest launches of
// the forward) conv_roll turns the slice axis into a software pipeline instead:
//   * a workgroup owns a TY x TX column of ONE sample and walks its output slices front to back;
//   * LDS holds a ring of 4 input slices (footprint (TY+2) x (TX+2), hi and lo planes of the split-bf16 records);
//     while slice z is contracted out of ring slots z-1, z, z+1 the LDS-DMA of slice z+2 is in flight into the 4th
//     slot, so a slice is fetched once per column (halo 1.4x in-plane, none along z) and its latency is covered by
//     the workgroup's own MFMAs;
//   * the whole filter (27 taps x 16 channels x <= 16 output channels, hi + lo) lives in registers as 15 MFMA
//     A-fragments per part: the inner loop issues only ds_read_b128 + v_mfma, no weight stream from L1/L2, no tap
//     table (the contraction is fully unrolled, tap offsets are immediates, only the ring rotation is a register);
//   * one barrier per slice.
// K order: [dz][5 chunks of 32] with chunk k5 = in-slice taps 2*k5, 2*k5+1 (tap 9 = zero weights) x 16 channels, so
// the slice of a chunk (= its ring slot) is wave-uniform.  Epilogue = conv_tile's (dffw_device.h).
#include <cstdio>
#include <algorithm>
#include <cstdlib>

#include "dffw_conv_roll.h"
#include "dffw_device.h"

namespace dffw {

// RES: the layer adds a residual volume (its pieces are prefetched by hand, see below); costs 8 VGPRs, paid for with a
// shallower operand pipeline so that both variants stay at two waves per SIMD
// PAIR: layers with <= 8 output channels.  A plain 16-row MFMA result tile would carry 8 dead rows; instead a GEMM
// column is a PAIR of horizontally adjacent pixels: result rows 0-7 are the 8 channels of the even pixel, rows 8-15
// those of the odd one, and the contraction runs over the 4 input columns the pair touches (K per slice and filter
// row = 4 x 16 = 64 = two chunks, the filter's unused corner entries are zeros): 18 chunks per 32 pixels instead of
// 2 x 15, i.e. 1.67x fewer MFMAs.  LDS rows are stored even columns first, odd columns second, so that the 8 pairs of
// an operand read stay on consecutive addresses; a wave's tile is 2 rows x 8 pairs.
// LEAN: the launch's epilogue is "out = [relu](acc [+ residual])" (roll_lean(), dffw_conv_roll.h): only the straight-line routine
// epilogue_lean is compiled in (with the generic one beside it the 16-output kernel needed 263 registers = one wave per SIMD)
template <int PREC, int TY, int TX, int NWAVES, int RING, bool RES, bool PAIR, bool LEAN>
__global__ __launch_bounds__(NWAVES * 64) void conv_roll(const ConvArgs a, const RollArgs t) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int CIN = 16, PIXB = CIN * 2;
    constexpr int FY = TY + 2, FX = TX + 2, FPIX = FY * FX;
    constexpr int NPIECE = (FPIX * 2 + 63) / 64;          // 1 KiB wave instructions per plane
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    static_assert(RING >= 4 && RING <= 8, "3 slices being read + at least one being filled");
    constexpr int MTW = TY * TX / (PAIR ? 32 : 16) / NWAVES;   // operand tiles per wave: one 16-pixel row, or (PAIR) 2 rows x 8 pixel pairs
    constexpr int NCH = PAIR ? 18 : 15;                      // 32-deep contraction chunks
    static_assert(!PAIR || (MTW == 1 && FX % 2 == 0), "pair layout: one tile of two rows per wave, even row length");
    static_assert(TX == 16, "one operand tile = one 16-pixel row");
    static_assert(TY % NWAVES == 0, "rows split evenly over the waves");
    constexpr int NP = PARTS * NPIECE;                    // DMA pieces per slice
    constexpr int PPW = (NP + NWAVES - 1) / NWAVES;       // pieces per wave per slice
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (the counted vmcnt waits rely on it)");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOTB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;

    // ---- this workgroup's columns.  A "unit" is one column of one sample (or of one of its zsplit slice ranges);
    // units are numbered x fastest, then y, slice range, sample.  XCD x (= blockIdx % 8) owns a contiguous range of them
    // and its workgroups take them round-robin, so the workgroups running at the same time on an XCD walk
    // neighbouring columns and share their halos in that XCD's L2.
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

    // ---- the slice stream.  The units of this workgroup form ONE stream of input slices: unit u contributes its
    // nz+2 slices (one above, one below the outputs; zero pages outside the volume), then the next unit follows.  The
    // fill cursor runs RING-1 slices ahead of the window that is being contracted and simply keeps going across unit
    // boundaries, so only the first unit of a workgroup ever waits for a cold ring.
    // Per-lane fill state of the unit being fetched: piece p = wave*PPW + k of a slice covers 64 16-byte chunks (pixel,
    // channel octet) of one plane; its source address inside input slice 0 of the sample is decoded once per unit.
    const int ps0 = PARTS * a.C0;
    const int slice_elems = a.Hi * a.Wi * ps0;            // == Hi*Wi*ps1 when a second input exists (C0 == C1, checked by the host)
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0, fslices = 0, fz0 = 0;        // fill cursor: unit, slice inside it, its slice count, first slice
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fslices = c.nz + 2;
        fz0 = c.zbeg - 1;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, pix = ci >> 1, oct = ci & 1;
            const int fy = pix / FX, sx = pix - fy * FX;
            const int fx = PAIR ? (sx < FX / 2 ? 2 * sx : 2 * (sx - FX / 2) + 1) : sx;   // PAIR: even columns first, then odd
            const int iy = c.gy0 - 1 + fy, ix = c.gx0 - 1 + fx;
            const int ch = oct * 8;
            const bool second = ch >= a.C0;
            const int cc = second ? ch - a.C0 : ch;
            const int csrc = second ? a.C1 : a.C0;
            fok[k] = p < NP && pix < FPIX && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const uint16_t *sp = (second ? a.in1 : a.in0) + (int64_t)c.b * a.Ni * slice_elems;
            fsrc[k] = sp + (int64_t)(iy * a.Wi + ix) * (PARTS * csrc) + part * csrc + cc;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {   // queue the next slice of the stream into the next ring slot (zero page once the stream is over)
        const int iz = fz0 + fq;
        const bool zin = (unsigned)iz < (unsigned)a.Ni && fu < uend;
        unsigned char *slot = smem + fslot * SLOTB;
        const int64_t zo = (int64_t)iz * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RING) ? 0 : fslot + 1;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // ---- per-lane operand addressing: row j of this wave, column r; K-octet g picks (tap parity, channel octet) ----
    int inoff[5];
#pragma unroll
    for (int k5 = 0; k5 < 5; ++k5) {
        const int tap9 = 2 * k5 + (g >> 1);
        const int dy = tap9 < 9 ? tap9 / 3 : 0, dx = tap9 < 9 ? tap9 % 3 : 0;
        inoff[k5] = (dy * FX + dx) * PIXB + (g & 1) * 16;
    }
    int pofs[MTW], voff[MTW];
    const int lanepart = (PARTS == 2) ? (g & 1) * a.Cout + (g >> 1) * 8 : g * 4;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int ty = wave * MTW + j;
        pofs[j] = (ty * FX + r) * PIXB;
        voff[j] = (ty * a.Wo + r) * (PARTS * a.Cout) + lanepart;
    }
    // PAIR: column r of the wave's tile = pair pp of row 2*wave + rr; lane group g reads input column
    // 2*pair + 2*half + (g >> 1) (even ones in the first half of the LDS row, odd ones in the second), channel octet
    // g & 1, and ends up with channels (g & 1)*4.. of pixel 2*pair + (g >> 1): everything below is that one pixel's.
    // The bits of r are dealt to (row, pair) so that the 16 lanes of every ds_read_b128 service group land on 16
    // distinct 16-byte bank groups with the 576-byte row pitch (rr = r & 1, pair = r bits 2,3,1: found by enumeration;
    // the natural r = row*8 + pair order measured 50 % of all LDS cycles as bank conflicts)
    int prow = 0, pcol = 0;
    if constexpr (PAIR) {
        static_assert(FX * PIXB == 576, "the conflict-free lane order was derived for this row pitch");
        const int rr = r & 1, pp = ((r >> 2) & 3) | (((r >> 1) & 1) << 2);
        prow = wave * 2 + rr;
        pcol = 2 * pp + (g >> 1);
        pofs[0] = (prow * FX + ((g >> 1) ? FX / 2 : 0) + pp) * PIXB + (g & 1) * 16;
        voff[0] = (prow * a.Wo + pcol) * (PARTS * 8) + ((PARTS == 2) ? (g & 1) * 8 : (g & 1) * 4);
    }
    const bool packed = !PAIR && (a.Cout == 8 && !a.outf && MTW % 2 == 0);   // two 8-channel result tiles share one epilogue
    static_assert(!LEAN || PARTS == 2, "the straight-line epilogue exists for split-bf16 storage");
    const bool lean_relu = a.relu == 1;

    // Ring protocol.  Window n of the stream reads slices n, n+1, n+2 (ring slots n, n+1, n+2 mod RING).  The prologue
    // queues slices 0 .. RING-2; iteration n queues slice n+RING-1 into the slot of slice n-1 (every wave left it
    // before the barrier of iteration n-1) -- ALWAYS, also past the end of the stream (zero page into a slot nobody
    // reads again), so the number of DMA pieces issued after any given slice is the same in every iteration.  vmcnt
    // retires in issue order, so "at most (RING-4)*PPW operations outstanding" means slice n+3 and everything older
    // (stores and residual loads of earlier iterations included) has landed while the RING-4 youngest slices stay in
    // flight across the barrier.  The barrier is the raw s_barrier: __syncthreads() would drain the DMA queue.
    // With a residual (RES) the step's residual pieces are requested right before its slice, so only that one youngest
    // slice may stay in flight at the wait (the pieces must have arrived: they are consumed right after it).
    constexpr int INFLIGHT = RES ? (RING > 4 ? PPW : 0) : (RING - 4) * PPW;
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) issue_next();

    // ---- the filter: 15 A-fragments per part, resident for the whole walk --------------------------------------
    short8 w[NCH][PARTS];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) w[c][pt] = wp[(c * PARTS + pt) * 64];
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + (PAIR ? (g & 1) : g) * 4);
    // everything queued so far (prologue slices, filter, bias) is waited for with a compiler-visible vmcnt(0): beside
    // LDS-DMA hipcc cannot count ordinary loads and would otherwise drain the queue at the first MFMA of every step
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int sidx = 0;   // ring slot of the window's first slice
    // Software pipeline over the 15 chunks, operand fragments DEPTH chunks ahead of the MFMAs.  The LDS reads and
    // their waits are inline asm: while an LDS-DMA is outstanding hipcc degrades every lgkmcnt wait to
    // lgkmcnt(0) (it models global_load_lds as a FLAT access that may also return through LGKM), which would
    // serialise read -> wait -> MFMA.  DS operations retire in order, so "at most (chunks still ahead) * RPC
    // reads outstanding" is exactly "chunk c has arrived"; the wait is tied to the fragment registers so the
    // MFMAs stay behind it.  The first DEPTH chunks of a window lie in its first slice, resident since two steps: they are
    // requested right behind the previous step's barrier and travel under its epilogue (round 5).
    constexpr int RPC = MTW * PARTS;   // ds_read_b128 per chunk
    constexpr int DEPTH = PAIR ? 2 : 1;
    short8 x[DEPTH + 1][MTW][PARTS];
    int sb[3];   // slot byte offsets of the three slices the window at sidx reads
    auto set_window = [&]() {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            int sl = sidx + dz;
            if (sl >= RING) sl -= RING;
            sb[dz] = sl * SLOTB;
        }
    };
    auto fetch = [&](int c, short8 (&dst)[MTW][PARTS]) {
        if constexpr (PAIR) {
            // chunk c = (slice c/6, filter row (c%6)/2, half c%2): a compile-time offset from the lane's base
            const unsigned ad = lds0 + sb[c / 6] + pofs[0];
            const int imm = (((c % 6) / 2) * FX + (c % 2)) * PIXB;
            switch (imm) {   // the offset must be an immediate: one case per (filter row, half)
#define DFFW_PAIR_RD(I)                                                                                                  \
    case I:                                                                                                              \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0][0]) : "v"(ad), "n"(I));                              \
        if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0][1]) : "v"(ad), "n"(I + PLANEB)); \
        break;
                DFFW_PAIR_RD(0)
                DFFW_PAIR_RD(PIXB)
                DFFW_PAIR_RD(FX * PIXB)
                DFFW_PAIR_RD(FX * PIXB + PIXB)
                DFFW_PAIR_RD(2 * FX * PIXB)
                DFFW_PAIR_RD(2 * FX * PIXB + PIXB)
#undef DFFW_PAIR_RD
            }
        } else {
            const unsigned ko = lds0 + sb[c / 5] + inoff[c % 5];
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                const unsigned ad = ko + pofs[j];
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[j][0]) : "v"(ad));
                if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[j][1]) : "v"(ad), "n"(PLANEB));
            }
        }
    };
    set_window();
#pragma unroll
    for (int c = 0; c < DEPTH; ++c) fetch(c, x[c]);   // (the prologue's slices have landed: barrier above)
    StepTrace trc(a.trace, wave, lane, NWAVES);
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t obase0 = (((int64_t)U.b * a.No + U.zbeg) * a.Ho + U.gy0) * a.Wo + U.gx0;
        for (int st = 0; st < U.nz + 2; ++st) {
            const bool live = st < U.nz;   // windows starting on the unit's last two slices straddle two units: no output
            const int64_t obase = obase0 + (int64_t)st * a.Ho * a.Wo;
            const int64_t ubase = obase * (PARTS * a.Cout);
            trc.stamp(0);
            // residual pieces of this step's outputs: requested BEFORE this iteration's slice so that the counted wait
            // below covers them (an ordinary load would make hipcc drain the whole DMA queue at its first use)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 rq[RES ? MTW : 1];
            if constexpr (RES && PARTS == 2) {
#pragma unroll
                for (int j = 0; j < MTW; ++j) rq[j] = u32x4{0, 0, 0, 0};
            }
            if (RES && live && PARTS == 2) {
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    if (packed && (j & 1)) continue;
                    const int vo = (packed && lane >= 32) ? voff[(j + 1) % MTW] - 8 : voff[j];
                    const uint16_t *rp = a.res0 + ubase + vo;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[j]) : "v"(rp) : "memory");
                }
            }
            if (!(a.dbg & 1)) issue_next();
            trc.stamp(1);

            f32x4 acc[MTW];
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MTW; ++j) acc[j] = bias4;
            if (live && !(a.dbg & 2)) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + DEPTH < NCH) fetch(c + DEPTH, x[(c + DEPTH) % (DEPTH + 1)]);
                    auto &xc = x[c % (DEPTH + 1)];
                    const int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * RPC;   // compile-time after unrolling
                    if (ahead == 2 * RPC && DEPTH == 2) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0][0]) : "n"(2 * RPC));
                    else if (ahead == RPC) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0][0]) : "n"(RPC));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0][0]));
#pragma unroll
                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt)
                            if (j + pt) asm volatile("" : "+v"(xc[j][pt]));
                    if constexpr (MTW == 1 && PARTS == 2) {
                        // one operand tile per wave: the three products of a chunk would form one dependent chain (a dependent
                        // 16x16x32 MFMA issues every ~26 cycles, an independent one every 16): the cross terms go to a second
                        // accumulator, summed after the loop
                        acc2 = mma<F16>(w[c][1], xc[0][0], acc2);
                        acc[0] = mma<F16>(w[c][0], xc[0][0], acc[0]);
                        acc2 = mma<F16>(w[c][0], xc[0][1], acc2);
                    } else {
                        if constexpr (PARTS == 2) {
#pragma unroll
                            for (int j = 0; j < MTW; ++j) acc[j] = mma<F16>(w[c][1], xc[j][0], acc[j]);
#pragma unroll
                            for (int j = 0; j < MTW; ++j) acc[j] = mma<F16>(w[c][0], xc[j][1], acc[j]);
                        }
#pragma unroll
                        for (int j = 0; j < MTW; ++j) acc[j] = mma<F16>(w[c][0], xc[j][0], acc[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MTW == 1 && PARTS == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[0][i] += acc2[i];
                }
            }

            trc.stamp(2);
            // slice n+3 has landed for this wave's pieces (and this step's residual pieces, which are older); after the
            // barrier for everyone's, and everyone is done reading slice n (its slot is the next DMA target).  The stores
            // below get a whole iteration to drain.
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
            if constexpr (RES && PARTS == 2) {
#pragma unroll
                for (int j = 0; j < MTW; ++j) asm volatile("" : "+v"(rq[j]));
            }
            sidx = (sidx + 1 == RING) ? 0 : sidx + 1;
            set_window();
#pragma unroll
            for (int c = 0; c < DEPTH; ++c) fetch(c, x[c]);   // the next window's first chunks (also behind the last step: the slots exist)
            trc.stamp(3);
            if (!live) {
                trc.next();
                continue;
            }
            if ((a.dbg & 4) && acc[0][0] != 12345.f) continue;

            // ---- epilogue of output slice zbeg + st (shared with conv_tile / conv_igemm) ----------------------------
            if constexpr (LEAN) {
                {
                    uint16_t *ob = a.out ? a.out + ubase : nullptr, *obp = a.out_pre ? a.out_pre + ubase : nullptr;
                    const f32x4 nocls = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (PAIR) {
                        epilogue_lean<PREC, RES>(ob, obp, voff[0], acc[0], RES ? make_uint4(rq[0][0], rq[0][1], rq[0][2], rq[0][3]) : uint4{}, lean_relu, nocls);
                    } else if (packed) {
#pragma unroll
                        for (int j = 0; j + 1 < MTW; j += 2) {
                            f32x4 q;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i]), __float_as_uint(acc[j + 1][i]), false, false);
                                q[i] = __uint_as_float(sw[0]);
                            }
                            const int vo = (lane >= 32) ? voff[j + 1] - 8 : voff[j];
                            epilogue_lean<PREC, RES>(ob, obp, vo, q, RES ? make_uint4(rq[j][0], rq[j][1], rq[j][2], rq[j][3]) : uint4{}, lean_relu, nocls);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < MTW; ++j)
                            epilogue_lean<PREC, RES>(ob, obp, voff[j], acc[j], RES ? make_uint4(rq[j][0], rq[j][1], rq[j][2], rq[j][3]) : uint4{}, lean_relu, nocls);
                    }
                    trc.stamp(4);
                    trc.next();
                    continue;
                }
            } else if constexpr (PAIR) {
                // lane rows 0-1 hold the even pixel's 8 channels, rows 2-3 the odd pixel's: both are "rows g & 1" of their
                // own pixel record, exactly the packed form of the 8-channel epilogue (no register shuffling needed)
                const int64_t opix = obase + (int64_t)prow * a.Wo + pcol;
                float cls = 0.f;
                if constexpr (RES && PARTS == 2) epilogue_quad<PREC, true, true, false>(a, acc[0], 0, g & 1, opix, true, cls, make_uint4(rq[0][0], rq[0][1], rq[0][2], rq[0][3]), uint4{}, ubase, voff[0]);
                else epilogue_quad<PREC, false, true, false>(a, acc[0], 0, g & 1, opix, true, cls, uint4{}, uint4{}, ubase, voff[0]);
                epilogue_cls(a, cls, g, opix, true, 2);
            } else if (packed) {
                // 8 output channels occupy lane rows 0-1 only: rows 2-3 take rows 0-1 of the next operand tile
#pragma unroll
                for (int j = 0; j + 1 < MTW; j += 2) {
                    f32x4 q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i]), __float_as_uint(acc[j + 1][i]), false, false);
                        q[i] = __uint_as_float(sw[0]);
                    }
                    const bool up = lane >= 32;
                    const int ty = wave * MTW + j + (up ? 1 : 0);
                    const int64_t opix = obase + (int64_t)ty * a.Wo + r;
                    const int vo = up ? voff[j + 1] - 8 : voff[j];
                    float cls = 0.f;
                    if constexpr (RES && PARTS == 2) epilogue_quad<PREC, true, true, false>(a, q, 0, g & 1, opix, true, cls, make_uint4(rq[j][0], rq[j][1], rq[j][2], rq[j][3]), uint4{}, ubase, vo);
                    else epilogue_quad<PREC, false, true, false>(a, q, 0, g & 1, opix, true, cls, uint4{}, uint4{}, ubase, vo);
                    epilogue_cls(a, cls, g, opix, true, 2);
                }
            } else {
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    const int64_t opix = obase + (int64_t)(wave * MTW + j) * a.Wo + r;
                    float cls = 0.f;
                    if constexpr (RES && PARTS == 2) epilogue_quad<PREC, true, true, false>(a, acc[j], 0, g, opix, true, cls, make_uint4(rq[j][0], rq[j][1], rq[j][2], rq[j][3]), uint4{}, ubase, voff[j]);
                    else epilogue_quad<PREC, false, true, false>(a, acc[j], 0, g, opix, true, cls, uint4{}, uint4{}, ubase, voff[j]);
                    epilogue_cls(a, cls, g, opix, true);
                }
            }
            trc.stamp(4);
            trc.next();
        }
    }
    // the slices queued past the end of the stream are still in flight: a wave must not retire before its LDS-DMA has
    // landed, or the pieces arrive in the LDS of whichever workgroup is given these bytes next
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- conv_roll_t: the same rolling window for the transposed 3x3x3 conv (stride (1,2,2), 16 -> 8 channels: `deconv_3`,
// `dres4.conv6`, DEN.py:41-48, 262-264) ---------------------------------------------------------------------------------
// out[oz, 2y+py, 2x+px] = sum over the taps of sub-pixel phase (py, px) (pack_conv: phase 0 along an axis uses filter index
// 1 at input i, phase 1 uses index 2 at input i and index 0 at input i+1).  conv_tile runs the four phases as four passes
// with 8 dead result rows each; here the two x phases of an input column ARE the two halves of the result tile: rows 0-7 =
// the 8 channels of output pixel 2x, rows 8-15 = of pixel 2x+1, contracting over input columns x and x+1 (32 = one chunk
// per slice and row tap).  9 chunks per input pixel instead of 14, no dead rows, and a lane pair writes 64 contiguous bytes
// (two adjacent output pixels), a wave 2 full rows of 1 KiB.  Two passes (py = 0, 1) per input slice, each with its own
// epilogue.  Streaming skeleton (column stream, ring of LDS slices, counted waits, inline-asm operand reads, residual
// prefetch) as in conv_roll above; the input footprint needs one extra row / column on the high side only.
template <int PREC, int TY, int TX, int NWAVES, int RING, bool RES, bool LEAN>
__global__ __launch_bounds__(NWAVES * 64) void conv_roll_t(const ConvArgs a, const RollArgs t) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int PIXB = 32;
    constexpr int FY = TY + 1, FX = TX + 1, FPIX = FY * FX;
    constexpr int NPIECE = 6;                              // 1 KiB wave instructions per plane (5 would do: 6 keeps 3 per wave)
    static_assert(NPIECE * 64 >= FPIX * 2, "plane holds the footprint");
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int MTW = TY / NWAVES;                       // input rows (16-column operand tiles) per wave
    constexpr int NCH = 9;                                 // chunks: py = 0: 3 slices; py = 1: 3 slices x 2 row taps
    static_assert(TX == 16 && TY % NWAVES == 0, "one operand tile = one 16-pixel input row");
    constexpr int NP = PARTS * NPIECE;
    constexpr int PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (the counted vmcnt waits rely on it)");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOTB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    auto decode = [&](int u) {   // units = columns of the INPUT grid
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

    const int ps0 = PARTS * a.C0;
    const int slice_elems = a.Hi * a.Wi * ps0;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0, fslices = 0, fz0 = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fslices = c.nz + 2;
        fz0 = c.zbeg - 1;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, pix = ci >> 1, oct = ci & 1;
            const int fy = pix / FX, fx = pix - fy * FX;
            const int iy = c.gy0 + fy, ix = c.gx0 + fx;          // halo on the high side only
            fok[k] = p < NP && pix < FPIX && iy < a.Hi && ix < a.Wi;
            const uint16_t *sp = a.in0 + (int64_t)c.b * a.Ni * slice_elems;
            fsrc[k] = sp + (int64_t)(iy * a.Wi + ix) * ps0 + part * a.C0 + oct * 8;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const int iz = fz0 + fq;
        const bool zin = (unsigned)iz < (unsigned)a.Ni && fu < uend;
        unsigned char *slot = smem + fslot * SLOTB;
        const int64_t zo = (int64_t)iz * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RING) ? 0 : fslot + 1;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // operand addressing: row j of this wave, input column r + (g >> 1), channel octet g & 1; the lane ends up with
    // channels (g & 1)*4.. of output pixel (2*row + py, 2*r + (g >> 1))
    int pofs[MTW], voff[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int ty = wave * MTW + j;
        pofs[j] = (ty * FX + r + (g >> 1)) * PIXB + (g & 1) * 16;
        voff[j] = (2 * ty * a.Wo + 2 * r + (g >> 1)) * (PARTS * 8) + ((PARTS == 2) ? (g & 1) * 8 : (g & 1) * 4);
    }
    const int rowstep = a.Wo * (PARTS * 8);   // elements from output row 2*ty to 2*ty + 1
    static_assert(!LEAN || PARTS == 2, "the straight-line epilogue exists for split-bf16 storage");
    const bool lean_relu = a.relu == 1, lean_cls = a.cls_w != nullptr;
    f32x4 clsw = {0.f, 0.f, 0.f, 0.f};
    if (LEAN && a.cls_w) clsw = *reinterpret_cast<const f32x4 *>(a.cls_w + (g & 1) * 4);

    constexpr int INFLIGHT = RES ? (RING > 4 ? PPW : 0) : (RING - 4) * PPW;
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) issue_next();

    short8 w[NCH][PARTS];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) w[c][pt] = wp[(c * PARTS + pt) * 64];
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + (g & 1) * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int sidx = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t obase0 = (((int64_t)U.b * a.No + U.zbeg) * a.Ho + 2 * U.gy0) * a.Wo + 2 * U.gx0;
        for (int st = 0; st < U.nz + 2; ++st) {
            const bool live = st < U.nz;
            const int64_t obase = obase0 + (int64_t)st * a.Ho * a.Wo;
            const int64_t ubase = obase * (PARTS * 8);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 rq[RES ? 2 * MTW : 1];
            if constexpr (RES && PARTS == 2) {
#pragma unroll
                for (int k = 0; k < 2 * MTW; ++k) rq[k] = u32x4{0, 0, 0, 0};
            }
            if (RES && live && PARTS == 2) {
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        const uint16_t *rp = a.res0 + ubase + voff[j] + py * rowstep;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[py * MTW + j]) : "v"(rp) : "memory");
                    }
            }
            if (!(a.dbg & 1)) issue_next();

            f32x4 acc[2][MTW];
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int j = 0; j < MTW; ++j) acc[py][j] = bias4;
            if (live && !(a.dbg & 2)) {
                int sb[3];
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    int sl = sidx + dz;
                    if (sl >= RING) sl -= RING;
                    sb[dz] = sl * SLOTB;
                }
                constexpr int RPC = MTW * PARTS;
                constexpr int DEPTH = 2;
                short8 x[DEPTH + 1][MTW][PARTS];
                // chunk c: c < 3: py = 0, slice c, input row y; c >= 3: py = 1, slice (c-3)/2, input row y + (c-3)%2
                auto fetch = [&](int c, short8 (&dst)[MTW][PARTS]) {
                    const int dz = c < 3 ? c : (c - 3) / 2;
                    const bool down = c >= 3 && ((c - 3) & 1);
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        const unsigned ad = lds0 + sb[dz] + pofs[j];
                        if (!down) {
                            asm volatile("ds_read_b128 %0, %1" : "=v"(dst[j][0]) : "v"(ad));
                            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[j][1]) : "v"(ad), "n"(PLANEB));
                        } else {
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[j][0]) : "v"(ad), "n"(FX * PIXB));
                            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[j][1]) : "v"(ad), "n"(FX * PIXB + PLANEB));
                        }
                    }
                };
#pragma unroll
                for (int c = 0; c < DEPTH; ++c) fetch(c, x[c]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + DEPTH < NCH) fetch(c + DEPTH, x[(c + DEPTH) % (DEPTH + 1)]);
                    auto &xc = x[c % (DEPTH + 1)];
                    const int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * RPC;
                    if (ahead == 2 * RPC) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0][0]) : "n"(2 * RPC));
                    else if (ahead == RPC) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0][0]) : "n"(RPC));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0][0]));
#pragma unroll
                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt)
                            if (j + pt) asm volatile("" : "+v"(xc[j][pt]));
                    const int py = c < 3 ? 0 : 1;
                    if constexpr (PARTS == 2) {
#pragma unroll
                        for (int j = 0; j < MTW; ++j) acc[py][j] = mma<F16>(w[c][1], xc[j][0], acc[py][j]);
#pragma unroll
                        for (int j = 0; j < MTW; ++j) acc[py][j] = mma<F16>(w[c][0], xc[j][1], acc[py][j]);
                    }
#pragma unroll
                    for (int j = 0; j < MTW; ++j) acc[py][j] = mma<F16>(w[c][0], xc[j][0], acc[py][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }

            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
            if constexpr (RES && PARTS == 2) {
#pragma unroll
                for (int k = 0; k < 2 * MTW; ++k) asm volatile("" : "+v"(rq[k]));
            }
            sidx = (sidx + 1 == RING) ? 0 : sidx + 1;
            if (!live) continue;
            if ((a.dbg & 4) && acc[0][0][0] != 12345.f) continue;

            // ---- epilogues: output rows 2*row (py = 0) and 2*row + 1 (py = 1); lane rows 0-1 = pixel 2*r, rows 2-3 = pixel 2*r + 1
            if constexpr (LEAN) {
                {   // straight-line routine (dffw_device.h: epilogue_lean), with or without the fused classifier
                    uint16_t *ob = a.out ? a.out + ubase : nullptr, *obp = a.out_pre ? a.out_pre + ubase : nullptr;
#pragma unroll
                    for (int py = 0; py < 2; ++py)
#pragma unroll
                        for (int j = 0; j < MTW; ++j) {
                            const int vo = voff[j] + py * rowstep;
                            uint4 q4 = uint4{};
                            if constexpr (RES) {
                                const u32x4 q = rq[py * MTW + j];
                                q4 = make_uint4(q[0], q[1], q[2], q[3]);
                            }
                            if (lean_cls) {
                                const float cls = epilogue_lean<PREC, RES, true>(ob, obp, vo, acc[py][j], q4, lean_relu, clsw);
                                const int64_t opix = obase + (int64_t)(2 * (wave * MTW + j) + py) * a.Wo + 2 * r + (g >> 1);
                                epilogue_cls(a, cls, g, opix, true, 2);
                            } else {
                                epilogue_lean<PREC, RES, false>(ob, obp, vo, acc[py][j], q4, lean_relu, clsw);
                            }
                        }
                    continue;
                }
            }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    const int ty = wave * MTW + j;
                    const int64_t opix = obase + (int64_t)(2 * ty + py) * a.Wo + 2 * r + (g >> 1);
                    const int vo = voff[j] + py * rowstep;
                    float cls = 0.f;
                    if constexpr (RES && PARTS == 2) {
                        const u32x4 q = rq[py * MTW + j];
                        epilogue_quad<PREC, true, true, false>(a, acc[py][j], 0, g & 1, opix, true, cls, make_uint4(q[0], q[1], q[2], q[3]), uint4{}, ubase, vo);
                    } else {
                        epilogue_quad<PREC, false, true, false>(a, acc[py][j], 0, g & 1, opix, true, cls, uint4{}, uint4{}, ubase, vo);
                    }
                    epilogue_cls(a, cls, g, opix, true, 2);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave (see conv_roll)
}

// ---- conv_roll_t32: transposed 3x3x3 conv, 32 -> 16 channels (`deconv_2`, `dres3.conv6`), one output row phase per launch ----
// 16 output channels fill the result rows, so the x phases cannot share a tile, and the filter of all four phases (27 taps x 32
// channels x 16 = 27 chunks, 216 VGPRs) does not fit beside the pipeline.  Two launches: sweep PY computes the output rows
// 2y + PY (both x phases) with only that row phase's taps resident (9 / 18 chunks = 72 / 144 VGPRs); each sweep streams the
// input once (it is a quarter of the output's size).  A chunk = one tap x 32 channels (K octet g = channel octet g), so every
// operand address is the lane's base plus an immediate.  Skeleton as conv_roll_t; ring of 4 slices (20 KB each).
template <int PREC, int PY, int RING, bool RES, bool LEAN>
__global__ __launch_bounds__(256) void conv_roll_t32(const ConvArgs a, const RollArgs t) {
    constexpr int TY = PY ? 4 : 8, TX = 16, NWAVES = 4;   // phase 1 carries twice the filter: one input row per wave instead of two
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int PIXB = 64;
    constexpr int FY = TY + 1, FX = TX + 1, FPIX = FY * FX;
    constexpr int NPIECE = PY ? 6 : (PARTS == 2 ? 10 : 12);   // 1 KiB wave instructions per plane: (TY+1) x 17 pixels x 4 channel octets, rounded so that every wave issues the same count
    static_assert(NPIECE * 64 >= FPIX * 4, "plane holds the footprint");
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int MTW = TY / NWAVES;                       // input rows (16-column operand tiles) per wave
    constexpr int NROW = PY ? 2 : 1;                       // row taps of this phase: py = 0: filter row 1 at input row y; py = 1: row 2 at y, row 0 at y+1
    constexpr int NCH0 = 3 * NROW, NCH1 = 6 * NROW, NCH = NCH0 + NCH1;   // chunks of x phase 0 (1 column tap) and x phase 1 (2 column taps)
    static_assert(TX == 16 && TY % NWAVES == 0, "one operand tile = one 16-pixel input row");
    constexpr int NP = PARTS * NPIECE;
    constexpr int PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (the counted vmcnt waits rely on it)");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOTB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    auto decode = [&](int u) {   // units = columns of the INPUT grid
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

    const int ps0 = PARTS * a.C0;
    const int slice_elems = a.Hi * a.Wi * ps0;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0, fslices = 0, fz0 = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fslices = c.nz + 2;
        fz0 = c.zbeg - 1;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, pix = ci >> 2, oct = ci & 3;
            const int fy = pix / FX, fx = pix - fy * FX;
            const int iy = c.gy0 + fy, ix = c.gx0 + fx;          // halo on the high side only
            // (a 16-channel input runs on the same kernel: its channel octets 2, 3 read the zero page and carry zero weights)
            fok[k] = p < NP && pix < FPIX && iy < a.Hi && ix < a.Wi && oct * 8 < a.C0;
            const uint16_t *sp = a.in0 + (int64_t)c.b * a.Ni * slice_elems;
            fsrc[k] = sp + (int64_t)(iy * a.Wi + ix) * ps0 + part * a.C0 + oct * 8;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const int iz = fz0 + fq;
        const bool zin = (unsigned)iz < (unsigned)a.Ni && fu < uend;
        unsigned char *slot = smem + fslot * SLOTB;
        const int64_t zo = (int64_t)iz * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RING) ? 0 : fslot + 1;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // operand addressing: row j of this wave, input column r, channel octet g; the lane ends up with channels 4g.. of output
    // pixels (2*row + PY, 2*r) and (2*row + PY, 2*r + 1)
    int pofs[MTW], voff[MTW];
    const int lanepart = (PARTS == 2) ? (g & 1) * 16 + (g >> 1) * 8 : g * 4;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int ty = wave * MTW + j;
        pofs[j] = (ty * FX + r) * PIXB + g * 16;
        voff[j] = ((2 * ty + PY) * a.Wo + 2 * r) * (PARTS * 16) + lanepart;
    }
    constexpr int pxstep_c = 16;   // (x PARTS) elements from output pixel 2r to 2r + 1
    static_assert(!LEAN || PARTS == 2, "the straight-line epilogue exists for split-bf16 storage");
    f32x4 clsw = {0.f, 0.f, 0.f, 0.f};
    if (LEAN && a.cls_w) clsw = *reinterpret_cast<const f32x4 *>(a.cls_w + g * 4);

    constexpr int INFLIGHT = RES ? (RING > 4 ? PPW : 0) : (RING - 4) * PPW;
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) issue_next();

    short8 w[NCH][PARTS];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) w[c][pt] = wp[(c * PARTS + pt) * 64];
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + g * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int sidx = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t obase0 = (((int64_t)U.b * a.No + U.zbeg) * a.Ho + 2 * U.gy0) * a.Wo + 2 * U.gx0;
        for (int st = 0; st < U.nz + 2; ++st) {
            const bool live = st < U.nz;
            const int64_t obase = obase0 + (int64_t)st * a.Ho * a.Wo;
            const int64_t ubase = obase * (PARTS * 16);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 rq[RES ? 2 * MTW : 1];
            if constexpr (RES && PARTS == 2) {
#pragma unroll
                for (int k = 0; k < 2 * MTW; ++k) rq[k] = u32x4{0, 0, 0, 0};
            }
            if (RES && live && PARTS == 2) {
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        const uint16_t *rp = a.res0 + ubase + voff[j] + px * (PARTS * pxstep_c);
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[px * MTW + j]) : "v"(rp) : "memory");
                    }
            }
            if (!(a.dbg & 1)) issue_next();

            f32x4 acc[2][MTW];   // [x phase][row]
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int j = 0; j < MTW; ++j) acc[px][j] = bias4;
            if (live && !(a.dbg & 2)) {
                int sb[3];
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    int sl = sidx + dz;
                    if (sl >= RING) sl -= RING;
                    sb[dz] = sl * SLOTB;
                }
                constexpr int RPC = MTW * PARTS;
                constexpr int DEPTH = 2;
                short8 x[DEPTH + 1][MTW][PARTS];
                // chunk c < NCH0: x phase 0, (slice d, row tap rt) = (c / NROW, c % NROW), input column x;
                // c >= NCH0: x phase 1, e = c - NCH0: (d, rt, column tap ct) = (e / (2*NROW), (e / 2) % NROW, e % 2): ct = 0 -> column x, 1 -> x+1
                auto fetch = [&](int c, short8 (&dst)[MTW][PARTS]) {
                    const int e = c - NCH0;
                    const int dz = c < NCH0 ? c / NROW : e / (2 * NROW);
                    const int rt = c < NCH0 ? c % NROW : (e / 2) % NROW;
                    const int ct = c < NCH0 ? 0 : e % 2;
                    const int dy = (PY && rt == 1) ? 1 : 0;
                    const int imm = (dy * FX + ct) * PIXB;
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        const unsigned ad = lds0 + sb[dz] + pofs[j] + imm;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[j][0]) : "v"(ad));
                        if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[j][1]) : "v"(ad), "n"(PLANEB));
                    }
                };
#pragma unroll
                for (int c = 0; c < DEPTH; ++c) fetch(c, x[c]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + DEPTH < NCH) fetch(c + DEPTH, x[(c + DEPTH) % (DEPTH + 1)]);
                    auto &xc = x[c % (DEPTH + 1)];
                    const int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * RPC;
                    if (ahead == 2 * RPC) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0][0]) : "n"(2 * RPC));
                    else if (ahead == RPC) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0][0]) : "n"(RPC));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0][0]));
#pragma unroll
                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt)
                            if (j + pt) asm volatile("" : "+v"(xc[j][pt]));
                    const int px = c < NCH0 ? 0 : 1;
                    if constexpr (PARTS == 2) {
#pragma unroll
                        for (int j = 0; j < MTW; ++j) acc[px][j] = mma<F16>(w[c][1], xc[j][0], acc[px][j]);
#pragma unroll
                        for (int j = 0; j < MTW; ++j) acc[px][j] = mma<F16>(w[c][0], xc[j][1], acc[px][j]);
                    }
#pragma unroll
                    for (int j = 0; j < MTW; ++j) acc[px][j] = mma<F16>(w[c][0], xc[j][0], acc[px][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }

            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
            if constexpr (RES && PARTS == 2) {
#pragma unroll
                for (int k = 0; k < 2 * MTW; ++k) asm volatile("" : "+v"(rq[k]));
            }
            sidx = (sidx + 1 == RING) ? 0 : sidx + 1;
            if (!live) continue;
            if ((a.dbg & 4) && acc[0][0][0] != 12345.f) continue;

            // ---- epilogues: output pixels 2*r (px = 0) and 2*r + 1 (px = 1) of row 2*row + PY ------------------------------------
            if constexpr (LEAN) {   // straight-line routine (dffw_device.h: epilogue_lean), with or without the fused classifier
                uint16_t *ob = a.out ? a.out + ubase : nullptr, *obp = a.out_pre ? a.out_pre + ubase : nullptr;
                const bool lean_relu = a.relu == 1;
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        const int vo = voff[j] + px * (PARTS * pxstep_c);
                        uint4 q4 = uint4{};
                        if constexpr (RES) {
                            const u32x4 q = rq[px * MTW + j];
                            q4 = make_uint4(q[0], q[1], q[2], q[3]);
                        }
                        if (a.cls_w) {
                            const float cls = epilogue_lean<PREC, RES, true>(ob, obp, vo, acc[px][j], q4, lean_relu, clsw);
                            const int64_t opix = obase + (int64_t)(2 * (wave * MTW + j) + PY) * a.Wo + 2 * r + px;
                            epilogue_cls(a, cls, g, opix, true);
                        } else {
                            epilogue_lean<PREC, RES, false>(ob, obp, vo, acc[px][j], q4, lean_relu, clsw);
                        }
                    }
                continue;
            }
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    const int ty = wave * MTW + j;
                    const int64_t opix = obase + (int64_t)(2 * ty + PY) * a.Wo + 2 * r + px;
                    const int vo = voff[j] + px * (PARTS * pxstep_c);
                    float cls = 0.f;
                    if constexpr (RES && PARTS == 2) {
                        const u32x4 q = rq[px * MTW + j];
                        epilogue_quad<PREC, true, true, false>(a, acc[px][j], 0, g, opix, true, cls, make_uint4(q[0], q[1], q[2], q[3]), uint4{}, ubase, vo);
                    } else {
                        epilogue_quad<PREC, false, true, false>(a, acc[px][j], 0, g, opix, true, cls, uint4{}, uint4{}, ubase, vo);
                    }
                    epilogue_cls(a, cls, g, opix, true);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave (see conv_roll)
}

// ---- conv_roll_efd: the EFD block of the 8-channel stage (DEN.py:306-315, `FM_conv1.0`) as one rolling kernel ------------
//     out = relu( BN(conv3x3x3 stride (1,2,2) (x)) + BN(conv3x3x3 (maxpool(1,2,2)(x))) ),   8 -> 16 channels, half resolution
// As two launches the strided branch writes its 16-channel result and the pooled branch reads it back as a residual.  Both
// contractions feed the SAME accumulators here (the two BatchNorm shifts add up in the accumulator init): a ring slot holds
// the footprints of one slice of x (9 x 33 pixels for a 4 x 16 output tile, even columns first so that the stride-2 operand
// reads stay contiguous) and of the pooled volume (6 x 18), K = [x: 3 slices x 3 chunks of 4 taps x 8 channels | pooled: same].
// With DUAL = false it is the plain strided conv (dres4.conv1).  Streaming skeleton as conv_roll.
template <int PREC, int RING, bool DUAL, bool LEAN>
__global__ __launch_bounds__(256) void conv_roll_efd(const ConvArgs a, const RollArgs t) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int TY = 4, TX = 16, NWAVES = 4, PIXB = 16;
    constexpr int XY = 2 * TY + 1, XX = 2 * TX + 1, XPIX = XY * XX, XEV = TX + 1;   // x footprint, XEV even columns per row
    constexpr int PY = TY + 2, PX = TX + 2, PPIX = PY * PX;                         // pooled footprint
    constexpr int XPIECES = 6, PPIECES = DUAL ? 2 : 0;                               // 1 KiB wave instructions per plane
    static_assert(XPIECES * 64 >= XPIX && (!DUAL || PPIECES * 64 >= PPIX), "plane holds the footprints");
    constexpr int NPIECE = XPIECES + PPIECES, PLANEB = NPIECE * 1024, SLOTB = PARTS * PLANEB, POFF = XPIECES * 1024;
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (the counted vmcnt waits rely on it)");
    constexpr int NCH = DUAL ? 18 : 9;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOTB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    auto decode = [&](int u) {   // columns of the OUTPUT grid
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

    const int rec = PARTS * 8;
    const int xslice = a.Hi * a.Wi * rec, pslice = a.Ho * a.Wo * rec;
    const uint16_t *fsrc[PPW];
    int fstride[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0, fslices = 0, fz0 = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fslices = c.nz + 2;
        fz0 = c.zbeg - 1;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            if (i < XPIECES) {
                const int sl = i * 64 + lane;
                const int fy = sl / XX, pos = sl - fy * XX;
                const int cx = pos < XEV ? 2 * pos : 2 * (pos - XEV) + 1;   // even columns first, then the odd ones
                const int iy = 2 * c.gy0 - 1 + fy, ix = 2 * c.gx0 - 1 + cx;
                fok[k] = p < NP && sl < XPIX && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
                fsrc[k] = a.in0 + (int64_t)c.b * a.Ni * xslice + (int64_t)(iy * a.Wi + ix) * rec + part * 8;
                fstride[k] = xslice;
            } else {
                const int sl = (i - XPIECES) * 64 + lane;
                const int fy = sl / PX, fx = sl - fy * PX;
                const int iy = c.gy0 - 1 + fy, ix = c.gx0 - 1 + fx;
                fok[k] = p < NP && sl < PPIX && (unsigned)iy < (unsigned)a.Ho && (unsigned)ix < (unsigned)a.Wo;
                fsrc[k] = a.in1 + (int64_t)c.b * a.Ni * pslice + (int64_t)(iy * a.Wo + ix) * rec + part * 8;
                fstride[k] = pslice;
            }
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const int iz = fz0 + fq;
        const bool zin = (unsigned)iz < (unsigned)a.Ni && fu < uend;
        unsigned char *slot = smem + fslot * SLOTB;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + (int64_t)iz * fstride[k] : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RING) ? 0 : fslot + 1;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // operand addressing: wave w = output row w of the column, lane column r; K octet g of chunk k3 = filter tap 4*k3 + g
    // (taps 9..11 carry zero weights).  x branch: input pixel (2*row + ky, 2*r + kx); pooled branch: (row + ky, r + kx).
    int tapX[3], tapP[3];
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
        const int tap = 4 * k3 + g;
        const int ky = tap < 9 ? tap / 3 : 0, kx = tap < 9 ? tap % 3 : 0;
        tapX[k3] = (ky * XX + (kx == 1 ? XEV : (kx == 2 ? 1 : 0))) * PIXB;
        tapP[k3] = (ky * PX + kx) * PIXB;
    }
    const int baseX = (2 * wave * XX + r) * PIXB, baseP = POFF + (wave * PX + r) * PIXB;
    const int lanepart = (PARTS == 2) ? (g & 1) * 16 + (g >> 1) * 8 : g * 4;
    const int voff = (wave * a.Wo + r) * (PARTS * 16) + lanepart;

    constexpr int INFLIGHT = (RING - 4) * PPW;
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) issue_next();

    short8 w[NCH][PARTS];
    {
        const short8 *wa = reinterpret_cast<const short8 *>(t.wroll) + lane;
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) w[c][pt] = wa[(c * PARTS + pt) * 64];
        if constexpr (DUAL) {
            const short8 *wb = reinterpret_cast<const short8 *>(t.wroll2) + lane;
#pragma unroll
            for (int c = 0; c < 9; ++c)
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) w[9 + c][pt] = wb[(c * PARTS + pt) * 64];
        }
    }
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + g * 4);
    if constexpr (DUAL) {
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(t.bias2 + g * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] += bb[i];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int sidx = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t obase0 = (((int64_t)U.b * a.No + U.zbeg) * a.Ho + U.gy0) * a.Wo + U.gx0;
        for (int st = 0; st < U.nz + 2; ++st) {
            const bool live = st < U.nz;
            const int64_t obase = obase0 + (int64_t)st * a.Ho * a.Wo;
            const int64_t ubase = obase * (PARTS * 16);
            if (!(a.dbg & 1)) issue_next();

            f32x4 acc = bias4;
            if (live && !(a.dbg & 2)) {
                int sb[3];
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    int sl = sidx + dz;
                    if (sl >= RING) sl -= RING;
                    sb[dz] = sl * SLOTB;
                }
                constexpr int DEPTH = 2;
                short8 x[DEPTH + 1][PARTS];
                auto fetch = [&](int c, short8 (&dst)[PARTS]) {
                    const int cc = c % 9, dz = cc / 3, k3 = cc % 3;
                    const unsigned ad = lds0 + sb[dz] + (c < 9 ? baseX + tapX[k3] : baseP + tapP[k3]);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                    if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "n"(PLANEB));
                };
#pragma unroll
                for (int c = 0; c < DEPTH; ++c) fetch(c, x[c]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + DEPTH < NCH) fetch(c + DEPTH, x[(c + DEPTH) % (DEPTH + 1)]);
                    auto &xc = x[c % (DEPTH + 1)];
                    const int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * PARTS;
                    if (ahead == 2 * PARTS) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0]) : "n"(2 * PARTS));
                    else if (ahead == PARTS) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0]) : "n"(PARTS));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0]));
                    if constexpr (PARTS == 2) {
                        asm volatile("" : "+v"(xc[1]));
                        acc = mma<F16>(w[c][1], xc[0], acc);
                        acc = mma<F16>(w[c][0], xc[1], acc);
                    }
                    acc = mma<F16>(w[c][0], xc[0], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
            sidx = (sidx + 1 == RING) ? 0 : sidx + 1;
            if (!live) continue;
            if ((a.dbg & 4) && acc[0] != 12345.f) continue;
            if constexpr (LEAN) {
                epilogue_lean<PREC, false>(a.out ? a.out + ubase : nullptr, a.out_pre ? a.out_pre + ubase : nullptr, voff, acc, uint4{}, a.relu == 1,
                                           f32x4{0.f, 0.f, 0.f, 0.f});
            } else {
                const int64_t opix = obase + (int64_t)wave * a.Wo + r;
                float cls = 0.f;
                epilogue_quad<PREC, false, true, false>(a, acc, 0, g, opix, true, cls, uint4{}, uint4{}, ubase, voff);
                epilogue_cls(a, cls, g, opix, true);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave (see conv_roll)
}

// ---- conv_roll_s2: 3x3x3 stride (1,2,2) over 16 input channels as a rolling window (`FM_conv2.0.stride_conv`, `dres3.conv1`:
// 16 -> 32; `dres4.conv3`: 16 -> 16; DEN.py:306-315, 252-256) -------------------------------------------------------------------------
// On conv_tile these layers stage their 5 x 4 x 16 tile as TWO 8-channel stages of a 7 x 9 x 34-pixel footprint: every 128-byte
// line is fetched once per stage, the slice halo 7/5 and the in-plane halo 1.7x on top -- measured 2.1x the algorithmic HBM bytes,
// 2.5 TB/s algorithmic (profiles/r01_hbm_traffic.json, r02_fetch_size_calibration.txt).  Here, as in conv_roll_efd for the 8-channel
// stage: a workgroup walks the output slices of a column; a ring slot holds ONE input slice of the column's (2 TY + 1) x (2 TX + 1)
// footprint with ALL 16 channels (whole 64-byte pixel records: every line is fetched once), even columns first so that the stride-2
// operand reads stay contiguous; the filter lives in registers -- 3 slices x 5 chunks of (2 taps x 16 channels) for ONE 16-channel
// output tile per wave (120 VGPRs).  NT = 2 (32 output channels): the four waves are (output tile, operand tile) pairs over a
// 4 x 8 column; NT = 1: wave w = output row w of a 4 x 16 column.  Streaming skeleton (column stream, counted vmcnt, raw s_barrier,
// inline-asm operand reads) as conv_roll.
// KH = 2: 32 input channels.  The filter of one output tile no longer fits a wave (27 taps x 32 channels = 216 VGPRs), so the
// contraction is split between wave pairs by channel half: the workgroup has 8 waves = (channel half, output tile, operand tile),
// each with its half filter (120 VGPRs); the odd half hands its partial tile to its partner through LDS in front of the step's
// barrier (double-buffered by step parity) and the partner runs the epilogue.  t.pair = first 16-channel output tile of this launch
// (64 output channels = two launches over the same input).
template <int PREC, int NT, int KH, int RING, bool LEAN>
__global__ __launch_bounds__(NT * KH == 4 ? 512 : 256) void conv_roll_s2(const ConvArgs a, const RollArgs t) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int CIN = 16 * KH, NWAVES = (NT * KH == 4) ? 8 : 4;
    static_assert(KH == 1 || NT == 2, "the channel-split form is built for two output tiles (8 waves)");
    constexpr int TY = 4, TX = NT == 2 ? 8 : 16, PIXB = 2 * CIN, OCT = 2 * KH;
    constexpr int XY = 2 * TY + 1, XXR = 2 * TX + 1, XEV = TX + 1;   // footprint of a slice (XXR columns per row); XEV even columns per row
    // LDS row pitch: with two output rows per operand tile (NT = 2: rows 2*tl, 2*tl + 1 = input rows two apart) the 16 lanes of a ds_read_b128 row group
    // only cover 16 distinct bank groups when those rows are a multiple of 256 bytes apart, i.e. when the pitch is a multiple of 4 pixels (17-pixel
    // rows: every read collided two ways, lds_conflict 0.50 for three rounds); the pad columns read the zero page
    constexpr int XX = (NT == 2) ? (XXR + 3) / 4 * 4 : XXR, XPIX = XY * XX;
    constexpr int NPIECE = (XPIX * OCT + 63) / 64;                                   // 1 KiB wave instructions per plane (64 x (pixel, octet))
    constexpr int PLANEB = NPIECE * 1024, SLOTB = PARTS * PLANEB;
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;             // wave w issues pieces [w * PPW, min(NP, (w + 1) * PPW))
    constexpr int NCH = 15;
    static_assert(RING >= 4, "3 slices being read + at least one being filled");
    constexpr int XCHB = KH == 2 ? 2 * (NWAVES / 2) * 1024 : 0;                      // partial tiles of the odd channel halves, two step parities
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOTB + XCHB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        int b, gy0, gx0;
    };
    auto decode = [&](int u) {   // columns of the OUTPUT grid
        Unit c;
        const int txi = u % t.tiles_x;
        const int tt = u / t.tiles_x;
        c.b = tt / t.tiles_y;
        c.gy0 = (tt % t.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };

    const int rec = PARTS * CIN;
    const int xslice = a.Hi * a.Wi * rec;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0;
    const int fslices = a.No + 2;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, sl = ci / OCT, oct = ci % OCT;
            const int fy = sl / XX, pos = sl - fy * XX;
            const int cx = pos < XEV ? 2 * pos : 2 * (pos - XEV) + 1;   // even columns first, then the odd ones (pos >= XXR: pad column)
            const int iy = 2 * c.gy0 - 1 + fy, ix = 2 * c.gx0 - 1 + cx;
            fok[k] = p < NP && sl < XPIX && pos < XXR && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            fsrc[k] = a.in0 + (int64_t)c.b * a.Ni * xslice + (int64_t)(iy * a.Wi + ix) * rec + part * CIN + oct * 8;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const int iz = fq - 1;
        const bool zin = (unsigned)iz < (unsigned)a.Ni && fu < uend;
        unsigned char *slot = smem + fslot * SLOTB;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + (int64_t)iz * xslice : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RING) ? 0 : fslot + 1;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // this wave's channel half `kh`, output tile `nt` and operand tile `tl`: KH = 1, NT = 1: row `wave`, column r;
    // NT = 2: rows 2*tl + (r >> 3), column r & 7 with (nt, tl) = (wave & 1, wave >> 1), or (kh, nt, tl) = (wave & 1, (wave >> 1) & 1, wave >> 2)
    const int kh = KH == 2 ? (wave & 1) : 0;
    const int ntl = NT == 2 ? ((KH == 2 ? wave >> 1 : wave) & 1) : 0;
    const int tl = NT == 2 ? (KH == 2 ? wave >> 2 : wave >> 1) : wave;
    const int nt = t.pair + ntl;
    const int orow = NT == 2 ? 2 * tl + (r >> 3) : tl, ocol = NT == 2 ? (r & 7) : r;
    // K octet g of chunk k5 = (in-slice tap 2*k5 + (g >> 1), channel octet g & 1 of this wave's half); tap 9 carries zero weights
    int tapo[5];
#pragma unroll
    for (int k5 = 0; k5 < 5; ++k5) {
        const int tap = 2 * k5 + (g >> 1);
        const int ky = tap < 9 ? tap / 3 : 0, kx = tap < 9 ? tap % 3 : 0;
        tapo[k5] = (ky * XX + (kx == 1 ? XEV : (kx == 2 ? 1 : 0))) * PIXB + kh * 32 + (g & 1) * 16;
    }
    const int base = (2 * orow * XX + ocol) * PIXB;
    const int Cout = a.Cout;
    const int lanepart = (PARTS == 2) ? (g & 1) * Cout + (g >> 1) * 8 : g * 4;
    const int voff = (orow * a.Wo + ocol) * (PARTS * Cout) + lanepart;

    // counted wait for this wave's own fill pieces: everything but the pieces of the newest RING - 4 slices has landed.  The
    // waves do not all issue the same number of pieces per slice, so the count is per wave (wave-uniform branch).
    const int mine = max(0, min(PPW, NP - wave * PPW));
    auto fill_wait = [&]() {
        if (mine == PPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 4) * PPW) : "memory");
        else if (PPW >= 2 && mine == PPW - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 4) * (PPW > 1 ? PPW - 1 : 0)) : "memory");
        else if (PPW >= 3 && mine == PPW - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 4) * (PPW > 2 ? PPW - 2 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // any other count: wait for everything (always sufficient)
    };
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) issue_next();

    short8 w[NCH][PARTS];
    {
        const short8 *wa = reinterpret_cast<const short8 *>(t.wroll) + ((int64_t)nt * KH + kh) * NCH * PARTS * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) w[c][pt] = wa[(c * PARTS + pt) * 64];
    }
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + nt * 16 + g * 4);
    if (kh) bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int sidx = 0, parity = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t obase0 = (((int64_t)U.b * a.No) * a.Ho + U.gy0) * a.Wo + U.gx0;
        for (int st = 0; st < a.No + 2; ++st) {
            const bool live = st < a.No;
            const int64_t obase = obase0 + (int64_t)st * a.Ho * a.Wo;
            const int64_t ubase = obase * (PARTS * Cout);
            if (!(a.dbg & 1)) issue_next();

            f32x4 acc = bias4;
            if (live && !(a.dbg & 2)) {
                int sb[3];
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    int sl = sidx + dz;
                    if (sl >= RING) sl -= RING;
                    sb[dz] = sl * SLOTB;
                }
                constexpr int DEPTH = 2;
                short8 x[DEPTH + 1][PARTS];
                auto fetch = [&](int c, short8 (&dst)[PARTS]) {
                    const unsigned ad = lds0 + sb[c / 5] + base + tapo[c % 5];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                    if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "n"(PLANEB));
                };
#pragma unroll
                for (int c = 0; c < DEPTH; ++c) fetch(c, x[c]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + DEPTH < NCH) fetch(c + DEPTH, x[(c + DEPTH) % (DEPTH + 1)]);
                    auto &xc = x[c % (DEPTH + 1)];
                    const int ahead = (NCH - 1 - c < DEPTH ? NCH - 1 - c : DEPTH) * PARTS;
                    if (ahead == 2 * PARTS) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0]) : "n"(2 * PARTS));
                    else if (ahead == PARTS) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xc[0]) : "n"(PARTS));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0]));
                    if constexpr (PARTS == 2) {
                        asm volatile("" : "+v"(xc[1]));
                        acc = mma<F16>(w[c][1], xc[0], acc);
                        acc = mma<F16>(w[c][0], xc[1], acc);
                    }
                    acc = mma<F16>(w[c][0], xc[0], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (KH == 2) {   // the odd channel half hands its partial tile to its partner (the wave below it)
                const unsigned xo = lds0 + RING * SLOTB + (parity * (NWAVES / 2) + (wave >> 1)) * 1024 + lane * 16;
                if (kh) asm volatile("ds_write_b128 %0, %1" ::"v"(xo), "v"(acc) : "memory");
                fill_wait();
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (!kh && live) {
                    f32x4 o;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(xo) : "memory");
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] += o[i];
                }
                parity ^= 1;
            } else {
                fill_wait();
                asm volatile("s_barrier" ::: "memory");
            }
            sidx = (sidx + 1 == RING) ? 0 : sidx + 1;
            if (!live || kh) continue;
            if ((a.dbg & 4) && acc[0] != 12345.f) continue;
            if constexpr (LEAN) {
                epilogue_lean<PREC, false>(a.out ? a.out + ubase : nullptr, a.out_pre ? a.out_pre + ubase : nullptr, voff + nt * 16, acc, uint4{},
                                           a.relu == 1, f32x4{0.f, 0.f, 0.f, 0.f});
            } else {
                const int64_t opix = obase + (int64_t)orow * a.Wo + ocol;
                float cls = 0.f;
                epilogue_quad<PREC, false, true, false>(a, acc, nt, g, opix, true, cls, uint4{}, uint4{}, ubase, voff);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave (see conv_roll)
}

// ---- host side ----------------------------------------------------------------------------------------------------
#define DFFW_ROLL_TY 8
#define DFFW_ROLL_TX 16
#define DFFW_ROLL_NW 4
#define DFFW_ROLL_RING 6

// The launch's epilogue fits the straight-line routine (epilogue_lean, dffw_device.h): split-bf16 storage, something to write,
// no relu-before-residual, no second / broadcast residual, no fp32 planar output; a residual only where the kernel family
// prefetches it (`res_variant`); whole 16-channel (or packed 8-channel) result tiles.
bool roll_lean(int prec, const ConvArgs &a, bool res_variant) {
    return prec == P_BF16X3 && (a.out || a.out_pre || a.cls_w) && !a.outf && !a.res1 && !a.res_bcast && a.relu != 2 && (res_variant || !a.res0) &&
           (a.Cout == 8 || a.Cout % 16 == 0) && !(a.dbg & DFFW_ARGS_NO_LEAN_ROLL);   // (the switch: generic epilogue, for A/B and the parity tests)
}
static const char *tf(bool b) { return b ? "true" : "false"; }

void roll_tile(int *ty, int *tx) {
    *ty = DFFW_ROLL_TY;
    *tx = DFFW_ROLL_TX;
}

void conv_roll_kernel_name(int prec, const ConvArgs &a, bool pair, char *buf, int n) {
    if (rollx_pair_ok(prec, a, pair)) return conv_rollx_pair_kernel_name(a, buf, n);
    const bool res = a.res0 != nullptr && prec == P_BF16X3;
    snprintf(buf, n, "dffw::conv_roll<%d, %d, %d, %d, %d, %s, %s, %s>", prec, DFFW_ROLL_TY, DFFW_ROLL_TX, DFFW_ROLL_NW, DFFW_ROLL_RING, tf(res), tf(pair),
             tf(roll_lean(prec, a, true) && !a.cls_w));
}

void conv_roll_t_kernel_name(int prec, const ConvArgs &a, char *buf, int n) {
    const bool res = a.res0 != nullptr && prec == P_BF16X3;
    snprintf(buf, n, "dffw::conv_roll_t<%d, %d, %d, %d, %d, %s, %s>", prec, DFFW_ROLL_TY, DFFW_ROLL_TX, DFFW_ROLL_NW, DFFW_ROLL_RING, tf(res), tf(roll_lean(prec, a, true)));
}

hipError_t launch_conv_roll_t(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 512;
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(DFFW_ROLL_NW * 64);
    const bool res = a.res0 != nullptr && prec == P_BF16X3;
    const bool lean = roll_lean(prec, a, true);
#define DFFW_ROLLT_LAUNCH(P, R, L) hipLaunchKernelGGL((conv_roll_t<P, DFFW_ROLL_TY, DFFW_ROLL_TX, DFFW_ROLL_NW, DFFW_ROLL_RING, R, L>), grid, block, 0, s, a, t)
    switch (prec) {
        case P_BF16X3:
            if (res && lean) DFFW_ROLLT_LAUNCH(P_BF16X3, true, true);
            else if (res) DFFW_ROLLT_LAUNCH(P_BF16X3, true, false);
            else if (lean) DFFW_ROLLT_LAUNCH(P_BF16X3, false, true);
            else DFFW_ROLLT_LAUNCH(P_BF16X3, false, false);
            break;
        case P_FP16: DFFW_ROLLT_LAUNCH(P_FP16, false, false); break;
        case P_BF16: DFFW_ROLLT_LAUNCH(P_BF16, false, false); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_ROLLT_LAUNCH
    return hipGetLastError();
}

void conv_roll_t32_kernel_name(int prec, int py, const ConvArgs &a, char *buf, int n) {
    const bool res = a.res0 != nullptr && prec == P_BF16X3;
    snprintf(buf, n, "dffw::conv_roll_t32<%d, %d, %d, %s, %s>", prec, py, py ? 6 : 4, tf(res), tf(roll_lean(prec, a, true)));
}

hipError_t launch_conv_roll_t32(int prec, int py, const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 512;
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
    const bool res = a.res0 != nullptr && prec == P_BF16X3;
    const bool lean = roll_lean(prec, a, true);
#define DFFW_T32_LAUNCH(P, R, L)                                                                      \
    do {                                                                                              \
        if (py) hipLaunchKernelGGL((conv_roll_t32<P, 1, 6, R, L>), grid, block, 0, s, a, t);         \
        else hipLaunchKernelGGL((conv_roll_t32<P, 0, 4, R, L>), grid, block, 0, s, a, t);            \
    } while (0)
    switch (prec) {
        case P_BF16X3:
            if (res && lean) DFFW_T32_LAUNCH(P_BF16X3, true, true);
            else if (res) DFFW_T32_LAUNCH(P_BF16X3, true, false);
            else if (lean) DFFW_T32_LAUNCH(P_BF16X3, false, true);
            else DFFW_T32_LAUNCH(P_BF16X3, false, false);
            break;
        case P_FP16: DFFW_T32_LAUNCH(P_FP16, false, false); break;
        case P_BF16: DFFW_T32_LAUNCH(P_BF16, false, false); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_T32_LAUNCH
    return hipGetLastError();
}

void roll_t32_tile(int py, int *ty, int *tx) {
    *ty = py ? 4 : 8;
    *tx = 16;
}

void efd_roll_tile(int *ty, int *tx) {
    *ty = 4;
    *tx = 16;
}

void conv_roll_efd_kernel_name(int prec, const ConvArgs &a, bool dual, char *buf, int n) {
    snprintf(buf, n, "dffw::conv_roll_efd<%d, %d, %s, %s>", prec, 5, tf(dual), tf(roll_lean(prec, a, false) && !a.cls_w));
}

hipError_t launch_conv_roll_efd(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 512;
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
    const bool dual = t.wroll2 != nullptr;
    const bool lean = roll_lean(prec, a, false) && !a.cls_w;
#define DFFW_EFD_LAUNCH(P, L)                                                                       \
    do {                                                                                            \
        if (dual) hipLaunchKernelGGL((conv_roll_efd<P, 5, true, L>), grid, block, 0, s, a, t);     \
        else hipLaunchKernelGGL((conv_roll_efd<P, 5, false, L>), grid, block, 0, s, a, t);         \
    } while (0)
    switch (prec) {
        case P_BF16X3:
            if (lean) DFFW_EFD_LAUNCH(P_BF16X3, true);
            else DFFW_EFD_LAUNCH(P_BF16X3, false);
            break;
        case P_FP16: DFFW_EFD_LAUNCH(P_FP16, false); break;
        case P_BF16: DFFW_EFD_LAUNCH(P_BF16, false); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_EFD_LAUNCH
    return hipGetLastError();
}

void s2_roll_tile(int nt, int *ty, int *tx) {
    *ty = 4;
    *tx = nt == 2 ? 8 : 16;
}

void conv_roll_s2_kernel_name(int prec, int nt, int kh, const ConvArgs &a, char *buf, int n) {
    snprintf(buf, n, "dffw::conv_roll_s2<%d, %d, %d, %d, %s>", prec, nt, kh, kh == 2 ? 5 : (nt == 2 ? 6 : 4), tf(roll_lean(prec, a, false) && !a.cls_w));
}

hipError_t launch_conv_roll_s2(int prec, int nt, int kh, const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    // ring depth by LDS: 16 channels, 4 x 8 column: 10 KiB per slice -> 6 slots = 60 KiB (two workgroups per CU); 4 x 16 column:
    // 20 KiB -> 4 slots (two workgroups); 32 channels, 4 x 8 column, 8 waves: 24 KiB -> 5 slots + the exchange area = 128 KiB (one)
    const int want = t.wgs > 0 ? t.wgs : (kh == 2 ? 256 : 512);
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8))));
    const bool lean = roll_lean(prec, a, false) && !a.cls_w;
#define DFFW_S2_LAUNCH(P, L)                                                                                   \
    do {                                                                                                       \
        if (kh == 2) hipLaunchKernelGGL((conv_roll_s2<P, 2, 2, 5, L>), grid, dim3(512), 0, s, a, t);          \
        else if (nt == 2) hipLaunchKernelGGL((conv_roll_s2<P, 2, 1, 6, L>), grid, dim3(256), 0, s, a, t);     \
        else hipLaunchKernelGGL((conv_roll_s2<P, 1, 1, 4, L>), grid, dim3(256), 0, s, a, t);                  \
    } while (0)
    switch (prec) {
        case P_BF16X3:
            if (lean) DFFW_S2_LAUNCH(P_BF16X3, true);
            else DFFW_S2_LAUNCH(P_BF16X3, false);
            break;
        case P_FP16: DFFW_S2_LAUNCH(P_FP16, false); break;
        case P_BF16: DFFW_S2_LAUNCH(P_BF16, false); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_S2_LAUNCH
    return hipGetLastError();
}

hipError_t launch_conv_roll(int prec, const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    if (rollx_pair_ok(prec, a, t.pair != 0)) return launch_conv_rollx_pair(a, t, s);
    // persistent grid: two resident workgroups per CU (72 KiB of LDS each), a multiple of the 8 XCDs, never more
    // workgroups than an XCD has columns
    const int want = t.wgs > 0 ? t.wgs : 512;
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(DFFW_ROLL_NW * 64);
    // the hand-prefetched residual exists for the split-bf16 storage only; fp16/bf16 layers with a residual load it in the epilogue
    const bool res = a.res0 != nullptr && prec == P_BF16X3;
    const bool lean = roll_lean(prec, a, true) && !a.cls_w;
#define DFFW_ROLL_LAUNCH(P, R, Q, L) hipLaunchKernelGGL((conv_roll<P, DFFW_ROLL_TY, DFFW_ROLL_TX, DFFW_ROLL_NW, DFFW_ROLL_RING, R, Q, L>), grid, block, 0, s, a, t)
#define DFFW_ROLL_LAUNCH_Q(P, R, L)                   \
    do {                                              \
        if (t.pair) DFFW_ROLL_LAUNCH(P, R, true, L);  \
        else DFFW_ROLL_LAUNCH(P, R, false, L);        \
    } while (0)
    switch (prec) {
        case P_BF16X3:
            if (res && lean) DFFW_ROLL_LAUNCH_Q(P_BF16X3, true, true);
            else if (res) DFFW_ROLL_LAUNCH_Q(P_BF16X3, true, false);
            else if (lean) DFFW_ROLL_LAUNCH_Q(P_BF16X3, false, true);
            else DFFW_ROLL_LAUNCH_Q(P_BF16X3, false, false);
            break;
        case P_FP16: DFFW_ROLL_LAUNCH_Q(P_FP16, false, false); break;
        case P_BF16: DFFW_ROLL_LAUNCH_Q(P_BF16, false, false); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_ROLL_LAUNCH_Q
#undef DFFW_ROLL_LAUNCH
    return hipGetLastError();
}

}  // namespace dffw
