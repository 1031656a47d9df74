// conv_slice32: the per-slice 1x3x3 convolution with 32 input and 32 output channels as a persistent streaming kernel (round 5), gfx950 / MI355X.
// `FM_conv2.1.Focus_Measure.conv.{0,2}` (DEN.py:295-304: the two convs of the 32-channel SRD block's resnet_block_2d) and the 32 -> 32 convs of
// the alignment heads of End_to_End (E2E.py:33-61: `optical_flow_aggregation.conv2.{2,4}.0`).
//
// On conv_tile these layers spend half their time staging footprints (profiles/r05_ablation_conv_tile_phases.txt: fill alone 0.063 of 0.120 ms):
// a 5 x 4 x 16 block of a per-slice conv has no reuse along the slices, so every block pays a full (y, x) halo for 64-byte half-line requests,
// and fill -> barrier -> contract -> store run one after the other.  Here
//   * the 9 x 32 x 32 split-bf16 filter (37 KB) is small enough for EVERY wave to hold all of it: 9 chunks (one tap x 32 channels) x 2 output
//     tiles x (hi, lo) = 144 VGPRs -- no weight stream and no split of the contraction, hence no exchange of partial sums;
//   * a workgroup of 4 waves walks the slices of 8 x 16 columns (a slice is an independent 2-D convolution; walking them gives the stream its
//     length): slice images go through a ring of 3 LDS slots by buffer-addressed LDS-DMA two slices ahead of the one being contracted, one
//     barrier per slice; wave w contracts rows 2w, 2w + 1 of the column (two operand tiles of 16 pixels) and runs their epilogue itself;
//   * LDS slice image [16-channel group][part][row][pixel][octet] in 16-byte entries (conv_rollk's): lane rows g, g + 1 of an operand read take
//     the two octets of the same tap of 16 consecutive pixels -- 16 distinct 16-byte bank groups per ds_read_b128 service group, no padding;
//   * the operand fragments of chunk c + 1 are requested in front of chunk c's MFMAs, chunk 0 of the NEXT slice (already resident) in front of the
//     last chunk's, so the matrix pipe restarts right behind the barrier; two workgroups per CU run out of phase.
// Epilogue: out = [relu](acc + BatchNorm shift [+ residual]) in split-bf16 storage (epilogue_lean: the arithmetic and order of conv_tile's LEAN
// epilogue; the contraction order differs from conv_tile's stage walk only in the position of the channel halves inside a chunk).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "dffw_conv_roll.h"
#include "dffw_device.h"

namespace dffw {

namespace slice32 {
constexpr int TY = DFFW_SLICE_TY, TX = DFFW_SLICE_TX, FY = TY + 2, FX = TX + 2, NW = 4, RING = 3, NCH = SLICE32_CHUNKS;
static_assert(TY == 2 * NW && TX == 16, "a wave contracts two 16-pixel rows of the column");
constexpr int PARTE = FY * FX * 2;        // entries of one part of a 16-channel group: [row][pixel][octet]
constexpr int CQE = 2 * PARTE;            // ... of a group: [part][row][pixel][octet]
constexpr int SLOTE = 2 * CQE;
constexpr int NPIECE = (SLOTE + 63) / 64;
constexpr int SLOTB = NPIECE * 1024;
constexpr int PPW = (NPIECE + NW - 1) / NW;
constexpr int LDSB = RING * SLOTB;
static_assert(CQE % 16 == 0 && SLOTB % 256 == 0 && 2 * LDSB <= 160 * 1024, "LDS layout (two workgroups per CU)");
}   // namespace slice32

// SUMS: the row-sums variant (third conv of an alignment head, End_to_End.py:41-46, whose ReLU'd result only feeds the head's last conv + plane mean
// = plane sums): nothing is stored but, per output row segment of 16 pixels, its sum and its first and last pixel -- conv_tile's row-sums epilogue,
// same layout: a.outf[((plane * Ho + y) * tiles_x + tile column) * 3 + {sum, first, last}][32] fp32 (an operand tile here is such a row segment)
template <bool RELU, bool RES, bool SUMS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_slice32(const ConvArgs a, const RollArgs t) {
    static_assert(!SUMS || (RELU && !RES), "row sums: relu(acc), no residual");
    using namespace slice32;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDSB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;

    // ---- this workgroup's units (8 x 16 columns of one sample): XCD x owns a contiguous range, as conv_roll -------------
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
    auto decode = [&](int u) {
        Unit c;
        const int txi = u % t.tiles_x;
        const int tt = u / t.tiles_x;
        c.gx0 = txi * TX;
        c.gy0 = (tt % t.tiles_y) * TY;
        c.b = tt / t.tiles_y;
        return c;
    };

    // ---- fill: pieces of 64 consecutive 16-byte entries of the slot; per lane the byte offset from the unit's footprint origin (out-of-image
    // and padding lanes pushed out of range: the buffer range check writes their zeros = the conv's padding) ---------------------------------
    constexpr int recb = 128, partb = 64;              // a pixel record of the 32-channel source: [hi 32][lo 32]
    const char *tb = reinterpret_cast<const char *>(a.in0);
    const int slice_bytes = a.Hi * a.Wi * recb;
    int fvo[PPW];
    const char *fb = tb;
    int fu = ufirst, fz = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fz = 0;
        fb = tb + ((int64_t)c.b * a.Ni * a.Hi * a.Wi + (int64_t)(c.gy0 - 1) * a.Wi + (c.gx0 - 1)) * recb;
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // (opaque: hipcc would hoist the decode below out of the unit loop and keep its results in registers)
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int e = (k * NW + wave) * 64 + ln;       // entry inside the slot: [group][part][row][pixel][octet]
            const int cq = e / CQE, e2 = e - cq * CQE;
            const int part = e2 / PARTE, e3 = e2 - part * PARTE;
            const int fy = e3 / (2 * FX), e4 = e3 - fy * (2 * FX);
            const int fx = e4 >> 1, oct = e4 & 1;
            const int iy = c.gy0 - 1 + fy, ix = c.gx0 - 1 + fx;
            fvo[k] = (e < SLOTE && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) ? (fy * a.Wi + fx) * recb + part * partb + (cq * 2 + oct) * 16
                                                                                                 : (int)0x80000000;
        }
    };
    setup_fill();
    int fslotb = 0;
    auto issue_piece = [&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const int p = k * NW + wave;
        if (p >= NPIECE) return;                           // (wave-uniform)
        const bool zin = fu < uend;                        // past the end of the stream: zeros (the slot is never read)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb), 0, zin ? (int)0x80000000 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + fslotb + p * 1024), 16, fvo[k], zin ? fz * slice_bytes : 0, 0, 0);
    };
    auto advance_fill = [&]() {
        fslotb = (fslotb + SLOTB == RING * SLOTB) ? 0 : fslotb + SLOTB;
        if (++fz == a.Ni && fu < uend) {
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {                          // the fill runs two slices ahead
        static_for<PPW>([&](auto K) { issue_piece(K); });
        advance_fill();
    }

    // ---- operand addressing: K octet g of a chunk = input channels 8g .. 8g + 7 = (group g >> 1, octet g & 1) of the chunk's tap; lane r of
    // operand tile j = pixel (row 2 * wave + j, column r): tap, tile and part are instruction immediates
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned abase = lds0 + (unsigned)(((g >> 1) * CQE + (2 * wave) * 2 * FX + r * 2 + (g & 1)) * 16);
    // output: the lane's 16-byte piece (part g & 1 of channel octet nt * 2 + (g >> 1)) of pixel (row 2 * wave + j, column r)
    int vob[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) vob[j][nt] = ((2 * wave + j) * a.Wo + r) * 64 + (g & 1) * 32 + (nt * 2 + (g >> 1)) * 8;

    // ---- the whole filter: 9 chunks x 2 output tiles x (hi, lo), resident for the whole walk ----
    short8 w[NCH][2][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                w[c][nt][0] = wp[((c * 2 + nt) * 2 + 0) * 64];
                w[c][nt][1] = wp[((c * 2 + nt) * 2 + 1) * 64];
            }
    }
    f32x4 bias4[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bias4[nt] = *reinterpret_cast<const f32x4 *>(a.bias + nt * 16 + g * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue slices, filter, bias (compiler-visible, so that no later wait is invented)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) asm volatile("" : "+v"(w[c][nt][0]), "+v"(w[c][nt][1]));   // (pinned: never re-loaded in front of an MFMA)
    asm volatile("s_barrier" ::: "memory");

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // operand fragments of one chunk: [operand tile][part]; two buffers in rotation over the chunk sequence, which runs on across slices (9 chunks
    // per slice: chunk c of a slice of parity PAR sits in buffer (PAR + c) & 1)
    short8 x[2][2][2];
    auto fetch = [](auto BUF, auto C, short8 (&xx)[2][2][2], const unsigned ad) __attribute__((always_inline)) {
        constexpr int b = decltype(BUF)::value, c = decltype(C)::value;
        constexpr int tapo = ((c / 3) * 2 * FX + (c % 3) * 2) * 16, row1 = 2 * FX * 16, pb = PARTE * 16;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][0]) : "v"(ad), "n"(tapo));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][1]) : "v"(ad), "n"(tapo + pb));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][0]) : "v"(ad), "n"(tapo + row1));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][1]) : "v"(ad), "n"(tapo + row1 + pb));
    };

    int sidxb = 0;                        // byte offset of the ring slot of the slice being contracted
    // One slice.  PAR: parity of the stream position (selects the fragment buffers); PRE: its chunk 0 was requested by the slice in front.
    auto step = [&](auto PAR_, auto PRE_, char *optr, const char *rptr, float *srow) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
        constexpr bool PRE = decltype(PRE_)::value;
        // the slice two ahead goes into the slot the previous step left
        static_for<PPW>([&](auto K) { issue_piece(K); });
        // residual pieces of this slice's outputs: requested behind chunk 1's MFMAs (at the top of the step they cost spills; later their latency shows: 16 registers live over most of the
        // contraction), waited for -- together with the DMA pieces in front of them -- before the epilogue
        u32x4 rq[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) rq[j][nt] = u32x4{0, 0, 0, 0};
        const unsigned cur0 = abase + (unsigned)sidxb;
        const unsigned nxtb = abase + (unsigned)(sidxb + SLOTB == RING * SLOTB ? 0 : sidxb + SLOTB);
        if constexpr (!PRE) fetch(std::integral_constant<int, PAR & 1>{}, std::integral_constant<int, 0>{}, x, cur0);
        f32x4 n[4];   // [operand tile][output tile]
#pragma unroll
        for (int u = 0; u < 4; ++u) n[u] = bias4[u & 1];
        static_for<NCH>([&](auto C) __attribute__((always_inline)) {
            constexpr int c = decltype(C)::value;
            constexpr int cur = (PAR + c) & 1, nxt = cur ^ 1;
            if constexpr (c + 1 < NCH) fetch(std::integral_constant<int, nxt>{}, std::integral_constant<int, (c + 1 < NCH ? c + 1 : 0)>{}, x, cur0);
            else fetch(std::integral_constant<int, nxt>{}, std::integral_constant<int, 0>{}, x, nxtb);   // the next slice's chunk 0: resident since the last barrier
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[cur][0][0]), "+v"(x[cur][0][1]), "+v"(x[cur][1][0]), "+v"(x[cur][1][1]));
            // product-major over the four accumulators: consecutive MFMAs never share one
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) n[j * 2 + nt] = mma<false>(w[c][nt][1], x[cur][j][0], n[j * 2 + nt]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) n[j * 2 + nt] = mma<false>(w[c][nt][0], x[cur][j][1], n[j * 2 + nt]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) n[j * 2 + nt] = mma<false>(w[c][nt][0], x[cur][j][0], n[j * 2 + nt]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RES && c == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        // (wave-uniform base in SGPRs + the lane's 32-bit byte offset: no 64-bit address per lane and piece)
                        const unsigned ro = (unsigned)(vob[j][nt] * 2);
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq[j][nt]) : "v"(ro), "s"(rptr) : "memory");
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // epilogue of the wave's four result tiles.  The queue is drained in front of its stores: the slice's DMA pieces (issued a whole contraction ago) and the
        // residual pieces have landed.  (A counted wait behind the stores -- "everything but this step's four stores" -- would assume that LDS-DMA loads and
        // stores retire in one order, which they do not, see conv_slice64_head below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the tie is a statement of its own BEHIND the wait and a scheduling barrier: as "+v" operands of the wait itself hipcc may copy the registers in front of
        // it -- conv_slice64_head below met that)
        if constexpr (RES) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(rq[0][0]), "+v"(rq[0][1]), "+v"(rq[1][0]), "+v"(rq[1][1]));
        }
        if constexpr (!SUMS) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 q4 = make_uint4(rq[j][nt][0], rq[j][nt][1], rq[j][nt][2], rq[j][nt][3]);
                    (void)epilogue_lean<P_BF16X3, RES, false>(reinterpret_cast<uint16_t *>(optr), nullptr, vob[j][nt], n[j * 2 + nt], q4, RELU, zero4);
                    __builtin_amdgcn_sched_barrier(0);   // (one tile's epilogue at a time: interleaved, the four need their temporaries at once)
                }
        } else {
            // lane (g, r) holds channels 4g .. 4g + 3 of pixel r of its row segment: sums over the 16 lanes of a row by DPP (quad xor 1, xor 2,
            // half-row mirror, row mirror); lanes r = 0 / r = 15 are the segment's first / last pixel
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float *rp = srow + ((int64_t)(2 * wave + j) * t.tiles_x * 3) * 32 + g * 4;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 v, rs;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = relu_bits(n[j * 2 + nt][i]);
                        float q = v[i];
                        q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                        q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                        q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x141, 0xF, 0xF, true));   // row_half_mirror
                        q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x140, 0xF, 0xF, true));   // row_mirror
                        rs[i] = q;
                    }
                    if (r == 0) {
                        *reinterpret_cast<f32x4 *>(rp + nt * 16) = rs;
                        *reinterpret_cast<f32x4 *>(rp + 32 + nt * 16) = v;
                    }
                    if (r == 15) *reinterpret_cast<f32x4 *>(rp + 64 + nt * 16) = v;
                }
            }
        }
        // the slice queued in this step has landed (drained in front of the epilogue); this step's stores stay in flight across the barrier.  All
        // waves are done reading this slice's slot: the step after the next one refills it.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            constexpr int nb = (PAR + NCH) & 1;
            asm volatile("" : "+v"(x[nb][0][0]), "+v"(x[nb][0][1]), "+v"(x[nb][1][0]), "+v"(x[nb][1][1]));
        }
        sidxb = (sidxb + SLOTB == RING * SLOTB) ? 0 : sidxb + SLOTB;
        advance_fill();
    };

    const int64_t ostride = (int64_t)a.Ho * a.Wo * 128;   // bytes per output slice (32 channels, hi + lo)
    const int64_t sstride = (int64_t)a.Ho * t.tiles_x * 3 * 32;   // row-sum floats per slice
    int par = 0;
    bool first = true;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t o0 = (((int64_t)U.b * a.No * a.Ho + U.gy0) * a.Wo + U.gx0) * 128;
        char *optr = SUMS ? nullptr : reinterpret_cast<char *>(a.out) + o0;
        const char *rp = RES ? reinterpret_cast<const char *>(a.res0) + o0 : nullptr;
        float *sp = SUMS ? a.outf + (((int64_t)U.b * a.No * a.Ho + U.gy0) * t.tiles_x + U.gx0 / TX) * 3 * 32 : nullptr;
        for (int z = 0; z < a.No; ++z) {
            using T = std::true_type;
            using F = std::false_type;
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            if (first) step(I0{}, F{}, optr, rp, sp);
            else if (par) step(I1{}, T{}, optr, rp, sp);
            else step(I0{}, T{}, optr, rp, sp);
            first = false;
            par ^= 1;
            if (!SUMS) optr += ostride;
            if (RES) rp += ostride;
            if (SUMS) sp += sstride;
        }
    }
    // the slices queued past the end of the stream are still in flight: a wave must not retire before its LDS-DMA has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- conv_slice64: the same per-slice convolution with 64 input and 64 output channels (End_to_End's level-3 alignment head, E2E.py:33-46:
// `optical_flow_aggregation.conv1.{2,4}.0` at quarter resolution) ------------------------------------------------------------------------------
// The 9 x 64 x 64 split-bf16 filter is 147 KB: a wave holds the share of ONE 16-channel output tile (18 chunks of (one tap x 32 channels) x (hi, lo)
// = 144 VGPRs) and contracts it against all of its pixels' K -- the split is over the outputs, so there are no partial sums to exchange.  Eight waves
// per workgroup walk the slices of an 8 x 16 column: wave (nt, rh) = output tile nt of rows 4 rh .. 4 rh + 3, two rows (operand tiles) at a time, so a
// slice is two passes of 18 chunks over the resident filter; an operand fragment feeds 3 MFMAs (conv_slice32: 6) -- every output tile's wave reads the
// image itself: ~0.4 of the LDS's read rate at the matrix pipe's sustained rate.  Slice image [group 4][part][row][pixel][octet] (45 KB), ring of 3 slots
// two slices ahead (135 KB: one workgroup per CU, two waves per SIMD), one barrier per slice; the fragments of the next chunk -- of the next pass, of
// the next slice's first chunk -- are requested in front of every chunk's MFMAs, as in conv_slice32.
// HEAD: the first conv of the same head over [warped features 32 | flow 2 | pad 6] records (`conv1.0.0#cur`, E2E.py:88-101; the volume flow_volume_kernel
// writes) plus the slice-broadcast reference part as a residual: 9 chunks of (tap x 32 feature channels) + 3 chunks over the record's fifth channel octet
// (K octet g of chunk 9 + k = tap 4k + g; taps 9-11 carry zero weights) = 12 chunks, 96 VGPRs of filter per wave; the slice image has a third channel
// group of which only octet 0 is filled; the residual pieces of a column (one slice per sample) are loaded once per column and stay in registers.
namespace slice64 {
constexpr int TY = DFFW_SLICE_TY, TX = DFFW_SLICE_TX, FY = TY + 2, FX = TX + 2, NW = 8, RING = 3;
static_assert(TY == 8 && TX == 16, "wave (nt, rh): rows 4 rh .. 4 rh + 3");
constexpr int PARTE = FY * FX * 2;        // entries of one part of a 16-channel group: [row][pixel][octet]
constexpr int CQE = 2 * PARTE;            // ... of a group: [part][row][pixel][octet]
// MODE 0: 64 -> 64; 1: HEAD; 2: CAT -- a 32 -> 32 conv over t with the block's 1x1x1 shortcut over a second 32-channel tensor x folded in as a tenth chunk
// (`OF_feature2.1.conv.2`, End_to_End.py:135-145: out = relu(conv.2(t) + feature(x))): the image holds [t | x] as the four groups of a 64-channel virtual concat (x's
// block starts on a DMA-piece boundary: a piece has one source), chunks 0-8 = the taps over t, chunk 9 = the centre tap over x; 2 output tiles: wave (nt, rg) = rows
// 2 rg, 2 rg + 1, one pass per slice.
template <int MODE>
struct Lay {
    static constexpr bool HEAD = MODE == 1, CAT = MODE == 2;
    static constexpr int NG = HEAD ? 3 : 4;
    static constexpr int NCH = HEAD ? SLICE64_HEAD_CHUNKS : CAT ? SLICE32_CAT_CHUNKS : SLICE64_CHUNKS;
    static constexpr int SRC1E = CAT ? (2 * CQE + 63) / 64 * 64 : 2 * CQE;   // entry at which groups 2, 3 start
    static constexpr int SLOTE = SRC1E + (NG - 2) * CQE;
    static constexpr int PPW = ((SLOTE + 63) / 64 + NW - 1) / NW;
    static constexpr int NPIECE = PPW * NW;   // every wave issues PPW pieces per slice (the counted vmcnt waits rely on it): the slot is padded to whole rounds
    static constexpr int SLOTB = NPIECE * 1024;
    static constexpr int LDSB = RING * SLOTB;
    static constexpr int RECB = HEAD ? 160 : CAT ? 128 : 256, PARTB = RECB / 2, NOCT = RECB / 32;   // a pixel record [hi C][lo C] of a source
    static constexpr int NTW = CAT ? 2 : 4, PASSES = CAT ? 1 : 2, COUT = NTW * 16;   // output tiles; passes of two rows per wave and slice
    static_assert(NCH % 2 == 0, "an even chunk count keeps the fragment buffers' parity over passes and slices");
    static_assert(CQE % 16 == 0 && SLOTB % 256 == 0 && LDSB <= 160 * 1024, "LDS layout (one workgroup per CU)");
};
}   // namespace slice64

// (the body is a device function template and the kernels thin wrappers: with the layout's dependent constants directly inside a __global__ template, hipcc's host
// pass silently emits no launch stub)
template <bool RELU, bool SUMS, int MODE>
__device__ __forceinline__ void slice64_body(const ConvArgs &a, const RollArgs &t, unsigned char *smem) {
    constexpr bool HEAD = MODE == 1, CAT = MODE == 2;
    static_assert(!SUMS || (RELU && MODE == 0), "row sums: relu(acc), no residual");
    using namespace slice64;
    using L = Lay<MODE>;
    constexpr int NCH = L::NCH, PPW = L::PPW, SLOTB = L::SLOTB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int nt = wave & (L::NTW - 1), rh = wave / L::NTW;   // (CAT: rh = row pair 0..3)

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
    auto decode = [&](int u) {
        Unit c;
        const int txi = u % t.tiles_x;
        const int tt = u / t.tiles_x;
        c.gx0 = txi * TX;
        c.gy0 = (tt % t.tiles_y) * TY;
        c.b = tt / t.tiles_y;
        return c;
    };

    // ---- fill (as conv_slice32) ----
    constexpr int recb = L::RECB, partb = L::PARTB;
    const char *tb = reinterpret_cast<const char *>(a.in0);
    const char *tb1 = CAT ? reinterpret_cast<const char *>(a.in1) : tb;
    const int slice_bytes = a.Hi * a.Wi * recb;
    int fvo[PPW];
    const char *fb = tb, *fb1 = tb1;
    int fu = ufirst, fz = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fz = 0;
        fb = tb + ((int64_t)c.b * a.Ni * a.Hi * a.Wi + (int64_t)(c.gy0 - 1) * a.Wi + (c.gx0 - 1)) * recb;
        fb1 = tb1 + ((int64_t)c.b * a.Ni * a.Hi * a.Wi + (int64_t)(c.gy0 - 1) * a.Wi + (c.gx0 - 1)) * recb;
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // (opaque: no hoisting of the decode out of the unit loop)
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int e = (k * NW + wave) * 64 + ln;       // entry inside the slot: [group][part][row][pixel][octet] (CAT: groups 2, 3 from entry SRC1E)
            const bool s1 = CAT && e >= L::SRC1E;
            const int es = s1 ? e - L::SRC1E : e;
            const int cqs = es / CQE, e2 = es - cqs * CQE; // group inside its source
            const int cq = cqs + (s1 ? 0 : 0);
            const int part = e2 / PARTE, e3 = e2 - part * PARTE;
            const int fy = e3 / (2 * FX), e4 = e3 - fy * (2 * FX);
            const int fx = e4 >> 1, oct = e4 & 1;
            const int iy = c.gy0 - 1 + fy, ix = c.gx0 - 1 + fx;
            fvo[k] = (e < L::SLOTE && (!CAT || s1 || e < 2 * CQE) && cq * 2 + oct < L::NOCT && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                         ? (fy * a.Wi + fx) * recb + part * partb + (cq * 2 + oct) * 16
                         : (int)0x80000000;
        }
    };
    setup_fill();
    int fslotb = 0;
    auto issue_piece = [&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const int p = k * NW + wave;
        const bool zin = fu < uend;                        // past the end of the stream: zeros (the slot is never read)
        const char *fbp = (CAT && p * 64 >= L::SRC1E) ? fb1 : fb;   // (wave-uniform: a piece has one source)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fbp), 0, zin ? (int)0x80000000 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + fslotb + p * 1024), 16, fvo[k], zin ? fz * slice_bytes : 0, 0, 0);
    };
    auto advance_fill = [&]() {
        fslotb = (fslotb + SLOTB == RING * SLOTB) ? 0 : fslotb + SLOTB;
        if (++fz == a.Ni && fu < uend) {
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {                          // the fill runs two slices ahead
        static_for<PPW>([&](auto K) { issue_piece(K); });
        advance_fill();
    }

    // ---- operand addressing.  64 -> 64: chunk c = (tap c / 2, channel half c % 2), K octet g = channels 32 (c % 2) + 8g .. = (group 2 (c % 2) + (g >> 1), octet g & 1)
    // of the tap.  HEAD: chunk c < 9 = tap c over the 32 feature channels (groups 0, 1); chunk 9 + k: K octet g = octet 0 of group 2 at tap 4k + g (taps >= 9: zero
    // weights, tap 8's operands).  Lane r of operand tile j of pass p = pixel (row 4 rh + 2 p + j, column r): tap, half, pass, tile and part are immediates.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    constexpr int RPW = 2 * L::PASSES;   // rows per wave
    const unsigned abase = lds0 + (unsigned)(((g >> 1) * CQE + (RPW * rh) * 2 * FX + r * 2 + (g & 1)) * 16);
    unsigned afl[3] = {0, 0, 0};
    if constexpr (HEAD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int tap = 4 * k + g < 9 ? 4 * k + g : 8;
            afl[k] = lds0 + (unsigned)((2 * CQE + (4 * rh + tap / 3) * 2 * FX + (r + tap % 3) * 2) * 16);
        }
    }
    // output: the lane's 16-byte piece (part g & 1 of channel octet nt * 2 + (g >> 1)) of pixel (row 4 rh + 2 p + j, column r)
    int vob[2][2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int j = 0; j < 2; ++j) vob[ps][j] = ((RPW * rh + 2 * ps + j) * a.Wo + r) * (2 * L::COUT) + (g & 1) * L::COUT + (nt * 2 + (g >> 1)) * 8;

    // ---- this output tile's filter: NCH chunks x (hi, lo), resident for the whole walk ----
    short8 w[NCH][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + (size_t)nt * NCH * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[c][0] = wp[(c * 2 + 0) * 64];
            w[c][1] = wp[(c * 2 + 1) * 64];
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + nt * 16 + g * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue slices, filter, bias
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(w[c][0]), "+v"(w[c][1]));   // (pinned: never re-loaded in front of an MFMA)
    asm volatile("s_barrier" ::: "memory");

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // operand fragments of one chunk: [operand tile][part]; chunk c of any pass sits in buffer c & 1 (an even number of chunks per pass)
    short8 x[2][2][2];
    auto fetch = [](auto BUF, auto C, auto PS, short8 (&xx)[2][2][2], const unsigned ad, const unsigned (&af)[3]) __attribute__((always_inline)) {
        constexpr int b = decltype(BUF)::value, c = decltype(C)::value, ps = decltype(PS)::value;
        constexpr int row1 = 2 * FX * 16, pb = PARTE * 16;
        if constexpr (HEAD && c >= 9) {
            constexpr int off = (2 * ps) * 2 * FX * 16;
            const unsigned adf = af[c - 9 < 3 ? c - 9 : 0];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][0]) : "v"(adf), "n"(off));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][1]) : "v"(adf), "n"(off + pb));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][0]) : "v"(adf), "n"(off + row1));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][1]) : "v"(adf), "n"(off + row1 + pb));
        } else {
            constexpr int tap = HEAD ? c : CAT ? (c < 9 ? c : 4) : c / 2, hf = HEAD ? 0 : CAT ? (c < 9 ? 0 : 1) : c % 2;
            constexpr int tapo = ((tap / 3 + 2 * ps) * 2 * FX + (tap % 3) * 2 + hf * L::SRC1E) * 16;
            static_assert(tapo + row1 + pb < 65536, "ds_read immediate");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][0]) : "v"(ad), "n"(tapo));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][1]) : "v"(ad), "n"(tapo + pb));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][0]) : "v"(ad), "n"(tapo + row1));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][1]) : "v"(ad), "n"(tapo + row1 + pb));
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // HEAD: the column's residual pieces (the sample's one reference slice), [pass][operand tile]; requested in front of the column's first slice and waited
    // for (vmcnt(0)) in front of its first epilogue
    u32x4 rq[2][2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int j = 0; j < 2; ++j) rq[ps][j] = u32x4{0, 0, 0, 0};

    int sidxb = 0;                        // byte offset of the ring slot of the slice being contracted
    // One pass = two rows of the wave's four.  PRE: its chunk 0 was requested by the pass in front (all but the kernel's first).
    auto pass = [&](auto PS_, auto PRE_, const unsigned cur0, const unsigned nxtb, const unsigned slotd, char *optr, float *srow, const bool fresh) __attribute__((always_inline)) {
        constexpr int ps = decltype(PS_)::value;
        constexpr bool PRE = decltype(PRE_)::value;
        unsigned afc[3], afn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            afc[k] = afl[k] + (unsigned)sidxb;
            afn[k] = afl[k] + slotd;
        }
        if constexpr (!PRE) fetch(I0{}, I0{}, std::integral_constant<int, ps>{}, x, cur0, afc);
        f32x4 n[2] = {bias4, bias4};   // [operand tile]
        static_for<NCH>([&](auto C) __attribute__((always_inline)) {
            constexpr int c = decltype(C)::value;
            constexpr int cur = c & 1, nxt = cur ^ 1;
            if constexpr (c + 1 < NCH) fetch(std::integral_constant<int, nxt>{}, std::integral_constant<int, (c + 1 < NCH ? c + 1 : 0)>{}, std::integral_constant<int, ps>{}, x, cur0, afc);
            else if constexpr (ps + 1 < L::PASSES) fetch(std::integral_constant<int, nxt>{}, I0{}, I1{}, x, cur0, afc);   // the second pass's chunk 0
            else fetch(std::integral_constant<int, nxt>{}, I0{}, I0{}, x, nxtb, afn);                          // the next slice's: resident since the last barrier
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[cur][0][0]), "+v"(x[cur][0][1]), "+v"(x[cur][1][0]), "+v"(x[cur][1][1]));
            // product-major over the two accumulators
            n[0] = mma<false>(w[c][1], x[cur][0][0], n[0]);
            n[1] = mma<false>(w[c][1], x[cur][1][0], n[1]);
            n[0] = mma<false>(w[c][0], x[cur][0][1], n[0]);
            n[1] = mma<false>(w[c][0], x[cur][1][1], n[1]);
            n[0] = mma<false>(w[c][0], x[cur][0][0], n[0]);
            n[1] = mma<false>(w[c][0], x[cur][1][0], n[1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        // Every slice's first pass drains the queue in front of its stores: the slice's DMA pieces were issued a pass ago, the previous slice's stores long before
        // (round 6, ADVICE r05: the end-of-step wait used to be a counted one behind the stores).  HEAD: the column's residual pieces were requested in front of its
        // first slice's DMA pieces; a counted wait (vmcnt(PPW): "everything but this slice's pieces") is NOT enough for them -- measured: wrong results that vary from
        // run to run once a workgroup walks more than one column; VGPR loads, LDS-DMA loads and the previous step's stores do not retire in one order.  (The wait
        // comes first and names the registers: the epilogue's arithmetic on them must not be scheduled in front of it.)
        if constexpr (ps == 0) {
#if defined(DFFW_SLICE_HAZARD) && DFFW_SLICE_HAZARD == 1   // development only: round 5's first form -- a COUNTED wait for the residual pieces ("everything but this slice's DMA pieces")
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            // The tie is a statement of its own BEHIND the wait, with a scheduling barrier between them.  The residual registers stay live for the column's ten slices
            // and the epilogue's lane swaps are destructive, so hipcc gives the tie's results other registers than its operands -- i.e. copies rq in front of the tie.
            // As "+v" operands of the wait itself, or without the barrier, (half of) these copies were scheduled in front of the wait: a read of a load that has not
            // landed -- measured: wrong by 2e-3, varying from run to run (profiles/r06_wait_tie_hazard.txt).
            if constexpr (HEAD) {
#if !(defined(DFFW_SLICE_HAZARD) && DFFW_SLICE_HAZARD == 2)   // development only, 2: the tie straight behind the wait, no scheduling barrier (the copies move in front of the wait)
                __builtin_amdgcn_sched_barrier(0);
#endif
                asm volatile("" : "+v"(rq[0][0]), "+v"(rq[0][1]), "+v"(rq[1][0]), "+v"(rq[1][1]));
            }
        }
        if constexpr (!SUMS) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 q4 = make_uint4(rq[ps][j][0], rq[ps][j][1], rq[ps][j][2], rq[ps][j][3]);
                (void)epilogue_lean<P_BF16X3, HEAD, false>(reinterpret_cast<uint16_t *>(optr), nullptr, vob[ps][j], n[j], q4, RELU, zero4);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // lane (g, r) holds channels 16 nt + 4g .. + 3 of pixel r of its row segment: sums over the 16 lanes of a row by DPP; lanes r = 0 / r = 15 are
            // the segment's first / last pixel (conv_tile's row-sums layout: [(row * tiles_x + tile column) * 3 + {sum, first, last}][64])
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float *rp = srow + ((int64_t)(4 * rh + 2 * ps + j) * t.tiles_x * 3) * 64 + nt * 16 + g * 4;
                f32x4 v, rs;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = relu_bits(n[j][i]);
                    float q = v[i];
                    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x141, 0xF, 0xF, true));   // row_half_mirror
                    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x140, 0xF, 0xF, true));   // row_mirror
                    rs[i] = q;
                }
                if (r == 0) {
                    *reinterpret_cast<f32x4 *>(rp) = rs;
                    *reinterpret_cast<f32x4 *>(rp + 64) = v;
                }
                if (r == 15) *reinterpret_cast<f32x4 *>(rp + 128) = v;
            }
        }
    };
    auto step = [&](auto PRE_, char *optr, float *srow, const bool fresh) __attribute__((always_inline)) {
        static_for<PPW>([&](auto K) { issue_piece(K); });   // the slice two ahead goes into the slot the previous step left
        const unsigned slotd = (unsigned)(sidxb + SLOTB == RING * SLOTB ? 0 : sidxb + SLOTB);
        const unsigned cur0 = abase + (unsigned)sidxb;
        const unsigned nxtb = abase + slotd;
        pass(I0{}, PRE_, cur0, nxtb, slotd, optr, srow, fresh);
        if constexpr (L::PASSES == 2) pass(I1{}, std::true_type{}, cur0, nxtb, slotd, optr, srow, false);
        // the slice queued in this step has landed (drained in front of the first pass's epilogue); this step's stores may stay in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("" : "+v"(x[0][0][0]), "+v"(x[0][0][1]), "+v"(x[0][1][0]), "+v"(x[0][1][1]));
        sidxb = (sidxb + SLOTB == RING * SLOTB) ? 0 : sidxb + SLOTB;
        advance_fill();
    };

    const int64_t ostride = (int64_t)a.Ho * a.Wo * 4 * L::COUT;   // bytes per output slice (hi + lo)
    const int64_t sstride = (int64_t)a.Ho * t.tiles_x * 3 * 64;   // row-sum floats per slice
    bool first = true;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t o0 = (((int64_t)U.b * a.No * a.Ho + U.gy0) * a.Wo + U.gx0) * 4 * L::COUT;
        char *optr = SUMS ? nullptr : reinterpret_cast<char *>(a.out) + o0;
        float *sp = SUMS ? a.outf + (((int64_t)U.b * a.No * a.Ho + U.gy0) * t.tiles_x + U.gx0 / TX) * 3 * 64 : nullptr;
        if constexpr (HEAD) {
            const char *rb = reinterpret_cast<const char *>(a.res0) + (((int64_t)U.b * a.Ho + U.gy0) * a.Wo + U.gx0) * 256;   // one reference slice per sample
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned ro = (unsigned)(vob[ps][j] * 2);
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq[ps][j]) : "v"(ro), "s"(rb) : "memory");
                }
        }
        for (int z = 0; z < a.No; ++z) {
            if (first) step(std::false_type{}, optr, sp, z == 0);
            else step(std::true_type{}, optr, sp, z == 0);
            first = false;
            if (!SUMS) optr += ostride;
            if (SUMS) sp += sstride;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave
}

template <bool RELU, bool SUMS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_slice64(const ConvArgs a, const RollArgs t) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[slice64::Lay<0>::LDSB];
    slice64_body<RELU, SUMS, 0>(a, t, smem);
}
template <bool RELU>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_slice64_head(const ConvArgs a, const RollArgs t) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[slice64::Lay<1>::LDSB];
    slice64_body<RELU, false, 1>(a, t, smem);
}
template <bool RELU>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_slice32_cat(const ConvArgs a, const RollArgs t) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[slice64::Lay<2>::LDSB];
    slice64_body<RELU, false, 2>(a, t, smem);
}

void slice32_tile(int *ty, int *tx) {
    *ty = slice32::TY;
    *tx = slice32::TX;
}

bool slice32_ok(int prec, const ConvArgs &a) {
    if (prec != P_BF16X3 || (a.dbg & DFFW_ARGS_NO_SLICE32)) return false;
    if (a.dbg & DFFW_ARGS_SUMS) {   // row-sums variant: nothing stored, a.outf receives the row vectors
        if (!a.outf || a.relu != 1 || a.res0) return false;
    } else if (!a.out || a.outf) return false;
    if (a.out_pre || a.res1 || a.res_bcast || a.cls_w || a.relu == 2 || a.Cout != 32 || a.C0 != 32 || a.C1 != 0) return false;
    if (a.Ho % slice32::TY || a.Wo % slice32::TX || a.Ho != a.Hi || a.Wo != a.Wi || a.No != a.Ni) return false;
    // 32-bit buffer offsets: a sample's input volume (+ one footprint) stays below 2^31 bytes
    return (int64_t)(a.Ni + 1) * a.Hi * a.Wi * 128 < (1ll << 31);
}

hipError_t launch_conv_slice32(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 512;   // two 4-wave workgroups per CU
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(slice32::NW * 64);
    const bool relu = a.relu == 1, res = a.res0 != nullptr;
    if (a.dbg & DFFW_ARGS_SUMS) hipLaunchKernelGGL((conv_slice32<true, false, true>), grid, block, 0, s, a, t);
    else if (relu && res) hipLaunchKernelGGL((conv_slice32<true, true>), grid, block, 0, s, a, t);
    else if (relu) hipLaunchKernelGGL((conv_slice32<true, false>), grid, block, 0, s, a, t);
    else if (res) hipLaunchKernelGGL((conv_slice32<false, true>), grid, block, 0, s, a, t);
    else hipLaunchKernelGGL((conv_slice32<false, false>), grid, block, 0, s, a, t);
    return hipGetLastError();
}

void conv_slice32_kernel_name(const ConvArgs &a, char *buf, int n) {
    if (a.dbg & DFFW_ARGS_SUMS) snprintf(buf, n, "dffw::conv_slice32<true, false, true>");
    else snprintf(buf, n, "dffw::conv_slice32<%s, %s, false>", a.relu == 1 ? "true" : "false", a.res0 ? "true" : "false");   // (rocprofv3's spelling)
}

bool slice64_ok(int prec, const ConvArgs &a) {
    if (prec != P_BF16X3 || (a.dbg & DFFW_ARGS_NO_SLICE32)) return false;
    if (a.dbg & DFFW_ARGS_SUMS) {   // row-sums variant: nothing stored, a.outf receives the row vectors
        if (!a.outf || a.relu != 1) return false;
    } else if (!a.out || a.outf) return false;
    const bool head = a.res_bcast && a.res0 && a.C0 == 40 && !(a.dbg & DFFW_ARGS_SUMS);   // the level-3 head's first conv over [features 32 | flow 2 | pad 6]
    const bool cat = a.C0 == 32 && a.C1 == 32 && a.in1 && a.Cout == 32 && !a.res0 && !a.res_bcast && !(a.dbg & DFFW_ARGS_SUMS);   // 32 -> 32 with the folded shortcut over a second tensor
    if ((a.res0 || a.res_bcast) && !head) return false;
    if (a.out_pre || a.res1 || a.cls_w || a.relu == 2 || (a.Cout != 64 && !cat) || (a.C0 != 64 && !head && !cat) || (a.C1 != 0 && !cat)) return false;
    if (a.Ho % slice64::TY || a.Wo % slice64::TX || a.Ho != a.Hi || a.Wo != a.Wi || a.No != a.Ni) return false;
    // 32-bit buffer offsets: a sample's input volume (+ one footprint) stays below 2^31 bytes
    return (int64_t)(a.Ni + 1) * a.Hi * a.Wi * 256 < (1ll << 31) && (int64_t)a.Ho * a.Wo * 256 < (1ll << 31);
}

hipError_t launch_conv_slice64(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 256;   // one 8-wave workgroup per CU
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(slice64::NW * 64);
    if (a.C1 == 32) {
        if (a.relu == 1) hipLaunchKernelGGL((conv_slice32_cat<true>), grid, block, 0, s, a, t);
        else hipLaunchKernelGGL((conv_slice32_cat<false>), grid, block, 0, s, a, t);
    } else if (a.res_bcast) {
        if (a.relu == 1) hipLaunchKernelGGL((conv_slice64_head<true>), grid, block, 0, s, a, t);
        else hipLaunchKernelGGL((conv_slice64_head<false>), grid, block, 0, s, a, t);
    } else if (a.dbg & DFFW_ARGS_SUMS) hipLaunchKernelGGL((conv_slice64<true, true>), grid, block, 0, s, a, t);
    else if (a.relu == 1) hipLaunchKernelGGL((conv_slice64<true, false>), grid, block, 0, s, a, t);
    else hipLaunchKernelGGL((conv_slice64<false, false>), grid, block, 0, s, a, t);
    return hipGetLastError();
}

void conv_slice64_kernel_name(const ConvArgs &a, char *buf, int n) {
    if (a.C1 == 32) snprintf(buf, n, "dffw::conv_slice32_cat<%s>", a.relu == 1 ? "true" : "false");
    else if (a.res_bcast) snprintf(buf, n, "dffw::conv_slice64_head<%s>", a.relu == 1 ? "true" : "false");
    else snprintf(buf, n, "dffw::conv_slice64<%s, %s>", (a.dbg & DFFW_ARGS_SUMS) || a.relu == 1 ? "true" : "false", (a.dbg & DFFW_ARGS_SUMS) ? "true" : "false");
}

}  // namespace dffw
