// conv_rollk: rolling-window 3x3x3 stride-1 convolution for the 32- and 64-input-channel layers with 32 output channels per launch
// (round 5), gfx950 / MI355X.  `dres2.conv0` (64 -> 32 over a virtual concat), `dres3.conv2`, `dres3.conv4`, the pyramid's `dres8_*`,
// `confidence.0` (32 -> 32), and as two launches over the output-channel halves `dres0.0` (32 -> 64), `dres0.2`, `dres2.conv2` (64 -> 64):
// DEN.py:33-40, 155-192, 243-258.
//
// conv_tile stages a 5 x 4 x 16 block's footprint (2.36x its outputs) per 16-channel stage, streams the filter from L2 once per wave
// and chunk, and runs fill -> barrier -> contract -> store as a serial chain whose phases add up (profiles/r04_ablation_conv_tile_phases.txt).
// Here, as in conv_rollx_k2, a persistent workgroup walks the focus slices of 8 x 8 columns with a ring of LDS slice images filled by
// LDS-DMA one slice ahead (in-plane halo 1.56x, none along the slices) and keeps the WHOLE filter in registers -- which for 64 x 27 x 32
// split-bf16 weights (221 KB) takes all eight waves of a CU: the contraction is split over K.
//   * wave (cq, th) owns input channels 16*cq .. +15 and the filter taps 14*th .. 14*th + 13 (27 taps + one zero): 7 chunks of (2 taps x
//     16 channels) x 2 output tiles x (hi, lo) = 112 VGPRs of A fragments.  Every wave contracts ALL 64 pixels of the column's slice against
//     its K share; a B fragment (one ds_read_b128 per part) feeds 6 MFMAs.
//   * a step (one output slice) is two half-steps of 32 pixels (2 operand tiles x 2 output tiles = 4 units of 16 x 16 fp32 per wave).  After a
//     half-step every wave hands its partial of the units it does not own to their owners through LDS (one ds_write_b128 per unit) and
//     the owners -- waves 0-3 for the first half-step, 4-7 for the second (Cin = 32: four waves, each owns one unit of either half) -- add
//     the NW - 1 partials in wave order INSIDE the next half-step's contraction and run the epilogue there.  The exchange area is double
//     buffered by half-step parity, so one barrier per half-step orders writes after the previous content's reads by construction.
//   * LDS slice image: [source][row][16-channel group][part][pixel][octet] in 16-byte entries (a source = one tensor of a virtual concat,
//     or one channel half of a single tensor: a DMA piece has one buffer descriptor; the one piece per slice that straddles the two
//     sources is issued twice under complementary lane masks).  An operand tile is rows j and j + 4 of the column: their distance is a
//     multiple of 256 bytes and lane rows g, g + 1 take the two octets of the same tap, so the 16 lanes of every ds_read_b128 service
//     group fall on 16 distinct 16-byte bank groups with no padding.
//   * ring of 4 slots: three being read, the fourth filled during the step and waited for (vmcnt(0)) at the step's last barrier.
// Epilogue: out = [relu](acc + BatchNorm shift [+ residual]) in split-bf16 storage (epilogue_lean).  Sums are re-associated with respect to
// conv_tile (K split over waves): results agree to ~1e-6 relative, not bitwise.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "dffw_conv_roll.h"
#include "dffw_device.h"

namespace dffw {

#ifndef DFFW_ROLLK_SKEW
#define DFFW_ROLLK_SKEW 0
#endif
namespace rollk {
constexpr int TY = DFFW_ROLLK_TY, TX = DFFW_ROLLK_TX, FY = TY + 2, FX = TX + 2, RING = 4, NCH = ROLLK_CHUNKS, NTAPH = 2 * NCH;
static_assert(TY == 8 && TX == 8, "operand tiles are rows (j, j + 4) of an 8 x 8 column");
template <int NW>
struct Lay {
    static constexpr int NCQ = NW / 2;           // 16-channel groups of the input
    static constexpr int CQS = NCQ / 2;          // ... per source
    static constexpr int PARTE = 2 * FX;         // entries of one part of a group's row: [pixel][octet]
    static constexpr int CQE = 2 * PARTE;        // ... of a group's row: [part][pixel][octet]
    static constexpr int ROWE = CQS * CQE;       // row pitch inside a source block
    static constexpr int SRCE = FY * ROWE;       // source block
    static constexpr int SLOTE = 2 * SRCE;
    static constexpr int NPIECE = (SLOTE + 63) / 64;
    static constexpr int SLOTB = NPIECE * 1024;
    static constexpr int PPW = (NPIECE + NW - 1) / NW;    // DMA pieces per wave and slice (the last one exists for the first waves only)
    static constexpr int XCH_OFF = RING * SLOTB;
    static constexpr int XCHB = 4 * (NW - 1) * 1024;      // one half-step's exchange: 4 units x (NW - 1) partials of 1 KiB
    static constexpr int LDSB = XCH_OFF + 2 * XCHB;
    static_assert((4 * ROWE) % 16 == 0, "an operand tile's two rows must be a multiple of 256 bytes apart");
    static_assert(SLOTB % 256 == 0 && LDSB <= 160 * 1024, "LDS layout");
};
// tap slot sl of a wave's half = filter tap 14 * TH + sl in [slice][row][column] order; slot 27 does not exist (zero weights): it reads tap 26's operands
template <int NW, int TH>
struct TapT {
    using L = Lay<NW>;
    static constexpr int tap(int sl) { return TH * NTAPH + sl > 26 ? 26 : TH * NTAPH + sl; }
    static constexpr int dz(int sl) { return tap(sl) / 9; }
    static constexpr int off(int sl) { return (((tap(sl) % 9) / 3) * L::ROWE + (tap(sl) % 3) * 2) * 16; }   // byte offset inside the slice image
};
// orders every later use of p[] behind the (volatile) asm statements before this point: the counted waits that cover their ds_reads
template <int N>
__device__ __forceinline__ void tie(f32x4 (&p)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(p[i]));
}
}   // namespace rollk

// OWN0 / OWN1: this wave owns a unit of the first / second half-step (NW = 8: waves 0-3 / 4-7; NW = 4: every wave owns one of each)
// ABL (development only, `make ABL=1` + DFFW_ROLLK_ABL): timing ablations -- 1 no operand reads, 2 no MFMAs, 4 no fill, 8 no exchange, 16 no
// half-step barrier, 32 no epilogue / store (results are wrong with any bit set)
template <int NW, bool RELU, bool RES, bool OWN0, bool OWN1, int TH, int ABL, int SKEW>
__device__ __forceinline__ void rollk_body(const ConvArgs &a, const RollArgs &t, unsigned char *smem, const int lane, const int wave) {
    using namespace rollk;
    using L = Lay<NW>;
    const int g = lane >> 4, r = lane & 15;
    const int cq = wave >> 1;                      // (the tap half TH = wave & 1 is a template parameter: every tap offset is an immediate)
    const int src = cq / L::CQS, cql = cq % L::CQS;
    const int myu = wave & 3;                      // the unit of a half-step this wave owns (if it owns one): operand tile myu >> 1, output tile myu & 1

    // ---- this workgroup's units (8 x 8 columns of one sample / slice range): XCD x owns a contiguous range, as conv_roll -------------
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

    // ---- fill.  Sources: the two tensors of a virtual concat, or the two channel halves of one tensor (records [hi C][lo C]) --------
    const bool two = a.C1 != 0;
    const int partb = a.C0 * 2;                        // bytes of one part of a pixel record of a source tensor
    const int recb = 2 * partb;
    const char *tb0 = reinterpret_cast<const char *>(a.in0);
    const char *tb1 = two ? reinterpret_cast<const char *>(a.in1) : reinterpret_cast<const char *>(a.in0) + a.C0;   // second half: C0 / 2 channels on
    const int slice_bytes = a.Hi * a.Wi * recb;
    // per lane and piece: the byte offset of its 16-byte entry from the unit's footprint origin (out-of-image and padding lanes pushed out of
    // range), recomputed per unit (every ~12 steps) rather than kept in 8 more registers
    int fvo[L::PPW];
    const char *fb0 = tb0, *fb1 = tb1;                 // descriptor bases of the unit being fetched: its footprint origin in slice 0, per source
    int fu = ufirst, fq = 0, fslices = 0, fz = 0;      // fill cursor: unit, slice inside it, its slice count, input slice index
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        // the unit's input slices: its output range and one halo slice on either side where the volume has one (no zero slices in the ring:
        // the windows at the volume's ends skip the taps that would read them)
        fz = c.zbeg > 0 ? c.zbeg - 1 : 0;
        fslices = (c.zbeg + c.nz < a.Ni ? c.zbeg + c.nz : a.Ni - 1) - fz + 1;
        const int64_t o = ((int64_t)c.b * a.Ni * a.Hi * a.Wi + (int64_t)(c.gy0 - 1) * a.Wi + (c.gx0 - 1)) * recb;
        fb0 = tb0 + o;
        fb1 = tb1 + o;
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // (opaque: hipcc would hoist the decode below out of the unit loop and keep its results in 8 registers)
#pragma unroll
        for (int k = 0; k < L::PPW; ++k) {
            const int e = (k * NW + wave) * 64 + ln;       // 16-byte entry inside the slot: [source][row][group][part][pixel][octet]
            const int s = e / L::SRCE, e2 = e - s * L::SRCE;
            const int fy = e2 / L::ROWE, e3 = e2 - fy * L::ROWE;
            const int gq = e3 / L::CQE, e4 = e3 - gq * L::CQE;
            const int part = e4 / L::PARTE, e5 = e4 - part * L::PARTE;
            const int fx = e5 >> 1, oct = e5 & 1;
            const int iy = c.gy0 - 1 + fy, ix = c.gx0 - 1 + fx;
            fvo[k] = (e < L::SLOTE && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) ? (fy * a.Wi + fx) * recb + part * partb + (gq * 2 + oct) * 16
                                                                                                    : (int)0x80000000;
        }
    };
    setup_fill();
    int fslotb = 0;                                    // byte offset of the ring slot the next slice goes to
    auto issue_piece = [&](auto K) {
        constexpr int k = decltype(K)::value;
        const int p = k * NW + wave;
        if (p >= L::NPIECE) return;                    // (wave-uniform)
        const bool zin = fu < uend;                    // past the end of the stream: zeros (the slot is never read)
        const int nrec = zin ? (int)0x80000000 : 0, so = zin ? fz * slice_bytes : 0;
        auto dst = (__attribute__((address_space(3))) void *)(smem + fslotb + p * 1024);
        // One descriptor per piece, chosen with SCALAR selects (written as a lane-dependent choice hipcc turns every piece into a waterfall
        // loop over the two descriptors).  The one piece per slice that straddles the two sources -- entry SRCE is not a multiple of 64 --
        // is issued once per source under complementary lane masks (exec), the other source's lanes sitting out.
        constexpr int PSTR = (L::SRCE % 64) ? L::SRCE / 64 : -1;
        const bool second = p * 64 >= L::SRCE;         // (wave-uniform)
        const char *fb = second ? fb1 : fb0;
        fb = reinterpret_cast<const char *>(((uint64_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)fb >> 32)) << 32) |
                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)fb));
        if (p != PSTR) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb), 0, nrec, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, fvo[k], so, 0, 0);
        } else {
            const bool lo = p * 64 + lane < L::SRCE;
            if (lo) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb0), 0, nrec, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, fvo[k], so, 0, 0);
            }
            asm volatile("" ::: "memory");             // (keeps the two halves two instructions)
            if (!lo) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb1), 0, nrec, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, fvo[k], so, 0, 0);
            }
        }
    };
    auto advance_fill = [&]() {
        fslotb = (fslotb + L::SLOTB == RING * L::SLOTB) ? 0 : fslotb + L::SLOTB;
        ++fz;
        if (++fq == fslices && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {                   // the fill runs two slices ahead of the window's centre
        static_for<L::PPW>([&](auto K) { issue_piece(K); });
        advance_fill();
    }

    // ---- operand addressing.  Tap slot s of this wave's half = filter tap 14*th + s (tap 27 does not exist: zero weights, reads tap 26's
    // operands); chunk c = slots 2c, 2c + 1; K octet g = (slot 2c + (g >> 1), channel octet g & 1).  Lane r of operand tile j = pixel
    // (row j + 4*(r >> 3), column r & 7): the tile index and the part are instruction immediates.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // per lane: the entry of its pixel / channel octet inside a slice image at tap (0, 0).  The two tap slots of a chunk (lanes g < 2 / g >= 2)
    // differ by a compile-time byte distance -- plus, for the one chunk that straddles two window slices (TH = 0: taps 8 | 9), the distance of their
    // ring slots -- which enters under the lane mask `gmask`; the first slot's own offset is the ds_read immediate.
    const unsigned abase = lds0 + (unsigned)((src * L::SRCE + 4 * (r >> 3) * L::ROWE + cql * L::CQE + (r & 7) * 2 + (g & 1)) * 16);
    const unsigned gmask = (g >> 1) ? 0xFFFFFFFFu : 0u;
    // output: the lane's 16-byte piece of its pixel's record in the unit it owns (after the hi / lo exchange lane row g holds part g & 1 of
    // channel octet g >> 1 of the output tile)
    const int ntb = (t.pair < 0 ? 0 : t.pair) + 2 * (int)blockIdx.y;   // first 16-channel output tile of this workgroup (RollArgs::pair < 0: grid.y = the 32-channel output halves, one launch)
    auto vob_of = [&](int half) {                      // element offset inside the output slice's column
        const int row = 2 * half + (myu >> 1) + 4 * (r >> 3), col = r & 7;
        return (row * a.Wo + col) * (2 * a.Cout) + (g & 1) * a.Cout + ((ntb + (myu & 1)) * 2 + (g >> 1)) * 8;
    };
    const int vob0 = vob_of(0), vob1 = vob_of(1);
    // exchange: unit u of half-step h is owned by wave 4h + u (NW = 8) / wave u (NW = 4); writer w's partial is the owner's contribution w (w < owner) or w - 1
    const unsigned xlane = lds0 + L::XCH_OFF + lane * 16;
    // writer w's partial of unit u of half-step h goes to slot w (w < owner) or w - 1 of the unit's NW - 1: base + wave KiB, one KiB less behind the owner
    const unsigned xwb = xlane + wave * 1024, xwb1 = xwb - 1024;
    const unsigned xrd = xlane + myu * (NW - 1) * 1024;   // + h * XCHB: the NW - 1 partials of my unit, 1 KiB apart

    // ---- the filter share: 7 chunks x 2 output tiles x (hi, lo), resident for the whole walk ----
    short8 w[NCH][2][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + ((size_t)blockIdx.y * NW + wave) * NCH * 4 * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                w[c][nt][0] = wp[((c * 2 + nt) * 2 + 0) * 64];
                w[c][nt][1] = wp[((c * 2 + nt) * 2 + 1) * 64];
            }
    }
    // the BatchNorm shift of the lane's 4 channels of the output tile this wave owns: the owner's sum starts from it
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + (ntb + (myu & 1)) * 16 + g * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue slices, filter, bias (compiler-visible, so that no later wait is invented)
    asm volatile("s_barrier" ::: "memory");

    int sidx = RING - 1;                  // ring slot of the window's first slice = global step - 1 (the stream's slices sit in consecutive slots)
    f32x4 mine0 = {0.f, 0.f, 0.f, 0.f}, mine1 = {0.f, 0.f, 0.f, 0.f};   // this wave's own partial of the unit it owns in half-step 0 / 1
    // partial columns at the volume's bottom / right edge (H or W not a multiple of 8: the 1/8-resolution volumes of 480 x 640 stacks): the fill's
    // range check zero-pads them, the epilogue's store (and residual read) is predicated on the lane's pixel lying inside: pv0 / pv1 for the unit
    // this wave owns in half-step 0 / 1 of the current column, ppv1 = pv1 of the step before (another column at a column change)
    int pvm = 7;                          // bit 0: pv0, bit 1: pv1, bit 2: ppv1 (one register for the three flags: the residual instantiations sit at the 256-register cap)
    char *pptr = nullptr;                 // where the previous step's output slice starts (wave-uniform)
    // the residual volume has the output's geometry: its slices are addressed through the output pointers + this distance (two fewer loop-carried 64-bit values:
    // at the 256-register cap the 8-wave residual instantiations spilled 7 VGPRs in their prologue)
    const int64_t rdelta = RES ? reinterpret_cast<const char *>(a.res0) - reinterpret_cast<const char *>(a.out) : 0;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Operand fragments of one chunk: [operand tile][part].  Two buffers in rotation over the chunk sequence, which runs on across half-steps
    // and steps (7 chunks per half-step: chunk c of half-step h sits in buffer (h + c) & 1), so that chunk 0 of a half-step is requested
    // BEFORE the barrier in front of it -- during the last chunk of the previous half-step -- and the matrix pipe restarts right behind the
    // barrier.  That is safe for the ring: chunk 0 reads window slice 0 (th = 0) or 1 (th = 1), never the slice still being filled.
    short8 x[2][2][2];
    auto fetch = [](auto BUF, auto H2, auto IMM, short8 (&xx)[2][2][2], const unsigned ad) {
        constexpr int b = decltype(BUF)::value, hh = decltype(H2)::value, im = decltype(IMM)::value;
        constexpr int i0 = im + (2 * hh) * L::ROWE * 16, i1 = im + (2 * hh + 1) * L::ROWE * 16, pb = L::PARTE * 16;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][0]) : "v"(ad), "n"(i0));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][1]) : "v"(ad), "n"(i0 + pb));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][0]) : "v"(ad), "n"(i1));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][1]) : "v"(ad), "n"(i1 + pb));
    };
    using Tap = rollk::TapT<NW, TH>;
    // register part of chunk c's operand address in the window that starts at ring slot `first` (the immediate part is Tap::off(2c))
    auto chunk_addr = [&](auto C, int first) -> unsigned {
        constexpr int c = decltype(C)::value;
        const unsigned s0 = (unsigned)(((first + Tap::dz(2 * c)) & (RING - 1)) * L::SLOTB);
        unsigned delta = (unsigned)(Tap::off(2 * c + 1) - Tap::off(2 * c));
        if constexpr (Tap::dz(2 * c + 1) != Tap::dz(2 * c)) delta += (unsigned)(((first + Tap::dz(2 * c + 1)) & (RING - 1)) * L::SLOTB) - s0;
        return abase + s0 + (delta & gmask);
    };

    // One half-step.  LIVE: contract this window's operand tiles 2h, 2h + 1 (PRE: its chunk 0 was requested by the previous half-step).
    // `nofront` / `noback` (wave-uniform): the window's centre is the volume's first / last slice -- the ring holds real slices only, so the taps
    // of the missing slice (TH = 0: chunks 0-3 and the first tap of chunk 4; TH = 1: chunks 2-6) contract zeroed operands instead of whatever
    // the ring slot holds.  FIN: this wave finishes the unit it owns in the PREVIOUS half-step (partials in exchange buffer h ^ 1, its own in
    // `mine`), output slice at optr_f / residual slice at rptr_f.  LAST: the step ends with this half-step.
    auto half = [&](auto H_, auto LIVE_, auto FIN_, auto LAST_, auto PRE_, const bool nofront, const bool noback, const f32x4 &mine, char *optr_f,
                    const char *rptr_f, int vob_f, int pv_f, f32x4 &mine_out) {
        constexpr int h = decltype(H_)::value;
        constexpr bool LIVE = decltype(LIVE_)::value, FIN = decltype(FIN_)::value, LAST = decltype(LAST_)::value, PRE = decltype(PRE_)::value;
        // the partials of my unit come in two batches (at most 4 + 3: 16 registers instead of 28), each requested behind a chunk's operands
        // and landed by the NEXT chunk's wait (DS operations retire in order)
        constexpr int NPA = FIN ? (NW - 1 < 4 ? NW - 1 : 4) : 0, NPB = FIN ? NW - 1 - NPA : 0;
        f32x4 part[NPA ? NPA : 1];
        u32x4 rq = {0, 0, 0, 0};
        if constexpr (FIN && RES) {
            // (a lane outside the volume reads the column's first pixel instead: always inside; wave-uniform base in SGPRs + the lane's 32-bit byte offset)
            const unsigned ro = (unsigned)(pv_f ? vob_f * 2 : 0);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq) : "v"(ro), "s"(rptr_f) : "memory");
        }
        f32x4 v = zero4;
        auto fin_read = [&](auto B_) {
            constexpr int bt = decltype(B_)::value, cnt = bt ? NPB : NPA, first = bt ? NPA : 0;
            if constexpr (FIN && cnt > 0 && !(ABL & 8)) {
                const unsigned ad = xrd + (h ^ 1) * L::XCHB;
                // (asm operands are passed in: clang rejects captured locals of an enclosing generic lambda as asm operands)
                auto rd = [](auto I, f32x4 (&pp)[NPA ? NPA : 1], const unsigned adr) {
                    constexpr int i = decltype(I)::value;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pp[i]) : "v"(adr), "n"((first + i) * 1024));
                };
                static_for<cnt>([&](auto I) { rd(I, part, ad); });
            }
            if constexpr (FIN && cnt > 0 && (ABL & 8)) static_for<cnt>([&](auto I) { part[decltype(I)::value] = mine; });
        };
        // this wave's own partial, then the others' in wave order: a fixed order per unit (results do not depend on timing)
        auto fin_sum = [&](auto B_) {
            constexpr int bt = decltype(B_)::value, cnt = bt ? NPB : NPA;
            if constexpr (FIN) {
                if constexpr (bt == 0) v = bias4 + mine;
                if constexpr (cnt > 0) {
                    if constexpr (!(ABL & 8) && !(ABL & 1)) rollk::tie(part);
                    static_for<cnt>([&](auto I) { v += part[decltype(I)::value]; });
                }
            }
        };
        auto fin_store = [&]() {
            if constexpr (FIN) {
                if constexpr (RES) {   // (wait, THEN tie: as a "+v" operand of the wait hipcc may copy rq into the operand register in front of it)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+v"(rq));
                }
                uint4 rq4 = make_uint4(rq[0], rq[1], rq[2], rq[3]);
                float cls = 0.f;
                if constexpr (!(ABL & 32))
                    epilogue_lean_t<P_BF16X3>(reinterpret_cast<uint16_t *>(optr_f), nullptr, vob_f, v[0], v[1], v[2], v[3], RES, rq4, RELU, false, zero4, cls, pv_f != 0);
                else asm volatile("" ::"v"(v), "v"(rq4.x));
            }
        };
        if constexpr (LIVE) {
            // The two waves of a SIMD (wave w and w + 4 of the 8-wave workgroup) leave every barrier together, interleave their MFMA bursts one by
            // one and then reach the instructions between two chunks -- operand requests, addresses, side work -- at the same time, with the matrix
            // pipe idle.  The second one starts SKEW x 64 cycles late: its bursts then run under the first one's gaps and vice versa.
            if constexpr (NW == 8 && OWN1 && SKEW > 0) __builtin_amdgcn_s_sleep(SKEW);
            // chunk 0 of the NEXT half-step: the same window's other operand tiles, or (h = 1) the next step's window
            const unsigned nxt0 = chunk_addr(std::integral_constant<int, 0>{}, h == 0 ? sidx : sidx + 1);
            if constexpr (!PRE && !(ABL & 1))
                fetch(std::integral_constant<int, h & 1>{}, H_, std::integral_constant<int, Tap::off(0)>{}, x, chunk_addr(std::integral_constant<int, 0>{}, sidx));
            f32x4 n[4];   // [operand tile][output tile]
#pragma unroll
            for (int u = 0; u < 4; ++u) n[u] = zero4;
            static_for<NCH>([&](auto C) {
                constexpr int c = decltype(C)::value;
                constexpr int cur = (h + c) & 1, nxt = cur ^ 1;
                if constexpr (!(ABL & 1)) {
                    if constexpr (c + 1 < NCH)
                        fetch(std::integral_constant<int, nxt>{}, H_, std::integral_constant<int, Tap::off(2 * (c + 1 < NCH ? c + 1 : 0))>{}, x,
                              chunk_addr(std::integral_constant<int, (c + 1 < NCH ? c + 1 : 0)>{}, sidx));
                    else fetch(std::integral_constant<int, nxt>{}, std::integral_constant<int, h ^ 1>{}, std::integral_constant<int, Tap::off(0)>{}, x, nxt0);
                    if constexpr (c == 1) fin_read(std::integral_constant<int, 0>{});
                    if constexpr (c == 3) fin_read(std::integral_constant<int, 1>{});
                    constexpr int ahead = 4 + ((c == 1 && !(ABL & 8)) ? NPA : 0) + ((c == 3 && !(ABL & 8)) ? NPB : 0);
                    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[cur][0][0]), "+v"(x[cur][0][1]), "+v"(x[cur][1][0]), "+v"(x[cur][1][1]) : "n"(ahead));
                } else {
                    if constexpr (c == 1) fin_read(std::integral_constant<int, 0>{});
                    if constexpr (c == 3) fin_read(std::integral_constant<int, 1>{});
                    asm volatile("" : "+v"(x[cur][0][0]), "+v"(x[cur][0][1]), "+v"(x[cur][1][0]), "+v"(x[cur][1][1]));
                }
                // taps in a slice the volume does not have: zero operands (a uniform branch, taken by two steps per unit)
                if constexpr (TH == 0 && c <= 4) {
                    if (nofront) {
                        const unsigned keep = c == 4 ? gmask : 0u;   // chunk 4 = taps 8 | 9: only its first tap (lanes g < 2) lies in the missing slice
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt) {
                                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                                u32x4v q = __builtin_bit_cast(u32x4v, x[cur][j][pt]);
                                q &= keep;
                                x[cur][j][pt] = __builtin_bit_cast(short8, q);
                            }
                    }
                }
                if constexpr (TH == 1 && c >= 2) {
                    if (noback) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt) x[cur][j][pt] = short8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
                if constexpr (!(ABL & 2)) {
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
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(x[cur][j][0]), "v"(x[cur][j][1]), "v"(w[c][0][0]), "v"(w[c][0][1]), "v"(w[c][1][0]), "v"(w[c][1][1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                // side work in the chunk gaps: the pending unit's sum and epilogue, then (first half-step) the fill of the ring's free slot
                // (chunk c's wait has passed: the batch requested behind chunk c - 1's operands has landed)
                if constexpr (c == 2) fin_sum(std::integral_constant<int, 0>{});
                if constexpr (c == 4) {
                    fin_sum(std::integral_constant<int, 1>{});
                    fin_store();
                }
                if constexpr (h == 0 && c >= 7 - L::PPW && !(ABL & 4)) issue_piece(std::integral_constant<int, c - (7 - L::PPW)>{});
                if constexpr (c == 2 || c == 4 || (h == 0 && c >= 7 - L::PPW)) __builtin_amdgcn_sched_barrier(0);
            });
            // hand the partials to their owners, keep mine.  (The MFMA -> DS wait states: hipcc does not see that an asm blob reads an
            // accumulator, dffw_srd_roll.hip's v_med3 trap -- so they are spelled out.)
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]));
            constexpr bool OWN = h == 0 ? OWN0 : OWN1;
            if constexpr (!(ABL & 8)) {
                auto xw = [](auto U, auto BEHIND, const unsigned (&bb)[2], const f32x4 (&nn)[4]) {
                    constexpr int u = decltype(U)::value, imm = h * L::XCHB + u * (NW - 1) * 1024;
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(bb[decltype(BEHIND)::value ? 1 : 0]), "v"(nn[u]), "n"(imm) : "memory");
                };
                const unsigned bb[2] = {xwb, xwb1};
                static_for<4>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    if constexpr (OWN) {
                        // the owners of this half-step's units are this wave's group: wave w and owner u + 4h differ as myu and u do
                        if (myu > u) xw(U, std::true_type{}, bb, n);
                        else if (myu < u) xw(U, std::false_type{}, bb, n);
                    } else {
                        // (NW = 8) the owners are the other group: all in front of this wave (h = 0, waves 4-7 writing) or all behind it (h = 1)
                        xw(U, std::integral_constant<bool, h == 0>{}, bb, n);
                    }
                });
            } else {
                asm volatile("" ::"v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]));
            }
            if constexpr (OWN) {
                mine_out = n[TH];                      // myu = wave & 3 is TH or TH + 2
                if (myu >= 2) mine_out = n[TH + 2];
            }
        } else {
            if constexpr (FIN) {
                fin_read(std::integral_constant<int, 0>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fin_sum(std::integral_constant<int, 0>{});
                fin_read(std::integral_constant<int, 1>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fin_sum(std::integral_constant<int, 1>{});
                fin_store();
            }
            if constexpr (!(ABL & 4)) static_for<L::PPW>([&](auto K) { issue_piece(K); });
        }
        if constexpr (LAST) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (!(ABL & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the operands requested for the next half-step have landed (lgkmcnt(0) above): from here on they are ordinary values
        if constexpr (LIVE) {
            constexpr int nb = (h + NCH) & 1;
            asm volatile("" : "+v"(x[nb][0][0]), "+v"(x[nb][0][1]), "+v"(x[nb][1][0]), "+v"(x[nb][1][1]));
        }
    };

    // One step of the stream.  LIVE: this window produces an output slice (nofront / noback: see half); PEND: the previous one did (its second
    // half-step's units are finished now).
    auto step = [&](auto LIVE_, auto PEND_, const bool nofront, const bool noback, char *optr) {
        constexpr bool LIVE = decltype(LIVE_)::value, PEND = decltype(PEND_)::value;
        using T = std::true_type;
        using F = std::false_type;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        f32x4 dummy;
        if constexpr (LIVE) {
            // (a live step behind a live step finds its chunk 0 requested by that step's second half-step)
            half(I0{}, T{}, std::integral_constant<bool, PEND && OWN1>{}, F{}, std::integral_constant<bool, PEND>{}, nofront, noback, mine1, pptr, pptr + rdelta, vob1, pvm & 4, OWN0 ? mine0 : dummy);
            half(I1{}, T{}, std::integral_constant<bool, OWN0>{}, T{}, T{}, nofront, noback, mine0, optr, optr + rdelta, vob0, pvm & 1, OWN1 ? mine1 : dummy);
        } else {
            half(I0{}, F{}, std::integral_constant<bool, PEND && OWN1>{}, T{}, F{}, false, false, mine1, pptr, pptr + rdelta, vob1, pvm & 4, dummy);
        }
        sidx = (sidx + 1) & (RING - 1);
        advance_fill();
        pptr = optr;
        pvm = (pvm & 3) | ((pvm & 2) << 1);
    };
    auto dispatch = [&](bool live, bool pend, bool nofront, bool noback, char *optr) {
        using T = std::true_type;
        using F = std::false_type;
        if (live) {
            if (pend) step(T{}, T{}, nofront, noback, optr);
            else step(T{}, F{}, nofront, noback, optr);
        } else {
            if (pend) step(F{}, T{}, false, false, optr);
            else step(F{}, F{}, false, false, optr);
        }
    };

    bool prev_live = false;
    const int64_t ostride = (int64_t)a.Ho * a.Wo * a.Cout * 4;   // bytes per output slice
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t o0 = ((((int64_t)U.b * a.No + U.zbeg) * a.Ho + U.gy0) * a.Wo + U.gx0) * a.Cout * 4;
        // Step st of the unit's ns input slices contracts the window centred on its slice st = output slice st - h0 (h0: the unit has a halo slice
        // in front, i.e. its range starts inside the volume): a range that starts / ends inside the volume costs a dead step there, one that
        // reaches the volume's end none -- a unit that covers all slices is ns live steps.
        {
            const int col = U.gx0 + (r & 7), row = U.gy0 + (myu >> 1) + 4 * (r >> 3);
            pvm = (pvm & 4) | ((col < a.Wo && row < a.Ho) ? 1 : 0) | ((col < a.Wo && row + 2 < a.Ho) ? 2 : 0);
        }
        const int zlo = U.zbeg > 0 ? U.zbeg - 1 : 0, h0 = U.zbeg - zlo;
        const int ns = (U.zbeg + U.nz < a.Ni ? U.zbeg + U.nz : a.Ni - 1) - zlo + 1;
        char *optr = reinterpret_cast<char *>(a.out) + o0 - (int64_t)h0 * ostride;
        for (int st = 0; st < ns; ++st) {
            const bool live = st >= h0 && st - h0 < U.nz;
            dispatch(live, prev_live, st == 0, st == ns - 1, optr);
            prev_live = live;
            optr += ostride;
        }
    }
    // the last live step's second half-step is finished by the step behind it: past the end of the stream, one more (dead) step
    if (prev_live) dispatch(false, true, false, false, nullptr);
    // ... and the slices queued past the end of the stream are still in flight: a wave must not retire before its LDS-DMA has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NW, bool RELU, bool RES, int ABL = 0, int SKEW = DFFW_ROLLK_SKEW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rollk(const ConvArgs a, const RollArgs t) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[rollk::Lay<NW>::LDSB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the body is specialised on the wave's tap half (every tap offset an immediate) and, for 8 waves, on the group that owns half-step 0 / 1's units
    if constexpr (NW == 8) {
        if (wave < 4) {
            if (wave & 1) rollk_body<NW, RELU, RES, true, false, 1, ABL, SKEW>(a, t, smem, lane, wave);
            else rollk_body<NW, RELU, RES, true, false, 0, ABL, SKEW>(a, t, smem, lane, wave);
        } else {
            if (wave & 1) rollk_body<NW, RELU, RES, false, true, 1, ABL, SKEW>(a, t, smem, lane, wave);
            else rollk_body<NW, RELU, RES, false, true, 0, ABL, SKEW>(a, t, smem, lane, wave);
        }
    } else {
        if (wave & 1) rollk_body<NW, RELU, RES, true, true, 1, ABL, SKEW>(a, t, smem, lane, wave);
        else rollk_body<NW, RELU, RES, true, true, 0, ABL, SKEW>(a, t, smem, lane, wave);
    }
}

void rollk_tile(int *ty, int *tx) {
    *ty = rollk::TY;
    *tx = rollk::TX;
}

// nw = waves per workgroup the layer needs: 8 for 64 input channels, 4 for 32 (0: not covered)
int rollk_waves(int prec, const ConvArgs &a) {
    if (prec != P_BF16X3 || (a.dbg & DFFW_ARGS_NO_ROLLK)) return 0;
    if (!a.out || a.out_pre || a.outf || a.res1 || a.res_bcast || a.cls_w || a.relu == 2 || a.Cout % 32) return 0;
    const int cin = a.C0 + a.C1;
    if (cin != 32 && cin != 64) return 0;
    if (a.C1 && a.C1 != a.C0) return 0;
    // 32-bit buffer offsets: a sample's input volume (+ one footprint) stays below 2^31 bytes
    const int64_t recb = (int64_t)a.C0 * 4;
    if ((int64_t)(a.Ni + 1) * a.Hi * a.Wi * recb >= (1ll << 31)) return 0;
    return cin / 8;
}

hipError_t launch_conv_rollk(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int nw = (a.C0 + a.C1) / 8;
    const int ny = t.pair < 0 ? a.Cout / 32 : 1;                  // (RollArgs::pair < 0: every 32-channel output half in ONE launch, as grid.y)
    const int want = (t.wgs > 0 ? t.wgs : (nw == 8 ? 256 : 512)) / ny;   // 16 waves per CU either way
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8))), (unsigned)ny), block(nw * 64);
    const bool relu = a.relu == 1, res = a.res0 != nullptr;
#define DFFW_ROLLK_LAUNCH(NW, RL, RS) hipLaunchKernelGGL((conv_rollk<NW, RL, RS>), grid, block, 0, s, a, t)
#ifdef DFFW_ABL_BUILD   // development (make ABL=1): timing ablations of the two most used instantiations (results are wrong)
    if (const char *z = getenv("DFFW_ROLLK_SKEW")) {
        const int sk = atoi(z);
        if (nw == 8 && relu && !res) {
            if (sk == 0) { hipLaunchKernelGGL((conv_rollk<8, true, false, 0, 0>), grid, block, 0, s, a, t); return hipGetLastError(); }
            if (sk == 1) { hipLaunchKernelGGL((conv_rollk<8, true, false, 0, 1>), grid, block, 0, s, a, t); return hipGetLastError(); }
            if (sk == 3) { hipLaunchKernelGGL((conv_rollk<8, true, false, 0, 3>), grid, block, 0, s, a, t); return hipGetLastError(); }
            if (sk == 4) { hipLaunchKernelGGL((conv_rollk<8, true, false, 0, 4>), grid, block, 0, s, a, t); return hipGetLastError(); }
            if (sk == 6) { hipLaunchKernelGGL((conv_rollk<8, true, false, 0, 6>), grid, block, 0, s, a, t); return hipGetLastError(); }
        }
    }
    if (const char *z = getenv("DFFW_ROLLK_ABL")) {
        const int abl = atoi(z);
#define DFFW_ROLLK_ABL_CASE(A)                                                                                  \
    if (abl == A && relu && !res) {                                                                             \
        if (nw == 8) hipLaunchKernelGGL((conv_rollk<8, true, false, A>), grid, block, 0, s, a, t);             \
        else hipLaunchKernelGGL((conv_rollk<4, true, false, A>), grid, block, 0, s, a, t);                     \
        return hipGetLastError();                                                                               \
    }
        DFFW_ROLLK_ABL_CASE(1) DFFW_ROLLK_ABL_CASE(2) DFFW_ROLLK_ABL_CASE(3) DFFW_ROLLK_ABL_CASE(4) DFFW_ROLLK_ABL_CASE(8) DFFW_ROLLK_ABL_CASE(16)
        DFFW_ROLLK_ABL_CASE(24) DFFW_ROLLK_ABL_CASE(32) DFFW_ROLLK_ABL_CASE(44) DFFW_ROLLK_ABL_CASE(47) DFFW_ROLLK_ABL_CASE(63)
#undef DFFW_ROLLK_ABL_CASE
    }
#endif
    if (nw == 8) {
        if (relu && res) DFFW_ROLLK_LAUNCH(8, true, true);
        else if (relu) DFFW_ROLLK_LAUNCH(8, true, false);
        else if (res) DFFW_ROLLK_LAUNCH(8, false, true);
        else DFFW_ROLLK_LAUNCH(8, false, false);
    } else {
        if (relu && res) DFFW_ROLLK_LAUNCH(4, true, true);
        else if (relu) DFFW_ROLLK_LAUNCH(4, true, false);
        else if (res) DFFW_ROLLK_LAUNCH(4, false, true);
        else DFFW_ROLLK_LAUNCH(4, false, false);
    }
#undef DFFW_ROLLK_LAUNCH
    return hipGetLastError();
}

void conv_rollk_kernel_name(const ConvArgs &a, char *buf, int n) {
    // (rocprofv3's spelling, defaulted template arguments included: tools/hbm_traffic.py and the PMC summaries match kernels by name)
    snprintf(buf, n, "dffw::conv_rollk<%d, %s, %s, 0, %d>", (a.C0 + a.C1) / 8, a.relu == 1 ? "true" : "false", a.res0 ? "true" : "false", DFFW_ROLLK_SKEW);
}

}  // namespace dffw
