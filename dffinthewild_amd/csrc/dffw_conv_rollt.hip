// conv_rollt: ConvTranspose3d k3 s(1,2,2) p1 op(0,1,1) over 32 / 64 input channels as a persistent streaming kernel (round 6), gfx950 / MI355X.
// `deconv_1` (64 -> 32), `dres2.conv5` (64 -> 64: grid.y = 2 output halves), `dres2.conv6` (64 -> 32 + skip + second output + fused classifier),
// `dres3.conv5` (32 -> 32 + residual), the pyramid's `conv9` (64 -> 32): DEN.py:41-42, 194-200, 260-264.
//
// conv_tile runs these layers as four sub-pixel passes over one 5 x 4 x 16 LDS image per workgroup: the footprint fill is 48 % of the family's time and
// overlaps nothing, the filter streams from L2 once per wave and chunk, and the counted HBM traffic is 1.63x the algorithmic bytes
// (profiles/r05_ablation_conv_tile_phases.txt, r05_pmc_conv_kernels.txt).  Here, as in conv_rollk / conv_slice64:
//   * a persistent workgroup walks the focus slices of 8 x 8 columns of the INPUT grid (16 x 16 output pixels per slice): input slice images go
//     through a ring of 4 LDS slots by buffer-addressed LDS-DMA one slice ahead of the window (in-plane halo 81 / 64, none along the slices);
//   * the filter is resident in registers and split over the waves by OUTPUT PHASE (py, px) and 16-channel output tile -- the phases own 3 / 6 / 6 / 12
//     of the 27 taps and disjoint output pixels, so a wave that holds a phase's taps for all of K finishes its outputs alone (conv_slice64's
//     no-exchange scheme).  64 input channels: phase (1,1) is 24 weight units (192 VGPRs) per output tile, too many for one wave at two waves per
//     SIMD, so the wave pair A / B splits it (and the 3-tap phase (0,0), which balances the load) over the two 32-channel K halves and exchanges ONE
//     partial tile per operand tile through LDS (conv_rollx_k2's one-unit exchange): A finishes (0,0), B finishes (1,1); waves C / D hold (0,1) / (1,0)
//     for all of K.  15 / 15 / 12 / 12 units of 8 VGPRs.  32 input channels: A32 = (1,1) + (0,0), C32 = (0,1) + (1,0), no exchange, four waves;
//   * a step (one output slice) is two passes over two operand tiles (rows j, j + 4 of the column: conv_rollk's conflict-free tile) each; an operand
//     fragment set (4 ds_read_b128) feeds 6 or 12 MFMAs; the results of a pass are finished -- partner's partial added, BatchNorm shift (in the
//     accumulator init), residual, ReLU, classifier, stores -- inside the NEXT pass's contraction, one barrier per pass, the exchange area double
//     buffered by pass parity (writes are ordered behind the previous content's reads by that barrier, no timing assumption);
//   * LDS slice image [16-channel group][part][row 9][pixel 9 + 1 pad][octet] in 16-byte entries: row pitch 20, so rows j and j + 4 of an operand tile
//     are a multiple of 256 bytes apart and lane rows g, g + 1 take the two octets of one tap: 16 distinct bank groups per ds_read_b128 service group.
// Epilogue arithmetic = epilogue_lean_t (conv_tile's LEAN epilogue).  The fused classifier's dot spans both output tiles of a pixel, i.e. two waves:
// each adds its 16-channel partial to the (zeroed) score volume with one atomic add per pixel -- two addends, so the sum does not depend on their order.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "dffw_conv_roll.h"
#include "dffw_device.h"

namespace dffw {

namespace rollt {
constexpr int TY = DFFW_ROLLT_TY, FY = TY + 1;
// WIDE (32 -> 16 channels: `deconv_2`, `dres3.conv6`): one 16-channel output tile only, so the workgroup's eight waves are (role A32 / C32) x FOUR pixel
// sub-blocks of an 8 x 16 column -- sub-block s = operand rows 2 (s >> 1), +1 (and +4) of column half s & 1 -- and a step is ONE pass per wave; ring of 5
// slots (the slice issued in a step is waited for in front of the NEXT step's epilogue: a one-pass step is too short to cover the DMA latency)
template <int CIN, bool WIDE>
struct Lay {
    static constexpr int TX = WIDE ? 2 * DFFW_ROLLT_TX : DFFW_ROLLT_TX;
    static constexpr int ROWP = 2 * (TX + 2);     // row pitch in entries: [pixel TX + 1 (+ 1 pad)][octet]
    static constexpr int RING = WIDE ? 5 : 4;
    static constexpr int NW = WIDE ? 8 : CIN / 8; // waves per workgroup
    static constexpr int NG = CIN / 16;           // 16-channel groups of the input (16 channels: one group, read by both K octet pairs)
    static constexpr int PARTE = FY * ROWP;       // entries of one part of a group: [row][pixel][octet]
    static constexpr int CQE = 2 * PARTE;         // ... of a group: [part][row][pixel][octet]
    static constexpr int SLOTE = NG * CQE;
    static constexpr int NPIECE = (SLOTE + 63) / 64;
    static constexpr int SLOTB = NPIECE * 1024;
    static constexpr int PPW = (NPIECE + NW - 1) / NW;   // DMA pieces per wave and slice (the last one exists for the first waves only)
    static constexpr int XCH_OFF = RING * SLOTB;
    static constexpr int XCHB = (CIN == 64 && !WIDE) ? 8 * 1024 : 0;   // one pass's exchange: [output tile 2][direction 2][operand tile 2] partials of 1 KiB
    static constexpr int LDSB = XCH_OFF + 2 * XCHB;
    static_assert(DFFW_ROLLT_TY == 8 && DFFW_ROLLT_TX == 8 && (4 * ROWP) % 16 == 0, "operand tiles are rows (j, j + 4) x 8 columns, a multiple of 256 bytes apart");
    static_assert(SLOTB % 256 == 0 && LDSB <= 160 * 1024, "LDS layout");
    static_assert(!WIDE || CIN == 32 || CIN == 16, "the wide form is built for 32 input channels (16: `dres4.conv5`, the upper K octets carry zero weights)");
    static_assert(CIN == 16 ? WIDE : true, "16 input channels: wide form only");
};
// orders every later use of p[] behind the (volatile) asm statements before this point: the counted waits that cover their ds_reads
template <int N>
__device__ __forceinline__ void tie(f32x4 (&p)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(p[i]));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// every outstanding vector-memory operation has completed; the residual pieces p[] requested by asm loads are ordinary values from here on
template <int N>
__device__ __forceinline__ void wait_vm0(u32x4 (&p)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // (no copy hipcc makes for the ties below may be scheduled in front of the wait: dffw_conv_slice.hip, conv_slice64_head)
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(p[i]));
}
// a residual piece: wave-uniform base in SGPRs + the lane's 32-bit byte offset (no 64-bit address per lane); completed by wait_vm0
__device__ __forceinline__ void res_load(u32x4 &dst, const unsigned ro, const char *base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(ro), "s"(base) : "memory");
}
}   // namespace rollt

// MODE 0: out = [relu](acc); 1: out = [relu](acc + residual); 2: + second output (the value before the residual) + fused 1x1x1 classifier
template <int CIN, int ROLE, int MODE, bool WIDE>
__device__ __forceinline__ void rollt_body(const ConvArgs &a, const RollArgs &t, unsigned char *smem, const int lane, const int wave) {
    using namespace rollt;
    using L = Lay<CIN, WIDE>;
    using P = Prog<ROLE>;
    constexpr bool RES = MODE >= 1, FULL = MODE == 2;
    constexpr int NW = L::NW, NS = P::NS, NACC = P::NACC, NOWN = P::NOWN, NPT = 2 * NOWN, SLOTB = L::SLOTB, TX = L::TX, ROWP = L::ROWP, RING = L::RING;
    static_assert(!WIDE || !P::XCH, "wide form: roles A32 / C32");
    const int g = lane >> 4, r = lane & 15;
    const int nt = WIDE ? 0 : (wave >> 1) & 1;
    const int ntg = WIDE ? 0 : (int)blockIdx.y * 2 + nt;   // this wave's 16-channel output tile
    const int psw = WIDE ? wave >> 2 : 0, hw = WIDE ? (wave >> 1) & 1 : 0;   // wide form: this wave's operand row pair and column half
    const bool relu = a.relu == 1;

    // ---- this workgroup's units (8 x 8 columns of one sample's input grid): XCD x owns a contiguous range, as conv_roll -------------
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

    // ---- fill: pieces of 64 consecutive 16-byte entries of the slot; per lane the byte offset from the unit's footprint origin (out-of-image,
    // padding and tail lanes pushed out of range: the buffer range check writes their zeros = the taps beyond the bottom / right edge) ------------
    const int partb = CIN * 2, recb = 2 * partb;       // a pixel record [hi CIN][lo CIN]
    const char *tb = reinterpret_cast<const char *>(a.in0);
    const int slice_bytes = a.Hi * a.Wi * recb;
    int fvo[L::PPW];
    const char *fb = tb;
    int fu = ufirst, fz = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fz = 0;
        fb = tb + (((int64_t)c.b * a.Ni * a.Hi + c.gy0) * a.Wi + c.gx0) * recb;
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // (opaque: no hoisting of the decode out of the unit loop)
#pragma unroll
        for (int k = 0; k < L::PPW; ++k) {
            const int e = (k * NW + wave) * 64 + ln;       // entry inside the slot: [group][part][row][pixel][octet]
            const int gq = e / L::CQE, e2 = e - gq * L::CQE;
            const int part = e2 / L::PARTE, e3 = e2 - part * L::PARTE;
            const int fy = e3 / ROWP, e4 = e3 - fy * ROWP;
            const int fx = e4 >> 1, oct = e4 & 1;
            const int iy = c.gy0 + fy, ix = c.gx0 + fx;
            fvo[k] = (e < L::SLOTE && fx <= TX && iy < a.Hi && ix < a.Wi) ? (fy * a.Wi + fx) * recb + part * partb + (gq * 2 + oct) * 16 : (int)0x80000000;
        }
    };
    setup_fill();
    int fslotb = 0;
    auto issue_piece = [&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const int p = k * NW + wave;
        if (p >= L::NPIECE) return;                        // (wave-uniform)
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
    for (int q = 0; q < RING - 2; ++q) {                   // the fill runs two (wide form: three) slices ahead of the window's centre
        static_for<L::PPW>([&](auto K) { issue_piece(K); });
        advance_fill();
    }

    // ---- operand addressing: K octet g of a chunk = channels 32 chunk + 8 g .. = (group 2 chunk + (g >> 1), octet g & 1) of the set's tap; lane r of
    // operand tile j = input pixel (row j + 4 (r >> 3), column r & 7): window slice -> one of three address registers, everything else an immediate
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned abase = lds0 + (unsigned)(((CIN >= 32 ? (g >> 1) : 0) * L::CQE + (4 * (r >> 3) + 2 * psw) * ROWP + (8 * hw + (r & 7)) * 2 + (g & 1)) * 16);
    // output: own slot k = phase (py, px): the lane's 16-byte piece (part g & 1 of channel octet 2 ntg + (g >> 1)) of output pixel (2 row + py, 2 col + px)
    // of operand tile 0; tile j adds 2 j output rows (wave-uniform).  vcls: the pixel's position in the fp32 score volume
    int vob[NOWN], vcls[NOWN];
#pragma unroll
    for (int k = 0; k < NOWN; ++k) {
        const int ph = P::phase(P::own(k)), py = ph >> 1, px = ph & 1;
        vcls[k] = (8 * (r >> 3) + py) * a.Wo + 2 * (8 * hw + (r & 7)) + px;
        vob[k] = vcls[k] * (2 * a.Cout) + (g & 1) * a.Cout + (ntg * 2 + (g >> 1)) * 8;
    }
    const int tstride = 8 * a.Wo * a.Cout;               // bytes between the output rows of consecutive operand tiles (2 rows of 2 Cout 16-bit elements)
    // exchange (K-split pair): direction 0 = A's partial of phase (1,1) for B, 1 = B's partial of phase (0,0) for A
    const unsigned xlane = lds0 + L::XCH_OFF + lane * 16;
    const unsigned xwr = xlane + (unsigned)(((nt * 2 + (P::SEND > 0 ? 1 : 0)) * 2) * 1024);
    const unsigned xrd = xlane + (unsigned)(((nt * 2 + (P::SEND > 0 ? 0 : 1)) * 2) * 1024);

    // ---- the filter share: NU units x (hi, lo), resident for the whole walk ----
    short8 w[P::NU][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wroll) + (WIDE ? (size_t)(wave & 1) : (size_t)blockIdx.y * NW + wave) * MAXU * 2 * 64 + lane;
#pragma unroll
        for (int u = 0; u < P::NU; ++u) {
            w[u][0] = wp[(u * 2 + 0) * 64];
            w[u][1] = wp[(u * 2 + 1) * 64];
        }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + ntg * 16 + g * 4);
    f32x4 clsw = zero4;
    if constexpr (FULL) {
        if (a.cls_w) clsw = *reinterpret_cast<const f32x4 *>(a.cls_w + ntg * 16 + g * 4);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue slices, filter, bias (compiler-visible, so that no later wait is invented)
#pragma unroll
    for (int u = 0; u < P::NU; ++u) asm volatile("" : "+v"(w[u][0]), "+v"(w[u][1]));   // (pinned: never re-loaded in front of an MFMA)
    asm volatile("s_barrier" ::: "memory");

    int sidx = RING - 1;                  // ring slot of the window's first slice (the stream's slices sit in consecutive slots)
    f32x4 pend[NOWN][2];                  // the own accumulators of the previous pass (BatchNorm shift included)
#pragma unroll
    for (int k = 0; k < NOWN; ++k) pend[k][0] = pend[k][1] = zero4;

    // Operand fragments of one set: [operand tile][part].  Two buffers in rotation over the set sequence, which runs on across passes and steps
    // (set i of pass ps sits in buffer (ps NS + i) & 1), so that set 0 of a pass is requested BEFORE the barrier in front of it -- during the last
    // set of the previous pass -- and the matrix pipe restarts right behind the barrier.  That is safe for the ring: set 0 reads window slice 0.
    short8 x[2][2][2];
    auto fetch = [](auto BUF, auto IMM, short8 (&xx)[2][2][2], const unsigned ad) __attribute__((always_inline)) {
        constexpr int b = decltype(BUF)::value, im = decltype(IMM)::value, t1 = ROWP * 16, pb = L::PARTE * 16;
        static_assert(im + t1 + pb < 65536, "ds_read immediate");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][0]) : "v"(ad), "n"(im));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][1]) : "v"(ad), "n"(im + pb));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][0]) : "v"(ad), "n"(im + t1));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][1]) : "v"(ad), "n"(im + t1 + pb));
    };

    // One pass = operand tiles 2 ps, 2 ps + 1 of the step's window.  LIVE: contract (PRE: its set 0 was requested by the pass in front); PEND: finish the
    // own tiles of the PREVIOUS pass (tiles 2 (ps ^ 1) ..  of the step whose output / residual / second-output / score slices start at o_f / res_f /
    // pre_f / cls_f; rowlim_f: the lane's rows left inside the volume, 0 for a column outside it).  `nofront` / `noback` (wave-uniform): the window's
    // centre is the volume's first / last slice -- the ring holds real slices only, so the sets of the missing slice contract zeroed operands.
    auto pass = [&](auto PS_, auto LIVE_, auto PEND_, auto PRE_, const bool nofront, const bool noback, char *o_f, char *pre_f, const char *res_f,
                    float *cls_f, const int rowlim_f) __attribute__((always_inline)) {
        constexpr int ps = decltype(PS_)::value, pps = ps ^ 1;   // (wide form: ps = the step's parity -- it only selects the fragment buffers)
        const int tg0 = WIDE ? 2 * psw : 2 * pps;               // first operand tile (row) of the pending pass
        constexpr bool LIVE = decltype(LIVE_)::value, PEND = decltype(PEND_)::value, PRE = decltype(PRE_)::value;
        auto bufof = [](int q, int i) constexpr { return (q * NS + i) & 1; };
        auto immof = [](int q, int i) constexpr { return (P::chunk(i) * 2 * L::CQE + ((WIDE ? 0 : 2 * q) + P::dy(i)) * ROWP + P::dx(i) * 2) * 16; };
        // residual pieces of the pending tiles (a lane outside the volume reads the column's first pixel instead: always inside)
        u32x4 rq[NPT];
#pragma unroll
        for (int e = 0; e < NPT; ++e) rq[e] = u32x4{0, 0, 0, 0};
        if constexpr (RES && PEND) {
            static_for<NPT>([&](auto E) __attribute__((always_inline)) {
                constexpr int e = decltype(E)::value, k = e / 2;
                const int tg = tg0 + e % 2;
                const unsigned ro = tg < rowlim_f ? (unsigned)(vob[k] * 2 + tg * tstride) : 0u;
                rollt::res_load(rq[e], ro, res_f);
            });
        }
        f32x4 part[2] = {zero4, zero4};   // the partner's partials of the pending own tiles
        auto part_read = [](auto TT, f32x4 (&pp)[2], const unsigned adr) __attribute__((always_inline)) {
            constexpr int tt = decltype(TT)::value;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pp[tt]) : "v"(adr), "n"(pps * L::XCHB + tt * 1024));
        };
        auto epi = [&](auto E) __attribute__((always_inline)) {
            constexpr int e = decltype(E)::value, k = e / 2, tt = e % 2;
            const int tg = tg0 + tt;
            f32x4 vv = pend[k][tt];
            if constexpr (P::XCH) vv += part[tt];
            const bool pv = tg < rowlim_f;
            char *ob = o_f ? o_f + tg * tstride : nullptr;
            char *pb = (FULL && pre_f) ? pre_f + tg * tstride : nullptr;
            const uint4 rq4 = make_uint4(rq[e][0], rq[e][1], rq[e][2], rq[e][3]);
            float cls = 0.f;
            epilogue_lean_t<P_BF16X3>(reinterpret_cast<uint16_t *>(ob), reinterpret_cast<uint16_t *>(pb), vob[k], vv[0], vv[1], vv[2], vv[3], RES, rq4, relu,
                                      FULL && cls_f != nullptr, clsw, cls, pv);
            if constexpr (FULL) {
                if (cls_f) {
                    cls += __shfl_xor(cls, 16);
                    cls += __shfl_xor(cls, 32);
                    if (g == 0 && pv) unsafeAtomicAdd(cls_f + tg * 2 * a.Wo + vcls[k], cls);
                }
            }
        };
        if constexpr (LIVE) {
            unsigned adw[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) adw[d] = abase + (unsigned)((sidx + d >= RING ? sidx + d - RING : sidx + d) * SLOTB);
            if constexpr (!PRE) fetch(std::integral_constant<int, bufof(ps, 0)>{}, std::integral_constant<int, immof(ps, 0)>{}, x, adw[0]);
            f32x4 n[NACC][2];
#pragma unroll
            for (int s = 0; s < NACC; ++s) n[s][0] = n[s][1] = (s == P::SEND) ? zero4 : bias4;
            static_for<NS>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                constexpr int cur = bufof(ps, i);
                // the next set's fragments: of this pass, of the step's second pass, or of the next step's window (its slice 0 = this window's slice 1)
                if constexpr (i + 1 < NS)
                    fetch(std::integral_constant<int, bufof(ps, i + 1 < NS ? i + 1 : 0)>{}, std::integral_constant<int, immof(ps, i + 1 < NS ? i + 1 : 0)>{}, x, adw[P::d(i + 1 < NS ? i + 1 : 0)]);
                else if constexpr (ps == 0 && !WIDE) fetch(std::integral_constant<int, bufof(1, 0)>{}, std::integral_constant<int, immof(1, 0)>{}, x, adw[0]);
                else fetch(std::integral_constant<int, bufof(ps + 1, 0)>{}, std::integral_constant<int, immof(0, 0)>{}, x, adw[1]);
                constexpr bool PR = P::XCH && PEND && i == 1;
                if constexpr (PR) {
                    part_read(std::integral_constant<int, 0>{}, part, xrd);
                    part_read(std::integral_constant<int, 1>{}, part, xrd);
                }
                // (the tie is a statement of its own BEHIND the wait: as "+v" operands of the wait itself hipcc may copy the registers into the asm's operand
                // registers in front of it, i.e. read a fragment that has not landed -- dffw_conv_slice.hip's conv_slice64_head met exactly that)
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PR ? 6 : 4));
                asm volatile("" : "+v"(x[cur][0][0]), "+v"(x[cur][0][1]), "+v"(x[cur][1][0]), "+v"(x[cur][1][1]));
                // sets of a slice the volume does not have: zero operands (a uniform branch, taken by two steps per unit)
                if constexpr (P::d(i) != 1) {
                    if (P::d(i) == 0 ? nofront : noback) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt) x[cur][j][pt] = short8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
                // product-major over the set's accumulators: consecutive MFMAs never share one
                constexpr int fm = P::feeds(i), ub = P::ubase(i);
                static_for<3>([&](auto PR_) __attribute__((always_inline)) {
                    constexpr int pr = decltype(PR_)::value, wp = pr == 0 ? 1 : 0, xp = pr == 1 ? 1 : 0;   // w_lo x_hi, w_hi x_lo, w_hi x_hi
                    static_for<NACC>([&](auto S_) __attribute__((always_inline)) {
                        constexpr int sl = decltype(S_)::value;
                        if constexpr ((fm >> sl) & 1) {
                            constexpr int u = ub + ((sl == 1 && (fm & 1)) ? 1 : 0);
                            n[sl][0] = mma<false>(w[u][wp], x[cur][0][xp], n[sl][0]);
                            n[sl][1] = mma<false>(w[u][wp], x[cur][1][xp], n[sl][1]);
                        }
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
                // side work in the set gaps: the pending tiles' epilogues, then (first pass) the fill of the ring's free slot
                if constexpr (PEND) {
                    if constexpr (P::XCH && i == 2) rollt::tie(part);   // (set 2's wait has passed: the partials requested behind set 2's operands have landed)
                    if constexpr (i >= 3 && i - 3 < NPT) {
                        if constexpr ((RES || WIDE) && i == 3) rollt::wait_vm0(rq);   // (wide form: also the slice queued in the previous step)
                        epi(std::integral_constant<int, (i >= 3 && i - 3 < NPT) ? i - 3 : 0>{});
                    }
                }
                if constexpr ((ps == 0 || WIDE) && i >= NS - L::PPW) issue_piece(std::integral_constant<int, (i >= NS - L::PPW) ? i - (NS - L::PPW) : 0>{});
                if constexpr ((PEND && i >= 2 && i - 3 < NPT) || ((ps == 0 || WIDE) && i >= NS - L::PPW)) __builtin_amdgcn_sched_barrier(0);
            });
            // hand the partial to the partner, keep the own tiles for the next pass.  (The MFMA -> DS wait states: hipcc does not see that an asm
            // blob reads an accumulator, so they are spelled out.)
            if constexpr (P::XCH) {
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(n[P::SEND < 0 ? 0 : P::SEND][0]), "+v"(n[P::SEND < 0 ? 0 : P::SEND][1]));
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(xwr), "v"(n[P::SEND < 0 ? 0 : P::SEND][0]), "n"(ps * L::XCHB) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(xwr), "v"(n[P::SEND < 0 ? 0 : P::SEND][1]), "n"(ps * L::XCHB + 1024) : "memory");
            }
#pragma unroll
            for (int k = 0; k < NOWN; ++k) {
                pend[k][0] = n[P::own(k)][0];
                pend[k][1] = n[P::own(k)][1];
            }
            if constexpr (ps == 1 && !WIDE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the operands requested for the next pass have landed (lgkmcnt(0) above): from here on they are ordinary values
            {
                constexpr int nb = bufof(ps + 1, 0);
                asm volatile("" : "+v"(x[nb][0][0]), "+v"(x[nb][0][1]), "+v"(x[nb][1][0]), "+v"(x[nb][1][1]));
            }
        } else if constexpr (PEND) {
            // past the end of the stream: the last pass's tiles
            if constexpr (P::XCH) {
                part_read(std::integral_constant<int, 0>{}, part, xrd);
                part_read(std::integral_constant<int, 1>{}, part, xrd);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rollt::tie(part);
            }
            if constexpr (RES || WIDE) rollt::wait_vm0(rq);
            static_for<NPT>([&](auto E) __attribute__((always_inline)) { epi(E); });
        }
    };

    using T = std::true_type;
    using F = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // where the previous step's output / second-output / residual / score slices start (wave-uniform), and its row limit
    char *pptr = nullptr, *ppre = nullptr;
    const char *pres = nullptr;
    float *pcls = nullptr;
    int prowlim = 0;
    const int64_t ostride = (int64_t)a.Ho * a.Wo * a.Cout * 4;   // bytes per output slice
    const int64_t cstride = (int64_t)a.Ho * a.Wo;                // score floats per slice
    bool first = true;
    int par = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        const int64_t p0 = (((int64_t)U.b * a.No * a.Ho + 2 * U.gy0) * a.Wo + 2 * U.gx0);   // the column's first output pixel in slice 0
        char *optr = a.out ? reinterpret_cast<char *>(a.out) + p0 * a.Cout * 4 : nullptr;
        char *preptr = (FULL && a.out_pre) ? reinterpret_cast<char *>(a.out_pre) + p0 * a.Cout * 4 : nullptr;
        const char *resptr = RES ? reinterpret_cast<const char *>(a.res0) + p0 * a.Cout * 4 : nullptr;
        float *clsptr = (FULL && a.cls_w) ? a.cls_out + p0 : nullptr;
        const int rowlim = (U.gx0 + 8 * hw + (r & 7) < a.Wi) ? a.Hi - U.gy0 - 4 * (r >> 3) : 0;
        for (int z = 0; z < a.No; ++z) {
            const bool nofront = z == 0, noback = z == a.No - 1;
            if constexpr (WIDE) {   // one pass per step; the template index is the step's parity (fragment buffers)
                if (first) pass(I0{}, T{}, F{}, F{}, nofront, noback, pptr, ppre, pres, pcls, prowlim);
                else if (par) pass(I1{}, T{}, T{}, T{}, nofront, noback, pptr, ppre, pres, pcls, prowlim);
                else pass(I0{}, T{}, T{}, T{}, nofront, noback, pptr, ppre, pres, pcls, prowlim);
                par ^= 1;
            } else {
                if (first) pass(I0{}, T{}, F{}, F{}, nofront, noback, pptr, ppre, pres, pcls, prowlim);
                else pass(I0{}, T{}, T{}, T{}, nofront, noback, pptr, ppre, pres, pcls, prowlim);
                pass(I1{}, T{}, T{}, T{}, nofront, noback, optr, preptr, resptr, clsptr, rowlim);
            }
            first = false;
            sidx = sidx + 1 == RING ? 0 : sidx + 1;
            advance_fill();
            pptr = optr;
            ppre = preptr;
            pres = resptr;
            pcls = clsptr;
            prowlim = rowlim;
            if (optr) optr += ostride;
            if (FULL && preptr) preptr += ostride;
            if (RES) resptr += ostride;
            if (FULL && clsptr) clsptr += cstride;
        }
    }
    // the last step's second pass is finished past the end of the stream
    pass(I0{}, F{}, T{}, F{}, false, false, pptr, ppre, pres, pcls, prowlim);
    // ... and the slices queued past the end of the stream are still in flight: a wave must not retire before its LDS-DMA has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CIN, int MODE, bool WIDE = false>
__global__ __launch_bounds__((rollt::Lay<CIN, WIDE>::NW * 64)) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rollt(const ConvArgs a, const RollArgs t) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[rollt::Lay<CIN, WIDE>::LDSB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the body is specialised on the wave's role: its sets, tap offsets and accumulator slots are compile-time constants
    if constexpr (CIN == 64) {
        if (wave < 4) {
            if (wave & 1) rollt_body<CIN, rollt::R_B, MODE, false>(a, t, smem, lane, wave);
            else rollt_body<CIN, rollt::R_A, MODE, false>(a, t, smem, lane, wave);
        } else {
            if (wave & 1) rollt_body<CIN, rollt::R_D, MODE, false>(a, t, smem, lane, wave);
            else rollt_body<CIN, rollt::R_C, MODE, false>(a, t, smem, lane, wave);
        }
    } else {
        if (wave & 1) rollt_body<CIN, rollt::R_C32, MODE, WIDE>(a, t, smem, lane, wave);
        else rollt_body<CIN, rollt::R_A32, MODE, WIDE>(a, t, smem, lane, wave);
    }
}

static int rollt_mode(const ConvArgs &a) { return (a.out_pre || a.cls_w) ? 2 : a.res0 ? 1 : 0; }
static bool rollt_wide(const ConvArgs &a) { return a.Cout == 16; }

void rollt_tile(int cout, int *ty, int *tx) {
    *ty = DFFW_ROLLT_TY;
    *tx = cout == 16 ? 2 * DFFW_ROLLT_TX : DFFW_ROLLT_TX;
}

bool rollt_ok(int prec, const ConvArgs &a) {
    if (prec != P_BF16X3 || (a.dbg & DFFW_ARGS_NO_ROLLT)) return false;
    if (a.outf || a.res1 || a.res_bcast || a.relu == 2 || a.C1 != 0 || (a.C0 != 16 && a.C0 != 32 && a.C0 != 64)) return false;
    if (!(((a.Cout == 32 || a.Cout == 64) && a.C0 >= 32) || (a.Cout == 16 && a.C0 <= 32))) return false;
    if (a.No != a.Ni || a.Ho != 2 * a.Hi || a.Wo != 2 * a.Wi) return false;
    const int mode = rollt_mode(a);
    // (the classifier's partial dots are ADDED to the zeroed score volume: one or two addends, so the sum does not depend on their order)
    if (mode == 2 && (!a.res0 || (a.cls_w && (!a.cls_out || a.Cout > 32)))) return false;
    if (mode != 2 && !a.out) return false;
    // 32-bit buffer / lane offsets: a sample's input volume (+ one footprint) and an output slice stay below 2^31 bytes
    return (int64_t)(a.Ni + 1) * a.Hi * a.Wi * a.C0 * 4 < (1ll << 31) && (int64_t)a.Ho * a.Wo * a.Cout * 4 < (1ll << 31);
}

hipError_t launch_conv_rollt(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const bool wide = rollt_wide(a);
    const int nw = wide ? 8 : a.C0 / 8, ny = wide ? 1 : a.Cout / 32;
    const int want = (t.wgs > 0 ? t.wgs : (nw == 8 ? 256 : 512)) / ny;   // 16 waves per CU either way
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8))), (unsigned)ny), block(nw * 64);
    const int mode = rollt_mode(a);
#define DFFW_ROLLT_LAUNCH(CI, MD, WD) hipLaunchKernelGGL((conv_rollt<CI, MD, WD>), grid, block, 0, s, a, t)
    if (wide && a.C0 == 16) {
        if (mode == 2) DFFW_ROLLT_LAUNCH(16, 2, true);
        else if (mode == 1) DFFW_ROLLT_LAUNCH(16, 1, true);
        else DFFW_ROLLT_LAUNCH(16, 0, true);
    } else if (wide) {
        if (mode == 2) DFFW_ROLLT_LAUNCH(32, 2, true);
        else if (mode == 1) DFFW_ROLLT_LAUNCH(32, 1, true);
        else DFFW_ROLLT_LAUNCH(32, 0, true);
    } else if (nw == 8) {
        if (mode == 2) DFFW_ROLLT_LAUNCH(64, 2, false);
        else if (mode == 1) DFFW_ROLLT_LAUNCH(64, 1, false);
        else DFFW_ROLLT_LAUNCH(64, 0, false);
    } else {
        if (mode == 2) DFFW_ROLLT_LAUNCH(32, 2, false);
        else if (mode == 1) DFFW_ROLLT_LAUNCH(32, 1, false);
        else DFFW_ROLLT_LAUNCH(32, 0, false);
    }
#undef DFFW_ROLLT_LAUNCH
    return hipGetLastError();
}

void conv_rollt_kernel_name(const ConvArgs &a, char *buf, int n) {
    snprintf(buf, n, "dffw::conv_rollt<%d, %d, %s>", a.C0, rollt_mode(a), rollt_wide(a) ? "true" : "false");
}

}  // namespace dffw
