// conv_efd16: the EFD block of the 16-channel stage (DEN.py:306-315, `FM_conv2.0`) as ONE persistent streaming kernel (round 6), gfx950 / MI355X.
//     out = relu( BN(conv3x3x3 stride (1,2,2) (x)) + BN(conv3x3x3 (maxpool(1,2,2)(x))) ),   16 -> 32 channels, half resolution
// As two launches (conv_roll_s2 for the strided branch, conv_tile for the pooled one) the strided branch writes its 32-channel result and the pooled branch
// reads it back as a residual: 924 MB at batch 32 for 588 MB of block input + output, the second launch on conv_tile's fill-serial path.  conv_roll_efd
// does the 8 -> 16 block of the stage above in one kernel with both filters in every wave (144 VGPRs); here the two 27 x 16 x 32 split-bf16 filters are
// 221 KB, i.e. the register files of a whole CU, so the eight waves of a workgroup are
//     (branch: strided | pooled) x (16-channel output tile) x (pixel half of an 8 x 8 output column),
// each with ONE branch's filter for ONE output tile resident (3 slices x 5 chunks of (2 in-slice taps x 16 channels), tap 9 = zero weights: 120 VGPRs,
// conv_roll_s2's packing).  The two branches of a pixel are partial sums of the same output: the wave pair (strided, pooled) of an (output tile, pixel
// half) exchanges ONE partial tile per step through LDS (conv_rollt's A / B exchange): of its two operand tiles each wave finishes one -- adds the partner's
// partial (both BatchNorm shifts are in the accumulator inits), ReLU, store -- inside the NEXT step's contraction; the exchange area is double buffered by
// step parity, one barrier per step.
//   * a step = one output slice of the column: 15 chunks x 2 operand tiles (rows j, j + 4 of the column: conv_rollk's conflict-free tile) x 3 products
//     = 90 MFMAs per wave, operand fragments one chunk ahead, chunk 0 of the next step requested in front of the last chunk's MFMAs;
//   * ring of 4 slice images, filled by buffer-addressed LDS-DMA one slice ahead (issued at the top of a step, waited for at its barrier); an image holds the
//     x footprint (17 x 17 pixels, EVEN columns first so that the stride-2 operand reads are contiguous; row pitch 34 entries: input rows 8 apart are a
//     multiple of 256 bytes apart) and, from a DMA-piece boundary on, the pooled footprint (10 x 10, pitch 20): [part][row][pixel][octet] in 16-byte entries;
//   * real slices only: the window at the volume's first / last slice contracts zeroed operands for the chunks of the missing slice (conv_rollk).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "dffw_conv_roll.h"
#include "dffw_device.h"

namespace dffw {

namespace efd16 {
constexpr int TY = 8, TX = 8, NW = 8, RING = 4, NCH = ROLL_CHUNKS;
constexpr int XR = 2 * TY + 1, XP = 2 * (2 * TX + 1);     // x footprint: 17 rows, row pitch 34 entries ([9 even pixels | 8 odd pixels][octet])
constexpr int PR = TY + 2, PP = 2 * (TX + 2);             // pooled footprint: 10 rows, row pitch 20 entries
constexpr int XPARTE = XR * XP, PPARTE = PR * PP;         // entries of one part
constexpr int XE = 2 * XPARTE, PE = 2 * PPARTE;
constexpr int XPIECES = (XE + 63) / 64, PPIECES = (PE + 63) / 64, NPIECE = XPIECES + PPIECES;
constexpr int POFFE = XPIECES * 64;                        // the pooled region starts on a DMA-piece boundary (a piece has one source)
constexpr int SLOTB = NPIECE * 1024, PPW = (NPIECE + NW - 1) / NW;
constexpr int XCH_OFF = RING * SLOTB, XCHB = NW * 1024, LDSB = XCH_OFF + 2 * XCHB;
static_assert((8 * XP) % 16 == 0 && (4 * PP) % 16 == 0, "the two rows of an operand tile are a multiple of 256 bytes apart");
static_assert(LDSB <= 160 * 1024, "LDS layout");
}   // namespace efd16

// BR 0: the strided branch over x (a.in0, (B, N, 2 Ho, 2 Wo, 16)); 1: the stride-1 branch over the pooled volume (a.in1, (B, N, Ho, Wo, 16))
template <int BR>
__device__ __forceinline__ void efd16_body(const ConvArgs &a, const RollArgs &t, unsigned char *smem, const int lane, const int wave) {
    using namespace efd16;
    const int g = lane >> 4, r = lane & 15;
    const int nt = (wave >> 1) & 1, ph = wave >> 2;   // this wave's 16-channel output tile and pixel half (operand tiles 2 ph, 2 ph + 1)

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
    auto decode = [&](int u) {   // 8 x 8 columns of the OUTPUT grid
        Unit c;
        const int txi = u % t.tiles_x;
        const int tt = u / t.tiles_x;
        c.gx0 = txi * TX;
        c.gy0 = (tt % t.tiles_y) * TY;
        c.b = tt / t.tiles_y;
        return c;
    };

    // ---- fill: pieces of 64 consecutive 16-byte entries; pieces below XPIECES come from x, the others from the pooled volume (records [hi 16][lo 16] = 64 bytes);
    // per lane the byte offset from the unit's footprint origin in its source (out-of-image and padding lanes out of range: zeros = the convs' padding)
    const char *tbx = reinterpret_cast<const char *>(a.in0), *tbp = reinterpret_cast<const char *>(a.in1);
    const int xslice_bytes = a.Hi * a.Wi * 64, pslice_bytes = a.Ho * a.Wo * 64;
    int fvo[PPW];
    const char *fbx = tbx, *fbp = tbp;
    int fu = ufirst, fz = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
        fz = 0;
        fbx = tbx + (((int64_t)c.b * a.Ni * a.Hi + (2 * c.gy0 - 1)) * a.Wi + (2 * c.gx0 - 1)) * 64;
        fbp = tbp + (((int64_t)c.b * a.Ni * a.Ho + (c.gy0 - 1)) * a.Wo + (c.gx0 - 1)) * 64;
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // (opaque: no hoisting of the decode out of the unit loop)
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int e = (k * NW + wave) * 64 + ln;
            int off = (int)0x80000000;
            if (e < POFFE) {
                const int part = e / XPARTE, e3 = e - part * XPARTE;
                const int fy = e3 / XP, e4 = e3 - fy * XP;
                const int pos = e4 >> 1, oct = e4 & 1;
                const int fx = pos < TX + 1 ? 2 * pos : 2 * (pos - (TX + 1)) + 1;   // even columns first, then the odd ones
                const int iy = 2 * c.gy0 - 1 + fy, ix = 2 * c.gx0 - 1 + fx;
                if (e < XE && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) off = (fy * a.Wi + fx) * 64 + part * 32 + oct * 16;
            } else {
                const int ep = e - POFFE;
                const int part = ep / PPARTE, e3 = ep - part * PPARTE;
                const int fy = e3 / PP, e4 = e3 - fy * PP;
                const int fx = e4 >> 1, oct = e4 & 1;
                const int iy = c.gy0 - 1 + fy, ix = c.gx0 - 1 + fx;
                if (ep < PE && fx < TX + 2 && (unsigned)iy < (unsigned)a.Ho && (unsigned)ix < (unsigned)a.Wo) off = (fy * a.Wo + fx) * 64 + part * 32 + oct * 16;
            }
            fvo[k] = off;
        }
    };
    setup_fill();
    int fslotb = 0;
    auto issue_piece = [&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const int p = k * NW + wave;
        if (p >= NPIECE) return;                           // (wave-uniform)
        const bool zin = fu < uend;                        // past the end of the stream: zeros (the slot is never read)
        const bool px = p >= XPIECES;                      // (wave-uniform: a piece has one source)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(px ? fbp : fbx), 0, zin ? (int)0x80000000 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + fslotb + p * 1024), 16, fvo[k],
                                                 zin ? fz * (px ? pslice_bytes : xslice_bytes) : 0, 0, 0);
    };
    auto advance_fill = [&]() {
        fslotb = (fslotb + SLOTB == RING * SLOTB) ? 0 : fslotb + SLOTB;
        if (++fz == a.Ni && fu < uend) {
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {                          // the fill runs two slices ahead of the window's centre
        static_for<PPW>([&](auto K) { issue_piece(K); });
        advance_fill();
    }

    // ---- operand addressing.  Chunk c = (window slice c / 5, in-slice taps 2 (c % 5), + 1); K octet g = (tap 2 k5 + (g >> 1), channel octet g & 1); tap 9 carries zero
    // weights (it reads tap 8's operands).  Lane r of operand tile j = output pixel (row j + 4 (r >> 3), column r & 7) = input pixel (2 row + ky, 2 col + kx) of x
    // (even columns first: kx = 0 -> entry col, 1 -> 9 + col, 2 -> col + 1) or (row + ky, col + kx) of the pooled footprint; the tap's offset is a register per k5,
    // window slice, operand tile and part are register / immediates.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    constexpr int ROWE = BR ? PP : 2 * XP;                 // entries per OUTPUT row step
    constexpr int PARTB = (BR ? PPARTE : XPARTE) * 16;
    const unsigned abase = lds0 + (unsigned)(((BR ? POFFE : 0) + (4 * (r >> 3) + 2 * ph) * ROWE + (r & 7) * 2 + (g & 1)) * 16);
    unsigned tapo[5];
#pragma unroll
    for (int k5 = 0; k5 < 5; ++k5) {
        const int tap = 2 * k5 + (g >> 1) < 9 ? 2 * k5 + (g >> 1) : 8;
        const int ky = tap / 3, kx = tap % 3;
        tapo[k5] = (unsigned)((BR ? ky * PP + kx * 2 : ky * XP + (kx == 1 ? 2 * (TX + 1) : kx == 2 ? 2 : 0)) * 16);
    }
    // output: the lane's 16-byte piece (part g & 1 of channel octet 2 nt + (g >> 1)) of pixel (row j + 4 (r >> 3), column r & 7) of the tile this wave finishes:
    // operand tile 2 ph + BR (the strided branch's wave finishes the first tile of the pair, the pooled branch's the second)
    const int town = 2 * ph + BR;
    const int vob = ((town + 4 * (r >> 3)) * a.Wo + (r & 7)) * 64 + (g & 1) * 32 + (nt * 2 + (g >> 1)) * 8;
    // exchange: pair (nt, ph) = wave >> 1; slot [parity][wave]: this wave writes its partial of the tile it does NOT finish into its own slot, its partner reads it
    const unsigned xlane = lds0 + XCH_OFF + lane * 16;
    const unsigned xwr = xlane + wave * 1024, xrd = xlane + (wave ^ 1) * 1024;

    // ---- this branch's filter for this output tile: 15 chunks x (hi, lo), resident for the whole walk ----
    short8 w[NCH][2];
    {
        const short8 *wp = reinterpret_cast<const short8 *>(BR ? t.wroll2 : t.wroll) + (size_t)nt * NCH * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[c][0] = wp[(c * 2 + 0) * 64];
            w[c][1] = wp[(c * 2 + 1) * 64];
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>((BR ? t.bias2 : a.bias) + nt * 16 + g * 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue slices, filter, bias (compiler-visible, so that no later wait is invented)
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(w[c][0]), "+v"(w[c][1]));   // (pinned: never re-loaded in front of an MFMA)
    asm volatile("s_barrier" ::: "memory");

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool relu = a.relu == 1;
    int sidx = RING - 1;                  // ring slot of the window's first slice
    f32x4 mine = zero4;                   // this wave's own partial of the tile it finishes, from the previous step
    short8 x[2][2][2];                    // operand fragments of one chunk: [buffer][operand tile][part]; chunk c of a step of parity PAR sits in buffer (PAR + c) & 1
    auto fetch = [](auto BUF, auto T1, auto PB, short8 (&xx)[2][2][2], const unsigned ad) __attribute__((always_inline)) {
        constexpr int b = decltype(BUF)::value, t1 = decltype(T1)::value, pb = decltype(PB)::value;
        asm volatile("ds_read_b128 %0, %1" : "=v"(xx[b][0][0]) : "v"(ad));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][0][1]) : "v"(ad), "n"(pb));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][0]) : "v"(ad), "n"(t1));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xx[b][1][1]) : "v"(ad), "n"(t1 + pb));
    };
    using T1 = std::integral_constant<int, ROWE * 16>;
    using PB = std::integral_constant<int, PARTB>;

    // One step.  PAR: its parity (fragment buffers, exchange buffer); PEND: finish the tile of the PREVIOUS step (output slice at o_f); PRE: chunk 0 was requested by the
    // step in front.  nofront / noback (wave-uniform): the window's centre is the volume's first / last slice.
    auto step = [&](auto PAR_, auto PEND_, auto PRE_, const bool nofront, const bool noback, char *o_f) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
        constexpr bool PEND = decltype(PEND_)::value, PRE = decltype(PRE_)::value;
        unsigned adw[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) adw[d] = abase + (unsigned)(((sidx + d) & (RING - 1)) * SLOTB);
        if constexpr (!PRE) fetch(std::integral_constant<int, PAR & 1>{}, T1{}, PB{}, x, adw[0] + tapo[0]);
        f32x4 n[2] = {bias4, bias4};
        f32x4 part = zero4;
        static_for<NCH>([&](auto C) __attribute__((always_inline)) {
            constexpr int c = decltype(C)::value;
            constexpr int cur = (PAR + c) & 1, nxt = cur ^ 1;
            // the next chunk's fragments; behind the last chunk: chunk 0 of the next step's window (its slice 0 = this window's slice 1, resident)
            if constexpr (c + 1 < NCH) fetch(std::integral_constant<int, nxt>{}, T1{}, PB{}, x, adw[(c + 1 < NCH ? c + 1 : 0) / 5] + tapo[(c + 1 < NCH ? c + 1 : 0) % 5]);
            else fetch(std::integral_constant<int, nxt>{}, T1{}, PB{}, x, adw[1] + tapo[0]);
            constexpr bool PR = PEND && c == 1;
            if constexpr (PR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(part) : "v"(xrd), "n"((PAR ^ 1) * XCHB));
            // (wait, THEN tie: dffw_conv_slice.hip, conv_slice64_head)
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PR ? 5 : 4));
            asm volatile("" : "+v"(x[cur][0][0]), "+v"(x[cur][0][1]), "+v"(x[cur][1][0]), "+v"(x[cur][1][1]));
            if constexpr (c / 5 != 1) {   // chunks of a slice the volume does not have: zero operands (a uniform branch, taken by two steps per unit)
                if (c / 5 == 0 ? nofront : noback) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) x[cur][j][pt] = short8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
            // product-major over the two accumulators
            n[0] = mma<false>(w[c][1], x[cur][0][0], n[0]);
            n[1] = mma<false>(w[c][1], x[cur][1][0], n[1]);
            n[0] = mma<false>(w[c][0], x[cur][0][1], n[0]);
            n[1] = mma<false>(w[c][0], x[cur][1][1], n[1]);
            n[0] = mma<false>(w[c][0], x[cur][0][0], n[0]);
            n[1] = mma<false>(w[c][0], x[cur][1][0], n[1]);
            __builtin_amdgcn_sched_barrier(0);
            // side work in the chunk gaps: the fill of the ring's free slot, the pending tile's epilogue
            if constexpr (c < PPW) issue_piece(std::integral_constant<int, c < PPW ? c : 0>{});
            if constexpr (PEND && c == 2) asm volatile("" : "+v"(part));   // (chunk 2's wait has passed: the partial requested behind chunk 2's operands has landed)
            if constexpr (PEND && c == 3) {
                const f32x4 vv = mine + part;
                float cls = 0.f;
                epilogue_lean_t<P_BF16X3>(reinterpret_cast<uint16_t *>(o_f), nullptr, vob, vv[0], vv[1], vv[2], vv[3], false, make_uint4(0, 0, 0, 0), relu, false, zero4, cls, true);
            }
            if constexpr (c < PPW || (PEND && (c == 2 || c == 3))) __builtin_amdgcn_sched_barrier(0);
        });
        // hand the partial of the tile the partner finishes over, keep the other.  (The MFMA -> DS wait states: hipcc does not see that an asm blob reads an accumulator.)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(n[BR ? 0 : 1]));
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(xwr), "v"(n[BR ? 0 : 1]), "n"(PAR * XCHB) : "memory");
        mine = n[BR ? 1 : 0];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            constexpr int nb = (PAR + NCH) & 1;
            asm volatile("" : "+v"(x[nb][0][0]), "+v"(x[nb][0][1]), "+v"(x[nb][1][0]), "+v"(x[nb][1][1]));
        }
        sidx = (sidx + 1) & (RING - 1);
        advance_fill();
    };

    using T = std::true_type;
    using F = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const int64_t ostride = (int64_t)a.Ho * a.Wo * 128;   // bytes per output slice
    char *pptr = nullptr;
    bool first = true;
    int par = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        char *optr = reinterpret_cast<char *>(a.out) + ((((int64_t)U.b * a.No) * a.Ho + U.gy0) * a.Wo + U.gx0) * 128;
        for (int z = 0; z < a.No; ++z) {
            const bool nofront = z == 0, noback = z == a.No - 1;
            if (first) step(I0{}, F{}, F{}, nofront, noback, pptr);
            else if (par) step(I1{}, T{}, T{}, nofront, noback, pptr);
            else step(I0{}, T{}, T{}, nofront, noback, pptr);
            first = false;
            par ^= 1;
            pptr = optr;
            optr += ostride;
        }
    }
    // the last step's tile is finished past the end of the stream (its partial sits in exchange buffer par ^ 1)
    {
        f32x4 part;
        if (par) asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(part) : "v"(xrd), "n"(0) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(part) : "v"(xrd), "n"(XCHB) : "memory");
        const f32x4 vv = mine + part;
        float cls = 0.f;
        epilogue_lean_t<P_BF16X3>(reinterpret_cast<uint16_t *>(pptr), nullptr, vob, vv[0], vv[1], vv[2], vv[3], false, make_uint4(0, 0, 0, 0), relu, false, zero4, cls, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_efd16(const ConvArgs a, const RollArgs t) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[efd16::LDSB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave & 1) efd16_body<1>(a, t, smem, lane, wave);
    else efd16_body<0>(a, t, smem, lane, wave);
}

void efd16_tile(int *ty, int *tx) {
    *ty = efd16::TY;
    *tx = efd16::TX;
}

// a = the block's launch: in0 = x (16 channels, (B, N, 2 Ho, 2 Wo)), in1 = its (1,2,2) max-pool, 32 outputs, relu; t.wroll / t.wroll2 = the strided / pooled branch's
// filter in conv_roll_s2's order ([output tile][15 chunks]), a.bias / t.bias2 their BatchNorm shifts
bool efd16_ok(int prec, const ConvArgs &a, const RollArgs &t) {
    if (prec != P_BF16X3 || !a.out || !a.in1 || !t.wroll || !t.wroll2 || !t.bias2) return false;
    if (a.C0 != 16 || a.C1 != 16 || a.Cout != 32 || a.outf || a.out_pre || a.res0 || a.res1 || a.cls_w || a.relu == 2) return false;
    if (a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || a.No != a.Ni || a.Ho % efd16::TY || a.Wo % efd16::TX) return false;
    return (int64_t)(a.Ni + 1) * a.Hi * a.Wi * 64 < (1ll << 31) && (int64_t)a.Ho * a.Wo * 128 < (1ll << 31);
}

hipError_t launch_conv_efd16(const ConvArgs &a, const RollArgs &t, hipStream_t s) {
    const int want = t.wgs > 0 ? t.wgs : 256;   // one 8-wave workgroup per CU
    const int per_xcd = (t.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(efd16::NW * 64);
    hipLaunchKernelGGL(conv_efd16, grid, block, 0, s, a, t);
    return hipGetLastError();
}

void conv_efd16_kernel_name(char *buf, int n) { snprintf(buf, n, "dffw::conv_efd16"); }

}  // namespace dffw
