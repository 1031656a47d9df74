// srd_roll: the whole SRD block of the 8-channel full-resolution stage (DEN.py:317-330 with resnet_block_2d :295-304,
// `FM_measure.Focus_extraction.2`) in ONE persistent kernel.
//
//     t    = relu(BN(conv1x3x3(x)))                       per slice          (Focus_Measure.conv.0)
//     feat = relu(x + BN(conv1x3x3(t)))                   per slice          (Focus_Measure.conv.2)
//     out  = feat + relu(conv1x1x1(relu(conv3x1x1(feat))))  across slices    (N_ch_attention.0 / .2, no BN, no bias)
//     pooled = maxpool(1,2,2)(out)                        (what the following EFD block reads, DEN.py:310)
//
// As three launches every stage is a full pass over the 8-channel volume bound by HBM: x is read twice, t and feat are
// written and read back (7 volume passes, 4.7 GB at batch 32).  Here a workgroup owns a column of 8 x 16 pixels of one
// sample and walks its slices; t and feat never leave LDS:
//   * x slices (12 x 20 footprint: two 3x3 halos) stream through a FIFO of LDS slots by LDS-DMA, several slices ahead
//     (counted vmcnt, raw s_barrier, as in conv_roll);
//   * stage A: conv.0 on the 10 x 18 region conv.2 needs -> t as hi/lo records in LDS (zero outside the image: it is
//     conv.2's padding);  stage B: conv.2 + x + ReLU on the 8 x 16 pixels -> feat in fp32 in a ring of 3 LDS slices;
//     both on the matrix cores with the two filters (2 x 3 chunks) resident in registers;
//   * stage C (two slices behind): the attention over feat[z-1], feat[z], feat[z+1] on the matrix cores as well (as 256
//     fp32 FMAs per pixel on the VALU it was the longest phase of a step): conv3x1x1 = 2 chunks over (pixel of the pair,
//     slice); its ReLU'd result, split to hi/lo in registers, IS the operand block of conv1x1x1 (K octet = the lane's own
//     4 channels hi | lo, filter rows [w_hi w_hi] and [w_lo 0]: two MFMAs, no data movement); + feat[z] (kept in fp32
//     registers by the lane that produced it), split to the storage format, stored; 2x2 max via DPP / permlane for the
//     pooled copy.
// HBM traffic: x once (halo from L2), out once, pooled once.  t and feat are rounded to hi+lo (the storage format) as
// operands, exactly as in the three-launch form.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "dffw_device.h"
#include "dffw_srd_roll.h"

namespace dffw {

// ABL (development only, DFFW_SRD_ABL): timing ablations -- 1 no stage C, 2 no stage A, 4 no stage B, 8 no fill, 16 no barriers, 32 no global stores
template <int PREC, bool POOL, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void srd_roll_kernel(const SrdArgs a, const float *__restrict__ w3g, const float *__restrict__ w1g) {
    // (w3g / w1g = a.w3 / a.w1 as separate read-only parameters: only then does hipcc fetch the attention weights with
    // scalar loads; through the struct it used vector loads inside the loop, and beside LDS-DMA every use of a vector
    // load drains the whole DMA queue.  For the same reason the LDS stores below are inline asm: hipcc puts a
    // vmcnt(0) in front of every ordinary ds_write while an LDS-DMA is in flight.)
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 8, TY = 8, TX = 16, NWAVES = 4;
    constexpr int XY = TY + 4, XX = TX + 4, XPIX = XY * XX;        // x footprint
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;    // region of t that conv.2 needs
    constexpr int PIXB = C * 2;                                    // bytes per pixel per plane
    constexpr int NPIECE = (XPIX + 63) / 64;                       // 1 KiB wave instructions per plane (one 16-byte chunk per pixel)
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int RX = 4;                                          // x FIFO depth
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (counted vmcnt waits)");
    constexpr int TPLANEB = (TPIX * PIXB + 15) / 16 * 16;
    constexpr int FPLANEB = TY * TX * PIXB;                        // one slice of feat as records (even pixels of a row first)
    constexpr int FSLOTB = PARTS * FPLANEB;
    constexpr int X_OFF = 0, T_OFF = RX * SLOTB, F_OFF = T_OFF + PARTS * TPLANEB;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[F_OFF + 3 * FSLOTB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto lds_store8 = [&](unsigned byte_off, uint32_t v0, uint32_t v1) {
        const u32x2 d = {v0, v1};
        asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + byte_off), "v"(d) : "memory");
    };
    auto lds_store16 = [&](unsigned byte_off, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + byte_off), "v"(v) : "memory"); };

    // ---- columns of this workgroup: as conv_roll (XCD-contiguous ranges, round-robin inside the XCD) ----------------
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };

    // ---- x FIFO: the slices of the workgroup's columns as one stream (N per column) ---------------------------------
    const int rec = PARTS * C;                                     // 16-bit elements per pixel record
    const int slice_elems = a.H * a.W * rec;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int pix = i * 64 + lane;
            const int fy = pix / XX, sx = pix - fy * XX;
            const int fx = sx < XX / 2 ? 2 * sx : 2 * (sx - XX / 2) + 1;   // LDS rows hold the even columns first, then the odd ones
            const int iy = c.gy0 - 2 + fy, ix = c.gx0 - 2 + fx;
            fok[k] = p < NP && pix < XPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            fsrc[k] = a.x + (int64_t)c.b * a.N * slice_elems + (int64_t)(iy * a.W + ix) * rec + part * C;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const bool zin = fu < uend;
        unsigned char *slot = smem + X_OFF + fslot * SLOTB;
        const int64_t zo = (int64_t)fq * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RX) ? 0 : fslot + 1;
        if (++fq == a.N && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // ---- per-lane constants ------------------------------------------------------------------------------------------
    // Both convs have 8 output channels: a GEMM column is a PAIR of horizontally adjacent pixels (result rows 0-7 = even
    // pixel, 8-15 = odd pixel) contracting per filter row over the 4 input columns the pair touches (4 x 8 channels = one
    // 32-deep chunk; K octet g = input column 2*pair + g), as in conv_roll's pair form: no dead rows, half the tiles.
    // LDS rows (x and t) keep the even columns first, then the odd ones, so the pairs of a tile read consecutive 16-byte slots.
    // stage A: the 10 x 18 t pixels are 90 pairs = 6 operand tiles (the last one partly idle): waves 2, 3 take two tiles
    // each, waves 0, 1 one each (those two waves also run stage C of an earlier slice in the same phase)
    constexpr int TA = 2;
    const int nA = wave < 2 ? 2 : 1;
    constexpr int APAIRS = TYT * (TXT / 2);
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        const int tile = j == 0 ? wave : 4 + wave;
        int pi = tile * 16 + r;
        ta_ok[j] = pi < APAIRS;
        if (pi >= APAIRS) pi = APAIRS - 1;   // idle columns recompute the last pair, nothing is stored for them
        const int row = pi / (TXT / 2), pc = pi - row * (TXT / 2);
        ta_y[j] = row;
        ta_x[j] = 2 * pc + (g >> 1);        // the t pixel this lane ends up with (channels (g & 1)*4 ..)
        pa[j] = (row * XX + ((g & 1) ? XX / 2 : 0) + pc + (g >> 1)) * PIXB;   // input column 2*pc + g of row `row`
        ta_st[j] = T_OFF + (row * TXT + ((g >> 1) ? TXT / 2 : 0) + pc) * PIXB + (g & 1) * 8;
    }
    // stage B: the 8 x 16 feat pixels are 64 pairs = 4 tiles, one per wave
    const int pb_pi = wave * 16 + r, pb_y = pb_pi / (TX / 2), pb_pc = pb_pi % (TX / 2), pb_x = 2 * pb_pc + (g >> 1);
    const int pbo = (pb_y * TXT + ((g & 1) ? TXT / 2 : 0) + pb_pc + (g >> 1)) * PIXB;
    const int pb_res = ((pb_y + 2) * XX + (((pb_x + 2) & 1) ? XX / 2 : 0) + ((pb_x + 2) >> 1)) * PIXB + (g & 1) * 8;
    const int pb_f = (pb_y * TX + (g >> 1) * (TX / 2) + pb_pc) * PIXB;   // the lane's pixel inside a feat plane
    // attention filters as MFMA A-fragments (pack_conv): conv3x1x1 = 2 chunks, conv1x1x1 = [w_hi w_hi] and [w_lo 0]
    short8 w3f[2][PARTS], w1f[PARTS];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w3f[k][pt] = reinterpret_cast<const short8 *>(a.w3f)[(k * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) w1f[pt] = reinterpret_cast<const short8 *>(a.w1f)[pt * 64 + lane];
    // the two filters as MFMA A-fragments (3 chunks each) and their BatchNorm shifts
    short8 w0[3][PARTS], w2[3][PARTS];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
            w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
        }
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + (g & 1) * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + (g & 1) * 4);
    // one operand tile: 3 chunks (filter rows) x hi/lo, read by inline asm (hipcc degrades every lgkmcnt wait to 0 and adds
    // vmcnt(0) in front of reads of DMA-filled slots while an LDS-DMA is outstanding) and contracted as they arrive
    auto tile_mma = [&](unsigned base, auto rowB_c, auto loB_c, const short8 (&wf)[3][PARTS], f32x4 acc) {
        constexpr int rowB = decltype(rowB_c)::value, loB = decltype(loB_c)::value;   // immediates of the reads: no address arithmetic per chunk
        short8 xh[3], xl[3];
#define DFFW_SRD_READ(k)                                                                                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xh[k]) : "v"(base), "n"(k * rowB));                                               \
    if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k]) : "v"(base), "n"(k * rowB + loB));               \
    else xl[k] = short8{0, 0, 0, 0, 0, 0, 0, 0};   /* (single-part storage: never contracted; NOT a copy of the in-flight hi fragment, tools/isa_wait_lint.py) */
        DFFW_SRD_READ(0)
        DFFW_SRD_READ(1)
        DFFW_SRD_READ(2)
#undef DFFW_SRD_READ
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xh[0]), "+v"(xl[0]) : "n"(2 * PARTS));
            else if (k == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xh[1]), "+v"(xl[1]) : "n"(PARTS));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[2]), "+v"(xl[2]));
            if constexpr (PARTS == 2) {
                acc = mma<F16>(wf[k][1], xh[k], acc);
                acc = mma<F16>(wf[k][0], xl[k], acc);
            }
            acc = mma<F16>(wf[k][0], xh[k], acc);
        }
        return acc;
    };

    constexpr int INFLIGHT = (RX - 2) * PPW;   // slices that may stay in flight when the next one is needed
#pragma unroll
    for (int q = 0; q < RX - 1; ++q) issue_next();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // compiler-visible vmcnt(0): filters, shifts and the first slices
    asm volatile("s_barrier" ::: "memory");

    int xslot = 0;
    f32x4 vq0 = f32x4{0.f, 0.f, 0.f, 0.f}, vq1 = vq0;
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v fop = {0u, 0u, 0u, 0u};
#ifdef DFFW_TRACE_BUILD
    StepTrace trc(a.trace, wave, lane, NWAVES);
#else
    StepTrace trc(nullptr, wave, lane, NWAVES);
#endif
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        // Step s of a column (s = 0 .. N): phase 1 = stage A of slice s (t = conv.0; waves 0-1 two operand tiles, waves 2-3 one); phase 2 = stage B
        // of slice s (feat[s] = conv.2 + x) and, on the same wave right behind it, stage C of slice s-1 (attention out of feat[s-2..s]).
        // A wave's stage B and stage C work on the same 32 pixels, so the feat ring (3 slices as hi/lo records) is wave-private: stage C follows
        // stage B without a barrier, one slice behind, and the column drains in ONE extra step that holds nothing but stage C of the last slice.
        // (Rounds 1-3 ran stage C two slices behind in phase 1, next to stage A: 2 + 1 operand tiles on the critical path of phase 1, one in
        // phase 2, N + 2 steps per column.)
        for (int s = 0; s <= a.N; ++s) {
            const bool produce = s < a.N;
            const unsigned fslot_off = F_OFF + (s % 3) * FSLOTB;
            trc.stamp(0);
            if (produce) {
                // (1) this step's x slice has landed (for every wave after the barrier)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
                if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            trc.stamp(1);
            if (s == 0) {   // feat[-1] = 0 (ring slot 2), the wave's own pixels
                lds_store8(F_OFF + 2 * FSLOTB + pb_f + (g & 1) * 8, 0u, 0u);
                if constexpr (PARTS == 2) lds_store8(F_OFF + 2 * FSLOTB + FPLANEB + pb_f + (g & 1) * 8, 0u, 0u);
            }
            // ---- stage A: t = relu(conv.0(x) + shift) on the 10 x 18 region, zero outside the image ---------------------
            if (produce && !(ABL & 2)) {
                auto store_t = [&](int j, const f32x4 acc) {
                    const int iy = U.gy0 - 1 + ta_y[j], ix = U.gx0 - 1 + ta_x[j];
                    const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    if (ta_ok[j]) {
                        uint32_t h01, h23, l01, l23;
                        Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                        Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                        lds_store8(ta_st[j], h01, h23);
                        if constexpr (PARTS == 2) lds_store8(ta_st[j] + TPLANEB, l01, l23);
                    }
                };
                if (nA == 2) {
                    // two operand tiles: chunk k of the second one is requested as soon as chunk k of the first has been contracted (same registers), so its
                    // reads travel under the first tile's MFMAs and epilogue (DS operations retire in order: 2 chunks' reads behind the one waited for)
                    const unsigned ab0 = lds0 + X_OFF + xslot * SLOTB + pa[0], ab1 = lds0 + X_OFF + xslot * SLOTB + pa[1];
                    short8 ah[3], al[3];
#define DFFW_SRD_RDA(base, k)                                                                                                             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[k]) : "v"(base), "n"(k * XX * PIXB));                                           \
    if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[k]) : "v"(base), "n"(k * XX * PIXB + PLANEB));
#define DFFW_SRD_MMA(k, W)                                 \
    if constexpr (PARTS == 2) {                            \
        acc = mma<F16>(w0[k][1], ah[k], acc);              \
        acc = mma<F16>(w0[k][0], al[k], acc);              \
    }                                                      \
    acc = mma<F16>(w0[k][0], ah[k], acc);
                    DFFW_SRD_RDA(ab0, 0)
                    DFFW_SRD_RDA(ab0, 1)
                    DFFW_SRD_RDA(ab0, 2)
                    f32x4 acc = b0;
                    if constexpr (PARTS == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[0]), "+v"(al[0])); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[0]));
                    DFFW_SRD_MMA(0, w0)
                    DFFW_SRD_RDA(ab1, 0)
                    if constexpr (PARTS == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[1]), "+v"(al[1])); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[1]));
                    DFFW_SRD_MMA(1, w0)
                    DFFW_SRD_RDA(ab1, 1)
                    if constexpr (PARTS == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[2]), "+v"(al[2])); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[2]));
                    DFFW_SRD_MMA(2, w0)
                    DFFW_SRD_RDA(ab1, 2)
                    store_t(0, acc);
                    acc = b0;
                    // (behind the second tile's reads: at most its later chunks -- the first tile's stores come after them and only make the wait longer)
                    if constexpr (PARTS == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[0]), "+v"(al[0])); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[0]));
                    DFFW_SRD_MMA(0, w0)
                    if constexpr (PARTS == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[1]), "+v"(al[1])); else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ah[1]));
                    DFFW_SRD_MMA(1, w0)
                    if constexpr (PARTS == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[2]), "+v"(al[2])); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[2]));
                    DFFW_SRD_MMA(2, w0)
#undef DFFW_SRD_RDA
#undef DFFW_SRD_MMA
                    store_t(1, acc);
                } else {
                    store_t(0, tile_mma(lds0 + X_OFF + xslot * SLOTB + pa[0], std::integral_constant<int, XX * PIXB>{}, std::integral_constant<int, PLANEB>{}, w0, b0));
                }
            }
            trc.stamp(2);
            if (produce) {
                // (2) t complete
                if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                trc.stamp(3);
                // ---- stage B: feat[s] = relu(conv.2(t) + shift + x) ----------------------------------------------------
                if constexpr (!(ABL & 4)) {
                    const f32x4 acc = tile_mma(lds0 + T_OFF + pbo, std::integral_constant<int, TXT * PIXB>{}, std::integral_constant<int, TPLANEB>{}, w2, b2);
                    const unsigned xp = lds0 + X_OFF + xslot * SLOTB + pb_res;
                    u32x2 xh, xl = {0u, 0u};
                    asm volatile("ds_read_b64 %0, %1" : "=v"(xh) : "v"(xp));
                    if constexpr (PARTS == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xl) : "v"(xp), "n"(PLANEB));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh), "+v"(xl));
                    float r0, r1, r2, r3;
                    Fmt<PREC>::join2(xh[0], xl[0], r0, r1);
                    Fmt<PREC>::join2(xh[1], xl[1], r2, r3);
                    f32x4 v;
                    v[0] = relu_bits(acc[0] + r0);
                    v[1] = relu_bits(acc[1] + r1);
                    v[2] = relu_bits(acc[2] + r2);
                    v[3] = relu_bits(acc[3] + r3);
                    uint32_t fh01, fh23, fl01, fl23;
                    Fmt<PREC>::split2(v[0], v[1], fh01, fl01);
                    Fmt<PREC>::split2(v[2], v[3], fh23, fl23);
                    lds_store8(fslot_off + pb_f + (g & 1) * 8, fh01, fh23);
                    if constexpr (PARTS == 2) lds_store8(fslot_off + FPLANEB + pb_f + (g & 1) * 8, fl01, fl23);
                    fop = u32x4v{fh01, fh23, fl01, fl23};   // feat[s] of the lane's own 4 channels as [hi x4 | lo x4]: stage C's K octet for the slice behind z
                    vq1 = vq0;   // fp32 feat of this lane's pixel / channels, two steps deep: stage C adds it back
                    vq0 = v;
                }
            } else {
                vq1 = vq0;
                fop = u32x4v{0u, 0u, 0u, 0u};   // feat[N] = 0
            }
            trc.stamp(4);
            // ---- stage C: attention for slice z = s-1 out of feat[z-1], feat[z], feat[z+1] = the slice stage B just wrote (or the zeros behind the
            // last one); wave w = pairs of rows 2w, 2w+1 = the pixels ITS stage B produced: feat never crosses waves, no barrier in between ----
            if (s >= 1 && !(ABL & 1)) {
                const int z = s - 1;
                const unsigned sm = F_OFF + ((z + 2) % 3) * FSLOTB, sc = F_OFF + (z % 3) * FSLOTB;
                // chunk 0: K octet g = (pixel g >> 1 of the pair, slice z-1 + (g & 1)) out of the ring; chunk 1: K octet g = the lane's own 4 channels of
                // feat[z+1] as [hi x4 | lo x4], straight from stage B's registers (filter fragments [w_hi w_hi] and [w_lo 0] as for the 1x1x1 conv: two MFMAs)
                const unsigned ad0 = lds0 + ((g & 1) ? sc : sm) + pb_f;
                short8 fh0, fl0;
                asm volatile("ds_read_b128 %0, %1" : "=v"(fh0) : "v"(ad0));
                if constexpr (PARTS == 2) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl0) : "v"(ad0), "n"(FPLANEB));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh0), "+v"(fl0));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh0));
                }
                const short8 f1op = __builtin_bit_cast(short8, fop);
                f32x4 at = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (PARTS == 2) {
                    at = mma<F16>(w3f[1][1], f1op, at);
                    at = mma<F16>(w3f[0][1], fh0, at);
                    at = mma<F16>(w3f[0][0], fl0, at);
                }
                at = mma<F16>(w3f[1][0], f1op, at);
                at = mma<F16>(w3f[0][0], fh0, at);
                // ReLU, split: {hi of the lane's 4 channels | lo of them} is the lane's K octet of the 1x1x1 conv
                uint32_t ah01, ah23, al01, al23;
                Fmt<PREC>::split2(relu_bits(at[0]), relu_bits(at[1]), ah01, al01);
                Fmt<PREC>::split2(relu_bits(at[2]), relu_bits(at[3]), ah23, al23);
                const u32x4v bq = {ah01, ah23, al01, al23};
                const short8 b2op = __builtin_bit_cast(short8, bq);
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (PARTS == 2) o = mma<F16>(w1f[1], b2op, o);
                o = mma<F16>(w1f[0], b2op, o);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = vq1[i] + relu_bits(o[i]);
                uint32_t h01, h23, l01, l23;
                Fmt<PREC>::split2(v[0], v[1], h01, l01);
                Fmt<PREC>::split2(v[2], v[3], h23, l23);
                const int64_t pix = (((int64_t)U.b * a.N + z) * a.H + U.gy0 + pb_y) * a.W + U.gx0 + pb_x;
                if constexpr (POOL) {   // max over the 2x2 block of the values as stored: other pixel of the pair = lane rows g ^ 2, other row = column r ^ 8
                    float m[4];
                    Fmt<PREC>::join2(h01, l01, m[0], m[1]);
                    Fmt<PREC>::join2(h23, l23, m[2], m[3]);
                    uint32_t ph01, ph23, pl01, pl23;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {   // the values are sums of two ReLU results: non-negative, so the maxima are taken on the bit patterns (no canonicalising v_max_f32)
                        uint32_t mu = __float_as_uint(m[i]);
                        const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                        mu = max(mu, lane < 32 ? sw[1] : sw[0]);
                        mu = max(mu, (uint32_t)__builtin_amdgcn_mov_dpp((int)mu, 0x128, 0xF, 0xF, true));   // row_ror:8
                        m[i] = __uint_as_float(mu);
                    }
                    Fmt<PREC>::split2(m[0], m[1], ph01, pl01);
                    Fmt<PREC>::split2(m[2], m[3], ph23, pl23);
                    if constexpr (PARTS == 2) {
                        swap16(ph01, pl01);
                        swap16(ph23, pl23);
                    }
                    if (g < 2 && r < 8 && !(ABL & 32)) {
                        const int64_t pp = (((int64_t)U.b * a.N + z) * (a.H / 2) + (U.gy0 / 2 + wave)) * (a.W / 2) + U.gx0 / 2 + pb_pc;
                        if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(a.pooled + pp * rec + (g & 1) * C) = make_uint4(ph01, ph23, pl01, pl23);
                        else *reinterpret_cast<uint2 *>(a.pooled + pp * rec + (g & 1) * 4) = make_uint2(ph01, ph23);
                    }
                }
                if constexpr (PARTS == 2) {
                    swap16(h01, l01);
                    swap16(h23, l23);
                    if ((ABL & 32) == 0 || h01 == 0x12345u) *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C) = make_uint4(h01, h23, l01, l23);
                } else {
                    *reinterpret_cast<uint2 *>(a.out + pix * rec + (g & 1) * 4) = make_uint2(h01, h23);
                }
            }
            trc.stamp(5);
            if (produce) {
                // (3) everyone's reads of x and t are done: the x slot is free, queue the slice RX-1 ahead into it
                if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                trc.stamp(6);
                if constexpr (!(ABL & 8)) issue_next();
                xslot = (xslot + 1 == RX) ? 0 : xslot + 1;
            }
            trc.next();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave
}

// ---- of_roll8: the 8-channel stride-1 residual blocks of the alignment network (`OF_feature.0`, `OF_feature.1`, full resolution) --
// As of_roll_kernel, in srd_roll_kernel's pixel-pair form (8 output channels): stage A = conv.0 -> t in LDS, stage B = conv.2 over
// t (3 chunks) + one chunk for the 1x1x1 shortcut (K octet 0 = the even pixel's 8 input channels, octet 1 = the odd pixel's), ReLU,
// stores.  a.w2 = conv.2 as 3 pair-form chunks + the shortcut chunk (pack_conv).
template <int PREC>
__global__ __launch_bounds__(256) void of_roll8_kernel(const SrdArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 8, TY = 8, TX = 16, NWAVES = 4;
    constexpr int XY = TY + 4, XX = TX + 4, XPIX = XY * XX;        // x footprint
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;    // region of t that conv.2 needs
    constexpr int PIXB = C * 2;                                    // bytes per pixel per plane
    constexpr int NPIECE = (XPIX + 63) / 64;                       // 1 KiB wave instructions per plane (one 16-byte chunk per pixel)
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int RX = 4;                                          // x FIFO depth
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (counted vmcnt waits)");
    constexpr int TPLANEB = (TPIX * PIXB + 15) / 16 * 16;
    constexpr int X_OFF = 0, T_OFF = RX * SLOTB;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[T_OFF + PARTS * TPLANEB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto lds_store8 = [&](unsigned byte_off, uint32_t v0, uint32_t v1) {
        const u32x2 d = {v0, v1};
        asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + byte_off), "v"(d) : "memory");
    };

    // ---- columns of this workgroup: as conv_roll (XCD-contiguous ranges, round-robin inside the XCD) ----------------
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };

    // ---- x FIFO: the slices of the workgroup's columns as one stream (N per column) ---------------------------------
    const int rec = PARTS * C;                                     // 16-bit elements per pixel record
    const int slice_elems = a.H * a.W * rec;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int pix = i * 64 + lane;
            const int fy = pix / XX, sx = pix - fy * XX;
            const int fx = sx < XX / 2 ? 2 * sx : 2 * (sx - XX / 2) + 1;   // LDS rows hold the even columns first, then the odd ones
            const int iy = c.gy0 - 2 + fy, ix = c.gx0 - 2 + fx;
            fok[k] = p < NP && pix < XPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            fsrc[k] = a.x + (int64_t)c.b * a.N * slice_elems + (int64_t)(iy * a.W + ix) * rec + part * C;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const bool zin = fu < uend;
        unsigned char *slot = smem + X_OFF + fslot * SLOTB;
        const int64_t zo = (int64_t)fq * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RX) ? 0 : fslot + 1;
        if (++fq == a.N && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // ---- per-lane constants ------------------------------------------------------------------------------------------
    // Both convs have 8 output channels: a GEMM column is a PAIR of horizontally adjacent pixels (result rows 0-7 = even
    // pixel, 8-15 = odd pixel) contracting per filter row over the 4 input columns the pair touches (4 x 8 channels = one
    // 32-deep chunk; K octet g = input column 2*pair + g), as in conv_roll's pair form: no dead rows, half the tiles.
    // LDS rows (x and t) keep the even columns first, then the odd ones, so the pairs of a tile read consecutive 16-byte slots.
    // stage A: the 10 x 18 t pixels are 90 pairs = 6 operand tiles (the last one partly idle): waves 2, 3 take two tiles
    // each, waves 0, 1 one each (those two waves also run stage C of an earlier slice in the same phase)
    constexpr int TA = 2;
    const int nA = wave < 2 ? 2 : 1;
    constexpr int APAIRS = TYT * (TXT / 2);
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        const int tile = j == 0 ? wave : 4 + wave;
        int pi = tile * 16 + r;
        ta_ok[j] = pi < APAIRS;
        if (pi >= APAIRS) pi = APAIRS - 1;   // idle columns recompute the last pair, nothing is stored for them
        const int row = pi / (TXT / 2), pc = pi - row * (TXT / 2);
        ta_y[j] = row;
        ta_x[j] = 2 * pc + (g >> 1);        // the t pixel this lane ends up with (channels (g & 1)*4 ..)
        pa[j] = (row * XX + ((g & 1) ? XX / 2 : 0) + pc + (g >> 1)) * PIXB;   // input column 2*pc + g of row `row`
        ta_st[j] = T_OFF + (row * TXT + ((g >> 1) ? TXT / 2 : 0) + pc) * PIXB + (g & 1) * 8;
    }
    // stage B: the 8 x 16 feat pixels are 64 pairs = 4 tiles, one per wave
    const int pb_pi = wave * 16 + r, pb_y = pb_pi / (TX / 2), pb_pc = pb_pi % (TX / 2), pb_x = 2 * pb_pc + (g >> 1);
    const int pbo = (pb_y * TXT + ((g & 1) ? TXT / 2 : 0) + pb_pc + (g >> 1)) * PIXB;
    // shortcut operand: K octet g < 2 = the 8 input channels of pixel 2*pair + g at the centre tap (octets 2, 3: zero weights)
    const int pb_sc = ((pb_y + 2) * XX + ((g & 1) ? XX / 2 : 0) + pb_pc + 1) * PIXB;
    // the two filters as MFMA A-fragments (3 chunks each) and their BatchNorm shifts
    short8 w0[3][PARTS], w2[3][PARTS], wsc[PARTS];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
            w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
        }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) wsc[pt] = reinterpret_cast<const short8 *>(a.w2)[(3 * PARTS + pt) * 64 + lane];
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + (g & 1) * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + (g & 1) * 4);
    // one operand tile: 3 chunks (filter rows) x hi/lo, read by inline asm (hipcc degrades every lgkmcnt wait to 0 and adds
    // vmcnt(0) in front of reads of DMA-filled slots while an LDS-DMA is outstanding) and contracted as they arrive
    auto tile_mma = [&](unsigned base, auto rowB_c, auto loB_c, const short8 (&wf)[3][PARTS], f32x4 acc) {
        constexpr int rowB = decltype(rowB_c)::value, loB = decltype(loB_c)::value;   // immediates of the reads: no address arithmetic per chunk
        short8 xh[3], xl[3];
#define DFFW_SRD_READ(k)                                                                                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xh[k]) : "v"(base), "n"(k * rowB));                                               \
    if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k]) : "v"(base), "n"(k * rowB + loB));               \
    else xl[k] = short8{0, 0, 0, 0, 0, 0, 0, 0};   /* (single-part storage: never contracted; NOT a copy of the in-flight hi fragment, tools/isa_wait_lint.py) */
        DFFW_SRD_READ(0)
        DFFW_SRD_READ(1)
        DFFW_SRD_READ(2)
#undef DFFW_SRD_READ
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xh[0]), "+v"(xl[0]) : "n"(2 * PARTS));
            else if (k == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xh[1]), "+v"(xl[1]) : "n"(PARTS));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[2]), "+v"(xl[2]));
            if constexpr (PARTS == 2) {
                acc = mma<F16>(wf[k][1], xh[k], acc);
                acc = mma<F16>(wf[k][0], xl[k], acc);
            }
            acc = mma<F16>(wf[k][0], xh[k], acc);
        }
        return acc;
    };

    constexpr int INFLIGHT = (RX - 2) * PPW;   // slices that may stay in flight when the next one is needed
#pragma unroll
    for (int q = 0; q < RX - 1; ++q) issue_next();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // compiler-visible vmcnt(0): filters, shifts and the first slices
    asm volatile("s_barrier" ::: "memory");

    int xslot = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        for (int s = 0; s < a.N; ++s) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
            // ---- stage A: t = relu(conv.0(x) + shift) on the 10 x 18 region, zero outside the image --------------------------------
#pragma unroll
            for (int j = 0; j < TA; ++j) {
                if (j >= nA) break;
                const f32x4 acc = tile_mma(lds0 + X_OFF + xslot * SLOTB + pa[j], std::integral_constant<int, XX * PIXB>{}, std::integral_constant<int, PLANEB>{}, w0, b0);
                const int iy = U.gy0 - 1 + ta_y[j], ix = U.gx0 - 1 + ta_x[j];
                const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                if (ta_ok[j]) {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                    Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                    lds_store8(ta_st[j], h01, h23);
                    if constexpr (PARTS == 2) lds_store8(ta_st[j] + TPLANEB, l01, l23);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // ---- stage B: out = relu(conv.2(t) + shift + shortcut(x)) -------------------------------------------------------------
            {
                f32x4 acc = tile_mma(lds0 + T_OFF + pbo, std::integral_constant<int, TXT * PIXB>{}, std::integral_constant<int, TPLANEB>{}, w2, b2);
                const unsigned xp = lds0 + X_OFF + xslot * SLOTB + pb_sc;
                short8 sh, sl;
                asm volatile("ds_read_b128 %0, %1" : "=v"(sh) : "v"(xp));
                if constexpr (PARTS == 2) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sl) : "v"(xp), "n"(PLANEB));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sh), "+v"(sl));
                    acc = mma<F16>(wsc[1], sh, acc);
                    acc = mma<F16>(wsc[0], sl, acc);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sh));
                }
                acc = mma<F16>(wsc[0], sh, acc);
                uint32_t h01, h23, l01, l23;
                Fmt<PREC>::split2(relu_bits(acc[0]), relu_bits(acc[1]), h01, l01);
                Fmt<PREC>::split2(relu_bits(acc[2]), relu_bits(acc[3]), h23, l23);
                const int64_t pix = (((int64_t)U.b * a.N + s) * a.H + U.gy0 + pb_y) * a.W + U.gx0 + pb_x;
                if constexpr (PARTS == 2) {
                    swap16(h01, l01);
                    swap16(h23, l23);
                    *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C) = make_uint4(h01, h23, l01, l23);
                } else {
                    *reinterpret_cast<uint2 *>(a.out + pix * rec + (g & 1) * 4) = make_uint2(h01, h23);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_next();
            xslot = (xslot + 1 == RX) ? 0 : xslot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave
}

// ---- of_first: the first residual block of the alignment network (`OF_feature.0`, 3 -> 8 channels) straight from the fp32 stack ---
// of_roll8's arithmetic (pixel-pair form, same filter packing, same operation order: bit-identical results) with the block input
// taken from the planar fp32 focal stack (B,3,N,H,W) instead of the 8-channel record volume: thread p < 240 owns pixel p of the
// 12 x 20 footprint, requests its three colour values for slice s+1 before the contraction of slice s, splits them afterwards into the
// record [c0 c1 c2 0 0 0 0 0] (what from_ncdhw_pad wrote) and stores it into one of two LDS slots, even columns of a row first.  Saves
// the record volume's write and read (32 B per pixel each; the stack is 12 B per pixel).  Plain loads only: hipcc counts every wait.
template <int PREC>
__global__ __launch_bounds__(256) void of_first_kernel(const SrdArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 8, TY = 8, TX = 16;
    constexpr int XY = TY + 4, XX = TX + 4, XPIX = XY * XX;        // x footprint
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;    // region of t that conv.2 needs
    constexpr int PIXB = C * 2;
    constexpr int PLANEB = XPIX * PIXB, SLOTB = PARTS * PLANEB;
    constexpr int TPLANEB = (TPIX * PIXB + 15) / 16 * 16;
    constexpr int T_OFF = 2 * SLOTB;
    static_assert(XPIX <= 256, "one footprint pixel per thread");
    __shared__ __attribute__((aligned(16))) unsigned char smem[T_OFF + PARTS * TPLANEB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };
    const int rec = PARTS * C;
    const float *FS = a.w3;                                         // the fp32 focal stack (B,3,N,H,W)
    const int64_t plane = (int64_t)a.H * a.W, cplane = (int64_t)a.N * plane;

    // ---- fill side ---------------------------------------------------------------------------------------------------
    const bool gth = tid < XPIX;
    const int fy = tid / XX, fx = tid - fy * XX;
    const int lpos = (fy * XX + ((fx & 1) ? XX / 2 + (fx >> 1) : (fx >> 1))) * PIXB;   // even columns of a row first
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    auto issue = [&](const Unit &U, int n) {
        const int iy = U.gy0 - 2 + fy, ix = U.gx0 - 2 + fx;
        c0 = c1 = c2 = 0.f;
        if (gth && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const float *sp = FS + (int64_t)U.b * 3 * cplane + (int64_t)n * plane + (int64_t)iy * a.W + ix;
            c0 = sp[0];
            c1 = sp[cplane];
            c2 = sp[2 * cplane];
        }
    };
    auto land = [&](int slot) {
        if (!gth) return;
        uint4 h = make_uint4(0, 0, 0, 0), l = h;
        Fmt<PREC>::split2(c0, c1, h.x, l.x);
        Fmt<PREC>::split2(c2, 0.f, h.y, l.y);
        *reinterpret_cast<uint4 *>(smem + slot * SLOTB + lpos) = h;
        if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(smem + slot * SLOTB + PLANEB + lpos) = l;
    };

    // ---- per-lane constants (of_roll8's) --------------------------------------------------------------------------------
    constexpr int TA = 2;
    const int nA = wave < 2 ? 2 : 1;
    constexpr int APAIRS = TYT * (TXT / 2);
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        const int tile = j == 0 ? wave : 4 + wave;
        int pi = tile * 16 + r;
        ta_ok[j] = pi < APAIRS;
        if (pi >= APAIRS) pi = APAIRS - 1;
        const int row = pi / (TXT / 2), pc = pi - row * (TXT / 2);
        ta_y[j] = row;
        ta_x[j] = 2 * pc + (g >> 1);
        pa[j] = (row * XX + ((g & 1) ? XX / 2 : 0) + pc + (g >> 1)) * PIXB;
        ta_st[j] = T_OFF + (row * TXT + ((g >> 1) ? TXT / 2 : 0) + pc) * PIXB + (g & 1) * 8;
    }
    const int pb_pi = wave * 16 + r, pb_y = pb_pi / (TX / 2), pb_pc = pb_pi % (TX / 2), pb_x = 2 * pb_pc + (g >> 1);
    const int pbo = (pb_y * TXT + ((g & 1) ? TXT / 2 : 0) + pb_pc + (g >> 1)) * PIXB;
    const int pb_sc = ((pb_y + 2) * XX + ((g & 1) ? XX / 2 : 0) + pb_pc + 1) * PIXB;
    short8 w0[3][PARTS], w2[3][PARTS], wsc[PARTS];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
            w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
        }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) wsc[pt] = reinterpret_cast<const short8 *>(a.w2)[(3 * PARTS + pt) * 64 + lane];
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + (g & 1) * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + (g & 1) * 4);
    auto tile_mma = [&](const unsigned char *base, int rowB, int loB, const short8 (&wf)[3][PARTS], f32x4 acc) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const short8 xh = *reinterpret_cast<const short8 *>(base + k * rowB);
            if constexpr (PARTS == 2) {
                const short8 xl = *reinterpret_cast<const short8 *>(base + k * rowB + loB);
                acc = mma<F16>(wf[k][1], xh, acc);
                acc = mma<F16>(wf[k][0], xl, acc);
            }
            acc = mma<F16>(wf[k][0], xh, acc);
        }
        return acc;
    };

    Unit U = decode(ufirst);
    issue(U, 0);
    int slot = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit Ucur = U;
        for (int s = 0; s < a.N; ++s) {
            land(slot);
            const bool more = s + 1 < a.N || cu + wgs_per_xcd < uend;
            if (s + 1 == a.N && more) U = decode(cu + wgs_per_xcd);
            if (more) issue(U, s + 1 < a.N ? s + 1 : 0);
            __syncthreads();
            const unsigned char *xs = smem + slot * SLOTB;
            // ---- stage A: t = relu(conv.0(x) + shift) on the 10 x 18 region, zero outside the image ----------------------------
#pragma unroll
            for (int j = 0; j < TA; ++j) {
                if (j >= nA) break;
                const f32x4 acc = tile_mma(xs + pa[j], XX * PIXB, PLANEB, w0, b0);
                const int iy = Ucur.gy0 - 1 + ta_y[j], ix = Ucur.gx0 - 1 + ta_x[j];
                const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                if (ta_ok[j]) {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                    Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                    *reinterpret_cast<uint2 *>(smem + ta_st[j]) = make_uint2(h01, h23);
                    if constexpr (PARTS == 2) *reinterpret_cast<uint2 *>(smem + ta_st[j] + TPLANEB) = make_uint2(l01, l23);
                }
            }
            __syncthreads();
            // ---- stage B: out = relu(conv.2(t) + shift + shortcut(x)) -------------------------------------------------------------
            {
                f32x4 acc = tile_mma(smem + T_OFF + pbo, TXT * PIXB, TPLANEB, w2, b2);
                const short8 sh = *reinterpret_cast<const short8 *>(xs + pb_sc);
                if constexpr (PARTS == 2) {
                    const short8 sl = *reinterpret_cast<const short8 *>(xs + PLANEB + pb_sc);
                    acc = mma<F16>(wsc[1], sh, acc);
                    acc = mma<F16>(wsc[0], sl, acc);
                }
                acc = mma<F16>(wsc[0], sh, acc);
                uint32_t h01, h23, l01, l23;
                Fmt<PREC>::split2(relu_bits(acc[0]), relu_bits(acc[1]), h01, l01);
                Fmt<PREC>::split2(relu_bits(acc[2]), relu_bits(acc[3]), h23, l23);
                const int64_t pix = (((int64_t)Ucur.b * a.N + s) * a.H + Ucur.gy0 + pb_y) * a.W + Ucur.gx0 + pb_x;
                if constexpr (PARTS == 2) {
                    swap16(h01, l01);
                    swap16(h23, l23);
                    *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C) = make_uint4(h01, h23, l01, l23);
                } else {
                    *reinterpret_cast<uint2 *>(a.out + pix * rec + (g & 1) * 4) = make_uint2(h01, h23);
                }
            }
            slot ^= 1;
        }
    }
}

void of_first_kernel_name(int prec, char *buf, int n) { snprintf(buf, n, "dffw::of_first_kernel<%d>", prec); }

hipError_t launch_of_first(int prec, const SrdArgs &a, hipStream_t s) {
    const int want = a.wgs > 0 ? a.wgs : 1024;
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
    switch (prec) {
        case P_BF16X3: hipLaunchKernelGGL((of_first_kernel<P_BF16X3>), grid, block, 0, s, a); break;
        case P_FP16: hipLaunchKernelGGL((of_first_kernel<P_FP16>), grid, block, 0, s, a); break;
        case P_BF16: hipLaunchKernelGGL((of_first_kernel<P_BF16>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- srd_roll16: the same block for the 16-channel half-resolution stage (`FM_conv1.1`) -----------------------------------
// 16 output channels fill the MFMA result rows, so no pixel pairs: a GEMM column is one pixel, the 1x3x3 convs contract over
// 5 chunks of (2 taps x 16 channels) (tap 9 = zeros), records are 32 bytes per plane (natural column order).  Columns are
// 4 x 16 pixels (LDS: 4 x-slices of 8 x 20 pixels + t + the feat ring = 67 KB, two workgroups per CU).  Stage B / C tiles
// are 2 rows x 8 pixels per wave so that the 2x2 max-pool stays inside a wave.  Streaming skeleton, counted waits, inline-asm
// LDS access and the MFMA attention (its split result = the 1x1x1 conv's operand in place) as in srd_roll_kernel.
// ABL (development only, DFFW_SRD_ABL): timing ablations -- 1 no stage C, 2 no stage A, 4 no stage B, 8 no fill, 16 no barriers, 32 no global stores
template <int PREC, bool POOL, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void srd_roll16_kernel(const SrdArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 16, TY = 4, TX = 16, NWAVES = 4;
    constexpr int XY = TY + 4, XX = TX + 4, XPIX = XY * XX;
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;
    // t rows at a pitch of 24 pixels = 768 bytes: a stage-B tile is 2 rows x 8 pixels, and the 16 lanes of a ds_read_b128 row group only cover 16 distinct
    // 16-byte bank groups when its two rows are a multiple of 256 bytes apart (18-pixel rows, 576 bytes: two-way conflicts on every stage-B read)
    constexpr int TXTP = 24;
    constexpr int PIXB = C * 2;
    constexpr int NPIECE = 6;                                      // 1 KiB wave instructions per plane (5 hold the 160 pixels; 6 keeps 3 per wave)
    static_assert(NPIECE * 32 >= XPIX, "plane holds the footprint");
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int RX = 4;
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (counted vmcnt waits)");
    constexpr int TPLANEB = TYT * TXTP * PIXB;
    constexpr int FPLANEB = TY * TX * PIXB;
    constexpr int FSLOTB = PARTS * FPLANEB;
    constexpr int X_OFF = 0, T_OFF = RX * SLOTB, F_OFF = T_OFF + PARTS * TPLANEB;
    constexpr int NCH = 5;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[F_OFF + 3 * FSLOTB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto lds_store8 = [&](unsigned byte_off, uint32_t v0, uint32_t v1) {
        const u32x2 d = {v0, v1};
        asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + byte_off), "v"(d) : "memory");
    };
    auto lds_store16 = [&](unsigned byte_off, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + byte_off), "v"(v) : "memory"); };

    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };

    const int rec = PARTS * C;
    const int slice_elems = a.H * a.W * rec;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, pix = ci >> 1, oct = ci & 1;
            const int fy = pix / XX, fx = pix - fy * XX;
            const int iy = c.gy0 - 2 + fy, ix = c.gx0 - 2 + fx;
            fok[k] = p < NP && pix < XPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            fsrc[k] = a.x + (int64_t)c.b * a.N * slice_elems + (int64_t)(iy * a.W + ix) * rec + part * C + oct * 8;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const bool zin = fu < uend;
        unsigned char *slot = smem + X_OFF + fslot * SLOTB;
        const int64_t zo = (int64_t)fq * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RX) ? 0 : fslot + 1;
        if (++fq == a.N && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // stage A: the 6 x 18 t pixels are 7 operand tiles: tiles 0-5 = the first 16 pixels of row 0-5 (16 consecutive pixels of ONE row: conflict-free
    // operand reads; tiles of 16 consecutive indices of the 6 x 18 region wrapped rows and collided two ways), tile 6 = the two remaining pixels of
    // each row (12 of its 16 lanes busy); waves 0-2 take two tiles, wave 3 one
    constexpr int TA = 2;
    const int nA = wave < 3 ? 2 : 1;
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        const int tile = j == 0 ? wave : 4 + wave;
        ta_ok[j] = tile < TYT || r < 2 * TYT;
        ta_y[j] = tile < TYT ? tile : (r < 2 * TYT ? r >> 1 : TYT - 1);
        ta_x[j] = tile < TYT ? r : TX + (r & 1);
        pa[j] = (ta_y[j] * XX + ta_x[j]) * PIXB + (g & 1) * 16;
        ta_st[j] = T_OFF + (ta_y[j] * TXTP + ta_x[j]) * PIXB + g * 8;
    }
    // K octet g of chunk k = (filter tap 2k + (g >> 1), channel octet g & 1); tap 9 carries zero weights
    int tapA[NCH], tapB[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int tap = 2 * k + (g >> 1);
        const int dy = tap < 9 ? tap / 3 : 0, dx = tap < 9 ? tap % 3 : 0;
        tapA[k] = (dy * XX + dx) * PIXB;
        tapB[k] = (dy * TXTP + dx) * PIXB;
    }
    // stage B / C: wave w = rows 2*(w >> 1), +1 x columns 8*(w & 1) .. +7 (the 2x2 pooling blocks stay inside the wave)
    const int pb_y = 2 * (wave >> 1) + (r >> 3), pb_x = 8 * (wave & 1) + (r & 7);
    const int pbo = (pb_y * TXTP + pb_x) * PIXB + (g & 1) * 16;
    const int pb_res = ((pb_y + 2) * XX + pb_x + 2) * PIXB + g * 8;
    const int pb_f = (pb_y * TX + pb_x) * PIXB;
    short8 w0[NCH][PARTS], w2[NCH][PARTS], w3f[2][PARTS], w1f[PARTS];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
            w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
        }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w3f[k][pt] = reinterpret_cast<const short8 *>(a.w3f)[(k * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) w1f[pt] = reinterpret_cast<const short8 *>(a.w1f)[pt * 64 + lane];
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + g * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + g * 4);
    auto tile_mma = [&](unsigned base, const int (&tapo)[NCH], auto loB_c, const short8 (&wf)[NCH][PARTS], f32x4 acc) {
        constexpr int loB = decltype(loB_c)::value;   // the lo plane as an immediate of the read
        short8 xh[NCH], xl[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const unsigned ad = base + tapo[k];
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[k]) : "v"(ad));
            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k]) : "v"(ad), "n"(loB));
            else xl[k] = short8{0, 0, 0, 0, 0, 0, 0, 0};   /* (single-part storage: never contracted; NOT a copy of the in-flight hi fragment, tools/isa_wait_lint.py) */
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int left = (NCH - 1 - k) * PARTS;   // reads still allowed in flight (compile-time after unrolling)
            if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xh[k]), "+v"(xl[k]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[k]), "+v"(xl[k]));
            if constexpr (PARTS == 2) {
                acc = mma<F16>(wf[k][1], xh[k], acc);
                acc = mma<F16>(wf[k][0], xl[k], acc);
            }
            acc = mma<F16>(wf[k][0], xh[k], acc);
        }
        return acc;
    };

    // two operand tiles side by side (stage A of the waves that own two): the reads of both run three chunks ahead of the MFMAs and the two
    // accumulator chains alternate, so one tile's LDS latency and MFMA dependency gaps are covered by the other's work
    auto tile_mma2 = [&](unsigned base0, unsigned base1, const int (&tapo)[NCH], auto loB_c, const short8 (&wf)[NCH][PARTS], f32x4 &acc0, f32x4 &acc1) {
        static_assert(PARTS == 2 || PARTS == 1, "");
        constexpr int NBUF = 3;
        short8 xh[NBUF][2], xl[NBUF][2];
        auto fetch = [&](int k) {
            const unsigned a0 = base0 + tapo[k], a1 = base1 + tapo[k];
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[k % NBUF][0]) : "v"(a0));
            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k % NBUF][0]) : "v"(a0), "n"(decltype(loB_c)::value));
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[k % NBUF][1]) : "v"(a1));
            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k % NBUF][1]) : "v"(a1), "n"(decltype(loB_c)::value));
        };
        fetch(0);
        fetch(1);
        fetch(2);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int left = (NCH - 1 - k < 2 ? NCH - 1 - k : 2) * 2 * PARTS;   // reads of later chunks that may still be in flight
            auto &h0 = xh[k % NBUF][0], &h1 = xh[k % NBUF][1], &l0 = xl[k % NBUF][0], &l1 = xl[k % NBUF][1];
            if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            if constexpr (PARTS == 1) {
                l0 = h0;
                l1 = h1;
            }
            if constexpr (PARTS == 2) {
                acc0 = mma<F16>(wf[k][1], h0, acc0);
                acc1 = mma<F16>(wf[k][1], h1, acc1);
                acc0 = mma<F16>(wf[k][0], l0, acc0);
                acc1 = mma<F16>(wf[k][0], l1, acc1);
            }
            acc0 = mma<F16>(wf[k][0], h0, acc0);
            acc1 = mma<F16>(wf[k][0], h1, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 3 < NCH) fetch(k + 3);
        }
    };

    constexpr int INFLIGHT = (RX - 2) * PPW;
#pragma unroll
    for (int q = 0; q < RX - 1; ++q) issue_next();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    int xslot = 0;
    f32x4 vq0 = f32x4{0.f, 0.f, 0.f, 0.f}, vq1 = vq0;
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v fop = {0u, 0u, 0u, 0u};
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        // a step as in srd_roll_kernel: stage A | barrier | stage B of slice s, then stage C of slice s-1 on the same wave (its own pixels: the feat
        // ring is wave-private); N + 1 steps per column, the last one holds only stage C of the last slice
        for (int s = 0; s <= a.N; ++s) {
            const bool produce = s < a.N;
            const unsigned fslot_off = F_OFF + (s % 3) * FSLOTB;
            if (produce) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
                if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (s == 0) {   // feat[-1] = 0 (ring slot 2), the wave's own pixels
                lds_store8(F_OFF + 2 * FSLOTB + pb_f + g * 8, 0u, 0u);
                if constexpr (PARTS == 2) lds_store8(F_OFF + 2 * FSLOTB + FPLANEB + pb_f + g * 8, 0u, 0u);
            }
            // ---- stage A ---------------------------------------------------------------------------------------------------------
            if (produce && !(ABL & 2)) {
                f32x4 accA[TA];
                if (nA == 2) {
                    accA[0] = b0;
                    accA[1] = b0;
                    tile_mma2(lds0 + X_OFF + xslot * SLOTB + pa[0], lds0 + X_OFF + xslot * SLOTB + pa[1], tapA, std::integral_constant<int, PLANEB>{}, w0, accA[0], accA[1]);
                } else {
                    accA[0] = tile_mma(lds0 + X_OFF + xslot * SLOTB + pa[0], tapA, std::integral_constant<int, PLANEB>{}, w0, b0);
                    accA[1] = b0;
                }
#pragma unroll
                for (int j = 0; j < TA; ++j) {
                    if (j >= nA) break;
                    const f32x4 acc = accA[j];
                    const int iy = U.gy0 - 1 + ta_y[j], ix = U.gx0 - 1 + ta_x[j];
                    const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    if (ta_ok[j]) {
                        uint32_t h01, h23, l01, l23;
                        Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                        Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                        lds_store8(ta_st[j], h01, h23);
                        if constexpr (PARTS == 2) lds_store8(ta_st[j] + TPLANEB, l01, l23);
                    }
                }
            }
            if (produce) {
                if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                // ---- stage B ---------------------------------------------------------------------------------------------------------
                if constexpr (!(ABL & 4)) {
                    const f32x4 acc = tile_mma(lds0 + T_OFF + pbo, tapB, std::integral_constant<int, TPLANEB>{}, w2, b2);
                    const unsigned xp = lds0 + X_OFF + xslot * SLOTB + pb_res;
                    u32x2 xh, xl = {0u, 0u};
                    asm volatile("ds_read_b64 %0, %1" : "=v"(xh) : "v"(xp));
                    if constexpr (PARTS == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xl) : "v"(xp), "n"(PLANEB));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh), "+v"(xl));
                    float r0, r1, r2, r3;
                    Fmt<PREC>::join2(xh[0], xl[0], r0, r1);
                    Fmt<PREC>::join2(xh[1], xl[1], r2, r3);
                    f32x4 v;
                    v[0] = relu_bits(acc[0] + r0);
                    v[1] = relu_bits(acc[1] + r1);
                    v[2] = relu_bits(acc[2] + r2);
                    v[3] = relu_bits(acc[3] + r3);
                    uint32_t fh01, fh23, fl01, fl23;
                    Fmt<PREC>::split2(v[0], v[1], fh01, fl01);
                    Fmt<PREC>::split2(v[2], v[3], fh23, fl23);
                    lds_store8(fslot_off + pb_f + g * 8, fh01, fh23);
                    if constexpr (PARTS == 2) lds_store8(fslot_off + FPLANEB + pb_f + g * 8, fl01, fl23);
                    fop = u32x4v{fh01, fh23, fl01, fl23};   // feat[s], channels 4g..4g+3 as [hi x4 | lo x4]: stage C's K octet for the slice behind z
                    vq1 = vq0;
                    vq0 = v;
                }
            } else {
                vq1 = vq0;
                fop = u32x4v{0u, 0u, 0u, 0u};   // feat[N] = 0
            }
            // ---- stage C: attention for slice z = s-1 (feat[z+1] = what stage B just wrote, or the zeros behind the last slice) ----------------------------------------------------------------------
            if (s >= 1 && !(ABL & 1)) {
                const int z = s - 1;
                const unsigned sm = F_OFF + ((z + 2) % 3) * FSLOTB, sc = F_OFF + (z % 3) * FSLOTB;
                // chunk 0: K octet g = (slice z-1 + (g >> 1), channel octet g & 1) out of the ring; chunk 1: K octet g = channels 4g..4g+3 of feat[z+1] as
                // [hi x4 | lo x4], straight from stage B's registers (fragments [w_hi w_hi] and [w_lo 0]: two MFMAs)
                const unsigned ad0 = lds0 + ((g >> 1) ? sc : sm) + pb_f + (g & 1) * 16;
                short8 fh0, fl0;
                asm volatile("ds_read_b128 %0, %1" : "=v"(fh0) : "v"(ad0));
                if constexpr (PARTS == 2) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl0) : "v"(ad0), "n"(FPLANEB));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh0), "+v"(fl0));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh0));
                }
                const short8 f1op = __builtin_bit_cast(short8, fop);
                f32x4 at = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (PARTS == 2) {
                    at = mma<F16>(w3f[1][1], f1op, at);
                    at = mma<F16>(w3f[0][1], fh0, at);
                    at = mma<F16>(w3f[0][0], fl0, at);
                }
                at = mma<F16>(w3f[1][0], f1op, at);
                at = mma<F16>(w3f[0][0], fh0, at);
                uint32_t ah01, ah23, al01, al23;
                Fmt<PREC>::split2(relu_bits(at[0]), relu_bits(at[1]), ah01, al01);
                Fmt<PREC>::split2(relu_bits(at[2]), relu_bits(at[3]), ah23, al23);
                const u32x4v bq = {ah01, ah23, al01, al23};
                const short8 b2op = __builtin_bit_cast(short8, bq);
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (PARTS == 2) o = mma<F16>(w1f[1], b2op, o);
                o = mma<F16>(w1f[0], b2op, o);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = vq1[i] + relu_bits(o[i]);
                uint32_t h01, h23, l01, l23;
                Fmt<PREC>::split2(v[0], v[1], h01, l01);
                Fmt<PREC>::split2(v[2], v[3], h23, l23);
                const int64_t pix = (((int64_t)U.b * a.N + z) * a.H + U.gy0 + pb_y) * a.W + U.gx0 + pb_x;
                if constexpr (POOL) {   // 2x2 block: column neighbour = lane r ^ 1, row neighbour = lane r ^ 8
                    float m[4];
                    Fmt<PREC>::join2(h01, l01, m[0], m[1]);
                    Fmt<PREC>::join2(h23, l23, m[2], m[3]);
                    uint32_t ph01, ph23, pl01, pl23;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {   // non-negative values (sums of two ReLU results): maxima on the bit patterns
                        uint32_t mu = __float_as_uint(m[i]);
                        mu = max(mu, (uint32_t)__builtin_amdgcn_mov_dpp((int)mu, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                        mu = max(mu, (uint32_t)__builtin_amdgcn_mov_dpp((int)mu, 0x128, 0xF, 0xF, true));   // row_ror:8
                        m[i] = __uint_as_float(mu);
                    }
                    Fmt<PREC>::split2(m[0], m[1], ph01, pl01);
                    Fmt<PREC>::split2(m[2], m[3], ph23, pl23);
                    if constexpr (PARTS == 2) {
                        swap16(ph01, pl01);
                        swap16(ph23, pl23);
                    }
                    if ((r & 9) == 0 && !(ABL & 32)) {
                        const int64_t pp = (((int64_t)U.b * a.N + z) * (a.H / 2) + (U.gy0 / 2 + (wave >> 1))) * (a.W / 2) + U.gx0 / 2 + 4 * (wave & 1) + ((r & 7) >> 1);
                        if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(a.pooled + pp * rec + (g & 1) * C + (g >> 1) * 8) = make_uint4(ph01, ph23, pl01, pl23);
                        else *reinterpret_cast<uint2 *>(a.pooled + pp * rec + g * 4) = make_uint2(ph01, ph23);
                    }
                }
                if constexpr (PARTS == 2) {
                    swap16(h01, l01);
                    swap16(h23, l23);
                    if ((ABL & 32) == 0 || h01 == 0x12345u) *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C + (g >> 1) * 8) = make_uint4(h01, h23, l01, l23);
                } else {
                    *reinterpret_cast<uint2 *>(a.out + pix * rec + g * 4) = make_uint2(h01, h23);
                }
            }
            if (produce) {
                if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if constexpr (!(ABL & 8)) issue_next();
                xslot = (xslot + 1 == RX) ? 0 : xslot + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- srd_pipe16 (round 6): srd_roll16's block as a software pipeline over the workgroup's slice stream -----------------------------------------
// srd_roll16 walks a column slice by slice as  barrier | stage A | barrier | stage B, stage C | barrier | fill : three barriers and three dependent
// stage chains per step that add up (profiles/r04_ablation_srd_roll.txt: no A -36 %, no B -25 %, no C -24 %, no barriers only -8 %), plus one drain step
// per column.  Here the stages of ONE step belong to three consecutive stream positions -- stage A of position p (t into one of TWO t buffers), stage B
// of position p - 1 (its residual pixels of x were read into registers a step earlier, so x[p - 1]'s slot is free), stage C of position p - 2 -- so
// nothing a wave does inside a step depends on what another wave does in the same step: ONE barrier per step (t[p - 1] complete, x[p] landed, the
// slot of x[p - 1] and the buffer t[p] free), the three chains are independent work for the scheduler and the SIMD's other wave, and the stream runs
// on across columns (positions carry their own column / slice; the attention's zero slices at a column's ends are a zeroed ring slot / a zero operand),
// so a column costs N steps, not N + 1.  Same arithmetic in the same order as srd_roll16: bit-identical results (tested).
// 16 output channels fill the MFMA result rows, so no pixel pairs: a GEMM column is one pixel, the 1x3x3 convs contract over
// 5 chunks of (2 taps x 16 channels) (tap 9 = zeros), records are 32 bytes per plane (natural column order).  Columns are
// 4 x 16 pixels (LDS: 4 x-slices of 8 x 20 pixels + t + the feat ring = 67 KB, two workgroups per CU).  Stage B / C tiles
// are 2 rows x 8 pixels per wave so that the 2x2 max-pool stays inside a wave.  Streaming skeleton, counted waits, inline-asm
// LDS access and the MFMA attention (its split result = the 1x1x1 conv's operand in place) as in srd_roll_kernel.
// ABL (development only, DFFW_SRD_ABL): timing ablations -- 1 no stage C, 2 no stage A, 4 no stage B, 8 no fill, 16 no barriers, 32 no global stores
template <int PREC, bool POOL, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void srd_pipe16_kernel(const SrdArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 16, TY = 4, TX = 16, NWAVES = 4;
    constexpr int XY = TY + 4, XX = TX + 4, XPIX = XY * XX;
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;
    // t rows at a pitch of 24 pixels = 768 bytes: a stage-B tile is 2 rows x 8 pixels, and the 16 lanes of a ds_read_b128 row group only cover 16 distinct
    // 16-byte bank groups when its two rows are a multiple of 256 bytes apart (18-pixel rows, 576 bytes: two-way conflicts on every stage-B read)
    constexpr int TXTP = 24;
    constexpr int PIXB = C * 2;
    constexpr int NPIECE = 6;                                      // 1 KiB wave instructions per plane (5 hold the 160 pixels; 6 keeps 3 per wave)
    static_assert(NPIECE * 32 >= XPIX, "plane holds the footprint");
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int RX = 4;
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (counted vmcnt waits)");
    constexpr int TPLANEB = TYT * TXTP * PIXB;
    constexpr int FPLANEB = TY * TX * PIXB;
    constexpr int FSLOTB = PARTS * FPLANEB;
    constexpr int X_OFF = 0, T_OFF = RX * SLOTB, TBUFB = PARTS * TPLANEB, F_OFF = T_OFF + 2 * TBUFB;   // two t buffers (positions of either parity)
    constexpr int NCH = 5;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[F_OFF + 3 * FSLOTB];
    static_assert(2 * (F_OFF + 3 * FSLOTB) <= 160 * 1024, "two workgroups per CU");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto lds_store8 = [&](unsigned byte_off, uint32_t v0, uint32_t v1) {
        const u32x2 d = {v0, v1};
        asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + byte_off), "v"(d) : "memory");
    };
    auto lds_store16 = [&](unsigned byte_off, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + byte_off), "v"(v) : "memory"); };

    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };

    const int rec = PARTS * C;
    const int slice_elems = a.H * a.W * rec;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, pix = ci >> 1, oct = ci & 1;
            const int fy = pix / XX, fx = pix - fy * XX;
            const int iy = c.gy0 - 2 + fy, ix = c.gx0 - 2 + fx;
            fok[k] = p < NP && pix < XPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            fsrc[k] = a.x + (int64_t)c.b * a.N * slice_elems + (int64_t)(iy * a.W + ix) * rec + part * C + oct * 8;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const bool zin = fu < uend;
        unsigned char *slot = smem + X_OFF + fslot * SLOTB;
        const int64_t zo = (int64_t)fq * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RX) ? 0 : fslot + 1;
        if (++fq == a.N && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // stage A: the 6 x 18 t pixels are 7 operand tiles: tiles 0-5 = the first 16 pixels of row 0-5 (16 consecutive pixels of ONE row: conflict-free
    // operand reads; tiles of 16 consecutive indices of the 6 x 18 region wrapped rows and collided two ways), tile 6 = the two remaining pixels of
    // each row (12 of its 16 lanes busy); waves 0-2 take two tiles, wave 3 one
    constexpr int TA = 2;
    const int nA = wave < 3 ? 2 : 1;
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        const int tile = j == 0 ? wave : 4 + wave;
        ta_ok[j] = tile < TYT || r < 2 * TYT;
        ta_y[j] = tile < TYT ? tile : (r < 2 * TYT ? r >> 1 : TYT - 1);
        ta_x[j] = tile < TYT ? r : TX + (r & 1);
        pa[j] = (ta_y[j] * XX + ta_x[j]) * PIXB + (g & 1) * 16;
        ta_st[j] = T_OFF + (ta_y[j] * TXTP + ta_x[j]) * PIXB + g * 8;
    }
    // K octet g of chunk k = (filter tap 2k + (g >> 1), channel octet g & 1); tap 9 carries zero weights
    int tapA[NCH], tapB[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int tap = 2 * k + (g >> 1);
        const int dy = tap < 9 ? tap / 3 : 0, dx = tap < 9 ? tap % 3 : 0;
        tapA[k] = (dy * XX + dx) * PIXB;
        tapB[k] = (dy * TXTP + dx) * PIXB;
    }
    // stage B / C: wave w = rows 2*(w >> 1), +1 x columns 8*(w & 1) .. +7 (the 2x2 pooling blocks stay inside the wave)
    const int pb_y = 2 * (wave >> 1) + (r >> 3), pb_x = 8 * (wave & 1) + (r & 7);
    const int pbo = (pb_y * TXTP + pb_x) * PIXB + (g & 1) * 16;
    const int pb_res = ((pb_y + 2) * XX + pb_x + 2) * PIXB + g * 8;
    const int pb_f = (pb_y * TX + pb_x) * PIXB;
    short8 w0[NCH][PARTS], w2[NCH][PARTS], w3f[2][PARTS], w1f[PARTS];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
            w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
        }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w3f[k][pt] = reinterpret_cast<const short8 *>(a.w3f)[(k * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) w1f[pt] = reinterpret_cast<const short8 *>(a.w1f)[pt * 64 + lane];
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + g * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + g * 4);
    auto tile_mma = [&](unsigned base, const int (&tapo)[NCH], auto loB_c, const short8 (&wf)[NCH][PARTS], f32x4 acc) {
        constexpr int loB = decltype(loB_c)::value;   // the lo plane as an immediate of the read
        short8 xh[NCH], xl[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const unsigned ad = base + tapo[k];
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[k]) : "v"(ad));
            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k]) : "v"(ad), "n"(loB));
            else xl[k] = short8{0, 0, 0, 0, 0, 0, 0, 0};   /* (single-part storage: never contracted; NOT a copy of the in-flight hi fragment, tools/isa_wait_lint.py) */
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int left = (NCH - 1 - k) * PARTS;   // reads still allowed in flight (compile-time after unrolling)
            if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xh[k]), "+v"(xl[k]));
            else if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xh[k]), "+v"(xl[k]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[k]), "+v"(xl[k]));
            if constexpr (PARTS == 2) {
                acc = mma<F16>(wf[k][1], xh[k], acc);
                acc = mma<F16>(wf[k][0], xl[k], acc);
            }
            acc = mma<F16>(wf[k][0], xh[k], acc);
        }
        return acc;
    };

    // two operand tiles side by side (stage A of the waves that own two): the reads of both run three chunks ahead of the MFMAs and the two
    // accumulator chains alternate, so one tile's LDS latency and MFMA dependency gaps are covered by the other's work
    auto tile_mma2 = [&](unsigned base0, unsigned base1, const int (&tapo)[NCH], auto loB_c, const short8 (&wf)[NCH][PARTS], f32x4 &acc0, f32x4 &acc1) {
        static_assert(PARTS == 2 || PARTS == 1, "");
        constexpr int NBUF = 3;
        short8 xh[NBUF][2], xl[NBUF][2];
        auto fetch = [&](int k) {
            const unsigned a0 = base0 + tapo[k], a1 = base1 + tapo[k];
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[k % NBUF][0]) : "v"(a0));
            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k % NBUF][0]) : "v"(a0), "n"(decltype(loB_c)::value));
            asm volatile("ds_read_b128 %0, %1" : "=v"(xh[k % NBUF][1]) : "v"(a1));
            if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xl[k % NBUF][1]) : "v"(a1), "n"(decltype(loB_c)::value));
        };
        fetch(0);
        fetch(1);
        fetch(2);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int left = (NCH - 1 - k < 2 ? NCH - 1 - k : 2) * 2 * PARTS;   // reads of later chunks that may still be in flight
            auto &h0 = xh[k % NBUF][0], &h1 = xh[k % NBUF][1], &l0 = xl[k % NBUF][0], &l1 = xl[k % NBUF][1];
            if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
            if constexpr (PARTS == 1) {
                l0 = h0;
                l1 = h1;
            }
            if constexpr (PARTS == 2) {
                acc0 = mma<F16>(wf[k][1], h0, acc0);
                acc1 = mma<F16>(wf[k][1], h1, acc1);
                acc0 = mma<F16>(wf[k][0], l0, acc0);
                acc1 = mma<F16>(wf[k][0], l1, acc1);
            }
            acc0 = mma<F16>(wf[k][0], h0, acc0);
            acc1 = mma<F16>(wf[k][0], h1, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 3 < NCH) fetch(k + 3);
        }
    };

    constexpr int INFLIGHT = (RX - 2) * PPW;
#pragma unroll
    for (int q = 0; q < RX - 1; ++q) issue_next();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    int xslot = 0;
    f32x4 vq0 = f32x4{0.f, 0.f, 0.f, 0.f}, vq1 = vq0;
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v fop = {0u, 0u, 0u, 0u};
    // the residual pixels of x for stage B: requested while their slice is stage A's (rn*), used one step later (rc*)
    u32x2 rnh = {0u, 0u}, rnl = {0u, 0u}, rch = {0u, 0u}, rcl = {0u, 0u};
    // stream positions: stage A works on (UA, zA), stage C on (UC, zC); P positions in all
    const int ncol = (uend - ufirst + wgs_per_xcd - 1) / wgs_per_xcd;
    const int P = ncol * a.N;
    int cuA = ufirst, zA = 0;
    Unit UA = decode(cuA), UB = UA, UC = UA;
    int zB = 0, zC = 0;
    for (int p = 0; p < P + 2; ++p) {
        const bool hasA = p < P, hasB = p >= 1 && p <= P, hasC = p >= 2;
        // ---- the step's one barrier: x[p] has landed (this wave's pieces; after the barrier everyone's), t[p - 1] is complete, x[p - 1]'s slot and t[p]'s buffer are free
        if (hasA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(rnh), "+v"(rnl));   // (requested in the previous step, landed by the lgkmcnt(0) above)
        rch = rnh;
        rcl = rnl;
        if (hasA && !(ABL & 8)) issue_next();      // slice p + RX - 1 into the slot x[p - 1] left
        const unsigned tA = T_OFF + (p & 1) * TBUFB, tB = T_OFF + ((p & 1) ^ 1) * TBUFB;
        // ---- stage A of position p -------------------------------------------------------------------------------------------------
        if (hasA && !(ABL & 2)) {
            f32x4 accA[TA];
            if (nA == 2) {
                accA[0] = b0;
                accA[1] = b0;
                tile_mma2(lds0 + X_OFF + xslot * SLOTB + pa[0], lds0 + X_OFF + xslot * SLOTB + pa[1], tapA, std::integral_constant<int, PLANEB>{}, w0, accA[0], accA[1]);
            } else {
                accA[0] = tile_mma(lds0 + X_OFF + xslot * SLOTB + pa[0], tapA, std::integral_constant<int, PLANEB>{}, w0, b0);
                accA[1] = b0;
            }
#pragma unroll
            for (int j = 0; j < TA; ++j) {
                if (j >= nA) break;
                const f32x4 acc = accA[j];
                const int iy = UA.gy0 - 1 + ta_y[j], ix = UA.gx0 - 1 + ta_x[j];
                const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                if (ta_ok[j]) {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                    Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                    lds_store8(tA + ta_st[j] - T_OFF, h01, h23);
                    if constexpr (PARTS == 2) lds_store8(tA + ta_st[j] - T_OFF + TPLANEB, l01, l23);
                }
            }
        }
        if (hasA) {   // stage B's residual pixels of this slice, for the next step
            const unsigned xp = lds0 + X_OFF + xslot * SLOTB + pb_res;
            asm volatile("ds_read_b64 %0, %1" : "=v"(rnh) : "v"(xp));
            if constexpr (PARTS == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rnl) : "v"(xp), "n"(PLANEB));
        }
        // ---- stage B of position p - 1 ---------------------------------------------------------------------------------------------
        if (hasB) {
            if constexpr (!(ABL & 4)) {
                const unsigned fslot_off = F_OFF + ((p - 1) % 3) * FSLOTB;
                const f32x4 acc = tile_mma(lds0 + tB + pbo, tapB, std::integral_constant<int, TPLANEB>{}, w2, b2);
                float r0, r1, r2, r3;
                Fmt<PREC>::join2(rch[0], rcl[0], r0, r1);
                Fmt<PREC>::join2(rch[1], rcl[1], r2, r3);
                f32x4 v;
                v[0] = relu_bits(acc[0] + r0);
                v[1] = relu_bits(acc[1] + r1);
                v[2] = relu_bits(acc[2] + r2);
                v[3] = relu_bits(acc[3] + r3);
                uint32_t fh01, fh23, fl01, fl23;
                Fmt<PREC>::split2(v[0], v[1], fh01, fl01);
                Fmt<PREC>::split2(v[2], v[3], fh23, fl23);
                lds_store8(fslot_off + pb_f + g * 8, fh01, fh23);
                if constexpr (PARTS == 2) lds_store8(fslot_off + FPLANEB + pb_f + g * 8, fl01, fl23);
                fop = u32x4v{fh01, fh23, fl01, fl23};   // feat of position p - 1, channels 4g..4g+3 as [hi x4 | lo x4]: stage C's K octet for the slice behind it
                vq1 = vq0;
                vq0 = v;
            }
        } else {
            vq1 = vq0;
            fop = u32x4v{0u, 0u, 0u, 0u};
        }
        // ---- stage C of position q = p - 2: attention for slice zC of column UC (feat[zC + 1] = what stage B just wrote, unless the column ends here) ----
        if (hasC && !(ABL & 1)) {
            const int q = p - 2;
            const unsigned sm = F_OFF + ((q + 2) % 3) * FSLOTB, sc = F_OFF + (q % 3) * FSLOTB;
            if (zC == 0) {   // feat[-1] = 0: the ring slot behind holds the previous column's last slice, which nothing needs any more (the wave's own pixels)
                lds_store8(sm + pb_f + g * 8, 0u, 0u);
                if constexpr (PARTS == 2) lds_store8(sm + FPLANEB + pb_f + g * 8, 0u, 0u);
            }
            // chunk 0: K octet g = (slice zC - 1 + (g >> 1), channel octet g & 1) out of the ring; chunk 1: K octet g = channels 4g..4g+3 of feat[zC + 1] as
            // [hi x4 | lo x4], straight from stage B's registers (fragments [w_hi w_hi] and [w_lo 0]: two MFMAs)
            const unsigned ad0 = lds0 + ((g >> 1) ? sc : sm) + pb_f + (g & 1) * 16;
            short8 fh0, fl0;
            asm volatile("ds_read_b128 %0, %1" : "=v"(fh0) : "v"(ad0));
            if constexpr (PARTS == 2) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl0) : "v"(ad0), "n"(FPLANEB));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh0), "+v"(fl0));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh0));
            }
            const u32x4v fz = zC == a.N - 1 ? u32x4v{0u, 0u, 0u, 0u} : fop;   // feat[N] = 0
            const short8 f1op = __builtin_bit_cast(short8, fz);
            f32x4 at = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (PARTS == 2) {
                at = mma<F16>(w3f[1][1], f1op, at);
                at = mma<F16>(w3f[0][1], fh0, at);
                at = mma<F16>(w3f[0][0], fl0, at);
            }
            at = mma<F16>(w3f[1][0], f1op, at);
            at = mma<F16>(w3f[0][0], fh0, at);
            uint32_t ah01, ah23, al01, al23;
            Fmt<PREC>::split2(relu_bits(at[0]), relu_bits(at[1]), ah01, al01);
            Fmt<PREC>::split2(relu_bits(at[2]), relu_bits(at[3]), ah23, al23);
            const u32x4v bq = {ah01, ah23, al01, al23};
            const short8 b2op = __builtin_bit_cast(short8, bq);
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (PARTS == 2) o = mma<F16>(w1f[1], b2op, o);
            o = mma<F16>(w1f[0], b2op, o);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = vq1[i] + relu_bits(o[i]);
            uint32_t h01, h23, l01, l23;
            Fmt<PREC>::split2(v[0], v[1], h01, l01);
            Fmt<PREC>::split2(v[2], v[3], h23, l23);
            const int64_t pix = (((int64_t)UC.b * a.N + zC) * a.H + UC.gy0 + pb_y) * a.W + UC.gx0 + pb_x;
            if constexpr (POOL) {   // 2x2 block: column neighbour = lane r ^ 1, row neighbour = lane r ^ 8
                float m[4];
                Fmt<PREC>::join2(h01, l01, m[0], m[1]);
                Fmt<PREC>::join2(h23, l23, m[2], m[3]);
                uint32_t ph01, ph23, pl01, pl23;
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // non-negative values (sums of two ReLU results): maxima on the bit patterns
                    uint32_t mu = __float_as_uint(m[i]);
                    mu = max(mu, (uint32_t)__builtin_amdgcn_mov_dpp((int)mu, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                    mu = max(mu, (uint32_t)__builtin_amdgcn_mov_dpp((int)mu, 0x128, 0xF, 0xF, true));   // row_ror:8
                    m[i] = __uint_as_float(mu);
                }
                Fmt<PREC>::split2(m[0], m[1], ph01, pl01);
                Fmt<PREC>::split2(m[2], m[3], ph23, pl23);
                if constexpr (PARTS == 2) {
                    swap16(ph01, pl01);
                    swap16(ph23, pl23);
                }
                if ((r & 9) == 0 && !(ABL & 32)) {
                    const int64_t pp = (((int64_t)UC.b * a.N + zC) * (a.H / 2) + (UC.gy0 / 2 + (wave >> 1))) * (a.W / 2) + UC.gx0 / 2 + 4 * (wave & 1) + ((r & 7) >> 1);
                    if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(a.pooled + pp * rec + (g & 1) * C + (g >> 1) * 8) = make_uint4(ph01, ph23, pl01, pl23);
                    else *reinterpret_cast<uint2 *>(a.pooled + pp * rec + g * 4) = make_uint2(ph01, ph23);
                }
            }
            if constexpr (PARTS == 2) {
                swap16(h01, l01);
                swap16(h23, l23);
                if ((ABL & 32) == 0 || h01 == 0x12345u) *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C + (g >> 1) * 8) = make_uint4(h01, h23, l01, l23);
            } else {
                *reinterpret_cast<uint2 *>(a.out + pix * rec + g * 4) = make_uint2(h01, h23);
            }
        }
        // ---- the positions move on --------------------------------------------------------------------------------------------------------
        UC = UB;
        zC = zB;
        UB = UA;
        zB = zA;
        if (hasA) {
            xslot = (xslot + 1 == RX) ? 0 : xslot + 1;
            if (++zA == a.N) {
                zA = 0;
                cuA += wgs_per_xcd;
                if (cuA < uend) UA = decode(cuA);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one operand fragment (hi [+ lo] plane) of an of_roll tile, and the counted wait that releases it (DS operations retire in order: `left` =
// operations requested after it that may still be in flight; the "+v" ties keep the MFMAs behind the wait)
template <int PARTS, int LOB>
__device__ __forceinline__ void of_read(unsigned ad, short8 &h, short8 &l) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(h) : "v"(ad));
    if constexpr (PARTS == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l) : "v"(ad), "n"(LOB));
}
template <int PARTS>
__device__ __forceinline__ void of_wait(int left, short8 &h, short8 &l) {
    if constexpr (PARTS == 1) {
        (void)l;
        if (left >= 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(h));
        else if (left >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(h));
        else if (left >= 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(h));
        else if (left >= 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(h));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h));
        return;
    }
    if (left >= 10) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(h), "+v"(l));
    else if (left >= 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(h), "+v"(l));
    else if (left >= 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(h), "+v"(l));
    else if (left >= 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(h), "+v"(l));
    else if (left >= 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(h), "+v"(l));
    else if (left >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(h), "+v"(l));
    else if (left >= 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(h), "+v"(l));
    else if (left >= 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(h), "+v"(l));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h), "+v"(l));
}

// ---- of_roll: a stride-1 residual block of the alignment network (End_to_End.py:135-145, `OF_feature.0`, `OF_feature.1`) -------
//     out = relu( conv1x1x1(x) + BN(conv1x3x3(relu(BN(conv1x3x3(x))))) ),   8 (3 real) or 16 -> 16 channels, full resolution
// As two launches t = relu(BN(conv(x))) went through HBM and the second conv re-read x for the folded shortcut (two 16-channel
// stages + a mostly empty third one).  Here, as in srd_roll16: x slices stream through an LDS FIFO, stage A leaves t in LDS,
// stage B contracts t (5 chunks) plus ONE extra chunk for the 1x1x1 shortcut (the centre pixel of x, already in LDS) and stores
// the block's output.  Slices are independent (no attention), so a step is A -> barrier -> B -> barrier.
// SUMS (the pair of 16 -> 16 convs in the middle of the level-1 alignment head, whose result only feeds the head's last conv + plane
// mean = plane sums, dffw_kernels.hip "alpha head tail"): the block's output is not stored; instead every (column, slice) leaves 18
// 16-channel fp32 vectors in a.out (as float[(plane * tiles + tile) * 288 + k * 16 + c], plane = b * N + slice): k = 3w, 3w+1, 3w+2 the
// sum over wave w's two rows of the 8 x 16 tile, over their first and over their last pixel; 12 / 13 the tile's first / last row;
// 14..17 its corner pixels (top-left, top-right, bottom-left, bottom-right); head_tail_finish_tiles_kernel adds them up in a fixed order.
template <int PREC, bool CIN8, bool SUMS = false>
__global__ __launch_bounds__(256) void of_roll_kernel(const SrdArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 16, CI = CIN8 ? 8 : 16, TY = 8, TX = 16, NWAVES = 4;
    constexpr int XY = TY + 4, XX = TX + 4, XPIX = XY * XX;
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;
    constexpr int PIXB = C * 2, XPIXB = CI * 2, XOCT = CI / 8;
    constexpr int NPIECE = (XPIX * XOCT + 63) / 64;                // 1 KiB wave instructions per plane
    constexpr int PLANEB = NPIECE * 1024;
    constexpr int SLOTB = PARTS * PLANEB;
    constexpr int RX = CIN8 ? 4 : 3;
    constexpr int NP = PARTS * NPIECE, PPW = (NP + NWAVES - 1) / NWAVES;
    static_assert(NP % PPW == 0, "every wave issues PPW pieces or none (counted vmcnt waits)");
    constexpr int TPLANEB = TPIX * PIXB;
    constexpr int X_OFF = 0, T_OFF = RX * SLOTB;
    constexpr int NCHA = CIN8 ? 3 : 5, NCHB = 5;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[T_OFF + PARTS * TPLANEB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto lds_store8 = [&](unsigned byte_off, uint32_t v0, uint32_t v1) {
        const u32x2 d = {v0, v1};
        asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + byte_off), "v"(d) : "memory");
    };

    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };

    const int rec = PARTS * C, xrec = PARTS * CI;
    const int slice_elems = a.H * a.W * xrec;
    const uint16_t *fsrc[PPW];
    bool fok[PPW];
    int fu = ufirst, fq = 0;
    auto setup_fill = [&]() {
        const Unit c = decode(fu);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            const int part = p / NPIECE, i = p % NPIECE;
            const int ci = i * 64 + lane, pix = ci / XOCT, oct = ci % XOCT;
            const int fy = pix / XX, fx = pix - fy * XX;
            const int iy = c.gy0 - 2 + fy, ix = c.gx0 - 2 + fx;
            fok[k] = p < NP && pix < XPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            fsrc[k] = a.x + (int64_t)c.b * a.N * slice_elems + (int64_t)(iy * a.W + ix) * xrec + part * CI + oct * 8;
        }
    };
    setup_fill();
    int fslot = 0;
    auto issue_next = [&]() {
        const bool zin = fu < uend;
        unsigned char *slot = smem + X_OFF + fslot * SLOTB;
        const int64_t zo = (int64_t)fq * slice_elems;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = wave * PPW + k;
            if (p >= NP) break;
            const int part = p / NPIECE, i = p % NPIECE;
            const uint16_t *src = (zin && fok[k]) ? fsrc[k] + zo : a.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(slot + part * PLANEB + i * 1024), 16, 0, 0);
        }
        fslot = (fslot + 1 == RX) ? 0 : fslot + 1;
        if (++fq == a.N && fu < uend) {
            fq = 0;
            fu += wgs_per_xcd;
            if (fu < uend) setup_fill();
        }
    };

    // stage A: the 10 x 18 t pixels are 12 operand tiles, three per wave: tiles 0-9 = the first 16 pixels of row 0-9 (16 consecutive pixels of ONE
    // row: conflict-free operand reads; tiles of 16 consecutive indices of the region wrapped rows and collided two ways, as in srd_roll16 before
    // round 4), tiles 10-11 = the two remaining pixels of each row (the last one a quarter busy)
    constexpr int TA = 3;
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        const int tile = wave * TA + j;
        const int q = (tile - TYT) * 16 + r;                       // index among the 2 * TYT left-over pixels
        ta_ok[j] = tile < TYT || q < 2 * TYT;
        ta_y[j] = tile < TYT ? tile : (q < 2 * TYT ? q >> 1 : TYT - 1);
        ta_x[j] = tile < TYT ? r : TX + (q & 1);
        pa[j] = (ta_y[j] * XX + ta_x[j]) * XPIXB + (CIN8 ? 0 : (g & 1) * 16);
        ta_st[j] = T_OFF + (ta_y[j] * TXT + ta_x[j]) * PIXB + g * 8;
    }
    // K octets: 8 input channels: chunk k, octet g = filter tap 4k + g; 16 channels: (tap 2k + (g >> 1), channel octet g & 1);
    // taps >= 9 carry zero weights
    int tapA[NCHA], tapB[NCHB];
#pragma unroll
    for (int k = 0; k < NCHA; ++k) {
        const int tap = CIN8 ? 4 * k + g : 2 * k + (g >> 1);
        const int dy = tap < 9 ? tap / 3 : 0, dx = tap < 9 ? tap % 3 : 0;
        tapA[k] = (dy * XX + dx) * XPIXB;
    }
#pragma unroll
    for (int k = 0; k < NCHB; ++k) {
        const int tap = 2 * k + (g >> 1);
        const int dy = tap < 9 ? tap / 3 : 0, dx = tap < 9 ? tap % 3 : 0;
        tapB[k] = (dy * TXT + dx) * PIXB;
    }
    // stage B: wave w = output rows 2w, 2w+1
    constexpr int TB = 2;
    int pbo[TB], pbx[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int fy = wave * TB + j;
        pbo[j] = (fy * TXT + r) * PIXB + (g & 1) * 16;
        pbx[j] = ((fy + 2) * XX + r + 2) * XPIXB + (g < XOCT ? g : 0) * 16;   // shortcut chunk: channel octet g of the centre pixel of x
    }
    short8 w0[NCHA][PARTS], w2[NCHB + 1][PARTS];   // conv.2: 5 chunks over t + the shortcut chunk over x
#pragma unroll
    for (int k = 0; k < NCHA; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int k = 0; k < NCHB + 1; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + g * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + g * 4);
    constexpr int INFLIGHT = (RX - 2) * PPW;
#pragma unroll
    for (int q = 0; q < RX - 1; ++q) issue_next();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_barrier" ::: "memory");

    int xslot = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit U = decode(cu);
        for (int s = 0; s < a.N; ++s) {
            // (1) this step's x slice has landed (for every wave after the barrier); stage B of the previous step has read t
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
            const unsigned xs = lds0 + X_OFF + xslot * SLOTB;
            // ---- stage A: t = relu(conv.0(x) + shift) on the 10 x 18 region, zero outside the image (conv.2's padding) ------------
            // operand reads run one tile ahead: chunk k of tile j+1 is requested as soon as chunk k of tile j has been contracted (into the same
            // registers), so only the stage's first tile waits for the LDS.  DS operations retire in order: behind the reads of (j, k) there are
            // the 2 (NC-1-k) reads of the tile's later chunks, the 2k already requested for tile j+1 and at most the 2 stores of tile j-1's
            // epilogue -- lgkmcnt(2 (NC-1)) covers (j, k) whether or not the stores were issued.
            short8 fxh[NCHB + 1], fxl[NCHB + 1];
#pragma unroll
            for (int k = 0; k < NCHA; ++k) of_read<PARTS, PLANEB>(xs + pa[0] + tapA[k], fxh[k], fxl[k]);
#pragma unroll
            for (int j = 0; j < TA; ++j) {
                f32x4 acc = b0;
#pragma unroll
                for (int k = 0; k < NCHA; ++k) {
                    of_wait<PARTS>(j + 1 < TA ? (NCHA - 1) * PARTS : (NCHA - 1 - k) * PARTS, fxh[k], fxl[k]);
                    if constexpr (PARTS == 2) {
                        acc = mma<F16>(w0[k][1], fxh[k], acc);
                        acc = mma<F16>(w0[k][0], fxl[k], acc);
                    }
                    acc = mma<F16>(w0[k][0], fxh[k], acc);
                    if (j + 1 < TA) of_read<PARTS, PLANEB>(xs + pa[j + 1] + tapA[k], fxh[k], fxl[k]);
                }
                const int iy = U.gy0 - 1 + ta_y[j], ix = U.gx0 - 1 + ta_x[j];
                const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                if (ta_ok[j]) {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                    Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                    lds_store8(ta_st[j], h01, h23);
                    if constexpr (PARTS == 2) lds_store8(ta_st[j] + TPLANEB, l01, l23);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // ---- stage B: out = relu(conv.2(t) + shift + shortcut(x)) -----------------------------------------------------------
            f32x4 sum_t = {0.f, 0.f, 0.f, 0.f}, sum_c = sum_t;   // SUMS only
            // (chunk NCHB = the shortcut chunk: centre pixel of x, its channel octets as K octets -- weights of absent octets are zeros; reads one
            // tile ahead as in stage A: 2 NCHB operations behind the reads of (j, k), no DS stores in this stage)
#pragma unroll
            for (int k = 0; k < NCHB; ++k) of_read<PARTS, TPLANEB>(lds0 + T_OFF + pbo[0] + tapB[k], fxh[k], fxl[k]);
            of_read<PARTS, PLANEB>(xs + pbx[0], fxh[NCHB], fxl[NCHB]);
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                f32x4 acc = b2;
#pragma unroll
                for (int k = 0; k <= NCHB; ++k) {
                    of_wait<PARTS>(j + 1 < TB ? NCHB * PARTS : (NCHB - k) * PARTS, fxh[k], fxl[k]);
                    if constexpr (PARTS == 2) {
                        acc = mma<F16>(w2[k][1], fxh[k], acc);
                        acc = mma<F16>(w2[k][0], fxl[k], acc);
                    }
                    acc = mma<F16>(w2[k][0], fxh[k], acc);
                    if (j + 1 < TB) {
                        if (k < NCHB) of_read<PARTS, TPLANEB>(lds0 + T_OFF + pbo[j + 1] + tapB[k], fxh[k], fxl[k]);
                        else of_read<PARTS, PLANEB>(xs + pbx[j + 1], fxh[k], fxl[k]);
                    }
                }
                if constexpr (SUMS) {
                    // this lane: channels 4g..4g+3 of pixel (row 2*wave + j, column r).  Row sums over the 16 lanes of the row group by
                    // DPP (quad xor 1, xor 2, half-row mirror, row mirror: every lane ends up with the sum); lanes r = 0 / r = 15 are
                    // the tile's first / last column.  Everything leaves straight from registers (16-byte stores of lanes r = 0 / 15).
                    f32x4 v, rs;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = relu_bits(acc[i]);
                        float t = v[i];
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x141, 0xF, 0xF, true));   // row_half_mirror
                        t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x140, 0xF, 0xF, true));   // row_mirror
                        rs[i] = t;
                    }
                    float *trec = reinterpret_cast<float *>(a.out) +
                                  (((int64_t)U.b * a.N + s) * (a.tiles_y * a.tiles_x) + (U.gy0 / TY) * a.tiles_x + U.gx0 / TX) * (18 * C) + g * 4;
                    if (j == 0) {
                        sum_t = rs;
                        sum_c = v;
                        if (wave == 0 && r == 0) *reinterpret_cast<f32x4 *>(trec + 12 * C) = rs;          // first row of the tile
                        if (wave == 0 && (r == 0 || r == 15)) *reinterpret_cast<f32x4 *>(trec + (r == 0 ? 14 : 15) * C) = v;   // TL, TR
                    } else {
                        sum_t += rs;
                        sum_c += v;
                        if (wave == NWAVES - 1 && r == 0) *reinterpret_cast<f32x4 *>(trec + 13 * C) = rs;  // last row
                        if (wave == NWAVES - 1 && (r == 0 || r == 15)) *reinterpret_cast<f32x4 *>(trec + (r == 0 ? 16 : 17) * C) = v;   // BL, BR
                        if (r == 0) *reinterpret_cast<f32x4 *>(trec + (wave * 3 + 0) * C) = sum_t;         // this wave's two rows
                        if (r == 0 || r == 15) *reinterpret_cast<f32x4 *>(trec + (wave * 3 + (r == 0 ? 1 : 2)) * C) = sum_c;   // ... their first / last column
                    }
                } else {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_bits(acc[0]), relu_bits(acc[1]), h01, l01);
                    Fmt<PREC>::split2(relu_bits(acc[2]), relu_bits(acc[3]), h23, l23);
                    const int64_t pix = (((int64_t)U.b * a.N + s) * a.H + U.gy0 + wave * TB + j) * a.W + U.gx0 + r;
                    if constexpr (PARTS == 2) {
                        swap16(h01, l01);
                        swap16(h23, l23);
                        *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C + (g >> 1) * 8) = make_uint4(h01, h23, l01, l23);
                    } else {
                        *reinterpret_cast<uint2 *>(a.out + pix * rec + g * 4) = make_uint2(h01, h23);
                    }
                }
            }
            // (3) the x slot is free: queue the slice RX-1 ahead into it
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_next();
            xslot = (xslot + 1 == RX) ? 0 : xslot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- head_warp: the first conv of an alignment head on the FOV-warped features, without the warped volume -------------------------
// (End_to_End.py:88-101: FE = FOV_warp(FE, alpha); conv over [ref | cur | flow]; the ref part enters as `ref`, see run_e2e)
//     y0[b,n] = relu( BN(conv1x3x3([warp(fe)[b,n] (CF) | flow_x, flow_y])) + ref[b] )       CF + 2 -> 2 CF channels
// CF = 8: level 1 (full resolution), CF = 16: level 2 (half resolution).  As two launches flow_volume wrote the volume
// [cur | flow | pad] (1.6 GB at 8 x 10 x 480 x 640 for level 1) and the conv read it back.  Here a workgroup walks the slices of a
// column of 8 x 16 output pixels: every channel octet of every pixel of the 10 x 18 footprint has its thread, which gathers the four
// bilinear corners (hi and lo piece each) ONE STEP AHEAD into registers -- the loads of slice s+1 travel under the contraction and
// the stores of slice s -- blends them with the operation order of flow_volume_kernel (warp_octet), splits to the storage format and
// writes its octet of the record [CF channels | flow_x flow_y 0..] into one of two LDS slots; the contraction (CF = 8: 4 waves, wave
// w = output rows 2w, 2w+1; CF = 16: 8 waves, wave w = row w, two 16-channel output tiles; K octet g of chunk k = o = 4k + g ->
// (tap o / OCT, channel octet o % OCT), OCT = CF / 8 + 1; filter resident in LDS) and the epilogue (+ ref, held in registers for all
// slices of the column, ReLU, split, 16-byte stores) follow after one barrier.  Plain loads only (no LDS-DMA): hipcc counts every wait.
#ifndef DFFW_HW_ABL
#define DFFW_HW_ABL 0   // dev-only ablations (tools/build_variant_lib.sh): 1 no corner loads, 2 no output stores, 4 no MFMAs, 8 no blend
#endif
template <int PREC, int CF>
__global__ __launch_bounds__(CF == 8 ? 256 : 512) __attribute__((amdgpu_waves_per_eu(4))) void head_warp_kernel(const HeadWarpArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int GO = CF / 8, OCT = GO + 1, NT = CF / 8, C = 16 * NT;
    constexpr int NWAVES = CF == 8 ? 4 : 8, NTHR = NWAVES * 64;
    constexpr int TY = 8, TX = 16, XY = TY + 2, XX = TX + 2, XPIX = XY * XX;
    constexpr int PIXB = OCT * 16, PLANEB = XPIX * PIXB, SLOTB = PARTS * PLANEB;
    constexpr int NCH = (9 * OCT + 3) / 4, TB = TY / NWAVES;
    constexpr int W_OFF = 2 * SLOTB, WB = NCH * NT * PARTS * 1024;   // the filter: [chunk][output tile][part][64 lanes][16 bytes]
    // the warp parameters of every (sample, slice): (alpha0 + fov, alpha1, alpha2), read from LDS in issue().  As global loads (wave-uniform addresses, but
    // hipcc issues vector loads for them) every step waited vmcnt(0) for them -- with the previous step's result stores in the same queue: a store
    // acknowledgement per step on the critical path (with every load, store, MFMA and blend taken out the kernel still ran 0.51 of its 0.73 ms:
    // profiles/r06_head_warp.txt).  B N <= head_warp_max_planes(); the engine keeps the two-launch form beyond.
    constexpr int PRM_OFF = W_OFF + WB, PRM_MAX = head_warp_max_planes();
    static_assert(XPIX * GO <= NTHR, "one gather item per thread");
    __shared__ __attribute__((aligned(16))) unsigned char smem[PRM_OFF + PRM_MAX * 12];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
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
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };
    const int rec = PARTS * C, frec = PARTS * CF;
    for (int i = tid; i < WB / 16; i += NTHR) reinterpret_cast<uint4 *>(smem + W_OFF)[i] = reinterpret_cast<const uint4 *>(a.w)[i];
    float *prm = reinterpret_cast<float *>(smem + PRM_OFF);
    for (int i = tid; i < a.B * a.N; i += NTHR) {
        const int b = i / a.N, n = i - b * a.N;
        prm[i * 3 + 0] = a.alpha[b * 3 * a.N + n] + a.fov[i];
        prm[i * 3 + 1] = a.alpha[b * 3 * a.N + a.N + n];
        prm[i * 3 + 2] = a.alpha[b * 3 * a.N + 2 * a.N + n];
    }
    __syncthreads();

    // ---- gather side: thread t < 180 * GO owns channel octet t / 180 of footprint pixel t % 180 ---------------------
    const bool gth = tid < XPIX * GO;
    const int goct = GO == 1 ? 0 : tid / XPIX, gp = tid - goct * XPIX;
    const int fy = gp / XX, fx = gp - fy * XX;
    uint4 q[4][PARTS];          // corner k: [hi, lo]
    float wgt[4], flx = 0.f, fly = 0.f;
    bool pin = false;
    auto issue = [&](const Unit &U, int n) {
        const int iy = U.gy0 - 1 + fy, ix = U.gx0 - 1 + fx;
        pin = gth && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        if (!pin) return;
        const int pi = (U.b * a.N + n) * 3;
        const WarpPoint wp = warp_point(ix, iy, a.H, a.W, prm[pi], prm[pi + 1], prm[pi + 2]);
        flx = wp.fx;
        fly = wp.fy;
        const float x0f = floorf(wp.sx), y0f = floorf(wp.sy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float wx1 = wp.sx - x0f, wy1 = wp.sy - y0f;
        const float wx[2] = {1.0f - wx1, wx1}, wy[2] = {1.0f - wy1, wy1};
        const uint16_t *slice = a.fe + ((int64_t)(U.b * a.N + n) * a.H * a.W) * frec + goct * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yc = y0 + (k >> 1), xc = x0 + (k & 1);
            const bool ok = (unsigned)yc < (unsigned)a.H && (unsigned)xc < (unsigned)a.W;
            wgt[k] = ok ? wx[k & 1] * wy[k >> 1] : 0.f;
            const uint16_t *rp = slice + (ok ? (yc * a.W + xc) * frec : 0);
#pragma unroll
            for (int i = 0; i < PARTS; ++i) {
                if constexpr (DFFW_HW_ABL & 1) q[k][i] = make_uint4(ix, iy, n, k);
                else q[k][i] = *reinterpret_cast<const uint4 *>(rp + i * CF);
            }
        }
    };
    auto land = [&](int slot) {     // blend the corners requested by the last issue(), write the octet (octet-0 threads: also the flow record)
        if (!gth) return;
        unsigned char *dst = smem + slot * SLOTB + gp * PIXB;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pin && !(DFFW_HW_ABL & 8)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (wgt[k] == 0.f) continue;      // corner outside the image (or weight exactly 0): skipped, as warp_octet skips it
                const uint4 h = q[k][0];
                uint4 l = make_uint4(0, 0, 0, 0);
                if constexpr (PARTS == 2) l = q[k][1];
                float x, y;
                Fmt<PREC>::join2(h.x, l.x, x, y); v[0] += x * wgt[k]; v[1] += y * wgt[k];
                Fmt<PREC>::join2(h.y, l.y, x, y); v[2] += x * wgt[k]; v[3] += y * wgt[k];
                Fmt<PREC>::join2(h.z, l.z, x, y); v[4] += x * wgt[k]; v[5] += y * wgt[k];
                Fmt<PREC>::join2(h.w, l.w, x, y); v[6] += x * wgt[k]; v[7] += y * wgt[k];
            }
        }
        uint4 h, l;
        Fmt<PREC>::split2(v[0], v[1], h.x, l.x);
        Fmt<PREC>::split2(v[2], v[3], h.y, l.y);
        Fmt<PREC>::split2(v[4], v[5], h.z, l.z);
        Fmt<PREC>::split2(v[6], v[7], h.w, l.w);
        *reinterpret_cast<uint4 *>(dst + goct * 16) = h;
        if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(dst + PLANEB + goct * 16) = l;
        if (goct == 0) {
            uint4 fh = make_uint4(0, 0, 0, 0), fl = fh;
            if (pin) Fmt<PREC>::split2(flx, fly, fh.x, fl.x);
            *reinterpret_cast<uint4 *>(dst + GO * 16) = fh;
            if constexpr (PARTS == 2) *reinterpret_cast<uint4 *>(dst + PLANEB + GO * 16) = fl;
        }
    };

    // ---- contraction side ----------------------------------------------------------------------------------------
    int pofs[TB], tapo[NCH];
#pragma unroll
    for (int j = 0; j < TB; ++j) pofs[j] = ((wave * TB + j) * XX + r) * PIXB;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int o = 4 * k + g, tap = o < 9 * OCT ? o / OCT : 0, oct = o < 9 * OCT ? o % OCT : 0;   // (octets >= 9 OCT carry zero weights)
        tapo[k] = ((tap / 3) * XX + tap % 3) * PIXB + oct * 16;
    }
    const unsigned char *wl = smem + W_OFF + lane * 16;
    f32x4 b0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b0[nt] = *reinterpret_cast<const f32x4 *>(a.bias + nt * 16 + g * 4);

    Unit U = decode(ufirst);
    issue(U, 0);
    int slot = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        // the reference part of this column: 4 channels per output tile of the lane's output pixels, added in front of the ReLU of every slice
        f32x4 rv[TB][NT];
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint16_t *rp = a.ref + (((int64_t)U.b * a.H + U.gy0 + wave * TB + j) * a.W + U.gx0 + r) * rec + nt * 16 + g * 4;
                const uint2 h = *reinterpret_cast<const uint2 *>(rp);
                uint2 l = make_uint2(0, 0);
                if constexpr (PARTS == 2) l = *reinterpret_cast<const uint2 *>(rp + C);
                float r0, r1, r2, r3;
                Fmt<PREC>::join2(h.x, l.x, r0, r1);
                Fmt<PREC>::join2(h.y, l.y, r2, r3);
                rv[j][nt] = f32x4{r0, r1, r2, r3} + b0[nt];
            }
        const Unit Ucur = U;
        for (int s = 0; s < a.N; ++s) {
            land(slot);
            __builtin_amdgcn_sched_barrier(0);
            // next step's corners: the following slice of this column, or the first slice of the workgroup's next column
            const bool more = s + 1 < a.N || cu + wgs_per_xcd < uend;
            if (s + 1 == a.N && more) U = decode(cu + wgs_per_xcd);
            if (more) issue(U, s + 1 < a.N ? s + 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            const unsigned char *xs = smem + slot * SLOTB;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = rv[j][nt];
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const short8 xh = *reinterpret_cast<const short8 *>(xs + pofs[j] + tapo[k]);
                    short8 xl = xh;
                    if constexpr (PARTS == 2) xl = *reinterpret_cast<const short8 *>(xs + PLANEB + pofs[j] + tapo[k]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const short8 wh = *reinterpret_cast<const short8 *>(wl + ((k * NT + nt) * PARTS) * 1024);
                        if constexpr (DFFW_HW_ABL & 4) {
                            acc[nt][0] += __builtin_bit_cast(float, (int)xh[0] + (int)xl[1] + (int)wh[0]);
                            continue;
                        }
                        if constexpr (PARTS == 2) {
                            const short8 wlo = *reinterpret_cast<const short8 *>(wl + ((k * NT + nt) * PARTS + 1) * 1024);
                            acc[nt] = mma<F16>(wlo, xh, acc[nt]);
                            acc[nt] = mma<F16>(wh, xl, acc[nt]);
                        }
                        acc[nt] = mma<F16>(wh, xh, acc[nt]);
                    }
                    if (CF == 8 ? k == 2 : ((k & 1) == 0 && k > 0)) __builtin_amdgcn_sched_barrier(0);   // (every fragment of the row in flight at once: +40 registers)
                }
                const int64_t pix = (((int64_t)Ucur.b * a.N + s) * a.H + Ucur.gy0 + wave * TB + j) * a.W + Ucur.gx0 + r;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_bits(acc[nt][0]), relu_bits(acc[nt][1]), h01, l01);
                    Fmt<PREC>::split2(relu_bits(acc[nt][2]), relu_bits(acc[nt][3]), h23, l23);
                    if ((DFFW_HW_ABL & 2) && h01 != 0x12345u) continue;
                    if constexpr (PARTS == 2) {
                        swap16(h01, l01);
                        swap16(h23, l23);
                        *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C + nt * 16 + (g >> 1) * 8) = make_uint4(h01, h23, l01, l23);
                    } else {
                        *reinterpret_cast<uint2 *>(a.out + pix * rec + nt * 16 + g * 4) = make_uint2(h01, h23);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            slot ^= 1;
        }
    }
}

void head_warp_kernel_name(int prec, int cf, char *buf, int n) { snprintf(buf, n, "dffw::head_warp_kernel<%d, %d>", prec, cf); }

hipError_t launch_head_warp(int prec, int cf, const HeadWarpArgs &a, hipStream_t s) {
    if ((int64_t)a.B * a.N > head_warp_max_planes()) return hipErrorInvalidValue;
    const int want = a.wgs > 0 ? a.wgs : (cf == 8 ? 1024 : 512);
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8))));
#define DFFW_HW_LAUNCH(P)                                                                             \
    do {                                                                                              \
        if (cf == 8) hipLaunchKernelGGL((head_warp_kernel<P, 8>), grid, dim3(256), 0, s, a);          \
        else if (cf == 16) hipLaunchKernelGGL((head_warp_kernel<P, 16>), grid, dim3(512), 0, s, a);   \
        else return hipErrorInvalidValue;                                                             \
    } while (0)
    switch (prec) {
        case P_BF16X3: DFFW_HW_LAUNCH(P_BF16X3); break;
        case P_FP16: DFFW_HW_LAUNCH(P_FP16); break;
        case P_BF16: DFFW_HW_LAUNCH(P_BF16); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_HW_LAUNCH
    return hipGetLastError();
}

// ---- of_s2: the down-sampling residual block of the alignment network at 8 -> 16 channels (End_to_End.py:135-145 with stride 2,
// `OF_feature1.0`) as ONE streaming kernel ---------------------------------------------------------------------------------------
//     out = relu( conv1x1x1_s2(x) + BN(conv1x3x3(relu(BN(conv1x3x3_s2(x))))) )
// As three launches (strided conv on conv_tile, the 1x1x1 shortcut on the gather kernel, the second conv with the shortcut as a
// residual) the full-resolution input was read twice and the two half-resolution intermediates went through HBM.  Here a workgroup
// walks the slices of a column of 8 x 16 OUTPUT pixels: the 21 x 37 input footprint of a slice (two 3x3 halos, the inner one at stride
// 2) is fetched one step ahead into registers (plain 16-byte loads, one (pixel, part) piece per thread and pass) and written into one
// of two LDS slots with the even columns of a row first, so that the stride-2 operand reads of 16 neighbouring pixels stay
// contiguous; stage A computes t = relu(BN(conv.0)) on the 10 x 18 region conv.2 needs (3 chunks, K octet g of chunk k = tap 4k + g)
// into LDS records, zero outside the image; stage B contracts t (5 chunks, srd_roll16's order) plus one chunk for the shortcut (the
// input pixel under the output pixel, already in LDS) and stores the block's output.  No LDS-DMA: hipcc counts every wait itself.
template <int PREC>
__global__ __launch_bounds__(256) void of_s2_kernel(const SrdArgs a) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int CI = 8, C = 16, TY = 8, TX = 16;
    constexpr int TYT = TY + 2, TXT = TX + 2, TPIX = TYT * TXT;               // t region
    constexpr int XY = 2 * TYT + 1, XX = 2 * TXT + 1, XPIX = XY * XX, XEV = TXT + 1;   // input footprint; XEV even columns per row
    constexpr int XPIXB = CI * 2, PIXB = C * 2;
    constexpr int XPLANEB = XPIX * XPIXB, XSLOTB = PARTS * XPLANEB, TPLANEB = TPIX * PIXB;
    constexpr int T_OFF = 2 * XSLOTB;
    constexpr int NITEM = XPIX * PARTS, NPASS = (NITEM + 255) / 256;
    constexpr int NCHA = 3, NCHB = 5, TA = 3, TB = 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[T_OFF + PARTS * TPLANEB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int ufirst, uend;
    {
        const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
        const int xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        uend = xs + q + (xcd < rem ? 1 : 0);
        ufirst = xs + widx;
    }
    if (ufirst >= uend) return;
    struct Unit {
        int b, gy0, gx0;
    };
    auto decode = [&](int u) {      // columns of the OUTPUT grid (a.H x a.W = output size; the input is 2 a.H x 2 a.W)
        Unit c;
        const int txi = u % a.tiles_x;
        const int tt = u / a.tiles_x;
        c.b = tt / a.tiles_y;
        c.gy0 = (tt % a.tiles_y) * TY;
        c.gx0 = txi * TX;
        return c;
    };
    const int Hi = 2 * a.H, Wi = 2 * a.W;
    const int rec = PARTS * C, xrec = PARTS * CI;

    // ---- fill side: item = (footprint pixel, part); thread t takes items t, t + 256, ... -------------------------------
    uint4 q[NPASS];
    int ldst[NPASS];      // LDS byte offset inside a slot of the item's piece (even columns of a row first)
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int item = tid + k * 256, part = item / XPIX, pix = item - part * XPIX;
        const int fy = pix / XX, fx = pix - fy * XX;
        ldst[k] = part * XPLANEB + (fy * XX + ((fx & 1) ? XEV + (fx >> 1) : (fx >> 1))) * XPIXB;
    }
    auto issue = [&](const Unit &U, int n) {
        const uint16_t *slice = a.x + ((int64_t)(U.b * a.N + n) * Hi * Wi) * xrec;
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int item = tid + k * 256, part = item / XPIX, pix = item - part * XPIX;
            const int fy = pix / XX, fx = pix - fy * XX;
            const int iy = 2 * U.gy0 - 3 + fy, ix = 2 * U.gx0 - 3 + fx;
            const bool ok = item < NITEM && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            q[k] = make_uint4(0, 0, 0, 0);
            if (ok) q[k] = *reinterpret_cast<const uint4 *>(slice + (iy * Wi + ix) * xrec + part * CI);
        }
    };
    auto land = [&](int slot) {
#pragma unroll
        for (int k = 0; k < NPASS; ++k)
            if (tid + k * 256 < NITEM) *reinterpret_cast<uint4 *>(smem + slot * XSLOTB + ldst[k]) = q[k];
    };

    // ---- stage A: the 10 x 18 t pixels are 12 operand tiles (the last one partly idle), three per wave ----------------
    int pa[TA], ta_y[TA], ta_x[TA], ta_st[TA];
    bool ta_ok[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        int p = (wave * TA + j) * 16 + r;
        ta_ok[j] = p < TPIX;
        if (p >= TPIX) p = TPIX - 1;
        ta_y[j] = p / TXT;
        ta_x[j] = p - ta_y[j] * TXT;
        pa[j] = (2 * ta_y[j] * XX + ta_x[j]) * XPIXB;             // footprint pixel (2 ty, 2 tx): even column tx of row 2 ty
        ta_st[j] = T_OFF + p * PIXB + g * 8;
    }
    int tapA[NCHA], tapB[NCHB];
#pragma unroll
    for (int k = 0; k < NCHA; ++k) {
        const int tap = 4 * k + g;                                 // taps >= 9 carry zero weights
        const int dy = tap < 9 ? tap / 3 : 0, dx = tap < 9 ? tap % 3 : 0;
        tapA[k] = (dy * XX + (dx == 1 ? XEV : (dx == 2 ? 1 : 0))) * XPIXB;
    }
#pragma unroll
    for (int k = 0; k < NCHB; ++k) {
        const int tap = 2 * k + (g >> 1);
        const int dy = tap < 9 ? tap / 3 : 0, dx = tap < 9 ? tap % 3 : 0;
        tapB[k] = (dy * TXT + dx) * PIXB + (g & 1) * 16;
    }
    // ---- stage B: wave w = output rows 2w, 2w+1 ---------------------------------------------------------------------
    int pbo[TB], pbx[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int oy = wave * TB + j;
        pbo[j] = (oy * TXT + r) * PIXB;
        pbx[j] = ((2 * oy + 3) * XX + XEV + r + 1) * XPIXB;       // input pixel (2 oy + 3, 2 r + 3): odd column r + 1
    }
    short8 w0[NCHA][PARTS], w2[NCHB][PARTS], wsc[PARTS];
#pragma unroll
    for (int k = 0; k < NCHA; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w0[k][pt] = reinterpret_cast<const short8 *>(a.w0)[(k * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int k = 0; k < NCHB; ++k)
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) w2[k][pt] = reinterpret_cast<const short8 *>(a.w2)[(k * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) wsc[pt] = reinterpret_cast<const short8 *>(a.w3f)[pt * 64 + lane];
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.b0 + g * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + g * 4);

    Unit U = decode(ufirst);
    issue(U, 0);
    int slot = 0;
    for (int cu = ufirst; cu < uend; cu += wgs_per_xcd) {
        const Unit Ucur = U;
        for (int s = 0; s < a.N; ++s) {
            land(slot);
            __builtin_amdgcn_sched_barrier(0);
            const bool more = s + 1 < a.N || cu + wgs_per_xcd < uend;
            if (s + 1 == a.N && more) U = decode(cu + wgs_per_xcd);
            if (more) issue(U, s + 1 < a.N ? s + 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();          // the slice is in LDS; stage B of the previous step has read t
            const unsigned char *xs = smem + slot * XSLOTB;
            // ---- stage A: t = relu(conv.0(x) + shift) on the 10 x 18 region, zero outside the image (conv.2's padding) ----------
#pragma unroll
            for (int j = 0; j < TA; ++j) {
                f32x4 acc = b0;
#pragma unroll
                for (int k = 0; k < NCHA; ++k) {
                    const short8 xh = *reinterpret_cast<const short8 *>(xs + pa[j] + tapA[k]);
                    if constexpr (PARTS == 2) {
                        const short8 xl = *reinterpret_cast<const short8 *>(xs + XPLANEB + pa[j] + tapA[k]);
                        acc = mma<F16>(w0[k][1], xh, acc);
                        acc = mma<F16>(w0[k][0], xl, acc);
                    }
                    acc = mma<F16>(w0[k][0], xh, acc);
                }
                const int iy = Ucur.gy0 - 1 + ta_y[j], ix = Ucur.gx0 - 1 + ta_x[j];
                const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                if (ta_ok[j]) {
                    uint32_t h01, h23, l01, l23;
                    Fmt<PREC>::split2(relu_lim_bits(acc[0], inside ? 0x7f800000 : 0), relu_lim_bits(acc[1], inside ? 0x7f800000 : 0), h01, l01);
                    Fmt<PREC>::split2(relu_lim_bits(acc[2], inside ? 0x7f800000 : 0), relu_lim_bits(acc[3], inside ? 0x7f800000 : 0), h23, l23);
                    *reinterpret_cast<uint2 *>(smem + ta_st[j]) = make_uint2(h01, h23);
                    if constexpr (PARTS == 2) *reinterpret_cast<uint2 *>(smem + ta_st[j] + TPLANEB) = make_uint2(l01, l23);
                }
            }
            __syncthreads();
            // ---- stage B: out = relu(conv.2(t) + shift + shortcut(x)) ------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                f32x4 acc = b2;
#pragma unroll
                for (int k = 0; k < NCHB; ++k) {
                    const short8 th = *reinterpret_cast<const short8 *>(smem + T_OFF + pbo[j] + tapB[k]);
                    if constexpr (PARTS == 2) {
                        const short8 tl = *reinterpret_cast<const short8 *>(smem + T_OFF + TPLANEB + pbo[j] + tapB[k]);
                        acc = mma<F16>(w2[k][1], th, acc);
                        acc = mma<F16>(w2[k][0], tl, acc);
                    }
                    acc = mma<F16>(w2[k][0], th, acc);
                }
                {   // the shortcut chunk: K octet 0 = the 8 channels of the input pixel under the output pixel (octets 1..3: zero weights)
                    const short8 sh = *reinterpret_cast<const short8 *>(xs + pbx[j]);
                    if constexpr (PARTS == 2) {
                        const short8 sl = *reinterpret_cast<const short8 *>(xs + XPLANEB + pbx[j]);
                        acc = mma<F16>(wsc[1], sh, acc);
                        acc = mma<F16>(wsc[0], sl, acc);
                    }
                    acc = mma<F16>(wsc[0], sh, acc);
                }
                uint32_t h01, h23, l01, l23;
                Fmt<PREC>::split2(relu_bits(acc[0]), relu_bits(acc[1]), h01, l01);
                Fmt<PREC>::split2(relu_bits(acc[2]), relu_bits(acc[3]), h23, l23);
                const int64_t pix = (((int64_t)Ucur.b * a.N + s) * a.H + Ucur.gy0 + wave * TB + j) * a.W + Ucur.gx0 + r;
                if constexpr (PARTS == 2) {
                    swap16(h01, l01);
                    swap16(h23, l23);
                    *reinterpret_cast<uint4 *>(a.out + pix * rec + (g & 1) * C + (g >> 1) * 8) = make_uint4(h01, h23, l01, l23);
                } else {
                    *reinterpret_cast<uint2 *>(a.out + pix * rec + g * 4) = make_uint2(h01, h23);
                }
            }
            slot ^= 1;
        }
    }
}

void of_s2_kernel_name(int prec, char *buf, int n) { snprintf(buf, n, "dffw::of_s2_kernel<%d>", prec); }

hipError_t launch_of_s2(int prec, const SrdArgs &a, hipStream_t s) {
    const int want = a.wgs > 0 ? a.wgs : 512;
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
    switch (prec) {
        case P_BF16X3: hipLaunchKernelGGL((of_s2_kernel<P_BF16X3>), grid, block, 0, s, a); break;
        case P_FP16: hipLaunchKernelGGL((of_s2_kernel<P_FP16>), grid, block, 0, s, a); break;
        case P_BF16: hipLaunchKernelGGL((of_s2_kernel<P_BF16>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- srd_attention_mfma: the attention tail of the 32-channel SRD block (`FM_conv2.1.N_ch_attention`) on the matrix cores ------
//     out = feat + relu(conv1x1x1(relu(conv3x1x1(feat))))     (DEN.py:322-329; no BatchNorm, no bias)
// The fused VALU kernel (srd_attention_kernel) stops at 16 channels (C*C*4 FMAs per pixel); at 32 channels the two convs ran as
// two gather-GEMM launches with the intermediate in HBM.  Pointwise in space, so no LDS: a wave owns 16 consecutive pixels of a
// row and walks the slices with feat[z-1], feat[z], feat[z+1] as MFMA operand fragments in registers (each record is read once);
// conv3x1x1 = 3 chunks (one per slice) x 2 output tiles; its ReLU'd result, split to hi/lo in registers, is the operand of the
// 1x1x1 conv in place (K octet = the lane's own 4 channels as [hi | lo], fragments [w_hi w_hi] and [w_lo 0]).
template <int PREC>
__global__ __launch_bounds__(256) void srd_attention_mfma(const uint16_t *__restrict__ feat, uint16_t *__restrict__ out,
                                                          const uint16_t *__restrict__ w3f, const uint16_t *__restrict__ w1f, int B, int N,
                                                          int H, int W) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int C = 32, REC = PARTS * C;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int64_t strip = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strips_per_row = W / 16;
    const int64_t nstrips = (int64_t)B * H * strips_per_row;
    if (strip >= nstrips) return;
    const int sx = (int)(strip % strips_per_row);
    const int64_t by = strip / strips_per_row;
    const int y = (int)(by % H), b = (int)(by / H);
    const int64_t hw = (int64_t)H * W;
    const int64_t pix0 = ((int64_t)b * N * H + y) * W + sx * 16 + r;   // the lane's pixel in slice 0

    short8 w3[3][2][PARTS], w1[2][2][2];   // [slice chunk][output tile][part], [channel chunk][fragment][output tile]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) w3[k][nt][pt] = reinterpret_cast<const short8 *>(w3f)[((k * 2 + nt) * PARTS + pt) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int fr = 0; fr < PARTS; ++fr)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) w1[c][fr][nt] = reinterpret_cast<const short8 *>(w1f)[((c * PARTS + fr) * 2 + nt) * 64 + lane];

    auto load = [&](int z, short8 (&f)[PARTS]) {   // the lane's K octet (channels 8g..8g+7) of feat[z]; zeros outside the stack
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) f[pt] = short8{0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)z < (unsigned)N) {
            const uint16_t *p = feat + (pix0 + (int64_t)z * hw) * REC + g * 8;
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) f[pt] = *reinterpret_cast<const short8 *>(p + pt * C);
        }
    };
    // feat[z] at the lane's RESULT channels nt*16 + 4g .. +3 (the residual; the operand octet holds other channels): requested one slice ahead
    auto load_res = [&](int z, uint2 (&rh)[2], uint2 (&rl)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            rh[nt] = make_uint2(0, 0);
            rl[nt] = make_uint2(0, 0);
            if (z < N) {
                const uint16_t *cp = feat + (pix0 + (int64_t)z * hw) * REC + nt * 16 + g * 4;
                rh[nt] = *reinterpret_cast<const uint2 *>(cp);
                if constexpr (PARTS == 2) rl[nt] = *reinterpret_cast<const uint2 *>(cp + C);
            }
        }
    };
    // The slice walk is a chain of dependent loads unless they run ahead: operand records two slices ahead, the residual one slice ahead (the
    // kernel has only 8 waves of work per SIMD, 3 resident: with each slice's loads consumed in the iteration that issued them it ran at 3.1 TB/s)
    short8 f[4][PARTS];
    uint2 rh[2][2], rl[2][2];
    load(-1, f[0]);
    load(0, f[1]);
    load(1, f[2]);
    load_res(0, rh[0], rl[0]);
    for (int z = 0; z < N; ++z) {
        load(z + 2, f[3]);
        load_res(z + 1, rh[1], rl[1]);
        f32x4 at[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            at[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if constexpr (PARTS == 2) {
                    at[nt] = mma<F16>(w3[k][nt][1], f[k][0], at[nt]);
                    at[nt] = mma<F16>(w3[k][nt][0], f[k][1], at[nt]);
                }
                at[nt] = mma<F16>(w3[k][nt][0], f[k][0], at[nt]);
            }
        }
        short8 b2[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t ah01, ah23, al01, al23;
            Fmt<PREC>::split2(relu_bits(at[c][0]), relu_bits(at[c][1]), ah01, al01);
            Fmt<PREC>::split2(relu_bits(at[c][2]), relu_bits(at[c][3]), ah23, al23);
            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
            const u32x4v bq = {ah01, ah23, al01, al23};
            b2[c] = __builtin_bit_cast(short8, bq);
        }
        const int64_t pix = pix0 + (int64_t)z * hw;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (PARTS == 2) o = mma<F16>(w1[c][1][nt], b2[c], o);
                o = mma<F16>(w1[c][0][nt], b2[c], o);
            }
            float c0, c1, c2, c3;
            Fmt<PREC>::join2(rh[0][nt].x, rl[0][nt].x, c0, c1);
            Fmt<PREC>::join2(rh[0][nt].y, rl[0][nt].y, c2, c3);
            uint32_t h01, h23, l01, l23;
            Fmt<PREC>::split2(c0 + relu_bits(o[0]), c1 + relu_bits(o[1]), h01, l01);
            Fmt<PREC>::split2(c2 + relu_bits(o[2]), c3 + relu_bits(o[3]), h23, l23);
            if constexpr (PARTS == 2) {
                swap16(h01, l01);
                swap16(h23, l23);
                *reinterpret_cast<uint4 *>(out + pix * REC + (g & 1) * C + (nt * 2 + (g >> 1)) * 8) = make_uint4(h01, h23, l01, l23);
            } else {
                *reinterpret_cast<uint2 *>(out + pix * REC + nt * 16 + g * 4) = make_uint2(h01, h23);
            }
        }
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            f[0][pt] = f[1][pt];
            f[1][pt] = f[2][pt];
            f[2][pt] = f[3][pt];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            rh[0][nt] = rh[1][nt];
            rl[0][nt] = rl[1][nt];
        }
    }
}

hipError_t launch_srd_attention_mfma(int prec, const uint16_t *feat, uint16_t *out, const uint16_t *w3f, const uint16_t *w1f, int B, int N,
                                     int H, int W, hipStream_t s) {
    if (W % 16) return hipErrorInvalidValue;
    const int64_t nstrips = (int64_t)B * H * (W / 16);
    const dim3 grid((unsigned)((nstrips + 3) / 4)), block(256);
    switch (prec) {
        case P_BF16X3: hipLaunchKernelGGL((srd_attention_mfma<P_BF16X3>), grid, block, 0, s, feat, out, w3f, w1f, B, N, H, W); break;
        case P_FP16: hipLaunchKernelGGL((srd_attention_mfma<P_FP16>), grid, block, 0, s, feat, out, w3f, w1f, B, N, H, W); break;
        case P_BF16: hipLaunchKernelGGL((srd_attention_mfma<P_BF16>), grid, block, 0, s, feat, out, w3f, w1f, B, N, H, W); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

void srd_roll_tile(int *ty, int *tx) {
    *ty = 8;
    *tx = 16;
}

void srd_roll_kernel_name(int prec, bool pool, char *buf, int n) { snprintf(buf, n, "dffw::srd_roll_kernel<%d, %s>", prec, pool ? "true" : "false"); }

void srd_roll16_tile(int *ty, int *tx) {
    *ty = 4;
    *tx = 16;
}

void srd_roll16_kernel_name(int prec, bool pool, char *buf, int n) { snprintf(buf, n, "dffw::srd_roll16_kernel<%d, %s>", prec, pool ? "true" : "false"); }
void srd_pipe16_kernel_name(int prec, bool pool, char *buf, int n) { snprintf(buf, n, "dffw::srd_pipe16_kernel<%d, %s, 0>", prec, pool ? "true" : "false"); }

hipError_t launch_srd_pipe16(int prec, const SrdArgs &a, hipStream_t s) {
    const int want = a.wgs > 0 ? a.wgs : 512;   // two resident workgroups per CU
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
#define DFFW_SRDP16_LAUNCH(P)                                                                     \
    do {                                                                                          \
        if (a.pooled) hipLaunchKernelGGL((srd_pipe16_kernel<P, true>), grid, block, 0, s, a);     \
        else hipLaunchKernelGGL((srd_pipe16_kernel<P, false>), grid, block, 0, s, a);             \
    } while (0)
    switch (prec) {
        case P_BF16X3: DFFW_SRDP16_LAUNCH(P_BF16X3); break;
        case P_FP16: DFFW_SRDP16_LAUNCH(P_FP16); break;
        case P_BF16: DFFW_SRDP16_LAUNCH(P_BF16); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_SRDP16_LAUNCH
    return hipGetLastError();
}

hipError_t launch_srd_roll16(int prec, const SrdArgs &a, hipStream_t s) {
    const int want = a.wgs > 0 ? a.wgs : 512;   // two resident workgroups per CU
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
#ifdef DFFW_ABL_BUILD   // development (make ABL=1): timing ablations, selected with DFFW_SRD_ABL (wrong results with any bit set)
    const char *az = getenv("DFFW_SRD_ABL");
    const int abl = az ? atoi(az) : 0;
#define DFFW_SRD16_ABL_CASES                                                                                      \
    case 1: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 1>), grid, block, 0, s, a); break;              \
    case 2: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 2>), grid, block, 0, s, a); break;              \
    case 4: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 4>), grid, block, 0, s, a); break;              \
    case 7: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 7>), grid, block, 0, s, a); break;              \
    case 8: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 8>), grid, block, 0, s, a); break;              \
    case 16: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 16>), grid, block, 0, s, a); break;            \
    case 32: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 32>), grid, block, 0, s, a); break;            \
    case 40: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 40>), grid, block, 0, s, a); break;            \
    case 23: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 23>), grid, block, 0, s, a); break;            \
    case 6: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true, 6>), grid, block, 0, s, a); break;
#else
    const int abl = 0;
#define DFFW_SRD16_ABL_CASES
#endif
#define DFFW_SRD16_LAUNCH(P)                                                                \
    do {                                                                                    \
        if (a.pooled && P == P_BF16X3 && abl) {                                             \
            switch (abl) {                                                                  \
                DFFW_SRD16_ABL_CASES                                                        \
                default: hipLaunchKernelGGL((srd_roll16_kernel<P_BF16X3, true>), grid, block, 0, s, a);            \
            }                                                                               \
        } else if (a.pooled) hipLaunchKernelGGL((srd_roll16_kernel<P, true>), grid, block, 0, s, a);  \
        else hipLaunchKernelGGL((srd_roll16_kernel<P, false>), grid, block, 0, s, a);          \
    } while (0)
    switch (prec) {
        case P_BF16X3: DFFW_SRD16_LAUNCH(P_BF16X3); break;
        case P_FP16: DFFW_SRD16_LAUNCH(P_FP16); break;
        case P_BF16: DFFW_SRD16_LAUNCH(P_BF16); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_SRD16_LAUNCH
#undef DFFW_SRD16_ABL_CASES
    return hipGetLastError();
}

void of_roll8_kernel_name(int prec, char *buf, int n) { snprintf(buf, n, "dffw::of_roll8_kernel<%d>", prec); }

hipError_t launch_of_roll8(int prec, const SrdArgs &a, hipStream_t s) {
    const int want = a.wgs > 0 ? a.wgs : 768;
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
    switch (prec) {
        case P_BF16X3: hipLaunchKernelGGL((of_roll8_kernel<P_BF16X3>), grid, block, 0, s, a); break;
        case P_FP16: hipLaunchKernelGGL((of_roll8_kernel<P_FP16>), grid, block, 0, s, a); break;
        case P_BF16: hipLaunchKernelGGL((of_roll8_kernel<P_BF16>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

void of_roll_kernel_name(int prec, bool cin8, char *buf, int n, bool sums) {
    if (sums) snprintf(buf, n, "dffw::of_roll_kernel<%d, %s, true>", prec, cin8 ? "true" : "false");
    else snprintf(buf, n, "dffw::of_roll_kernel<%d, %s>", prec, cin8 ? "true" : "false");
}

hipError_t launch_of_roll(int prec, bool cin8, const SrdArgs &a, hipStream_t s, bool sums) {
    const int want = a.wgs > 0 ? a.wgs : (cin8 ? 768 : 512);
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
    if (sums && cin8) return hipErrorInvalidValue;
#define DFFW_OF_LAUNCH(P)                                                                   \
    do {                                                                                    \
        if (sums) hipLaunchKernelGGL((of_roll_kernel<P, false, true>), grid, block, 0, s, a); \
        else if (cin8) hipLaunchKernelGGL((of_roll_kernel<P, true>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((of_roll_kernel<P, false>), grid, block, 0, s, a);          \
    } while (0)
    switch (prec) {
        case P_BF16X3: DFFW_OF_LAUNCH(P_BF16X3); break;
        case P_FP16: DFFW_OF_LAUNCH(P_FP16); break;
        case P_BF16: DFFW_OF_LAUNCH(P_BF16); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_OF_LAUNCH
    return hipGetLastError();
}

hipError_t launch_srd_roll(int prec, const SrdArgs &a, hipStream_t s) {
    const int want = a.wgs > 0 ? a.wgs : 768;   // three resident workgroups per CU
    const int per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * std::min(per_xcd, std::max(1, want / 8)))), block(256);
#ifdef DFFW_ABL_BUILD   // development (make ABL=1): timing ablations, selected with DFFW_SRD_ABL (wrong results with any bit set)
    const char *az = getenv("DFFW_SRD_ABL");
    const int abl = az ? atoi(az) : 0;
#define DFFW_SRD_ABL_CASES \
    case 1: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 1>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 2: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 2>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 4: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 4>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 6: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 6>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 7: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 7>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 8: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 8>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 16: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 16>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 32: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 32>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 40: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 40>), grid, block, 0, s, a, a.w3, a.w1); break; \
    case 23: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true, 23>), grid, block, 0, s, a, a.w3, a.w1); break; \

#else
    const int abl = 0;
#define DFFW_SRD_ABL_CASES
#endif
#define DFFW_SRD_LAUNCH(P)                                                                    \
    do {                                                                                      \
        if (a.pooled && P == P_BF16X3 && abl) {                                               \
            switch (abl) {                                                                    \
                DFFW_SRD_ABL_CASES                                                            \
                default: hipLaunchKernelGGL((srd_roll_kernel<P_BF16X3, true>), grid, block, 0, s, a, a.w3, a.w1);            \
            }                                                                                 \
        } else if (a.pooled) hipLaunchKernelGGL((srd_roll_kernel<P, true>), grid, block, 0, s, a, a.w3, a.w1);   \
        else hipLaunchKernelGGL((srd_roll_kernel<P, false>), grid, block, 0, s, a, a.w3, a.w1);           \
    } while (0)
    switch (prec) {
        case P_BF16X3: DFFW_SRD_LAUNCH(P_BF16X3); break;
        case P_FP16: DFFW_SRD_LAUNCH(P_FP16); break;
        case P_BF16: DFFW_SRD_LAUNCH(P_BF16); break;
        default: return hipErrorInvalidValue;
    }
#undef DFFW_SRD_LAUNCH
#undef DFFW_SRD_ABL_CASES
    return hipGetLastError();
}

}  // namespace dffw
