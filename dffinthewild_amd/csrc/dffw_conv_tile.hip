// conv_tile: LDS-tiled implicit-GEMM convolution for gfx950 (MI355X).
//
// One workgroup (4 waves) produces a TZ x TY x TX block of output grid points of one focal-stack
// sample, for ALL output channels.  Per channel group ("stage") of CG input channels it
//   1. stages the input footprint of the block (block + filter halo, zero-filled outside the
//      volume) into LDS by LDS-DMA (global_load_lds_dwordx4, every load of the stage in flight at once,
//      no VGPR round trip), hi and lo halves of split-bf16 in separate planes;
//   2. runs the contraction over (tap, channel) in 32-deep chunks on the matrix cores: the
//      activation operand of v_mfma_f32_16x16x32 is a single ds_read_b128 at
//      (grid-point base + tap offset), so each byte fetched from HBM/L2 is reused by all 27 taps
//      out of LDS instead of being re-gathered from the vector L1;
//      the weight operand comes pre-packed in fragment order from L2 (one 16-byte load per lane,
//      register double-buffered one chunk ahead).
// The transposed conv runs its 4 sub-pixel output phases as 4 passes over the same LDS image.
// Stride-2 convs store the footprint with even/odd columns de-interleaved so that the 16 lanes of
// an operand read stay on consecutive LDS addresses (bank-conflict free).
// Epilogue as in conv_igemm: BatchNorm shift, residual adds, ReLU, split to the storage format,
// 8-byte channels-last stores.  Workgroup -> tile mapping is XCD-aware: each of the 8 XCDs walks a
// contiguous range of tiles so neighbouring tiles' halos hit in that XCD's L2.
#include <cstdio>
#include <cstdlib>

#include "dffw_conv_geom.h"

#ifndef DFFW_TILE_PREC
#error "compile with -DDFFW_TILE_PREC=0|1|2 (arithmetic of this object, see the Makefile)"
#endif

namespace dffw {

// NWAVES waves per workgroup: 4 for the 320-point tiles, 8 for the "wide" 640-point variants (same work per
// wave, one more resident wave per SIMD for the same LDS, smaller halo share)
// SPLITK: the split-K variant (raw fp32 partial sums, stage range from blockIdx.z).  Compile-time because as a
// runtime branch its partial-store path cost every kernel ~50 VGPRs at the peak (one resident wave per SIMD on the
// 64-channel kernels); only the configurations that few-tile layers actually use are instantiated with it.
// LEAN: the launch's epilogue is one the straight-line routine covers (tile_lean(): split-bf16 storage, out / out_pre / fused
// classifier, at most one residual in the output's geometry, ReLU after it): epilogue_lean_t instead of epilogue_quad's run-time
// option tree (which costs ~1000 cycles per operand tile and 16-channel group)
// KT (round 6, "teams"): the split of the contraction depth INSIDE the workgroup -- KT teams of NWAVES waves, team z owns the channel-group stages
// [z nstage / KT, (z + 1) nstage / KT) (the split-K partition) and its own LDS image; the teams' accumulators meet in LDS (team 0 adds them in team
// order and runs the epilogue): no fp32 partials through memory and no splitk_finish launch behind the kernel, which is what the few-tile layers of a
// batch-1 forward pay for split-K (28 of its 89 launches, 5-6 us each on a chain of dependent 11 us launches).  nstage % KT == 0 (the teams meet at
// the same workgroup barriers; the engine launches one stage per team), ONE pass per workgroup: the transposed conv only with its passes split over grid.z
// (TileArgs::pass_split), where a team per 32-channel stage replaces the walk over the stages.
template <int PREC, int GEO, int NT, int TZ, int TY, int TX, int CG, int PIPE, int NWAVES = 4, bool SPLITK = false, bool LEAN = false, int KT = 1>
__global__ __launch_bounds__(NWAVES * KT * 64) void conv_tile(const ConvArgs a, const TileArgs t) {
    using T = TileT<GEO, TZ, TY, TX, CG>;
    using G = GeoT<GEO>;
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int MTW = T::MT / NWAVES;  // operand tiles (16 grid points) per wave
    constexpr bool STEM = (GEO == G2D || GEO == G2P);   // the stem, per pixel or (STEMP) per pixel pair
    constexpr bool STEMP = (GEO == G2P);
    // the per-slice 1x3x3 geometry never splits K: its "SPLITK" instantiations are the row-sums variant (third conv of an alignment
    // head, End_to_End.py:41-46, whose result only feeds the head's last conv + plane mean = plane sums, dffw_kernels.hip "alpha head
    // tail"): nothing is stored but, per output row segment of 16 pixels, its sum and its first and last pixel (a.outf)
    constexpr bool SUMS = (GEO == G2S1) && SPLITK;
    constexpr int CG8 = CG / 8;

    // LDS image: PARTS planes (hi, lo) of [footprint pixel][CG channels], 16-bit.  With CG = 8 a pixel is
    // one 16-byte slot; with CG = 16 the lane groups 0/1 (2/3) of an operand read take the two channel
    // octets of the SAME tap, so for the hardware's ds_read_b128 service groups the 16 lanes always fall
    // on 16 distinct 16-byte bank groups: conflict-free without padding or swizzle, and the operand
    // address is linear: (pixel of the grid point + pixel offset of the tap) * PIXB + octet*16.
    // The image is filled by LDS-DMA (global_load_lds_dwordx4; destination = wave-uniform base +
    // lane*16, i.e. linear in chunk order [part][pixel][octet]).
    constexpr int PIXB = CG * 2;                          // bytes per pixel per plane
    // plane stride: padded to whole wave instructions (1 KiB) so the tail of the hi plane's last DMA piece
    // (zeros) can never land on lo-plane pixels
    constexpr int PLANEB = (T::FPIX * PIXB + 1023) / 1024 * 1024;
    constexpr int LDSB = PARTS * PLANEB;
    static_assert(PLANEB < 65536, "lo-plane offset must fit the ds_read immediate");
    static_assert(KT == 1 || (!SPLITK && GEO != G2D && GEO != G2P), "teams: conv geometries (the transposed conv only with its passes split over grid.z), no split-K / raw variants");
    static_assert(KT * LDSB <= 160 * 1024, "the teams' LDS images must fit one CU");
    __shared__ __attribute__((aligned(1024))) unsigned char smem_all[KT * LDSB];

    const int team = KT > 1 ? (int)threadIdx.x / (NWAVES * 64) : 0;       // (wave-uniform)
    unsigned char *const smem = smem_all + team * LDSB;                   // this team's image
    const int tid = KT > 1 ? (int)threadIdx.x % (NWAVES * 64) : (int)threadIdx.x;   // thread, wave within the team
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    // output-channel split (few-tile layers): blockIdx.y selects which NT of the layer's t.nt_total 16-channel
    // tiles this workgroup produces, so small grids still spread over the chip
    const int ntb = blockIdx.y * NT;
    const int NTT = t.nt_total;

    // ---- this workgroup's tile.  XCD x (= blockIdx % 8) owns a contiguous range of the tile sequence (x
    // fastest, then y, z, sample) so neighbouring tiles' halos hit in that XCD's L2.  (A variant walking
    // several tiles per workgroup, with the next tile's DMA queued before the epilogue, was measured slower:
    // hipcc keeps the DMA address state live across tiles, 104 -> 163 VGPRs, one wave per SIMD lost.) ------
    int tile;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = t.total_tiles >> 3, rem = t.total_tiles & 7;
        const int xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        const int xe = xs + q + (xcd < rem ? 1 : 0);
        tile = xs + idx;
        if (tile >= xe) return;
    }

    struct Coord {
        int b, gz0, gy0, gx0;
    };
    auto decode = [&](int tile) {
        Coord c;
        const int txi = tile % t.tiles_x;
        int tt = tile / t.tiles_x;
        const int tyi = tt % t.tiles_y;
        tt /= t.tiles_y;
        const int tzi = tt % t.tiles_z;
        c.b = tt / t.tiles_z;
        c.gz0 = tzi * TZ;
        c.gy0 = tyi * TY;
        c.gx0 = txi * TX;
        return c;
    };

    // ---- per-lane constants of each of this wave's operand tiles: LDS byte offset of its grid point, its
    // pixel offset inside the output volume, and its packed tile coordinates (for edge tiles only) --------
    int pofs[MTW], voff[MTW], tcrd[MTW];
    const int lanepart = (PARTS == 2) ? (g & 1) * a.Cout + (g >> 1) * 8 : g * 4;   // this lane's piece of a pixel record
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (wave * MTW + j) * 16 + r;
        if constexpr (STEMP) {
            // pair k of a row = pixels (x, x+2) with x = (k & 1) + 4 * (k >> 1); its records sit at packed column k (+ jx per
            // filter column pair).  Lane rows 0-1 (g = 0, 1) end up with pixel x, rows 2-3 with pixel x+2: each as "row g & 1"
            // of its own 8-channel record.
            constexpr int HP = TX / 2;
            const int k = p % HP, ty = (p / HP) % TY, tz = p / (HP * TY);
            const int tx = (k & 1) + 4 * (k >> 1) + 2 * (g >> 1);
            pofs[j] = ((tz * T::FY + ty) * T::FXL + k) * PIXB;
            voff[j] = ((tz * a.Ho + ty) * a.Wo + tx) * (PARTS * 8) + ((PARTS == 2) ? (g & 1) * 8 : (g & 1) * 4);
            tcrd[j] = tx | (ty << 8) | (tz << 16);
        } else {
            const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
            pofs[j] = ((tz * T::FY + ty * G::S) * T::FXL + tx) * PIXB;
            voff[j] = ((tz * a.Ho + ty * G::OS) * a.Wo + tx * G::OS) * (PARTS * a.Cout) + lanepart;
            tcrd[j] = tx | (ty << 8) | (tz << 16);
        }
    }
    // BatchNorm shift of this lane's 4 output channels per 16-channel tile: the accumulators start from it
    // (not for the 4-pass transposed conv: there hipcc then allocates three accumulator sets)
    constexpr bool BIAS_IN_ACC = (GEO != G3T);
    f32x4 bias4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias4[nt] = *reinterpret_cast<const f32x4 *>(a.bias + (ntb + nt) * 16 + g * 4);

    const int ps0 = PARTS * a.C0, ps1 = PARTS * a.C1;
    const int64_t samp0 = (int64_t)a.Ni * a.Hi * a.Wi * ps0, samp1 = (int64_t)a.Ni * a.Hi * a.Wi * ps1;

    // ---- LDS-DMA of the footprint of channel group `st` of tile `c` (issue only: no wait, no barrier) ----
    // One wave instruction fills 64 consecutive 16-byte chunks of one plane = PPW consecutive footprint
    // pixels.  The (expensive) pixel decode + bounds check + global address is done once per lane and
    // feeds the DMA of the hi plane and of the lo plane.
    auto issue_fill = [&](const Coord &c, int st) {
        constexpr int PPW = 64 / CG8;                                  // pixels per wave instruction
        constexpr int NPI = (T::FPIX + PPW * NWAVES - 1) / (PPW * NWAVES);  // iterations over the pixel list
        const int iz0 = c.gz0 + G::MINZ, iy0 = c.gy0 * G::S + G::MINY, ix0 = c.gx0 * G::S + G::MINX;  // footprint origin
        const int c8 = lane % CG8;
        const int ch = st * CG + c8 * 8;
        const bool second = ch >= a.C0;
        const int cc = second ? ch - a.C0 : ch;
        const int csrc = second ? a.C1 : a.C0;
        const uint16_t *sp = second ? a.in1 + c.b * samp1 : a.in0 + c.b * samp0;
        // footprint coordinates of this lane's pixel: decoded once, then advanced incrementally by the constant
        // step of PPW*NWAVES pixels per iteration (no div/mod in the loop); row/column bounds tests are skipped
        // for tiles whose footprint lies inside the image (wave-uniform), the slice test is always per lane
        constexpr int STEP = PPW * NWAVES;
        constexpr int DLX = STEP % T::FXL, DFY = (STEP / T::FXL) % T::FY, DFZ = STEP / (T::FXL * T::FY);
        const int p0 = wave * PPW + lane / CG8;
        int lx = p0 % T::FXL, fy = (p0 / T::FXL) % T::FY, fz = p0 / (T::FXL * T::FY);
        const bool yx_in = iy0 >= 0 && iy0 + T::FY <= a.Hi && ix0 >= 0 && ix0 + T::FX <= a.Wi;
        const int pst = PARTS * csrc;
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int pbase = (it * NWAVES + wave) * PPW;              // wave-uniform first pixel
            if (pbase >= T::FPIX) break;
            const int fx = (G::S == 2) ? (lx < T::FXL / 2 ? 2 * lx : 2 * (lx - T::FXL / 2) + 1) : lx;
            const int iz = iz0 + fz, iy = iy0 + fy, ix = ix0 + fx;
            bool ok = (unsigned)iz < (unsigned)a.Ni;
            if ((it + 1) * STEP > T::FPIX) ok = ok && (pbase + lane / CG8 < T::FPIX);   // only the last iteration can run past the image
            if (T::FXL > T::FX) ok = ok && fx < T::FX;                                   // pad column of the de-interleaved layout
            if (!yx_in) ok = ok && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const uint16_t *gp = sp + ((int64_t)(((iz * a.Hi + iy) * a.Wi + ix) * pst) + cc);
#pragma unroll
            for (int part = 0; part < PARTS; ++part) {
                const uint16_t *src = ok ? gp + part * csrc : a.zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + part * PLANEB + pbase * PIXB), 16, 0, 0);
            }
            lx += DLX;
            if (lx >= T::FXL) {
                lx -= T::FXL;
                fy += 1;
            }
            fy += DFY;
            if (fy >= T::FY) {
                fy -= T::FY;
                fz += 1;
            }
            fz += DFZ;
        }
    };

    // debug timeline (DFFW_TRACE_LAYER): wave 0 stamps s_memtime at the phase boundaries of its tile
    // (compiled in only with -DDFFW_TRACE_BUILD: the checks cost a few percent on the large layers)
    auto stamp = [&](int k) {
#ifdef DFFW_TRACE_BUILD
        if (a.trace && tid == 0 && team == 0) a.trace[(int64_t)tile * 8 + k] = __builtin_amdgcn_s_memtime();
#else
        (void)k;
#endif
    };
    stamp(0);
    const Coord cur = decode(tile);
    // split-K (few-tile layers with a deep contraction, see Run::conv): blockIdx.z owns a contiguous range of the
    // channel-group stages and writes raw fp32 partial sums; splitk_finish adds them up and runs the epilogue
    const bool splitk = SPLITK && !STEM && !SUMS && t.ksplit > 1;
    const int st_lo = KT > 1 ? team * t.nstage / KT : splitk ? (int)blockIdx.z * t.nstage / t.ksplit : 0;
    const int st_hi = KT > 1 ? (team + 1) * t.nstage / KT : splitk ? ((int)blockIdx.z + 1) * t.nstage / t.ksplit : t.nstage;
    // pass split (transposed conv, few tiles): the 4 sub-pixel passes write disjoint output phases, so they can be
    // 4 workgroups instead of a 4x longer chain in one
    const int pass_lo = (G::NPASS > 1 && t.pass_split) ? (int)blockIdx.z : 0;
    const int pass_hi = (G::NPASS > 1 && t.pass_split) ? pass_lo + 1 : G::NPASS;
    // ---- stem: footprint straight from the fp32 focal stack.  Record q of the virtual (W+2)-wide paired volume is
    // RGB(pixel q-2) | RGB(pixel q) (see stack_in_kernel); each thread gathers its records' six values from the
    // three colour planes (consecutive lanes = consecutive columns), splits them and writes hi/lo to the two LDS
    // planes.  Saves writing and re-reading the 32 B/pixel record volume (12 B/pixel of stack instead). ----
    auto fill_from_stack = [&](const Coord &c) {
        const int W = a.Wi - 2;
        const int64_t plane = (int64_t)a.Ni * a.Hi * W;
        const float *src = a.fs32 + (int64_t)c.b * 3 * plane + (int64_t)c.gz0 * a.Hi * W;
        constexpr bool rawmode = STEM && SPLITK;   // the stem's "SPLITK" instantiation is its raw-source variant (it never splits K)
        RawStack rs{};
        if constexpr (rawmode) rs = *reinterpret_cast<const RawStack *>(a.fs32);   // wave-uniform: scalar loads
        const int64_t rbase = c.b * rs.sb + c.gz0 * rs.sn;
        // value of colour ch at (row iy, column q) of the padded stack; zero outside it (the conv's padding)
        auto px = [&](int ch, int iy, int q) -> float {
            if ((unsigned)q >= (unsigned)W) return 0.f;
            if constexpr (!rawmode) return src[(int64_t)iy * W + ch * plane + q];
            if (iy >= rs.h || q >= rs.w) return -1.f;     // the loaders' pad value (test_Dataloader.py:126-137)
            const int64_t o = rbase + iy * rs.sy + q * rs.sx + ch * rs.sc;
            const float v = (rs.dtype & 1) == 0 ? (float)reinterpret_cast<const uint8_t *>(rs.p)[o] : reinterpret_cast<const float *>(rs.p)[o];
            // the FS6 loader normalises in float64 and rounds once (DFFW_RAW_NORM_F64, test_Dataloader.py:31-39) ...
            if (rs.dtype & DFFW_RAW_NORM_F64_BIT) return (float)__dsub_rn(__ddiv_rn((double)v, 127.5), 1.0);
            return __fsub_rn(__fdiv_rn(v, 127.5f), 1.0f);       // ... the others: float32 divide, then subtract (as dffw_pack_stack)
        };
        const int iy0 = c.gy0 + G::MINY, ix0 = c.gx0 + G::MINX;
        if constexpr (STEMP && !rawmode) {
            // pair form from the fp32 stack: the records at packed columns 2m, 2m+1 of a row are pixels (4m-2, 4m) and
            // (4m-1, 4m+1) past the footprint origin = the 4 consecutive, 16-byte aligned pixels from column gx0 - 8 + 4m:
            // one float4 load per colour plane makes two records, and a quad is inside the image or outside it as a whole
            constexpr int QR = T::FXL / 2, NQ = T::FY * QR;
            constexpr int NITQ = (NQ + NWAVES * 64 - 1) / (NWAVES * 64);
#pragma unroll
            for (int it = 0; it < NITQ; ++it) {
                const int p2 = tid + it * NWAVES * 64;
                if (p2 >= NQ) break;
                const int fy = p2 / QR, m = p2 - fy * QR;
                const int iy = iy0 + fy, x = c.gx0 - 8 + 4 * m;
                const bool in = (unsigned)iy < (unsigned)a.Hi && (unsigned)x < (unsigned)W;
                short8 ha = short8{0, 0, 0, 0, 0, 0, 0, 0}, la = ha, hb = ha, lb = ha;
                if (in) {
                    const float *sp = src + (int64_t)iy * W + x;
                    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(sp), c1 = *reinterpret_cast<const f32x4 *>(sp + plane),
                                c2 = *reinterpret_cast<const f32x4 *>(sp + 2 * plane);
                    // record = [c0 c1 c2 0] of its first pixel | [c0 c1 c2 0] of its second: four packed pairs per part, split with the
                    // hardware pack (split2: v_cvt_pk_bf16_f32, 5 instructions per pair against ~10 per value in software)
                    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
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
                if constexpr (PARTS == 2) {
                    *reinterpret_cast<short8 *>(dst + PLANEB) = la;
                    *reinterpret_cast<short8 *>(dst + PLANEB + PIXB) = lb;
                }
            }
            return;
        }
        constexpr int NIT = (T::FPIX + NWAVES * 64 - 1) / (NWAVES * 64);
#pragma unroll   // constant trip count: the loads of all iterations can be in flight together
        for (int it = 0; it < NIT; ++it) {
            const int p = tid + it * NWAVES * 64;
            if (p >= T::FPIX) break;
            const int fy = p / T::FXL, fl = p - fy * T::FXL;
            const int fx = STEMP ? 4 * (fl >> 1) + (fl & 1) : fl;   // pair form keeps the footprint columns = 0,1 mod 4 only
            const int iy = iy0 + fy, q = ix0 + fx;
            short8 h = short8{0, 0, 0, 0, 0, 0, 0, 0}, l = h;
            if ((unsigned)iy < (unsigned)a.Hi) {
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                uint32_t hh[4], ll[4];   // same packing as the float4 path above (the two must round alike: forward_raw == pack_stack + forward)
                Fmt<PREC>::split2(px(0, iy, q - 2), px(1, iy, q - 2), hh[0], ll[0]);
                Fmt<PREC>::split2(px(2, iy, q - 2), 0.f, hh[1], ll[1]);
                Fmt<PREC>::split2(px(0, iy, q), px(1, iy, q), hh[2], ll[2]);
                Fmt<PREC>::split2(px(2, iy, q), 0.f, hh[3], ll[3]);
                h = __builtin_bit_cast(short8, (u32x4v){hh[0], hh[1], hh[2], hh[3]});
                l = __builtin_bit_cast(short8, (u32x4v){ll[0], ll[1], ll[2], ll[3]});
            }
            *reinterpret_cast<short8 *>(smem + p * PIXB) = h;
            if constexpr (PARTS == 2) *reinterpret_cast<short8 *>(smem + PLANEB + p * PIXB) = l;
        }
    };
    bool from_stack = false;
    if constexpr (STEM) from_stack = STEMP || a.fs32 != nullptr;
    if (from_stack) {
        if constexpr (STEM) fill_from_stack(cur);
    } else if (!(a.dbg & 1)) {
        issue_fill(cur, st_lo);
    }
    if (t.warm) {
        // Few-tile launches (batch 1, the low-resolution pyramid): the contraction loop requests a chunk's weight fragments only one chunk
        // ahead, which covers an L2 hit but not the ~1-2 us of a miss -- and with a handful of workgroups per layer nobody else has pulled
        // the filter into this XCD's L2 (measured r03, one 10x256x256 stack: SPP conv8 = 32 chunks per workgroup in 42 us).  So every
        // thread first touches its share of the 128-byte lines of the workgroup's whole walk (all its passes, stages, chunks, its output
        // tiles); the touches travel with the footprint DMA and are waited for with it.
        uint32_t sink = 0;
        const int wstride = NTT * PARTS * 64;
        for (int pass = pass_lo; pass < pass_hi; ++pass) {
            const int KCp = t.KC[pass];
            const int nlines = (st_hi - st_lo) * KCp * NT * PARTS * 8;
            const unsigned char *wb = reinterpret_cast<const unsigned char *>(t.wpk[pass]);
            if (tid * 128 < KCp * 16) {   // ... and of the pass's tap-offset table (4 ints per chunk)
                const unsigned char *p = reinterpret_cast<const unsigned char *>(t.tab[pass]) + tid * 128;
                asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p));
            }
            for (int i = tid; i < nlines; i += NWAVES * 64) {
                const int l8 = i & 7, f = (i >> 3) % (NT * PARTS), c = (i >> 3) / (NT * PARTS);
                const int kc = c % KCp, st = st_lo + c / KCp;
                const unsigned char *p = wb + ((((int64_t)st * KCp + kc) * wstride + ntb * PARTS * 64 + f * 64) * 16 + l8 * 128);
                // ("+v": the destination stays live from one touch to the next -- as a plain output hipcc would hand the register to the next
                // iteration's address arithmetic while the load is still in flight)
                asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
    }
    stamp(1);

    {
        f32x4 acc[NT][MTW];
#pragma unroll 1   // one accumulator set live at a time
        for (int pass = pass_lo; pass < pass_hi; ++pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < MTW; ++j) acc[nt][j] = (BIAS_IN_ACC && !splitk && team == 0) ? bias4[nt] : f32x4{0.f, 0.f, 0.f, 0.f};

            const int KC = t.KC[pass];
            const int *tab = t.tab[pass] + g;

            for (int sti = st_lo; sti < st_hi; ++sti) {
                // transposed conv over several channel-group stages: odd passes walk the stages backwards, so every pass
                // after the first starts on the image the previous pass ended on (5 fills per tile instead of 8 for two
                // stages; measured on dres2.conv6: a fill is 10.7k cycles of issue + 6.2k of wait per workgroup)
                const bool backwards = (GEO == G3T) && ((pass - pass_lo) & 1);
                const int st = backwards ? st_hi - 1 - (sti - st_lo) : sti;
                const bool resident = (GEO == G3T) && pass > pass_lo && sti == st_lo && st_hi - st_lo > 1;   // staged by the previous pass
                const bool prefilled = (pass == pass_lo && sti == st_lo);   // queued by the prologue
                // first chunk's weight fragments and tap offset: requested BEFORE waiting for the footprint DMA so
                // that their L2 latency overlaps it
                const int wstride = NTT * PARTS * 64;   // fragments (16 B per lane) per 32-deep chunk
                const short8 *wp = reinterpret_cast<const short8 *>(t.wpk[pass]) + (int64_t)st * KC * wstride + ntb * PARTS * 64 + lane;
                short8 wfirst[NT][PARTS];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int pt = 0; pt < PARTS; ++pt) wfirst[nt][pt] = wp[(nt * PARTS + pt) * 64];
                const int tfirst = tab[0];
                // team configurations: weight fragments and tap offsets WD chunks ahead (a ring of WD register sets, requested in front of the wait for the image).
                // These launches run one or two workgroups per CU on a mostly idle chip: one chunk ahead, every chunk of 6-12 MFMAs waited ~600 cycles
                // for its fragments' L2 round trip -- 6.1k cycles for a 14-chunk stage (phase timeline of dres8_1.0 at batch 1).
                constexpr int WD = KT > 1 ? 4 : 1;
                short8 wr[WD][NT][PARTS];
                int tr[WD];
                if constexpr (KT > 1) {
#pragma unroll
                    for (int q = 0; q < WD; ++q) {
                        const int kq = q < KC ? q : KC - 1;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int pt = 0; pt < PARTS; ++pt) wr[q][nt][pt] = wp[(int64_t)kq * wstride + (nt * PARTS + pt) * 64];
                        tr[q] = tab[kq * 4];
                    }
                }
                if ((pass == pass_lo || st_hi - st_lo > 1) && !resident && !(a.dbg & 1)) {
                    if (!prefilled) {
                        __syncthreads();  // everyone is done reading the previous image
                        issue_fill(cur, st);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (prefilled) stamp(2);
                }

                // ---- contraction over (tap, channel-in-group) ------------------------------------------
                if constexpr (PIPE == 0) {
                    // Lean loop for bandwidth-bound layers: few registers -> 4-5 workgroups per CU hide the
                    // fill / residual / store latencies by occupancy instead of by software pipelining.
                    short8 wf[NT][PARTS], wn[NT][PARTS];
                    int toff = tfirst, tn = 0;
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
    #pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt) wf[nt][pt] = wfirst[nt][pt];
                    for (int kc = 0; kc < ((a.dbg & 2) ? 1 : KC); ++kc) {
                        if (kc + 1 < KC) {   // next chunk's fragments travel under this chunk's MFMAs
    #pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
    #pragma unroll
                                for (int pt = 0; pt < PARTS; ++pt) wn[nt][pt] = wp[(int64_t)(kc + 1) * wstride + (nt * PARTS + pt) * 64];
                            tn = tab[(kc + 1) * 4];
                        }
    #pragma unroll
                        for (int j = 0; j < MTW; ++j) {
                            const unsigned char *lp = smem + pofs[j] + toff;
                            const short8 xh = *reinterpret_cast<const short8 *>(lp);
                            short8 xl = xh;
                            if constexpr (PARTS == 2) xl = *reinterpret_cast<const short8 *>(lp + PLANEB);
    #pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if constexpr (PARTS == 2) {
                                    acc[nt][j] = mma<F16>(wf[nt][1], xh, acc[nt][j]);
                                    acc[nt][j] = mma<F16>(wf[nt][0], xl, acc[nt][j]);
                                }
                                acc[nt][j] = mma<F16>(wf[nt][0], xh, acc[nt][j]);
                            }
                        }
    #pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
    #pragma unroll
                            for (int pt = 0; pt < PARTS; ++pt) wf[nt][pt] = wn[nt][pt];
                        toff = tn;
                    }
                } else if constexpr (KT > 1) {
                    // team loop: the operands of chunk k + 1 are read while chunk k is contracted, the fragments of chunk k + WD requested behind it
                    short8 xc[MTW][PARTS], xn[MTW][PARTS];
#pragma unroll
                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt) xc[j][pt] = *reinterpret_cast<const short8 *>(smem + pofs[j] + tr[0] + pt * PLANEB);
                    for (int kc0 = 0; kc0 < ((a.dbg & 2) ? 1 : KC); kc0 += WD) {
#pragma unroll
                        for (int q = 0; q < WD; ++q) {
                            const int kc = kc0 + q;
                            if (kc < KC) {
                                if (kc + 1 < KC) {
#pragma unroll
                                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                                        for (int pt = 0; pt < PARTS; ++pt)
                                            xn[j][pt] = *reinterpret_cast<const short8 *>(smem + pofs[j] + tr[(q + 1) % WD] + pt * PLANEB);
                                }
                                if constexpr (PARTS == 2) {
#pragma unroll
                                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                                        for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wr[q][nt][1], xc[j][0], acc[nt][j]);
#pragma unroll
                                    for (int j = 0; j < MTW; ++j)
#pragma unroll
                                        for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wr[q][nt][0], xc[j][1], acc[nt][j]);
                                }
#pragma unroll
                                for (int j = 0; j < MTW; ++j)
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wr[q][nt][0], xc[j][0], acc[nt][j]);
                                if (kc + WD < KC) {
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                        for (int pt = 0; pt < PARTS; ++pt) wr[q][nt][pt] = wp[(int64_t)(kc + WD) * wstride + (nt * PARTS + pt) * 64];
                                    tr[q] = tab[(kc + WD) * 4];
                                }
#pragma unroll
                                for (int j = 0; j < MTW; ++j)
#pragma unroll
                                    for (int pt = 0; pt < PARTS; ++pt) xc[j][pt] = xn[j][pt];
                            }
                        }
                    }
                } else {
                    // Software pipeline: the wave's MTW operand tiles are split into two groups; while the matrix
                    // cores work on one group's operands the ds_reads of the other group (same chunk or the next
                    // one) are in flight.  Within a group the three split-bf16 products are issued product-major so
                    // consecutive MFMAs never share an accumulator.
                    constexpr int GA = MTW / 2, GB = MTW - GA;
                    short8 wcur[NT][PARTS], wnxt[NT][PARTS];
                    short8 xa[GA][PARTS], xb[GB][PARTS];
                    // tap offsets run two chunks ahead: the one for chunk kc+1 is consumed in the MIDDLE of iteration kc
                    // (group A's next operands), so a load issued at the top of the same iteration would be waited
                    // for ~100 cycles later together with the weight fragments behind it
                    int tcur = tfirst, tnxt = (KC > 1) ? tab[4] : 0, tn2 = 0;
        #pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
        #pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt) wcur[nt][pt] = wfirst[nt][pt];
        #pragma unroll
                    for (int j = 0; j < GA; ++j)
        #pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt) xa[j][pt] = *reinterpret_cast<const short8 *>(smem + pofs[j] + tcur + pt * PLANEB);
                    for (int kc = 0; kc < ((a.dbg & 2) ? 1 : KC); ++kc) {
                        const bool more = kc + 1 < KC;
                        if (more) {
                            const short8 *wn = wp + (int64_t)(kc + 1) * wstride;
        #pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
        #pragma unroll
                                for (int pt = 0; pt < PARTS; ++pt) wnxt[nt][pt] = wn[(nt * PARTS + pt) * 64];
                            if (kc + 2 < KC) tn2 = tab[(kc + 2) * 4];
                        }
                        // operands of group B for this chunk: in flight during group A's MFMAs
        #pragma unroll
                        for (int j = 0; j < GB; ++j)
        #pragma unroll
                            for (int pt = 0; pt < PARTS; ++pt) xb[j][pt] = *reinterpret_cast<const short8 *>(smem + pofs[GA + j] + tcur + pt * PLANEB);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (PARTS == 2) {
        #pragma unroll
                            for (int j = 0; j < GA; ++j)
        #pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wcur[nt][1], xa[j][0], acc[nt][j]);
        #pragma unroll
                            for (int j = 0; j < GA; ++j)
        #pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wcur[nt][0], xa[j][1], acc[nt][j]);
                        }
        #pragma unroll
                        for (int j = 0; j < GA; ++j)
        #pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wcur[nt][0], xa[j][0], acc[nt][j]);
                        __builtin_amdgcn_sched_barrier(0);
                        // operands of group A for the next chunk: in flight during group B's MFMAs
                        if (more) {
        #pragma unroll
                            for (int j = 0; j < GA; ++j)
        #pragma unroll
                                for (int pt = 0; pt < PARTS; ++pt) xa[j][pt] = *reinterpret_cast<const short8 *>(smem + pofs[j] + tnxt + pt * PLANEB);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (PARTS == 2) {
        #pragma unroll
                            for (int j = 0; j < GB; ++j)
        #pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt][GA + j] = mma<F16>(wcur[nt][1], xb[j][0], acc[nt][GA + j]);
        #pragma unroll
                            for (int j = 0; j < GB; ++j)
        #pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt][GA + j] = mma<F16>(wcur[nt][0], xb[j][1], acc[nt][GA + j]);
                        }
        #pragma unroll
                        for (int j = 0; j < GB; ++j)
        #pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][GA + j] = mma<F16>(wcur[nt][0], xb[j][0], acc[nt][GA + j]);
                        __builtin_amdgcn_sched_barrier(0);
        #pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
        #pragma unroll
                            for (int pt = 0; pt < PARTS; ++pt) wcur[nt][pt] = wnxt[nt][pt];
                        tcur = tnxt;
                        tnxt = tn2;
                    }
                }
            }

            if constexpr (KT > 1) {
                // ---- the teams' partial sums meet in LDS: [team - 1][wave][output tile][operand tile][64 lanes] x 16 bytes over the images, which
                // nobody reads any more behind the first barrier; team 0 adds them in team order (fixed order: run-to-run identical) ----
                static_assert((KT - 1) * NWAVES * NT * MTW * 1024 <= KT * LDSB, "reduction area");
                __syncthreads();
                if (team > 0) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < MTW; ++j)
                            *reinterpret_cast<f32x4 *>(smem_all + ((((team - 1) * NWAVES + wave) * NT + nt) * MTW + j) * 1024 + lane * 16) = acc[nt][j];
                }
                __syncthreads();
                if (team > 0) return;
#pragma unroll
                for (int z = 1; z < KT; ++z)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < MTW; ++j)
                            acc[nt][j] += *reinterpret_cast<const f32x4 *>(smem_all + ((((z - 1) * NWAVES + wave) * NT + nt) * MTW + j) * 1024 + lane * 16);
            }
            // ---- epilogue of this pass (shared with conv_igemm, see dffw_device.h) ---------------------------
            if (pass == G::NPASS - 1) stamp(3);
            const int ooy = t.ooy[pass], oox = t.oox[pass];
            // output location of operand tile j = tile base (wave-uniform) + the lane's precomputed offset; the
            // bounds test is only evaluated for tiles that stick out of the volume
            const int64_t obase = (((int64_t)cur.b * a.No + cur.gz0) * a.Ho + (cur.gy0 * G::OS + ooy)) * a.Wo + (cur.gx0 * G::OS + oox);
            const int64_t ubase = obase * (PARTS * a.Cout);
            const bool interior = cur.gz0 + TZ <= a.Ng && cur.gy0 + TY <= a.Hg && cur.gx0 + TX <= a.Wg;
            auto where = [&](int j, int64_t &opix) -> bool {
                const int c = tcrd[j];
                opix = obase + (((c >> 16) * a.Ho + ((c >> 8) & 255) * G::OS) * a.Wo + (c & 255) * G::OS);   // only read by the score paths
                bool ok = true;
                if (!interior) ok = cur.gz0 + (c >> 16) < a.Ng && cur.gy0 + ((c >> 8) & 255) < a.Hg && cur.gx0 + (c & 255) < a.Wg;
                if ((a.dbg & 4) && acc[0][j][0] != 12345.f) ok = false;
                return ok;
            };
            if constexpr (LEAN) {
                static_assert(!LEAN || (PARTS == 2 && !SPLITK), "straight-line epilogue: split-bf16 storage, no split-K / raw / row-sums variant");
                const bool relu1 = a.relu == 1, has_res = a.res0 != nullptr, has_cls = a.cls_w != nullptr;
                uint16_t *ob = a.out ? a.out + ubase : nullptr, *obp = a.out_pre ? a.out_pre + ubase : nullptr;
                const uint16_t *rb = has_res ? a.res0 + ubase : nullptr;
                auto valid = [&](int j) -> bool {
                    bool ok = true;
                    if (!interior) {
                        const int c = tcrd[j];
                        ok = cur.gz0 + (c >> 16) < a.Ng && cur.gy0 + ((c >> 8) & 255) < a.Hg && cur.gx0 + (c & 255) < a.Wg;
                    }
                    if ((a.dbg & 4) && acc[0][j][0] != 12345.f) ok = false;
                    return ok;
                };
                auto pixel = [&](int j) -> int64_t {   // only the classifier's score store needs it
                    const int c = tcrd[j];
                    return obase + (((c >> 16) * a.Ho + ((c >> 8) & 255) * G::OS) * a.Wo + (c & 255) * G::OS);
                };
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (STEMP) {
                    // pixel-pair stem: lane rows 0-1 hold pixel x's 8 channels, rows 2-3 pixel x+2's -- each lane is "row g & 1" of its own record
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        float cls = 0.f;
                        f32x4 v = acc[0][j];
                        if constexpr (!BIAS_IN_ACC) v += *reinterpret_cast<const f32x4 *>(a.bias + (g & 1) * 4);
                        epilogue_lean_t<PREC>(ob, obp, voff[j], v[0], v[1], v[2], v[3], false, uint4{}, relu1, false, zero4, cls, valid(j));
                    }
                } else if (NT == 1 && a.Cout == 8) {
                    // 8 output channels occupy only lane rows 0-1 of a result tile: operand tiles j and j+1 share one epilogue
#pragma unroll
                    for (int j = 0; j < MTW; j += 2) {
                        if (j + 1 < MTW) {
                            float q[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][j][i]), __float_as_uint(acc[0][j + 1][i]), false, false);
                                q[i] = __uint_as_float(sw[0]);   // lanes 0-31: tile j, lanes 32-63: lanes 0-31 of tile j+1
                            }
                            const bool p0 = valid(j), p1 = valid(j + 1);
                            const bool up = lane >= 32;
                            const bool pv = up ? p1 : p0;
                            const int vo = up ? voff[j + 1] - 8 : voff[j];
                            if constexpr (!BIAS_IN_ACC) {
                                const f32x4 bb = *reinterpret_cast<const f32x4 *>(a.bias + (g & 1) * 4);
#pragma unroll
                                for (int i = 0; i < 4; ++i) q[i] += bb[i];
                            }
                            uint4 rq = uint4{};
                            if (has_res && pv) rq = *reinterpret_cast<const uint4 *>(rb + vo);
                            float cls = 0.f;
                            f32x4 cw = zero4;
                            if (has_cls) cw = *reinterpret_cast<const f32x4 *>(a.cls_w + (g & 1) * 4);
                            epilogue_lean_t<PREC>(ob, obp, vo, q[0], q[1], q[2], q[3], has_res, rq, relu1, has_cls, cw, cls, pv);
                            if (has_cls) epilogue_cls(a, cls, g, up ? pixel(j + 1) : pixel(j), pv, 2);
                        } else {
                            const bool pv = valid(j) && g < 2;   // the odd last tile: its rows 2-3 carry no channels
                            f32x4 v = acc[0][j];
                            if constexpr (!BIAS_IN_ACC) v += bias4[0];
                            uint4 rq = uint4{};
                            if (has_res && pv) rq = *reinterpret_cast<const uint4 *>(rb + voff[j]);
                            float cls = 0.f;
                            f32x4 cw = zero4;
                            if (has_cls && g < 2) cw = *reinterpret_cast<const f32x4 *>(a.cls_w + g * 4);
                            epilogue_lean_t<PREC>(ob, obp, voff[j], v[0], v[1], v[2], v[3], has_res, rq, relu1, has_cls, cw, cls, pv);
                            if (has_cls) epilogue_cls(a, cls, g, pixel(j), valid(j));
                        }
                    }
                } else {
                    // residual pieces one operand tile ahead of their use where the registers are there (<= 2 output tiles per
                    // workgroup), else requested right before it (what the generic routine does)
                    constexpr bool AHEAD = NT <= 2 && NWAVES == 4;
                    uint4 rn[AHEAD ? NT : 1];
                    bool pvn = valid(0);
                    if constexpr (AHEAD) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            rn[nt] = uint4{};
                            if (has_res && pvn) rn[nt] = *reinterpret_cast<const uint4 *>(rb + (voff[0] + (ntb + nt) * 16));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < MTW; ++j) {
                        const bool pv = pvn;
                        uint4 rq[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if constexpr (AHEAD) rq[nt] = rn[nt];
                            else {
                                rq[nt] = uint4{};
                                if (has_res && pv) rq[nt] = *reinterpret_cast<const uint4 *>(rb + (voff[j] + (ntb + nt) * 16));
                            }
                        }
                        if (j + 1 < MTW) {
                            pvn = valid(j + 1);
                            if constexpr (AHEAD) {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) {
                                    rn[nt] = uint4{};
                                    if (has_res && pvn) rn[nt] = *reinterpret_cast<const uint4 *>(rb + (voff[j + 1] + (ntb + nt) * 16));
                                }
                            }
                        }
                        float cls = 0.f;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            f32x4 v = acc[nt][j];
                            if constexpr (!BIAS_IN_ACC) v += bias4[nt];
                            f32x4 cw = zero4;
                            if (has_cls) cw = *reinterpret_cast<const f32x4 *>(a.cls_w + (ntb + nt) * 16 + g * 4);
                            epilogue_lean_t<PREC>(ob, obp, voff[j] + (ntb + nt) * 16, v[0], v[1], v[2], v[3], has_res, rq[nt], relu1, has_cls, cw, cls, pv);
                        }
                        if (has_cls) epilogue_cls(a, cls, g, pixel(j), pv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if (SPLITK && splitk) {
                // raw partial sums: 4 consecutive channels of the lane's pixel as one 16-byte store
                float *pz = t.partial + (int64_t)blockIdx.z * t.partial_stride;
                const int cpad = NTT * 16;
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    int64_t opix;
                    if (where(j, opix)) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4 *>(pz + opix * cpad + (ntb + nt) * 16 + g * 4) = acc[nt][j];
                    }
                }
            } else if (STEMP) {
                // pixel-pair stem: lane rows 0-1 hold pixel x's 8 channels, rows 2-3 pixel x+2's -- each lane is "row g & 1" of
                // its own pixel's record, the packed form of the 8-channel epilogue with no register shuffling
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    int64_t opix;
                    const bool pv = where(j, opix);
                    float cls = 0.f;
                    epilogue_quad<PREC, false, true, !BIAS_IN_ACC>(a, acc[0][j], 0, g & 1, opix, pv, cls, uint4{}, uint4{}, ubase, voff[j]);
                    epilogue_cls(a, cls, g, opix, pv, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (NT == 1 && a.Cout == 8 && !a.outf) {
                // 8 output channels occupy only lane rows 0-1 of a result tile: pack operand tiles j and j+1 into
                // one register set (rows 2-3 <- rows 0-1 of tile j+1, v_permlane32_swap) and run ONE epilogue for both
#pragma unroll
                for (int j = 0; j < MTW; j += 2) {
                    if (j + 1 < MTW) {
                        f32x4 q;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][j][i]), __float_as_uint(acc[0][j + 1][i]), false, false);
                            q[i] = __uint_as_float(sw[0]);   // lanes 0-31: tile j, lanes 32-63: lanes 0-31 of tile j+1
                        }
                        int64_t o0, o1;
                        const bool p0 = where(j, o0), p1 = where(j + 1, o1);
                        const bool up = lane >= 32;
                        const int64_t opix = up ? o1 : o0;
                        const bool pv = up ? p1 : p0;
                        // lanes 32-63 carry rows 0-1 of tile j+1: their piece of the record is that of row g-2
                        const int vo = up ? voff[j + 1] - ((PARTS == 2) ? 8 : 8) : voff[j];
                        float cls = 0.f;
                        epilogue_quad<PREC, false, true, !BIAS_IN_ACC>(a, q, 0, g & 1, opix, pv, cls, uint4{}, uint4{}, ubase, vo);
                        epilogue_cls(a, cls, g, opix, pv, 2);
                        __builtin_amdgcn_sched_barrier(0);   // keep the unrolled iterations' live ranges apart (occupancy)
                    } else {
                        int64_t opix;
                        const bool pv = where(j, opix);
                        float cls = 0.f;
                        epilogue_quad<PREC, false, true, !BIAS_IN_ACC>(a, acc[0][j], 0, g, opix, pv, cls, uint4{}, uint4{}, ubase, voff[j]);
                        epilogue_cls(a, cls, g, opix, pv);
                    }
                }
            } else if (GEO == G3T && PIPE == 1 && NT <= 2 && PARTS == 2 && a.res0 && !a.res1 && !a.outf && !a.res_bcast) {
                // transposed convs with a skip connection (dres*.conv5/6): all residual pieces of the pass are requested before the
                // first is consumed.  The loop below requests one, waits, stores, requests the next -- measured ~35k cycles per
                // pass on dres2.conv6, mostly load latency; these kernels are LDS-limited to two workgroups per CU, so the 40
                // registers are there.
                uint4 rq[NT][MTW];
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    int64_t opix;
                    const bool pv = where(j, opix);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool wv = pv && ((ntb + nt) * 2 + (g >> 1)) * 8 < a.Cout;
                        rq[nt][j] = make_uint4(0, 0, 0, 0);
                        if (wv) rq[nt][j] = *reinterpret_cast<const uint4 *>(a.res0 + ubase + (int64_t)(voff[j] + (ntb + nt) * 16));
                    }
                }
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    int64_t opix;
                    const bool pv = where(j, opix);
                    float cls = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) epilogue_quad<PREC, true, true, !BIAS_IN_ACC>(a, acc[nt][j], ntb + nt, g, opix, pv, cls, rq[nt][j], uint4{}, ubase, voff[j]);
                    epilogue_cls(a, cls, g, opix, pv);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (SUMS) {
                // an operand tile is one row segment of 16 pixels (TX == 16): lane (g, r) holds channels 4g..4g+3 of pixel r.  Row sums
                // over the 16 lanes by DPP (quad xor 1, xor 2, half-row mirror, row mirror); lanes r = 0 / r = 15 are the segment's
                // first / last pixel.  rows[((plane * Ho + y) * tiles_x + tile column) * 3 + {sum, first, last}][Cout] fp32.
                static_assert(!SUMS || TX == 16, "row-sums variant: one operand tile = one row segment");
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    int64_t opix;
                    const bool pv = where(j, opix);
                    const int c = tcrd[j];
                    const int64_t row = ((int64_t)cur.b * a.No + cur.gz0 + (c >> 16)) * a.Ho + cur.gy0 + ((c >> 8) & 255);
                    float *rp = a.outf + ((row * t.tiles_x + cur.gx0 / TX) * 3) * a.Cout + g * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x4 v, rs;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = pv ? relu_bits(acc[nt][j][i]) : 0.f;
                            float q = v[i];
                            q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                            q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                            q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x141, 0xF, 0xF, true));   // row_half_mirror
                            q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x140, 0xF, 0xF, true));   // row_mirror
                            rs[i] = q;
                        }
                        // (a row of an edge tile past the volume: its lane r = 0 is past it too, nothing is written)
                        if (pv && r == 0) {
                            *reinterpret_cast<f32x4 *>(rp + (ntb + nt) * 16) = rs;
                            *reinterpret_cast<f32x4 *>(rp + a.Cout + (ntb + nt) * 16) = v;
                        }
                        if (pv && r == 15) *reinterpret_cast<f32x4 *>(rp + 2 * a.Cout + (ntb + nt) * 16) = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    int64_t opix;
                    const bool pv = where(j, opix);
                    float cls = 0.f;
                    // slice-broadcast residual (alignment heads): same (y,x) record of the sample's single residual slice
                    // (only the per-slice 1x3x3 geometry is ever launched with it, so only that one carries the arithmetic)
                    int64_t rbase = ubase;
                    int rvoff = voff[j];
                    if constexpr (GEO == G2S1) {
                        rbase = ubase - ((int64_t)cur.b * (a.No - 1) + cur.gz0) * a.Ho * a.Wo * (PARTS * a.Cout);
                        rvoff = voff[j] - (tcrd[j] >> 16) * a.Ho * a.Wo * (PARTS * a.Cout);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) epilogue_quad<PREC, false, true, !BIAS_IN_ACC>(a, acc[nt][j], ntb + nt, g, opix, pv, cls, uint4{}, uint4{}, ubase, voff[j], rbase, rvoff);
                    epilogue_cls(a, cls, g, opix, pv);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#ifdef DFFW_TRACE_BUILD
    if (a.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores of this wave issued AND acknowledged
        stamp(4);
        if (tid == 0 && team == 0) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            a.trace[(int64_t)tile * 8 + 5] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#endif
}

// ---- configuration table -----------------------------------------------------------------------
//        id  geo   NT  TZ TY  TX  CG
#define DFFW_TILE_CONFIGS(X)         \
    X(0, G3S1, 1, 5, 4, 16, 16, 1)   \
    X(1, G3S1, 1, 5, 4, 16, 8, 1)    \
    X(2, G3S1, 2, 5, 4, 16, 16, 1)   \
    X(3, G3S1, 4, 5, 4, 16, 16, 1)   \
    X(4, G3S1, 8, 4, 4, 8, 16, 1)    \
    X(5, G3S2, 1, 5, 4, 16, 8, 1)    \
    X(6, G3S2, 2, 5, 4, 16, 8, 1)    \
    X(7, G3S2, 4, 5, 4, 16, 8, 1)    \
    X(8, G3S2, 8, 4, 4, 8, 8, 1)     \
    X(9, G3T, 1, 5, 4, 16, 16, 0)    \
    X(10, G3T, 2, 5, 4, 16, 16, 1)   \
    X(11, G3T, 4, 5, 4, 16, 16, 1)   \
    X(12, G2S1, 1, 5, 4, 16, 8, 0)   \
    X(13, G2S1, 1, 5, 4, 16, 16, 0)  \
    X(14, G2S1, 2, 5, 8, 16, 16, 1)  \
    X(15, G2D, 1, 1, 16, 32, 8, 0)   \
    X(16, G3S1, 2, 4, 4, 8, 16, 1)   \
    X(17, G3S2, 2, 4, 4, 8, 8, 1)    \
    X(18, G2S1, 2, 5, 4, 16, 8, 1)   \
    X(19, G2S1, 4, 5, 4, 16, 16, 1)  \
    X(20, G2S1, 4, 5, 4, 16, 8, 1)   \
    X(21, G2S2, 1, 5, 4, 16, 8, 0)   \
    X(22, G2S2, 2, 5, 4, 16, 8, 0)   \
    X(23, G3T, 1, 5, 4, 16, 32, 0)   \
    X(24, G3T, 2, 5, 4, 16, 32, 1)   \
    X(25, G3T, 4, 5, 4, 16, 32, 1)   \
    X(34, G2S1, 2, 5, 4, 16, 32, 1)  \
    X(37, G3S2, 4, 4, 4, 8, 16, 1)   \
    X(38, G3S2, 2, 4, 4, 8, 16, 1)   \
    X(39, G3S2, 1, 4, 4, 8, 16, 1)   \
    X(32, G2P, 1, 1, 16, 32, 8, 0)   \
    X(40, G3S1, 1, 5, 8, 8, 16, 1)   \
    X(41, G3S1, 2, 5, 8, 8, 16, 1)   \
    X(42, G3S1, 4, 5, 8, 8, 16, 1)   \
    X(43, G3T, 1, 5, 8, 8, 32, 0)    \
    X(44, G3T, 2, 5, 8, 8, 32, 1)    \
    X(45, G3T, 4, 5, 8, 8, 32, 1)
// (40-45, round 4: 5 x 8 x 8 blocks for grids at most 8 wide -- the 1/32-resolution pyramid layers at 256 x 256: on the 5 x 4 x 16 block half of
// every operand tile lies outside an 8 x 8 grid, and the 4 x 4 x 8 block re-streams the filter for 128 grid points at a time; tile_cfg_find_shape)
// 8-wave "wide" variants: 640-point tiles, same work per wave.  Measured +8..17 % on the bandwidth-bound
// single-stage layers with <= 16 output channels (one more resident wave per SIMD for the same LDS, 17 % less
// halo per output), -10 % on the 32-channel / multi-stage ones, so the engine asks for them only for the former.
#define DFFW_TILE_CONFIGS_W8(X)      \
    X(26, G3S1, 1, 5, 8, 16, 16, 1)  \
    X(27, G3T, 1, 5, 8, 16, 16, 0)   \
    X(28, G2S1, 1, 5, 8, 16, 8, 0)   \
    X(29, G2S1, 1, 5, 8, 16, 16, 0)  \
    X(30, G2S1, 2, 5, 8, 16, 16, 0)  \
    X(31, G2D, 1, 1, 32, 32, 8, 0)   \
    X(33, G2P, 1, 1, 32, 32, 8, 0)

// the 4-wave configurations few-tile layers end up on after the channel split (3x3x3 at stride 1 and 2, <= 32 output
// channels per workgroup): these also exist as split-K kernels
#define DFFW_TILE_CONFIGS_SPLITK(X)  \
    X(0, G3S1, 1, 5, 4, 16, 16, 1)   \
    X(1, G3S1, 1, 5, 4, 16, 8, 1)    \
    X(2, G3S1, 2, 5, 4, 16, 16, 1)   \
    X(16, G3S1, 2, 4, 4, 8, 16, 1)   \
    X(5, G3S2, 1, 5, 4, 16, 8, 1)    \
    X(6, G3S2, 2, 5, 4, 16, 8, 1)    \
    X(17, G3S2, 2, 4, 4, 8, 8, 1)    \
    X(38, G3S2, 2, 4, 4, 8, 16, 1)   \
    X(39, G3S2, 1, 4, 4, 8, 16, 1)

// "team" configurations (KT teams of NW waves per workgroup, see the kernel header): what a few-tile layer runs on INSTEAD of a split-K launch + splitk_finish.
// Same (geo, NT, TY, TX, CG) as a split-K configuration -- the packs and tap tables do not depend on TZ.  Small blocks (64 grid points, two waves per team):
// a team launch has 1 / KT of the split-K launch's workgroups, and these layers are latency chains on a mostly idle chip -- first measured with the split-K
// blocks (320 points, four waves per team): the teams of a workgroup share their SIMDs' matrix pipes, 14.1 us against 11.5 + the finish launch.
//        id  geo   NT  TZ TY  TX  CG  PIPE NW KT
#define DFFW_TILE_CONFIGS_TEAM(X)           \
    X(50, G3S1, 1, 1, 4, 16, 16, 1, 2, 2)   \
    X(51, G3S1, 1, 1, 4, 16, 16, 1, 2, 4)   \
    X(52, G3S2, 1, 1, 4, 16, 8, 1, 2, 2)    \
    X(53, G3S2, 1, 1, 4, 16, 8, 1, 2, 4)    \
    X(58, G3T, 1, 1, 4, 16, 32, 1, 2, 2)    \
    X(59, G3T, 1, 1, 4, 16, 32, 1, 2, 4)    \
    X(60, G3T, 2, 1, 4, 16, 32, 1, 2, 2)    \
    X(61, G3T, 2, 1, 4, 16, 32, 1, 2, 4)
// (built and dropped, batch 1 / 2 layer tables in profiles/r06_batch1_teams.txt: the 2 x 4 x 8 blocks of the 8 x 8-grid layers with 128-192 input channels, 4 and 8
// teams of two or three stages each -- 40-96 workgroups walk what split-K spreads over 192: combine2 27.9 against 24.1 us, conv4 21-41 against 19-22)

#if DFFW_TILE_PREC == 0 && !defined(DFFW_TILE_LEAN)   // configuration table and look-ups live in one of the per-precision objects
bool tile_cfg_has_sums(const TileCfg *c) { return c && c->geo == G2S1 && c->nw == 4 && (c->id == 34 || c->id == 19); }
bool tile_cfg_has_splitk(const TileCfg *c) {
    switch (c->id) {
#define X_HAS(ID, GEO, NT, TZ, TY, TX, CG, PIPE) case ID:
        DFFW_TILE_CONFIGS_SPLITK(X_HAS)
#undef X_HAS
        return c->nw == 4;
        default: return false;
    }
}

#define X_CFG(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                             \
    TileCfg{ID, GEO, NT, CG, TZ, TY, TX, TileT<GEO, TZ, TY, TX, CG>::FZ, TileT<GEO, TZ, TY, TX, CG>::FY, \
            TileT<GEO, TZ, TY, TX, CG>::FX, TileT<GEO, TZ, TY, TX, CG>::FXL, PIPE, 4},
#define X_CFG8(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                            \
    TileCfg{ID, GEO, NT, CG, TZ, TY, TX, TileT<GEO, TZ, TY, TX, CG>::FZ, TileT<GEO, TZ, TY, TX, CG>::FY, \
            TileT<GEO, TZ, TY, TX, CG>::FX, TileT<GEO, TZ, TY, TX, CG>::FXL, PIPE, 8},
#define X_CFGT(ID, GEO, NT, TZ, TY, TX, CG, PIPE, NW, KT)                                                    \
    TileCfg{ID, GEO, NT, CG, TZ, TY, TX, TileT<GEO, TZ, TY, TX, CG>::FZ, TileT<GEO, TZ, TY, TX, CG>::FY, \
            TileT<GEO, TZ, TY, TX, CG>::FX, TileT<GEO, TZ, TY, TX, CG>::FXL, PIPE, NW, KT},
static const TileCfg g_cfgs[] = {DFFW_TILE_CONFIGS(X_CFG) DFFW_TILE_CONFIGS_W8(X_CFG8) DFFW_TILE_CONFIGS_TEAM(X_CFGT)};
#undef X_CFG
#undef X_CFG8
#undef X_CFGT

int tile_cfg_count() { return (int)(sizeof(g_cfgs) / sizeof(g_cfgs[0])); }
const TileCfg *tile_cfg_at(int i) { return (i >= 0 && i < tile_cfg_count()) ? &g_cfgs[i] : nullptr; }
const TileCfg *tile_cfg_find(int geo, int nt, int cg, bool wide) {
    if (wide)
        for (const TileCfg &c : g_cfgs)
            if (c.nw == 8 && c.kt <= 1 && c.geo == geo && c.nt == nt && c.cg == cg) return &c;
    for (const TileCfg &c : g_cfgs)
        if (c.nw == 4 && c.kt <= 1 && c.geo == geo && c.nt == nt && c.cg == cg) return &c;
    return nullptr;
}

const TileCfg *tile_cfg_find_shape(int geo, int nt, int cg, int tz, int ty, int tx) {
    for (const TileCfg &c : g_cfgs)
        if (c.nw == 4 && c.kt <= 1 && c.geo == geo && c.nt == nt && c.cg == cg && c.tz == tz && c.ty == ty && c.tx == tx) return &c;
    return nullptr;
}

const TileCfg *tile_cfg_find_like(const TileCfg *base, int nt) {
    for (const TileCfg &c : g_cfgs)
        if (c.kt <= 1 && c.geo == base->geo && c.cg == base->cg && c.nt == nt && c.tz == base->tz && c.ty == base->ty && c.tx == base->tx && c.nw == base->nw) return &c;
    return nullptr;
}

// the team configuration that replaces a split-K launch of `base` over `nstage` channel-group stages: same geometry, channel group, output tiles per workgroup
// and TY x TX (the pack and its tap offsets are those of `base`), ONE stage per team (a team that walks several stages re-fills its image between them, and every
// such layer measured slower than its split-K launch pair); nullptr: none
const TileCfg *tile_cfg_find_team(const TileCfg *base, int nstage, int nt) {
    for (const TileCfg &c : g_cfgs)
        if (c.kt == nstage && c.kt > 1 && c.geo == base->geo && c.cg == base->cg && c.nt == nt && c.ty == base->ty && c.tx == base->tx) return &c;
    return nullptr;
}

// the launch takes the LEAN instantiation (straight-line epilogue): split-bf16 storage, no split-K / raw-stack / row-sums variant,
// something to write, ReLU (if any) after the residual, at most one residual in the output's own geometry, and EVERY 16-channel
// result tile of the layer a real one (the straight-line epilogue has no per-tile channel guard: Cout = 48 / 80 / 96 / 112 are packed as 4 / 8
// tiles whose padding tiles must not store; a channel-split launch, c->nt < nt_total, walks real tiles only) or the packed 8-channel form
bool tile_lean(int prec, const TileCfg *c, const ConvArgs &a, const TileArgs &t) {
    return prec == P_BF16X3 && t.ksplit <= 1 && !(a.dbg & (DFFW_ARGS_RAW | DFFW_ARGS_SUMS)) && (a.out || a.out_pre || a.cls_w) && !a.outf && !a.res1 &&
           !a.res_bcast && a.relu != 2 && (a.Cout == t.nt_total * 16 || (a.Cout == 8 && c->nt == 1)) && !(a.dbg & DFFW_ARGS_NO_LEAN_TILE) &&
           c->id != 23 && c->id != 27;   // (these two transposed-conv configurations need 7 / 10 registers more with it: a wave per SIMD lost)
}

void conv_tile_kernel_name(int prec, const TileCfg *c, bool splitk, bool lean, char *buf, int n) {
    // exactly as rocprofv3 --kernel-trace prints the instantiation (all twelve template arguments)
    snprintf(buf, n, "dffw::conv_tile<%d, %d, %d, %d, %d, %d, %d, %d, %d, %s, %s, %d>", prec, c->geo, c->nt, c->tz, c->ty, c->tx, c->cg, c->pipe, c->nw,
             splitk ? "true" : "false", lean ? "true" : "false", c->kt > 1 ? c->kt : 1);
}

#endif

#ifdef DFFW_TILE_LEAN
// ---- the LEAN instantiations of the split-bf16 arithmetic: their own object (dffw_conv_tile_p0l.o), built beside the others ----
hipError_t launch_conv_tile_lean0(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    switch (cfg->id) {
#define X_LAUNCH(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                                       \
    case ID:                                                                                                        \
        hipLaunchKernelGGL((conv_tile<P_BF16X3, GEO, NT, TZ, TY, TX, CG, PIPE, 4, false, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)(t.pass_split ? 4 : 1)), dim3(256), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS(X_LAUNCH)
#undef X_LAUNCH
#define X_LAUNCH8(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                                      \
    case ID:                                                                                                        \
        hipLaunchKernelGGL((conv_tile<P_BF16X3, GEO, NT, TZ, TY, TX, CG, PIPE, 8, false, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)(t.pass_split ? 4 : 1)), dim3(512), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS_W8(X_LAUNCH8)
#undef X_LAUNCH8
#define X_LAUNCHT(ID, GEO, NT, TZ, TY, TX, CG, PIPE, NW, KT)                                                              \
    case ID:                                                                                                        \
        hipLaunchKernelGGL((conv_tile<P_BF16X3, GEO, NT, TZ, TY, TX, CG, PIPE, NW, false, true, KT>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)(t.pass_split ? 4 : 1)), dim3(64 * NW * KT), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS_TEAM(X_LAUNCHT)
#undef X_LAUNCHT
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#else
template <int PREC>
static hipError_t launch_conv_tile_p(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    switch (t.ksplit > 1 ? 1000 + cfg->id : ((a.dbg & DFFW_ARGS_SUMS) ? 3000 + cfg->id : ((a.dbg & DFFW_ARGS_RAW) ? 2000 + cfg->id : cfg->id))) {
#define X_LAUNCH(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                                       \
    case ID:                                                                                                        \
        hipLaunchKernelGGL((conv_tile<PREC, GEO, NT, TZ, TY, TX, CG, PIPE>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)(t.ksplit > 1 ? t.ksplit : (t.pass_split ? 4 : 1))), dim3(256), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS(X_LAUNCH)
#undef X_LAUNCH
#define X_LAUNCH8(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                                      \
    case ID:                                                                                                        \
        hipLaunchKernelGGL((conv_tile<PREC, GEO, NT, TZ, TY, TX, CG, PIPE, 8>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)(t.ksplit > 1 ? t.ksplit : (t.pass_split ? 4 : 1))), dim3(512), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS_W8(X_LAUNCH8)
#undef X_LAUNCH8
#define X_LAUNCHK(ID, GEO, NT, TZ, TY, TX, CG, PIPE)                                                                      \
    case 1000 + ID:                                                                                                 \
        hipLaunchKernelGGL((conv_tile<PREC, GEO, NT, TZ, TY, TX, CG, PIPE, 4, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)t.ksplit), dim3(256), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS_SPLITK(X_LAUNCHK)
#undef X_LAUNCHK
#define X_LAUNCHT(ID, GEO, NT, TZ, TY, TX, CG, PIPE, NW, KT)                                                              \
    case ID:                                                                                                        \
        hipLaunchKernelGGL((conv_tile<PREC, GEO, NT, TZ, TY, TX, CG, PIPE, NW, false, false, KT>), dim3((unsigned)t.grid, (unsigned)t.nsplit, (unsigned)(t.pass_split ? 4 : 1)), dim3(64 * NW * KT), 0, s, a, t); \
        break;
        DFFW_TILE_CONFIGS_TEAM(X_LAUNCHT)
#undef X_LAUNCHT
        // the row-sums variant of the per-slice 1x3x3 configurations the alignment heads' third conv runs on
        case 3000 + 34:
            hipLaunchKernelGGL((conv_tile<PREC, G2S1, 2, 5, 4, 16, 32, 1, 4, true>), dim3((unsigned)t.grid, 1, 1), dim3(256), 0, s, a, t);
            break;
        case 3000 + 19:
            hipLaunchKernelGGL((conv_tile<PREC, G2S1, 4, 5, 4, 16, 16, 1, 4, true>), dim3((unsigned)t.grid, 1, 1), dim3(256), 0, s, a, t);
            break;
        // the stem reading a raw (uint8 / 0..255) stack: its "SPLITK" instantiations
        case 2000 + 15:
            hipLaunchKernelGGL((conv_tile<PREC, G2D, 1, 1, 16, 32, 8, 0, 4, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, 1), dim3(256), 0, s, a, t);
            break;
        case 2000 + 31:
            hipLaunchKernelGGL((conv_tile<PREC, G2D, 1, 1, 32, 32, 8, 0, 8, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, 1), dim3(512), 0, s, a, t);
            break;
        case 2000 + 32:
            hipLaunchKernelGGL((conv_tile<PREC, G2P, 1, 1, 16, 32, 8, 0, 4, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, 1), dim3(256), 0, s, a, t);
            break;
        case 2000 + 33:
            hipLaunchKernelGGL((conv_tile<PREC, G2P, 1, 1, 32, 32, 8, 0, 8, true>), dim3((unsigned)t.grid, (unsigned)t.nsplit, 1), dim3(512), 0, s, a, t);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// This source is compiled three times (-DDFFW_TILE_PREC=0/1/2, see the Makefile): one object per arithmetic, so the
// ~40 kernel instantiations of each build in parallel.
#define DFFW_CAT2(a, b) a##b
#define DFFW_CAT(a, b) DFFW_CAT2(a, b)
hipError_t DFFW_CAT(launch_conv_tile_prec, DFFW_TILE_PREC)(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    return launch_conv_tile_p<DFFW_TILE_PREC>(cfg, a, t, s);
}

#if DFFW_TILE_PREC == 0
hipError_t launch_conv_tile_prec1(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s);
hipError_t launch_conv_tile_prec2(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s);
hipError_t launch_conv_tile_lean0(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s);
hipError_t launch_conv_tile(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    if (tile_lean(prec, cfg, a, t)) return launch_conv_tile_lean0(cfg, a, t, s);
    switch (prec) {
        case P_BF16X3: return launch_conv_tile_prec0(cfg, a, t, s);
        case P_FP16: return launch_conv_tile_prec1(cfg, a, t, s);
        case P_BF16: return launch_conv_tile_prec2(cfg, a, t, s);
    }
    return hipErrorInvalidValue;
}
#endif
#endif   // DFFW_TILE_LEAN

}  // namespace dffw
