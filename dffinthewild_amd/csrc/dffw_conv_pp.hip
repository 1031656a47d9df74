// conv_pp: "ping-pong" form of the LDS-tiled implicit-GEMM convolution for the 32..192-channel 3x3x3 layers (the cost
// aggregation of DEN.py:145-284: SPP stacks, dres0, hourglass conv0/2/4, and their stride-(1,2,2) siblings).
//
// conv_tile walks fill -> barrier -> contract -> epilogue once per workgroup and leaves the overlap of those phases to
// whatever other workgroups share the CU.  Measured: they do not overlap -- the co-resident workgroups
// drift into step (all contracting, then all filling), the matrix pipe is 41 % busy and the phases' times ADD.
// Here the alternation is explicit.  One persistent workgroup per CU holds TWO groups of four waves (one wave of each
// group per SIMD) and two LDS images; time is cut into slots separated by a workgroup barrier, and in every slot
//     one group CONTRACTS a channel-group stage of its tile out of its image (ds_read + MFMA, weights from L2), while
//     the other group runs the epilogue of the tile it just finished and/or stages its next stage by LDS-DMA,
// then the roles swap.  So each SIMD always has one wave feeding the matrix pipe and one wave doing memory work, the DMA
// and store latencies of one group lie under the MFMAs of the other by construction, and the in-order vmcnt problem of a
// double-buffered single group (weight-fragment waits draining the DMA queue) does not arise: a wave never has DMA in
// flight while it contracts.  Tiles, tap tables, weight packing and the epilogue are conv_tile's (same TileCfg), so a
// layer can run on either kernel from one packing.
//
// Work units: (tile, output-channel slab).  Few-tile layers split their output channels over units (t.nsplit) so that
// both groups -- and all CUs -- have work; the two groups of a workgroup then take the two slabs of the SAME tile.
// XCD x (= blockIdx % 8) owns a contiguous range of units and its workgroups interleave over it (halos hit in that L2).
#include <cstdio>
#include <cstdlib>

#include "dffw_conv_geom.h"
#include "dffw_conv_pp.h"

namespace dffw {

// NG: wave groups per workgroup.  2: one group contracts while the other loads.  3: a stage's contraction spans two
// slots, so TWO groups (two waves per SIMD) feed the matrix pipe at any time while the third loads -- a single contracting
// wave per SIMD was measured at half the matrix peak (its ds_read / weight-load latencies are exposed), two cover each other.
template <int PREC, int GEO, int NT, int TZ, int TY, int TX, int CG, int NG>
__global__ __launch_bounds__(NG * 256) void conv_pp(const ConvArgs a, const TileArgs t) {
    using T = TileT<GEO, TZ, TY, TX, CG>;
    using G = GeoT<GEO>;
    static_assert(G::NPASS == 1, "single-pass geometries (3x3x3 stride 1 / stride (1,2,2))");
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int GW = 4;                       // waves per group
    constexpr int MTW = T::MT / GW;             // operand tiles (16 grid points) per wave
    constexpr int CG8 = CG / 8;
    constexpr int PIXB = CG * 2;
    constexpr int PLANEB = (T::FPIX * PIXB + 1023) / 1024 * 1024;
    constexpr int LDSB = PARTS * PLANEB;        // one group's image
    static_assert(PLANEB < 65536, "lo-plane offset must fit the ds_read immediate");
    static_assert(NG == 2 || NG == 3, "two or three wave groups");
    static_assert(NG * LDSB <= 160 * 1024, "one image per group must fit the 160 KiB LDS");
    constexpr int NMP = NG - 1;                 // matrix slots per step
    __shared__ __attribute__((aligned(1024))) unsigned char smem_all[NG * LDSB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = wave >> 2, gw = wave & 3;
    const int g = lane >> 4, r = lane & 15;
    unsigned char *smem = smem_all + grp * LDSB;
    const int NTT = t.nt_total;
    const int S = t.nstage;

    // ---- this group's units ---------------------------------------------------------------------------------------
    const int total_units = t.total_tiles * t.nsplit;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    int xs, xe;
    {
        const int q = total_units >> 3, rem = total_units & 7;
        xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        xe = xs + q + (xcd < rem ? 1 : 0);
    }
    // unit k of group grp: xs + (k*wpx + idx)*NG + grp
    auto nunits_of = [&](int gr) {
        const int first = xs + idx * NG + gr;
        return first < xe ? (xe - first + NG * wpx - 1) / (NG * wpx) : 0;
    };
    const int n_steps = nunits_of(grp) * S;            // this group's (unit, stage) steps
    const int n_slots = NG * nunits_of(0) * S + NG;    // group 0 never has fewer units than the others
    if (n_slots == NG) return;

    struct Coord {
        int b, gz0, gy0, gx0, ntb;
    };
    auto decode = [&](int k) {
        const int u = xs + (k * wpx + idx) * NG + grp;
        const int tile = u / t.nsplit;
        Coord c;
        c.ntb = (u - tile * t.nsplit) * NT;
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

    // ---- per-lane constants of this wave's operand tiles (as in conv_tile) ------------------------------------------
    // (the output-side constants -- tile coordinates and the lane's element offset -- are recomputed in the epilogue rather
    // than held in registers across the contraction: this kernel lives at 3 waves per SIMD, 168 registers)
    int pofs[MTW];
    const int lanepart = (PARTS == 2) ? (g & 1) * a.Cout + (g >> 1) * 8 : g * 4;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (gw * MTW + j) * 16 + r;
        const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
        pofs[j] = ((tz * T::FY + ty * G::S) * T::FXL + tx) * PIXB;
    }

    const int ps0 = PARTS * a.C0, ps1 = PARTS * a.C1;
    const int64_t samp0 = (int64_t)a.Ni * a.Hi * a.Wi * ps0, samp1 = (int64_t)a.Ni * a.Hi * a.Wi * ps1;

    // ---- LDS-DMA of the footprint of channel group `st` of unit `c` into this group's image (issue only) -------------
    auto issue_fill = [&](const Coord &c, int st) {
        constexpr int PPW = 64 / CG8;
        constexpr int NPI = (T::FPIX + PPW * GW - 1) / (PPW * GW);
        const int iz0 = c.gz0 + G::MINZ, iy0 = c.gy0 * G::S + G::MINY, ix0 = c.gx0 * G::S + G::MINX;
        const int c8 = lane % CG8;
        const int ch = st * CG + c8 * 8;
        const bool second = ch >= a.C0;
        const int cc = second ? ch - a.C0 : ch;
        const int csrc = second ? a.C1 : a.C0;
        const uint16_t *sp = second ? a.in1 + c.b * samp1 : a.in0 + c.b * samp0;
        constexpr int STEP = PPW * GW;
        constexpr int DLX = STEP % T::FXL, DFY = (STEP / T::FXL) % T::FY, DFZ = STEP / (T::FXL * T::FY);
        const int p0 = gw * PPW + lane / CG8;
        int lx = p0 % T::FXL, fy = (p0 / T::FXL) % T::FY, fz = p0 / (T::FXL * T::FY);
        const bool yx_in = iy0 >= 0 && iy0 + T::FY <= a.Hi && ix0 >= 0 && ix0 + T::FX <= a.Wi;
        const int pst = PARTS * csrc;
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int pbase = (it * GW + gw) * PPW;
            if (pbase >= T::FPIX) break;
            const int fx = (G::S == 2) ? (lx < T::FXL / 2 ? 2 * lx : 2 * (lx - T::FXL / 2) + 1) : lx;
            const int iz = iz0 + fz, iy = iy0 + fy, ix = ix0 + fx;
            bool ok = (unsigned)iz < (unsigned)a.Ni;
            if ((it + 1) * STEP > T::FPIX) ok = ok && (pbase + lane / CG8 < T::FPIX);
            if (T::FXL > T::FX) ok = ok && fx < T::FX;
            if (!yx_in) ok = ok && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const uint16_t *gp = sp + ((int64_t)(((iz * a.Hi + iy) * a.Wi + ix) * pst) + cc);
#pragma unroll
            for (int part = 0; part < PARTS; ++part) {
                const uint16_t *src = ok ? gp + part * csrc : a.zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + part * PLANEB + pbase * PIXB), 16, 0, 0);
            }
            if (t.ksplit > 1) {   // pacing knob (DFFW_PP_PACE): spread the DMA over the slot so the contracting group's weight loads slip in
                switch (t.ksplit) {
                    case 2: __builtin_amdgcn_s_sleep(2); break;
                    case 4: __builtin_amdgcn_s_sleep(4); break;
                    case 8: __builtin_amdgcn_s_sleep(8); break;
                    default: __builtin_amdgcn_s_sleep(1); break;
                }
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

    const int KC = t.KC[0];
    const int *tab0 = t.tab[0] + g;
    const int wstride = NTT * PARTS * 64;   // fragments (16 B per lane) per 32-deep chunk

    f32x4 acc[NT][MTW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < MTW; ++j) acc[nt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue of one finished unit (shared with conv_tile / conv_igemm, dffw_device.h) ---------------------------
    auto epilogue = [&](const Coord &cur) {
        const int64_t obase = (((int64_t)cur.b * a.No + cur.gz0) * a.Ho + cur.gy0) * a.Wo + cur.gx0;
        const int64_t ubase = obase * (PARTS * a.Cout);
        const bool interior = cur.gz0 + TZ <= a.Ng && cur.gy0 + TY <= a.Hg && cur.gx0 + TX <= a.Wg;
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (gw * MTW + j) * 16 + r;
            const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
            const int pixo = (tz * a.Ho + ty) * a.Wo + tx;
            const int voff = pixo * (PARTS * a.Cout) + lanepart;
            const int64_t opix = obase + pixo;
            bool pv = true;
            if (!interior) pv = cur.gz0 + tz < a.Ng && cur.gy0 + ty < a.Hg && cur.gx0 + tx < a.Wg;
            if ((a.dbg & 4) && acc[0][j][0] != 12345.f) pv = false;
            float cls = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) epilogue_quad<PREC, false, true, false>(a, acc[nt][j], cur.ntb + nt, g, opix, pv, cls, uint4{}, uint4{}, ubase, voff);
            epilogue_cls(a, cls, g, opix, pv);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- contraction of chunks [kc_lo, kc_hi) of one stage.  ONE wave per SIMD feeds the matrix pipe here, so nothing else
    // covers this wave's own issue gaps: measured with conv_tile's loop (operand reads fenced into bursts between MFMA
    // blocks) a bare 30-MFMA chunk took 22 cycles per MFMA instead of 16-17.  So: operand fragments of chunk k+1 (LDS) and
    // the weight fragments of chunk k+1 (L2) are requested WHILE chunk k's MFMAs issue, spread one request per few MFMAs
    // (sched_group_barrier pattern below), into the other half of register double buffers (the loop is unrolled by two
    // so that the halves alternate without copies); every chunk starts with all its operands already in registers. ------
    auto contract = [&](const Coord &cur, int st, int kc_lo, int kc_hi) {
        const short8 *wp = reinterpret_cast<const short8 *>(t.wpk[0]) + ((int64_t)st * KC + kc_lo) * wstride + cur.ntb * PARTS * 64 + lane;
        const int *tab = tab0 + kc_lo * 4;
        const int nkc = (a.dbg & 2) ? 1 : kc_hi - kc_lo;
        short8 w[2][NT][PARTS];
        short8 x[2][MTW][PARTS];
        auto load_w = [&](int kc, short8 (&dst)[NT][PARTS]) {
            const short8 *wn = wp + (int64_t)kc * wstride;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) dst[nt][pt] = wn[(nt * PARTS + pt) * 64];
        };
        auto load_x = [&](int toff, short8 (&dst)[MTW][PARTS]) {
#pragma unroll
            for (int j = 0; j < MTW; ++j)
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) dst[j][pt] = *reinterpret_cast<const short8 *>(smem + pofs[j] + toff + pt * PLANEB);
        };
        auto mfmas = [&](short8 (&wc)[NT][PARTS], short8 (&xc)[MTW][PARTS]) {
            // product-major inside a tile pair so that consecutive MFMAs never share an accumulator
            if constexpr (PARTS == 2) {
#pragma unroll
                for (int j = 0; j < MTW; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wc[nt][1], xc[j][0], acc[nt][j]);
#pragma unroll
                for (int j = 0; j < MTW; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wc[nt][0], xc[j][1], acc[nt][j]);
            }
#pragma unroll
            for (int j = 0; j < MTW; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(wc[nt][0], xc[j][0], acc[nt][j]);
        };
        // one chunk: MFMAs of chunk kc out of (wc, xc) with the requests for chunk kn (= kc+1, or kc again at the end of the
        // range: the requests are unconditional so that the chunk is ONE basic block -- hipcc then counts its waits exactly
        // and the issue pattern below can interleave) into (wn, xn) spread between them
        auto chunk = [&](int kn, int tnext, short8 (&wc)[NT][PARTS], short8 (&xc)[MTW][PARTS], short8 (&wn)[NT][PARTS], short8 (&xn)[MTW][PARTS]) {
            __builtin_amdgcn_sched_barrier(0);
#ifdef DFFW_PP_EXPERIMENT
            if (!(a.dbg & 16)) load_w(kn, wn);
            if (!(a.dbg & 32)) load_x(tnext, xn);
#else
            load_w(kn, wn);
            load_x(tnext, xn);
#endif
            mfmas(wc, xc);
            // issue pattern: the weight fragments of the next chunk first, one per MFMA (they come from L2: they need the
            // whole chunk as cover), then its operand tiles, PARTS ds_reads per two MFMAs; the remaining MFMAs follow bare
#pragma unroll
            for (int i = 0; i < NT * PARTS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read
            }
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);        // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, PARTS, 0);    // DS read
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        const int last = nkc - 1;
        int t1 = tab[min(1, last) * 4];
        load_w(0, w[0]);
        load_x(tab[0], x[0]);
        int kc = 0;
        for (; kc + 1 < nkc; kc += 2) {
            const int t2 = tab[min(kc + 2, last) * 4];
            chunk(kc + 1, t1, w[0], x[0], w[1], x[1]);
            t1 = tab[min(kc + 3, last) * 4];
            chunk(min(kc + 2, last), t2, w[1], x[1], w[0], x[0]);
        }
        if (kc < nkc) {   // odd count: the last chunk on its own (its operands are in the first halves)
            __builtin_amdgcn_sched_barrier(0);
            mfmas(w[0], x[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the slot loop.  Local slot j = slot - grp of a group, phase j % NG: phase 0 = loader slot of step j / NG (epilogue
    // of the unit finished by the previous step when that was a unit's last stage, then the DMA of this step's stage), phases
    // 1 .. NG-1 = matrix slots of the same step (the stage's chunks in NG-1 parts).  Group g lags g slots behind group 0,
    // so in every slot exactly one group loads and the others contract.  Every wave executes every barrier. ------------------
    // debug timeline (make TRACE=1, DFFW_TRACE_LAYER): lane 0 of each group's first wave stamps s_memtime per slot:
    // [workgroup][slot < 512][group < 4][4] = slot start, after epilogue + DMA issue (loader) , phase done, after the barrier
    auto stamp = [&](int slot, int k) {
#ifdef DFFW_TRACE_BUILD
        if (a.trace && gw == 0 && lane == 0 && slot < 512) a.trace[(((int64_t)blockIdx.x * 512 + slot) * 4 + grp) * 4 + k] = __builtin_amdgcn_s_memtime();
#else
        (void)slot; (void)k;
#endif
    };
    for (int slot = 0; slot < n_slots; ++slot) {
        const int j = slot - grp;
        stamp(slot, 0);
        if (j >= 0) {
            const int q = j / NG, ph = j - q * NG;
            if (ph == 0) {
                if (q > 0 && q <= n_steps && q % S == 0 && !(a.dbg & 64)) epilogue(decode(q / S - 1));
                if (q < n_steps && !(a.dbg & (1 | 64))) issue_fill(decode(q / S), q % S);
                stamp(slot, 1);
                // the image must have landed before the barrier hands it to this group's matrix slots (compiler-visible wait:
                // hipcc then knows no DMA is pending while the group contracts)
                __builtin_amdgcn_s_waitcnt(0);
            } else if (q < n_steps) {
                if (t.pass_split) __builtin_amdgcn_s_setprio(3);
                const Coord cur = decode(q / S);
                if (q % S == 0 && ph == 1) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + (cur.ntb + nt) * 16 + g * 4);
#pragma unroll
                        for (int jj = 0; jj < MTW; ++jj) acc[nt][jj] = b4;
                    }
                }
                contract(cur, q % S, (ph - 1) * KC / NMP, ph * KC / NMP);
                if (t.pass_split) __builtin_amdgcn_s_setprio(0);
            }
        }
        stamp(slot, 2);
        __syncthreads();
        stamp(slot, 3);
    }
}

// ---- configurations: the conv_tile configurations (same ids, same packing) that have a ping-pong instantiation ------
//        id  geo   NT  TZ TY  TX  CG  NG
#define DFFW_PP_CONFIGS(X)         \
    X(0, G3S1, 1, 5, 4, 16, 16, 2) \
    X(1, G3S1, 1, 5, 4, 16, 8, 2)  \
    X(2, G3S1, 2, 5, 4, 16, 16, 2) \
    X(16, G3S1, 2, 4, 4, 8, 16, 3) \
    X(5, G3S2, 1, 5, 4, 16, 8, 2)  \
    X(6, G3S2, 2, 5, 4, 16, 8, 2)  \
    X(12, G2S1, 1, 5, 4, 16, 8, 2) \
    X(13, G2S1, 1, 5, 4, 16, 16, 2) \
    X(18, G2S1, 2, 5, 4, 16, 8, 2)

#ifndef DFFW_PP_NG2
#define DFFW_PP_NG2 2
#endif
#ifndef DFFW_PP_PREC
#define DFFW_PP_PREC 0
#endif

bool conv_pp_has(const TileCfg *c) {
    if (!c || c->nw != 4) return false;
    switch (c->id) {
#define X_HAS(ID, GEO, NT, TZ, TY, TX, CG, NG) case ID:
        DFFW_PP_CONFIGS(X_HAS)
#undef X_HAS
        return true;
        default: return false;
    }
}

int conv_pp_groups(const TileCfg *c) {
    switch (c ? c->id : -1) {
#define X_NG(ID, GEO, NT, TZ, TY, TX, CG, NG) \
    case ID: return NG;
        DFFW_PP_CONFIGS(X_NG)
#undef X_NG
        default: return 0;
    }
}

void conv_pp_kernel_name(int prec, const TileCfg *c, char *buf, int n) {
    snprintf(buf, n, "dffw::conv_pp<%d, %d, %d, %d, %d, %d, %d, %d>", prec, c->geo, c->nt, c->tz, c->ty, c->tx, c->cg, conv_pp_groups(c));
}

template <int PREC>
static hipError_t launch_conv_pp_p(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    switch (cfg->id) {
#define X_LAUNCH(ID, GEO, NT, TZ, TY, TX, CG, NG)                                                                           \
    case ID:                                                                                                                \
        hipLaunchKernelGGL((conv_pp<PREC, GEO, NT, TZ, TY, TX, CG, NG>), dim3((unsigned)t.grid), dim3(NG * 256), 0, s, a, t); \
        break;
        DFFW_PP_CONFIGS(X_LAUNCH)
#undef X_LAUNCH
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_pp(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    switch (prec) {
        case P_BF16X3: return launch_conv_pp_p<P_BF16X3>(cfg, a, t, s);
        case P_FP16: return launch_conv_pp_p<P_FP16>(cfg, a, t, s);
        case P_BF16: return launch_conv_pp_p<P_BF16>(cfg, a, t, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dffw
