// conv_stream: persistent, warp-specialised form of the LDS-tiled implicit-GEMM convolution.
//
// conv_tile runs fill -> barrier -> contract -> epilogue once per workgroup; rocprofv3 PMC showed
// its waves parked ~48 % of their life and the matrix pipe 30 % busy: per-workgroup launch/setup,
// DMA and store-drain latencies that <= 3 resident workgroups cannot hide.  Here ONE workgroup per
// CU stays resident and walks a list of tiles:
//   * waves 4..7 ("loaders") only issue LDS-DMA: they stage the footprint of work item i+1
//     (tile, pass, channel group) into the other half of a double-buffered LDS image while
//   * waves 0..3 ("MFMA waves") contract item i out of the current half — their instruction
//     stream is ds_read + v_mfma (+ the weight fragments from L2, one chunk ahead) and the epilogue.
// Each wave has its own vmcnt, so the loaders' outstanding DMA never blocks the MFMA waves'
// weight loads or stores.  One s_barrier per work item hands the buffers over.
// Tile order is XCD-aware as in conv_tile: the workgroups of one XCD interleave over a contiguous
// range of tiles, so tiles in flight at the same time on one XCD are neighbours (halos hit in L2).
#include <cstdio>

#include "dffw_conv_geom.h"

namespace dffw {

constexpr int S_NCW = 4;                       // MFMA (consumer) waves
constexpr int S_NPW = 4;                       // loader (producer) waves
constexpr int S_THREADS = (S_NCW + S_NPW) * 64;

template <int PREC, int GEO, int NT, int TZ, int TY, int TX, int CG>
__global__ __launch_bounds__(S_THREADS) void conv_stream(const ConvArgs a, const TileArgs t) {
    using T = TileT<GEO, TZ, TY, TX, CG>;
    using G = GeoT<GEO>;
    constexpr int PARTS = Fmt<PREC>::PARTS;
    constexpr bool F16 = (PREC == P_FP16);
    constexpr int MTW = T::MT / S_NCW;
    constexpr int CG8 = CG / 8;
    constexpr int PIXB = CG * 2;
    constexpr int PLANEB = (T::FPIX * PIXB + 1023) / 1024 * 1024;
    constexpr int BUFB = PARTS * PLANEB;       // one image
    static_assert(PLANEB < 65536, "lo-plane offset must fit the ds_read immediate");
    static_assert(2 * BUFB <= 160 * 1024, "double-buffered image must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUFB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;

    // ---- this workgroup's tiles: XCD x owns the contiguous range [xs, xe); its workgroups interleave ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int wpx = gridDim.x >> 3;            // workgroups per XCD (grid is a multiple of 8)
    const int q = t.total_tiles >> 3, rem = t.total_tiles & 7;
    const int xs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const int xe = xs + q + (xcd < rem ? 1 : 0);
    const int ntiles = (xe - xs - idx + wpx - 1) / wpx;   // tiles xs+idx, xs+idx+wpx, ...
    if (xs + idx >= xe) return;

    struct Coord {
        int b, gz0, gy0, gx0;
    };
    auto decode = [&](int k) {   // k-th tile of this workgroup
        const int tile = xs + idx + k * wpx;
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

    const int nstage = t.nstage;
    // "fill items": with one channel group the image of a tile serves all passes; otherwise every
    // (pass, stage) restages its own channel group
    const int ipt = nstage == 1 ? 1 : G::NPASS * nstage;
    const int nitems = ntiles * ipt;

    if (wave >= S_NCW) {
        // =================================== loader waves ==========================================
        const int pw = wave - S_NCW;
        const int ps0 = PARTS * a.C0, ps1 = PARTS * a.C1;
        const int64_t samp0 = (int64_t)a.Ni * a.Hi * a.Wi * ps0, samp1 = (int64_t)a.Ni * a.Hi * a.Wi * ps1;
        auto issue_fill = [&](int item) {
            const Coord c = decode(item / ipt);
            const int st = nstage == 1 ? 0 : (item % ipt) % nstage;
            unsigned char *buf = smem + (item & 1) * BUFB;
            constexpr int PPW = 64 / CG8;                                      // pixels per wave instruction
            constexpr int NPI = (T::FPIX + PPW * S_NPW - 1) / (PPW * S_NPW);
            const int iz0 = c.gz0 + G::MINZ, iy0 = c.gy0 * G::S + G::MINY, ix0 = c.gx0 * G::S + G::MINX;
            const int c8 = lane % CG8;
            const int ch = st * CG + c8 * 8;
            const bool second = ch >= a.C0;
            const int cc = second ? ch - a.C0 : ch;
            const int csrc = second ? a.C1 : a.C0;
            const uint16_t *sp = second ? a.in1 + c.b * samp1 : a.in0 + c.b * samp0;
#pragma unroll
            for (int it = 0; it < NPI; ++it) {
                const int pbase = (it * S_NPW + pw) * PPW;                    // wave-uniform first pixel
                if (pbase >= T::FPIX) break;
                const int p = pbase + lane / CG8;
                const int lx = p % T::FXL;
                const int fy = (p / T::FXL) % T::FY;
                const int fz = p / (T::FXL * T::FY);
                const int fx = (G::S == 2) ? (lx < T::FXL / 2 ? 2 * lx : 2 * (lx - T::FXL / 2) + 1) : lx;
                const int iz = iz0 + fz, iy = iy0 + fy, ix = ix0 + fx;
                const bool ok = p < T::FPIX && fx < T::FX && (unsigned)iz < (unsigned)a.Ni && (unsigned)iy < (unsigned)a.Hi &&
                                (unsigned)ix < (unsigned)a.Wi;
                const uint16_t *gp = sp + ((int64_t)((iz * a.Hi + iy) * a.Wi + ix) * (PARTS * csrc) + cc);
#pragma unroll
                for (int part = 0; part < PARTS; ++part) {
                    const uint16_t *src = ok ? gp + part * csrc : a.zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(buf + part * PLANEB + pbase * PIXB), 16, 0, 0);
                }
            }
        };
        if (!(a.dbg & 1)) issue_fill(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // barrier 0: item 0 is in LDS
        for (int i = 0; i < nitems; ++i) {
            if (i + 1 < nitems && !(a.dbg & 1)) {
                issue_fill(i + 1);                         // overlaps the MFMA waves' work on item i
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                               // barrier i+1: item i+1 landed, item i's buffer is free
        }
        return;
    }

    // ====================================== MFMA waves =============================================
    int pofs[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (wave * MTW + j) * 16 + r;
        const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
        pofs[j] = ((tz * T::FY + ty * G::S) * T::FXL + tx) * PIXB;
    }
    constexpr int GA = MTW / 2, GB = MTW - GA;

    __syncthreads();                                       // barrier 0
    int item = 0;
    for (int k = 0; k < ntiles; ++k) {
        const Coord cur = decode(k);
        for (int pass = 0; pass < G::NPASS; ++pass) {
            f32x4 acc[NT][MTW];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < MTW; ++j) acc[nt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int KC = t.KC[pass];
            const int *tab = t.tab[pass] + g;

            for (int st = 0; st < nstage; ++st) {
                const unsigned char *img = smem + (item & 1) * BUFB;
                // Weight fragments and tap offsets travel through a 4-slot register ring, fetched three
                // 32-deep chunks ahead: with ONE MFMA wave per SIMD nothing else hides the L2 latency
                // (~700 cycles) of these loads behind the 240*NT cycles of MFMA work per chunk.
                const short8 *wp = reinterpret_cast<const short8 *>(t.wpk[pass]) + (int64_t)st * KC * (NT * PARTS * 64) + lane;
                short8 w0[NT][PARTS], w1[NT][PARTS], w2[NT][PARTS], w3[NT][PARTS];
                int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                short8 xa[GA][PARTS], xb[GB][PARTS];
                auto fetch = [&](int kc, short8 (&w)[NT][PARTS], int &tv) {
                    if (kc < KC) {
                        const short8 *wn = wp + (int64_t)kc * (NT * PARTS * 64);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int pt = 0; pt < PARTS; ++pt) w[nt][pt] = wn[(nt * PARTS + pt) * 64];
                        tv = tab[kc * 4];
                    }
                };
                // one chunk: operands of group B for this chunk and of group A for the next one are read
                // under the other group's MFMAs (as in conv_tile's pipelined loop)
                auto step = [&](int kc, const short8 (&w)[NT][PARTS], int tcur, int tnxt) {
                    if (kc >= KC) return;
#pragma unroll
                    for (int j = 0; j < GB; ++j)
#pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt) xb[j][pt] = *reinterpret_cast<const short8 *>(img + pofs[GA + j] + tcur + pt * PLANEB);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (PARTS == 2) {
#pragma unroll
                        for (int j = 0; j < GA; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(w[nt][1], xa[j][0], acc[nt][j]);
#pragma unroll
                        for (int j = 0; j < GA; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(w[nt][0], xa[j][1], acc[nt][j]);
                    }
#pragma unroll
                    for (int j = 0; j < GA; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt][j] = mma<F16>(w[nt][0], xa[j][0], acc[nt][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kc + 1 < KC) {
#pragma unroll
                        for (int j = 0; j < GA; ++j)
#pragma unroll
                            for (int pt = 0; pt < PARTS; ++pt) xa[j][pt] = *reinterpret_cast<const short8 *>(img + pofs[j] + tnxt + pt * PLANEB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (PARTS == 2) {
#pragma unroll
                        for (int j = 0; j < GB; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][GA + j] = mma<F16>(w[nt][1], xb[j][0], acc[nt][GA + j]);
#pragma unroll
                        for (int j = 0; j < GB; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt][GA + j] = mma<F16>(w[nt][0], xb[j][1], acc[nt][GA + j]);
                    }
#pragma unroll
                    for (int j = 0; j < GB; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt][GA + j] = mma<F16>(w[nt][0], xb[j][0], acc[nt][GA + j]);
                    __builtin_amdgcn_sched_barrier(0);
                };
                fetch(0, w0, t0);
                fetch(1, w1, t1);
                fetch(2, w2, t2);
#pragma unroll
                for (int j = 0; j < GA; ++j)
#pragma unroll
                    for (int pt = 0; pt < PARTS; ++pt) xa[j][pt] = *reinterpret_cast<const short8 *>(img + pofs[j] + t0 + pt * PLANEB);
                for (int kc = 0; kc < ((a.dbg & 2) ? 1 : KC); kc += 4) {
                    fetch(kc + 3, w3, t3);
                    step(kc, w0, t0, t1);
                    fetch(kc + 4, w0, t0);
                    step(kc + 1, w1, t1, t2);
                    fetch(kc + 5, w1, t1);
                    step(kc + 2, w2, t2, t3);
                    fetch(kc + 6, w2, t2);
                    step(kc + 3, w3, t3, t0);
                }
                if (nstage > 1) {          // this channel group's image is consumed: hand the buffer back
                    __syncthreads();
                    ++item;
                }
            }

            // ---- epilogue of this pass (dffw_device.h) -------------------------------------------------------
            const int ooy = t.ooy[pass], oox = t.oox[pass];
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                const int p = (wave * MTW + j) * 16 + r;
                const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
                const int gz = cur.gz0 + tz, gy = cur.gy0 + ty, gx = cur.gx0 + tx;
                bool pv = gz < a.Ng && gy < a.Hg && gx < a.Wg;
                if ((a.dbg & 4) && acc[0][j][0] != 12345.f) pv = false;
                const int64_t opix = (((int64_t)cur.b * a.No + gz) * a.Ho + (gy * G::OS + ooy)) * a.Wo + (gx * G::OS + oox);
                float cls = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) epilogue_quad<PREC, false>(a, acc[nt][j], nt, g, opix, pv, cls, uint4{}, uint4{});
                epilogue_cls(a, cls, g, opix, pv);
            }
        }
        if (nstage == 1) {                 // all passes of this tile have read the image
            __syncthreads();
            ++item;
        }
    }
}

// ---- configuration table (ids are independent of conv_tile's) ---------------------------------------
//        id  geo   NT  TZ TY  TX  CG
#define DFFW_STREAM_CONFIGS(X)    \
    X(0, G3S1, 1, 5, 4, 16, 16)   \
    X(1, G3S1, 1, 5, 8, 16, 8)    \
    X(2, G3S1, 2, 5, 4, 16, 16)   \
    X(3, G3T, 1, 5, 4, 16, 16)    \
    X(4, G3T, 2, 5, 4, 16, 16)    \
    X(5, G2S1, 1, 5, 8, 16, 8)    \
    X(6, G2S1, 1, 5, 8, 16, 16)   \
    X(7, G2S1, 2, 5, 8, 16, 16)   \
    X(8, G3S2, 1, 5, 4, 16, 8)    \
    X(9, G3S2, 2, 5, 4, 16, 8)

#define X_CFG(ID, GEO, NT, TZ, TY, TX, CG)                                                             \
    TileCfg{ID, GEO, NT, CG, TZ, TY, TX, TileT<GEO, TZ, TY, TX, CG>::FZ, TileT<GEO, TZ, TY, TX, CG>::FY, \
            TileT<GEO, TZ, TY, TX, CG>::FX, TileT<GEO, TZ, TY, TX, CG>::FXL, 2},
static const TileCfg g_scfgs[] = {DFFW_STREAM_CONFIGS(X_CFG)};
#undef X_CFG

const TileCfg *stream_cfg_find(int geo, int nt, int cg) {
    for (const TileCfg &c : g_scfgs)
        if (c.geo == geo && c.nt == nt && c.cg == cg) return &c;
    return nullptr;
}

void conv_stream_kernel_name(int prec, const TileCfg *c, char *buf, int n) {
    snprintf(buf, n, "dffw::conv_stream<%d, %d, %d, %d, %d, %d, %d>", prec, c->geo, c->nt, c->tz, c->ty, c->tx, c->cg);
}

template <int PREC>
static hipError_t launch_conv_stream_p(const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    switch (cfg->id) {
#define X_LAUNCH(ID, GEO, NT, TZ, TY, TX, CG)                                                                          \
    case ID:                                                                                                           \
        hipLaunchKernelGGL((conv_stream<PREC, GEO, NT, TZ, TY, TX, CG>), dim3((unsigned)t.grid), dim3(S_THREADS), 0, s, a, t); \
        break;
        DFFW_STREAM_CONFIGS(X_LAUNCH)
#undef X_LAUNCH
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_stream(int prec, const TileCfg *cfg, const ConvArgs &a, const TileArgs &t, hipStream_t s) {
    switch (prec) {
        case P_BF16X3: return launch_conv_stream_p<P_BF16X3>(cfg, a, t, s);
        case P_FP16: return launch_conv_stream_p<P_FP16>(cfg, a, t, s);
        case P_BF16: return launch_conv_stream_p<P_BF16>(cfg, a, t, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dffw
