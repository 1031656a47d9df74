// conv_wino32: a 3x3x3 stride-1 padding-1 conv over 32 input channels with the in-plane 3x3 taps in Winograd F(2x2, 3x3) form
// (the reference's nn.Conv3d + folded BatchNorm3d of the 32-channel hourglass / SPP layers, Depth_Estimation_Network/DEN.py:5-11, 96-120
// and submodule.py:117-160).  16 MFMA multiplies per 2x2 outputs and slice tap instead of 36:
//     U = G g G^T        per (cout, cin, dz): done once at weight-pack time in float64, split into bf16 hi + lo (pack_conv)
//     V = B^T d B        per 4x4 input patch: fp32 from the stored hi + lo activations, re-split into hi + lo         (T)
//     M[pos] += U[pos][dz] . V[pos][z + dz - 1]    three MFMA products each, fp32 accumulation, 16 positions          (M)
//     y = A^T M A        fp32, then the usual epilogue (BatchNorm shift, residual, ReLU, hi + lo split)               (O)
// The arithmetic is not the direct kernels' (sums are re-associated by the transforms), so results differ from conv_tile's in the
// last bits; tools/winograd_emulation.py measured pred3 on the nine goldens at 0.97e-5 .. 4.2e-5 rel-L2 (direct: 0.65e-5 .. 3.7e-5).
//
// One workgroup of 8 waves walks one column of 4 x 16 output pixels (16 blocks of 2 x 2) through the slices, slice-stationary: input
// slice z is transformed once and contributes to the three output slices z+1, z, z-1 (three accumulator sets whose roles rotate with
// the step; the step body is instantiated per rotation so that no register moves).  Wave w owns positions 2w and 2w+1 for all 16
// blocks: its filter fragments U[2][3 dz][2 nt][hi, lo] = 96 VGPRs stay in registers for the whole column and a step is 36 MFMAs per
// wave for 4 LDS fragment reads.  LDS (134 KB): V of two slices (16 positions x 16 blocks x 32 channels x hi + lo = 32.5 KB each; 64-byte
// block pitch with the channel octets XOR-swizzled by the block's upper half: conflict-free ds_read_b128 without padding), the 42 KB fp32
// hand-over of finished M values (160-byte block pitch, positions 128 B out of phase), two raw slice footprints (6 x 18 pixel records,
// piece pairs XOR-swizzled by the pixel's (x >> 1, y >> 1) parities for the transposing reads; filled by LDS-DMA two steps ahead).  A full step runs three independent parts in one instruction stream:
//     M(s):   the wave contracts V[s & 1] (its two positions, three slice taps)
//     T(s+1): the input transform of the next slice ON THE MATRIX CORE: wave w takes blocks 2w, 2w+1; the data operand (8 patch pixels of one
//             channel per lane) comes through ds_read_b64_tr_b16, the other operand is the constant 0 / +-1 matrix B^T (x) B^T, and adding
//             the hi and lo parts in the fp32 accumulator is the join -- 8 reads + 4 MFMAs + the hi/lo re-split per wave
//     O(s-2): every thread turns 3 x 3 hand-over values into one output pixel x 4 channels (output transform + epilogue)
// then barrier, hand-over write of the finished accumulator set, barrier.  Every LDS access is inline asm with counted waits: behind an
// outstanding LDS-DMA hipcc would put vmcnt(0) in front of it, i.e. wait for the slice just requested.  Measured: 9-13 %
// under conv_tile per layer, level on the whole forward; opt-in through DFFW_WINO_MIN_UNITS.
#include "dffw_conv_wino.h"
#include "dffw_device.h"

#include <cstdio>
#include <type_traits>

namespace dffw {
namespace {

constexpr int WB = 16;                          // blocks per column step (2 rows x 8)
constexpr int W_PITCH = 64;                     // bytes per (position, block) in one V plane: 32 channels x 2 B; channel octet o of block r
                                                // sits at 16-byte slot o ^ ((r >> 3) << 1): conflict-free ds_read_b128 without padding
constexpr int W_VPOS = WB * W_PITCH + 16;       // bytes per position: 16 B out of phase, so that T's writes (a lane per position) spread over the banks
constexpr int W_VPLANE = 16 * W_VPOS;           // one part (hi or lo) of V
constexpr int W_VBUF = 2 * W_VPLANE;            // V of one slice
constexpr int W_MOFF = 2 * W_VBUF;              // hand-over buffer: [position][block][32 output channels] fp32
constexpr int W_MROW = 160, W_MPOS = WB * W_MROW + 128;   // 128 B of channels per block at a 160-byte pitch (conflict-free b128 writes),
                                                           // positions 128 B out of phase (the two pixels of an O row read both halves at once)
constexpr int W_MBYTES = 16 * W_MPOS;
constexpr int W_FX = 18, W_FPIX = 6 * W_FX;     // a slice's input footprint: (4 + 2) x (16 + 2) pixels
constexpr int W_RAWOFF = W_MOFF + W_MBYTES;
constexpr int W_RAWSLOT = W_FPIX * 128;         // ... as 128-byte records; two slots
constexpr int W_BIASOFF = W_RAWOFF + 2 * W_RAWSLOT;   // the slab's 32 BatchNorm shifts (fp32)
constexpr int W_TCOFF = W_BIASOFF + 128;           // the transform's constant operand, one 16-byte fragment per lane (the same in every wave)
constexpr int W_LDS = W_TCOFF + 1024;
static_assert(W_LDS <= 160 * 1024, "LDS budget");

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2w;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4w;

}  // namespace

template <int PREC, bool RES>
__global__ __launch_bounds__(512) void conv_wino32(const ConvArgs a, const WinoArgs t) {
    static_assert(PREC == P_BF16X3, "the Winograd path exists for the split-bf16 storage only");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int slab = blockIdx.y;
    int u = blockIdx.x;
    const int tx = u % t.tiles_x;
    u /= t.tiles_x;
    const int ty = u % t.tiles_y;
    const int bs = u / t.tiles_y;
    const int y0 = ty * WINO_TY, x0 = tx * WINO_TX;
    const int N = a.Ni, H = a.Hi, W = a.Wi;

    // ---- the wave's filter fragments and accumulators -----------------------------------------------------------------
    short8 U[2][3][2][2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    U[pp][dz][nt][part] = *reinterpret_cast<const short8 *>(
                        t.u + ((((((size_t)slab * 16 + (2 * wave + pp)) * 3 + dz) * 2 + nt) * 2 + part) * 512) + lane * 8);
    f32x4 P[3][2][2];   // three accumulator sets x (position of the wave) x (16-channel output tile)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) P[k][pp][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // M: the wave's V fragments (positions 2w, 2w+1; block r, channel octet g) and its hand-over rows
    const unsigned vrd = lds0 + (2 * wave) * W_VPOS + r * W_PITCH + (g ^ ((r >> 3) << 1)) * 16;
    const unsigned mwr = lds0 + W_MOFF + (2 * wave) * W_MPOS + r * W_MROW + g * 16;

    // T runs on the matrix core too: for one block and 16 channels, D[channel][position] = sum over the 32 keys (part, patch pixel) of
    // d_part[pixel][channel] * (B^T[xi][i] B^T[nu][j]) -- the constant operand holds 0 / +-1, exact in bf16, and summing both parts in the fp32
    // accumulator IS the hi + lo join.  The data operand wants, per lane, 8 pixels of ONE channel from records that are channel-contiguous:
    // ds_read_b64_tr_b16 (each lane of a 16-lane group addresses one 8-byte piece [key row i >> 2][channel chunk i & 3]; lane n receives
    // column n of the group's 4 x 16 block).  Wave w transforms blocks 2w, 2w+1 (both channel groups): 8 reads + 4 MFMAs per step.
    // (row 2 of B^T and of G are both negated against the textbook matrices: the products U.V do not change)
    const int li = lane & 15;
    const int tby = wave >> 2, tbx0 = 2 * (wave & 3);
    // (raw records are stored with their four 32-byte piece pairs [part][channel group] XOR-ed by ((x >> 1) & 1) | (((y >> 1) & 1) << 1) of the
    // footprint pixel: the 8 pixels x 32 B one half-wave of a transposing read covers -- 4 columns x 2 rows two apart -- then fall on
    // 8 different bank groups instead of 2.  Channel group 1 and block 1 each flip bit 0 of that index: address ^ 32.)
    const int tsw = ((tbx0 + (li >> 3)) & 1) | (((tby + (g & 1)) & 1) << 1);
    const unsigned trd = lds0 + W_RAWOFF + ((2 * tby + (g & 1) * 2) * W_FX + 2 * tbx0 + (li >> 2)) * 128 + ((((g >> 1) * 2) ^ tsw) * 32) + (li & 3) * 8;
    // write side: lane (g, n) holds channels 4g .. 4g+3 of the group for position n
    const unsigned twr = lds0 + r * W_VPOS + (2 * wave) * W_PITCH + ((g >> 1) ^ (tby << 1)) * 16 + (g & 1) * 8;
    const int tcg = tby ? -32 : 32;   // the second channel group's octets: slot bit 1 flipped
    short8 tconst0;
    {
        const int xi = r >> 2, nu = r & 3;
        auto bt = [](int x, int i) { return x == 0 ? (i == 0 ? 1 : i == 2 ? -1 : 0) : x == 1 ? (i == 1 || i == 2 ? 1 : 0) : x == 2 ? (i == 1 ? 1 : i == 2 ? -1 : 0) : (i == 1 ? 1 : i == 3 ? -1 : 0); };
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int v = bt(xi, (g & 1) * 2 + (e >> 2)) * bt(nu, e & 3);
            tconst0[e] = (short)(v == 0 ? 0 : v > 0 ? 0x3F80 : 0xBF80);
        }
    }

    // slice fill by LDS-DMA: 108 pixels x 8 pieces of 16 B, two rounds of 512 lanes; one wave instruction = 1 KB of the slot
    uint32_t foff[2];
    bool fok[2], fin[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = k * 512 + tid, pix = p >> 3, fy = pix / W_FX, fx = pix - fy * W_FX;
        const int iy = y0 - 1 + fy, ix = x0 - 1 + fx;
        fin[k] = pix < W_FPIX;
        fok[k] = fin[k] && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int fsw = ((fx >> 1) & 1) | (((fy >> 1) & 1) << 1);               // the slot's piece pair holds the record's pair (slot ^ fsw)
        foff[k] = (((uint32_t)(bs * N) * H + iy) * W + ix) * 128u + (((((p >> 1) & 3) ^ fsw) << 1) | (p & 1)) * 16u;   // (the launcher checks the volume is < 4 GB)
    }
    const uint32_t slice_b = (uint32_t)H * W * 128u;
    const unsigned char *inb = reinterpret_cast<const unsigned char *>(a.in0);
    auto fill = [&](int z) __attribute__((always_inline)) {
        unsigned char *slot = smem + W_RAWOFF + (z & 1) * W_RAWSLOT;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const void *src = fok[k] ? (const void *)(inb + (foff[k] + (uint32_t)z * slice_b)) : (const void *)a.zero;
            if (fin[k])
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + (k * 8 + wave) * 1024), 16, 0, 0);
        }
    };

    // O: (block, output row, pixel of the row, output-channel quad): one output pixel x 4 channels from 3 x 3 hand-over values.
    // A^T row 0 = (1, 1, 1, 0), row 1 = (0, 1, -1, -1): output row orow uses M rows orow .. orow+2, pixel px columns px .. px+2
    const int oq = tid & 7, opx = (tid >> 3) & 1, orow = (tid >> 4) & 1, ob = tid >> 5;
    const unsigned mrd = lds0 + W_MOFF + (orow * 4 + opx) * W_MPOS + ob * W_MROW + oq * 16;
    const float sgr = orow ? -1.f : 1.f, sgc = opx ? -1.f : 1.f;
    const int oco = slab * 32 + oq * 4;
    const int C = a.Cout;
    const unsigned brd = lds0 + W_BIASOFF + oq * 16;
    const unsigned tcrd = lds0 + W_TCOFF + lane * 16;
    const float rfloor = a.relu ? 0.f : -__builtin_inff();
    // element offset of the thread's output pixel in slice 0 (the launcher checks the output volume is < 2^31 elements)
    const uint32_t ooff0 = ((((uint32_t)(bs * N) * H + (y0 + 2 * (ob >> 3) + orow)) * W + (x0 + 2 * (ob & 7) + opx)) * 2u * C) + oco;
    const uint32_t oslice = (uint32_t)H * W * 2u * C;

    // T of slice z into V[z & 1] (raw slice in raw[z & 1])
    auto transform = [&](int z) __attribute__((always_inline)) {
        const unsigned rs = trd + (z & 1) * W_RAWSLOT;
        const unsigned ws = twr + (z & 1) * W_VBUF;
        u32x2w ta[4][2];
        short8 tconst;
        asm volatile("ds_read_b64_tr_b16 %0, %9\n\tds_read_b64_tr_b16 %1, %9 offset:%11\n\t"
                     "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %10 offset:%11\n\t"
                     "ds_read_b64_tr_b16 %4, %10 offset:256\n\tds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %6, %9 offset:256\n\tds_read_b64_tr_b16 %7, %9 offset:%12\n\tds_read_b128 %8, %13\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ta[0][0]), "=&v"(ta[0][1]), "=&v"(ta[1][0]), "=&v"(ta[1][1]), "=&v"(ta[2][0]), "=&v"(ta[2][1]), "=&v"(ta[3][0]), "=&v"(ta[3][1]),
                       "=&v"(tconst)
                     : "v"(rs), "v"(rs ^ 32u), "n"(W_FX * 128), "n"(256 + W_FX * 128), "v"(tcrd));
#define DFFW_WINO_TITEM(IT, BLK, CG)                                                                                                   \
        {                                                                                                                              \
            const short8 av = __builtin_bit_cast(short8, (u32x4w){ta[IT][0].x, ta[IT][0].y, ta[IT][1].x, ta[IT][1].y});                \
            const f32x4 d = mma<false>(av, tconst, f32x4{0.f, 0.f, 0.f, 0.f});                                                         \
            uint32_t h0, l0, h1, l1;                                                                                                   \
            Fmt<PREC>::split2(d[0], d[1], h0, l0);                                                                                     \
            Fmt<PREC>::split2(d[2], d[3], h1, l1);                                                                                     \
            const u32x2w vh = {h0, h1}, vl = {l0, l1};                                                                                 \
            asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4"                                             \
                         ::"v"((CG) ? ws + tcg : ws), "v"(vh), "v"(vl), "n"((BLK) * W_PITCH), "n"((BLK) * W_PITCH + W_VPLANE));         \
        }
        DFFW_WINO_TITEM(0, 0, 0)
        DFFW_WINO_TITEM(1, 0, 1)
        DFFW_WINO_TITEM(2, 1, 0)
        DFFW_WINO_TITEM(3, 1, 1)
#undef DFFW_WINO_TITEM
    };

#define DFFW_WINO_M3(E, M0, M1, M2)                                                                                                    \
    asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"                      \
                 : "=&v"(M0), "=&v"(M1), "=&v"(M2)                                                                                      \
                 : "v"(mrd), "n"(((E) * 4 + 0) * W_MPOS), "n"(((E) * 4 + 1) * W_MPOS), "n"(((E) * 4 + 2) * W_MPOS))

    // O of output slice zo (its M values are in the hand-over buffer); rh / rl: the thread's residual piece (loaded by the caller)
    auto output = [&](int zo) __attribute__((always_inline)) {
        f32x4 y;
        const uint32_t off = ooff0 + (uint32_t)zo * oslice;
        u32x2w rh = {0, 0}, rl = {0, 0};
        if constexpr (RES) {   // (requested here, used after the output transform: the slice fill is older and long done, so the wait is for these two only)
            rh = *reinterpret_cast<const u32x2w *>(a.res0 + off);
            rl = *reinterpret_cast<const u32x2w *>(a.res0 + off + C);
        }
        asm volatile("ds_read_b128 %0, %1" : "=&v"(y) : "v"(brd));   // (arrives with the first hand-over values below)
#define DFFW_WINO_OE(E, SG)                                                                                                            \
        {                                                                                                                              \
            f32x4 m0, m1, m2;                                                                                                          \
            DFFW_WINO_M3(E, m0, m1, m2);                                                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(y));                                               \
            y += (SG) * (m0 + sgc * (m1 + m2));                                                                                        \
        }
        DFFW_WINO_OE(0, 1.f)
        DFFW_WINO_OE(1, sgr)
        DFFW_WINO_OE(2, sgr)
#undef DFFW_WINO_OE
        if constexpr (RES) {
            float r0, r1;
            Fmt<PREC>::join2(rh.x, rl.x, r0, r1); y[0] += r0; y[1] += r1;
            Fmt<PREC>::join2(rh.y, rl.y, r0, r1); y[2] += r0; y[3] += r1;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = fmaxf(y[c], rfloor);
        uint2 vh, vl;
        Fmt<PREC>::split2(y[0], y[1], vh.x, vl.x);
        Fmt<PREC>::split2(y[2], y[3], vh.y, vl.y);
        *reinterpret_cast<uint2 *>(a.out + off) = vh;
        *reinterpret_cast<uint2 *>(a.out + off + C) = vl;
    };

    // ---- prologue: slices 0 and 1 requested, slice 0 transformed -------------------------------------------------------
    if (tid < 32) *reinterpret_cast<float *>(smem + W_BIASOFF + tid * 4) = a.bias[slab * 32 + tid];
    if (tid < 64) *reinterpret_cast<short8 *>(smem + W_TCOFF + tid * 16) = tconst0;   // (kept in LDS, not in 4 VGPRs: the kernel sits at the 256-register limit)
    if (N > 0) fill(0);
    if (N > 1) fill(1);
    // (the builtin, not asm: hipcc's wait-count pass must see that the filter fragments have arrived, or it waits vmcnt(0) -- i.e. for the
    // slice in flight -- in front of every step's first MFMA)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    asm volatile("s_barrier" ::: "memory");
    if (N > 0) transform(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    StepTrace tr((blockIdx.x < 512 && blockIdx.y == 0) ? a.trace : nullptr, wave, lane, 8);
    tr.no_skip();
    // one step; ROT = s % 3 fixes which accumulator set plays which role: filter slice dz accumulates into P[(dz - ROT) mod 3]
    // (dz = 0: output slice s+1, started here; dz = 1: s; dz = 2: s-1, finished here).  FULL: 2 <= s <= N-3, every part runs.
    auto step = [&](auto rot, auto full, int s) __attribute__((always_inline)) {
        constexpr int ROT = decltype(rot)::value;
        constexpr bool FULL = decltype(full)::value;
        tr.stamp(0);
        const bool do_o = FULL || (s >= 2), do_t = FULL || (s + 1 < N), do_m = FULL || (s < N);
        if (FULL || s + 2 < N) fill(s + 2);   // into the slot T(s) read one step ago; lands during this step
        auto contract = [&]() __attribute__((always_inline)) {
            const unsigned vr = vrd + (s & 1) * W_VBUF;
#define DFFW_WINO_POS(PP)                                                                                                              \
            {                                                                                                                          \
                short8 xh, xl;                                                                                                         \
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"                 \
                             : "=&v"(xh), "=&v"(xl)                                                                                    \
                             : "v"(vr), "n"((PP) * W_VPOS), "n"((PP) * W_VPOS + W_VPLANE));                                            \
                _Pragma("unroll") for (int dz = 0; dz < 3; ++dz) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                    \
                    const int k = (dz - ROT + 3) % 3;                                                                                  \
                    f32x4 c = dz == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : P[k][PP][nt];                                                      \
                    c = mma<false>(U[PP][dz][nt][1], xh, c);                                                                           \
                    c = mma<false>(U[PP][dz][nt][0], xl, c);                                                                           \
                    c = mma<false>(U[PP][dz][nt][0], xh, c);                                                                           \
                    P[k][PP][nt] = c;                                                                                                  \
                }                                                                                                                      \
            }
            DFFW_WINO_POS(0)
            DFFW_WINO_POS(1)
#undef DFFW_WINO_POS
        };
        if constexpr (FULL) {
            // every LDS read of the step is requested up front (they return in order): the MFMAs start when the four fragments are in,
            // and the transform's and the output's operands arrive behind them
            const unsigned vr = vrd + (s & 1) * W_VBUF;
            const unsigned rs = trd + ((s + 1) & 1) * W_RAWSLOT;
            short8 xh[2], xl[2];
            u32x2w ta[4][2];
            short8 tconst;
            f32x4 y, m[2][3];
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7"
                         : "=&v"(xh[0]), "=&v"(xl[0]), "=&v"(xh[1]), "=&v"(xl[1])
                         : "v"(vr), "n"(W_VPLANE), "n"(W_VPOS), "n"(W_VPOS + W_VPLANE));
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
                         "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:%10\n\t"
                         "ds_read_b64_tr_b16 %4, %9 offset:256\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
                         "ds_read_b64_tr_b16 %6, %8 offset:256\n\tds_read_b64_tr_b16 %7, %8 offset:%11"
                         : "=&v"(ta[0][0]), "=&v"(ta[0][1]), "=&v"(ta[1][0]), "=&v"(ta[1][1]), "=&v"(ta[2][0]), "=&v"(ta[2][1]), "=&v"(ta[3][0]), "=&v"(ta[3][1])
                         : "v"(rs), "v"(rs ^ 32u), "n"(W_FX * 128), "n"(256 + W_FX * 128));
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]));
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int k = (dz - ROT + 3) % 3;
                        f32x4 c = dz == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : P[k][pp][nt];
                        c = mma<false>(U[pp][dz][nt][1], xh[pp], c);
                        c = mma<false>(U[pp][dz][nt][0], xl[pp], c);
                        c = mma<false>(U[pp][dz][nt][0], xh[pp], c);
                        P[k][pp][nt] = c;
                    }
            // the residual of the pixel O finishes below: requested behind the contraction (its registers are free now), used ~1000 cycles on
            u32x2w rh = {0, 0}, rl = {0, 0};
            const uint32_t off = ooff0 + (uint32_t)(s - 2) * oslice;
            if constexpr (RES) {
                rh = *reinterpret_cast<const u32x2w *>(a.res0 + off);
                rl = *reinterpret_cast<const u32x2w *>(a.res0 + off + C);
            }
            tr.stamp(1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0][0]), "+v"(ta[0][1]), "+v"(ta[1][0]), "+v"(ta[1][1]), "+v"(ta[2][0]), "+v"(ta[2][1]), "+v"(ta[3][0]), "+v"(ta[3][1]));
            // (the LGKM counter has 4 bits: never more than 15 requests in flight -- the hand-over reads go out only now, and land behind the
            // transform; in front of them the transform's constant operand, which is needed at once)
            asm volatile("ds_read_b128 %0, %1" : "=&v"(tconst) : "v"(tcrd));
            asm volatile("ds_read_b128 %0, %1" : "=&v"(y) : "v"(brd));
            DFFW_WINO_M3(0, m[0][0], m[0][1], m[0][2]);
            DFFW_WINO_M3(1, m[1][0], m[1][1], m[1][2]);
            asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(tconst));
            const unsigned ws = twr + ((s + 1) & 1) * W_VBUF;
            u32x2w tvh[4], tvl[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const short8 av = __builtin_bit_cast(short8, (u32x4w){ta[it][0].x, ta[it][0].y, ta[it][1].x, ta[it][1].y});
                const f32x4 d = mma<false>(av, tconst, f32x4{0.f, 0.f, 0.f, 0.f});
                uint32_t h0, l0, h1, l1;
                Fmt<PREC>::split2(d[0], d[1], h0, l0);
                Fmt<PREC>::split2(d[2], d[3], h1, l1);
                tvh[it] = u32x2w{h0, h1};
                tvl[it] = u32x2w{l0, l1};
            }
            tr.stamp(2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y), "+v"(m[0][0]), "+v"(m[0][1]), "+v"(m[0][2]), "+v"(m[1][0]), "+v"(m[1][1]), "+v"(m[1][2]));
            asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %0, %3 offset:%10\n\tds_write_b64 %1, %4\n\tds_write_b64 %1, %5 offset:%10\n\t"
                         "ds_write_b64 %0, %6 offset:%11\n\tds_write_b64 %0, %7 offset:%12\n\tds_write_b64 %1, %8 offset:%11\n\tds_write_b64 %1, %9 offset:%12"
                         ::"v"(ws), "v"(ws + tcg), "v"(tvh[0]), "v"(tvl[0]), "v"(tvh[1]), "v"(tvl[1]), "v"(tvh[2]), "v"(tvl[2]), "v"(tvh[3]), "v"(tvl[3]),
                           "n"(W_VPLANE), "n"(W_PITCH), "n"(W_PITCH + W_VPLANE));
            // (the third row of hand-over values goes into the first row's registers once that row is summed: 24 registers instead of 36)
            y += m[0][0] + sgc * (m[0][1] + m[0][2]);
            DFFW_WINO_M3(2, m[0][0], m[0][1], m[0][2]);
            f32x4 yr = m[1][0] + sgc * (m[1][1] + m[1][2]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m[0][0]), "+v"(m[0][1]), "+v"(m[0][2]));
            yr += m[0][0] + sgc * (m[0][1] + m[0][2]);
            y += sgr * yr;
            if constexpr (RES) {
                float r0, r1;
                Fmt<PREC>::join2(rh.x, rl.x, r0, r1); y[0] += r0; y[1] += r1;
                Fmt<PREC>::join2(rh.y, rl.y, r0, r1); y[2] += r0; y[3] += r1;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = fmaxf(y[c], rfloor);
            uint2 vh, vl;
            Fmt<PREC>::split2(y[0], y[1], vh.x, vl.x);
            Fmt<PREC>::split2(y[2], y[3], vh.y, vl.y);
            *reinterpret_cast<uint2 *>(a.out + off) = vh;
            *reinterpret_cast<uint2 *>(a.out + off + C) = vl;
        } else {
            if (do_m) contract();
            tr.stamp(1);
            if (do_t) transform(s + 1);
            tr.stamp(2);
            if (do_o) output(s - 2);
        }
        tr.stamp(3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every thread has read its hand-over values
        if (s >= 1 && (FULL || s <= N)) {
            constexpr int k2 = (2 - ROT + 3) % 3;
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:64\n\tds_write_b128 %0, %3 offset:%5\n\tds_write_b128 %0, %4 offset:%6"
                         ::"v"(mwr), "v"(P[k2][0][0]), "v"(P[k2][0][1]), "v"(P[k2][1][0]), "v"(P[k2][1][1]), "n"(W_MPOS), "n"(W_MPOS + 64));
        }
        tr.stamp(4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slice s+2 has landed, V[(s+1) & 1] and the hand-over are written
        tr.stamp(5);
        tr.next();
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    auto partial = [&](int s) __attribute__((always_inline)) {   // the column's first two and last four steps: parts switched off at run time
        switch (s % 3) {
            case 0: step(R0{}, std::false_type{}, s); break;
            case 1: step(R1{}, std::false_type{}, s); break;
            default: step(R2{}, std::false_type{}, s); break;
        }
    };
    for (int s = 0; s <= N + 1;) {
        if (s >= 2 && s + 4 < N && s % 3 == 2) {   // three full steps (2 <= s and s + 2 + 2 < N), straight-line
            step(R2{}, std::true_type{}, s);
            step(R0{}, std::true_type{}, s + 1);
            step(R1{}, std::true_type{}, s + 2);
            s += 3;
        } else {
            partial(s);
            ++s;
        }
    }
#undef DFFW_WINO_M3
}

hipError_t launch_conv_wino32(int prec, const ConvArgs &a, const WinoArgs &t, hipStream_t s) {
    if ((int64_t)a.B * a.Ni * a.Hi * a.Wi * 128 >= (int64_t)1 << 32 || (int64_t)a.B * a.Ni * a.Hi * a.Wi * 2 * a.Cout >= (int64_t)1 << 31) return hipErrorInvalidValue;
    if (prec != P_BF16X3 || a.C0 != 32 || a.C1 != 0 || a.Cout % 32 || a.Hi % WINO_TY || a.Wi % WINO_TX || !a.out) return hipErrorInvalidValue;
    auto k = a.res0 ? conv_wino32<P_BF16X3, true> : conv_wino32<P_BF16X3, false>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[a.res0 ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        if (e != hipSuccess) return e;
        attr_done[a.res0 ? 1 : 0] = true;
    }
    const dim3 grid((unsigned)(a.B * t.tiles_y * t.tiles_x), (unsigned)(a.Cout / 32));
    hipLaunchKernelGGL(k, grid, dim3(512), W_LDS, s, a, t);
    return hipGetLastError();
}

void conv_wino32_kernel_name(int prec, const ConvArgs &a, char *buf, int n) { snprintf(buf, n, "dffw::conv_wino32<%d, %s>", prec, a.res0 ? "true" : "false"); }

}  // namespace dffw
