// Device-side helpers shared by the gfx950 kernels: 16-bit storage formats and the MFMA wrapper.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "dffw_internal.h"

namespace dffw {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- for bodies whose asm immediates, wait counts
// or scheduling hints must be constants of the iteration
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- 16-bit number formats ---------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even (finite inputs)
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float h2f(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }

template <int PREC>
struct Fmt {
    static constexpr int PARTS = (PREC == P_BF16X3) ? 2 : 1;
    // value of channel c of a pixel whose storage starts at p (layout [part][C])
    static __device__ __forceinline__ float load(const uint16_t *p, int C, int c) {
        if constexpr (PREC == P_BF16X3) return bf2f(p[c]) + bf2f(p[C + c]);
        else if constexpr (PREC == P_FP16) return h2f(p[c]);
        else return bf2f(p[c]);
    }
    static __device__ __forceinline__ void split(float v, uint16_t &hi, uint16_t &lo) {
        if constexpr (PREC == P_BF16X3) {
            hi = f2bf(v);
            lo = f2bf(v - bf2f(hi));
        } else if constexpr (PREC == P_FP16) {
            hi = f2h(v);
            lo = 0;
        } else {
            hi = f2bf(v);
            lo = 0;
        }
    }
    static __device__ __forceinline__ float join(uint16_t hi, uint16_t lo) {
        if constexpr (PREC == P_BF16X3) return bf2f(hi) + bf2f(lo);
        else if constexpr (PREC == P_FP16) return h2f(hi);
        else return bf2f(hi);
    }
    // two values at once: packed 16-bit pairs (x in the low half) of the hi and lo parts.  bf16 uses the
    // hardware round-to-nearest-even pack (v_cvt_pk_bf16_f32): 5 instructions per pair instead of ~20.
    static __device__ __forceinline__ void split2(float x, float y, uint32_t &hi, uint32_t &lo) {
        typedef __attribute__((ext_vector_type(2))) float f2;
        if constexpr (PREC == P_FP16) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x, y}, h2));
            lo = 0;
        } else {
            typedef __attribute__((ext_vector_type(2))) __bf16 b2;
            hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x, y}, b2));
            lo = 0;
            if constexpr (PREC == P_BF16X3) {
                const float rx = x - __uint_as_float(hi << 16), ry = y - __uint_as_float(hi & 0xFFFF0000u);
                lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{rx, ry}, b2));
            }
        }
    }
    static __device__ __forceinline__ void join2(uint32_t hi, uint32_t lo, float &x, float &y) {
        if constexpr (PREC == P_FP16) {
            x = h2f((uint16_t)(hi & 0xFFFF));
            y = h2f((uint16_t)(hi >> 16));
        } else {
            x = __uint_as_float(hi << 16);
            y = __uint_as_float(hi & 0xFFFF0000u);
            if constexpr (PREC == P_BF16X3) {
                x += __uint_as_float(lo << 16);
                y += __uint_as_float(lo & 0xFFFF0000u);
            }
        }
    }
    static __device__ __forceinline__ void store(uint16_t *p, int C, int c, float v) {
        uint16_t hi, lo;
        split(v, hi, lo);
        p[c] = hi;
        if constexpr (PARTS == 2) p[C + c] = lo;
    }
};

template <bool F16>
__device__ __forceinline__ f32x4 mma(short8 a, short8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- step timeline of the persistent streaming kernels (make TRACE=1; DFFW_TRACE_LAYER / DFFW_TRACE_OUT, tools/trace_steps.py) ----
// Lane 0 of every wave stamps s_memtime at up to 7 points of DFFW_TRACE_STEPS consecutive steps (after skipping the first
// DFFW_TRACE_SKIP, i.e. the cold ring of the workgroup's first column); slot 7 of step 0 = XCC_ID << 32 | HW_ID.
// Buffer: [workgroup][wave][step][8] u64.  Compiled out of production builds.
#define DFFW_TRACE_STEPS 32
#define DFFW_TRACE_SKIP 14
#define DFFW_TRACE_MAX_WGS 1024
struct StepTrace {
#ifdef DFFW_TRACE_BUILD
    unsigned long long *p;
    int n;
    __device__ __forceinline__ StepTrace(unsigned long long *base, int wave, int lane, int nwaves) {
        // (the engine sizes the buffer for DFFW_TRACE_MAX_WGS workgroups: a test grid beyond that records nothing instead of writing past it)
        p = (base && lane == 0 && blockIdx.x < DFFW_TRACE_MAX_WGS) ? base + ((int64_t)blockIdx.x * nwaves + wave) * (DFFW_TRACE_STEPS * 8) : nullptr;
        n = -DFFW_TRACE_SKIP;
    }
    __device__ __forceinline__ void no_skip() { n = 0; }   // short streams: record from the first step
    __device__ __forceinline__ void stamp(int k) {
        if (p && n >= 0 && n < DFFW_TRACE_STEPS) p[n * 8 + k] = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void next() {
        if (p && n == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p[7] = ((unsigned long long)xcc << 32) | hwid;
        }
        ++n;
    }
#else
    __device__ __forceinline__ StepTrace(unsigned long long *, int, int, int) {}
    __device__ __forceinline__ void no_skip() {}
    __device__ __forceinline__ void stamp(int) {}
    __device__ __forceinline__ void next() {}
#endif
};

// ---- conv epilogue shared by conv_igemm and conv_tile ---------------------------------------------------
// One call handles, for every lane of the wave, the 4 output channels c0 = nt*16 + g*4 .. +3 of the
// lane's grid point (the v_mfma_f32_16x16x32 result layout): BatchNorm shift, optional copy of the
// pre-residual value, up to two residual adds, ReLU, optional 1x1x1 classifier partial dot, store.
// MUST be called by all 64 lanes (no divergence around it): in split-bf16 storage the hi and lo
// halves of lane rows g and g^1 are exchanged with v_permlane16_swap so that every lane moves one
// 16-byte piece (8 channels of one half) per access instead of two 8-byte pieces, i.e. a wave store
// covers 16 pixels x 64 contiguous bytes.  `pvalid`: the lane's grid point exists.
// relu as a signed-integer max on the bit pattern: one instruction, no float canonicalisation in front
__device__ __forceinline__ float relu_bits(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
// ReLU and a position mask in one v_med3_f32: lim = +inf keeps relu(v), lim = 0 gives 0 (t outside the image = the next conv's padding).
// (Not inline asm on the accumulator: hipcc does not see an asm blob's reads, so the MFMA -> VALU wait states would be missing.)
__device__ __forceinline__ float relu_lim_bits(float v, int lim) { return __builtin_amdgcn_fmed3f(v, 0.f, __int_as_float(lim)); }

__device__ __forceinline__ void swap16(uint32_t &a, uint32_t &b) {
    // a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3} (rows = 16-lane groups)
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// raw residual piece of one lane (16 bytes in split-bf16 storage, 8 bytes otherwise); zero when absent
template <int PREC>
__device__ __forceinline__ uint4 epilogue_res_load(const ConvArgs &a, const uint16_t *base, int nt, int g, int64_t opix, bool pvalid) {
    uint4 q = make_uint4(0, 0, 0, 0);
    if (!base) return q;
    const int Cout = a.Cout;
    if constexpr (Fmt<PREC>::PARTS == 2) {
        const int oct = nt * 2 + (g >> 1);
        if (pvalid && oct * 8 < Cout) q = *reinterpret_cast<const uint4 *>(base + opix * (2 * Cout) + (g & 1) * Cout + oct * 8);
    } else {
        const int c0 = nt * 16 + g * 4;
        if (pvalid && c0 < Cout) {
            const uint2 h = *reinterpret_cast<const uint2 *>(base + opix * Cout + c0);
            q.x = h.x;
            q.y = h.y;
        }
    }
    return q;
}

// PRE: the residual pieces were fetched earlier with epilogue_res_load (pre0/pre1), else they are loaded here.
// FAST (conv_tile): the BatchNorm shift is already in the accumulator (it was initialised with it) and the
// lane's element offset inside the output volume is `voff` (32-bit, precomputed once per kernel: pixel
// offset inside the tile * record size + this lane's 16-byte piece) relative to the wave-uniform element
// offset `ubase` of the tile, so a store costs no per-lane 64-bit address arithmetic.
template <int PREC, bool PRE, bool FAST = false, bool ADD_BIAS = !FAST>
__device__ __forceinline__ void epilogue_quad(const ConvArgs &a, const f32x4 &accq, int nt, int g, int64_t opix, bool pvalid,
                                              float &cls_partial, uint4 pre0, uint4 pre1, int64_t ubase = 0, int voff = 0,
                                              int64_t rbase = 0, int rvoff = 0) {   // FAST + a.res_bcast: residual's own base/offset
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const int Cout = a.Cout;
    const int c0 = nt * 16 + g * 4;
    const bool cvalid = c0 < Cout;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ADD_BIAS ? accq[i] + a.bias[c0 + i] : accq[i];   // bias is zero-padded to the kernel's NT*16 channels
    if (a.outf) {  // fp32 planar output: 1-channel score volume (B,No,Ho,Wo), or the <= 4 channels of an alignment head
        if (pvalid && c0 == 0) {
            if (a.outf_ch <= 1) {
                a.outf[opix] = (a.relu == 1) ? fmaxf(v[0], 0.f) : v[0];
            } else {
                const int64_t b = opix / a.outf_plane, q = opix - b * a.outf_plane;
                for (int i = 0; i < a.outf_ch; ++i) a.outf[(b * a.outf_ch + i) * a.outf_plane + q] = v[i];
            }
        }
        return;
    }
    if constexpr (PARTS == 2) {
        const int oct = nt * 2 + (g >> 1);                 // which 8-channel group this lane moves
        const bool wvalid = pvalid && oct * 8 < Cout;
        const int64_t eo = FAST ? (int64_t)(voff + nt * 16) : opix * (2 * Cout) + (g & 1) * Cout + oct * 8;
        auto wide_store = [&](uint16_t *base_) {
            uint16_t *base = FAST ? base_ + ubase : base_;
            uint32_t h01, h23, l01, l23;
            Fmt<PREC>::split2(v[0], v[1], h01, l01);
            Fmt<PREC>::split2(v[2], v[3], h23, l23);
            swap16(h01, l01);
            swap16(h23, l23);
            if (wvalid) *reinterpret_cast<uint4 *>(base + eo) = make_uint4(h01, h23, l01, l23);
        };
        auto wide_add = [&](const uint16_t *base_, uint4 q) {
            const uint16_t *base = FAST ? base_ + (a.res_bcast ? rbase : ubase) : base_;
            int64_t ro = eo;
            if (a.res_bcast) {   // one residual slice per sample, shared by all output slices
                if constexpr (FAST) ro = (int64_t)(rvoff + nt * 16);
                else {
                    const int64_t hw = (int64_t)a.Ho * a.Wo;
                    ro = ((opix / a.outf_plane) * hw + opix % hw) * (2 * Cout) + (g & 1) * Cout + oct * 8;
                }
            }
            if constexpr (!PRE) {
                q = make_uint4(0, 0, 0, 0);
                if (wvalid) q = *reinterpret_cast<const uint4 *>(base + ro);
            }
            swap16(q.x, q.z);
            swap16(q.y, q.w);
            float r0, r1, r2, r3;
            Fmt<PREC>::join2(q.x, q.z, r0, r1);
            Fmt<PREC>::join2(q.y, q.w, r2, r3);
            v[0] += r0;
            v[1] += r1;
            v[2] += r2;
            v[3] += r3;
        };
        if (a.out_pre) wide_store(a.out_pre);
        if (a.relu == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_bits(v[i]);
        }
        if (a.res0) wide_add(a.res0, pre0);
        if (a.res1) wide_add(a.res1, pre1);
        if (a.relu == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_bits(v[i]);
        }
        if (a.cls_w && cvalid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cls_partial = fmaf(a.cls_w[c0 + i], v[i], cls_partial);
        }
        if (a.out) wide_store(a.out);
    } else {
        const bool ok = pvalid && cvalid;
        const int64_t eo = FAST ? (int64_t)(voff + nt * 16) : opix * Cout + c0;
        auto store4 = [&](uint16_t *base_) {
            uint16_t *base = FAST ? base_ + ubase : base_;
            uint32_t h01, h23, l01, l23;
            Fmt<PREC>::split2(v[0], v[1], h01, l01);
            Fmt<PREC>::split2(v[2], v[3], h23, l23);
            if (ok) *reinterpret_cast<uint2 *>(base + eo) = make_uint2(h01, h23);
        };
        auto add4 = [&](const uint16_t *base_, uint4 q) {
            const uint16_t *base = FAST ? base_ + (a.res_bcast ? rbase : ubase) : base_;
            int64_t ro = eo;
            if (a.res_bcast) {
                if constexpr (FAST) ro = (int64_t)(rvoff + nt * 16);
                else {
                    const int64_t hw = (int64_t)a.Ho * a.Wo;
                    ro = ((opix / a.outf_plane) * hw + opix % hw) * Cout + c0;
                }
            }
            if constexpr (!PRE) {
                q = make_uint4(0, 0, 0, 0);
                if (ok) {
                    const uint2 h = *reinterpret_cast<const uint2 *>(base + ro);
                    q.x = h.x;
                    q.y = h.y;
                }
            }
            float r0, r1, r2, r3;
            Fmt<PREC>::join2(q.x, 0, r0, r1);
            Fmt<PREC>::join2(q.y, 0, r2, r3);
            v[0] += r0;
            v[1] += r1;
            v[2] += r2;
            v[3] += r3;
        };
        if (a.out_pre) store4(a.out_pre);
        if (a.relu == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_bits(v[i]);
        }
        if (a.res0) add4(a.res0, pre0);
        if (a.res1) add4(a.res1, pre1);
        if (a.relu == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_bits(v[i]);
        }
        if (a.cls_w && cvalid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cls_partial = fmaf(a.cls_w[c0 + i], v[i], cls_partial);
        }
        if (a.out) store4(a.out);
    }
}

// the streaming kernels' 16-byte result store.  (Round 6, measured and not kept: as a NON-TEMPORAL store -- a wave's store covers 32-byte pieces of its pixels'
// 128-byte records, the other pieces follow from the same or a neighbouring wave, and it is the L2 that merges them: with nt the pieces go out one by one,
// conv_slice32 +31 %, conv_rollt +18 ... 37 %, the K-split kernels -2 %, the forward +1.7 %; profiles/r06_nt_stores.txt)
__device__ __forceinline__ void dffw_store16(char *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }

// Lean epilogue of the streaming kernels in split-bf16 storage: the same arithmetic and the same instruction sequence per value
// as epilogue_quad<PREC, PRE, FAST> (bit-identical results), but straight-line.  The generic routine tests nine ConvArgs fields
// per call; its ~25 taken branches, SGPR spills and AGPR round trips cost ~1100 cycles per operand tile in the rolling kernels
// (step timeline, profiles/r03_step_timeline.txt), a third of a step.  Covers
//     pre  = acc                      -> out_pre (if non-null)
//     v    = [relu](acc [+ residual]) -> out     (if non-null)
//     cls  = sum_i clsw[i] * v[i]     (CLS; returned, the caller finishes it with epilogue_cls)
// i.e. every epilogue of the network except relu-before-residual (relu == 2), a second residual, broadcast residuals and fp32
// planar outputs, which stay on epilogue_quad (host side: roll_lean(), dffw_conv_roll.hip, picks the LEAN instantiation).  `out` /
// `out_pre` = wave-uniform pointers of the step's first output element, `voff` = the lane's element offset (its 16-byte piece).
// MUST be called by all 64 lanes (v_permlane16_swap).
template <int PREC, bool RES, bool CLS = false>
__device__ __forceinline__ float epilogue_lean(uint16_t *__restrict__ out, uint16_t *__restrict__ out_pre, int voff, const f32x4 &acc, uint4 rq,
                                               bool relu, const f32x4 &clsw) {
    static_assert(Fmt<PREC>::PARTS == 2, "split-bf16 storage only");
    float v0 = acc[0], v1 = acc[1], v2 = acc[2], v3 = acc[3];
    auto store = [&](uint16_t *base) {
        uint32_t h01, h23, l01, l23;
        Fmt<PREC>::split2(v0, v1, h01, l01);
        Fmt<PREC>::split2(v2, v3, h23, l23);
        swap16(h01, l01);
        swap16(h23, l23);
        dffw_store16(reinterpret_cast<char *>(base) + (uint32_t)(voff * 2), make_uint4(h01, h23, l01, l23));
    };
    if (out_pre) store(out_pre);
    if constexpr (RES) {
        swap16(rq.x, rq.z);
        swap16(rq.y, rq.w);
        float r0, r1, r2, r3;
        Fmt<PREC>::join2(rq.x, rq.z, r0, r1);
        Fmt<PREC>::join2(rq.y, rq.w, r2, r3);
        v0 += r0;
        v1 += r1;
        v2 += r2;
        v3 += r3;
    }
    if (relu) {
        v0 = relu_bits(v0);
        v1 = relu_bits(v1);
        v2 = relu_bits(v2);
        v3 = relu_bits(v3);
    }
    float cls = 0.f;
    if constexpr (CLS) {
        cls = fmaf(clsw[0], v0, cls);
        cls = fmaf(clsw[1], v1, cls);
        cls = fmaf(clsw[2], v2, cls);
        cls = fmaf(clsw[3], v3, cls);
    }
    if (out) store(out);
    return cls;
}

// conv_tile's form of the lean epilogue: residual, classifier and the two stores are wave-uniform run-time options (one uniform
// branch each instead of a template parameter: conv_tile already has ~50 instantiations per arithmetic), `pv` predicates the
// stores of edge tiles, `cls` is the classifier partial carried across the 16-channel tiles of a pixel.  Same arithmetic and
// order as epilogue_quad (out_pre store, + residual, ReLU, classifier, out store); MUST be called by all 64 lanes.
template <int PREC>
__device__ __forceinline__ void epilogue_lean_t(uint16_t *__restrict__ out, uint16_t *__restrict__ out_pre, int voff, float v0, float v1, float v2,
                                                float v3, bool has_res, uint4 rq, bool relu, bool has_cls, const f32x4 &clsw, float &cls, bool pv) {
    static_assert(Fmt<PREC>::PARTS == 2, "split-bf16 storage only");
    auto store = [&](uint16_t *base) {
        uint32_t h01, h23, l01, l23;
        Fmt<PREC>::split2(v0, v1, h01, l01);
        Fmt<PREC>::split2(v2, v3, h23, l23);
        swap16(h01, l01);
        swap16(h23, l23);
        if (pv) dffw_store16(reinterpret_cast<char *>(base) + (uint32_t)(voff * 2), make_uint4(h01, h23, l01, l23));
    };
    if (out_pre) store(out_pre);
    if (has_res) {
        swap16(rq.x, rq.z);
        swap16(rq.y, rq.w);
        float r0, r1, r2, r3;
        Fmt<PREC>::join2(rq.x, rq.z, r0, r1);
        Fmt<PREC>::join2(rq.y, rq.w, r2, r3);
        v0 += r0;
        v1 += r1;
        v2 += r2;
        v3 += r3;
    }
    if (relu) {
        v0 = relu_bits(v0);
        v1 = relu_bits(v1);
        v2 = relu_bits(v2);
        v3 = relu_bits(v3);
    }
    if (has_cls) {
        cls = fmaf(clsw[0], v0, cls);
        cls = fmaf(clsw[1], v1, cls);
        cls = fmaf(clsw[2], v2, cls);
        cls = fmaf(clsw[3], v3, cls);
    }
    if (out) store(out);
}

// finish the fused 1x1x1 classifier: sum the partial dots of the 4 lane rows, row 0 writes the score
// `rows`: how many 16-lane rows hold channels of the SAME pixel (4 normally; 2 when two 8-channel operand
// tiles were packed into one register set, then rows 0-1 and 2-3 are different pixels)
__device__ __forceinline__ void epilogue_cls(const ConvArgs &a, float partial, int g, int64_t opix, bool pvalid, int rows = 4) {
    if (!a.cls_w) return;
    partial += __shfl_xor(partial, 16);
    if (rows == 4) partial += __shfl_xor(partial, 32);
    if ((g & (rows - 1)) == 0 && pvalid) a.cls_out[opix] = partial;
}

// ---- FOV warp of the alignment network (End_to_End.py:106-134), shared by fov_warp_kernel, flow_volume_kernel and conv_tile's
// warp-fill variant -------------------------------------------------------------------------------------------------------
// flow of grid point (xx,yy) of a slice with scale f = FOV + a0 and shifts a1, a2, and the un-normalised sample
// position (sx,sy) grid_sample(align_corners=True) derives from it
struct WarpPoint {
    float fx, fy, sx, sy;
};
__device__ __forceinline__ WarpPoint warp_point(int xx, int yy, int H, int W, float f, float a1, float a2) {
    const float stepx = 2.0f / (float)(W > 1 ? W - 1 : 1), stepy = 2.0f / (float)(H > 1 ? H - 1 : 1);
    // torch.linspace(-1, 1, steps): start + i*step in the first half, end - (steps-1-i)*step in the second
    const float lx = xx < W / 2 ? -1.0f + (float)xx * stepx : 1.0f - (float)(W - 1 - xx) * stepx;
    const float ly = yy < H / 2 ? -1.0f + (float)yy * stepy : 1.0f - (float)(H - 1 - yy) * stepy;
    WarpPoint p;
    p.fx = (float)(W / 2) * (f - 1.0f) * lx + a1;
    p.fy = (float)(H / 2) * (f - 1.0f) * ly + a2;
    // normalised grid, then grid_sample(align_corners=True): ((g + 1) / 2) * (size - 1)
    const float gx = 2.0f * ((float)xx - p.fx) / (float)(W > 1 ? W - 1 : 1) - 1.0f;
    const float gy = 2.0f * ((float)yy - p.fy) / (float)(H > 1 ? H - 1 : 1) - 1.0f;
    p.sx = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
    p.sy = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    return p;
}


// bilinear sample (zeros outside, grid_sample align_corners=True) of 8 consecutive channels of a channels-last slice at the
// warped position of `wp`; `slice` points at channel octet's first element of pixel (0,0).  v must be zero-initialised.
template <int PREC>
__device__ __forceinline__ void warp_octet(const uint16_t *__restrict__ slice, int C, int H, int W, const WarpPoint &wp, float (&v)[8]) {
    constexpr int PARTS = Fmt<PREC>::PARTS;
    const float x0f = floorf(wp.sx), y0f = floorf(wp.sy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = wp.sx - x0f, wy1 = wp.sy - y0f;
    const float wx[2] = {1.0f - wx1, wx1}, wy[2] = {1.0f - wy1, wy1};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int yc = y0 + dy;
        if (yc < 0 || yc >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xc = x0 + dx;
            if (xc < 0 || xc >= W) continue;
            const uint16_t *rec = slice + ((int64_t)yc * W + xc) * (PARTS * C);
            const uint4 h = *reinterpret_cast<const uint4 *>(rec);
            uint4 l = make_uint4(0, 0, 0, 0);
            if constexpr (PARTS == 2) l = *reinterpret_cast<const uint4 *>(rec + C);
            const float wgt = wx[dx] * wy[dy];
            float a, c;
            Fmt<PREC>::join2(h.x, l.x, a, c); v[0] += a * wgt; v[1] += c * wgt;
            Fmt<PREC>::join2(h.y, l.y, a, c); v[2] += a * wgt; v[3] += c * wgt;
            Fmt<PREC>::join2(h.z, l.z, a, c); v[4] += a * wgt; v[5] += c * wgt;
            Fmt<PREC>::join2(h.w, l.w, a, c); v[6] += a * wgt; v[7] += c * wgt;
        }
    }
}

}  // namespace dffw
