// Sustained issue rate of the two bf16 MFMA shapes on all CUs: v_mfma_f32_16x16x32_bf16 (what the conv kernels use) against
// v_mfma_f32_32x32x16_bf16, for 1 / 2 / 3 waves per SIMD and ACC independent accumulators per wave, operands held in registers
// (no memory traffic at all).  build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o /tmp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int ACC>
__global__ void probe(float *out, int iters, unsigned seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        unsigned h = (threadIdx.x * 8 + i) * 2654435761u + seed;
        a[i] = (__bf16)(float)((h >> 8) & 255) * (__bf16)0.01f;
        b[i] = (__bf16)(float)((h >> 16) & 255) * (__bf16)0.01f;
    }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[ACC];
        for (int k = 0; k < ACC; ++k) acc[k] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < ACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
        }
        for (int k = 0; k < ACC; ++k) s += acc[k][0] + acc[k][3];
    } else {
        f32x16 acc[ACC];
        for (int k = 0; k < ACC; ++k)
            for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < ACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        }
        for (int k = 0; k < ACC; ++k) s += acc[k][0] + acc[k][15];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int SHAPE, int ACC>
void run(int waves_per_simd, float *out) {
    const int iters = 4000;
    const int threads = 256 * 1;            // 4 waves per block = one per SIMD
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, ACC>), dim3(blocks), dim3(threads), 0, 0, out, 100, 1u);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, ACC>), dim3(blocks), dim3(threads), 0, 0, out, iters, 7u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop_per = SHAPE == 16 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
    const double total = (double)blocks * 4 * iters * ACC * flop_per;
    const double tf = total / (best * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at a nominal 2.4 GHz, for reference only (the chip clocks to its power budget)
    const double mfma_per_simd = (double)waves_per_simd * iters * ACC;
    printf("shape %2d  acc %2d  waves/SIMD %d : %8.1f TFLOP/s  %.3f ms  (%.1f ns per MFMA per SIMD)\n", SHAPE, ACC, waves_per_simd, tf, best,
           best * 1e6 / mfma_per_simd);
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    // dependent chains: ACC = 1 is one accumulator fed back to back (what a 3-product split-bf16 tile loop does when a wave owns
    // ONE operand tile), 2 / 3 = that many independent chains
    for (int w = 1; w <= 3; ++w) {
        run<16, 1>(w, out);
        run<16, 2>(w, out);
        run<16, 3>(w, out);
    }
    for (int w = 1; w <= 3; ++w) {
        run<16, 4>(w, out);
        run<16, 10>(w, out);
        run<32, 2>(w, out);
        run<32, 5>(w, out);
    }
    return 0;
}
