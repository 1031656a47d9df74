// How much do concurrent LDS operand reads cost the matrix pipe?  Per loop iteration a wave issues 12 v_mfma_f32_16x16x32_bf16 (4
// accumulators x 3, as a split-bf16 tile loop) and R ds_read_b128 whose results feed the NEXT iteration's MFMA operands (so they are
// real operand traffic: R = 0 .. 12; the conv kernels read 2 fragments per 3 MFMAs at one 16-channel output tile per operand tile,
// 2 per 6 at two).  Also the 32x32x16 shape with the same operand bytes per MAC halved.  3 / 2 / 1 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_lds_probe.hip -o tools/mfma_lds_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int R, int SHAPE>
__global__ __launch_bounds__(256) void probe(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + (threadIdx.x & 63) * 16;
    short8 x[12];
    for (int k = 0; k < 12; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[k]) : "v"(base), "n"(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
            short8 y[12];
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k < R) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y[k]) : "v"(base + ((it & 7) << 10)), "n"(1024 * 0));
#pragma unroll
            for (int k = 0; k < 12; ++k)
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x[k]), __builtin_bit_cast(bf16x8, x[(k + 1) % 12]), acc[k & 3], 0, 0, 0);
            if (R > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k < R) x[k] = y[k];
        }
        for (int k = 0; k < 4; ++k) s += acc[k][0];
    } else {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it) {
            short8 y[12];
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k < R) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y[k]) : "v"(base + ((it & 7) << 10)), "n"(0));
#pragma unroll
            for (int k = 0; k < 6; ++k)
                acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x[k]), __builtin_bit_cast(bf16x8, x[(k + 1) % 12]), acc[k & 1], 0, 0, 0);
            if (R > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k < R) x[k] = y[k];
        }
        for (int k = 0; k < 2; ++k) s += acc[k][0];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int R, int SHAPE>
void run(int wps, float *out) {
    const int iters = 3000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<R, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, 50);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<R, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)blocks * 4 * iters * (SHAPE == 16 ? 12 * 16384.0 : 6 * 32768.0);
    printf("shape %2d  waves/SIMD %d  ds_read_b128 per 12 (6) MFMAs %2d : %7.1f TFLOP/s   LDS read %6.1f B/clk/CU-equivalent at 2.3 GHz\n", SHAPE, wps, R,
           flop / (best * 1e-3) / 1e12, (double)blocks * 4 * iters * R * 1024.0 / (best * 1e-3) / 256 / 2.3e9);
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    for (int w = 3; w >= 1; --w) {
        run<0, 16>(w, out);
        run<2, 16>(w, out);
        run<4, 16>(w, out);
        run<8, 16>(w, out);
        run<12, 16>(w, out);
        run<0, 32>(w, out);
        run<4, 32>(w, out);
        run<8, 32>(w, out);
    }
    return 0;
}
