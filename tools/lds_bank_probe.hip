// Which lanes of a wave share an LDS service pass?  (gfx950; build: hipcc --offload-arch=gfx950 -O2 tools/lds_bank_probe.hip -o lds_bank_probe)
// Lanes 0 and j are the only active lanes and read 16 (or 8) bytes each, either from the same banks at different addresses
// (collide) or from different banks; the cycle difference per read tells whether the two lanes are served in the same pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BYTES>
__global__ void probe(unsigned long long *out, int j, int collide, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[32768];
    const int lane = threadIdx.x;
    for (int i = lane; i < 32768 / 4; i += 64) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // lane 0 reads offset 0; lane j reads 4096 (same banks, other address) or 4096 + 512 + BYTES (other banks)
    const unsigned ad = base + (lane == 0 ? 0u : (collide ? 4096u : 4096u + 128u));
    unsigned acc = 0;
    if (lane == 0 || lane == j) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if constexpr (BYTES == 16) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad));
                acc += v[0];
            } else {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                u2 v;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad));
                acc += v[0];
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[0] = t1 - t0;
    }
    if (acc == 0xdeadbeef) out[1] = acc;
}

// full-wave patterns: every lane reads 16 bytes at slot(lane) * 16; returns cycles per read
__global__ void pattern(unsigned long long *out, const int *slot, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
    const int lane = threadIdx.x;
    for (int i = lane; i < 65536 / 4; i += 64) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned ad = base + slot[lane] * 16;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad));
        acc += v[0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    if (acc == 0xdeadbeef) out[1] = acc;
}

int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 20000;
    for (int bytes : {16, 8}) {
        printf("ds_read_b%d: lanes served in the same pass as lane 0 (extra cycles per read when colliding):\n", bytes * 8);
        for (int j = 1; j < 64; ++j) {
            double t[2];
            for (int c = 0; c < 2; ++c) {
                if (bytes == 16) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, d, j, c, iters);
                else hipLaunchKernelGGL(probe<8>, dim3(1), dim3(64), 0, 0, d, j, c, iters);
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                t[c] = (double)h[0] / iters;
            }
            if (t[1] - t[0] > 0.5) printf(" %d(+%.1f)", j, t[1] - t[0]);
        }
        printf("\n");
    }
    // named full-wave patterns (slots of 16 bytes)
    struct P { const char *name; std::vector<int> s; };
    std::vector<P> ps;
    auto mk = [&](const char *n, auto f) { P p; p.name = n; for (int l = 0; l < 64; ++l) p.s.push_back(f(l)); ps.push_back(p); };
    mk("linear: slot = lane", [](int l) { return l; });
    mk("same-g rows 18 apart (srd stage B, t pitch 18): slot = 18*(r/8) + r%8 + 9*(g&1) + (g>>1)", [](int l) { int r = l & 15, g = l >> 4; return 18 * (r / 8) + r % 8 + 9 * (g & 1) + (g >> 1); });
    mk("t pitch 24, halves 12 apart", [](int l) { int r = l & 15, g = l >> 4; return 24 * (r / 8) + r % 8 + 12 * (g & 1) + (g >> 1); });
    mk("srd stage A (x pitch 20, halves 10 apart, 9 pairs per row)", [](int l) { int r = l & 15, g = l >> 4; return 20 * (r / 9) + r % 9 + 10 * (g & 1) + (g >> 1); });
    mk("x halves 120 apart, row pitch 10", [](int l) { int r = l & 15, g = l >> 4; return 10 * (r / 9) + r % 9 + 120 * (g & 1) + (g >> 1); });
    mk("conv_roll pair natural: 2*(18*(r/8) + 9*(g>>1) + r%8) + (g&1)", [](int l) { int r = l & 15, g = l >> 4; return 2 * (18 * (r / 8) + 9 * (g >> 1) + r % 8) + (g & 1); });
    mk("conv_roll pair enumerated order", [](int l) { int r = l & 15, g = l >> 4; int rr = r & 1, pp = ((r >> 2) & 3) | (((r >> 1) & 1) << 2); return 2 * (18 * rr + 9 * (g >> 1) + pp) + (g & 1); });
    mk("feat ring, two slices 256 slots apart: slot = 16*(r/8) + r%8 + 8*(g>>1) + 256*(g&1)", [](int l) { int r = l & 15, g = l >> 4; return 16 * (r / 8) + r % 8 + 8 * (g >> 1) + 256 * (g & 1); });
    mk("feat ring with parity swizzle: slot = 16*(r/8) + r%8 + 8*((g>>1)^(g&1)) + 256*(g&1)", [](int l) { int r = l & 15, g = l >> 4; return 16 * (r / 8) + r % 8 + 8 * ((g >> 1) ^ (g & 1)) + 256 * (g & 1); });
    int *ds;
    hipMalloc(&ds, 256);
    for (auto &p : ps) {
        hipMemcpy(ds, p.s.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(pattern, dim3(1), dim3(64), 0, 0, d, ds, iters);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%6.1f cycles/read  %s\n", (double)h[0] / iters, p.name);
    }
    return 0;
}
