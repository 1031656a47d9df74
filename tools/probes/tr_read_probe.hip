// ds_read_b64_tr_b16 semantics probe: each lane supplies the address of one 8-byte piece (4 x 16-bit); prints which LDS elements each
// lane receives.  hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe && ./tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__global__ void probe(uint16_t *out) {
    __shared__ uint16_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, gq = l >> 4, i = l & 15;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)lds;
    const unsigned addr = base + gq * 4096 + (i >> 2) * 512 + (i & 3) * 8;   // row (i >> 2) of a 4-row matrix with a 512-byte row stride, chunk i & 3
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr));
    out[l * 4 + 0] = v.x & 0xffff;
    out[l * 4 + 1] = v.x >> 16;
    out[l * 4 + 2] = v.y & 0xffff;
    out[l * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int expect = (l >> 4) * 2048 + j * 256 + (l & 15);
            printf(" %5d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "!");
            ok &= h[l * 4 + j] == expect;
        }
        printf("\n");
    }
    printf("hypothesis (lane n of a 16-lane group, element j = row j, column n of the 4 x 16 block whose 8-byte pieces the group's lanes address row-major) %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
