// Calibration of rocprofv3's FETCH_SIZE on this code base's access patterns (MI355X_MICROARCH.md: FETCH_SIZE reports half the
// bytes of a wide coalesced stream on gfx950; other widths are uncalibrated).  Each kernel reads a KNOWN number of bytes once
// from a buffer far larger than the 256 MiB Infinity Cache:
//   stream16      every lane 16 B, consecutive lanes consecutive addresses (the calibrated case: 16 B/lane coalesced)
//   piece16of64   every lane the first 16 B of its own 64-byte record (one 8-channel stage of a 16-channel split-bf16 pixel:
//                 what conv_tile's stride-2 stages fetch) -> useful bytes = 1/4 of the records' span
//   piece32of128  two lanes the first 32 B of a 128-byte record (a 16-channel stage of one part of a 32-channel pixel)
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib.out ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./tools/fetch_calib.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void stream16(const u32x4 *p, unsigned *out, long n) {
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = p[i];
        acc ^= v[0] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void piece16of64(const u32x4 *p, unsigned *out, long nrec) {
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = p[i * 4];
        acc ^= v[0] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void piece32of128(const u32x4 *p, unsigned *out, long nrec) {
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * nrec; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = p[(i >> 1) * 8 + (i & 1)];
        acc ^= v[0] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const long bytes = 2L << 30;   // 2 GiB
    char *buf;
    unsigned *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, 0, (const u32x4 *)buf, out, bytes / 16);
    hipLaunchKernelGGL(piece16of64, dim3(256 * 16), dim3(256), 0, 0, (const u32x4 *)buf, out, bytes / 64);
    hipLaunchKernelGGL(piece32of128, dim3(256 * 16), dim3(256), 0, 0, (const u32x4 *)buf, out, bytes / 128);
    hipDeviceSynchronize();
    printf("span %ld bytes: stream16 reads %ld, piece16of64 reads %ld useful bytes (%ld of 64-B sectors), piece32of128 reads %ld useful (%ld of sectors)\n",
           bytes, bytes, bytes / 4, bytes, bytes / 4, bytes / 2);
    return 0;
}
