/* Minimal C host of libdffw.so: no PyTorch, no C++ — HIP runtime + the C ABI of include/dffw.h only.
 *
 *   c_host <weights.bin> <input.bin> <output.bin> [e2e]
 *
 * weights.bin : int32 n, then n x { int32 name_len, name bytes, int64 numel, numel x float32 }
 * input.bin   : int32 B, N, H, W, then FS (B,3,N,H,W) float32, then focus_dists (B,N) float32 (broadcast over H,W);
 *               with `e2e`: then the relative fields of view (B,N) float32
 * output.bin  : 4 x (B,H,W) float32 = mid_out, pred1, pred2, pred3; with `e2e`: then the aligned stack (B,3,N,H,W)
 *
 * `e2e` = the End_to_End variant: a DFFW_NET_E2E engine (522-entry state dict) and dffw_forward_e2e in place of
 * `mid, p1, p2, p3, aligned = model(FS, focus_dists, FOVs)` (End_to_End/TRS.py:44).
 *
 * Build (tests/test_c_host.py does this):
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_host.c -I/opt/rocm/include -Iinclude -Ldffinthewild_amd -ldffw \
 *       -L/opt/rocm/lib -lamdhip64 -o c_host
 * This is the call sequence a non-Python deployment binds in place of the reference's
 * Network() / load_state_dict / model(FS, focus_dists) (Depth_Estimation_Test/test.py:30,78,118).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dffw.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_DFFW(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, dffw_last_error()); return 3; } } while (0)

static int read_all(FILE *f, void *dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char **argv) {
    if (argc != 4 && argc != 5) { fprintf(stderr, "usage: %s weights.bin input.bin output.bin [e2e]\n", argv[0]); return 1; }
    const int e2e = argc == 5;
    FILE *fw = fopen(argv[1], "rb");
    if (!fw) { perror(argv[1]); return 1; }
    int32_t n = 0;
    if (read_all(fw, &n, 4) || n <= 0) { fprintf(stderr, "bad weights file\n"); return 1; }
    dffw_tensor *tensors = (dffw_tensor *)calloc((size_t)n, sizeof(dffw_tensor));
    for (int i = 0; i < n; ++i) {
        int32_t len = 0;
        int64_t numel = 0;
        if (read_all(fw, &len, 4)) return 1;
        char *name = (char *)calloc((size_t)len + 1, 1);
        if (read_all(fw, name, (size_t)len) || read_all(fw, &numel, 8)) return 1;
        float *data = (float *)malloc((size_t)numel * sizeof(float));
        if (read_all(fw, data, (size_t)numel * sizeof(float))) return 1;
        tensors[i].name = name;
        tensors[i].data = data;
        tensors[i].numel = numel;
    }
    fclose(fw);

    FILE *fi = fopen(argv[2], "rb");
    if (!fi) { perror(argv[2]); return 1; }
    int32_t dims[4];
    if (read_all(fi, dims, sizeof dims)) return 1;
    const int B = dims[0], N = dims[1], H = dims[2], W = dims[3];
    const size_t fs_elems = (size_t)B * 3 * N * H * W, fd_elems = (size_t)B * N, map_elems = (size_t)B * H * W;
    float *h_fs = (float *)malloc(fs_elems * sizeof(float)), *h_fd = (float *)malloc(fd_elems * sizeof(float));
    float *h_fov = (float *)malloc(fd_elems * sizeof(float));
    if (read_all(fi, h_fs, fs_elems * sizeof(float)) || read_all(fi, h_fd, fd_elems * sizeof(float))) return 1;
    if (e2e && read_all(fi, h_fov, fd_elems * sizeof(float))) return 1;
    fclose(fi);

    dffw_engine *eng = NULL;
    CHECK_HIP(hipSetDevice(0));
    CHECK_DFFW(dffw_engine_create(0, e2e ? DFFW_NET_E2E : DFFW_NET_DEPTH, tensors, n, DFFW_PREC_BF16X3, &eng));
    const int64_t ws_bytes = dffw_workspace_bytes(eng, B, N, H, W);
    if (ws_bytes < 0) { fprintf(stderr, "workspace: %s\n", dffw_last_error()); return 3; }

    float *d_fs = NULL, *d_fd = NULL, *d_fov = NULL, *d_aligned = NULL, *d_out[4] = {NULL, NULL, NULL, NULL};
    void *d_ws = NULL;
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc((void **)&d_fs, fs_elems * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_fd, fd_elems * sizeof(float)));
    CHECK_HIP(hipMalloc(&d_ws, (size_t)ws_bytes));
    for (int k = 0; k < 4; ++k) CHECK_HIP(hipMalloc((void **)&d_out[k], map_elems * sizeof(float)));
    CHECK_HIP(hipMemcpyAsync(d_fs, h_fs, fs_elems * sizeof(float), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(d_fd, h_fd, fd_elems * sizeof(float), hipMemcpyHostToDevice, stream));

    const int64_t fd_strides[4] = {N, 1, 0, 0}; /* (B,N) values broadcast over rows and columns */
    if (e2e) {
        CHECK_HIP(hipMalloc((void **)&d_fov, fd_elems * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&d_aligned, fs_elems * sizeof(float)));
        CHECK_HIP(hipMemcpyAsync(d_fov, h_fov, fd_elems * sizeof(float), hipMemcpyHostToDevice, stream));
        CHECK_DFFW(dffw_forward_e2e(eng, d_fs, d_fd, fd_strides, d_fov, B, N, H, W, d_out, d_aligned, d_ws, ws_bytes, stream, NULL, 0));
    } else {
        CHECK_DFFW(dffw_forward(eng, d_fs, d_fd, fd_strides, B, N, H, W, d_out, d_ws, ws_bytes, stream));
    }
    CHECK_HIP(hipStreamSynchronize(stream));

    FILE *fo = fopen(argv[3], "wb");
    if (!fo) { perror(argv[3]); return 1; }
    float *h_map = (float *)malloc(map_elems * sizeof(float));
    for (int k = 0; k < 4; ++k) {
        CHECK_HIP(hipMemcpy(h_map, d_out[k], map_elems * sizeof(float), hipMemcpyDeviceToHost));
        fwrite(h_map, sizeof(float), map_elems, fo);
    }
    if (e2e) {
        float *h_al = (float *)malloc(fs_elems * sizeof(float));
        CHECK_HIP(hipMemcpy(h_al, d_aligned, fs_elems * sizeof(float), hipMemcpyDeviceToHost));
        fwrite(h_al, sizeof(float), fs_elems, fo);
    }
    fclose(fo);
    dffw_engine_destroy(eng);
    printf("%s: %d stacks of %dx%dx%d -> 4 depth maps%s, workspace %lld bytes\n", dffw_version(), B, N, H, W, e2e ? " + aligned stack" : "",
           (long long)ws_bytes);
    return 0;
}
