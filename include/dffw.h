/* dffw.h — C ABI of the MI355X (gfx950) depth-from-focus engine, libdffw.so.
 *
 * The reference (wcy199705/DfFintheWild) has no FFI/plugin layer: its boundary is the Python
 * nn.Module call `Network()(FS, focus_dists)` (Depth_Estimation_Test/test.py:30,118,
 * Depth_Estimation_Test/Depth_Estimation_Network.py:7-13) whose arithmetic runs inside PyTorch
 * operators.  This header is what a binding for that call binds instead: plain pointers and
 * sizes, no torch types.  dffinthewild_amd/engine.py is the ctypes stub.
 *
 * Conventions: every function returns 0 on success or a negative DFFW_E* code; the message is
 * available from dffw_last_error() (thread-local).  No C++ exception crosses this boundary.
 * "device" pointers are HIP device memory on the engine's device; "host" pointers are ordinary
 * memory.  All launches are enqueue-only on the caller's HIP stream (no implicit sync), so
 * timing around a call behaves like the reference's asynchronous model() call (test.py:117-119).
 */
#ifndef DFFW_H
#define DFFW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFFW_OK 0
#define DFFW_EINVAL (-1)  /* bad argument / shape contract violated */
#define DFFW_EHIP (-2)    /* a HIP runtime call failed */
#define DFFW_ENOMEM (-3)  /* workspace too small */
#define DFFW_EMISSING (-4)/* a required state-dict tensor was not supplied */

/* Arithmetic of the conv contractions (accumulation is always fp32 in the MFMA). */
#define DFFW_PREC_BF16X3 0 /* split-bf16: x = hi+lo, 3 MFMA products hi*hi+hi*lo+lo*hi (default; ~1e-5 rel-L2) */
#define DFFW_PREC_FP16 1   /* single fp16 product (~1e-3 rel-L2, weight dependent) */
#define DFFW_PREC_BF16 2   /* single bf16 product (~5e-3..1e-2 rel-L2; does not meet the 1e-3 target) */

/* Which network of the reference an engine implements. */
#define DFFW_NET_DEPTH 0 /* Depth_Estimation_Network.Network  (DEN.py:7-13): DFF_net alone, 384 entries */
#define DFFW_NET_E2E 1   /* End_to_End.Network (End_to_End/End_to_End.py:9-16): FlowNetwork alignment + DFF_net, 522 entries */
#define DFFW_E2E_SLICES 10 /* the alignment heads end in AdaptiveAvgPool3d((10,1,1)) (End_to_End.py:46,57,68) */

typedef struct dffw_engine dffw_engine;

/* One state-dict entry handed to the engine: the reference's key (an optional leading "module."
 * is accepted, test.py:32,78 vs train_code_Defocus.py:66), host fp32 data in PyTorch layout
 * (Conv3d (Cout,Cin,kd,kh,kw); ConvTranspose3d (Cin,Cout,kd,kh,kw); BatchNorm vectors). */
typedef struct dffw_tensor {
    const char *name;
    const float *data;
    int64_t numel;
} dffw_tensor;

/* Optional debug tap: after the forward, the named intermediate volume is written to `dst`
 * (device, fp32, reference layout (B,C,N,h,w), or (B,N,h,w) for the 1-channel score volumes).
 * Names: V1 V2 V3 FS_volume conf cost1 cost2 cost3; for dffw_forward_e2e also
 * head3 head2 head1 (each alpha head's output before the 0.001 damping, (B,3,N)) and alpha (the accumulated
 * warp parameters the stack is finally warped with, (B,3,N)). */
typedef struct dffw_tap {
    const char *name;
    float *dst;
    int64_t numel;
} dffw_tap;

const char *dffw_version(void);
const char *dffw_last_error(void);
/* Name of the kernel instantiation the calling thread's most recent convolution launch used (as rocprofv3 spells
 * it, e.g. "dffw::conv_roll<0, 8, 16, 4>"); "" before any.  Parity tests use it to prove which kernel they hit. */
const char *dffw_last_conv_kernel(void);

/* The weight contract (replaces nn.Module.state_dict() of DEN.py:7-57): number of state-dict
 * entries of network `net`, and entry `index` in the reference's registration order.
 * flags bit0: buffer (BatchNorm running stats) rather than parameter; bit1: int64
 * num_batches_tracked scalar; bit2: dead (present in checkpoints, never read by forward). */
int dffw_param_count(int net);
int dffw_param_info(int net, int index, const char **name, int64_t shape[5], int *ndim, int *flags);

/* Replaces `Network(); model.load_state_dict(...); model.cuda()` (test.py:30,78,80): folds
 * BatchNorm into the convs, packs weights into MFMA fragment order for `precision` and uploads
 * them to `device`.  The engine owns only that packed copy. */
int dffw_engine_create(int device, int net, const dffw_tensor *tensors, int n_tensors,
                       int precision, dffw_engine **out);
void dffw_engine_destroy(dffw_engine *e);
int dffw_engine_precision(const dffw_engine *e);

/* Bytes of scratch the engine's forward (dffw_forward for DFFW_NET_DEPTH, dffw_forward_e2e for
 * DFFW_NET_E2E) needs for one (B,N,H,W) call; the caller allocates it (torch's caching allocator in the
 * Python binding) so nothing is hipMalloc'ed per call. */
int64_t dffw_workspace_bytes(const dffw_engine *e, int B, int N, int H, int W);

/* Replaces `mid, p1, p2, p3 = model(FS, focus_dists)` (test.py:118; DFF_net.forward DEN.py:74-127).
 *   FS            device fp32, contiguous (B,3,N,H,W) — channel BEFORE slice (test_Dataloader.py:39)
 *   focus_dists   device fp32, element strides fd_strides[4] over (B,N,H,W); stride 0 = broadcast
 *                 ((B,N,1,1) of End_to_End/Test_dataloader.py:52-54, or dense (B,N,H,W))
 *   out[4]        device fp32 (B,H,W) each: mid_out, pred1, pred2, pred3 (DEN.py:127); entries may
 *                 be NULL to skip storing that map
 *   H, W          multiples of 32 (else DFFW_EINVAL, as the reference fails in torch.cat / pooling)
 */
int dffw_forward(dffw_engine *e, const float *FS, const float *focus_dists,
                 const int64_t fd_strides[4], int B, int N, int H, int W, float *const out[4],
                 void *workspace, int64_t workspace_bytes, void *hip_stream);

/* Same, additionally exporting intermediate volumes for parity debugging. */
int dffw_forward_taps(dffw_engine *e, const float *FS, const float *focus_dists,
                      const int64_t fd_strides[4], int B, int N, int H, int W, float *const out[4],
                      void *workspace, int64_t workspace_bytes, void *hip_stream,
                      const dffw_tap *taps, int n_taps);

/* Replaces `mid, p1, p2, p3, aligned = model(FS, focus_dists, FOVs)` of the End_to_End variant
 * (End_to_End/End_to_End.py:13-16; call site End_to_End/TRS.py:44): FlowNetwork.forward (End_to_End.py:71-105)
 * estimates per-slice warp parameters coarse to fine and warps the stack (FOV_warp, End_to_End.py:106-134),
 * DFF_net runs on the aligned stack.  Engine must have been created with DFFW_NET_E2E.
 *   FS, focus_dists, fd_strides, out[4], workspace, hip_stream   as for dffw_forward, N must be 10
 *   fovs          device fp32 (B,N): relative field of view of every slice (Test_dataloader.py:56-70)
 *   aligned       device fp32 (B,3,N,H,W): receives the aligned focal stack (5th return value); required
 *   taps          optional (NULL, 0)
 * Batch > 1 uses per-sample warp parameters, i.e. equals a stack of batch-1 reference calls (the reference's
 * own batch>1 path broadcasts sample 0's scale term by accident and is only ever run with batch 1, TRS.py:23). */
int dffw_forward_e2e(dffw_engine *e, const float *FS, const float *focus_dists,
                     const int64_t fd_strides[4], const float *fovs, int B, int N, int H, int W,
                     float *const out[4], float *aligned, void *workspace, int64_t workspace_bytes,
                     void *hip_stream, const dffw_tap *taps, int n_taps);

/* ---- per-launch timing (bench.py's roofline figures) -------------------------------------------
 * With profiling enabled every kernel launch of the next forward is bracketed by HIP events on the
 * caller's stream; dffw_profile_collect waits for them and returns one entry per launch.  Strings
 * stay valid until the next forward or the engine's destruction. */
typedef struct dffw_prof_entry {
    const char *kernel; /* kernel name as rocprofv3 --kernel-trace prints it, e.g. "dffw::conv_igemm<0, 1, 4>" */
    const char *layer;  /* state-dict prefix of the conv (e.g. "DFF_net.dres4.conv0.0.0") or the op name */
    double flops;       /* algorithmic FLOPs of the launch: 2*MAC of the layer's definition (DEN.py), no packing padding */
    double bytes;       /* algorithmic HBM bytes: activations read once + written once (+ residuals) in storage format */
    float ms;           /* elapsed time between the two events */
} dffw_prof_entry;
int dffw_profile_enable(dffw_engine *e, int on);
int dffw_profile_collect(dffw_engine *e, dffw_prof_entry *out, int capacity); /* returns the number of entries */

/* ---- single-operator entry points (parity tests of each kernel family; they allocate their own
 * temporaries with hipMalloc and synchronise the stream before returning) -------------------- */

/* y = [relu]( BN(conv(x)) [+ residual] ) in the engine's arithmetic.  Replaces nn.Conv3d /
 * nn.ConvTranspose3d (+ nn.BatchNorm3d, DEN.py:286-289).  x, residual, y: device fp32 in the
 * reference layout (B,C,N,H,W).  weight: host fp32, PyTorch layout.  bn: host fp32 4*Cout
 * values gamma|beta|mean|var, or NULL.  conv_bias: host fp32 Cout or NULL.  For transposed != 0
 * the geometry is fixed to the reference's only form: k3, stride (1,2,2), pad 1, output_padding
 * (0,1,1).  relu: 0 none, 1 relu(acc+res), 2 relu(acc)+res. */
int dffw_op_conv3d(int device, int precision, const float *x, int B, int Cin, int N, int H, int W,
                   const float *weight, int Cout, const int kernel[3], const int stride[3],
                   const int pad[3], const int dilation[3], int transposed, const float *bn,
                   const float *conv_bias, const float *residual, int relu, float *y,
                   void *hip_stream);

/* mode 0: max-pool (1,k,k) stride (1,k,k) (DEN.py:310); mode 1: average-pool (DEN.py:149-153). */
/* The same operator with the two extras the hourglass's last layer uses (DEN.py:96-97, 260-284: `out_in = x + conv6(...)`,
 * `cost = classif(out_in)`): y_pre (device fp32, y's shape, or NULL) receives BN(conv(x)) BEFORE the residual add; cls_weight (host fp32
 * Cout values, or NULL) is a bias-free 1x1x1 Cout -> 1 classifier applied to the final value y, its scores written to cls_score (device
 * fp32 (B,No,Ho,Wo)).  Needs Cout % 8 == 0. */
int dffw_op_conv3d_ex(int device, int precision, const float *x, int B, int Cin, int N, int H, int W,
                      const float *weight, int Cout, const int kernel[3], const int stride[3],
                      const int pad[3], const int dilation[3], int transposed, const float *bn,
                      const float *conv_bias, const float *residual, int relu, float *y, float *y_pre,
                      const float *cls_weight, float *cls_score, void *hip_stream);

int dffw_op_pool(int device, int precision, int mode, int k, const float *x, int B, int C, int N,
                 int H, int W, float *y, void *hip_stream);

/* The regression block of DEN.py:86-90: bilinear resize (align_corners=False) of score (B,N,h,w) to
 * (H,W), softplus+1e-6, normalise over N, sum_N focus_dists*p -> depth (B,H,W).  All device fp32. */
int dffw_op_regress(int device, const float *score, int B, int N, int h, int w, int H, int W,
                    const float *focus_dists, const int64_t fd_strides[4], float *depth,
                    void *hip_stream);

/* The warp operator of the End_to_End alignment path on its own.
 * Replaces FlowNetwork.FOV_warp (End_to_End/End_to_End.py:106-134): warps x (B,C,N,H,W)
 * by the per-slice field-of-view scale and translation.  alpha: device fp32 (B,3,N) = (scale offset,
 * x shift, y shift) per slice; fovs: device fp32 (B,N); out: (B,C,N,H,W); flow: (B,2,N,H,W) or NULL (the
 * pixel-unit flow the reference returns as its second value).  alpha_from_sample0 != 0 reproduces the
 * reference's batch>1 broadcast quirk (End_to_End.py:112: `alpha[:,0,:,:] + FOVs` broadcasts to (B,B,N,1,1) and
 * `[:,0]` keeps alpha[0,n] + FOVs[b,n]: every sample uses SAMPLE 0's scale offset with its OWN FOV). */
int dffw_op_fov_warp(int device, const float *x, int B, int C, int N, int H, int W, const float *alpha,
                     const float *fovs, int alpha_from_sample0, float *out, float *flow, void *hip_stream);

/* ---- input pipeline ---------------------------------------------------------------
 * Replaces the NumPy tensor assembly of the reference's loaders: `FS/127.5 - 1.0`, the transpose to (3,N,H,W) and the
 * bottom/right padding to multiples of 32 with -1 (Depth_Estimation_Test/test_Dataloader.py:80-89 HCI, :122-141 DDFF,
 * :197-228 Smartphone; End_to_End/Test_dataloader.py:56-75 Real_Scenes: all float32 arrays, i.e. a float32 divide followed
 * by a float32 subtract).  The FS6 / DefocusNet loader (test_Dataloader.py:31-39) differs: it concatenates its images onto
 * np.zeros((256,256,3,0)), a FLOAT64 array, so its `/127.5 - 1.0` runs in float64 and torch.Tensor() rounds once at the end
 * (128 of the 256 byte values then differ by one float32 ulp from the float32 form): OR DFFW_RAW_NORM_F64 into `dtype` to get
 * that arithmetic, (float)((double)v / 127.5 - 1.0).
 *   raw      device uint8 (DFFW_RAW_U8) or fp32 0..255 (DFFW_RAW_F32) stack in ANY source layout, described by
 *            element strides {batch, slice, row, col, channel}: (N,H,W,3) hdf5 stacks, (H,W,3,N) / (H,W,N,3) image
 *            arrays; a crop (test_Dataloader.py:205, Test_dataloader.py:58) is a pointer offset by the caller
 *   h, w     rows / cols taken from the source;  Hp, Wp  padded size, multiples of 32, >= h, w
 *   FS       device fp32 (B,3,N,Hp,Wp): exactly the tensor the reference's loader yields (float32 divide, then
 *            subtract; -1 in the padding).  Enqueue-only on hip_stream. */
#define DFFW_RAW_U8 0
#define DFFW_RAW_F32 1
#define DFFW_RAW_NORM_F64 16 /* flag OR-ed into dtype: normalise in float64, round once (the FS6 loader) */
int dffw_pack_stack(int device, const void *raw, int dtype, const int64_t strides[5], int B, int N, int h, int w,
                    int Hp, int Wp, float *FS, void *hip_stream);

/* dffw_forward on the raw stack: the stem kernel's loader applies `x/127.5 - 1` and the -1 padding while it stages its
 * tiles, so the normalised fp32 stack (4x the bytes of a uint8 source) is never written or read.  Bit-identical to
 * dffw_pack_stack followed by dffw_forward.  raw / dtype / raw_strides / h / w as for dffw_pack_stack; H, W are the
 * padded sizes (multiples of 32, >= h, w).  DFFW_NET_DEPTH engines.  Workspace: dffw_workspace_bytes(B,N,H,W) plus
 * B*3*N*H*W*4 bytes (used only when the shape is too small for the tiled stem kernel and the stack is expanded first). */
int dffw_forward_raw(dffw_engine *e, const void *raw, int dtype, const int64_t raw_strides[5], int h, int w,
                     const float *focus_dists, const int64_t fd_strides[4], int B, int N, int H, int W,
                     float *const out[4], void *workspace, int64_t workspace_bytes, void *hip_stream);

/* ---- output post-processing -------------------------------------------------------
 * dffw_colorize replaces the crop + normalise + `cm.get_cmap('jet')` + uint8 pass of Depth_Estimation_Test/test.py:124-133
 * and End_to_End/test_real_scenes.py:40-52:  rgb[b,y,x,:] = jet((depth[b,y,x] - lo) / (hi - lo)) for y < h, x < w.
 *   depth    device fp32 (B,H,W) as returned by dffw_forward (pred3)
 *   mode     DFFW_RANGE_FIXED: lo, hi given (test.py:132, the data set's depth range);
 *            DFFW_RANGE_MINMAX: each map's own min / max over the whole padded map (test_real_scenes.py:40)
 *   range    device fp32 (B,2) scratch AND result: the (lo, hi) pair used for every map
 *   rgb      device uint8 (B,h,w,3), RGB, `(255 * colour).astype(uint8)` truncation (test_real_scenes.py:48-49)
 * Colour-map semantics are matplotlib's: index = trunc(float32(x * 256)), x == 1 -> 255, x < 0 / x > 1 clamp, NaN -> black.
 * dffw_jet_lut writes the 256 x 3 uint8 table the kernel uses (host memory) for inspection / tests. */
#define DFFW_RANGE_FIXED 0
#define DFFW_RANGE_MINMAX 1
int dffw_colorize(int device, const float *depth, int B, int H, int W, int h, int w, int mode, float lo, float hi,
                  float *range, uint8_t *rgb, void *hip_stream);
int dffw_jet_lut(uint8_t *lut768);

/* dffw_unpack_stack replaces the warped-stack export of End_to_End/test_real_scenes.py:42-47:
 *     np.squeeze(127.5 * (warp + 1.0)).astype(np.uint8), transposed to (H,W,3,N) and written slice by slice cropped to (h,w).
 *   warp     device fp32 (B,3,N,H,W): the aligned stack dffw_forward returns in out[4] for DFFW_NET_E2E engines
 *   images   device uint8 (B,N,h,w,3): slice n of stack b as an interleaved image in the channel order of the input (the
 *            reference reads BGR with cv2 and writes it back with cv2)
 * Arithmetic: float32 `127.5f * (v + 1.0f)`, truncated toward zero; values in [-1,1] land in 0..255.  Outside that NumPy's cast is
 * platform-defined; this entry point follows x86 NumPy (conversion to int32, low byte kept; NaN and |t| >= 2^31 give 0). */
int dffw_unpack_stack(int device, const float *warp, int B, int N, int H, int W, int h, int w, uint8_t *images, void *hip_stream);

/* dffw_metrics replaces the masked NumPy metrics of Depth_Estimation_Test/metrics.py:90-127 as called from
 * test.py:144-158: est = pred3 (B,H,W) cropped to (h,w) (test.py:124-126); gt fp32 (B,h,w); mask uint8 (B,h,w)
 * (non-zero = valid); conf fp32 (B,h,w) or NULL (Smartphone confidence, test.py:145-146).
 *   out      device fp64 (B, DFFW_N_METRICS): valid pixels, abs_rel, sq_rel, mse, mae, rmse, rmse_log,
 *            accuracy(1.25), accuracy(1.25^2), accuracy(1.25^3), mse_w_conf, mae_w_conf (the last two NaN without conf)
 *   scratch  device memory of dffw_metrics_scratch_bytes(B)
 * Per-pixel terms are float32 like the NumPy expressions; sums are float64 in a fixed order (run-to-run identical). */
#define DFFW_N_METRICS 12
int64_t dffw_metrics_scratch_bytes(int B);
int dffw_metrics(int device, const float *est, int B, int H, int W, const float *gt, const uint8_t *mask,
                 const float *conf, int h, int w, double *out, void *scratch, int64_t scratch_bytes, void *hip_stream);

/* ---- measured ceilings of the GPU at hand (bench.py prints them beside the datasheet peaks) -------------------------------
 * mfma_tflops: v_mfma_f32_16x16x32_bf16 issued back to back out of registers on every SIMD (no memory traffic);
 * hbm_gbs: float4 streaming (copy of 1 GiB with read + write counted, or a pure read of 2 GiB: the better).  Synchronises;
 * allocates 2 GiB for the duration of the call. */
int dffw_probe_peaks(int device, float *mfma_tflops, float *hbm_gbs, void *hip_stream);

/* ---- multi-GPU: RCCL all-gather of the depth maps ---------------------------------------------
 * The forward has no cross-sample operation, so a batch is sharded on dim 0 over the GPUs of a node with no data-path
 * collective; what the reference's nn.DataParallel does after the replicas finish (Depth_Estimation_Test/test.py:32:
 * gather the outputs on device 0) becomes ONE ncclAllGather (RCCL over xGMI) of every rank's (b,H,W) fp32 maps, enqueued
 * on the compute stream behind the last head kernel.  librccl.so is bound with dlopen at the first call (override the
 * path with the environment variable DFFW_RCCL_LIB); nothing here is needed for single-GPU use.
 *   one process per GPU:  rank 0 calls dffw_comm_unique_id and hands the 128 bytes to the other ranks (file, pipe, MPI ...);
 *                         every rank calls dffw_comm_init_rank(device, nranks, rank, id, &comm)
 *   one process, n GPUs:  dffw_comm_init_all(n, devices, comms); calls for several comms from one thread go between
 *                         dffw_comm_group_start / dffw_comm_group_end
 *   dffw_allgather        recv[r*count .. (r+1)*count) on every rank = rank r's send[0 .. count) (device fp32); equal
 *                         counts on all ranks; enqueue-only on hip_stream. */
#define DFFW_COMM_ID_BYTES 128
typedef struct dffw_comm dffw_comm;
int dffw_comm_unique_id(char id[DFFW_COMM_ID_BYTES]);
int dffw_comm_init_rank(int device, int nranks, int rank, const char id[DFFW_COMM_ID_BYTES], dffw_comm **out);
int dffw_comm_init_all(int ndev, const int *devices, dffw_comm **out /* [ndev] */);
void dffw_comm_destroy(dffw_comm *c);
int dffw_comm_rank(const dffw_comm *c);
int dffw_comm_size(const dffw_comm *c);
int dffw_allgather(dffw_comm *c, const float *send, float *recv, int64_t count, void *hip_stream);
int dffw_comm_group_start(void);
int dffw_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* DFFW_H */
