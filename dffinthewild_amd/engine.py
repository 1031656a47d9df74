"""ctypes binding of ``libdffw.so`` (C ABI: ``include/dffw.h``).

This is the only place Python touches the native library.  There is no CPU or PyTorch fallback:
if the shared object is missing the import of this module raises, and every forward runs the
hand-written gfx950 kernels.  PyTorch is used for device memory (inputs, outputs, the workspace
come from its caching allocator) and for the current HIP stream, nothing else.
"""
import ctypes
import os
import threading
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFFW_LIB_PATH: load another build of the same library (A/B measurements of kernel variants on one box); in-tree by default
LIB_PATH = os.environ.get("DFFW_LIB_PATH") or os.path.join(_HERE, "libdffw.so")

PRECISIONS = {"bf16x3": 0, "fp16": 1, "bf16": 2}
NET_DEPTH = 0   # Depth_Estimation_Network.Network: DFF_net alone
NET_E2E = 1     # End_to_End.Network: alignment network + FOV warp + DFF_net


class DffwError(RuntimeError):
    pass


class _Tensor(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", POINTER(c_float)), ("numel", c_int64)]


class _Tap(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("dst", c_void_p), ("numel", c_int64)]


class _Prof(ctypes.Structure):
    _fields_ = [("kernel", c_char_p), ("layer", c_char_p), ("flops", ctypes.c_double), ("bytes", ctypes.c_double),
                ("ms", c_float)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise DffwError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C dffinthewild_amd/csrc). "
            "dffinthewild_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.dffw_version.restype = c_char_p
    lib.dffw_last_error.restype = c_char_p
    lib.dffw_last_conv_kernel.restype = c_char_p
    lib.dffw_param_count.argtypes = [c_int]
    lib.dffw_param_info.argtypes = [c_int, c_int, POINTER(c_char_p), POINTER(c_int64), POINTER(c_int), POINTER(c_int)]
    lib.dffw_engine_create.argtypes = [c_int, c_int, POINTER(_Tensor), c_int, c_int, POINTER(c_void_p)]
    lib.dffw_engine_destroy.argtypes = [c_void_p]
    lib.dffw_engine_destroy.restype = None
    lib.dffw_engine_precision.argtypes = [c_void_p]
    lib.dffw_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
    lib.dffw_workspace_bytes.restype = c_int64
    fwd = [c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_int, c_int, c_int, c_int,
           POINTER(c_void_p), c_void_p, c_int64, c_void_p]
    lib.dffw_forward.argtypes = fwd
    lib.dffw_forward_taps.argtypes = fwd + [POINTER(_Tap), c_int]
    lib.dffw_forward_e2e.argtypes = [c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_int, c_int, c_int, c_int,
                                     POINTER(c_void_p), c_void_p, c_void_p, c_int64, c_void_p, POINTER(_Tap), c_int]
    lib.dffw_forward_raw.argtypes = [c_void_p, c_void_p, c_int, POINTER(c_int64), c_int, c_int, c_void_p, POINTER(c_int64), c_int, c_int, c_int,
                                     c_int, POINTER(c_void_p), c_void_p, c_int64, c_void_p]
    lib.dffw_profile_enable.argtypes = [c_void_p, c_int]
    lib.dffw_profile_collect.argtypes = [c_void_p, POINTER(_Prof), c_int]
    lib.dffw_op_conv3d.argtypes = [c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_int,
                                   POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_int,
                                   POINTER(c_float), POINTER(c_float), c_void_p, c_int, c_void_p, c_void_p]
    lib.dffw_op_conv3d_ex.argtypes = [c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_int,
                                      POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_int,
                                      POINTER(c_float), POINTER(c_float), c_void_p, c_int, c_void_p, c_void_p, POINTER(c_float), c_void_p, c_void_p]
    lib.dffw_op_pool.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.dffw_op_fov_warp.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p]
    lib.dffw_op_regress.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    POINTER(c_int64), c_void_p, c_void_p]
    c_u8p = POINTER(ctypes.c_uint8)
    lib.dffw_pack_stack.argtypes = [c_int, c_void_p, c_int, POINTER(c_int64), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.dffw_unpack_stack.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.dffw_colorize.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]
    lib.dffw_jet_lut.argtypes = [c_u8p]
    lib.dffw_metrics_scratch_bytes.argtypes = [c_int]
    lib.dffw_metrics_scratch_bytes.restype = c_int64
    lib.dffw_metrics.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                 c_int64, c_void_p]
    lib.dffw_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.dffw_comm_init_rank.argtypes = [c_int, c_int, c_int, ctypes.c_char_p, POINTER(c_void_p)]
    lib.dffw_comm_init_all.argtypes = [c_int, POINTER(c_int), POINTER(c_void_p)]
    lib.dffw_comm_destroy.argtypes = [c_void_p]
    lib.dffw_comm_destroy.restype = None
    lib.dffw_comm_rank.argtypes = [c_void_p]
    lib.dffw_comm_size.argtypes = [c_void_p]
    lib.dffw_allgather.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.dffw_probe_peaks.argtypes = [c_int, POINTER(c_float), POINTER(c_float), c_void_p]
    return lib


lib = _load()
COMM_ID_BYTES = 128
RAW_NORM_F64 = 16   # DFFW_RAW_NORM_F64: the FS6 loader's float64 normalisation

# every symbol include/dffw.h declares (tests check the library exports each one)
ABI_SYMBOLS = (
    "dffw_version", "dffw_last_error", "dffw_param_count", "dffw_param_info", "dffw_engine_create",
    "dffw_engine_destroy", "dffw_engine_precision", "dffw_workspace_bytes", "dffw_forward",
    "dffw_forward_taps", "dffw_profile_enable", "dffw_profile_collect", "dffw_op_conv3d", "dffw_op_conv3d_ex", "dffw_op_pool", "dffw_op_regress",
    "dffw_op_fov_warp", "dffw_forward_e2e", "dffw_last_conv_kernel",
    "dffw_forward_raw", "dffw_pack_stack", "dffw_unpack_stack", "dffw_colorize", "dffw_jet_lut", "dffw_metrics_scratch_bytes", "dffw_metrics",
    "dffw_comm_unique_id", "dffw_comm_init_rank", "dffw_comm_init_all", "dffw_comm_destroy", "dffw_comm_rank", "dffw_comm_size",
    "dffw_allgather", "dffw_comm_group_start", "dffw_comm_group_end", "dffw_probe_peaks",
)


def _check(rc, what):
    if rc < 0:
        msg = lib.dffw_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise DffwError(f"{what} failed ({rc}): {msg}")
    return rc


def param_table(net=NET_DEPTH):
    """The library's view of the weight contract: list of (key, shape, flags)."""
    n = _check(lib.dffw_param_count(net), "dffw_param_count")
    out = []
    for i in range(n):
        name, shape, ndim, flags = c_char_p(), (c_int64 * 5)(), c_int(), c_int()
        _check(lib.dffw_param_info(net, i, byref(name), shape, byref(ndim), byref(flags)), "dffw_param_info")
        out.append((name.value.decode(), tuple(shape[:ndim.value]), flags.value))
    return out


def _stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32(t):
    return ctypes.cast(t.data_ptr(), POINTER(c_float))


class Engine:
    """Owns one ``dffw_engine`` (packed weights on one GPU).  Thread-compatible: calls on engines
    of different devices may run concurrently (ctypes releases the GIL)."""

    def __init__(self, state_dict, device, precision="bf16x3", net=NET_DEPTH):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise DffwError("the HIP engine needs a GPU device (no CPU fallback)")
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.precision = precision
        keep, arr = [], (_Tensor * len(state_dict))()
        i = 0
        for key, t in state_dict.items():
            if not torch.is_floating_point(t):
                continue  # num_batches_tracked counters carry no arithmetic
            h = t.detach().to("cpu", torch.float32).contiguous()
            keep.append(h)
            arr[i] = _Tensor(key.encode(), _f32(h), h.numel())
            i += 1
        self._h = c_void_p()
        _check(lib.dffw_engine_create(self.index, net, arr, i, PRECISIONS[precision], byref(self._h)), "dffw_engine_create")
        self._ws = {}
        self._ws_stream = None      # torch stream the cached workspace was last used on
        self._lock = threading.Lock()

    # an Engine owns a dffw_engine* and a device workspace: a second owner would free the handle twice
    def __copy__(self):
        raise TypeError("dffinthewild_amd.engine.Engine cannot be copied: build a new one from the state dict")

    def __deepcopy__(self, memo):
        raise TypeError("dffinthewild_amd.engine.Engine cannot be copied: build a new one from the state dict")

    def __reduce__(self):
        raise TypeError("dffinthewild_amd.engine.Engine cannot be pickled: save the model's state_dict instead")

    def __del__(self, _destroy=lib.dffw_engine_destroy):
        h = getattr(self, "_h", None)
        if h:
            _destroy(h)
            self._h = None

    def profile(self, on=True):
        """Bracket every kernel launch of the following forwards with HIP events (bench.py)."""
        _check(lib.dffw_profile_enable(self._h, int(on)), "dffw_profile_enable")

    def profile_collect(self):
        """[(kernel, layer, flops, bytes, ms)] for the launches of the last profiled forward."""
        n = _check(lib.dffw_profile_collect(self._h, None, 0), "dffw_profile_collect")
        arr = (_Prof * n)()
        _check(lib.dffw_profile_collect(self._h, arr, n), "dffw_profile_collect")
        return [(e.kernel.decode(), e.layer.decode(), e.flops, e.bytes, e.ms) for e in arr]

    def workspace_bytes(self, B, N, H, W):
        return _check(lib.dffw_workspace_bytes(self._h, B, N, H, W), "dffw_workspace_bytes")

    def _workspace(self, B, N, H, W, extra=0):
        key = (B, N, H, W, extra)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.workspace_bytes(B, N, H, W) + extra
            self._ws.clear()  # one resident workspace: shapes change rarely (test.py runs one dataset per process)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _call_ws(self, B, N, H, W, extra, call):
        """Run ``call(ws)`` (a C-ABI forward returning its status) on the cached workspace.  The size is cached per shape, but
        the engine's allocation path also depends on its DFFW_* switches (read per call): if it reports the workspace too
        small (-3) the size is asked for again under the present switches and the forward repeated once."""
        cur = torch.cuda.current_stream(self.index)
        if self._ws_stream is not None and self._ws_stream != cur and self._ws:
            # the workspace is one buffer shared by successive forwards: a forward issued under another torch stream must
            # not start before the previous one (on the old stream) is done with it
            ev = torch.cuda.Event()
            ev.record(self._ws_stream)
            cur.wait_event(ev)
        self._ws_stream = cur
        rc = call(self._workspace(B, N, H, W, extra))
        if rc == -3:
            # kernels of the failed attempt (main and internal side streams were joined by the engine) may still be running:
            # let them finish before the block goes back to the allocator
            cur.synchronize()
            self._ws.clear()
            rc = call(self._workspace(B, N, H, W, extra))
        return rc

    def forward(self, FS, focus_dists, taps=None):
        """FS (B,3,N,H,W) float32 on this engine's device; focus_dists broadcastable to (B,N,H,W).
        Returns (mid_out, pred1, pred2, pred3); with ``taps`` (list of names) also a dict of
        intermediate volumes in the reference's layout."""
        B, C, N, H, W = FS.shape
        FS = FS.contiguous()
        fd = focus_dists.expand(B, N, H, W)
        strides = (c_int64 * 4)(*fd.stride())
        outs = [torch.empty((B, H, W), dtype=torch.float32, device=FS.device) for _ in range(4)]
        optrs = (c_void_p * 4)(*[o.data_ptr() for o in outs])
        with self._lock, torch.cuda.device(self.index):
            def args(ws):
                return [self._h, c_void_p(FS.data_ptr()), c_void_p(fd.data_ptr()), strides, B, N, H, W, optrs,
                        c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(self.index)]
            if not taps:
                _check(self._call_ws(B, N, H, W, 0, lambda ws: lib.dffw_forward(*args(ws))), "dffw_forward")
                return tuple(outs)
            shapes = _tap_shapes(B, N, H, W)
            bufs = {nm: torch.empty(shapes[nm], dtype=torch.float32, device=FS.device) for nm in taps}
            tarr = (_Tap * len(bufs))(*[_Tap(nm.encode(), c_void_p(t.data_ptr()), t.numel()) for nm, t in bufs.items()])
            _check(self._call_ws(B, N, H, W, 0, lambda ws: lib.dffw_forward_taps(*args(ws), tarr, len(bufs))), "dffw_forward_taps")
            return tuple(outs), bufs


    def forward_raw(self, raw, strides, dtype, h, w, focus_dists, B, N, H, W):
        """dffw_forward_raw: the stem normalises and pads the raw (uint8 / 0..255 float32) stack on the fly.
        raw: tensor whose storage holds the stack; strides: element strides (sample, slice, row, col, channel) and the
        data pointer already points at the crop origin; H, W: padded sizes."""
        fd = focus_dists.expand(B, N, H, W)
        fst = (c_int64 * 4)(*fd.stride())
        outs = [torch.empty((B, H, W), dtype=torch.float32, device=fd.device) for _ in range(4)]
        optrs = (c_void_p * 4)(*[o.data_ptr() for o in outs])
        with self._lock, torch.cuda.device(self.index):
            st5 = (c_int64 * 5)(*strides)
            _check(self._call_ws(B, N, H, W, B * 3 * N * H * W * 4 + 256,
                                 lambda ws: lib.dffw_forward_raw(self._h, c_void_p(raw), dtype, st5, h, w, c_void_p(fd.data_ptr()), fst,
                                                                 B, N, H, W, optrs, c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(self.index))),
                   "dffw_forward_raw")
        return tuple(outs)

    def forward_e2e(self, FS, focus_dists, fovs, taps=None):
        """End_to_End.Network.forward: FS (B,3,10,H,W), focus_dists broadcastable to (B,10,H,W), fovs with B*10
        elements in (sample, slice) order.  Returns (mid_out, pred1, pred2, pred3, aligned FS); with ``taps``
        also a dict of intermediate values (head3/head2/head1/alpha as (B,3,N), plus the DFF_net taps)."""
        B, C, N, H, W = FS.shape
        FS = FS.contiguous()
        fd = focus_dists.expand(B, N, H, W)
        fov = fovs.reshape(B, N).contiguous()
        strides = (c_int64 * 4)(*fd.stride())
        outs = [torch.empty((B, H, W), dtype=torch.float32, device=FS.device) for _ in range(4)]
        aligned = torch.empty_like(FS)
        optrs = (c_void_p * 4)(*[o.data_ptr() for o in outs])
        with self._lock, torch.cuda.device(self.index):
            bufs, tarr, nt = {}, None, 0
            if taps:
                shapes = _tap_shapes(B, N, H, W)
                bufs = {nm: torch.empty(shapes[nm], dtype=torch.float32, device=FS.device) for nm in taps}
                tarr = (_Tap * len(bufs))(*[_Tap(nm.encode(), c_void_p(t.data_ptr()), t.numel()) for nm, t in bufs.items()])
                nt = len(bufs)
            _check(self._call_ws(B, N, H, W, 0,
                                 lambda ws: lib.dffw_forward_e2e(self._h, c_void_p(FS.data_ptr()), c_void_p(fd.data_ptr()), strides,
                                                                 c_void_p(fov.data_ptr()), B, N, H, W, optrs, c_void_p(aligned.data_ptr()),
                                                                 c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(self.index), tarr, nt)),
                   "dffw_forward_e2e")
        res = tuple(outs) + (aligned,)
        return (res, bufs) if taps else res


def _tap_shapes(B, N, H, W):
    return {
        "head3": (B, 3, N), "head2": (B, 3, N), "head1": (B, 3, N), "alpha": (B, 3, N),
        "V1": (B, 8, N, H, W), "V2": (B, 16, N, H // 2, W // 2), "V3": (B, 32, N, H // 4, W // 4),
        "FS_volume": (B, 32, N, H // 8, W // 8), "conf": (B, N, H // 8, W // 8),
        "cost1": (B, N, H // 4, W // 4), "cost2": (B, N, H // 2, W // 2), "cost3": (B, N, H, W),
    }


# ---- single-operator wrappers (used by the kernel parity tests) ----------------------------------
def _i3(v):
    v = (v, v, v) if isinstance(v, int) else tuple(v)
    return (c_int * 3)(*v)


def op_conv3d(x, weight, *, stride=1, pad=0, dilation=1, transposed=False, bn=None, bias=None,
              residual=None, relu=0, precision="bf16x3", want_pre=False, cls_weight=None):
    """y = [relu](BN(conv(x)) [+ residual]) through the MFMA implicit-GEMM kernel.  ``x`` (B,C,N,H,W)
    float32 on the GPU; ``weight`` CPU/GPU float32 in PyTorch layout; ``bn`` = (gamma, beta, mean, var).
    ``want_pre`` / ``cls_weight`` (dffw_op_conv3d_ex): also return BN(conv(x)) before the residual add and / or the scores
    of a bias-free 1x1x1 Cout -> 1 classifier applied to y (the hourglass's last layer, DEN.py:96-97): the result is then
    the tuple (y, y_pre or None, scores or None)."""
    B, Cin, N, H, W = x.shape
    w = weight.detach().to("cpu", torch.float32).contiguous()
    Cout = w.shape[1] if transposed else w.shape[0]
    k = tuple(w.shape[2:])
    s, p, d = _i3(stride), _i3(pad), _i3(dilation)
    if transposed:
        No, Ho, Wo = N, 2 * H, 2 * W
    else:
        No = N + 2 * p[0] - (k[0] - 1)
        Ho = (H + 2 * p[1] - d[1] * (k[1] - 1) - 1) // s[1] + 1
        Wo = (W + 2 * p[2] - d[2] * (k[2] - 1) - 1) // s[2] + 1
    bnh = None
    if bn is not None:
        bnh = torch.cat([t.detach().to("cpu", torch.float32).reshape(-1) for t in bn]).contiguous()
    bh = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
    y = torch.empty((B, No, Ho, Wo) if Cout == 1 else (B, Cout, No, Ho, Wo), dtype=torch.float32, device=x.device)
    x = x.contiguous()
    res = residual.contiguous() if residual is not None else None
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    ex = want_pre or cls_weight is not None
    y_pre = torch.empty_like(y) if want_pre else None
    cw = cls_weight.detach().to("cpu", torch.float32).reshape(-1).contiguous() if cls_weight is not None else None
    score = torch.empty((B, No, Ho, Wo), dtype=torch.float32, device=x.device) if cw is not None else None
    with torch.cuda.device(dev):
        args = (dev, PRECISIONS[precision], c_void_p(x.data_ptr()), B, Cin, N, H, W, _f32(w), Cout,
                (c_int * 3)(*k), s, p, d, int(transposed),
                _f32(bnh) if bnh is not None else None, _f32(bh) if bh is not None else None,
                c_void_p(res.data_ptr()) if res is not None else None, relu, c_void_p(y.data_ptr()))
        if ex:
            _check(lib.dffw_op_conv3d_ex(*args, c_void_p(y_pre.data_ptr()) if want_pre else None, _f32(cw) if cw is not None else None,
                                         c_void_p(score.data_ptr()) if cw is not None else None, _stream_ptr(dev)), "dffw_op_conv3d_ex")
            return y, y_pre, score
        _check(lib.dffw_op_conv3d(*args, _stream_ptr(dev)), "dffw_op_conv3d")
    return y


def probe_peaks(device=0):
    """(sustained bf16 MFMA TFLOP/s, sustained HBM copy GB/s) of this GPU, measured now (dffw_probe_peaks)."""
    m, h = c_float(), c_float()
    with torch.cuda.device(device):
        _check(lib.dffw_probe_peaks(device, byref(m), byref(h), _stream_ptr(device)), "dffw_probe_peaks")
    return m.value, h.value


def last_conv_kernel():
    """Kernel instantiation used by this thread's most recent convolution launch (rocprofv3 spelling)."""
    return lib.dffw_last_conv_kernel().decode()


def op_pool(x, k, mode="max", precision="bf16x3"):
    B, C, N, H, W = x.shape
    y = torch.empty((B, C, N, H // k, W // k), dtype=torch.float32, device=x.device)
    x = x.contiguous()
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(dev):
        _check(lib.dffw_op_pool(dev, PRECISIONS[precision], 0 if mode == "max" else 1, k, c_void_p(x.data_ptr()),
                                B, C, N, H, W, c_void_p(y.data_ptr()), _stream_ptr(dev)), "dffw_op_pool")
    return y


def op_regress(score, focus_dists, H, W):
    B, N, h, w = score.shape
    score = score.contiguous()
    fd = focus_dists.expand(B, N, H, W)
    depth = torch.empty((B, H, W), dtype=torch.float32, device=score.device)
    dev = score.device.index if score.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(dev):
        _check(lib.dffw_op_regress(dev, c_void_p(score.data_ptr()), B, N, h, w, H, W, c_void_p(fd.data_ptr()),
                                   (c_int64 * 4)(*fd.stride()), c_void_p(depth.data_ptr()), _stream_ptr(dev)),
               "dffw_op_regress")
    return depth


def op_fov_warp(x, alpha, fovs, compat_batch_alpha0=False):
    """FlowNetwork.FOV_warp of the reference's End_to_End path (End_to_End.py:106-134) on the GPU.
    x (B,C,N,H,W), alpha (B,3,N,1,1) or (B,3,N), fovs (B,1,N,1,1) or (B,N); returns (warped, flow (B,2,N,H,W))."""
    B, C, N, H, W = x.shape
    x = x.contiguous()
    a = alpha.reshape(B, 3, N).contiguous().float()
    f = fovs.reshape(B, N).contiguous().float()
    out = torch.empty_like(x)
    flow = torch.empty((B, 2, N, H, W), dtype=torch.float32, device=x.device)
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(dev):
        _check(lib.dffw_op_fov_warp(dev, c_void_p(x.data_ptr()), B, C, N, H, W, c_void_p(a.data_ptr()), c_void_p(f.data_ptr()),
                                    int(compat_batch_alpha0), c_void_p(out.data_ptr()), c_void_p(flow.data_ptr()),
                                    _stream_ptr(dev)), "dffw_op_fov_warp")
    return out, flow
