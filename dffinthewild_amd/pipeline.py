"""Input and output side of the forward on the GPU: host-side mirror of what the
reference's loaders and scripts do in NumPy around `model(FS, focus_dists)`.

    FS = pack_stack(raw_u8, "NHWC")                       # test_Dataloader.py:121-141: /127.5-1, transpose, pad to x32 with -1
    fd = focus_dists(values, B)                           # (B,N,1,1): broadcast instead of np.tile(..., [1,H,W]) (test_Dataloader.py:24)
    _, _, _, pred3 = model(FS, fd)
    rgb = colorize(pred3, size=(H, W), vrange=(lo, hi))   # test.py:124-133;  vrange=None: test_real_scenes.py:40-52
    m = masked_metrics(pred3, gt, mask)                   # metrics.py:90-127 as called from test.py:144-158

Everything stays in device memory; the kernels live in libdffw.so (csrc/dffw_io.hip) and are reached through the C ABI
(include/dffw.h: dffw_pack_stack, dffw_colorize, dffw_metrics).  No CPU fallback: CPU tensors raise."""
from ctypes import c_void_p, c_int64

import numpy as np
import torch

from . import engine
from .engine import lib, _check, _stream_ptr

# element order of every source layout the reference's loaders build, as (slice, row, col, channel) axis positions
_LAYOUTS = {
    "NHWC": (0, 1, 2, 3),   # hdf5 stacks (N,H,W,3): DDFF test_Dataloader.py:121, HCI :78
    "HWCN": (3, 0, 1, 2),   # image arrays (H,W,3,N): FS6 test_Dataloader.py:31-35, Real_Scenes Test_dataloader.py:24
    "HWNC": (2, 0, 1, 3),   # image arrays (H,W,N,3): Smartphone test_Dataloader.py:197
}
METRIC_NAMES = ("valid", "abs_rel", "sq_rel", "mse", "mae", "rmse", "rmse_log", "accuracy_1", "accuracy_2", "accuracy_3",
                "mse_w_conf", "mae_w_conf")


def _dev(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{what} must be a CUDA (ROCm) tensor: dffinthewild_amd has no CPU path")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _norm_flag(norm):
    if norm not in ("f32", "f64"):
        raise ValueError(f"norm must be 'f32' or 'f64', got {norm!r}")
    return engine.RAW_NORM_F64 if norm == "f64" else 0


def pack_stack(raw, layout="NHWC", crop=None, multiple=32, norm="f32"):
    """raw: uint8 or float32 (0..255) CUDA tensor in `layout`, with or without a leading batch dim (any strides: a
    view of a larger buffer is fine).  crop = (y0, x0, h, w).  Returns float32 (B,3,N,Hp,Wp) = raw/127.5-1, padded at
    the bottom/right to multiples of 32 with -1: the tensor the reference's loaders hand to the model.
    norm="f32": float32 divide then subtract (the DDFF / HCI / Smartphone / Real_Scenes loaders); norm="f64": the FS6 /
    DefocusNet loader's arithmetic (test_Dataloader.py:31-39 builds a float64 array, torch.Tensor() rounds once)."""
    nflag = _norm_flag(norm)
    if layout not in _LAYOUTS:
        raise ValueError(f"unknown layout {layout!r} (one of {sorted(_LAYOUTS)})")
    dev = _dev(raw, "raw stack")
    if raw.dtype not in (torch.uint8, torch.float32):
        raise ValueError(f"raw stack must be uint8 or float32, got {raw.dtype}")
    if raw.dim() == 4:
        raw = raw.unsqueeze(0)
    if raw.dim() != 5:
        raise ValueError(f"raw stack must have 4 or 5 dims, got {tuple(raw.shape)}")
    an, ay, ax, ac = (1 + a for a in _LAYOUTS[layout])
    if raw.shape[ac] != 3:
        raise ValueError(f"layout {layout}: expected 3 colour channels on axis {ac}, got {raw.shape[ac]}")
    B, N, H, W = raw.shape[0], raw.shape[an], raw.shape[ay], raw.shape[ax]
    y0, x0, h, w = (0, 0, H, W) if crop is None else crop
    if y0 < 0 or x0 < 0 or h < 1 or w < 1 or y0 + h > H or x0 + w > W:
        raise ValueError(f"crop {crop} does not fit the {H}x{W} source")
    Hp, Wp = -(-h // multiple) * multiple, -(-w // multiple) * multiple
    st = raw.stride()
    strides = (c_int64 * 5)(st[0], st[an], st[ay], st[ax], st[ac])
    off = (y0 * st[ay] + x0 * st[ax]) * raw.element_size()
    FS = torch.empty((B, 3, N, Hp, Wp), dtype=torch.float32, device=raw.device)
    with torch.cuda.device(dev):
        _check(lib.dffw_pack_stack(dev, c_void_p(raw.data_ptr() + off), (0 if raw.dtype == torch.uint8 else 1) | nflag, strides, B, N, h, w,
                                   Hp, Wp, c_void_p(FS.data_ptr()), _stream_ptr(dev)), "dffw_pack_stack")
    return FS


def focus_dists(values, batch=1, device="cuda"):
    """(batch,N,1,1) float32 focus distances: the forward broadcasts them over the map, so the (N,H,W) tile of
    test_Dataloader.py:24,71,113,166 is never materialised."""
    v = torch.as_tensor(values, dtype=torch.float32, device=device).reshape(1, -1, 1, 1)
    return v.expand(batch, -1, -1, -1).contiguous()


def real_scene_crop(height, width):
    """(y0, x0, h, w) of the border crop End_to_End/Test_dataloader.py:20-23 applies to every slice (1/12 of each side, integer
    division) - the `crop=` argument of pack_stack."""
    cy, cx = height // 12, width // 12
    if cy < 1 or cx < 1:
        raise ValueError(f"the loader's [c:-c] crop is empty for a {height}x{width} image")    # NumPy: x[0:-0] is empty
    return cy, cx, height - 2 * cy, width - 2 * cx


def real_scene_inputs(focus_distances, focal_length, device="cuda"):
    """The two small inputs of End_to_End.Network besides the stack, from the values of a scene's focus_distance.txt /
    focal_length.txt, as End_to_End/Test_dataloader.py:37-53 builds them (float64 arithmetic, rounded to float32 by torch.Tensor)
    and the batch-1 DataLoader of test_real_scenes.py:24 stacks them:
        focus_dists  (1,N,1,1) float32 = 1 / d          relative_fov  (1,1,N,1,1) float32 = (1/f - 1/d) / min(1/f - 1/d)"""
    d = np.asarray([float(v) for v in focus_distances], dtype=np.float64)
    if d.ndim != 1 or d.size < 1:
        raise ValueError("focus_distances must be a non-empty sequence")
    rel = 1 / float(focal_length) - 1 / d
    rel = rel / np.min(rel)
    fd = torch.from_numpy((1 / d).astype(np.float32)).reshape(1, -1, 1, 1).to(device)
    fov = torch.from_numpy(rel.astype(np.float32)).reshape(1, 1, -1, 1, 1).to(device)
    return fd, fov


def unpack_stack(warp, size=None):
    """The aligned stack End_to_End.Network returns, (B,3,N,H,W) float32 CUDA in [-1,1] -> uint8 (B,N,h,w,3) slice images:
    `127.5 * (warp + 1.0)` truncated to uint8, cropped to size=(h, w), channel order kept (test_real_scenes.py:42-47)."""
    dev = _dev(warp, "warp")
    if warp.dim() != 5 or warp.shape[1] != 3 or warp.dtype != torch.float32:
        raise ValueError(f"warp must be float32 (B,3,N,H,W), got {warp.dtype} {tuple(warp.shape)}")
    x = warp.contiguous()
    B, _, N, H, W = x.shape
    h, w = (H, W) if size is None else size
    img = torch.empty((B, N, h, w, 3), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(dev):
        _check(lib.dffw_unpack_stack(dev, c_void_p(x.data_ptr()), B, N, H, W, h, w, c_void_p(img.data_ptr()), _stream_ptr(dev)), "dffw_unpack_stack")
    return img


def colorize(depth, size=None, vrange=None, return_range=False):
    """depth (B,H,W) or (H,W) float32 CUDA -> uint8 (B,h,w,3) RGB through matplotlib's 'jet' semantics.
    vrange=(lo, hi): fixed normalisation range (test.py:130-132); None: each map's own min/max over the whole padded
    map (test_real_scenes.py:40).  size=(h, w): crop (test.py:124-126, test_real_scenes.py:52)."""
    dev = _dev(depth, "depth")
    squeeze = depth.dim() == 2
    d = depth.unsqueeze(0) if squeeze else depth
    if d.dim() != 3 or d.dtype != torch.float32:
        raise ValueError(f"depth must be float32 (B,H,W), got {d.dtype} {tuple(d.shape)}")
    d = d.contiguous()
    B, H, W = d.shape
    h, w = (H, W) if size is None else size
    rng = torch.empty((B, 2), dtype=torch.float32, device=d.device)
    rgb = torch.empty((B, h, w, 3), dtype=torch.uint8, device=d.device)
    lo, hi = (0.0, 0.0) if vrange is None else (float(vrange[0]), float(vrange[1]))
    with torch.cuda.device(dev):
        _check(lib.dffw_colorize(dev, c_void_p(d.data_ptr()), B, H, W, h, w, 1 if vrange is None else 0, lo, hi,
                                 c_void_p(rng.data_ptr()), c_void_p(rgb.data_ptr()), _stream_ptr(dev)), "dffw_colorize")
    rgb = rgb[0] if squeeze else rgb
    return (rgb, rng) if return_range else rgb


def masked_metrics(est, gt, mask, conf=None):
    """est (B,H,W) float32 (pred3, still padded), gt (B,h,w) float32, mask (B,h,w) bool/uint8, conf (B,h,w) float32 or
    None.  Returns a float64 CUDA tensor (B,12) in METRIC_NAMES order (metrics.py:90-127)."""
    dev = _dev(est, "est")
    for name, t in (("gt", gt), ("mask", mask)) + ((("conf", conf),) if conf is not None else ()):
        if _dev(t, name) != dev:
            raise ValueError(f"{name} is on another device")
    if est.dim() == 2:
        est, gt, mask = est.unsqueeze(0), gt.unsqueeze(0), mask.unsqueeze(0)
        conf = conf.unsqueeze(0) if conf is not None else None
    est, gt = est.contiguous(), gt.contiguous().float()
    mask = mask.contiguous().to(torch.uint8)
    B, H, W = est.shape
    h, w = gt.shape[-2:]
    if mask.shape != gt.shape or (conf is not None and conf.shape != gt.shape) or gt.shape[0] != B:
        raise ValueError("gt / mask / conf shapes differ")
    conf = conf.contiguous().float() if conf is not None else None
    out = torch.empty((B, len(METRIC_NAMES)), dtype=torch.float64, device=est.device)
    nbytes = lib.dffw_metrics_scratch_bytes(B)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=est.device)
    with torch.cuda.device(dev):
        _check(lib.dffw_metrics(dev, c_void_p(est.data_ptr()), B, H, W, c_void_p(gt.data_ptr()), c_void_p(mask.data_ptr()),
                                c_void_p(conf.data_ptr()) if conf is not None else None, h, w, c_void_p(out.data_ptr()),
                                c_void_p(scratch.data_ptr()), nbytes, _stream_ptr(dev)), "dffw_metrics")
    return out
