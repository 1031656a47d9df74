"""Drop-in for the reference's ``Depth_Estimation_Network.Network`` on MI355X.

Keeps the model-call API of ``Depth_Estimation_Test/test.py`` (reference lines 30-32, 78-86, 118):

    from dffinthewild_amd.Depth_Estimation_Network import Network
    model = Network(); model = nn.DataParallel(model.cpu())
    model.module.load_state_dict(torch.load(path)); model = model.cuda(); model.eval()
    with torch.no_grad():
        mid_out, pred1, pred2, pred3 = model(FS, focus_dists)     # FS (B,3,N,H,W), each out (B,H,W)

The module holds the reference's 384 state-dict entries under the reference's names (so its
checkpoints load unchanged, dead entries included) but owns no PyTorch operators: ``forward`` hands
the stack to the HIP engine (``libdffw.so``), which runs the whole graph of ``DFF_net.forward``
(DEN.py:74-127) in hand-written gfx950 kernels.  There is no CPU path: CPU tensors raise.
"""
import math
import os
import threading

import torch
import torch.nn as nn

from . import engine as _engine
from . import graph as _graph

__all__ = ["Network"]


class _Scope(nn.Module):
    """Name-space node of the parameter tree (stands in for the reference's nn.Sequential nesting)."""


def _plant(root: nn.Module, dotted: str, value, is_buffer: bool):
    node = root
    parts = dotted.split(".")
    for name in parts[:-1]:
        child = node._modules.get(name)
        if child is None:
            child = _Scope()
            node.add_module(name, child)
        node = child
    if is_buffer:
        node.register_buffer(parts[-1], value)
    else:
        node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))


class Network(nn.Module):
    """``Network()(FS, focus_dists) -> (mid_out, pred1, pred2, pred3)`` (DEN.py:7-13,127).

    precision: arithmetic of the conv contractions, ``"bf16x3"`` (default; split-bf16 with fp32
    accumulation, ~1e-5 rel-L2 to the fp32 reference), ``"fp16"`` or ``"bf16"`` (faster, reported
    with their measured error).  Default can be set with the environment variable DFFW_PRECISION.
    """

    _NET = _engine.NET_DEPTH

    def __init__(self, precision=None):
        super().__init__()
        self.precision = precision or os.environ.get("DFFW_PRECISION", "bf16x3")
        if self.precision not in _engine.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_engine.PRECISIONS)}")
        self._convs = self._conv_rows()
        for key, shape, role, is_buffer in _graph.param_entries(self._convs):
            _plant(self, key, self._initial(shape, role), is_buffer)
        self._engines = {}          # device index -> (fingerprint, Engine); shared with DataParallel replicas
        self._token = [0]           # bumped by load_state_dict / .to() / .cuda()
        self._guard = threading.Lock()
        self._inherited_fp = None   # set on DataParallel replicas
        self._tensors = None        # cached list of parameters/buffers for _fingerprint

    # ---- copy / pickle: the usual nn.Module idioms (copy.deepcopy for EMA / SWA clones, torch.save(model)) ----------
    # The packed-weight cache holds ctypes engine handles and a lock: neither may be copied (a second owner of a
    # dffw_engine* would free it twice).  A copy therefore carries the parameters only and packs its own engine at its
    # first forward.  (DataParallel replicas are NOT copies in this sense: _replicate_for_data_parallel below shares
    # the cache on purpose.)
    _RUNTIME_STATE = ("_engines", "_token", "_guard", "_inherited_fp", "_tensors")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._RUNTIME_STATE:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._engines = {}
        self._token = [0]
        self._guard = threading.Lock()
        self._inherited_fp = None
        self._tensors = None

    # ---- weight contract ------------------------------------------------------------------------
    @staticmethod
    def _conv_rows():
        return _graph.dff_net_convs("DFF_net")

    @staticmethod
    def _initial(shape, role):
        """Same initial statistics as the reference constructor (DEN.py:59-73): He-normal convs with
        n = kd*kh*kw*C_out, BatchNorm gamma=1 beta=0 (running stats at PyTorch defaults)."""
        if role in (_graph.ROLE_CONV, _graph.ROLE_CONVT):
            cout = shape[1] if role == _graph.ROLE_CONVT else shape[0]
            n = shape[2] * shape[3] * shape[4] * cout
            return torch.empty(shape).normal_(0.0, math.sqrt(2.0 / n))
        if role in (_graph.ROLE_BN_W, _graph.ROLE_BN_VAR):
            return torch.ones(shape)
        if role == _graph.ROLE_BN_NBT:
            return torch.zeros((), dtype=torch.long)
        return torch.zeros(shape)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts the reference's keys with or without DataParallel's ``module.`` prefix
        (train_code_Defocus.py:66 saves the wrapped dict, train_code_DDFF.py:65 the inner one)."""
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate()
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        if hasattr(self, "_token"):
            self.invalidate()
        return out

    def invalidate(self):
        """Forget packed weights (call after editing parameters in place)."""
        self._token[0] += 1
        self._engines.clear()
        self._tensors = None

    def _fingerprint(self):
        """Cheap change detector for the packed-weight cache: a token bumped by load_state_dict/.to() plus
        the sum of the tensors' in-place version counters (the tensor list itself is cached: walking
        state_dict() costs milliseconds per call, which would dominate the batch-1 latency)."""
        if self._inherited_fp is not None:
            return self._inherited_fp
        ts = self._tensors
        if ts is None:
            ts = self._tensors = list(self.state_dict(keep_vars=True).values())
        v = 0
        for t in ts:
            v += t._version
        return (self._token[0], v)

    def _replicate_for_data_parallel(self):
        fp = self._fingerprint()
        replica = super()._replicate_for_data_parallel()
        replica._inherited_fp = fp
        return replica

    def _engine_on(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        fp = self._fingerprint()
        with self._guard:
            hit = self._engines.get(idx)
            if hit is None or hit[0] != fp:
                sd = {k: v for k, v in self.state_dict().items()}
                hit = (fp, _engine.Engine(sd, torch.device("cuda", idx), self.precision, self._NET))
                self._engines[idx] = hit
        return hit[1]

    # ---- the model call -------------------------------------------------------------------------
    def _check_inputs(self, FS, focus_dists):
        if not (torch.is_tensor(FS) and torch.is_tensor(focus_dists)):
            raise TypeError("FS and focus_dists must be tensors")
        _graph.check_stack_shape(FS.shape, focus_dists.shape)
        if self.training:
            raise RuntimeError("dffinthewild_amd.Network is an inference engine (BatchNorm folded with running "
                               "statistics): call model.eval() first, as test.py:85 does")
        if not FS.is_cuda:
            raise RuntimeError("dffinthewild_amd.Network runs only on a ROCm GPU (HIP kernels, no CPU fallback): "
                               "move the model and inputs with .cuda() as test.py:80,115-116 does")
        if focus_dists.device != FS.device:
            raise RuntimeError(f"FS is on {FS.device} but focus_dists on {focus_dists.device}")
        if FS.dtype != torch.float32:
            raise TypeError(f"FS must be float32 (the reference loaders yield float32), got {FS.dtype}")
        if focus_dists.dtype != torch.float32:
            focus_dists = focus_dists.float()
        return FS, focus_dists

    def forward(self, FS, focus_dists):
        FS, focus_dists = self._check_inputs(FS, focus_dists)
        return self._engine_on(FS.device).forward(FS, focus_dists)

    def forward_raw(self, raw, focus_dists, layout="NHWC", crop=None, norm="f32"):
        """The same forward fed with the stack as the loaders hold it BEFORE `FS/127.5 - 1.0` (uint8 or float32 0..255
        CUDA tensor in one of pipeline._LAYOUTS, optional crop (y0,x0,h,w)): normalisation, transpose and the -1
        padding to multiples of 32 (test_Dataloader.py:122-141) happen inside the stem kernel's loader.  Bit-identical
        to `self(pipeline.pack_stack(raw, layout, crop, norm=norm), focus_dists)`; output maps have the padded size.
        norm="f64" selects the FS6 loader's float64 normalisation (test_Dataloader.py:31-39), see pipeline.pack_stack."""
        from . import pipeline as _pl
        nflag = _pl._norm_flag(norm)
        if layout not in _pl._LAYOUTS:
            raise ValueError(f"unknown layout {layout!r}")
        if self.training:
            raise RuntimeError("dffinthewild_amd.Network is an inference engine: call model.eval() first, as test.py:85 does")
        if not (torch.is_tensor(raw) and raw.is_cuda):
            raise RuntimeError("dffinthewild_amd.Network runs only on a ROCm GPU (HIP kernels, no CPU fallback)")
        if raw.dtype not in (torch.uint8, torch.float32):
            raise TypeError(f"raw stack must be uint8 or float32, got {raw.dtype}")
        if raw.dim() == 4:
            raw = raw.unsqueeze(0)
        an, ay, ax, ac = (1 + a for a in _pl._LAYOUTS[layout])
        if raw.dim() != 5 or raw.shape[ac] != 3:
            raise ValueError(f"layout {layout}: bad raw stack shape {tuple(raw.shape)}")
        B, N, Hs, Ws = raw.shape[0], raw.shape[an], raw.shape[ay], raw.shape[ax]
        y0, x0, h, w = (0, 0, Hs, Ws) if crop is None else crop
        if y0 < 0 or x0 < 0 or h < 1 or w < 1 or y0 + h > Hs or x0 + w > Ws:
            raise ValueError(f"crop {crop} does not fit the {Hs}x{Ws} source")
        H, W = -(-h // 32) * 32, -(-w // 32) * 32
        _graph.check_stack_shape((B, 3, N, H, W), focus_dists.shape)
        if focus_dists.device != raw.device:
            raise RuntimeError(f"raw stack is on {raw.device} but focus_dists on {focus_dists.device}")
        st = raw.stride()
        ptr = raw.data_ptr() + (y0 * st[ay] + x0 * st[ax]) * raw.element_size()
        return self._engine_on(raw.device).forward_raw(ptr, (st[0], st[an], st[ay], st[ax], st[ac]), (0 if raw.dtype == torch.uint8 else 1) | nflag,
                                                       h, w, focus_dists.float(), B, N, H, W)

    def forward_with_taps(self, FS, focus_dists, names):
        """Debug variant: also returns {name: tensor} for intermediate volumes (V1, V2, V3,
        FS_volume, conf, cost1, cost2, cost3) in the reference's layout."""
        FS, focus_dists = self._check_inputs(FS, focus_dists)
        return self._engine_on(FS.device).forward(FS, focus_dists, taps=list(names))
