"""Drop-in for the reference's ``End_to_End.Network`` (End_to_End/End_to_End.py:9-16) on MI355X.

Keeps the model-call API of ``End_to_End/TRS.py`` (reference lines 31-37, 44):

    from dffinthewild_amd.End_to_End import Network
    model = Network(); model = nn.DataParallel(model.cpu())
    model.load_state_dict(torch.load(path)); model = model.cuda(); model.eval()
    with torch.no_grad():
        mid_out, pred1, pred2, pred3, aligned = model(FS, focus_dists, FOVs)

FS (B,3,10,H,W) padded to multiples of 32 with -1, focus_dists (B,10,1,1) or dense, FOVs (B,1,10,1,1)
relative field of view per slice (Test_dataloader.py:39-70).  The module holds the reference's 522 state-dict
entries (``DFF_net.*`` then ``optical_flow_aggregation.*``) and hands the whole forward — alignment network,
FOV warp, depth network — to the HIP engine.  Batch > 1 uses per-sample warp parameters (= a stack of
batch-1 reference calls; the reference itself only runs batch 1, TRS.py:23).  No CPU path.
"""
import torch

from . import engine as _engine
from . import graph as _graph
from .Depth_Estimation_Network import Network as _DepthNetwork

__all__ = ["Network"]


class Network(_DepthNetwork):
    """``Network()(FS, focus_dists, FOVs) -> (mid_out, pred1, pred2, pred3, aligned_FS)`` (End_to_End.py:13-16,259)."""

    _NET = _engine.NET_E2E

    @staticmethod
    def _conv_rows():
        return _graph.e2e_convs()

    def _check_e2e(self, FS, focus_dists, FOVs):
        if not torch.is_tensor(FOVs):
            raise TypeError("FOVs must be a tensor")
        if not (torch.is_tensor(FS) and torch.is_tensor(focus_dists)):
            raise TypeError("FS and focus_dists must be tensors")
        _graph.check_e2e_shape(FS.shape, focus_dists.shape, FOVs.shape)
        FS, focus_dists = self._check_inputs(FS, focus_dists)
        if FOVs.device != FS.device:
            raise RuntimeError(f"FS is on {FS.device} but FOVs on {FOVs.device}")
        if FOVs.dtype != torch.float32:
            FOVs = FOVs.float()
        return FS, focus_dists, FOVs

    def forward(self, FS, focus_dists, FOVs):
        FS, focus_dists, FOVs = self._check_e2e(FS, focus_dists, FOVs)
        return self._engine_on(FS.device).forward_e2e(FS, focus_dists, FOVs)

    def forward_with_taps(self, FS, focus_dists, FOVs, names):
        """Debug variant: also returns {name: tensor} — head3, head2, head1 (each alpha head before damping) and
        alpha, all (B,3,N), plus the DFF_net taps of the depth module."""
        FS, focus_dists, FOVs = self._check_e2e(FS, focus_dists, FOVs)
        return self._engine_on(FS.device).forward_e2e(FS, focus_dists, FOVs, taps=list(names))
