"""FOV_warp of the End_to_End alignment path: oracle restatement vs goldens made by
the reference's own FlowNetwork.FOV_warp (CPU test), HIP kernel vs the same goldens and vs the oracle (GPU)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from oracle.make_goldens_e2e import case_inputs

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "e2e_fov_warp_*.npz")))


def load(path):
    g = np.load(path)
    x, alpha, fov = case_inputs(int(g["C"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    return g, torch.from_numpy(x), torch.from_numpy(alpha), torch.from_numpy(fov)


def test_fov_warp_goldens_present():
    assert len(GOLDEN) == 3


@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_oracle_fov_warp_matches_reference(path):
    g, x, alpha, fov = load(path)
    out, flow = cpu_ref.fov_warp(x, alpha, fov)
    assert out.shape == g["out"].shape and flow.shape == g["flow"].shape
    assert float((flow - torch.from_numpy(g["flow"])).abs().max()) <= 1e-5
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_hip_fov_warp_matches_reference(lib_built, path):
    from dffinthewild_amd import engine
    g, x, alpha, fov = load(path)
    out, flow = engine.op_fov_warp(x.cuda(), alpha.cuda(), fov.cuda())
    assert float((flow.cpu() - torch.from_numpy(g["flow"])).abs().max()) <= 2e-5
    # the flow (pixels) feeds a bilinear sample of values in [-1,1]: 1e-5 px of flow error moves a sample by <= 2e-5
    assert float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) <= 1e-4
    assert cpu_ref.rel_l2(out.cpu(), g["out"]) <= 1e-5


def _leak_inputs():
    from oracle.make_goldens_e2e import batch2_inputs
    x, alpha, fov = (torch.from_numpy(a) for a in batch2_inputs())
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_fovwarp_batch2_quirk.npz"))
    # what the reference's broadcast amounts to (End_to_End.py:112): sample 0's scale offset, each sample's own FOV
    alpha_leak = alpha.clone()
    alpha_leak[1, 0] = alpha[0, 0]
    return x, alpha, fov, alpha_leak, g


def test_oracle_reproduces_reference_batch2_quirk():
    """Golden = the reference's own FOV_warp called with batch 2: equals per-sample warps with alpha[0]'s scale offset."""
    x, alpha, fov, alpha_leak, g = _leak_inputs()
    out, flow = cpu_ref.fov_warp(x, alpha_leak, fov)
    assert float((flow - torch.from_numpy(g["flow"])).abs().max()) <= 1e-5
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5
    out_ps, _ = cpu_ref.fov_warp(x, alpha, fov)          # the per-sample semantics differ from it for sample 1 only
    assert torch.equal(out_ps[0], out[0]) and not torch.allclose(out_ps[1], out[1], atol=1e-3)


@pytest.mark.gpu
def test_hip_fov_warp_batch_semantics(lib_built):
    """Per-sample semantics for batch > 1 (= stack of batch-1 reference calls); the compat switch reproduces the reference's
    batch>1 broadcast as the reference itself computes it (golden from its FOV_warp at batch 2): sample 0's scale offset,
    every sample's own FOV."""
    from dffinthewild_amd import engine
    x, alpha, fov, alpha_leak, g = _leak_inputs()
    ref, _ = cpu_ref.fov_warp(x, alpha, fov)
    out, _ = engine.op_fov_warp(x.cuda(), alpha.cuda(), fov.cuda())
    assert cpu_ref.rel_l2(out.cpu(), ref) <= 1e-5
    out_leak, flow_leak = engine.op_fov_warp(x.cuda(), alpha.cuda(), fov.cuda(), compat_batch_alpha0=True)
    assert float((flow_leak.cpu() - torch.from_numpy(g["flow"])).abs().max()) <= 2e-5
    assert cpu_ref.rel_l2(out_leak.cpu(), g["out"]) <= 1e-5
