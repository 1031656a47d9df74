"""ORACLE tooling — golden vectors for FlowNetwork.FOV_warp (End_to_End/End_to_End.py:106-134), produced by
calling the REFERENCE's own method in the build container (imported in place, never copied).

    python oracle/make_goldens_e2e.py        # rewrites tests/golden/e2e_fov_warp_*.npz

Inputs come from dffinthewild_amd.synth (recipe stored in the fixture), with large distinct
alpha/beta/gamma per slice so that the 0.001 scale the network applies to alpha does not hide errors
(SURVEY.md section 8c).  Batch 1 only: the reference's batch>1 path has the alpha-broadcast bug.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dffinthewild_amd import synth  # noqa: E402

CASES = [("rgb_32x48", 3, 10, 32, 48, 2001), ("feat_16x16", 8, 10, 16, 16, 2002), ("wide_24x64", 3, 10, 24, 64, 2003)]


def case_inputs(C, N, H, W, seed):
    x = (2.0 * synth.uniform01("fov_warp_x", C * N * H * W, seed) - 1.0).astype(np.float32).reshape(1, C, N, H, W)
    u = synth.uniform01("fov_warp_alpha", 3 * N, seed).reshape(3, N)
    alpha = np.stack([0.08 * (u[0] - 0.5), 6.0 * (u[1] - 0.5), 5.0 * (u[2] - 0.5)]).astype(np.float32).reshape(1, 3, N, 1, 1)
    fov = (1.0 + 0.06 * np.arange(N, dtype=np.float64)[::-1] / max(N - 1, 1)).astype(np.float32).reshape(1, 1, N, 1, 1)
    return x, alpha, np.ascontiguousarray(fov)


def main():
    sys.path.insert(0, "/root/reference/End_to_End")
    warnings.filterwarnings("ignore")
    from End_to_End import FlowNetwork  # the reference, imported in place
    net = FlowNetwork(8).eval()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, C, N, H, W, seed in CASES:
        x, alpha, fov = case_inputs(C, N, H, W, seed)
        with torch.no_grad():
            out, flow = net.FOV_warp(torch.from_numpy(x), torch.from_numpy(alpha), torch.from_numpy(fov))
        path = os.path.join(out_dir, f"e2e_fov_warp_{name}.npz")
        np.savez_compressed(path, C=C, N=N, H=H, W=W, seed=seed, out=out.numpy(), flow=flow.numpy())
        print(name, out.shape, flow.shape, float(out.abs().mean()), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
