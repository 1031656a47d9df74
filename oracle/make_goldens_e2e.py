"""ORACLE tooling — golden vectors for the End_to_End path, produced by calling the REFERENCE itself in
the build container (imported in place, never copied): FlowNetwork.FOV_warp alone
(End_to_End/End_to_End.py:106-134) and the whole End_to_End.Network forward (End_to_End.py:9-16) on the
synthetic 522-entry state dict of dffinthewild_amd.synth.

    python oracle/make_goldens_e2e.py        # rewrites tests/golden/e2e_*.npz

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
from dffinthewild_amd import graph, synth  # noqa: E402

CASES = [("rgb_32x48", 3, 10, 32, 48, 2001), ("feat_16x16", 8, 10, 16, 16, 2002), ("wide_24x64", 3, 10, 24, 64, 2003)]


def case_inputs(C, N, H, W, seed):
    x = (2.0 * synth.uniform01("fov_warp_x", C * N * H * W, seed) - 1.0).astype(np.float32).reshape(1, C, N, H, W)
    u = synth.uniform01("fov_warp_alpha", 3 * N, seed).reshape(3, N)
    alpha = np.stack([0.08 * (u[0] - 0.5), 6.0 * (u[1] - 0.5), 5.0 * (u[2] - 0.5)]).astype(np.float32).reshape(1, 3, N, 1, 1)
    fov = (1.0 + 0.06 * np.arange(N, dtype=np.float64)[::-1] / max(N - 1, 1)).astype(np.float32).reshape(1, 1, N, 1, 1)
    return x, alpha, np.ascontiguousarray(fov)


# name, H, W, weight profile, weight seed, input seed, keep all five outputs?
NET_CASES = [
    ("smooth_64x96", 64, 96, "smooth", 0, 3001, True),
    ("he_64x64", 64, 64, "he", 0, 3002, True),
    ("seed1_96x128", 96, 128, "smooth", 1, 3003, False),
    # BASELINE config 5's stack size (the batch-8 test places it at two batch positions): pred3 whole, aligned stack sampled
    ("smooth_480x640", 480, 640, "smooth", 0, 3004, False),
]
ALIGNED_SAMPLE = (slice(None), slice(None), slice(0, 10, 3), slice(0, None, 16), slice(0, None, 16))   # of (1,3,10,H,W)


def net_inputs(H, W, seed, N=10):
    """Batch-1 inputs in the layout of End_to_End/Test_dataloader.py:39-70: FS (1,3,10,H,W), focus_dists
    (1,10,1,1), relative FOVs (1,1,10,1,1) decreasing to 1 at the last (reference) slice."""
    FS = synth.focal_stack(1, N, H, W, seed=seed)
    fd = synth.focus_dists(1, N, 1, 1)
    fov = (1.0 + 0.06 * np.arange(N, dtype=np.float64)[::-1] / (N - 1)).astype(np.float32).reshape(1, 1, N, 1, 1)
    return FS, fd, np.ascontiguousarray(fov)


def network_goldens(out_dir):
    from End_to_End import Network  # the reference, imported in place
    entries = list(graph.param_entries(graph.e2e_convs()))
    model = Network().eval()
    ref_sd = model.state_dict()
    assert [k for k, *_ in entries] == list(ref_sd.keys()), "state-dict keys/order differ from the reference"
    for k, shape, *_ in entries:
        assert tuple(ref_sd[k].shape) == tuple(shape), (k, ref_sd[k].shape, shape)
    heads = {}
    fa = model.optical_flow_aggregation
    for tag, mod in (("head3", fa.conv1), ("head2", fa.conv2), ("head1", fa.conv3)):
        # the reference damps the head output in place afterwards (End_to_End.py:86): clone in the hook
        mod.register_forward_hook(lambda m, i, o, tag=tag: heads.__setitem__(tag, o.detach().clone()))
    only = set(sys.argv[1:])          # optional: regenerate the named network cases only
    for name, H, W, profile, wseed, iseed, keep_all in NET_CASES:
        if only and name not in only:
            continue
        sd = synth.state_dict_numpy(entries, seed=wseed, profile=profile)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        FS, fd, fov = net_inputs(H, W, iseed)
        with torch.no_grad():
            outs = model(torch.from_numpy(FS), torch.from_numpy(fd), torch.from_numpy(fov))
        payload = dict(H=H, W=W, profile=profile, wseed=wseed, iseed=iseed,
                       **{k: v.numpy().reshape(3, 10) for k, v in heads.items()})
        for nm, t in zip(("mid_out", "pred1", "pred2", "pred3", "aligned"), outs):
            if keep_all or nm == "pred3":
                payload[nm] = t.numpy().astype(np.float32)
        if H * W >= 480 * 640:
            payload["aligned_sample"] = outs[4].numpy().astype(np.float32)[ALIGNED_SAMPLE]
        path = os.path.join(out_dir, f"e2e_net_{name}.npz")
        np.savez_compressed(path, **payload)
        print(name, tuple(outs[3].shape), "head1", payload["head1"][:, 0], os.path.getsize(path) // 1024, "KiB")


def main():
    sys.path.insert(0, "/root/reference/End_to_End")
    warnings.filterwarnings("ignore")
    from End_to_End import FlowNetwork  # the reference, imported in place
    network_goldens(os.path.join(ROOT, "tests", "golden"))
    if len(sys.argv) > 1:
        return
    net = FlowNetwork(8).eval()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, C, N, H, W, seed in CASES:
        x, alpha, fov = case_inputs(C, N, H, W, seed)
        with torch.no_grad():
            out, flow = net.FOV_warp(torch.from_numpy(x), torch.from_numpy(alpha), torch.from_numpy(fov))
        path = os.path.join(out_dir, f"e2e_fov_warp_{name}.npz")
        np.savez_compressed(path, C=C, N=N, H=H, W=W, seed=seed, out=out.numpy(), flow=flow.numpy())
        print(name, out.shape, flow.shape, float(out.abs().mean()), os.path.getsize(path) // 1024, "KiB")
    batch2_quirk_golden(out_dir)


def batch2_inputs():
    """Two different samples (inputs, warp parameters AND fields of view differ) for the batch>1 quirk golden."""
    x0, a0, f0 = case_inputs(3, 10, 16, 32, 2004)
    x1, a1, f1 = case_inputs(3, 10, 16, 32, 2005)
    return np.concatenate([x0, x1]), np.concatenate([a0, 0.5 * a1]), np.concatenate([f0, (f1 * 1.01).astype(np.float32)])


def batch2_quirk_golden(out_dir):
    """The reference's FOV_warp called with batch 2 (End_to_End.py:112 broadcasts sample 0's scale offset to every sample,
    each sample keeps its own FOV): what dffw_op_fov_warp(alpha_from_sample0=1) must reproduce."""
    from End_to_End import FlowNetwork
    x, alpha, fov = batch2_inputs()
    with torch.no_grad():
        out, flow = FlowNetwork(8).eval().FOV_warp(torch.from_numpy(x), torch.from_numpy(alpha), torch.from_numpy(fov))
    path = os.path.join(out_dir, "e2e_fovwarp_batch2_quirk.npz")
    np.savez_compressed(path, out=out.numpy(), flow=flow.numpy())
    print("batch2 quirk", out.shape, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
