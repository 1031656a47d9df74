"""ORACLE tooling — generates ``tests/golden/*.npz`` by running the REFERENCE ITSELF.

Runs only in the build container, where ``/root/reference`` is mounted: it imports the reference's
``Depth_Estimation_Network.Network`` (never copies it), loads the synthetic state dict produced by
``dffinthewild_amd.synth`` (same recipe the tests use to rebuild the inputs), runs the reference
CPU forward and stores the outputs (plus a few intermediate volumes for the tiniest case, captured
with forward hooks).  The fixtures hold data only: case parameters and expected outputs.

    python oracle/make_goldens.py            # rewrites tests/golden/den_*.npz
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dffinthewild_amd import graph, synth  # noqa: E402

REF_DIR = "/root/reference/Depth_Estimation_Test"

# name, B, N, H, W, focus layout, weight profile, weight seed, input seed, store taps?
CASES = [
    ("tiny_taps",   1, 4, 32, 32, "dense", "smooth", 0, 1000, True),
    ("one_slice",   1, 1, 32, 32, "dense", "smooth", 0, 1001, False),
    ("batch2_bcast", 2, 5, 64, 96, "bcast", "smooth", 0, 1002, False),
    ("n15_wide",    1, 15, 32, 64, "dense", "smooth", 1, 1003, False),
    ("he_n10_64",   1, 10, 64, 64, "dense", "he", 0, 1004, False),
    ("ddff_5x224",  1, 5, 224, 224, "bcast", "smooth", 0, 1005, False),
    ("full_10x256", 1, 10, 256, 256, "dense", "smooth", 0, 1006, False),
    ("he_10x256",   1, 10, 256, 256, "bcast", "he", 0, 1007, False),
    # a second full-size stack under the weights of "full_10x256": the two sit at different positions of the batch-32
    # test of BASELINE config 3 (tests/test_gpu_forward.py::test_config3_batch32_...)
    ("full2_10x256", 1, 10, 256, 256, "dense", "smooth", 0, 1008, False),
]


def case_inputs(B, N, H, W, layout, in_seed):
    FS = synth.focal_stack(B, N, H, W, seed=in_seed)
    fd = synth.focus_dists(B, N, H, W) if layout == "dense" else synth.focus_dists(B, N, 1, 1)
    return FS, fd


def main():
    sys.path.insert(0, REF_DIR)
    warnings.filterwarnings("ignore")
    from Depth_Estimation_Network import Network  # the reference, imported in place
    torch.manual_seed(0)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    entries = list(graph.param_entries(graph.dff_net_convs()))
    model = Network().eval()
    ref_sd = model.state_dict()
    assert [k for k, *_ in entries] == list(ref_sd.keys()), "state-dict keys/order differ from the reference"
    for k, shape, *_ in entries:
        assert tuple(ref_sd[k].shape) == tuple(shape), (k, ref_sd[k].shape, shape)

    only = set(sys.argv[1:])          # optional: regenerate the named cases only
    for name, B, N, H, W, layout, profile, wseed, iseed, want_taps in CASES:
        if only and name not in only:
            continue
        sd = synth.state_dict_numpy(entries, seed=wseed, profile=profile)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        FS, fd = case_inputs(B, N, H, W, layout, iseed)
        taps = {}
        hooks = []
        if want_taps:
            net = model.DFF_net
            for tag, mod in (("V1", net.FM_measure), ("V2", net.FM_conv1), ("V3", net.FM_conv2),
                             ("FS_volume", net.SPP_module), ("conf", net.confidence),
                             ("cost1", net.classif1), ("cost2", net.classif2), ("cost3", net.classif3)):
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, tag=tag: taps.__setitem__(tag, o.detach().clone())))
        with torch.no_grad():
            outs = model(torch.from_numpy(FS), torch.from_numpy(fd))
        for h in hooks:
            h.remove()
        payload = dict(B=B, N=N, H=H, W=W, layout=layout, profile=profile, wseed=wseed, iseed=iseed)
        names = ("mid_out", "pred1", "pred2", "pred3")
        big = H * W * B > 64 * 96 * 2
        for nm, t in zip(names, outs):
            if big and nm != "pred3":
                continue                      # large cases keep only the final depth map
            payload[nm] = t.numpy().astype(np.float32)
        for tag, t in taps.items():
            t = t.squeeze(1) if t.dim() == 5 and t.shape[1] == 1 else t
            payload["tap_" + tag] = t.numpy().astype(np.float32)
        path = os.path.join(out_dir, f"den_{name}.npz")
        np.savez_compressed(path, **payload)
        p3 = outs[3].numpy()
        print(f"{name:14s} {B}x3x{N}x{H}x{W} {layout:5s} {profile:6s} pred3 range [{p3.min():.4f},{p3.max():.4f}] "
              f"std {p3.std():.4f} -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path)//1024} KiB)")


if __name__ == "__main__":
    main()
