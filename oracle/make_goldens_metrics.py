#!/usr/bin/env python3
"""Golden vectors for the masked metrics: outputs of the REFERENCE's own functions
`/root/reference/Depth_Estimation_Test/metrics.py:90-127` (mask_abs_rel ... mask_mae_w_conf) on seeded inputs, in the call order
of test.py:144-158.

`import metrics` fails in this image: the module's first lines import skimage (absent; used only by get_bumpiness, which is not
on the path).  So this script reads the reference file where it lies, takes the FunctionDef nodes that do not touch `skf`, and
compiles exactly those, unmodified, into a namespace holding numpy and torch - it executes the reference's code, it does not
restate it, and nothing of it is written into this repo.  Run here only (TEST INFRASTRUCTURE; /root/reference does not exist on
the GPU box):   python oracle/make_goldens_metrics.py   ->   tests/golden/io_metrics.npz
"""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference/Depth_Estimation_Test/metrics.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "io_metrics.npz")
NAMES = ["mask_abs_rel", "mask_sq_rel", "mask_mse", "mask_mae", "mask_rmse", "mask_rmse_log", "mask_accuracy_k", "mask_mse_w_conf",
         "mask_mae_w_conf"]


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            assert not any(isinstance(n, ast.Name) and n.id == "skf" for n in ast.walk(node)), node.name
            keep.append(node)
    assert sorted(n.name for n in keep) == sorted(NAMES), [n.name for n in keep]
    ns = {"np": np, "torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def reference_row(f, est, gt, mask, conf):
    """the 12 numbers of dffw_metrics (include/dffw.h) from the reference's functions"""
    row = [float(np.sum(mask)), f["mask_abs_rel"](est, gt, mask), f["mask_sq_rel"](est, gt, mask), f["mask_mse"](est, gt, mask),
           f["mask_mae"](est, gt, mask), f["mask_rmse"](est, gt, mask), f["mask_rmse_log"](est, gt, mask)]
    row += [f["mask_accuracy_k"](est, gt, k, mask) for k in (1, 2, 3)]
    row += [f["mask_mse_w_conf"](est, gt, conf, mask), f["mask_mae_w_conf"](est, gt, conf, mask)]
    return np.asarray(row, np.float64)


def main():
    f = reference_functions()
    rng = np.random.default_rng(20261002)
    cases = {}
    # (name, h, w, fraction of valid pixels, depth range) - test.py hands float32 arrays of the cropped size and a bool mask
    for name, h, w, valid, lo, hi in [("ddff_like", 96, 128, 0.7, 0.02, 0.28), ("fs6_like", 128, 128, 1.0, 0.1, 3.0),
                                      ("ragged", 61, 90, 0.4, 0.1, 1.5), ("one_pixel", 17, 9, 0.0, 0.5, 2.0)]:
        gt = (rng.random((h, w)) * (hi - lo) + lo).astype(np.float32)
        est = (gt * (1.0 + 0.35 * rng.standard_normal((h, w)))).astype(np.float32)
        est = np.maximum(est, np.float32(lo * 0.25))
        mask = rng.random((h, w)) < valid
        if not mask.any():
            mask[h // 2, w // 3] = True
        conf = rng.random((h, w)).astype(np.float32)
        cases[name] = (est, gt, mask, conf)
    out = {}
    for name, (est, gt, mask, conf) in cases.items():
        out[name + "/est"], out[name + "/gt"], out[name + "/mask"], out[name + "/conf"] = est, gt, mask, conf
        out[name + "/want"] = reference_row(f, est, gt, mask, conf)
    np.savez_compressed(OUT, **out)
    for name in cases:
        print(name, out[name + "/want"])


if __name__ == "__main__":
    sys.exit(main())
