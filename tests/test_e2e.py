"""End_to_End path (BASELINE config 5): alignment network + FOV warp + DFF_net.

CPU: the oracle restatement (oracle/cpu_ref.py: flow_forward, e2e_forward) against goldens made by running the
reference's End_to_End.Network itself (oracle/make_goldens_e2e.py), and the 522-key weight contract.
GPU: the HIP engine behind dffinthewild_amd.End_to_End.Network against the same goldens and the oracle.
"""
import glob
import os

import numpy as np
import pytest
import torch

from dffinthewild_amd import graph, synth
from oracle import cpu_ref
from oracle.make_goldens_e2e import net_inputs

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "e2e_net_*.npz")))
OUT_NAMES = ("mid_out", "pred1", "pred2", "pred3", "aligned")


def load(path):
    g = np.load(path)
    entries = list(graph.param_entries(graph.e2e_convs()))
    sd = synth.state_dict_numpy(entries, seed=int(g["wseed"]), profile=str(g["profile"]))
    FS, fd, fov = net_inputs(int(g["H"]), int(g["W"]), int(g["iseed"]))
    return g, sd, torch.from_numpy(FS), torch.from_numpy(fd), torch.from_numpy(fov)


def test_e2e_goldens_present():
    assert len(GOLDEN) == 3


def test_e2e_weight_contract():
    """522 entries: the 384 of DFF_net first, then the 138 of optical_flow_aggregation (End_to_End.py:9-12)."""
    entries = list(graph.param_entries(graph.e2e_convs()))
    keys = [k for k, *_ in entries]
    assert len(keys) == 522 and len(set(keys)) == 522
    assert keys[0].startswith("DFF_net.") and keys[383].startswith("DFF_net.")
    assert all(k.startswith("optical_flow_aggregation.") for k in keys[384:])
    shapes = {k: s for k, s, *_ in entries}
    assert shapes["optical_flow_aggregation.conv1.0.0.weight"] == (64, 66, 1, 3, 3)
    assert shapes["optical_flow_aggregation.conv3.6.weight"] == (3, 16, 1, 3, 3)
    assert shapes["optical_flow_aggregation.conv2.6.bias"] == (3,)
    assert shapes["optical_flow_aggregation.OF_feature1.0.feature.weight"] == (16, 8, 1, 1, 1)


@pytest.mark.parametrize("path", GOLDEN, ids=os.path.basename)
def test_oracle_e2e_matches_reference(path):
    g, sd, FS, fd, fov = load(path)
    taps = {}
    with torch.no_grad():
        outs = cpu_ref.e2e_forward(cpu_ref.to_torch_state(sd), FS, fd, fov, taps)
    for tag in ("head3", "head2", "head1"):
        assert float((taps[tag].reshape(3, 10) - torch.from_numpy(g[tag])).abs().max()) <= 1e-4, tag
    checked = 0
    for name, o in zip(OUT_NAMES, outs):
        if name in g.files:
            assert tuple(o.shape) == g[name].shape
            assert cpu_ref.rel_l2(o, g[name]) <= 1e-5, name
            checked += 1
    assert checked >= 1


def test_oracle_e2e_batch_is_per_sample():
    """Batch > 1 in the oracle = a stack of batch-1 reference calls (the reference's own batch>1 path
    broadcasts sample 0's alpha; it is only ever run with batch 1, TRS.py:23)."""
    g, sd, FS, fd, fov = load(GOLDEN[0])
    sd = cpu_ref.to_torch_state(sd)
    FS2 = torch.cat([FS, FS.flip(-1)], 0)
    fov2 = torch.cat([fov, 1.0 + (fov - 1.0) * 0.5], 0)
    fd2 = fd.expand(2, -1, -1, -1)
    with torch.no_grad():
        both = cpu_ref.e2e_forward(sd, FS2, fd2, fov2)
        second = cpu_ref.e2e_forward(sd, FS2[1:], fd, fov2[1:])
    assert cpu_ref.rel_l2(both[3][:1], g["pred3"]) <= 1e-5
    assert cpu_ref.rel_l2(both[3][1:], second[3]) <= 2e-5      # conv summation order differs with batch size
    assert cpu_ref.rel_l2(both[4][1:], second[4]) <= 2e-5


def test_e2e_shape_contract():
    graph.check_e2e_shape((1, 3, 10, 64, 96), (1, 10, 1, 1), (1, 1, 10, 1, 1))
    graph.check_e2e_shape((2, 3, 10, 64, 96), (2, 10, 64, 96), (2, 10))
    with pytest.raises(ValueError):
        graph.check_e2e_shape((1, 3, 5, 64, 96), (1, 5, 1, 1), (1, 1, 5, 1, 1))      # not 10 slices
    with pytest.raises(ValueError):
        graph.check_e2e_shape((1, 3, 10, 64, 96), (1, 10, 1, 1), (1, 1, 9, 1, 1))     # FOV count
    with pytest.raises(ValueError):
        graph.check_e2e_shape((1, 3, 10, 60, 96), (1, 10, 1, 1), (1, 1, 10, 1, 1))    # H not a multiple of 32
