"""Diagnostic: identical stacks in one batch must give bit-identical maps (position of a column in a workgroup's stream must not matter)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dffinthewild_amd import graph, synth, engine
from dffinthewild_amd.Depth_Estimation_Network import Network

B, N, H, W = 4, 10, 256, 256
entries = list(graph.param_entries(graph.dff_net_convs()))
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries, 0, "smooth").items()}
model = Network(); model.load_state_dict(sd); model = model.cuda().eval()
one = torch.from_numpy(synth.focal_stack(1, N, H, W, seed=1006))
FS = one.repeat(B, 1, 1, 1, 1).cuda()
fd = torch.from_numpy(synth.focus_dists(B, N, 1, 1)).cuda()
with torch.no_grad():
    outs = model(FS, fd)
for k, o in enumerate(outs):
    for b in range(1, B):
        d = (o[0] - o[b]).abs()
        nz = int((d > 0).sum())
        ys, xs = np.nonzero(d.cpu().numpy() > 0)
        print("out", k, "sample", b, "max", float(d.max()), "ndiff", nz, "rows", (ys.min(), ys.max()) if nz else None, "cols", (xs.min(), xs.max()) if nz else None)
