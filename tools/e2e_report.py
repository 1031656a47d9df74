"""Print the parity numbers of the End_to_End path against the committed goldens (GPU box)."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dffinthewild_amd import graph, synth  # noqa: E402
from dffinthewild_amd.End_to_End import Network  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from oracle.make_goldens_e2e import net_inputs  # noqa: E402

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "e2e_net_*.npz"))):
    g = np.load(path)
    entries = list(graph.param_entries(graph.e2e_convs()))
    sd = synth.state_dict_numpy(entries, seed=int(g["wseed"]), profile=str(g["profile"]))
    FS, fd, fov = (torch.from_numpy(a) for a in net_inputs(int(g["H"]), int(g["W"]), int(g["iseed"])))
    for prec in (sys.argv[1:] or ["bf16x3"]):
        m = Network(precision=prec)
        m.load_state_dict(cpu_ref.to_torch_state(sd))
        m = m.cuda().eval()
        with torch.no_grad():
            outs, taps = m.forward_with_taps(FS.cuda(), fd.cuda(), fov.cuda(), ["head3", "head2", "head1", "alpha"])
        line = [os.path.basename(path), prec]
        for tag in ("head3", "head2", "head1"):
            line.append(f"{tag}={cpu_ref.rel_l2(taps[tag].cpu().reshape(3, 10), g[tag]):.2e}")
        for name, o in zip(("mid_out", "pred1", "pred2", "pred3", "aligned"), outs):
            if name in g.files:
                line.append(f"{name}={cpu_ref.rel_l2(o.cpu(), g[name]):.2e}")
        print(" ".join(line), flush=True)
