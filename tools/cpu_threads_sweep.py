"""Times the oracle forward (B=1, 10x256x256) for several torch thread counts on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dffinthewild_amd import graph, synth
from oracle import cpu_ref
entries = list(graph.param_entries(graph.dff_net_convs()))
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries, 0).items()}
FS = torch.from_numpy(synth.focal_stack(1, 10, 256, 256, seed=1000)); fd = torch.from_numpy(synth.focus_dists(1, 10, 256, 256))
for B in (1, 8):
    FSb, fdb = FS.repeat(B, 1, 1, 1, 1), fd.repeat(B, 1, 1, 1)
    for th in (8, 16, 32, 64, 128):
        torch.set_num_threads(th)
        with torch.no_grad():
            cpu_ref.dff_forward(sd, FSb, fdb)
            t = time.perf_counter(); cpu_ref.dff_forward(sd, FSb, fdb); dt = time.perf_counter() - t
        print(f"B={B} threads={th:4d}  {dt:.3f} s  {B/dt:.3f} stacks/s", flush=True)
