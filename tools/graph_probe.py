"""Probe: does capturing the forward in a HIP graph (torch.cuda.CUDAGraph) cut batch-1 latency?"""
import sys
import time

import torch

sys.path.insert(0, ".")
from bench import build_model  # noqa: E402
from dffinthewild_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
for (B, N, S) in ((1, 10, 256), (1, 5, 224), (8, 10, 256)):
    model, sd = build_model("bf16x3", dev)
    FS = torch.from_numpy(synth.focal_stack(B, N, S, S, seed=1000)).to(dev)
    fd = torch.from_numpy(synth.focus_dists(B, N, 1, 1)).to(dev)
    with torch.no_grad():
        for _ in range(5):
            ref = model(FS, fd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            model(FS, fd)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model(FS, fd)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = model(FS, fd)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 50
    err = float((out[3] - ref[3]).abs().max())
    print(f"B={B} N={N} {S}x{S}: eager {eager*1e3:.3f} ms  graph {graph*1e3:.3f} ms  max|diff| {err:.2e}", flush=True)
