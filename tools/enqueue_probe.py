"""Is the batch-1 forward bound by the host's launch rate?  Times the enqueue of K forwards (no sync) and the K forwards to completion.
usage: enqueue_probe.py [batch=1] [slices=10] [size=256] [K=200]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
K = int(sys.argv[4]) if len(sys.argv) > 4 else 200
from dffinthewild_amd import synth
dev = torch.device("cuda", 0)
model, _ = bench.build_model("bf16x3", dev)
FS = torch.from_numpy(synth.focal_stack(B, N, S, S, seed=1000)).to(dev)
fd = torch.from_numpy(synth.focus_dists(B, N, S, S)).to(dev)
with torch.no_grad():
    for _ in range(20):
        model(FS, fd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        model(FS, fd)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"batch {B} {N}x{S}x{S}: enqueue {1e3*(t1-t0)/K:.3f} ms per forward, to completion {1e3*(t2-t0)/K:.3f} ms per forward")
