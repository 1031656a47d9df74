"""Does running two half-batches on two HIP streams beat one full batch?  (inter-forward overlap: the tail of one kernel under the
head of another, MFMA-bound layers of one forward beside HBM-bound layers of the other)
usage: python tools/two_stream_probe.py [total batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dffinthewild_amd import graph, synth
from dffinthewild_amd.Depth_Estimation_Network import Network

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
entries = list(graph.param_entries(graph.dff_net_convs()))
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries, 0, "smooth").items()}


def model():
    m = Network()
    m.load_state_dict(sd)
    return m.cuda().eval()


FS = torch.from_numpy(synth.focal_stack(B, 10, 256, 256, seed=1000)).cuda()
fd = torch.from_numpy(synth.focus_dists(B, 10, 256, 256)).cuda()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


with torch.no_grad():
    m0 = model()
    print(f"one stream, batch {B}: {timed(lambda: m0(FS, fd)):.1f} stacks/s")
    for parts in (2, 4):
        ms = [model() for _ in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        bs = B // parts
        chunks = [(FS[i * bs:(i + 1) * bs].contiguous(), fd[i * bs:(i + 1) * bs].contiguous()) for i in range(parts)]

        def run():
            for m, s, (x, f) in zip(ms, streams, chunks):
                with torch.cuda.stream(s):
                    m(x, f)

        print(f"{parts} streams x batch {bs}: {timed(run):.1f} stacks/s")
        print(f"   (one stream, batch {bs}: {timed(lambda: ms[0](*chunks[0])) / parts * parts / 1:.1f} stacks/s)".replace("stacks/s)", f"stacks/s of {bs})"))
