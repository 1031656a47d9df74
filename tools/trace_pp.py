"""Summarise a conv_pp slot timeline (make TRACE=1; DFFW_TRACE_LAYER=<layer> DFFW_TRACE_OUT=<file>): per slot and group
[start, issued, done, after barrier] in shader cycles.  usage: python tools/trace_pp.py <file> [workgroup]"""
import sys
import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 512, 4, 4).astype(np.int64)
nwg = d.shape[0]
used = (d[:, :, 0, 0] != 0).sum(1)
print(f"{nwg} workgroups, slots per workgroup: min {used.min()} max {used.max()}")
rows = {"loader: epilogue+issue": [], "loader: wait": [], "loader: barrier wait": [], "matrix: contract": [], "matrix: barrier wait": [], "slot": []}
NG = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for w in range(nwg):
    n = used[w]
    for s in range(NG, n - NG - 1):
        for g in range(NG):
            t0, t1, t2, t3 = d[w, s, g]
            if t0 == 0:
                continue
            loader = ((s - g) % NG) == 0
            if loader:
                rows["loader: epilogue+issue"].append(t1 - t0)
                rows["loader: wait"].append(t2 - t1)
                rows["loader: barrier wait"].append(t3 - t2)
            else:
                rows["matrix: contract"].append(t2 - t0)
                rows["matrix: barrier wait"].append(t3 - t2)
        rows["slot"].append(d[w, s + 1, 0, 0] - d[w, s, 0, 0])
for k, v in rows.items():
    v = np.array(v)
    if len(v):
        print(f"{k:26s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  (n={len(v)})")
if len(sys.argv) > 2:
    w = int(sys.argv[2])
    base = d[w, 0, 0, 0]
    for s in range(min(used[w], 24)):
        print(s, *[" ".join(f"{int(x - base):8d}" if x > 0 else "       -" for x in d[w, s, g]) for g in range(NG)], sep="  |  ")
