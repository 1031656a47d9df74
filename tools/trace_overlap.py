"""From a conv_tile phase timeline (tools/trace_report.py format): per CU, how the contraction phases of the
co-resident workgroups line up.  Prints the share of CU time with 0 / 1 / 2+ workgroups in their contraction phase."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[(a[:, 4] > 0) & (a[:, 0] > 0)]
t = a[:, :5].astype(np.int64)
hw = a[:, 5]
key = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw >> np.uint64(8)) & np.uint64(0xFF))
acc = np.zeros(4)
busy_tot = 0
for k in np.unique(key)[:64]:
    m = key == k
    tt = t[m]
    lo, hi = tt[:, 0].min(), tt[:, 4].max()
    ev = []
    for r in tt:
        ev.append((r[2], 1))
        ev.append((r[3], -1))
    ev.sort()
    cur, last, n = 0, lo, np.zeros(4)
    for x, d in ev:
        n[min(cur, 3)] += x - last
        last = x
        cur += d
    n[0] += hi - last
    acc += n / (hi - lo)
    res = [(r[0], 1) for r in tt] + [(r[4], -1) for r in tt]
acc /= min(64, len(np.unique(key)))
print("share of CU time with k workgroups in contraction: " + "  ".join(f"k={i}{'+' if i == 3 else ''}: {v:.2f}" for i, v in enumerate(acc)))
# start-time offsets between workgroups co-resident on a CU (first CU): sorted start stamps modulo the tile period
k0 = np.unique(key)[0]
tt = t[key == k0]
starts = np.sort(tt[:, 0])
print("first CU: tile starts (first 12, relative):", (starts[:12] - starts[0]).tolist())
print("first CU: tile durations (first 6):", (tt[np.argsort(tt[:, 0])][:6, 4] - tt[np.argsort(tt[:, 0])][:6, 0]).tolist())
