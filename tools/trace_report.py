"""Summarise a conv_tile phase timeline written with DFFW_TRACE_LAYER / DFFW_TRACE_OUT (see dffw_engine.cpp).

Per tile: s_memtime at 0 start, 1 fill issued, 2 fill landed (+barrier), 3 contraction done, 4 stores acknowledged;
slot 5 = XCC_ID << 32 | HW_ID.  Prints phase medians, the kernel span and the average number of workgroups resident
per CU (sweep over the [start, end] intervals of the tiles that ran on each CU)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[(a[:, 4] > 0) & (a[:, 0] > 0)]
t = a[:, :5].astype(np.int64)
t0 = t[:, 0].min()
span = t[:, 4].max() - t0
ph = np.diff(t, axis=1)
names = ["issue fill", "wait fill+barrier", "contraction", "epilogue+store ack"]
print(f"tiles {len(a)}  kernel span {span} ticks")
tot = t[:, 4] - t[:, 0]
for i, n in enumerate(names):
    print(f"  {n:22s} median {np.median(ph[:, i]):8.0f}  mean {ph[:, i].mean():8.1f}  p90 {np.percentile(ph[:, i], 90):8.0f}  share {ph[:, i].sum() / tot.sum():.2f}")
print(f"  {'tile total':22s} median {np.median(tot):8.0f}  mean {tot.mean():8.1f}")
hw = a[:, 5]
key = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw >> np.uint64(8)) & np.uint64(0xFF))
cus = np.unique(key)
res = []
for k in cus:
    m = key == k
    busy = (t[m, 4] - t[m, 0]).sum()
    res.append(busy / span)
print(f"CUs seen {len(cus)}  tiles per CU {len(a) / len(cus):.1f}  avg resident workgroups per CU {np.mean(res):.2f}")
print(f"per-CU tile throughput: one tile every {span / (len(a) / len(cus)):.0f} ticks; per-tile latency/throughput ratio {np.mean(tot) / (span / (len(a) / len(cus))):.2f}")
