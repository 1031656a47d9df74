"""Summarise a step timeline of a persistent streaming kernel (StepTrace in csrc/dffw_device.h; library built with
tools/build_trace_lib.sh, DFFW_TRACE_LAYER=<layer> DFFW_TRACE_OUT=<file>).

File: [workgroup][wave][32 steps][8] u64 of s_memtime stamps, slot 7 of step 0 = XCC_ID << 32 | HW_ID.
usage: trace_steps.py file.bin [waves_per_workgroup=4] [name0,name1,...]   (names of the intervals between consecutive stamps)
Prints, per wave index, the median length of every interval between consecutive non-zero stamps of a step, the step period
(stamp 0 of step n+1 minus stamp 0 of step n) and how far apart the waves of one workgroup reach each stamp."""
import sys

import numpy as np

path = sys.argv[1]
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
names = sys.argv[3].split(",") if len(sys.argv) > 3 else None
a = np.fromfile(path, dtype=np.uint64).reshape(-1, nw, 32, 8)
hw = a[:, :, 0, 7].copy()
t = a[..., :7].astype(np.int64)
live = (t[:, :, :, 0] > 0).all(axis=(1, 2))          # workgroups that recorded all 32 steps on every wave
t = t[live]
hw = hw[live]
print(f"workgroups with a full record: {live.sum()} of {len(live)}")
if not live.any():
    sys.exit(0)
nst = int((t[0, 0, 5] > 0).sum())                      # stamps used per step
period = np.diff(t[:, :, :, 0], axis=2)                # (wg, wave, 31)
print(f"stamps per step {nst}; step period median {np.median(period):.0f} ticks, mean {period.mean():.0f}, p10 {np.percentile(period, 10):.0f}, p90 {np.percentile(period, 90):.0f}")
for w in range(nw):
    row = []
    for k in range(nst - 1):
        d = t[:, w, :, k + 1] - t[:, w, :, k]
        d = d[(t[:, w, :, k + 1] > 0) & (t[:, w, :, k] > 0)]
        lab = names[k] if names and k < len(names) else f"{k}->{k + 1}"
        row.append(f"{lab} {np.median(d):6.0f}")
    # tail: last stamp of the step to stamp 0 of the next step
    d = t[:, w, 1:, 0] - t[:, w, :-1, nst - 1]
    d = d[t[:, w, :-1, nst - 1] > 0]
    row.append(f"tail {np.median(d):6.0f}")
    print(f"  wave {w}: " + " | ".join(row))
# skew between the waves of a workgroup at every stamp
for k in range(nst):
    x = t[:, :, :, k]
    ok = (x > 0).all(axis=1)
    sk = (x.max(axis=1) - x.min(axis=1))[ok]
    if sk.size:
        print(f"  stamp {k}: spread over the workgroup's waves median {np.median(sk):6.0f}  p90 {np.percentile(sk, 90):6.0f}")
simd = (hw >> np.uint64(4)) & np.uint64(3)
cu = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw >> np.uint64(8)) & np.uint64(0xFF))
print(f"CUs seen {len(np.unique(cu[:, 0]))}; workgroups per CU {live.sum() / max(1, len(np.unique(cu[:, 0]))):.2f}; SIMD of waves 0..{nw - 1} of the first workgroup: {simd[0].tolist()}")
