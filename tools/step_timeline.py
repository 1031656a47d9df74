"""Timeline of ONE forward from a rocprofv3 --kernel-trace CSV: every dispatch with its queue, start / end / duration in us from the step's first kernel, and the
time the chip sat idle before it (no kernel of any queue running) -- the gaps behind hipEventRecord / hipStreamWaitEvent on the main stream show up there.
usage: step_timeline.py <..._kernel_trace.csv> [first-kernel substring, default stem_pipe] [which step from the end, default 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "stem_pipe"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0, i1 = idx[-back - 1], idx[-back]
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
print(f"step: {len(step)} dispatches, {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us from its first kernel to the next step's")
busy_until, idle = 0, 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = max(0, s - busy_until) if busy_until else 0
    idle += gap
    name = r["Kernel_Name"].replace("void dffw::", "").split("(")[0]
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print(f"{s / 1e3:8.2f} {e / 1e3:8.2f} {(e - s) / 1e3:6.2f}  q{r['Queue_Id']}  idle {gap / 1e3:5.2f}  {name[:66]:66s} {wgs}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wgs of {r['Workgroup_Size_X']}")
    busy_until = max(busy_until, e)
print(f"chip idle between kernels: {idle / 1e3:.1f} us")
