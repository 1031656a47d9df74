"""Timeline of one forward from a rocprofv3 --kernel-trace CSV: start offset, duration and kernel of every launch of the LAST complete
forward in the trace, the union-busy time, and per kernel the time during which it was the ONLY kernel running.
usage: timeline.py kernel_trace.csv [min_launches=55]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
minl = int(sys.argv[2]) if len(sys.argv) > 2 else 55
ts = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
groups, cur = [], [ts[0]]
for a, b in zip(ts, ts[1:]):
    if b[0] - max(x[1] for x in cur) > 100000:
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
gs = [g for g in groups if len(g) >= minl]
g = gs[-2] if len(gs) > 1 else gs[-1]
t0 = g[0][0]; tend = max(x[1] for x in g)
iv = sorted((a, b) for a, b, _ in g); busy = 0; ce = iv[0][0]
for a, b in iv:
    if b > ce: busy += b - max(a, ce); ce = b
print(f"launches {len(g)}  span {(tend-t0)/1e3:.1f} us  union busy {busy/1e3:.1f}  sum of durations {sum(b-a for a,b,_ in g)/1e3:.1f}")
for a, b, n in g:
    conc = [m for (x, y, m) in g if x < b and y > a and (x, y, m) != (a, b, n)]
    print(f"{(a-t0)/1e3:9.1f} {(b-a)/1e3:8.1f}  {n[:90]}  || {len(conc)}")
