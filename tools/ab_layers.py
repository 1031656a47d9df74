"""Compare two per-launch layer tables written by bench.py --dump-layers: per layer (matched by name) ms, totals."""
import collections
import csv
import sys


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path), delimiter="\t"):
        key = r["layer"] or r["kernel"]
        k, i = key, 1
        while k in d:
            i += 1
            k = f"{key}#{i}"
        d[k] = (float(r["ms"]), r["kernel"][6:])
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
for k in list(a) + [k for k in b if k not in a]:
    ma, ka = a.get(k, (0.0, "-"))
    mb, kb = b.get(k, (0.0, "-"))
    if abs(ma - mb) > 0.02 * max(ma, mb) + 0.004:
        print(f"{ma:8.3f} -> {mb:8.3f}  {k:52s} {ka}" + (f" -> {kb}" if kb != ka else ""))
print(f"total {sum(v[0] for v in a.values()):.3f} -> {sum(v[0] for v in b.values()):.3f}")
