"""HBM traffic per launch of the conv kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected
separately, tools/round_measurements.sh), corrected as MI355X_MICROARCH.md (HBM section) prescribes: counters are
KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B -> x2.
Writes the JSON bench.py picks `roofline.traffic` from.

    python tools/hbm_traffic.py <pmc_fetch.csv> <pmc_write.csv> <bench.json> <out.json>
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k] += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
bench = json.load(open(sys.argv[3]))
dom = bench["roofline"]["kernel"]


def entry(k):
    f = 2.0 * fetch[k] / max(nf[k], 1)
    w = write[k] / max(nw[k], 1)
    return {"kernel": k, "launches_averaged": nf[k], "fetch_bytes_per_launch_corrected": f, "write_bytes_per_launch": w,
            "hbm_bytes_per_launch": f + w}


out = entry(dom)
out["method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1` (batch 32, "
                 "default precision); counters are KiB -> *1024; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16 B/lane "
                 "coalesced streams as 64 B, MI355X_MICROARCH.md HBM section); averaged over the kernel's launches (2 forwards)")
others = sorted((k for k in fetch if k != dom and "dffw::" in k), key=lambda k: -(fetch[k] + write[k]))[:40]
out["other_kernels"] = [entry(k) for k in others]
json.dump(out, open(sys.argv[4], "w"), indent=1)
alg = bench["roofline"]["algorithmic_gb_per_launch"] * 1e9
print(dom, "HBM bytes/launch", round(out["hbm_bytes_per_launch"]), "= %.2fx algorithmic" % (out["hbm_bytes_per_launch"] / alg))
