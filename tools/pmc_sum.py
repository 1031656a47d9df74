"""Sum a rocprofv3 counter_collection.csv per kernel: python tools/pmc_sum.py <dir-or-csv> [substring]"""
import collections, csv, glob, os, sys
path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if flt in k:
            d[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
for k, cs in d.items():
    print(f"{k[:70]:70s} launches {len(n[k]):4d} " + "  ".join(f"{c} {v:.6g}" for c, v in sorted(cs.items())))
