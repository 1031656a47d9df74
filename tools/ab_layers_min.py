"""A/B of per-layer times on one box: runs bench.py's profiled forward several times per variant (alternating) and compares the
per-layer minimum.  usage: ab_layers_min.py REPS name=ENV1=V1,ENV2=V2 name2=...   (an empty env list = the default library)"""
import csv, os, subprocess, sys, tempfile
reps = int(sys.argv[1])
variants = []
for spec in sys.argv[2:]:
    name, _, envs = spec.partition("=")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    variants.append((name, env))
best = {n: {} for n, _ in variants}
order = []
for r in range(reps):
    for name, env in variants:
        with tempfile.NamedTemporaryFile(suffix=".tsv", delete=False) as f:
            path = f.name
        subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-configs", "--sustain-seconds", "0", "--steps", "3", "--warmup", "2",
                        "--dump-layers", path], env=dict(os.environ, **env), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        rows = list(csv.reader(open(path), delimiter="\t"))[1:]
        os.unlink(path)
        seen = {}
        for row in rows:
            seen[row[1]] = seen.get(row[1], 0) + 1
            key = (seen[row[1]], row[1])                     # layer name + its occurrence (the three pooling launches share a name)
            if key not in order:
                order.append(key)
            best[name][key] = min(best[name].get(key, 1e9), float(row[4]))
names = [n for n, _ in variants]
print("layer".ljust(46) + "".join(n.rjust(10) for n in names))
tot = {n: 0.0 for n in names}
for key in order:
    vals = [best[n].get(key, float("nan")) for n in names]
    for n, v in zip(names, vals):
        if v == v:
            tot[n] += v
    have = [v for v in vals if v == v]
    flag = " *" if len(have) > 1 and max(have) - min(have) > 0.03 * max(have) and max(have) > 0.02 else ""
    print(key[1][:45].ljust(46) + "".join(f"{v:10.4f}" for v in vals) + flag)
print("TOTAL".ljust(46) + "".join(f"{tot[n]:10.3f}" for n in names))
