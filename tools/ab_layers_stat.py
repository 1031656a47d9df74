"""A/B of selected layers on one box with run-to-run statistics: REPS profiled forwards per variant in ABBA order, per layer the median, minimum and
the quartiles of the per-run times.  usage: ab_layers_stat.py REPS 'substr1,substr2' name=ENV=V,... name2=...   (empty env list = the default library; extra bench.py
arguments, e.g. "--workload e2e", through the environment variable AB_BENCH_ARGS)"""
import csv, os, statistics, subprocess, sys, tempfile
reps = int(sys.argv[1])
keys = [k for k in sys.argv[2].split(",") if k]
variants = []
for spec in sys.argv[3:]:
    name, _, envs = spec.partition("=")
    variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
vals = {n: {} for n, _ in variants}
tot = {n: [] for n, _ in variants}
for r in range(reps):
    order = variants if r % 2 == 0 else variants[::-1]
    for name, env in order:
        with tempfile.NamedTemporaryFile(suffix=".tsv", delete=False) as f:
            path = f.name
        subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-configs", "--sustain-seconds", "0", "--steps", "3", "--warmup", "2",
                        "--dump-layers", path] + os.environ.get("AB_BENCH_ARGS", "").split(), env=dict(os.environ, **env), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        rows = list(csv.reader(open(path), delimiter="\t"))[1:]
        os.unlink(path)
        tot[name].append(sum(float(row[4]) for row in rows))
        per = {}
        for row in rows:   # a layer that runs as several launches ("... (lower output channels)") counts as their sum
            if any(k in row[1] for k in keys):
                base = row[1].split(" (")[0]
                per[base] = per.get(base, 0.0) + float(row[4])
        for base, v in per.items():
            vals[name].setdefault(base, []).append(v)
def q(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))]
print("layer".ljust(44) + "".join(f"{n + ' med':>12}{'min':>8}{'q25':>8}{'q75':>8}" for n, _ in variants))
layers = list(vals[variants[0][0]].keys())
for l in layers:
    print(l[:43].ljust(44) + "".join((f"{statistics.median(vals[n][l]):12.4f}{min(vals[n][l]):8.4f}{q(vals[n][l], .25):8.4f}{q(vals[n][l], .75):8.4f}" if l in vals[n] else " " * 36) for n, _ in variants))
print("TOTAL (all layers)".ljust(44) + "".join(f"{statistics.median(tot[n]):12.3f}{min(tot[n]):8.3f}{q(tot[n], .25):8.3f}{q(tot[n], .75):8.3f}" for n, _ in variants))
