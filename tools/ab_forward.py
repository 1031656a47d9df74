"""A/B of the WHOLE forward on one box: bench.py's timed + sustained region per variant, variants alternating, REPS rounds; prints every
sustained value and the per-variant median.  usage: ab_forward.py REPS [--seconds S] name=ENV1=V1,ENV2=V2 name2=...   (empty env list = default)
The chip runs this forward at ~95 % of its 1400 W socket limit (tools/power_probe.sh), so a kernel-level gain need not show up one to one."""
import json, os, statistics, subprocess, sys
args = sys.argv[1:]
reps = int(args.pop(0))
secs = "4"
if args and args[0] == "--seconds":
    args.pop(0); secs = args.pop(0)
variants = []
for spec in args:
    name, _, envs = spec.partition("=")
    variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
vals = {n: [] for n, _ in variants}
for r in range(reps):
    for name, env in variants:
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-configs", "--no-roofline", "--sustain-seconds", secs, "--steps", "10", "--warmup", "5"],
                             env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        vals[name].append(d["sustained"]["ms_per_step"])
for name, _ in variants:
    v = vals[name]
    print(f"{name:12s} ms/step sustained: " + " ".join(f"{x:.3f}" for x in v) + f"   median {statistics.median(v):.3f}  min {min(v):.3f}")
