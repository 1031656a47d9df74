#!/bin/bash
# per-layer times of selected layers with parts of the conv kernels switched off (DFFW_DEBUG_FLAGS: 1 no fill, 2 no MFMA loop, 4 no stores)
# usage: tools/ablate_layers.sh "<grep -E pattern>" ["flag list"]
pat=${1:-"deconv_|conv5|conv6"}
flags=${2:-"0 2 4 6"}
for f in $flags; do
  DFFW_NO_ROLL=${DFFW_NO_ROLL:-0} DFFW_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 2 --warmup 1 --dump-layers gpurun_out/abl_$f.tsv > /dev/null 2>&1
done
python - "$pat" "$flags" <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[1])
flags = [int(f) for f in sys.argv[2].split()]
tabs = {}
for f in flags:
    try:
        tabs[f] = {r[1]: float(r[4]) for r in list(csv.reader(open(f"gpurun_out/abl_{f}.tsv"), delimiter="\t"))[1:]}
    except Exception as e:
        tabs[f] = {}
print("layer".ljust(40), *[f"f={f}".rjust(8) for f in tabs])
for k in tabs[flags[0]]:
    if pat.search(k):
        print(k[:40].ljust(40), *[("%.3f" % tabs[f][k]).rjust(8) if k in tabs[f] else "     n/a" for f in tabs])
PY
