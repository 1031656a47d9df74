"""pred3 rel-L2 of every precision mode against the committed goldens (run on the GPU box)."""
import glob, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_gpu_forward import GOLDEN, case, model_for   # noqa: E402
from oracle import cpu_ref                               # noqa: E402  (checker only)

for path in GOLDEN:
    g, meta, FS, fd, sd = case(path)
    row = [os.path.basename(path)]
    for prec in ("bf16x3", "fp16", "bf16"):
        model = model_for(sd, (meta["wseed"], meta["profile"]), prec)
        with torch.no_grad():
            outs = model(FS.cuda(), fd.cuda())
        row.append("%s %.2e" % (prec, cpu_ref.rel_l2(outs[3].cpu(), g["pred3"])))
    print("  ".join(row))
