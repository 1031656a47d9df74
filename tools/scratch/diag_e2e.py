import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_e2e as T
from dffinthewild_amd import synth
from oracle import cpu_ref
g, sd, FS1, fd1, fov1 = T.load(T.BIG[0])
B, H, W = 8, int(g["H"]), int(g["W"])
FS = torch.from_numpy(synth.focal_stack(B, 10, H, W, seed=555))
FS[1] = FS1[0]; FS[6] = FS1[0]
fov = fov1.expand(B, -1, -1, -1, -1).clone()
fd = fd1.expand(B, -1, -1, -1).contiguous()
m = T._model(sd)
FSd, fdd, fovd = FS.cuda(), fd.cuda(), fov.cuda()
def run(env):
    for k, v in env.items(): os.environ[k] = v
    errs = []
    for rep in range(4):
        with torch.no_grad():
            outs = m(FSd, fdd, fovd)
        torch.cuda.synchronize()
        e1 = cpu_ref.rel_l2(outs[3][1:2].cpu(), g["pred3"]); e6 = cpu_ref.rel_l2(outs[3][6:7].cpu(), g["pred3"])
        errs.append("%.2e/%.2e%s" % (e1, e6, "" if torch.equal(outs[3][1], outs[3][6]) else "!"))
    for k in env: del os.environ[k]
    print(env, errs, flush=True)
run({})
