"""Single-layer probe for the K-split rolling-window kernel (dffw_conv_rollk.hip): runs the three shapes it serves in the batch-32 forward through
dffw_op_conv3d a few times, so that `rocprofv3 --kernel-trace --stats` / `--pmc ...` of THIS script give the kernel's own duration and counters in
seconds instead of a whole-forward profile.  usage: python tools/rollk_probe.py [reps]   (DFFW_NO_ROLLK=1: conv_tile on the same shapes)"""
import sys
import torch
from dffinthewild_amd import engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [("dres2.conv0", 32, 64, 32, 10, 64, 64), ("dres3.conv2", 32, 32, 32, 10, 64, 64), ("dres0.2", 32, 64, 64, 10, 32, 32)]
g = torch.Generator().manual_seed(0)
for name, B, cin, cout, N, H, W in shapes:
    x = (torch.rand(B, cin, N, H, W, generator=g) * 2 - 1).cuda()
    w = (torch.rand(cout, cin, 3, 3, 3, generator=g) * 2 - 1) * (2.0 / (cin * 27)) ** 0.5
    bn = (0.5 + torch.rand(cout, generator=g), torch.rand(cout, generator=g) - 0.5, torch.rand(cout, generator=g) - 0.5, 0.5 + torch.rand(cout, generator=g))
    for _ in range(reps):
        y = engine.op_conv3d(x, w, pad=1, bn=bn, relu=1, precision="bf16x3")
    torch.cuda.synchronize()
    print(name, engine.last_conv_kernel(), float(y.abs().mean()))
