"""One 3x3x3 stride-1 conv through the C ABI per input width (16 / 32 / 64 / 128 channels -> 32), for rocprofv3 --pmc runs that
compare the L1 -> L2 read requests of conv_tile's footprint staging with the bytes it stages (partial use of 128-byte lines by
16-channel stages of wide pixel records).  usage: fill_probe.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dffinthewild_amd import engine as eng
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, N, H, W = 16, 10, 64, 64
torch.manual_seed(0)
for cin in (16, 32, 64, 128):
    x = torch.randn(B, cin, N, H, W, device="cuda")
    w = torch.randn(32, cin, 3, 3, 3) * 0.05
    for _ in range(reps):
        y = eng.op_conv3d(x, w, pad=1, relu=1)
    torch.cuda.synchronize()
    print(cin, eng.last_conv_kernel(), float(y.abs().mean()))
