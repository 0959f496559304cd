import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
B = 32
shapes = [(512, 512, 32), (64, 64, 256)] if len(sys.argv) < 4 else [tuple(int(v) for v in sys.argv[1:4])]
for ci, co, H in shapes:
    x = torch.randn(B, ci, H, H, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    qf, qd = ops.pack3x3_winograd4(w)
    for _ in range(10):
        ops.conv3x3_winograd4(x, qf, co)
    torch.cuda.synchronize()
print("done")
