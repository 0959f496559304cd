import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
B, ci, co, H = 32, 512, 512, 32
x = torch.randn(B, ci, H, H, device="cuda")
w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
qf, qd = ops.pack3x3_winograd(w)
for _ in range(10):
    ops.conv3x3_winograd(x, qf, co)
torch.cuda.synchronize()
print("done")
