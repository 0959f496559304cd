import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
B = 32
ci, co, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 512, 32)
x = torch.relu(torch.randn(B, ci, H, H, device="cuda")); dz = torch.randn(B, co, H, H, device="cuda")
for _ in range(10):
    ops.conv3x3_winograd_wgrad(x, dz, (co, ci, 3, 3))
torch.cuda.synchronize()
print("done")
