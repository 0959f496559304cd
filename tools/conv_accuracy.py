"""Forward error of the conv kernels against fp64 on a realistic deep layer (post-ReLU-like inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from onet_amd import ops
torch.manual_seed(0)
for (B, ci, co, H) in ((4, 64, 64, 64), (4, 256, 256, 32), (4, 1024, 1024, 16)):
    x = torch.relu(torch.randn(B, ci, H, H)) ; w = torch.randn(co, ci, 3, 3) * (2.0 / (ci * 9)) ** 0.5
    z64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    zc = F.conv2d(x, w, None, 1, 1)
    xd, wd = x.cuda(), w.cuda()
    zd = ops.conv_fwd(xd, ops.pack3x3(wd)[0], co, 3).cpu()
    zw = ops.conv3x3_winograd4(xd, ops.pack3x3_winograd4(wd)[0], co).cpu()
    rel = lambda a: float((a.double() - z64).norm() / z64.norm())
    print(f"Cin {ci:4d} @{H:3d}: cpu32 {rel(zc):.2e}  direct(two-level) {rel(zd):.2e}  winograd F(4x4) {rel(zw):.2e}")
