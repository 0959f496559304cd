import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from onet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, Cin, Cout, H, W = 1, 64, 64, 16, 64
rb = lambda t: t.to(torch.bfloat16).float()
x = rb(torch.randn(B, Cin, H, W, device=dev)); g = rb(torch.randn(B, Cout, H, W, device=dev))
wz = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
F.conv2d(x.double().cpu(), wz, None, 1, 1).backward(g.double().cpu())
ref = wz.grad
xP1, gP1 = ops.split_pack_act(x, parts=1), ops.split_pack_act(g, parts=1)
xP2, gP2 = ops.split_pack_act(x, f16=False), ops.split_pack_act(g, f16=False)      # bf16 hi | mid (mid = 0: x is bf16-exact)
d1 = ops.conv3x3_split_wgrad_pre(xP1, gP1, (Cout, Cin, 3, 3)).double().cpu()
d2 = ops.conv3x3_split_wgrad_pre(xP2, gP2, (Cout, Cin, 3, 3)).double().cpu()
print("2-part err", float((d2 - ref).abs().max() / ref.abs().max()), " plain err", float((d1 - ref).abs().max() / ref.abs().max()))
e = (d1 - ref).abs()
print("err by tap", e.amax(dim=(0, 1)).tolist())
print("err by co block of 8", e.amax(dim=(1, 2, 3)).view(-1, 8).amax(1).tolist())
print("err by ci block of 8", e.amax(dim=(0, 2, 3)).view(-1, 8).amax(1).tolist())
print("ratio d1/ref sample", (d1[0, 0] / ref[0, 0]).tolist())
# delta tests: x = delta at (c=3, y=5, x=7); g = delta at (c=2, y=5, x=7) -> dw[2,3,1,1] = 1
x0 = torch.zeros_like(x); g0 = torch.zeros_like(g); x0[0, 3, 5, 7] = 1; g0[0, 2, 5, 7] = 1
dd = ops.conv3x3_split_wgrad_pre(ops.split_pack_act(x0, parts=1), ops.split_pack_act(g0, parts=1), (Cout, Cin, 3, 3))
nz = dd.nonzero().tolist()
print("delta nonzeros", nz[:10], [float(dd[tuple(i)]) for i in nz[:10]])
