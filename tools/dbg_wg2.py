import sys, torch
sys.path.insert(0, ".")
from onet_amd import ops
dev = torch.device("cuda:0")
B, Ci, Co, H, W = 1, 64, 64, 4, 16
x = torch.ones(B, Ci, H, W); g = torch.ones(B, Co, H, W)
dw = ops.conv3x3_wgrad_bf16(x.to(dev), g.to(dev), (Co, Ci, 3, 3)).cpu()
print("ones:", dw[0, 0].tolist(), "expect [[45,48,45],[60,64,60],[45,48,45]]")
print("ones co=37 ci=50:", dw[37, 50].tolist())
x = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, Ci, H, W).contiguous()
for (py, px) in [(1, 5), (2, 8), (0, 0), (3, 15)]:
    g = torch.zeros(B, Co, H, W); g[:, :, py, px] = 1
    dw = ops.conv3x3_wgrad_bf16(x.to(dev), g.to(dev), (Co, Ci, 3, 3)).cpu()
    print("delta at", (py, px), dw[0, 0].tolist())
x = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1).expand(B, Ci, H, W).contiguous() + 1
g = torch.zeros(B, Co, H, W); g[:, :, 1, 5] = 1
dw = ops.conv3x3_wgrad_bf16(x.to(dev), g.to(dev), (Co, Ci, 3, 3)).cpu()
print("rows: delta (1,5)", dw[0, 0].tolist(), "expect rows [1,1,1],[2,2,2],[3,3,3]")
