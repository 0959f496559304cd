import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, Cin, h, w = 2, 128, 16, 32
Ct = Cin // 2
x = torch.randn(B, Cin, h, w, device=dev); wt = torch.randn(Cin, Ct, 2, 2, device=dev) * 0.05; bias = torch.randn(Ct, device=dev)
wf = ops.packT2x2_fused(wt)
ref = torch.empty(B, Ct, 2 * h, 2 * w, device=dev)
ops.convT2x2_fwd(x, wf, bias, ref, Ct, 0, 0)
P = ops.p16_empty(B, Ct, 2 * h, 2 * w, dev); P.zero_()
print(ops.convT2x2_fwd_p(x, wf, bias, P, Ct, 0, 0))
R = ops.split_pack_act(ref, f16=True)          # [B, C8, H, 2, W, 8]
bad = (P != R)
print("by part", bad.sum(dim=(0, 1, 2, 4, 5)).tolist())
print("by row parity", bad[:, :, 0::2].sum().item(), bad[:, :, 1::2].sum().item())
print("by col parity", bad[:, :, :, :, 0::2].sum().item(), bad[:, :, :, :, 1::2].sum().item())
print("by channel in slot", bad.sum(dim=(0, 1, 2, 3, 4)).tolist())
print("by c8", bad.sum(dim=(0, 2, 3, 4, 5)).tolist())
# decode values: hi + mid
val = P[:, :, :, 0].float() + P[:, :, :, 1].float()     # [B, C8, H, W, 8]
val = val.permute(0, 1, 4, 2, 3).reshape(B, Ct, 2 * h, 2 * w)
print("decoded max err vs ref", float((val - ref).abs().max()), "ref scale", float(ref.abs().max()))
# is it a permutation?  compare sorted
print("sorted equal-ish:", float((val.flatten().sort().values - ref.flatten().sort().values).abs().max()))
for b_, c_, y_, x_ in [(0, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1), (0, 1, 1, 1)]:
    v = float(val[b_, c_, y_, x_])
    loc = (ref - v).abs().flatten().argmin().item()
    print((b_, c_, y_, x_), "got", v, "want", float(ref[b_, c_, y_, x_]), "found at", (loc // (Ct * 4 * h * w), (loc // (4 * h * w)) % Ct, (loc // (2 * w)) % (2 * h), loc % (2 * w)))
