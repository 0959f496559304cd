import sys, torch
sys.path.insert(0, '.')
from onet_amd import ops
import torch.nn.functional as F
torch.manual_seed(0)
for xs in (1e-4, 1e-2, 1.0, 100.0, 3e4):
    for ws in (1e-4, 0.05, 5.0):
        B, Cin, Cout, H, W = 2, 128, 64, 32, 64
        x = (torch.randn(B, Cin, H, W) * xs).clamp(-6e4, 6e4); w = torch.randn(Cout, Cin, 3, 3) * ws
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        qf, qd = ops.pack3x3_split(w.cuda())
        z = ops.conv3x3_split(x.cuda(), qf, Cout).cpu().double()
        ops.SPLIT_F16 = False
        qb, _ = ops.pack3x3_split(w.cuda())
        zb = ops.conv3x3_split(x.cuda(), qb, Cout).cpu().double()
        ops.SPLIT_F16 = True
        sc = float(ref.abs().max())
        print(f"x {xs:7.0e} w {ws:7.0e}: fp16 split rms {float((z-ref).pow(2).mean().sqrt())/sc:.2e} max {float((z-ref).abs().max())/sc:.2e} | bf16 split rms {float((zb-ref).pow(2).mean().sqrt())/sc:.2e} max {float((zb-ref).abs().max())/sc:.2e}  finite {bool(torch.isfinite(z).all())}")
