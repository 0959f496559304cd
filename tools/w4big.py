"""debug: F(4x4) kernel on the 256x256 shapes of the model against the direct fp32 kernel (onet_conv_fwd)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
torch.manual_seed(0)
for (B, ci, co, H, W) in [(8, 64, 64, 256, 256), (8, 128, 64, 256, 256), (8, 64, 128, 128, 128), (8, 128, 128, 128, 128), (2, 64, 64, 256, 256), (8, 64, 64, 64, 256), (8, 64, 64, 256, 64)]:
    x = torch.randn(B, ci, H, W, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (ci * 9)) ** 0.5
    qf, qd = ops.pack3x3_winograd4(w)
    pf, pd = ops.pack3x3(w)
    ref = ops.conv_fwd(x, pf, co, 3)
    for rep in range(12):
        z = ops.conv3x3_winograd4(x, qf, co)
        err = (z - ref).abs()
        bad = (err > 1e-3 * ref.abs().max()).nonzero()
        if bad.shape[0] or rep == 0: print(B, ci, co, H, W, "rep", rep, "max err", float(err.max()), "scale", float(ref.abs().max()), "bad elems", bad.shape[0],
              bad[:3].tolist() if bad.shape[0] else "")
