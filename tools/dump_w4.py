"""Dump F(4x4) outputs (forward + fused statistics records) for a few shapes: bitwise comparison of two library builds."""
import sys, torch
from onet_amd import ops, _lib
dev = torch.device("cuda:0")
out = {}
for i, (B, Ci, Co, H, W) in enumerate([(2, 64, 64, 64, 64), (3, 16, 72, 16, 16), (2, 8, 128, 40, 48), (4, 128, 64, 32, 32)]):
    g = torch.Generator().manual_seed(i)
    x = torch.randn(B, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(dev)
    qf, qd = ops.pack3x3_winograd4(w)
    out[f"z{i}"] = ops.conv3x3_winograd4(x, qf, Co).cpu()
    n = int(_lib.load().onet_conv3x3_winograd4_nparts(B, H, W))
    if n:
        z = torch.empty(B, Co, H, W, device=dev)
        cm = torch.zeros(Co, n, 3, device=dev)
        _lib.call("onet_conv3x3_winograd4_fwd_stats", x.data_ptr(), Ci * H * W, qf.data_ptr(), z.data_ptr(), Co * H * W,
                  cm.data_ptr(), B, Ci, Co, H, W, torch.cuda.current_stream().cuda_stream)
        out[f"cm{i}"] = cm.cpu()
        out[f"zs{i}"] = z.cpu()
torch.save(out, sys.argv[1])
