"""HIP-event timing of the fp32 Winograd F(4x4,3x3) forward kernel, layer by layer (B images of the U-Net's level shapes).
ONET_HIP_LIB selects a variant build.   B=64 N=10 python tools/time_wino4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops

B = int(os.environ.get("B", "64"))
N = int(os.environ.get("N", "10"))
tot = 0.0
for ci, co, H in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    qf, qd = ops.pack3x3_winograd4(w)
    fn = lambda: ops.conv3x3_winograd4(x, qf, co)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / N
    tot += ms
    fl = 2.0 * B * H * H * ci * co * 9 / 1e9
    print(f"{ci:5d}->{co:5d} @{H:3d}^2  fwd {ms:7.3f} ms {fl / ms:7.1f} TF (direct-equivalent)", flush=True)
print(f"sum {tot:.3f} ms")
