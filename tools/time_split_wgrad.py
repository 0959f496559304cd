"""HIP-event timing of the split-bf16 weight gradient against the fp32 Winograd weight-gradient kernels, layer by layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
B = int(os.environ.get("B", "64")); N = int(os.environ.get("N", "10"))
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
tot = [0.0, 0.0]
for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64), (512, 256, 64),
                  (256, 512, 32), (512, 512, 32), (1024, 512, 32), (512, 1024, 16), (1024, 1024, 16)]:
    if H < int(os.environ.get("MINW", "16")):
        continue
    x = torch.randn(B, ci, H, H, device="cuda"); g = torch.randn(B, co, H, H, device="cuda")
    shp = (co, ci, 3, 3)
    tw = timeit(lambda: ops.conv3x3_winograd4_wgrad(x, g, shp) if ops.winograd4_wgrad_ok(x, g) else ops.conv_wgrad(x, g, shp, 3))
    ts = timeit(lambda: ops.conv3x3_split_wgrad(x, g, shp))
    fl = 2.0 * B * H * H * ci * co * 9 / 1e9
    tot[0] += tw; tot[1] += ts
    print(f"{ci:5d}->{co:5d} @{H:3d}^2  winograd wgrad {tw:7.3f} ms {fl / tw:6.1f} TF | split {ts:7.3f} ms {fl / ts:6.1f} TF (x3 issued: {3 * fl / ts / 2500:5.3f} of bf16 peak)", flush=True)
print(f"sum winograd {tot[0]:.3f} ms, split {tot[1]:.3f} ms")
