import os, sys
sys.path.insert(0, '.')
import torch
from onet_amd import ops
B = 64; N = 10
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
tot = 0
for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64), (512, 256, 64), (256, 512, 32), (512, 512, 32), (1024, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    sf, sd = ops.pack3x3_split(w)
    out = torch.empty(B, co, H, H, device="cuda")
    t = timeit(lambda: ops.conv3x3_split(x, sf, co, out=out)); tot += t
    print(f"{ci:5d}->{co:5d} @{H:3d}^2 {t:7.3f} ms {2.0*B*H*H*ci*co*9/1e9/t:7.1f} TF", flush=True)
print(f"sum {tot:.3f}")
