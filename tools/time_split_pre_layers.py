"""HIP-event time of the pre-split forward kernel per U-Net layer shape (ONET_HIP_LIB selects a variant build: same-box A/B).
   B=64 N=10 python tools/time_split_pre_layers.py"""
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
res = []
for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64), (512, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    sf, sd = ops.pack3x3_split(w)
    out = torch.empty(B, co, H, H, device="cuda")
    xs = ops.split_pack_act(x, f16=True)
    res.append(timeit(lambda: ops.conv3x3_split_pre(xs, sf, co, out=out)))
    del x, out, xs
print(os.path.basename(os.environ.get("ONET_HIP_LIB", "default")), " ".join(f"{t:7.4f}" for t in res), f"sum {sum(res):.4f}", flush=True)
