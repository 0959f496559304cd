"""What the BatchNorm-statistics epilogue costs the pre-split forward kernel: HIP-event time with and without it per layer shape.
   B=64 N=10 python tools/time_split_pre_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops, _lib
B = int(os.environ.get("B", "64")); N = int(os.environ.get("N", "10"))
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 256, 64), (512, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    sf, sd = ops.pack3x3_split(w)
    out = torch.empty(B, co, H, H, device="cuda")
    xs = ops.split_pack_act(x, f16=True)
    nparts = int(_lib.load().onet_conv3x3_split_pre_nparts(B, H, H))
    cm = torch.empty((co, nparts, 3), device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: ops.conv3x3_split_pre(xs, sf, co, out=out))
        t1 = timeit(lambda: ops.conv3x3_split_pre(xs, sf, co, out=out, stats=cm))
        print(f"{ci:4d}->{co:4d} @{H:3d}^2  plain {t0:.4f} ms   with statistics {t1:.4f} ms  ({t1 / t0 - 1:+.1%})", flush=True)
    del x, out, xs
