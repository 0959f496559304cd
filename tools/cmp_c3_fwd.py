"""Probe: the forward kernel of BASELINE configs[2] (conv_bf16.hip, bf16 copies or fp32 rounded on load) against the split
kernel's structure issuing ONE bf16 MFMA per term (variant build -DSP_ONE_TERM=1 via ONET_HIP_LIB: same rounding points as the
bf16 path), at the configuration's own batch.   B=256 N=5 ONET_HIP_LIB=.../libonet_hip_one.so python tools/cmp_c3_fwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
B = int(os.environ.get("B", "256")); N = int(os.environ.get("N", "5"))
def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
ops.SPLIT_F16 = False
tot = [0.0, 0.0, 0.0]
for ci, co, H in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda"); x16 = x.to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    bf, _ = ops.pack3x3_bf16(w); sb, _ = ops.pack3x3_split(w)
    out = torch.empty(B, co, H, H, device="cuda")
    t = [timeit(lambda: ops.conv3x3_bf16(None, bf, co, out=out, x16=x16)), timeit(lambda: ops.conv3x3_bf16(x, bf, co, out=out)),
         timeit(lambda: ops.conv3x3_split(x, sb, co, out=out))]
    for i in range(3): tot[i] += t[i]
    print(f"{ci:4d}->{co:4d} @{H:3d}^2 B={B}: bf16 copy {t[0]:7.3f} ms | fp32 rounded on load {t[1]:7.3f} | split structure {t[2]:7.3f}", flush=True)
    del x, x16, out
print("sum", *[f"{v:.3f}" for v in tot])
