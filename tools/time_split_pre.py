"""Pre-split (LDS-DMA staged) forward kernel against the round-3 kernel that splits its fp32 input in staging: bit identity and
HIP-event time per U-Net layer shape.   B=64 N=10 python tools/time_split_pre.py"""
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
tot = [0, 0, 0]
for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64), (512, 256, 64), (256, 512, 32), (512, 512, 32), (1024, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    sf, sd = ops.pack3x3_split(w)
    out = torch.empty(B, co, H, H, device="cuda"); out2 = torch.empty_like(out)
    xs = ops.split_pack_act(x, f16=True)
    t0 = timeit(lambda: ops.conv3x3_split(x, sf, co, out=out))
    t1 = timeit(lambda: ops.conv3x3_split_pre(xs, sf, co, out=out2))
    tp = timeit(lambda: ops.split_pack_act(x, f16=True, out=xs))
    same = bool(torch.equal(out, out2))
    # dgrad arithmetic (bf16 parts)
    xb = ops.split_pack_act(x[:, :co] if co <= ci else x, f16=False)
    tot[0] += t0; tot[1] += t1; tot[2] += tp
    fl = 2.0 * B * H * H * ci * co * 9 / 1e9
    print(f"{ci:5d}->{co:5d} @{H:3d}^2  fp32-in {t0:7.3f} ms {fl/t0:7.1f} TF | pre-split {t1:7.3f} ms {fl/t1:7.1f} TF ({3*fl/t1/2500:5.3f} issued) "
          f"{t1/t0-1:+6.1%}  identical={same}  pack {tp:6.3f} ms {x.numel()*8/tp/1e6:6.0f} GB/s", flush=True)
print(f"sum fp32-in {tot[0]:.3f}  pre-split {tot[1]:.3f} ({tot[1]/tot[0]-1:+.1%})  pack {tot[2]:.3f}")
# bf16 parts + statistics epilogue, one shape
x = torch.randn(8, 64, 64, 64, device="cuda"); w = torch.randn(128, 64, 3, 3, device="cuda") * 0.05
sf, sd = ops.pack3x3_split(w)
import onet_amd.ops as o
keep = o.SPLIT_F16
o.SPLIT_F16 = False
sfb, _ = ops.pack3x3_split(w)
o.SPLIT_F16 = keep
z0 = ops.conv3x3_split(x, sfb, 128); z1 = ops.conv3x3_split_pre(ops.split_pack_act(x, f16=False), sfb, 128)
print("bf16 parts identical:", bool(torch.equal(z0, z1)))
