"""HIP-event timing + accuracy (vs fp64) of the split-bf16 fp32 convolution against the fp32 Winograd F(4x4,3x3) kernel, layer by
layer (B images of the U-Net's level shapes).   B=64 N=10 python tools/time_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from onet_amd import ops

B = int(os.environ.get("B", "64"))
N = int(os.environ.get("N", "10"))


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


tot = [0.0, 0.0]
for ci, co, H in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (9 * ci)) ** 0.5
    qf, qd = ops.pack3x3_winograd4(w)
    sf, sd = ops.pack3x3_split(w)
    t4 = timeit(lambda: ops.conv3x3_winograd4(x, qf, co))
    ts = timeit(lambda: ops.conv3x3_split(x, sf, co))
    # accuracy on two images against fp64
    ref = F.conv2d(x[:2].double().cpu(), w.double().cpu(), None, 1, 1)
    sc = float(ref.abs().max())
    e4 = (ops.conv3x3_winograd4(x[:2], qf, co).double().cpu() - ref)
    es = (ops.conv3x3_split(x[:2], sf, co).double().cpu() - ref)
    fl = 2.0 * B * H * H * ci * co * 9 / 1e9
    tot[0] += t4
    tot[1] += ts
    print(f"{ci:5d}->{co:5d} @{H:3d}^2  F(4x4) {t4:7.3f} ms {fl / t4:6.1f} TF rms {float(e4.pow(2).mean().sqrt()) / sc:.2e} max {float(e4.abs().max()) / sc:.2e}"
          f" | split {ts:7.3f} ms {fl / ts:6.1f} TF (x3 issued: {3 * fl / ts / 2500e0 * 1:5.3f} of bf16 peak) rms {float(es.pow(2).mean().sqrt()) / sc:.2e} max {float(es.abs().max()) / sc:.2e}",
          flush=True)
print(f"sum F(4x4) {tot[0]:.3f} ms, split {tot[1]:.3f} ms")
