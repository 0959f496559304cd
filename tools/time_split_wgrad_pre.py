"""Pre-split weight-gradient kernel (LDS-DMA staging, ds_read_b64_tr_b16 fragments) against the round-3 kernel that splits fp32
operands in staging: error against fp64 and HIP-event time per U-Net layer shape.   B=64 N=10 python tools/time_split_wgrad_pre.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from onet_amd import ops
B = int(os.environ.get("B", "64")); N = int(os.environ.get("N", "10"))
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
def ref_wgrad(x, dz):
    x = x.double().cpu().requires_grad_(False); dz = dz.double().cpu()
    w = torch.zeros(dz.shape[1], x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dz)
    return w.grad
# correctness on small shapes (both 16-bit types; W = 32 pairs images)
for (b, ci, co, H, W) in [(2, 64, 64, 64, 64), (2, 64, 128, 32, 64), (3, 72, 64, 16, 96), (4, 64, 128, 32, 32), (2, 128, 64, 32, 32), (1, 64, 64, 8, 128)]:
    x = torch.randn(b, ci, H, W, device="cuda"); dz = torch.randn(b, co, H, W, device="cuda")
    ref = ref_wgrad(x, dz); sc = float(ref.abs().max())
    for f16 in (True, False):
        got = ops.conv3x3_split_wgrad_pre(ops.split_pack_act(x, f16=f16), ops.split_pack_act(dz, f16=f16), (co, ci, 3, 3))
        err = float((got.double().cpu() - ref).abs().max()) / sc
        print(f"check B={b} {ci}->{co} {H}x{W} {'f16' if f16 else 'bf16'}: max err {err:.2e} of scale", flush=True)
        assert err < (2e-6 if f16 else 3e-5), err
    old = ops.conv3x3_split_wgrad(x, dz, (co, ci, 3, 3)) if ops.split_wgrad_ok(x, dz) else None
    if old is not None:
        print(f"      round-3 kernel (bf16 parts): {float((old.double().cpu() - ref).abs().max()) / sc:.2e}")
tot = [0, 0, 0]
for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64), (512, 256, 64), (256, 512, 32), (512, 512, 32), (1024, 512, 32)]:
    x = torch.randn(B, ci, H, H, device="cuda"); dz = torch.randn(B, co, H, H, device="cuda")
    xs = ops.split_pack_act(x, f16=True); dzs = ops.split_pack_act(dz, f16=True)
    dw = torch.empty(co, ci, 3, 3, device="cuda"); dw2 = torch.empty_like(dw)
    t0 = timeit(lambda: ops.conv3x3_split_wgrad(x, dz, (co, ci, 3, 3), out=dw))
    t1 = timeit(lambda: ops.conv3x3_split_wgrad_pre(xs, dzs, (co, ci, 3, 3), out=dw2))
    rel = float((dw - dw2).abs().max() / dw.abs().max())
    tot[0] += t0; tot[1] += t1
    fl = 2.0 * B * H * H * ci * co * 9 / 1e9
    print(f"{ci:5d}->{co:5d} @{H:3d}^2  fp32-in {t0:7.3f} ms {fl/t0:7.1f} TF | pre-split {t1:7.3f} ms {fl/t1:7.1f} TF ({3*fl/t1/2500:5.3f} issued) "
          f"{t1/t0-1:+6.1%}  diff {rel:.1e}", flush=True)
print(f"sum fp32-in {tot[0]:.3f}  pre-split {tot[1]:.3f} ({tot[1]/tot[0]-1:+.1%})")
