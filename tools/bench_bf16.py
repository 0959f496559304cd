"""Per-layer timing of the bf16-operand conv kernel next to the fp32 F(4x4) kernel (forward orientation)."""
import sys, torch
sys.path.insert(0, ".")
from onet_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layers = [("inc.c2", 64, 64, 256), ("d1.c2", 128, 128, 128), ("d2.c2", 256, 256, 64), ("d3.c2", 512, 512, 32),
          ("u1.c1", 1024, 512, 32), ("u3.c1", 256, 128, 128), ("u4.c1", 128, 64, 256)]
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, ci, co, h in layers:
    x = torch.randn(B, ci, h, h, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    qf, _ = ops.pack3x3_bf16(w); q4, _ = ops.pack3x3_winograd4(w)
    fl = 2.0 * B * h * h * ci * co * 9 / 1e9
    tb = timeit(lambda: ops.conv3x3_bf16(x, qf, co)); t4 = timeit(lambda: ops.conv3x3_winograd4(x, q4, co))
    print(f"{name:8s} {ci:5d}->{co:5d} @{h:3d}  bf16 {tb:6.3f} ms {fl/tb:6.0f} TF | fp32 F(4x4) {t4:6.3f} ms {fl/t4:6.0f} TF")
print("weight gradient")
for name, ci, co, h in layers:
    x = torch.randn(B, ci, h, h, device=dev); g = torch.randn(B, co, h, h, device=dev)
    fl = 2.0 * B * h * h * ci * co * 9 / 1e9
    tb = timeit(lambda: ops.conv3x3_wgrad_bf16(x, g, (co, ci, 3, 3))); t4 = timeit(lambda: ops.conv3x3_winograd_wgrad(x, g, (co, ci, 3, 3)))
    print(f"{name:8s} {ci:5d}->{co:5d} @{h:3d}  bf16 {tb:6.3f} ms {fl/tb:6.0f} TF | fp32 F(2x2) {t4:6.3f} ms {fl/t4:6.0f} TF")
