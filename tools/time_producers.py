"""HIP-event times of the pre-split producers against the fp32 passes they replace, at the B = 64 twin-batch shapes of the benchmark."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
dev = torch.device("cuda:0")
N = 10
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
tot = [0.0] * 6
for (B, C, H) in [(64, 64, 256), (64, 128, 128), (64, 256, 64), (64, 512, 32)]:
    z = torch.randn(B, C, H, H, device=dev); da = torch.randn(B, C, H, H, device=dev) * 1e-4
    save = torch.rand(4, C, device=dev) + 0.5
    a = torch.empty_like(z); xs = ops.p16_empty(B, C, H, H, dev); y = torch.empty(B, C, H // 2, H // 2, device=dev); ys = ops.p16_empty(B, C, H // 2, H // 2, dev)
    t = [timeit(lambda: ops.bn_relu_apply(z, save, out=a)), timeit(lambda: ops.bn_relu_apply_split(z, save, xs)),
         timeit(lambda: ops.bn_relu_apply_pool(z, save, a, None, y, None)), timeit(lambda: ops.bn_relu_apply_pool_split(z, save, xs, None, ys, None)),
         timeit(lambda: ops.bn_relu_bwd(da, z, save, True)), timeit(lambda: ops.bn_relu_bwd_split(da, z, save.view(1, 4, C), True))]
    for i in range(6): tot[i] += t[i]
    gb = z.numel() * 8 / 1e6
    print(f"{B}x{C}x{H}^2  apply {t[0]:.3f} / split {t[1]:.3f} ms ({gb/t[1]:.0f} GB/s) | apply+pool {t[2]:.3f} / split {t[3]:.3f} | bwd (reduce+apply) {t[4]:.3f} / split {t[5]:.3f}", flush=True)
print("sum  apply %.3f / %.3f | pool %.3f / %.3f | bwd %.3f / %.3f" % tuple(tot))
