import sys, torch
sys.path.insert(0, ".")
from onet_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, ci, co, H, W) in [(64, 1024, 1024, 16, 16), (64, 1024, 1024, 20, 16), (64, 1024, 1024, 12, 16), (64, 992, 1024, 16, 16), (64, 1024, 1024, 8, 32), (64, 1024, 1024, 12, 32)]:
    x = torch.randn(B, ci, H, W, device=dev); g = torch.randn(B, co, H, W, device=dev)
    fl = 2.0 * B * H * W * ci * co * 9 / 1e9
    t4 = timeit(lambda: ops.conv3x3_winograd4_wgrad(x, g, (co, ci, 3, 3)))
    t2 = timeit(lambda: ops.conv3x3_winograd_wgrad(x, g, (co, ci, 3, 3)))
    print((B, ci, co, H, W), f"F(4x4) {t4:.3f} ms {fl/t4:.0f} TF | F(2x2) {t2:.3f} ms {fl/t2:.0f} TF")
