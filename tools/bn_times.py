"""Streaming-kernel timing on the U-Net's largest activation: BatchNorm apply / backward kernels, GB/s."""
import sys, torch
sys.path.insert(0, ".")
from onet_amd import ops
dev = torch.device("cuda:0")
B, C, H = 32, 64, 256
z = torch.randn(B, C, H, H, device=dev); da = torch.randn(B, C, H, H, device=dev)
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
save = ops.bn_train_coeffs(z, g, b, torch.zeros(C, device=dev), torch.ones(C, device=dev), 0.1, 1e-5)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = z.numel() * 4 / 1e9
out = torch.empty_like(z)
t = timeit(lambda: ops.bn_relu_apply(z, save, out=out)); print(f"bn_relu_apply      {t*1e3:7.1f} us  {2*N/t*1e3:6.0f} GB/s")
t = timeit(lambda: ops.bn_relu_bwd(da, z, save, True, out=out)); print(f"bn_relu_bwd (r+a)  {t*1e3:7.1f} us  {5*N/t*1e3:6.0f} GB/s")
t = timeit(lambda: ops.bn_train_coeffs(z, g, b, None, None, 0.1, 1e-5)); print(f"bn_stats+finalize  {t*1e3:7.1f} us  {N/t*1e3:6.0f} GB/s")
