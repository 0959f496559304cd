"""HIP-event time of the BatchNorm passes that read the conv output z, with z stored as fp32 and as bf16 (plain-bf16 slots, np = 1):
backward apply (da, z -> dz slots), forward apply (z -> a slots), backward reduce.   python tools/time_z16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
st = torch.cuda.current_stream().cuda_stream
for (B, C, H) in [(256, 64, 256), (256, 128, 128), (256, 256, 64)]:
    zf = torch.randn(B, C, H, H, device=dev); da = torch.randn(B, C, H, H, device=dev) * 1e-3
    zb = zf.to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    save = ops.bn_train_coeffs(zf, gamma, beta, None, None, 0.1, 1e-5).view(1, 4, C).contiguous()
    coef = torch.zeros(1, 4, C, device=dev)
    dzP = ops.p16_empty(B, C, H, H, dev, 1); aP = ops.p16_empty(B, C, H, H, dev, 1)
    n = zf.numel()
    line = f"B={B} C={C} {H}x{H}:"
    for name, z in (("fp32", zf), ("bf16", zb)):
        zs = z.element_size()
        t = timeit(lambda: lib.onet_bn_relu_bwd_apply_split(da.data_ptr(), C * H * H, z.data_ptr(), int(zs == 2), C * H * H, save.data_ptr(), coef.data_ptr(),
                                                            dzP.data_ptr(), ops._pbs(dzP), None, 1, 0, B, C, H, H, st))
        line += f"  bwd_apply[{name}] {t:.3f} ms {n * (4 + zs + 2) / t / 1e9:.0f} GB/s"
        t = timeit(lambda: lib.onet_bn_relu_apply_split(z.data_ptr(), int(zs == 2), C * H * H, aP.data_ptr(), ops._pbs(aP), None, 0, save.data_ptr(), None, 1, 0,
                                                        B, C, H, H, st))
        line += f"  apply[{name}] {t:.3f} ms {n * (zs + 2) / t / 1e9:.0f} GB/s"
    # the default path's form: fp32 z, fp16 (hi | mid) slots scaled by a bound
    dzP2 = ops.p16_empty(B, C, H, H, dev, 2)
    bound = ops.new_amax(dev); bound.view(torch.float32)[::32] = 1e-2
    t = timeit(lambda: lib.onet_bn_relu_bwd_apply_split(da.data_ptr(), C * H * H, zf.data_ptr(), 0, C * H * H, save.data_ptr(), coef.data_ptr(),
                                                        dzP2.data_ptr(), ops._pbs(dzP2), bound.data_ptr(), 2, 0, B, C, H, H, st))
    line += f"  bwd_apply[fp32 z, hi|mid] {t:.3f} ms {n * 12 / t / 1e6:.0f} GB/s"
    print(os.path.basename(os.environ.get("ONET_HIP_LIB", "default")), line, flush=True)
