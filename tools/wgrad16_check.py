"""conv3x3_wgrad_pre16_kernel (v_mfma_f32_16x16x32, K = 32 pixels) against the fp64 weight gradient -- every unit shape (64-pixel strips,
32-pixel maps = two images per unit, 16-pixel maps = four), both channel tilings (64 / 128 output channels per block), fp16 parts,
bf16 parts and plain bf16 -- then HIP-event time per U-Net layer shape (fp16 split and plain bf16).   python tools/wgrad16_check.py [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from onet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(3)
rb = lambda t: t.to(torch.bfloat16).double()
def ref_wgrad(x, dz):
    w = torch.zeros(dz.shape[1], x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.cpu(), w, None, 1, 1).backward(dz.cpu())
    return w.grad
bad = 0
for (b, ci, co, H, W) in [(2, 64, 64, 64, 64), (2, 64, 128, 32, 64), (3, 72, 64, 16, 96), (4, 64, 128, 32, 32), (2, 128, 64, 32, 32), (1, 64, 64, 8, 128),
                          (4, 64, 128, 16, 16), (8, 128, 64, 32, 16), (4, 256, 256, 16, 16), (2, 64, 256, 24, 128), (6, 96, 40, 32, 32)]:
    x = torch.randn(b, ci, H, W, device=dev); dz = torch.randn(b, co, H, W, device=dev)
    ref = ref_wgrad(x.double(), dz.double()); sc = float(ref.abs().max())
    line = f"B={b} {ci}->{co} {H}x{W}:"
    for f16, tol in ((True, 2e-6), (False, 3e-5)):
        got = ops.conv3x3_split_wgrad_pre(ops.split_pack_act(x, f16=f16), ops.split_pack_act(dz, f16=f16), (co, ci, 3, 3))
        err = float((got.double().cpu() - ref).abs().max()) / sc
        line += f" {'f16' if f16 else 'bf16'} {err:.1e}"
        bad += not (err < tol)
        again = ops.conv3x3_split_wgrad_pre(ops.split_pack_act(x, f16=f16), ops.split_pack_act(dz, f16=f16), (co, ci, 3, 3))
        bad += not torch.equal(got, again)
    refb = ref_wgrad(rb(x), rb(dz))
    got = ops.conv3x3_split_wgrad_pre(ops.split_pack_act(x, parts=1), ops.split_pack_act(dz, parts=1), (co, ci, 3, 3))
    err = float((got.double().cpu() - refb).abs().max()) / float(refb.abs().max())
    line += f" plain {err:.1e}"
    bad += not (err < 2e-6)
    print(line, flush=True)
print("FAILURES:", bad)
if len(sys.argv) > 1:
    N = 10; B = 64
    def timeit(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / N
    for mode in ("f16", "plain"):
        res = []
        for ci, co, H in [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64), (512, 256, 64), (256, 512, 32), (512, 512, 32), (1024, 512, 32), (512, 1024, 16), (1024, 1024, 16)]:
            x = torch.randn(B, ci, H, H, device=dev); dz = torch.randn(B, co, H, H, device=dev)
            xs, dzs = (ops.split_pack_act(x, f16=True), ops.split_pack_act(dz, f16=True)) if mode == "f16" else (ops.split_pack_act(x, parts=1), ops.split_pack_act(dz, parts=1))
            dw = torch.empty(co, ci, 3, 3, device=dev)
            res.append(timeit(lambda: ops.conv3x3_split_wgrad_pre(xs, dzs, (co, ci, 3, 3), out=dw)))
            del x, dz, xs, dzs
        print(os.path.basename(os.environ.get("ONET_HIP_LIB", "default")), "wgrad", mode, " ".join(f"{t:7.4f}" for t in res), f"sum {sum(res):.4f}", flush=True)
