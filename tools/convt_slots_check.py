"""ConvTranspose2d GEMMs on slot operands (round 5) against fp64 and against the fp32-operand GEMMs: error and HIP-event time per decoder
level of the 256 x 256 U-Net.   python tools/convt_slots_check.py [B] [parts]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from onet_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(3)


def nchw(P):
    v = P.float().sum(3) if P.shape[3] == 2 else P[:, :, :, 0].float()
    return v.permute(0, 1, 4, 2, 3).reshape(P.shape[0], P.shape[1] * 8, P.shape[2], P.shape[4])


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = [0.0, 0.0]
with ops.using(ops.Settings(conv="bf16" if parts == 1 else "auto")):
    for Cin, h in ((1024, 16), (512, 32), (256, 64), (128, 128)):
        Ct = Cin // 2
        x = torch.relu(torch.randn(B, Cin, h, h, device=dev))
        w = (torch.rand(Cin, Ct, 2, 2, device=dev) * 2 - 1) / (Ct * 4) ** 0.5
        bias = (torch.rand(Ct, device=dev) * 2 - 1) / (Ct * 4) ** 0.5
        xr = x.to(torch.bfloat16).double() if parts == 1 else x.double()
        wr = w.to(torch.bfloat16).double() if parts == 1 else w.double()
        ref = F.conv_transpose2d(xr[:4].cpu(), wr.cpu(), bias.double().cpu(), stride=2)
        xP = ops.split_pack_act(x, f16=True, parts=parts)
        wP = ops.packT2x2_slots(w, parts)
        outP = ops.p16_empty(B, Ct, 2 * h, 2 * h, dev, parts=parts)
        ok = ops.convT2x2_fwd_slots(xP, wP, bias, outP, Ct)
        got = nchw(outP[:4]).double().cpu()
        e = float((got - ref).abs().max() / ref.abs().max())
        wq = ops.packT2x2_fused(w)
        outQ = ops.p16_empty(B, Ct, 2 * h, 2 * h, dev, parts=parts)
        ok2 = ops.convT2x2_fwd_p(x, wq, bias, outQ, Ct, 0, 0)
        e2 = float((nchw(outQ[:4]).double().cpu() - ref).abs().max() / ref.abs().max())
        t1 = timeit(lambda: ops.convT2x2_fwd_slots(xP, wP, bias, outP, Ct))
        t0 = timeit(lambda: ops.convT2x2_fwd_p(x, wq, bias, outQ, Ct, 0, 0))
        tot[0] += t0; tot[1] += t1
        fl = 2.0 * B * h * h * Cin * 4 * Ct * (3 if parts == 2 else 1)
        print(f"fwd {Cin:4d}->{Ct:3d} @{h:3d}^2 B={B}: slots taken {ok} err {e:.1e} {t1:.3f} ms ({fl / t1 / 1e9 / 2500:.3f} of peak) | fp32 operands taken {ok2} err {e2:.1e} {t0:.3f} ms", flush=True)
print(f"forward sum: fp32 operands {tot[0]:.3f} ms, slots {tot[1]:.3f} ms")
