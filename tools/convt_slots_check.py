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

# ---------------------------------------------------------------- backward GEMMs on slot operands (fp16 parts)
if parts == 2:
    import math
    print("backward:")
    tb = [0.0, 0.0, 0.0, 0.0]
    for Cin, h in ((1024, 16), (512, 32), (256, 64), (128, 128)):
        Ct = Cin // 2
        x = torch.relu(torch.randn(B, Cin, h, h, device=dev))
        w = (torch.rand(Cin, Ct, 2, 2, device=dev) * 2 - 1) / (Ct * 4) ** 0.5
        dy = torch.randn(B, Ct, 2 * h, 2 * h, device=dev) * 3e-5
        nb = min(B, 2)
        xr = x[:nb].double().cpu().requires_grad_(True)
        wr = w.double().cpu().requires_grad_(True)
        (F.conv_transpose2d(xr, wr, None, stride=2) * dy[:nb].double().cpu()).sum().backward()
        dy_slots = ops.absmax_slots(dy)
        k = 13 - math.floor(math.log2(float(dy.abs().max())))
        dyP = ops.split_pack_act(dy, f16=True, scale=2.0 ** k)
        xP = ops.split_pack_act(x, f16=True)
        wdP = ops.packT2x2_slots(w, dgrad=True)[1]
        dx = ops.convT2x2_dgrad_slots(dyP, wdP, Cin, dy_slots=dy_slots)
        e1 = float((dx[:nb].double().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
        wd = ops.packT2x2(w)[1]
        dx0 = ops.convT2x2_dgrad(dy, wd, Cin, h, h, 0, 0)
        e0 = float((dx0[:nb].double().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
        t1 = timeit(lambda: ops.convT2x2_dgrad_slots(dyP, wdP, Cin, dy_slots=dy_slots))
        t0 = timeit(lambda: ops.convT2x2_dgrad(dy, wd, Cin, h, h, 0, 0))
        # weight gradient over the whole batch against fp64 on the first images only when B is small: use a B-image reference on the CPU for B <= 8
        got = ops.convT2x2_wgrad_slots(xP, dyP, (Cin, Ct, 2, 2), dy_slots=dy_slots, want_dbias=True)
        ref_dw = torch.einsum("bcij,bdiqjr->cdqr", x.double(), dy.double().view(B, Ct, h, 2, h, 2)).cpu()
        ref_db = dy.double().sum((0, 2, 3)).cpu()
        ew = float((got[0].double().cpu() - ref_dw).abs().max() / ref_dw.abs().max())
        eb = float((got[1].double().cpu() - ref_db).abs().max() / dy.double().abs().sum((0, 2, 3)).max().cpu())
        dw0, db0 = ops.convT2x2_wgrad(x, dy, (Cin, Ct, 2, 2), 0, 0, want_dbias=True)
        ew0 = float((dw0.double().cpu() - ref_dw).abs().max() / ref_dw.abs().max())
        t3 = timeit(lambda: ops.convT2x2_wgrad_slots(xP, dyP, (Cin, Ct, 2, 2), dy_slots=dy_slots, want_dbias=True))
        t2 = timeit(lambda: ops.convT2x2_wgrad(x, dy, (Cin, Ct, 2, 2), 0, 0, want_dbias=True))
        tb[0] += t0; tb[1] += t1; tb[2] += t2; tb[3] += t3
        print(f"bwd {Cin:4d}->{Ct:3d} @{h:3d}^2 B={B}: dgrad slots err {e1:.1e} {t1:.3f} ms | fp32 operands err {e0:.1e} {t0:.3f} ms || wgrad slots err {ew:.1e} dbias {eb:.1e} {t3:.3f} ms | fp32 operands err {ew0:.1e} {t2:.3f} ms", flush=True)
    print(f"dgrad sum: fp32 operands {tb[0]:.3f} ms, slots {tb[1]:.3f} ms; wgrad sum: fp32 operands {tb[2]:.3f} ms, slots {tb[3]:.3f} ms")
    # the 3x3 input gradient that writes the up-sampled half of the concat gradient pre-split
    for (Bq, Cd, Ca, H, W) in ((4, 64, 128, 32, 64), (2, 128, 256, 16, 32)):
        dz = torch.randn(Bq, Cd, H, W, device=dev) * 1e-3
        w3 = torch.randn(Cd, Ca, 3, 3, device=dev) * (2.0 / (9 * Ca)) ** 0.5
        _, qd = ops.pack3x3_split(w3)
        sl = ops.absmax_slots(dz)
        k = 13 - math.floor(math.log2(float(dz.abs().max())))
        dzP = ops.split_pack_act(dz, f16=True, scale=2.0 ** k)
        ref = ops.conv3x3_split_pre(dzP, qd, Ca, slots=sl, always=True)
        bound = ops.conv3x3_dgrad_bound(w3, sl, Ca // 2)
        bv = float(torch.tensor(bound.cpu().numpy().view("float32")).max())
        assert bv >= float(ref[:, Ca // 2:].abs().max()), "bound below the gradient"
        da, daP = ops.conv3x3_split_dgrad_pre_slots(dzP, qd, Ca, Ca // 2, bound, slots=sl, always=True)
        kb = 13 - math.floor(math.log2(bv))
        same_lo = bool(torch.equal(da[:, :Ca // 2], ref[:, :Ca // 2]))
        want = ops.split_pack_act(ref[:, Ca // 2:].contiguous(), f16=True, scale=2.0 ** kb)
        same_up = bool(torch.equal(daP, want))
        if not same_up:
            dd = (nchw(daP).double() - nchw(want).double()).abs()
            bad = (daP != want)
            print("   mismatch: max |diff| of hi + mid", float(dd.max()), "of", float(nchw(want).abs().max()), "; differing halfs", int(bad.sum()), "of", bad.numel(),
                  "; hi parts differ", int(bad[:, :, :, 0].sum()), "mid parts differ", int(bad[:, :, :, 1].sum()))
            idx = bad.nonzero()[:4]
            for t in idx:
                print("   at", [int(v) for v in t], float(daP[tuple(t)]), float(want[tuple(t)]))
        print(f"3x3 dgrad with slot output {Bq}x{Cd}->{Ca} {H}x{W}: skip half identical {same_lo}, up-sampled half == split of the fp32 result {same_up} (bound / max {bv / float(ref[:, Ca // 2:].abs().max()):.1f})")
