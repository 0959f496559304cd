"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs the CPU oracle in fp32 and
vs the SAME oracle in fp64 ("truth"), to separate implementation error from fp32 conditioning."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import onet_oracle as orc
from onet_amd import Onet

def run(B, C, H, W):
    X = orc.det_input(B, C, H, W)
    top32 = orc.clone_state(orc.det_state_dict(C, 1981))
    (o32, l32, g32) = orc.train_mode_step(X, top32)
    sd64 = orc.det_state_dict(C, 1981)
    top64 = orc.clone_state({k: (v.double() if v.is_floating_point() else v) for k, v in sd64.items()})
    (o64, l64, g64) = orc.train_mode_step(X.double(), top64)
    m = Onet(C, True, True)
    m.load_state_dict(orc.onet_state_dict(C, 1981, True))
    m = m.cuda().train()
    Lt, Vt, Ld, Vd, S = m(X.cuda())
    loss = m.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
    loss.backward()
    print(f"== B{B} C{C} {H}x{W}: loss gpu {loss.item():.7f} cpu32 {float(l32):.7f} cpu64 {float(l64):.7f}")
    def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
    print("   fwd Vt: gpu-vs-64 %.2e cpu32-vs-64 %.2e | Lt: gpu %.2e cpu32 %.2e" % (
        rel(Vt.detach().cpu(), o64[1].detach()), rel(o32[1].detach(), o64[1].detach()),
        rel(Lt.detach().cpu(), o64[0].detach()), rel(o32[0].detach(), o64[0].detach())))
    named = dict(m.topu.named_parameters())
    worst = []
    for k in g32:
        eg = rel(named[k].grad.detach().cpu(), g64[k])
        ec = rel(g32[k], g64[k])
        egc = rel(named[k].grad.detach().cpu(), g32[k])
        worst.append((eg, ec, egc, k))
    for eg, ec, egc, k in worst:
        flag = " <<<" if eg > 1e-3 else ""
        print(f"   {k:50s} gpu-vs-64 {eg:.2e}  cpu32-vs-64 {ec:.2e}  gpu-vs-cpu32 {egc:.2e}{flag}")

if __name__ == "__main__":
    torch.set_num_threads(16)
    for shp in [(2, 1, 16, 16), (2, 1, 32, 32), (4, 1, 64, 48), (2, 1, 256, 256)]:
        run(*shp)
