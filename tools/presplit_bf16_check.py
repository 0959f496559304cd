"""conv == "bf16" (BASELINE configs[2]) through the pre-split kernels with ONE part of plain bf16 operands against round 3's bf16 kernels
(conv_bf16.hip, bf16 storage): same operand roundings, other kernels -> results agree to accumulation-order rounding.
Kernel level against fp64 on the rounded operands, then one training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from onet_amd import Onet, ops
from oracle import onet_oracle as orc
dev = torch.device("cuda:0")
rb = lambda t: t.to(torch.bfloat16).double()
for (B, Cin, Cout, H, W) in [(2, 64, 64, 64, 64), (2, 64, 128, 32, 64), (4, 128, 64, 32, 32), (2, 96, 64, 16, 96)]:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; g = torch.randn(B, Cout, H, W, device=dev)
    xr = rb(x).cpu().requires_grad_(True); wr = rb(w).cpu().requires_grad_(True)
    zr = F.conv2d(xr, wr, None, 1, 1)
    wf, wd = ops.pack3x3_plain16(w)
    xP = ops.split_pack_act(x, parts=1)
    z = ops.conv3x3_split_pre(xP, wf, Cout)
    print(f"{B}x{Cin}->{Cout} {H}x{W}: fwd err {float((z.double().cpu() - zr.detach()).abs().max() / zr.detach().abs().max()):.2e}", end="  ")
    F.conv2d(xr, rb(w).cpu(), None, 1, 1).backward(rb(g).cpu())
    dx_ref = xr.grad.clone()
    gP = ops.split_pack_act(g, parts=1)
    dx = ops.conv3x3_split_pre(gP, wd, Cin)
    print(f"dgrad err {float((dx.double().cpu() - dx_ref).abs().max() / dx_ref.abs().max()):.2e}", end="  ")
    wz = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(rb(x).cpu(), wz, None, 1, 1).backward(rb(g).cpu())
    dw = ops.conv3x3_split_wgrad_pre(xP, gP, (Cout, Cin, 3, 3))
    print(f"wgrad err {float((dw.double().cpu() - wz.grad).abs().max() / wz.grad.abs().max()):.2e}", flush=True)
B, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 128)
X = orc.det_input(B, 1, H, H, seed=23).to(dev)
res = {}
for name, pre in (("round-3 bf16 kernels", False), ("pre-split plain bf16", True)):
    ops.PRESPLIT_BF16 = pre
    m = Onet(in_chns=1, binit=True, bshare=True)
    m.load_state_dict(orc.onet_state_dict(1, 1981, True, head_gain=0.3))
    m = m.to(dev).train()
    m.settings = ops.Settings(conv="bf16")
    m.zero_grad()
    Lt, Vt, Ld, Vd, S = m(X)
    loss = m.compute_loss(Lt, S[:, 0:1], Ld, S[:, 1:2])
    loss.backward()
    torch.cuda.synchronize()
    res[name] = (float(loss), Vt.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    print(name, "loss", float(loss))
a, b = res["round-3 bf16 kernels"], res["pre-split plain bf16"]
print("loss diff %.2e  Vt diff %.2e of scale" % (abs(a[0] - b[0]) / abs(a[0]), float((a[1] - b[1]).abs().max() / a[1].abs().max())))
print("worst relative gradient difference: %.2e (%s)" % max(((float((a[2][k] - b[2][k]).norm() / a[2][k].norm()), k) for k in a[2])))
