"""Pre-split storage (Settings.presplit) against the default path on one training step: forward outputs must be bit-identical (same
parts, same MFMA order), gradients agree to rounding.   python tools/presplit_check.py [B H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import Onet, ops
from oracle import onet_oracle as orc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
X = orc.det_input(B, 1, H, H, seed=23).to(dev)
res = {}
for name, st in (("default", ops.Settings(presplit=False, bn_on_load=False)), ("presplit", ops.Settings(presplit=True))):
    m = Onet(in_chns=1, binit=True, bshare=True)
    m.load_state_dict(orc.onet_state_dict(1, 1981, True, head_gain=0.3))
    m = m.to(dev).train()
    m.settings = st
    m.zero_grad()
    Lt, Vt, Ld, Vd, S = m(X)
    loss = m.compute_loss(Lt, S[:, 0:1], Ld, S[:, 1:2])
    loss.backward()
    torch.cuda.synchronize()
    res[name] = (loss.detach().clone(), Lt.detach().clone(), Vt.detach().clone(), S.detach().clone(),
                 {k: p.grad.detach().clone() for k, p in m.named_parameters()}, [b.detach().clone() for b in m.buffers()])
    print(name, "loss", float(loss))
a, b = res["default"], res["presplit"]
print("forward bit-identical:", bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])))
print("buffers bit-identical:", all(torch.equal(p, q) for p, q in zip(a[5], b[5])))
worst = max(((float((a[4][k] - b[4][k]).norm() / a[4][k].norm()), k) for k in a[4]))
print("worst relative gradient difference: %.2e (%s)" % worst)
