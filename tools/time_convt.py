"""HIP-event timing of the three ConvTranspose2d GEMMs per decoder level (B images; ONET_HIP_LIB selects a variant build).
   B=64 N=10 python tools/time_convt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops

B = int(os.environ.get("B", "64"))
N = int(os.environ.get("N", "10"))
if os.environ.get("ALGO"):
    ops.CONV_ALGO = os.environ["ALGO"]
tot = [0.0, 0.0, 0.0]
for cin, h in ((128, 128), (256, 64), (512, 32), (1024, 16)):
    ct = cin // 2
    x1 = torch.randn(B, cin, h, h, device="cuda")
    wt = torch.randn(cin, ct, 2, 2, device="cuda") * 0.05
    bias = torch.zeros(ct, device="cuda")
    wq = ops.packT2x2_fused(wt)
    wd = ops.packT2x2(wt)[1]
    cat = torch.empty(B, 2 * ct, 2 * h, 2 * h, device="cuda")
    dcat = torch.randn(B, 2 * ct, 2 * h, 2 * h, device="cuda")
    res = []
    for fn in (lambda: ops.convT2x2_fwd(x1, wq, bias, cat[:, ct:], ct, 0, 0),
               lambda: ops.convT2x2_dgrad(dcat[:, ct:], wd, cin, h, h, 0, 0),
               lambda: ops.convT2x2_wgrad(x1, dcat[:, ct:], (cin, ct, 2, 2), 0, 0, False)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / N)
    fl = 2.0 * B * h * h * cin * 4 * ct / 1e9
    for i in range(3):
        tot[i] += res[i]
    print(f"{cin:5d}->{ct:4d} @{h:3d}^2  fwd {res[0]:6.3f} ms {fl / res[0]:6.1f} TF  dgrad {res[1]:6.3f} ms {fl / res[1]:6.1f} TF  "
          f"wgrad {res[2]:6.3f} ms {fl / res[2]:6.1f} TF", flush=True)
print(f"sum fwd {tot[0]:.3f}  dgrad {tot[1]:.3f}  wgrad {tot[2]:.3f} ms")
