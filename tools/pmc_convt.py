"""Single launches of the slot-operand ConvTranspose2d GEMMs (forward, input gradient, weight gradient) for rocprofv3 --pmc passes:
the four decoder levels of the 256 x 256 U-Net at the benchmark's twin batch.   LEVELS=0,1,2,3 WHICH=fwd,dgrad,wgrad python tools/pmc_convt.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops
B = int(os.environ.get("B", "64"))
N = int(os.environ.get("N", "4"))
which = os.environ.get("WHICH", "fwd,dgrad,wgrad").split(",")
levels = [int(v) for v in os.environ.get("LEVELS", "0,1,2,3").split(",")]
dev = torch.device("cuda:0")
for li, (Cin, h) in enumerate(((1024, 16), (512, 32), (256, 64), (128, 128))):
    if li not in levels:
        continue
    Ct = Cin // 2
    x = torch.relu(torch.randn(B, Cin, h, h, device=dev))
    w = (torch.rand(Cin, Ct, 2, 2, device=dev) * 2 - 1) / (Ct * 4) ** 0.5
    dy = torch.randn(B, Ct, 2 * h, 2 * h, device=dev) * 3e-5
    dy_slots = ops.absmax_slots(dy)
    k = 13 - math.floor(math.log2(float(dy.abs().max())))
    dyP = ops.split_pack_act(dy, f16=True, scale=2.0 ** k)
    xP = ops.split_pack_act(x, f16=True)
    wP, wdP = ops.packT2x2_slots(w), ops.packT2x2_slots(w, dgrad=True)[1]
    outP = ops.p16_empty(B, Ct, 2 * h, 2 * h, dev, parts=2)
    for _ in range(N + 2):
        if "fwd" in which:
            ops.convT2x2_fwd_slots(xP, wP, None, outP, Ct)
        if "dgrad" in which:
            ops.convT2x2_dgrad_slots(dyP, wdP, Cin, dy_slots=dy_slots)
        if "wgrad" in which:
            ops.convT2x2_wgrad_slots(xP, dyP, (Cin, Ct, 2, 2), dy_slots=dy_slots, want_dbias=True)
    torch.cuda.synchronize()
