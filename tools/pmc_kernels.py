"""Single-layer launches of the dominant MFMA kernels for rocprofv3 --pmc passes (tools/pmc_diag.sh): the 64-image twin batch
of the B=32 benchmark, shapes of the 256x256 U-Net.  Each kernel is launched N times after 2 warm-ups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops

B = int(os.environ.get("B", "64"))
N = int(os.environ.get("N", "4"))
which = os.environ.get("WHICH", "wino4,wino4w,presplit,convT").split(",")
shapes = [(64, 64, 256), (128, 128, 128), (512, 512, 32)]


def rep(fn):
    for _ in range(N + 2):
        fn()
    torch.cuda.synchronize()


for ci, co, H in shapes:
    x = torch.randn(B, ci, H, H, device="cuda")
    dz = torch.randn(B, co, H, H, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    if "wino4" in which:
        qf, qd = ops.pack3x3_winograd4(w)
        rep(lambda: ops.conv3x3_winograd4(x, qf, co))
    if "wino4w" in which and ops.winograd4_wgrad_ok(x, dz):
        rep(lambda: ops.conv3x3_winograd4_wgrad(x, dz, (co, ci, 3, 3)))
    if "split" in which:                    # the round-3 default kernels: forward (fp16 parts), input gradient, weight gradient
        sf, sd = ops.pack3x3_split(w)
        rep(lambda: ops.conv3x3_split(x, sf, co))
        rep(lambda: ops.conv3x3_split(dz, sd, ci))
        rep(lambda: ops.conv3x3_split_wgrad(x, dz, (co, ci, 3, 3)))
    if "presplit" in which:                 # round 4: pre-split operands, LDS-DMA staged kernels (fp16 hi | mid parts)
        import onet_amd.ops as _o
        keep = _o.SPLIT_GRAD_F16
        _o.SPLIT_GRAD_F16 = True
        pf, pd = ops.pack3x3_split(w)
        _o.SPLIT_GRAD_F16 = keep
        xP, dzP = ops.split_pack_act(x, f16=True), ops.split_pack_act(dz, f16=True)
        rep(lambda: ops.conv3x3_split_pre(xP, pf, co))
        rep(lambda: ops.conv3x3_split_pre(dzP, pd, ci))
        rep(lambda: ops.conv3x3_split_wgrad_pre(xP, dzP, (co, ci, 3, 3)))
if "convT" in which:
    for cin, h in ((128, 128), (512, 32)):
        ct = cin // 2
        x1 = torch.randn(B, cin, h, h, device="cuda")
        wt = torch.randn(cin, ct, 2, 2, device="cuda") * 0.05
        bias = torch.zeros(ct, device="cuda")
        wq = ops.packT2x2_fused(wt)
        wd = ops.packT2x2(wt)[1]
        cat = torch.empty(B, 2 * ct, 2 * h, 2 * h, device="cuda")
        dcat = torch.randn(B, 2 * ct, 2 * h, 2 * h, device="cuda")
        rep(lambda: ops.convT2x2_fwd(x1, wq, bias, cat[:, ct:], ct, 0, 0))
        rep(lambda: ops.convT2x2_dgrad(dcat[:, ct:], wd, cin, h, h, 0, 0))
        rep(lambda: ops.convT2x2_wgrad(x1, dcat[:, ct:], (cin, ct, 2, 2), 0, 0, False))
print("done")
