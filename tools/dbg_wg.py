import sys, torch
sys.path.insert(0, ".")
from onet_amd import ops
dev = torch.device("cuda:0")
for (B, Ci, Co, H, W) in [(1, 64, 64, 4, 16), (1, 64, 64, 8, 32), (2, 64, 64, 32, 32)]:
    torch.manual_seed(0)
    x = torch.randn(B, Ci, H, W); g = torch.randn(B, Co, H, W)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float64)
    dw = ops.conv3x3_wgrad_bf16(x.to(dev), g.to(dev), (Co, Ci, 3, 3)).cpu().double()
    ref = torch.nn.grad.conv2d_weight(rb(x), (Co, Ci, 3, 3), rb(g), stride=1, padding=1)
    err = (dw - ref).abs()
    print((B, Ci, Co, H, W), "scale", float(ref.abs().max()), "max err per tap", [round(float(err[:, :, t // 3, t % 3].max()), 4) for t in range(9)])
    bad = (err > 1e-3 * ref.abs().max()).nonzero()
    print("  bad entries", bad.shape[0], bad[:6].tolist())
    print("  |dw| mean", float(dw.abs().mean()), "|ref| mean", float(ref.abs().mean()), "corr", float((dw * ref).sum() / (dw.norm() * ref.norm() + 1e-30)))
    # which single-tap shift correlates best with tap (1,1)?
    for t in range(9):
        print("   tap", t, "corr with ref taps", [round(float((dw[:, :, t // 3, t % 3] * ref[:, :, u // 3, u % 3]).sum() / (dw[:, :, t // 3, t % 3].norm() * ref[:, :, u // 3, u % 3].norm() + 1e-30)), 2) for u in range(9)])
    refT = ref.transpose(0, 1)
    print("  corr with transposed (co<->ci)", float((dw * refT).sum() / (dw.norm() * refT.norm() + 1e-30)))
    break
