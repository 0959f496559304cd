"""Unit checks of the pre-split producers against onet_split_pack_act of the fp32 passes' results (must be bit-identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
def eq(a, b, what):
    ok = bool(torch.equal(a, b))
    print(what, "OK" if ok else "MISMATCH %d of %d" % (int((a != b).sum()), a.numel()), flush=True)
    return ok
for (B, C, H, W) in [(2, 64, 32, 64), (3, 16, 16, 32), (2, 128, 64, 32)]:
    z = torch.randn(B, C, H, W, device=dev) * 2 + 0.3
    save = torch.empty(4, C, device=dev)
    save[0] = torch.randn(C, device=dev) * 0.3; save[1] = torch.rand(C, device=dev) + 0.5
    save[2] = save[1] * (torch.rand(C, device=dev) + 0.5); save[3] = torch.randn(C, device=dev) * 0.5 + 0.2
    a = ops.bn_relu_apply(z, save)
    xs = ops.p16_empty(B, C, H, W, dev); a2 = torch.empty_like(a)
    ops.bn_relu_apply_split(z, save, xs, a=a2)
    eq(xs, ops.split_pack_act(a, f16=True), f"apply_split {B,C,H,W} slots"); eq(a2, a, "   fp32 copy")
    # into the leading groups of a wider buffer
    wide = ops.p16_empty(B, 2 * C, H, W, dev); wide.zero_()
    ops.bn_relu_apply_split(z, save, wide[:, :C // 8])
    eq(wide[:, :C // 8], ops.split_pack_act(a, f16=True), "   into a concat buffer's skip groups")
    # pooling
    y = ops.maxpool2_fwd(a)
    xs2 = ops.p16_empty(B, C, H, W, dev); ys = ops.p16_empty(B, C, H // 2, W // 2, dev); a3 = torch.empty_like(a); yf = torch.empty_like(y)
    assert ops.bn_relu_apply_pool_split(z, save, xs2, a3, ys, None)
    eq(xs2, ops.split_pack_act(a, f16=True), f"apply_pool_split {B,C,H,W} slots"); eq(a3, a, "   fp32 copy"); eq(ys, ops.split_pack_act(y, f16=True), "   pooled slots")
    assert ops.bn_relu_apply_pool_split(z, save, None, a3, None, yf)
    eq(yf, y, "   pooled fp32")
    # backward apply
    da = torch.randn(B, C, H, W, device=dev) * 1e-4
    dz, dg, db = ops.bn_relu_bwd(da, z, save, True)
    dzP, slots, dg2, db2 = ops.bn_relu_bwd_split(da, z, save.view(1, 4, C), True)
    bound = float(slots.view(torch.float32).max()); amax = float(dz.abs().max())
    k = 13 - (torch.tensor(bound).log2().floor().item())
    eq(dzP, ops.split_pack_act(dz, f16=True, scale=2.0 ** k), f"bwd_apply_split {B,C,H,W} (bound {bound:.3e} vs max |dz| {amax:.3e}, k = {k})")
    eq(dg2, dg, "   dgamma"); eq(db2, db, "   dbeta")
# ConvTranspose2d forward
for (B, Cin, h, w) in [(2, 128, 16, 32), (4, 256, 16, 16), (2, 1024, 16, 16)]:
    Ct = Cin // 2
    x = torch.randn(B, Cin, h, w, device=dev); wt = torch.randn(Cin, Ct, 2, 2, device=dev) * 0.05; bias = torch.randn(Ct, device=dev)
    wf = ops.packT2x2_fused(wt)
    ref = torch.empty(B, Ct, 2 * h, 2 * w, device=dev)
    ops.convT2x2_fwd(x, wf, bias, ref, Ct, 0, 0)
    cat = ops.p16_empty(B, 2 * Ct, 2 * h, 2 * w, dev); cat.zero_()
    took = ops.convT2x2_fwd_p(x, wf, bias, cat[:, Ct // 8:], Ct, 0, 0)
    print("convT fwd_p took:", took)
    if took:
        eq(cat[:, Ct // 8:], ops.split_pack_act(ref, f16=True), f"convT2x2_fwd_p {B,Cin,h,w}")
        print("   skip groups untouched:", float(cat[:, :Ct // 8].float().abs().max()) == 0.0)
