"""Diagnostic (GPU box): gradient error per ACTIVATION (x1..x5, y4..y1 of the first U-Net pass) of the
HIP path and of the fp32 CPU oracle, both against the fp64 oracle: localises where error enters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import onet_oracle as orc
from onet_amd import Onet

def oracle_pass(x, st, store, tag):
    feats = [orc._double_conv(x, st, "inc", True)]
    for name, _, _ in orc.ENC[1:]:
        feats.append(orc._double_conv(F.max_pool2d(feats[-1], 2), st, name, True))
    ys = []
    y = feats[-1]
    for lvl, (name, _, _) in enumerate(orc.DEC):
        y = orc._double_conv(orc.upsample_cat(y, feats[3 - lvl], st, name), st, name, True)
        ys.append(y)
    for i, t in enumerate(feats):
        t.register_hook(lambda g, k=f"{tag}x{i+1}": store.__setitem__(k, g.detach().clone()))
    for i, t in enumerate(ys):
        t.register_hook(lambda g, k=f"{tag}y{4-i}": store.__setitem__(k, g.detach().clone()))
    return feats[0], y

def oracle_step(X, st, store):
    Lt, Ht = oracle_pass(X, st, store, "t")
    Vt = orc.head(Lt, Ht)
    Ld, Hd = oracle_pass(torch.clip(1 - X, 0, 1), st, store, "d")
    Vd = orc.head(Ld, Hd)
    S = torch.softmax(torch.cat([Vt, Vd], 1), 1)
    loss = orc.compute_loss(Lt, S[:, 0:1], Ld, S[:, 1:2])
    loss.backward()

def gpu_step(X, m, store):
    def unet(u, x, tag):
        x1 = u.inc(x); x2 = u.down1(x1); x3 = u.down2(x2); x4 = u.down3(x3); x5 = u.down4(x4)
        y4 = u.up1(x5, x4); y3 = u.up2(y4, x3); y2 = u.up3(y3, x2); y1 = u.up4(y2, x1)
        for k, t in (("x1", x1), ("x2", x2), ("x3", x3), ("x4", x4), ("x5", x5), ("y4", y4), ("y3", y3), ("y2", y2), ("y1", y1)):
            t.register_hook(lambda g, kk=tag + k: store.__setitem__(kk, g.detach().cpu().clone()))
        return x1, y1
    from onet_amd import functional as Fn
    Lt, Ht = unet(m.topu, X, "t")
    Ld, Hd = unet(m.dwnu, Fn.ComplementClipFn.apply(X, 0.0), "d")
    Vt, Vd, S = Fn.HeadSoftmaxFn.apply(Lt, Ht, Ld, Hd)
    loss = m.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
    loss.backward()

def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))

def run(B, C, H, W):
    X = orc.det_input(B, C, H, W)
    s32, s64, sg = {}, {}, {}
    oracle_step(X, orc.clone_state(orc.det_state_dict(C, 1981)), s32)
    oracle_step(X.double(), orc.clone_state({k: (v.double() if v.is_floating_point() else v) for k, v in orc.det_state_dict(C, 1981).items()}), s64)
    m = Onet(C, True, True); m.load_state_dict(orc.onet_state_dict(C, 1981, True)); m = m.cuda().train()
    gpu_step(X.cuda(), m, sg)
    print(f"== B{B} C{C} {H}x{W}")
    for k in ["ty1", "ty2", "ty3", "ty4", "tx5", "tx4", "tx3", "tx2", "tx1", "dy1", "dy2", "dy3", "dy4", "dx5", "dx4", "dx3", "dx2", "dx1"]:
        print(f"   d/d{k}: gpu-vs-64 {rel(sg[k], s64[k]):.2e}   cpu32-vs-64 {rel(s32[k], s64[k]):.2e}   shape {tuple(s64[k].shape)}")

if __name__ == "__main__":
    torch.set_num_threads(16)
    run(2, 1, 32, 32)
    run(2, 1, 40, 40)
