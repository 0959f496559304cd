"""GPU micro-benchmark: every conv layer shape of the 256x256 U-Net (B per GPU = 32) through
onet_conv_fwd (fwd and dgrad orientation) and onet_conv_wgrad; prints TFLOP/s per layer.
ONET_CONV_CFG=P|C|D|E forces a forward tile configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops

B = int(os.environ.get("B", "32"))
S = int(os.environ.get("S", "256"))
L = [("inc.c1", 1, 64, 1), ("inc.c2", 64, 64, 1), ("down1.c1", 64, 128, 2), ("down1.c2", 128, 128, 2),
     ("down2.c1", 128, 256, 4), ("down2.c2", 256, 256, 4), ("down3.c1", 256, 512, 8), ("down3.c2", 512, 512, 8),
     ("down4.c1", 512, 1024, 16), ("down4.c2", 1024, 1024, 16), ("up1.c1", 1024, 512, 8), ("up1.c2", 512, 512, 8),
     ("up2.c1", 512, 256, 4), ("up2.c2", 256, 256, 4), ("up3.c1", 256, 128, 2), ("up3.c2", 128, 128, 2),
     ("up4.c1", 128, 64, 1), ("up4.c2", 64, 64, 1)]

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0], "wino": [0, 0], "winod": [0, 0], "winow": [0, 0], "wino4": [0, 0], "wino4d": [0, 0]}
which = os.environ.get("WHICH", "fwd,dgrad,wgrad").split(",")
for name, ci, co, d in L:
    H = S // d
    x = torch.randn(B, ci, H, H, device="cuda"); dz = torch.randn(B, co, H, H, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wf, wd = ops.pack3x3(w)
    fl = 2.0 * B * H * H * ci * co * 9
    out = [f"{name:9s} {ci:4d}->{co:4d} @{H:3d}"]
    if "fwd" in which:
        t = timeit(lambda: ops.conv_fwd(x, wf, co, 3)); tot["fwd"][0] += fl; tot["fwd"][1] += t
        out.append(f"fwd {t:7.3f} ms {fl/t/1e9:6.1f} TF")
    if "dgrad" in which and ci >= 32:
        t = timeit(lambda: ops.conv_fwd(dz, wd, ci, 3)); tot["dgrad"][0] += fl; tot["dgrad"][1] += t
        out.append(f"dgrad {t:7.3f} ms {fl/t/1e9:6.1f} TF")
    if "wino" in which and ci >= 8:
        qf, qd = ops.pack3x3_winograd(w)
        t = timeit(lambda: ops.conv3x3_winograd(x, qf, co)); tot["wino"][0] += fl; tot["wino"][1] += t
        out.append(f"wino {t:7.3f} ms {fl/t/1e9:6.1f} TF")
        t = timeit(lambda: ops.conv3x3_winograd(dz, qd, ci)); tot["winod"][0] += fl; tot["winod"][1] += t
        out.append(f"wino-dgrad {t:7.3f} ms {fl/t/1e9:6.1f} TF")
    if "wino4" in which and ci >= 8:
        q4f, q4d = ops.pack3x3_winograd4(w)
        t = timeit(lambda: ops.conv3x3_winograd4(x, q4f, co)); tot["wino4"][0] += fl; tot["wino4"][1] += t
        out.append(f"wino4 {t:7.3f} ms {fl/t/1e9:6.1f} TF")
        t = timeit(lambda: ops.conv3x3_winograd4(dz, q4d, ci)); tot["wino4d"][0] += fl; tot["wino4d"][1] += t
        out.append(f"wino4-dgrad {t:7.3f} ms {fl/t/1e9:6.1f} TF")
    if "winow" in which and ci >= 16:
        t = timeit(lambda: ops.conv3x3_winograd_wgrad(x, dz, (co, ci, 3, 3))); tot["winow"][0] += fl; tot["winow"][1] += t
        out.append(f"wino-wgrad {t:7.3f} ms {fl/t/1e9:6.1f} TF")
    if "wgrad" in which:
        t = timeit(lambda: ops.conv_wgrad(x, dz, (co, ci, 3, 3), 3)); tot["wgrad"][0] += fl; tot["wgrad"][1] += t
        out.append(f"wgrad {t:7.3f} ms {fl/t/1e9:6.1f} TF")
    print("  ".join(out), flush=True)
for k, (f, t) in tot.items():
    if t: print(f"TOTAL {k}: {t:.2f} ms  {f/t/1e9:.1f} TF")
