"""Per-layer timing of the F(4x4,3x3) forward / dgrad kernel and the F(2x2) weight-gradient kernel on the U-Net's
3x3 layer shapes (twin batch of 64 images of 256x256): where the convolution time of a step goes."""
import sys, torch
sys.path.insert(0, ".")
from onet_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 256
layers = [("inc.c2", 64, 64, 1), ("d1.c1", 64, 128, 2), ("d1.c2", 128, 128, 2), ("d2.c1", 128, 256, 4), ("d2.c2", 256, 256, 4),
          ("d3.c1", 256, 512, 8), ("d3.c2", 512, 512, 8), ("d4.c1", 512, 1024, 16), ("d4.c2", 1024, 1024, 16),
          ("u1.c1", 1024, 512, 8), ("u1.c2", 512, 512, 8), ("u2.c1", 512, 256, 4), ("u2.c2", 256, 256, 4),
          ("u3.c1", 256, 128, 2), ("u3.c2", 128, 128, 2), ("u4.c1", 128, 64, 1), ("u4.c2", 64, 64, 1)]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = [0, 0, 0]
print(f"{'layer':8s} {'Cin':>5s} {'Cout':>5s} {'HxW':>5s} | fwd ms   TF | dgrad ms   TF | wgrad ms   TF")
for name, ci, co, div in layers:
    h = S // div
    x = torch.randn(B, ci, h, h, device=dev); g = torch.randn(B, co, h, h, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    pk = ops.Packed3x3(w)
    fl = 2.0 * B * h * h * ci * co * 9 / 1e9
    tf = timeit(lambda: ops.conv3x3_auto(x, pk, 0))
    td = timeit(lambda: ops.conv3x3_auto(g, pk, 1))
    tw = timeit(lambda: ops.conv3x3_wgrad_auto(x, g, (co, ci, 3, 3)))
    tw4 = timeit(lambda: ops.conv3x3_winograd4_wgrad(x, g, (co, ci, 3, 3))) if ops.winograd4_wgrad_ok(x, g) else float("nan")
    tot[0] += tf; tot[1] += td; tot[2] += tw
    print(f"{name:8s} {ci:5d} {co:5d} {h:5d} | {tf:6.3f} {fl/tf:5.0f} | {td:6.3f} {fl/td:5.0f} | {tw:6.3f} {fl/tw:5.0f} | F(4x4)-wgrad {tw4:6.3f} {fl/tw4:5.0f}   algo {ops.conv3x3_algo(B, ci, co, h, h)}/{ops.conv3x3_algo(B, co, ci, h, h)}")
print("totals ms: fwd %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
print("ConvTranspose2d GEMMs (fwd / dgrad / wgrad), 68.7 GFLOP each at B=64:")
for name, cin, div in (("up1", 1024, 16), ("up2", 512, 8), ("up3", 256, 4), ("up4", 128, 2)):
    h = S // div
    ct = cin // 2
    x1 = torch.randn(B, cin, h, h, device=dev)
    wt = torch.randn(cin, ct, 2, 2, device=dev) * 0.05
    bias = torch.zeros(ct, device=dev)
    wq, wd = ops.packT2x2_fused(wt), ops.packT2x2(wt)[1]
    cat = torch.empty(B, 2 * ct, 2 * h, 2 * h, device=dev)
    dcat = torch.randn(B, 2 * ct, 2 * h, 2 * h, device=dev)
    fl = 2.0 * B * h * h * cin * 4 * ct / 1e9
    tf = timeit(lambda: ops.convT2x2_fwd(x1, wq, bias, cat[:, ct:], ct, 0, 0))
    td = timeit(lambda: ops.convT2x2_dgrad(dcat[:, ct:], wd, cin, h, h, 0, 0))
    tw = timeit(lambda: ops.convT2x2_wgrad(x1, dcat[:, ct:], (cin, ct, 2, 2), 0, 0, False))
    print(f"{name:8s} {cin:5d} {ct:5d} {h:5d} | {tf:6.3f} {fl/tf:5.0f} | {td:6.3f} {fl/td:5.0f} | {tw:6.3f} {fl/tw:5.0f}")
