"""Per-launch table of the MFMA kernels INSIDE the benchmark step (B=32 twin batch, 1x256x256): HIP-event time of every launch of a
step, averaged over N steps by position in the step -> markdown on stdout.   python tools/step_layer_table.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import Onet_vanilla_20240606 as OV
from onet_amd import ops
from onet_amd.trainer import FlatAdam, train_step
from onet_amd.data import make_clutter_batch_gpu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(os.environ.get("B", "32"))
dev = torch.device("cuda", 0)
torch.manual_seed(1981)
net = OV.Onet(in_chns=1, binit=True, bshare=True).to(dev).train()
net.settings = ops.Settings(lazy_nan=True)            # as bench.py runs it
opt = FlatAdam(net, lr=5e-6, world_size=1)
X = make_clutter_batch_gpu(B, 256, 256, seed=1981, device=dev).contiguous()
for _ in range(3):
    train_step(net, opt, X)
torch.cuda.synchronize()
order = []
real_end = ops._prof_end
def spy(kind, flops, e0, nbytes=0.0):
    if e0 is not None:
        order.append(kind)
    return real_end(kind, flops, e0, nbytes)
ops._prof_end = spy
ops.profile_start(everything=False)
for _ in range(N):
    train_step(net, opt, X)
torch.cuda.synchronize()
prof, _ = ops.profile_stop()
ops._prof_end = real_end
per = len(order) // N
assert order[:per] * N == order, "launch order differs between steps"
cursor = {k: 0 for k in prof}
rows = []
for pos in range(per):
    kind = order[pos]
    ms, fl = 0.0, 0.0
    for s in range(N):
        r = prof[kind][cursor[kind] + s * (len(prof[kind]) // N)]
        ms += r[1].elapsed_time(r[2]); fl = r[0]
    cursor[kind] += 1
    rows.append((pos, kind, fl, ms / N))
# names of the launches of the standard U-Net step (65 MFMA-side launches): forward in module order, backward in reverse (weight
# gradient, then input gradient of each unit; ConvTranspose2d: input gradient, then weight gradient)
enc = ["inc.c2", "down1.c1", "down1.c2", "down2.c1", "down2.c2", "down3.c1", "down3.c2", "down4.c1", "down4.c2"]
dec = [("up1.T", "up1.c1", "up1.c2"), ("up2.T", "up2.c1", "up2.c2"), ("up3.T", "up3.c1", "up3.c2"), ("up4.T", "up4.c1", "up4.c2")]
names = ["inc.c1 (stem) fwd"] + [n + " fwd" for n in enc] + [n + " fwd" for t in dec for n in t]
for t in reversed(dec):
    names += [t[2] + " wgrad", t[2] + " dgrad", t[1] + " wgrad", t[1] + " dgrad", t[0] + " dgrad", t[0] + " wgrad"]
for n in reversed(enc):
    names += [n + " wgrad", n + " dgrad"]
names += ["inc.c1 (stem) wgrad + BN bwd"]
rows = [r for r in rows if r[2] > 0]          # (an entry point that declined a shape leaves a zero-FLOP record: the caller launched another)
rows = [(i,) + r[1:] for i, r in enumerate(rows)]
if len(names) != len(rows):
    names = [""] * len(rows)
SPLIT = {"conv3x3_split_pre_kernel", "conv3x3_split_wgrad_pre_kernel", "convt_gemm_kernel", "convt_wgrad_gemm_kernel"}
print(f"| # | launch | kernel | direct GFLOP | ms | direct TF | issued fraction of 2.5 PF |\n|---:|---|---|---:|---:|---:|---:|")
tot = {}
for pos, kind, fl, ms in rows:
    issued = 3.0 * fl if kind in SPLIT else fl
    print(f"| {pos} | {names[pos]} | `{kind}` | {fl / 1e9:.1f} | {ms:.3f} | {fl / ms / 1e9:.0f} | {issued / ms / 1e9 / 2500:.3f} |" if kind in SPLIT else
          f"| {pos} | {names[pos]} | `{kind}` | {fl / 1e9:.1f} | {ms:.3f} | {fl / ms / 1e9:.0f} | |")
    t = tot.setdefault(kind, [0, 0.0, 0.0]); t[0] += 1; t[1] += ms; t[2] += issued
print()
for k, (n, ms, iss) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"* `{k}`: {n} launches, {ms:.2f} ms/step" + (f", {iss / ms / 1e9 / 2500:.3f} of 2.5 PF issued" if k in SPLIT else ""))
