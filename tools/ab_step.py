"""Same-process A/B of the benchmark step (B images of 1x256x256, twin U-Net fwd + loss + bwd + fused Adam): interleaved rounds of the
default configuration and variants that flip module switches of onet_amd.ops, HIP-event time per step, median and min over the rounds.
   python tools/ab_step.py FUSE_DGRAD_REDUCE=0 [NAME=VALUE ...]      (each argument = one variant; '+'-joined: several switches)
   B=32 CONV=auto ROUNDS=5 STEPS=6 are environment knobs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import Onet, ops
from onet_amd import data as odata
from onet_amd.trainer import FlatAdam, train_step
B = int(os.environ.get("B", "32")); ROUNDS = int(os.environ.get("ROUNDS", "5")); STEPS = int(os.environ.get("STEPS", "6"))
conv = os.environ.get("CONV", "auto")
dev = torch.device("cuda:0")
torch.manual_seed(1981)
onet = Onet(in_chns=1, binit=True, bshare=True).to(dev)
onet.settings = ops.Settings(conv=conv, lazy_nan=True)
opt = FlatAdam(onet, lr=5e-6)
onet.train()
X = odata.make_clutter_batch_gpu(B, 256, 256, seed=1981, device=dev).contiguous()
variants = [("default", {})]
for arg in sys.argv[1:]:
    kv = {}
    for item in arg.split("+"):
        k, v = item.split("=")
        kv[k] = {"0": False, "1": True}.get(v, v)
    variants.append((arg, kv))
def run(kv, n):
    old = {k: getattr(ops, k) for k in kv}
    for k, v in kv.items():
        setattr(ops, k, v)
    try:
        train_step(onet, opt, X)                      # (switch warm-up: packs, allocator)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            loss = train_step(onet, opt, X)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, float(loss)
    finally:
        for k, v in old.items():
            setattr(ops, k, v)
for name, kv in variants:
    run(kv, 2)
times = {name: [] for name, _ in variants}
for r in range(ROUNDS):
    for name, kv in variants:
        t, loss = run(kv, STEPS)
        times[name].append(t)
for name, _ in variants:
    ts = times[name]
    print(f"{name:40s} median {statistics.median(ts):8.3f} ms  min {min(ts):8.3f}  ({' '.join('%.2f' % t for t in ts)})  loss {loss:.5f}", flush=True)
