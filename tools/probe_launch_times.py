"""Per-launch HIP-event times of one pre-split forward layer over a long run of back-to-back launches (does the time drift?).
   B=64 python tools/probe_launch_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops, _lib
B = int(os.environ.get("B", "64"))
ci, co, H = [int(v) for v in os.environ.get("LAYER", "64,64,256").split(",")]
x = torch.randn(B, ci, H, H, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
sf, sd = ops.pack3x3_split(w)
out = torch.empty(B, co, H, H, device="cuda")
xs = ops.split_pack_act(x, f16=True)
nparts = int(_lib.load().onet_conv3x3_split_pre_nparts(B, H, H))
cm = torch.empty((co, nparts, 3), device="cuda")
def run(n, stats):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        ops.conv3x3_split_pre(xs, sf, co, out=out, stats=cm if stats else None)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for label, stats in (("plain", False), ("stats", True), ("plain", False), ("plain", False), ("stats", True)):
    t = run(24, stats)
    print(label, " ".join(f"{v:.3f}" for v in t), flush=True)
