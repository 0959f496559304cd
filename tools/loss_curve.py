"""Loss over N optimizer steps on one fixed synthetic batch (same init): the default dispatch (split 16-bit-operand kernels), the
fp32-MFMA-only kernels (Settings(split=False)) and the bf16-operand conv path."""
import sys, torch
sys.path.insert(0, ".")
from onet_amd import Onet, ops
from onet_amd import data as odata
from onet_amd.trainer import FlatAdam, train_step
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
X = torch.from_numpy(odata.make_clutter_batch(8, 256, 256, seed=7, channels=1)).to(dev)
curves = {}
for algo in ("auto", "f32mfma", "bf16"):
    ops.CONV_ALGO = "bf16" if algo == "bf16" else "auto"
    torch.manual_seed(1981)
    m = Onet(in_chns=1, binit=True, bshare=True).to(dev); m.train()
    if algo == "f32mfma":
        m.settings = ops.Settings(split=False)
    opt = FlatAdam(m, lr=1e-4, world_size=1)
    curves[algo] = [float(train_step(m, opt, X).item()) for _ in range(N)]
for i in range(0, N, max(1, N // 10)):
    a, f, b = curves["auto"][i], curves["f32mfma"][i], curves["bf16"][i]
    print(f"step {i:3d}  default {a:10.5f}  fp32-MFMA only {f:10.5f} (rel diff {abs(a-f)/abs(f):.1e})  bf16 {b:10.5f} (rel diff {abs(b-f)/abs(f):.1e})")
print("final", curves["auto"][-1], curves["f32mfma"][-1], curves["bf16"][-1])
