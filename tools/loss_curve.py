"""Loss over N optimizer steps on one fixed synthetic batch, fp32 kernels vs the bf16-operand conv path (same init)."""
import sys, torch
sys.path.insert(0, ".")
from onet_amd import Onet, ops
from onet_amd import data as odata
from onet_amd.trainer import FlatAdam, train_step
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
X = torch.from_numpy(odata.make_clutter_batch(8, 256, 256, seed=7, channels=1)).to(dev)
curves = {}
for algo in ("auto", "bf16"):
    ops.CONV_ALGO = algo
    torch.manual_seed(1981)
    m = Onet(in_chns=1, binit=True, bshare=True).to(dev); m.train()
    opt = FlatAdam(m, lr=1e-4, world_size=1)
    curves[algo] = [float(train_step(m, opt, X).item()) for _ in range(N)]
for i in range(0, N, max(1, N // 10)):
    a, b = curves["auto"][i], curves["bf16"][i]
    print(f"step {i:3d}  fp32 {a:10.5f}  bf16 {b:10.5f}  rel diff {abs(a-b)/abs(a):.1e}")
print("final", curves["auto"][-1], curves["bf16"][-1])
