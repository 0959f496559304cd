"""Per-step caching-allocator trace of the training step (run on the GPU box):
   python tools/alloc_trace.py [--batch 32] [--conv bf16] [--steps 12] [--profile-from 6]
prints, per step: wall ms, hipMalloc calls made during the step, reserved / allocated / peak GB."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onet_amd import ops, data as odata
from onet_amd.modules import Onet
from onet_amd.trainer import FlatAdam, train_step

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--conv", default=None)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--profile-from", type=int, default=-1)
ap.add_argument("--no-sync", action="store_true", help="let the host run ahead of the GPU between steps")
ap.add_argument("--hold-loss", action="store_true", help="keep the previous step's loss alive, as a training loop does")
a = ap.parse_args()
if a.conv:
    ops.CONV_ALGO = a.conv
ops.LAZY_NAN_CHECK = True
dev = torch.device("cuda", 0)
torch.manual_seed(1981)
onet = Onet(in_chns=1, binit=True, bshare=True).to(dev)
opt = FlatAdam(onet, lr=5e-6, world_size=1)
onet.train()
X = torch.from_numpy(odata.make_clutter_batch(min(a.batch, 8), a.size, a.size, seed=1981)).to(dev)
X = X.repeat((a.batch + X.shape[0] - 1) // X.shape[0], 1, 1, 1)[:a.batch].contiguous()
st = lambda k: int(torch.cuda.memory_stats(dev).get(k, 0))
for i in range(a.steps):
    if i == a.profile_from:
        ops.profile_start(everything=True)
    n0 = st("num_device_alloc")
    f0 = st("num_device_free")
    torch.cuda.reset_peak_memory_stats(dev)
    if not a.no_sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = train_step(onet, opt, X)
    if a.hold_loss:
        loss = r
    del r
    if not a.no_sync:
        torch.cuda.synchronize()
    print(f"step {i:2d}  {1e3 * (time.perf_counter() - t0):8.2f} ms  mallocs {st('num_device_alloc') - n0:3d}  frees "
          f"{st('num_device_free') - f0:3d}  reserved {torch.cuda.memory_reserved(dev) / 2**30:7.2f}  allocated "
          f"{torch.cuda.memory_allocated(dev) / 2**30:7.2f}  peak {torch.cuda.max_memory_allocated(dev) / 2**30:7.2f} GiB",
          flush=True)
if a.hold_loss:
    import gc
    big = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() > 2**22]
    seen = set()
    for o in big:
        k = o.untyped_storage().data_ptr()
        if k in seen:
            continue
        seen.add(k)
        print("live:", tuple(o.shape), o.dtype, f"{o.untyped_storage().nbytes() / 2**20:.1f} MiB", flush=True)
