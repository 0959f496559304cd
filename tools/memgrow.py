import sys, torch
sys.path.insert(0, ".")
from onet_amd import Onet, ops
from onet_amd import data as odata
from onet_amd.trainer import FlatAdam, train_step
dev = torch.device("cuda:0")
ops.LAZY_NAN_CHECK = True
B = int(sys.argv[1]); N = int(sys.argv[2])
X = torch.from_numpy(odata.make_clutter_batch(B, 256, 256, seed=7, channels=1)).to(dev)
m = Onet(in_chns=1, binit=True, bshare=True).to(dev); m.train()
opt = FlatAdam(m, lr=5e-6, world_size=1)
for i in range(N):
    loss = train_step(m, opt, X)      # held across the next step, as a training loop does until loss.item()
    if i % 4 == 3 or i < 3:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        print(i, "reserved GB %.1f" % (torch.cuda.memory_reserved(dev) / 2**30), "allocated peak GB %.1f" % (torch.cuda.max_memory_allocated(dev) / 2**30),
              "device allocs", st.get("num_device_alloc", 0), "frees", st.get("num_device_free", 0), flush=True)
