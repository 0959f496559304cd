"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes): a calibration copy with
a KNOWN byte count in each access width the kernels use (4 B/lane and 16 B/lane), then the modal
conv layers (64->64 @256^2, 512->512 @32^2, B=32) forward, dgrad-shaped and wgrad.  The parser
(tools/pmc_parse.py) turns the counters into bytes per launch with the calibration applied."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops, _lib

B = 32
dev = "cuda"
# calibration: 512 MiB strided copies, float4 path (16 B/lane) and scalar path (4 B/lane: odd element count)
n4 = 128 * 1024 * 1024
src = torch.randn(n4 + 1, device=dev); dst = torch.empty(n4 + 1, device=dev)
for _ in range(2):
    _lib.call("onet_copy_strided", src.data_ptr(), n4, dst.data_ptr(), n4, 1, n4, None)          # float4 kernel
    _lib.call("onet_copy_strided", src.data_ptr(), n4 + 1, dst.data_ptr(), n4 + 1, 1, n4 + 1, None)  # scalar kernel
torch.cuda.synchronize()
del src, dst
for ci, co, H in ((64, 64, 256), (512, 512, 32)):
    x = torch.randn(B, ci, H, H, device=dev); dz = torch.randn(B, co, H, H, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    wf, wd = ops.pack3x3(w)
    for _ in range(2):
        ops.conv_fwd(x, wf, co, 3)
        ops.conv_wgrad(x, dz, (co, ci, 3, 3), 3)
    torch.cuda.synchronize()
print("done")
