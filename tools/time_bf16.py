"""HIP-event timing of the bf16 convolution kernels, layer by layer (the 512-image twin batch of BASELINE config 3 by default
costs too much memory for a quick probe: B images of the U-Net's 256x256 level shapes).  ONET_HIP_LIB selects a variant build.
   python tools/time_bf16.py [B=64] [N=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onet_amd import ops

B = int(os.environ.get("B", "64"))
N = int(os.environ.get("N", "10"))
shapes = [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 1024, 16)]
tot_f = tot_w = 0.0
for ci, co, H in shapes:
    x16 = torch.randn(B, ci, H, H, device="cuda").to(torch.bfloat16)
    dz16 = torch.randn(B, co, H, H, device="cuda").to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    bf, bd = ops.pack3x3_bf16(w)
    if os.environ.get("BLOCKED"):      # channel-blocked copy [C/8][H][W][8] through the experimental entry point
        from onet_amd import _lib
        xb = x16.view(B, ci // 8, 8, H, H).permute(0, 1, 3, 4, 2).contiguous()
        zb = torch.empty(B, co, H, H, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        z0 = ops.conv3x3_bf16(None, bf, co, x16=x16)
        _lib.call("onet_conv3x3_bf16_fwd_blk", xb.data_ptr(), ci * H * H, bf.data_ptr(), zb.data_ptr(), co * H * H, B, ci, co, H, H, st)
        assert torch.equal(z0, zb), "blocked-layout forward differs"
        fwd = lambda: _lib.call("onet_conv3x3_bf16_fwd_blk", xb.data_ptr(), ci * H * H, bf.data_ptr(), zb.data_ptr(), co * H * H, B, ci, co, H, H, st)
    else:
        fwd = lambda: ops.conv3x3_bf16(None, bf, co, x16=x16)
    res = []
    for fn in (fwd,
               lambda: ops.conv3x3_wgrad_bf16(None, None, (co, ci, 3, 3), x16=x16, dz16=dz16)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / N)
    fl = 2.0 * B * H * H * ci * co * 9 / 1e9
    tot_f += res[0]
    tot_w += res[1]
    print(f"{ci:5d}->{co:5d} @{H:3d}^2  fwd {res[0]:7.3f} ms {fl / res[0]:7.1f} TF   wgrad {res[1]:7.3f} ms {fl / res[1]:7.1f} TF", flush=True)
print(f"sum fwd {tot_f:.3f} ms  wgrad {tot_w:.3f} ms")
