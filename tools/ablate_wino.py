"""Timing-only ablation of the Winograd forward kernel (ONET_WINO_DBG builds; outputs are garbage):
one child process per build because the library reads the variable once.
usage: python tools/ablate_wino.py [Cin Cout H]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(here))
    import torch
    from onet_amd import ops
    ci, co, H = (int(v) for v in sys.argv[2:5])
    B = 32
    x = torch.randn(B, ci, H, H, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    qf, _ = ops.pack3x3_winograd(w)
    for _ in range(3):
        ops.conv3x3_winograd(x, qf, co)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        ops.conv3x3_winograd(x, qf, co)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"{t:.4f} ms  {2.0 * B * H * H * ci * co * 9 / t / 1e9:.1f} eff-TF")
    sys.exit(0)
shape = sys.argv[1:4] if len(sys.argv) >= 4 else ["512", "512", "32"]
names = {0: "full kernel", 1: "no global staging", 2: "no input transform", 4: "no A-operand LDS reads",
         8: "no patch LDS reads", 14: "MFMA + staging only", 15: "MFMA skeleton"}
for dbg, nm in names.items():
    env = dict(os.environ, ONET_WINO_DBG=str(dbg))
    out = subprocess.run([sys.executable, __file__, "--child", *shape], env=env, capture_output=True, text=True)
    print(f"DBG={dbg:2d} {nm:26s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
