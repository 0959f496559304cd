"""Headline benchmark: training images/s of the Onet twin 256x256 pass on N MI355X
(BASELINE.json metric), one process per GPU, weak scaling (per-GPU batch fixed).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = zero_grad + twin forward + JSD loss + backward + [RCCL all-reduce] + fused Adam on one
synthetic K-distributed clutter batch already resident in HBM (TS:209-219 order).  Prints ONE JSON
line on rank 0 with `roofline` (dominant MFMA kernel, HIP-event timed inside the timed region)
and, at N=1, `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_bench_latest.json")   # written by tools/pmc_parse.py


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS
    command (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950
    correction).  PMC counters cannot be collected inside the timed run, so the number comes from
    profiles/; None if the file is absent."""
    try:
        d = json.load(open(PMC_JSON))
    except Exception:
        return None
    tot_b = tot_n = 0
    for k, v in d.items():
        if k.startswith(kernel_prefix) and "hbm_bytes_per_launch" in v:
            n = v["FETCH_SIZE"]["launches"]
            tot_b += v["hbm_bytes_per_launch"] * n
            tot_n += n
    return (tot_b / tot_n) if tot_n else None


def cpu_baseline(X_cpu, seconds_budget):
    """fwd + loss + bwd + Adam of the CPU oracle (PyTorch CPU fp32, same ATen kernels the reference
    runs) on a B=2 sample of the SAME synthetic batch."""
    from oracle import onet_oracle as orc
    xs = X_cpu[:2].contiguous()
    top = orc.clone_state(orc.det_state_dict(xs.shape[1], 1981, randomize_running=False))
    params = [v for v in top.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-6, betas=(0.9, 0.999), eps=1e-8)
    times = []
    t_all = time.time()
    for i in range(6):
        t0 = time.time()
        opt.zero_grad()
        orc.train_mode_step(xs, top)
        opt.step()
        dt = time.time() - t0
        if i > 0:
            times.append(dt)
        if time.time() - t_all > seconds_budget and len(times) >= 1:
            break
    mean = sum(times) / len(times)
    return {"value": round(xs.shape[0] / mean, 4), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "B=2 of the same 1x%dx%d K-clutter batch, %d timed full training steps "
            "(fwd+loss+bwd+Adam), 1 warm-up" % (xs.shape[2], xs.shape[3], len(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (weak scaling)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--chans", type=int, default=1)
    ap.add_argument("--conv", default=None, choices=["auto", "bf16", "winograd4", "winograd", "direct"],
                    help="ops.CONV_ALGO for this run (default: ONET_CONV_ALGO or auto = the fp32 kernels); bf16 = BASELINE "
                         "config 3's bf16-operand MFMA path for forward / input gradient")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam instead of the fused flat Adam")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one gradient all-reduce after backward instead of bucketed all-reduces overlapped with it")
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    args = ap.parse_args()

    from onet_amd import Onet, _lib, ops
    from onet_amd import data as odata
    from onet_amd.trainer import FlatAdam, init_distributed, train_step
    _lib.load()                      # fail loudly without the HIP library
    if args.conv:
        ops.CONV_ALGO = args.conv
    bf16 = ops.CONV_ALGO == "bf16"
    # the loop owns the optimizer step: the loss's NaN assertion (OV:234) is evaluated by FlatAdam.step() before the update
    # instead of by a device synchronisation between forward and backward (ops.LAZY_NAN_CHECK)
    ops.LAZY_NAN_CHECK = not args.torch_adam

    rank, world, local = init_distributed("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    torch.manual_seed(1981)
    onet = Onet(in_chns=args.chans, binit=True, bshare=True).to(dev)
    opt = (torch.optim.Adam(onet.parameters(), lr=5e-6, betas=(0.9, 0.999), eps=1e-8) if args.torch_adam
           else FlatAdam(onet, lr=5e-6, world_size=world))
    if not args.torch_adam:
        opt.broadcast_params(0)
    distributed = dist.is_available() and dist.is_initialized()
    overlap = distributed and not args.torch_adam and not args.no_overlap
    if overlap:
        opt.enable_overlap(args.bucket_mb)
    onet.train()

    X_cpu = torch.from_numpy(odata.make_clutter_batch(args.batch, args.size, args.size, seed=1981 + rank,
                                                      channels=args.chans))
    X = X_cpu.to(dev)                # inputs resident in HBM before the timed region

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(onet, opt, X)
    barrier()
    ops.PROFILE = {}
    dev_allocs0 = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(onet, opt, X)
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        kern = {}
        for kind, recs in prof.items():
            ms = sum(e0.elapsed_time(e1) for _, e0, e1 in recs)
            fl = sum(f for f, _, _ in recs)
            kern[kind] = {"launches": len(recs), "ms_total": round(ms, 3), "avg_ms": round(ms / len(recs), 4),
                          "tflops": round(fl / (ms * 1e-3) / 1e12, 2)}
        dom = max(kern, key=lambda k: kern[k]["ms_total"])
        algo = {"conv_wino_kernel": "Winograd F(2x2,3x3) on fp32 MFMA: achieved = direct-convolution FLOPs / time "
                                    "(the kernel issues 2.25x fewer MFMA FLOPs, so frac can exceed the MFMA busy fraction)",
                "conv_wino4_kernel": "Winograd F(4x4,3x3) on fp32 MFMA: achieved = direct-convolution FLOPs / time "
                                     "(the kernel issues 4x fewer MFMA FLOPs, so frac can exceed 1; hardware_frac = issued MFMA FLOPs / time / "
                                     "peak, which the SQ_VALU_MFMA_BUSY_CYCLES profile under profiles/ confirms)",
                "conv_wino_wgrad_kernel": "Winograd F(2x2,3x3) weight gradient on fp32 MFMA: achieved = direct-convolution "
                                          "FLOPs / time (2.25x fewer MFMA FLOPs issued)",
                "conv3x3_bf16_kernel": "direct implicit GEMM, bf16 operands (rounded on the way into LDS) on "
                                       "v_mfma_f32_32x32x16_bf16, fp32 accumulation; peak = dense bf16 MFMA",
                "conv_fwd_kernel": "direct implicit GEMM on fp32 MFMA", "conv_wgrad_kernel": "split-K MFMA wgrad"}
        # multiplies the algorithm issues relative to direct convolution: F(4x4,3x3) 36 per 16 outputs x 9 taps -> 1/4,
        # F(2x2,3x3) 16 per 4 x 9 -> 1/2.25
        reduction = {"conv_wino4_kernel": 4.0, "conv_wino_kernel": 2.25, "conv_wino_wgrad_kernel": 2.25}.get(dom, 1.0)
        peak = BF16_MFMA_PEAK_TFLOPS if dom == "conv3x3_bf16_kernel" else FP32_MFMA_PEAK_TFLOPS
        roofline = {"kernel": dom, "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": peak,
                    "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / peak, 4),
                    "traffic": pmc_traffic(dom), "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc, profiles/)",
                    "mfma_flop_reduction": reduction,
                    "hardware_frac": round(kern[dom]["tflops"] / reduction / peak, 4),
                    "launches_timed": kern[dom]["launches"], "avg_launch_ms": kern[dom]["avg_ms"], "algorithm": algo.get(dom, dom),
                    "all": kern}
        imgs = args.batch * world * args.steps
        out = {"metric": "training images/sec (twin 256x256 pass)", "value": round(imgs / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16 conv operands (fwd/dgrad/wgrad), f32 accumulate/storage" if bf16 else "f32",
               "data": "synthetic",
               "config": {"workload": "%s: batch=%d/GPU %dx%dx%d synthetic K-clutter, %s, twin U-Net "
                                      "fwd+JSD loss+bwd+Adam" % ("configs[2]" if bf16 else "configs[1]", args.batch, args.chans,
                                                                 args.size, args.size, "bf16 MFMA conv path" if bf16 else "fp32"),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "optimizer": "torch.optim.Adam" if args.torch_adam else "fused flat Adam (HIP)",
                          "grad_allreduce": ("none (1 process)" if not distributed else
                                             "%g MB buckets overlapped with backward" % args.bucket_mb if overlap
                                             else "single all-reduce after backward")},
               "loss": loss_val, "hbm_peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
               "hbm_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
               "alloc_retries": int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)),
               "device_allocs_in_timed_steps": int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0)) - dev_allocs0,
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X_cpu, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
