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
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_bench_latest.json")   # written by tools/pmc_parse.py


def pmc_traffic(kernel_prefix, bf16=False, batch=32):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS
    command (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950
    correction).  PMC counters cannot be collected inside the timed run, so the number comes from
    profiles/ (the default fp32 B=32 run: pmc_bench_latest.json; other configurations:
    pmc_bench_<f32|bf16>_b<batch>_latest.json); None if the file is absent."""
    path = PMC_JSON if (not bf16 and batch == 32) else os.path.join(
        ROOT, "profiles", "pmc_bench_%s_b%d_latest.json" % ("bf16" if bf16 else "f32", batch))
    try:
        d = json.load(open(path))
    except Exception:
        return None
    tot_b = tot_n = 0
    for k, v in d.items():
        if k.startswith(kernel_prefix) and "hbm_bytes_per_launch" in v:
            n = v["FETCH_SIZE"]["launches"]
            tot_b += v["hbm_bytes_per_launch"] * n
            tot_n += n
    return (tot_b / tot_n) if tot_n else None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each job a
    share of the host; os.cpu_count() reports the whole machine and 128 threads on a 16-core share run 3x slower than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(X_cpu, seconds_budget):
    """fwd + loss + bwd + Adam of the CPU oracle (PyTorch CPU fp32: the same ATen / oneDNN kernels the reference runs
    on CPU) on the host cores of this box, for n = all cores and n = 8 threads.  The sample is the first
    B_s images of the SAME synthetic batch, B_s = the largest power of two <= the benchmark batch whose 3 timed steps fit
    the budget of a thread setting (half of --cpu-seconds), estimated from a B=2 warm-up step."""
    from oracle import onet_oracle as orc
    ncores = _host_cores()
    settings = [ncores] + ([8] if ncores > 8 else [])
    per_setting = seconds_budget / len(settings)
    by_threads, sample = {}, {}
    for n in settings:
        torch.set_num_threads(n)
        top = orc.clone_state(orc.det_state_dict(X_cpu.shape[1], 1981, randomize_running=False))
        params = [v for v in top.values() if v.requires_grad]
        opt = torch.optim.Adam(params, lr=5e-6, betas=(0.9, 0.999), eps=1e-8)

        def step(xs):
            t0 = time.perf_counter()
            opt.zero_grad()
            orc.train_mode_step(xs, top)
            opt.step()
            return time.perf_counter() - t0

        t_warm = step(X_cpu[:2].contiguous())                 # warm-up (allocator, oneDNN primitive cache) + estimate
        bs = 2
        while bs * 2 <= X_cpu.shape[0] and 3 * t_warm * (bs * 2) / 2 <= per_setting - t_warm:
            bs *= 2
        xs = X_cpu[:bs].contiguous()
        times = []
        for _ in range(3):
            times.append(step(xs))
            sys.stderr.write("[cpu_baseline] n=%d B=%d step %.2f s\n" % (n, bs, times[-1]))
            sys.stderr.flush()
            if sum(times) > 2.0 * per_setting:                # the estimate was off: stop at what we have
                break
        by_threads[str(n)] = round(bs / (sum(times) / len(times)), 4)
        sample[str(n)] = bs
    torch.set_num_threads(ncores)
    n0 = str(settings[0])
    return {"value": by_threads[n0], "unit": "images/s", "cores": settings[0], "kind": "port",
            "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "by_threads": by_threads,
            "sample": "first %s images (n=%s threads) of the same %dx%dx%d K-clutter batch; per thread setting 1 warm-up "
                      "step at B=2 + up to 3 timed full training steps (zero_grad+fwd+loss+bwd+Adam) of the CPU oracle"
                      % ("/".join(str(sample[k]) for k in by_threads), "/".join(by_threads), X_cpu.shape[1],
                         X_cpu.shape[2], X_cpu.shape[3])}


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
    ap.add_argument("--mfma-events-only", action="store_true",
                    help="HIP-event brackets around the MFMA launches only (default: around every launch of the library, "
                         "which the whole-step breakdown and the streaming-kernel roofline need)")
    ap.add_argument("--cpu-seconds", type=float, default=60.0,
                    help="budget of the CPU-oracle baseline (split over the thread settings n = all cores and n = 8)")
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

    loss = None
    for _ in range(args.warmup):
        loss = train_step(onet, opt, X)      # (held like in the timed loop: the caching allocator sees the same liveness pattern)
    barrier()
    ops.profile_start(everything=not args.mfma_events_only)
    dev_allocs0 = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0))
    t0 = time.perf_counter()
    allocs_per_step = []
    for _ in range(args.steps):
        loss = train_step(onet, opt, X)
        allocs_per_step.append(int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0)))     # host-side counter, no sync
    barrier()
    elapsed = time.perf_counter() - t0
    prof, prof_all = ops.profile_stop()
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        # multiplies the algorithm issues relative to direct convolution: F(4x4,3x3) 36 per 16 outputs x 9 taps -> 1/4,
        # F(2x2,3x3) 16 per 4 x 9 -> 1/2.25 (the weight-gradient kernels F(3x3,4x4) / F(3x3,2x2) likewise)
        REDUCTION = {"conv_wino4_kernel": 4.0, "conv_wino_kernel": 2.25, "conv_wino_wgrad_kernel": 2.25,
                     "conv_wino4_wgrad_kernel": 4.0}
        BF16 = ("conv3x3_bf16_kernel", "conv3x3_wgrad_bf16_kernel")
        if bf16 and ops.CONVT_BF16:     # the ConvTranspose2d GEMMs take bf16 operands too (priced against the bf16 peak)
            BF16 += ("convt_gemm_kernel", "convt_wgrad_gemm_kernel")
        ALGO = {"conv_wino_kernel": "Winograd F(2x2,3x3) on fp32 MFMA (fwd + dgrad)",
                "conv_wino4_kernel": "Winograd F(4x4,3x3) on fp32 MFMA (fwd + dgrad; BatchNorm statistics / backward-reduce "
                                     "epilogues)",
                "conv_wino_wgrad_kernel": "Winograd F(2x2,3x3) weight gradient on fp32 MFMA, deterministic split-K",
                "conv_wino4_wgrad_kernel": "Winograd F(3x3,4x4) weight gradient on fp32 MFMA, deterministic split-K",
                "conv3x3_bf16_kernel": "direct implicit GEMM on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate)",
                "conv3x3_wgrad_bf16_kernel": "split-K weight gradient on v_mfma_f32_32x32x16_bf16",
                "conv_fwd_kernel": "direct implicit GEMM on fp32 MFMA (stem, tiny maps)",
                "conv_wgrad_kernel": "direct split-K weight gradient on fp32 MFMA (stem, tiny maps)",
                "stem_conv_stats_kernel": "stem convolution (Cin = n_channels) + BatchNorm statistics, one streaming VALU pass "
                                          "(bound by writing z: see hbm_frac)",
                "convt_gemm_kernel": "ConvTranspose2d forward / input-gradient GEMMs, 128x128 DMA-fed tiles (fp32 MFMA; bf16 "
                                     "operands under --conv bf16)",
                "convt_wgrad_gemm_kernel": "ConvTranspose2d weight-gradient GEMM, split-K (fp32 MFMA; bf16 operands under "
                                           "--conv bf16)"}
        kern = {}
        for kind, recs in prof.items():
            ms = sum(r[1].elapsed_time(r[2]) for r in recs)
            fl = sum(r[0] for r in recs)
            by = sum(r[3] for r in recs)
            peak = BF16_MFMA_PEAK_TFLOPS if kind in BF16 else FP32_MFMA_PEAK_TFLOPS
            issued = fl / REDUCTION.get(kind, 1.0)
            kern[kind] = {"launches": len(recs), "ms_total": round(ms, 3), "avg_ms": round(ms / len(recs), 4),
                          "direct_equivalent_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                          "issued_mfma_tflops": round(issued / (ms * 1e-3) / 1e12, 2),
                          "mfma_frac": round(issued / (ms * 1e-3) / 1e12 / peak, 4),
                          "compulsory_gbps": round(by / (ms * 1e-3) / 1e9, 1),
                          "hbm_frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        dom = max(kern, key=lambda k: kern[k]["ms_total"])
        d = kern[dom]
        traffic = pmc_traffic(dom, bf16, args.batch)
        if dom in BF16:
            # in bf16 the 64- and 128-channel layers sit below the ridge (312 FLOP/B): the launch is priced
            # against HBM; `achieved` = compulsory bytes (operands once + result once, fp32 storage) / time
            roofline = {"kernel": dom, "bound": "hbm", "achieved": d["compulsory_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": d["hbm_frac"], "mfma_frac": d["mfma_frac"]}
        else:
            # fp32: every dense layer is far above the ridge (19.7 FLOP/B): priced against the fp32 MFMA peak with the MFMA
            # FLOPs the kernel ISSUES (Winograd issues 4x / 2.25x fewer than the direct algorithm performs)
            roofline = {"kernel": dom, "bound": "mfma", "achieved": d["issued_mfma_tflops"], "peak": FP32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": d["mfma_frac"], "hbm_frac": d["hbm_frac"]}
        roofline.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc, profiles/)",
                         "compulsory_bytes_per_launch": round(d["compulsory_gbps"] * 1e9 * d["avg_ms"] * 1e-3),
                         "direct_equivalent_tflops": d["direct_equivalent_tflops"],
                         "mfma_flop_reduction": REDUCTION.get(dom, 1.0), "launches_timed": d["launches"],
                         "avg_launch_ms": d["avg_ms"], "algorithm": ALGO.get(dom, dom), "kernels": kern})
        if prof_all:
            stream = {}
            for name, recs in prof_all.items():
                ms = sum(r[1].elapsed_time(r[2]) for r in recs)
                by = sum(r[0] for r in recs if r[0])
                stream[name] = {"launches": len(recs), "ms_total": round(ms, 3), "avg_ms": round(ms / len(recs), 4),
                                "gbps": round(by / (ms * 1e-3) / 1e9, 1) if by else None}
            priced = {k: v for k, v in stream.items() if v["gbps"]}
            sdom = max(priced, key=lambda k: priced[k]["ms_total"])
            roofline["streaming"] = {"kernel": sdom, "bound": "hbm", "achieved": priced[sdom]["gbps"], "peak": HBM_PEAK_GBPS,
                                     "unit": "GB/s", "frac": round(priced[sdom]["gbps"] / HBM_PEAK_GBPS, 4),
                                     "launches_timed": priced[sdom]["launches"], "avg_launch_ms": priced[sdom]["avg_ms"],
                                     "kernels": dict(sorted(stream.items(), key=lambda kv: -kv[1]["ms_total"])[:12])}
            mfma_ms = sum(v["ms_total"] for v in kern.values()) / args.steps
            hbm_ms = sum(v["ms_total"] for v in priced.values()) / args.steps
            small_ms = sum(v["ms_total"] for k, v in stream.items() if k not in priced) / args.steps
            wall = elapsed / args.steps * 1e3
            roofline["step"] = {"wall_ms": round(wall, 3), "mfma_kernels_ms": round(mfma_ms, 3),
                                "hbm_kernels_ms": round(hbm_ms, 3), "small_kernels_ms": round(small_ms, 3),
                                "unattributed_ms": round(wall - mfma_ms - hbm_ms - small_ms, 3),
                                "mfma_share": round(mfma_ms / wall, 4), "hbm_share": round(hbm_ms / wall, 4),
                                "note": "per-launch HIP-event brackets on the launch stream, inside the timed region; "
                                        "unattributed = torch glue kernels, event overhead and launch gaps"}
        imgs = args.batch * world * args.steps
        if args.size == 256 and args.chans == 1:
            cfg_name = "configs[2]" if bf16 else ("configs[3]" if world > 1 and args.batch * world == 256 else "configs[1]")
        elif args.size == 512 and args.chans == 3:
            cfg_name = "configs[4]-shaped (3x512x512 tiles)"
        else:
            cfg_name = "non-BASELINE shape"
        out = {"metric": "training images/sec (twin %dx%d pass)" % (args.size, args.size), "value": round(imgs / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "dtype": "bf16" if bf16 else "f32",
               "precision": ("bf16 MFMA operands (3x3 conv fwd/dgrad/wgrad, ConvTranspose2d GEMMs), f32 accumulation, f32 conv outputs, "
                             "BatchNorm, loss, master weights and optimizer") if bf16 else "f32 throughout",
               "data": "synthetic",
               "config": {"workload": "%s: batch=%d/GPU %dx%dx%d synthetic K-clutter, %s, twin U-Net "
                                      "fwd+JSD loss+bwd+Adam" % (cfg_name, args.batch, args.chans,
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
               "device_allocs_per_timed_step": [b - a for a, b in zip([dev_allocs0] + allocs_per_step, allocs_per_step)],
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X_cpu, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:      # noqa: BLE001 -- any failure (RCCL included) ends THIS rank at once with a non-zero code:
        if isinstance(e, SystemExit) and e.code in (0, None):      # no in-process retry, no waiting in a collective's
            raise                                                  # destructor; the launcher then stops the other ranks
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
