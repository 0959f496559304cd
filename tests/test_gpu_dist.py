"""N > 1 rehearsal on the one-GPU box: two ranks of bench.py share cuda:0 and exchange gradients over gloo
(RCCL refuses two ranks on one device).  Everything but the collective library is the code the driver's
multi-GPU run executes: torchrun environment, broadcast of the flat parameters, bucketed all-reduce issued from
the backward hooks, max-over-ranks timing, the single JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script, *args):
    env = dict(os.environ, ONET_DIST_BACKEND="gloo", ONET_FORCE_LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), script] + list(args)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shape", ["c2_small", "c5_3x512x512"])
def test_bench_two_ranks_sharing_one_gpu_over_gloo(shape):
    """c2_small: the benchmark program at a small size; c5_3x512x512: BASELINE configs[4]'s tile shape (3-channel
    512x512, 4 images per rank) through the same two-rank path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    extra = ["--batch", "4", "--size", "64"] if shape == "c2_small" else ["--batch", "4", "--size", "512", "--chans", "3"]
    out = _torchrun(os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                    "--bucket-mb", "8", *extra)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    assert "overlapped with backward" in d["config"]["grad_allreduce"]
    assert d["value"] > 0 and d["loss"] == d["loss"]               # finite
    assert 0 < d["roofline"]["frac"] <= 1.0


@pytest.mark.timeout(600)
def test_bench_two_ranks_at_the_configs3_per_gpu_workload():
    """The step the 8-GPU node will run, rehearsed before it gets there: BASELINE configs[3]'s PER-GPU workload (32 images of
    1x256x256 per rank) with the DEFAULT bucket schedule (the 124 MB flat gradient in four 32 MB buckets, reduced from the backward
    hooks), two ranks sharing cuda:0 over gloo.  Asserted: the replicas hold bit-identical parameters after the timed steps
    (checksums all-gathered by bench.py), every rank's loss is finite and its HBM peak is reported, the with / without-overlap
    loops ran and `allreduce_exposed_ms` is in the line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _torchrun(os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and "batch=32/GPU 1x256x256" in d["config"]["workload"]
    assert d["config"]["grad_allreduce"] == "32 MB buckets overlapped with backward"
    assert d["replicas_identical"] is True
    assert len(d["per_rank"]["loss"]) == 2 and all(v == v for v in d["per_rank"]["loss"])
    assert len(d["per_rank"]["hbm_peak_gb"]) == 2 and all(10 < v < 60 for v in d["per_rank"]["hbm_peak_gb"])
    assert len(d["per_rank_ms"]) == 2
    assert {"overlap_ms", "no_overlap_ms", "no_allreduce_ms", "allreduce_exposed_ms", "bucket_mb"} <= set(d["comm"])
    assert d["comm"]["bucket_mb"] == 32.0
    assert d["dist_backend"] == "gloo" and d["rccl_world"] == 0        # the rehearsal's backend; the node's line must say nccl / N


@pytest.mark.timeout(600)
def test_two_ranks_syncbn_equals_one_rank():
    """tests/dist_worker.py: replicas bit-identical after 2 steps; 2-rank SyncBN == 1 rank on the concatenated batch
    (ops._gather_partials over a real process group)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _torchrun(os.path.join(ROOT, "tests", "dist_worker.py"))
    if out.returncode != 0:
        print(out.stdout[-3000:])
        print(out.stderr[-6000:])
    assert out.returncode == 0
    assert "DIST_OK" in out.stdout, out.stdout[-1500:]


@pytest.mark.timeout(600)
def test_bench_one_rank_over_rccl():
    """The RCCL code path before the 8-GPU node sees it: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with the
    DEFAULT backend (nccl = RCCL; no ONET_DIST_BACKEND / ONET_FORCE_LOCAL_RANK in the environment) -- init_process_group(nccl,
    device_id=...), broadcast of the flat parameters and BatchNorm buffers, the bucketed asynchronous all-reduces issued from the
    backward hooks on RCCL's stream, the device-tensor MIN all-reduce that settles the gloo side group, the all-gathered per-rank
    times and the with / without-overlap loops behind the timed region."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("ONET_DIST_BACKEND", "ONET_FORCE_LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "4", "--size", "64",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--bucket-mb", "8"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["dist_backend"] == "nccl" and d["rccl_world"] == 1 and d["n_gpus"] == 1
    assert d["replicas_identical"] is True and len(d["per_rank"]["hbm_peak_gb"]) == 1
    assert "overlapped with backward" in d["config"]["grad_allreduce"]
    assert len(d["per_rank_ms"]) == 1 and d["per_rank_ms"][0] > 0
    assert {"overlap_ms", "no_overlap_ms", "no_allreduce_ms", "allreduce_exposed_ms"} <= set(d["comm"])
    assert d["config"]["per_step_host_sync"] is False and d["config"]["inputs_resident"] is True
    assert "generated on the GPU" in d["data_source"]
    assert d["value"] > 0 and d["loss"] == d["loss"]


def test_bench_exits_nonzero_on_a_failure_inside_main():
    """No in-process restart: any exception in a rank's main() (an RCCL error included) ends the process with code 1.
    Provoked here by a launcher environment that contradicts --gpus (RANK set, so bench.py does not launch ranks itself)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--no-cpu-baseline"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                  MASTER_PORT=str(_free_port()), ONET_DIST_BACKEND="gloo"))
    assert out.returncode != 0
    assert "WORLD_SIZE=1" in out.stderr


@pytest.mark.timeout(600)
def test_bench_refuses_a_group_that_is_not_rccl_of_the_requested_size():
    """`--gpus N` must mean N ranks over RCCL: outside the one-GPU rehearsal (ONET_FORCE_LOCAL_RANK) a process group on another
    backend ends every rank with a non-zero code instead of printing a line that looks like a multi-GPU measurement."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k != "ONET_FORCE_LOCAL_RANK"}
    env.update(ONET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "4", "--size", "64",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-comm-curve"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode != 0
    assert "not an RCCL run of the requested size" in out.stderr, out.stderr[-2000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the GPU-free parent starts torch.distributed.run as a child,
    relays rank 0's single JSON line and the exit code (two ranks sharing cuda:0 over gloo, as above)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ONET_DIST_BACKEND="gloo", ONET_FORCE_LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--bucket-mb", "8", "--batch", "4", "--size", "64"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert "secondary" not in d and "cpu_baseline" not in d          # N = 1 only
