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


@pytest.mark.timeout(600)
def test_bench_two_ranks_sharing_one_gpu_over_gloo():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, ONET_DIST_BACKEND="gloo", ONET_FORCE_LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "4", "--size", "64", "--no-cpu-baseline", "--bucket-mb", "8"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    assert "overlapped with backward" in d["config"]["grad_allreduce"]
    assert d["value"] > 0 and d["loss"] == d["loss"]               # finite
