"""N>1 data-parallel path on CPU: world_size-2 `gloo` processes exercise the SAME FlatAdam /
shard_batch / all-reduce / broadcast code the RCCL run uses (only the fused Adam launch itself
needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        torch.set_num_threads(2)
        from onet_amd.trainer import FlatAdam, init_distributed, shard_batch
        from oracle import onet_oracle as orc
        r, w, _ = init_distributed("gloo")
        assert (r, w) == (rank, world)

        # (1) flat-buffer all-reduce == sum over ranks; broadcast makes replicas identical
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        opt = FlatAdam(net, lr=1e-3, world_size=world)
        opt.broadcast_params(0)
        ref0 = [torch.zeros_like(p) for p in net.parameters()]
        for t, p in zip(ref0, net.parameters()):
            t.copy_(p.detach())
            dist.broadcast(t, src=0)
            assert torch.equal(t, p.detach()), "params differ from rank 0 after broadcast"
        X = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
        opt.zero_grad()
        net(shard_batch(X, rank, world)).pow(2).sum().backward()
        mine = opt.gflat.clone()
        opt.all_reduce_grads()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.allclose(opt.gflat, sum(both), rtol=1e-6, atol=1e-7)

        # (1b) bucketed all-reduce issued from the backward hooks (overlap path) == the single all-reduce, bit for bit;
        #      tiny buckets so that several are in flight, one parameter left without a gradient on purpose
        net2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
        unused = torch.nn.Parameter(torch.ones(7))
        net2.register_parameter("unused", unused)
        opt2 = FlatAdam(net2, lr=1e-3, world_size=world)
        opt2.broadcast_params(0)
        opt2.zero_grad()
        net2(shard_batch(X, rank, world)).pow(2).sum().backward()
        local = opt2.gflat.clone()
        opt2.all_reduce_grads()
        want = opt2.gflat.clone()
        opt2.enable_overlap(bucket_mb=1e-4)
        assert len(opt2._buckets) >= 3
        for _ in range(2):                                   # twice: the bucket state must reset between steps
            opt2.zero_grad()
            net2(shard_batch(X, rank, world)).pow(2).sum().backward()
            assert any(bk["launched"] for bk in opt2._buckets), "no bucket was reduced during backward"
            opt2.all_reduce_grads()
            assert torch.equal(opt2.gflat, want), (opt2.gflat - want).abs().max()
        assert not torch.equal(local, want)

        # (2) DP semantics on the real model (CPU oracle): averaged shard gradients == the mean of the
        #     per-shard reference runs (local-BN / DDP semantics)
        B, C, H, W = 4, 1, 16, 16
        Xo = orc.det_input(B, C, H, W)
        top = orc.clone_state(orc.det_state_dict(C, 1981))
        _, loss, grads = orc.train_mode_step(shard_batch(Xo, rank, world), top)
        flat = torch.cat([g.reshape(-1) for g in grads.values()])
        dist.all_reduce(flat)
        flat /= world
        acc = None
        for rr in range(world):
            t2 = orc.clone_state(orc.det_state_dict(C, 1981))
            _, _, g2 = orc.train_mode_step(shard_batch(Xo, rr, world), t2)
            f2 = torch.cat([g.reshape(-1) for g in g2.values()])
            acc = f2 if acc is None else acc + f2
        assert torch.allclose(flat, acc / world, rtol=1e-5, atol=1e-8)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise e


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
