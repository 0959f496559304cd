"""Worker of tests/test_gpu_dist.py::test_two_ranks_syncbn_equals_one_rank (run under torch.distributed.run, 2 ranks
sharing cuda:0 over gloo): the data-parallel training path of onet_amd.trainer with SyncBN over a REAL process group.

Checks, after 2 steps of (zero_grad, forward, loss, backward, bucketed all-reduce, fused Adam):
  * parameters (the flat buffer) and every BatchNorm buffer are bit-identical on the two ranks;
  * with SyncBN the 2-rank run equals ONE rank on the concatenated batch: loss, first-step gradient (all-reduced
    sum / N), running statistics."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from onet_amd import Onet, ops
    from onet_amd.trainer import FlatAdam, init_distributed, shard_batch
    from oracle import onet_oracle as orc
    rank, world, local = init_distributed("nccl")
    assert world == 2
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ops.LAZY_NAN_CHECK = True
    B, C, H, W = 4, 1, 64, 64
    X = orc.det_input(B, C, H, W, seed=31).to(dev)

    def build(seed_offset):
        torch.manual_seed(7 + seed_offset)                 # ranks start DIFFERENT on purpose: broadcast must fix it
        m = Onet(in_chns=C, binit=True, bshare=True).to(dev).train()
        return m

    def step(m, opt, x):
        opt.zero_grad()
        Lt, Vt, Ld, Vd, S = m(x)
        loss = m.compute_loss(Lt, S[:, 0:1], Ld, S[:, 1:2])
        loss.backward()
        return loss

    # ---- two ranks, SyncBN, overlapped buckets
    # (SyncBN takes the stem through the direct kernel + all-gathered partials; the 1-rank comparison run below must take the same
    # stem kernel: z differs by 1e-7 between the two, which flips the sign of rounding-level gradients of the 9-element stem filters,
    # and Adam turns a sign flip into a full lr-sized step -- 1e-3 of the stem BatchNorm's running mean after two steps)
    # For the same reason both runs stay on the fp32-MFMA kernels (Settings(split=False)): the split-bf16 kernels carry 16-bit
    # operands (8e-7 rms per layer against 2e-7), their unit / tile partition depends on the per-rank batch, and every gradient
    # element whose sign that moves is a full lr-sized difference after Adam's first step.  What is compared here is the
    # data-parallel mechanics; the split kernels under two ranks run in test_bench_two_ranks_sharing_one_gpu_over_gloo (default dispatch).
    ops.STEM_FUSED = False
    ops.set_sync_bn(True)
    m = build(rank)
    m.settings = ops.Settings(split=False)
    opt = FlatAdam(m, lr=1e-4, world_size=world)
    opt.broadcast_params(0)
    opt.enable_overlap(bucket_mb=8)
    xs = shard_batch(X, rank, world)
    loss1 = step(m, opt, xs)
    opt.all_reduce_grads()                                  # finish the buckets so that the gradient can be looked at
    g1 = opt.gflat.clone() / world
    for bk in opt._buckets:                                 # step() must not reduce a second time
        bk.update(launched=True, work=None)
    opt.step()
    loss2 = step(m, opt, xs)
    opt.step()
    torch.cuda.synchronize()
    flats = [torch.empty_like(opt.flat) for _ in range(world)]
    dist.all_gather(flats, opt.flat)
    assert torch.equal(flats[0], flats[1]), "parameters differ between ranks after 2 steps"
    for n, b in m.named_buffers():
        both = [torch.empty_like(b) for _ in range(world)]
        dist.all_gather(both, b.contiguous())
        assert torch.equal(both[0], both[1]), f"buffer {n} differs between ranks"
    lsum = torch.stack([loss1.detach(), loss2.detach()]).clone()
    dist.all_reduce(lsum)
    lsum /= world                                           # mean of the shard means = mean over the global batch

    # ---- one rank on the concatenated batch (no collective in it: both ranks compute it, rank 0 compares)
    ops.set_sync_bn(False)
    m1 = build(0)
    m1.settings = ops.Settings(split=False)
    opt1 = FlatAdam(m1, lr=1e-4, world_size=1)
    opt1.all_reduce_grads = lambda: None                    # a single replica inside an initialised group: nothing to reduce
    r1 = step(m1, opt1, X)
    g_ref = opt1.gflat.clone()
    opt1.step()
    r2 = step(m1, opt1, X)
    opt1.step()
    torch.cuda.synchronize()
    if rank == 0:
        assert abs(float(lsum[0]) - float(r1)) <= 1e-5 * abs(float(r1)), (float(lsum[0]), float(r1))
        e = float((g1 - g_ref).norm() / g_ref.norm())
        assert e <= 2e-3, f"2-rank SyncBN gradient vs 1-rank gradient: {e}"
        assert abs(float(lsum[1]) - float(r2)) <= 1e-4 * abs(float(r2)), (float(lsum[1]), float(r2))
        for (n, b), (_, b1) in zip(m.named_buffers(), m1.named_buffers()):
            if b.is_floating_point():
                assert torch.allclose(b, b1, rtol=1e-4, atol=1e-6), n
            else:
                assert torch.equal(b, b1), n
        dp = float((opt.flat - opt1.flat).norm() / opt1.flat.norm())
        assert dp <= 1e-4, dp        # two Adam steps of lr 1e-4: an update differs where a gradient is at rounding level
        print("DIST_OK grad_err=%.2e param_err=%.2e" % (e, dp), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
