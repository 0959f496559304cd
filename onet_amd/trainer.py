"""Training harness for the HIP Onet: the build's counterpart of the reference's training loops
(harness contract) plus the data-parallel layer the reference does not have.

* per step (TS:209-219, TZ:110-121):  zero_grad -> X.to(device) -> forward -> slice S ->
  compute_loss -> backward -> [RCCL all-reduce of the flat gradient] -> Adam -> loss.item()
* sim-clutter schedule (TS:181-182, TS:248-249): Adam(lr=5e-6), lr *= 0.5 at epochs 100, 200, ...
* ZY-3 schedule (TZ:89-90, TZ:128): Adam(lr=1e-4) + CosineAnnealingWarmRestarts(T_0=300, T_mult=2,
  eta_min=1e-6) stepped once per epoch
* checkpoints (TS:255-266, TZ:145-149): {'net': state_dict (232 keys), 'epoch' | 'save_epoch': e}

Data parallelism: one process per GPU, each rank takes B/N images of the batch,
full weight replica; ONE all-reduce (sum) of the 31 036 416-element flat fp32 gradient buffer per
step over RCCL/xGMI, folded into the fused Adam launch as grad_scale = 1/N.  BatchNorm statistics
are local to the rank (PyTorch-DDP semantics)."""
from __future__ import annotations

import math
import os
import time

import torch
import torch.distributed as dist

from . import ops
from .modules import invalidate_packed


# ----------------------------------------------------------------------------- schedules
def sim_lr(epoch: int, base_lr: float = 5e-6) -> float:
    """learning rate in force DURING `epoch` of the sim-clutter loop (TS:248-249): halved after
    epochs 100, 200, ... have finished."""
    return base_lr * (0.5 ** max(0, (epoch - 1) // 100)) if epoch > 0 else base_lr


def cosine_warm_restarts_lr(epoch: int, base_lr: float = 1e-4, T_0: int = 300, T_mult: int = 2,
                            eta_min: float = 1e-6) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingWarmRestarts after `epoch` scheduler.step() calls (TZ:89-90,128)."""
    t, Ti = epoch, T_0
    while t >= Ti:
        t -= Ti
        Ti *= T_mult
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / Ti)) / 2


# ----------------------------------------------------------------------------- flat Adam
class FlatAdam:
    """torch.optim.Adam semantics (TS:181-182) on ONE flat parameter buffer:
    parameters and gradients of the model are re-pointed into flat fp32 buffers (so the RCCL
    all-reduce is a single call and the update is a single fused HIP launch, onet_adam_step)."""

    def __init__(self, model, lr=5e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None,
                 world_size=1):
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: model has no trainable parameters")
        dev = self.params[0].device
        # 16-byte aligned slots so every parameter view stays float4-addressable
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.gflat[off:off + p.numel()].view_as(p)
                # backward kernels may write this slice directly (ops.grad_slot_if_free) when .grad is None
                p._onet_gslot = (self.gflat, off, p.numel(), tuple(p.shape))
        self.direct_grads = bool(self.gflat.is_cuda)
        invalidate_packed(model)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.step_count = 0           # steps taken by a parameter that never skipped one (torch: state['step'])
        self._steps = None            # per-parameter counts, kept only once some parameter has skipped a step
        self.pg = process_group
        self.world_size = world_size
        self._buckets = None          # set by enable_overlap()
        self.skip_allreduce = False   # measurement only (bench.py's compute-only loop): the replicas drift apart
        # host-side group for the per-step NaN verdict (see _check_nan_all_ranks); created collectively, here
        self._flag_pg = None
        # (also in an initialised group of ONE rank -- `torch.distributed.run --nproc-per-node 1` -- so that a one-GPU box runs
        # the same code the 8-GPU node does: tests/test_gpu_dist.py::test_bench_one_rank_over_rccl)
        if dist.is_available() and dist.is_initialized() and (world_size > 1 or dist.get_world_size() == 1):
            err = None
            try:
                self._flag_pg = dist.new_group(backend="gloo")
            except Exception as e:      # no gloo in this build / rendezvous refused on this rank
                err = e
            # the ranks must AGREE on whether the side group exists: one that holds it would wait in the per-step all-reduce for
            # peers that skip it.  One MIN over the default group (a device tensor where that group is RCCL) settles it.
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32,
                              device=dev if dist.get_backend(process_group) == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=process_group)
            if int(ok.item()) == 0:
                import warnings
                warnings.warn("onet_amd: no host-side group for the shared NaN verdict on at least one rank"
                              + (f" (here: {err})" if err is not None else "") + "; a NaN loss raises on its own rank only")
                self._flag_pg = None

    # ------------------------------------------------------------------ bucketed all-reduce overlapped with backward
    def enable_overlap(self, bucket_mb: float = 32.0):
        """instead of ONE all-reduce after backward, reduce contiguous slices of the flat gradient
        buffer as soon as backward has produced them.  Buckets are cut from the END of the parameter list
        (autograd reaches the last layers first), each at least `bucket_mb` MB; a bucket's all-reduce is issued
        asynchronously from the post-accumulate-grad hook of its last missing parameter -- with shared weights
        (`dwnu is topu`) autograd sums both passes' contributions before that hook fires once -- and runs on the
        collective library's own stream while backward continues.  `step()` waits for all of them.  The reduced
        values are identical to the single all-reduce (same elementwise sums), so is the update."""
        if self._buckets is not None:
            return
        cap = max(1, int(bucket_mb * 2 ** 20 / 4))
        self._buckets, end, members = [], self.numel, []
        for i in reversed(range(len(self.params))):
            members.append(i)
            start = self.offsets[i]
            if end - start >= cap or i == 0:
                self._buckets.append({"start": start, "end": end, "params": members, "pending": len(members),
                                      "work": None, "launched": False})
                end, members = start, []
        self._bucket_of = {i: b for b, bk in enumerate(self._buckets) for i in bk["params"]}
        self._hooks = [p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))
                       for i, p in enumerate(self.params)]

    def disable_overlap(self):
        """Back to ONE all-reduce of the flat buffer after backward (pending asynchronous buckets are waited for first)."""
        if self._buckets is None:
            return
        for bk in self._buckets:
            if bk["work"] is not None:
                bk["work"].wait()
        for h in self._hooks:
            h.remove()
        self._buckets = self._hooks = None

    def _launch_bucket(self, bk, async_op=True):
        view = self.gflat[bk["start"]:bk["end"]]
        bk["work"] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        bk["launched"] = True

    def _on_grad(self, i):
        p, off = self.params[i], self.offsets[i]
        if p.grad is not None and p.grad.data_ptr() != self.gflat.data_ptr() + 4 * off:
            view = self.gflat[off:off + p.numel()].view_as(p)      # someone detached .grad: put it back first
            view.copy_(p.grad)
            p.grad = view
        bk = self._buckets[self._bucket_of[i]]
        bk["pending"] -= 1
        if bk["pending"] == 0 and not bk["launched"] and not self.skip_allreduce and dist.is_available() and dist.is_initialized():
            self._launch_bucket(bk)

    def _finish_overlap(self):
        for bk in self._buckets:
            if not bk["launched"]:                  # a parameter that got no gradient this step kept it open
                self._launch_bucket(bk, async_op=False)
            elif bk["work"] is not None:
                bk["work"].wait()
            bk.update(pending=len(bk["params"]), work=None, launched=False)

    def zero_grad(self, set_to_none: bool = False):
        if self.gflat.is_cuda:
            ops.fill(self.gflat, 0.0)
        else:                       # host-side tests of the flat-buffer / all-reduce plumbing only
            self.gflat.zero_()
        if self.direct_grads:
            # .grad = None: the first gradient of a parameter is WRITTEN into its slice of the (zeroed) flat buffer by
            # the producing kernel and adopted by autograd as-is; a second one (two-pass mode) is added by autograd
            for p in self.params:
                p.grad = None
                p._onet_gslot_taken = False
        else:
            for p, off in zip(self.params, self.offsets):   # re-attach if someone set .grad = None
                if p.grad is None or p.grad.data_ptr() != self.gflat.data_ptr() + 4 * off:
                    p.grad = self.gflat[off:off + p.numel()].view_as(p)
        if self._buckets is not None:
            for bk in self._buckets:
                bk.update(pending=len(bk["params"]), work=None, launched=False)

    def all_reduce_grads(self):
        if self.skip_allreduce:
            if self._buckets is not None:
                for bk in self._buckets:
                    bk.update(pending=len(bk["params"]), work=None, launched=False)
            return
        if self.world_size > 1 or (dist.is_available() and dist.is_initialized()):
            if self._buckets is not None:
                self._finish_overlap()
            else:
                dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=self.pg)

    def _gather_stray_grads(self):
        """`model.zero_grad()` (set_to_none) detaches .grad from the flat buffer: copy such grads back."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is not None and p.grad.data_ptr() != self.gflat.data_ptr() + 4 * off:
                view = self.gflat[off:off + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view

    def _check_nan_all_ranks(self):
        """The loss's OV:234 verdict (Settings.lazy_nan), BEFORE the update is applied.  With several ranks the verdict
        is shared first (one byte over a host-side gloo group), so that every rank raises in the same step instead of one rank
        leaving its peers waiting in the all-reduce.  Cost: `check_deferred_nan` waits for the event behind the loss kernel and
        the host collective then lines the ranks' hosts up once per step -- the backward launches are already queued by then,
        so the devices keep running, but the hosts cannot run further ahead than one step."""
        try:
            ops.check_deferred_nan()
            bad = None
        except AssertionError as e:
            bad = e
        if self._flag_pg is not None:
            flag = torch.tensor([0 if bad is None else 1], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._flag_pg)
            if bad is None and int(flag.item()):
                bad = AssertionError("jsd is NaN (on another rank)")
        if bad is not None:
            if self._buckets is not None:            # leave no asynchronous all-reduce behind
                for bk in self._buckets:
                    if bk["work"] is not None:
                        bk["work"].wait()
                    bk.update(pending=len(bk["params"]), work=None, launched=False)
            raise bad

    def _segments_with_grads(self):
        """torch.optim.Adam skips parameters whose .grad is None (frozen or unused branches keep p, m and v untouched) and
        bias-corrects every parameter with ITS OWN step count (state['step']).  -> [(start, end, step)]: element ranges of the
        maximal runs of parameters that have a gradient and share a step count, with the count this update is their n-th;
        one range over the whole buffer as long as no parameter has ever skipped a step."""
        have = [p.grad is not None for p in self.params]
        if self._steps is None:
            if all(have):
                self.step_count += 1
                return [(0, self.numel, self.step_count)]
            self._steps = [self.step_count] * len(self.params)
        segs, start, cur = [], None, None
        for i, h in enumerate(have):
            if h:
                self._steps[i] += 1
            st = self._steps[i] if h else None
            if start is not None and st != cur:
                segs.append((start, self.offsets[i], cur))
                start = None
            if h and start is None:
                start, cur = self.offsets[i], st
        if start is not None:
            segs.append((start, self.numel, cur))
        self.step_count = max(self._steps)
        return segs

    def step(self):
        self._check_nan_all_ranks()
        g = self.param_groups[0]
        segs = self._segments_with_grads()
        self._gather_stray_grads()
        self.all_reduce_grads()
        for a, b, n in segs:
            ops.adam_step(self.flat[a:b], self.gflat[a:b], self.m[a:b], self.v[a:b], g["lr"], g["betas"][0], g["betas"][1],
                          g["eps"], g["weight_decay"], n, grad_scale=1.0 / self.world_size)
        invalidate_packed(self.model)

    def broadcast_params(self, src=0):
        if self.world_size > 1 or (dist.is_available() and dist.is_initialized()):
            dist.broadcast(self.flat, src=src, group=self.pg)
            for b in self.model.buffers():
                dist.broadcast(b, src=src, group=self.pg)
            invalidate_packed(self.model)


# ----------------------------------------------------------------------------- distributed
def init_distributed(backend: str | None = None):
    """-> (rank, world_size, local_rank).  Reads the torchrun environment; single process otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("ONET_FORCE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))
    # initialise whenever a launcher set RANK (also for world == 1: it exercises the RCCL path on one GPU)
    if "RANK" in os.environ and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("ONET_DIST_BACKEND", backend)     # rehearsal: N ranks sharing ONE GPU over gloo
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(X, rank: int, world: int, drop_remainder: bool = False):
    """rank r takes X[r*B/N:(r+1)*B/N] (equal shards; the mean of shard means is the global mean).
    drop_remainder: the reference's DataLoaders keep the last, smaller batch of an epoch (no drop_last, DS:144); equal
    shards need B % N == 0, so the loop drops the B % N trailing images of such a batch instead of aborting."""
    B = X.shape[0]
    if B % world:
        if not drop_remainder or B < world:
            raise ValueError(f"batch {B} not divisible by world size {world}")
        B -= B % world
    per = B // world
    return X[rank * per:(rank + 1) * per]


# ----------------------------------------------------------------------------- the step and the loops
def train_step(onet, opt, X):
    """One iteration exactly as TS:210-219 orders it.  Returns the loss tensor (0-dim, on device)."""
    opt.zero_grad()
    Lt, Vt, Ld, Vd, S = onet(X)
    St = S[:, 0, :, :].unsqueeze(dim=1)
    Sd = S[:, 1, :, :].unsqueeze(dim=1)
    loss = onet.compute_loss(Lt, St, Ld, Sd)
    loss.backward()
    opt.step()
    return loss


def save_checkpoint(onet, path, epoch, key="epoch"):
    """{'net': state_dict, 'epoch': e} (TS:264-266) or {'net', 'save_epoch'} (TZ:145-149): onet_amd.io."""
    from . import io
    io.save_checkpoint(onet, path, epoch, zy3=(key == "save_epoch"))


def load_checkpoint(onet, path, map_location=None):
    """resume = load_state_dict(torch.load(f)['net']) (TZ:77-82, TS:492-493): onet_amd.io."""
    from . import io
    return io.load_checkpoint(onet, path, map_location=map_location)


def fit(onet, train_loader, device, epochs, schedule="sim", base_lr=None, eval_fn=None, eval_every=None,
        out_root=None, model_name="Onet", fused_adam=True, rank=0, world=1, log=print):
    """Epoch loop of TS:201-266 (schedule='sim') / TZ:99-153 (schedule='zy3').
    `train_loader` yields (X, ...) with X a CPU or GPU float32 [B,C,H,W] tensor in [0,1]."""
    before = getattr(onet, "settings", None)
    if fused_adam and before is not None:
        # OV:234's assertion is raised by FlatAdam.step(), before the update, without a mid-step sync (this model only)
        onet.settings = before.replace(lazy_nan=True)
    try:
        return _fit(onet, train_loader, device, epochs, schedule, base_lr, eval_fn, eval_every, out_root, model_name,
                    fused_adam, rank, world, log)
    finally:
        if before is not None:
            onet.settings = before


def _fit(onet, train_loader, device, epochs, schedule, base_lr, eval_fn, eval_every, out_root, model_name, fused_adam,
         rank, world, log):
    if base_lr is None:
        base_lr = 5e-6 if schedule == "sim" else 1e-4
    if fused_adam:
        opt = FlatAdam(onet, lr=base_lr, world_size=world)
        opt.broadcast_params(0)
        if world > 1:
            opt.enable_overlap()
    else:
        opt = torch.optim.Adam(onet.parameters(), lr=base_lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    if eval_every is None:
        eval_every = 50 if schedule == "sim" else 1
    history = []
    for epoch in range(epochs):
        onet.train()
        lr = sim_lr(epoch, base_lr) if schedule == "sim" else cosine_warm_restarts_lr(epoch, base_lr)
        opt.param_groups[0]["lr"] = lr
        losses, n_img, t0 = [], 0, time.time()
        for batch in train_loader:
            X = batch[0] if isinstance(batch, (tuple, list)) else batch
            if X.shape[0] < world:        # an epoch's short last batch with fewer images than ranks (the reference's loaders have
                continue                  # no drop_last): every rank sees the same global batch, so every rank skips it
            X = shard_batch(X, rank, world, drop_remainder=True).to(device, non_blocking=True)
            loss = train_step(onet, opt, X)
            losses.append(loss.item())            # device->host sync every step, as TS:219
            n_img += X.shape[0] * world
        ep_loss = float(sum(losses) / max(1, len(losses)))
        rec = {"epoch": epoch, "loss": ep_loss, "lr": lr, "images_per_s": n_img / max(1e-9, time.time() - t0)}
        if eval_fn is not None and epoch % eval_every == 0:
            onet.eval()
            with torch.no_grad():
                rec["eval"] = eval_fn(onet, epoch)
        history.append(rec)
        if rank == 0:
            log("%s===Epoch: %04d loss: %.5f, lr: %.10f, %.1f img/s" % (model_name, epoch, ep_loss, lr, rec["images_per_s"]))
        last = epoch == epochs - 1
        if out_root and rank == 0 and (last or epoch == 300):          # TS:255 and TZ:141: the final epoch and epoch 300
            save_checkpoint(onet, os.path.join(out_root, "%s_epoch_%d.pytorch" % (model_name, epoch)), epoch,
                            key="epoch" if schedule == "sim" else "save_epoch")
    return history
