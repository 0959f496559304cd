"""torch.autograd glue: each Function is one fused unit of the Onet hot path whose forward
and backward run hand-written HIP kernels through the C ABI (onet_amd.ops).  Autograd only
orchestrates (graph walk, gradient accumulation of multiply-used tensors); it computes nothing
of the model itself.  Reference call sites are cited per Function ("OV" =
source_code/Onet_vanilla_20240606.py)."""
from __future__ import annotations

import torch

from . import ops


def _carries_settings(cls):
    """Class decorator for the Functions below: forward remembers the `ops.Settings` record active on the calling thread (the
    model's, set by `Onet.forward` / `compute_loss`), backward -- run by autograd's own thread, any time later -- re-activates it,
    so that algorithm / precision choices made in backward are the model's and not whatever another model left behind."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kw):
        ctx._onet_settings = ops.active_settings()
        return fwd(ctx, *args, **kw)

    def backward(ctx, *grads):
        with ops.using(getattr(ctx, "_onet_settings", None)):
            return bwd(ctx, *grads)

    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


@_carries_settings
class ConvBNReLUFn(torch.autograd.Function):
    """Conv2d(3x3, pad 1, no bias) -> BatchNorm2d -> ReLU  (OV:47-49 / OV:51-53).

    forward : conv (MFMA; the F(4x4) kernel also emits the BN statistics records) -> finalize -> normalise+ReLU
    backward: BN+ReLU backward (reduce + apply) -> wgrad (MFMA split-K) -> dgrad (MFMA)

    `link_out` / `link_in`: the dict DoubleConv shares between its two units.  The first unit publishes its
    pre-activation and BN coefficients in it; the second unit's backward then folds the first unit's BN-backward reduce
    pass into its own dgrad launch (whose output IS the first unit's `da`) and leaves the records in the dict."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, packed, out, groups=1,
                link_out=None, link_in=None, aux=None, p16=None):
        ops.require_gpu(x, weight, gamma, beta)
        if p16 is not None and (p16.get("x") is not None or p16.get("want")):
            return ConvBNReLUFn._forward_pre(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, packed,
                                             groups, link_out, link_in, p16)
        ctx.pre = False
        # an encoder output that is max-pooled next: a request left in the link dict by UNet.forward (read before the dict is refilled
        # below); the pooled tensor goes back through the same dict for SkipPoolFn
        want_pool = link_out.pop("want_pool", None) if link_out is not None else None
        # normalise on load (ops.norm_on_load_ok, decided by DoubleConv for the pair): as the FIRST unit, this launch computes
        # z and the BatchNorm coefficients and stops there -- the second unit's convolution applies BatchNorm + ReLU in its own
        # operand staging, forward and weight gradient, and a placeholder stands in for the activation in the graph; as the
        # SECOND unit, x is that placeholder and the operand comes from (z, save) of the first
        defer = bool(link_out.pop("defer", False)) if link_out is not None else False
        defer = defer and training and out is None and want_pool is None
        norm = None
        if link_in is not None and link_in.get("deferred"):
            norm = (link_in["z"], link_in["save"])
        # the magnitude slots x's producer left on it (ops.tag_amax): the fp16-split kernel's overflow guard
        # (aux: {"x_amax": slots of x or None} in, {"a_amax": slots of the output} out -- DoubleConv._unit tags the tensors)
        x_amax = aux.get("x_amax") if (aux is not None and norm is None) else None
        # training: the F(4x4) kernel emits the BatchNorm statistics records from its epilogue (cm), where it can
        if norm is not None:
            z, cm = ops.conv3x3_fwd_bn_partials(None, packed, norm=norm)
        else:
            z, cm = ops.conv3x3_fwd_bn_partials(x, packed, amax=x_amax) if training else (ops.conv3x3_auto(x, packed, 0, amax=x_amax), None)
        # `out` is None or a 1-tuple holding a plane-contiguous destination view (kept out of autograd's sight)
        dst = None if out is None else out[0]
        G = groups if (training and groups > 1) else 1
        C = z.shape[1]
        save_all = torch.empty((G, 4, C), dtype=torch.float32, device=z.device)
        Bz, _, Hz, Wz = z.shape
        pooled = None
        # magnitude slots of the activation written here (all statistics groups share them), for the convolution that consumes it
        a_amax = ops.new_amax(z.device) if (not defer and z.is_cuda and ops.split_f16() and ops.split_enabled()
                                            and ops.conv_algo() in ("auto", "split", "bf16")) else None
        if want_pool is not None and ops.FUSE_POOL and Hz % 2 == 0 and Wz % 4 == 0:
            pooled = (torch.empty((Bz, C, Hz // 2, Wz // 2), dtype=torch.float32, device=z.device), None)

        def apply(zg, sv, o, sl):
            """BatchNorm + ReLU of one statistics group (batch slice sl), with the pooled output where it was asked for"""
            nonlocal pooled
            if defer:
                return None
            if pooled is not None:
                if o is None:
                    o = torch.empty_like(zg)
                if ops.bn_relu_apply_pool(zg, sv, o, pooled[0][sl], amax=a_amax):
                    return o
                pooled = None                       # shape not taken: the separate pooling pass runs as before
            return ops.bn_relu_apply(zg, sv, out=o, amax=a_amax)

        if G == 1:
            if training:
                ops.bn_train_coeffs(z, gamma, beta, running_mean, running_var, momentum, eps,
                                    cm=None if cm is None else (cm, 0, cm.shape[1]), save=save_all[0])
            else:
                ops.bn_eval_coeffs(gamma, beta, running_mean, running_var, eps, save=save_all[0])
            a = apply(z, save_all[0], dst, slice(0, Bz))
            if defer:
                a = ops.fp32_placeholder(z.shape, z.device)
        else:
            # twin batch: the G batch slices are separate BatchNorm batches (own statistics, running stats updated
            # slice after slice, exactly as G consecutive forward passes would)
            B = z.shape[0]
            Bg = B // G
            a = dst if dst is not None else (ops.fp32_placeholder(z.shape, z.device) if defer else torch.empty_like(z))
            for g in range(G):
                zg = z[g * Bg:(g + 1) * Bg]
                npg = 0 if cm is None else cm.shape[1] // G
                ops.bn_train_coeffs(zg, gamma, beta, running_mean, running_var, momentum, eps,
                                    cm=None if cm is None else (cm, g * npg, npg), save=save_all[g])
                apply(zg, save_all[g], None if defer else a[g * Bg:(g + 1) * Bg], slice(g * Bg, (g + 1) * Bg))
        if aux is not None:
            aux["a_amax"] = a_amax
        ctx.twin = ops.twin_src_of(x)       # a virtual twin batch (placeholder + (X, bias)): backward re-attaches the tag
        ctx.save_for_backward(x, z, save_all, None if norm is None else norm[0], None if norm is None else norm[1])
        ctx.training = training
        ctx.packed = packed
        ctx.wshape = tuple(weight.shape)
        ctx.params = (weight, gamma, beta)          # for ops.grad_slot_if_free in backward
        ctx.link_out = ctx.link_in = None
        if training and link_out is not None:
            link_out.clear()
            link_out.update(z=z, save=save_all)
            if defer:
                link_out["deferred"] = True
            ctx.link_out = link_out
        if training and link_in is not None and "z" in link_in:
            ctx.link_in = link_in
        if pooled is not None and link_out is not None:
            link_out["pooled"] = pooled
        return a

    @staticmethod
    def _forward_pre(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, packed, groups, link_out,
                     link_in, p16):
        """Pre-split storage (Settings.presplit).  p16 in: "x" = the pre-split form of x (then the convolution, its input gradient
        and weight gradient run on pre-split operands, ops.pre_layer_ok) or None; "want": write the output activation pre-split
        ("out": its destination, e.g. the skip groups of a concat buffer; else allocated); "keep_fp32": write the fp32 activation too
        (it has fp32 readers); with a pooling request in link_out (UNet.forward: {"p16": the pooled tensor is wanted pre-split}).
        p16 out: "a" = the pre-split output.  The graph carries placeholders where no fp32 tensor exists."""
        want_pool = link_out.pop("want_pool", None) if link_out is not None else None
        if link_out is not None:
            link_out.pop("defer", None)
        xP = p16.get("x")
        x_slots = p16.get("x_slots") if xP is not None else None
        if xP is not None:
            z, cm = ops.conv3x3_pre_bn_partials(xP, packed, x_slots)
        else:
            z, cm = ops.conv3x3_fwd_bn_partials(x, packed) if training else (ops.conv3x3_auto(x, packed, 0), None)
        G = groups if (training and groups > 1) else 1
        Bz, C, Hz, Wz = z.shape
        Bg = Bz // G
        save_all = torch.empty((G, 4, C), dtype=torch.float32, device=z.device)
        want = bool(p16.get("want"))
        keep = bool(p16.get("keep_fp32")) or not want or ops.PRESPLIT_KEEP_FP32
        aP = None
        if want:
            aP = p16.get("out")
            if aP is None:
                aP = ops.p16_empty(Bz, C, Hz, Wz, z.device)
        # the network's last unit (round 5): its activation has ONE reader, the head, which forms relu(bn(z)) on load from z and the
        # coefficients -- no activation tensor, no BatchNorm + ReLU pass (a placeholder goes through the graph)
        head = p16.get("head_link")
        defer_head = bool(head is not None and training and G == 2 and not want and want_pool is None and z.dtype == torch.float32
                          and ops.HEAD_NORM and not ops.PRESPLIT_KEEP_FP32)
        a = torch.empty(z.shape, dtype=torch.float32, device=z.device) if (keep and not defer_head) else ops.fp32_placeholder(z.shape, z.device)      # (z may be stored as bf16)
        # magnitude slots: of a pre-split output the BOUND |gamma| sqrt(N - 1) + |beta| (written by the statistics finalize, before
        # the pass that needs it as the guard scale); of an fp32-only output the exact maximum (recorded by the pass that writes
        # it: what bounds a ConvTranspose2d's pre-split output downstream)
        parts2 = ops.p16_parts() == 2
        act_slots = ops.new_amax(z.device) if (want and parts2 and training) else None
        a_amax = ops.new_amax(z.device) if (not want and parts2) else None
        pooled = None
        if want_pool is not None and Hz % 2 == 0 and Wz % 2 == 0:
            if want_pool.get("p16"):
                pooled = (None, None, ops.p16_empty(Bz, C, Hz // 2, Wz // 2, z.device))
            else:
                pooled = (torch.empty((Bz, C, Hz // 2, Wz // 2), dtype=torch.float32, device=z.device), None, None)
        # the statistics groups of a twin batch (consecutive batch slices) go through every pass in ONE launch: the kernels pick a
        # group's coefficients from save_all [G][4][C] by image index
        if training and cm is not None:
            ops.bn_train_coeffs(z[:Bg], gamma, beta, running_mean, running_var, momentum, eps, cm=(cm, 0, cm.shape[1] // G), save=save_all,
                                act_slots=act_slots, groups=G)
        else:
            for g in range(G):
                if training:
                    ops.bn_train_coeffs(z[g * Bg:(g + 1) * Bg], gamma, beta, running_mean, running_var, momentum, eps, save=save_all[g],
                                        act_slots=act_slots)
                else:
                    ops.bn_eval_coeffs(gamma, beta, running_mean, running_var, eps, save=save_all[g])
        gi = Bg if G > 1 else 0
        if defer_head:
            head["z"], head["save"] = z, save_all
            a_amax = None
        elif pooled is not None:
            if not ops.bn_relu_apply_pool_split(z, save_all, aP, a if keep else None, pooled[2], pooled[0], slots=act_slots, group_images=gi):
                raise RuntimeError("onet_amd: pre-split BatchNorm + pooling pass refused a shape ops.pre_layer_ok accepted")
        elif aP is not None:
            ops.bn_relu_apply_split(z, save_all, aP, a=a if keep else None, slots=act_slots, group_images=gi)
        else:
            ops.bn_relu_apply(z, save_all, out=a, amax=a_amax, group_images=gi)
        p16["a"], p16["a_slots"], p16["a_amax"] = aP, act_slots, a_amax
        ctx.x_slots = x_slots
        ctx.up_link = p16.get("up_link")    # (first convolution of a decoder block: see UpConvTCatFn -- the up-sampled half of dx leaves pre-split)
        ctx.twin = ops.twin_src_of(x)       # a virtual twin batch (placeholder + (X, bias)): backward re-attaches the tag
        ctx.save_for_backward(x, z, save_all, xP)
        ctx.pre = True
        ctx.training = training
        ctx.packed = packed
        ctx.wshape = tuple(weight.shape)
        ctx.params = (weight, gamma, beta)
        ctx.link_out = ctx.link_in = None
        if training and link_out is not None:
            link_out.clear()
            link_out.update(z=z, save=save_all)
            ctx.link_out = link_out
        if training and link_in is not None and "z" in link_in:
            ctx.link_in = link_in
        if pooled is not None and link_out is not None:
            link_out["pooled"] = pooled
        return a

    @staticmethod
    def _retag_twin(ctx, x):
        """x as saved by forward; a virtual twin batch gets its (X, bias) tag back should the saved object have lost it."""
        tw = getattr(ctx, "twin", None)
        if tw is not None and x is not None and ops.twin_src_of(x) is None:
            x._onet_twin_src = [tw[0], tw[1], None]
        return x

    @staticmethod
    def _backward_pre(ctx, da):
        x, z, save_all, xP = ctx.saved_tensors
        x = ConvBNReLUFn._retag_twin(ctx, x)
        need_x, need_w, need_g, need_b = ctx.needs_input_grad[:4]
        pw, pg, pb = ctx.params
        aff = (ops.grad_slot_if_free(pg) if need_g else None, ops.grad_slot_if_free(pb) if need_b else None)
        rec4, da_amax = None, None
        lk = ctx.link_out
        if lk is not None and "rec4" in lk:
            rda, rec4, da_amax = lk.pop("da"), lk.pop("rec4", None), lk.pop("da_amax", None)
            lk.pop("rec", None)
            if rda.data_ptr() != da.data_ptr() or rda.shape != da.shape or rda.stride() != da.stride():
                rec4 = da_amax = None
        z_below = save_below = None
        if ctx.link_in is not None:                 # the unit below's (z, save) must not outlive this backward
            z_below, save_below = ctx.link_in.pop("z", None), ctx.link_in.pop("save", None)
        nones = (None,) * 12
        if xP is None:
            # fp32 input (the stem): dz in fp32 for the fp32-input kernels; no input gradient path on pre-split operands
            if need_w and not need_x and x.shape[1] <= 4 and ops.STEM_WGRAD_BN and (not ops.is_placeholder(x) or ops.twin_src_of(x) is not None):
                # the stem: dz has one reader, the weight gradient -- formed on load there, never written
                dw, dgamma, dbeta = ops.stem_wgrad_bn(x, da, z, save_all, ctx.training, ctx.wshape, affine_out=aff, rec4=rec4,
                                                      out=ops.grad_slot_if_free(pw))
                return (None, dw, (dgamma if need_g else None), (dbeta if need_b else None)) + nones
            dz, dgamma, dbeta = ops.bn_relu_bwd_groups(da, z, save_all, ctx.training, affine_out=aff, rec4=rec4)
            dw = ops.conv3x3_wgrad_auto(x, dz, ctx.wshape, out=ops.grad_slot_if_free(pw)) if need_w else None
            dx = ops.conv3x3_auto(dz, ctx.packed, 1) if need_x else None
            return (dx, dw, (dgamma if need_g else None), (dbeta if need_b else None)) + nones
        dzP, dz_slots, dgamma, dbeta = ops.bn_relu_bwd_split(da, z, save_all, ctx.training, need_affine_grads=(need_g or need_b),
                                                             affine_out=aff, rec4=rec4, da_amax=da_amax)
        s1, s2, sc = ops._slots3(getattr(ctx, "x_slots", None))
        dw = ops.conv3x3_split_wgrad_pre(xP, dzP, ctx.wshape, out=ops.grad_slot_if_free(pw), x_slots=s1, dz_slots=dz_slots, x_slots2=s2,
                                         split_ch=sc) if need_w else None
        dpack = ctx.packed.get_pack("split" if dzP.shape[3] == 2 else "plain16")[1]
        dx = None
        if need_x and z_below is not None and save_below is not None and ctx.training:
            # the unit below's BatchNorm-backward reduce rides in this launch's epilogue (its da IS this dx): the records (and max |da|)
            # go back through the shared dict, where that unit's backward looks for them
            fused = ops.conv3x3_split_dgrad_pre_bnreduce(dzP, dpack, ctx.wshape[1], z_below, save_below, slots=dz_slots,
                                                         always=dz_slots is not None, want_amax=ops.p16_parts() == 2)
            if fused is not None:
                dx, rec_below, am_below = fused
                ctx.link_in["da"], ctx.link_in["rec4"] = dx, rec_below
                if am_below is not None:
                    ctx.link_in["da_amax"] = am_below
        up = getattr(ctx, "up_link", None)
        if need_x and dx is None and up is not None and up.get("want") and (dz_slots is not None or dzP.shape[3] == 1):
            # the input of this convolution is a concat buffer whose up-sampled half came out of a ConvTranspose2d: that half of dx is
            # read only by the ConvTranspose2d backward GEMMs, and leaves this launch in THEIR operand form (fp16 hi | mid slots scaled by
            # a bound from dz's bound and the weights); the fp32 tensor keeps its shape, its upper channels are never written
            C2, Ct = up["want"]
            bound = ops.conv3x3_dgrad_bound(pw, dz_slots, C2) if dzP.shape[3] == 2 else None      # (plain bf16 parts are not scaled)
            dx, dyP = ops.conv3x3_split_dgrad_pre_slots(dzP, dpack, ctx.wshape[1], C2, bound, slots=dz_slots, always=dz_slots is not None)
            up["dyP"], up["dy_slots"], up["da"] = dyP, bound, dx
        if need_x and dx is None:
            dx = ops.conv3x3_split_pre(dzP, dpack, ctx.wshape[1], slots=dz_slots, always=dz_slots is not None)
        return (dx, dw, (dgamma if need_g else None), (dbeta if need_b else None)) + nones

    @staticmethod
    def backward(ctx, da):
        if ctx.pre:
            return ConvBNReLUFn._backward_pre(ctx, da)
        x, z, save_all, nz, nsave = ctx.saved_tensors
        x = ConvBNReLUFn._retag_twin(ctx, x)
        need_x, need_w, need_g, need_b = ctx.needs_input_grad[:4]
        pw, pg, pb = ctx.params
        aff = (ops.grad_slot_if_free(pg) if need_g else None, ops.grad_slot_if_free(pb) if need_b else None)
        G = save_all.shape[0]
        # the reduce records of THIS layer, if the unit above wrote them while producing exactly this `da`
        rec = rec4 = None
        lk = ctx.link_out
        if lk is not None and ("rec" in lk or "rec4" in lk):
            rda, rec, rec4 = lk.pop("da"), lk.pop("rec", None), lk.pop("rec4", None)
            if rda.data_ptr() != da.data_ptr() or rda.shape != da.shape or rda.stride() != da.stride():
                rec = rec4 = None                   # autograd handed over something else (another consumer, a hook)
        Bz, Cz, Hz, Wz = z.shape
        # fp16-split gradient kernels (Settings.grad_f16): the BatchNorm backward records max |dz| on its way out, and the input- and
        # weight-gradient kernels that consume dz scale it by a power of two chosen from that before splitting it into fp16 parts
        dz_amax = ops.new_amax(z.device) if (ops.grad_f16() and ops.conv_algo() in ("auto", "split", "bf16")
                                             and ops.split_enabled() and Wz >= 16) else None
        if G == 1:
            dz, dgamma, dbeta = ops.bn_relu_bwd(da, z, save_all[0], ctx.training, need_affine_grads=(need_g or need_b),
                                                affine_out=aff, red=None if rec is None else (rec, 0, rec.shape[1]),
                                                red4=None if rec4 is None else (rec4, 0, rec4.shape[0]), amax=dz_amax)
        else:
            Bg = z.shape[0] // G
            dz = torch.empty_like(z)
            dgamma = dbeta = None
            for g in range(G):
                sl = slice(g * Bg, (g + 1) * Bg)
                npg = 0 if rec is None else rec.shape[1] // G
                np4 = 0 if rec4 is None else rec4.shape[0] // G
                _, dgamma, dbeta = ops.bn_relu_bwd(da[sl], z[sl], save_all[g], ctx.training, need_affine_grads=True,
                                                   out=dz[sl], acc=None if g == 0 else (dgamma, dbeta),
                                                   affine_out=aff if g == 0 else None,
                                                   red=None if rec is None else (rec, g * npg, npg),
                                                   red4=None if rec4 is None else (rec4, g * np4, np4), amax=dz_amax)
        if need_w and nz is not None:      # normalise on load: the operand is relu(bn(nz)) of the unit below, applied in the staging
            dw = ops.conv3x3_split_wgrad(nz, dz, ctx.wshape, out=ops.grad_slot_if_free(pw), norm=nsave, dz_amax=dz_amax)
        else:
            dw = ops.conv3x3_wgrad_auto(x, dz, ctx.wshape, out=ops.grad_slot_if_free(pw), dz_amax=dz_amax) if need_w else None
        dx = None
        if need_x:
            lk = ctx.link_in
            fused = None
            if lk is not None and "z" in lk:
                # (z, save) leave the dict here: the graph -- and with it this ctx and the dict -- lives as long as the
                # caller holds the loss tensor, i.e. into the next step; a whole set of pre-activations must not
                fused = ops.conv3x3_dgrad_bnreduce(dz, ctx.packed, lk.pop("z"), lk.pop("save"))
            if fused is not None:
                dx, r = fused
                lk["da"], lk["rec"] = dx, r
            else:
                dx = ops.conv3x3_auto(dz, ctx.packed, 1, amax=dz_amax)
        if ctx.link_in is not None:                 # whatever path ran: the unit below's (z, save) must not outlive this backward
            ctx.link_in.pop("z", None)
            ctx.link_in.pop("save", None)
        return (dx, dw, (dgamma if need_g else None), (dbeta if need_b else None)) + (None,) * 12


@_carries_settings
class Conv3x3Fn(torch.autograd.Function):
    """Bare 3x3 convolution (standalone use of the conv parameter holder)."""

    @staticmethod
    def forward(ctx, x, weight, packed):
        ops.require_gpu(x, weight)
        z = ops.conv3x3_auto(x, packed, 0)
        ctx.save_for_backward(x)
        ctx.packed = packed
        ctx.wshape = tuple(weight.shape)
        return z

    @staticmethod
    def backward(ctx, dz):
        (x,) = ctx.saved_tensors
        dw = ops.conv3x3_wgrad_auto(x, dz, ctx.wshape) if ctx.needs_input_grad[1] else None
        dx = ops.conv3x3_auto(dz, ctx.packed, 1) if ctx.needs_input_grad[0] else None
        return dx, dw, None


@_carries_settings
class BNReLUFn(torch.autograd.Function):
    """BatchNorm2d -> ReLU on an existing pre-activation (standalone use)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, training, momentum, eps):
        ops.require_gpu(z, gamma, beta)
        if training:
            save = ops.bn_train_coeffs(z, gamma, beta, running_mean, running_var, momentum, eps)
        else:
            save = ops.bn_eval_coeffs(gamma, beta, running_mean, running_var, eps)
        a = ops.bn_relu_apply(z, save)
        ctx.save_for_backward(z, save)
        ctx.training = training
        return a

    @staticmethod
    def backward(ctx, da):
        z, save = ctx.saved_tensors
        dz, dgamma, dbeta = ops.bn_relu_bwd(da, z, save, ctx.training, True)
        return dz, dgamma, dbeta, None, None, None, None, None


@_carries_settings
class MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2)  (OV:67)."""

    @staticmethod
    def forward(ctx, x):
        ops.require_gpu(x)
        y = ops.maxpool2_fwd(x)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(x, dy)


@_carries_settings
class SkipPoolFn(torch.autograd.Function):
    """x -> (x, maxpool2(x)[, x]) for an encoder output that feeds both the next Down (OV:67) and an Up's skip concat
    (OV:100) -- and, for the first one, also leaves the U-Net as its first output (OV:152): backward sums the two or
    three gradients inside the pooling-backward kernel (one pass instead of pool backward + autograd's full-tensor
    adds).  `link`: the dict the producing Conv-BN-ReLU unit published its (z, save) in; the sum written here IS that
    unit's activation gradient, so its BatchNorm-backward reduce records are taken in the same kernel and left in the
    dict for the unit's backward."""

    @staticmethod
    def forward(ctx, x, returned=False, link=None, carry=None):
        """carry: a dict that takes the pooled tensor's pre-split form back to the caller ("yP"), where its producer wrote one."""
        ops.require_gpu(x)
        pooled = link.pop("pooled", None) if link is not None else None
        if pooled is not None and len(pooled) == 3:
            # pre-split storage: the producing pass wrote the pooled tensor pre-split (pooled[2]) or in fp32 (pooled[0])
            shp = (x.shape[0], x.shape[1], x.shape[2] // 2, x.shape[3] // 2)
            y = pooled[0] if pooled[0] is not None else ops.fp32_placeholder(shp, x.device)
            if carry is not None:
                carry["yP"] = pooled[2]
            ctx.pre = True
        elif pooled is not None and pooled[0] is not None and tuple(pooled[0].shape) == (x.shape[0], x.shape[1], x.shape[2] // 2, x.shape[3] // 2):
            y = pooled[0]                   # the producing BatchNorm + ReLU pass already wrote the pooled tensor (onet_bn_relu_apply_pool)
        else:
            y = ops.maxpool2_fwd(x)
        ctx.save_for_backward(x)
        ctx.link = link if (link is not None and "z" in link) else None
        return (x.view_as(x), y, x.view_as(x)) if returned else (x.view_as(x), y)

    @staticmethod
    def backward(ctx, g_skip, g_pool, g_ret=None):
        (x,) = ctx.saved_tensors
        if g_pool is None:
            gs = [g for g in (g_skip, g_ret) if g is not None]
            return (sum(gs[1:], gs[0]) if gs else None), None, None, None
        lk = ctx.link
        if lk is not None and "z" in lk:
            am = ops.new_amax(g_pool.device) if (getattr(ctx, "pre", False) and ops.p16_parts() == 2) else None
            dx, part2 = ops.maxpool2_bwd(x, g_pool, add=g_skip, add2=g_ret, bn=(lk.pop("z"), lk.pop("save")), dx_amax=am)
            if part2 is not None:
                lk["da"], lk["rec4"] = dx, part2
                if am is not None:
                    lk["da_amax"] = am
            return dx, None, None, None
        return ops.maxpool2_bwd(x, g_pool, add=g_skip, add2=g_ret), None, None, None


def _pad_offsets(x1_hw, x2_hw):
    """F.pad amounts of OV:92-96: (top, left); right/bottom are implied by the skip size."""
    dY = x2_hw[0] - 2 * x1_hw[0]
    dX = x2_hw[1] - 2 * x1_hw[1]
    if dY < 0 or dX < 0:
        raise ValueError("Up: skip tensor smaller than the upsampled tensor (negative F.pad is not supported)")
    return dY // 2, dX // 2


@_carries_settings
class UpConvTCatFn(torch.autograd.Function):
    """ConvTranspose2d(C, C/2, k=2, s=2) + F.pad + cat([skip, up], 1)  (OV:86, OV:91-100).

    The transposed conv is a 1x1 MFMA convolution to 4*Ct sub-pixel channels followed by a
    pixel-shuffle that writes straight into the second half of the concat buffer."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, packed, cat_holder=None, p16=None):
        """p16 (pre-split storage): {"catP": the pre-split concat buffer, skip groups written by the encoder}: the up-sampled groups
        are written here and NO fp32 concat exists (a placeholder goes through the graph)."""
        ops.require_gpu(x1, x2, weight, bias)
        wp_fused, wp_dgrad = packed               # (ops.PackedT hands out lazy handles: a classic fp32 pack is made only if it is used)
        B, Cin, h, w = x1.shape
        Ct = weight.shape[1]
        C2, Ho, Wo = x2.shape[1], x2.shape[2], x2.shape[3]
        pt, pl = _pad_offsets((h, w), (Ho, Wo))
        if p16 is not None:
            catP = p16["catP"]
            assert (Ho, Wo) == (2 * h, 2 * w) and catP.shape[1] * 8 == C2 + Ct and C2 % 8 == 0
            # round 5: x1 pre-split by its producer and the weights in slot form -- both GEMM operands are LDS-DMA copies
            x1P = p16.get("x1P")
            done = False
            if x1P is not None and hasattr(packed, "slots") and (pt, pl) == (0, 0):
                done = ops.convT2x2_fwd_slots(x1P, packed.slots(x1P.shape[3]), bias, catP[:, C2 // 8:], Ct, x_slots=p16.get("x1_slots"),
                                              slots=p16.get("up_slots"))
            if not done and ops.is_placeholder(x1):
                raise RuntimeError("onet_amd: ConvTranspose2d input kept only pre-split reached a shape the slot-operand GEMM does not take")
            # the GEMM's epilogue writes whole pre-split slots; shapes outside its fast path: fp32 + one conversion pass
            if not done and not ops.convT2x2_fwd_p(x1, ops.pack_of(wp_fused), bias, catP[:, C2 // 8:], Ct, pt, pl, slots=p16.get("up_slots")):
                # (e.g. an 8 x 8 input map: h w % 128 != 0.)  The conversion pass applies the guard scale of the same slots the
                # fused epilogue would have used, so the consumer's rescale stays right
                up = torch.empty((B, Ct, Ho, Wo), dtype=torch.float32, device=x1.device)
                ops.convT2x2_fwd(x1, ops.pack_of(wp_fused), bias, up, Ct, pt, pl)
                ops.split_pack_act(up, out=catP[:, C2 // 8:], slots=p16.get("up_slots"))
            ctx.save_for_backward(x1)
            ctx.wp_dgrad = wp_dgrad
            ctx.meta = (C2, Ct, h, w, pt, pl, tuple(weight.shape), bias is not None)
            ctx.params = (weight, bias)
            # round 5: the backward GEMMs on slot operands too -- the consumer of the concat buffer (the decoder block's first convolution)
            # is told, through the shared dict, to hand the up-sampled half of its input gradient over pre-split
            lk = p16.get("up_link")
            ctx.up_link = None
            if lk is not None and done:
                lk["want"] = (C2, Ct)
                ctx.up_link, ctx.slot_ops = lk, (x1P, p16.get("x1_slots"), packed)
            elif ops.is_placeholder(x1):
                raise RuntimeError("onet_amd: ConvTranspose2d input kept only pre-split, but the slot-operand backward is not set up")
            return ops.fp32_placeholder((B, C2 + Ct, Ho, Wo), x1.device)
        cat = None if cat_holder is None else cat_holder[0]
        in_place = (cat is not None and tuple(cat.shape) == (B, C2 + Ct, Ho, Wo) and C2 > 0
                    and x2.data_ptr() == cat.data_ptr() and x2.stride() == cat[:, :C2].stride())
        if not in_place:       # x2 is an ordinary tensor: torch.cat's copy of the skip half
            cat = torch.empty((B, C2 + Ct, Ho, Wo), dtype=torch.float32, device=x1.device)
            if C2 > 0:
                ops.copy_strided(x2, cat[:, :C2])
        if (Ho, Wo) != (2 * h, 2 * w):
            for bi in range(B):                      # F.pad border (only when H or W is not a multiple of 16); raw
                ops.fill(cat[bi, C2:], 0.0)          # fills: a torch in-place op on the base of the skip view is forbidden
        ops.convT2x2_fwd(x1, ops.pack_of(wp_fused), bias, cat[:, C2:], Ct, pt, pl)
        ctx.save_for_backward(x1)
        ctx.wp_dgrad = wp_dgrad
        ctx.meta = (C2, Ct, h, w, pt, pl, tuple(weight.shape), bias is not None)
        ctx.params = (weight, bias)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        (x1,) = ctx.saved_tensors
        C2, Ct, h, w, pt, pl, wshape, has_bias = ctx.meta
        need_x1, need_x2, need_w, need_b = ctx.needs_input_grad[:4]
        dx2 = dcat[:, :C2] if need_x2 else None
        dx1 = dw = db = None
        lk = getattr(ctx, "up_link", None)
        if lk is not None:
            dyP, dy_slots, rda = lk.pop("dyP", None), lk.pop("dy_slots", None), lk.pop("da", None)
            lk.pop("want", None)
            if dyP is not None and rda is not None and rda.data_ptr() == dcat.data_ptr() and rda.shape == dcat.shape:
                # the up-sampled half of dcat exists only pre-split (written by the 3x3 input gradient that produced dcat)
                x1P, x1_slots, packed = ctx.slot_ops
                want_db = need_b and has_bias
                if need_x1:
                    dx1 = ops.convT2x2_dgrad_slots(dyP, packed.dgrad_slots(dyP.shape[3]), wshape[0], dy_slots=dy_slots)
                got = ops.convT2x2_wgrad_slots(x1P, dyP, wshape, x_slots=x1_slots, dy_slots=dy_slots, want_dbias=want_db,
                                               out=ops.grad_slot_if_free(ctx.params[0]),
                                               db_out=ops.grad_slot_if_free(ctx.params[1]) if want_db else None) if (need_w or want_db) else (None, None)
                if (need_x1 and dx1 is None) or got is None:
                    raise RuntimeError("onet_amd: the slot-operand ConvTranspose2d backward refused a shape ops.convt_bwd_slots_ok accepted")
                return dx1, dx2, (got[0] if need_w else None), (got[1] if want_db else None), None, None, None
            if ops.is_placeholder(x1):
                raise RuntimeError("onet_amd: ConvTranspose2d backward: the input exists only pre-split and the pre-split gradient did not arrive")
        wp_dgrad = ops.pack_of(ctx.wp_dgrad) if need_x1 else None     # (the weight is what it was in forward: no optimizer step in between)
        if (need_x1 or need_w or need_b) and ops.convT2x2_bwd_fusable(Ct):
            # the GEMM kernels gather dy from the concat gradient themselves: no space-to-depth tensor
            dup = dcat[:, C2:]
            want_db = need_b and has_bias
            db_slot = ops.grad_slot_if_free(ctx.params[1]) if want_db else None
            if need_x1:
                # the dgrad GEMM stages every dy row anyway: the bias gradient is summed on the way where its fast path applies
                dx1, db = ops.convT2x2_dgrad(dup, wp_dgrad, wshape[0], h, w, pt, pl, want_dbias=want_db, db_out=db_slot) \
                    if want_db else (ops.convT2x2_dgrad(dup, wp_dgrad, wshape[0], h, w, pt, pl), None)
            if need_w or (want_db and db is None):
                dw, db2 = ops.convT2x2_wgrad(x1, dup, wshape, pt, pl, want_dbias=(want_db and db is None),
                                             out=ops.grad_slot_if_free(ctx.params[0]), db_out=db_slot)
                db = db if db is not None else db2
        elif need_x1 or need_w or need_b:
            dsub, db = ops.space_to_depth2(dcat[:, C2:], h, w, pt, pl, want_dbias=(need_b and has_bias))
            if need_w:
                dw = ops.conv_wgrad(x1, dsub, wshape, 1, out_layout=1)
            if need_x1:
                dx1 = ops.conv_fwd(dsub, wp_dgrad, wshape[0], 1)
        return dx1, dx2, dw, db, None, None, None


@_carries_settings
class UpBilinearCatFn(torch.autograd.Function):
    """nn.Upsample(x2, bilinear, align_corners=True) + F.pad + cat  (OV:83, OV:91-100)."""

    @staticmethod
    def forward(ctx, x1, x2):
        ops.require_gpu(x1, x2)
        B, C1, h, w = x1.shape
        C2, Ho, Wo = x2.shape[1], x2.shape[2], x2.shape[3]
        pt, pl = _pad_offsets((h, w), (Ho, Wo))
        cat = torch.empty((B, C2 + C1, Ho, Wo), dtype=torch.float32, device=x1.device)
        if C2 > 0:
            ops.copy_strided(x2, cat[:, :C2])
        ops.bilinear2x_fwd(x1, cat[:, C2:], pt, pl)
        ctx.meta = (C2, h, w, pt, pl)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        C2, h, w, pt, pl = ctx.meta
        dx2 = dcat[:, :C2] if ctx.needs_input_grad[1] else None
        dx1 = ops.bilinear2x_bwd(dcat[:, C2:], h, w, pt, pl) if ctx.needs_input_grad[0] else None
        return dx1, dx2


@_carries_settings
class HeadSoftmaxFn(torch.autograd.Function):
    """V = einsum('bpxy,bpxy->bxy', L, H) for both branches + Softmax2d(cat[Vt,Vd])  (OV:176-189)."""

    @staticmethod
    def forward(ctx, Lt, Ht, Ld, Hd):
        Vt, Vd, S = ops.head_softmax_fwd(Lt, Ht, Ld, Hd)
        ctx.save_for_backward(Lt, Ht, Ld, Hd, S)
        return Vt, Vd, S

    @staticmethod
    def backward(ctx, dVt, dVd, dS):
        Lt, Ht, Ld, Hd, S = ctx.saved_tensors
        dLt, dHt, dLd, dHd = ops.head_softmax_bwd(dVt, dVd, dS, S, Lt, Ht, Ld, Hd)
        return dLt, dHt, dLd, dHd


@_carries_settings
class TwinInputFn(torch.autograd.Function):
    """[X ; clip(1 - X + bias, 0, 1)] as one batch of 2B (OV:180 for the second half)."""

    @staticmethod
    def forward(ctx, x, bias):
        ops.require_gpu(x)
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.bias = bias
        if ops.TWIN_VIRTUAL:
            # K7 as SURVEY 2 states it: the complement half is formed by the stem kernels while they load X; the 2B-image tensor
            # exists only if some other reader asks for it (ops.plane materialises it once)
            return ops.twin_virtual(x, bias)
        return ops.twin_materialize(src=(x, bias))

    @staticmethod
    def backward(ctx, g):
        # only reached when the caller asks for d/dX (never on the training path): plumbing
        (x,) = ctx.saved_tensors
        B = x.shape[0]
        y = ops.complement_clip(x, ctx.bias)
        return g[:B] - g[B:] * ((y > 0) & (y < 1)).to(g.dtype), None


@_carries_settings
class TwinSplitFn(torch.autograd.Function):
    """full [2B, ...] -> (top half, down half) as views.  The module's own head and loss differentiate the FULL
    tensor directly; this node only carries gradient when a caller uses Lt / Ld in a graph of their own."""

    @staticmethod
    def forward(ctx, full):
        B = full.shape[0] // 2
        ctx.shape = tuple(full.shape)
        return full[:B], full[B:]

    @staticmethod
    def backward(ctx, g0, g1):
        B = ctx.shape[0] // 2
        ref = g0 if g0 is not None else g1
        out = torch.zeros(ctx.shape, dtype=ref.dtype, device=ref.device) if (g0 is None or g1 is None) else \
            torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        if g0 is not None:
            out[:B].copy_(g0)
        if g1 is not None:
            out[B:].copy_(g1)
        return out


@_carries_settings
class HeadSoftmaxTwinFn(torch.autograd.Function):
    """HeadSoftmaxFn on the twin batch: L = [Lt ; Ld], H = [Ht ; Hd] (each [2B, 64, H, W]).  Also returns the per-pixel
    channel sums of Lt and Ld -- all the loss needs of L (OV:231-232: the einsum with a 1-channel S is S * sum_c L) --
    so that compute_loss works on [B,1,H,W] tensors and its gradient re-enters here as two small maps."""

    @staticmethod
    def forward(ctx, L, H, head_link=None):
        """head_link (round 5): {"z", "save"} published by the network's last Conv-BatchNorm-ReLU unit, which then wrote NO activation
        (H is a placeholder): the head normalises and rectifies z on load -- the same bits -- in forward and backward."""
        B = L.shape[0] // 2
        ctx.h_save = None
        if head_link is not None and "z" in head_link:
            z, save = head_link.pop("z"), head_link.pop("save")
            assert z.shape == H.shape and z.dtype == torch.float32 and save.shape[0] == 2
            ctx.h_save = save
            H = z
        elif ops.is_placeholder(H):
            raise RuntimeError("onet_amd: the head received a placeholder for the last activation without its pre-activation")
        nrm = None if ctx.h_save is None else (ctx.h_save[0], ctx.h_save[1])
        Vt, Vd, S, sLt, sLd = ops.head_softmax_fwd(L[:B], H[:B], L[B:], H[B:], want_sums=True, h_norm=nrm)
        ctx.save_for_backward(L, H, S)
        return Vt, Vd, S, sLt, sLd

    @staticmethod
    def backward(ctx, dVt, dVd, dS, gsLt, gsLd):
        L, H, S = ctx.saved_tensors
        B = L.shape[0] // 2
        nrm = None if ctx.h_save is None else (ctx.h_save[0], ctx.h_save[1])
        dL, dH = ops.head_softmax_bwd(dVt, dVd, dS, S, L[:B], H[:B], L[B:], H[B:], twin=True, gsums=(gsLt, gsLd), h_norm=nrm)
        return dL, dH, None


@_carries_settings
class JSDSumsFn(torch.autograd.Function):
    """(jsd(Lt, St, Sd), jsd(Ld, Sd, St)) of Onet.compute_loss (OV:253-267) from the channel sums sLt, sLd."""

    @staticmethod
    def forward(ctx, sLt, sLd, St, Sd):
        top, _ = ops.jsd_fwd(None, St, Sd, sums=sLt)
        dwn, _ = ops.jsd_fwd(None, Sd, St, sums=sLd)
        ctx.save_for_backward(sLt, sLd, St, Sd)
        return top, dwn

    @staticmethod
    def backward(ctx, g_top, g_dwn):
        sLt, sLd, St, Sd = ctx.saved_tensors
        B, _, H, W = sLt.shape
        half = (B, 1, H, W)
        zero = torch.zeros((), dtype=torch.float32, device=St.device)
        gLt, dSt_a, dSd_a = ops.jsd_bwd(zero if g_top is None else g_top, sLt.contiguous().view(-1), St, Sd, half)
        gLd, dSd_b, dSt_b = ops.jsd_bwd(zero if g_dwn is None else g_dwn, sLd.contiguous().view(-1), Sd, St, half)
        return gLt, gLd, dSt_a + dSt_b, dSd_a + dSd_b


@_carries_settings
class JSDFn(torch.autograd.Function):
    """Onet.jensen_shannon_divergence(Li, Si, Sprime)  (OV:221-235) incl. the log1pexp quirk."""

    @staticmethod
    def forward(ctx, Li, Si, Sp):
        out, sums = ops.jsd_fwd(Li, Si, Sp)
        ctx.save_for_backward(sums, Si, Sp)
        ctx.shape = tuple(Li.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        sums, Si, Sp = ctx.saved_tensors
        gL, dSi, dSp = ops.jsd_bwd(g, sums, Si, Sp, ctx.shape)
        # d jsd / dL is identical for all 64 channels: hand autograd a stride-0 view, not a copy
        return gL.expand(ctx.shape), dSi, dSp


@_carries_settings
class Log1pExpFn(torch.autograd.Function):
    """Onet.log1pexp: mutates its argument in place and returns it  (OV:237-251)."""

    @staticmethod
    def forward(ctx, x):
        x0 = x.detach().clone()
        ops.log1pexp_(x)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x0)
        return x

    @staticmethod
    def backward(ctx, g):
        (x0,) = ctx.saved_tensors
        return ops.log1pexp_bwd(x0, g)


@_carries_settings
class ComplementClipFn(torch.autograd.Function):
    """Xd = clip(1 - X + bias, 0, 1)  (OV:180)."""

    @staticmethod
    def forward(ctx, x, bias):
        y = ops.complement_clip(x, bias)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        # only reached when the caller asks for d/dX (never on the training path): plumbing
        (y,) = ctx.saved_tensors
        return -g * ((y > 0) & (y < 1)).to(g.dtype), None
