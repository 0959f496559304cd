"""Host-side mirror of the reference's model interface (source_code/Onet_vanilla_20240606.py,
"OV"): same class names, constructor signatures, attribute tree, parameter names and
state_dict keys (232 entries for Onet), same method names / argument meaning / error
behaviour -- but every tensor operation runs a hand-written gfx950 HIP kernel through the
C ABI (include/onet_hip.h).

The leaf parameter holders subclass torch's own module types (``nn.Conv2d``,
``nn.BatchNorm2d``, ``nn.ConvTranspose2d`` ...) ONLY to inherit parameter registration,
default initialisation, ``state_dict`` layout and ``isinstance`` behaviour (the reference's
``_initialize_weights`` and its model-summary hooks dispatch on those types).  Their
``forward`` is overridden: ATen/MIOpen convolution, batch-norm, pooling, upsampling are never
called.  There is no CPU fallback: a CPU tensor raises."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops


class _Packable:
    """Caches GEMM-packed copies of ``self.weight``; re-packed when the parameter changes."""

    _pack = None
    _pack_key = None

    def _pack_fn(self, w):
        raise NotImplementedError

    def packed(self):
        w = self.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._pack_key != key:
            with torch.no_grad():
                self._pack = self._pack_fn(w)
            self._pack_key = key
        return self._pack

    def invalidate_packed(self):
        self._pack_key = None


def invalidate_packed(model: nn.Module):
    """Call after parameters were updated behind autograd's back (fused Adam, NCCL broadcast)."""
    for m in model.modules():
        if isinstance(m, _Packable):
            m.invalidate_packed()


class Conv3x3(nn.Conv2d, _Packable):
    """nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False) parameter holder (OV:47,51)."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, kernel_size=3, padding=1, bias=False)

    def _pack_fn(self, w):
        return ops.pack3x3_auto(w)

    def forward(self, x):
        return Fn.Conv3x3Fn.apply(x, self.weight, self.packed())


class BatchNormReLU2d(nn.BatchNorm2d):
    """nn.BatchNorm2d parameter/buffer holder (OV:48,52).  In this network it is always followed
    by ReLU (OV:49,53) and the HIP kernel fuses the two, so the ReLU entry of the Sequential is a
    pass-through marker (see DoubleConv)."""

    def forward(self, z):
        if z.dim() != 4:
            raise ValueError(f"expected 4D input (got {z.dim()}D input)")
        training = self.training or (self.running_mean is None)
        if training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return Fn.BNReLUFn.apply(z, self.weight, self.bias, self.running_mean, self.running_var, training,
                                 self.momentum, self.eps)


class FusedReLU(nn.ReLU):
    """nn.ReLU(inplace=True) slot of the reference's Sequential (OV:49,53): the activation is applied
    inside the BN kernel, so this module is the identity on already-rectified data."""

    def __init__(self):
        super().__init__(inplace=True)

    def forward(self, x):
        return x


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2   -- OV:39-58."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            Conv3x3(in_channels, mid_channels),
            BatchNormReLU2d(mid_channels),
            FusedReLU(),
            Conv3x3(mid_channels, out_channels),
            BatchNormReLU2d(out_channels),
            FusedReLU(),
        )

    def pre_capable(self):
        """Does forward() take its pre-split branch (the only one that reads or writes pre-split tensors)?  UNet._forward allocates a
        concat buffer in pre-split form ONLY where both the encoder block that writes its skip groups and the decoder block that
        reads it answer yes -- a block with a frozen (eval) BatchNorm or without running statistics keeps fp32 tensors."""
        s = self.double_conv
        return bool(ops.presplit() and s[1].training and s[4].training and s[1].running_mean is not None)

    @staticmethod
    def _unit(x, conv, bn, out=None, groups=1, link_out=None, link_in=None, p16=None):
        training = bn.training or (bn.running_mean is None)
        if x.dim() != 4:
            raise ValueError(f"expected 4D input (got {x.dim()}D input)")
        if x.shape[1] != conv.in_channels:
            raise RuntimeError(f"Given groups=1, weight of size {list(conv.weight.shape)}, expected input"
                               f"{list(x.shape)} to have {conv.in_channels} channels, but got {x.shape[1]} channels instead")
        if training and (x.shape[0] // groups) * x.shape[2] * x.shape[3] <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                             f"{[x.shape[0] // groups, conv.out_channels, x.shape[2], x.shape[3]]}")
        if training and bn.track_running_stats:
            ops.count_batches(bn.num_batches_tracked, groups)
        # magnitude slots (ops.tag_amax): what x's producer recorded of it goes in, what this unit records of its output comes out
        aux = {"x_amax": ops.amax_of(x)}
        a = Fn.ConvBNReLUFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  training, bn.momentum, bn.eps, conv.packed(), out, groups, link_out, link_in, aux, p16)
        if p16 is not None:
            ops.tag_p16(a, p16.get("a"), p16.get("a_slots"))      # pre-split storage: the output's pre-split form (+ its scale slots) rides on it
            if p16.get("a_amax") is not None:
                return ops.tag_amax(a, p16["a_amax"])
        return ops.tag_amax(a, aux.get("a_amax"))

    def forward(self, x, out=None, groups=1, pool_link=None, p16_out=None, up_link=None, head_link=None):
        """`out`: optional plane-contiguous [B, Cout, H, W] destination view (the skip half of a concat buffer);
        `groups`: the batch holds that many independent BatchNorm batches (twin pass); `pool_link`: dict the second
        unit publishes its (z, save) in for the SkipPoolFn that consumes the block's output; `p16_out` (pre-split storage): {"out": pre-split destination of the
        block's output (the skip groups of a pre-split concat buffer), "keep_fp32": the fp32 tensor is needed too}."""
        s = self.double_conv
        link = {}       # unit 1 -> unit 2: lets unit 2's dgrad launch take unit 1's BatchNorm-backward reduce pass with it
        if x.dim() == 4 and x.is_cuda and self.pre_capable():
            # pre-split storage: a tensor is written pre-split exactly where ops.pre_layer_ok says its consuming convolution runs
            # all three of its kernels on pre-split operands (the producer of x decided with the same function)
            B, _, H, W = x.shape
            xP = ops.p16_of(x)
            pre2 = ops.pre_layer_ok(B, s[3].in_channels, s[3].out_channels, H, W)
            if xP is not None or pre2 or p16_out is not None:
                if xP is None and ops.is_placeholder(x) and ops.twin_src_of(x) is None:
                    raise RuntimeError("onet_amd: a tensor kept only pre-split reached a DoubleConv without its pre-split form")
                p1 = {"x": xP, "x_slots": ops.p16_slots(x), "want": pre2, "up_link": up_link}
                a1 = self._unit(x, s[0], s[1], None, groups, link_out=link, p16=p1)
                p2 = {"x": p1.get("a"), "x_slots": p1.get("a_slots"), "want": p16_out is not None, "head_link": head_link,
                      "out": None if p16_out is None else p16_out.get("out"),
                      "keep_fp32": True if p16_out is None else bool(p16_out.get("keep_fp32"))}
                return self._unit(a1, s[3], s[4], None, groups, link_out=pool_link, link_in=link, p16=p2)
        # fp32 model, split-bf16 kernels: unit 2's convolution (forward and weight gradient) applies unit 1's BatchNorm + ReLU
        # on load -- unit 1 writes no activation (functional.ConvBNReLUFn; bit-identical to the materialised form)
        if (x.dim() == 4 and x.is_cuda and s[1].training and s[4].training and s[1].running_mean is not None
                and ops.norm_on_load_ok(x.shape[0], s[3].in_channels, s[3].out_channels, x.shape[2], x.shape[3], groups)):
            link["defer"] = True
        a1 = self._unit(x, s[0], s[1], None, groups, link_out=link)
        return self._unit(a1, s[3], s[4], None if out is None else (out,), groups, link_out=pool_link, link_in=link)


_SKIPPOOL = ops._flag("SKIPPOOL", True)


class MaxPool2(nn.MaxPool2d):
    """nn.MaxPool2d(2) (OV:67)."""

    def __init__(self):
        super().__init__(2)

    def forward(self, x):
        return Fn.MaxPool2Fn.apply(x)


class Down(nn.Module):
    """Downscaling with maxpool then double conv -- OV:61-72."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(MaxPool2(), DoubleConv(in_channels, out_channels))

    def forward(self, x, out=None, groups=1, pooled=None, pool_link=None, p16_out=None):
        """`pooled`: maxpool2(x) when the caller already has it (UNet.forward pools skip tensors with SkipPoolFn);
        `pool_link`, `p16_out`: see DoubleConv.forward."""
        p = self.maxpool_conv[0](x) if pooled is None else pooled
        return self.maxpool_conv[1](p, out=out, groups=groups, pool_link=pool_link, p16_out=p16_out)


class ConvT2x2(nn.ConvTranspose2d, _Packable):
    """nn.ConvTranspose2d(cin, cin//2, kernel_size=2, stride=2) parameter holder (OV:86)."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, kernel_size=2, stride=2)

    def _pack_fn(self, w):
        return ops.PackedT(w)

    def forward(self, x, output_size=None):
        # standalone use: up-sample only (no skip): concat with an empty skip
        B, _, h, w = x.shape
        empty = torch.empty((B, 0, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
        return Fn.UpConvTCatFn.apply(x, empty, self.weight, self.bias, self.packed())


class BilinearUp2x(nn.Upsample):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (OV:83)."""

    def __init__(self):
        super().__init__(scale_factor=2, mode="bilinear", align_corners=True)

    def forward(self, x):
        B, _, h, w = x.shape
        empty = torch.empty((B, 0, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
        return Fn.UpBilinearCatFn.apply(x, empty)


class Up(nn.Module):
    """Upscaling then double conv -- OV:75-101.  up -> F.pad to the skip size -> cat([x2, x1]) -> DoubleConv."""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = BilinearUp2x()
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = ConvT2x2(in_channels, in_channels // 2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2, cat=None, groups=1, catP=None, p16_out=None, head_link=None):
        """`cat`: optional concat buffer whose first channels already ARE x2 (UNet.forward lets the encoder write its
        skip outputs there, so torch.cat's copy of the skip tensor, OV:100, never happens); `catP` (pre-split storage): the pre-split concat
        buffer, skip groups already written; `p16_out`: see DoubleConv.forward (the block's output feeds the next Up's slot-operand GEMM)."""
        if isinstance(self.up, ConvT2x2) and catP is not None:
            # scales of the two producers of the concat buffer: the skip groups' (the encoder's BatchNorm bound), the up-sampled groups'
            # (a bound from this layer's weights and a bound of x1: the exact maximum its producer recorded, or -- x1 written pre-split
            # for the slot-operand GEMM -- the BatchNorm bound it was scaled by)
            x1P = ops.p16_of(x1)
            x1_slots = ops.p16_slots(x1) if x1P is not None else None
            s_skip, x1_amax = ops.p16_slots(x2), (x1_slots if x1P is not None else ops.amax_of(x1))
            s_up = ops.convT2x2_out_bound(self.up.weight, self.up.bias, x1_amax) if (x1_amax is not None and catP.shape[3] == 2) else None
            p16 = {"catP": catP, "up_slots": s_up, "x1P": x1P, "x1_slots": x1_slots}
            # the backward GEMMs on slot operands (round 5): this Up's ConvTranspose2d and the first convolution of its DoubleConv share a
            # dict through which the up-sampled half of the concat gradient travels pre-split
            B, Cin, h, w = x1.shape
            up_link = {} if (x1P is not None and self.training and self.conv.pre_capable() and
                             ops.convt_bwd_slots_ok(B, Cin, self.up.out_channels, x2.shape[1], h, w)) else None
            p16["up_link"] = up_link
            x = Fn.UpConvTCatFn.apply(x1, x2, self.up.weight, self.up.bias, self.up.packed(), None, p16)
            ops.tag_p16(x, catP, (s_skip, s_up, x2.shape[1]) if (s_skip is not None or s_up is not None) else None)
            return self.conv(x, groups=groups, p16_out=p16_out, up_link=up_link, head_link=head_link)
        if isinstance(self.up, ConvT2x2):
            x = Fn.UpConvTCatFn.apply(x1, x2, self.up.weight, self.up.bias, self.up.packed(), None if cat is None else (cat,))
        else:
            x = Fn.UpBilinearCatFn.apply(x1, x2)
        return self.conv(x, groups=groups, p16_out=p16_out)


class UNet(nn.Module):
    """OV:104-153: 5-level encoder (64..1024), 4-level decoder, no 1x1 head; returns (x1, y1)."""

    def __init__(self, n_channels=1, n_classes=1, binit=False, bilinear=False):
        super().__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear

        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        factor = 2 if bilinear else 1
        self.down4 = Down(512, 1024 // factor)
        self.up1 = Up(1024, 512 // factor, bilinear)
        self.up2 = Up(512, 256 // factor, bilinear)
        self.up3 = Up(256, 128 // factor, bilinear)
        self.up4 = Up(128, 64, bilinear)
        if binit:
            self._initialize_weights()

    def _initialize_weights(self, mode="fan_in"):
        """OV:125-140: Kaiming-normal on every nn.Conv2d (ConvTranspose2d is NOT an nn.Conv2d, so it
        keeps torch's default init and non-zero bias -- SURVEY.md §8a-4), BN weight 1 / bias 0."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode=mode, nonlinearity="relu")
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def forward(self, x, groups=1, head_link=None):
        """head_link (Onet.forward, twin batch): a dict through which the LAST unit hands its pre-activation and coefficients to the head
        instead of writing its activation (the second returned tensor is then a placeholder: Onet's head is its only reader)."""
        # the 18 num_batches_tracked counters of a pass take their increments in one multi-tensor launch when the pass is through
        with ops.counting_batches():
            return self._forward(x, groups, head_link)

    def _forward(self, x, groups=1, head_link=None):
        # ConvTranspose path: the four skip tensors are produced directly inside the first half of their concat
        # buffers (allocated here, before the encoder runs), the decoder fills the second half
        cats = [None] * 4
        if not self.bilinear and x.dim() == 4 and x.is_cuda:
            B, h, w = x.shape[0], x.shape[2], x.shape[3]
            for k, enc in enumerate((self.inc, self.down1.maxpool_conv[1], self.down2.maxpool_conv[1],
                                     self.down3.maxpool_conv[1])):
                C = enc.double_conv[3].out_channels
                if h > 0 and w > 0:
                    cats[k] = torch.empty((B, 2 * C, h, w), dtype=torch.float32, device=x.device)
                h, w = h // 2, w // 2

        # pre-split storage: where the decoder's first convolution runs on pre-split operands (ops.pre_layer_ok), the concat buffer
        # exists ONLY pre-split: the encoder's BatchNorm pass writes the skip groups, the Up block the up-sampled ones
        catsP = [None] * 4
        pool_p = [False] * 4                                # the pooled tensor of encoder level k feeds a pre-split convolution
        if not self.bilinear and x.dim() == 4 and x.is_cuda and ops.presplit() and self.training:
            B, h, w = x.shape[0], x.shape[2], x.shape[3]
            for k, enc in enumerate((self.inc, self.down1.maxpool_conv[1], self.down2.maxpool_conv[1], self.down3.maxpool_conv[1])):
                C = enc.double_conv[3].out_channels
                dec = (self.up4, self.up3, self.up2, self.up1)[k].conv.double_conv[0]
                nxt = (self.down1, self.down2, self.down3, self.down4)[k].maxpool_conv[1].double_conv[0]
                dec_blk = (self.up4, self.up3, self.up2, self.up1)[k].conv
                nxt_blk = (self.down1, self.down2, self.down3, self.down4)[k].maxpool_conv[1]
                # (producer AND consumer must take their pre-split branches: DoubleConv.pre_capable -- a partially frozen network keeps
                # fp32 tensors at the levels concerned, as it did before pre-split storage)
                if h % 16 == 0 and w % 16 == 0 and dec.in_channels == 2 * C and ops.pre_layer_ok(B, dec.in_channels, dec.out_channels, h, w) \
                        and enc.pre_capable() and dec_blk.pre_capable():
                    catsP[k] = ops.p16_empty(B, 2 * C, h, w, x.device)
                    cats[k] = None
                pool_p[k] = h % 2 == 0 and w % 2 == 0 and ops.pre_layer_ok(B, nxt.in_channels, nxt.out_channels, h // 2, w // 2) \
                    and nxt_blk.pre_capable()
                h, w = h // 2, w // 2
        # round 5: the input of a ConvTranspose2d whose output goes into a pre-split concat buffer is written pre-split by the block below
        # (down4, up1 .. up3) where the slot-operand GEMM takes the shape; its fp32 tensor stays for the weight gradient
        upP = [None] * 4                                    # [k]: p16_out of the block that feeds level k's Up
        if any(c is not None for c in catsP):
            B, h, w = x.shape[0], x.shape[2], x.shape[3]
            for k in range(4):
                up = (self.up4, self.up3, self.up2, self.up1)[k]
                src = (self.up3.conv, self.up2.conv, self.up1.conv, self.down4.maxpool_conv[1])[k]
                hk, wk = h >> (k + 1), w >> (k + 1)         # the map that level k's ConvTranspose2d reads
                if catsP[k] is not None and isinstance(up.up, ConvT2x2) and src.pre_capable() and \
                        ops.convt_slots_ok(B, up.up.in_channels, up.up.out_channels, hk, wk):
                    # (the fp32 tensor stays only where the weight gradient still reads it: Up.forward decides with the same function)
                    bwd = up.training and up.conv.pre_capable() and \
                        ops.convt_bwd_slots_ok(B, up.up.in_channels, up.up.out_channels, up.up.out_channels, hk, wk)
                    upP[k] = {"keep_fp32": not bwd}

        def skip(k, C):
            return None if cats[k] is None else cats[k][:, :C]

        def skipP(k, C, keep_fp32=False):
            return None if catsP[k] is None else {"out": catsP[k][:, :C // 8], "keep_fp32": keep_fp32}

        g = groups

        def fork(t, returned=False, link=None, nxt=None):
            # skip tensors feed the next Down's pooling AND an Up's concat (x1 also leaves as the first output): one
            # node, so that their gradients are summed inside the pooling-backward kernel
            if t.is_cuda and _SKIPPOOL:
                am = ops.amax_of(t)
                carry = {} if ops.presplit() else None        # (carries the pooled tensor's pre-split form back: "yP")
                outs = Fn.SkipPoolFn.apply(t, returned, link, carry)
                if carry is not None:
                    ops.tag_p16(outs[1], carry.get("yP"), ops.p16_slots(t))  # the pooled tensor's pre-split form (max-pooling keeps the bound)
                    ops.tag_p16(outs[0], ops.p16_of(t), ops.p16_slots(t))     # the skip view keeps the tensor's pre-split form
                ops.tag_amax(outs[1], am)                    # max-pooling keeps the maximum: the same slots bound the pooled tensor
                return outs
            return (t, None, t) if returned else (t, None)

        # pl[i]: the producing block's second unit -> the SkipPoolFn of its output (BatchNorm-backward reduce records)
        pl = [{} for _ in range(4)]
        c0 = self.inc.double_conv[3].out_channels
        c1 = self.down1.maxpool_conv[1].double_conv[3].out_channels
        c2 = self.down2.maxpool_conv[1].double_conv[3].out_channels
        c3 = self.down3.maxpool_conv[1].double_conv[3].out_channels
        if x.dim() == 4 and x.is_cuda and _SKIPPOOL and ops.FUSE_POOL:
            # the four encoder outputs are max-pooled next: their BatchNorm + ReLU pass writes the pooled tensor too
            for i in range(4):
                pl[i]["want_pool"] = {"p16": pool_p[i]}
        # (pre-split storage: x1 leaves the U-Net and feeds the head, so it keeps its fp32 tensor beside the pre-split skip groups)
        x1 = self.inc(x, out=skip(0, c0), groups=g, pool_link=pl[0], p16_out=skipP(0, c0, True))
        x1, p1, x1_out = fork(x1, True, pl[0], self.down1)
        x2 = self.down1(x1, out=skip(1, c1), groups=g, pooled=p1, pool_link=pl[1], p16_out=skipP(1, c1))
        x2, p2 = fork(x2, False, pl[1], self.down2)
        x3 = self.down2(x2, out=skip(2, c2), groups=g, pooled=p2, pool_link=pl[2], p16_out=skipP(2, c2))
        x3, p3 = fork(x3, False, pl[2], self.down3)
        x4 = self.down3(x3, out=skip(3, c3), groups=g, pooled=p3, pool_link=pl[3], p16_out=skipP(3, c3))
        x4, p4 = fork(x4, False, pl[3], self.down4)
        x5 = self.down4(x4, groups=g, pooled=p4, p16_out=upP[3])
        y4 = self.up1(x5, x4, cat=cats[3], groups=g, catP=catsP[3], p16_out=upP[2])
        y3 = self.up2(y4, x3, cat=cats[2], groups=g, catP=catsP[2], p16_out=upP[1])
        y2 = self.up3(y3, x2, cat=cats[1], groups=g, catP=catsP[1], p16_out=upP[0])
        y1 = self.up4(y2, x1, cat=cats[0], groups=g, catP=catsP[0], head_link=head_link)
        return x1_out, y1


class Softmax2(nn.Softmax2d):
    """nn.Softmax2d slot (OV:171): 2-channel softmax; Onet.forward fuses it into the head kernel,
    get_label() calls it directly on cat([Vt, Vd])."""

    def forward(self, V):
        if V.dim() != 4 or V.shape[1] != 2:
            raise ValueError("Softmax2: expected a [B,2,H,W] logit tensor")
        one = torch.ones((V.shape[0], 1, V.shape[2], V.shape[3]), dtype=V.dtype, device=V.device)
        # softmax([Vt,Vd]) == head(L=V-channel, H=1): reuse the fused head kernel
        _, _, S = Fn.HeadSoftmaxFn.apply(V[:, 0:1], one, V[:, 1:2], one)
        return S


class Onet(nn.Module):
    """OV:156-267.  forward(X) -> (Lt, Vt, Ld, Vd, S); compute_loss(Lt, St, Ld, Sd) -> scalar."""

    def __init__(self, in_chns=1, binit=False, bshare=True):
        super().__init__()
        self.topu = UNet(n_channels=in_chns, n_classes=1, bilinear=False, binit=binit)
        if bshare:
            self.dwnu = self.topu           # weight sharing: the SAME module object (OV:163-164)
        else:
            self.dwnu = UNet(n_channels=in_chns, n_classes=1, bilinear=False, binit=binit)
        self.softmax = Softmax2()
        self.bias = 0                       # background bias in [0,1], read every forward (OV:172,180)
        self.check_finite = True            # OV:234 asserts the loss is not NaN (forces a device sync)
        # this model's switches (conv algorithm / precision, twin batching, deferred NaN check); None fields = process defaults.
        # Active for everything forward / compute_loss launch and -- captured by the autograd Functions -- for their backward.
        self.settings = ops.Settings()

    def forward(self, X):
        with ops.using(self.settings):
            return self._forward(X)

    def _forward(self, X):
        if X.dim() == 4 and X.is_cuda and ops.split_enabled():
            ops.amax_arena_reset(X.device)          # this step's magnitude slots: one fill instead of one per tensor
        if self.dwnu is self.topu and ops.twin_enabled() and X.dim() == 4 and X.is_cuda:
            # shared weights: X and 1-X go through every convolution as ONE batch of 2B (twice the blocks per launch,
            # weights packed / weight gradients reduced once); BatchNorm treats the halves as two batches, in the
            # reference's order (X first).  Results: identical activations, weight gradients summed in one
            # split-K reduction instead of two plus autograd's add.
            XX = Fn.TwinInputFn.apply(X, float(self.bias))
            hl = {} if (ops.HEAD_NORM and self.training) else None     # (the last activation is formed by the head from z: no tensor)
            L, H = self.topu(XX, groups=2, head_link=hl)
            Vt, Vd, S, sLt, sLd = Fn.HeadSoftmaxTwinFn.apply(L, H, hl)
            Lt, Ld = Fn.TwinSplitFn.apply(L)
            # compute_loss may work on the channel sums of L -- if it is handed exactly these halves, in this order and
            # unmodified (role and tensor version are checked there)
            tw = (L, sLt, sLd)
            Lt._onet_twin, Ld._onet_twin = (tw, 0, Lt._version), (tw, 1, Ld._version)
            return Lt, Vt, Ld, Vd, S
        Lt, Ht = self.topu(X)
        Xd = Fn.ComplementClipFn.apply(X, float(self.bias))
        Ld, Hd = self.dwnu(Xd)
        Vt, Vd, S = Fn.HeadSoftmaxFn.apply(Lt, Ht, Ld, Hd)
        return Lt, Vt, Ld, Vd, S

    def predict_label(self, S):
        with torch.no_grad():
            assert (S.dim() == 4)
            return ops.argmax2(S)

    def get_label(self, Vt, Vd):
        with torch.no_grad():
            one = torch.ones_like(Vt)
            _, _, V = Fn.HeadSoftmaxFn.apply(Vt, one, Vd, one)
            return ops.argmax2(V), V

    def jensen_shannon_divergence(self, Li, Si, Sprime):
        assert (Li.dim() == 4 and Si.dim() == 4 and Sprime.dim() == 4)
        jsd = Fn.JSDFn.apply(Li, Si, Sprime)
        if self.check_finite:
            assert (torch.isnan(jsd) == False)  # noqa: E712  (mirrors OV:234)
        return jsd

    def log1pexp(self, x):
        return Fn.Log1pExpFn.apply(x)

    def compute_loss(self, Lt, St, Ld, Sd):
        with ops.using(self.settings):
            return self._compute_loss(Lt, St, Ld, Sd)

    def _compute_loss(self, Lt, St, Ld, Sd):
        tag_t, tag_d = getattr(Lt, "_onet_twin", None), getattr(Ld, "_onet_twin", None)
        twin = None
        if (tag_t is not None and tag_d is not None and tag_t[0] is tag_d[0] and tag_t[1] == 0 and tag_d[1] == 1
                and tag_t[2] == Lt._version and tag_d[2] == Ld._version):
            twin = tag_t[0]       # Lt is the top half, Ld the down half of ONE forward, neither modified in place since
        if (twin is not None and twin[0].shape[0] == 2 * Lt.shape[0]
                and type(self).jensen_shannon_divergence is Onet.jensen_shannon_divergence
                and "jensen_shannon_divergence" not in self.__dict__):
            # Lt / Ld are the halves of this module's own twin batch: the JSD terms only need sum_c L, which the head
            # kernel already produced; their gradient flows back INTO the head's backward kernel as two [B,1,H,W]
            # maps (no second read of L, no full-tensor gradient add)
            assert (Lt.dim() == 4 and St.dim() == 4 and Sd.dim() == 4)
            jsd_top, jsd_dwn = Fn.JSDSumsFn.apply(twin[1], twin[2], St, Sd)
            if self.check_finite:
                # OV:234 asserts each term right after computing it (two device syncs); both terms exist here
                # already, so one check answers for the pair -- in place, or (settings.lazy_nan, set by training
                # loops that own the optimizer step) deferred to FlatAdam.step() / the next compute_loss
                ops.check_deferred_nan()
                flag = torch.isnan(jsd_top + jsd_dwn)
                if ops.lazy_nan_check():
                    ops.defer_nan_check(flag)
                else:
                    assert not bool(flag), "jsd is NaN"
            return -(jsd_top + jsd_dwn) / 2
        jsd = getattr(self, "jensen_shannon_divergence", None)
        if callable(jsd):
            jsd_top = jsd(Lt, St, Sd)
            jsd_dwn = jsd(Ld, Sd, St)
            return -(jsd_top + jsd_dwn) / 2
