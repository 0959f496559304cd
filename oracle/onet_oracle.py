"""CPU ORACLE for the Onet twin-U-Net hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.  The product path (``onet_amd``) never does:
it fails loudly when the HIP extension is missing.

What this is
------------
A table-driven, functional restatement (PyTorch *CPU* fp32 ops: ``F.conv2d``,
``F.batch_norm`` ... i.e. the same ATen/oneDNN arithmetic the reference runs on
CPU) of ``/root/reference/source_code/Onet_vanilla_20240606.py`` ("OV"):

* ``unet_pass``        <- OV:39-58 (DoubleConv), OV:61-72 (Down), OV:75-101 (Up),
                          OV:104-153 (UNet wiring; no ``outc`` head, returns (x1, y1))
* ``onet_forward``     <- OV:174-191 (twin pass on X and clip(1-X+bias,0,1), head
                          einsum, 2-channel softmax)
* ``log1pexp_quirk``   <- OV:237-251 (the in-place, order-dependent piecewise
                          function: x <= -37 maps to ln 2)
* ``jsd`` / ``compute_loss`` <- OV:221-235, OV:253-267
* ``predict_label``    <- OV:193-202

Pinning
-------
The reference ships no tests or golden vectors for this path (SURVEY.md §4,
§8c: "parity unpinned" by the reference itself).  This oracle is therefore
pinned against OUTPUTS OF THE REFERENCE ITSELF, produced in the build
container by ``tests/golden/make_golden.py`` (which imports OV with inert
stubs for its unused imports) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them.

Weights are never stored: both sides regenerate them with ``det_state_dict``
(NumPy PCG64 keyed by (seed, crc32(name)) -- independent of torch's RNG).

Routing-controlled evaluation (``Routing``)
-------------------------------------------
The network is piecewise linear in its activations: every ReLU (OV:49,53) and every
MaxPool2d window (OV:67) takes a discrete decision, and the GRADIENT is discontinuous in
those decisions.  Two correct fp32 evaluations whose pre-activations differ in the last
bits decide a handful of elements differently, and a single decision at the last layer
moves every upstream gradient by ~1/sqrt(#elements) -- measured here on the reference's
own ATen kernels: fp32 vs fp64 gradients differ by 3e-3 .. 5e-3 (relative L2, EVERY
parameter, every input size from 16^2 to 256^2, saturated head or not), and by 4e-6 once
the fp64 evaluation is made to take the fp32 run's decisions (tests/test_oracle_routing.py).
``Routing`` records those decisions (ReLU masks + pooling indices) from one evaluation and
replays them in another, which turns gradient parity into a continuous -- hence tightly
testable -- statement, plus an audit of the decisions themselves.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default (OV:48,52)
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default

ENC = [("inc", None, 64), ("down1", 64, 128), ("down2", 128, 256),
       ("down3", 256, 512), ("down4", 512, 1024)]
DEC = [("up1", 1024, 512), ("up2", 512, 256), ("up3", 256, 128), ("up4", 128, 64)]


def _dc_prefix(block: str) -> str:
    """state_dict prefix of the DoubleConv inside a UNet block (OV:111-120)."""
    if block == "inc":
        return "inc.double_conv"
    if block.startswith("down"):
        return f"{block}.maxpool_conv.1.double_conv"
    return f"{block}.conv.double_conv"


def unet_channels(bilinear: bool = False):
    """(encoder rows, decoder rows) of UNet.__init__ (OV:111-120): encoder (block, cin, cout); decoder (block, cin, cout,
    mid).  bilinear=True (OV:115,83-84): factor 2 -- down4 ends at 512 channels, every Up halves its output once more and its
    DoubleConv gets mid_channels = in_channels // 2; there is no ConvTranspose2d."""
    if not bilinear:
        return ENC, [(n, ci, co, co) for n, ci, co in DEC]
    enc = ENC[:4] + [("down4", 512, 512)]
    dec = [("up1", 1024, 256, 512), ("up2", 512, 128, 256), ("up3", 256, 64, 128), ("up4", 128, 64, 64)]
    return enc, dec


def param_table(in_chns: int = 1, bilinear: bool = False):
    """Ordered (name, shape, kind) for one UNet, in nn.Module registration order.

    kinds: conv | bn_w | bn_b | bn_rm | bn_rv | bn_nbt | convT_w | convT_b.
    116 entries (SURVEY.md §5 checkpoint row); bilinear=True (OV:83-84): no `up.*` rows, 108 entries."""
    rows = []

    def dc(block, cin, cout, mid=None):
        p = _dc_prefix(block)
        mid = cout if mid is None else mid
        for idx, (ci, co) in ((0, (cin, mid)), (3, (mid, cout))):
            rows.append((f"{p}.{idx}.weight", (co, ci, 3, 3), "conv"))
            b = idx + 1
            rows.append((f"{p}.{b}.weight", (co,), "bn_w"))
            rows.append((f"{p}.{b}.bias", (co,), "bn_b"))
            rows.append((f"{p}.{b}.running_mean", (co,), "bn_rm"))
            rows.append((f"{p}.{b}.running_var", (co,), "bn_rv"))
            rows.append((f"{p}.{b}.num_batches_tracked", (), "bn_nbt"))

    enc, dec = unet_channels(bilinear)
    for name, cin, cout in enc:
        dc(name, in_chns if cin is None else cin, cout)
    for name, cin, cout, mid in dec:
        if not bilinear:
            rows.append((f"{name}.up.weight", (cin, cin // 2, 2, 2), "convT_w"))
            rows.append((f"{name}.up.bias", (cin // 2,), "convT_b"))
        dc(name, cin, cout, mid)
    return rows


HEAD_BN = ("inc.double_conv.4", "up4.conv.double_conv.4")     # the BatchNorms whose outputs are L and H of the head


def det_state_dict(in_chns: int = 1, seed: int = 1981, randomize_running: bool = True, head_gain: float = 1.0,
                   bilinear: bool = False):
    """Deterministic UNet state (un-prefixed keys), independent of torch RNG.

    head_gain < 1 scales gamma and beta of the two BatchNorms that feed the head (OV:176-177), i.e. L and H and with
    them |V| ~ head_gain^2: at head_gain = 1 the initial logits reach |V| ~ 47 and the 2-way softmax is saturated on
    a fifth of the pixels; 0.3 gives |V| <= 5 (no pixel beyond S = 0.95)."""
    sd = OrderedDict()
    for name, shape, kind in param_table(in_chns, bilinear):
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if kind == "conv":
            fan_in = shape[1] * 9
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif kind == "convT_w":
            v = rng.standard_normal(shape) * np.sqrt(1.0 / (shape[0] * 4)) * 2.0
        elif kind == "convT_b":
            v = 0.1 * rng.standard_normal(shape)
        elif kind == "bn_w":
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "bn_b":
            v = 0.1 * rng.standard_normal(shape)
        elif kind == "bn_rm":
            v = 0.1 * rng.standard_normal(shape) if randomize_running else np.zeros(shape)
        elif kind == "bn_rv":
            v = 1.0 + 0.1 * np.abs(rng.standard_normal(shape)) if randomize_running else np.ones(shape)
        elif kind == "bn_nbt":
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        if head_gain != 1.0 and kind in ("bn_w", "bn_b") and name.rsplit(".", 1)[0] in HEAD_BN:
            v = v * head_gain
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def onet_state_dict(in_chns=1, seed=1981, bshare=True, randomize_running=True, head_gain=1.0):
    """The 232-key Onet state_dict: topu.* + dwnu.* (aliases when shared, OV:163-164)."""
    top = det_state_dict(in_chns, seed, randomize_running, head_gain)
    dwn = top if bshare else det_state_dict(in_chns, seed + 1, randomize_running, head_gain)
    sd = OrderedDict()
    for k, v in top.items():
        sd["topu." + k] = v
    for k, v in dwn.items():
        sd["dwnu." + k] = v
    return sd


def det_input(B, C, H, W, seed=7):
    rng = np.random.Generator(np.random.PCG64([seed, B, C, H, W]))
    return torch.from_numpy(rng.random((B, C, H, W), dtype=np.float32))


# --------------------------------------------------------------------------
# functional model
# --------------------------------------------------------------------------
class Routing:
    """The discrete decisions of one twin forward, in call order: 36 ReLU masks (18 Conv-BN-ReLU units per
    U-Net pass, X pass first) and 8 max-pool index maps (4 per pass).

    mode "record": the evaluation decides freely and its decisions are stored.
    mode "replay": the evaluation TAKES the stored decisions (relu(y) -> y * mask, max-pool -> gather at the stored
    indices) and `audit` lists, per decision site, how many elements the free decision would have taken differently
    and how far from the switching point the worst of them lies (|y| at a flipped ReLU / the gap between the window
    maximum and the routed element at a re-routed pool window), in units of the tensor's largest magnitude."""

    def __init__(self):
        self.mode = "record"
        self.masks, self.pools = [], []
        self.audit = []
        self._i = self._j = 0

    def replay(self):
        self.mode, self._i, self._j, self.audit = "replay", 0, 0, []
        return self

    @classmethod
    def from_activations(cls, acts, pooled_units=(1, 3, 5, 7), units_per_pass=18):
        """Decisions of ANOTHER implementation from its post-ReLU activations `acts` (list of [B,C,H,W] tensors in
        call order, 18 per pass): mask = a > 0; pooling indices = first maximum of each 2x2 window of the outputs
        of inc / down1 / down2 / down3 (units 1, 3, 5, 7 of a pass), as nn.MaxPool2d takes it."""
        r = cls()
        for n, a in enumerate(acts):
            r.masks.append(a > 0)
            if n % units_per_pass in pooled_units:
                r.pools.append(F.max_pool2d(a.float(), 2, return_indices=True)[1])
        return r.replay()

    def relu(self, y):
        if self.mode == "record":
            self.masks.append((y > 0).detach())
            return F.relu(y)
        m = self.masks[self._i]
        self._i += 1
        yd = y.detach()
        flip = (yd > 0) != m
        n = int(flip.sum())
        self.audit.append(("relu", n, m.numel(), float(yd[flip].abs().max() / yd.abs().max()) if n else 0.0))
        # torch.where, not `y * mask`: same function and same gradient (g where the mask is set, 0 elsewhere), but its backward
        # hands on a CONTIGUOUS gradient.  A multiply keeps the layout of the gradient it receives, and at batch size 1 the
        # backward of the reference's head einsum (OV:176) delivers shape [1,64,H,W] with strides (1,1,64W,64); ATen's CPU
        # batch_norm backward misreads such a tensor (dgamma, dbeta and dx off by O(1): reproduced standalone on torch 2.10) --
        # which is why the module-swap replay of the REAL reference gave wrong gradients at B = 1 in round 2.
        return torch.where(m, y, torch.zeros((), dtype=y.dtype))

    def maxpool2(self, a):
        if self.mode == "record":
            out, idx = F.max_pool2d(a, 2, return_indices=True)
            self.pools.append(idx)
            return out
        idx = self.pools[self._j]
        self._j += 1
        B, C = a.shape[0], a.shape[1]
        out = a.reshape(B, C, -1).gather(2, idx.reshape(B, C, -1)).view(idx.shape)
        free, fidx = F.max_pool2d(a.detach(), 2, return_indices=True)
        moved = fidx != idx
        n = int(moved.sum())
        gap = float((free - out.detach())[moved].max() / a.detach().abs().max()) if n else 0.0
        self.audit.append(("pool", n, idx.numel(), gap))
        return out


# ----------------------------------------------------------------------------- bf16-operand evaluation (BASELINE configs[2])
# The bf16 conv path of the HIP implementation rounds the OPERANDS of its matrix products to bf16 (nearest-even, ==
# tensor.bfloat16()) and accumulates in fp32: forward conv(bf16(x), bf16(w)); input gradient from bf16(dz) and bf16(w); weight
# gradient from bf16(x) and bf16(dz); biases and everything else unrounded.  That is NOT the derivative of the rounded forward
# (rounding has zero derivative almost everywhere), so it is stated as two autograd Functions.  `operand_rounding(rule)` switches
# them in for the convolutions rule(kind, x.shape, w.shape) selects (kind "conv3x3" | "convT2x2"; True = all three products, or
# the subset of {"fwd", "dgrad", "wgrad"} that takes rounded operands -- the kernels' shape conditions differ): the checker of the bf16
# gradient-parity test (tests/test_gpu_gradients.py), in fp64 so that only the operand rounding -- not summation -- is modelled.
#
# `replay`: rounding is a DECISION like a ReLU sign, and a chaotic one -- two evaluations whose operands differ in the last
# fp32 bits round ~3e-4 of them to different bf16 neighbours (4e-3 relative each), which after one layer moves 1e-2 of the next
# layer's roundings, and so on up to full bf16 noise: measured 1.2e-2 of the head logits' scale between the HIP run and this
# evaluation with free rounding, under identical ReLU / pooling decisions.  So, as `Routing` does for those, the evaluation can
# TAKE the other implementation's rounded operands: replay = {"conv3x3": [(x_r, g_r)] per call, "convT2x2": [(x_r, g_r)] per
# call} (already rounded, in this module's call order; weights are identical numbers in both and round identically).  What
# remains is a smooth function of the same operands, and the two must agree to summation accuracy.
#
# Round 5 -- STORED bf16 conv outputs (Settings.z_bf16 of the HIP path under conv == "bf16"): the convolution rounds its output z once,
# to bf16, when it stores it; the BatchNorm STATISTICS are taken from the unrounded accumulators, everything that READS z (normalise,
# ReLU decision, both backward passes) sees the rounded value.  kind "bn_z" of the rule says which units do that; the replay list
# "bn_z" holds the HIP run's stored z per unit (None: that unit kept fp32).  `_BNStoredZ` states the arithmetic -- forward from
# (mean, var)(z) and z_r, backward by the usual BatchNorm formula evaluated at xhat_r = (z_r - mean) invstd, which is what the
# kernels compute; it is NOT the derivative of the rounded forward (same remark as for the operand rounding above).
_ROUNDING_RULE = None
_ROUNDING_REPLAY = None
_ROUNDING_CALLS = {"conv3x3": 0, "convT2x2": 0, "bn_z": 0}


def _rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


class operand_rounding:
    def __init__(self, rule, replay=None):
        self.rule, self.replay = rule, replay

    def __enter__(self):
        global _ROUNDING_RULE, _ROUNDING_REPLAY
        self.prev = (_ROUNDING_RULE, _ROUNDING_REPLAY)
        _ROUNDING_RULE, _ROUNDING_REPLAY = self.rule, self.replay
        _ROUNDING_CALLS.update(conv3x3=0, convT2x2=0, bn_z=0)
        return self

    def __exit__(self, *exc):
        global _ROUNDING_RULE, _ROUNDING_REPLAY
        _ROUNDING_RULE, _ROUNDING_REPLAY = self.prev
        return False


def _replayed(kind):
    """-> (x_r, g_r) of this call from the replay lists (None, None without replay); counts every call of `kind`."""
    i = _ROUNDING_CALLS[kind]
    _ROUNDING_CALLS[kind] = i + 1
    return (None, None) if _ROUNDING_REPLAY is None else _ROUNDING_REPLAY[kind][i]


def _products(sel):
    """rule result -> the set of matrix products evaluated on rounded operands: True = all three"""
    return {"fwd", "dgrad", "wgrad"} if sel is True else set(sel)


class _RoundedConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, prods, x_r=None, g_r=None):
        xr, wr = (_rb(x) if x_r is None else x_r.to(x.dtype)), _rb(w)
        assert xr.shape == x.shape
        ctx.save_for_backward(x, w, xr, wr)
        ctx.g_r, ctx.prods = g_r, prods
        return F.conv2d(xr, wr, None, 1, 1) if "fwd" in prods else F.conv2d(x, w, None, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, w, xr, wr = ctx.saved_tensors
        gr = _rb(g) if ctx.g_r is None else ctx.g_r.to(g.dtype)
        assert gr.shape == g.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv2d_input(x.shape, wr, gr, padding=1) if "dgrad" in ctx.prods else \
                torch.nn.grad.conv2d_input(x.shape, w, g, padding=1)
        dw = torch.nn.grad.conv2d_weight(xr, w.shape, gr, padding=1) if "wgrad" in ctx.prods else \
            torch.nn.grad.conv2d_weight(x, w.shape, g, padding=1)
        return dx, dw, None, None, None


class _RoundedConvT2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, prods, x_r=None, g_r=None):
        xr, wr = (_rb(x) if x_r is None else x_r.to(x.dtype)), _rb(w)
        assert xr.shape == x.shape
        ctx.save_for_backward(x, w, xr, wr)
        ctx.g_r, ctx.prods = g_r, prods
        return F.conv_transpose2d(xr, wr, b, stride=2) if "fwd" in prods else F.conv_transpose2d(x, w, b, stride=2)

    @staticmethod
    def backward(ctx, g):
        x, w, xr, wr = ctx.saved_tensors
        gr = _rb(g) if ctx.g_r is None else ctx.g_r.to(g.dtype)
        assert gr.shape == g.shape
        # y = convT(x, w)  <=>  x-gradient = conv2d(g, w, stride 2), whose weight gradient (with x as the output gradient) is dW
        dx = F.conv2d(gr, wr, None, 2) if "dgrad" in ctx.prods else F.conv2d(g, w, None, 2)
        dw = torch.nn.grad.conv2d_weight(gr, w.shape, xr, stride=2) if "wgrad" in ctx.prods else \
            torch.nn.grad.conv2d_weight(g, w.shape, x, stride=2)
        # ("dbias": the gradient itself is STORED rounded -- a path that hands the up-sampled half of the concat gradient to the
        # ConvTranspose2d backward as bf16 sums the bias gradient from those values too, as torch.autocast's bf16 grad_output does)
        return dx, dw, (gr if "dbias" in ctx.prods else g).sum((0, 2, 3)), None, None, None


def _conv3x3(x, w):
    if _ROUNDING_RULE is not None:
        x_r, g_r = _replayed("conv3x3")
        sel = _ROUNDING_RULE("conv3x3", tuple(x.shape), tuple(w.shape))
        if sel:
            return _RoundedConv3x3.apply(x, w, _products(sel), x_r, g_r)
    return F.conv2d(x, w, None, 1, 1)


class _BNStoredZ(torch.autograd.Function):
    """Train-mode BatchNorm whose input was stored rounded (z_r) while its statistics come from the exact z."""

    @staticmethod
    def forward(ctx, z, z_r, gamma, beta, eps):
        mean = z.mean((0, 2, 3), keepdim=True)
        var = z.var((0, 2, 3), unbiased=False, keepdim=True)
        invstd = (var + eps).rsqrt()
        xhat = (z_r - mean) * invstd
        ctx.save_for_backward(xhat, invstd, gamma)
        return xhat * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        xhat, invstd, gamma = ctx.saved_tensors
        c1 = g.mean((0, 2, 3), keepdim=True)
        c2 = (g * xhat).mean((0, 2, 3), keepdim=True)
        dz = gamma.view(1, -1, 1, 1) * invstd * (g - c1 - xhat * c2)
        return dz, None, (g * xhat).sum((0, 2, 3)), g.sum((0, 2, 3)), None


def _stored_z(z, w):
    """-> the rounded z this unit stored (replayed, or bf16(z) under free rounding), or None where the unit keeps its output in fp32."""
    if _ROUNDING_RULE is None:
        return None
    i = _ROUNDING_CALLS["bn_z"]
    _ROUNDING_CALLS["bn_z"] = i + 1
    if _ROUNDING_REPLAY is not None and "bn_z" in _ROUNDING_REPLAY:
        z_r = _ROUNDING_REPLAY["bn_z"][i]
        return None if z_r is None else z_r.to(z.dtype)
    return _rb(z.detach()) if _ROUNDING_RULE("bn_z", tuple(z.shape), tuple(w.shape)) else None


def _conv_bn_relu(x, st, p, idx, training, routing=None):
    """conv3x3(pad 1, no bias) -> BN -> ReLU  (OV:47-49 / OV:51-53)."""
    z = _conv3x3(x, st[f"{p}.{idx}.weight"])
    b = idx + 1
    z_r = _stored_z(z, st[f"{p}.{idx}.weight"]) if training else None
    if z_r is not None:
        assert z_r.shape == z.shape
        with torch.no_grad():       # running statistics: from the exact z, F.batch_norm's update (unbiased variance, momentum 0.1)
            n = z.numel() // z.shape[1]
            st[f"{p}.{b}.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * z.detach().mean((0, 2, 3)))
            st[f"{p}.{b}.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * z.detach().var((0, 2, 3), unbiased=True) if n > 1 else 0.0)
        y = _BNStoredZ.apply(z, z_r, st[f"{p}.{b}.weight"], st[f"{p}.{b}.bias"], BN_EPS)
        st[f"{p}.{b}.num_batches_tracked"] += 1
        return F.relu(y) if routing is None else routing.relu(y)
    y = F.batch_norm(z, st[f"{p}.{b}.running_mean"], st[f"{p}.{b}.running_var"],
                     st[f"{p}.{b}.weight"], st[f"{p}.{b}.bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        st[f"{p}.{b}.num_batches_tracked"] += 1
    return F.relu(y) if routing is None else routing.relu(y)


def _double_conv(x, st, block, training, routing=None):
    p = _dc_prefix(block)
    return _conv_bn_relu(_conv_bn_relu(x, st, p, 0, training, routing), st, p, 3, training, routing)


def upsample_cat(x1, x2, st, block, bilinear=False):
    """Up.forward up to (and including) the concat: OV:91-100."""
    if bilinear:
        u = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        w = st[f"{block}.up.weight"]
        x_r, g_r = _replayed("convT2x2") if _ROUNDING_RULE is not None else (None, None)
        sel = _ROUNDING_RULE("convT2x2", tuple(x1.shape), tuple(w.shape)) if _ROUNDING_RULE is not None else False
        if sel:
            u = _RoundedConvT2x2.apply(x1, w, st[f"{block}.up.bias"], _products(sel), x_r, g_r)
        else:
            u = F.conv_transpose2d(x1, w, st[f"{block}.up.bias"], stride=2)
    dy = x2.shape[2] - u.shape[2]
    dx = x2.shape[3] - u.shape[3]
    u = F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return torch.cat([x2, u], dim=1)       # skip first, upsampled second (OV:100)


def unet_pass(x, st, training=True, routing=None, bilinear=False):
    """UNet.forward (OV:142-153): returns (x1, y1) = (first-block features, last-block features).
    bilinear: the nn.Upsample variant of Up (OV:83-84; Onet itself always builds bilinear=False, OV:162)."""
    feats = [_double_conv(x, st, "inc", training, routing)]
    for name, _, _ in ENC[1:]:
        pooled = F.max_pool2d(feats[-1], 2) if routing is None else routing.maxpool2(feats[-1])
        feats.append(_double_conv(pooled, st, name, training, routing))
    y = feats[-1]
    for lvl, (name, _, _) in enumerate(DEC):
        skip = feats[3 - lvl]
        y = _double_conv(upsample_cat(y, skip, st, name, bilinear), st, name, training, routing)
    return feats[0], y


def head(L, Hf):
    """V = einsum('bpxy,bpxy->bxy', L, H).unsqueeze(1)  (OV:176-177)."""
    return (L * Hf).sum(dim=1, keepdim=True)


def onet_forward(X, top, dwn=None, training=True, bias=0.0, routing=None):
    """Onet.forward (OV:174-191).  ``top``/``dwn`` are un-prefixed UNet state
    dicts of tensors (parameters may require grad); ``dwn is None`` == shared."""
    dwn = top if dwn is None else dwn
    Lt, Ht = unet_pass(X, top, training, routing)
    Vt = head(Lt, Ht)
    Xd = torch.clip(1 - X + bias, 0, 1)
    Ld, Hd = unet_pass(Xd, dwn, training, routing)
    Vd = head(Ld, Hd)
    S = torch.softmax(torch.cat([Vt, Vd], dim=1), dim=1)     # Softmax2d
    return Lt, Vt, Ld, Vd, S


def log1pexp_quirk(x):
    """Effective function of Onet.log1pexp (OV:237-251), autograd-equivalent.

    The reference mutates x in place between range tests, so step-1 outputs
    (exp(x) ~ 0 for x <= -37) are re-captured by step 2 and become
    log(1+exp(exp(x))) = ln 2.  Piecewise on the ORIGINAL x:
        x <= -37        : log(1 + exp(exp(x)))     (= ln 2;  d/dx ~ 0.5*exp(x))
        -37 < x <= 18   : log(1 + exp(x))
        18 < x < 33.3   : x + exp(-x)
        otherwise       : x
    (step-2 outputs never exceed 18 in fp32, so step 3 cannot re-capture them)."""
    lo = x <= -37.0
    mid = (x > -37.0) & (x <= 18.0)
    hi = (x > 18.0) & (x < 33.3)
    xs = torch.where(mid, x, torch.zeros_like(x))
    f_lo = torch.log(1 + torch.exp(torch.exp(torch.where(lo, x, torch.full_like(x, -40.0)))))
    f_mid = torch.log(1 + torch.exp(xs))
    xh = torch.where(hi, x, torch.full_like(x, 20.0))
    f_hi = xh + torch.exp(-xh)
    return torch.where(lo, f_lo, torch.where(mid, f_mid, torch.where(hi, f_hi, x)))


def jsd(Li, Si, Sp):
    """Onet.jensen_shannon_divergence (OV:221-235); the p=64 x p=1 einsum
    broadcast equals Si * sum_p Li (verified against the reference, §8c)."""
    sL = Li.sum(dim=1)
    LS = sL * Si[:, 0]
    LSp = sL * Sp[:, 0]
    return -log1pexp_quirk(-LS).mean() - log1pexp_quirk(LSp).mean()


def compute_loss(Lt, St, Ld, Sd):
    """Onet.compute_loss (OV:253-267)."""
    return -(jsd(Lt, St, Sd) + jsd(Ld, Sd, St)) / 2


def predict_label(S):
    """Onet.predict_label (OV:193-202): argmax over the 2 channels, ties -> 0."""
    return torch.argmax(S, dim=1)


def clone_state(sd, requires_grad=True):
    """Fresh leaf tensors from a (un-prefixed) UNet state dict."""
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and "running" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def train_mode_step(X, top, dwn=None, routing=None):
    """fwd + loss + backward; returns (outputs, loss, grads dict)."""
    Lt, Vt, Ld, Vd, S = onet_forward(X, top, dwn, training=True, routing=routing)
    loss = compute_loss(Lt, S[:, 0:1], Ld, S[:, 1:2])
    loss.backward()
    grads = OrderedDict((k, v.grad) for k, v in top.items() if v.requires_grad)
    if dwn is not None:
        for k, v in dwn.items():
            if v.requires_grad:
                grads["dwnu." + k] = v.grad
    return (Lt, Vt, Ld, Vd, S), loss, grads
