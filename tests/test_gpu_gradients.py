"""Gradient parity of the HIP path at 2e-4 -- five times tighter than the north-star's 1e-3 -- on EVERY element of EVERY
parameter gradient, including the B=32 256x256 benchmark dispatch (F(4x4,3x3) with both fused epilogues, F(3x3,4x4)
and F(2x2,3x3) weight gradients, fused ConvTranspose2d GEMMs, pooling backward with the BatchNorm reduce).

Why "routing-controlled": the Onet gradient is discontinuous in every ReLU / max-pool decision.  The reference's own fp32
gradients are 3e-3..7e-3 from its fp64 gradients on every parameter at every size (goldens `routed_*`, minted from the real
reference; tests/test_oracle_routing.py), so no fp32 implementation can be held to 1e-3 against a FREE fp64 evaluation --
and a bound loose enough to pass would also pass a 1 % systematic kernel error.  Here the fp64 oracle takes the discrete
decisions of the HIP run (oracle.Routing.from_activations: ReLU masks and pooling indices rebuilt from the HIP
activations); what remains is a smooth function, and the HIP gradients must match it to fp32 rounding.  The decisions
themselves are audited: wherever the free fp64 evaluation would decide differently, the pre-activation (or the window
gap) must lie within rounding distance of the switching point, and such elements must be rare.

Forward outputs / loss / labels / running statistics stay on the plain 1e-3 against the reference goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import onet_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
GRAD_TOL = 2e-4          # relative L2 per parameter tensor, HIP fp32 vs routed fp64 oracle (measured: <= 4.4e-5 on the
                         # unsaturated cases, 1.01e-4 on the saturated B=32 benchmark batch)
FLIP_DIST = 2e-4         # a decision may differ from the free fp64 one only this close to its switch (of the tensor's max)
FLIP_FRAC = 2e-4         # ... and on at most this fraction of a tensor's elements


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _to64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def _record_units(monkeypatch, B, keep, device):
    """Capture the output of every Conv-BN-ReLU unit for the images `keep`.  -> (rec, finish): `finish()` restores the
    hook and returns the activations in the ORACLE's call order (18 units of the X pass, then 18 of the 1-X pass),
    whichever way the model ran: twin batch (18 launches of 2B images: both halves in one) or two passes (36 launches of B)."""
    from onet_amd import functional as Fn
    from onet_amd import ops
    idx = torch.tensor(list(keep), device=device)
    rec = []
    real = Fn.ConvBNReLUFn.apply
    # every unit's activation must exist to be read: normalise-on-load (which leaves a placeholder where the first unit of a
    # DoubleConv would write its output) is switched off; it is bit-identical (test_bn_on_load_model_step_is_bit_identical)
    monkeypatch.setattr(ops, "BN_ON_LOAD", False)
    # ... and under pre-split storage (the default) every unit also leaves its fp32 activation -- the tensor its pre-split parts are
    # split from -- beside the pre-split form its consumers read (ops.PRESPLIT_KEEP_FP32: the kernels that run are unchanged)
    monkeypatch.setattr(ops, "PRESPLIT_KEEP_FP32", True)

    def apply(*a, **k):
        out = real(*a, **k)
        o = out.detach()
        if o.shape[0] == 2 * B:
            rec.append((o[idx].cpu(), o[idx + B].cpu()))
        else:
            assert o.shape[0] == B
            rec.append((o[idx].cpu(),))
        return out

    monkeypatch.setattr(Fn.ConvBNReLUFn, "apply", staticmethod(apply))

    def finish():
        monkeypatch.setattr(Fn.ConvBNReLUFn, "apply", real)
        if all(len(r) == 2 for r in rec):
            return [r[0] for r in rec] + [r[1] for r in rec]
        assert all(len(r) == 1 for r in rec)
        return [r[0] for r in rec]

    return rec, finish


def _hip_step_recording(m, X, monkeypatch, keep):
    """One HIP training step (TS:210-217 order) with every unit's output captured -> (outputs, loss, acts in the oracle's
    call order)."""
    rec, finish = _record_units(monkeypatch, X.shape[0], keep, X.device)
    m.zero_grad()
    Lt, Vt, Ld, Vd, S = m(X)
    loss = m.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
    loss.backward()
    acts = finish()
    assert len(acts) == 36
    return (Lt, Vt, Ld, Vd, S), loss, acts


def _routed_oracle(Xcpu, C, seed, gain, acts, bshare=True):
    r = orc.Routing.from_activations(acts)
    top = orc.clone_state(_to64(orc.det_state_dict(C, seed, head_gain=gain)))
    dwn = None if bshare else orc.clone_state(_to64(orc.det_state_dict(C, seed + 1, head_gain=gain)))
    outs, loss, grads = orc.train_mode_step(Xcpu.double(), top, dwn, routing=r)
    return outs, loss, grads, r


def _check(m, grads64, routing, what, named=None, tol=None, flip_dist=None, flip_frac=None):
    """grads64: the oracle's dict -- un-prefixed keys = the top (or only) U-Net, "dwnu." + key = an unshared down U-Net"""
    if named is None:
        named = {(k[5:] if k.startswith("topu.") else k): v for k, v in m.named_parameters()}
    assert set(grads64) == set(named), set(grads64) ^ set(named)
    worst = (0.0, "")
    for k, t in grads64.items():
        a = named[k].grad.detach().cpu().double()
        e = float((a - t).norm() / t.norm())
        worst = max(worst, (e, k))
        assert e <= (tol or GRAD_TOL), (what, k, e)
        # element-wise too: no single element further off than GRAD_TOL of the tensor's largest magnitude x sqrt(n) share
        assert float((a - t).abs().max()) <= 20 * (tol or GRAD_TOL) * float(t.abs().max()), (what, k)
    flips = sum(a[1] for a in routing.audit)
    dist = max(a[3] for a in routing.audit)
    frac = max(a[1] / a[2] for a in routing.audit)
    assert dist <= (flip_dist or FLIP_DIST), (what, "a decision differs from the exact one far from its switching point", dist)
    assert frac <= (flip_frac or FLIP_FRAC), (what, "too many decisions differ from the exact ones", frac)
    print(f"{what}: worst gradient error {worst[0]:.2e} ({worst[1]}); {flips} of "
          f"{sum(a[2] for a in routing.audit)} decisions differ from the exact ones, farthest {dist:.1e} from the switch")


def _model(C, gain, dev, seed=1981, bshare=True):
    import Onet_vanilla_20240606 as ov
    m = ov.Onet(in_chns=C, binit=True, bshare=bshare)
    m.load_state_dict(orc.onet_state_dict(C, seed, bshare, head_gain=gain))
    return m.to(dev).train()


CASES = {
    # tag: (B, C, H, W, head_gain, algorithms)
    # (round 5: the default dispatch at every size; one forced run per fp32-MFMA family -- F(4x4) and the direct kernels at 64 pixels
    # (F(4x4) at 256 pixels: the golden forward tests and bench.py's f32_mfma_only loop); the default dispatch at 256 pixels with the benchmark's batch is the benchmark-dispatch test below, the non-default
    # Settings.grad_f16 of the in-staging kernels is held at op level (test_gpu_ops.py).  The suite's wall time is bounded by the fp64
    # CPU oracle of these cases: 20-35 s each at 128 / 256 pixels)
    "b8_c1_128": (8, 1, 128, 128, 0.3, ("auto",)),
    "b4_c1_256": (4, 1, 256, 256, 0.3, ("auto",)),
    "b2_c3_64_saturated": (2, 3, 64, 64, 1.0, ("auto", "winograd4", "direct")),
    "b3_c1_40_padpath": (3, 1, 40, 40, 1.0, ("auto",)),
    "b1_c3_512": (1, 3, 512, 512, 0.3, ("auto",)),        # BASELINE configs[4]'s tile shape (3-channel 512 x 512), batch statistics over ONE image
}


@pytest.mark.parametrize("tag,algo", [(t, a) for t, c in CASES.items() for a in c[5]])
def test_every_gradient_element_vs_routed_fp64_oracle(dev, tag, algo, monkeypatch):
    from onet_amd import ops
    B, C, H, W, gain, algos = CASES[tag]
    monkeypatch.setattr(ops, "CONV_ALGO", algo)
    X = orc.det_input(B, C, H, W)
    m = _model(C, gain, dev)
    (Lt, Vt, Ld, Vd, S), loss, acts = _hip_step_recording(m, X.to(dev), monkeypatch, range(B))
    (oLt, oVt, oLd, oVd, oS), oloss, g64, r = _routed_oracle(X, C, 1981, gain, acts)
    assert abs(loss.item() - float(oloss)) <= 1e-5 * abs(float(oloss))
    assert float((Vt.detach().cpu().double() - oVt.detach()).abs().max()) <= 1e-4 * float(oVt.detach().abs().max())
    # S = softmax([Vt, Vd]): |dS| <= |dV| / 4, and |V| reaches 35 in the saturated cases
    assert float((S.detach().cpu().double() - oS.detach()).abs().max()) <= 1e-4 * max(1.0, float(oVt.detach().abs().max()) / 4)
    _check(m, g64, r, f"{tag}/{algo}")
    g = os.path.join(G, f"onet_routed_{tag}.npz")
    if os.path.exists(g):                       # the same case as recorded from the REAL reference (fp32): forward at 1e-3
        g = np.load(g)
        assert abs(loss.item() - g["losses"][0]) <= 1e-3 * abs(g["losses"][0])
        for name, t in (("Vt", Vt), ("Vd", Vd), ("S", S)):
            ref = g[name]
            assert np.abs(t.detach().cpu().numpy()[:, :, ::37, :] - ref).max() <= 1e-3 * np.abs(ref).max(), name
        lab = m.predict_label(S).cpu().numpy().astype(np.uint8)[:, ::37, :]
        tie = np.abs(g["S"][:, 0] - g["S"][:, 1]) < 1e-3
        assert np.array_equal(lab[~tie], g["label"][~tie])
        rm = torch.cat([b.reshape(-1) for n, b in m.topu.named_buffers() if n.endswith("running_mean")]).cpu().numpy()
        rv = torch.cat([b.reshape(-1) for n, b in m.topu.named_buffers() if n.endswith("running_var")]).cpu().numpy()
        assert np.abs(rm - g["bn_rm"][:rm.size]).max() <= 1e-3 * np.abs(g["bn_rm"]).max()
        assert np.abs(rv - g["bn_rv"][:rv.size]).max() <= 1e-3 * np.abs(g["bn_rv"]).max()
        # gradients of the reference's fp32 run: its decisions differ from ours at a few dozen elements, so this
        # comparison carries the decision noise the reference has against ITSELF in fp64 (measured in the golden)
        offs = g["grad_offs"]
        v32, v64 = g["grad_vals"].astype(np.float64), g["grad_vals64"]
        named = dict(m.named_parameters())
        eps_ref = max(np.linalg.norm(v32[offs[i]:offs[i + 1]] - v64[offs[i]:offs[i + 1]]) /
                      np.linalg.norm(v64[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1))
        for i, n in enumerate(str(s) for s in g["grad_names"]):
            gr = named[n].grad.detach().reshape(-1).double().cpu()
            v = (gr if gr.numel() <= 4096 else gr[:: gr.numel() // 1024][:1024]).numpy()
            e = np.linalg.norm(v - v32[offs[i]:offs[i + 1]]) / np.linalg.norm(v32[offs[i]:offs[i + 1]])
            # (two independent fp32 evaluations each sit eps_ref from exact; measured 1.4-3.1 x eps_ref over the builds of round 2,
            # 4.3 x with the split-bf16 kernels of round 3 -- which parameter is worst, and by how much, moves with the last bit
            # of any convolution: this is decision noise, a sanity bound.  The strict statement is _check above.)
            assert e <= 6 * eps_ref, (n, e, eps_ref)


def test_benchmark_dispatch_b32_256_every_gradient_element(dev, monkeypatch):
    """BASELINE configs[1] as benchmarked: B=32, 1x256x256, default dispatch.  The batch is the B=2 golden input tiled
    16 times: BatchNorm statistics, loss and every parameter gradient of a tiled batch equal those of the tile (means
    over B*H*W), so the fp64 oracle runs at B=2 under the decisions the HIP run took on the first two images of each
    half -- after checking that all 16 copies decided identically."""
    from onet_amd import ops
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~25 GB of free HBM")
    g = np.load(os.path.join(G, "onet_b2_c1_256.npz"))
    x2 = orc.det_input(2, 1, 256, 256)
    X = x2.to(dev).repeat(16, 1, 1, 1)
    m = _model(1, 1.0, dev)
    used = {}
    for name in ("conv3x3_fwd_bn_partials", "conv3x3_dgrad_bnreduce", "conv3x3_split", "conv3x3_split_wgrad", "conv3x3_winograd4",
                 "conv3x3_winograd4_wgrad", "convT2x2_wgrad", "convT2x2_dgrad", "conv3x3_pre_bn_partials",
                 "conv3x3_split_pre", "conv3x3_split_wgrad_pre", "bn_relu_bwd_split", "convT2x2_fwd_p", "convT2x2_fwd_slots", "conv3x3_split_dgrad_pre_bnreduce",
                 "conv3x3_split_dgrad_pre_slots", "convT2x2_dgrad_slots", "convT2x2_wgrad_slots"):
        real = getattr(ops, name)

        def spy(*a, _real=real, _name=name, **k):
            out = _real(*a, **k)
            if out is not None and not (isinstance(out, tuple) and out[-1] is None and _name == "conv3x3_fwd_bn_partials"):
                used[_name] = used.get(_name, 0) + 1
            return out

        monkeypatch.setattr(ops, name, spy)
    copies = []
    from onet_amd import functional as Fn
    real_apply = Fn.ConvBNReLUFn.apply

    def apply(*a, **k):
        out = real_apply(*a, **k)
        o = out.detach()
        copies.append(bool(torch.equal(o[0:2] > 0, o[30:32] > 0)) and bool(torch.equal(o[32:34] > 0, o[46:48] > 0)))
        return out

    monkeypatch.setattr(Fn.ConvBNReLUFn, "apply", staticmethod(apply))
    (Lt, Vt, Ld, Vd, S), loss, acts = _hip_step_recording(m, X, monkeypatch, range(2))
    assert all(copies) and len(copies) == 18, "tile copies took different ReLU decisions"
    # the benchmark's kernels really ran: forward with fused statistics (split-bf16 kernel on the 32-pixel-and-wider levels, fp32
    # Winograd F(4x4) on the 16-pixel level), split-bf16 input gradients, split-bf16 weight gradients of all 17 layers with
    # >= 16 input channels (round 3: also the 32- and 16-pixel levels), the ConvTranspose2d GEMMs
    diag = bool(os.environ.get("ONET_DIAG"))        # diagnostic runs with parts of the dispatch switched off (ONET_SPLIT_DGRAD=0 ...)
    if ops.PRESPLIT and not diag:
        # round 4, pre-split storage: the 15 layers on maps >= 32 pixels wide with >= 16 input channels run forward (with fused
        # statistics), input gradient and weight gradient on pre-split operands (LDS-DMA staged kernels); the stem and the 16-pixel
        # level keep the fp32-operand kernels (2 split weight gradients there); the up-sampled concat halves leave the GEMM pre-split
        # (+ the two layers of the 16-pixel level with ops.PRESPLIT_W16, the default: two images side by side per forward tile)
        n = 17 if ops.PRESPLIT_W16 else 15
        assert used.get("conv3x3_pre_bn_partials", 0) == n and used.get("conv3x3_split_wgrad_pre", 0) == n, used
        # (round 5: the second convolution of each DoubleConv on maps >= 32 pixels runs its input gradient with the first unit's
        # BatchNorm-backward reduce in the epilogue: 8 of the 17 input-gradient launches; the fused launch does not count as
        # conv3x3_split_pre)
        fused = used.get("conv3x3_split_dgrad_pre_bnreduce", 0)
        assert fused == (8 if ops.FUSE_DGRAD_REDUCE else 0), used
        # (... and the first convolution of each decoder block hands the up-sampled half of its input gradient to the ConvTranspose2d
        # backward GEMMs pre-split: 4 launches of conv3x3_split_dgrad_pre_slots)
        slotted = used.get("conv3x3_split_dgrad_pre_slots", 0)
        assert slotted == (4 if (ops.CONVT_SLOTS and ops.CONVT_BWD_SLOTS) else 0), used
        assert used.get("conv3x3_split_pre", 0) + fused + slotted == n + n and used.get("bn_relu_bwd_split", 0) == n, used
        # (round 5: the four ConvTranspose2d layers run all three GEMMs on slot operands)
        assert used.get("convT2x2_fwd_slots" if ops.CONVT_SLOTS else "convT2x2_fwd_p", 0) == 4, used
        assert used.get("convT2x2_dgrad_slots", 0) == used.get("convT2x2_wgrad_slots", 0) == slotted and \
            used.get("convT2x2_dgrad", 0) == used.get("convT2x2_wgrad", 0) == 4 - slotted, used
        assert used.get("conv3x3_split_wgrad", 0) == 17 - n and used.get("conv3x3_winograd4_wgrad", 0) == 0, used
        assert used.get("conv3x3_winograd4", 0) == (0 if ops.PRESPLIT_W16 else 4), used
    else:
        assert diag or (used.get("conv3x3_fwd_bn_partials", 0) >= 14 and used.get("conv3x3_split", 0) >= 10), used
        if ops.SPLIT_AUTO and not diag:
            assert used.get("conv3x3_split_wgrad", 0) == 17 and used.get("conv3x3_winograd4_wgrad", 0) == 0, used
    assert used.get("convT2x2_wgrad", 0) + used.get("convT2x2_wgrad_slots", 0) == 4 and (ops.PRESPLIT or used.get("conv3x3_winograd4", 0) >= 1), used
    assert abs(loss.item() - g["losses"][0]) <= 1e-3 * abs(g["losses"][0])
    assert np.abs(Vt.detach().cpu().numpy()[:2, :, ::37, :] - g["Vt"]).max() <= 1e-3 * np.abs(g["Vt"]).max()
    _, oloss, g64, r = _routed_oracle(x2, 1, 1981, 1.0, acts)
    assert abs(loss.item() - float(oloss)) <= 1e-5 * abs(float(oloss))
    # This batch has the SATURATED head of the reference's initialisation (|V| up to 47: a fifth of the pixels at S = 0 / 1 to fp32
    # precision), which amplifies every rounding difference of the FORWARD pass: the fp32-MFMA Winograd dispatch of round 2 measured
    # 1.0e-4 here; the split kernels with bf16 parts in the forward convolution (16-bit operands) 3.3e-4 -- all of it from the forward
    # kernel, the split input / weight gradients and ConvTranspose2d GEMMs add nothing (ONET_SPLIT_DGRAD=0 etc.: 3.26e-4) -- and with
    # fp16 parts in the forward convolution (22-bit operands, the default) 1.0e-4 again.  The bound is the common 2e-4.
    _check(m, g64, r, "B=32 256x256 benchmark dispatch", tol=None if ops.SPLIT_F16 else 4e-4)


@pytest.mark.parametrize("mode", ["noshare", "two-pass", "two-pass-winograd4"])
def test_every_gradient_element_two_pass_and_unshared(dev, mode, monkeypatch):
    """The reference's own order -- topu(X), then dwnu(1 - X) (OV:174-185) -- instead of the twin batch, at the same 2e-4
    on every element: "noshare" = `Onet(bshare=False)` (OV:165-166: two U-Nets, 124 parameter gradients, no sum over the
    passes), "two-pass" = shared weights with `settings.twin = False` (ONET_TWIN=0: autograd sums the two passes'
    contributions; the weight-gradient slots are taken once and added to once)."""
    from onet_amd import ops
    B, C, H, W, gain = 4, 1, 64, 64, 0.3
    bshare = mode != "noshare"
    X = orc.det_input(B, C, H, W)
    m = _model(C, gain, dev, bshare=bshare)
    m.settings = ops.Settings(twin=False, conv="winograd4" if mode.endswith("winograd4") else None)
    (Lt, Vt, Ld, Vd, S), loss, acts = _hip_step_recording(m, X.to(dev), monkeypatch, range(B))
    (oLt, oVt, oLd, oVd, oS), oloss, g64, r = _routed_oracle(X, C, 1981, gain, acts, bshare=bshare)
    assert len(g64) == (62 if bshare else 124)
    assert abs(loss.item() - float(oloss)) <= 1e-5 * abs(float(oloss))
    assert float((Vd.detach().cpu().double() - oVd.detach()).abs().max()) <= 1e-4 * float(oVd.detach().abs().max())
    _check(m, g64, r, mode)


def _bf16_rule(kind, xs, ws):
    """Which matrix products of the model the bf16 conv path evaluates with bf16-rounded operands: the 3x3 layers that run the
    pre-split kernels on one part of plain bf16 (ops.pre_layer_ok under conv == "bf16": 32-channel chunks, maps made of full
    16 x 32-pixel tiles; the 16-pixel level joins them only at batches that fill the chip -- not at this file's sizes; the stem, the
    smaller maps and odd shapes run the fp32 dispatch, round 5) and the 128 x 128 ConvTranspose2d GEMMs.  The test checks the rule
    against the run's dispatch (the units whose weight gradient took pre-split operands)."""
    if kind == "bn_z":       # which units STORE their conv output as bf16 (ops.z16_storage): the same layers
        _, Cout, H, W = xs
        return ws[1] % 32 == 0 and Cout % 32 == 0 and H % 16 == 0 and W % 32 == 0
    if kind == "conv3x3":
        _, Cin, H, W = xs
        return Cin % 32 == 0 and ws[0] % 32 == 0 and H % 16 == 0 and W % 32 == 0
    # ConvTranspose2d: the forward and input-gradient GEMMs tile 128 pixels, the weight-gradient GEMM 32 (convt_gemm.hip) --
    # on an 8 x 8 map only the weight gradient takes the bf16-operand fast path
    _, Cin, h, w = xs
    if Cin % 128 or ws[1] % 32:
        return False
    if (Cin, h, w) in CONVT_SLOT_SHAPES:      # round 5, slot-operand backward: dy itself is stored as bf16 -- the bias gradient sums those values
        return {"fwd", "dgrad", "wgrad", "dbias"}
    return {"fwd", "dgrad", "wgrad"} if (h * w) % 128 == 0 else ({"wgrad"} if (h * w) % 32 == 0 else False)


CONVT_SLOT_SHAPES = set()   # (Cin, h, w) of the ConvTranspose2d layers whose backward GEMMs read slot operands in the recorded run (filled by
                            # _record_bf16_operands' spy on ops.convT2x2_wgrad_slots: the dispatch is the run's, the rule follows it)


BF16_GRAD_TOL = 1e-4     # per parameter tensor, relative L2, with the HIP run's rounded operands replayed (measured 7.1e-6 at
                         # B=8 128 x 128, 1.3e-5 at B=4 256 x 256; free rounding: 2e-2; against the unrounded gradients: 5e-2)


def _record_bf16_operands(monkeypatch, B):
    """Capture the fp32 operands of every 3x3 weight-gradient call (x = the unit's forward input, dz = the gradient of its
    pre-activation: the operands of all three products of the unit) and of every ConvTranspose2d weight-gradient call (x1,
    the concat-gradient window), rounded to bf16 as the kernels round them on load.  -> finish() -> replay lists in the
    ORACLE's call order (18 units / 4 Up blocks of the X pass, then of the 1-X pass; the twin batch holds both passes)."""
    from onet_amd import ops
    rb = lambda t: t.detach().to(torch.bfloat16).cpu()          # RNE, == v_cvt_pk_bf16_f32 (tests/test_gpu_ops.py)
    conv, convt, zs, pre = [], [], [], []
    CONVT_SLOT_SHAPES.clear()
    real_w, real_t, real_p, real_ts = ops.conv3x3_wgrad_auto, ops.convT2x2_wgrad, ops.conv3x3_split_wgrad_pre, ops.convT2x2_wgrad_slots
    real_zp, real_zf = ops.conv3x3_pre_bn_partials, ops.conv3x3_fwd_bn_partials

    # the conv output z of every unit AS STORED (round 5: bf16 where ops.z16_storage() says so; None for a unit that keeps fp32): the
    # oracle normalises the stored values with the statistics of its own exact z (oracle._BNStoredZ)
    def z_pre(*a, **k):
        z, cm = real_zp(*a, **k)
        assert z.shape[0] == 2 * B
        zs.append(z.detach().cpu() if z.dtype == torch.bfloat16 else None)
        return z, cm

    def z_fp32(*a, **k):
        z, cm = real_zf(*a, **k)
        zs.append(None)
        return z, cm

    monkeypatch.setattr(ops, "conv3x3_pre_bn_partials", z_pre)
    monkeypatch.setattr(ops, "conv3x3_fwd_bn_partials", z_fp32)

    def wgrad(x, dz, *a, **k):
        assert x is not None and dz is not None and x.shape[0] == 2 * B
        conv.append((rb(x), rb(dz)))
        pre.append(False)
        return real_w(x, dz, *a, **k)

    def wgrad_pre(xP, dzP, *a, **k):
        # pre-split storage with ONE part of plain bf16 (round 4): the operands the kernels consumed, as their producers rounded them
        assert xP.shape[0] == 2 * B and xP.shape[3] == 1 and xP.dtype == torch.bfloat16
        nchw = lambda P: P.detach()[:, :, :, 0].permute(0, 1, 4, 2, 3).reshape(P.shape[0], P.shape[1] * 8, P.shape[2], P.shape[4]).cpu()
        conv.append((nchw(xP), nchw(dzP)))
        pre.append(True)
        return real_p(xP, dzP, *a, **k)

    def wgrad_t(x, dy, *a, **k):
        assert x.shape[0] == 2 * B
        convt.append((rb(x), rb(dy)))
        return real_t(x, dy, *a, **k)

    def wgrad_ts(xP, dyP, *a, **k):
        # round 5: the ConvTranspose2d backward GEMMs on slot operands -- one part of plain bf16 here: what the kernels consumed
        assert xP.shape[0] == 2 * B and xP.shape[3] == 1 and dyP.shape[3] == 1
        nchw = lambda P: P.detach()[:, :, :, 0].permute(0, 1, 4, 2, 3).reshape(P.shape[0], P.shape[1] * 8, P.shape[2], P.shape[4]).cpu()
        convt.append((nchw(xP), nchw(dyP)))
        CONVT_SLOT_SHAPES.add((xP.shape[1] * 8, xP.shape[2], xP.shape[4]))
        return real_ts(xP, dyP, *a, **k)

    monkeypatch.setattr(ops, "conv3x3_wgrad_auto", wgrad)
    monkeypatch.setattr(ops, "convT2x2_wgrad", wgrad_t)
    monkeypatch.setattr(ops, "convT2x2_wgrad_slots", wgrad_ts)
    monkeypatch.setattr(ops, "conv3x3_split_wgrad_pre", wgrad_pre)
    # the stem's dz is recorded too: materialise it (by default its weight gradient forms it on load -- the same bits,
    # tests/test_gpu_ops.py::test_stem_wgrad_with_bn_backward_on_load)
    monkeypatch.setattr(ops, "STEM_WGRAD_BN", False)

    def finish():
        monkeypatch.setattr(ops, "conv3x3_wgrad_auto", real_w)
        monkeypatch.setattr(ops, "convT2x2_wgrad", real_t)
        monkeypatch.setattr(ops, "convT2x2_wgrad_slots", real_ts)
        monkeypatch.setattr(ops, "conv3x3_split_wgrad_pre", real_p)
        monkeypatch.setattr(ops, "conv3x3_pre_bn_partials", real_zp)
        monkeypatch.setattr(ops, "conv3x3_fwd_bn_partials", real_zf)
        assert len(conv) == 18 and len(convt) == 4 and len(zs) == 18, (len(conv), len(convt), len(zs))
        out = {"conv3x3": [], "convT2x2": [], "bn_z": [], "pre": list(reversed(pre)) * 2}
        for p in range(2):                                       # backward visits the units last to first
            out["conv3x3"] += [(x[p * B:(p + 1) * B], g[p * B:(p + 1) * B]) for x, g in reversed(conv)]
            out["convT2x2"] += [(x[p * B:(p + 1) * B], g[p * B:(p + 1) * B]) for x, g in reversed(convt)]
            out["bn_z"] += [None if z is None else z[p * B:(p + 1) * B] for z in zs]     # (forward order)
        return out

    return finish


@pytest.mark.parametrize("tag", ["b8_c1_128", "b4_c1_256"])
def test_bf16_path_every_gradient_element_vs_routed_rounded_oracle(dev, tag, monkeypatch):
    """BASELINE configs[2]'s arithmetic held to the fp32 path's element-wise statement instead of "gradient norms within 10 %": the
    fp64 oracle evaluates the function the bf16 conv path computes -- every bf16 matrix product (forward, input gradient, weight
    gradient of the 3x3 layers and of the ConvTranspose2d GEMMs the path takes) on bf16-rounded operands, everything else
    unrounded (oracle.operand_rounding) -- under the HIP run's DECISIONS: its ReLU masks and pooling indices, and its roundings.
    The roundings must be replayed for the same reason as the masks: free rounding is chaotic (the oracle's note; the test
    measures it: with free rounding the two evaluations differ by ~1e-2 of the head logits' scale), while on the same rounded
    operands the two differ by fp32 summation only."""
    from onet_amd import ops
    B, C, H, W, gain, _ = CASES[tag]
    monkeypatch.setattr(ops, "CONV_ALGO", "bf16")
    # the layers >= 32 pixels wide run the LDS-DMA staged kernels on ONE part of plain bf16 operands written by their producers (nearest
    # even); the others take the fp32 dispatch
    X = orc.det_input(B, C, H, W)
    m = _model(C, gain, dev)
    finish_ops = _record_bf16_operands(monkeypatch, B)
    ops.profile_start(everything=False)
    try:
        (Lt, Vt, Ld, Vd, S), loss, acts = _hip_step_recording(m, X.to(dev), monkeypatch, range(B))
    finally:
        prof, _ = ops.profile_stop()
    replay = finish_ops()
    # (round 5: the pre-split layers store their conv output as bf16 -- 11 of 18 units per pass at 128 x 128 with B = 8: the levels
    # 32 pixels wide and wider except the stem)
    assert sum(z is not None for z in replay["bn_z"]) >= 20, [None if z is None else tuple(z.shape) for z in replay["bn_z"]]
    assert len(prof.get("conv3x3_split_pre_kernel", [])) >= 10 and len(prof.get("conv3x3_split_wgrad_pre_kernel", [])) >= 5, {k: len(v) for k, v in prof.items()}
    # the rule the oracle rounds by == the run's dispatch, unit by unit (and the stored-z units are the same ones)
    ran_pre = replay.pop("pre")
    for (x_r, g_r), was_pre, z_r in zip(replay["conv3x3"], ran_pre, replay["bn_z"]):
        want = _bf16_rule("conv3x3", tuple(x_r.shape), (g_r.shape[1], x_r.shape[1], 3, 3))
        assert want == was_pre == (z_r is not None), (tuple(x_r.shape), tuple(g_r.shape), want, was_pre, z_r is not None)
    with orc.operand_rounding(_bf16_rule, replay):
        (oLt, oVt, oLd, oVd, oS), oloss, g64, r = _routed_oracle(X, C, 1981, gain, acts)
    assert abs(loss.item() - float(oloss)) <= 1e-5 * abs(float(oloss))
    assert float((Vt.detach().cpu().double() - oVt.detach()).abs().max()) <= 1e-4 * float(oVt.detach().abs().max())
    _check(m, g64, r, f"bf16 path {tag}, roundings replayed", tol=BF16_GRAD_TOL)
    if tag != "b8_c1_128":
        return                                       # (the two extra fp64 evaluations below cost a minute at 256 x 256)
    # for the record (and so that the replay is not vacuous): the same evaluation with FREE rounding, and without any rounding
    with orc.operand_rounding(_bf16_rule):
        (_, fVt, _, _, _), _, gfree, _ = _routed_oracle(X, C, 1981, gain, acts)
    free_v = float((fVt.detach() - oVt.detach()).abs().max() / oVt.detach().abs().max())
    free_g = max(float((gfree[k] - g64[k]).norm() / g64[k].norm()) for k in g64)
    print(f"bf16 path {tag}: free rounding moves the head logits by {free_v:.1e} and the gradients by up to {free_g:.1e}")
    assert free_g > 10 * BF16_GRAD_TOL, "rounding does not matter here: the check would be vacuous"


@pytest.mark.parametrize("algo", ["auto", "direct"])
def test_every_gradient_element_bilinear_unet(dev, algo, monkeypatch):
    """A full `UNet(bilinear=True)` (OV:83-84 inside OV:104-153; incl. the F.pad path at 40 x 40) on its own: forward,
    a scalar coupling both outputs, backward -- every element of its 54 parameter gradients against the fp64 oracle under the
    HIP run's decisions; forward against the golden recorded from the REAL reference's UNet(bilinear=True)."""
    import Onet_vanilla_20240606 as ov
    from onet_amd import ops
    from test_oracle_routing import unet_loss
    monkeypatch.setattr(ops, "CONV_ALGO", algo)
    g = np.load(os.path.join(G, "unet_bilinear_b4_c1_40.npz"))
    B, C, H, W = [int(v) for v in g["meta"]]
    X = orc.det_input(B, C, H, W)
    m = ov.UNet(n_channels=C, n_classes=1, binit=True, bilinear=True)
    m.load_state_dict(orc.det_state_dict(C, 1981, bilinear=True))
    m = m.to(dev).train()
    rec, finish = _record_units(monkeypatch, B, range(B), dev)
    m.zero_grad()
    x1, y1 = m(X.to(dev))
    loss = unet_loss(x1, y1)
    loss.backward()
    acts = finish()
    assert len(acts) == 18
    assert abs(loss.item() - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    assert np.abs(y1.detach().cpu().numpy() - g["y1"]).max() <= 1e-3 * np.abs(g["y1"]).max()
    r = orc.Routing.from_activations(acts)
    sd = orc.clone_state(_to64(orc.det_state_dict(C, 1981, bilinear=True)))
    ox1, oy1 = orc.unet_pass(X.double(), sd, True, r, True)
    oloss = unet_loss(ox1, oy1)
    oloss.backward()
    assert abs(loss.item() - float(oloss)) <= 1e-5 * abs(float(oloss))
    _check(m, {k: v.grad for k, v in sd.items() if v.requires_grad}, r, f"UNet(bilinear=True)/{algo}", named=dict(m.named_parameters()))
