"""GPU parity, op by op: every HIP kernel (through the C ABI via onet_amd.ops / autograd
Functions) against the same op of the CPU oracle's arithmetic (PyTorch CPU fp32) on identical
seeded inputs.  Tolerance: BASELINE north_star says 1e-3 relative in fp32; single ops are held
to 2e-4 of the output scale (summation order is the only difference)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib_load():
    from onet_amd import _lib
    return _lib.load()

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from onet_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = np.random.Generator(np.random.PCG64([seed, *shape]))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def close(a, b, tol=2e-4, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


CONV_SHAPES = [
    # B, Cin, Cout, H, W, ks
    (2, 64, 64, 40, 48, 3),      # pixel-heavy tile, ragged edges
    (2, 16, 64, 16, 16, 3),      # 16-wide tile
    (1, 32, 128, 64, 64, 3),     # channel-heavy tile
    (2, 64, 256, 32, 32, 3),     # deep tile, 32 wide
    (2, 128, 256, 16, 16, 3),    # deep tile, 16 wide
    (64, 8, 2048, 16, 16, 3),    # 128-channel tile on 16-wide images
    (2, 3, 10, 17, 23, 3),       # odd everything (masked channels, scalar weight path)
    (2, 1, 64, 32, 32, 3),       # stem, Cin = 1
    (3, 20, 36, 9, 7, 3),        # tiny image
    (2, 128, 256, 12, 12, 1),    # 1x1 (convT sub-pixel GEMM)
    (2, 64, 64, 40, 40, 1),
    (2, 1024, 512, 4, 4, 3),     # very deep, tiny
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks", CONV_SHAPES)
def test_conv_fwd_dgrad_wgrad(dev, B, Cin, Cout, H, W, ks):
    from onet_amd import ops
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, ks, ks, seed=2, scale=(2.0 / (Cin * ks * ks)) ** 0.5)
    g = rnd(B, Cout, H, W, seed=3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    zr = F.conv2d(xr, wr, None, 1, ks // 2)
    zr.backward(g)

    xd, wd_, gd = x.to(dev), w.to(dev), g.to(dev)
    if ks == 3:
        wf, wdg = ops.pack3x3(wd_)
    else:
        # 1x1: packed layouts are plain transposes
        wf = wd_.reshape(Cout, Cin).t().contiguous().reshape(-1)
        wdg = wd_.reshape(Cout, Cin).contiguous().reshape(-1)
    z = ops.conv_fwd(xd, wf, Cout, ks)
    close(z, zr, what="fwd")
    dx = ops.conv_fwd(gd, wdg, Cin, ks)
    close(dx, xr.grad, what="dgrad")
    dw = ops.conv_wgrad(xd, gd, (Cout, Cin, ks, ks), ks)
    close(dw, wr.grad, tol=3e-4, what="wgrad")


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 40, 48), (2, 16, 64, 16, 16), (1, 32, 128, 64, 64),
                                             (3, 128, 256, 16, 16), (2, 24, 36, 9, 7), (3, 8, 64, 33, 31),
                                             (2, 1024, 512, 4, 4), (2, 4, 12, 17, 23), (1, 64, 128, 32, 96)])
def test_conv3x3_winograd4_fwd_dgrad(dev, B, Cin, Cout, H, W):
    """Winograd F(4x4,3x3) MFMA path against the direct CPU convolution (fwd and dgrad orientation);
    ragged edges, odd batch with the two-images-per-block layout (W <= 16), channel tails."""
    from onet_amd import ops
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(2.0 / (Cin * 9)) ** 0.5)
    g = rnd(B, Cout, H, W, seed=3)
    xr = x.clone().requires_grad_(True)
    zr = F.conv2d(xr, w, None, 1, 1)
    zr.backward(g)
    qf, qd = ops.pack3x3_winograd4(w.to(dev))
    close(ops.conv3x3_winograd4(x.to(dev), qf, Cout), zr, tol=2e-4, what="winograd4 fwd")
    close(ops.conv3x3_winograd4(g.to(dev), qd, Cin), xr.grad, tol=2e-4, what="winograd4 dgrad")


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 24, 64), (1, 32, 128, 64, 64), (3, 16, 36, 33, 68), (2, 128, 80, 16, 192),
                                             (5, 48, 64, 37, 128), (2, 24, 20, 9, 100),
                                             (4, 64, 64, 32, 32), (6, 80, 48, 19, 32), (8, 128, 64, 16, 16), (12, 32, 96, 7, 16),
                                             (2, 64, 128, 24, 64), (3, 40, 256, 17, 128), (4, 48, 256, 20, 32), (4, 64, 128, 9, 16)])
def test_conv3x3_split_wgrad(dev, B, Cin, Cout, H, W):
    """conv3x3_split_wgrad_kernel (both operands split into bf16 hi + mid, three MFMAs per term, row-streaming units,
    deterministic split-K) against the fp64 weight gradient at 2e-5 of its scale; ragged strips (W % 64 != 0), odd heights,
    channel tails, runs crossing image boundaries; 32- and 16-pixel-wide maps (2 / 4 images side by side per unit, each with
    its own zero halo); Cout % 128 == 0 on maps >= 32 pixels wide: the 128-channel block tile; bitwise reproducible."""
    from onet_amd import ops
    x = rnd(B, Cin, H, W, seed=11)
    g = rnd(B, Cout, H, W, seed=12)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(g.double())
    assert ops.split_wgrad_ok(x.to(dev), g.to(dev))
    dw = ops.conv3x3_split_wgrad(x.to(dev), g.to(dev), (Cout, Cin, 3, 3))
    close(dw, w.grad, tol=2e-5, what="split wgrad")
    assert torch.equal(dw, ops.conv3x3_split_wgrad(x.to(dev), g.to(dev), (Cout, Cin, 3, 3)))
    if W >= 64:     # narrower maps: only 32 / 16 pixels wide with a whole number of image groups
        assert not ops.split_wgrad_ok(x[..., :24].contiguous().to(dev), g[..., :24].contiguous().to(dev))
        assert ops.split_wgrad_ok(x[..., :32].contiguous().to(dev), g[..., :32].contiguous().to(dev)) == (B % 2 == 0)
    # a channel slice of a wider buffer as x (the decoder's concat buffers): batch stride != Cin * H * W
    wide = torch.zeros(B, Cin + 16, H, W)
    wide[:, 16:] = x
    assert torch.equal(dw, ops.conv3x3_split_wgrad(wide.to(dev)[:, 16:], g.to(dev), (Cout, Cin, 3, 3)))
    # round 4: fp16 parts of the scaled operands (dz at a gradient's magnitude, scaled from its magnitude slots): 2e-6 of scale
    tiny = (3e-6 * g).to(dev)
    dwh = ops.conv3x3_split_wgrad(x.to(dev), tiny, (Cout, Cin, 3, 3), dz_amax=ops.absmax_slots(tiny))
    close(dwh / 3e-6, w.grad, tol=2e-6, what="split wgrad, fp16 parts")
    assert torch.equal(dwh, ops.conv3x3_split_wgrad(x.to(dev), tiny, (Cout, Cin, 3, 3), dz_amax=ops.absmax_slots(tiny)))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 64, 64), (2, 64, 128, 32, 64), (3, 72, 64, 16, 96), (4, 64, 128, 32, 32),
                                             (2, 128, 64, 32, 32), (1, 64, 64, 8, 128), (4, 64, 128, 16, 16), (8, 128, 64, 32, 16)])
def test_conv3x3_split_presplit_kernels(dev, B, Cin, Cout, H, W):
    """Round 4, pre-split operands (slot layout [B][C/8][H][hi|mid][W][8] of 16-bit parts, written by onet_split_pack_act):
    conv3x3_split_pre_kernel (staging = LDS-DMA copy) must reproduce conv3x3_split_kernel on the same parts BIT FOR BIT, for fp16
    and bf16 parts; conv3x3_split_wgrad_pre_kernel (LDS-DMA + ds_read_b64_tr_b16 fragments, another summation order) is held to
    the fp64 weight gradient: 2e-6 of scale with fp16 parts, 3e-5 with bf16 parts; 32-pixel maps pair two images per unit."""
    from onet_amd import ops
    x = rnd(B, Cin, H, W, seed=51).to(dev)
    g = rnd(B, Cout, H, W, seed=52).to(dev)
    w = rnd(Cout, Cin, 3, 3, seed=53, scale=(2.0 / (Cin * 9)) ** 0.5).to(dev)
    if Cin % 16 == 0 and W > 16:
        qf, _ = ops.pack3x3_split(w)
        ref = ops.conv3x3_split(x, qf, Cout)
        got = ops.conv3x3_split_pre(ops.split_pack_act(x, f16=True), qf, Cout)
        assert torch.equal(ref, got), float((ref - got).abs().max())
    if W == 16:
        # maps 16 pixels wide (two images side by side per tile; the fp32-operand split kernel has no such variant): against fp64,
        # forward with the statistics epilogue == the separate statistics pass, and the input-gradient orientation
        qf, qd = ops.pack3x3_split(w)
        zr = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
        nparts = int(_lib_load().onet_conv3x3_split_pre_nparts(B, H, W))
        assert nparts == (B // 2) * (H // 16)
        cm = torch.full((Cout, nparts, 3), float("nan"), device=dev)
        z = ops.conv3x3_split_pre(ops.split_pack_act(x, f16=True), qf, Cout, stats=cm)
        close(z, zr, tol=2e-6, what="pre-split fwd, 16-pixel maps")
        assert torch.isfinite(cm).all() and float(cm[:, :, 0].sum(1).min()) == float(cm[:, :, 0].sum(1).max()) == B * H * W
        gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        s0 = ops.bn_train_coeffs(z, gamma, beta, None, None, 0.1, 1e-5)
        s1 = ops.bn_train_coeffs(z, gamma, beta, None, None, 0.1, 1e-5, cm=(cm, 0, nparts))
        assert float((s0[0] - s1[0]).abs().max()) <= 2e-6 * float(z.std()) and float(((s0[1] - s1[1]) / s0[1]).abs().max()) <= 1e-5
    wz = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wz, None, 1, 1).backward(g.double().cpu())
    for f16, tol in ((True, 2e-6), (False, 3e-5)):
        dw = ops.conv3x3_split_wgrad_pre(ops.split_pack_act(x, f16=f16), ops.split_pack_act(g, f16=f16), (Cout, Cin, 3, 3))
        close(dw, wz.grad, tol=tol, what="pre-split wgrad " + ("fp16" if f16 else "bf16"))


def test_split_f16_range_guard(dev):
    """Round 4 (review item: fp16 range unguarded).  (a) Weights beyond fp16's range times the old fixed 2^8 (|w| = 400 > 255.9):
    the pack now takes its power of two from the tensor's largest magnitude.  (b) An activation beyond 65504 (a loaded checkpoint
    with a large BatchNorm gamma): with the magnitude slots its producer recorded, the kernel scales it into range; without them
    the fp16 parts overflow (shown, so that the guard is known to be what saves the result).  Both against the fp32-MFMA direct
    kernel at 1e-5 of the output scale."""
    from onet_amd import ops
    B, Cin, Cout, H, W = 2, 64, 64, 32, 64
    x = rnd(B, Cin, H, W, seed=61).to(dev)
    w = (rnd(Cout, Cin, 3, 3, seed=62) * 400.0 / 4.0).to(dev)
    assert float(w.abs().max()) > 255.9
    wd, _ = ops.pack3x3(w)
    ref = ops.conv_fwd(x, wd, Cout, 3)
    qf, qd = ops.pack3x3_split(w)
    close(ops.conv3x3_split(x, qf, Cout), ref.double().cpu(), tol=1e-5, what="split fwd, |w| = 400")
    xb = torch.relu(x) * 3e5
    assert float(xb.max()) > 65504
    w1 = rnd(Cout, Cin, 3, 3, seed=63, scale=(2.0 / (Cin * 9)) ** 0.5).to(dev)
    wd1, _ = ops.pack3x3(w1)
    ref = ops.conv_fwd(xb, wd1, Cout, 3)
    q1, _ = ops.pack3x3_split(w1)
    close(ops.conv3x3_split(xb, q1, Cout, amax=ops.absmax_slots(xb)), ref.double().cpu(), tol=1e-5, what="split fwd, |x| = 3e5, guarded")
    assert not torch.isfinite(ops.conv3x3_split(xb, q1, Cout)).all()        # unguarded: fp16(3e5) = inf
    # an ordinary tensor is untouched by the guard: bit-identical with and without its slots
    assert torch.equal(ops.conv3x3_split(x, q1, Cout), ops.conv3x3_split(x, q1, Cout, amax=ops.absmax_slots(x)))


@pytest.mark.parametrize("B,Cin,Cout,H,W,G", [(4, 64, 64, 32, 64, 2), (2, 32, 48, 20, 40, 1), (4, 128, 128, 16, 32, 2), (6, 48, 256, 33, 96, 1),
                                               (8, 64, 128, 16, 16, 2), (2, 16, 64, 48, 128, 2)])
def test_conv3x3_split_norm_on_load_is_bit_identical(dev, B, Cin, Cout, H, W, G):
    """Normalise on load: the split forward kernel and the split weight-gradient kernel given the PRE-activation of the unit
    below plus its BatchNorm coefficients (one or two statistics groups) must produce, bit for bit, what they produce on the
    activation bn_relu_apply materialises -- interior, zero padding (the halo must stay zero, not relu(shift)), ragged tiles,
    the 128-channel weight-gradient tiles, 32- / 16-pixel-wide maps (weight gradient only at 16)."""
    from onet_amd import ops
    zp = rnd(B, Cin, H, W, seed=41).to(dev)
    save = torch.empty(G, 4, Cin)
    g = torch.Generator().manual_seed(42)
    save[:, 0] = torch.randn(G, Cin, generator=g) * 0.3                  # mean
    save[:, 1] = torch.rand(G, Cin, generator=g) + 0.5                   # invstd
    save[:, 2] = save[:, 1] * (torch.rand(G, Cin, generator=g) + 0.5)    # scale = gamma * invstd
    save[:, 3] = torch.randn(G, Cin, generator=g) * 0.5 + 0.2            # shift = beta (positive on average: relu(shift) != 0)
    save = save.to(dev)
    Bg = B // G
    a = torch.cat([ops.bn_relu_apply(zp[i * Bg:(i + 1) * Bg], save[i]) for i in range(G)])
    w = rnd(Cout, Cin, 3, 3, seed=43, scale=(2.0 / (Cin * 9)) ** 0.5).to(dev)
    dz = rnd(B, Cout, H, W, seed=44).to(dev)
    if W > 16:
        qf, _ = ops.pack3x3_split(w)
        ref = ops.conv3x3_split(a, qf, Cout)
        got = ops.conv3x3_split(zp, qf, Cout, norm=save)
        assert torch.equal(got, ref), float((got - ref).abs().max())
    if ops.split_wgrad_ok(a, dz):
        ref = ops.conv3x3_split_wgrad(a, dz, (Cout, Cin, 3, 3))
        got = ops.conv3x3_split_wgrad(zp, dz, (Cout, Cin, 3, 3), norm=save)
        assert torch.equal(got, ref), float((got - ref).abs().max())
    else:
        assert W < 64 and W != 32          # (the weight-gradient kernel takes maps >= 64 or exactly 32 / 16 pixels wide)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 40, 48), (1, 32, 128, 64, 64), (3, 16, 36, 33, 28), (2, 128, 80, 16, 96),
                                             (9, 48, 64, 32, 32), (1, 512, 64, 17, 40)])
def test_conv3x3_split_fwd_dgrad_stats(dev, B, Cin, Cout, H, W, monkeypatch):
    """conv_split.hip -- fp32 convolution on the bf16 matrix cores by operand splitting (x = hi + mid, w = hi + mid, three MFMAs
    per term) -- against the fp64 convolution: forward and input-gradient orientation at 2e-5 of the output scale (measured
    5e-6; the fp32 Winograd F(4x4) kernel it replaces in the default dispatch: 1e-5 .. 4e-5), ragged edges, channel tails,
    several tiles per persistent block (B = 9), long reductions; fused BatchNorm statistics == the separate pass."""
    from onet_amd import _lib, ops
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(2.0 / (Cin * 9)) ** 0.5)
    g = rnd(B, Cout, H, W, seed=3)
    xr = x.double().clone().requires_grad_(True)
    zr = F.conv2d(xr, w.double(), None, 1, 1)
    zr.backward(g.double())
    qf, qd = ops.pack3x3_split(w.to(dev))
    z = ops.conv3x3_split(x.to(dev), qf, Cout)
    # fp16 parts (22 significant bits) in both orientations -- 2e-6 of the output scale (measured 6e-7 .. 1.9e-6 max, 5e-8 .. 1.5e-7
    # rms: the fp32 direct kernel's level; round 4: the input gradient too, on a power-of-two-scaled operand); bf16 parts
    # (16 bits; Settings.split_f16 / grad_f16 = False, round 3's gradients): 2e-5 (measured 5e-6)
    monkeypatch.setattr(ops, "SPLIT_GRAD_F16", True)
    qf, qd = ops.pack3x3_split(w.to(dev))
    assert qf.dtype == torch.float16 and qd.dtype == torch.float16
    close(z, zr, tol=2e-6 if Cin <= 128 else 4e-6, what="split fwd")
    monkeypatch.setattr(ops, "SPLIT_F16", False)
    monkeypatch.setattr(ops, "SPLIT_GRAD_F16", False)
    qb, qdb = ops.pack3x3_split(w.to(dev))
    monkeypatch.setattr(ops, "SPLIT_F16", True)
    assert qb.dtype == torch.bfloat16 and qdb.dtype == torch.bfloat16
    close(ops.conv3x3_split(x.to(dev), qb, Cout), zr, tol=2e-5, what="split fwd, bf16 parts")
    if Cout % 16 == 0:
        close(ops.conv3x3_split(g.to(dev), qdb, Cin), xr.grad, tol=2e-5, what="split dgrad, bf16 parts")
        # the gradient operand at a gradient's magnitude: 1e-7 g underflows fp16 unscaled; scaled from its magnitude slots
        # (computed here, always=True; in the model bn_relu_bwd_apply records them) the error stays at the fp16-parts level
        tiny = (1e-7 * g).to(dev)
        dgt = ops.conv3x3_split(tiny, qd, Cin, always=True)
        close(dgt * 1e7, xr.grad, tol=2e-6 if Cout <= 128 else 4e-6, what="split dgrad, fp16 parts of the scaled operand")
        am = ops.absmax_slots(tiny)
        assert abs(float(am.view(torch.float32).max()) - float(tiny.abs().max())) == 0.0
        assert torch.equal(dgt, ops.conv3x3_split(tiny, qd, Cin, amax=am, always=True))
    assert torch.equal(z, ops.conv3x3_split(x.to(dev), qf, Cout)), "bitwise reproducible"
    nparts = int(_lib.load().onet_conv3x3_split_nparts(B, H, W))
    if W % 32 or H % 16:
        assert nparts == 0
        return
    assert nparts == B * (H // 16) * (W // 32)
    z1 = torch.empty_like(z)
    cm = torch.full((Cout, nparts, 3), float("nan"), device=dev)
    _lib.call("onet_conv3x3_split_fwd_stats", x.to(dev).data_ptr(), Cin * H * W, qf.data_ptr(), 1, z1.data_ptr(), Cout * H * W,
              cm.data_ptr(), B, Cin, Cout, H, W, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(z, z1) and torch.isfinite(cm).all()
    assert float(cm[:, :, 0].sum(1).min()) == float(cm[:, :, 0].sum(1).max()) == B * H * W
    gamma, beta = (1 + 0.1 * rnd(Cout, seed=33)).to(dev), (0.1 * rnd(Cout, seed=34)).to(dev)
    rm0, rv0 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    rm1, rv1 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    s0 = ops.bn_train_coeffs(z, gamma, beta, rm0, rv0, 0.1, 1e-5)
    s1 = ops.bn_train_coeffs(z, gamma, beta, rm1, rv1, 0.1, 1e-5, cm=(cm, 0, nparts))
    sd = float(z.std())
    assert float((s0[0] - s1[0]).abs().max()) <= 2e-6 * sd and float(((s0[1] - s1[1]) / s0[1]).abs().max()) <= 1e-5


def test_conv_on_channel_slices(dev):
    """inputs/outputs that are channel-slices of wider (concat) buffers: batch stride != C*H*W"""
    from onet_amd import ops
    B, Cin, Cout, H, W = 2, 32, 64, 24, 40
    big = rnd(B, Cin + 16, H, W, seed=5)
    w = rnd(Cout, Cin, 3, 3, seed=6, scale=0.1)
    ref = F.conv2d(big[:, 16:], w, None, 1, 1)
    bigd = big.to(dev)
    wf, _ = ops.pack3x3(w.to(dev))
    outbuf = torch.zeros((B, Cout + 8, H, W), device=dev)
    ops.conv_fwd(bigd[:, 16:], wf, Cout, 3, out=outbuf[:, 8:])
    close(outbuf[:, 8:], ref, what="sliced fwd")
    assert float(outbuf[:, :8].abs().max()) == 0.0


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,C,H,W", [(2, 64, 32, 32), (3, 10, 7, 9), (2, 128, 130, 130)])
def test_bn_relu(dev, training, B, C, H, W):
    from onet_amd import functional as Fn
    z = rnd(B, C, H, W, seed=7) * 2 + 0.7
    gamma = 1 + 0.1 * rnd(C, seed=8)
    beta = 0.1 * rnd(C, seed=9)
    rm = 0.1 * rnd(C, seed=10)
    rv = 1 + 0.1 * rnd(C, seed=11).abs()
    g = rnd(B, C, H, W, seed=12)

    zr, gr, br = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    yr = F.relu(F.batch_norm(zr, rmr, rvr, gr, br, training, 0.1, 1e-5))
    yr.backward(g)

    zd, gd_, bd = z.to(dev).requires_grad_(True), gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    y = Fn.BNReLUFn.apply(zd, gd_, bd, rmd, rvd, training, 0.1, 1e-5)
    y.backward(g.to(dev))
    close(y, yr, what="bn fwd")
    close(rmd, rmr, tol=1e-5, what="running_mean")
    close(rvd, rvr, tol=1e-5, what="running_var")
    close(zd.grad, zr.grad, tol=5e-4, what="dz")
    close(gd_.grad, gr.grad, tol=5e-4, what="dgamma")
    close(bd.grad, br.grad, tol=5e-4, what="dbeta")


@pytest.mark.parametrize("B,C,H,W", [(2, 8, 16, 16), (2, 3, 9, 11), (1, 64, 130, 34)])
def test_maxpool(dev, B, C, H, W):
    from onet_amd import functional as Fn
    x = rnd(B, C, H, W, seed=13)
    x[:, :, :4, :4] = 0.0          # ties: first maximum must win
    x = torch.relu(x)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2)
    g = rnd(*yr.shape, seed=14)
    yr.backward(g)
    xd = x.to(dev).requires_grad_(True)
    y = Fn.MaxPool2Fn.apply(xd)
    y.backward(g.to(dev))
    assert torch.equal(y.cpu(), yr.detach())
    assert torch.equal(xd.grad.cpu(), xr.grad)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 8, 64, 32, 32), (3, 16, 72, 16, 64), (2, 4, 12, 48, 32),
                                             (4, 64, 128, 64, 64), (2, 8, 8, 32, 96)])
def test_winograd4_fused_bn_statistics(dev, B, Cin, Cout, H, W):
    """F(4x4) forward with the BatchNorm statistics records written by its epilogue (maps made of full 16 x 32-pixel
    blocks; channel tails): z bit-identical to the plain kernel, and finalize(records) == finalize(separate statistics
    pass) for the whole batch and for a batch slice (the twin batch's statistics groups).  Ragged maps report
    nparts == 0 and keep the separate statistics pass."""
    from onet_amd import _lib, ops
    for h, w in [(40, 40), (16, 16), (20, 64), (32, 48)]:
        assert int(_lib.load().onet_conv3x3_winograd4_nparts(B, h, w)) == 0
    x = rnd(B, Cin, H, W, seed=31) + 0.7
    w = rnd(Cout, Cin, 3, 3, seed=32) / (3.0 * Cin ** 0.5)
    xd = x.to(dev)
    qf, _ = ops.pack3x3_winograd4(w.to(dev))
    z0 = ops.conv3x3_winograd4(xd, qf, Cout)
    nparts = int(_lib.load().onet_conv3x3_winograd4_nparts(B, H, W))
    assert nparts == 2 * B * (W // 32) * (H // 16)          # one record per tile half (16 x 16 pixels) of a 16 x 32-pixel block
    z1 = torch.empty_like(z0)
    cm = torch.full((Cout, nparts, 3), float("nan"), device=dev)
    _lib.call("onet_conv3x3_winograd4_fwd_stats", xd.data_ptr(), Cin * H * W, qf.data_ptr(), z1.data_ptr(),
              Cout * H * W, cm.data_ptr(), B, Cin, Cout, H, W, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(z0, z1)
    assert torch.isfinite(cm).all()
    assert float(cm[:, :, 0].sum(1).min()) == float(cm[:, :, 0].sum(1).max()) == B * H * W
    gamma = (1 + 0.1 * rnd(Cout, seed=33)).to(dev)
    beta = (0.1 * rnd(Cout, seed=34)).to(dev)
    for lo, hi in [(0, B), (B // 2, B)]:
        zs = z0[lo:hi]
        rm0, rv0 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        rm1, rv1 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        s0 = ops.bn_train_coeffs(zs, gamma, beta, rm0, rv0, 0.1, 1e-5)
        npi = nparts // B
        s1 = ops.bn_train_coeffs(zs, gamma, beta, rm1, rv1, 0.1, 1e-5, cm=(cm, lo * npi, (hi - lo) * npi))
        sd = float(zs.std())
        assert float((s0[0] - s1[0]).abs().max()) <= 2e-6 * sd, "mean"
        assert float(((s0[1] - s1[1]) / s0[1]).abs().max()) <= 1e-5, "invstd"
        assert float((rm0 - rm1).abs().max()) <= 1e-6 * sd and float(((rv0 - rv1) / rv0).abs().max()) <= 1e-5
        # and against torch in fp64
        zr = zs.double().cpu()
        assert float((s1[0].cpu().double() - zr.mean((0, 2, 3))).abs().max()) <= 2e-6 * sd
        ir = 1.0 / torch.sqrt(zr.var((0, 2, 3), unbiased=False) + 1e-5)
        assert float(((s1[1].cpu().double() - ir) / ir).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,C,H,W", [(2, 8, 16, 32), (3, 5, 6, 12), (2, 64, 64, 64), (1, 3, 2, 4)])
def test_bn_relu_apply_pool_fused(dev, B, C, H, W):
    """onet_bn_relu_apply_pool: BatchNorm + ReLU and the 2x2 max-pooling of its output in one pass -- activation and pooled tensor
    bit-identical to the two separate kernels, with and without the magnitude slots (which then hold max a); shapes it does not take
    (odd H, W % 4) return False."""
    from onet_amd import ops
    z = (rnd(B, C, H, W, seed=81) * 2 + 0.3).to(dev)
    gamma, beta = (1 + 0.2 * rnd(C, seed=82)).to(dev), (0.2 * rnd(C, seed=83)).to(dev)
    save = ops.bn_train_coeffs(z, gamma, beta, torch.zeros(C, device=dev), torch.ones(C, device=dev), 0.1, 1e-5)
    a_ref = ops.bn_relu_apply(z, save)
    y_ref = ops.maxpool2_fwd(a_ref)
    for slots in (None, ops.new_amax(dev)):
        a = torch.full_like(z, float("nan"))
        y = torch.full((B, C, H // 2, W // 2), float("nan"), device=dev)
        assert ops.bn_relu_apply_pool(z, save, a, y, amax=slots)
        assert torch.equal(a, a_ref) and torch.equal(y, y_ref)
        if slots is not None:
            assert float(torch.tensor(slots.cpu().numpy().view("float32")).max()) == float(a_ref.max())
    zz = z[:, :, : H - 1 if H > 2 else H, :].contiguous() if H > 2 else z[..., : W - 2].contiguous()
    o = torch.empty_like(zz)
    yy = torch.empty((B, C, zz.shape[2] // 2, zz.shape[3] // 2), device=dev)
    assert not ops.bn_relu_apply_pool(zz, save, o, yy)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 1, 64, 32, 64), (3, 3, 64, 48, 128), (2, 1, 64, 256, 256), (4, 2, 40, 16, 64),
                                             (2, 4, 128, 32, 192)])
def test_stem_conv_fused_bn_statistics(dev, B, Cin, Cout, H, W):
    """stem.hip: the first convolution (Cin = n_channels in 1..4) with its BatchNorm statistics records in one streaming pass:
    z against the fp64 convolution at fp32 accuracy (a 9 Cin-term FMA chain), finalize(records) == finalize(separate pass) for
    the whole batch and a batch slice, and against fp64; maps not made of 16 x 64 tiles, Cin > 4 or Cout > 128 report 0."""
    from onet_amd import _lib, ops
    lib = _lib.load()
    for args in [(B, 1, 64, 40, 64), (B, 1, 64, 32, 96), (B, 5, 64, 32, 64), (B, 1, 136, 32, 64)]:
        assert int(lib.onet_conv3x3_stem_nparts(*args)) == 0
    x = rnd(B, Cin, H, W, seed=71) + 0.3
    w = rnd(Cout, Cin, 3, 3, seed=72) / (3.0 * Cin ** 0.5)
    xd, wd = x.to(dev), w.to(dev)
    nparts = int(lib.onet_conv3x3_stem_nparts(B, Cin, Cout, H, W))
    assert nparts == B * (H // 16) * (W // 64)
    z = torch.full((B, Cout, H, W), float("nan"), device=dev)
    cm = torch.full((Cout, nparts, 3), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("onet_conv3x3_stem_fwd_stats", xd.data_ptr(), Cin * H * W, wd.data_ptr(), z.data_ptr(), Cout * H * W, cm.data_ptr(),
              B, Cin, Cout, H, W, 0, 0.0, st)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert float((z.cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    z2 = torch.empty_like(z)                                  # convolution only
    _lib.call("onet_conv3x3_stem_fwd_stats", xd.data_ptr(), Cin * H * W, wd.data_ptr(), z2.data_ptr(), Cout * H * W, None,
              B, Cin, Cout, H, W, 0, 0.0, st)
    assert torch.equal(z, z2)
    # K7 on load (round 5): the twin batch [X ; clip(1 - X + bias, 0, 1)] from X alone == the kernel on the materialised twin batch
    from onet_amd import ops
    bias = 0.125
    xx = torch.cat([xd, ops.complement_clip(xd, bias)], 0).contiguous()
    np2 = int(lib.onet_conv3x3_stem_nparts(2 * B, Cin, Cout, H, W))
    zt, zm = torch.empty((2 * B, Cout, H, W), device=dev), torch.empty((2 * B, Cout, H, W), device=dev)
    cmt, cmm = torch.empty((Cout, np2, 3), device=dev), torch.empty((Cout, np2, 3), device=dev)
    _lib.call("onet_conv3x3_stem_fwd_stats", xd.data_ptr(), Cin * H * W, wd.data_ptr(), zt.data_ptr(), Cout * H * W, cmt.data_ptr(),
              2 * B, Cin, Cout, H, W, B, bias, st)
    _lib.call("onet_conv3x3_stem_fwd_stats", xx.data_ptr(), Cin * H * W, wd.data_ptr(), zm.data_ptr(), Cout * H * W, cmm.data_ptr(),
              2 * B, Cin, Cout, H, W, 0, 0.0, st)
    assert torch.equal(zt, zm) and torch.equal(cmt, cmm)
    assert torch.isfinite(cm).all()
    assert float(cm[:, :, 0].sum(1).min()) == float(cm[:, :, 0].sum(1).max()) == B * H * W
    gamma = (1 + 0.1 * rnd(Cout, seed=33)).to(dev)
    beta = (0.1 * rnd(Cout, seed=34)).to(dev)
    for lo, hi in [(0, B), (B // 2, B)]:
        zs = z[lo:hi]
        rm0, rv0 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        rm1, rv1 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        s0 = ops.bn_train_coeffs(zs, gamma, beta, rm0, rv0, 0.1, 1e-5)
        npi = nparts // B
        s1 = ops.bn_train_coeffs(zs, gamma, beta, rm1, rv1, 0.1, 1e-5, cm=(cm, lo * npi, (hi - lo) * npi))
        sd = float(zs.std())
        assert float((s0[0] - s1[0]).abs().max()) <= 2e-6 * sd, "mean"
        assert float(((s0[1] - s1[1]) / s0[1]).abs().max()) <= 1e-5, "invstd"
        assert float((rm0 - rm1).abs().max()) <= 1e-6 * sd and float(((rv0 - rv1) / rv0).abs().max()) <= 1e-5
        zr = zs.double().cpu()
        assert float((s1[0].cpu().double() - zr.mean((0, 2, 3))).abs().max()) <= 2e-6 * sd
        ir = 1.0 / torch.sqrt(zr.var((0, 2, 3), unbiased=False) + 1e-5)
        assert float(((s1[1].cpu().double() - ir) / ir).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 64, 64), (2, 64, 128, 32, 64), (4, 128, 64, 32, 32), (2, 96, 64, 16, 96),
                                             (4, 64, 128, 16, 16), (8, 128, 64, 32, 16), (2, 256, 64, 48, 32), (1, 64, 64, 8, 128)])
def test_conv3x3_plain_bf16_presplit_kernels(dev, B, Cin, Cout, H, W):
    """BASELINE configs[2]'s bf16 MFMA conv path (conv == "bf16"): the pre-split kernels on ONE part of plain bf16 operands
    (conv3x3_pre16_kernel on v_mfma_f32_16x16x32_bf16 with 32-channel chunks; the 16x16x32 weight-gradient kernel).  Each launch
    must equal the fp64 convolution of the bf16-ROUNDED operands to fp32 summation accuracy (2e-6 of scale) -- forward with its
    BatchNorm statistics epilogue, input-gradient orientation, weight gradient -- the statistics records must finalize to the
    separate statistics pass, and with bf16 STORAGE of z (z16) the stored tensor is the fp32 result rounded once, bit for bit."""
    from onet_amd import ops
    rb = lambda t: t.to(torch.bfloat16).double()
    x, g = rnd(B, Cin, H, W, seed=61), rnd(B, Cout, H, W, seed=62)
    w = rnd(Cout, Cin, 3, 3, seed=63, scale=(2.0 / (Cin * 9)) ** 0.5)
    xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    zr = F.conv2d(xr, wr, None, 1, 1)
    zr.backward(rb(g))
    wf, wd = ops.pack3x3_plain16(w.to(dev))
    xP, gP = ops.split_pack_act(x.to(dev), parts=1), ops.split_pack_act(g.to(dev), parts=1)
    assert xP.dtype == torch.bfloat16 and xP.shape == (B, Cin // 8, H, 1, W, 8)
    nparts = int(_lib_load().onet_conv3x3_split_pre_nparts(B, H, W))
    cm = torch.full((Cout, nparts, 3), float("nan"), device=dev) if nparts > 0 else None
    z = ops.conv3x3_split_pre(xP, wf, Cout, stats=cm)
    close(z, zr.detach(), tol=2e-6, what="plain-bf16 forward")
    if cm is not None:
        assert torch.isfinite(cm).all() and float(cm[:, :, 0].sum(1).min()) == float(cm[:, :, 0].sum(1).max()) == B * H * W
        gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        s0 = ops.bn_train_coeffs(z, gamma, beta, None, None, 0.1, 1e-5)
        s1 = ops.bn_train_coeffs(z, gamma, beta, None, None, 0.1, 1e-5, cm=(cm, 0, nparts))
        assert float((s0[0] - s1[0]).abs().max()) <= 2e-6 * float(z.std()) and float(((s0[1] - s1[1]) / s0[1]).abs().max()) <= 1e-5
        cm2 = torch.full_like(cm, float("nan"))
        z16 = ops.conv3x3_split_pre(xP, wf, Cout, stats=cm2, z16=True)
        assert z16.dtype == torch.bfloat16 and torch.equal(z16, z.to(torch.bfloat16)), "bf16 storage of z: one rounding of the fp32 result"
        assert torch.equal(cm, cm2), "the statistics come from the fp32 accumulators, whatever z is stored as"
    dx = ops.conv3x3_split_pre(gP, wd, Cin)
    close(dx, xr.grad, tol=2e-6, what="plain-bf16 input gradient")
    dw = ops.conv3x3_split_wgrad_pre(xP, gP, (Cout, Cin, 3, 3))
    close(dw, wr.grad, tol=4e-6, what="plain-bf16 weight gradient")


@pytest.mark.parametrize("B,Cd,Ca,H,W,G,mode", [(4, 64, 64, 32, 64, 2, "f16"), (2, 128, 64, 64, 32, 1, "f16"), (6, 64, 64, 16, 64, 3, "f16"),
                                                 (4, 128, 128, 48, 96, 2, "plain"), (8, 128, 128, 16, 16, 2, "plain")])
def test_dgrad_with_bn_backward_reduce_epilogue(dev, B, Cd, Ca, H, W, G, mode):
    """onet_conv3x3_split_dgrad_pre_bnreduce (round 5): the input gradient of a DoubleConv's second convolution with the first unit's
    BatchNorm-backward REDUCE in the epilogue (the accumulators are that unit's da).  da must equal the unfused launch to summation
    order, the records -- (hi, lo) pairs in bn_relu_bwd_reduce's format, per statistics group -- must sum to the fp64 sums of
    dy = da * (relu mask) and dy * xhat formed from the SAME da and z (2e-6 of sum |dy|), and the magnitude slots hold max |da|."""
    import math
    from onet_amd import ops
    dz = (rnd(B, Cd, H, W, seed=71) * 1e-3).to(dev)
    w = rnd(Cd, Ca, 3, 3, seed=72, scale=(2.0 / (9 * Ca)) ** 0.5).to(dev)
    zp = (rnd(B, Ca, H, W, seed=73) * 1.5 + 0.2).to(dev)
    gamma, beta = (1 + 0.1 * rnd(Ca, seed=74)).to(dev), (0.1 * rnd(Ca, seed=75)).to(dev)
    save = torch.empty(G, 4, Ca, device=dev)
    Bg = B // G
    for gi in range(G):
        ops.bn_train_coeffs(zp[gi * Bg:(gi + 1) * Bg], gamma, beta, None, None, 0.1, 1e-5, save=save[gi])
    if mode == "f16":
        _, qd = ops.pack3x3_split(w)
        sl = ops.absmax_slots(dz)
        k = 13 - math.floor(math.log2(float(dz.abs().max())))          # the producers' rule: 2^k dz with max |.| in [2^13, 2^14)
        dzP = ops.split_pack_act(dz, f16=True, scale=2.0 ** k)
        kw = dict(slots=sl, always=True)
    else:
        _, qd = ops.pack3x3_plain16(w)
        dzP = ops.split_pack_act(dz, parts=1)
        kw = {}
    ref = ops.conv3x3_split_pre(dzP, qd, Ca, **kw)
    got = ops.conv3x3_split_dgrad_pre_bnreduce(dzP, qd, Ca, zp, save, want_amax=True, **kw)
    assert got is not None, "shape not taken by the fused kernel"
    da, rec4, am = got
    close(da, ref.double().cpu(), tol=2e-6, what="da vs the unfused launch")
    r = rec4.double().view(G, -1, Ca, 4).sum(1).cpu()
    for gi in range(G):
        s_ = slice(gi * Bg, (gi + 1) * Bg)
        mean, inv, scl, sh = (save[gi, i].view(1, -1, 1, 1) for i in range(4))
        mask = torch.addcmul(sh, zp[s_] - mean, scl) > 0                    # the kernel's fp32 expression fma(z - mean, scale, shift) > 0
        dy = (da[s_].double() * mask).cpu()
        xhat = ((zp[s_].double() - mean.double()) * inv.double()).cpu()
        s1, s2, n1 = dy.sum((0, 2, 3)), (dy * xhat).sum((0, 2, 3)), dy.abs().sum((0, 2, 3))
        assert float(((r[gi, :, 0] + r[gi, :, 1] - s1).abs() / n1).max()) <= 2e-6, "sum dy"
        assert float(((r[gi, :, 2] + r[gi, :, 3] - s2).abs() / n1).max()) <= 2e-6, "sum dy * xhat"
    assert float(torch.tensor(am.cpu().numpy().view("float32")).max()) == float(da.abs().max())


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 32, 64, 4, 32), (2, 64, 64, 32, 32), (3, 32, 72, 20, 64), (2, 128, 256, 64, 64),
                                             (4, 64, 128, 16, 96), (2, 32, 64, 4, 16), (6, 64, 96, 16, 16), (4, 256, 128, 8, 16)])
def test_conv3x3_winograd4_wgrad(dev, B, Cin, Cout, H, W):
    """Winograd F(3x3,4x4) weight gradient (conv_wino4w.hip) against the fp64 weight gradient: single unit, several
    units and split-K plans, channel tails in Cout, borders (zero padding on all four sides)."""
    from onet_amd import ops
    x = rnd(B, Cin, H, W, seed=91)
    g = rnd(B, Cout, H, W, seed=92)
    assert ops.winograd4_wgrad_ok(x.to(dev), g.to(dev))
    dw = ops.conv3x3_winograd4_wgrad(x.to(dev), g.to(dev), (Cout, Cin, 3, 3)).cpu().double()
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), g.double(), stride=1, padding=1)
    sc = float(ref.abs().max())
    err = float((dw - ref).abs().max())
    assert err <= 2e-4 * sc, (err, sc)


@pytest.mark.parametrize("B,Cin,Cmid,Cout,H,W,G", [(4, 8, 64, 32, 32, 32, 1), (4, 16, 72, 64, 32, 64, 2),
                                                    (2, 16, 24, 16, 48, 32, 1)])
def test_double_conv_fused_bn_backward_reduce(dev, B, Cin, Cmid, Cout, H, W, G, monkeypatch):
    """DoubleConv (OV:39-58) under forced F(4x4): unit 2's dgrad launch also writes unit 1's BatchNorm-backward reduce
    records (sum dy, sum dy*xhat per 16 x 32 block).  (a) the records against fp64 sums of the same quantities;
    (b) every gradient of the block against the unfused path (separate reduce kernel), one and two statistics groups."""
    from onet_amd import ops, modules
    monkeypatch.setattr(ops, "CONV_ALGO", "winograd4")
    torch.manual_seed(5)
    m = modules.DoubleConv(Cin, Cout, Cmid).to(dev)
    with torch.no_grad():
        for bn in (m.double_conv[1], m.double_conv[4]):
            bn.weight.copy_(1 + 0.2 * rnd(bn.weight.numel(), seed=41))
            bn.bias.copy_(0.2 * rnd(bn.bias.numel(), seed=42))
    x = rnd(B, Cin, H, W, seed=43).to(dev)
    gy = rnd(B, Cout, H, W, seed=44).to(dev)

    def run(fuse):
        monkeypatch.setattr(ops, "FUSE_BN_REDUCE", fuse)
        m.zero_grad()
        xd = x.clone().requires_grad_(True)
        y = m(xd, groups=G)
        y.backward(gy)
        return [xd.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    seen = []
    real = ops.conv3x3_dgrad_bnreduce
    monkeypatch.setattr(ops, "conv3x3_dgrad_bnreduce", lambda *a: seen.append(real(*a)) or seen[-1])
    fused, plain = run(True), run(False)
    assert seen and seen[0] is not None, "the fused dgrad launch was not taken"
    for i, (a, b) in enumerate(zip(fused, plain)):
        close(a, b, tol=2e-5, what=f"fused vs separate reduce, tensor {i}")
    # (a) records vs fp64
    da, rec = seen[0]
    link_z = {}
    monkeypatch.setattr(ops, "FUSE_BN_REDUCE", False)
    s = m.double_conv
    with torch.no_grad():
        z1 = ops.conv3x3_auto(x, s[0].packed(), 0)
    Bg = B // G
    tot = rec.double().cpu().reshape(Cmid, G, -1, 2).sum(2)              # [C, G, 2]
    for g in range(G):
        zg = z1[g * Bg:(g + 1) * Bg].double().cpu()
        mean = zg.mean((0, 2, 3), keepdim=True)
        inv = 1.0 / torch.sqrt(zg.var((0, 2, 3), unbiased=False, keepdim=True) + s[1].eps)
        xh = (zg - mean) * inv
        act = xh * s[1].weight.double().cpu().view(1, -1, 1, 1) + s[1].bias.double().cpu().view(1, -1, 1, 1)
        dy = da[g * Bg:(g + 1) * Bg].double().cpu() * (act > 0)
        scale = float(dy.abs().sum((0, 2, 3)).max())
        assert float((tot[:, g, 0] - dy.sum((0, 2, 3))).abs().max()) <= 2e-5 * scale
        assert float((tot[:, g, 1] - (dy * xh).sum((0, 2, 3))).abs().max()) <= 2e-5 * scale * float(xh.abs().max())


@pytest.mark.parametrize("B,C,H,W", [(2, 3, 8, 8), (3, 5, 7, 9), (2, 4, 16, 12), (1, 2, 6, 10)])
def test_skip_pool(dev, B, C, H, W):
    """A skip tensor feeds both the pooling (OV:67) and the concat (OV:100): SkipPoolFn sums the two gradients inside
    the pooling backward; the skip gradient arrives as a channel slice of a wider (concat) gradient."""
    from onet_amd import functional as Fn
    x = torch.relu(rnd(B, C, H, W, seed=15))
    x[:, :, :4, :4] = 0.0
    gp = rnd(B, C, H // 2, W // 2, seed=16)
    gcat = rnd(B, 2 * C, H, W, seed=17)
    xr = x.clone().requires_grad_(True)
    (F.max_pool2d(xr, 2) * gp).sum().backward()
    want = xr.grad + gcat[:, :C]
    xd = x.to(dev).requires_grad_(True)
    s, p = Fn.SkipPoolFn.apply(xd)
    assert torch.equal(p.cpu(), F.max_pool2d(x, 2)) and torch.equal(s.cpu(), x)
    torch.autograd.backward([s, p], [gcat.to(dev)[:, :C], gp.to(dev)])
    assert torch.equal(xd.grad.cpu(), want)
    # three consumers (the first encoder output is pooled, concatenated and returned)
    gret = rnd(B, C, H, W, seed=18)
    xd.grad = None
    s, p, r = Fn.SkipPoolFn.apply(xd, True)
    torch.autograd.backward([s, p, r], [gcat.to(dev)[:, :C], gp.to(dev), gret.to(dev)])
    close(xd.grad, want + gret, tol=1e-6, what="skip + pool + returned")
    xd.grad = None
    s, p, r = Fn.SkipPoolFn.apply(xd, True)
    torch.autograd.backward([p, r], [gp.to(dev), gret.to(dev)])
    assert torch.equal(xd.grad.cpu(), xr.grad + gret)
    # only one consumer
    xd.grad = None
    s, p = Fn.SkipPoolFn.apply(xd)
    p.backward(gp.to(dev))
    assert torch.equal(xd.grad.cpu(), xr.grad)


@pytest.mark.parametrize("B,C,H,W,G", [(4, 6, 16, 16, 1), (4, 5, 12, 40, 2), (2, 3, 64, 256, 1)])
def test_pool_backward_with_bn_reduce(dev, B, C, H, W, G):
    """onet_maxpool2_bwd_add_bnreduce: dx identical to the plain pooling backward (+ adds), and the BatchNorm-backward
    records it takes on the way equal to those of the reduce kernel run on that dx (same fp64 sums), per group."""
    from onet_amd import ops
    z = rnd(B, C, H, W, seed=71).to(dev)
    gamma = (1 + 0.3 * rnd(C, seed=72)).to(dev)
    beta = (0.2 * rnd(C, seed=73)).to(dev)
    Bg = B // G
    save = torch.empty(G, 4, C, device=dev)
    a = torch.empty_like(z)
    for g in range(G):
        ops.bn_train_coeffs(z[g * Bg:(g + 1) * Bg], gamma, beta, None, None, 0.1, 1e-5, save=save[g])
        ops.bn_relu_apply(z[g * Bg:(g + 1) * Bg], save[g], out=a[g * Bg:(g + 1) * Bg])
    gp = rnd(B, C, H // 2, W // 2, seed=74).to(dev)
    gs = rnd(B, 2 * C, H, W, seed=75).to(dev)[:, :C]
    gr = rnd(B, C, H, W, seed=76).to(dev)
    want = ops.maxpool2_bwd(a, gp, add=gs, add2=gr)
    dx, part2 = ops.maxpool2_bwd(a, gp, add=gs, add2=gr, bn=(z, save))
    assert part2 is not None and torch.equal(dx, want)
    npg = part2.shape[0] // G
    for g in range(G):
        sl = slice(g * Bg, (g + 1) * Bg)
        _, dg0, db0 = ops.bn_relu_bwd(want[sl], z[sl], save[g], True)
        _, dg1, db1 = ops.bn_relu_bwd(want[sl], z[sl], save[g], True, red4=(part2, g * npg, npg))
        close(dg1, dg0, tol=1e-6, what="dgamma")
        close(db1, db0, tol=1e-6, what="dbeta")


@pytest.mark.parametrize("h,w,Ho,Wo,Cin,Ct", [(8, 8, 16, 16, 64, 32), (12, 12, 25, 25, 64, 32), (5, 7, 11, 16, 64, 32),
                                              (32, 32, 64, 64, 128, 64), (17, 40, 35, 80, 24, 12),
                                              (9, 20, 19, 41, 256, 128), (16, 16, 32, 32, 128, 64),
                                              (16, 16, 32, 32, 256, 128), (8, 32, 16, 64, 384, 96), (64, 64, 128, 128, 128, 32)])
def test_up_convT_cat(dev, h, w, Ho, Wo, Cin, Ct):
    """ConvTranspose2d(k=2, s=2) + F.pad + cat through the fused shuffle-epilogue GEMM (odd pad offsets, a wide map,
    channel counts that do not fill the 64-row GEMM tile) and its backward (Ct % 64 == 0: the gather-fused dgrad / wgrad;
    otherwise space-to-depth + plain 1x1 GEMMs).  Maps of 128 k pixels with Cin % 128 == 0, Ct % 32 == 0 and no F.pad take
    the DMA-fed 128 x 128 GEMMs of convt_gemm.hip (16x16 and wider maps, several K chunks, split-K in the wgrad)."""
    from onet_amd import functional as Fn
    from onet_amd import ops
    B, C2 = 2, Ct
    x1, x2 = rnd(B, Cin, h, w, seed=15), rnd(B, C2, Ho, Wo, seed=16)
    wt, bt = rnd(Cin, Ct, 2, 2, seed=17, scale=0.1), rnd(Ct, seed=18, scale=0.1)
    a, b_, c, d = [t.clone().requires_grad_(True) for t in (x1, x2, wt, bt)]
    u = F.conv_transpose2d(a, c, d, stride=2)
    dy, dx = Ho - u.shape[2], Wo - u.shape[3]
    u = F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    ref = torch.cat([b_, u], 1)
    g = rnd(*ref.shape, seed=19)
    ref.backward(g)
    A, Bt, Cw, Db = [t.to(dev).requires_grad_(True) for t in (x1, x2, wt, bt)]
    out = Fn.UpConvTCatFn.apply(A, Bt, Cw, Db, (ops.packT2x2_fused(Cw), ops.packT2x2(Cw)[1]))
    out.backward(g.to(dev))
    close(out, ref, what="up fwd")
    close(A.grad, a.grad, what="dx1")
    close(Bt.grad, b_.grad, tol=1e-6, what="dx2")
    close(Cw.grad, c.grad, tol=3e-4, what="dW")
    close(Db.grad, d.grad, tol=3e-4, what="dbias")


@pytest.mark.parametrize("h,w,Cin,Ct", [(32, 32, 128, 64), (16, 16, 128, 64), (16, 16, 256, 128), (64, 64, 128, 64), (8, 32, 384, 192)])
def test_up_convT_cat_bf16_operands(dev, h, w, Cin, Ct, monkeypatch):
    """BASELINE configs[2]: under the bf16 conv path the three ConvTranspose2d GEMMs (128 x 128 fast path) round their MFMA
    operands to bf16 and accumulate in fp32 -- each must equal the fp64 result for the bf16-ROUNDED operands to fp32 summation
    accuracy (1e-5 of scale; the bias gradient is summed from the unrounded fp32 rows), must differ from the fp32 kernels'
    result (the bf16 kernels really ran), and the switch must not leak into fp32 calls made afterwards.  (Ct % 64 == 0: the
    shapes whose forward AND backward take the GEMM path.)"""
    from onet_amd import functional as Fn
    from onet_amd import ops
    B, C2, Ho, Wo = 2, Ct, 2 * h, 2 * w
    x1, x2 = rnd(B, Cin, h, w, seed=15), rnd(B, C2, Ho, Wo, seed=16)
    wt, bt = rnd(Cin, Ct, 2, 2, seed=17, scale=0.1), rnd(Ct, seed=18, scale=0.1)
    g = rnd(B, C2 + Ct, Ho, Wo, seed=19)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float64)
    a, c = rb(x1).requires_grad_(True), rb(wt).requires_grad_(True)
    u = F.conv_transpose2d(a, c, bt.double(), stride=2)
    u.backward(rb(g[:, C2:]))
    db_ref = g[:, C2:].double().sum((0, 2, 3))

    def run():
        A, Bt, Cw, Db = [t.to(dev).requires_grad_(True) for t in (x1, x2, wt, bt)]
        out = Fn.UpConvTCatFn.apply(A, Bt, Cw, Db, (ops.packT2x2_fused(Cw), ops.packT2x2(Cw)[1]))
        out.backward(g.to(dev))
        return out.detach()[:, C2:].cpu().double(), A.grad.cpu().double(), Cw.grad.cpu().double(), Db.grad.cpu().double()

    f32 = run()
    monkeypatch.setattr(ops, "CONV_ALGO", "bf16")
    b16 = run()
    monkeypatch.setattr(ops, "CONV_ALGO", "auto")
    again = run()
    for got, ref, what in ((b16[0], u.detach(), "up fwd"), (b16[1], a.grad, "dx1"), (b16[2], c.grad, "dW"), (b16[3], db_ref, "dbias")):
        sc = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= (1e-5 if what != "dbias" else 1e-4) * sc, what
    for i, what in enumerate(("up fwd", "dx1", "dW")):
        assert float((b16[i] - f32[i]).abs().max()) > 1e-4 * float(f32[i].abs().max()), what + ": the bf16 kernels did not run"
        assert torch.equal(again[i], f32[i]), what + ": bf16 switch leaked into an fp32 call"


@pytest.mark.parametrize("h,w,Cin,Ct", [(32, 32, 128, 64), (16, 16, 256, 128), (64, 64, 128, 64), (8, 32, 384, 192), (16, 16, 1024, 512)])
def test_up_convT_cat_split_operands(dev, h, w, Cin, Ct, monkeypatch):
    """Round 3 default of the fp32 model: the three ConvTranspose2d GEMMs (128 x 128 fast path) on the bf16 matrix cores with
    SPLIT operands (hi + mid bf16 parts, three MFMAs per term) -- each must equal the fp64 result of the UNROUNDED operands to
    2e-5 of its scale (measured 1e-6 .. 5e-6; plain bf16 operands: 2e-3), the split kernels must really have run (results differ
    from the fp32-MFMA kernels' in the last bits) and ONET_CONVT_SPLIT=0 / Settings(split=False) must give the fp32-MFMA result."""
    from onet_amd import functional as Fn
    from onet_amd import ops
    B, C2, Ho, Wo = 2, Ct, 2 * h, 2 * w
    x1, x2 = rnd(B, Cin, h, w, seed=15), rnd(B, C2, Ho, Wo, seed=16)
    wt, bt = rnd(Cin, Ct, 2, 2, seed=17, scale=0.1), rnd(Ct, seed=18, scale=0.1)
    g = rnd(B, C2 + Ct, Ho, Wo, seed=19)
    a, c = x1.double().requires_grad_(True), wt.double().requires_grad_(True)
    u = F.conv_transpose2d(a, c, bt.double(), stride=2)
    u.backward(g[:, C2:].double())

    def run(settings):
        with ops.using(settings):
            A, Bt, Cw, Db = [t.to(dev).requires_grad_(True) for t in (x1, x2, wt, bt)]
            out = Fn.UpConvTCatFn.apply(A, Bt, Cw, Db, (ops.packT2x2_fused(Cw), ops.packT2x2(Cw)[1]))
            out.backward(g.to(dev))
        return out.detach()[:, C2:].cpu().double(), A.grad.cpu().double(), Cw.grad.cpu().double()

    monkeypatch.setattr(ops, "CONVT_SPLIT_MIN_BLOCKS", 0)       # (the dispatch keeps the fp32 kernels on problems this small)
    sp = run(ops.Settings(conv="auto", split=True))
    f32 = run(ops.Settings(conv="auto", split=False))
    monkeypatch.setattr(ops, "CONVT_SPLIT", False)
    off = run(ops.Settings(conv="auto", split=True))
    for got, ref, what in ((sp[0], u.detach(), "up fwd"), (sp[1], a.grad, "dx1"), (sp[2], c.grad, "dW")):
        sc = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-5 * sc, what
    for i, what in enumerate(("up fwd", "dx1", "dW")):
        assert not torch.equal(sp[i], f32[i]), what + ": the split kernels did not run"
        assert torch.equal(off[i], f32[i]), what + ": ONET_CONVT_SPLIT=0 did not restore the fp32 kernels"


@pytest.mark.parametrize("h,w,Ho,Wo", [(8, 8, 16, 16), (12, 12, 25, 25), (1, 1, 2, 2)])
def test_up_bilinear_cat(dev, h, w, Ho, Wo):
    from onet_amd import functional as Fn
    B, C1, C2 = 2, 16, 8
    x1, x2 = rnd(B, C1, h, w, seed=20), rnd(B, C2, Ho, Wo, seed=21)
    a, b_ = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    u = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = Ho - u.shape[2], Wo - u.shape[3]
    ref = torch.cat([b_, F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])], 1)
    g = rnd(*ref.shape, seed=22)
    ref.backward(g)
    A, Bt = x1.to(dev).requires_grad_(True), x2.to(dev).requires_grad_(True)
    out = Fn.UpBilinearCatFn.apply(A, Bt)
    out.backward(g.to(dev))
    close(out, ref, tol=1e-5, what="bilinear fwd")
    close(A.grad, a.grad, tol=1e-5, what="bilinear dx1")
    close(Bt.grad, b_.grad, tol=1e-6, what="bilinear dx2")


def test_head_softmax(dev):
    from onet_amd import functional as Fn
    B, C, H, W = 2, 64, 20, 24
    ts = [rnd(B, C, H, W, seed=30 + i).abs() * 0.5 for i in range(4)]
    r = [t.clone().requires_grad_(True) for t in ts]
    Vt = (r[0] * r[1]).sum(1, keepdim=True)
    Vd = (r[2] * r[3]).sum(1, keepdim=True)
    S = torch.softmax(torch.cat([Vt, Vd], 1), 1)
    gV, gS = rnd(B, 1, H, W, seed=40), rnd(B, 2, H, W, seed=41)
    (Vt * gV).sum().add((Vd * gV * 0.5).sum()).add((S * gS).sum()).backward()
    d = [t.to(dev).requires_grad_(True) for t in ts]
    vt, vd, s = Fn.HeadSoftmaxFn.apply(*d)
    ((vt * gV.to(dev)).sum() + (vd * gV.to(dev) * 0.5).sum() + (s * gS.to(dev)).sum()).backward()
    close(vt, Vt, what="Vt")
    close(vd, Vd, what="Vd")
    close(s, S, tol=1e-5, what="S")
    for i in range(4):
        close(d[i].grad, r[i].grad, what=f"head grad {i}")


def test_log1pexp_table_and_inplace(dev):
    from onet_amd import Onet
    g = np.load(os.path.join(G, "log1pexp.npz"))
    m = Onet(1)
    x = torch.tensor(g["x"], device=dev, requires_grad=True)
    xin = x * 1.0
    y = m.log1pexp(xin)
    assert y.data_ptr() == xin.data_ptr()          # mutates and returns its argument (OV:237-251)
    y.sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["dy"], rtol=2e-5, atol=1e-30)


def test_jsd_loss_extreme_golden(dev):
    """compute_loss + all four gradients against the REAL reference on inputs that hit every
    log1pexp branch (x<=-37 -> ln2, softplus, x+exp(-x), identity)."""
    from onet_amd import Onet
    g = np.load(os.path.join(G, "loss_extreme.npz"))
    m = Onet(1)
    Lt, Ld, Vt, Vd = [torch.tensor(g[k], device=dev, requires_grad=True) for k in ("Lt", "Ld", "Vt", "Vd")]
    S = m.softmax(torch.cat([Vt, Vd], 1))
    loss = m.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for t, k in ((Lt, "dLt"), (Ld, "dLd"), (Vt, "dVt"), (Vd, "dVd")):
        close(t.grad, torch.tensor(g[k]), tol=1e-4, what=k)


def test_argmax_ties(dev):
    from onet_amd import Onet
    S = torch.tensor([[[[0.5, 0.2, 0.8]], [[0.5, 0.8, 0.2]]]], device=dev)
    Y = Onet(1).predict_label(S)
    assert Y.dtype == torch.int64 and Y.tolist() == [[[0, 1, 0]]]


def test_adam_matches_torch(dev):
    from onet_amd import ops
    n = 10007
    p0, g0 = rnd(n, seed=50), rnd(n, seed=51)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    p = torch.zeros(n + 1, device=dev)[:n]
    p.copy_(p0)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        gk = g0 * step
        pr.grad = gk.clone()
        opt.step()
        ops.adam_step(p, gk.to(dev), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
    close(p, pr, tol=1e-6, what="adam params")


def test_complement_clip(dev):
    from onet_amd import ops
    x = rnd(2, 1, 16, 16, seed=60) * 0.8 + 0.5
    close(ops.complement_clip(x.to(dev), 0.1), torch.clip(1 - x + 0.1, 0, 1), tol=1e-7, what="clip")


def test_cpu_tensor_is_refused():
    from onet_amd import Onet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Onet(1)(torch.zeros(2, 1, 16, 16))


def test_conv_kernels_random_shapes_fuzz(dev):
    """40 random layer shapes (ragged maps down to 1x1, odd batches, channel counts off the 32/64 tile grid) through all
    four 3x3 kernels -- direct, F(2x2,3x3), F(4x4,3x3), and both weight-gradient kernels -- against the CPU convolution."""
    from onet_amd import ops
    rng = np.random.default_rng(20240606)
    for it in range(40):
        B = int(rng.integers(1, 6))
        Cin = 4 * int(rng.integers(1, 41))
        Cout = 4 * int(rng.integers(1, 41))
        H = int(rng.integers(1, 71))
        W = int(rng.integers(1, 71))
        x = rnd(B, Cin, H, W, seed=1000 + it)
        w = rnd(Cout, Cin, 3, 3, seed=2000 + it, scale=(2.0 / (Cin * 9)) ** 0.5)
        g = rnd(B, Cout, H, W, seed=3000 + it)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        zr = F.conv2d(xr, wr, None, 1, 1)
        zr.backward(g)
        xd, wd_, gd = x.to(dev), w.to(dev), g.to(dev)
        tag = f"#{it} B{B} {Cin}->{Cout} {H}x{W}"
        pf, pd = ops.pack3x3(wd_)
        close(ops.conv_fwd(xd, pf, Cout, 3), zr, what="direct fwd " + tag)
        close(ops.conv_fwd(gd, pd, Cin, 3), xr.grad, what="direct dgrad " + tag)
        q4f, q4d = ops.pack3x3_winograd4(wd_)
        close(ops.conv3x3_winograd4(xd, q4f, Cout), zr, what="F(4x4) fwd " + tag)
        close(ops.conv3x3_winograd4(gd, q4d, Cin), xr.grad, what="F(4x4) dgrad " + tag)
        close(ops.conv_wgrad(xd, gd, (Cout, Cin, 3, 3), 3), wr.grad, tol=3e-4, what="direct wgrad " + tag)


def test_gpu_clutter_generator_vs_reference_fixture(dev):
    """csrc/clutter.hip against statistics of frames made by the REFERENCE's own generator functions (KD:469-526 /
    RG:177-216 run by tests/golden/make_golden.py::run_clutter_stats after np.random.seed(1981): tests/golden/clutter_stats.npz):
    texture moments and ACF at lags 1..60 along both axes (the Hermite-polynomial ACF mapping of KD:121-164), speckle power /
    PSD slope / real-to-imaginary power, K-amplitude moments E[a^2], E[a^4]/E[a^2]^2 and intensity ACF, and the frames with
    targets: label area fraction, signal-to-clutter ratio as RG:277-294 computes it, how much of the label area the compositing
    rule `template > background` raises.  Random streams differ (Philox vs MT19937), so every statistic's mean over the frames
    must agree within 4.5 standard errors of the difference (+ 2 %); the same estimators (tests/clutter_stats.py) on both
    sides.  Also: frames are bit-reproducible from (seed, frame) and differ across seeds."""
    import clutter_stats as cs
    from onet_amd import data
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clutter_stats.npz"))
    ref = {k: g[k] for k in g.files}
    B, n = 24, data.FRAME
    parts = data.make_clutter_batch_gpu(B, n, n, seed=7, device=dev, with_labels=True, normalise=False, parts=True,
                                        snr_choices=(0, 1, 2))
    rows = []
    for i in range(B):
        tau, s, amp = (parts[k][i].cpu().numpy() for k in ("texture", "speckle", "amplitude"))
        r = {}
        r.update(cs.texture_stats(tau))
        r.update(cs.speckle_stats(s))
        r.update(cs.amplitude_stats(amp))
        r.update(cs.frame_stats(parts["frames"][i, 0].cpu().numpy(), parts["labels"][i].cpu().numpy(), amp))
        rows.append(r)
    got = cs.collect(rows)
    bad = cs.compare(got, ref, "GPU generator", skip=("poly_coeffs", "root_rule_max_abs_err", "quadrature_coeffs", "snr"))
    assert not bad, "\n".join(bad)
    # the whole-frame min-max normalisation and the crop: [0, 1], reproducible, seed-dependent, labels binary
    H = W = 256
    X, lab = data.make_clutter_batch_gpu(6, H, W, seed=7, device=dev, with_labels=True)
    X2 = data.make_clutter_batch_gpu(6, H, W, seed=7, device=dev)
    assert torch.equal(X, X2)                                    # counter-based generator + fixed-order sums: bit-reproducible
    assert X.shape == (6, 1, H, W) and float(X.min()) == 0.0 and abs(float(X.max()) - 1.0) < 1e-6
    assert not torch.equal(X, data.make_clutter_batch_gpu(6, H, W, seed=8, device=dev))
    assert set(lab.unique().tolist()) <= {0.0, 1.0}
    c0 = (n - H) // 2                                            # the crop is the centre of the 400 x 400 frame (RG:302)
    full = data.make_clutter_batch_gpu(6, n, n, seed=7, device=dev, normalise=False)
    crop = data.make_clutter_batch_gpu(6, H, W, seed=7, device=dev, normalise=False)
    assert torch.equal(full[:, :, c0:c0 + H, c0:c0 + W], crop)
    # targets raise the amplitude inside their labels
    assert float(X[:, 0][lab > 0].mean()) > float(X[:, 0][lab == 0].mean())



@pytest.mark.parametrize("B,C,H,W", [(2, 64, 32, 64), (3, 16, 16, 32), (2, 128, 64, 32)])
def test_presplit_producers_are_bit_identical_to_the_fp32_passes(dev, B, C, H, W):
    """Round 4, pre-split storage: the BatchNorm passes that write their result as fp16 (hi | mid) slots for the split convolution
    kernels must produce exactly onet_split_pack_act of what the fp32 passes write -- forward apply (whole tensor / leading groups of
    a concat buffer / + fp32 copy), apply + 2 x 2 pooling (slots or fp32 for either output), backward apply (scaled by the power of
    two its magnitude slots select; the slots hold an upper bound of |dz| within a small factor of the true maximum)."""
    from onet_amd import ops
    g = torch.Generator().manual_seed(7)
    z = (torch.randn(B, C, H, W, generator=g) * 2 + 0.3).to(dev)
    save = torch.empty(4, C)
    save[0] = torch.randn(C, generator=g) * 0.3
    save[1] = torch.rand(C, generator=g) + 0.5
    save[2] = save[1] * (torch.rand(C, generator=g) + 0.5)
    save[3] = torch.randn(C, generator=g) * 0.5 + 0.2
    save = save.to(dev)
    a = ops.bn_relu_apply(z, save)
    aP = ops.split_pack_act(a, f16=True)
    xs, a2 = ops.p16_empty(B, C, H, W, dev), torch.empty_like(a)
    ops.bn_relu_apply_split(z, save, xs, a=a2)
    assert torch.equal(xs, aP) and torch.equal(a2, a)
    wide = torch.zeros((B, 2 * C // 8, H, 2, W, 8), dtype=torch.float16, device=dev)
    ops.bn_relu_apply_split(z, save, wide[:, :C // 8])
    assert torch.equal(wide[:, :C // 8], aP) and float(wide[:, C // 8:].float().abs().max()) == 0.0
    y = ops.maxpool2_fwd(a)
    xs2, ys, a3, yf = ops.p16_empty(B, C, H, W, dev), ops.p16_empty(B, C, H // 2, W // 2, dev), torch.empty_like(a), torch.empty_like(y)
    assert ops.bn_relu_apply_pool_split(z, save, xs2, a3, ys, None)
    assert torch.equal(xs2, aP) and torch.equal(a3, a) and torch.equal(ys, ops.split_pack_act(y, f16=True))
    assert ops.bn_relu_apply_pool_split(z, save, None, a3, None, yf) and torch.equal(yf, y)
    da = (torch.randn(B, C, H, W, generator=g) * 1e-4).to(dev)
    dz, dg, db = ops.bn_relu_bwd(da, z, save, True)
    dzP, slots, dg2, db2 = ops.bn_relu_bwd_split(da, z, save.view(1, 4, C), True)
    bound, amax = float(slots.view(torch.float32).max()), float(dz.abs().max())
    assert amax <= bound <= 8 * amax, (bound, amax)
    k = 13 - int(np.floor(np.log2(bound)))
    assert torch.equal(dzP, ops.split_pack_act(dz, f16=True, scale=2.0 ** k)) and torch.equal(dg2, dg) and torch.equal(db2, db)
    # two statistics groups share the slots and the scale
    if B % 2 == 0:
        save2 = torch.stack([save, save * 1.1])
        dzP2, slots2, _, _ = ops.bn_relu_bwd_split(da, z, save2, True)
        ref = torch.cat([ops.bn_relu_bwd(da[:B // 2], z[:B // 2], save2[0], True)[0], ops.bn_relu_bwd(da[B // 2:], z[B // 2:], save2[1], True)[0]])
        k2 = 13 - int(np.floor(np.log2(float(slots2.view(torch.float32).max()))))
        assert torch.equal(dzP2, ops.split_pack_act(ref, f16=True, scale=2.0 ** k2))


@pytest.mark.parametrize("B,Cin,h,w", [(2, 128, 16, 32), (4, 256, 16, 16), (2, 1024, 16, 16)])
def test_convT_presplit_epilogue(dev, B, Cin, h, w):
    """ConvTranspose2d forward writing the up-sampled channel groups of a pre-split concat buffer (whole slots per lane after a
    v_permlane32_swap exchange in the GEMM's epilogue): bit-identical to splitting the fp32 output; the skip groups stay untouched."""
    from onet_amd import ops
    Ct = Cin // 2
    x, wt, bias = rnd(B, Cin, h, w, seed=71).to(dev), rnd(Cin, Ct, 2, 2, seed=72, scale=0.05).to(dev), rnd(Ct, seed=73).to(dev)
    wf = ops.packT2x2_fused(wt)
    ref = torch.empty(B, Ct, 2 * h, 2 * w, device=dev)
    ops.convT2x2_fwd(x, wf, bias, ref, Ct, 0, 0)
    cat = torch.zeros((B, 2 * Ct // 8, 2 * h, 2, 2 * w, 8), dtype=torch.float16, device=dev)
    assert ops.convT2x2_fwd_p(x, wf, bias, cat[:, Ct // 8:], Ct, 0, 0)
    assert torch.equal(cat[:, Ct // 8:], ops.split_pack_act(ref, f16=True))
    assert float(cat[:, :Ct // 8].float().abs().max()) == 0.0


@pytest.mark.parametrize("B,Cin,h,w,parts", [(2, 128, 16, 32, 2), (4, 256, 16, 16, 2), (2, 1024, 16, 16, 2), (1, 128, 64, 64, 2),
                                             (2, 128, 16, 32, 1), (3, 512, 32, 8, 1)])
def test_convT_slot_operands(dev, B, Cin, h, w, parts):
    """Round 5: the ConvTranspose2d forward with BOTH operands in slot form (x pre-split by its producer, the weights packed by
    onet_convT2x2_pack_weights_slots; every MFMA fragment one 16-byte LDS read of a DMA-copied slot).  fp16 (hi | mid) parts: the
    pre-split output must equal the fp64 ConvTranspose2d of the fp32 operands to 2e-6 of scale (22-bit operands, fp32 accumulation --
    the fp32-operand GEMM with bf16 parts holds 1e-5), also with an input written under a guard scale and an output that needs one;
    plain bf16: the fp64 result of the bf16-rounded operands, rounded once to bf16.  The skip groups of the concat buffer stay untouched."""
    from onet_amd import ops
    Ct = Cin // 2
    x = torch.relu(rnd(B, Cin, h, w, seed=81))
    wt, bias = rnd(Cin, Ct, 2, 2, seed=82, scale=0.05), rnd(Ct, seed=83)
    rb = (lambda t: t.to(torch.bfloat16).double()) if parts == 1 else (lambda t: t.double())

    def nchw(P):
        v = P.float().sum(3) if P.shape[3] == 2 else P[:, :, :, 0].float()
        return v.permute(0, 1, 4, 2, 3).reshape(P.shape[0], P.shape[1] * 8, P.shape[2], P.shape[4])

    for gain in ((1.0, 3.0e4) if parts == 2 else (1.0,)):            # 3e4: max |x| and max |y| beyond fp16's range -> guard scales
        xg = x * gain
        ref = F.conv_transpose2d(rb(xg), rb(wt), bias.double(), stride=2)
        xd = xg.to(dev)
        x_slots = ops.absmax_slots(xd) if parts == 2 else None
        xP = ops.split_pack_act(xd, f16=True, parts=parts, slots=x_slots)
        wP = ops.packT2x2_slots(wt.to(dev), parts)
        y_slots = ops.convT2x2_out_bound(wt.to(dev), bias.to(dev), x_slots) if parts == 2 else None
        cat = torch.zeros((B, 2 * Ct // 8, 2 * h, parts, 2 * w, 8), dtype=xP.dtype, device=dev)
        assert ops.convT2x2_fwd_slots(xP, wP, bias.to(dev), cat[:, Ct // 8:], Ct, x_slots=x_slots, slots=y_slots)
        got = nchw(cat[:, Ct // 8:]).double().cpu()
        if parts == 2 and gain > 1:
            k = 13 - int(np.floor(np.log2(float(torch.tensor(y_slots.cpu().numpy().view("float32")).max()))))
            assert k < 0, "this case is meant to need the output guard scale"
            got = got * 2.0 ** -k
        if parts == 2:
            close(got, ref, tol=2e-6, what=f"slot-operand ConvTranspose2d, gain {gain}")
        else:
            assert torch.equal(got.float(), ref.float().to(torch.bfloat16).float()) or \
                float((got - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()), "plain bf16: one output rounding"
            close(got, ref, tol=2.0 ** -8, what="slot-operand ConvTranspose2d, plain bf16")
        assert float(cat[:, :Ct // 8].float().abs().max()) == 0.0
    # shapes outside the fast path are refused, nothing written
    if parts == 2:
        xs = ops.split_pack_act(torch.relu(rnd(1, 64, 8, 8, seed=84)).to(dev), f16=True)
        out = torch.zeros((1, 4, 16, 2, 16, 8), dtype=torch.float16, device=dev)
        assert not ops.convT2x2_fwd_slots(xs, ops.packT2x2_slots(rnd(64, 32, 2, 2, seed=85).to(dev)), None, out, 32)
        assert float(out.float().abs().max()) == 0.0


@pytest.mark.parametrize("B,Cin,h,w,parts", [(2, 128, 16, 32, 2), (4, 256, 16, 16, 2), (2, 1024, 16, 16, 2), (1, 128, 64, 64, 2), (3, 512, 8, 16, 2),
                                             (2, 128, 16, 32, 1), (4, 256, 16, 16, 1), (2, 512, 32, 32, 1)])
def test_convT_backward_slot_operands(dev, B, Cin, h, w, parts):
    """Round 5: the ConvTranspose2d input-gradient and weight-gradient GEMMs on slot operands (fp16 hi | mid parts): dy pre-split under
    the `always` rule of a BOUND (here 20 x its maximum, as the bound of the 3x3 input gradient that writes it is that loose), x
    pre-split by its producer, the weights in K-slot form; transposed LDS reads for the weight gradient, the bias gradient as a row of
    ones in the same launch.  Each result against fp64 on the fp32 operands: 4e-6 of scale (the fp32-operand GEMMs with bf16 parts hold
    1e-5); both block-tile configurations of the weight gradient occur (Cin % 256 == 0 or not).  parts = 1 (BASELINE configs[2]): one
    part of plain bf16, unscaled -- against fp64 on the bf16-ROUNDED operands, to fp32 summation accuracy."""
    from onet_amd import ops
    Ct = Cin // 2
    rb = (lambda t: t.to(torch.bfloat16).double()) if parts == 1 else (lambda t: t.double())
    x = torch.relu(rnd(B, Cin, h, w, seed=91))
    wt = rnd(Cin, Ct, 2, 2, seed=92, scale=0.05)
    dy = rnd(B, Ct, 2 * h, 2 * w, seed=93, scale=3e-5)
    xr, wr = rb(x).requires_grad_(True), rb(wt).requires_grad_(True)
    (F.conv_transpose2d(xr, wr, None, stride=2) * rb(dy)).sum().backward()
    db_ref = rb(dy).sum((0, 2, 3))
    dyd = dy.to(dev)
    if parts == 2:
        bound = ops.new_amax(dev)
        bound.view(torch.float32)[::32] = 20.0 * float(dy.abs().max())
        k = 13 - int(np.floor(np.log2(20.0 * float(dy.abs().max()))))
        dyP = ops.split_pack_act(dyd, f16=True, scale=2.0 ** k)
    else:
        bound, dyP = None, ops.split_pack_act(dyd, parts=1)
    xP = ops.split_pack_act(x.to(dev), f16=True, parts=parts)
    dx = ops.convT2x2_dgrad_slots(dyP, ops.packT2x2_slots(wt.to(dev), parts, dgrad=True)[1], Cin, dy_slots=bound)
    assert dx is not None
    close(dx, xr.grad, tol=4e-6, what="slot-operand ConvTranspose2d input gradient")
    got = ops.convT2x2_wgrad_slots(xP, dyP, (Cin, Ct, 2, 2), dy_slots=bound, want_dbias=True)
    assert got is not None
    close(got[0], wr.grad, tol=4e-6, what="slot-operand ConvTranspose2d weight gradient")
    assert float((got[1].double().cpu() - db_ref).abs().max()) <= 4e-6 * float(rb(dy).abs().sum((0, 2, 3)).max()), "bias gradient"


@pytest.mark.parametrize("B,Cd,Ca,H,W,parts", [(4, 64, 128, 32, 64, 2), (2, 128, 256, 16, 32, 2), (2, 64, 128, 48, 96, 2),
                                               (4, 64, 128, 32, 64, 1), (2, 128, 256, 16, 32, 1), (2, 64, 128, 48, 96, 1)])
def test_conv3x3_dgrad_with_presplit_upper_half(dev, B, Cd, Ca, H, W, parts):
    """onet_conv3x3_split_dgrad_pre_slots (round 5): the input gradient of a decoder block's first convolution writes the channels of
    the up-sampled half pre-split for the ConvTranspose2d backward GEMMs -- the skip half must equal the plain launch bit for bit, the
    slots must equal the split of the plain launch's fp32 values under the scale of the bound (onet_conv3x3_dgrad_bound), and the bound
    must really bound.  parts = 1: plain bf16 operands (conv3x3_pre16_kernel; an 8 x 8 transposition over eight lanes in its epilogue):
    the slots are the bf16 roundings of the plain launch's fp32 values."""
    import math
    from onet_amd import ops
    dz = (rnd(B, Cd, H, W, seed=95) * 1e-3).to(dev)
    w3 = rnd(Cd, Ca, 3, 3, seed=96, scale=(2.0 / (9 * Ca)) ** 0.5).to(dev)
    if parts == 1:
        _, qd = ops.pack3x3_plain16(w3)
        dzP = ops.split_pack_act(dz, parts=1)
        ref = ops.conv3x3_split_pre(dzP, qd, Ca)
        da, daP = ops.conv3x3_split_dgrad_pre_slots(dzP, qd, Ca, Ca // 2, None)
        assert torch.equal(da[:, :Ca // 2], ref[:, :Ca // 2])
        assert torch.equal(daP, ops.split_pack_act(ref[:, Ca // 2:].contiguous(), parts=1))
        return
    _, qd = ops.pack3x3_split(w3)
    sl = ops.absmax_slots(dz)
    k = 13 - math.floor(math.log2(float(dz.abs().max())))
    dzP = ops.split_pack_act(dz, f16=True, scale=2.0 ** k)
    ref = ops.conv3x3_split_pre(dzP, qd, Ca, slots=sl, always=True)
    bound = ops.conv3x3_dgrad_bound(w3, sl, Ca // 2)
    bv = float(torch.tensor(bound.cpu().numpy().view("float32")).max())
    assert float(ref[:, Ca // 2:].abs().max()) <= bv <= 200 * float(ref[:, Ca // 2:].abs().max())
    da, daP = ops.conv3x3_split_dgrad_pre_slots(dzP, qd, Ca, Ca // 2, bound, slots=sl, always=True)
    assert torch.equal(da[:, :Ca // 2], ref[:, :Ca // 2])
    kb = 13 - math.floor(math.log2(bv))
    assert torch.equal(daP, ops.split_pack_act(ref[:, Ca // 2:].contiguous(), f16=True, scale=2.0 ** kb))


@pytest.mark.parametrize("Cout,Cin", [(64, 64), (128, 64), (72, 48), (512, 256), (40, 16)])
def test_split_weight_packs_bit_for_bit(dev, Cout, Cin, monkeypatch):
    """The weight packs of conv_split.hip ([K/16][part][tap][half][N][8], written by pack3x3_split_kernel) against their definition in
    torch: forward (K = Cin) and input gradient (K = Cout rounded up to 16, taps rotated) orientation; fp16 (hi, mid) parts of
    2^k w with the stored scale putting max |w| into [2^13, 2^14), bf16 (hi, mid) parts, plain bf16; channel counts that do not fill
    the 64-channel tiles."""
    from onet_amd import ops
    w = rnd(Cout, Cin, 3, 3, seed=5, scale=0.07)
    w[3, 5, 1, 1] = 3.1                                   # the maximum the fp16 scale is chosen from

    def layout(v):                                        # v [K][N][9] -> [K/16][tap][half][N][8]
        K, N = v.shape[0], v.shape[1]
        return v.reshape(K // 16, 2, 8, N, 9).permute(0, 4, 1, 3, 2).contiguous()

    def operand(which):
        if which == 0:
            return w.reshape(Cout, Cin, 9).permute(1, 0, 2)
        K = -(-Cout // 16) * 16
        v = torch.zeros(K, Cin, 9)
        v[:Cout] = w.reshape(Cout, Cin, 9).flip(-1)
        return v

    def check(pack, which, mode):
        v = layout(operand(which))
        n = v.numel()
        if mode == 2:
            assert torch.equal(pack[:n].cpu().view(v.shape), v.to(torch.bfloat16))
            return
        got = pack[:2 * n].cpu().view(v.shape[0], 2, *v.shape[1:])
        if mode == 1:
            scale, inv = [float(t) for t in pack[2 * n:2 * n + 4].view(torch.float32).cpu()]
            assert scale * inv == 1.0 and 2.0 ** 13 <= 3.1 * scale < 2.0 ** 14
            vs = v * scale
            hi = vs.to(torch.float16)
            mid = (vs - hi.float()).to(torch.float16)
        else:
            hi = v.to(torch.bfloat16)
            mid = (v - hi.float()).to(torch.bfloat16)
        assert torch.equal(got[:, 0], hi) and torch.equal(got[:, 1], mid)

    monkeypatch.setattr(ops, "SPLIT_GRAD_F16", True)
    qf, qd = ops.pack3x3_split(w.to(dev))
    if Cin % 16 == 0:
        check(qf, 0, 1)
    check(qd, 1, 1)
    monkeypatch.setattr(ops, "SPLIT_F16", False)
    monkeypatch.setattr(ops, "SPLIT_GRAD_F16", False)
    monkeypatch.setattr(ops, "PRESPLIT", False)
    qf, qd = ops.pack3x3_split(w.to(dev))
    assert qd.dtype == torch.bfloat16
    if Cin % 16 == 0:
        check(qf, 0, 0)
    check(qd, 1, 0)
    if Cin % 32 == 0 and Cout % 32 == 0:
        pf, pd = ops.pack3x3_plain16(w.to(dev))
        check(pf, 0, 2)
        check(pd, 1, 2)


@pytest.mark.parametrize("B,Cin,H,W,G", [(4, 1, 64, 64, 2), (3, 3, 40, 56, 1), (2, 4, 32, 32, 2)])
def test_stem_wgrad_with_bn_backward_on_load(dev, B, Cin, H, W, G):
    """conv3x3_stem_wgrad_kernel<.., true>: the stem's weight gradient formed from (da, z) and the BatchNorm-backward coefficients
    without materialising dz -- bit for bit the weight gradient (and dgamma / dbeta) of the apply pass followed by the stem kernel,
    one and two statistics groups, train and eval coefficients."""
    from onet_amd import ops
    Cout = 24
    x, z, da = rnd(B, Cin, H, W, seed=11).to(dev), rnd(B, Cout, H, W, seed=12).to(dev), rnd(B, Cout, H, W, seed=13, scale=1e-3).to(dev)
    gamma, beta = (1.0 + 0.1 * rnd(Cout, seed=14)).to(dev), (0.1 * rnd(Cout, seed=15)).to(dev)
    save = torch.empty(G, 4, Cout, device=dev)
    for g in range(G):
        ops.bn_train_coeffs(z[g * (B // G):(g + 1) * (B // G)], gamma, beta, None, None, 0.1, 1e-5, save=save[g])
    for training in (True, False):
        dz, dg0, db0 = ops.bn_relu_bwd_groups(da, z, save, training)
        dw0 = ops.conv_wgrad(x, dz, (Cout, Cin, 3, 3), 3)
        dw1, dg1, db1 = ops.stem_wgrad_bn(x, da, z, save, training, (Cout, Cin, 3, 3))
        assert torch.equal(dw0, dw1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    ref = F.conv2d(x.double().cpu().requires_grad_(True), torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True), None, 1, 1)
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wr, None, 1, 1).backward(dz.double().cpu())
    close(dw1, wr.grad, tol=2e-5, what="stem weight gradient")
    # K7 on load (round 5): x = the first half of a twin batch -- the same bits as the kernel on the materialised twin batch
    if B % 2 == 0 and G in (1, 2):
        bias = 0.25
        xh = x[:B // 2].contiguous()
        xx = torch.cat([xh, ops.complement_clip(xh, bias)], 0).contiguous()
        dwm, _, _ = ops.stem_wgrad_bn(xx, da, z, save, True, (Cout, Cin, 3, 3))
        dwt, _, _ = ops.stem_wgrad_bn(ops.twin_virtual(xh, bias), da, z, save, True, (Cout, Cin, 3, 3))
        assert torch.equal(dwm, dwt)
