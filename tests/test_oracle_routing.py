"""Routing-controlled evaluation of the CPU oracle, pinned to the REAL reference (CPU only).

The Onet gradient is discontinuous in every ReLU / max-pool decision, so "fp32 gradient within 1e-3 of the exact
one" is not a property ANY fp32 evaluation has -- the reference's own ATen kernels sit 3e-3..7e-3 from their fp64
selves on every parameter (golden `routed_*`: recorded from the reference, reproduced here).  With the decisions
of the fp32 run replayed in the fp64 evaluation (oracle.Routing) the same comparison gives ~2e-5.  The GPU parity
tests use exactly this: fp64 oracle under the HIP path's own decisions, every gradient element, tolerance 1e-4,
plus an audit that the decisions differ from the free fp64 ones only at rounding distance from the switch."""
import os

import numpy as np
import pytest
import torch

from oracle import onet_oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")


def _to64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def _samples(grads, names, offs, n_big=1024):
    out = []
    for n in names:
        g = grads[n[5:]].detach().reshape(-1).double()
        out.append((g if g.numel() <= 4096 else g[:: g.numel() // n_big][:n_big]).numpy())
    v = np.concatenate(out)
    assert v.size == offs[-1]
    return v


def _worst(a, b, offs):
    return max(np.linalg.norm(a[offs[i]:offs[i + 1]] - b[offs[i]:offs[i + 1]]) /
               (np.linalg.norm(b[offs[i]:offs[i + 1]]) + 1e-300) for i in range(len(offs) - 1))


@pytest.mark.parametrize("tag", ["routed_b8_c1_128", "routed_b4_c1_256"])
def test_routed_oracle_matches_routed_reference(tag):
    g = np.load(os.path.join(G, f"onet_{tag}.npz"))
    B, C, H, W = [int(v) for v in g["meta"][:4]]
    gain = float(g["head_gain"])
    names = [str(n) for n in g["grad_names"]]
    offs = g["grad_offs"]
    X = orc.det_input(B, C, H, W)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    r = orc.Routing()
    top = orc.clone_state(orc.det_state_dict(C, 1981, head_gain=gain))
    (Lt, Vt, Ld, Vd, S), loss, g32 = orc.train_mode_step(X, top, routing=r)
    # fp32: same ATen ops as the reference -> same numbers, same decisions
    assert abs(float(loss) - g["losses"][0]) <= 2e-6 * abs(g["losses"][0])
    np.testing.assert_allclose(Vt.detach().numpy()[:, :, ::37, :], g["Vt"], rtol=1e-4, atol=1e-4)
    assert float(Vt.detach().abs().max()) < 6.0                      # the head is NOT saturated in these cases
    v32 = _samples(g32, names, offs)
    assert _worst(v32, g["grad_vals"].astype(np.float64), offs) <= 2e-4
    # fp64 under the fp32 run's decisions: the reference's routed evaluation, to fp64 accuracy
    top64 = orc.clone_state(_to64(orc.det_state_dict(C, 1981, head_gain=gain)))
    _, loss64r, g64r = orc.train_mode_step(X.double(), top64, routing=r.replay())
    assert abs(float(loss64r) - float(g["loss64r"])) <= 1e-10 * abs(float(g["loss64r"]))
    v64r = _samples(g64r, names, offs)
    assert _worst(v64r, g["grad_vals64r"], offs) <= 1e-7
    for i, n in enumerate(names):
        assert abs(float(g64r[n[5:]].norm()) - g["grad_norms64r"][i]) <= 1e-7 * g["grad_norms64r"][i], n
    # the statement itself, on the reference's numbers: routed 1e-4 or better, free 1e-3 or worse
    e_routed = _worst(g["grad_vals"].astype(np.float64), g["grad_vals64r"], offs)
    e_free = _worst(g["grad_vals"].astype(np.float64), g["grad_vals64"], offs)
    assert e_routed <= 1e-4 and e_free >= 1e-3, (e_routed, e_free)
    # and every decision the fp64 run would have taken differently sits at rounding distance from its switch
    audit = g["routing_audit"]
    assert audit[:, 0].sum() > 0 and audit[:, 2].max() <= 1e-5
    assert (audit[:, 0] / audit[:, 1]).max() <= 1e-4


def test_routing_from_activations_equals_recorded_routing():
    """Routing.from_activations (how the GPU tests obtain the HIP path's decisions) rebuilds the recorded decisions
    from the post-ReLU activations alone."""
    B, C, H, W = 2, 1, 32, 32
    X = orc.det_input(B, C, H, W)
    acts = []
    real = orc.Routing.relu

    class Rec(orc.Routing):
        def relu(self, y):
            out = real(self, y)
            acts.append(out.detach().clone())
            return out

    r = Rec()
    top = orc.clone_state(orc.det_state_dict(C, 1981))
    _, _, g_rec = orc.train_mode_step(X, top, routing=r)
    r2 = orc.Routing.from_activations(acts)
    assert len(r2.masks) == 36 and len(r2.pools) == 8
    for a, b in zip(r.masks, r2.masks):
        assert torch.equal(a, b)
    for a, b in zip(r.pools, r2.pools):
        assert torch.equal(a, b)
    top2 = orc.clone_state(orc.det_state_dict(C, 1981))
    _, _, g_rep = orc.train_mode_step(X, top2, routing=r2)
    assert all(a[1] == 0 for a in r2.audit)
    for k in g_rec:
        assert torch.allclose(g_rec[k], g_rep[k], rtol=1e-6, atol=1e-9), k


def unet_loss(x1, y1):
    """the scalar tests/golden/make_golden.py::unet_loss differentiates for a bare UNet"""
    return (x1 * y1).mean() + 0.5 * (y1 * y1).mean()


def unet_step(X, sd, bilinear, routing=None):
    x1, y1 = orc.unet_pass(X, sd, True, routing, bilinear)
    loss = unet_loss(x1, y1)
    loss.backward()
    return x1, y1, loss, {k: v.grad for k, v in sd.items() if v.requires_grad}


def test_bilinear_unet_oracle_matches_reference():
    """oracle.unet_pass(bilinear=True) -- the nn.Upsample variant of Up inside a FULL UNet (OV:83-84, 115-120; F.pad path
    2 -> 4 against a 5-pixel skip) -- pinned to the real reference's `UNet(bilinear=True)`: parameter table == its state_dict
    (the golden was minted with a strict load_state_dict), fp32 outputs / loss / gradients, and the fp64 evaluation under the
    fp32 run's decisions."""
    g = np.load(os.path.join(G, "unet_bilinear_b4_c1_40.npz"))
    B, C, H, W = [int(v) for v in g["meta"]]
    names, offs = [str(n) for n in g["grad_names"]], g["grad_offs"]
    X = orc.det_input(B, C, H, W)
    r = orc.Routing()
    sd = orc.clone_state(orc.det_state_dict(C, 1981, bilinear=True))
    assert [k for k, v in sd.items() if v.requires_grad] == names and len(sd) == 108
    x1, y1, loss, g32 = unet_step(X, sd, True, r)
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    np.testing.assert_allclose(x1.detach().numpy(), g["x1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y1.detach().numpy(), g["y1"], rtol=1e-4, atol=1e-5)

    def samples(grads):
        return np.concatenate([(lambda t: (t if t.numel() <= 4096 else t[:: t.numel() // 1024][:1024]).numpy())(
            grads[n].detach().reshape(-1).double()) for n in names])

    assert _worst(samples(g32), g["grad_vals"], offs) <= 2e-4
    sd64 = orc.clone_state(_to64(orc.det_state_dict(C, 1981, bilinear=True)))
    _, _, loss64r, g64r = unet_step(X.double(), sd64, True, r.replay())
    assert abs(float(loss64r) - float(g["loss64r"])) <= 1e-10 * abs(float(g["loss64r"]))
    assert _worst(samples(g64r), g["grad_vals64r"], offs) <= 1e-7
    e_routed, e_free = _worst(g["grad_vals"], g["grad_vals64r"], offs), _worst(g["grad_vals"], g["grad_vals64"], offs)
    assert e_routed <= 1e-4 and e_free >= 1e-3, (e_routed, e_free)


def test_rounded_operand_functions_reduce_to_autograd_on_bf16_representable_operands():
    """oracle.operand_rounding's two autograd Functions (the checker of the bf16 gradient-parity test): on operands that are
    already bf16 numbers rounding is the identity, so forward, input gradient, weight gradient and bias gradient must equal
    autograd's -- for every subset of products that takes rounded operands; on general operands the forward must equal the
    convolution of the rounded operands, and a replayed operand must be used instead of the function's own rounding."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    rb = lambda t: t.to(torch.bfloat16).double()
    x, w, g = rb(torch.randn(2, 5, 8, 8)).requires_grad_(True), rb(torch.randn(7, 5, 3, 3)).requires_grad_(True), rb(torch.randn(2, 7, 8, 8))
    F.conv2d(x, w, None, 1, 1).backward(g)
    ref = (x.grad.clone(), w.grad.clone())
    for prods in ({"fwd", "dgrad", "wgrad"}, {"wgrad"}, {"fwd"}):
        x.grad = w.grad = None
        orc._RoundedConv3x3.apply(x, w, prods).backward(g)
        assert torch.equal(x.grad, ref[0]) and torch.equal(w.grad, ref[1])
    xt, wt, bt = rb(torch.randn(2, 6, 4, 4)).requires_grad_(True), rb(torch.randn(6, 3, 2, 2)).requires_grad_(True), torch.randn(3, dtype=torch.float64, requires_grad=True)
    gt = rb(torch.randn(2, 3, 8, 8))
    F.conv_transpose2d(xt, wt, bt, stride=2).backward(gt)
    ref = (xt.grad.clone(), wt.grad.clone(), bt.grad.clone())
    for prods in ({"fwd", "dgrad", "wgrad"}, {"wgrad"}):
        xt.grad = wt.grad = bt.grad = None
        orc._RoundedConvT2x2.apply(xt, wt, bt, prods).backward(gt)
        assert torch.allclose(xt.grad, ref[0], rtol=0, atol=1e-12) and torch.allclose(wt.grad, ref[1], rtol=0, atol=1e-12) and torch.equal(bt.grad, ref[2])
    y, v = torch.randn(1, 4, 6, 6, dtype=torch.float64), torch.randn(3, 4, 3, 3, dtype=torch.float64)
    assert torch.equal(orc._RoundedConv3x3.apply(y, v, {"fwd"}), F.conv2d(rb(y), rb(v), None, 1, 1))
    other = rb(torch.randn(1, 4, 6, 6))
    assert torch.equal(orc._RoundedConv3x3.apply(y, v, {"fwd"}, other, None), F.conv2d(other, rb(v), None, 1, 1))
