"""Mint golden vectors from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference/source_code/Onet_vanilla_20240606.py ("OV") with inert
stub modules for its unused third-party imports (skimage, albumentations: see
SURVEY.md §8c), loads the deterministic weights of ``oracle.onet_oracle`` into
it with ``load_state_dict`` and records outputs of the reference's own
``forward`` / ``compute_loss`` / ``backward`` / ``log1pexp`` / ``predict_label``.

Only DATA is written (``tests/golden/*.npz``): inputs are regenerated from
seeds, weights from ``det_state_dict``.  Nothing of the reference's source is
copied.  The GPU box has no /root/reference; tests there read the npz files.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import onet_oracle as orc  # noqa: E402

REF = "/root/reference/source_code"


def import_reference():
    for name in ("skimage", "skimage.transform", "albumentations"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["skimage.transform"].resize = lambda *a, **k: None
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    argv, sys.argv = sys.argv, ["x"]
    cwd = os.getcwd()
    sys.path.insert(0, REF)
    try:
        os.chdir("/tmp")
        import Onet_vanilla_20240606 as ov
    finally:
        os.chdir(cwd)
        sys.argv = argv
    torch.set_num_threads(8)
    return ov


def grad_digest(named_grads):
    names, norms, heads = [], [], []
    for n, g in named_grads:
        g = g.detach().reshape(-1).double()
        names.append(n)
        norms.append(float(g.norm()))
        h = np.zeros(64, dtype=np.float32)
        k = min(64, g.numel())
        h[:k] = g[:k].float().numpy()
        heads.append(h)
    return np.array(names), np.array(norms, dtype=np.float64), np.stack(heads)


def bn_digest(model):
    rm, rv, nbt = [], [], []
    for n, b in model.named_buffers():
        if n.startswith("dwnu.") and model.dwnu is model.topu:
            continue
        if n.endswith("running_mean"):
            rm.append(b.detach().numpy().copy())
        elif n.endswith("running_var"):
            rv.append(b.detach().numpy().copy())
        elif n.endswith("num_batches_tracked"):
            nbt.append(int(b))
    return np.concatenate(rm), np.concatenate(rv), np.array(nbt)


def run_case(ov, tag, B, C, H, W, bshare=True, train=True, full=False, steps=1):
    torch.manual_seed(0)
    model = ov.Onet(in_chns=C, binit=True, bshare=bshare)
    model.load_state_dict(orc.onet_state_dict(C, 1981, bshare))
    model.train(train)
    X = orc.det_input(B, C, H, W)
    out = {"meta": np.array([B, C, H, W, int(bshare), int(train), steps])}
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, betas=(0.9, 0.999), eps=1e-8)
    losses = []
    for s in range(steps):
        model.zero_grad()
        if train:
            Lt, Vt, Ld, Vd, S = model(X)
            loss = model.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
            loss.backward()
        else:
            with torch.no_grad():
                Lt, Vt, Ld, Vd, S = model(X)
                loss = model.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
        losses.append(float(loss))
        if s == 0:
            out["Vt"] = Vt.detach().numpy() if full or H <= 40 else Vt.detach().numpy()[:, :, ::37, :]
            out["Vd"] = Vd.detach().numpy() if full or H <= 40 else Vd.detach().numpy()[:, :, ::37, :]
            out["S"] = S.detach().numpy() if full or H <= 40 else S.detach().numpy()[:, :, ::37, :]
            out["Lt_chsum"] = Lt.detach().sum(1).numpy() if H <= 40 else Lt.detach().sum(1).numpy()[:, ::37, :]
            out["Ld_chsum"] = Ld.detach().sum(1).numpy() if H <= 40 else Ld.detach().sum(1).numpy()[:, ::37, :]
            if full:
                out["Lt"] = Lt.detach().numpy()
                out["Ld"] = Ld.detach().numpy()
            out["label"] = model.predict_label(S).numpy().astype(np.uint8) if H <= 40 else \
                model.predict_label(S).numpy().astype(np.uint8)[:, ::37, :]
            if train:
                named = [(n, p.grad) for n, p in model.named_parameters()]
                out["grad_names"], out["grad_norms"], out["grad_heads"] = grad_digest(named)
                out["bn_rm"], out["bn_rv"], out["bn_nbt"] = bn_digest(model)
        if train and steps > 1:
            opt.step()
    out["losses"] = np.array(losses, dtype=np.float64)
    if train and steps == 1:
        # fp64 evaluation of the SAME reference graph ("truth"): tells how well-conditioned each
        # quantity is, i.e. how far the reference's own fp32 result is from exact arithmetic.
        torch.manual_seed(0)
        m64 = ov.Onet(in_chns=C, binit=True, bshare=bshare)
        m64.load_state_dict(orc.onet_state_dict(C, 1981, bshare))
        m64 = m64.double().train()
        Lt, Vt, Ld, Vd, S = m64(X.double())
        l64 = m64.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
        l64.backward()
        out["loss64"] = np.float64(l64.item())
        names, norms, heads = [], [], []
        for n, p_ in m64.named_parameters():
            g = p_.grad.detach().reshape(-1)
            norms.append(float(g.norm()))
            h = np.zeros(64, dtype=np.float64)
            k = min(64, g.numel())
            h[:k] = g[:k].numpy()
            heads.append(h)
        out["grad_norms64"] = np.array(norms, dtype=np.float64)
        out["grad_heads64"] = np.stack(heads)
        out["Vt64"] = Vt.detach().numpy() if H <= 40 else Vt.detach().numpy()[:, :, ::37, :]
        out["S64"] = S.detach().numpy() if H <= 40 else S.detach().numpy()[:, :, ::37, :]
    if steps > 1:
        out["bn_rm_end"], out["bn_rv_end"], out["bn_nbt_end"] = bn_digest(model)
        pn = [(n, p.detach()) for n, p in model.named_parameters()]
        out["param_names"], out["param_norms"], out["param_heads"] = grad_digest(pn)
    np.savez_compressed(os.path.join(HERE, f"onet_{tag}.npz"), **out)
    print(tag, "losses", losses)


def grad_samples(named_grads, n_big=1024):
    """Per parameter: L2 norm, and the gradient itself where it has <= 4096 elements (every BatchNorm gamma / beta,
    ConvTranspose2d bias, the stem convolution) or a strided sample of n_big elements spread over the whole tensor."""
    norms, vals, offs = [], [], [0]
    for _, g in named_grads:
        g = g.detach().reshape(-1).double()
        norms.append(float(g.norm()))
        v = g if g.numel() <= 4096 else g[:: g.numel() // n_big][:n_big]
        vals.append(v.numpy())
        offs.append(offs[-1] + v.numel())
    return np.array(norms), np.concatenate(vals), np.array(offs, dtype=np.int64)


def install_routing(model, routing):
    """Swap every nn.ReLU / nn.MaxPool2d of a reference model INSTANCE (OV:49,53,67) for a module that records or
    replays its decisions through `routing` (oracle.Routing): the reference's own graph, its own conv / BN / convT /
    cat / head / loss code, with only the discrete decisions taken from outside."""
    import torch.nn as nn

    class RoutedReLU(nn.Module):
        def forward(self, y):
            return routing.relu(y)

    class RoutedPool(nn.Module):
        def forward(self, a):
            return routing.maxpool2(a)

    for mod in list(model.modules()):
        if isinstance(mod, nn.Sequential):
            for i, child in enumerate(mod):
                if isinstance(child, nn.ReLU):
                    mod[i] = RoutedReLU()
                elif isinstance(child, nn.MaxPool2d):
                    mod[i] = RoutedPool()


def run_routed_case(ov, tag, B, C, H, W, head_gain=0.3, with_routed=True):
    """Well-conditioned head (|V| <= 5), larger batch.  Three evaluations of the REAL reference graph:
    fp32 (decisions recorded), fp64 deciding freely ("truth"), fp64 taking the fp32 run's decisions ("routed").
    fp32 vs truth differ by 3e-3..5e-3 on every parameter (ReLU / pooling decisions flip at rounding level), fp32 vs
    routed by ~1e-5: that is what tests/test_oracle_routing.py and the routing-controlled GPU tests build on."""
    X = orc.det_input(B, C, H, W)
    sd = orc.onet_state_dict(C, 1981, True, head_gain=head_gain)

    def build(dtype, routing=None):
        torch.manual_seed(0)
        m = ov.Onet(in_chns=C, binit=True, bshare=True)
        m.load_state_dict(sd)
        m = m.to(dtype).train()
        if routing is not None:
            install_routing(m, routing)
        Lt, Vt, Ld, Vd, S = m(X.to(dtype))
        loss = m.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
        loss.backward()
        return m, (Lt, Vt, Ld, Vd, S), loss

    r = orc.Routing()
    m32, o32, l32 = build(torch.float32, r)
    out = {"meta": np.array([B, C, H, W, 1, 1, 1]), "head_gain": np.float64(head_gain),
           "losses": np.array([float(l32)]), "Vt": o32[1].detach().numpy()[:, :, ::37, :],
           "Vd": o32[3].detach().numpy()[:, :, ::37, :], "S": o32[4].detach().numpy()[:, :, ::37, :],
           "Lt_chsum": o32[0].detach().sum(1).numpy()[:, ::37, :], "Ld_chsum": o32[2].detach().sum(1).numpy()[:, ::37, :],
           "label": m32.predict_label(o32[4]).numpy().astype(np.uint8)[:, ::37, :]}
    named = [(n, p.grad) for n, p in m32.named_parameters()]
    out["grad_names"] = np.array([n for n, _ in named])
    out["grad_norms"], out["grad_vals"], out["grad_offs"] = grad_samples(named)
    out["bn_rm"], out["bn_rv"], out["bn_nbt"] = bn_digest(m32)
    del m32, o32
    m64, o64, l64 = build(torch.float64)
    out["loss64"] = np.float64(l64.item())
    out["grad_norms64"], out["grad_vals64"], _ = grad_samples([(n, p.grad) for n, p in m64.named_parameters()])
    out["Vt64"] = o64[1].detach().numpy()[:, :, ::37, :]
    del m64, o64
    if with_routed:
        m64r, o64r, l64r = build(torch.float64, r.replay())
        out["loss64r"] = np.float64(l64r.item())
        out["grad_norms64r"], out["grad_vals64r"], _ = grad_samples([(n, p.grad) for n, p in m64r.named_parameters()])
        out["routing_audit"] = np.array([[a[1], a[2], a[3]] for a in r.audit], dtype=np.float64)
    out["grad_vals"] = out["grad_vals"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f"onet_{tag}.npz"), **out)

    def worst(a, b):
        o = out["grad_offs"]
        return max(np.linalg.norm(a[o[i]:o[i + 1]] - b[o[i]:o[i + 1]]) / np.linalg.norm(b[o[i]:o[i + 1]])
                   for i in range(len(o) - 1))
    if not with_routed:
        print(tag, "loss", float(l32), float(l64), "| fp32 vs truth %.2e" % worst(out["grad_vals"].astype(np.float64), out["grad_vals64"]))
        return
    print(tag, "loss", float(l32), float(l64), float(l64r), "| fp32 vs truth %.2e, fp32 vs routed %.2e, flips %d" % (
        worst(out["grad_vals"].astype(np.float64), out["grad_vals64"]),
        worst(out["grad_vals"].astype(np.float64), out["grad_vals64r"]), int(out["routing_audit"][:, 0].sum())))


def run_log1pexp(ov):
    model = ov.Onet(in_chns=1)
    xs = np.array([-100, -50, -37.0001, -37, -36.9999, -20, -10, -1, 0, 1, 10, 17.9999, 18, 18.0001,
                   25, 33.2999, 33.3, 33.3001, 50, 100], dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(5))
    xs = np.concatenate([xs, (rng.random(236, dtype=np.float32) * 120 - 60)])
    x = torch.tensor(xs, requires_grad=True)
    y = model.log1pexp(-1 * (-1 * x))     # non-leaf copy, mutated in place like OV:232
    y.sum().backward()
    np.savez_compressed(os.path.join(HERE, "log1pexp.npz"), x=xs, y=y.detach().numpy(), dy=x.grad.numpy())
    print("log1pexp", y[:5].tolist())


def run_loss_extreme(ov):
    """compute_loss + grads on hand-built L, S that hit all four log1pexp branches."""
    model = ov.Onet(in_chns=1)
    rng = np.random.Generator(np.random.PCG64(11))
    B, C, H, W = 2, 64, 8, 8
    Lt = torch.tensor((rng.random((B, C, H, W), dtype=np.float32) * 2.0), requires_grad=True)
    Ld = torch.tensor((rng.random((B, C, H, W), dtype=np.float32) * 1.0), requires_grad=True)
    Vt = torch.tensor(rng.standard_normal((B, 1, H, W)).astype(np.float32) * 3, requires_grad=True)
    Vd = torch.tensor(rng.standard_normal((B, 1, H, W)).astype(np.float32) * 3, requires_grad=True)
    S = torch.softmax(torch.cat([Vt, Vd], 1), 1)
    loss = model.compute_loss(Lt, S[:, 0].unsqueeze(1), Ld, S[:, 1].unsqueeze(1))
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "loss_extreme.npz"),
                        Lt=Lt.detach().numpy(), Ld=Ld.detach().numpy(), Vt=Vt.detach().numpy(),
                        Vd=Vd.detach().numpy(), loss=np.float64(loss.item()),
                        dLt=Lt.grad.numpy(), dLd=Ld.grad.numpy(), dVt=Vt.grad.numpy(), dVd=Vd.grad.numpy())
    print("loss_extreme", float(loss))


def run_up_block(ov, bilinear, tag, h=12, w=12, H=25, W=25):
    """Up block alone (OV:75-101) incl. the F.pad path (12->24 vs skip 25, NAU-rain shape)."""
    torch.manual_seed(0)
    up = ov.Up(128, 64, bilinear=bilinear)
    sd = up.state_dict()
    new = {}
    import zlib   # deterministic fill keyed by crc32(name) (python hash() is salted)
    for k, v in sd.items():
        rng = np.random.Generator(np.random.PCG64([3, zlib.crc32(k.encode())]))
        if v.dtype == torch.long:
            new[k] = torch.tensor(0)
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy((1 + 0.1 * np.abs(rng.standard_normal(tuple(v.shape)))).astype(np.float32))
        elif v.dim() == 4:
            fan = v.shape[1] * v.shape[2] * v.shape[3]
            new[k] = torch.from_numpy((rng.standard_normal(tuple(v.shape)) * np.sqrt(2.0 / fan)).astype(np.float32))
        elif k.endswith(".weight"):
            new[k] = torch.from_numpy((1 + 0.1 * rng.standard_normal(tuple(v.shape))).astype(np.float32))
        else:
            new[k] = torch.from_numpy((0.1 * rng.standard_normal(tuple(v.shape))).astype(np.float32))
    up.load_state_dict(new)
    up.train()
    x1 = orc.det_input(2, 64 if bilinear else 128, h, w, seed=21).requires_grad_(True)
    x2 = orc.det_input(2, 64, H, W, seed=22).requires_grad_(True)
    y = up(x1, x2)
    g = orc.det_input(*y.shape, seed=23) - 0.5
    (y * g).sum().backward()
    out = {"y": y.detach().numpy(), "dx1": x1.grad.numpy(), "dx2": x2.grad.numpy(),
           "meta": np.array([h, w, H, W, int(bilinear)])}
    named = [(n, p.grad) for n, p in up.named_parameters()]
    out["grad_names"], out["grad_norms"], out["grad_heads"] = grad_digest(named)
    np.savez_compressed(os.path.join(HERE, f"up_{tag}.npz"), **out)
    print("up", tag, float(y.abs().mean()))


def eval_side_cases():
    """Label maps for the eval-side fixtures (stored in the fixture next to the outputs): random, complement of the ground
    truth, all-background prediction, empty ground truth, both empty, tie between the two assignments."""
    rng = np.random.Generator(np.random.PCG64(2024))
    H, W = 24, 40
    gt = (rng.random((H, W)) > 0.8).astype(np.int64)
    cases = [("random", (rng.random((H, W)) > 0.7).astype(np.int64), gt),
             ("complement", 1 - gt, gt),
             ("mostly_right", np.where(rng.random((H, W)) > 0.1, gt, 1 - gt), gt),
             ("pred_all_bg", np.zeros((H, W), np.int64), gt),
             ("gt_all_bg", (rng.random((H, W)) > 0.5).astype(np.int64), np.zeros((H, W), np.int64)),
             ("both_all_bg", np.zeros((H, W), np.int64), np.zeros((H, W), np.int64)),
             ("both_all_fg", np.ones((H, W), np.int64), np.ones((H, W), np.int64))]
    cases.append(("disjoint", np.zeros((H, W), np.int64), np.ones((H, W), np.int64)))   # UT:152 raises AttributeError
    tie_p = np.zeros((H, W), np.int64)
    tie_p[:, : W // 2] = 1
    tie_g = np.zeros((H, W), np.int64)
    tie_g[: H // 2] = 1
    cases.append(("tie", tie_p, tie_g))
    return cases


def run_eval_side():
    """Eval-side helpers of the reference (utils_20231218.py, "UT") on fixed label maps / frames:
    _acc, _miou, _target_iou, _detection_rate, _false_alarm_rate (UT:100-192), re_assign_label (UT:410-453),
    reorder_segmentation / _hungarian_match (UT:258-285, 360-375), tensor_normal_per_frame (UT:673-689)."""
    import utils_20231218 as ut
    out = {}
    names = []
    for name, p, g in eval_side_cases():
        names.append(name)
        out[f"{name}_pred"], out[f"{name}_gt"] = p.astype(np.uint8), g.astype(np.uint8)
        P, Gt = torch.from_numpy(p), torch.from_numpy(g)
        Gf = Gt.to(torch.float32)                      # the reference's label tensors are float32 (RG:283)
        vals = []
        for fn in (lambda: ut._acc(P, Gt, 2), lambda: ut._miou(P, Gt, 2), lambda: ut._target_iou(P, Gt),
                   lambda: ut._detection_rate(P, Gt), lambda: ut._false_alarm_rate(P, Gt)):
            try:
                vals.append(float(fn()))
            except AttributeError:                     # UT:152: `.item()` on a python float when no class is mixed
                vals.append(np.nan)
        out[f"{name}_metrics"] = np.array(vals, dtype=np.float64)
        out[f"{name}_reassign"] = ut.re_assign_label(P, Gf).numpy().astype(np.int64)
        out[f"{name}_reorder"] = ut.reorder_segmentation(P.clone(), Gt.clone()).numpy().astype(np.int64)
    out["names"] = np.array(names)
    rng = np.random.Generator(np.random.PCG64(77))
    X = (rng.standard_normal((3, 2, 19, 23)) * 5 + 2).astype(np.float32)
    X[1, 0] = 0.75                                     # constant frame: max == min
    out["norm_in"] = X
    out["norm_out"] = ut.tensor_normal_per_frame(torch.from_numpy(X)).numpy()
    np.savez_compressed(os.path.join(HERE, "eval_side.npz"), **out)
    print("eval_side", {n: out[f"{n}_metrics"].round(4).tolist() for n in names})


def unet_loss(x1, y1):
    """a scalar the bare-UNet cases differentiate (the reference has no loss for a UNet on its own): couples both outputs"""
    return (x1 * y1).mean() + 0.5 * (y1 * y1).mean()


def run_unet_bilinear(ov, tag="unet_bilinear_b4_c1_40", B=4, C=1, H=40, W=40):
    """A full `UNet(bilinear=True)` of the REAL reference (OV:104-153 with the nn.Upsample variant of Up, OV:83-84, incl. the
    F.pad path: 2 -> 4 against a 5-pixel skip): train-mode forward, `unet_loss`, backward -- in fp32 (decisions recorded),
    fp64 deciding freely and fp64 replaying the fp32 decisions.  Pins oracle.unet_pass(bilinear=True) and its routing."""
    X = orc.det_input(B, C, H, W)
    sd = orc.det_state_dict(C, 1981, bilinear=True)

    def build(dtype, routing=None):
        m = ov.UNet(n_channels=C, n_classes=1, binit=True, bilinear=True)
        m.load_state_dict(sd)                       # strict: the oracle's 108-row parameter table IS the reference's key set
        m = m.to(dtype).train()
        if routing is not None:
            install_routing(m, routing)
        x1, y1 = m(X.to(dtype))
        loss = unet_loss(x1, y1)
        loss.backward()
        return m, x1, y1, loss

    r = orc.Routing()
    m32, x1, y1, l32 = build(torch.float32, r)
    named = [(n, p.grad) for n, p in m32.named_parameters()]
    out = {"meta": np.array([B, C, H, W]), "loss": np.float64(l32.item()), "x1": x1.detach().numpy(), "y1": y1.detach().numpy(),
           "grad_names": np.array([n for n, _ in named])}
    out["grad_norms"], out["grad_vals"], out["grad_offs"] = grad_samples(named)
    out["bn_rm"], out["bn_rv"], out["bn_nbt"] = bn_digest(m32)
    m64, _, _, l64 = build(torch.float64)
    out["loss64"] = np.float64(l64.item())
    out["grad_norms64"], out["grad_vals64"], _ = grad_samples([(n, p.grad) for n, p in m64.named_parameters()])
    m64r, _, _, l64r = build(torch.float64, r.replay())
    out["loss64r"] = np.float64(l64r.item())
    out["grad_norms64r"], out["grad_vals64r"], _ = grad_samples([(n, p.grad) for n, p in m64r.named_parameters()])
    out["routing_audit"] = np.array([[a[1], a[2], a[3]] for a in r.audit], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    o = out["grad_offs"]

    def worst(a, b):
        return max(np.linalg.norm(a[o[i]:o[i + 1]] - b[o[i]:o[i + 1]]) / np.linalg.norm(b[o[i]:o[i + 1]]) for i in range(len(o) - 1))
    print(tag, "loss", float(l32), float(l64), float(l64r), "| fp32 vs truth %.2e, fp32 vs routed %.2e, flips %d" % (
        worst(out["grad_vals"], out["grad_vals64"]), worst(out["grad_vals"], out["grad_vals64r"]), int(out["routing_audit"][:, 0].sum())))


def import_generators():
    """-> (KD, RG): the reference's clutter / frame generator modules (torchvision is not installed: the one transform RG:302
    uses, CenterCrop, is supplied)."""
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, t):
            h, w = self.size
            H, W = t.shape[-2:]
            return t[..., (H - h) // 2:(H - h) // 2 + h, (W - w) // 2:(W - w) // 2 + w]

    tvt.CenterCrop = CenterCrop
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    cwd = os.getcwd()
    try:
        os.chdir("/tmp")
        import K_distributed_SeaClutter_Simulation_20210919 as kd
        import Rayleigh_bg_Gaussian_EOT_generator_20230208 as rg
    finally:
        os.chdir(cwd)
    return kd, rg


def run_clutter_stats(n_frames=24):
    """SURVEY 8f row 4 pinned to the reference: `n_frames` K-clutter frames with targets made by the reference's OWN frame
    function `get_k_frame` (RG:177-216 -> KD:469-526 `generate_K_distributed_noise`, KD:270-297, RG:63-175) after
    np.random.seed(1981), PSNR cycling over TS:668's (0, 1, 2); the texture, speckle and clutter fields are picked up on the
    way out of the reference's functions.  Stored: one value per frame of every statistic of tests/clutter_stats.py, the
    per-frame polynomial coefficients of KD:491-492, and how far this build's closed-form root rule (onet_amd.data.
    gaussian_acf_field) is from the reference's numpy.roots field for the same coefficients (frame 0)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import clutter_stats as cs
    from onet_amd import data as odata
    kd, rg = import_generators()
    seen = {}
    real_k, real_s, real_c, real_solve = (kd.generate_K_distributed_noise, kd.generate_correlated_Gaussian_via_expdecay,
                                          kd.coeff_acf_polyn, kd.solve_acf_polyn)

    def k_noise(*a, **k):
        amp, tau = real_k(*a, **k)
        seen["amp"], seen["tau"] = amp.copy(), tau.copy()
        return amp, tau

    def speckle(*a, **k):
        seen["speckle"] = real_s(*a, **k)
        return seen["speckle"]

    def coeffs(*a, **k):
        c = real_c(*a, **k)
        seen["coeffs"] = np.array(c) / c[-1]
        return c

    def solve(gamma_acf, co):
        out = real_solve(gamma_acf, co)
        seen.setdefault("solve0", (np.array(co, dtype=np.float64), out.copy()))
        return out

    kd.generate_K_distributed_noise, kd.generate_correlated_Gaussian_via_expdecay = k_noise, speckle
    kd.coeff_acf_polyn, kd.solve_acf_polyn = coeffs, solve
    rows, coefs = [], []
    np.random.seed(1981)
    try:
        for i in range(n_frames):
            snr = (0, 1, 2)[i % 3]
            frame, mask = rg.get_k_frame(snr)
            r = {"snr": float(snr)}
            r.update(cs.texture_stats(seen["tau"]))
            r.update(cs.speckle_stats(seen["speckle"]))
            r.update(cs.amplitude_stats(seen["amp"].astype(np.float32)))
            r.update(cs.frame_stats(frame, np.asarray(mask, dtype=np.float64), seen["amp"].astype(np.float32)))
            rows.append(r)
            coefs.append(seen["coeffs"])
            print("clutter frame", i, {k: round(v, 4) for k, v in r.items() if k in ("tex_var", "amp_m4_ratio", "lab_frac", "scr_db")},
                  flush=True)
    finally:
        kd.generate_K_distributed_noise, kd.generate_correlated_Gaussian_via_expdecay = real_k, real_s
        kd.coeff_acf_polyn, kd.solve_acf_polyn = real_c, real_solve
    # the closed-form root rule against numpy.roots, on the reference's own field and coefficients (frame 0)
    co0, field0 = seen["solve0"]
    odata._COEFFS[5.0] = (co0[0], co0[1], 1.0)
    mine = odata.gaussian_acf_field(400, 5.0)
    del odata._COEFFS[5.0]
    root_rule_err = float(np.abs(mine - field0).max())
    out = {k: v for k, v in cs.collect(rows).items()}
    out["poly_coeffs"] = np.asarray(coefs, dtype=np.float64)
    out["root_rule_max_abs_err"] = np.float64(root_rule_err)
    out["quadrature_coeffs"] = np.asarray(odata.acf_poly_coeffs(5.0), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "clutter_stats.npz"), **out)
    print("clutter_stats:", n_frames, "frames; coefficient mean", out["poly_coeffs"].mean(0), "std", out["poly_coeffs"].std(0),
          "quadrature", out["quadrature_coeffs"], "root rule err", root_rule_err)


def run_wire_formats(ov):
    """(1) key list of the REAL Onet state_dict (names, shapes, dtypes; 232 entries, TS:264) and of a checkpoint dict as
    the trainers write it; (2) a small data file written by the reference's own writer `prepare_data` (RG:300-324) --
    with the frame count per PSNR cut from 150 to 2 and a 48x48 centre crop so that the fixture stays small (the
    writer's hard-coded `psnrs.extend([psnr] * 150)` is left as it is: 1650 PSNR entries for 22 frames)."""
    import json
    sd = ov.Onet(in_chns=1, binit=True, bshare=True).state_dict()
    sd3 = ov.Onet(in_chns=3, binit=True, bshare=False).state_dict()
    spec = {"onet_c1_share": [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()],
            "onet_c3_noshare": [[k, list(v.shape), str(v.dtype)] for k, v in sd3.items()],
            "checkpoint_keys_sim": ["net", "epoch"], "checkpoint_keys_zy3": ["net", "save_epoch"]}
    json.dump(spec, open(os.path.join(HERE, "state_dict_keys.json"), "w"))
    rg = import_generators()[1]
    real = rg.prepare_frames
    rg.prepare_frames = lambda type="rayleigh", fnums=4, snr=10: real(type=type, fnums=2, snr=snr)
    np.random.seed(1981)
    path = os.path.join(HERE, "rayleigh_writer_sample.pt")
    rg.prepare_data(img_sz=(48, 48), bg_type="rayleigh", file_name=path)
    d = torch.load(path)
    print("wire formats:", len(spec["onet_c1_share"]), "keys;", {k: (tuple(v.shape) if hasattr(v, "shape") else len(v))
                                                                  for k, v in d.items()})


def run_train_behaviour(ov, tag="train_b4_c3_64", B=4, C=3, H=64, W=64, steps=40):
    """The reachable part of BASELINE configs[4]'s "IoU parity": the REAL reference `Onet(in_chns=3)` trained for `steps` epochs of
    ONE batch each in the order of TZ:99-128 -- onet.train(); zero_grad; forward; slice S; compute_loss; backward; Adam(lr 1e-4)
    step; loss.item(); CosineAnnealingWarmRestarts(T_0=300, T_mult=2, eta_min=1e-6).step() per epoch -- on synthetic bright-blob
    tiles (SURVEY 8d's C5 stand-in, onet_amd.data.make_blob_tiles: the ZY-3 tiles are not in the image), then evaluated as
    UZ:160-181 does: eval(); no_grad forward; predict_label; reorder_segmentation; evaluate_segmentation -> (acc, mIoU), with the
    reference's OWN utils_20231218 functions.  Stored: the fp32 loss sequence, the same loop in fp64 (how far fp32 training itself
    drifts from exact arithmetic: the yardstick for the build's tolerance), final label maps, acc / mIoU of both, lr sequence."""
    import utils_20231218 as ut
    from onet_amd.data import make_blob_tiles
    Xn, Mn = make_blob_tiles(B, H, W, seed=4242, channels=C)
    out = {"meta": np.array([B, C, H, W, steps]), "x_sum": np.float64(Xn.astype(np.float64).sum()), "mask": Mn.astype(np.uint8)}
    for dtype, key in ((torch.float32, "32"), (torch.float64, "64")):
        torch.manual_seed(0)
        onet = ov.Onet(in_chns=C, binit=True, bshare=True)
        onet.load_state_dict(orc.onet_state_dict(C, 1981, True))
        onet = onet.to(dtype)
        X, label = torch.from_numpy(Xn).to(dtype), torch.from_numpy(Mn)
        opt = torch.optim.Adam(onet.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, amsgrad=False)
        sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=300, T_mult=2, eta_min=0.000001, last_epoch=-1)
        losses, lrs = [], []
        for epoch in range(steps):
            onet.train()
            lrs.append(opt.param_groups[0]["lr"])
            onet.zero_grad()
            Lt, Vt, Ld, Vd, S = onet(X)
            St = S[:, 0, :, :].unsqueeze(dim=1)
            Sd = S[:, 1, :, :].unsqueeze(dim=1)
            loss = onet.compute_loss(Lt, St, Ld, Sd)
            loss.backward()
            opt.step()
            losses.append(loss.item())
            sch.step()
        onet.eval()
        with torch.no_grad():
            Lt, Vt, Ld, Vd, S = onet(X)
            pred = onet.predict_label(S)
            Y = ut.reorder_segmentation(pred, label)
            tl = onet.compute_loss(Lt, S[:, 0:1], Ld, S[:, 1:2]).item()
            acc, miou = ut.evaluate_segmentation(Y, label, gt_k=2)
        out["losses" + key], out["lrs"] = np.array(losses, dtype=np.float64), np.array(lrs, dtype=np.float64)
        out["eval_loss" + key], out["acc" + key], out["miou" + key] = np.float64(tl), np.float64(acc), np.float64(miou)
        out["pred" + key], out["Y" + key] = pred.numpy().astype(np.uint8), Y.numpy().astype(np.uint8)
        out["Vt_eval" + key] = Vt.float().numpy()
        print("train", tag, key, "loss %.5f -> %.5f  eval %.5f  acc %.4f miou %.4f" % (losses[0], losses[-1], tl, acc, miou), flush=True)
    out["loss_drift_32_vs_64"] = np.float64(np.max(np.abs(out["losses32"] - out["losses64"]) / np.abs(out["losses64"])))
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    print("train", tag, "max relative loss drift fp32 vs fp64 over %d steps: %.2e; labels differing: %d px"
          % (steps, float(out["loss_drift_32_vs_64"]), int((out["pred32"] != out["pred64"]).sum())))


if __name__ == "__main__":
    ov = import_reference()
    if os.environ.get("GOLDEN_ONLY") == "train":         # configs[4]: training behaviour + IoU on blob tiles
        run_train_behaviour(ov)
        sys.exit(0)
    if os.environ.get("GOLDEN_ONLY") == "evalside":
        run_eval_side()
        run_wire_formats(ov)
        sys.exit(0)
    if os.environ.get("GOLDEN_ONLY") == "bilinear":
        run_unet_bilinear(ov)
        sys.exit(0)
    if os.environ.get("GOLDEN_ONLY") == "clutter":
        run_clutter_stats()
        sys.exit(0)
    if os.environ.get("GOLDEN_ONLY") == "up":
        run_up_block(ov, False, "convT_pad")
        run_up_block(ov, True, "bilinear_pad")
        sys.exit(0)
    if os.environ.get("GOLDEN_ONLY") == "c5":            # BASELINE configs[4]'s tile shape: 3 x 512 x 512 (ZY-3 cloud tiles)
        run_routed_case(ov, "routed_b1_c3_512", 1, 3, 512, 512)
        sys.exit(0)
    if os.environ.get("GOLDEN_ONLY") == "routed":
        run_routed_case(ov, "routed_b8_c1_128", 8, 1, 128, 128)
        run_routed_case(ov, "routed_b4_c1_256", 4, 1, 256, 256)
        sys.exit(0)
    run_log1pexp(ov)
    run_loss_extreme(ov)
    run_case(ov, "b2_c1_16", 2, 1, 16, 16, full=True)
    run_case(ov, "b2_c1_32", 2, 1, 32, 32)
    run_case(ov, "b2_c3_32", 2, 3, 32, 32)
    run_case(ov, "b2_c1_40", 2, 1, 40, 40)            # F.pad path (40/16 not integral)
    run_case(ov, "b3_c1_32", 3, 1, 32, 32)            # odd batch
    run_case(ov, "b2_c1_32_eval", 2, 1, 32, 32, train=False)
    run_case(ov, "b2_c1_32_noshare", 2, 1, 32, 32, bshare=False)
    run_case(ov, "b2_c1_32_adam4", 2, 1, 32, 32, steps=4)   # harness contract
    run_case(ov, "b2_c1_256", 2, 1, 256, 256)         # BASELINE config C1 shape
    run_up_block(ov, False, "convT_pad")
    run_up_block(ov, True, "bilinear_pad")
    run_unet_bilinear(ov)
    run_routed_case(ov, "routed_b8_c1_128", 8, 1, 128, 128)   # unsaturated head, large N, full / strided gradients
    run_routed_case(ov, "routed_b4_c1_256", 4, 1, 256, 256)
    run_routed_case(ov, "routed_b1_c3_512", 1, 3, 512, 512)   # BASELINE configs[4]'s tile shape, batch statistics over ONE image
    run_eval_side()
    run_wire_formats(ov)
    run_clutter_stats()
    run_train_behaviour(ov)
