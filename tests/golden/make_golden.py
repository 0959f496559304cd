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


if __name__ == "__main__":
    ov = import_reference()
    if os.environ.get("GOLDEN_ONLY") == "up":
        run_up_block(ov, False, "convT_pad")
        run_up_block(ov, True, "bilinear_pad")
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
