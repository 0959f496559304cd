"""Statistics of synthetic K-clutter frames (test infrastructure): the SAME estimators are applied to frames produced by
the reference's own generator functions (tests/golden/make_golden.py::run_clutter_stats -> tests/golden/clutter_stats.npz), to
the NumPy statement of the recipe (onet_amd/data.py) and to the GPU generator (csrc/clutter.hip).  One value per frame, so that
a fixture carries the frame-to-frame spread its comparison bound is built from."""
import numpy as np

LAGS = (1, 2, 5, 10, 20, 60)


def _acf(field, lag, axis):
    """normalised autocovariance of a real [n, n] field at `lag` along `axis` (non-circular estimator)."""
    a = field - field.mean()
    if axis == 1:
        num = (a[:, :-lag] * a[:, lag:]).mean()
    else:
        num = (a[:-lag, :] * a[lag:, :]).mean()
    return float(num / (a * a).mean())


def texture_stats(tau):
    """Gamma texture [n, n] (KD:503): first two moments, normalised second moment, ACF at LAGS along x and y."""
    tau = np.asarray(tau, dtype=np.float64)
    out = {"tex_mean": float(tau.mean()), "tex_var": float(tau.var()), "tex_m2": float((tau ** 2).mean() / tau.mean() ** 2)}
    for lag in LAGS:
        out[f"tex_acf_x{lag}"] = _acf(tau, lag, 1)
        out[f"tex_acf_y{lag}"] = _acf(tau, lag, 0)
    return out


def speckle_stats(s):
    """complex speckle [n, n] (KD:270-297): power, the log-log slope of its periodogram against the recipe's own frequency
    grid f = hypot(fx, fy), fx = fy = linspace(0.1, n / 10, n) (-0.6 by construction), the share of its power in the lowest
    5 x 5 frequency bins, |s|^2 ACF at lags 1, 5, 20 along x."""
    s = np.asarray(s, dtype=np.complex128)
    n = s.shape[0]
    P = np.abs(np.fft.fft2(s)) ** 2
    f = np.linspace(0.1, n / 10.0, num=n, endpoint=True)
    fx, fy = np.meshgrid(f, f)
    slope = float(np.polyfit(np.log(np.hypot(fx, fy)).ravel(), np.log(P + 1e-300).ravel(), 1)[0])
    out = {"spk_power": float((np.abs(s) ** 2).mean()), "spk_psd_slope": slope, "spk_low_share": float(P[:5, :5].sum() / P.sum()),
           "spk_re_im_ratio": float((s.real ** 2).mean() / (s.imag ** 2).mean())}
    inten = np.abs(s) ** 2
    for lag in (1, 5, 20):
        out[f"spk_acf_x{lag}"] = _acf(inten, lag, 1)
    return out


def amplitude_stats(a):
    """K amplitude [n, n] (KD:519-520): E[a^2], E[a^4] / E[a^2]^2, E[a] / sqrt(E[a^2]), intensity ACF at lags 1, 5, 20 (x, y)."""
    a = np.asarray(a, dtype=np.float64)
    m2 = (a ** 2).mean()
    out = {"amp_m2": float(m2), "amp_m4_ratio": float((a ** 4).mean() / m2 ** 2), "amp_m1_ratio": float(a.mean() / np.sqrt(m2))}
    for lag in (1, 5, 20):
        out[f"amp_acf_x{lag}"] = _acf(a ** 2, lag, 1)
        out[f"amp_acf_y{lag}"] = _acf(a ** 2, lag, 0)
    return out


def frame_stats(frame, label, clutter, crop=256):
    """frame with targets [n, n] (RG:177-216), its label mask and the clutter it was built on: label area fraction (full
    frame and the central crop), signal-to-clutter ratio as RG:277-294 computes it on the min-max normalised frame, the raw
    peak over the clutter's rms, the fraction of label pixels the compositing rule (template > background) actually raised."""
    frame, label, clutter = (np.asarray(v, dtype=np.float64) for v in (frame, label, clutter))
    n = frame.shape[0]
    c0 = (n - crop) // 2
    nrm = (frame - frame.min()) / (frame.max() - frame.min() + np.spacing(1))
    sig = ((label * nrm) ** 2).sum() / max(1.0, (label == 1).sum())
    noi = (((1 - label) * nrm) ** 2).sum() / max(1.0, (label == 0).sum())
    raised = frame > clutter
    return {"lab_frac": float(label.mean()), "lab_frac_crop": float(label[c0:c0 + crop, c0:c0 + crop].mean()),
            "scr_db": float(10 * np.log10(sig / noi)), "peak_over_rms": float(frame.max() / np.sqrt((clutter ** 2).mean())),
            "raised_frac": float(raised.mean()), "raised_in_label": float(raised[label > 0].mean()) if label.sum() else 0.0,
            "mean_gain": float(frame.mean() / clutter.mean())}


def collect(rows):
    """list of per-frame dicts -> {name: float64 [frames]}"""
    keys = rows[0].keys()
    return {k: np.asarray([r[k] for r in rows], dtype=np.float64) for k in keys}


def compare(got, ref, what, z=4.5, rel_floor=0.02, abs_floor=2e-3, skip=()):
    """Means of two sets of per-frame statistics agree within z standard errors of their difference (plus a small floor for
    statistics whose frame-to-frame spread is nearly zero).  -> list of failure strings (empty = agreement)."""
    bad = []
    for k, r in ref.items():
        if k in skip or k not in got:
            continue
        g = np.asarray(got[k], dtype=np.float64)
        r = np.asarray(r, dtype=np.float64)
        se = np.sqrt(g.var(ddof=1) / g.size + r.var(ddof=1) / r.size)
        tol = z * se + rel_floor * abs(r.mean()) + abs_floor
        if not abs(g.mean() - r.mean()) <= tol:
            bad.append(f"{what}: {k}: {g.mean():.5g} vs reference {r.mean():.5g} (tol {tol:.3g}, se {se:.3g})")
    return bad
