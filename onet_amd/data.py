"""Synthetic inputs for benchmarks and tests (SURVEY.md §8d "Synthetic inputs").

K-distributed sea clutter in the build's own NumPy, restating the recipe of the reference's
generators (KD = K_distributed_SeaClutter_Simulation_20210919.py, RG =
Rayleigh_bg_Gaussian_EOT_generator_20230208.py):

  texture  tau = Gamma(nu) field obtained by the memoryless non-linear transform
           gammaincinv(nu, Phi(g)) (KD:83-91) of a spatially coloured unit Gaussian field g;
  speckle  s   = complex white Gaussian field coloured by the PSD |f|^-0.6 (KD:270-297);
  clutter  a   = |s * sqrt(tau)|                                    (KD:519-520);
  targets  20 rotated 2-D Gaussian blobs, centre ~ N(centre,(30,24)), size ~ N((10,18),2), peak
           amplitude sqrt(10^(snr/10) * E[a^2])                     (RG:63-175,198-209);
  frame    400x400 centre-cropped to HxW, per-frame min-max normalised to [0,1] (UT:673-689).

SIMPLIFICATION (stated, allowed by SURVEY §8d): the Gaussian field g is coloured with a fixed
exponential ACF exp(-(|dx|+|dy|)/10) instead of solving the Hermite-polynomial ACF mapping of
KD:121-164 per pixel.  The marginal statistics (K-distributed amplitude, shape nu) are the same;
only the texture's second-order structure differs slightly.  Data content does not affect the
kernels' work (dense fp32 arithmetic), only realism."""
from __future__ import annotations

import numpy as np
from scipy import special as ss


def _coloured_gaussian(rng, n, corr_len=10.0):
    """unit-variance real Gaussian field with ACF ~ exp(-(|dx|+|dy|)/corr_len) via FFT colouring."""
    d = np.minimum(np.arange(n), n - np.arange(n)).astype(np.float64)
    acf = np.exp(-(d[:, None] + d[None, :]) / corr_len)
    psd = np.maximum(np.real(np.fft.fft2(acf)), 0.0)
    white = rng.standard_normal((n, n))
    g = np.real(np.fft.ifft2(np.fft.fft2(white) * np.sqrt(psd)))
    return g / g.std()


def _speckle(rng, n):
    fs = n / 10.0
    f = np.linspace(0.1, fs, num=n, endpoint=True)
    fx, fy = np.meshgrid(f, f)
    psd = np.sqrt(fx ** 2 + fy ** 2) ** (-0.6)
    white = rng.standard_normal((n, n))
    return np.fft.ifft2(np.fft.fft2(white) * np.sqrt(psd))


def k_clutter_frame(rng, n=400, nu=5.0):
    g = _coloured_gaussian(rng, n)
    u = 1.0 - ss.erfc(g / np.sqrt(2.0)) / 2.0                # Phi(g)
    u = np.clip(u, 1e-12, 1 - 1e-12)
    tau = ss.gammaincinv(nu, u)
    return np.abs(_speckle(rng, n) * np.sqrt(tau))


def add_targets(rng, frame, snr_db, n_targets=20):
    n = frame.shape[0]
    power = float(np.mean(frame ** 2))
    peak = np.sqrt(10.0 ** (snr_db / 10.0) * power)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    label = np.zeros_like(frame, dtype=np.float32)
    out = frame.copy()
    for _ in range(n_targets):
        cx, cy = rng.normal(n / 2, 30 * n / 400 * 3), rng.normal(n / 2, 24 * n / 400 * 3)
        sx, sy = np.abs(rng.normal(10, 2)) / 2.5 + 1.0, np.abs(rng.normal(18, 2)) / 2.5 + 1.0
        th = rng.uniform(0, np.pi)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        blob = np.exp(-0.5 * ((xr / sx) ** 2 + (yr / sy) ** 2))
        out = np.maximum(out, 0) + peak * blob
        label = np.maximum(label, (blob > np.exp(-2.0)).astype(np.float32))
    return out, label


def normalise_per_frame(x):
    """(x - min) / (max - min) per frame (UT:673-689)."""
    lo = x.min(axis=(-2, -1), keepdims=True)
    hi = x.max(axis=(-2, -1), keepdims=True)
    return (x - lo) / np.maximum(hi - lo, 1e-12)


def make_clutter_batch(B, H=256, W=256, seed=1981, snr_choices=(0, 1, 2), channels=1, with_labels=False):
    """-> float32 [B, channels, H, W] in [0,1] (and labels [B,H,W] if asked)."""
    n = max(400, H + 16, W + 16)
    rng = np.random.Generator(np.random.PCG64(seed))
    frames, labels = [], []
    for _ in range(B):
        chans = []
        for _c in range(channels):
            f = k_clutter_frame(rng, n)
            f, lab = add_targets(rng, f, float(rng.choice(snr_choices)))
            y0, x0 = (n - H) // 2, (n - W) // 2
            chans.append(f[y0:y0 + H, x0:x0 + W])
        labels.append(lab[y0:y0 + H, x0:x0 + W])
        frames.append(normalise_per_frame(np.stack(chans)))
    X = np.stack(frames).astype(np.float32)
    if with_labels:
        return X, np.stack(labels)
    return X


# ----------------------------------------------------------------------------- on the GPU (csrc/clutter.hip)
def target_params(rng, frames, n, n_targets=20):
    """Host-side draw of the extended targets' geometry, as `add_targets` draws it (RG:198-209 scaled to an n-pixel
    frame): -> float32 [frames, n_targets, 6] = (cx, cy, sigma_x, sigma_y, cos theta, sin theta)."""
    t = np.empty((frames, n_targets, 6), dtype=np.float32)
    for f in range(frames):
        for k in range(n_targets):
            cx, cy = rng.normal(n / 2, 30 * n / 400 * 3), rng.normal(n / 2, 24 * n / 400 * 3)
            sx, sy = np.abs(rng.normal(10, 2)) / 2.5 + 1.0, np.abs(rng.normal(18, 2)) / 2.5 + 1.0
            th = rng.uniform(0, np.pi)
            t[f, k] = (cx, cy, sx, sy, np.cos(th), np.sin(th))
    return t


def make_clutter_batch_gpu(B, H=256, W=256, seed=1981, snr_choices=(0, 1, 2), device="cuda", with_labels=False,
                           n_targets=20, corr_len=10.0, normalise=True):
    """The recipe of `make_clutter_batch` synthesised ON the GPU (white Philox fields, FFT colouring, MNLT, speckle,
    targets, crop, per-frame normalisation): -> float32 [B, 1, H, W] device tensor in [0,1] (and labels [B,H,W]).
    Frames depend on (seed, frame index) only, so each data-parallel rank can make its own shard with its own seed;
    the random streams differ from the NumPy generator's (statistics agree: tests/test_gpu_ops.py)."""
    import torch
    from . import _lib
    from .ops import _p, _stream
    dev = torch.device(device)
    n = int(_lib.load().onet_clutter_frame_size())
    if H > n or W > n:
        raise ValueError(f"make_clutter_batch_gpu: H, W <= {n}")
    rng = np.random.Generator(np.random.PCG64(seed))
    tg = torch.from_numpy(target_params(rng, B, n, n_targets)).to(dev)
    snr = torch.from_numpy(rng.choice(np.asarray(snr_choices, dtype=np.float32), size=B).astype(np.float32)).to(dev)
    ws = torch.empty(int(_lib.load().onet_clutter_ws_bytes(B)) // 4 + 4, dtype=torch.float32, device=dev)
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    lab = torch.empty((B, H, W), dtype=torch.float32, device=dev) if with_labels else None
    _lib.call("onet_clutter_generate", _p(out), _p(lab), _p(tg), _p(snr), n_targets, B, H, W, int(seed) & (2 ** 64 - 1),
              float(corr_len), _p(ws), ws.numel() * 4, _stream())
    if normalise:
        from .metrics import tensor_normal_per_frame
        out = tensor_normal_per_frame(out)
    return (out, lab) if with_labels else out
