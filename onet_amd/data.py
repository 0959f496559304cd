"""Synthetic inputs for benchmarks and tests (SURVEY.md §8d "Synthetic inputs", §8f row 4).

K-distributed sea clutter with extended targets: the build's own NumPy statement of the FRAME RECIPE of the reference's
generators (KD = K_distributed_SeaClutter_Simulation_20210919.py, RG = Rayleigh_bg_Gaussian_EOT_generator_20230208.py), step by
step and with the recipe's own quirks -- the statistics of these frames (and of `make_clutter_batch_gpu`, csrc/clutter.hip) are
pinned to frames produced by the reference's own functions (tests/golden/clutter_stats.npz, tests/test_clutter_stats.py):

  texture  (KD:469-503 `generate_K_distributed_noise`)
           Gamma-process ACF on the grid xs = ys = linspace(10, n, n):  R_T = 1 + exp(-(x + y) / 10) cos(pi y / 8) / nu  (KD:489);
           mapped to a "Gaussian ACF" R_G by solving, per pixel, c2 r^2 + c1 r + (1 - R_T) = 0 and taking numpy.roots' FIRST root
           (KD:141-164) -- the root of larger magnitude where the roots are real, the one with positive imaginary part where
           they are complex; the coefficients c_n = (E_n / E_0)^2 / (n! 2^n), E_n = E_{x~N(0,1)}[exp(-x^2) H_n(x) g(x)],
           g = gammaincinv(nu, Phi(x)), H_n the physicists' Hermite polynomials (KD:93-139, orders 2, 1, 0).  The reference
           estimates E_n from the frame's own 160 000 white samples (c1, c2 scatter by 1.5 % from frame to frame: measured);
           this module uses their expectation (quadrature): c2 = 0.045805, c1 = 0.045408 for nu = 5;
           g_field = Re ifft2( fft2(white) * sqrt(fft2(R_G)) )  (complex square root; NOT normalised to unit variance: its
           variance is 1.109 for n = 400) and  tau = gammaincinv(nu, Phi(g_field))  (KD:83-91 `mnlt`, KD:499-503);
  speckle  (KD:270-297 `generate_correlated_Gaussian_via_expdecay`)  s = ifft2( fft2(white') * sqrt(F) ), complex,
           F = (fx^2 + fy^2)^-0.3 on fx = fy = linspace(0.1, n / 10, n) (index grid: no Hermitian symmetry);
  clutter  a = | s * sqrt(tau) |  as float32  (KD:519-520, RG:189-190);
  targets  (RG:177-216 `get_k_frame`, RG:63-175 `add_gaussian_template_on_clutter_v3`, swerling type 0)  20 per frame,
           centre ~ N((n/2, n/2), (30, 24)), w ~ N(10, 2), h ~ N(18, 2), theta = 180 U(0,1) used AS RADIANS (RG:204, 46-50),
           sigma = (size / 2 - 0.5) / 2, window half-size int(2.5 sigma + 0.5), un-normalised rotated Gaussian `kgauss`, peak
           amplitude sqrt(10^(snr/10) * mean(a^2)); added ONE AFTER THE OTHER as  bg += (template > bg) * template  (RG:156-158),
           label |= kgauss > max(kgauss) - 2 std(kgauss)  (RG:155,166);
  frame    n = 400 (RG:186) centre-cropped to H x W (RG:302,308), min-max normalised per frame (RG:262, UT:673-689).

Random streams differ from the reference's (its global MT19937 state vs a PCG64 / Philox stream here): only statistics are
comparable, and those are what the fixture pins."""
from __future__ import annotations

import math

import numpy as np
from scipy import special as ss

FRAME = 400          # RG:186 img_sz = (400, 400)
NU = 5.0             # RG:189 gamma_shape = 5


# ----------------------------------------------------------------------------- texture: the ACF mapping (KD:93-164)
_COEFFS = {}


def acf_poly_coeffs(nu=NU):
    """(c2, c1, c0 = 1): coefficients of the polynomial in R_G (KD:121-139, normalised as KD:492) in expectation over the white
    samples.  E_n = int phi(x) exp(-x^2) H_n(x) gammaincinv(nu, Phi(x)) dx by the trapezoidal rule on [-8, 8]."""
    key = float(nu)
    if key not in _COEFFS:
        x = np.linspace(-8.0, 8.0, 160001)
        phi = np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
        g = ss.gammaincinv(nu, np.clip(1.0 - ss.erfc(x / math.sqrt(2.0)) / 2.0, 1e-300, 1.0 - 1e-16))
        herm = (np.ones_like(x), 2.0 * x, 4.0 * x * x - 2.0)
        E = [float(np.sum(phi * np.exp(-x * x) * h * g) * (x[1] - x[0])) for h in herm]
        _COEFFS[key] = tuple((E[n] / E[0]) ** 2 / (math.factorial(n) * 2 ** n) for n in (2, 1, 0))
    return _COEFFS[key]


def gaussian_acf_field(n=FRAME, nu=NU):
    """R_G [n, n] complex (KD:483-497): first root, in numpy.roots' order, of c2 r^2 + c1 r + (1 - R_T) per pixel -- for a
    quadratic that is the eigenvalue LAPACK returns first for the companion matrix: -(c1 + sqrt(D)) / (2 c2) where the
    discriminant D = c1^2 - 4 c2 (1 - R_T) is non-negative, (-c1 + i sqrt(-D)) / (2 c2) where it is negative (checked
    element by element against numpy.roots on the reference's own field: tests/golden/make_golden.py)."""
    c2, c1, _ = acf_poly_coeffs(nu)
    xs = np.linspace(10.0, n, num=n, endpoint=True)
    XS, YS = np.meshgrid(xs, xs)
    r_t = 1.0 + np.exp(-(XS + YS) / 10.0) * np.cos(np.pi * YS / 8.0) / nu
    D = c1 * c1 - 4.0 * c2 * (1.0 - r_t)
    root = np.where(D >= 0, -(c1 + np.sqrt(np.maximum(D, 0.0))) / (2 * c2) + 0j, (-c1 + 1j * np.sqrt(np.maximum(-D, 0.0))) / (2 * c2))
    return root


_FILTERS = {}


def texture_filter(n=FRAME, nu=NU):
    """sqrt(fft2(R_G)) [n, n] complex (KD:500,502)."""
    key = ("t", n, float(nu))
    if key not in _FILTERS:
        _FILTERS[key] = np.sqrt(np.fft.fft2(gaussian_acf_field(n, nu)))
    return _FILTERS[key]


def speckle_filter(n=FRAME):
    """sqrt(F) [n, n], F = (fx^2 + fy^2)^-0.3 on linspace(0.1, n / 10, n) (KD:276-296: L = 10, fs = n / L)."""
    key = ("s", n)
    if key not in _FILTERS:
        f = np.linspace(0.1, n / 10.0, num=n, endpoint=True)
        fx, fy = np.meshgrid(f, f)
        _FILTERS[key] = np.sqrt(np.sqrt(fx ** 2 + fy ** 2) ** (-0.6))
    return _FILTERS[key]


def mnlt(x, nu=NU):
    """KD:83-91: gammaincinv(nu, Phi(x))."""
    return ss.gammaincinv(nu, 1.0 - ss.erfc(x / math.sqrt(2.0)) / 2.0)


def texture_field(rng, n=FRAME, nu=NU):
    g = np.real(np.fft.ifft2(np.fft.fft2(rng.standard_normal((n, n))) * texture_filter(n, nu)))      # KD:499-502
    return mnlt(g, nu)                                                                                 # KD:503


def speckle_field(rng, n=FRAME):
    return np.fft.ifft2(np.fft.fft2(rng.standard_normal((n, n))) * speckle_filter(n))                  # KD:286-297


def k_clutter_frame(rng, n=FRAME, nu=NU, parts=False):
    """|speckle * sqrt(texture)| [n, n] float32 (KD:517-520, RG:190); parts: -> (amplitude, texture, speckle)."""
    tau = texture_field(rng, n, nu)
    s = speckle_field(rng, n)
    a = np.abs(s * np.sqrt(tau)).astype(np.float32)
    return (a, tau, s) if parts else a


# ----------------------------------------------------------------------------- targets (RG:28-60, 63-175, 198-209)
N_TARGETS = 20
TARGET_FIELDS = 8     # lx, ly, kernel_wr, kernel_hr, a, b, c, label threshold


def target_params(rng, frames, n=FRAME, n_targets=N_TARGETS):
    """Draw the extended targets of `frames` frames as RG:198-204 draws them and reduce each to what placing it needs:
    float32 [frames, n_targets, 8] = (lx, ly, kernel_wr, kernel_hr, a, b, c, thr): window origin (RG:77-80), window
    half-sizes (RG:36-37), the quadratic form of the rotated Gaussian (RG:46-57) and the label threshold
    max(kgauss) - 2 std(kgauss) of its window (RG:155)."""
    t = np.empty((frames, n_targets, TARGET_FIELDS), dtype=np.float32)
    for f in range(frames):
        cx, cy = rng.normal(n / 2, 30, n_targets), rng.normal(n / 2, 24, n_targets)
        w, h = rng.normal(10, 2, n_targets), rng.normal(18, 2, n_targets)
        theta = rng.random(n_targets) * 180
        for k in range(n_targets):
            sx, sy = (w[k] / 2 - 0.5) / 2, (h[k] / 2 - 0.5) / 2
            wr, hr = int(np.int32(sx * 2.5 + 0.5)), int(np.int32(sy * 2.5 + 0.5))
            th = -theta[k]
            a = math.cos(th) ** 2 / (2 * sx ** 2) + math.sin(th) ** 2 / (2 * sy ** 2)
            b = -math.sin(2 * th) / (4 * sx ** 2) + math.sin(2 * th) / (4 * sy ** 2)
            c = math.sin(th) ** 2 / (2 * sx ** 2) + math.cos(th) ** 2 / (2 * sy ** 2)
            KX, KY = np.meshgrid(np.arange(-wr, wr + 1), np.arange(-hr, hr + 1))
            kg = np.exp(-(a * KX ** 2 + 2 * b * KX * KY + c * KY ** 2))
            ly, lx = int(cy[k] - (2 * hr + 1 - 1) / 2), int(cx[k] - (2 * wr + 1 - 1) / 2)
            t[f, k] = (lx, ly, wr, hr, a, b, c, kg.max() - 2 * kg.std())
    return t


def add_targets(frame, params, snr_db):
    """RG:189-209 + RG:63-175 on one frame [n, n] float32 with params [n_targets, 8]: -> (frame with targets, label float32)."""
    out = frame.astype(np.float32).copy()
    label = np.zeros(out.shape, dtype=bool)
    peak = math.sqrt(10.0 ** (snr_db / 10.0) * float(np.sum(out.astype(np.float64) ** 2) / out.size))     # RG:192, 89
    for lx, ly, wr, hr, a, b, c, thr in params.astype(np.float64):
        lx, ly, wr, hr = int(lx), int(ly), int(wr), int(hr)
        if ly < 0 or lx < 0 or ly + 2 * hr > out.shape[0] or lx + 2 * wr > out.shape[1]:
            raise ValueError("template location is beyond the image boundaries!")                          # RG:83-84
        KX, KY = np.meshgrid(np.arange(-wr, wr + 1), np.arange(-hr, hr + 1))
        kg = np.exp(-(a * KX ** 2 + 2 * b * KX * KY + c * KY ** 2))
        roi = out[ly:ly + 2 * hr + 1, lx:lx + 2 * wr + 1]
        template = (kg * peak).astype(np.float32)
        roi += (template > roi) * template                                                                  # RG:156-158
        label[ly:ly + 2 * hr + 1, lx:lx + 2 * wr + 1] |= kg > thr                                          # RG:155,166-167
    return out, label.astype(np.float32)


def normalise_per_frame(x):
    """(x - min) / (max - min) per frame (UT:673-689)."""
    lo = x.min(axis=(-2, -1), keepdims=True)
    hi = x.max(axis=(-2, -1), keepdims=True)
    return (x - lo) / np.maximum(hi - lo, 1e-12)


def centre_crop(a, H, W):
    """torchvision.transforms.CenterCrop on the last two axes (RG:302)."""
    n0, n1 = a.shape[-2:]
    y0, x0 = int(round((n0 - H) / 2.0)), int(round((n1 - W) / 2.0))
    return a[..., y0:y0 + H, x0:x0 + W]


def make_clutter_batch(B, H=256, W=256, seed=1981, snr_choices=(0, 1, 2), channels=1, with_labels=False, n_targets=N_TARGETS):
    """-> float32 [B, channels, H, W] in [0,1] (and labels [B,H,W] if asked).  Frames of max(400, H, W) pixels a side
    (400 x 400 for the BASELINE 256 x 256 configurations, as RG:186), centre-cropped; PSNR per frame from `snr_choices`
    (TS:668 trains on PSNR 0, 1, 2).  channels > 1 (the 3 x 512 x 512 tile SHAPE of BASELINE configs[4]): one clutter frame
    per channel."""
    n = max(FRAME, H, W)
    rng = np.random.Generator(np.random.PCG64(seed))
    frames, labels = [], []
    for _ in range(B):
        chans = []
        for _c in range(channels):
            f = k_clutter_frame(rng, n)
            f, lab = add_targets(f, target_params(rng, 1, n, n_targets)[0], float(rng.choice(snr_choices)))
            chans.append(centre_crop(f, H, W))
        labels.append(centre_crop(lab, H, W))
        frames.append(normalise_per_frame(np.stack(chans)))
    X = np.stack(frames).astype(np.float32)
    if with_labels:
        return X, np.stack(labels)
    return X


def make_blob_tiles(B, H=512, W=512, seed=4242, channels=3, n_blobs=2):
    """Stand-in for BASELINE configs[4]'s ZY-3 cloud tiles, which are not in the image (SURVEY 8d "C5": uniform-[0,1] RGB tiles
    with a synthetic bright-blob mask): per tile a dark textured background (0.35 * uniform noise per channel) plus `n_blobs`
    bright Gaussian blobs (sigma 6 % .. 12 % of the tile side, peak 0.55 .. 0.75, the same shape in every channel with a
    per-channel gain of 0.9 .. 1.0), clipped to [0, 1].  -> (X float32 [B, channels, H, W], mask float32 [B, H, W] = blob
    intensity > 0.25, the reference's label dtype).  Depends on (seed, B, H, W, channels, n_blobs) only."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    X = np.empty((B, channels, H, W), dtype=np.float32)
    M = np.empty((B, H, W), dtype=np.float32)
    side = float(min(H, W))
    for b in range(B):
        blob = np.zeros((H, W), dtype=np.float64)
        for _ in range(n_blobs):
            cy, cx = rng.uniform(0.2, 0.8) * H, rng.uniform(0.2, 0.8) * W
            sy, sx = rng.uniform(0.06, 0.12, size=2) * side
            blob = np.maximum(blob, rng.uniform(0.55, 0.75) * np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2)))
        gain = rng.uniform(0.9, 1.0, size=channels)
        bg = 0.35 * rng.random((channels, H, W))
        X[b] = np.clip(bg + gain[:, None, None] * blob[None], 0.0, 1.0).astype(np.float32)
        M[b] = (blob > 0.25).astype(np.float32)
    return X, M


# ----------------------------------------------------------------------------- on the GPU (csrc/clutter.hip)
def _embed_kernel(kern, N):
    """The reference's colouring is a CIRCULAR convolution of white noise on the n-torus with k = ifft2(filter).  The GPU
    generator works on a 2^k torus (N = 512 >= n): place k there with its lags wrapped (index i < n/2 -> lag i, else i - n).
    The texture kernel holds 99.6 % of its energy within 20 pixels of the origin and the window used is n x n, so the field
    inside the window has the reference's covariance up to the kernel's wrap-around tail (< 0.1 % of its energy)."""
    n = kern.shape[0]
    lag = np.where(np.arange(n) < (n + 1) // 2, np.arange(n), np.arange(n) - n) % N
    out = np.zeros((N, N), dtype=kern.dtype)
    out[np.ix_(lag, lag)] = kern
    return out


_GPU_FILTERS = {}


def gpu_filters(N, n=FRAME, nu=NU):
    """(texture, speckle) frequency responses [N, N] complex64 on the GPU generator's N-torus."""
    key = (N, n, float(nu))
    if key not in _GPU_FILTERS:
        kt = np.real(np.fft.ifft2(texture_filter(n, nu)))        # the white field is real: only Re(k) acts (KD:502 takes Re)
        ks = np.fft.ifft2(speckle_filter(n))
        _GPU_FILTERS[key] = (np.fft.fft2(_embed_kernel(kt, N)).astype(np.complex64),
                             np.fft.fft2(_embed_kernel(ks, N)).astype(np.complex64))
    return _GPU_FILTERS[key]


def make_clutter_batch_gpu(B, H=256, W=256, seed=1981, snr_choices=(0, 1, 2), device="cuda", with_labels=False,
                           n_targets=N_TARGETS, normalise=True, parts=False):
    """The recipe of `make_clutter_batch` synthesised ON the GPU (white Philox fields, FFT colouring with the reference's
    two filters, MNLT, speckle, sequential target compositing, centre crop of the 400 x 400 frame, per-frame normalisation):
    -> float32 [B, 1, H, W] device tensor in [0,1] (and labels [B,H,W]).  Frames depend on (seed, frame index) only and are
    bit-reproducible, so each data-parallel rank can make its own shard with its own seed; the random streams differ from the
    NumPy generator's and the reference's (statistics agree with the reference-minted fixture: tests/test_gpu_ops.py).
    parts: -> dict with the full 400 x 400 texture, speckle (complex) and amplitude fields as well (statistics tests)."""
    import torch
    from . import _lib
    from .ops import _p, _stream
    dev = torch.device(device)
    lib = _lib.load()
    N, n = int(lib.onet_clutter_fft_size()), int(lib.onet_clutter_frame_size())
    if n != FRAME or H > n or W > n:
        raise ValueError(f"make_clutter_batch_gpu: H, W <= {n}")
    rng = np.random.Generator(np.random.PCG64(seed))
    tg = torch.from_numpy(target_params(rng, B, n, n_targets) if n_targets else np.zeros((B, 0, TARGET_FIELDS), np.float32)).to(dev)
    snr = torch.from_numpy(rng.choice(np.asarray(snr_choices, dtype=np.float32), size=B).astype(np.float32)).to(dev)
    ft, fs = (torch.view_as_real(torch.from_numpy(f)).contiguous().to(dev) for f in gpu_filters(N, n))
    ws = torch.empty(int(lib.onet_clutter_ws_bytes(B)) // 4 + 4, dtype=torch.float32, device=dev)
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    lab = torch.empty((B, H, W), dtype=torch.float32, device=dev) if with_labels else None
    fields = torch.empty((B, 4, n, n), dtype=torch.float32, device=dev) if parts else None      # texture, Re s, Im s, amplitude
    _lib.call("onet_clutter_generate", _p(out), _p(lab), _p(fields), _p(tg), _p(snr), n_targets, B, H, W, int(seed) & (2 ** 64 - 1),
              _p(ft), _p(fs), _p(ws), ws.numel() * 4, _stream())
    if normalise:
        from .metrics import tensor_normal_per_frame
        out = tensor_normal_per_frame(out)
    if parts:
        return {"frames": out, "labels": lab, "texture": fields[:, 0], "speckle": torch.complex(fields[:, 1], fields[:, 2]),
                "amplitude": fields[:, 3]}
    return (out, lab) if with_labels else out
