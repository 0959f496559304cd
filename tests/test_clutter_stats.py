"""SURVEY 8f row 4 on the CPU: the NumPy statement of the reference's frame recipe (onet_amd/data.py) against statistics of
frames made by the reference's OWN generator functions (tests/golden/clutter_stats.npz, minted by make_golden.py::
run_clutter_stats from KD:469-526 / RG:177-216 after np.random.seed(1981)).  The GPU generator is held to the same fixture by
tests/test_gpu_ops.py::test_gpu_clutter_generator_vs_reference_fixture."""
import os

import numpy as np

import clutter_stats as cs
from onet_amd import data

G = os.path.join(os.path.dirname(__file__), "golden")


def _fixture():
    g = np.load(os.path.join(G, "clutter_stats.npz"))
    return {k: g[k] for k in g.files}


def numpy_recipe_rows(n_frames, seed=77):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = []
    for i in range(n_frames):
        snr = (0, 1, 2)[i % 3]
        amp, tau, s = data.k_clutter_frame(rng, data.FRAME, parts=True)
        frame, lab = data.add_targets(amp, data.target_params(rng, 1, data.FRAME)[0], snr)
        r = {"snr": float(snr)}
        r.update(cs.texture_stats(tau))
        r.update(cs.speckle_stats(s))
        r.update(cs.amplitude_stats(amp))
        r.update(cs.frame_stats(frame, lab, amp))
        rows.append(r)
    return cs.collect(rows)


def test_fixture_is_self_consistent():
    ref = _fixture()
    assert ref["tex_mean"].shape == (24,) and float(ref["root_rule_max_abs_err"]) < 1e-12
    # the reference estimates the polynomial coefficients from each frame's own white samples; this build uses their
    # expectation: within the frame-to-frame scatter of the reference's values
    co, q = ref["poly_coeffs"], ref["quadrature_coeffs"]
    assert np.all(np.abs(co.mean(0) - q) <= 3 * co.std(0, ddof=1) / np.sqrt(len(co)) + 1e-12), (co.mean(0), q)
    assert np.allclose(q, data.acf_poly_coeffs(5.0), rtol=1e-9)
    # Gamma(5) texture fed by a Gaussian field of variance 1.109 (not 1): mean above 5, and the recipe's speckle slope
    assert 5.0 < ref["tex_mean"].mean() < 5.4 and abs(ref["spk_psd_slope"].mean() + 0.6) < 0.02


def test_numpy_recipe_statistics_vs_reference_fixture():
    ref = _fixture()
    got = numpy_recipe_rows(18)
    bad = cs.compare(got, ref, "numpy recipe", skip=("poly_coeffs", "root_rule_max_abs_err", "quadrature_coeffs", "snr"))
    assert not bad, "\n".join(bad)


def test_compare_has_teeth():
    """the comparison rejects the round-2 simplifications: a unit-variance exponential-ACF texture and additive targets"""
    ref = _fixture()
    rng = np.random.Generator(np.random.PCG64(5))
    rows = []
    for _ in range(6):
        n = data.FRAME
        d = np.minimum(np.arange(n), n - np.arange(n)).astype(np.float64)
        psd = np.maximum(np.real(np.fft.fft2(np.exp(-(d[:, None] + d[None, :]) / 10.0))), 0.0)
        g = np.real(np.fft.ifft2(np.fft.fft2(rng.standard_normal((n, n))) * np.sqrt(psd)))
        tau = data.mnlt(g / g.std())
        rows.append(cs.texture_stats(tau))
    bad = cs.compare(cs.collect(rows), ref, "simplified texture")
    assert any("tex_acf" in b for b in bad) and any("tex_mean" in b or "tex_var" in b for b in bad), bad
