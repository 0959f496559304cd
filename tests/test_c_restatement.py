"""The plain-C restatement (oracle/onet_ref.c) agrees with the PyTorch-CPU oracle op by op: two
independent statements of the same reference arithmetic."""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

from oracle import onet_oracle as orc

ODIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def _lib():
    so = os.path.join(ODIR, "libonet_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", ODIR])
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_ops_match_oracle():
    lib = _lib()
    lib.ref_log1pexp.restype = ctypes.c_float
    lib.ref_log1pexp.argtypes = [ctypes.c_float]
    lib.ref_head_loss.restype = ctypes.c_double
    rng = np.random.Generator(np.random.PCG64(4))
    B, Ci, Co, H, W = 2, 5, 7, 9, 11
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, 3, 3)) * 0.2).astype(np.float32)
    z = np.zeros((B, Co, H, W), np.float32)
    lib.ref_conv3x3(_p(x), _p(w), _p(z), B, Ci, Co, H, W)
    np.testing.assert_allclose(z, F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, 1, 1).numpy(), rtol=1e-5, atol=1e-5)

    gamma = (1 + 0.1 * rng.standard_normal(Co)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(Co)).astype(np.float32)
    rm, rv = np.zeros(Co, np.float32), np.ones(Co, np.float32)
    a = np.zeros_like(z)
    lib.ref_bn_relu_train(_p(z), _p(gamma), _p(beta), _p(rm), _p(rv), _p(a), B, Co, H * W, ctypes.c_float(0.1), ctypes.c_float(1e-5))
    trm, trv = torch.zeros(Co), torch.ones(Co)
    ta = F.relu(F.batch_norm(torch.from_numpy(z), trm, trv, torch.from_numpy(gamma), torch.from_numpy(beta), True, 0.1, 1e-5))
    np.testing.assert_allclose(a, ta.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rm, trm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv, trv.numpy(), rtol=1e-5, atol=1e-6)

    y = np.zeros((B, Co, H // 2, W // 2), np.float32)
    lib.ref_maxpool2(_p(a), _p(y), B * Co, H, W)
    assert np.array_equal(y, F.max_pool2d(torch.from_numpy(a), 2).numpy())

    wt = (rng.standard_normal((Co, 3, 2, 2)) * 0.3).astype(np.float32)
    bt = (0.1 * rng.standard_normal(3)).astype(np.float32)
    u = np.zeros((B, 3, 2 * (H // 2), 2 * (W // 2)), np.float32)
    lib.ref_convT2x2(_p(y), _p(wt), _p(bt), _p(u), B, Co, 3, H // 2, W // 2)
    np.testing.assert_allclose(u, F.conv_transpose2d(torch.from_numpy(y), torch.from_numpy(wt), torch.from_numpy(bt), stride=2).numpy(), rtol=1e-5, atol=1e-5)

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "log1pexp.npz"))
    got = np.array([lib.ref_log1pexp(float(v)) for v in g["x"]], np.float32)
    np.testing.assert_allclose(got, g["y"], rtol=2e-6, atol=1e-7)       # vs the REAL reference's table

    C, HW = 64, 40
    Ls = [np.abs(rng.standard_normal((B, C, HW))).astype(np.float32) * s for s in (1.0, 0.3, 0.8, 0.4)]
    S = np.zeros((B, 2, HW), np.float32)
    loss = lib.ref_head_loss(_p(Ls[0]), _p(Ls[1]), _p(Ls[2]), _p(Ls[3]), _p(S), B, C, HW)
    T = [torch.from_numpy(v).reshape(B, C, HW, 1) for v in Ls]
    Vt, Vd = orc.head(T[0], T[1]), orc.head(T[2], T[3])
    So = torch.softmax(torch.cat([Vt, Vd], 1), 1)
    lo = orc.compute_loss(T[0], So[:, 0:1], T[2], So[:, 1:2])
    np.testing.assert_allclose(S, So.reshape(B, 2, HW).numpy(), rtol=1e-5, atol=1e-6)
    assert abs(loss - float(lo)) <= 1e-5 * abs(float(lo))
