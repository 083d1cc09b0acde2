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
