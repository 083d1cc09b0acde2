"""CPU parity of the product's device functions (run through tests/hostsim) against the oracle.

This is the `-m "not gpu"` stand-in for the kernel parity tests: same per-Gaussian / per-pixel code the HIP kernels
inline, executed sequentially on the host.  Integer outputs must match the oracle bit for bit."""
import numpy as np
import pytest

from gaussianhaircut_amd.utils import synthetic as syn
from tests import helpers as hp


def _ranges_from_tile_start(ts):
    ranges = np.stack([ts[:-1], ts[1:]], axis=1).astype(np.uint32)
    ranges[ts[:-1] == ts[1:]] = 0  # the reference leaves empty tiles at (0,0) (memset, rasterizer_impl.cu:314)
    return ranges


# rotated / rolled cameras (scene/cameras.py: parity_camera): with the front camera's R = I a transposed view rotation in
# computeCov2D or its backward would be invisible
ROTATED = [("tiny", "A", "ring5"), ("tiny", "B_sr", "ring13roll"), ("tiny", "B_cov", "ring5"),
           ("tiny", "B_cov", "ring13roll"), ("tiny_strands", "A_sr", "ring13roll"), ("ragged", "B_sr", "ring5")]
FRONT = [("tiny", "A"), ("ragged", "A"), ("tiny_strands", "A"), ("tiny", "B_sr"), ("tiny", "B_cov"), ("tiny_strands", "A_sr")]
FWD_CASES = [c + ("front",) for c in FRONT + [("cfg1", "A")]] + ROTATED
BWD_CASES = [c + ("front",) for c in FRONT] + ROTATED


@pytest.mark.parametrize("cfg,mode,cam", FWD_CASES)
def test_forward_matches_oracle(oracle_mod, hostsim, cfg, mode, cam):
    spec = syn.CONFIGS[cfg]
    ri = syn.raster_inputs(spec, cam=cam)
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, mode)
    st = hostsim.forward(ri, mode)
    try:
        # K1: bit-exact per-Gaussian state
        np.testing.assert_array_equal(st["radii"], radii_o)
        vis = radii_o > 0
        np.testing.assert_array_equal(st["depths"][vis].view(np.uint32), st_o.depths[vis].view(np.uint32))
        np.testing.assert_array_equal(st["rec"][vis, 0:2].view(np.uint32), st_o.xy[vis].view(np.uint32))
        np.testing.assert_array_equal(st["rec"][vis, 2:6].view(np.uint32), st_o.conic_opacity[vis].view(np.uint32))
        # binning: identical instance count, tile ranges and sorted lists
        assert st["R"] == st_o.num_rendered
        np.testing.assert_array_equal(_ranges_from_tile_start(st["tile_start"]), st_o.ranges)
        np.testing.assert_array_equal(st["point_list"], st_o.point_list)
        # K7
        frag = st_o.fragile.reshape(-1).astype(bool)
        assert frag.mean() < 2e-3
        ok = ~frag
        np.testing.assert_array_equal(st["n_contrib"][ok], st_o.n_contrib[ok])
        assert hp.image_close(st["final_T"][ok], st_o.final_T[ok]).all()
        H, W = spec.H, spec.W
        close = hp.image_close(st["out"].reshape(10, -1)[:, ok], out_o.reshape(10, -1)[:, ok])
        assert close.all(), "max err %g" % np.abs(st["out"].reshape(10, -1)[:, ok] - out_o.reshape(10, -1)[:, ok]).max()
    finally:
        hostsim.free(st)


@pytest.mark.parametrize("cfg,mode,cam", BWD_CASES)
def test_backward_matches_oracle(oracle_mod, hostsim, cfg, mode, cam):
    spec = syn.CONFIGS[cfg]
    ri = syn.raster_inputs(spec, cam=cam)
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, mode)
    dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)  # O(1) per-pixel gradients
    dL[:, st_o.fragile.astype(bool)] = 0.0
    ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, mode)
    st = hostsim.forward(ri, mode)
    try:
        got = hostsim.backward(st, dL)
        hp.assert_grads_close(got, ref)
        if mode.startswith("B"):
            assert np.abs(ref["dL_dmeans3D"]).sum() > 0
        if mode == "B_sr":
            assert np.abs(ref["dL_dscales"]).sum() > 0 and np.abs(ref["dL_drotations"]).sum() > 0
    finally:
        hostsim.free(st)


def test_xcd_tile_is_a_bijection(hostsim):
    for n in (1, 7, 8, 9, 63, 64, 65, 768, 8160, 8161, 8167):
        assert hostsim.L.ghrsim_xcd_bijective(n) == 1, n


def test_bitonic_network_any_length(hostsim):
    import ctypes
    rng = np.random.default_rng(0)
    for n in list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 1000, 4095, 4096, 4097, 5000]:
        keys = rng.integers(0, 1 << 40, size=max(n, 1), dtype=np.uint64)
        if n > 3:
            keys[: n // 2] = keys[0]  # heavy ties
        ref = np.sort(keys[:n])
        assert hostsim.L.ghrsim_bitonic(ctypes.c_void_p(keys.ctypes.data), ctypes.c_uint32(n)) == 1
        np.testing.assert_array_equal(keys[:n], ref)


@pytest.mark.parametrize("r", [1, 2, 3, 4])
def test_register_blocked_bitonic_network_any_length(hostsim, r):
    """k_tile_sort's wave path: a thread owns 2^r keys between LDS round trips and runs up to r steps on them in registers
    (csrc/ghr_binning.h bitonic_blocked).  Every length, ties included, must come out as np.sort leaves it."""
    import ctypes
    rng = np.random.default_rng(r)
    for n in list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 511, 512, 513, 700, 1000, 1023, 1024, 1025, 3000]:
        keys = rng.integers(0, 1 << 40, size=max(n, 1), dtype=np.uint64)
        if n > 3 and n % 3 == 0:
            keys[: n // 2] = keys[0]  # heavy ties
        ref = np.sort(keys[:n])
        assert hostsim.L.ghrsim_bitonic_blocked(ctypes.c_void_p(keys.ctypes.data), ctypes.c_uint32(n), r, 64) == 1, n
        np.testing.assert_array_equal(keys[:n], ref)


def test_alpha_bbox_is_conservative(hostsim):
    """The culling box AND the per-cell box+ellipse test of k_render_fwd / k_render_bwd must keep every pixel the per-pixel alpha test accepts,
    including needle-like conics, opacities at the 1/255 threshold, huge and degenerate splats."""
    import ctypes
    rng = np.random.default_rng(5)
    n = 400
    rec = np.zeros((n, 16), np.float32)
    W, H = 96, 80
    rec[:, 0] = rng.uniform(-20, W + 20, n)
    rec[:, 1] = rng.uniform(-20, H + 20, n)
    # random SPD conics over 6 decades of scale and strong anisotropy
    th = rng.uniform(0, np.pi, n)
    l1 = 10 ** rng.uniform(-4, 0.5, n)
    l2 = l1 * 10 ** rng.uniform(0, 3, n)
    c, s = np.cos(th), np.sin(th)
    rec[:, 2] = l1 * c * c + l2 * s * s
    rec[:, 3] = (l1 - l2) * c * s
    rec[:, 4] = l1 * s * s + l2 * c * c
    rec[:, 5] = np.concatenate([rng.uniform(0.002, 1.0, n - 60), np.full(20, 1 / 255), np.full(20, np.float32(1 / 255) * 1.0005),
                                np.full(10, 0.99), np.full(10, 1.0)])
    # a few degenerate conics (not positive definite / zero)
    rec[:5, 3] = 10.0
    rec[5:8, 2:5] = 0.0
    assert hostsim.L.ghrsim_bbox_violations(ctypes.c_void_p(rec.ctypes.data), n, W, H) == 0
