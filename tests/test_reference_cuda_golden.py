"""The oracle (oracle/ghr_oracle.c) and the HIP product against OUTPUTS OF THE REFERENCE'S OWN RASTERIZER: the CUDA
sources of /root/reference/ext/diff_gaussian_rasterization_hair/cuda_rasterizer, hipified and compiled for gfx950 by
oracle/Makefile.ref and run on an MI355X by tests/golden/make_reference_cuda_golden.py ->
tests/golden/reference_cuda_golden.npz (inputs, forward outputs, internal state, gradients; twelve cases over modes A /
A_sr / B_sr / B_cov, six of them through rotated / rolled ring cameras).  This is what pins the oracle.

Comparison: integers / indices bit for bit; floats to 1e-6 relative (the reference binary is built with the compiler's
default fp contraction like nvcc's, the oracle without, so single roundings may differ); gradients are sums of ~1e4
fp32 atomics in arbitrary order on the reference side: 2e-5 of the tensor's largest magnitude + 1e-5 relative.
Pixels the oracle flags as fragile (a discrete decision within 2e-5 of its threshold) are excluded from the image
comparison and carry dL/dpixel = 0 in the golden's backward (stored mask), exactly as in tests/test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

from tests import helpers as hp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_cuda_golden.npz")
from tests.golden.make_reference_cuda_golden import CASES, case_tag  # noqa: E402  (workload, mode, camera) incl. rotated / rolled views


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def case_inputs(gold, cfg, mode, cam="front"):
    """The golden's stored inputs as the dict synthetic.raster_inputs() returns (numpy -> torch)."""
    import torch
    t = case_tag(cfg, mode, cam)
    ri = {k: torch.from_numpy(gold[t + "in_" + k]) for k in ("means3D", "colors", "opacities", "cov3D", "conic", "scales",
                                                             "rotations", "bg", "viewmatrix", "projmatrix", "campos")}
    W, H, tx, ty = gold[t + "in_scalars"]
    ri.update(W=int(W), H=int(H), tanfovx=float(tx), tanfovy=float(ty), P=ri["means3D"].shape[0])
    return ri, t


def case_dL(gold, t, cfg, ri):
    from gaussianhaircut_amd.utils import synthetic as syn
    spec = syn.CONFIGS[cfg]
    frag = np.unpackbits(gold[t + "in_dL_mask"])[: ri["W"] * ri["H"]].astype(bool).reshape(ri["H"], ri["W"])
    dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)
    dL[:, frag] = 0.0
    assert abs(np.abs(dL.astype(np.float64)).sum() - float(gold[t + "in_dL_seed_abs_sum"])) < 1e-6 * float(
        gold[t + "in_dL_seed_abs_sum"]), "seeded dL/dout differs from the one the golden was made with"
    return dL, frag


def close(a, b, rel, floor=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= rel * np.abs(b) + floor


def check_state(gold, t, radii, st_depths, st_xy, st_conic_opacity, R, ranges, point_list, n_contrib, final_T, out, frag,
                vis_rows=None):
    """Forward state / outputs of an implementation against the reference's."""
    g = lambda k: gold[t + k]
    np.testing.assert_array_equal(radii, g("radii"))
    vis = g("radii") > 0
    np.testing.assert_array_equal(st_depths[vis].view(np.uint32), g("st_depths")[vis].view(np.uint32))
    assert close(st_xy[vis], g("st_means2D")[vis], 1e-6, 1e-5).all()          # pixel coordinates up to ~1e3
    co, co_ref = st_conic_opacity[vis].astype(np.float64), g("st_conic_opacity")[vis].astype(np.float64)
    # mode A: the conic is an input (bit-identical); mode B: computed in the kernel, where the reference binary
    # contracts a*b+c (2e-6 of the row's largest entry; measured 4e-7)
    assert (np.abs(co - co_ref) <= 2e-6 * np.abs(co_ref).max(axis=1, keepdims=True)).all()
    assert R == int(g("num_rendered"))
    np.testing.assert_array_equal(ranges, g("st_ranges").view(np.uint32))
    np.testing.assert_array_equal(point_list, g("st_point_list").view(np.uint32))
    ok = ~frag.reshape(-1)
    np.testing.assert_array_equal(n_contrib[ok], g("st_n_contrib").view(np.uint32)[ok])
    assert close(final_T[ok], g("st_final_T")[ok], 2e-6, 1e-7).all()
    a, b = out.reshape(10, -1)[:, ok], g("out_color").reshape(10, -1)[:, ok]
    assert close(a, b, 1e-5, 2e-6).all(), np.abs(a - b).max()


def check_grads(gold, t, got, rel=1e-5, of_max=2e-5):
    for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dscales",
              "dL_drotations"):
        a, b = np.asarray(got[k], np.float64).reshape(-1), gold[t + k].astype(np.float64).reshape(-1)
        assert np.isfinite(a).all(), k
        bad = np.abs(a - b) > rel * np.abs(b) + of_max * (np.abs(b).max() if b.size else 0.0) + 1e-30
        assert not bad.any(), (k, int(bad.sum()), float(np.abs(a - b).max()), float(np.abs(b).max()))


@pytest.mark.parametrize("cfg,mode,cam", CASES)
def test_oracle_matches_the_reference_cuda_rasterizer(oracle_mod, gold, cfg, mode, cam):
    ri, t = case_inputs(gold, cfg, mode, cam)
    if cam != "front":
        v = ri["viewmatrix"].numpy()[:3, :3]
        assert np.abs(v - np.eye(3)).max() > 0.3 and np.abs(v - v.T).max() > 0.1, "not a rotated view"
    out, radii, st = hp.oracle_forward(oracle_mod, ri, mode)
    dL, frag = case_dL(gold, t, cfg, ri)
    # the stored fragile mask is the oracle's own (same code, same inputs)
    np.testing.assert_array_equal(st.fragile.astype(bool).reshape(ri["H"], ri["W"]), frag)
    np.testing.assert_array_equal(st.tiles_touched, gold[t + "st_tiles_touched"].view(np.uint32))
    np.testing.assert_array_equal(st.point_offsets, gold[t + "st_point_offsets"].view(np.uint32))
    check_state(gold, t, radii, st.depths, st.xy, st.conic_opacity, st.num_rendered, st.ranges, st.point_list,
                st.n_contrib, st.final_T, out, frag)
    # the reference's 64-bit keys: tile << 32 | depth bits (rasterizer_impl.cu:88-108)
    np.testing.assert_array_equal(st.keys_sorted, gold[t + "st_keys"].view(np.uint64))
    check_grads(gold, t, hp.oracle_backward(oracle_mod, st, ri, dL, mode))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,mode,cam", CASES)
def test_hip_rasterizer_matches_the_reference_cuda_rasterizer(gold, cfg, mode, cam):
    """The product, through the C ABI, against the same reference outputs (no oracle in between)."""
    import torch
    from tests.gpu_helpers import GpuRun, to_dev
    ri, t = case_inputs(gold, cfg, mode, cam)
    dL, frag = case_dL(gold, t, cfg, ri)
    run = GpuRun(to_dev(ri, torch.device("cuda:0")), mode)
    ins = run.inspect()
    ts = ins["tile_start"]
    ranges = np.stack([ts[:-1], ts[1:]], axis=1).astype(np.uint32)
    ranges[ts[:-1] == ts[1:]] = 0
    check_state(gold, t, run.radii.cpu().numpy(), ins["depths"], ins["rec"][:, 0:2], ins["rec"][:, 2:6], run.R, ranges,
                ins["point_list"], ins["n_contrib"], ins["final_T"], run.out.cpu().numpy(), frag)
    check_grads(gold, t, run.backward(torch.from_numpy(dL)), rel=1e-4, of_max=1e-4)


@pytest.mark.parametrize("case", ["tiny@ring13roll", "tiny"])
def test_reference_binary_mode_b_conic_equals_the_reference_pythons_get_conic(gold, case):
    """every "reference binary" number goes through oracle/ref_shim/glm (a builder-written stand-in
    for the un-vendored glm).  This compares two artefacts neither of which involves the oracle or the product: the
    conic the reference's CUDA kernel computed in mode B (scales + rotations -> computeCov3D -> computeCov2D -> inverse,
    forward.cu:74-152,214-239; the strict build for the rolled camera) as stored by make_reference_cuda_golden.py, and
    the conic the reference's own PYTHON computes for the same Gaussians and camera (GaussianModel.get_conic,
    src/scene/gaussian_model.py:303-315; reference_host_golden.npz, made by importing the reference on the CPU).  With
    a full 3x3 view rotation (ring view 13 rolled 20 degrees) any transposed / row-major slip in the shim's mat3
    products would show at O(1).  Bound 2e-6 of the row's largest entry (measured 4.9e-7: fp32 rounding of two
    differently ordered product chains; the Python adds eps = 1e-12 to the determinant)."""
    host = np.load(os.path.join(os.path.dirname(GOLD), "reference_host_golden.npz"))
    t = case + "/B_sr/"
    m = host[case + "/mask"].astype(bool)  # the cull raster_inputs() applied before the op (filter_points)
    # same Gaussians, same camera on both sides
    assert m.sum() == gold[t + "in_scales"].shape[0]
    assert np.abs(gold[t + "in_scales"] - host[case + "/scaling"][m]).max() <= 2e-8
    assert np.array_equal(gold[t + "in_rotations"], host[case + "/rotation"][m])
    assert np.array_equal(gold[t + "in_viewmatrix"].reshape(-1), host[case + "/view"].reshape(-1))
    if case != "tiny":
        assert float(np.abs(host[case + "/view"].reshape(4, 4)[:3, :3] - np.eye(3)).min()) > 1e-3  # every entry of R is non-trivial
    vis = gold[t + "radii"] > 0
    assert vis.sum() > 1900
    bin_conic = gold[t + "st_conic_opacity"][vis, :3]
    py_conic = host[case + "/conic"][m][vis]
    rel = np.abs(bin_conic - py_conic) / np.abs(py_conic).max(axis=1, keepdims=True)
    assert rel.max() <= 2e-6, rel.max()
