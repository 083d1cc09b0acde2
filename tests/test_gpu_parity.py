"""`-m gpu` parity tests proper: the HIP path, called through the C ABI, against the oracle on the same seeded inputs.

Integer / index outputs are compared bit for bit; floats within 1e-4 (rules in tests/helpers.py)."""
import numpy as np
import pytest
import torch

from gaussianhaircut_amd.utils import synthetic as syn
from tests import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def _ranges(ts):
    r = np.stack([ts[:-1], ts[1:]], axis=1).astype(np.uint32)
    r[ts[:-1] == ts[1:]] = 0
    return r


def _check_forward(run, out_o, radii_o, st_o, frag_limit=2e-3):
    ins = run.inspect()
    np.testing.assert_array_equal(run.radii.cpu().numpy(), radii_o)
    vis = radii_o > 0
    np.testing.assert_array_equal(ins["depths"][vis].view(np.uint32), st_o.depths[vis].view(np.uint32))
    np.testing.assert_array_equal(ins["rec"][vis, 0:2].view(np.uint32), st_o.xy[vis].view(np.uint32))
    np.testing.assert_array_equal(ins["rec"][vis, 2:6].view(np.uint32), st_o.conic_opacity[vis].view(np.uint32))
    assert run.R == st_o.num_rendered
    np.testing.assert_array_equal(_ranges(ins["tile_start"]), st_o.ranges)
    np.testing.assert_array_equal(ins["point_list"], st_o.point_list)
    frag = st_o.fragile.reshape(-1).astype(bool)
    assert frag.mean() < frag_limit
    ok = ~frag
    np.testing.assert_array_equal(ins["n_contrib"][ok], st_o.n_contrib[ok])
    assert hp.image_close(ins["final_T"][ok], st_o.final_T[ok]).all()
    got = run.out.cpu().numpy().reshape(10, -1)[:, ok]
    ref = out_o.reshape(10, -1)[:, ok]
    assert np.isfinite(got).all()
    close = hp.image_close(got, ref)
    assert close.all(), "%d px-channels off, max err %g" % ((~close).sum(), np.abs(got - ref).max())
    # fragile pixels may differ, but only boundedly (one splat more or less)
    return ins


# cameras: scene/cameras.py:parity_camera -- the front camera has R = I, under which a transposed view rotation in K1 mode B
# (forward.cu:74-113) or K9 (backward.cu:144-274) would go unnoticed; "ring5" / "ring13roll" are rotated / rolled ring views
CASES = [(c, m, "front") for c, m in [("tiny", "A"), ("ragged", "A"), ("tiny_strands", "A"), ("tiny", "B_sr"),
                                      ("tiny", "B_cov"), ("tiny_strands", "A_sr"), ("cfg1", "A"), ("cfg1", "B_sr")]] + \
        [("tiny", "A", "ring5"), ("tiny", "B_sr", "ring13roll"), ("tiny", "B_cov", "ring5"), ("tiny", "B_cov", "ring13roll"),
         ("tiny_strands", "A_sr", "ring13roll"), ("ragged", "B_sr", "ring5"), ("cfg1", "A", "ring13roll"),
         ("cfg1", "B_sr", "ring5"), ("cfg1", "B_cov", "ring13roll")]


@pytest.mark.parametrize("cfg,mode,cam", CASES)
def test_forward_backward_vs_oracle(oracle_mod, dev, cfg, mode, cam):
    from tests.gpu_helpers import GpuRun, to_dev
    spec = syn.CONFIGS[cfg]
    ri = syn.raster_inputs(spec, cam=cam)
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, mode)
    run = GpuRun(to_dev(ri, dev), mode)
    _check_forward(run, out_o, radii_o, st_o)
    dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)
    dL[:, st_o.fragile.astype(bool)] = 0.0
    ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, mode)
    got = run.backward(torch.from_numpy(dL))
    hp.assert_grads_close(got, ref)


@pytest.mark.parametrize("mode,cam", [("A", "front"), ("B_sr", "front"), ("A", "ring5"), ("B_sr", "ring13roll"),
                                      ("B_cov", "ring13roll")])
def test_cfg2_full_size_vs_oracle(oracle_mod, dev, mode, cam):
    """BASELINE.json configs[1]: 100k Gaussians, 1920x1080, fwd+bwd vs the oracle at full size (tol 1e-4), in pipeline
    mode (A) and with the covariance computed in the kernel from scales + rotations (B_sr: K9 / K10 / cov3D backward) or
    from a given 3D covariance (B_cov); through the front camera and through rotated / rolled ring views."""
    from tests.gpu_helpers import GpuRun, to_dev
    spec = syn.CONFIGS["cfg2"]
    ri = syn.raster_inputs(spec, cam=cam)
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, mode)
    run = GpuRun(to_dev(ri, dev), mode, debug=False)
    _check_forward(run, out_o, radii_o, st_o)
    dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)
    dL[:, st_o.fragile.astype(bool)] = 0.0
    ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, mode)
    got = run.backward(torch.from_numpy(dL))
    hp.assert_grads_close(got, ref)


def test_cfg3_full_size_properties_and_oracle(oracle_mod, dev):
    """BASELINE.json configs[2] shape: 500k strand-aligned Gaussians at 1080p.  Oracle comparison plus the
    size-independent properties: per-tile sortedness, count conservation, determinism, linearity of the backward."""
    from tests.gpu_helpers import GpuRun, to_dev
    spec = syn.CONFIGS["cfg3"]
    ri = syn.raster_inputs(spec)
    rid = to_dev(ri, dev)
    run = GpuRun(rid, "A", debug=False)
    ins = run.inspect()
    ts = ins["tile_start"].astype(np.int64)
    assert ts[0] == 0 and ts[-1] == run.R and (np.diff(ts) >= 0).all()
    # rect areas sum to R (checksum of checksums)
    rects = ins["rects"]
    area = ((rects[:, 0] >> 16).astype(np.int64) - (rects[:, 0] & 0xffff)) * \
           ((rects[:, 1] >> 16).astype(np.int64) - (rects[:, 1] & 0xffff))
    assert area.sum() == run.R
    # gradient slots: the [base, base + area) ranges of the visible Gaussians partition [0, R)
    vis = area > 0
    base = rects[:, 2].astype(np.int64) + rects[:, 3].astype(np.int64)
    order = np.argsort(base[vis], kind="stable")
    b, a_ = base[vis][order], area[vis][order]
    assert b[0] == 0 and (b[1:] == b[:-1] + a_[:-1]).all() and b[-1] + a_[-1] == run.R
    # every tile list is sorted by (depth bits, idx) and refers to Gaussians whose rect contains the tile
    keys = ins["keys"]
    seg = np.repeat(np.arange(len(ts) - 1), np.diff(ts))
    same = seg[1:] == seg[:-1]
    assert (keys[1:][same] > keys[:-1][same]).all()
    assert (ins["point_list"] == (keys & np.uint64(0xffffffff)).astype(np.uint32)).all()
    assert (ins["n_contrib"] <= np.repeat(np.diff(ts), 1).max()).all()
    assert ((ins["final_T"] >= 0) & (ins["final_T"] <= 1)).all()
    # determinism / idempotence of the forward
    run2 = GpuRun(rid, "A", debug=False)
    assert torch.equal(run.out, run2.out) and torch.equal(run.radii, run2.radii)
    np.testing.assert_array_equal(ins["point_list"], run2.inspect()["point_list"])
    # oracle at full size
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, "A")
    _check_forward(run, out_o, radii_o, st_o)
    dL = syn.grad_image(spec, 303).numpy() * (spec.H * spec.W)
    dL[:, st_o.fragile.astype(bool)] = 0.0
    ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, "A")
    g1 = run.backward(torch.from_numpy(dL))
    hp.assert_grads_close(g1, ref)
    # linearity in dL/dpixel: bwd(2 dL) == 2 bwd(dL) up to fp32 atomic reordering
    g2 = run.backward(torch.from_numpy(2 * dL))
    hp.assert_grads_close({k: v for k, v in g2.items()}, {k: 2 * v for k, v in g1.items()})


def _manual_inputs(dev, xyz, scales, opac, W=64, H=48, colors=None):
    """A few hand-placed isotropic Gaussians in front of the SURVEY camera (mode B inputs)."""
    from gaussianhaircut_amd.scene.cameras import make_camera
    import math
    cam = make_camera(W, H, device="cpu")
    P = xyz.shape[0]
    rot = torch.zeros(P, 4)
    rot[:, 0] = 1
    colors = colors if colors is not None else torch.rand(P, 10, generator=torch.Generator().manual_seed(5))
    return dict(P=P, W=W, H=H, means3D=xyz.float(), means2D=torch.zeros(P, 3), colors=colors.float(),
                opacities=opac.reshape(P, 1).float(), cov3D=torch.zeros(P, 6), conic=torch.zeros(P, 3),
                scales=scales.float(), rotations=rot, bg=syn.background(), viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, tanfovx=math.tan(float(cam.FoVx) * 0.5),
                tanfovy=math.tan(float(cam.FoVy) * 0.5), campos=cam.camera_center)


def _run_manual(oracle_mod, dev, ri, check_bwd=True):
    from tests.gpu_helpers import GpuRun, to_dev
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, "B_sr")
    run = GpuRun(to_dev(ri, dev), "B_sr")
    _check_forward(run, out_o, radii_o, st_o, frag_limit=0.05)
    if check_bwd and ri["P"] > 0:
        g = torch.Generator().manual_seed(11)
        dL = torch.randn(10, ri["H"], ri["W"], generator=g).numpy()
        dL[:, st_o.fragile.astype(bool)] = 0
        ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, "B_sr")
        hp.assert_grads_close(run.backward(torch.from_numpy(dL)), ref)
    return run, st_o


def test_edge_empty_input(dev):
    from gaussianhaircut_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    ri = _manual_inputs(dev, torch.zeros(0, 3), torch.zeros(0, 3), torch.zeros(0))
    rs = GaussianRasterizationSettings(ri["H"], ri["W"], ri["tanfovx"], ri["tanfovy"], ri["bg"].to(dev), 1.0,
                                       ri["viewmatrix"].to(dev), ri["projmatrix"].to(dev), 3, ri["campos"].to(dev),
                                       True, False)
    e = torch.zeros(0, 3, device=dev)
    color, radii = GaussianRasterizer(rs)(means3D=e, means2D=e, opacities=torch.zeros(0, 1, device=dev),
                                          colors_precomp=torch.zeros(0, 10, device=dev), scales=e,
                                          rotations=torch.zeros(0, 4, device=dev))
    assert color.shape == (10, ri["H"], ri["W"]) and radii.numel() == 0
    assert (color == 0).all()  # the reference returns the zero-filled image when P == 0 (rasterize_points.cu:70,87)


def test_edge_all_culled_and_near_plane(oracle_mod, dev):
    # behind the camera, inside the near band (view z = 0.125 <= 0.2: culled), just past it (0.25), well in front
    z = torch.tensor([-10.0, -3.875, -3.75, -3.0])  # camera sits at z = -4 looking +z => view z = z + 4 (exact)
    xyz = torch.stack([torch.zeros(4), torch.zeros(4), z], dim=1)
    ri = _manual_inputs(dev, xyz, torch.full((4, 3), 0.05), torch.full((4,), 0.8))
    run, st = _run_manual(oracle_mod, dev, ri)
    r = run.radii.cpu().numpy()
    assert r[0] == 0 and r[1] == 0 and r[2] > 0 and r[3] > 0
    ri0 = _manual_inputs(dev, xyz[:2], torch.full((2, 3), 0.05), torch.full((2,), 0.8))
    run0, st0 = _run_manual(oracle_mod, dev, ri0, check_bwd=True)
    assert run0.R == 0
    assert torch.allclose(run0.out, ri0["bg"].to(dev)[:, None, None].expand_as(run0.out))


def test_edge_borders_huge_and_ragged_image(oracle_mod, dev):
    # Gaussians on the image border / outside it, one covering the whole frame; 70x37 is not a multiple of 16
    xyz = torch.tensor([[0.0, 0.0, 0.0], [1.4, 0.0, 0.0], [-1.45, 1.0, 0.0], [0.0, -1.05, 0.3], [5.0, 5.0, 0.0],
                        [0.2, 0.1, -1.0]])
    scales = torch.tensor([[3.0, 3.0, 3.0], [0.05, 0.2, 0.05], [0.1, 0.1, 0.1], [0.02, 0.02, 0.02],
                           [0.1, 0.1, 0.1], [0.3, 0.01, 0.01]])
    ri = _manual_inputs(dev, xyz, scales, torch.tensor([0.3, 0.9, 0.7, 1.0, 0.5, 0.99]), W=70, H=37)
    run, st = _run_manual(oracle_mod, dev, ri)
    assert run.R > 0


@pytest.mark.parametrize("P", [1100, 1300, 2100, 3300, 4090, 5000, 20000, 40000])
def test_edge_depth_ties_and_long_tile_list(oracle_mod, dev, P):
    """> 1024 / > 4096 instances in ONE tile with many exactly equal depths: exercises the dense-tile sorts (k_tile_sort_mid,
    round 6: lists of 1025 .. 4096 keys, 512 threads on the register-blocked network; k_tile_sort_big: one LDS block at 5000,
    three / five blocks with global flip and disperse steps at 20000 / 40000) and the tie-break by ascending Gaussian index;
    also the > 1024-instance path of the backward render kernel."""
    g = torch.Generator().manual_seed(3)
    xyz = torch.zeros(P, 3)
    xyz[:, :2] = (torch.rand(P, 2, generator=g) - 0.5) * 0.05
    xyz[:, 2] = torch.randint(0, 7, (P,), generator=g).float() * 0.01  # 7 distinct depths => massive ties
    ri = _manual_inputs(dev, xyz, torch.full((P, 3), 0.004), torch.full((P,), 0.02 * 5000 / P), W=64, H=64)
    run, st = _run_manual(oracle_mod, dev, ri)
    counts = np.diff(run.inspect()["tile_start"].astype(np.int64))
    assert counts.max() > 4096 * (P // 5000)
    assert counts.max() > 1024 and counts.sum() >= 256 * counts.size, "not a dense scene: the dense-tile kernels were not launched"
    if P < 4096:
        assert counts.max() <= 4096


def test_analytic_single_gaussian(dev):
    """One isotropic Gaussian at the image centre: alpha(d) = min(.99, o * exp(-d^2 / (2 s^2))) in closed form."""
    from tests.gpu_helpers import GpuRun, to_dev
    import math
    W = H = 65  # centre pixel 32 is exactly the projection of the origin: ndc 0 -> ((0+1)*65-1)/2 = 32
    o, s_world = 0.8, 0.1
    ri = _manual_inputs(dev, torch.zeros(1, 3), torch.full((1, 3), s_world), torch.tensor([o]), W=W, H=H,
                        colors=torch.ones(1, 10))
    run = GpuRun(to_dev(ri, dev), "B_sr")
    focal = H / (2 * ri["tanfovy"])
    var = (s_world * focal / 4.0) ** 2 + 0.3  # cov2D = (f s / z)^2 + 0.3 low-pass
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs - 32.0) ** 2 + (ys - 32.0) ** 2
    alpha = np.minimum(0.99, o * np.exp(-0.5 * d2 / var))
    alpha[alpha < 1 / 255] = 0
    radius = math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))  # forward.cu:255: max(0.1, mid^2 - det) under the root
    out = run.out.cpu().numpy()
    # inside the splat's tile rect the image is alpha * 1 (+ T * bg, bg = 0 for channel 0)
    rect_ok = (np.abs(xs - 32) <= radius) & (np.abs(ys - 32) <= radius)
    assert np.abs(out[0][rect_ok] - alpha[rect_ok]).max() < 2e-5
    assert int(run.radii.cpu()[0]) == radius


def test_api_rejects_cpu_tensors_and_bad_args(dev):
    from gaussianhaircut_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    ri = _manual_inputs(dev, torch.zeros(2, 3), torch.full((2, 3), 0.1), torch.full((2,), 0.5))
    rs = GaussianRasterizationSettings(ri["H"], ri["W"], ri["tanfovx"], ri["tanfovy"], ri["bg"].to(dev), 1.0,
                                       ri["viewmatrix"].to(dev), ri["projmatrix"].to(dev), 3, ri["campos"].to(dev),
                                       True, False)
    r = GaussianRasterizer(rs)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=ri["means3D"], means2D=ri["means2D"], opacities=ri["opacities"], colors_precomp=ri["colors"],
          scales=ri["scales"], rotations=ri["rotations"])
    d = {k: v.to(dev) for k, v in ri.items() if isinstance(v, torch.Tensor)}
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(means3D=d["means3D"], means2D=d["means2D"], opacities=d["opacities"], scales=d["scales"],
          rotations=d["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        r(means3D=d["means3D"], means2D=d["means2D"], opacities=d["opacities"], colors_precomp=d["colors"])
    with pytest.raises(RuntimeError, match="provide precomputed Gaussian colors"):
        r(means3D=d["means3D"], means2D=d["means2D"], opacities=d["opacities"], shs=torch.zeros(2, 16, 3, device=dev),
          scales=d["scales"], rotations=d["rotations"])
    vis = r.markVisible(torch.tensor([[0.0, 0, 0], [0, 0, -10.0]], device=dev))
    assert vis.tolist() == [True, False]


def test_render_api_end_to_end_grads(oracle_mod, dev):
    """render() -> loss -> backward through the PyTorch projection graph: leaf gradients are finite, and the image
    equals the oracle driven with the same rasterizer-level inputs."""
    from gaussianhaircut_amd.gaussian_renderer import render
    from gaussianhaircut_amd.trainer import PIPE
    spec = syn.CONFIGS["tiny"]
    model = syn.make_model(spec, dev)
    cam = syn.make_view(spec, dev)
    pkg = render(cam, model, PIPE, syn.background(dev))
    assert pkg["render"].shape == (3, spec.H, spec.W) and pkg["mask"].shape == (2, spec.H, spec.W)
    assert pkg["orient_angle"].shape == (1, spec.H, spec.W) and pkg["radii"].shape == (spec.P,)
    loss = pkg["render"].mean() + pkg["mask"].mean() + (pkg["orient_angle"] * pkg["orient_conf"]).mean()
    loss.backward()
    for p in model.leaf_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert pkg["viewspace_points"].grad is not None
    ri = syn.raster_inputs(spec, "cpu", syn.make_model(spec, "cpu"), syn.make_view(spec, "cpu"))
    out_o, _, st_o = hp.oracle_forward(oracle_mod, ri, "A")
    ok = ~st_o.fragile.astype(bool)
    img = pkg["render"].detach().cpu().numpy()
    # GPU-side torch projection vs CPU-side torch projection differ by rounding: loose tolerance, same picture
    assert np.abs(img[:, ok] - out_o[:3][:, ok]).mean() < 1e-3


def test_non_finite_feature_of_a_non_contributing_gaussian_does_not_leak(oracle_mod, dev):
    """The branch-free backward step evaluates pairs that do not contribute with alpha = 0; a non-finite feature of
    such a Gaussian (here: one whose opacity is below 1/255, so it never contributes anywhere) must not poison the
    gradients of the others (the reference skips the pair, backward.cu:494-505)."""
    from tests.gpu_helpers import GpuRun, to_dev
    spec = syn.CONFIGS["tiny"]
    ri = syn.raster_inputs(spec)
    bad = torch.arange(0, ri["P"], 7)
    ri["opacities"] = ri["opacities"].clone()
    ri["opacities"][bad] = 1e-4
    clean = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in ri.items()}
    ri["colors"] = ri["colors"].clone()
    ri["colors"][bad, 2] = float("inf")
    ri["colors"][bad, 5] = float("nan")
    dL = syn.grad_image(spec, 101) * (spec.H * spec.W)
    outs = []
    for inp in (clean, ri):
        run = GpuRun(to_dev(inp, dev), "A", debug=False)
        g = run.backward(dL)
        outs.append((run.out.cpu().numpy(), g))
    assert np.isfinite(outs[1][0]).all() and np.array_equal(outs[0][0], outs[1][0])
    keep = np.ones(ri["P"], bool)
    keep[bad.numpy()] = False
    for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors"):
        a, b = outs[1][1][k][keep], outs[0][1][k][keep]
        assert np.isfinite(a).all(), k
        assert np.allclose(a, b, rtol=1e-4, atol=1e-6 * np.abs(b).max()), k


def test_deterministic_backward_is_bit_reproducible(oracle_mod, dev):
    """ghr_set_deterministic (optional per-tile-ordered backward): with it the gradients of repeated
    backward passes over the same state are bit-identical (cfg2-like blobs overlap heavily: many cells add into every
    gradient line), and they are the same gradients (oracle tolerance)."""
    from gaussianhaircut_amd import _lib
    from tests.gpu_helpers import GpuRun, to_dev
    spec = syn.CONFIGS["cfg1"]
    ri = syn.raster_inputs(spec)
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, "A")
    run = GpuRun(to_dev(ri, dev), "A")
    dL = syn.grad_image(spec, 7).numpy() * (spec.H * spec.W)
    dL[:, st_o.fragile.astype(bool)] = 0.0
    ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, "A")
    L = _lib.lib()
    assert L.ghr_set_deterministic(1) == 0
    try:
        runs = [run.backward(torch.from_numpy(dL)) for _ in range(4)]
    finally:
        assert L.ghr_set_deterministic(0) == 1
    hp.assert_grads_close(runs[0], ref)
    for other in runs[1:]:
        for k in runs[0]:
            assert np.array_equal(np.asarray(runs[0][k]).view(np.uint32), np.asarray(other[k]).view(np.uint32)), k


def test_op_backward_twice_on_one_forward(oracle_mod, dev):
    """The op hands the backward's scratch to the forward pass, whose tile sort zeroes the gradient lines
    (ghr_forward_stage2 grad_scratch): only the FIRST backward over that state may rely on it.  retain_graph + a second
    backward must give the same gradients again (the render kernel then zeroes for itself), both equal to the oracle's."""
    from gaussianhaircut_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    spec = syn.CONFIGS["tiny"]
    ri = syn.raster_inputs(spec)
    out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, "B_sr")
    dL = syn.grad_image(spec, 5).numpy() * (spec.H * spec.W)
    dL[:, st_o.fragile.astype(bool)] = 0.0
    ref = hp.oracle_backward(oracle_mod, st_o, ri, dL, "B_sr")
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in ri.items()}
    rs = GaussianRasterizationSettings(ri["H"], ri["W"], ri["tanfovx"], ri["tanfovy"], d["bg"], 1.0, d["viewmatrix"],
                                       d["projmatrix"], 3, d["campos"], True, False)
    leaves = {k: d[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    color, radii = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=d["means2D"], opacities=leaves["opacities"],
                                          colors_precomp=leaves["colors"], scales=leaves["scales"],
                                          rotations=leaves["rotations"])
    w = torch.from_numpy(dL).to(dev)
    grads = []
    for rep in range(3):
        for t in leaves.values():
            t.grad = None
        (color * w).sum().backward(retain_graph=True)
        torch.cuda.synchronize()
        grads.append({k: t.grad.detach().cpu().numpy().copy() for k, t in leaves.items()})
    names = dict(means3D="dL_dmeans3D", colors="dL_dcolors", opacities="dL_dopacity", scales="dL_dscales",
                 rotations="dL_drotations")
    for g in grads:
        hp.assert_grads_close({names[k]: v.reshape(ref[names[k]].shape) for k, v in g.items()},
                              {names[k]: ref[names[k]] for k in g})


def test_two_states_sharing_one_scratch_backward_in_reverse_order(oracle_mod, dev):
    """VERDICT r2 weak #8 / ADVICE: the library keeps no record of buffers (ABI 13: `prezeroed` is an explicit argument of
    the backward entry points).  Two states share ONE gradient scratch, as include/ghr.h allows: F1(S), F2(S), B2(S),
    B1(S).  By the sharing rule only B2 -- the state whose stage 2 was the last to be handed S, first backward to touch
    it -- may pass prezeroed = 1; B1 passes 0 and zero-fills for itself.  Both must equal the oracle (with the old host
    map B1 skipped its zero-fill and accumulated on top of B2's lines)."""
    from tests.gpu_helpers import GpuRun, to_dev
    specs = [syn.CONFIGS["tiny"], syn.CONFIGS["ragged"]]
    ris = [syn.raster_inputs(s) for s in specs]
    refs, dLs = [], []
    for spec, ri in zip(specs, ris):
        out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, "A")
        dL = syn.grad_image(spec, 77).numpy() * (spec.H * spec.W)
        dL[:, st_o.fragile.astype(bool)] = 0.0
        refs.append(hp.oracle_backward(oracle_mod, st_o, ri, dL, "A"))
        dLs.append(dL)
    # one scratch large enough for either state, poisoned so that a missing zero-fill cannot pass by luck
    probe = [GpuRun(to_dev(ri, dev), "A") for ri in ris]
    rows = max(p.R for p in probe)
    S = torch.full((rows, 16), float("nan"), dtype=torch.float32, device=dev)
    r1 = GpuRun(to_dev(ris[0], dev), "A", scratch=S)
    r2 = GpuRun(to_dev(ris[1], dev), "A", scratch=S)
    g2 = r2.backward(torch.from_numpy(dLs[1]), scratch=S, prezeroed=1)
    g1 = r1.backward(torch.from_numpy(dLs[0]), scratch=S, prezeroed=0)
    hp.assert_grads_close(g2, refs[1])
    hp.assert_grads_close(g1, refs[0])
    # and again over the same states (a second backward never claims prezeroed)
    hp.assert_grads_close(r2.backward(torch.from_numpy(dLs[1]), scratch=S, prezeroed=0), refs[1])


def test_deterministic_mode_fails_loudly_where_it_cannot_be_honoured(dev):
    """ghr_set_deterministic(1) used to fall back to the unordered walk without a word when the ordered kernel's 32-bit
    offsets do not reach.  The size check is on the host, so a fake row count is enough to see the error."""
    import ctypes
    from gaussianhaircut_amd import _lib
    L = _lib.lib()
    assert L.ghr_set_deterministic(1) == 0
    try:
        one = torch.zeros(64, dtype=torch.float32, device=dev)
        rows = (1 << 26) + 256  # rows * 64 B >= 4 GiB
        p = ctypes.c_void_p(one.data_ptr())
        rc = L.ghr_render_backward(None, rows, 16, 16, 1, p, p, p, p, p, p, 0)
        assert rc == _lib.GHR_E_INVALID and b"deterministic" in L.ghr_last_error()
    finally:
        assert L.ghr_set_deterministic(0) == 1


def test_fallback_gradient_walk_vs_oracle(oracle_mod, dev, monkeypatch):
    """k_render_bwd (round 1's cell-group form) stays in the library as the fallback beyond the 32-bit byte offsets of
    k_render_bwd_cells (>= 2^26 rows / instances: not reachable in a test); GHR_K8=cell selects it at any size."""
    from tests.gpu_helpers import GpuRun, to_dev
    monkeypatch.setenv("GHR_K8", "cell")
    for cfg, mode in (("cfg1", "A"), ("tiny", "B_sr")):
        spec = syn.CONFIGS[cfg]
        ri = syn.raster_inputs(spec)
        out_o, radii_o, st_o = hp.oracle_forward(oracle_mod, ri, mode)
        run = GpuRun(to_dev(ri, dev), mode)
        dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)
        dL[:, st_o.fragile.astype(bool)] = 0.0
        hp.assert_grads_close(run.backward(torch.from_numpy(dL)), hp.oracle_backward(oracle_mod, st_o, ri, dL, mode))


@pytest.mark.gpu
def test_rasterizer_op_recycles_its_image_workspace_and_changes_nothing():
    """Round 6: the GaussianRasterizer op takes its image workspace from the pool of completed passes of the same size and
    stream (ghr_view_args.img_ws_recycled: stage 1 then skips the per-tile counters' zero-fill launch).  Three calls in a row
    with different inputs against the same calls with recycling off: image, radii and instance count bit for bit."""
    import gaussianhaircut_amd.diff_gaussian_rasterization as dgr
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["cfg1"]
    outs = {}
    for recycle in (False, True):
        dgr.RECYCLE_IMG_WS = recycle
        dgr._ImgLease._pools.clear()
        got, seen = [], []
        for camname in ("front", "ring5", "ring13roll"):
            ri = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in syn.raster_inputs(spec, "cpu", cam=camname).items()}
            rs = dgr.GaussianRasterizationSettings(image_height=spec.H, image_width=spec.W, tanfovx=ri["tanfovx"],
                                                   tanfovy=ri["tanfovy"], bg=ri["bg"], scale_modifier=1.0,
                                                   viewmatrix=ri["viewmatrix"], projmatrix=ri["projmatrix"], sh_degree=3,
                                                   campos=ri["campos"], prefiltered=True, debug=False)
            leaves = {k: ri[k].clone().requires_grad_(True) for k in ("means3D", "means2D", "colors", "opacities", "conic")}
            color, radii = dgr.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=leaves["means2D"], shs=None,
                                                      colors_precomp=leaves["colors"], opacities=leaves["opacities"],
                                                      cov3D_precomp=ri["cov3D"], conic_precomp=leaves["conic"])
            color.sum().backward()
            got.append((color.detach().clone(), radii.clone(), dgr.LAST_STATS["num_rendered"]))
            del color, radii, leaves   # the graph is gone: the lease returns its buffer
            seen.append(sum(len(v) for v in dgr._ImgLease._pools.values()))
        outs[recycle] = got
        # a completed pass's workspace goes back to the pool (and the next pass takes it: never more than one there)
        assert seen == ([1, 1, 1] if recycle else [0, 0, 0]), seen
        torch.cuda.synchronize()
    dgr.RECYCLE_IMG_WS = True
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
