"""Host logic that needs no GPU: the C ABI loads and exports every declared symbol, argument validation mirrors the
reference's messages, the product path refuses CPU tensors (no fallback), render() plumbing (with the TEST-ONLY oracle
backend patched in), Adam groups, lr schedule."""
import ctypes
import math
import re
import os

import numpy as np
import pytest
import torch

from gaussianhaircut_amd import _lib
from gaussianhaircut_amd.diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer)
from gaussianhaircut_amd.gaussian_renderer import render
from gaussianhaircut_amd.scene.gaussian_model import GaussianModel, OptimizationParams
from gaussianhaircut_amd.trainer import PIPE, training_step, make_ground_truth
from gaussianhaircut_amd.utils import synthetic as syn
from tests.oracle_backend import oracle_rasterizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ghr.h")).read()
    declared = set(re.findall(r"\b(ghr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ghr_view_args", "ghr_ws_view"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.ghr_abi_version() == _lib.ABI_VERSION == 20


def test_workspace_sizes_and_error_codes():
    g, i = _lib.forward_sizes(1000, 1920, 1080, False)
    gb, _ = _lib.forward_sizes(1000, 1920, 1080, True)
    assert g >= 1000 * (64 + 4 + 8) and gb >= g + 1000 * 24
    assert i >= 1920 * 1080 * 8 + 8160 * 8
    assert _lib.binning_size(1000, 64, 64) >= 12000 and _lib.binning_size(0, 64, 64) > 0
    L = _lib.lib()
    a = _lib.ViewArgs()
    a.P, a.W, a.H, a.C = 4, 64, 64, 3
    assert L.ghr_forward_stage1(None, ctypes.byref(a), None, None, None, None) == _lib.GHR_E_INVALID
    assert b"GHR_NUM_CHANNELS" in L.ghr_last_error()
    a.C = 10
    assert L.ghr_forward_stage1(None, ctypes.byref(a), None, None, None, None) == _lib.GHR_E_NOCOLORS
    assert L.ghr_last_error() == b"For non-RGB, provide precomputed Gaussian colors!"
    with pytest.raises(RuntimeError, match="provide precomputed Gaussian colors"):
        _lib.check(_lib.GHR_E_NOCOLORS)


def test_model_args_sh_coeffs_must_be_a_square_of_the_degree_plus_one():
    """include/ghr.h: K = (max_sh_degree + 1)^2.  The projection kernels stage features_rest in 16-B pieces and count on
    rows of 0 or >= 9 floats: anything else is refused before a kernel is launched (no GPU needed to see the refusal)."""
    L = _lib.lib()
    m = _lib.ModelArgs()
    m.P, m.W, m.H, m.sh_degree, m.mode = 0, 64, 64, 0, 0
    r_host = ctypes.c_uint32(0)
    for k in (2, 3, 5, 8, 15):
        m.sh_coeffs = k
        rc = L.ghr_model_forward_stage1(None, ctypes.byref(m), None, None, None, None,
                                        ctypes.cast(ctypes.byref(r_host), ctypes.c_void_p))
        assert rc == _lib.GHR_E_INVALID, k
        assert b"sh_coeffs must be" in L.ghr_last_error(), L.ghr_last_error()


def _settings(ri):
    return GaussianRasterizationSettings(ri["H"], ri["W"], ri["tanfovx"], ri["tanfovy"], ri["bg"], 1.0,
                                         ri["viewmatrix"], ri["projmatrix"], 3, ri["campos"], True, False)


def test_argument_validation_messages_and_no_cpu_fallback():
    ri = syn.raster_inputs(syn.CONFIGS["tiny"])
    r = GaussianRasterizer(_settings(ri))
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(ri["means3D"], ri["means2D"], ri["opacities"], scales=ri["scales"], rotations=ri["rotations"])
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(ri["means3D"], ri["means2D"], ri["opacities"], shs=torch.zeros(1), colors_precomp=ri["colors"],
          scales=ri["scales"], rotations=ri["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(ri["means3D"], ri["means2D"], ri["opacities"], colors_precomp=ri["colors"], scales=ri["scales"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(ri["means3D"], ri["means2D"], ri["opacities"], colors_precomp=ri["colors"], scales=ri["scales"],
          rotations=ri["rotations"], cov3D_precomp=ri["cov3D"])
    # the product path must fail loudly on CPU tensors: there is no CPU / eager fallback
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(ri["means3D"], ri["means2D"], ri["opacities"], colors_precomp=ri["colors"], cov3D_precomp=ri["cov3D"],
          conic_precomp=ri["conic"])
    with pytest.raises(RuntimeError, match="ROCm device"):
        r.markVisible(ri["means3D"])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaussianhaircut_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "ghr_oracle" not in src, f


def test_render_dict_and_gradients_with_oracle_backend():
    spec = syn.CONFIGS["tiny"]
    model, cam = syn.make_model(spec), syn.make_view(spec)
    with oracle_rasterizer():
        pkg = render(cam, model, PIPE, syn.background())
        assert set(pkg) == {"render", "mask", "orient_angle", "orient_conf", "viewspace_points", "visibility_filter",
                            "radii"}
        assert pkg["render"].shape == (3, spec.H, spec.W) and pkg["mask"].shape == (2, spec.H, spec.W)
        assert pkg["orient_angle"].shape == (1, spec.H, spec.W) and pkg["orient_conf"].shape == (1, spec.H, spec.W)
        assert pkg["radii"].shape == (spec.P,) and pkg["radii"].dtype == torch.int32
        assert (pkg["visibility_filter"] == (pkg["radii"] > 0)).all()
        assert ((pkg["orient_angle"] >= 0) & (pkg["orient_angle"] <= 1)).all()
        (pkg["render"].sum() + pkg["mask"].sum() + pkg["orient_conf"].sum()).backward()
    for p in model.leaf_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert model._xyz.grad.abs().sum() > 0 and model._scaling.grad.abs().sum() > 0
    assert pkg["viewspace_points"].grad is not None  # densification signal (gaussian_renderer/__init__.py:30-34)


def test_adam_groups_and_lr_schedule():
    spec = syn.CONFIGS["tiny"]
    model = syn.make_model(spec)
    opt = OptimizationParams()
    model.training_setup(opt)
    names = [g["name"] for g in model.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "label", "scaling", "rotation", "orient_conf"]
    assert model.optimizer.defaults["eps"] == 1e-15
    lrs = {g["name"]: g["lr"] for g in model.optimizer.param_groups}
    assert lrs["f_rest"] == pytest.approx(opt.feature_lr / 20) and lrs["opacity"] == 0.05
    n_float = sum(p.numel() for g in model.optimizer.param_groups for p in g["params"]) // spec.P
    assert n_float == 3 + 3 + 45 + 1 + 1 + 3 + 4 + 1  # 61 fp32 per Gaussian
    lr0 = model.update_learning_rate(0)
    lr_end = model.update_learning_rate(opt.position_lr_max_steps)
    assert lr0 == pytest.approx(opt.position_lr_init) and lr_end == pytest.approx(opt.position_lr_final)
    mid = model.update_learning_rate(opt.position_lr_max_steps // 2)
    assert mid == pytest.approx(math.sqrt(opt.position_lr_init * opt.position_lr_final), rel=1e-6)


def test_training_step_decreases_loss_with_oracle_backend():
    spec = syn.CONFIGS["tiny"]
    model, cam = syn.make_model(spec), syn.make_view(spec)
    opt = OptimizationParams()
    with oracle_rasterizer():
        gt = syn.make_model(spec)
        with torch.no_grad():
            gt._features_dc.add_(0.3)
        make_ground_truth(gt, [cam], syn.background())
        model.training_setup(opt)
        losses = [float(training_step(model, [cam], syn.background(), opt, it + 1)) for it in range(6)]
    assert losses[-1] < losses[0]


def _hair_scene(dev="cpu", cam="front"):
    """A frozen head (free Gaussians, half of them labelled head) + explicit strands, as src/train_strands.py sets up."""
    from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands
    spec = syn.CONFIGS["tiny"]
    head = syn.make_model(spec, dev)
    with torch.no_grad():
        head._label[: spec.P // 2] = -4.0   # sigmoid < 0.5 -> head
        head._label[spec.P // 2:] = 4.0     # hair-labelled free Gaussians are dropped in the strand stage
    head.precompute_head()
    g = torch.Generator().manual_seed(9)
    S, n_seg = 40, 10
    origins = torch.nn.functional.normalize(torch.randn(S, 1, 3, generator=g), dim=-1) * 0.9
    dirs = torch.randn(S, n_seg, 3, generator=g) * 0.015 + torch.randn(S, 1, 3, generator=g) * 0.03
    feats = torch.randn(S * n_seg, 16, 3, generator=g) * 0.2
    hair = GaussianModelStrands(3).create_from_strands(origins.to(dev), dirs.to(dev), feats.to(dev))
    return spec, head, hair, syn.make_view(spec, dev, cam)


def test_render_hair_dict_gradients_and_plumbing_with_oracle_backend():
    """render_hair() (reference gaussian_renderer/__init__.py:116-214): head + hair concatenation, keep mask, radii
    scatter, gradients reach the strand parameters only."""
    import oracle
    from gaussianhaircut_amd.gaussian_renderer import render_hair
    from tests import helpers as hp
    spec, head, hair, cam = _hair_scene()
    n_head, n_hair = int(head.mask_precomp.sum()), hair.get_xyz.shape[0]
    with oracle_rasterizer():
        pkg = render_hair(cam, head, hair, PIPE, syn.background())
        assert set(pkg) == {"render", "mask", "orient_angle", "orient_conf", "viewspace_points", "visibility_filter",
                            "radii"}
        assert pkg["render"].shape == (3, spec.H, spec.W) and pkg["radii"].shape == (n_head + n_hair,)
        assert (pkg["visibility_filter"] == (pkg["radii"] > 0)).all() and pkg["radii"].max() > 0
        (pkg["render"].sum() + pkg["mask"].sum() * 0.5 + pkg["orient_conf"].sum() + pkg["orient_angle"].sum()).backward()
    assert hair._dirs.grad is not None and torch.isfinite(hair._dirs.grad).all() and hair._dirs.grad.abs().sum() > 0
    assert hair._features_dc.grad.abs().sum() > 0 and hair._orient_conf.grad.abs().sum() > 0
    # (as in the reference, the head's conic / depth are not detached; the head is frozen by not being optimised)
    assert not head.xyz_precomp.requires_grad and not head.shs_view.requires_grad
    # hair label channel: head Gaussians splat label 0, strands label 1 -> mask[0] <= mask[1] (foreground) everywhere
    m = pkg["mask"].detach()
    assert (m[0] <= m[1] + 1e-5).all() and m[0].max() > 0.1
    # plumbing: the same inputs straight through the oracle (mode A_sr: scales/rotations AND conic given)
    with torch.no_grad():
        xyz = torch.cat([head.xyz_precomp, hair.get_xyz])
        keep = torch.cat([head.filter_points(cam)[head.mask_precomp], hair.filter_points(cam)])
        conic = torch.cat([head.get_conic(cam)[head.mask_precomp], hair.get_conic(cam)])
        out, radii, _ = oracle.rasterize_forward(
            hp.np32(syn.background()), hp.np32(xyz[keep]), hp.np32(_hair_colors(head, hair, cam)[keep]),
            hp.np32(torch.cat([head.opacity_precomp, hair.get_opacity])[keep]), hp.np32(cam.world_view_transform),
            hp.np32(cam.full_proj_transform), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), spec.H, spec.W,
            scales=hp.np32(torch.cat([head.scaling_precomp, hair.get_scaling])[keep]),
            rotations=hp.np32(torch.cat([head.rotation_precomp, hair.get_rotation])[keep]), conic_precomp=hp.np32(conic[keep]))
    assert np.abs(out[:3] - pkg["render"].detach().numpy()).max() < 1e-6
    assert (pkg["radii"][keep].numpy() == radii).all() and (pkg["radii"][~keep] == 0).all()


def _hair_colors(head, hair, cam):
    """The 10 features render_hair() assembles (reference :160-186), restated for the plumbing check."""
    from gaussianhaircut_amd.gaussian_renderer import _sh_to_rgb
    m = head.mask_precomp
    xyz = torch.cat([head.xyz_precomp, hair.get_xyz])
    shs = torch.cat([head.shs_view, hair.get_features.transpose(1, 2).reshape(-1, 3, 16)])
    rgb = _sh_to_rgb(hair.active_sh_degree, shs, xyz, cam.camera_center)
    z1 = torch.zeros_like(head.xyz_precomp[:, :1])
    label = torch.cat([z1, hair.get_label])
    dir2d = torch.cat([torch.zeros_like(head.xyz_precomp), hair.get_direction_2d(cam)])
    conf = torch.cat([z1, hair.get_orient_conf])
    depth = torch.cat([head.get_depths(cam)[m], hair.get_depths(cam)])
    return torch.cat([rgb, label, torch.ones_like(label), dir2d, conf, depth], dim=-1)


def test_tan_half_fov_is_cached_per_tensor_version():
    """render() needs tan(FoV/2) as a host float; for a device-resident FoV tensor that is a blocking read, so the value
    is remembered on the tensor and refreshed only when the tensor is written (trainable FoV in the reference)."""
    import math
    from gaussianhaircut_amd.gaussian_renderer import _tan_half
    fov = torch.tensor(0.8)
    assert _tan_half(fov) == pytest.approx(math.tan(0.4))
    assert getattr(fov, "_ghr_tan_half")[0] == fov._version
    fov.fill_(1.0)  # in-place update bumps the version: the cache must not serve the old value
    assert _tan_half(fov) == pytest.approx(math.tan(0.5))
    assert _tan_half(0.6) == pytest.approx(math.tan(0.3))  # plain floats pass through


def test_capacity_guess_follows_the_largest_recent_count():
    """The speculative binning capacity is 1.25 x the largest instance count of the last 64 frames (+4096), rounded up to a
    coarse grid (so that the per-view workspaces keep their size while the count drifts: no allocator growth inside a step):
    a narrow view after a wide one must not shrink it, and it must forget counts that left the window."""
    import gaussianhaircut_amd.diff_gaussian_rasterization as dgr

    def want(m):
        g = m + m // 4 + 4096
        q = 1 << max(16, g.bit_length() - 5)
        return (g + q - 1) // q * q

    saved = (dict(dgr._R_HINT), dict(dgr._R_RECENT), dict(dgr.LAST_STATS))
    try:
        dgr._R_HINT.pop(7, None)
        dgr._R_RECENT.pop(7, None)
        dgr._note_count(7, 1_000_000, 10)
        assert dgr._R_HINT[7] == want(1_000_000) and dgr.LAST_STATS["num_rendered"] == 1_000_000
        assert 1_254_096 <= dgr._R_HINT[7] < 1_254_096 * 1.07
        dgr._note_count(7, 100_000, 10)                  # narrow view: the guess keeps covering the wide one
        assert dgr._R_HINT[7] == want(1_000_000) and dgr.LAST_STATS["num_rendered"] == 100_000
        for k in range(1, 20):                           # a count that creeps up keeps its capacity (same allocation sizes)
            dgr._note_count(7, 1_000_000 + 500 * k, 10)
        assert dgr._R_HINT[7] == want(1_000_000)
        dgr._note_count(7, 4_000_000, 10)
        assert dgr._R_HINT[7] == want(4_000_000) >= 5_004_096
        for _ in range(64):                              # the wide views leave the window
            dgr._note_count(7, 200_000, 10)
        assert dgr._R_HINT[7] == want(200_000)
        dgr._note_count(7, 1000, 10)
        for _ in range(64):
            dgr._note_count(7, 1000, 10)
        assert dgr._R_HINT[7] == 65536                   # (the grid's floor)
    finally:
        dgr._R_HINT.clear(); dgr._R_HINT.update(saved[0])
        dgr._R_RECENT.clear(); dgr._R_RECENT.update(saved[1])
        dgr.LAST_STATS.update(saved[2])


@pytest.mark.parametrize("n,chunks", [(1, 4), (1023, 4), (1024, 4), (4097, 4), (30_500_000, 4), (30_500_000, 1), (5000, 64)])
def test_adam_chunk_ranges_tile_the_buffer(n, chunks):
    """FusedAdam.step_chunked walks `_chunk_ranges`: consecutive, non-empty, 4-KiB aligned starts, exactly covering [0, n)
    with at most `chunks` pieces."""
    from types import SimpleNamespace
    from gaussianhaircut_amd.optim import FusedAdam
    ranges = FusedAdam._chunk_ranges(SimpleNamespace(flat_param=torch.empty(n)), chunks)
    assert 1 <= len(ranges) <= max(chunks, 1)
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    for (a, b), (c, d) in zip(ranges, ranges[1:]):
        assert b == c
    assert all(a < b and a % 1024 == 0 for a, b in ranges)


@pytest.mark.parametrize("deg", [0, 1, 2, 3, None])
def test_adam_reduce_plan_covers_the_buffer_and_skips_inactive_sh_bands(deg):
    """FusedAdam.step_chunked(reduce=True) walks `_reduce_plan`: consecutive ranges exactly covering the flat buffer; the
    f_rest group is sent whole (degree 3 / unknown), packed to its active coefficients (degree 1, 2) or not at all
    (degree 0); everything else is all-reduced in place."""
    from types import SimpleNamespace
    from gaussianhaircut_amd.optim import FusedAdam
    P = 1234
    shapes = [("xyz", (P, 3)), ("f_dc", (P, 1, 3)), ("f_rest", (P, 15, 3)), ("opacity", (P, 1)), ("label", (P, 1)),
              ("scaling", (P, 3)), ("rotation", (P, 4)), ("orient_conf", (P, 1))]
    groups = [{"name": n, "params": [torch.empty(s)]} for n, s in shapes]
    n = sum(g["params"][0].numel() for g in groups)
    act = None if deg is None else (deg + 1) ** 2 - 1
    fake = SimpleNamespace(flat_param=torch.empty(n), param_groups=groups, active_rest_coeffs=act)
    plan = FusedAdam._reduce_plan(fake, 4)
    assert plan[0][0] == 0 and plan[-1][1] == n
    for x, y in zip(plan, plan[1:]):
        assert x[1] == y[0] and x[0] < x[1]
    rest_a, rest_b = 6 * P, 6 * P + 45 * P
    sent = sum(b - a for a, b, how in plan if how == "sum") + sum(how[1] * how[3] * 3 for a, b, how in plan
                                                                if isinstance(how, tuple))
    if deg in (3, None):
        assert all(how == "sum" for _, _, how in plan) and sent == n
    else:
        special = [x for x in plan if x[2] != "sum"]
        assert len(special) == 1 and special[0][:2] == (rest_a, rest_b)
        assert special[0][2] == ("none" if deg == 0 else ("rest", P, 15, act))
        assert sent == n - 45 * P + 3 * P * act
    # SH gradients as per-view factors (ABI 19): neither f_dc nor f_rest is reduced, whatever the active degree
    fake._views = dict(buf=None, gather=True)
    plan = FusedAdam._reduce_plan(fake, 4)
    assert plan[0][0] == 0 and plan[-1][1] == n and all(x[1] == y[0] for x, y in zip(plan, plan[1:]))
    assert [x[:2] for x in plan if x[2] == "views"] == [(3 * P, 6 * P), (6 * P, 51 * P)]
    assert all(how == "sum" for a, b, how in plan if not (3 * P <= a < 51 * P))
    assert sum(b - a for a, b, how in plan if how == "sum") == 13 * P
    fake._views = dict(buf=None, gather=False)  # a rank folding only its own views: the plan is the usual one
    assert all(how == "sum" for _, _, how in FusedAdam._reduce_plan(SimpleNamespace(**{**vars(fake), "active_rest_coeffs": None}), 4))


def test_camera_requires_grad_detects_every_trainable_camera_tensor():
    """A trainable camera (reference default: src/arguments/__init__.py:61-62) is recognised whichever of its tensors carries
    the graph; since ABI 17 the fused path returns the camera's gradients itself (tests/test_camera_grads.py) instead of
    falling back to the autograd projection."""
    import torch
    from gaussianhaircut_amd.gaussian_renderer import camera_requires_grad
    from gaussianhaircut_amd.utils import synthetic as syn
    cam = syn.make_view(syn.CONFIGS["tiny"], "cpu")
    assert not camera_requires_grad(cam)
    for name in ("world_view_transform", "full_proj_transform", "camera_center", "FoVx", "FoVy"):
        c = syn.make_view(syn.CONFIGS["tiny"], "cpu")
        setattr(c, name, getattr(c, name).clone().requires_grad_(True))
        assert camera_requires_grad(c), name
        with torch.no_grad():
            assert not camera_requires_grad(c), name
    # non-leaf tensors computed from a trainable pose count as well
    c = syn.make_view(syn.CONFIGS["tiny"], "cpu")
    pose = c.world_view_transform.clone().requires_grad_(True)
    c.full_proj_transform = pose @ c.projection_matrix
    assert camera_requires_grad(c)


def test_strand_model_modules_export_the_reference_class_names():
    """R:src/scene/__init__.py:17-19 / gaussian_renderer/__init__.py:17 / train_strands.py:21 / train_latent_strands.py:21
    import GaussianModelCurves from scene.gaussian_model_strands and GaussianModelHair from
    scene.gaussian_model_latent_strands (and both from `scene`): the drop-in package answers to the same names."""
    from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelCurves, GaussianModelStrands
    from gaussianhaircut_amd.scene.gaussian_model_latent_strands import GaussianModelHair
    from gaussianhaircut_amd import scene
    assert GaussianModelCurves is GaussianModelStrands and issubclass(GaussianModelHair, GaussianModelStrands)
    assert scene.GaussianModelCurves is GaussianModelCurves and scene.GaussianModelHair is GaussianModelHair
    assert scene.GaussianModel.__name__ == "GaussianModel"


def test_fused_hair_check_does_not_evaluate_properties():
    """``_use_fused_hair`` asks whether the strand model HAS ``get_orient_conf`` / ``get_scaling`` / ``get_xyz``: with ``hasattr``
    that ran the properties -- an exp kernel over every strand Gaussian and its autograd node, twice per iteration (round 6)."""
    from gaussianhaircut_amd.gaussian_renderer import _has
    calls = []

    class M:
        attr_on_class = 1

        def __init__(self):
            self._dir = 0

        @property
        def get_orient_conf(self):
            calls.append("evaluated")
            return 1

    class Dyn:
        def __getattr__(self, name):
            if name == "answered":
                return 1
            raise AttributeError(name)

    m = M()
    assert _has(m, "get_orient_conf") and _has(m, "_dir") and _has(m, "attr_on_class") and not _has(m, "missing")
    assert calls == []
    assert _has(Dyn(), "answered") and not _has(Dyn(), "missing")
