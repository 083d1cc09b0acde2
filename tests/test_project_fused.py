"""Fused projection kernel (csrc/ghr_project.h), CPU side: the product's host+device functions vs the PyTorch
pipeline they replace (GaussianModel.get_conic / get_mean_2d / get_direction_2d / get_depths / filter_points + eval_sh,
itself pinned to the reference by tests/test_reference_golden.py).  Gradients are checked against fp64 autograd."""
import ctypes

import numpy as np
import pytest
import torch

from gaussianhaircut_amd.scene.gaussian_model import GaussianModel
from gaussianhaircut_amd.utils import synthetic as syn
from gaussianhaircut_amd.utils.sh_utils import eval_sh
from tests import helpers as hp


def torch_pipeline(model, cam):
    """colors_precomp / conic / means2D / opacity exactly as render() assembles them (gaussian_renderer:58-83)."""
    conic = model.get_conic(cam)
    means2D = model.get_mean_2d(cam)
    xyz = model.get_xyz
    K = (model.max_sh_degree + 1) ** 2
    shs_view = model.get_features.transpose(1, 2).reshape(-1, 3, K)
    d = xyz - cam.camera_center[None].to(xyz.dtype)
    d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(eval_sh(model.active_sh_degree, shs_view, d) + 0.5, 0.0)
    label = model.get_label
    colors = torch.cat([rgb, label, torch.ones_like(label), model.get_direction_2d(cam), model.get_orient_conf,
                        model.get_depths(cam)], dim=-1)
    keep = model.filter_points(cam)
    return conic, means2D, colors, model.get_opacity, keep


def to_double(model, cam):
    m = GaussianModel(model.max_sh_degree)
    m.active_sh_degree = model.active_sh_degree
    for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_label", "_orient_conf"):
        setattr(m, n, getattr(model, n).detach().double().requires_grad_(True))
    import copy
    c = copy.copy(cam)
    for k, v in list(c.__dict__.items()):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            setattr(c, k, v.double())
    return m, c


@pytest.mark.parametrize("cfg,deg", [("tiny", 3), ("tiny", 1), ("tiny_strands", 3), ("ragged", 0)])
def test_fused_projection_forward_and_backward(hostsim, cfg, deg):
    spec = syn.CONFIGS[cfg]
    model = syn.make_model(spec)
    model.active_sh_degree = deg
    cam = syn.make_view(spec)
    P = spec.P
    keep_alive = []
    a = hp.model_args_from(model, cam, keep_alive)
    assert hostsim.L.ghrsim_sizeof_model_args() == ctypes.sizeof(hp.ModelArgsC)

    rec = np.zeros((P, 16), np.float32)
    radii = np.zeros(P, np.int32)
    m2d = np.zeros((P, 3), np.float32)
    depths = np.zeros(P, np.float32)
    hostsim.L.ghrsim_project_forward(ctypes.byref(a), ctypes.c_void_p(rec.ctypes.data),
                                     ctypes.c_void_p(radii.ctypes.data), ctypes.c_void_p(m2d.ctypes.data),
                                     ctypes.c_void_p(depths.ctypes.data))
    md, cd = to_double(model, cam)
    conic, means2D, colors, opac, keep = torch_pipeline(md, cd)
    keep = keep.numpy()
    vis = radii > 0
    assert (vis == keep).mean() > 0.999
    both = vis & keep
    W, H = spec.W, spec.H
    pix = np.stack([((means2D[:, 0].detach().numpy() + 1) * W - 1) * 0.5,
                    ((means2D[:, 1].detach().numpy() + 1) * H - 1) * 0.5], -1)

    def close(x, ref, tol=5e-6):
        x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
        err = np.abs(x - ref) / (np.abs(ref).max(axis=-1, keepdims=True) + 1e-6)
        assert err.max() < tol, err.max()   # every element (measured: <= 1.2e-6)

    close(m2d[:, :2], means2D[:, :2].detach().numpy())
    close(rec[both, 0:2], pix[both])
    close(rec[both, 2:5], conic.detach().numpy()[both])  # fp32 cancellation in thin strands vs fp64
    close(rec[both, 5:6], opac.detach().numpy()[both])
    close(rec[both, 6:16], colors.detach().numpy()[both])

    # ---- backward: random cotangents on (conic, mean2D, colours, opacity) of the visible Gaussians
    g = torch.Generator().manual_seed(17)
    g_conic = torch.randn(P, 3, generator=g, dtype=torch.float64)
    g_m = torch.randn(P, 2, generator=g, dtype=torch.float64)
    g_col = torch.randn(P, 10, generator=g, dtype=torch.float64)
    g_op = torch.randn(P, 1, generator=g, dtype=torch.float64)
    mask = torch.from_numpy(both.astype(np.float64))[:, None]
    L = ((conic * g_conic + 0).sum(-1, keepdim=True) * mask).sum() + ((means2D[:, :2] * g_m).sum(-1, keepdim=True) * mask).sum() \
        + ((colors * g_col).sum(-1, keepdim=True) * mask).sum() + (opac * g_op * mask).sum()
    L.backward()
    gacc = np.zeros((P, 16), np.float32)
    gacc[:, 0:2] = g_m.numpy()
    gacc[:, 2] = g_conic[:, 0].numpy()
    gacc[:, 3] = 0.5 * g_conic[:, 1].numpy()  # the kernel stores HALF of d/db; the wrapper (and the fused op) doubles
    gacc[:, 4] = g_conic[:, 2].numpy()
    gacc[:, 5] = g_op[:, 0].numpy()
    gacc[:, 6:16] = g_col.numpy()
    gacc *= both[:, None]
    K = 16
    outs = dict(d_means2D=np.zeros((P, 3), np.float32), d_xyz=np.zeros((P, 3), np.float32),
                d_ls=np.zeros((P, 3), np.float32), d_rot=np.zeros((P, 4), np.float32), d_op=np.zeros(P, np.float32),
                d_label=np.zeros(P, np.float32), d_conf=np.zeros(P, np.float32),
                d_fdc=np.zeros((P, 1, 3), np.float32), d_frest=np.zeros((P, K - 1, 3), np.float32))
    radii_in = (both * 1).astype(np.int32)
    hostsim.L.ghrsim_project_backward(ctypes.byref(a), ctypes.c_void_p(radii_in.ctypes.data),
                                      ctypes.c_void_p(gacc.ctypes.data),
                                      *[ctypes.c_void_p(outs[k].ctypes.data) for k in
                                        ("d_means2D", "d_xyz", "d_ls", "d_rot", "d_op", "d_label", "d_conf", "d_fdc",
                                         "d_frest")])
    ref = dict(d_xyz=md._xyz.grad, d_ls=md._scaling.grad, d_rot=md._rotation.grad, d_op=md._opacity.grad[:, 0],
               d_label=md._label.grad[:, 0], d_conf=md._orient_conf.grad[:, 0], d_fdc=md._features_dc.grad,
               d_frest=md._features_rest.grad)
    for k, r in ref.items():
        r = r.numpy()
        got = outs[k].astype(np.float64)
        scale = np.abs(r).max() + 1e-30
        rowscale = np.abs(r.reshape(P, -1)).max(axis=1, keepdims=True).reshape((P,) + (1,) * (r.ndim - 1))
        err = np.abs(got - r) / (rowscale + 1e-3 * scale)
        # fp32 kernel vs fp64 autograd: per-row relative; thin strands amplify rounding in the conic chain
        assert err.max() < 1e-4 and np.isfinite(got).all(), (k, err.max())   # every element (measured: <= 1.5e-5)
    assert np.array_equal(outs["d_means2D"][:, :2], gacc[:, 0:2])


def test_fused_projection_explicit_mode_matches_strand_pipeline(hostsim):
    """mode 1 of the fused projection (explicit linear-space Gaussians = what render_hair() assembles for the strands:
    conic eps 1e-7, unit opacity / label, 2D direction = normalize(dir) @ T, linear confidence) against the PyTorch
    strand pipeline in fp64, forward and hand-derived backward incl. the strand-direction term."""
    from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands
    from tests.test_api_cpu import _hair_scene
    spec, _, hair, cam = _hair_scene()
    P = hair.get_xyz.shape[0]
    deg = 3
    hair.active_sh_degree = deg
    # fp64 twin whose per-Gaussian quantities are independent leaves
    md = GaussianModelStrands(3)
    md.active_sh_degree = deg
    with torch.no_grad():
        leaves = dict(_xyz=hair._xyz.double(), _scaling=hair._scaling.double(), _rotation=hair._rotation.double(),
                      _dir=hair._dir.double(), _features_dc=hair._features_dc.double(),
                      _features_rest=hair._features_rest.double(), conf=hair.get_orient_conf.double())
    leaves = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
    for k in ("_xyz", "_scaling", "_rotation", "_dir", "_features_dc", "_features_rest"):
        setattr(md, k, leaves[k])
    import copy
    cd = copy.copy(cam)
    for k, v in list(cd.__dict__.items()):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            setattr(cd, k, v.double())
    conic = md.get_conic(cd)
    means2D = md.get_mean_2d(cd)
    shs_view = md.get_features.transpose(1, 2).reshape(-1, 3, 16)
    d = md.get_xyz - cd.camera_center[None]
    d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(eval_sh(deg, shs_view, d) + 0.5, 0.0)
    ones = torch.ones(P, 1, dtype=torch.float64)
    colors = torch.cat([rgb, ones, ones, md.get_direction_2d(cd), leaves["conf"], md.get_depths(cd)], dim=-1)
    keep = md.filter_points(cd).numpy()

    # ---- product code (host-sim), mode 1
    keep_alive = []
    a = hp.ModelArgsC()
    arr = dict(xyz=hp.np32(hair._xyz), log_scales=hp.np32(hair._scaling), rotations=hp.np32(hair._rotation),
               dir3d=hp.np32(hair._dir), orient_conf_log=hp.np32(hair.get_orient_conf).reshape(-1),
               features_dc=hp.np32(hair._features_dc), features_rest=hp.np32(hair._features_rest),
               view=hp.np32(cam.world_view_transform).reshape(-1), proj=hp.np32(cam.full_proj_transform).reshape(-1),
               campos=hp.np32(cam.camera_center))
    keep_alive.append(arr)
    for k, v in arr.items():
        setattr(a, k, v.ctypes.data)
    a.opacity_logit, a.label_logit = None, None
    a.const_opacity, a.const_label, a.const_conf = 1.0, 1.0, 0.0
    a.P, a.W, a.H, a.sh_degree, a.sh_coeffs, a.mode, a.row0 = P, spec.W, spec.H, deg, 16, 1, 0
    import math
    a.scale_modifier = 1.0
    a.tan_fovx, a.tan_fovy = math.tan(float(cam.FoVx) * 0.5), math.tan(float(cam.FoVy) * 0.5)
    a.focal_x, a.focal_y = a.W / (2.0 * a.tan_fovx), a.H / (2.0 * a.tan_fovy)
    a.conic_eps = 1e-7
    rec = np.zeros((P, 16), np.float32)
    radii = np.zeros(P, np.int32)
    m2d = np.zeros((P, 3), np.float32)
    depths = np.zeros(P, np.float32)
    hostsim.L.ghrsim_project_forward(ctypes.byref(a), ctypes.c_void_p(rec.ctypes.data),
                                     ctypes.c_void_p(radii.ctypes.data), ctypes.c_void_p(m2d.ctypes.data),
                                     ctypes.c_void_p(depths.ctypes.data))
    vis = radii > 0
    assert (vis == keep).mean() > 0.999 and vis.sum() > 100
    both = vis & keep

    def close(x, ref, tol=5e-6):
        x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
        err = np.abs(x - ref) / (np.abs(ref).max(axis=-1, keepdims=True) + 1e-6)
        assert err.max() < tol, err.max()   # every element (measured: <= 1.2e-6)

    close(m2d[:, :2], means2D[:, :2].detach().numpy())
    close(rec[both, 2:5], conic.detach().numpy()[both])
    assert np.all(rec[both, 5] == 1.0) and np.all(rec[both, 9] == 1.0) and np.all(rec[both, 10] == 1.0)
    close(rec[both, 6:16], colors.detach().numpy()[both])

    # ---- backward
    g = torch.Generator().manual_seed(23)
    g_conic = torch.randn(P, 3, generator=g, dtype=torch.float64)
    g_m = torch.randn(P, 2, generator=g, dtype=torch.float64)
    g_col = torch.randn(P, 10, generator=g, dtype=torch.float64)
    mask = torch.from_numpy(both.astype(np.float64))[:, None]
    L = ((conic * g_conic).sum(-1, keepdim=True) * mask).sum() + ((means2D[:, :2] * g_m).sum(-1, keepdim=True) * mask).sum() \
        + ((colors * g_col).sum(-1, keepdim=True) * mask).sum()
    L.backward()
    gacc = np.zeros((P, 16), np.float32)
    gacc[:, 0:2] = g_m.numpy()
    gacc[:, 2] = g_conic[:, 0].numpy()
    gacc[:, 3] = 0.5 * g_conic[:, 1].numpy()
    gacc[:, 4] = g_conic[:, 2].numpy()
    gacc[:, 6:16] = g_col.numpy()
    gacc *= both[:, None]
    outs = dict(d_means2D=np.zeros((P, 3), np.float32), d_xyz=np.zeros((P, 3), np.float32),
                d_ls=np.zeros((P, 3), np.float32), d_rot=np.zeros((P, 4), np.float32), d_op=np.zeros(P, np.float32),
                d_label=np.zeros(P, np.float32), d_conf=np.zeros(P, np.float32),
                d_fdc=np.zeros((P, 1, 3), np.float32), d_frest=np.zeros((P, 15, 3), np.float32),
                d_dir=np.zeros((P, 3), np.float32))
    radii_in = (both * 1).astype(np.int32)
    hostsim.L.ghrsim_project_backward2(ctypes.byref(a), ctypes.c_void_p(radii_in.ctypes.data),
                                       ctypes.c_void_p(gacc.ctypes.data),
                                       *[ctypes.c_void_p(outs[k].ctypes.data) for k in
                                         ("d_means2D", "d_xyz", "d_ls", "d_rot", "d_op", "d_label", "d_conf", "d_fdc",
                                          "d_frest", "d_dir")])
    ref = dict(d_xyz=leaves["_xyz"].grad, d_ls=leaves["_scaling"].grad, d_rot=leaves["_rotation"].grad,
               d_conf=leaves["conf"].grad[:, 0], d_fdc=leaves["_features_dc"].grad,
               d_frest=leaves["_features_rest"].grad, d_dir=leaves["_dir"].grad)
    for k, r in ref.items():
        r = r.numpy()
        got = outs[k].astype(np.float64)
        scale = np.abs(r).max() + 1e-30
        rowscale = np.abs(r.reshape(P, -1)).max(axis=1, keepdims=True).reshape((P,) + (1,) * (r.ndim - 1))
        err = np.abs(got - r) / (rowscale + 1e-3 * scale)
        assert err.max() < 1e-4 and np.isfinite(got).all(), (k, err.max())   # every element (measured: <= 1.5e-5)


# ---------------------------------------------------------------------------------------------------------------------
# Camera cotangents (ABI 17): the reference's projection graph is differentiable w.r.t. the camera tensors
# (/root/reference/src/scene/gaussian_model.py:258-266,279-294,332-335; src/gaussian_renderer/__init__.py:59) -- the fused
# backward's per-Gaussian contributions, summed, against fp64 autograd of that graph with the camera tensors as leaves.
CAM_LAYOUT = {"view": [(4 * (c // 3) + c % 3) for c in range(12)],
              "proj": [(4 * (c // 3) + (3 if c % 3 == 2 else c % 3)) for c in range(12)]}


def cam_vector_to_tensors(cam32):
    """[32] partial-table column sums -> (d view [4,4], d proj [4,4], d campos [3], d tanfov [2]) like k_cam_fold"""
    dv, dp = np.zeros(16), np.zeros(16)
    dv[CAM_LAYOUT["view"]] = cam32[0:12]
    dp[CAM_LAYOUT["proj"]] = cam32[12:24]
    return dv.reshape(4, 4), dp.reshape(4, 4), cam32[26:29].copy(), cam32[24:26].copy()


class LeafCamera:
    """The tensors render() reads from a camera, as fp64 leaves; tan(FoV / 2) enters through FoVx / FoVy exactly like the
    reference's `torch.tan(viewpoint_camera.FoVx * 0.5)`."""

    def __init__(self, cam):
        self.image_width, self.image_height = cam.image_width, cam.image_height
        self.world_view_transform = cam.world_view_transform.detach().double().clone().requires_grad_(True)
        self.full_proj_transform = cam.full_proj_transform.detach().double().clone().requires_grad_(True)
        self.camera_center = cam.camera_center.detach().double().clone().requires_grad_(True)
        self.tanfov = torch.tan(torch.stack([cam.FoVx.detach().double(), cam.FoVy.detach().double()]) * 0.5).requires_grad_(True)
        self.FoVx = 2.0 * torch.atan(self.tanfov[0])
        self.FoVy = 2.0 * torch.atan(self.tanfov[1])


@pytest.mark.parametrize("cfg,camname,deg", [("tiny", "ring13roll", 3), ("tiny_strands", "ring13roll", 2), ("tiny", "front", 0)])
def test_fused_projection_camera_cotangents(hostsim, cfg, camname, deg):
    spec = syn.CONFIGS[cfg]
    model = syn.make_model(spec)
    model.active_sh_degree = deg
    cam = syn.make_view(spec, "cpu", camname)
    narrow = camname == "ring13roll" and cfg == "tiny"
    if narrow:
        # narrow the field of view so that part of the Gaussians sits outside the 1.3 tan(FoV / 2) clamp of tx / tz, ty / tz:
        # only those feed gradient into the clamp's tensor bounds
        import math
        from gaussianhaircut_amd.scene.cameras import Camera
        cam = Camera(cam.R, cam.T, math.radians(14.0), math.radians(9.0), spec.W, spec.H)
    P = spec.P
    keep_alive = []
    a = hp.model_args_from(model, cam, keep_alive)
    md, _ = to_double(model, cam)
    lc = LeafCamera(cam)
    conic, means2D, colors, opac, keep = torch_pipeline(md, lc)
    keep = keep.numpy()
    txtz = (md.get_xyz.detach() @ lc.world_view_transform.detach()[:3, :3] + lc.world_view_transform.detach()[3, :3])
    outside = ((txtz[:, 0] / txtz[:, 2]).abs() > 1.3 * lc.tanfov[0].item()) | ((txtz[:, 1] / txtz[:, 2]).abs() > 1.3 * lc.tanfov[1].item())
    if narrow:
        assert (outside.numpy() & keep).sum() > 10  # the clamp-bound terms are exercised
    g = torch.Generator().manual_seed(29)
    g_conic = torch.randn(P, 3, generator=g, dtype=torch.float64)
    g_m = torch.randn(P, 2, generator=g, dtype=torch.float64)
    g_col = torch.randn(P, 10, generator=g, dtype=torch.float64)
    g_op = torch.randn(P, 1, generator=g, dtype=torch.float64)
    mask = torch.from_numpy(keep.astype(np.float64))[:, None]
    gacc = np.zeros((P, 16), np.float32)
    gacc[:, 0:2] = g_m.numpy()
    gacc[:, 2] = g_conic[:, 0].numpy()
    gacc[:, 3] = 0.5 * g_conic[:, 1].numpy()
    gacc[:, 4] = g_conic[:, 2].numpy()
    gacc[:, 5] = g_op[:, 0].numpy()
    gacc[:, 6:16] = g_col.numpy()
    gacc *= keep[:, None]
    radii_in = (keep * 1).astype(np.int32)
    for detach in (0, 1):
        for t in (lc.world_view_transform, lc.full_proj_transform, lc.camera_center, lc.tanfov):
            t.grad = None
        m2 = means2D.detach() if detach else means2D
        L = ((conic * g_conic).sum(-1, keepdim=True) * mask).sum() + ((m2[:, :2] * g_m).sum(-1, keepdim=True) * mask).sum() \
            + ((colors * g_col).sum(-1, keepdim=True) * mask).sum() + (opac * g_op * mask).sum()
        L.backward(retain_graph=True)
        K = 16
        outs = dict(d_means2D=np.zeros((P, 3), np.float32), d_xyz=np.zeros((P, 3), np.float32),
                    d_ls=np.zeros((P, 3), np.float32), d_rot=np.zeros((P, 4), np.float32), d_op=np.zeros(P, np.float32),
                    d_label=np.zeros(P, np.float32), d_conf=np.zeros(P, np.float32),
                    d_fdc=np.zeros((P, 1, 3), np.float32), d_frest=np.zeros((P, K - 1, 3), np.float32))
        camv = np.full((P, 32), np.nan, np.float32)
        hostsim.L.ghrsim_project_backward3(ctypes.byref(a), ctypes.c_void_p(radii_in.ctypes.data),
                                           ctypes.c_void_p(gacc.ctypes.data),
                                           *[ctypes.c_void_p(outs[k].ctypes.data) for k in
                                             ("d_means2D", "d_xyz", "d_ls", "d_rot", "d_op", "d_label", "d_conf", "d_fdc",
                                              "d_frest")], None, ctypes.c_void_p(camv.ctypes.data), detach)
        assert np.isfinite(camv).all() and np.all(camv[:, 29:] == 0) and np.all(camv[~keep] == 0)
        dv, dp, dc, dt = cam_vector_to_tensors(camv.astype(np.float64).sum(0))
        gz = lambda t: np.zeros(tuple(t.shape)) if t.grad is None else t.grad.numpy()  # (degree 0: no view direction)
        ref = dict(view=gz(lc.world_view_transform), proj=np.zeros((4, 4)) if detach else gz(lc.full_proj_transform),
                   campos=gz(lc.camera_center), tanfov=gz(lc.tanfov))
        got = dict(view=dv, proj=dp, campos=dc, tanfov=dt)
        for k in ref:
            scale = np.abs(ref[k]).max()
            # fp32 per-Gaussian terms summed in double against fp64 autograd, relative to the tensor's largest entry
            assert np.abs(got[k] - ref[k]).max() <= 1e-4 * scale + 1e-30, (k, detach, got[k], ref[k])
        if narrow and not detach:
            assert np.abs(ref["tanfov"]).min() > 0 and np.abs(ref["campos"]).max() > 0
        # the parameter gradients are those of the camera-less instantiation, bit for bit
        outs2 = {k: np.zeros_like(v) for k, v in outs.items()}
        hostsim.L.ghrsim_project_backward3(ctypes.byref(a), ctypes.c_void_p(radii_in.ctypes.data),
                                           ctypes.c_void_p(gacc.ctypes.data),
                                           *[ctypes.c_void_p(outs2[k].ctypes.data) for k in
                                             ("d_means2D", "d_xyz", "d_ls", "d_rot", "d_op", "d_label", "d_conf", "d_fdc",
                                              "d_frest")], None, None, detach)
        for k in outs:
            assert np.array_equal(outs[k], outs2[k]), k
