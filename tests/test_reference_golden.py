"""Our host-side mirror vs golden vectors produced by the REFERENCE'S OWN Python code
(tests/golden/make_reference_golden.py imported /root/reference/src and wrote the .npz that is committed here)."""
import math
import os

import numpy as np
import pytest
import torch

from gaussianhaircut_amd.scene.gaussian_model import GaussianModel
from gaussianhaircut_amd.utils import general_utils as gu
from gaussianhaircut_amd.utils import graphics_utils as gr
from gaussianhaircut_amd.utils import loss_utils as lu
from gaussianhaircut_amd.utils import sh_utils as sh
from gaussianhaircut_amd.utils import synthetic as syn

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_host_golden.npz"))


def close(a, b, rtol=2e-5, atol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert (err <= 0).all(), "max violation %g at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))


# cameras: scene/cameras.py:parity_camera -- the rotated / rolled ring views have all nine entries of W non-trivial (with the
# front camera's W = I a transposed view rotation in T = W J, the 2D direction or the SH view direction would be invisible)
@pytest.mark.parametrize("cfg,camname", [("tiny", "front"), ("tiny_strands", "front"), ("tiny", "ring13roll"),
                                         ("tiny_strands", "ring5")])
def test_projection_helpers_match_reference(cfg, camname):
    spec = syn.CONFIGS[cfg]
    m = syn.make_model(spec)
    cam = syn.make_view(spec, cam=camname)
    cfg = cfg if camname == "front" else cfg + "@" + camname
    close(cam.world_view_transform.numpy(), G[cfg + "/view"])
    close(cam.full_proj_transform.numpy(), G[cfg + "/proj"])
    # ... and the camera itself against the reference's own construction (src/scene/cameras.py:72-80) from (R, T, FoV)
    from gaussianhaircut_amd.scene.cameras import Camera
    fx, fy = G[cfg + "/cam_fov"]
    c2 = Camera(G[cfg + "/cam_R"], G[cfg + "/cam_T"], fx, fy, spec.W, spec.H)
    close(c2.world_view_transform.numpy(), G[cfg + "/ref_view"], rtol=1e-6, atol=1e-7)
    close(c2.full_proj_transform.numpy(), G[cfg + "/ref_proj"], rtol=2e-6, atol=1e-6)
    close(c2.camera_center.numpy(), G[cfg + "/ref_center"], rtol=1e-5, atol=2e-6)
    if camname != "front":
        Rv = G[cfg + "/ref_view"][:3, :3]
        assert np.abs(Rv - np.eye(3)).max() > 0.3 and np.abs(Rv - Rv.T).max() > 0.1
    with torch.no_grad():
        conic = m.get_conic(cam)
        close(m.get_scaling.numpy(), G[cfg + "/scaling"])
        close(m.get_rotation.numpy(), G[cfg + "/rotation"])
        close(m.get_opacity.numpy(), G[cfg + "/opacity"])
        close(m.cov.numpy(), G[cfg + "/cov3D"], rtol=5e-5, atol=1e-9)
        # cov2D / conic are cancellation-prone for needle-like strands: scale the tolerance by the row magnitude
        ref2d = G[cfg + "/cov2d"]
        assert np.abs(m.cov2d.numpy() - ref2d).max() <= 2e-4 * np.abs(ref2d).max(axis=1, keepdims=True).max() + 1e-4
        refc = G[cfg + "/conic"]
        rel = np.abs(conic.numpy() - refc) / (np.abs(refc).max(axis=1, keepdims=True) + 1e-6)
        assert rel.max() < 1e-5, rel.max()   # every row (same torch ops as the reference's Python: 0 on the golden's machine)
        close(m.get_mean_2d(cam).numpy(), G[cfg + "/mean2d"], rtol=2e-5, atol=2e-6)
        close(m.get_depths(cam).numpy(), G[cfg + "/depths"])
        close(m.get_direction_2d(cam).numpy(), G[cfg + "/dir2d"], rtol=1e-4, atol=1e-4)
        mask = m.filter_points(cam).numpy()
        assert (mask == G[cfg + "/mask"]).all()
        K = 16
        shs_view = m.get_features.transpose(1, 2).reshape(-1, 3, K)
        d = m.get_xyz - cam.camera_center[None]
        d = d / d.norm(dim=1, keepdim=True)
        for deg in range(4):
            close(sh.eval_sh(deg, shs_view, d).numpy(), G[cfg + "/sh%d" % deg], rtol=1e-5, atol=2e-6)


def test_general_and_graphics_utils_match_reference():
    q = torch.from_numpy(G["util/q"])
    close(gu.build_rotation(q).numpy(), G["util/build_rotation"])
    s = torch.from_numpy(G["util/s"])
    L = gu.build_scaling_rotation(s, q)
    close(L.numpy(), G["util/scaling_rotation"])
    close(gu.strip_symmetric(L.transpose(1, 2) @ L).numpy(), G["util/strip_symmetric"])
    close(gu.parallel_transport(torch.from_numpy(G["util/pt_a"]), torch.from_numpy(G["util/pt_b"])).numpy(),
          G["util/parallel_transport"])
    f = gu.get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
    close([f(int(t)) for t in G["util/lr_steps"]], G["util/lr_values"], rtol=1e-12, atol=0)
    close(gr.getProjectionMatrix(0.01, 100.0, torch.tensor(0.9), torch.tensor(0.6)).numpy(), G["util/projection"])
    close(gr.getWorld2View2(G["util/w2v_R"], np.array([0.3, -0.2, 4.0])), G["util/w2v"])


def test_losses_match_reference():
    i1, i2, mask = (torch.from_numpy(G["loss/" + k]) for k in ("img1", "img2", "mask"))
    close(float(lu.l1_loss(i1, i2)), G["loss/l1"])
    close(float(lu.l1_loss(i1, i2, mask=mask)), G["loss/l1_masked"])
    close(float(lu.ssim(i1, i2)), G["loss/ssim"], rtol=1e-5)
    a1, a2, conf = (torch.from_numpy(G["loss/" + k]) for k in ("ang1", "ang2", "conf"))
    close(float(lu.or_loss(a1, a2, conf, weight=torch.ones_like(mask) * 0.7, mask=mask)), G["loss/or"])
    close(float(lu.or_loss(a1, a2)), G["loss/or_noconf"])


@pytest.mark.parametrize("camname", ["front", "ring13roll"])
def test_strand_model_matches_reference_strands_module(camname):
    """GaussianModelStrands vs the reference's src/scene/gaussian_model_strands.py (initialize_gaussians_hair,
    get_conic with eps 1e-7, get_direction_2d = normalize(dir) @ T, filter_points, unit opacity / label)."""
    from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands
    spec = syn.CONFIGS["tiny_strands"]
    cam = syn.make_view(spec, cam=camname)
    V = "strands/" if camname == "front" else "strands@" + camname + "/"
    m = GaussianModelStrands(3).create_from_strands(torch.from_numpy(G["strands/origins"]),
                                                    torch.from_numpy(G["strands/dirs"]),
                                                    torch.from_numpy(G["strands/features"]))
    with torch.no_grad():
        close(m.get_xyz.numpy(), G["strands/xyz"])
        close(m._rotation.numpy(), G["strands/rotation"], rtol=1e-5, atol=1e-6)
        close(m.get_scaling.numpy(), G["strands/scaling"])
        conic = m.get_conic(cam)
        close(m.cov.numpy(), G[V + "cov3D"], rtol=5e-5, atol=1e-9)
        ref2d = G[V + "cov2d"]
        assert np.abs(m.cov2d.numpy() - ref2d).max() <= 2e-4 * np.abs(ref2d).max() + 1e-4
        refc = G[V + "conic"]
        rel = np.abs(conic.numpy() - refc) / (np.abs(refc).max(axis=1, keepdims=True) + 1e-6)
        assert rel.max() < 1e-5, rel.max()   # every row (same torch ops as the reference's Python: 0 on the golden's machine)
        close(m.get_mean_2d(cam).numpy(), G[V + "mean2d"], rtol=2e-5, atol=2e-6)
        close(m.get_depths(cam).numpy(), G[V + "depths"])
        close(m.get_direction_2d(cam).numpy(), G[V + "dir2d"], rtol=1e-4, atol=1e-4)
        assert (m.filter_points(cam).numpy() == G[V + "mask"]).all()
        close(m.get_opacity.numpy(), G["strands/opacity"])
        close(m.get_label.numpy(), G["strands/label"])
        close(m.get_orient_conf.numpy(), G["strands/orient_conf"])


def _densify_sequence(model, opt, dev="cpu"):
    """The exact sequence tests/golden/make_reference_golden.py ran on the reference's GaussianModel."""
    gd = torch.Generator().manual_seed(4242)
    for it in range(2):
        for grp in model.optimizer.param_groups:
            prm = grp["params"][0]
            g = (torch.randn(prm.shape, generator=gd) * 1e-3).to(dev)
            if prm.grad is None:
                prm.grad = g
            else:
                prm.grad.copy_(g)
        model.update_learning_rate(it + 1)
        model.optimizer.step()
    model.xyz_gradient_accum = (torch.rand(model.get_xyz.shape[0], 1, generator=gd) * 1.2e-3).to(dev)
    model.denom = torch.randint(0, 4, (model.get_xyz.shape[0], 1), generator=gd).float().to(dev)
    model.max_radii2D = (torch.rand(model.get_xyz.shape[0], generator=gd) * 30).to(dev)
    return gd


def test_densify_prune_and_optimizer_surgery_match_reference():
    """densify_and_prune (clone + split + prune) and reset_opacity with torch.optim.Adam state surgery vs the
    reference's own implementation (gaussian_model.py:560-741), same seeds -> same tensors."""
    from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
    spec = syn.CONFIGS["tiny"]
    m = syn.make_model(spec)
    opt = OptimizationParams()
    m.training_setup(opt, fused=False)
    gd = _densify_sequence(m, opt)
    close(m.xyz_gradient_accum.numpy(), G["densify/accum"])
    torch.manual_seed(991)
    m.densify_and_prune(opt.densify_grad_threshold, 0.005, 2.5, 20)
    assert m.get_xyz.shape[0] == int(G["densify/n_after"]) != spec.P
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_label", "_scaling", "_rotation", "_orient_conf"):
        close(getattr(m, n).detach().numpy(), G["densify/after" + n], rtol=1e-6, atol=1e-7)
    for grp in m.optimizer.param_groups:
        st = m.optimizer.state[grp["params"][0]]
        close(st["exp_avg"].numpy(), G["densify/m_" + grp["name"]], rtol=1e-6, atol=1e-12)
        close(st["exp_avg_sq"].numpy(), G["densify/v_" + grp["name"]], rtol=1e-6, atol=1e-15)
    assert m.xyz_gradient_accum.shape == (m.get_xyz.shape[0], 1) and m.max_radii2D.shape == (m.get_xyz.shape[0],)
    with torch.no_grad():
        m.reset_opacity()
    close(m._opacity.detach().numpy(), G["densify/opacity_reset"], rtol=1e-6, atol=1e-7)
    close(m.optimizer.state[m._opacity]["exp_avg"].numpy(), G["densify/m_opacity_reset"])
    for grp in m.optimizer.param_groups:
        prm = grp["params"][0]
        prm.grad = torch.randn(prm.shape, generator=gd) * 1e-3
    m.optimizer.step()
    close(m._xyz.detach().numpy(), G["densify/xyz_after_step"], rtol=1e-6, atol=1e-7)


def test_ply_round_trip_and_reference_layout(tmp_path):
    """save_ply / load_ply (gaussian_model.py:458-579): attribute order, the two files, float32 payload, round trip."""
    from gaussianhaircut_amd.scene.ply_io import read_ply_vertices
    spec = syn.CONFIGS["tiny"]
    m = syn.make_model(spec)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    raw = os.path.join(os.path.dirname(path), "raw_point_cloud.ply")
    names_raw, v_raw = read_ply_vertices(raw)
    names_vis, v_vis = read_ply_vertices(path)
    exp = (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] +
           ["opacity", "orient_conf", "label_0"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)])
    assert names_raw == exp and names_vis == [n for n in exp if n != "label_0"]
    assert v_raw.shape == (spec.P,) and all(v_raw.dtype[n] == np.dtype("<f4") for n in names_raw)
    header = open(raw, "rb").read(64)
    assert header.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % spec.P)
    # f_dc / f_rest are stored channel-major: transpose(1, 2).flatten(1) (gaussian_model.py:486-487)
    fr = m._features_rest.detach().transpose(1, 2).flatten(start_dim=1).numpy()
    assert np.array_equal(v_raw["f_rest_17"], fr[:, 17]) and np.array_equal(v_raw["nx"], np.zeros(spec.P, np.float32))
    m2 = GaussianModel(3)
    m2.load_ply(raw)
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_label", "_orient_conf", "_scaling", "_rotation"):
        assert torch.equal(getattr(m, n).detach(), getattr(m2, n).detach()), n
    assert m2.active_sh_degree == 3 and m2._xyz.requires_grad
