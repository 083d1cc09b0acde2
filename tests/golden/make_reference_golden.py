"""Generates tests/golden/reference_host_golden.npz by IMPORTING THE REFERENCE'S OWN PYTHON (read-only, from
/root/reference/src) and running its host-side maths on seeded inputs.  Run once in the build container
(/root/reference does not exist on the GPU box; the committed .npz is what the tests read):

    python tests/golden/make_reference_golden.py

What is pinned: ``GaussianModel.get_covariance / get_conic / get_mean_2d / get_depths / get_direction_2d /
filter_points`` (src/scene/gaussian_model.py:143-393), ``eval_sh`` (src/utils/sh_utils.py:57-112), ``build_rotation``,
``strip_symmetric``, ``get_expon_lr_func``, ``parallel_transport`` (src/utils/general_utils.py),
``getProjectionMatrix`` / ``getWorld2View2`` (src/utils/graphics_utils.py), ``l1_loss`` / ``ssim`` / ``or_loss``
(src/utils/loss_utils.py), and the strand model's ``initialize_gaussians_hair`` / ``get_conic`` (eps 1e-7) /
``get_direction_2d`` / ``filter_points`` (src/scene/gaussian_model_strands.py:143-452), and ``densify_and_prune`` /
``reset_opacity`` with their ``torch.optim.Adam`` state surgery (src/scene/gaussian_model.py:560-741).  The reference modules hard-code device="cuda" and import plyfile / simple_knn, which do
not exist here: the script stubs those two modules and redirects "cuda" tensor factories to the CPU.  Nothing from
the reference is copied into the repository -- only numeric outputs.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _patch_cuda_factories():
    def wrap(fn):
        def inner(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    for name in ("zeros", "ones", "arange", "tensor", "empty", "full"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    _patch_cuda_factories()
    sys.modules["plyfile"] = types.SimpleNamespace(PlyData=None, PlyElement=None)
    knn = types.ModuleType("simple_knn")
    knn_c = types.ModuleType("simple_knn._C")
    knn_c.distCUDA2 = None
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = knn, knn_c
    # `utils` must resolve to the reference's utils package (namespace package under REF)
    sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    ref_general = importlib.import_module("utils.general_utils")
    ref_sh = importlib.import_module("utils.sh_utils")
    ref_graphics = importlib.import_module("utils.graphics_utils")
    ref_loss = importlib.import_module("utils.loss_utils")
    ref_gm = _load("ref_gaussian_model", os.path.join(REF, "scene", "gaussian_model.py"))

    from gaussianhaircut_amd.utils import synthetic as syn  # our generator only supplies the INPUT tensors
    out = {}
    # cameras: scene/cameras.py:parity_camera -- "front" has R = I; "ring5" / "ring13roll" are rotated / rolled ring views (all
    # nine entries of W non-trivial, as with the COLMAP poses of src/scene/cameras.py:72-80)
    for cfg, camname in (("tiny", "front"), ("tiny_strands", "front"), ("tiny", "ring13roll"), ("tiny_strands", "ring5")):
        spec = syn.CONFIGS[cfg]
        p = (syn.random_gaussian_params(spec.P, spec.seed, spec.log_scale_mean) if spec.kind == "random"
             else syn.strand_gaussian_params(spec.n_strands, spec.P, spec.seed))
        cam = syn.make_view(spec, cam=camname)
        cfg = cfg if camname == "front" else cfg + "@" + camname
        m = ref_gm.GaussianModel(3)
        m._xyz, m._scaling, m._rotation = p["xyz"], p["log_scales"], p["rotations"]
        m._opacity, m._label, m._orient_conf = p["opacity_logit"], p["label_logit"], p["orient_conf_log"]
        m._features_dc, m._features_rest = p["features"][:, :1], p["features"][:, 1:]
        conic = m.get_conic(cam)
        out[cfg + "/conic"] = conic.numpy()
        out[cfg + "/cov3D"] = m.cov.numpy()
        out[cfg + "/cov2d"] = m.cov2d.numpy()
        out[cfg + "/mean2d"] = m.get_mean_2d(cam).numpy()
        out[cfg + "/depths"] = m.get_depths(cam).numpy()
        out[cfg + "/dir2d"] = m.get_direction_2d(cam).numpy()
        out[cfg + "/mask"] = m.filter_points(cam).numpy()
        out[cfg + "/scaling"] = m.get_scaling.numpy()
        out[cfg + "/rotation"] = m.get_rotation.numpy()
        out[cfg + "/opacity"] = m.get_opacity.numpy()
        shs_view = m.get_features.transpose(1, 2).view(-1, 3, 16)
        d = p["xyz"] - cam.camera_center[None]
        d = d / d.norm(dim=1, keepdim=True)
        for deg in range(4):
            out[cfg + "/sh%d" % deg] = ref_sh.eval_sh(deg, shs_view, d).numpy()
        out[cfg + "/view"] = cam.world_view_transform.numpy()
        out[cfg + "/proj"] = cam.full_proj_transform.numpy()
        # the same camera built by the REFERENCE's own functions in the order of src/scene/cameras.py:72-80 from (R, T, FoV)
        wv = torch.tensor(ref_graphics.getWorld2View2(cam.R, cam.T)).transpose(0, 1)
        pm = ref_graphics.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=cam.FoVx, fovY=cam.FoVy).transpose(0, 1)
        out[cfg + "/cam_R"], out[cfg + "/cam_T"] = cam.R, cam.T
        out[cfg + "/cam_fov"] = np.array([float(cam.FoVx), float(cam.FoVy)])
        out[cfg + "/ref_view"] = wv.numpy()
        out[cfg + "/ref_proj"] = wv.unsqueeze(0).bmm(pm.unsqueeze(0)).squeeze(0).numpy()
        out[cfg + "/ref_center"] = wv.inverse()[3, :3].numpy()

    # ---- the strand-parametrised model (src/scene/gaussian_model_strands.py): its module imports trimesh / pysdf /
    # NeuralHaircut networks at import time and builds them in __init__; stub the imports, bypass __init__ and drive
    # the projection helpers (:143-452) directly on explicit strand tensors.
    for name in ("trimesh", "pysdf", "src", "src.hair_networks", "src.hair_networks.optimizable_textured_strands",
                 "src.hair_networks.strand_prior"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pysdf"].SDF = None
    sys.modules["src.hair_networks.optimizable_textured_strands"].OptimizableTexturedStrands = None
    sys.modules["src.hair_networks.strand_prior"].Decoder = None
    sys.modules["src.hair_networks.strand_prior"].Encoder = None
    ref_gms = _load("ref_gaussian_model_strands", os.path.join(REF, "scene", "gaussian_model_strands.py"))
    spec = syn.CONFIGS["tiny_strands"]
    cam = syn.make_view(spec)
    gs = torch.Generator().manual_seed(77)
    S, n_seg = 24, 12
    origins = torch.nn.functional.normalize(torch.randn(S, 1, 3, generator=gs), dim=-1) * 0.8
    dirs = torch.randn(S, n_seg, 3, generator=gs) * 0.02 + torch.randn(S, 1, 3, generator=gs) * 0.03
    feats = torch.randn(S * n_seg, 16, 3, generator=gs) * 0.1
    m = object.__new__(ref_gms.GaussianModelCurves)
    m.setup_functions()
    m.active_sh_degree = m.max_sh_degree = 3
    m.pts_origins, m._dirs = origins, dirs
    m._features_dc, m._features_rest = feats[:, :1], feats[:, 1:]
    m._orient_conf = torch.zeros(S * n_seg, 1)
    m.scale, m.use_sds = 1e-3, False
    m.initialize_gaussians_hair()
    out["strands/origins"], out["strands/dirs"], out["strands/features"] = origins.numpy(), dirs.numpy(), feats.numpy()
    out["strands/xyz"], out["strands/rotation"] = m._xyz.numpy(), m._rotation.numpy()
    out["strands/scaling"] = m._scaling.numpy()
    for camname in ("front", "ring13roll"):
        cam = syn.make_view(spec, cam=camname)
        tg = "strands/" if camname == "front" else "strands@" + camname + "/"
        conic = m.get_conic(cam)
        out[tg + "conic"] = conic.numpy()
        out[tg + "cov2d"] = m.cov.numpy()  # this class caches the 2D covariance in .cov (gaussian_model_strands.py:300)
        out[tg + "cov3D"] = m.get_covariance().numpy()
        out[tg + "mean2d"] = m.get_mean_2d(cam).numpy()
        out[tg + "depths"] = m.get_depths(cam).numpy()
        out[tg + "dir2d"] = m.get_direction_2d(cam).numpy()
        out[tg + "mask"] = m.filter_points(cam).numpy()
        out[tg + "view"], out[tg + "proj"] = cam.world_view_transform.numpy(), cam.full_proj_transform.numpy()
        out[tg + "campos"] = cam.camera_center.numpy()
    out["strands/opacity"], out["strands/label"] = m.get_opacity.numpy(), m.get_label.numpy()
    out["strands/orient_conf"] = m.get_orient_conf.numpy()

    # ---- densification with optimizer-state surgery (src/scene/gaussian_model.py:560-741) on the reference's own
    # GaussianModel + torch.optim.Adam: two Adam steps on seeded gradients to populate the moments, seeded
    # densification statistics, then densify_and_prune (clone + split with torch.normal + prune) and reset_opacity.
    from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
    spec = syn.CONFIGS["tiny"]
    p = syn.random_gaussian_params(spec.P, spec.seed, spec.log_scale_mean)
    m = ref_gm.GaussianModel(3)
    par = lambda t: torch.nn.Parameter(t.detach().clone().float().contiguous().requires_grad_(True))
    m._xyz, m._scaling, m._rotation = par(p["xyz"]), par(p["log_scales"]), par(p["rotations"])
    m._opacity, m._label = par(p["opacity_logit"].reshape(-1, 1)), par(p["label_logit"].reshape(-1, 1))
    m._orient_conf = par(p["orient_conf_log"].reshape(-1, 1))
    m._features_dc, m._features_rest = par(p["features"][:, :1]), par(p["features"][:, 1:])
    m.max_radii2D = torch.zeros(spec.P)
    m.spatial_lr_scale = 1.0
    opt = OptimizationParams()
    m.training_setup(opt)
    gd = torch.Generator().manual_seed(4242)
    for it in range(2):
        for grp in m.optimizer.param_groups:
            prm = grp["params"][0]
            prm.grad = torch.randn(prm.shape, generator=gd) * 1e-3
        m.update_learning_rate(it + 1)
        m.optimizer.step()
    out["densify/pre_xyz"], out["densify/pre_label"] = m._xyz.detach().numpy().copy(), m._label.detach().numpy().copy()
    m.xyz_gradient_accum = torch.rand(spec.P, 1, generator=gd) * 1.2e-3
    m.denom = torch.randint(0, 4, (spec.P, 1), generator=gd).float()
    m.max_radii2D = torch.rand(spec.P, generator=gd) * 30
    out["densify/accum"], out["densify/denom"] = m.xyz_gradient_accum.numpy(), m.denom.numpy()
    out["densify/max_radii"] = m.max_radii2D.numpy()
    torch.manual_seed(991)
    with torch.no_grad():
        m.densify_and_prune(opt.densify_grad_threshold, 0.005, 2.5, 20)
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_label", "_scaling", "_rotation", "_orient_conf")
    for nme in names:
        out["densify/after" + nme] = getattr(m, nme).detach().numpy().copy()
    for grp in m.optimizer.param_groups:
        st = m.optimizer.state[grp["params"][0]]
        out["densify/m_" + grp["name"]] = st["exp_avg"].numpy().copy()
        out["densify/v_" + grp["name"]] = st["exp_avg_sq"].numpy().copy()
    out["densify/n_after"] = np.int64(m.get_xyz.shape[0])
    with torch.no_grad():
        m.reset_opacity()
    out["densify/opacity_reset"] = m._opacity.detach().numpy().copy()
    out["densify/m_opacity_reset"] = m.optimizer.state[m._opacity]["exp_avg"].numpy().copy()
    # one more Adam step on the resized state (the surgery must leave a steppable optimizer)
    for grp in m.optimizer.param_groups:
        prm = grp["params"][0]
        prm.grad = torch.randn(prm.shape, generator=gd) * 1e-3
    m.optimizer.step()
    out["densify/xyz_after_step"] = m._xyz.detach().numpy()

    g = torch.Generator().manual_seed(123)
    q = torch.randn(64, 4, generator=g)
    out["util/q"] = q.numpy()
    out["util/build_rotation"] = ref_general.build_rotation(q).numpy()
    s = torch.rand(64, 3, generator=g) + 0.1
    L = ref_general.build_scaling_rotation(s, q)
    out["util/s"] = s.numpy()
    out["util/scaling_rotation"] = L.numpy()
    out["util/strip_symmetric"] = ref_general.strip_symmetric(L.transpose(1, 2) @ L).numpy()
    a, b = torch.randn(32, 3, generator=g), torch.randn(32, 3, generator=g)
    out["util/pt_a"], out["util/pt_b"] = a.numpy(), b.numpy()
    out["util/parallel_transport"] = ref_general.parallel_transport(a, b).numpy()
    f = ref_general.get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 10, 100, 1000, 15000, 29999, 30000, 40000])
    out["util/lr_steps"] = steps
    out["util/lr_values"] = np.array([f(int(t)) for t in steps], dtype=np.float64)
    fovx, fovy = torch.tensor(0.9), torch.tensor(0.6)
    out["util/projection"] = ref_graphics.getProjectionMatrix(0.01, 100.0, fovx, fovy).numpy()
    Rm = ref_general.build_rotation(q[:1])[0].numpy().astype(np.float64)
    out["util/w2v"] = ref_graphics.getWorld2View2(Rm, np.array([0.3, -0.2, 4.0]))
    out["util/w2v_R"] = Rm

    H, W = 48, 64
    img1, img2 = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.3).float()
    out["loss/img1"], out["loss/img2"], out["loss/mask"] = img1.numpy(), img2.numpy(), mask.numpy()
    out["loss/l1"] = np.float64(ref_loss.l1_loss(img1, img2))
    out["loss/l1_masked"] = np.float64(ref_loss.l1_loss(img1, img2, mask=mask))
    out["loss/ssim"] = np.float64(ref_loss.ssim(img1, img2))
    ang1, ang2 = torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)
    conf = torch.rand(1, H, W, generator=g) + 0.2
    out["loss/ang1"], out["loss/ang2"], out["loss/conf"] = ang1.numpy(), ang2.numpy(), conf.numpy()
    out["loss/or"] = np.float64(ref_loss.or_loss(ang1, ang2, conf, weight=torch.ones_like(mask) * 0.7, mask=mask))
    out["loss/or_noconf"] = np.float64(ref_loss.or_loss(ang1, ang2))

    path = os.path.join(HERE, "reference_host_golden.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
