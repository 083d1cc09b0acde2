"""Trainable cameras through the FUSED path (ABI 17) -- the reference trains camera pose and FoV by default
(/root/reference/src/arguments/__init__.py:61-62, run.sh:112-115; src/scene/cameras.py:85-151; optimizer
src/train_gaussians.py:45-60,183-196), so its projection graph's gradients w.r.t. ``world_view_transform``,
``full_proj_transform``, ``camera_center``, ``FoVx`` and ``FoVy`` are part of the hot path's contract.

The arbiter is tests/golden/reference_camera_golden.npz (tests/golden/make_reference_camera_golden.py): THE REFERENCE'S OWN
``render()`` / ``render_hair()`` with the five camera tensors as leaves, its fp32 gradients and the same chain in IEEE double.

* CPU, anywhere: this package's generic ``render()`` (PyTorch projection, CPU oracle behind the op) reproduces the golden.
* CPU, build container: the reference's own model class passes ``is_free_gaussian_model`` (the fused path's acceptance test).
* ``-m gpu``: the FUSED kernels -- ``k_project_bwd<CAM>`` + ``k_cam_fold`` through the C ABI -- against the golden: every camera
  gradient within 1e-4 of the tensor's largest entry of the double result, plus three times the distance the reference's own
  fp32 chain keeps from it; raw-parameter gradients unchanged by asking for camera gradients (bit for bit); a FOREIGN model
  class (not this package's) with a trainable camera takes the fused path.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gaussianhaircut_amd.gaussian_renderer import _use_fused, _use_fused_hair, is_free_gaussian_model, render, render_hair
from gaussianhaircut_amd.utils import synthetic as syn
from tests import helpers as hp
from tests.golden import make_reference_camera_golden as mk
from tests.golden.make_reference_render_golden import HAIR_PARAMS, PARAMS, functional, weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_camera_golden.npz")
TOL = 1e-4


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _sub(gold, prefix):
    return {k[len(prefix):]: gold[k] for k in gold.files if k.startswith(prefix)}


def _frag(ref, spec):
    return torch.from_numpy(np.unpackbits(ref["fragile"])[: spec.W * spec.H].astype(bool).reshape(spec.H, spec.W))


def _check_cam(cam, ref, tol=TOL, arbiter=True):
    """camera .grad vs the golden: |got - f64| <= tol * max|f64| + 3 |ref32 - f64| (elementwise); without the double chain
    (render_hair) against the reference's fp32 chain at 3 tol"""
    worst = {}
    for n in mk.CAM_LEAVES:
        g = getattr(cam, n).grad
        got = np.zeros_like(ref["gradcam_" + n]) if g is None else g.detach().cpu().numpy().astype(np.float64)
        r32 = ref["gradcam_" + n].astype(np.float64)
        if arbiter:
            r64 = ref["grad64cam_" + n].astype(np.float64)
            bar = tol * np.abs(r64).max() + 3.0 * np.abs(r32 - r64)
            err = np.abs(got - r64)
        else:
            bar = 3.0 * tol * np.abs(r32).max() + 0.0 * r32
            err = np.abs(got - r32)
        assert np.isfinite(got).all(), n
        assert (err <= bar + 1e-30).all(), (n, got, r32, ref.get("grad64cam_" + n))
        worst[n] = float((err / (np.abs(r32).max() + 1e-30)).max())
    return worst


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,camname", mk.CAMERA_CASES)
def test_generic_render_reproduces_the_reference_camera_gradients_on_cpu(oracle_mod, gold, cfg, camname):
    from tests import oracle_backend as ob
    ref = _sub(gold, mk.tag(cfg, camname))
    spec = syn.CONFIGS[cfg]
    model, cam = syn.make_model(spec, "cpu"), mk.leaf_camera(mk.camera_for(spec, camname))
    pipe = SimpleNamespace(debug=False, fused_projection=False)
    with ob.oracle_rasterizer():
        pkg = render(cam, model, pipe, syn.background("cpu"))
        w = weights(spec, 5)
        w[:, _frag(ref, spec)] = 0.0
        functional(pkg, w).backward()
    for n in mk.CAM_LEAVES:
        a, b = getattr(cam, n).grad.numpy(), ref["gradcam_" + n]
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max() + 1e-30, n
    assert np.abs(ref["gradcam_FoVx"]).max() > 0 and np.abs(ref["gradcam_camera_center"]).max() > 0


def test_reference_model_class_is_accepted_by_the_fused_path():
    """build container only: the reference's own scene.gaussian_model.GaussianModel (imported with plyfile / simple_knn
    stubbed, as tests/golden/make_reference_golden.py does) satisfies is_free_gaussian_model; a strand model does not."""
    ref_src = "/root/reference/src/scene/gaussian_model.py"
    if not os.path.isfile(ref_src):
        pytest.skip("needs /root/reference (build container)")
    import importlib
    import importlib.util
    import sys
    import types
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "utils" or k.startswith("utils.") or
             k in ("plyfile", "simple_knn", "simple_knn._C")}
    try:
        sys.modules["plyfile"] = types.SimpleNamespace(PlyData=None, PlyElement=None)
        knn, knn_c = types.ModuleType("simple_knn"), types.ModuleType("simple_knn._C")
        knn_c.distCUDA2 = None
        sys.modules["simple_knn"], sys.modules["simple_knn._C"] = knn, knn_c
        for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[m]
        sys.path.insert(0, "/root/reference/src")
        spec_ = importlib.util.spec_from_file_location("ref_gaussian_model_for_duck_typing", ref_src)
        mod = importlib.util.module_from_spec(spec_)
        spec_.loader.exec_module(mod)
    finally:
        sys.path.remove("/root/reference/src")
        for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k in ("plyfile", "simple_knn", "simple_knn._C")]:
            sys.modules.pop(m, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    spec = syn.CONFIGS["tiny"]
    ours = syn.make_model(spec, "cpu")
    m = mod.GaussianModel(3)
    assert not is_free_gaussian_model(m)  # empty tensors of the wrong shape: not yet a model
    for n in PARAMS:
        setattr(m, n, torch.nn.Parameter(getattr(ours, n).detach().clone()))
    m.active_sh_degree = 2
    assert type(m).__module__ != type(ours).__module__ and is_free_gaussian_model(m)
    m.scaling_activation = torch.abs  # another parametrisation: refused
    assert not is_free_gaussian_model(m)
    from tests.test_api_cpu import _hair_scene
    _, head, hair, _ = _hair_scene()
    assert not is_free_gaussian_model(hair) and is_free_gaussian_model(head)


class ForeignModel:
    """NOT this package's class: the interface of the reference's free-Gaussian model the fused path relies on (raw tensors,
    activations bound by identity, SH counters) -- what a reference user's own ``scene.GaussianModel`` instance looks like."""

    def __init__(self, src, requires_grad=True):
        self.max_sh_degree, self.active_sh_degree = src.max_sh_degree, src.active_sh_degree
        for n in PARAMS:
            setattr(self, n, getattr(src, n).detach().clone().requires_grad_(requires_grad))
        self.scaling_activation, self.opacity_activation, self.label_activation = torch.exp, torch.sigmoid, torch.sigmoid
        self.orient_conf_activation, self.rotation_activation = torch.exp, torch.nn.functional.normalize

    @property
    def get_xyz(self):
        return self._xyz


def test_duck_typing_of_the_fused_path_on_cpu():
    spec = syn.CONFIGS["tiny"]
    ours = syn.make_model(spec, "cpu")
    f = ForeignModel(ours)
    pipe = SimpleNamespace(debug=False)
    assert is_free_gaussian_model(f) and is_free_gaussian_model(ours)
    assert not _use_fused(f, pipe) and not _use_fused(ours, pipe)  # CPU tensors: never the HIP path
    f._features_rest = f._features_rest[:, :8]
    assert not is_free_gaussian_model(f)  # not (max_sh_degree + 1)^2 - 1 coefficients


# ---------------------------------------------------------------------------------------------------------------------
def _fused_was_taken(pkg):
    return getattr(pkg, "count", None) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("foreign", [False, True])
@pytest.mark.parametrize("cfg,camname", mk.CAMERA_CASES)
def test_gpu_fused_camera_gradients_replay_the_reference_golden(gold, cfg, camname, foreign):
    dev = torch.device("cuda:0")
    ref = _sub(gold, mk.tag(cfg, camname))
    spec = syn.CONFIGS[cfg]
    model = syn.make_model(spec, dev)
    if foreign:
        model = ForeignModel(model)
    cam = mk.leaf_camera(mk.camera_for(spec, camname, dev))
    pipe = SimpleNamespace(debug=False)
    assert _use_fused(model, pipe, cam)
    pkg = render(cam, model, pipe, syn.background(dev))
    assert _fused_was_taken(pkg)
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref["radii"])
    w = weights(spec, 5)
    w[:, _frag(ref, spec)] = 0.0
    functional(pkg, w.to(dev)).backward()
    worst = _check_cam(cam, ref)
    print("camera-gradient error / tensor max:", worst)
    g_cam = {n: getattr(model, n).grad.detach().cpu().numpy() for n in PARAMS}
    # raw-parameter gradients: the row criterion against the double chain (as tests/test_reference_dropin.py) ...
    for n in PARAMS:
        a, r32, r64 = g_cam[n].reshape(spec.P, -1), ref["grad" + n].reshape(spec.P, -1), ref["grad64" + n].reshape(spec.P, -1).astype(np.float64)
        rm = np.abs(r64).max(axis=1, keepdims=True)
        bad = np.abs(a - r64) > 3.0 * np.abs(r32 - r64) + hp.TOL * (np.abs(r64) + rm) + 2e-6 * np.abs(r64).max()
        assert not bad.any(), (n, int(bad.sum()))
    # ... and bit-identical to what the camera-less instantiation of the kernel writes (constant camera, same values)
    ghr = __import__("gaussianhaircut_amd")._lib.lib()
    prev = ghr.ghr_set_deterministic(1)
    try:
        outs = []
        for trainable in (True, False):
            m2 = syn.make_model(spec, dev)
            c2 = mk.camera_for(spec, camname, dev)
            if trainable:
                mk.leaf_camera(c2)
            p2 = render(c2, m2, pipe, syn.background(dev))
            functional(p2, w.to(dev)).backward()
            outs.append({n: getattr(m2, n).grad.detach().cpu().numpy() for n in PARAMS})
    finally:
        ghr.ghr_set_deterministic(prev)
    for n in PARAMS:
        assert np.array_equal(outs[0][n], outs[1][n]), n


@pytest.mark.gpu
@pytest.mark.parametrize("camname", mk.HAIR_CAMS)
def test_gpu_fused_render_hair_camera_gradients_replay_the_reference_golden(gold, camname):
    from tests.test_api_cpu import _hair_scene
    dev = torch.device("cuda:0")
    ref = _sub(gold, mk.hair_tag(camname))
    spec, head, hair, cam = _hair_scene(dev, camname)
    mk.leaf_camera(cam)
    hair.initialize_gaussians_hair()
    pipe = SimpleNamespace(debug=False)
    assert _use_fused_hair(head, hair, pipe, cam)
    pkg = render_hair(cam, head, hair, pipe, syn.background(dev))
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref["radii"])
    w = weights(spec, 3)
    w[:, _frag(ref, spec)] = 0.0
    functional(pkg, w.to(dev)).backward()
    _check_cam(cam, ref, arbiter=False)
    assert np.abs(ref["gradcam_world_view_transform"]).max() > 0
    for n in HAIR_PARAMS:
        a, b = getattr(hair, n).grad.detach().cpu().numpy(), ref["grad" + n]
        a, b = a.reshape(len(a), -1), b.reshape(len(b), -1)
        assert (np.abs(a - b) <= 3 * hp.TOL * (np.abs(b) + np.abs(b).max(axis=1, keepdims=True)) + 2e-5 * np.abs(b).max()).all(), n


@pytest.mark.gpu
def test_gpu_trainable_fov_is_never_read_back_by_the_host():
    """camera_inputs(): a FoV inside the autograd graph reaches the kernels as a device tensor (ghr_model_args.tanfov_dev);
    the image equals the constant-FoV render bit for bit when the values agree."""
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["tiny"]
    pipe = SimpleNamespace(debug=False)
    model = syn.make_model(spec, dev)
    c0 = mk.camera_for(spec, "ring13roll", dev)
    with torch.no_grad():
        img0 = render(c0, model, pipe, syn.background(dev)).renders_packed.clone()
    c1 = mk.leaf_camera(mk.camera_for(spec, "ring13roll", dev))
    pkg = render(c1, model, pipe, syn.background(dev))
    # tan on the device (fp32) and math.tan on the host (double, rounded) may differ in the last bit: 1e-6 of the image
    assert (pkg.renders_packed.detach() - img0).abs().max() <= 1e-5 * img0.abs().max()


@pytest.mark.gpu
def test_gpu_trainable_camera_inside_a_training_step_with_the_fused_update():
    """k_project_bwd<CAM, ADAM>: a training step whose camera tensors require grad AND whose last backward carries the optimizer
    update (trainer.training_step, one rank).  Camera gradients and the updated parameters must equal, bit for bit, those of the
    same step with the separate optimizer pass (deterministic gradient walk), and the camera gradients must be those of a plain
    render() + loss + backward without any optimizer."""
    from gaussianhaircut_amd import _lib
    from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
    from gaussianhaircut_amd.trainer import PIPE, make_ground_truth, training_step, view_loss
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["tiny_strands"]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    bg = syn.background(dev)
    gt = syn.make_model(spec, dev)
    with torch.no_grad():
        gt._features_dc.add_(0.3)
    base = mk.camera_for(spec, "ring5", dev)
    make_ground_truth(gt, [base], bg)

    def trainable():
        import copy
        return mk.leaf_camera(copy.copy(base))

    _lib.lib().ghr_set_deterministic(1)
    try:
        outs = {}
        for fused in (True, False):
            model = syn.make_model(spec, dev)
            model.training_setup(opt)
            cam = trainable()
            training_step(model, [cam], bg, opt, 1, fuse_adam=fused)
            torch.cuda.synchronize()
            assert model.optimizer.fused_steps == (1 if fused else 0)
            outs[fused] = ({n: getattr(cam, n).grad.detach().clone() for n in mk.CAM_LEAVES},
                           model.optimizer.flat_param.clone(), model.optimizer.exp_avg.clone())
        for n in mk.CAM_LEAVES:
            assert torch.equal(outs[True][0][n], outs[False][0][n]), n
        assert torch.equal(outs[True][1], outs[False][1]) and torch.equal(outs[True][2], outs[False][2])
        model = syn.make_model(spec, dev)
        cam = trainable()
        pkg = render(cam, model, PIPE, bg)
        view_loss(pkg, cam, opt).backward()
        for n in mk.CAM_LEAVES:
            assert torch.equal(cam.__dict__[n].grad, outs[True][0][n]), n
        assert float(outs[True][0]["world_view_transform"].abs().max()) > 0
    finally:
        _lib.lib().ghr_set_deterministic(0)
