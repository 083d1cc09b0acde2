"""Generates tests/golden/reference_camera_golden.npz: gradients with respect to the CAMERA of THE REFERENCE'S OWN
``render()`` / ``render_hair()`` (src/gaussian_renderer/__init__.py:23-113,116-214, imported read-only from /root/reference
by tests/refload.py, running on this repo's drop-in ``diff_gaussian_rasterization`` with the CPU oracle behind it).

    python tests/golden/make_reference_camera_golden.py        # build container only (needs /root/reference)

The reference trains its cameras by default (src/arguments/__init__.py:61-62, run.sh:112-115): ``world_view_transform``,
``full_proj_transform``, ``camera_center``, ``FoVx`` and ``FoVy`` are functions of trainable residuals
(src/scene/cameras.py:85-151) and the projection graph is differentiable with respect to all five
(src/scene/gaussian_model.py:258-266,279-294,332-335; src/gaussian_renderer/__init__.py:59).  Here the five tensors are
leaves that require grad; the golden holds their ``.grad`` after ``functional(pkg, w).backward()``, from the reference's own
fp32 chain ("gradcam_*") and from the same chain in IEEE double ("grad64cam_*": this repo's render() on float64 copies with
oracle/ghr_oracle64.c compositing over the fp32 oracle's lists) -- the arbiter the GPU replay is judged against
(tests/test_gpu_camera_grads.py).  Raw-parameter gradients ride along so that the replay also shows that asking for
camera gradients changes nothing else.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden.make_reference_render_golden import HAIR_PARAMS, PARAMS, PIPE, functional, weights  # noqa: E402

CAM_LEAVES = ("world_view_transform", "full_proj_transform", "camera_center", "FoVx", "FoVy")
# (workload, camera).  "ring13roll_narrow": view 13 of the rolled ring through a 14 x 9 degree field of view, so that part of
# the Gaussians lies outside the 1.3 tan(FoV / 2) clamp of tx / tz, ty / tz and the clamp's tensor bounds receive gradient
CAMERA_CASES = [("tiny", "ring13roll"), ("tiny", "ring13roll_narrow"), ("tiny_strands", "ring5")]
HAIR_CAMS = ["ring13roll"]


def camera_for(spec, name, device="cpu"):
    from gaussianhaircut_amd.scene.cameras import Camera
    from gaussianhaircut_amd.utils import synthetic as syn
    if name.endswith("_narrow"):
        c = syn.make_view(spec, "cpu", name[: -len("_narrow")])
        return Camera(c.R, c.T, math.radians(14.0), math.radians(9.0), spec.W, spec.H, device=device)
    return syn.make_view(spec, device, name)


def leaf_camera(cam):
    """every tensor of the camera the projection reads becomes a leaf that requires grad (in place)"""
    for n in CAM_LEAVES:
        setattr(cam, n, getattr(cam, n).detach().clone().requires_grad_(True))
    return cam


def tag(cfg, cam):
    return "render/%s@%s/" % (cfg, cam)


def hair_tag(cam):
    return "render_hair@%s/" % cam


def _cam_grads(cam, prefix, dst):
    for n in CAM_LEAVES:
        g = getattr(cam, n).grad
        dst[prefix + n] = (torch.zeros_like(getattr(cam, n)) if g is None else g).detach().numpy().astype(np.float32)


def run_render_cam(render_fn, cfg, camname):
    from gaussianhaircut_amd.gaussian_renderer import render as our_render
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import oracle_backend as ob
    spec = syn.CONFIGS[cfg]
    model, cam = syn.make_model(spec, "cpu"), leaf_camera(camera_for(spec, camname))
    out = {}
    with ob.oracle_rasterizer():
        pkg = render_fn(cam, model, PIPE, syn.background("cpu"))
        st32 = ob.LAST["state"]
        frag = st32.fragile.reshape(spec.H, spec.W).astype(bool)
        m64, c64 = ob.double_chain(model, cam, model.filter_points(cam))
        leaf_camera(c64)
        with ob.oracle_rasterizer64(st32):
            p64 = our_render(c64, m64, PIPE, syn.background("cpu").double())
            frag = frag | (ob.LAST["n_contrib64"] != st32.n_contrib).reshape(spec.H, spec.W)
        w = weights(spec, 5)
        w[:, torch.from_numpy(frag)] = 0.0
        functional(pkg, w).backward()
        with ob.oracle_rasterizer64(st32):
            functional(p64, w.double()).backward()
    _cam_grads(cam, "gradcam_", out)
    _cam_grads(c64, "grad64cam_", out)
    for n in PARAMS:
        out["grad" + n] = getattr(model, n).grad.numpy()
        out["grad64" + n] = getattr(m64, n).grad.numpy().astype(np.float32)
    out["radii"] = pkg["radii"].numpy()
    out["fragile"] = np.packbits(frag.reshape(-1))
    return out


def run_render_hair_cam(render_hair_fn, camname):
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import oracle_backend as ob
    from tests.test_api_cpu import _hair_scene
    spec, head, hair, cam = _hair_scene("cpu", camname)
    leaf_camera(cam)
    hair.initialize_gaussians_hair()
    out = {}
    with ob.oracle_rasterizer():
        pkg = render_hair_fn(cam, head, hair, PIPE, syn.background("cpu"))
        frag = ob.LAST["state"].fragile.reshape(spec.H, spec.W).astype(bool)
        w = weights(spec, 3)
        w[:, torch.from_numpy(frag)] = 0.0
        functional(pkg, w).backward()
    _cam_grads(cam, "gradcam_", out)
    for n in HAIR_PARAMS:
        out["grad" + n] = getattr(hair, n).grad.numpy()
    out["radii"] = pkg["radii"].numpy()
    out["fragile"] = np.packbits(frag.reshape(-1))
    return out


def main():
    from tests import refload
    assert refload.available(), "run in the build container (needs /root/reference)"
    ref = refload.load_reference_renderer()
    out = {}
    for cfg, cam in CAMERA_CASES:
        for k, v in run_render_cam(ref.render, cfg, cam).items():
            out[tag(cfg, cam) + k] = v
    for cam in HAIR_CAMS:
        for k, v in run_render_hair_cam(ref.render_hair, cam).items():
            out[hair_tag(cam) + k] = v
    dst = os.path.join(HERE, "reference_camera_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")
    for k in sorted(out):
        if "cam_" in k:
            print(k, np.abs(out[k]).max())


if __name__ == "__main__":
    main()
