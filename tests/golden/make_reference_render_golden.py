"""Generates tests/golden/reference_render_golden.npz: the return dict of THE REFERENCE'S OWN ``render()`` /
``render_hair()`` (src/gaussian_renderer/__init__.py:23-113,116-214, imported read-only from /root/reference by
tests/refload.py) calling this repo's drop-in ``diff_gaussian_rasterization`` package with the CPU oracle behind it, plus
the gradients of a seeded linear functional of the outputs with respect to the model's raw parameters.

    python tests/golden/make_reference_render_golden.py        # build container only (needs /root/reference)

Replayed on the GPU through the real library by tests/test_reference_dropin.py (``-m gpu``): that is the drop-in claim
of BASELINE.json's north_star ("so train_gaussians.py and train_strands.py call it as a drop-in") under test.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

PIPE = SimpleNamespace(debug=False, convert_SHs_python=False, compute_cov3D_python=False, fused_projection=False)
PARAMS = ("_xyz", "_scaling", "_rotation", "_opacity", "_label", "_orient_conf", "_features_dc", "_features_rest")
HAIR_PARAMS = ("_dirs", "_features_dc", "_features_rest", "_orient_conf")


def weights(spec, seed):
    return torch.randn(7, spec.H, spec.W, generator=torch.Generator().manual_seed(seed))


def functional(pkg, w):
    full = torch.cat([pkg["render"], pkg["mask"], pkg["orient_conf"], pkg["orient_angle"]], dim=0)
    return (full * w).sum()


def run_render(render_fn, cfg):
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import oracle_backend as ob
    spec = syn.CONFIGS[cfg]
    model, cam = syn.make_model(spec, "cpu"), syn.make_view(spec, "cpu")
    with ob.oracle_rasterizer():
        pkg = render_fn(cam, model, PIPE, syn.background("cpu"))
        frag = ob.LAST["state"].fragile.reshape(spec.H, spec.W).astype(bool)
        w = weights(spec, 5)
        w[:, torch.from_numpy(frag)] = 0.0
        functional(pkg, w).backward()
    out = {k: pkg[k].detach().numpy() for k in ("render", "mask", "orient_angle", "orient_conf", "viewspace_points",
                                                 "visibility_filter", "radii")}
    out["fragile"] = np.packbits(frag.reshape(-1))
    for n in PARAMS:
        out["grad" + n] = getattr(model, n).grad.numpy()
    out["grad_viewspace"] = pkg["viewspace_points"].grad.numpy()
    return out


def run_render_hair(render_hair_fn):
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import oracle_backend as ob
    from tests.test_api_cpu import _hair_scene
    spec, head, hair, cam = _hair_scene("cpu")
    hair.initialize_gaussians_hair()
    with ob.oracle_rasterizer():
        pkg = render_hair_fn(cam, head, hair, PIPE, syn.background("cpu"))
        frag = ob.LAST["state"].fragile.reshape(spec.H, spec.W).astype(bool)
        w = weights(spec, 3)
        w[:, torch.from_numpy(frag)] = 0.0
        functional(pkg, w).backward()
    out = {k: pkg[k].detach().numpy() for k in ("render", "mask", "orient_angle", "orient_conf", "viewspace_points",
                                                 "visibility_filter", "radii")}
    out["fragile"] = np.packbits(frag.reshape(-1))
    for n in HAIR_PARAMS:
        out["grad" + n] = getattr(hair, n).grad.numpy()
    out["grad_viewspace"] = pkg["viewspace_points"].grad.numpy()
    return out


def main():
    from tests import refload
    assert refload.available(), "run in the build container (needs /root/reference)"
    ref = refload.load_reference_renderer()
    out = {}
    for cfg in ("tiny", "tiny_strands"):
        for k, v in run_render(ref.render, cfg).items():
            out["render/%s/%s" % (cfg, k)] = v
    for k, v in run_render_hair(ref.render_hair).items():
        out["render_hair/%s" % k] = v
    dst = os.path.join(HERE, "reference_render_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
