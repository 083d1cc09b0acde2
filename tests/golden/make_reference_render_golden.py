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


# (workload, camera): scene/cameras.py:parity_camera -- "front" has R = I, the others are rotated / rolled ring views
RENDER_CASES = [("tiny", "front"), ("tiny_strands", "front"), ("tiny", "ring13roll"), ("tiny_strands", "ring5")]
HAIR_CAMS = ["front", "ring13roll"]


def render_tag(cfg, cam):
    return "render/%s/" % (cfg if cam == "front" else cfg + "@" + cam)


def hair_tag(cam):
    return "render_hair/" if cam == "front" else "render_hair@%s/" % cam


def run_render(render_fn, cfg, cam="front", arbiter=False):
    """arbiter: also the gradients of the SAME chain evaluated in IEEE double (this repo's render() on float64 copies of
    the parameters, oracle/ghr_oracle64.c over the fp32 oracle's lists) as "grad64_*": T <- T / (1 - alpha) amplifies
    last-bit differences of alpha by alpha / (1 - alpha), so on an ill-conditioned row two correct fp32 chains can sit
    several 1e-4 of the row apart (tiny@ring13roll, Gaussian 1300: the reference's own fp32 chain is 5.6 bars from the
    double result); the GPU replay is judged against the double result with the fp32 golden's own distance as allowance."""
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import oracle_backend as ob
    spec = syn.CONFIGS[cfg]
    model, cam = syn.make_model(spec, "cpu"), syn.make_view(spec, "cpu", cam)
    with ob.oracle_rasterizer():
        pkg = render_fn(cam, model, PIPE, syn.background("cpu"))
        st32 = ob.LAST["state"]
        frag = st32.fragile.reshape(spec.H, spec.W).astype(bool)
        g64, p64 = {}, None
        if arbiter:
            from gaussianhaircut_amd.gaussian_renderer import render as our_render
            m64, c64 = ob.double_chain(model, cam, model.filter_points(cam))
            with ob.oracle_rasterizer64(st32):
                p64 = our_render(c64, m64, PIPE, syn.background("cpu").double())
                # pixels whose stop decision falls the other way in double leave the functional on every side (they join
                # the stored "fragile" mask, which the replay applies to its weights)
                frag = frag | (ob.LAST["n_contrib64"] != st32.n_contrib).reshape(spec.H, spec.W)
        w = weights(spec, 5)
        w[:, torch.from_numpy(frag)] = 0.0
        functional(pkg, w).backward()
        if arbiter:
            with ob.oracle_rasterizer64(st32):
                functional(p64, w.double()).backward()
            for n in PARAMS:
                g64["grad64" + n] = getattr(m64, n).grad.numpy().astype(np.float32)   # (rounded once: 6e-8 << the 1e-4 bar)
            g64["grad64_viewspace"] = p64["viewspace_points"].grad.numpy().astype(np.float32)
    out = {k: pkg[k].detach().numpy() for k in ("render", "mask", "orient_angle", "orient_conf", "viewspace_points",
                                                 "visibility_filter", "radii")}
    out["fragile"] = np.packbits(frag.reshape(-1))
    for n in PARAMS:
        out["grad" + n] = getattr(model, n).grad.numpy()
    out["grad_viewspace"] = pkg["viewspace_points"].grad.numpy()
    out.update(g64)
    return out


def run_render_hair(render_hair_fn, cam="front"):
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import oracle_backend as ob
    from tests.test_api_cpu import _hair_scene
    spec, head, hair, cam = _hair_scene("cpu", cam)
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
    for cfg, cam in RENDER_CASES:
        for k, v in run_render(ref.render, cfg, cam, arbiter=True).items():
            out[render_tag(cfg, cam) + k] = v
    for cam in HAIR_CAMS:
        for k, v in run_render_hair(ref.render_hair, cam).items():
            out[hair_tag(cam) + k] = v
    dst = os.path.join(HERE, "reference_render_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
