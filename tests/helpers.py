"""Shared test helpers: oracle drivers, the CPU host-sim of the product's device functions, comparison rules.

Comparison rules (the parity bar of BASELINE.json's north_star, tolerance 1e-4):
* integer / index outputs (radii, tile ranges, sorted point lists, n_contrib): bit-exact;
* float images: |a - b| <= 1e-4 * max(1, |b|);
* gradients: per ROW (one Gaussian), |a - b| <= 1e-4 * (|b| + max|b| of the row) + 2e-6 * max|b| of the tensor  (round 1
  scaled by the tensor maximum, under which a small element could be 100 % wrong and pass; the floor covers the fp32
  cancellation noise of rows whose own gradient is ~0);
* "fragile" pixels -- where the oracle's walk took a discrete decision (alpha < 1/255, T < 1e-4) within a relative
  margin 2e-5 of its threshold, so a 1-ulp different exp() may legitimately decide otherwise -- are excluded from the
  exact image comparison (their count is bounded) and get dL/dpixel = 0 in gradient tests, which removes every
  contribution of those pixels from both sides.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def np32(t):
    import torch
    if isinstance(t, torch.Tensor):
        return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    return np.ascontiguousarray(t, dtype=np.float32)


def oracle_forward(oracle, ri, mode="A"):
    """ri: dict from synthetic.raster_inputs (torch tensors).  mode A: conic (+cov3D) precomputed; B_sr: scales+rot
    in kernel; B_cov: cov3D_precomp in kernel."""
    kw = _mode_kwargs(ri, mode)
    return oracle.rasterize_forward(np32(ri["bg"]), np32(ri["means3D"]), np32(ri["colors"]), np32(ri["opacities"]),
                                    np32(ri["viewmatrix"]), np32(ri["projmatrix"]), ri["tanfovx"], ri["tanfovy"],
                                    ri["H"], ri["W"], **kw)


def oracle_backward(oracle, st, ri, dL, mode="A"):
    kw = _mode_kwargs(ri, mode)
    return oracle.rasterize_backward(st, np32(ri["bg"]), np32(ri["means3D"]), np32(ri["colors"]),
                                     np32(ri["viewmatrix"]), np32(ri["projmatrix"]), ri["tanfovx"], ri["tanfovy"],
                                     np32(dL), **kw)


def _mode_kwargs(ri, mode):
    if mode == "A":
        return dict(cov3D_precomp=np32(ri["cov3D"]), conic_precomp=np32(ri["conic"]))
    if mode == "A_sr":  # render_hair(): scales/rot AND conic given (kernel ignores scales/rot)
        return dict(scales=np32(ri["scales"]), rotations=np32(ri["rotations"]), conic_precomp=np32(ri["conic"]))
    if mode == "B_sr":
        return dict(scales=np32(ri["scales"]), rotations=np32(ri["rotations"]))
    if mode == "B_cov":
        return dict(cov3D_precomp=np32(ri["cov3D"]))
    raise ValueError(mode)


def image_close(a, b, tol=TOL):
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


GRAD_FLOOR = 2e-6  # of the tensor's largest |ref|


def grad_close(a, b, tol=TOL, floor=GRAD_FLOOR):
    """a, b: [rows, ...] (one row per Gaussian)."""
    a2, b2 = a.reshape(len(a), -1) if a.ndim > 1 else a.reshape(-1, 1), b.reshape(len(b), -1) if b.ndim > 1 else b.reshape(-1, 1)
    rows = np.abs(b2).max(axis=1, keepdims=True) if b2.size else 0.0
    scale = np.abs(b2).max() if b2.size else 0.0
    return (np.abs(a2 - b2) <= tol * (np.abs(b2) + rows) + floor * scale + 1e-30).reshape(a.shape)


def assert_grads_close(got: dict, ref: dict, keys=None, tol=TOL):
    for k in (keys or ref.keys()):
        a, b = np.asarray(got[k], dtype=np.float32), np.asarray(ref[k], dtype=np.float32)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.isfinite(a).all(), k
        ok = grad_close(a, b, tol)
        if not ok.all():
            a1, b1, o1 = a.reshape(-1), b.reshape(-1), ok.reshape(-1)
            i = int(np.argmax(np.abs(a1 - b1) * ~o1))
            raise AssertionError("%s: %d / %d elements off; worst idx %d got %g ref %g (tensor max %g)" %
                                 (k, (~o1).sum(), o1.size, i, a1[i], b1[i], np.abs(b1).max()))


# --------------------------------------------------------------------------------------------------------------------
class _ViewArgs(ctypes.Structure):
    _fields_ = [
        ("P", ctypes.c_int32), ("W", ctypes.c_int32), ("H", ctypes.c_int32), ("C", ctypes.c_int32),
        ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("colors", ctypes.c_void_p),
        ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p), ("rotations", ctypes.c_void_p),
        ("cov3D_precomp", ctypes.c_void_p), ("conic_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
        ("projmatrix", ctypes.c_void_p), ("scale_modifier", ctypes.c_float), ("tan_fovx", ctypes.c_float),
        ("tan_fovy", ctypes.c_float), ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
        ("img_ws_recycled", ctypes.c_int32),
    ]


class HostSim:
    """CPU execution of the product's host+device functions (tests/hostsim/ghr_hostsim.cpp)."""

    @staticmethod
    def build():
        """Compile only (no dlopen): __graft_entry__.build() must not load a second HIP-linked library before torch."""
        return HostSim._build()

    def __init__(self):
        import torch  # noqa: F401  -- load torch's HIP runtime first so every HIP-linked .so shares ONE runtime
        so = HostSim._build()
        self._load(so)

    @staticmethod
    def _build():
        src = os.path.join(ROOT, "tests", "hostsim", "ghr_hostsim.cpp")
        out_dir = os.path.join(ROOT, "tests", "hostsim", "_build")
        so = os.path.join(out_dir, "libghr_hostsim.so")
        deps = [src] + [os.path.join(ROOT, "gaussianhaircut_amd", "csrc", f)
                        for f in os.listdir(os.path.join(ROOT, "gaussianhaircut_amd", "csrc")) if f.endswith(".h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            os.makedirs(out_dir, exist_ok=True)
            hipcc = "/opt/rocm/bin/hipcc"
            subprocess.run([hipcc, "--offload-arch=gfx950", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off",
                            "-fPIC", "-shared", "-o", so, src], check=True)
        return so

    def _load(self, so):
        L = ctypes.CDLL(so)
        L.ghrsim_forward.restype = ctypes.c_void_p
        L.ghrsim_num_rendered.restype = ctypes.c_uint32
        for n in ("rec", "depths", "tile_start", "point_list", "keys", "final_T", "n_contrib"):
            getattr(L, "ghrsim_" + n).restype = ctypes.c_void_p
            getattr(L, "ghrsim_" + n).argtypes = [ctypes.c_void_p]
        L.ghrsim_num_rendered.argtypes = [ctypes.c_void_p]
        L.ghrsim_free.argtypes = [ctypes.c_void_p]
        self.L = L

    def _args(self, ri, mode, keep):
        a = _ViewArgs()
        arrs = {k: np32(ri[k]) for k in ("bg", "means3D", "colors", "opacities", "viewmatrix", "projmatrix")}
        opt = {}
        if mode in ("A",):
            opt = dict(cov3D_precomp=np32(ri["cov3D"]), conic_precomp=np32(ri["conic"]))
        elif mode == "A_sr":
            opt = dict(scales=np32(ri["scales"]), rotations=np32(ri["rotations"]), conic_precomp=np32(ri["conic"]))
        elif mode == "B_sr":
            opt = dict(scales=np32(ri["scales"]), rotations=np32(ri["rotations"]))
        elif mode == "B_cov":
            opt = dict(cov3D_precomp=np32(ri["cov3D"]))
        keep.extend(arrs.values())
        keep.extend(opt.values())
        a.P, a.W, a.H, a.C = arrs["means3D"].shape[0], ri["W"], ri["H"], 10
        a.background = arrs["bg"].ctypes.data
        a.means3D = arrs["means3D"].ctypes.data
        a.colors = arrs["colors"].ctypes.data
        a.opacities = arrs["opacities"].ctypes.data
        a.viewmatrix = arrs["viewmatrix"].ctypes.data
        a.projmatrix = arrs["projmatrix"].ctypes.data
        for k in ("scales", "rotations", "cov3D_precomp", "conic_precomp"):
            setattr(a, k, opt[k].ctypes.data if k in opt else None)
        a.scale_modifier, a.tan_fovx, a.tan_fovy = 1.0, ri["tanfovx"], ri["tanfovy"]
        a.prefiltered, a.debug = 1, 0
        return a

    def forward(self, ri, mode="A"):
        keep = []
        a = self._args(ri, mode, keep)
        P, W, H = a.P, a.W, a.H
        radii = np.zeros(P, np.int32)
        out = np.zeros((10, H, W), np.float32)
        h = self.L.ghrsim_forward(ctypes.byref(a), ctypes.c_void_p(radii.ctypes.data), ctypes.c_void_p(out.ctypes.data))
        h = ctypes.c_void_p(h)
        R = int(self.L.ghrsim_num_rendered(h))
        T = ((W + 15) // 16) * ((H + 15) // 16)

        def view(name, dtype, n):
            p = getattr(self.L, "ghrsim_" + name)(h)
            if n == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))),
                                         shape=(n,)).copy()

        st = dict(handle=h, args=a, keep=keep, R=R, radii=radii, out=out,
                  rec=view("rec", np.float32, 16 * P).reshape(P, 16), depths=view("depths", np.float32, P),
                  tile_start=view("tile_start", np.uint32, T + 1), point_list=view("point_list", np.uint32, R),
                  keys=view("keys", np.uint64, R), final_T=view("final_T", np.float32, H * W),
                  n_contrib=view("n_contrib", np.uint32, H * W))
        return st

    def backward(self, st, dL):
        a = st["args"]
        P = a.P
        dL = np32(dL)
        outs = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
                    dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 10), np.float32),
                    dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
                    dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32))
        order = ["dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dscales",
                 "dL_drotations"]
        self.L.ghrsim_backward(st["handle"], ctypes.byref(a), ctypes.c_void_p(dL.ctypes.data),
                               *[ctypes.c_void_p(outs[k].ctypes.data) for k in order])
        return outs

    def free(self, st):
        self.L.ghrsim_free(st["handle"])


# --------------------------------------------------------------------------------------------------------------------
class ModelArgsC(ctypes.Structure):
    """Mirror of ghr::ModelArgs (gaussianhaircut_amd/csrc/ghr_project.h) for the host-sim."""
    _fields_ = [(n, ctypes.c_int) for n in ("P", "W", "H", "gx", "gy", "sh_degree", "sh_coeffs", "mode", "row0")] + \
               [(n, ctypes.c_void_p) for n in ("xyz", "log_scales", "rotations", "opacity_logit", "label_logit",
                                               "orient_conf_log", "dir3d")] + \
               [(n, ctypes.c_float) for n in ("const_opacity", "const_label", "const_conf")] + \
               [(n, ctypes.c_void_p) for n in ("features_dc", "features_rest", "view", "proj", "campos")] + \
               [(n, ctypes.c_float) for n in ("scale_modifier", "tan_fovx", "tan_fovy", "focal_x", "focal_y",
                                              "conic_eps")] + \
               [("fovx", ctypes.c_void_p), ("fovy", ctypes.c_void_p)] + \
               [(n, ctypes.c_void_p) for n in ("rec", "depths", "rects", "radii", "means2D", "tile_count", "pos", "slot_blk")]


def model_args_from(model, cam, keep, conic_eps=1e-12):
    """Fill ModelArgsC from a (CPU) GaussianModel + Camera; `keep` collects the numpy arrays backing the pointers."""
    import math
    a = ModelArgsC()
    arr = dict(xyz=np32(model._xyz), log_scales=np32(model._scaling), rotations=np32(model._rotation),
               opacity_logit=np32(model._opacity).reshape(-1), label_logit=np32(model._label).reshape(-1),
               orient_conf_log=np32(model._orient_conf).reshape(-1), features_dc=np32(model._features_dc),
               features_rest=np32(model._features_rest), view=np32(cam.world_view_transform).reshape(-1),
               proj=np32(cam.full_proj_transform).reshape(-1), campos=np32(cam.camera_center))
    keep.append(arr)
    for k, v in arr.items():
        setattr(a, k, v.ctypes.data)
    a.P, a.W, a.H = arr["xyz"].shape[0], int(cam.image_width), int(cam.image_height)
    a.sh_degree, a.sh_coeffs = int(model.active_sh_degree), (model.max_sh_degree + 1) ** 2
    a.scale_modifier = 1.0
    a.tan_fovx, a.tan_fovy = math.tan(float(cam.FoVx) * 0.5), math.tan(float(cam.FoVy) * 0.5)
    a.focal_x, a.focal_y = a.W / (2.0 * a.tan_fovx), a.H / (2.0 * a.tan_fovy)
    a.conic_eps = conic_eps
    a.mode, a.row0, a.dir3d = 0, 0, None
    return a
