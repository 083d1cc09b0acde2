"""ctypes binding of ``libghr_hip.so`` (C ABI in ``include/ghr.h``) and its in-tree build recipe.

The shared library is the product: there is **no** CPU or eager-PyTorch fallback.  If the library is missing and
cannot be built (no ``hipcc``) importing it raises, and every entry point refuses non-ROCm tensors.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GHR_LIB_PATH") or os.path.join(CSRC, "libghr_hip.so")  # override: kernel experiments
SOURCES = ["ghr_capi.hip"]
HEADERS = ["ghr_device.h", "ghr_preprocess.h", "ghr_binning.h", "ghr_render_fwd.h", "ghr_render_bwd.h", "ghr_render_bwd2.h", "ghr_render_bwd3.h",
           "ghr_geom_bwd.h", "ghr_project.h", "ghr_loss.h", "ghr_adam.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC",
               "-shared"]

NUM_CHANNELS = 10
GRAD_STRIDE = 16
CAM_PARTIALS = 32  # GHR_CAM_PARTIALS: rows of the camera-gradient partial table
CAM_GRADS = 37     # GHR_CAM_GRADS: d view[16] | d proj[16] | d camera_center[3] | d tanfov[2]
STRAND_MAX_SEG = 2048  # GHR_STRAND_MAX_SEG: longest strand ghr_strand_build takes
ADAM_STATE = 18  # GHR_ADAM_STATE
ABI_VERSION = 20  # GHR_ABI_VERSION of include/ghr.h this binding was written for

GHR_OK, GHR_E_INVALID, GHR_E_NOCOLORS, GHR_E_HIP = 0, -1, -2, -3


class GhrError(RuntimeError):
    pass


def _hipcc() -> Optional[str]:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(_HERE, "..", "include", "ghr.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``csrc/libghr_hip.so`` (cross-compiles without a GPU)."""
    if os.environ.get("GHR_LIB_PATH"):
        # an experiment library is used as it is: rebuilding it from the tree's sources (it is always "stale" after the
        # next edit) would silently turn an A/B into a comparison of the product with itself
        if not os.path.exists(LIB_PATH):
            raise GhrError("GHR_LIB_PATH=%s does not exist" % LIB_PATH)
        return LIB_PATH
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    if hipcc is None:
        raise GhrError("libghr_hip.so is missing/stale and hipcc was not found; the HIP extension is required "
                       "(there is no CPU fallback)")
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise GhrError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


class ViewArgs(ctypes.Structure):
    """``ghr_view_args`` (include/ghr.h)."""
    _fields_ = [
        ("P", ctypes.c_int32), ("W", ctypes.c_int32), ("H", ctypes.c_int32), ("C", ctypes.c_int32),
        ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("colors", ctypes.c_void_p),
        ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p), ("rotations", ctypes.c_void_p),
        ("cov3D_precomp", ctypes.c_void_p), ("conic_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
        ("projmatrix", ctypes.c_void_p), ("scale_modifier", ctypes.c_float), ("tan_fovx", ctypes.c_float),
        ("tan_fovy", ctypes.c_float), ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
        ("img_ws_recycled", ctypes.c_int32),
    ]


class ModelArgs(ctypes.Structure):
    """``ghr_model_args`` (include/ghr.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("P", "W", "H", "sh_degree", "sh_coeffs")] + \
               [(n, ctypes.c_void_p) for n in ("xyz", "log_scales", "rotations", "opacity_logit", "label_logit",
                                               "orient_conf_log", "features_dc", "features_rest", "viewmatrix",
                                               "projmatrix", "campos", "background")] + \
               [(n, ctypes.c_float) for n in ("scale_modifier", "tan_fovx", "tan_fovy", "conic_eps")] + \
               [("debug", ctypes.c_int32), ("mode", ctypes.c_int32), ("row0", ctypes.c_int32),
                ("dir3d", ctypes.c_void_p)] + \
               [(n, ctypes.c_float) for n in ("const_opacity", "const_label", "const_conf")] + \
               [("img_ws_recycled", ctypes.c_int32)] + \
               [("fovx_dev", ctypes.c_void_p), ("fovy_dev", ctypes.c_void_p), ("cam_partial", ctypes.c_void_p), ("cam_slot0", ctypes.c_int32),
                ("cam_slots", ctypes.c_int32), ("cam_only", ctypes.c_int32), ("detach_means2D", ctypes.c_int32),
                ("dens_grad_accum", ctypes.c_void_p), ("dens_denom", ctypes.c_void_p), ("dens_max_radii2D", ctypes.c_void_p),
                ("dens_img_ws", ctypes.c_void_p), ("overflow_raises_flag", ctypes.c_int32), ("adam_fuse", ctypes.c_void_p),
                ("d_rgb", ctypes.c_void_p)]


class AdamFuse(ctypes.Structure):
    """``ghr_adam_fuse`` (include/ghr.h): the optimizer update fused into the step's last projection backward."""
    _fields_ = [("n", ctypes.c_int64)] + \
               [(n, ctypes.c_void_p) for n in ("p_in", "m_in", "v_in", "p_out", "m_out", "v_out", "state", "flag", "flag_next")] + \
               [("n_groups", ctypes.c_int32), ("group_end_host", ctypes.c_void_p), ("lr_host", ctypes.c_void_p),
                ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_float)]


class LossArgs(ctypes.Structure):
    """``ghr_loss_args`` (include/ghr.h)."""
    _fields_ = [("W", ctypes.c_int32), ("H", ctypes.c_int32)] + \
               [(n, ctypes.c_void_p) for n in ("image", "mask", "dir2d", "orient_conf", "gt_image", "gt_mask",
                                               "gt_orient_angle", "gt_orient_conf")] + \
               [(n, ctypes.c_float) for n in ("w_l1", "w_ssim", "w_mask", "w_orient")] + \
               [("unmasked_colours", ctypes.c_int32), ("gt_stats", ctypes.c_void_p)]




def loss_sums_floats(W: int, H: int) -> int:
    """floats of the loss kernels' ``sums`` scratch for a W x H image (``ghr_loss_sums_floats``)."""
    return int(lib().ghr_loss_sums_floats(int(W), int(H)))


class WsView(ctypes.Structure):
    """``ghr_ws_view`` (include/ghr.h) -- test introspection."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("rec", "depths", "rects", "cov3D", "final_T", "n_contrib",
                                                "tile_start", "keys", "point_list")]


# Every symbol include/ghr.h declares (the CPU test suite checks the library exports all of them).
EXPORTS = ["ghr_last_error", "ghr_abi_version", "ghr_forward_sizes", "ghr_binning_size", "ghr_forward_stage1",
           "ghr_forward_stage2", "ghr_backward", "ghr_backward_ex", "ghr_mark_visible", "ghr_ws_inspect", "ghr_set_profile_events", "ghr_set_deterministic", "ghr_selftest_wave", "ghr_selftest_math", "ghr_model_forward_stage1",
           "ghr_model_backward", "ghr_model_forward_segment", "ghr_model_forward_finish", "ghr_render_backward",
           "ghr_model_backward_segment", "ghr_camera_slots", "ghr_camera_grad_fold", "ghr_strand_build", "ghr_strand_build_backward", "ghr_strand_build_backward_ex", "ghr_sh_grad_from_views", "ghr_loss_sums_floats", "ghr_loss_forward", "ghr_loss_gt_stats", "ghr_loss_backward", "ghr_adam_step",
           "ghr_adam_step_range", "ghr_adam_step_range_to", "ghr_adam_nan_scan", "ghr_adam_relay_rows", "ghr_adam_fused_finish"]

_lib = None


def lib() -> ctypes.CDLL:
    """Load (building first if needed) the HIP library.  torch must be imported first so that the HIP runtime the
    library binds to (soname libamdhip64.so.7) is the one torch already loaded -- streams and pointers are shared."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (load torch's libamdhip64 first)
    build_library()
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, u32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_float
    L.ghr_last_error.restype = ctypes.c_char_p
    L.ghr_abi_version.restype = ctypes.c_int
    got = int(L.ghr_abi_version())
    if got != ABI_VERSION:
        # a stale build or a GHR_LIB_PATH experiment library with other argument lists: calling it would read garbage
        raise GhrError("%s reports ABI %d, this binding needs %d (rebuild: gaussianhaircut_amd._lib.build_library(force=True))"
                       % (LIB_PATH, got, ABI_VERSION))
    L.ghr_forward_sizes.argtypes = [i32, i32, i32, i32, ctypes.POINTER(ctypes.c_size_t),
                                    ctypes.POINTER(ctypes.c_size_t)]
    L.ghr_binning_size.argtypes = [u32, i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.ghr_forward_stage1.argtypes = [vp, ctypes.POINTER(ViewArgs), vp, vp, vp, vp]
    L.ghr_forward_stage2.argtypes = [vp, ctypes.POINTER(ViewArgs), u32, vp, vp, vp, vp, vp]
    L.ghr_backward.argtypes = [vp, ctypes.POINTER(ViewArgs), u32] + [vp] * 14 + [i32]
    L.ghr_backward_ex.argtypes = [vp, ctypes.POINTER(ViewArgs), u32] + [vp] * 14 + [i32, vp]
    L.ghr_mark_visible.argtypes = [vp, i32, vp, vp, vp, vp]
    L.ghr_set_profile_events.argtypes = [vp, vp, vp, vp]
    L.ghr_set_deterministic.argtypes = [i32]
    L.ghr_selftest_wave.argtypes = [vp, vp, vp]
    L.ghr_selftest_math.argtypes = [vp, i32, vp, vp]
    L.ghr_model_forward_stage1.argtypes = [vp, ctypes.POINTER(ModelArgs), vp, vp, vp, vp, vp]
    L.ghr_loss_sums_floats.argtypes = [i32, i32]
    L.ghr_loss_sums_floats.restype = ctypes.c_size_t
    L.ghr_loss_forward.argtypes = [vp, ctypes.POINTER(LossArgs), vp, vp, vp]
    L.ghr_loss_gt_stats.argtypes = [vp, ctypes.POINTER(LossArgs), vp]
    L.ghr_loss_backward.argtypes = [vp, ctypes.POINTER(LossArgs)] + [vp] * 9
    L.ghr_adam_step.argtypes = [vp, ctypes.c_int64, vp, vp, vp, vp, vp, i32, ctypes.POINTER(ctypes.c_int64),
                                ctypes.POINTER(ctypes.c_float), ctypes.c_double, ctypes.c_double, f32, i32, i32, u32]
    L.ghr_adam_step_range.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, vp, vp, vp, i32,
                                      ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_float), ctypes.c_double,
                                      ctypes.c_double, f32, i32, i32, i32, u32]
    L.ghr_adam_step_range_to.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64] + [vp] * 9 + [
        i32, i32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_float), ctypes.c_double, ctypes.c_double, f32, i32, u32]
    L.ghr_model_backward.argtypes = [vp, ctypes.POINTER(ModelArgs), u32] + [vp] * 15 + [i32, vp, i32]
    L.ghr_model_forward_segment.argtypes = [vp, ctypes.POINTER(ModelArgs), i32, i32, vp, vp, vp, vp]
    L.ghr_model_forward_finish.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    L.ghr_render_backward.argtypes = [vp, i32, i32, i32, u32] + [vp] * 6 + [i32]
    L.ghr_model_backward_segment.argtypes = [vp, ctypes.POINTER(ModelArgs), i32] + [vp] * 13 + [i32, vp, u32, vp, u32]
    L.ghr_camera_slots.argtypes = [i32]
    L.ghr_camera_grad_fold.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ghr_sh_grad_from_views.argtypes = [vp, i32, i32, i32, vp, i32, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp, i32, vp,
                                         ctypes.c_int64]
    L.ghr_adam_relay_rows.argtypes = [vp, i32, vp, ctypes.c_int64, ctypes.c_int64] + [vp] * 10
    L.ghr_adam_fused_finish.argtypes = [vp, vp]
    L.ghr_adam_nan_scan.argtypes = [vp, vp, ctypes.c_int64, vp]
    L.ghr_strand_build.argtypes = [vp, i32, i32, vp, vp, f32, vp, vp, vp]
    L.ghr_strand_build_backward.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.ghr_strand_build_backward_ex.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.ghr_ws_inspect.argtypes = [i32, i32, i32, i32, u32, vp, vp, vp, ctypes.POINTER(WsView)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("ghr_last_error",):
            fn.restype = ctypes.c_int
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != GHR_OK:
        msg = lib().ghr_last_error().decode("utf-8", "replace")
        if rc == GHR_E_NOCOLORS:
            raise RuntimeError(msg)  # same text as the reference's std::runtime_error (rasterizer_impl.cu:246)
        raise GhrError("libghr_hip: %s (code %d)" % (msg, rc))


def forward_sizes(P: int, W: int, H: int, mode_b: bool):
    g, i = ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(lib().ghr_forward_sizes(P, W, H, int(mode_b), ctypes.byref(g), ctypes.byref(i)))
    return int(g.value), int(i.value)


def binning_size(R: int, W: int, H: int) -> int:
    b = ctypes.c_size_t(0)
    check(lib().ghr_binning_size(R, W, H, ctypes.byref(b)))
    return int(b.value)
