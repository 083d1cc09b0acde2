"""CPU oracle for the strand-aligned Gaussian rasterizer -- TEST INFRASTRUCTURE ONLY.

Thin numpy/ctypes front end over ``ghr_oracle.c`` (a plain-C restatement of
``ext/diff_gaussian_rasterization_hair/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu`` of the reference).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product package ``gaussianhaircut_amd`` never does.

Parity status: **pinned** to outputs of the reference's own CUDA rasterizer (hipified and compiled for gfx950 by
``oracle/Makefile.ref``, run once on an MI355X -> ``tests/golden/reference_cuda_golden.npz``; checked on the CPU by
``tests/test_reference_cuda_golden.py``: integers bit-identical, images / gradients to 1e-5).  See ``ghr_oracle.c``.

The two entry points mirror the reference's native boundary (``R:rasterize_points.h:18-69``):

* :func:`rasterize_forward`  ~ ``_C.rasterize_gaussians``           (R:rasterize_points.cu:35-123)
* :func:`rasterize_backward` ~ ``_C.rasterize_gaussians_backward``  (R:rasterize_points.cu:125-206)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libghr_oracle.so")
_SO64 = os.path.join(_HERE, "_build", "libghr_oracle64.so")  # the compositing walk in double (ghr_oracle64.c)
_SRC = os.path.join(_HERE, "ghr_oracle.c")
_lib = None
_lib64 = None

NUM_CHANNELS = 10  # R:cuda_rasterizer/config.h:15
FRAG_EPS = 2e-5  # relative decision margin used to flag "fragile" pixels (see ghro_render_forward)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    stale = any((not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(_SRC) for so in (_SO, _SO64))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ghro_scan.restype = ctypes.c_int64
        _lib.ghro_num_threads.restype = ctypes.c_int
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    if a is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(a.ctypes.data)


def _opt(a) -> Optional[np.ndarray]:
    """None / empty -> None (mirrors the reference's empty-tensor => nullptr convention,
    R:diff_gaussian_rasterization/__init__.py:210-222)."""
    if a is None:
        return None
    a = np.asarray(a)
    if a.size == 0:
        return None
    return _f32(a)


@dataclass
class ForwardState:
    """Everything the reference keeps in geomBuffer / binningBuffer / imgBuffer (R:rasterizer_impl.h:29-65)."""
    P: int
    W: int
    H: int
    C: int
    num_rendered: int
    depths: np.ndarray
    radii: np.ndarray
    xy: np.ndarray
    conic_opacity: np.ndarray
    cov3D: np.ndarray
    tiles_touched: np.ndarray
    point_offsets: np.ndarray
    keys_sorted: np.ndarray
    point_list: np.ndarray
    ranges: np.ndarray
    final_T: np.ndarray
    n_contrib: np.ndarray
    fragile: np.ndarray  # oracle-only: pixels whose discrete decisions sit within FRAG_EPS of a threshold


def preprocess(means3D, opacities, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, scales=None, rotations=None,
               scale_modifier=1.0, cov3D_precomp=None, conic_precomp=None):
    """K1 (R:cuda_rasterizer/forward.cu:155-282)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    opacities = _f32(opacities).reshape(-1)
    scales, rotations = _opt(scales), _opt(rotations)
    cov3D_precomp, conic_precomp = _opt(cov3D_precomp), _opt(conic_precomp)
    view, proj = _f32(viewmatrix).reshape(-1), _f32(projmatrix).reshape(-1)
    depths = np.zeros(P, np.float32)
    radii = np.zeros(P, np.int32)
    xy = np.zeros((P, 2), np.float32)
    conic_opacity = np.zeros((P, 4), np.float32)
    cov3D = np.zeros((P, 6), np.float32)
    tiles = np.zeros(P, np.uint32)
    L.ghro_preprocess(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(means3D), _p(opacities), _p(scales),
                      _p(rotations), ctypes.c_float(scale_modifier), _p(cov3D_precomp), _p(conic_precomp), _p(view),
                      _p(proj), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(depths), _p(radii), _p(xy),
                      _p(conic_opacity), _p(cov3D), _p(tiles))
    return depths, radii, xy, conic_opacity, cov3D, tiles


def binning(xy, depths, radii, tiles_touched, H, W):
    """K2-K6 (R:cuda_rasterizer/rasterizer_impl.cu:70-138,281-321)."""
    L = lib()
    P = radii.shape[0]
    offsets = np.zeros(P, np.uint32)
    R = int(L.ghro_scan(ctypes.c_int(P), _p(tiles_touched), _p(offsets)))
    T = ((W + 15) // 16) * ((H + 15) // 16)
    keys = np.zeros(max(R, 1), np.uint64)
    plist = np.zeros(max(R, 1), np.uint32)
    ranges = np.zeros((T, 2), np.uint32)
    L.ghro_binning(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(xy), _p(depths), _p(radii), _p(offsets),
                   ctypes.c_int64(R), _p(keys), _p(plist), _p(ranges))
    return R, offsets, keys[:R], plist[:R], ranges


def render_forward(ranges, point_list, xy, features, conic_opacity, bg, H, W, frag_eps=FRAG_EPS):
    """K7 (R:cuda_rasterizer/forward.cu:287-400)."""
    L = lib()
    features = _f32(features)
    C = features.shape[1]
    out = np.zeros((C, H, W), np.float32)
    final_T = np.zeros(H * W, np.float32)
    n_contrib = np.zeros(H * W, np.uint32)
    fragile = np.zeros(H * W, np.uint8)
    plist = point_list if point_list.size else np.zeros(1, np.uint32)
    L.ghro_render_forward(ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(C), _p(ranges), _p(plist), _p(xy),
                          _p(features), _p(conic_opacity), _p(_f32(bg)), _p(out), _p(final_T), _p(n_contrib),
                          _p(fragile), ctypes.c_float(frag_eps))
    return out, final_T, n_contrib, fragile


def rasterize_forward(bg, means3D, colors, opacities, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, scales=None,
                      rotations=None, scale_modifier=1.0, cov3D_precomp=None, conic_precomp=None):
    """Whole forward (R:cuda_rasterizer/rasterizer_impl.cu:198-340). Returns (out_color[C,H,W], radii[P], state)."""
    colors = _f32(colors)
    if colors.ndim != 2 or colors.shape[0] != np.asarray(means3D).shape[0]:
        raise ValueError("For non-RGB, provide precomputed Gaussian colors!")  # rasterizer_impl.cu:244-247
    depths, radii, xy, con_o, cov3D, tiles = preprocess(means3D, opacities, viewmatrix, projmatrix, tanfovx, tanfovy,
                                                        H, W, scales, rotations, scale_modifier, cov3D_precomp,
                                                        conic_precomp)
    R, offsets, keys, plist, ranges = binning(xy, depths, radii, tiles, H, W)
    out, final_T, n_contrib, fragile = render_forward(ranges, plist, xy, colors, con_o, bg, H, W)
    st = ForwardState(P=radii.shape[0], W=W, H=H, C=colors.shape[1], num_rendered=R, depths=depths, radii=radii,
                      xy=xy, conic_opacity=con_o, cov3D=cov3D, tiles_touched=tiles, point_offsets=offsets,
                      keys_sorted=keys, point_list=plist, ranges=ranges, final_T=final_T, n_contrib=n_contrib,
                      fragile=fragile.reshape(H, W))
    return out, radii, st


def rasterize_backward(st: ForwardState, bg, means3D, colors, viewmatrix, projmatrix, tanfovx, tanfovy, dL_dout,
                       scales=None, rotations=None, scale_modifier=1.0, cov3D_precomp=None, conic_precomp=None):
    """Whole backward (R:cuda_rasterizer/rasterizer_impl.cu:344-441).

    Returns the reference's 9-tuple order minus dL_dsh (dead path):
    dict(dL_dmeans2D[P,3], dL_dcolors[P,C], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dconic[P,2,2],
         dL_dscales[P,3], dL_drotations[P,4]).
    """
    L = lib()
    P, W, H, C = st.P, st.W, st.H, st.C
    means3D, colors = _f32(means3D), _f32(colors)
    dL_dout = _f32(dL_dout)
    scales, rotations = _opt(scales), _opt(rotations)
    cov3D_precomp, conic_precomp = _opt(cov3D_precomp), _opt(conic_precomp)
    view, proj = _f32(viewmatrix).reshape(-1), _f32(projmatrix).reshape(-1)
    g_mean2D = np.zeros((P, 3), np.float32)
    g_conic = np.zeros((P, 4), np.float32)
    g_opac = np.zeros((P, 1), np.float32)
    g_col = np.zeros((P, C), np.float32)
    g_mean3D = np.zeros((P, 3), np.float32)
    g_cov3D = np.zeros((P, 6), np.float32)
    g_scale = np.zeros((P, 3), np.float32)
    g_rot = np.zeros((P, 4), np.float32)
    plist = st.point_list if st.point_list.size else np.zeros(1, np.uint32)
    L.ghro_render_backward(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(C), _p(st.ranges),
                           _p(plist), _p(_f32(bg)), _p(st.xy), _p(st.conic_opacity), _p(colors), _p(st.final_T),
                           _p(st.n_contrib), _p(dL_dout), _p(g_mean2D), _p(g_conic), _p(g_opac), _p(g_col))
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(tanfovy))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(tanfovx))
    if conic_precomp is None:  # backward.cu:588-607
        cov3D_ptr = cov3D_precomp if cov3D_precomp is not None else st.cov3D
        L.ghro_cov2d_backward(ctypes.c_int(P), _p(means3D), _p(st.radii), _p(_f32(cov3D_ptr)),
                              ctypes.c_float(focal_x), ctypes.c_float(focal_y), ctypes.c_float(tanfovx),
                              ctypes.c_float(tanfovy), _p(view), _p(g_conic), _p(g_mean3D), _p(g_cov3D))
    L.ghro_preprocess_backward(ctypes.c_int(P), _p(means3D), _p(st.radii), _p(scales), _p(rotations),
                               ctypes.c_float(scale_modifier), _p(conic_precomp), _p(proj), _p(g_mean2D),
                               _p(g_mean3D), _p(g_cov3D), _p(g_scale), _p(g_rot))
    return dict(dL_dmeans2D=g_mean2D, dL_dcolors=g_col, dL_dopacity=g_opac, dL_dmeans3D=g_mean3D, dL_dcov3D=g_cov3D,
                dL_dconic=g_conic.reshape(P, 2, 2), dL_dscales=g_scale, dL_drotations=g_rot)


def mark_visible(means3D, viewmatrix, projmatrix):
    """R:cuda_rasterizer/rasterizer_impl.cu:54-66,141-153."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    L.ghro_mark_visible(ctypes.c_int(P), _p(means3D), _p(_f32(viewmatrix).reshape(-1)),
                        _p(_f32(projmatrix).reshape(-1)), _p(out))
    return out.astype(bool)


def num_threads() -> int:
    return int(lib().ghro_num_threads())


# ---- the compositing walk in IEEE double (ghr_oracle64.c): arbiter between two fp32 implementations ---------------
def lib64() -> ctypes.CDLL:
    global _lib64
    if _lib64 is None:
        build()
        _lib64 = ctypes.CDLL(_SO64)
    return _lib64


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def render_forward64(ranges, point_list, xy, features, conic_opacity, bg, H, W):
    """K7 in double over GIVEN lists (normally the fp32 oracle's).  Returns (out[C,H,W], final_T[N], n_contrib[N])."""
    L = lib64()
    features = _f64(features)
    C = features.shape[1]
    out = np.zeros((C, H, W), np.float64)
    final_T = np.zeros(H * W, np.float64)
    n_contrib = np.zeros(H * W, np.uint32)
    plist = point_list if point_list.size else np.zeros(1, np.uint32)
    L.ghro64_render_forward(ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(C), _p(np.ascontiguousarray(ranges, np.uint32)),
                            _p(plist), _p(_f64(xy)), _p(features), _p(_f64(conic_opacity)), _p(_f64(bg)), _p(out),
                            _p(final_T), _p(n_contrib), ctypes.c_void_p(0), ctypes.c_double(0.0))
    return out, final_T, n_contrib


def render_backward64(ranges, point_list, bg, xy, conic_opacity, colors, final_T, n_contrib, dL_dout, H, W):
    """K8 in double.  Returns dict(dL_dmeans2D[P,3], dL_dconic[P,2,2], dL_dopacity[P,1], dL_dcolors[P,C])."""
    L = lib64()
    colors = _f64(colors)
    P, C = colors.shape
    g_mean2D = np.zeros((P, 3), np.float64)
    g_conic = np.zeros((P, 4), np.float64)
    g_opac = np.zeros((P, 1), np.float64)
    g_col = np.zeros((P, C), np.float64)
    plist = point_list if point_list.size else np.zeros(1, np.uint32)
    L.ghro64_render_backward(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(C),
                             _p(np.ascontiguousarray(ranges, np.uint32)), _p(plist), _p(_f64(bg)), _p(_f64(xy)),
                             _p(_f64(conic_opacity)), _p(colors), _p(_f64(final_T)),
                             _p(np.ascontiguousarray(n_contrib, np.uint32)), _p(_f64(dL_dout)), _p(g_mean2D), _p(g_conic),
                             _p(g_opac), _p(g_col))
    return dict(dL_dmeans2D=g_mean2D, dL_dconic=g_conic.reshape(P, 2, 2), dL_dopacity=g_opac, dL_dcolors=g_col)
