"""GPU-side drivers for the parity tests: call libghr_hip.so through its C ABI with torch-owned device buffers and
read the workspaces back for stage-by-stage comparison with the oracle."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from gaussianhaircut_amd import _lib
from gaussianhaircut_amd.diff_gaussian_rasterization import _view_args, _ptr, _stream, _pinned
from types import SimpleNamespace


def to_dev(ri, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in ri.items()}


def mode_tensors(ri, mode):
    none = None
    if mode == "A":
        return dict(scales=none, rotations=none, cov3D=ri["cov3D"], conic=ri["conic"])
    if mode == "A_sr":
        return dict(scales=ri["scales"], rotations=ri["rotations"], cov3D=none, conic=ri["conic"])
    if mode == "B_sr":
        return dict(scales=ri["scales"], rotations=ri["rotations"], cov3D=none, conic=none)
    if mode == "B_cov":
        return dict(scales=none, rotations=none, cov3D=ri["cov3D"], conic=none)
    raise ValueError(mode)


def _slice(buf: torch.Tensor, ptr, nbytes, dtype):
    if nbytes == 0 or not ptr:
        return torch.zeros(0, dtype=dtype)
    off = int(ptr) - buf.data_ptr()
    assert 0 <= off and off + nbytes <= buf.numel(), (off, nbytes, buf.numel())
    return buf[off:off + nbytes].view(dtype).cpu()


class GpuRun:
    """One forward (+ optional backward) of the rasterizer through the C ABI, keeping every buffer."""

    def __init__(self, ri, mode="A", debug=True, scratch=None):
        """scratch: the gradient-line buffer a later ghr_backward will get; stage 2 then zeroes it under the tile sort
        (include/ghr.h) -- the product's op does this, the parity tests leave it out on purpose (both ways are tested:
        the op-level tests go through the product path)."""
        L = _lib.lib()
        self.L = L
        self.ri, self.mode = ri, mode
        dev = ri["means3D"].device
        self.dev = dev
        P, W, H = ri["means3D"].shape[0], ri["W"], ri["H"]
        self.P, self.W, self.H = P, W, H
        mt = mode_tensors(ri, mode)
        self.mode_b = mt["conic"] is None
        rs = SimpleNamespace(image_width=W, image_height=H, scale_modifier=1.0, tanfovx=ri["tanfovx"],
                             tanfovy=ri["tanfovy"], prefiltered=True, debug=debug)
        self.opac = ri["opacities"].reshape(-1).contiguous()
        self.args = _view_args(rs, P, ri["means3D"], ri["colors"], self.opac, mt["scales"], mt["rotations"],
                               mt["cov3D"], mt["conic"], ri["bg"], ri["viewmatrix"].contiguous(),
                               ri["projmatrix"].contiguous())
        self._keep = mt
        gb, ib = _lib.forward_sizes(P, W, H, self.mode_b)
        self.geom = torch.zeros(gb, dtype=torch.uint8, device=dev)
        self.img = torch.zeros(ib, dtype=torch.uint8, device=dev)
        self.radii = torch.full((P,), -1, dtype=torch.int32, device=dev)
        self.out = torch.full((10, H, W), float("nan"), dtype=torch.float32, device=dev)
        pinned = _pinned(dev)
        _lib.check(L.ghr_forward_stage1(_stream(), ctypes.byref(self.args), _ptr(self.geom), _ptr(self.img),
                                        _ptr(self.radii), ctypes.c_void_p(pinned.data_ptr())))
        torch.cuda.current_stream().synchronize()
        self.R = int(pinned[0].item()) if P > 0 else 0
        self.bin = torch.zeros(_lib.binning_size(self.R, W, H), dtype=torch.uint8, device=dev)
        _lib.check(L.ghr_forward_stage2(_stream(), ctypes.byref(self.args), self.R, _ptr(self.geom), _ptr(self.img),
                                        _ptr(self.bin), _ptr(self.out), _ptr(scratch)))
        torch.cuda.synchronize()

    def inspect(self):
        v = _lib.WsView()
        _lib.check(self.L.ghr_ws_inspect(self.P, self.W, self.H, int(self.mode_b), self.R, _ptr(self.geom),
                                         _ptr(self.img), _ptr(self.bin) if self.R else None, ctypes.byref(v)))
        P, N, R = self.P, self.W * self.H, self.R
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        return dict(
            rec=_slice(self.geom, v.rec, 64 * P, torch.float32).numpy().reshape(P, 16),
            depths=_slice(self.geom, v.depths, 4 * P, torch.float32).numpy(),
            rects=_slice(self.geom, v.rects, 16 * P, torch.int32).numpy().view(np.uint32).reshape(P, 4),
            final_T=_slice(self.img, v.final_T, 4 * N, torch.float32).numpy(),
            n_contrib=_slice(self.img, v.n_contrib, 4 * N, torch.int32).numpy().view(np.uint32),
            tile_start=_slice(self.img, v.tile_start, 4 * (T + 1), torch.int32).numpy().view(np.uint32),
            keys=_slice(self.bin, v.keys, 8 * R, torch.int64).numpy().view(np.uint64) if R else np.zeros(0, np.uint64),
            point_list=_slice(self.bin, v.point_list, 4 * R, torch.int32).numpy().view(np.uint32) if R else
            np.zeros(0, np.uint32),
        )

    def backward(self, dL: torch.Tensor, scratch=None, prezeroed=0):
        """scratch / prezeroed: see include/ghr.h, ghr_backward (default: a NaN-filled scratch the kernel zero-fills)."""
        P, dev = self.P, self.dev
        f = dict(dtype=torch.float32, device=dev)
        nan = float("nan")
        o = dict(dL_dmeans2D=torch.full((P, 3), nan, **f), dL_dconic=torch.full((P, 2, 2), nan, **f),
                 dL_dopacity=torch.full((P, 1), nan, **f), dL_dcolors=torch.full((P, 10), nan, **f),
                 dL_dmeans3D=torch.full((P, 3), nan, **f), dL_dcov3D=torch.full((P, 6), nan, **f),
                 dL_dscales=torch.full((P, 3), nan, **f), dL_drotations=torch.full((P, 4), nan, **f))
        if scratch is None:
            scratch = torch.full((max(self.R, 1), 16), nan, **f)
        dL = dL.to(dev).float().contiguous()
        # through ghr_backward_ex (ABI 14: ghr_backward is the same call with dL_dconic3 = NULL); the extra output must be
        # the reference wrapper's restack of dL_dconic (diff_gaussian_rasterization/__init__.py:149-153), bit for bit
        conic3 = torch.full((P, 3), nan, **f)
        _lib.check(self.L.ghr_backward_ex(_stream(), ctypes.byref(self.args), self.R, _ptr(self.radii), _ptr(self.geom),
                                          _ptr(self.img), _ptr(self.bin) if self.R else None, _ptr(dL), _ptr(scratch),
                                          _ptr(o["dL_dmeans2D"]), _ptr(o["dL_dconic"]), _ptr(o["dL_dopacity"]),
                                          _ptr(o["dL_dcolors"]), _ptr(o["dL_dmeans3D"]), _ptr(o["dL_dcov3D"]),
                                          _ptr(o["dL_dscales"]), _ptr(o["dL_drotations"]), int(prezeroed), _ptr(conic3)))
        torch.cuda.synchronize()
        c = o["dL_dconic"]
        assert torch.equal(conic3, torch.stack([c[:, 0, 0], 2 * c[:, 0, 1], c[:, 1, 1]], dim=-1))
        return {k: v.cpu().numpy() for k, v in o.items()}


def inspect_fused(renders_packed, P, W, H, R):
    """Workspaces of the FUSED render() path (k_project -> ... -> k_render_fwd), read through ghr_ws_inspect from the
    tensors its autograd node saved (gaussian_renderer/fused.py: ..., radii, geom, img, binb)."""
    saved = renders_packed.grad_fn.saved_tensors
    radii, geom, img, binb = saved[-4], saved[-3], saved[-2], saved[-1]
    # the binning workspace is laid out by the CAPACITY stage 2 ran with (a speculative launch uses the previous frames'
    # guess, include/ghr.h), which the autograd node remembers; R itself only bounds what is read back
    cap = int(getattr(renders_packed.grad_fn, "cap", R))
    v = _lib.WsView()
    _lib.check(_lib.lib().ghr_ws_inspect(P, W, H, 0, cap, _ptr(geom), _ptr(img), _ptr(binb) if R else None,
                                         ctypes.byref(v)))
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    return dict(
        radii=radii.cpu().numpy(),
        rec=_slice(geom, v.rec, 64 * P, torch.float32).numpy().reshape(P, 16),
        depths=_slice(geom, v.depths, 4 * P, torch.float32).numpy(),
        rects=_slice(geom, v.rects, 16 * P, torch.int32).numpy().view(np.uint32).reshape(P, 4),
        final_T=_slice(img, v.final_T, 4 * N, torch.float32).numpy(),
        n_contrib=_slice(img, v.n_contrib, 4 * N, torch.int32).numpy().view(np.uint32),
        tile_start=_slice(img, v.tile_start, 4 * (T + 1), torch.int32).numpy().view(np.uint32),
        point_list=_slice(binb, v.point_list, 4 * R, torch.int32).numpy().view(np.uint32) if R else
        np.zeros(0, np.uint32))
