"""Drop-in for the reference's ``diff_gaussian_rasterization`` package, backed by ``libghr_hip.so`` (MI355X / gfx950).

API surface kept verbatim (reference ``ext/diff_gaussian_rasterization_hair/diff_gaussian_rasterization/__init__.py``):

* ``GaussianRasterizationSettings`` -- the 12-field NamedTuple (:170-182)
* ``GaussianRasterizer(raster_settings)`` with ``forward(means3D, means2D, opacities, shs, colors_precomp, scales,
  rotations, cov3D_precomp, conic_precomp) -> (color[C,H,W], radii[P] int32)`` and ``markVisible`` (:184-236)
* ``rasterize_gaussians`` / ``_RasterizeGaussians`` with the same argument order, saved-tensor tuple and gradient
  tuple (:21-168), including the ``[xx, 2*xy, yy]`` restack of the conic gradient (:149-153).

Differences, all deliberate: ``means2D`` is only a gradient sink (the reference's null-check
on it is inverted and its values are never read, forward.cu:199-210); kernels run on torch's *current* stream;
a Gaussian failing the near test is culled silently instead of ``__trap()``-ing.  There is no CPU path: tensors
must live on a ROCm device and the HIP library must be present.
"""
from __future__ import annotations

import ctypes
import os
from typing import NamedTuple

import torch
import torch.nn as nn

try:  # imported as gaussianhaircut_amd.diff_gaussian_rasterization
    from .. import _lib
except ImportError:  # imported as top-level `diff_gaussian_rasterization` (reference-style sys.path layout)
    from gaussianhaircut_amd import _lib

NUM_CHANNELS = _lib.NUM_CHANNELS

# last forward's sizes (read by bench.py for the roofline's algorithmic byte count)
LAST_STATS = {"num_rendered": 0, "P": 0}

# pinned host words that receive num_rendered from stage 1: a ring per device, one word per forward (several forwards
# may be in flight before their counts are read, see PendingCount)
_PIN_SLOTS = 256
_pinned_R = {}


def _pinned(device: torch.device) -> torch.Tensor:
    key = device.index if device.index is not None else torch.cuda.current_device()
    ring = _pinned_R.get(key)
    if ring is None:
        ring = _pinned_R[key] = [torch.zeros(_PIN_SLOTS, dtype=torch.int32).pin_memory(), 0]
    i = ring[1]
    ring[1] = (i + 1) % _PIN_SLOTS
    return ring[0][i:i + 1]


def _ptr(t):
    """Device pointer, or NULL for the reference's 'absent optional' empty tensor (__init__.py:210-222)."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _dev_f32(t, name):
    if t is None or t.numel() == 0:
        return t
    if not t.is_cuda:
        raise RuntimeError("gaussianhaircut_amd: tensor '%s' is on %s; the HIP rasterizer has no CPU path "
                           "(tensors must be on a ROCm device)" % (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def _view_args(rs, P, means3D, colors, opacities, scales, rotations, cov3D, conic, bg, view, proj):
    a = _lib.ViewArgs()
    a.P, a.W, a.H, a.C = int(P), int(rs.image_width), int(rs.image_height), NUM_CHANNELS
    a.background = _ptr(bg)
    a.means3D = _ptr(means3D)
    a.colors = _ptr(colors)
    a.opacities = _ptr(opacities)
    a.scales = _ptr(scales)
    a.rotations = _ptr(rotations)
    a.cov3D_precomp = _ptr(cov3D)
    a.conic_precomp = _ptr(conic)
    a.viewmatrix = _ptr(view)
    a.projmatrix = _ptr(proj)
    a.scale_modifier = float(rs.scale_modifier)
    a.tan_fovx = float(rs.tanfovx)
    a.tan_fovy = float(rs.tanfovy)
    a.prefiltered = int(bool(rs.prefiltered))
    a.debug = int(bool(rs.debug))
    return a


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(dev):
    """``with torch.cuda.device(dev)`` that costs nothing when ``dev`` is the current device already (one process per GPU: always;
    the guard's set / restore pair was ~10 us of Python per use, four uses per view)."""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(dev)


def _stream():
    """The current HIP stream of the current device as a void*.  Through the raw accessor where this PyTorch has it: the
    public route builds a Stream object and resolves the device index in Python every time, ~10 us a call, several calls per
    view (tools/host_profile.py)."""
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ImgLease:
    """The image workspace of one fused forward pass, recycled between passes of the same size on the same stream.

    Stage 1 counts instances per tile into the workspace's counters and needs them at zero: for a fresh buffer that is a
    ~5-us zero-fill launch in front of every view.  A workspace that has been through a whole forward pass has its counters
    back at zero (``k_tile_scan`` turns the counts into append cursors starting at 0, stage 2's tile sort resets them), so a
    pass that gets such a buffer says so (``ghr_model_args.img_ws_recycled`` / ``ghr_view_args.img_ws_recycled``) and the launch is dropped.  The lease lives
    in the autograd node: the buffer goes back to the pool when the graph is freed (after backward, or when the outputs
    go out of scope; a retained graph keeps it), and only if stage 1 AND stage 2 actually ran their kernels on it (a pass
    over an empty model launches nothing: its buffer is never pooled).  Pools are per (device, W, H, stream): the next
    pass on the same stream is ordered behind everything that still reads the buffer."""
    _pools = {}
    MAX_POOLED = 4

    def __init__(self, dev, nbytes, W, H):
        # keyed on the image size, not the byte count: the carve offsets of the counters depend on W x H (two sizes may round
        # to the same number of bytes), and include/ghr.h only allows recycling between passes of the SAME W x H
        self.key = (dev.index, int(W), int(H), torch.cuda.current_stream(dev).cuda_stream)
        pool = _ImgLease._pools.get(self.key)
        if pool:
            self.buf, self.recycled = pool.pop(), True
        else:
            self.buf, self.recycled = torch.empty((nbytes,), dtype=torch.uint8, device=dev), False
        self.complete = False  # set once stage 2 has been launched on this buffer

    def __del__(self):
        try:
            if self.complete and self.buf is not None:
                pool = _ImgLease._pools.setdefault(self.key, [])
                if len(pool) < _ImgLease.MAX_POOLED:
                    pool.append(self.buf)
        except Exception:  # interpreter shutdown
            pass


RECYCLE_IMG_WS = os.environ.get("GHR_RECYCLE_IMG_WS", "1") != "0"  # the rasterizer op's own use of _ImgLease


CAPACITY_GRID = os.environ.get("GHR_CAPACITY_GRID", "1") != "0"  # (0: the exact 1.25 x + 4096 of rounds 1-5, for A/B runs)
_R_HINT = {}    # device index -> binning capacity guess: 1.25 x the largest instance count of the last 64 frames, on a grid
_R_RECENT = {}  # device index -> those counts (cameras of a training set differ; a guess from the last frame alone would
                # overflow every time a wide view follows a narrow one)


_R_P = {}       # device index -> the number of Gaussians (rows handed to the op) those counts belong to: a guess learnt on one
                # model says nothing about another (bench.py alternates a 500k and a 2M model: the 2M model's guess made every
                # view of the small one allocate, sort and gather for 6.5 M instances -- 2.1 ms per step instead of 0.75)


def _note_count(dev_index, R, P):
    import collections
    if _R_P.get(dev_index) != int(P):
        _R_RECENT.pop(dev_index, None)
        _R_P[dev_index] = int(P)
    recent = _R_RECENT.setdefault(dev_index, collections.deque(maxlen=64))
    recent.append(int(R))
    m = max(recent)
    # ... rounded UP to a coarse grid (1/32 .. 1/16 of itself, at least 64k instances): the binning workspace and the gradient
    # lines (76 B per instance: 0.5 GB at 2 M Gaussians) are allocated per view with this capacity, and a guess that creeps up
    # with every new maximum -- the instance count drifts as the model trains -- asks the caching allocator for a slightly
    # larger block every few steps: a hipMalloc of several milliseconds inside the step each time (round 6: bench.py's
    # config5_2M block read 2.0 ms per step, or 2.8 / 3.0 when it hit two of those in its ten steps) and one more cached block
    # nobody can reuse.  On the grid the sizes repeat.
    g = m + m // 4 + 4096
    q = 1 << max(16, g.bit_length() - 5) if CAPACITY_GRID else 1
    _R_HINT[dev_index] = (g + q - 1) // q * q
    LAST_STATS["num_rendered"], LAST_STATS["P"] = R, int(P)


class PendingCount:
    """Instance count of a forward whose stage 2 was launched with a guessed capacity and whose true count has not been
    read yet (``run_stage2(defer=True)``).  ``resolve()`` waits for stage 1 (long finished in practice), returns
    ``(num_rendered, overflow)``; with ``overflow`` the forward's outputs AND everything derived from them (loss,
    gradients) are invalid and must be recomputed -- the kernels stayed inside their buffers (include/ghr.h)."""

    def __init__(self, slot, event, cap, dev_index, P):
        self.slot, self.event, self.cap, self.dev_index, self.P = slot, event, cap, dev_index, P
        self.result = None

    def resolve(self):
        if self.result is None:
            self.event.synchronize()
            R = int(self.slot[0].item())
            self.result = (R, R > self.cap)
            _note_count(self.dev_index, R, self.P)
        return self.result


def run_stage2(dev, P, pinned, launch, defer=False):
    """Reads stage 1's instance count and runs stage 2 (``launch(capacity) -> binning buffer``).  After the first
    frame stage 2 is launched SPECULATIVELY with a capacity guessed from the previous frame before the 4-byte count is
    read back, so the GPU already works on scatter / sort / compositing while the host waits; it is relaunched only if
    the true count exceeds the guess (include/ghr.h, ghr_forward_stage2).  Returns (num_rendered, capacity, buffer).
    ``defer``: do not wait at all -- num_rendered comes back as a ``PendingCount`` the caller resolves later (the
    training step: once per step, after every view is queued)."""
    stream = torch.cuda.current_stream()
    # (no guess for a model of another size than the one the guess was learnt on: that frame waits for its count)
    hint = _R_HINT.get(dev.index) if P > 0 and _R_P.get(dev.index, int(P)) == int(P) else None
    if hint and defer:
        ev = torch.cuda.Event()
        ev.record(stream)
        return PendingCount(pinned, ev, hint, dev.index, P), hint, launch(hint)
    if hint:
        ev = torch.cuda.Event()
        ev.record(stream)      # completes when stage 1's count has landed in pinned memory ...
        binb = launch(hint)    # ... while stage 2 is already queued behind it
        ev.synchronize()
        R, cap = int(pinned[0].item()), hint
        if R > hint:
            binb, cap = launch(R), R
    else:
        stream.synchronize()  # the reference blocks on the same 4 bytes (rasterizer_impl.cu:284-285)
        R = int(pinned[0].item()) if P > 0 else 0
        binb, cap = launch(R), R
    if P > 0:
        _note_count(dev.index, R, P)
    return R, cap, binb


def rasterize_gaussians(means3D, means2D_precomp, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        conics_precomp, raster_settings):
    # autograd.Function.forward always runs with grad mode off and ctx.needs_input_grad ignores torch.no_grad(): whether
    # a backward pass can follow is decided here, outside (an eval render must not allocate and zero gradient lines)
    return _RasterizeGaussians.apply(means3D, means2D_precomp, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, conics_precomp, raster_settings, torch.is_grad_enabled())


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D_precomp, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                conics_precomp, raster_settings, grad_enabled=None):
        rs = raster_settings
        ctx.n_inputs = 10 if grad_enabled is None else 11  # the reference's ten arguments (+ ours)
        grad_enabled = True if grad_enabled is None else bool(grad_enabled)
        L = _lib.lib()
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:59-61
        if not means3D.is_cuda:
            raise RuntimeError("gaussianhaircut_amd: means3D is on %s; the HIP rasterizer has no CPU path" %
                               means3D.device)
        P = means3D.size(0)
        H, W = int(rs.image_height), int(rs.image_width)
        dev = means3D.device

        means3D_c = _dev_f32(means3D, "means3D")
        colors_c = _dev_f32(colors_precomp, "colors_precomp")
        opac_c = _dev_f32(opacities, "opacities")
        scales_c = _dev_f32(scales, "scales")
        rot_c = _dev_f32(rotations, "rotations")
        cov3D_c = _dev_f32(cov3Ds_precomp, "cov3D_precomp")
        conic_c = _dev_f32(conics_precomp, "conic_precomp")
        bg_c = _dev_f32(rs.bg, "bg")
        view_c = _dev_f32(rs.viewmatrix, "viewmatrix")
        proj_c = _dev_f32(rs.projmatrix, "projmatrix")
        if colors_c is None or colors_c.numel() == 0:
            # rasterizer_impl.cu:244-247 (NUM_CHANNELS != 3 forces precomputed colours; the in-kernel SH path is dead)
            if P != 0:
                raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")
        elif colors_c.dim() != 2 or colors_c.size(0) != P or colors_c.size(1) != NUM_CHANNELS:
            raise RuntimeError("colors_precomp must have dimensions (num_points, %d)" % NUM_CHANNELS)

        mode_b = conic_c is None or conic_c.numel() == 0
        with _on_device(dev):
            color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            gbytes, ibytes = _lib.forward_sizes(P, W, H, mode_b)
            geomBuffer = torch.empty((gbytes,), dtype=torch.uint8, device=dev)
            # (the image workspace of a completed earlier pass of the same size on this stream has its per-tile counters back
            # at zero: stage 1 then skips its zero-fill launch -- _ImgLease; round 6: the op as well as the fused path)
            lease = _ImgLease(dev, ibytes, W, H) if (RECYCLE_IMG_WS and P > 0 and not rs.debug) else None
            imgBuffer = lease.buf if lease is not None else torch.empty((ibytes,), dtype=torch.uint8, device=dev)
            args = _view_args(rs, P, means3D_c, colors_c, opac_c, scales_c, rot_c, cov3D_c, conic_c, bg_c, view_c,
                              proj_c)
            args.img_ws_recycled = int(lease is not None and lease.recycled)
            cpu_args = None
            if rs.debug:  # __init__.py:88-95: snapshot the inputs before they can be corrupted
                cpu_args = cpu_deep_copy_tuple((rs.bg, means3D, means2D_precomp, colors_precomp, opacities, scales,
                                                rotations, rs.scale_modifier, cov3Ds_precomp, conics_precomp,
                                                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                                                rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos,
                                                rs.prefiltered, rs.debug))
            try:
                pinned = _pinned(dev)
                _lib.check(L.ghr_forward_stage1(_stream(), ctypes.byref(args), _ptr(geomBuffer), _ptr(imgBuffer),
                                                _ptr(radii), ctypes.c_void_p(pinned.data_ptr())))

                want_grad = grad_enabled and any(ctx.needs_input_grad) and not os.environ.get("GHR_NO_PREZERO")
                f32 = dict(dtype=torch.float32, device=dev)

                def launch(cap):
                    b = torch.empty((_lib.binning_size(cap, W, H),), dtype=torch.uint8, device=dev)
                    # the backward pass's gradient lines, zeroed by stage 2 (include/ghr.h, ghr_forward_stage2)
                    sc = torch.empty((max(int(cap), 1), _lib.GRAD_STRIDE), **f32) if want_grad and cap > 0 else None
                    _lib.check(L.ghr_forward_stage2(_stream(), ctypes.byref(args), cap, _ptr(geomBuffer),
                                                    _ptr(imgBuffer), _ptr(b), _ptr(color), _ptr(sc)))
                    return b, sc

                num_rendered, bin_cap, (binningBuffer, ctx.scratch) = run_stage2(dev, P, pinned, launch)
                if lease is not None:
                    lease.complete = True   # stage 1 and stage 2 ran their kernels on it: the counters end at zero again
                    ctx.img_lease = lease   # (back to the pool when the graph is freed: the backward pass reads the workspace)
            except Exception as ex:
                if cpu_args is not None:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex

        LAST_STATS["num_rendered"], LAST_STATS["P"] = num_rendered, P
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.bin_cap = bin_cap  # layout of binningBuffer
        ctx.scratch_clean = ctx.scratch is not None  # stage 2 zeroed its lines and nothing has touched them since
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # no zeros_like(radii) fill for the integer output on every backward
        # same tuple as the reference (__init__.py:102); tensors are the contiguous fp32 versions the kernels read
        ctx.save_for_backward(colors_c if colors_c is not None else torch.empty(0), means3D_c,
                              scales_c if scales_c is not None else torch.empty(0),
                              rot_c if rot_c is not None else torch.empty(0),
                              cov3D_c if cov3D_c is not None else torch.empty(0),
                              conic_c if conic_c is not None else torch.empty(0), radii,
                              sh if sh is not None else torch.empty(0), geomBuffer, binningBuffer, imgBuffer,
                              opac_c, bg_c, view_c, proj_c)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        num_rendered = ctx.num_rendered
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, conics_precomp, radii, sh, geomBuffer,
         binningBuffer, imgBuffer, opacities, bg, view, proj) = ctx.saved_tensors
        L = _lib.lib()
        P = means3D.size(0)
        if grad_out_color is None:
            grad_out_color = torch.zeros((NUM_CHANNELS, int(rs.image_height), int(rs.image_width)),
                                         dtype=torch.float32, device=means3D.device)
        dev = means3D.device
        f32 = dict(dtype=torch.float32, device=dev)
        with _on_device(dev):
            # the nine gradient tensors as views of ONE allocation (each part 256-B aligned like a tensor of its own): eight
            # allocator calls less on the host's way to the gradient walk's launch -- on a slow host they are what leaves the GPU
            # idle between the forward pass's last kernel and the backward pass's first (profiles/r06f)
            widths = (3, NUM_CHANNELS, 1, 3, 6, 4, 3, 3, 4)
            offs, tot = [], 0
            for w_ in widths:
                offs.append(tot)
                tot += (P * w_ + 63) // 64 * 64
            flat = torch.empty((max(tot, 1),), **f32)
            part = lambda i, shape: flat[offs[i]:offs[i] + P * widths[i]].view(shape)
            grad_means2D = part(0, (P, 3))
            grad_colors_precomp = part(1, (P, NUM_CHANNELS))
            grad_opacities = part(2, (P, 1))
            grad_means3D = part(3, (P, 3))
            grad_cov3Ds_precomp = part(4, (P, 6))
            grad_conic = part(5, (P, 2, 2))
            grad_conics_precomp = part(6, (P, 3))  # [xx, 2 * xy, yy]: written by the library (ghr_backward_ex)
            grad_scales = part(7, (P, 3))
            grad_rotations = part(8, (P, 4))
            scratch = getattr(ctx, "scratch", None)  # one line per instance; zeroed by the forward pass if it made it
            if scratch is None:
                scratch = torch.empty((max(int(num_rendered), 1), _lib.GRAD_STRIDE), **f32)
            # include/ghr.h, ghr_backward: only the FIRST backward over the lines stage 2 zeroed may say so
            prezeroed = int(bool(getattr(ctx, "scratch_clean", False)))
            ctx.scratch_clean = False
            dL = grad_out_color
            if dL.dtype != torch.float32:
                dL = dL.float()
            dL = dL.contiguous()
            args = _view_args(rs, P, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                              conics_precomp, bg, view, proj)
            cpu_args = None
            if rs.debug:
                cpu_args = cpu_deep_copy_tuple((rs.bg, means3D, radii, colors_precomp, scales, rotations,
                                                rs.scale_modifier, cov3Ds_precomp, conics_precomp, rs.viewmatrix,
                                                rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh,
                                                rs.sh_degree, rs.campos, geomBuffer, num_rendered, binningBuffer,
                                                imgBuffer, rs.debug))
            try:
                if P > 0:
                    _lib.check(L.ghr_backward_ex(_stream(), ctypes.byref(args), ctx.bin_cap, _ptr(radii),
                                              _ptr(geomBuffer), _ptr(imgBuffer), _ptr(binningBuffer), _ptr(dL),
                                              _ptr(scratch), _ptr(grad_means2D), _ptr(grad_conic),
                                              _ptr(grad_opacities), _ptr(grad_colors_precomp), _ptr(grad_means3D),
                                              _ptr(grad_cov3Ds_precomp), _ptr(grad_scales), _ptr(grad_rotations),
                                              prezeroed, _ptr(grad_conics_precomp)))
            except Exception as ex:
                if cpu_args is not None:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex

        # __init__.py:149-153: the kernel stores half of d/d(conic.y) (backward.cu:554) and the reference's wrapper restacks
        # [xx, 2 * xy, yy] with three slices, a doubling and a stack; ghr_backward_ex writes that tensor itself
        if P == 0:
            grad_conics_precomp.zero_()

        def opt(g, ref):
            return g if ref.numel() != 0 else None

        grads = (
            grad_means3D,
            grad_means2D,
            None,  # sh: the SH path is dead in this fork (dL_dsh has M = 0 columns in the reference)
            grad_colors_precomp,
            grad_opacities,
            opt(grad_scales, scales),
            opt(grad_rotations, rotations),
            opt(grad_cov3Ds_precomp, cov3Ds_precomp),
            opt(grad_conics_precomp, conics_precomp),
            None,
            None,
        )
        return grads[:ctx.n_inputs]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of points passing the near-plane test (rasterizer_impl.cu:54-66)."""
        with torch.no_grad():
            rs = self.raster_settings
            if not positions.is_cuda:
                raise RuntimeError("gaussianhaircut_amd: markVisible needs a tensor on a ROCm device")
            pos = _dev_f32(positions, "positions")
            P = pos.size(0)
            with _on_device(pos.device):
                present = torch.zeros((P,), dtype=torch.bool, device=pos.device)
                if P:
                    _lib.check(_lib.lib().ghr_mark_visible(_stream(), P, _ptr(pos),
                                                           _ptr(_dev_f32(rs.viewmatrix, "viewmatrix")),
                                                           _ptr(_dev_f32(rs.projmatrix, "projmatrix")), _ptr(present)))
        return present

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, conic_precomp=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        conic_precomp = empty if conic_precomp is None else conic_precomp

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, conic_precomp, raster_settings)
