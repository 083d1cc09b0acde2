"""Fused fast path of ``render()`` for the free-Gaussian ``GaussianModel``: raw parameters in, 10-channel image out.

One autograd.Function around ``ghr_model_forward_stage1`` + ``ghr_forward_stage2`` / ``ghr_model_backward``
(include/ghr.h): projection, SH, feature assembly, culling, rasterization and -- in backward -- the whole chain back to
the leaf parameters run in hand-written HIP kernels instead of ~60 PyTorch kernels + autograd
(src/gaussian_renderer/__init__.py:29-96 of the reference).
"""
from __future__ import annotations

import ctypes
import os

import torch

from .. import _lib
from . import _tan_half
from ..diff_gaussian_rasterization import _ImgLease, _on_device, LAST_STATS, NUM_CHANNELS, _pinned, _ptr, _stream, run_stage2

RECYCLE_IMG_WS = os.environ.get("GHR_RECYCLE_IMG_WS", "1") != "0"  # see _ImgLease


def _model_args(P, W, H, sh_degree, K, tensors, view, proj, campos, bg, scale_modifier, tanfovx, tanfovy, eps, debug,
                fov_dev=None):
    m = _lib.ModelArgs()
    if fov_dev is not None:  # (FoVx, FoVy) as device scalars: the kernels take tan(FoV / 2) themselves
        m.fovx_dev, m.fovy_dev = _ptr(fov_dev[0]), _ptr(fov_dev[1])
    m.P, m.W, m.H, m.sh_degree, m.sh_coeffs = int(P), int(W), int(H), int(sh_degree), int(K)
    (m.xyz, m.log_scales, m.rotations, m.opacity_logit, m.label_logit, m.orient_conf_log, m.features_dc,
     m.features_rest) = [_ptr(t) for t in tensors]
    m.viewmatrix, m.projmatrix, m.campos, m.background = _ptr(view), _ptr(proj), _ptr(campos), _ptr(bg)
    m.scale_modifier, m.tan_fovx, m.tan_fovy, m.conic_eps = float(scale_modifier), float(tanfovx), float(tanfovy), eps
    m.debug = int(bool(debug))
    return m


class _RenderModelFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, log_scales, rotations, opacity_logit, label_logit, orient_conf_log, f_dc, f_rest,
                screenspace_points, view, proj, campos, fovx, fovy, cfg):
        L = _lib.lib()
        if not xyz.is_cuda:
            raise RuntimeError("gaussianhaircut_amd: parameters are on %s; the HIP renderer has no CPU path" % xyz.device)
        dev = xyz.device
        P, W, H = xyz.shape[0], cfg["W"], cfg["H"]
        params = [t.detach().float().contiguous() for t in (xyz, log_scales, rotations, opacity_logit, label_logit,
                                                            orient_conf_log, f_dc, f_rest)]
        K = 1 + f_rest.shape[1]
        # the camera's tensors are inputs of the op: when they are functions of trainable pose / FoV residuals
        # (src/scene/cameras.py:85-151) the backward pass returns their gradients (_camera_grads)
        cam_in = (view, proj, campos, fovx, fovy)
        view, proj = view.detach().float().contiguous(), proj.detach().float().contiguous()
        campos, bg = campos.detach().float().contiguous(), cfg["bg"].float().contiguous()
        fov = (fovx.detach().float().contiguous(), fovy.detach().float().contiguous()) if fovx is not None else None
        ctx.cam_meta = [(t.shape, t.dtype) if t is not None else None for t in cam_in]
        with _on_device(dev):
            color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            gbytes, ibytes = _lib.forward_sizes(P, W, H, False)
            geom = torch.empty((gbytes,), dtype=torch.uint8, device=dev)
            lease = _ImgLease(dev, ibytes, W, H) if RECYCLE_IMG_WS else None
            img = lease.buf if lease is not None else torch.empty((ibytes,), dtype=torch.uint8, device=dev)
            m = _model_args(P, W, H, cfg["sh_degree"], K, params, view, proj, campos, bg, cfg["scale_modifier"],
                            cfg["tanfovx"], cfg["tanfovy"], cfg["conic_eps"], cfg["debug"], fov)
            m.img_ws_recycled = int(lease is not None and lease.recycled)
            pinned = _pinned(dev)
            _lib.check(L.ghr_model_forward_stage1(_stream(), ctypes.byref(m), _ptr(geom), _ptr(img), _ptr(radii),
                                                  _ptr(screenspace_points.detach()), ctypes.c_void_p(pinned.data_ptr())))
            va = _lib.ViewArgs()
            va.P, va.W, va.H, va.C = P, W, H, NUM_CHANNELS
            va.background = _ptr(bg)
            va.debug = int(bool(cfg["debug"]))

            want_grad = (cfg.get("grad_enabled", True) and any(ctx.needs_input_grad) and
                         not os.environ.get("GHR_NO_PREZERO"))

            def launch(cap):
                b = torch.empty((_lib.binning_size(cap, W, H),), dtype=torch.uint8, device=dev)
                # the backward pass's gradient lines, zeroed by stage 2 (include/ghr.h, ghr_forward_stage2)
                sc = (torch.empty((max(int(cap), 1), _lib.GRAD_STRIDE), dtype=torch.float32, device=dev)
                      if want_grad and cap > 0 else None)
                _lib.check(L.ghr_forward_stage2(_stream(), ctypes.byref(va), cap, _ptr(geom), _ptr(img), _ptr(b),
                                                _ptr(color), _ptr(sc)))
                return b, sc

            # speculative (see diff_gaussian_rasterization.run_stage2); with cfg["defer_count"] R is a PendingCount
            R, cap, (binb, ctx.scratch) = run_stage2(dev, P, pinned, launch, defer=bool(cfg.get("defer_count")))
            if lease is not None:
                # stage 2 has been launched: the counters end up at zero again.  With P == 0 the library returns early from
                # both stages and never touches the workspace (an uninitialised buffer must not enter the pool)
                lease.complete = P > 0
                ctx.img_lease = lease  # (lives as long as the graph: the backward pass reads the workspace)
        cfg["count"] = R  # handed to the caller through render_model_fused (cfg is this call's private dict)
        ctx.cfg, ctx.R, ctx.K, ctx.cap = cfg, R, K, cap
        ctx.scratch_clean = ctx.scratch is not None  # zeroed by stage 2, untouched since
        # the leaf parameters themselves (not the detached views saved below): backward may add straight into their
        # .grad when those alias an optimizer's flat gradient buffer (cfg["grad_sink"])
        ctx.leaves = (xyz, log_scales, rotations, opacity_logit, label_logit, orient_conf_log, f_dc, f_rest)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # no zeros_like(radii) fill for the integer output on every backward
        ctx.fov = fov
        ctx.save_for_backward(*params, view, proj, campos, bg, radii, geom, img, binb)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _):
        L = _lib.lib()
        cfg, R, K = ctx.cfg, ctx.R, ctx.K
        *params, view, proj, campos, bg, radii, geom, img, binb = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((NUM_CHANNELS, cfg["H"], cfg["W"]), dtype=torch.float32, device=radii.device)
        xyz = params[0]
        dev, P = xyz.device, xyz.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        sink = cfg.get("grad_sink")
        direct = sink is not None and P > 0 and all(
            isinstance(t, torch.nn.Parameter) and t.requires_grad and t.grad is not None and t.grad.is_contiguous()
            and t.grad.dtype == torch.float32 and t.grad.shape == t.shape for t in ctx.leaves)
        with _on_device(dev):
            d_m2d = torch.empty((P, 3), **f32)
            if direct:
                # accumulate into the optimizer's flat gradient buffer: no per-parameter AccumulateGrad kernels
                d_xyz, d_ls, d_rot, d_op, d_label, d_conf, d_fdc, d_frest = [t.grad for t in ctx.leaves]
            else:
                d_xyz = torch.empty((P, 3), **f32)
                d_ls = torch.empty((P, 3), **f32)
                d_rot = torch.empty((P, 4), **f32)
                d_op = torch.empty((P, 1), **f32)
                d_label = torch.empty((P, 1), **f32)
                d_conf = torch.empty((P, 1), **f32)
                d_fdc = torch.empty((P, 1, 3), **f32)
                d_frest = torch.empty((P, K - 1, 3), **f32)
            # one line per instance; `cap` lines while the count is still pending (include/ghr.h, ghr_backward)
            scratch = getattr(ctx, "scratch", None)  # made (and zeroed) by the forward pass when it knew of a backward
            if scratch is None:
                rows = max(int(R), 1) if isinstance(R, int) else max(int(ctx.cap), 1)
                scratch = torch.empty((rows, _lib.GRAD_STRIDE), **f32)
            rows = scratch.shape[0]
            # include/ghr.h, ghr_backward: only the FIRST backward over the lines stage 2 zeroed may say so
            prezeroed = int(bool(getattr(ctx, "scratch_clean", False)))
            ctx.scratch_clean = False
            dL = grad_color.float().contiguous()
            m = _model_args(P, cfg["W"], cfg["H"], cfg["sh_degree"], K, params, view, proj, campos, bg,
                            cfg["scale_modifier"], cfg["tanfovx"], cfg["tanfovy"], cfg["conic_eps"], cfg["debug"],
                            ctx.fov)
            want_cam = any(ctx.needs_input_grad[9:14])
            cam_partial = _camera_partials(m, [P], dev) if want_cam and P > 0 else None
            dens = cfg.get("densify_stats")
            if dens is not None and P > 0:
                # the stage-1 loop's per-iteration statistics (train_gaussians.py:161-165) ride along in k_project_bwd
                m.dens_grad_accum, m.dens_denom, m.dens_max_radii2D = [_ptr(t) for t in dens]
                m.dens_img_ws = _ptr(img)  # (a view whose capacity guess overflowed leaves the statistics alone: it is redone)
            fuse = direct and P > 0 and getattr(sink, "_fuse_step", None) is not None
            if fuse:
                # a step whose LAST backward carries the optimizer update (optim.FusedAdam.begin_fused_step): every view checks
                # its instance count on the device (an overflowed speculative pass must raise the step's flag) ...
                m.dens_img_ws, m.overflow_raises_flag = _ptr(img), 1
                if cfg.get("fuse_adam"):  # ... and this one IS the last
                    m.adam_fuse = ctypes.addressof(sink._fuse_step["args"])
            # the step's first gradients into a buffer that is known to hold zeros are assigned, not added (optim.py)
            acc = 1
            if P > 0 and direct and sink.take_known_zero():
                acc = 0
            p_fdc, p_frest = _ptr(d_fdc), _ptr(d_frest)
            fold_first = False
            if P > 0 and direct and getattr(sink, "_views", None) is not None:
                if fuse and cfg.get("fuse_adam"):
                    # the view that carries the optimizer update needs the step's whole SH gradient in the flat buffer: the
                    # earlier views' tables are folded into it first (one launch), this view's terms are added by the kernel
                    fold_first = True
                else:
                    # the SH gradients of this view in factored form (optim.FusedAdam.begin_factored_views): its dL/d(rgb)
                    # table instead of 192 B per Gaussian read, added and written back in the flat gradient
                    m.d_rgb = sink.next_view_slot(campos)
                    p_fdc = p_frest = None
            if P > 0 and direct and sink.concurrent:
                # this view shares the GPU with its neighbours (trainer.training_step): only the kernel that adds into
                # the shared gradient buffer is ordered after the previous view's
                stream = torch.cuda.current_stream()
                _lib.check(L.ghr_render_backward(_stream(), P, cfg["W"], cfg["H"], ctx.cap, _ptr(bg), _ptr(geom),
                                                 _ptr(img), _ptr(binb), _ptr(dL), _ptr(scratch), prezeroed))
                sink.accumulate_begin(stream)
                if fold_first:
                    sink.fold_own_views()
                _lib.check(L.ghr_model_backward_segment(_stream(), ctypes.byref(m), P, _ptr(radii), _ptr(geom),
                                                        _ptr(scratch), _ptr(d_m2d), _ptr(d_xyz), _ptr(d_ls), _ptr(d_rot),
                                                        _ptr(d_op), _ptr(d_label), _ptr(d_conf), p_fdc,
                                                        p_frest, None, acc, sink.nan_flag_ptr(), rows,
                                                        _ptr(binb), ctx.cap))
                sink.accumulate_end(stream)
            elif P > 0:
                if fold_first:
                    sink.fold_own_views()
                _lib.check(L.ghr_model_backward(_stream(), ctypes.byref(m), ctx.cap, _ptr(radii), _ptr(geom), _ptr(img),
                                                _ptr(binb), _ptr(dL), _ptr(scratch), _ptr(d_m2d), _ptr(d_xyz),
                                                _ptr(d_ls), _ptr(d_rot), _ptr(d_op), _ptr(d_label), _ptr(d_conf),
                                                p_fdc, p_frest, acc if direct else 0,
                                                sink.nan_flag_ptr() if direct else None, prezeroed))
            d_cam = _camera_grads(cam_partial, ctx.cam_meta, ctx.needs_input_grad[9:14], dev, ctx.fov) if want_cam else (None,) * 5
        if direct:
            sink.note_direct_backward()
            if fuse and cfg.get("fuse_adam"):
                sink.note_fused_update()
            return (None, None, None, None, None, None, None, None, d_m2d) + d_cam + (None,)
        return (d_xyz, d_ls, d_rot, d_op, d_label, d_conf, d_fdc, d_frest, d_m2d) + d_cam + (None,)


def _camera_partials(m, seg_sizes, dev):
    """The camera-gradient partial table of one view (include/ghr.h, ghr_model_args.cam_partial): one column per 64 Gaussians
    of every segment; points `m` (the first segment's arguments) at its columns.  Returns (table, columns, [slot0 ...])."""
    L = _lib.lib()
    slots = [int(L.ghr_camera_slots(int(n))) for n in seg_sizes]
    total = sum(slots)
    table = torch.empty((_lib.CAM_PARTIALS, max(total, 1)), dtype=torch.float32, device=dev)
    starts = [sum(slots[:i]) for i in range(len(slots))]
    m.cam_partial, m.cam_slot0, m.cam_slots = _ptr(table), starts[0], total
    return table, total, starts


def _camera_grads(cam_partial, meta, needs, dev, fov):
    """Fold the partial table into dL/d(world_view_transform, full_proj_transform, camera_center, FoVx, FoVy) -- what autograd
    hands those tensors in the reference's render() -- shaped like the op's inputs.  One launch; the five results are views of
    one 37-float buffer."""
    d_cam = torch.empty((_lib.CAM_GRADS,), dtype=torch.float32, device=dev)
    if cam_partial is None:
        d_cam.zero_()
    else:
        table, total, _ = cam_partial
        _lib.check(_lib.lib().ghr_camera_grad_fold(_stream(), _ptr(table), total, _ptr(d_cam),
                                                   _ptr(fov[0]) if fov is not None else None,
                                                   _ptr(fov[1]) if fov is not None else None))
    parts = (d_cam[0:16], d_cam[16:32], d_cam[32:35], d_cam[35:36], d_cam[36:37])
    out = []
    for g, mt, need in zip(parts, meta, needs):
        if not need or mt is None:
            out.append(None)
            continue
        g = g.reshape(mt[0])
        out.append(g if mt[1] == torch.float32 else g.to(mt[1]))
    return tuple(out)


def render_model_fused(cam, pc, bg_color, scaling_modifier, debug, defer_count=False, densify_stats=False, fuse_adam=False):
    """Returns (renders[10,H,W], radii[P], screenspace_points[P,3] leaf whose .grad receives dL/d(NDC mean),
    num_rendered: int, or a PendingCount with ``defer_count``).  ``densify_stats``: the backward pass of this view also
    updates the model's ``xyz_gradient_accum`` / ``denom`` / ``max_radii2D`` (the reference's per-iteration bookkeeping,
    train_gaussians.py:161-165), inside k_project_bwd."""
    import math
    xyz = pc.get_xyz
    P = xyz.shape[0]
    # "zero tensor used to make pytorch return gradients of the 2D (screen-space) means" of the original 3DGS;
    # here it also carries the NDC means as values, like the reference's get_mean_2d() output.
    # k_project writes all P rows (culled Gaussians included), so no zero-fill is needed
    screenspace_points = torch.empty((P, 3), dtype=torch.float32, device=xyz.device).requires_grad_(True)
    view, proj, campos, fovx, fovy, tfx, tfy = camera_inputs(cam)
    cfg = dict(W=int(cam.image_width), H=int(cam.image_height), bg=bg_color,
               sh_degree=int(pc.active_sh_degree), scale_modifier=float(scaling_modifier), tanfovx=tfx, tanfovy=tfy,
               conic_eps=float(getattr(pc, "conic_eps", 1e-12)), debug=bool(debug), defer_count=bool(defer_count),
               grad_enabled=torch.is_grad_enabled())  # (inside Function.forward grad mode is always off)
    from ..optim import FusedAdam
    opt = getattr(pc, "optimizer", None)
    if isinstance(opt, FusedAdam) and opt.direct_grads:
        cfg["grad_sink"] = opt
    if fuse_adam and "grad_sink" in cfg:
        cfg["fuse_adam"] = True
    if densify_stats and torch.is_grad_enabled():
        stats = (pc.xyz_gradient_accum, pc.denom, pc.max_radii2D)
        if not all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and
                   t.numel() == P for t in stats):
            raise RuntimeError("densify_stats: xyz_gradient_accum / denom / max_radii2D must be contiguous fp32 device tensors "
                               "of P elements (GaussianModel.training_setup creates them)")
        cfg["densify_stats"] = stats
    renders, radii = _RenderModelFused.apply(xyz, pc._scaling, pc._rotation, pc._opacity, pc._label, pc._orient_conf,
                                             pc._features_dc, pc._features_rest, screenspace_points, view, proj, campos,
                                             fovx, fovy, cfg)
    return renders, radii, screenspace_points, cfg.get("count")


def camera_inputs(cam):
    """(world_view_transform, full_proj_transform, camera_center, FoVx | None, FoVy | None, tan_fovx, tan_fovy) of a camera, each
    property read ONCE (the reference's trainable Camera rebuilds its matrices on every access, src/scene/cameras.py:113-151).
    A FoV that is part of the autograd graph (trainable intrinsics, :93-105) goes to the kernels as two DEVICE scalars -- they
    take tan(FoV / 2) themselves, its gradient comes back through the op, and the host never reads the value; a constant FoV as
    two host floats."""
    if hasattr(cam, "tensors"):  # (scene.cameras.TrainableCamera: all five from one evaluation of the camera's graph)
        view, proj, campos, fx, fy = cam.tensors()[:5]
    else:
        view, proj, campos = cam.world_view_transform, cam.full_proj_transform, cam.camera_center
        fx, fy = cam.FoVx, cam.FoVy
    live = torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in (fx, fy))
    if live:
        dev = view.device
        fx, fy = [t if isinstance(t, torch.Tensor) and t.device == dev else torch.as_tensor(t, dtype=torch.float32, device=dev)
                  for t in (fx, fy)]
        return view, proj, campos, fx, fy, 1.0, 1.0  # (the floats are ignored: fovx_dev / fovy_dev override them)
    return view, proj, campos, None, None, _tan_half(fx), _tan_half(fy)


def _ptr_rows(t, rows: int, row_bytes: int):
    """Base pointer of a per-row array displaced by `rows` rows (may point in front of the allocation: the library only touches
    the rows of the segment it is called for, which lie inside; include/ghr.h, segmented form)."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr() + rows * row_bytes)


# ---------------------------------------------------------------------------------------------------------------------
# Strand stage: render_hair() = frozen head Gaussians + strand Gaussians (reference gaussian_renderer/__init__.py:116-214)
# as TWO segments of one rasterizer state, both projected by the fused kernel in its explicit mode (include/ghr.h,
# ghr_model_forward_segment): no concatenation of 60-float rows, no ~60 PyTorch projection kernels, and the backward
# goes straight to the strand quantities (xyz, scaling, rotation, direction, SH, confidence), from where autograd
# continues to the strand parameters (`_dirs`) through initialize_gaussians_hair().
def _seg_args(P, row0, W, H, sh_degree, K, t, cam_t, cfg, eps, consts):
    m = _lib.ModelArgs()
    m.P, m.W, m.H, m.sh_degree, m.sh_coeffs = int(P), int(W), int(H), int(sh_degree), int(K)
    m.xyz, m.log_scales, m.rotations = _ptr(t["xyz"]), _ptr(t["scaling"]), _ptr(t["rotation"])
    m.opacity_logit = _ptr(t["opacity"]) if t.get("opacity") is not None else None
    m.label_logit = None
    m.orient_conf_log = _ptr(t["conf"]) if t.get("conf") is not None else None
    m.dir3d = _ptr(t["dir"]) if t.get("dir") is not None else None
    m.features_dc, m.features_rest = _ptr(t["fdc"]), _ptr(t["frest"])
    m.viewmatrix, m.projmatrix, m.campos, m.background = [_ptr(x) for x in cam_t[:4]]
    m.scale_modifier, m.tan_fovx, m.tan_fovy = cfg["scale_modifier"], cfg["tanfovx"], cfg["tanfovy"]
    m.conic_eps = eps
    if len(cam_t) > 4 and cam_t[4] is not None:
        m.fovx_dev, m.fovy_dev = _ptr(cam_t[4][0]), _ptr(cam_t[4][1])
    m.debug = int(bool(cfg["debug"]))
    m.mode, m.row0 = 1, int(row0)
    m.const_opacity, m.const_label, m.const_conf = consts
    return m


class _RenderHairFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scaling, rotation, dirs, conf, f_dc, f_rest, screenspace_points, view, proj, campos, fovx, fovy,
                head, cfg):
        L = _lib.lib()
        if not xyz.is_cuda:
            raise RuntimeError("gaussianhaircut_amd: parameters are on %s; the HIP renderer has no CPU path" % xyz.device)
        dev = xyz.device
        W, H = cfg["W"], cfg["H"]
        hair = dict(xyz=xyz, scaling=scaling, rotation=rotation, dir=dirs, conf=conf, fdc=f_dc, frest=f_rest)
        hair = {k: v.detach().float().contiguous() for k, v in hair.items()}
        hair["conf"] = hair["conf"].reshape(-1)
        n_head, n_hair = head["xyz"].shape[0], xyz.shape[0]
        row0 = (n_head + 255) // 256 * 256
        rows = row0 + n_hair
        K = 1 + f_rest.shape[1]
        ctx.cam_meta = [(t.shape, t.dtype) if t is not None else None for t in (view, proj, campos, fovx, fovy)]
        cam_t = [t.detach().float().contiguous() for t in (view, proj, campos, cfg["bg"])]
        cam_t.append((fovx.detach().float().contiguous(), fovy.detach().float().contiguous()) if fovx is not None else None)
        with _on_device(dev):
            color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
            # The reference's per-Gaussian outputs are indexed [head rows, strand rows] WITHOUT the alignment padding between the
            # two segments.  The library indexes radii / means2D_out / d_means2D by workspace row (row0 + i), so the strand
            # segment is handed base pointers displaced by the padding (include/ghr.h allows it): the kernels then write the
            # compact arrays themselves -- three torch.cat, a copy and a 37-MB zero-fill per iteration before (round 6).
            # (radii keeps `rows` entries: the head segment zero-fills its padding rows there, the strand kernel overwrites them)
            pad = row0 - n_head
            radii_ws = torch.empty((rows,), dtype=torch.int32, device=dev)
            m2d = screenspace_points.detach()
            if not (m2d.is_contiguous() and m2d.dtype == torch.float32 and tuple(m2d.shape) == (n_head + n_hair, 3)):
                raise RuntimeError("render_hair: screenspace_points must be a contiguous fp32 [n_head + n_hair, 3] tensor")
            gbytes, ibytes = _lib.forward_sizes(rows, W, H, False)
            geom = torch.empty((gbytes,), dtype=torch.uint8, device=dev)
            img = torch.empty((ibytes,), dtype=torch.uint8, device=dev)
            m_head = _seg_args(n_head, 0, W, H, cfg["sh_degree"], K, head, cam_t, cfg, cfg["eps_head"], (1.0, 0.0, 0.0))
            m_hair = _seg_args(n_hair, row0, W, H, cfg["sh_degree"], K, hair, cam_t, cfg, cfg["eps_hair"], (1.0, 1.0, 0.0))
            pinned = _pinned(dev)
            _lib.check(L.ghr_model_forward_segment(_stream(), ctypes.byref(m_head), rows, 1, _ptr(geom), _ptr(img),
                                                   _ptr(radii_ws), _ptr(m2d)))
            _lib.check(L.ghr_model_forward_segment(_stream(), ctypes.byref(m_hair), rows, 0, _ptr(geom), _ptr(img),
                                                   _ptr_rows(radii_ws, -pad, 4), _ptr_rows(m2d, -pad, 12)))
            _lib.check(L.ghr_model_forward_finish(_stream(), rows, W, H, int(bool(cfg["debug"])), _ptr(geom), _ptr(img),
                                                  ctypes.c_void_p(pinned.data_ptr())))
            va = _lib.ViewArgs()
            va.P, va.W, va.H, va.C = rows, W, H, NUM_CHANNELS
            va.background = _ptr(cam_t[3])
            va.debug = int(bool(cfg["debug"]))

            want_grad = (cfg.get("grad_enabled", True) and any(ctx.needs_input_grad) and
                         not os.environ.get("GHR_NO_PREZERO"))

            def launch(cap):
                b = torch.empty((_lib.binning_size(cap, W, H),), dtype=torch.uint8, device=dev)
                sc = (torch.empty((max(int(cap), 1), _lib.GRAD_STRIDE), dtype=torch.float32, device=dev)
                      if want_grad and cap > 0 else None)
                _lib.check(L.ghr_forward_stage2(_stream(), ctypes.byref(va), cap, _ptr(geom), _ptr(img), _ptr(b),
                                                _ptr(color), _ptr(sc)))
                return b, sc

            R, cap, (binb, ctx.scratch) = run_stage2(dev, rows, pinned, launch)
            radii = radii_ws[:n_head + n_hair]
        LAST_STATS["num_rendered"], LAST_STATS["P"] = R, int(rows)
        ctx.cfg, ctx.R, ctx.K, ctx.cap, ctx.dims = cfg, R, K, cap, (n_head, n_hair, row0, rows)
        ctx.scratch_clean = ctx.scratch is not None
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # no zeros_like(radii) fill for the integer output on every backward
        ctx.fov = cam_t[4]
        ctx.head = head  # (the frozen head contributes to the camera's gradients)
        # the SH feature LEAVES themselves: backward may assign straight into their .grad when those alias an optimizer's flat
        # gradient buffer that is known to hold zeros (cfg["grad_sink"]) -- 48 of the strand model's 52 floats per Gaussian
        ctx.sh_leaves = (f_dc, f_rest)
        ctx.save_for_backward(*[hair[k] for k in ("xyz", "scaling", "rotation", "dir", "conf", "fdc", "frest")], *cam_t[:4],
                              radii_ws, geom, img, binb)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _):
        L = _lib.lib()
        cfg, R, K = ctx.cfg, ctx.R, ctx.K
        n_head, n_hair, row0, rows = ctx.dims
        xyz, scaling, rotation, dirs, conf, fdc, frest, view, proj, campos, bg, radii_ws, geom, img, binb = ctx.saved_tensors
        dev = xyz.device
        f32 = dict(dtype=torch.float32, device=dev)
        W, H = cfg["W"], cfg["H"]
        if grad_color is None:
            grad_color = torch.zeros((NUM_CHANNELS, H, W), **f32)
        with _on_device(dev):
            pad = row0 - n_head
            d_m2d = torch.empty((n_head + n_hair, 3), **f32)  # compact [head rows, strand rows]; see forward
            if n_head > 0:
                d_m2d[:n_head].zero_()  # the head is frozen
            d_xyz, d_sc = torch.empty((n_hair, 3), **f32), torch.empty((n_hair, 3), **f32)
            d_rot, d_dir = torch.empty((n_hair, 4), **f32), torch.empty((n_hair, 3), **f32)
            d_conf = torch.empty((n_hair, 1), **f32)
            # Direct gradients for the SH features (round 6): as leaves of a FusedAdam whose gradient buffer is known to be zero
            # (the step's first backward) they are ASSIGNED in place by the kernel -- autograd's AccumulateGrad otherwise reads
            # the 570 MB of zeros, adds and writes them back (0.43 ms per iteration at the reference's 30 000 strands) -- and the
            # kernel raises the optimizer's non-finite flag for everything it stores (the scan over 52 floats per Gaussian goes)
            sink = cfg.get("grad_sink")
            if sink is not None:
                # (a buffer left undefined by step(zero_grad="defer") is only made whole by a backward that assigns EVERY
                # group; this one assigns two of four: the others would be accumulated into garbage)
                sink.resolve_deferred()
            leaves_ok = (sink is not None and n_hair > 0 and all(
                isinstance(t, torch.nn.Parameter) and t.requires_grad and t.grad is not None and t.grad.is_contiguous() and
                t.grad.dtype == torch.float32 and t.grad.shape == t.shape for t in ctx.sh_leaves))
            # The optimizer update of the SH features inside this backward (round 6: trainer.strand_training_step opened a fused
            # step, FusedAdam.begin_fused_step): the kernel reads their moments and writes parameters + moments of the other
            # buffer set instead of 192 B of gradient per Gaussian for a separate Adam pass to read back; the caller finishes the
            # step once autograd has delivered the strand directions' gradients (finish_fused_step_with_late_groups)
            fuse = bool(leaves_ok and cfg.get("fuse_adam") and getattr(sink, "_fuse_step", None) is not None)
            direct = (not fuse and sink is not None and n_hair > 0 and all(
                isinstance(t, torch.nn.Parameter) and t.requires_grad and t.grad is not None and t.grad.is_contiguous() and
                t.grad.dtype == torch.float32 and t.grad.shape == t.shape for t in ctx.sh_leaves) and sink.take_known_zero())
            if direct:
                d_fdc, d_frest = ctx.sh_leaves[0].grad, ctx.sh_leaves[1].grad
            elif fuse:
                d_fdc = d_frest = None
            else:
                d_fdc, d_frest = torch.empty((n_hair, 1, 3), **f32), torch.empty((n_hair, K - 1, 3), **f32)
            scratch = getattr(ctx, "scratch", None)  # made (and zeroed) by the forward pass when it knew of a backward
            if scratch is None:
                scratch = torch.empty((max(int(R), 1), _lib.GRAD_STRIDE), **f32)
            prezeroed = int(bool(getattr(ctx, "scratch_clean", False)))
            ctx.scratch_clean = False
            dL = grad_color.float().contiguous()
            hair = dict(xyz=xyz, scaling=scaling, rotation=rotation, dir=dirs, conf=conf, fdc=fdc, frest=frest)
            cam_t = [view, proj, campos, bg, ctx.fov]
            m_hair = _seg_args(n_hair, row0, W, H, cfg["sh_degree"], K, hair, cam_t, cfg, cfg["eps_hair"], (1.0, 1.0, 0.0))
            if fuse:
                # (the kernel finds the parameters' moments through the offsets of the RAW parameter arrays inside the flat
                # buffer: the saved copies are the leaves' own storage -- .detach().float().contiguous() of a contiguous fp32
                # parameter is a view)
                m_hair.features_dc, m_hair.features_rest = _ptr(ctx.sh_leaves[0].data), _ptr(ctx.sh_leaves[1].data)
                m_hair.adam_fuse = ctypes.addressof(sink._fuse_step["args"])
            want_cam = any(ctx.needs_input_grad[8:13])
            cam_partial = None
            if want_cam and rows > 0:
                # the camera's gradients are sums over BOTH segments: the head is frozen but seen through the same camera
                # (its NDC means are detached in the reference, gaussian_renderer/__init__.py:136: no term through proj)
                m_head = _seg_args(n_head, 0, W, H, cfg["sh_degree"], K, ctx.head, cam_t, cfg, cfg["eps_head"], (1.0, 0.0, 0.0))
                cam_partial = _camera_partials(m_head, [n_head, n_hair], dev)
                m_head.cam_only, m_head.detach_means2D = 1, 1
                m_hair.cam_partial, m_hair.cam_slot0, m_hair.cam_slots = m_head.cam_partial, cam_partial[2][1], cam_partial[1]
            if rows > 0:
                _lib.check(L.ghr_render_backward(_stream(), rows, W, H, ctx.cap, _ptr(bg), _ptr(geom), _ptr(img),
                                                 _ptr(binb), _ptr(dL), _ptr(scratch), prezeroed))
            if cam_partial is not None and n_head > 0:
                _lib.check(L.ghr_model_backward_segment(_stream(), ctypes.byref(m_head), rows, _ptr(radii_ws), _ptr(geom),
                                                        _ptr(scratch), None, None, None, None, None, None, None, None, None,
                                                        None, 0, None, scratch.shape[0], _ptr(binb), ctx.cap))
            if n_hair > 0:
                _lib.check(L.ghr_model_backward_segment(_stream(), ctypes.byref(m_hair), rows, _ptr_rows(radii_ws, -pad, 4),
                                                        _ptr(geom), _ptr(scratch), _ptr_rows(d_m2d, -pad, 12), _ptr(d_xyz), _ptr(d_sc),
                                                        _ptr(d_rot), None, None, _ptr(d_conf),
                                                        None if fuse else _ptr(d_fdc), None if fuse else _ptr(d_frest),
                                                        _ptr(d_dir), 0, sink.nan_flag_ptr() if (direct or fuse) else None,
                                                        scratch.shape[0], _ptr(binb), ctx.cap))
            d_cam = _camera_grads(cam_partial, ctx.cam_meta, ctx.needs_input_grad[8:13], dev, ctx.fov) if want_cam else (None,) * 5
        if fuse:
            sink.note_direct_backward()
            sink.note_fused_update()
            return (d_xyz, d_sc, d_rot, d_dir, d_conf, None, None, d_m2d) + d_cam + (None, None)
        if direct:
            sink.note_direct_backward()
            return (d_xyz, d_sc, d_rot, d_dir, d_conf, None, None, d_m2d) + d_cam + (None, None)
        return (d_xyz, d_sc, d_rot, d_dir, d_conf, d_fdc, d_frest, d_m2d) + d_cam + (None, None)


def head_segment(pc):
    """Contiguous fp32 tensors of the frozen head Gaussians (the ``*_precomp`` attributes of the reference's
    train_strands.py:65-73) in the layout the fused kernel reads; cached on the model."""
    cache = getattr(pc, "_fused_head_cache", None)
    key = (pc.xyz_precomp.data_ptr(), pc.xyz_precomp.shape[0])
    if cache is None or cache[0] != key:
        sv = pc.shs_view  # [n, 3, K] -> features layout [n, K, 3]
        feats = sv.transpose(1, 2).contiguous().float()
        t = dict(xyz=pc.xyz_precomp.detach().float().contiguous(),
                 scaling=pc.scaling_precomp.detach().float().contiguous(),
                 rotation=pc.rotation_precomp.detach().float().contiguous(),
                 opacity=pc.opacity_precomp.detach().float().reshape(-1).contiguous(),
                 fdc=feats[:, :1].contiguous(), frest=feats[:, 1:].contiguous())
        pc._fused_head_cache = cache = (key, t)
    return cache[1]


def render_hair_fused(cam, pc, pc_hair, bg_color, scaling_modifier, debug, fuse_adam=False):
    """Returns (renders[10,H,W], radii[n_head + n_hair], screenspace_points leaf)."""
    import math
    head = head_segment(pc)
    xyz = pc_hair.get_xyz
    n = head["xyz"].shape[0] + xyz.shape[0]
    screenspace_points = torch.empty((n, 3), dtype=torch.float32, device=xyz.device).requires_grad_(True)
    view, proj, campos, fovx, fovy, tfx, tfy = camera_inputs(cam)
    cfg = dict(W=int(cam.image_width), H=int(cam.image_height), bg=bg_color,
               sh_degree=int(pc_hair.active_sh_degree), scale_modifier=float(scaling_modifier), tanfovx=tfx, tanfovy=tfy,
               eps_head=float(getattr(pc, "conic_eps", 1e-12)), eps_hair=float(getattr(pc_hair, "conic_eps", 1e-7)),
               debug=bool(debug), grad_enabled=torch.is_grad_enabled())
    from ..optim import FusedAdam
    opt = getattr(pc_hair, "optimizer", None)
    if isinstance(opt, FusedAdam) and opt.direct_grads and torch.is_grad_enabled():
        cfg["grad_sink"] = opt
        cfg["fuse_adam"] = bool(fuse_adam)
    renders, radii = _RenderHairFused.apply(xyz, pc_hair.get_scaling, pc_hair._rotation, pc_hair._dir,
                                            pc_hair.get_orient_conf, pc_hair._features_dc, pc_hair._features_rest,
                                            screenspace_points, view, proj, campos, fovx, fovy, head, cfg)
    return renders, radii, screenspace_points
