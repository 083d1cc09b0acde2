"""The measured gradient step: shape of the reference's stage-1 loop body (``src/train_gaussians.py:96-181``) without
densification / logging / GUI -- lr schedule, render, losses, backward, [grad all-reduce], NaN guard, Adam."""
from __future__ import annotations

import functools
import os
from types import SimpleNamespace
from typing import List, Optional

import torch

from .gaussian_renderer import render
from .parallel import FlatGradBucket
from .utils.loss_utils import l1_loss, or_loss, ssim

PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)


_ONES = {}


def _one_like(loss):
    """A cached scalar 1 to seed backward() with (saves autograd's ones_like fill kernel on every view)."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    return one


OVERLAP_ALL_REDUCE_WITH_ADAM = os.environ.get("GHR_OVERLAP_AR_ADAM", "1") != "0"
# One rank, fused path: the step's LAST projection backward applies the optimizer update itself (optim.FusedAdam.begin_fused_step)
FUSE_STRAND_ADAM = os.environ.get("GHR_FUSE_STRAND_ADAM", "1") != "0"  # strand stage: the SH features' update in the backward
FUSE_ADAM_INTO_BACKWARD = os.environ.get("GHR_FUSE_ADAM", "1") != "0"
DEFER_GRAD_ZEROING = os.environ.get("GHR_DEFER_GRAD_ZEROING", "1") != "0"
CACHE_GT_SSIM_STATS = True  # keep the SSIM window moments of every camera's ground truth (2*3*H*W floats per camera)


def _gt_stats(cam, gt_image, gt_mask, mask_colours):
    """Ground truth and mask are constants of a training view: their SSIM window moments are computed once per camera
    (fused_loss.gt_ssim_stats) and re-used until either tensor is replaced or written."""
    if not CACHE_GT_SSIM_STATS:
        return None
    key = (gt_image._version, gt_mask._version, bool(mask_colours))
    cached = getattr(cam, "_ghr_gt_stats", None)
    # the entry holds the two tensors themselves: identity (not an address that the allocator may hand out again)
    if cached is None or cached[0] is not gt_image or cached[1] is not gt_mask or cached[2] != key:
        from .fused_loss import gt_ssim_stats
        cached = (gt_image, gt_mask, key, gt_ssim_stats(gt_image, gt_mask, mask_colours))
        try:
            cam._ghr_gt_stats = cached
        except AttributeError:
            pass
    return cached[3]


def view_loss(render_pkg, cam, opt, fused=None, scale: float = 1.0):
    """train_gaussians.py:113-140.  On a ROCm device all four terms run as one fused HIP op."""
    image, mask = render_pkg["render"], render_pkg["mask"]
    gt_image, gt_mask = cam.original_image, cam.original_mask
    if fused is None:
        fused = image.is_cuda
    if fused and getattr(render_pkg, "renders_packed", None) is not None:
        # one HIP op on the rasterizer's packed [10,H,W] output, orientation term included (orient_weight =
        # ones_like(gt_mask[:1]) * gt_orient_conf, train_gaussians.py:130)
        from .fused_loss import stage1_loss
        # `scale` (1/V of a V-view batch) is folded into the weights: no separate elementwise kernels around the loss
        return stage1_loss(render_pkg.renders_packed, gt_image, gt_mask, cam.original_orient_angle,
                           cam.original_orient_conf, opt.lambda_dl1 * scale, opt.lambda_dssim * scale,
                           opt.lambda_dmask * scale, opt.lambda_dorient * scale,
                           gt_stats=_gt_stats(cam, gt_image, gt_mask, True))
    if fused and opt.lambda_dorient == 0.0:
        from .fused_loss import photometric_loss
        return photometric_loss(image, mask, gt_image, gt_mask, opt.lambda_dl1 * scale, opt.lambda_dssim * scale,
                                opt.lambda_dmask * scale)
    Ll1 = l1_loss(image, gt_image, mask=gt_mask[1:].detach())
    Lssim = 1.0 - ssim(image * gt_mask[1:], gt_image * gt_mask[1:])
    Lmask = l1_loss(mask, gt_mask)
    loss = Ll1 * opt.lambda_dl1 + Lssim * opt.lambda_dssim + Lmask * opt.lambda_dmask
    if opt.lambda_dorient != 0.0:
        w = torch.ones_like(gt_mask[:1]) * cam.original_orient_conf
        Lorient = or_loss(render_pkg["orient_angle"], cam.original_orient_angle, render_pkg["orient_conf"], weight=w,
                          mask=gt_mask[:1])
        # train_gaussians.py:133: `if torch.isnan(Lorient).any(): Lorient = torch.zeros_like(Ll1)` -- a host decision that
        # drops the term from the graph.  Masking the VALUE (torch.where) would still back-propagate 0/0 through a view
        # whose orientation weights are all zero and poison the other three terms' step.
        if bool(torch.isnan(Lorient).any()):
            Lorient = torch.zeros_like(Ll1)
        loss = loss + Lorient * opt.lambda_dorient
    return loss * scale if scale != 1.0 else loss


_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = (device.index, n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE_STREAMS[key]


def _views_forward_backward(gaussians, cams, background, opt, V, pipe, n_streams, sink, last_pipe=None):
    """render + loss + backward of every view; returns (detached losses, instance counts [int | PendingCount]).
    ``last_pipe``: the pipe of the step's LAST view (it may carry the fused optimizer update)."""
    losses, counts = [], []
    pipes = [pipe] * len(cams)
    if last_pipe is not None and cams:
        pipes[-1] = last_pipe
    if n_streams > 1:
        main = torch.cuda.current_stream(background.device)
        side = _side_streams(background.device, n_streams)
        for s in side:
            s.wait_stream(main)  # parameters as the previous optimizer step left them
        sink.concurrent = True
        try:
            for i, cam in enumerate(cams):
                with torch.cuda.stream(side[i % n_streams]):
                    pipe = pipes[i]
                    pkg = render(cam, gaussians, pipe, background)
                    loss = view_loss(pkg, cam, opt, scale=1.0 / V)
                    loss.backward(gradient=_one_like(loss))
                    _generic_densify_stats(gaussians, pkg, pipe)
                    ld = loss.detach()
                    ld.record_stream(main)
                    losses.append(ld)
                    counts.append(getattr(pkg, "count", None))
        finally:
            sink.concurrent = False
            for s in side:
                main.wait_stream(s)
    else:
        for cam, pipe in zip(cams, pipes):
            pkg = render(cam, gaussians, pipe, background)
            loss = view_loss(pkg, cam, opt, scale=1.0 / V)
            loss.backward(gradient=_one_like(loss))
            _generic_densify_stats(gaussians, pkg, pipe)
            losses.append(loss.detach())
            counts.append(getattr(pkg, "count", None))
    return losses, counts


@torch.no_grad()
def _generic_densify_stats(gaussians, pkg, pipe):
    """pipe.densify_stats on a view whose backward pass did not keep the statistics itself (generic path): the reference's
    PyTorch form (train_gaussians.py:161-165)."""
    if getattr(pipe, "densify_stats", False) and not getattr(pkg, "densify_stats_done", False):
        vis = pkg["visibility_filter"]
        gaussians.update_max_radii(pkg["radii"], vis)
        gaussians.add_densification_stats(pkg["viewspace_points"], vis)


def _training_step(gaussians, cams: List, background, opt, iteration: int, bucket: Optional[FlatGradBucket] = None,
                  global_views: Optional[int] = None, pipe=PIPE, streams: Optional[int] = None,
                  defer_counts: Optional[bool] = None, densify_stats: bool = False, fuse_adam: Optional[bool] = None,
                  views_per_rank: Optional[int] = None):
    """One global gradient step over this rank's views.  Returns the (detached) summed local loss.

    ``streams`` (default 2 with the fused path and more than one view): the views of the step are independent given
    the parameters, so consecutive views run on alternating HIP streams -- the next view's projection / binning /
    compositing / loss fills the CUs the current view's VALU-bound backward leaves idle and vice versa (measured
    3.38 -> 3.08 ms for the 4-view step).  Only the kernels that add into the shared flat gradient buffer are
    chained (FusedAdam.accumulate_begin/end); the optimizer step waits for every stream.

    ``defer_counts`` (default on with the fused path): no view waits for its ``num_rendered`` -- the forward runs with
    the capacity guessed from the previous frame and the counts are checked once, after everything is queued (the host
    never blocks inside the step: 3.10 -> 2.92 ms).  If a count exceeded its capacity the accumulated gradients are
    dropped and the step is recomputed view by view with exact capacities, so the result never depends on the guess.

    ``densify_stats``: every view's backward pass also keeps the per-iteration densification statistics of the stage-1 loop
    (``max_radii2D``, ``xyz_gradient_accum``, ``denom``: src/train_gaussians.py:161-165, half of the iterations of a run) --
    inside ``k_project_bwd`` on the fused path (no extra launch); ``densification_step(..., stats_done=True)`` then only
    densifies / prunes at its interval.  Views that take the generic path get the PyTorch form.

    ``fuse_adam`` (default on: one rank, every view on the fused path): the step's last ``k_project_bwd`` applies the Adam
    update from the gradients it holds in registers (``ghr_adam_fuse``): no 244 B of gradient per Gaussian written and read
    back, no separate optimizer pass.  Same parameters, bit for bit, as the separate pass; the NaN rule stays exact (the
    update is undone on the device when the step's flag is up).  After such a step ``p.grad`` still holds what the previous
    separate pass left there (``zero_grad="defer"`` semantics: undefined) -- pass ``fuse_adam=False`` to log gradient norms.

    ``views_per_rank`` (more than one rank): the LARGEST number of views any rank holds in this step, the same value on every
    rank; default ceil(global views / ranks), which is what ``parallel.shard_views`` deals.  Steps of up to
    ``optim.FACTORED_SH_MAX_VIEWS`` view slots in all send the SH gradients as per-view dL/d(rgb) tables (12 B per Gaussian and
    view, all-gathered; ``FusedAdam.begin_factored_views``) instead of summing 192 B per Gaussian: every rank opens that many
    slots, and a rank with more views than slots fails loudly before any collective."""
    from .optim import FusedAdam
    gaussians.update_learning_rate(iteration)
    # the loss of a view is scaled 1 / V: with more than one rank V defaults to the GLOBAL number of views (the
    # all-reduce sums the ranks' gradients), assuming every rank holds len(cams) of them
    V = global_views or len(cams) * _world_size()
    sink = gaussians.optimizer if isinstance(getattr(gaussians, "optimizer", None), FusedAdam) else None
    if sink is not None and sink.direct_grads:
        # the direct backward needs every leaf of the fused renderer inside the optimizer (e.g. train_orient_conf = False
        # leaves _orient_conf out): otherwise autograd accumulates and the scanning guard applies -- a property of the
        # configuration, identical on every rank
        leaves = (gaussians._xyz, gaussians._scaling, gaussians._rotation, gaussians._opacity, gaussians._label,
                  gaussians._orient_conf, gaussians._features_dc, gaussians._features_rest)
        if not all(isinstance(p, torch.nn.Parameter) and p.grad is not None for p in leaves):
            sink = None
    can_overlap = (sink is not None and sink.direct_grads and len(cams) > 1 and background.is_cuda and
                   not getattr(pipe, "debug", False))
    n_streams = (2 if streams is None else int(streams)) if can_overlap else 0
    fused_sink = (sink is not None and sink.direct_grads and background.is_cuda and not getattr(pipe, "debug", False))
    # every view of this step goes through the fused renderer's direct backward (render() decides per camera: a camera
    # whose tensors require grad takes the generic path, whose gradients arrive through autograd)
    from .gaussian_renderer import _use_fused
    all_direct = bool(fused_sink and cams and all(_use_fused(gaussians, pipe, c) for c in cams))
    if isinstance(getattr(gaussians, "optimizer", None), FusedAdam) and not all_direct:
        # the previous step may have left the gradient buffer undefined (zero_grad="defer" below): only a fused backward
        # redefines it -- a rank without views, or the generic / autograd path, needs the zeros the eager step leaves
        gaussians.optimizer.resolve_deferred()
    defer = fused_sink and (True if defer_counts is None else bool(defer_counts))
    run_pipe = pipe
    if defer or densify_stats:
        run_pipe = SimpleNamespace(**{**vars(pipe), "defer_count": bool(defer), "densify_stats": bool(densify_stats)})
    from .optim import collectives_on
    # (the fused update leaves the gradient buffer undefined, like zero_grad="defer": not for callers who switched that off)
    fuse = (FUSE_ADAM_INTO_BACKWARD and DEFER_GRAD_ZEROING if fuse_adam is None else bool(fuse_adam)) and all_direct and \
        bool(cams) and not collectives_on() and bucket is None
    # Data-parallel steps of few views send the SH gradients -- 48 of the 61 floats per Gaussian -- in factored form: every view's
    # backward leaves its dL/d(rgb) table (12 B per Gaussian) in a slot of the optimizer, the ranks all-gather the slots and each
    # rebuilds the f_dc / f_rest gradients (optim.FusedAdam.begin_factored_views).  The choice decides the sequence of
    # collectives: it depends on the configuration and on the GLOBAL number of views only, never on this rank's cameras.
    from . import optim as _optim
    if sink is not None:
        sink.end_factored_views()  # (a step that died between its backwards and its update)
    slots = int(views_per_rank) if views_per_rank else -(-V // max(_world_size(), 1))
    factored = bool(collectives_on() and fused_sink and OVERLAP_ALL_REDUCE_WITH_ADAM and bucket is None and
                    _optim.FACTORED_SH_REDUCE and 0 < slots * _world_size() <= _optim.FACTORED_SH_MAX_VIEWS and
                    sink.can_factor_views())
    if factored:
        if len(cams) > slots:
            raise RuntimeError("training_step: %d views on this rank, %d view slots per rank (%d global views on %d ranks): "
                               "pass views_per_rank = the largest number of views any rank holds, on every rank" %
                               (len(cams), slots, V, _world_size()))
        sink.begin_factored_views(slots, sh_degree=int(gaussians.active_sh_degree))
    elif (all_direct and bucket is None and _optim.FACTORED_SH_REDUCE and sink.can_factor_views() and
          len(cams) >= (3 if fuse else 2) and (not collectives_on() or OVERLAP_ALL_REDUCE_WITH_ADAM)):
        # one rank, or too many views in all for the gathered form: the rank folds ITS views' tables into the flat gradient
        # once (before the sums / the update; before the last view's backward when that one carries the update) and every other
        # view's backward skips the read-modify-write of 192 B of SH gradients per Gaussian (4 views per rank through the
        # collective branch: 2.74 -> 2.45 ms, profiles/r06k).  A rank's own choice: the sequence of collectives is the plain one.
        sink.begin_factored_views(len(cams), gather=False, sh_degree=int(gaussians.active_sh_degree))
    last_pipe = None
    if fuse:
        sink.cancel_skip()  # (as below: every backward of this step runs after any earlier surgery)
        fuse = sink.can_fuse_step()
    if fuse:
        sink.begin_fused_step()
        last_pipe = SimpleNamespace(**{**vars(run_pipe), "fuse_adam": True})
    try:
        losses, counts = _views_forward_backward(gaussians, cams, background, opt, V, run_pipe, n_streams, sink, last_pipe)
    finally:
        fused_done = sink.end_fused_step() if fuse else False
    overflow = [c.resolve()[1] for c in counts if hasattr(c, "resolve")]  # resolve every one: they feed the next guess
    if any(overflow):
        # a guessed capacity was too small: that view's image, loss and gradients are garbage (memory-safe garbage).
        # Drop everything this step accumulated and redo it with blocking, exactly sized forwards.
        sink._direct_backwards = 0
        sink.zero()
        sink.state_dev[1:2].zero_()
        sink._acc_event = None
        # (densify_stats: an overflowed view's backward pass left the statistics alone -- k_project_bwd checks the count on the
        # device -- but the step's OTHER views have been counted, and ALL views are recomputed below: theirs would be counted twice)
        if densify_stats and len(cams) > 1 and not all(overflow):
            raise RuntimeError("training_step(densify_stats=True): a capacity guess overflowed in a multi-view step; the "
                               "statistics of its other views cannot be rolled back -- use defer_counts=False for the first "
                               "step after the scene has grown")
        redo_pipe = SimpleNamespace(**{**vars(pipe), "densify_stats": True}) if densify_stats else pipe
        losses, counts = _views_forward_backward(gaussians, cams, background, opt, V, redo_pipe, 0, sink)
        if fused_done:  # (the overflowed views raised the step's flag: the fused update was undone on the device)
            sink.fused_steps -= 1
            fused_done = False
    if not losses:  # a rank without views in this step still takes part in the collectives and the update
        total = torch.zeros((), device=background.device)
    else:
        total = losses[0] if len(losses) == 1 else torch.stack(losses).sum()
    if fused_done:
        return total  # the last view's backward carried the update (k_adam_v4 did not run)
    if isinstance(gaussians.optimizer, FusedAdam):
        # Every backward of this step ran AFTER any earlier optimizer surgery (densification / opacity reset between two
        # calls), so all groups hold fresh gradients: the "parameters replaced since the last backward" marks never apply
        # here, whichever path produced the gradients (generic pipe, autograd accumulation, a rank without views).  They
        # only matter for a hand-written loop that runs backward -> surgery -> step (the reference's order), which calls
        # optimizer.step() itself.  Unconditional, hence identical on every rank.
        gaussians.optimizer.cancel_skip()
        # SH bands above the active degree carry exactly-zero gradients on every rank: not part of the all-reduce
        gaussians.optimizer.active_rest_coeffs = (int(gaussians.active_sh_degree) + 1) ** 2 - 1
        # grads already live in the optimizer's flat buffer; NaN guard + Adam + grad zeroing are one HIP pass
        # every view's gradients went through the fused renderer's direct backward (which keeps the NaN flag) and no
        # other rank contributes: the guard needs no scan over the gradients
        direct_local = gaussians.optimizer._direct_backwards == len(cams)
        # On the fused path the next training_step's first backward ASSIGNS the whole gradient buffer, so the Adam pass
        # need not zero it (an eighth of its traffic): FusedAdam.step(zero_grad="defer").  Anything else that touches the
        # buffer first either gets the zeros (resolve_deferred) or fails loudly (optim.py).
        zg = "defer" if (all_direct and direct_local and DEFER_GRAD_ZEROING) else True
        # The choice below must be the same on every rank (it decides the sequence of collectives): it only depends on
        # the configuration (`fused_sink`), and a rank whose gradients did not all come through the direct backward
        # fails loudly instead of silently taking the other branch.
        from .optim import collectives_on
        if collectives_on() and fused_sink and OVERLAP_ALL_REDUCE_WITH_ADAM:
            if not direct_local:
                raise RuntimeError("training_step: %d of %d views went through the fused backward on this rank" %
                                   (gaussians.optimizer._direct_backwards, len(cams)))
            # all-reduce in chunks, each chunk's Adam update as soon as its sum is there (FusedAdam.step_chunked)
            gaussians.optimizer.step_chunked(chunks=4, zero_grad=zg, reduce=True)
            return total
        gaussians.optimizer.all_reduce()
        gaussians.optimizer.step(zero_grad=zg, nan_scan=not (direct_local and not collectives_on()))
        return total
    if bucket is not None:
        bucket.all_reduce()
        bad = bucket.has_nan()
    else:
        bad = torch.stack([p.grad.isnan().any() for p in gaussians.leaf_parameters() if p.grad is not None]).any()
    # train_gaussians.py:174-181: a NaN anywhere skips the update -- the reference drops the gradients
    # (zero_grad(set_to_none=True)), so optimizer.step() touches nothing: no moment decay, no step count.
    # One host decision per step, like the reference's `if torch.isnan(...)` (the torch.optim.Adam path is not the measured
    # one; round 2's clone-everything-and-restore variant tripled the optimizer traffic and still synchronised on Adam's
    # CPU-resident step counters).
    if bool(bad):
        print('NaN during backprop was found, skipping iteration...')
        if bucket is not None:
            # the gradients alias the bucket and cannot be dropped to None: a step over zeroed gradients would still decay
            # the moments and move the parameters by lr m / sqrt(v) -- skip the whole step, as the reference's does in effect
            bucket.zero()
            return total
        gaussians.optimizer.zero_grad(set_to_none=True)
    gaussians.optimizer.step()
    if bucket is not None:
        bucket.zero()
    else:
        gaussians.optimizer.zero_grad(set_to_none=True)
    return total


@functools.wraps(_training_step)
def training_step(gaussians, cams: List, background, opt, iteration: int, *args, **kwargs):
    # (this wrapper only makes sure the optimizer's view slots never outlive the step)
    try:
        return _training_step(gaussians, cams, background, opt, iteration, *args, **kwargs)
    finally:
        o = getattr(gaussians, "optimizer", None)
        if hasattr(o, "end_factored_views"):
            o.end_factored_views()


training_step.__name__ = training_step.__qualname__ = "training_step"


@torch.no_grad()
def densification_step(gaussians, render_pkg, opt, iteration: int, cameras_extent: float,
                       white_background: bool = False, generator=None, stats_done: Optional[bool] = None):
    """The densification block of the stage-1 loop (src/train_gaussians.py:158-171), to be called between
    ``loss.backward()`` and the optimizer step of an iteration: image-space radius tracking, gradient statistics,
    densify_and_prune every ``densification_interval`` and the periodic opacity reset.  Under data parallelism every
    rank must call it with identical statistics (``parallel.all_reduce_densification_stats``) and the same
    ``generator`` seed so the replicas stay bit-identical."""
    if iteration >= opt.densify_until_iter:
        return False
    # ``stats_done`` (default: what the package says): this iteration's statistics were kept by the fused backward pass
    # (pipe.densify_stats / training_step(densify_stats=True)); ``render_pkg`` may then be None
    if stats_done is None:
        stats_done = bool(getattr(render_pkg, "densify_stats_done", False))
    if not stats_done:
        vis, radii = render_pkg["visibility_filter"], render_pkg["radii"]
        gaussians.update_max_radii(radii, vis)
        gaussians.add_densification_stats(render_pkg["viewspace_points"], vis)
    changed = False
    if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
        size_threshold = 20 if iteration > opt.opacity_reset_interval else None
        gaussians.densify_and_prune(opt.densify_grad_threshold, 0.005, cameras_extent, size_threshold,
                                    generator=generator)
        changed = True
    if iteration % opt.opacity_reset_interval == 0 or (white_background and iteration == opt.densify_from_iter):
        gaussians.reset_opacity()
    return changed


def strand_view_loss(render_pkg, cam, opt, fused=None, scale: float = 1.0):
    """The strand-stage loss (src/train_strands.py:121-147 without the diffusion-prior term, whose networks are out of
    scope): L1 and SSIM on the WHOLE image, mask L1, orientation loss weighted by the ground-truth confidence."""
    image, mask = render_pkg["render"], render_pkg["mask"]
    gt_image, gt_mask = cam.original_image, cam.original_mask
    if fused is None:
        fused = image.is_cuda
    if fused and getattr(render_pkg, "renders_packed", None) is not None and opt.train_orient_conf:
        from .fused_loss import stage1_loss
        w_conf = cam.original_orient_conf if opt.use_gt_orient_conf else torch.ones_like(gt_mask[:1])
        return stage1_loss(render_pkg.renders_packed, gt_image, gt_mask, cam.original_orient_angle, w_conf,
                           opt.lambda_dl1 * scale, opt.lambda_dssim * scale, opt.lambda_dmask * scale,
                           opt.lambda_dorient * scale, mask_colours=False,
                           gt_stats=_gt_stats(cam, gt_image, gt_mask, False))
    loss = l1_loss(image, gt_image) * opt.lambda_dl1 + (1.0 - ssim(image, gt_image)) * opt.lambda_dssim + \
        l1_loss(mask, gt_mask) * opt.lambda_dmask
    if opt.lambda_dorient != 0.0:
        w = torch.ones_like(gt_mask[:1])
        if opt.use_gt_orient_conf:
            w = w * cam.original_orient_conf
        conf = render_pkg["orient_conf"] if opt.train_orient_conf else None
        Lorient = or_loss(render_pkg["orient_angle"], cam.original_orient_angle, conf, weight=w, mask=gt_mask[:1])
        if bool(torch.isnan(Lorient).any()):  # train_strands.py:141: the term is dropped from the graph
            Lorient = torch.zeros_like(loss)
        loss = loss + Lorient * opt.lambda_dorient
    return loss * scale if scale != 1.0 else loss


def strand_training_step(gaussians, gaussians_hair, cams: List, background, opt, iteration: int, pipe=PIPE):
    """One iteration of the strand stage (src/train_strands.py:98-160): rebuild the strand Gaussians from the strand
    parameters, render head + hair, loss, backward, NaN guard on the strand parameters, Adam."""
    from .gaussian_renderer import _use_fused_hair, render_hair
    from .optim import FusedAdam
    gaussians_hair.initialize_gaussians_hair()
    gaussians_hair.update_learning_rate(iteration)
    V = len(cams)
    losses = []
    # One view, FusedAdam, the fused render_hair path: the optimizer update of the SH features (48 of the 52 floats per strand
    # Gaussian) rides in the projection backward (ghr_adam_fuse, strand segment): no 192 B of gradient per Gaussian written
    # for an Adam pass to read back.  The strand directions' and the confidence's gradients arrive through autograd after
    # that kernel; they are stepped -- and their NaN mark is added to the step's flag -- before the step is finished on the
    # device (FusedAdam.finish_fused_step_with_late_groups), which undoes everything when the flag is up.
    o = gaussians_hair.optimizer if isinstance(getattr(gaussians_hair, "optimizer", None), FusedAdam) else None
    fuse = bool(FUSE_STRAND_ADAM and o is not None and V == 1 and o.direct_grads and o.can_fuse_step() and
                not getattr(pipe, "debug", False) and _use_fused_hair(gaussians, gaussians_hair, pipe, cams[0]) and
                all(g["name"] in ("xyz", "f_dc", "f_rest", "orient_conf") for g in o.param_groups))
    run_pipe = pipe
    if fuse:
        o.resolve_deferred()
        o.begin_fused_step()
        run_pipe = SimpleNamespace(**{**vars(pipe), "fuse_adam": True})
    fused_done = False
    try:
        for cam in cams:
            pkg = render_hair(cam, gaussians, gaussians_hair, run_pipe, background)
            loss = strand_view_loss(pkg, cam, opt, scale=1.0 / V)
            loss.backward(gradient=_one_like(loss))
            losses.append(loss.detach())
            if cam is not cams[-1]:
                gaussians_hair.initialize_gaussians_hair()  # a fresh graph for the next view
        if fuse and o._fuse_step["done"]:
            o.finish_fused_step_with_late_groups([g["name"] for g in o.param_groups if g["name"] not in ("f_dc", "f_rest")])
    finally:
        if fuse:
            fused_done = o.end_fused_step(grads_zero=True)
    if fused_done:
        return losses[0]
    if isinstance(gaussians_hair.optimizer, FusedAdam):
        o = gaussians_hair.optimizer
        if o._direct_backwards == V and V > 0:
            # every view's SH-feature gradients (48 of the 52 floats per Gaussian) were assigned by the fused backward, which
            # raised the optimizer's flag for any non-finite value it stored; what reached the remaining parameters through
            # autograd (strand directions, confidence: 4 of the 52 floats per Gaussian) is scanned here -- not everything
            o.scan_groups_for_nan([g["name"] for g in o.param_groups if g["name"] not in ("f_dc", "f_rest")])
            o.step(zero_grad=True, nan_scan=False)
        else:
            o.step(zero_grad=True)  # device-side NaN guard: a scan over every strand parameter's gradient
    else:
        ps = [gaussians_hair._dirs, gaussians_hair._features_dc, gaussians_hair._features_rest]
        if any(p.grad is not None and bool(p.grad.isnan().any()) for p in ps):  # train_strands.py:151-155
            gaussians_hair.optimizer.zero_grad(set_to_none=True)
            print('NaN during backprop was found, skipping iteration...')
        gaussians_hair.optimizer.step()
        gaussians_hair.optimizer.zero_grad()
    return losses[0] if len(losses) == 1 else torch.stack(losses).sum()


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


@torch.no_grad()
def make_ground_truth(gaussians_gt, cams: List, background, pipe=PIPE):
    """Synthetic supervision: render a (perturbed) model and attach the maps the loss reads (cameras.py fields)."""
    for cam in cams:
        pkg = render(cam, gaussians_gt, pipe, background)
        cam.original_image = pkg["render"].clamp(0, 1).detach()
        cam.original_mask = pkg["mask"].clamp(0, 1).detach()
        cam.original_orient_angle = pkg["orient_angle"].detach()
        cam.original_orient_conf = torch.ones_like(pkg["orient_conf"]).detach()
