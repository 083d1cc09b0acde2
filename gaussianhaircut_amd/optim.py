"""``FusedAdam``: the reference's ``torch.optim.Adam(groups, lr=0.0, eps=1e-15)`` (src/scene/gaussian_model.py:431-444)
as ONE HIP kernel over flat buffers (csrc/ghr_adam.h).

Every parameter's storage is re-pointed into one contiguous fp32 buffer (``flat_param``) and every ``.grad`` into
``flat_grad`` (so the data-parallel all-reduce is one collective on ``flat_grad`` with no packing); moments live in
``exp_avg`` / ``exp_avg_sq``.  ``param_groups`` keeps the reference's shape (``name`` / ``lr`` / ``params``), so
``update_learning_rate`` and friends work unchanged.  The NaN guard of src/train_gaussians.py:174-181 runs on-device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List

import torch
import torch.distributed as dist

from . import _lib
from .diff_gaussian_rasterization import _on_device, _ptr, _stream


def _zero_grad_mode(zero_grad):
    """``zero_grad`` of step() / step_chunked(): True, False, or the string "defer".  Returns (zero, defer)."""
    if isinstance(zero_grad, str):
        if zero_grad != "defer":
            raise ValueError("FusedAdam: zero_grad must be True, False or 'defer' (got %r)" % (zero_grad,))
        return False, True
    return bool(zero_grad), False


# Optimizer-COMPUTE sharding over the ranks (the communication pattern of ZeRO-1: reduce-scatter the gradients, update 1 / G
# of the parameters per rank, all-gather the parameters -- the wire bytes of the all-reduce, 1 / G of the Adam pass).  It is
# NOT ZeRO-1's memory saving: both moment buffers stay allocated in full on every rank (488 B per Gaussian against 288 GB of
# HBM; densification re-lays them as whole tensors), only the slices a rank does not own go stale there -- see sync_moments().
SHARD_ADAM = True
# Data-parallel steps of up to this many views in all send the SH gradients in factored form (FusedAdam.begin_factored_views):
# 12 B per Gaussian and VIEW, all-gathered, instead of 192 B per Gaussian, summed -- on the wire of a ring V (G - 1) / G x 12 B
# against 2 (G - 1) / G x 192 B per GPU, i.e. a gain below V = 32 views, kept to where it is at least 2x.
FACTORED_SH_REDUCE = os.environ.get("GHR_FACTORED_SH_REDUCE", "1") != "0"
FACTORED_SH_MAX_VIEWS = 16
SHARD_WITH_GATHERED_VIEWS = os.environ.get("GHR_SHARD_WITH_GATHERED_VIEWS", "0") == "1"


class StaleMomentsError(RuntimeError):
    """The Adam moments of this rank are current only on the slices it owns (sharded update, step_chunked(shard=True)):
    reading them all needs sync_moments(), a COLLECTIVE every rank must enter.  Raised instead of starting that
    collective from a call that is usually made by one rank alone (``if rank == 0: torch.save(gaussians.capture())``)."""


def collectives_on() -> bool:
    """More than one rank -- or one rank with GHR_FORCE_COLLECTIVES=1, which sends every collective of the N > 1 path
    through the backend anyway (tests/test_gpu_dist_shared.py: ONE rank on backend "nccl" = RCCL runs the init, the
    async work handles and the stream ordering rules that gloo does not have, on a one-GPU box)."""
    import os
    return dist.is_available() and dist.is_initialized() and (
        dist.get_world_size() > 1 or os.environ.get("GHR_FORCE_COLLECTIVES") == "1")


_TORCH_ADAM_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                            differentiable=False, fused=None, decoupled_weight_decay=False)


class FusedAdam:
    def __init__(self, param_groups: List[Dict], betas=(0.9, 0.999), eps: float = 1e-15, nan_guard: bool = True,
                 direct_grads: bool = True):
        self.param_groups = [dict(g) for g in param_groups]
        self.betas, self.eps, self.nan_guard = betas, eps, nan_guard
        # direct_grads: the fused renderer's backward adds its parameter gradients straight into ``flat_grad`` and keeps
        # the NaN flag up to date, instead of returning 8 tensors for autograd to accumulate (8 kernels, 3x the traffic)
        self.direct_grads = direct_grads
        self._direct_backwards = 0
        self.concurrent = False  # set by the trainer while a step's views run on several streams
        self._acc_event = None
        params = [p for g in self.param_groups for p in g["params"]]
        assert params and all(p.is_cuda and p.dtype == torch.float32 for p in params), \
            "FusedAdam needs fp32 parameters on a ROCm device (use torch.optim.Adam elsewhere)"
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        # {step, nan flag, steps each group has sat out}
        self.state_dev = torch.zeros(_lib.ADAM_STATE, dtype=torch.int32, device=dev)
        # groups whose parameters were replaced since the last step (densification / opacity reset): the reference's new
        # nn.Parameters have grad None, so its optimizer.step() passes them by on that iteration -- no moment decay, no
        # update, no step count (train_gaussians.py:158-181).  Bit g = group g; consumed by the next step.  Contract: the
        # surgery sits between backward() and step() like the reference's densification block; a backward AFTER the
        # surgery through the fused renderer clears the marks again (note_direct_backward), a caller that fills .grad any
        # other way before stepping calls ``cancel_skip()``.
        self._skip_next = 0
        self._deferred = None  # see step(zero_grad="defer")
        # number of higher-order SH coefficients (rows of f_rest's middle axis) that can carry a gradient, i.e.
        # (active_sh_degree + 1)^2 - 1; None = all.  Set by trainer.training_step; only shortens the all-reduce.
        self.active_rest_coeffs = None
        off, ends = 0, []
        for g in self.param_groups:
            for p in g["params"]:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                off += k
            ends.append(off)
        self._ends = (ctypes.c_int64 * len(ends))(*ends)
        self.params = params
        self._mark_zero()

    # ---- "the gradient buffer is all zero" as a checked fact.  The first direct backward of a step may then ASSIGN
    # instead of accumulate (k_project_bwd skips reading 244 B of zeros per Gaussian).  The library's own kernels are
    # tracked here; anything PyTorch does to the buffer or to a parameter's .grad view in place (autograd accumulation of
    # another loss, zero_(), an all-reduce) bumps the tensor's version counter, which withdraws the fact.
    def _mark_zero(self):
        self._zero_version = self.flat_grad._version
        self._deferred = None

    def take_known_zero(self) -> bool:
        if getattr(self, "_deferred", None) is not None:  # see step(zero_grad="defer"): this backward defines them
            self._check_untouched()
            self._deferred = None
            self._zero_version = None
            return True
        ok = (self._zero_version is not None and self.flat_grad._version == self._zero_version and
              self._direct_backwards == 0)
        self._zero_version = None
        return ok

    # ---- "the gradients are UNDEFINED until the next fused backward assigns them" (step(zero_grad="defer")).  The step of
    # trainer.training_step is always followed by another training_step whose first backward ASSIGNS every element of the
    # buffer (k_project_bwd writes all P rows of every group), so zero-filling 4 B per parameter in the Adam pass (an
    # eighth of its traffic) buys nothing there.  Everything else that could consume the buffer goes through
    # resolve_deferred() first -- it zero-fills then, which is what the eager step would have left -- and an in-place write
    # by anyone else while the buffer is undefined (autograd accumulating another loss into .grad) is detected through the
    # tensor's version counter and fails loudly instead of stepping on garbage.
    def _check_untouched(self):
        if self.flat_grad._version != self._deferred:
            raise RuntimeError("FusedAdam: the gradient buffer was written in place while its contents were undefined "
                               "(the previous step ran with zero_grad='defer', as trainer.training_step does on the fused "
                               "path); call optimizer.zero_grad() before accumulating gradients by other means")

    def resolve_deferred(self):
        """Make the gradient buffer hold zeros if the last step left it undefined (no-op otherwise)."""
        if getattr(self, "_deferred", None) is not None:
            self._check_untouched()
            self.flat_grad.zero_()
            self._mark_zero()

    # ---- gradient-bucket interface (same as parallel.FlatGradBucket)
    @property
    def flat(self):
        """The flat gradient buffer AS A READER EXPECTS IT: after ``step(zero_grad="defer")`` its contents are undefined
        until the next fused backward, so reading through this property zero-fills first (what the eager step would have
        left).  Library code that writes the buffer uses ``flat_grad`` directly."""
        self.resolve_deferred()
        return self.flat_grad

    def zero(self):
        self.flat_grad.zero_()
        self._mark_zero()
        if self._views is not None:
            self._views["next"] = 0  # (a step that is recomputed: its views fill the slots again)

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()  # grads alias the flat buffer: never dropped
        self._mark_zero()

    def all_reduce(self, average_over=None, async_op=False):
        self.resolve_deferred()
        work = None
        if collectives_on():
            if not async_op and self.active_rest_coeffs is not None:
                for a, b, how in self._reduce_plan(1):  # inactive SH bands are not sent (see _reduce_plan)
                    if how == "sum":
                        dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM)
                    elif isinstance(how, tuple):
                        view = self.flat_grad[a:b].view(how[1], how[2], 3)[:, :how[3]]
                        packed = view.contiguous()
                        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
                        view.copy_(packed)
            else:
                work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)
        if average_over and average_over != 1 and not async_op:
            self.flat_grad.div_(average_over)
        return work

    def has_nan(self):
        self.resolve_deferred()
        return torch.isnan(self.flat_grad).any()

    # ---- optimizer-state surgery for densification (reference: gaussian_model.py:581-658 on torch.optim.Adam state)
    def _rebuild(self, new_params: List[torch.Tensor], new_m: List[torch.Tensor], new_v: List[torch.Tensor]):
        """Re-lay the flat buffers for a new set of per-group tensors (one parameter per group, the reference's layout).
        Returns {group name: new nn.Parameter}.  The step counter is kept (Adam's ``step`` is per optimizer here, as it
        effectively is in the reference where every group is stepped every iteration)."""
        dev = self.flat_param.device
        n = sum(t.numel() for t in new_params)
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        flat_m = torch.empty(n, dtype=torch.float32, device=dev)
        flat_v = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off, ends, out, params = 0, [], {}, []
        for g, t, m, v in zip(self.param_groups, new_params, new_m, new_v):
            k = t.numel()
            flat_p[off:off + k].copy_(t.reshape(-1))
            flat_m[off:off + k].copy_(m.reshape(-1))
            flat_v[off:off + k].copy_(v.reshape(-1))
            p = torch.nn.Parameter(flat_p[off:off + k].view(t.shape), requires_grad=True)
            p.grad = self.flat_grad[off:off + k].view(t.shape)
            g["params"] = [p]
            out[g["name"]] = p
            params.append(p)
            off += k
            ends.append(off)
        self.flat_param, self.exp_avg, self.exp_avg_sq = flat_p, flat_m, flat_v
        self._fuse = None  # (the second buffer set of the fused update is re-made at the new size when next needed)
        self._ends = (ctypes.c_int64 * len(ends))(*ends)
        self.params = params
        self._mark_zero()
        self._direct_backwards = 0
        self._skip_next = (1 << len(self.param_groups)) - 1  # every parameter is new: the coming step is a no-op
        return out

    def relay_rows(self, take: torch.Tensor, fresh: torch.Tensor, child: torch.Tensor = None, overrides: Dict = None):
        """One re-lay of the flat buffers for a densification event (``ghr_adam_relay_rows``): new row r of every group is old
        row ``take[r]`` -- or row ``child[r]`` of ``overrides[group name]`` where ``child[r] >= 0`` -- with zeroed moments
        where ``fresh[r]``.  Every group must hold one parameter of shape [P, ...].  Returns {group name: new nn.Parameter}
        like ``_rebuild`` (whose ~80 index_select / where / copy launches this replaces)."""
        self.sync_moments()
        dev = self.flat_param.device
        P_old = int(self.param_groups[0]["params"][0].shape[0])
        P_new = int(take.numel())
        widths, shapes = [], []
        for g in self.param_groups:
            p = g["params"][0]
            assert len(g["params"]) == 1 and p.shape[0] == P_old, "relay_rows: one [P, ...] parameter per group"
            widths.append(int(p.numel() // max(P_old, 1)) if P_old else int(torch.Size(p.shape[1:]).numel()))
            shapes.append(tuple(p.shape[1:]))
        n = sum(widths) * P_new
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        flat_m = torch.empty(n, dtype=torch.float32, device=dev)
        flat_v = torch.empty(n, dtype=torch.float32, device=dev)
        take = take.to(torch.int64).contiguous()
        fresh = fresh.to(torch.uint8).contiguous()
        child = None if child is None else child.to(torch.int64).contiguous()
        keep = []
        ovr = (ctypes.c_void_p * len(widths))()
        for i, g in enumerate(self.param_groups):
            t = (overrides or {}).get(g["name"])
            if t is not None:
                t = t.detach().to(torch.float32).contiguous()
                assert t.numel() == 0 or t.numel() // t.shape[0] == widths[i]
                keep.append(t)
                ovr[i] = t.data_ptr() if t.numel() else None
        w_arr = (ctypes.c_int32 * len(widths))(*widths)
        with _on_device(dev):
            _lib.check(_lib.lib().ghr_adam_relay_rows(
                _stream(), len(widths), w_arr, P_old, P_new, _ptr(take), _ptr(fresh), None if child is None else _ptr(child),
                ctypes.cast(ovr, ctypes.c_void_p) if keep else None, _ptr(self.flat_param), _ptr(self.exp_avg),
                _ptr(self.exp_avg_sq), _ptr(flat_p), _ptr(flat_m), _ptr(flat_v)))
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off, ends, out, params = 0, [], {}, []
        for g, w, shp in zip(self.param_groups, widths, shapes):
            k = w * P_new
            p = torch.nn.Parameter(flat_p[off:off + k].view((P_new,) + shp), requires_grad=True)
            p.grad = self.flat_grad[off:off + k].view((P_new,) + shp)
            g["params"] = [p]
            out[g["name"]] = p
            params.append(p)
            off += k
            ends.append(off)
        self.flat_param, self.exp_avg, self.exp_avg_sq = flat_p, flat_m, flat_v
        self._fuse = None
        self._ends = (ctypes.c_int64 * len(ends))(*ends)
        self.params = params
        self._mark_zero()
        self._direct_backwards = 0
        self._skip_next = (1 << len(self.param_groups)) - 1  # every parameter is new: the coming step is a no-op
        return out

    def _group_views(self):
        self.sync_moments()  # (sharded optimizer: the moments of the other ranks' slices are stale here)
        off = 0
        for g in self.param_groups:
            p = g["params"][0]
            k = p.numel()
            yield g, p, self.exp_avg[off:off + k].view(p.shape), self.exp_avg_sq[off:off + k].view(p.shape)
            off += k

    def prune(self, keep_mask: torch.Tensor):
        """``_prune_optimizer`` (gaussian_model.py:596-612): keep the rows where ``keep_mask`` is True, moments included."""
        ps, ms, vs = [], [], []
        for _, p, m, v in self._group_views():
            ps.append(p.data[keep_mask]); ms.append(m[keep_mask]); vs.append(v[keep_mask])
        return self._rebuild(ps, ms, vs)

    def extend(self, tensors_dict: Dict[str, torch.Tensor]):
        """``cat_tensors_to_optimizer`` (gaussian_model.py:634-654): append rows with zero moments."""
        ps, ms, vs = [], [], []
        for g, p, m, v in self._group_views():
            ext = tensors_dict[g["name"]].detach().to(p.dtype)
            ps.append(torch.cat((p.data, ext), dim=0))
            ms.append(torch.cat((m, torch.zeros_like(ext)), dim=0))
            vs.append(torch.cat((v, torch.zeros_like(ext)), dim=0))
        return self._rebuild(ps, ms, vs)

    def replace(self, tensor: torch.Tensor, name: str):
        """``replace_tensor_to_optimizer`` (gaussian_model.py:581-594): new values, zeroed moments, for one group."""
        out = {}
        self.resolve_deferred()  # (the group's gradient is zeroed in place below)
        for i, (g, p, m, v) in enumerate(self._group_views()):
            if g["name"] == name:
                p.data.copy_(tensor.detach().reshape(p.shape))
                m.zero_(); v.zero_()
                if p.grad is not None:
                    p.grad.zero_()  # the stale gradient of this iteration belongs to the replaced values
                self._skip_next |= 1 << i
                out[name] = p
        return out

    def cancel_skip(self):
        self._skip_next = 0

    # ---- direct-gradient sink used by gaussian_renderer.fused
    def nan_flag_ptr(self):
        if self._fuse_step is not None:  # a step whose last backward carries the update: its own flag word (see below)
            return self.fused_flag_ptr()
        return ctypes.c_void_p(self.state_dev.data_ptr() + 4)

    def scan_groups_for_nan(self, names):
        """Raise the device NaN flag if a gradient of the named groups holds a NaN (``ghr_adam_nan_scan``): the part of the
        scanning guard a step needs whose other groups' gradients were stored by the fused backward (which keeps the flag)."""
        off = 0
        with _on_device(self.flat_param.device):
            for g in self.param_groups:
                k = sum(p.numel() for p in g["params"])
                if g["name"] in names and k:
                    _lib.check(_lib.lib().ghr_adam_nan_scan(_stream(), ctypes.c_void_p(self.flat_grad.data_ptr() + 4 * off), k,
                                                            _ptr(self.state_dev)))
                off += k

    # ---- SH gradients in factored form (data-parallel steps; include/ghr.h ABI 19, csrc/ghr_project.h k_sh_grad_from_views).
    # A view's gradient of the 48 SH floats of a Gaussian is basis(dir) (x) d_rgb: the fused backward of every view of the step
    # leaves its d_rgb table [P,3] (+ the camera centre) in a slot here instead of adding 192 B per Gaussian into the flat
    # gradient; step_chunked all-gathers the slots of all ranks and every rank rebuilds the f_dc / f_rest gradients from them.
    _views = None

    def _group_range(self, name):
        off = 0
        for g in self.param_groups:
            k = sum(p.numel() for p in g["params"])
            if g["name"] == name:
                return off, off + k, g["params"][0]
            off += k
        return None

    def can_factor_views(self) -> bool:
        dc, rest, xyz = self._group_range("f_dc"), self._group_range("f_rest"), self._group_range("xyz")
        return (dc is not None and rest is not None and xyz is not None and dc[1] == rest[0] and rest[2].dim() == 3 and
                rest[2].shape[1] in (3, 8, 15) and xyz[2].dim() == 2 and xyz[2].shape[1] == 3 and
                dc[2].shape[0] == xyz[2].shape[0] == rest[2].shape[0])

    def begin_factored_views(self, n_local: int, gather: bool = True, sh_degree=None):
        """``n_local`` view slots for this rank's backwards of the coming step.  ``gather`` (the same on every rank, and then
        ``n_local`` too: the all-gather is of equal parts; a rank with fewer views leaves zero tables): the slots of all ranks
        are gathered and the f_dc / f_rest ranges are NOT reduced.  ``gather=False`` (a rank's own business: the collectives do
        not change): only this rank's views are folded, before the usual sums -- what is saved is the read-modify-write of
        192 B per Gaussian in every view's backward (a rank with at least two views: 12 B per view + one 192-B store)."""
        P = int(self._group_range("xyz")[2].shape[0])
        stride = -(-(3 * P + 4) // 4) * 4  # [d_rgb P x 3 | camera centre 3 | the rank's non-finite mark | pad]: 16-B multiples
        v = getattr(self, "_views_buf", None)
        if v is None or v["buf"].shape != (n_local, stride) or v["buf"].device != self.flat_param.device:
            v = dict(buf=torch.zeros((n_local, stride), dtype=torch.float32, device=self.flat_param.device), P=P,
                     stride=stride, gathered=None)
            self._views_buf = v
        v["next"] = 0
        v["gather"] = bool(gather)
        v["deg"] = None if sh_degree is None else int(sh_degree)  # the active SH degree of the step's backwards
        # whether the SH ranges of the gradient buffer hold nothing yet (zeros, or undefined after a deferred step): the fold then
        # ASSIGNS; otherwise (somebody else's gradients are in there) it adds -- the question take_known_zero() answers for the
        # step's first backward, asked here without consuming the answer
        v["assign"] = bool(getattr(self, "_deferred", None) is not None or
                           (self._zero_version is not None and self.flat_grad._version == self._zero_version and
                            self._direct_backwards == 0))
        self._views = v

    def end_factored_views(self):
        self._views = None

    def next_view_slot(self, campos: torch.Tensor):
        """Device pointer of the next free d_rgb table of the step; the view's camera centre goes behind it."""
        v = self._views
        i = v["next"]
        if i >= v["buf"].shape[0]:
            raise RuntimeError("FusedAdam: more backward passes than view slots in this step (%d)" % v["buf"].shape[0])
        v["next"] = i + 1
        v["buf"][i, 3 * v["P"]: 3 * v["P"] + 3].copy_(campos.detach().reshape(3).to(torch.float32))
        return ctypes.c_void_p(v["buf"].data_ptr() + 4 * i * v["stride"])

    def fold_own_views(self):
        """``gather=False``: this rank's filled slots folded into the SH ranges of the gradient buffer (no-op without any)."""
        v = self._views
        if v is None or v["gather"] or v["next"] == 0:
            return
        self._rebuild_sh_from_views(v["buf"][: v["next"]])
        v["next"] = 0

    def _rebuild_sh_from_views(self, gathered: torch.Tensor, flags: bool = False):
        """flat_grad[f_dc | f_rest] (+)= sum over the gathered views (rank-major, then slot order) -- ghr_sh_grad_from_views.
        ``flags``: the rows also carry their owners' non-finite marks (float 3 P + 3 of a row): OR-ed into this rank's flag."""
        v = self._views
        P, stride = v["P"], v["stride"]
        n_views = gathered.numel() // stride
        rows = gathered.view(n_views, stride)
        (a_dc, _, _), (a_rest, _, p_rest), (a_xyz, _, _) = self._group_range("f_dc"), self._group_range("f_rest"), self._group_range("xyz")
        K = int(p_rest.shape[1]) + 1
        act = K - 1 if self.active_rest_coeffs is None else int(self.active_rest_coeffs)
        deg = v["deg"] if v.get("deg") is not None else {0: 0, 3: 1, 8: 2, 15: 3}[act]
        base_g, base_p = self.flat_grad.data_ptr(), self.flat_param.data_ptr()
        with _on_device(self.flat_param.device):
            _lib.check(_lib.lib().ghr_sh_grad_from_views(
                _stream(), P, deg, K, ctypes.c_void_p(base_p + 4 * a_xyz), n_views,
                ctypes.c_void_p(rows.data_ptr() + 4 * 3 * P), stride, _ptr(rows), stride,
                ctypes.c_void_p(base_g + 4 * a_dc), ctypes.c_void_p(base_g + 4 * a_rest), 0 if v["assign"] else 1,
                ctypes.c_void_p(self.state_dev.data_ptr() + 4) if flags else None, 3 * P + 3))
        v["assign"] = False  # (anything folded later in the same step comes on top)
        self._views_keep = gathered  # (alive until the stream has consumed them: replaced by the next step's)

    def note_direct_backward(self):
        self._direct_backwards += 1
        self._zero_version = None
        self._skip_next = 0  # fresh gradients for the (new) parameters: they take part in the coming step

    # ---- views of one step on several HIP streams (trainer.training_step): the direct backward ACCUMULATES into the
    # flat gradient buffer with plain read-modify-writes, so the accumulating kernels of different views are chained
    # by events (everything else of a view -- forward, loss, render backward -- may overlap with its neighbours)
    def accumulate_begin(self, stream):
        if self.concurrent and self._acc_event is not None:
            stream.wait_event(self._acc_event)

    def accumulate_end(self, stream):
        if self.concurrent:
            ev = torch.cuda.Event()
            ev.record(stream)
            self._acc_event = ev

    # ---- the update fused into the step's last projection backward (include/ghr.h, ghr_adam_fuse; round 6) ----------------
    # One rank, every view through the fused renderer's direct backward, no group sitting the step out: the LAST view's
    # k_project_bwd holds the step's whole gradient in registers and applies the update itself -- p, m, v read from the current
    # buffers, written to a second set whose role is swapped with the first after the step -- so that 244 B of gradient per
    # Gaussian are neither written nor read back and k_adam_v4 does not run.  The skip-on-non-finite rule stays exact: the
    # kernel that follows undoes the update (copies the old values over the new) when the step's flag is up; a view whose
    # speculative forward pass overflowed its capacity raises the same flag, so the trainer's recovery finds the parameters
    # untouched.  The gradient buffer is left as it was (contents undefined, as with zero_grad="defer").
    _fuse = None           # {"p", "m", "v": the second set; "flags": two int32 words; "parity"; "clean"}
    _fuse_step = None      # while a fused step is in progress: its ghr_adam_fuse + what keeps the pointers alive
    fused_steps = 0

    def can_fuse_step(self) -> bool:
        return (self.nan_guard and self.direct_grads and self._skip_next == 0 and self._moment_shards is None and
                not collectives_on() and len(self.param_groups) <= 16 and
                all(len(g["params"]) == 1 for g in self.param_groups))

    def begin_fused_step(self):
        """Called by the trainer before the views of a step whose last backward will carry the update.  From here until
        ``end_fused_step`` every view's backward raises THIS step's flag word (``nan_flag_ptr``)."""
        n, dev = self.flat_param.numel(), self.flat_param.device
        f = self._fuse
        if f is None or f["p"].numel() != n:
            f = self._fuse = dict(p=torch.empty(n, dtype=torch.float32, device=dev), m=torch.empty(n, dtype=torch.float32, device=dev),
                                  v=torch.empty(n, dtype=torch.float32, device=dev),
                                  flags=torch.zeros(2, dtype=torch.int32, device=dev), parity=0, clean=True)
        if not f["clean"]:
            f["flags"].zero_()  # (the previous step was not a fused one: its finish kernel did not clear this step's word)
            f["clean"] = True
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        a = _lib.AdamFuse()
        a.n = n
        a.p_in, a.m_in, a.v_in = _ptr(self.flat_param), _ptr(self.exp_avg), _ptr(self.exp_avg_sq)
        a.p_out, a.m_out, a.v_out = _ptr(f["p"]), _ptr(f["m"]), _ptr(f["v"])
        a.state = _ptr(self.state_dev)
        base = f["flags"].data_ptr()
        a.flag, a.flag_next = base + 4 * f["parity"], base + 4 * (1 - f["parity"])
        a.n_groups = len(self.param_groups)
        a.group_end_host = ctypes.cast(self._ends, ctypes.c_void_p)
        a.lr_host = ctypes.cast(lrs, ctypes.c_void_p)
        a.beta1, a.beta2, a.eps = float(self.betas[0]), float(self.betas[1]), float(self.eps)
        self._fuse_step = dict(args=a, lrs=lrs, done=False)
        return a

    def fused_flag_ptr(self):
        return ctypes.c_void_p(int(self._fuse_step["args"].flag))

    def note_fused_update(self):
        """The step's last backward has launched the update (gaussian_renderer.fused)."""
        self._fuse_step["done"] = True

    def finish_fused_step_with_late_groups(self, late_groups):
        """A step whose update was carried by a STRAND segment's backward (``ghr_adam_fuse`` with mode 1: only the SH features are
        updated by the kernel): the groups in ``late_groups`` got their gradients through autograd, after the kernel.  Their
        NaN mark joins the step's flag, their ranges of the ``out`` set are written by ``ghr_adam_step_range_to`` (in set -> out set,
        guarded by the flag, gradients zeroed), then ``k_adam_fused_finish`` decides for everything."""
        st, f = self._fuse_step, self._fuse
        a = st["args"]
        lib = _lib.lib()
        n = self.flat_param.numel()
        ranges, off = [], 0
        for g in self.param_groups:
            k = sum(p.numel() for p in g["params"])
            if g["name"] in late_groups and k:
                ranges.append((off, k))
            off += k
        par = f["parity"]
        with _on_device(self.flat_param.device):
            # (ABI 20: one out-of-place pass per range -- in set -> out set, skipped under the step's own flag word, which the
            # pass raises itself for a NaN among the range's gradients: the `in` set is intact, the finish kernel undoes -- where
            # rounds up to 6 copied p, m, v across, stepped in place and moved the flag through state[1]: six 12-36 MB copies
            # and three one-word kernels per iteration at the reference's 30 000 strands)
            for o, k in ranges:
                _lib.check(lib.ghr_adam_step_range_to(_stream(), n, o, k, _ptr(self.flat_param), _ptr(self.exp_avg),
                                                      _ptr(self.exp_avg_sq), _ptr(f["p"]), _ptr(self.flat_grad), _ptr(f["m"]),
                                                      _ptr(f["v"]), _ptr(self.state_dev), ctypes.c_void_p(int(a.flag)), 1,
                                                      len(self.param_groups), self._ends, st["lrs"], self.betas[0],
                                                      self.betas[1], self.eps, 1, 0))
            _lib.check(lib.ghr_adam_fused_finish(_stream(), ctypes.byref(a)))

    def end_fused_step(self, grads_zero: bool = False) -> bool:
        """After the views: True when the update was carried by the last backward -- the two buffer sets then swap roles and
        the parameters are re-pointed (host work only); False: nothing happened, the caller steps the usual way.
        ``grads_zero``: the gradient buffer holds zeros afterwards (the strand-stage form: the SH ranges were never written,
        the late groups' were zeroed by their update) instead of being undefined."""
        st, self._fuse_step = self._fuse_step, None
        f = self._fuse
        if st is None or not st["done"]:
            if f is not None:
                f["clean"] = False  # views may have raised this step's word
            return False
        f["p"], self.flat_param = self.flat_param, f["p"]
        f["m"], self.exp_avg = self.exp_avg, f["m"]
        f["v"], self.exp_avg_sq = self.exp_avg_sq, f["v"]
        f["parity"] = 1 - f["parity"]  # (the finish kernel cleared the other word: clean for the next fused step)
        # the parameters' views into the two buffers are made once per pair of buffers, not once per step
        views = f.setdefault("views", {})
        mine = views.get(self.flat_param.data_ptr())
        if mine is None:
            mine, off = [], 0
            for g in self.param_groups:
                p = g["params"][0]
                k = p.numel()
                mine.append(self.flat_param[off:off + k].view(p.shape))
                off += k
            views[self.flat_param.data_ptr()] = mine
        for g, v in zip(self.param_groups, mine):
            g["params"][0].data = v
        self._direct_backwards = 0
        self._acc_event = None
        self._skip_next = 0
        if grads_zero:
            self._after_step(True, False)
        else:
            self._after_step(False, True)  # the gradient buffer was neither zeroed nor written: undefined until the next backward
        self.fused_steps += 1
        return True

    # ---- optimizer interface
    def step(self, zero_grad=True, nan_scan: bool = True):
        """``zero_grad``: True (the pass zeroes the gradients), False (left as they are) or "defer" (left UNDEFINED until
        the next fused backward assigns them; see ``resolve_deferred``).  ``nan_scan=False``: every gradient of this step was produced by the fused renderer's backward on THIS rank
        (which maintains the NaN flag), so the guard needs no pass over the gradients.  Must stay True after an
        all-reduce (another rank's NaN arrives through the sum) or when other losses touched ``.grad``."""
        if self._views is not None and self._views["gather"]:
            raise RuntimeError("FusedAdam.step: view slots opened for a gathered exchange need step_chunked(reduce=True)")
        self.fold_own_views()  # (before resolve_deferred: the fold defines the SH ranges of a deferred buffer)
        self.resolve_deferred()  # (a step without a backward since a deferred one: its gradients are zeros)
        zero_grad, defer = _zero_grad_mode(zero_grad)
        self._sync_before_replicated_update()
        if self._fuse is not None:
            self._fuse["clean"] = False
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        guard = 0 if not self.nan_guard else (1 if nan_scan or self._direct_backwards == 0 else 2)
        self._direct_backwards = 0
        self._acc_event = None
        skip, self._skip_next = self._skip_next, 0
        with _on_device(self.flat_param.device):
            _lib.check(_lib.lib().ghr_adam_step(_stream(), self.flat_param.numel(), _ptr(self.flat_param),
                                                _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                                _ptr(self.state_dev), len(self.param_groups), self._ends, lrs,
                                                self.betas[0], self.betas[1], self.eps, guard, int(zero_grad), skip))
        self._after_step(zero_grad, defer)

    def _after_step(self, zero_grad, defer):
        if defer:
            self._zero_version = None
            self._deferred = self.flat_grad._version  # undefined until the next fused backward assigns them
        elif zero_grad:
            self._mark_zero()  # (the kernel zeroes the gradients whether or not the guard lets the update through)
        else:
            self._zero_version = None
            self._deferred = None

    def _chunk_ranges(self, chunks: int):
        n = self.flat_param.numel()
        per = -(-n // max(int(chunks), 1))
        per = -(-per // 1024) * 1024  # whole 4-KiB pieces
        return [(a, min(a + per, n)) for a in range(0, n, per)]

    def _reduce_plan(self, chunks: int):
        """[(a, b, how)] covering the flat buffer in order: ``how`` = "sum" (all-reduce flat_grad[a:b]), "none" (no
        rank can hold a non-zero gradient there) or ("rest", P, K1, act) (only the first ``act`` of the K1 higher-order
        SH coefficients of every Gaussian can be non-zero: they are packed, reduced and scattered back).  While
        ``active_sh_degree`` < 3 -- the first 3000 iterations of the reference's schedule (oneupSHdegree every 1000) --
        the gradients of the SH bands above the active degree are exactly zero (sh_utils.eval_sh never reads them) and
        make up to 45 of the 61 floats per Gaussian of the message."""
        n = self.flat_param.numel()
        per = -(-n // max(int(chunks), 1))
        per = -(-per // 1024) * 1024  # whole 4-KiB pieces
        plan, run_a, off = [], 0, 0
        act = self.active_rest_coeffs

        def flush(b):
            for a in range(run_a, b, per):
                plan.append((a, min(a + per, b), "sum"))

        for g in self.param_groups:
            p = g["params"][0]
            k = p.numel()
            if (getattr(self, "_views", None) or {}).get("gather") and g.get("name") in ("f_dc", "f_rest"):
                # not reduced: rebuilt on every rank from the gathered per-view factors (begin_factored_views)
                flush(off)
                plan.append((off, off + k, "views"))
                run_a = off + k
            elif g.get("name") == "f_rest" and act is not None and p.dim() == 3 and act < p.shape[1]:
                flush(off)
                plan.append((off, off + k, "none" if act <= 0 else ("rest", p.shape[0], p.shape[1], int(act))))
                run_a = off + k
            off += k
        flush(n)
        return [x for x in plan if x[1] > x[0]]

    def step_chunked(self, chunks: int = 4, zero_grad=True, reduce: bool = False, shard=None):
        """The same update as ``step(nan_scan=False)`` applied range by range (``ghr_adam_step_range``).  With
        ``reduce`` (more than one rank) every range's gradient collective is started up front and the range is updated
        as soon as its sum has arrived, so all but the last chunk of the Adam pass runs under the remaining
        communication; ranges that cannot hold a non-zero gradient on any rank (inactive SH bands, ``_reduce_plan``)
        are not sent at all.  The skip-on-NaN decision must be known before the first range is touched: the flag the
        fused backward maintains is exact for this rank's gradients -- it is raised by any NON-FINITE value, so that
        +inf on one rank and -inf on another cannot meet as a NaN only inside the sum -- and is OR-ed over the ranks
        first (4 bytes).  Only valid when every gradient of the step came through the fused renderer's direct backward
        (as with ``nan_scan=False``).

        ``shard`` (default ``SHARD_ADAM``; needs ``reduce``): ZeRO-1.  A "sum" range [a, b) is cut into G equal slices;
        rank r receives the SUM of slice r only (``reduce_scatter_tensor``, in place), runs the Adam pass over that
        slice -- 1 / G of the range -- and the updated parameters are ``all_gather_into_tensor``-ed back in place: the
        wire bytes of the all-reduce (a ring all-reduce IS a reduce-scatter followed by an all-gather), 1 / G of the
        optimizer pass per rank (134 -> 17 us at 500k Gaussians on 8 ranks).  Every rank ends the step with the same
        parameters, bit for bit, as the replicated update gives (each element is updated once, by the same kernel, from
        the same reduced gradient and the same moments).  The moments of the slices a rank does not own go STALE on
        it: ``sync_moments()`` all-gathers them, and is called by everything that reads or re-lays them (state_dict,
        optimizer surgery, a change of the plan, a replicated ``step()``).  The tail of a range that does not divide by
        G x 256 and the packed SH-band ranges take the all-reduce + replicated update as before (a few KB)."""
        self.resolve_deferred()
        zero_grad, defer = _zero_grad_mode(zero_grad)
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        guard = 2 if self.nan_guard else 0
        self._direct_backwards = 0
        self._acc_event = None
        skip, self._skip_next = self._skip_next, 0
        comm = reduce and collectives_on()
        if shard and not (zero_grad or defer):
            # the in-place reduce-scatter leaves a rank's OWN slice holding the global sum and the other slices local
            # gradients / send-buffer remains: a caller that keeps the gradients (zero_grad=False) would read a mix
            raise ValueError("FusedAdam.step_chunked: shard=True needs zero_grad=True or 'defer' (the sharded update does "
                             "not leave the summed gradient on every rank); use shard=False to keep the gradients")
        # (default: shard only when the gradients are not kept -- the same on every rank, it only depends on the arguments;
        # and not when the SH ranges travel as gathered view tables: what is left to sum is 13 floats per Gaussian in two
        # ranges -- two all-reduces and 29 us of replicated update instead of reduce-scatter + all-gather pairs and their tails)
        gather_mode = self._views is not None and self._views["gather"]
        shard = comm and (SHARD_ADAM and (zero_grad or defer) and not (gather_mode and not SHARD_WITH_GATHERED_VIEWS)
                          if shard is None else bool(shard))
        G, r = (dist.get_world_size(), dist.get_rank()) if comm else (1, 0)
        plan = self._reduce_plan(chunks) if comm else [(a, b, "local") for a, b in self._chunk_ranges(chunks)]
        if shard:
            plan = self._shard_plan(plan, G)
            key = (G, tuple((a, b) for a, b, how in plan if how == "shard"))
            if self._moment_shards is not None and self._moment_shards != key:
                self.sync_moments()  # the slices change owners (SH degree went up): bring every rank up to date first
            self._moment_shards = key
        else:
            self._sync_before_replicated_update()
        works = [None] * len(plan)
        views_work, gathered = None, None
        if self._views is not None:
            v = self._views
            if not v["gather"]:
                self.fold_own_views()  # this rank's views folded; the ranges are summed like the others
            else:
                if v["next"] < v["buf"].shape[0]:
                    v["buf"][v["next"]:].zero_()  # slots no backward of this rank filled: tables without a gradient
                if comm:
                    # the per-view d_rgb tables of every rank, rank-major: started first, the f_dc / f_rest ranges wait for it.
                    # The rank's skip-the-step flag rides in its first row (the pad float behind the camera centre): the
                    # rebuild kernel ORs the gathered marks into the flag before any range is updated -- no 4-byte all-reduce
                    v["buf"][0, 3 * v["P"] + 3: 3 * v["P"] + 4].copy_(self.state_dev[1:2])
                    gathered = torch.empty((G * v["buf"].shape[0], v["buf"].shape[1]), dtype=torch.float32,
                                           device=v["buf"].device)
                    views_work = dist.all_gather_into_tensor(gathered, v["buf"], async_op=True)
                else:
                    self._rebuild_sh_from_views(v["buf"])
        if comm:
            flag_work = None
            if views_work is None:
                flag = self.state_dev[1:2]
                flag_work = dist.all_reduce(flag, op=dist.ReduceOp.MAX, async_op=True)
            for i, (a, b, how) in enumerate(plan):
                if how == "sum":
                    works[i] = (dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True), None, None)
                elif how == "shard":
                    L = (b - a) // G
                    works[i] = (dist.reduce_scatter_tensor(self.flat_grad[a + r * L: a + (r + 1) * L], self.flat_grad[a:b],
                                                           op=dist.ReduceOp.SUM, async_op=True), None, None)
                elif isinstance(how, tuple):
                    _, P, K1, act = how
                    view = self.flat_grad[a:b].view(P, K1, 3)[:, :act]
                    packed = view.contiguous()
                    works[i] = (dist.all_reduce(packed, op=dist.ReduceOp.SUM, async_op=True), view, packed)
            if flag_work is not None:
                flag_work.wait()
        n = self.flat_param.numel()
        gathers = []
        # The ranges rebuilt from the gathered views go FIRST: the rebuild evaluates the SH basis at the positions the forward
        # passes saw, i.e. before the xyz range of this very step is updated (and, sharded, before other ranks' updated
        # slices of it arrive); their gather was also issued first, so the sums of the other ranges travel under their update.
        order = [i for i, x in enumerate(plan) if x[2] == "views"] + [i for i, x in enumerate(plan) if x[2] != "views"]
        with _on_device(self.flat_param.device):
            for pos, i in enumerate(order):
                a, b, how = plan[i]
                if works[i] is not None:
                    w, view, packed = works[i]
                    w.wait()  # orders the current stream behind this chunk's collective
                    if view is not None:
                        view.copy_(packed)
                if how == "views" and views_work is not None:
                    views_work.wait()
                    views_work = None
                    self._rebuild_sh_from_views(gathered, flags=True)  # both ranges (f_dc, f_rest) in one launch
                lo, hi = a, b
                if how == "shard":
                    L = (b - a) // G
                    lo, hi = a + r * L, a + (r + 1) * L
                _lib.check(_lib.lib().ghr_adam_step_range(
                    _stream(), n, lo, hi - lo, _ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.exp_avg),
                    _ptr(self.exp_avg_sq), _ptr(self.state_dev), len(self.param_groups), self._ends, lrs,
                    self.betas[0], self.betas[1], self.eps, guard, int(zero_grad), int(pos == len(order) - 1), skip))
                if how == "shard":
                    # (issued behind the kernel above on the current stream; in place: the send slice lies in the receive
                    # buffer at rank * count, the layout the backends' in-place all-gather expects)
                    gathers.append(dist.all_gather_into_tensor(self.flat_param[a:b], self.flat_param[lo:hi], async_op=True))
                    if zero_grad:  # the slices of the other ranks still hold this rank's local gradients
                        self.flat_grad[a:lo].zero_()
                        self.flat_grad[hi:b].zero_()
            for w in gathers:
                w.wait()  # the next forward reads the gathered parameters
        self._after_step(zero_grad, defer)  # (every range of the plan has been through the kernel)

    # ---- ZeRO-1 helpers (step_chunked(shard=True))
    _moment_shards = None  # (G, ((a, b), ...)): the "shard" ranges whose moments are current only on their owner's slice

    @staticmethod
    def _shard_plan(plan, G):
        """Splits every "sum" range of a reduce plan into a part whose length divides by G x 256 (how = "shard": slice r of
        it belongs to rank r; 1-KiB multiples keep every slice on the 16-B aligned float4 path of k_adam_v4) and a tail
        that keeps the all-reduce.  G = 1 (one rank forced through the collectives) shards every range onto itself."""
        out = []
        for a, b, how in plan:
            if how != "sum":
                out.append((a, b, how))
                continue
            main = (b - a) // (G * 256) * (G * 256)
            if main:
                out.append((a, a + main, "shard"))
            if a + main < b:
                out.append((a + main, b, "sum"))
        return out

    def moments_stale(self) -> bool:
        """True when this rank's moments are current only on the slices it owns (after a sharded step on more than one
        rank): state_dict() / capture() then need sync_moments() -- on every rank -- first."""
        # (one rank under GHR_FORCE_COLLECTIVES=1 runs the collectives but owns every slice: nothing is stale there)
        return (self._moment_shards is not None and dist.is_available() and dist.is_initialized() and
                dist.get_world_size() > 1)

    def sync_moments(self):
        """All-gather the Adam moments of the sharded ranges so that every rank holds all of them again (no-op when the
        optimizer is not sharded or already in sync).  COLLECTIVE: every rank must call it at the same point.  That holds
        for the callers inside this class that do so implicitly -- optimizer surgery (prune / extend / replace:
        densification runs on every rank alike or the replicas diverge anyway), a change of the shard plan, a replicated
        ``step()`` -- but NOT for checkpointing, which is usually rank 0's business alone: ``state_dict()`` therefore
        never starts it on its own (StaleMomentsError; pass ``collective=True`` from every rank, or call this first)."""
        key, self._moment_shards = self._moment_shards, None
        if key is None or not collectives_on():
            return
        G, ranges = key
        r = dist.get_rank()
        for a, b in ranges:
            L = (b - a) // G
            for buf in (self.exp_avg, self.exp_avg_sq):
                dist.all_gather_into_tensor(buf[a:b], buf[a + r * L: a + (r + 1) * L])

    def _sync_before_replicated_update(self):
        if self._moment_shards is not None:
            self.sync_moments()

    def _param_ranges(self):
        off = 0
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                yield gi, off, off + p.numel(), p
                off += p.numel()

    def state_dict(self, collective: bool = False):
        """``torch.optim.Adam.state_dict()`` layout (what the reference's ``GaussianModel.capture`` stores,
        src/scene/gaussian_model.py:84-99): ``state[i] = {step, exp_avg, exp_avg_sq}`` per parameter index and
        ``param_groups`` with ``params`` as indices -- a checkpoint written here restores into torch.optim.Adam and the
        other way round.  ``step`` of a parameter is the optimizer's step count less the steps its group sat out.

        After a sharded step on more than one rank the moments are stale outside this rank's slices.  Bringing them in is a
        collective, so it only happens on request: ``collective=True`` -- EVERY rank must then make this call -- or an
        explicit ``sync_moments()`` on every rank beforehand.  Otherwise StaleMomentsError: the common
        ``if rank == 0: torch.save(gaussians.capture())`` fails loudly instead of hanging in an all-gather nobody else joins."""
        if self.moments_stale():
            if not collective:
                raise StaleMomentsError(
                    "FusedAdam.state_dict(): the Adam moments are sharded over %d ranks (step_chunked(shard=True)) and stale "
                    "on this one outside its own slices.  Call optimizer.sync_moments() on EVERY rank first (it is a "
                    "collective), or state_dict(collective=True) / GaussianModel.capture(collective=True) from every rank; "
                    "FusedAdam.SHARD off (optim.SHARD_ADAM = False) keeps every rank's moments complete."
                    % dist.get_world_size())
            self.sync_moments()
        st = self.state_dev.cpu()
        state = {}
        for i, (gi, a, b, p) in enumerate(self._param_ranges()):
            steps = int(st[0]) - int(st[2 + gi])
            if steps > 0 or bool(self.exp_avg_sq[a:b].any()):
                state[i] = {"step": torch.tensor(float(steps)), "exp_avg": self.exp_avg[a:b].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[a:b].view(p.shape).clone()}
        groups, i = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d.setdefault("betas", self.betas)
            d.setdefault("eps", self.eps)
            for k, v in _TORCH_ADAM_DEFAULTS.items():  # so torch.optim.Adam.load_state_dict finds every key it reads
                d.setdefault(k, v)
            d["params"] = list(range(i, i + len(g["params"])))
            i += len(g["params"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts ``torch.optim.Adam.state_dict()`` (also the one ``state_dict`` above writes) and round 1's flat
        layout.  Parameters are not part of an optimizer checkpoint (the reference restores them from the model tuple)."""
        if "flat_param" in sd:  # round-1 layout
            self._moment_shards = None
            self.flat_param.copy_(sd["flat_param"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            st = sd["state"]
            self.state_dev.zero_()
            self.state_dev[: st.numel()].copy_(st)  # checkpoints of ABI <= 10 carry {step, flag} only
            for g, lr in zip(self.param_groups, sd["lrs"]):
                g["lr"] = lr
            return
        assert len(sd["param_groups"]) == len(self.param_groups), "optimizer checkpoint has a different group layout"
        ranges = list(self._param_ranges())
        steps = [0] * len(self.param_groups)
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, (gi, a, b, p) in enumerate(ranges):
            e = sd["state"].get(i)
            if e is None:
                continue
            assert e["exp_avg"].numel() == b - a, "optimizer checkpoint does not match parameter %d" % i
            self.exp_avg[a:b].copy_(e["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(e["exp_avg_sq"].reshape(-1))
            steps[gi] = max(steps[gi], int(float(e["step"])))
        st = torch.zeros(_lib.ADAM_STATE, dtype=torch.int32)
        st[0] = max(steps)
        for gi, k in enumerate(steps):
            st[2 + gi] = max(steps) - k
        self.state_dev.copy_(st)
        for g, d in zip(self.param_groups, sd["param_groups"]):
            for k, v in d.items():
                if k != "params":
                    g[k] = v
        if "betas" in self.param_groups[0]:
            self.betas = tuple(self.param_groups[0]["betas"])
        if "eps" in self.param_groups[0]:
            self.eps = self.param_groups[0]["eps"]
        self._skip_next = 0
        self._moment_shards = None  # a checkpoint holds every moment: nothing is stale on any rank
