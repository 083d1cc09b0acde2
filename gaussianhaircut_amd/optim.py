"""``FusedAdam``: the reference's ``torch.optim.Adam(groups, lr=0.0, eps=1e-15)`` (src/scene/gaussian_model.py:431-444)
as ONE HIP kernel over flat buffers (csrc/ghr_adam.h).

Every parameter's storage is re-pointed into one contiguous fp32 buffer (``flat_param``) and every ``.grad`` into
``flat_grad`` (so the data-parallel all-reduce is one collective on ``flat_grad`` with no packing); moments live in
``exp_avg`` / ``exp_avg_sq``.  ``param_groups`` keeps the reference's shape (``name`` / ``lr`` / ``params``), so
``update_learning_rate`` and friends work unchanged.  The NaN guard of src/train_gaussians.py:174-181 runs on-device.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List

import torch
import torch.distributed as dist

from . import _lib
from .diff_gaussian_rasterization import _ptr, _stream


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
        self.state_dev = torch.zeros(2, dtype=torch.int32, device=dev)  # {step, nan flag}
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

    # ---- gradient-bucket interface (same as parallel.FlatGradBucket)
    @property
    def flat(self):
        return self.flat_grad

    def zero(self):
        self.flat_grad.zero_()

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()  # grads alias the flat buffer: never dropped

    def all_reduce(self, average_over=None, async_op=False):
        work = None
        if dist.is_initialized() and dist.get_world_size() > 1:
            work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)
        if average_over and average_over != 1 and not async_op:
            self.flat_grad.div_(average_over)
        return work

    def has_nan(self):
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
        self._ends = (ctypes.c_int64 * len(ends))(*ends)
        self.params = params
        self._direct_backwards = 0
        return out

    def _group_views(self):
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
        for g, p, m, v in self._group_views():
            if g["name"] == name:
                p.data.copy_(tensor.detach().reshape(p.shape))
                m.zero_(); v.zero_()
                out[name] = p
        return out

    # ---- direct-gradient sink used by gaussian_renderer.fused
    def nan_flag_ptr(self):
        return ctypes.c_void_p(self.state_dev.data_ptr() + 4)

    def note_direct_backward(self):
        self._direct_backwards += 1

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

    # ---- optimizer interface
    def step(self, zero_grad: bool = True, nan_scan: bool = True):
        """``nan_scan=False``: every gradient of this step was produced by the fused renderer's backward on THIS rank
        (which maintains the NaN flag), so the guard needs no pass over the gradients.  Must stay True after an
        all-reduce (another rank's NaN arrives through the sum) or when other losses touched ``.grad``."""
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        guard = 0 if not self.nan_guard else (1 if nan_scan or self._direct_backwards == 0 else 2)
        self._direct_backwards = 0
        self._acc_event = None
        with torch.cuda.device(self.flat_param.device):
            _lib.check(_lib.lib().ghr_adam_step(_stream(), self.flat_param.numel(), _ptr(self.flat_param),
                                                _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                                _ptr(self.state_dev), len(self.param_groups), self._ends, lrs,
                                                self.betas[0], self.betas[1], self.eps, guard, int(zero_grad)))

    def _chunk_ranges(self, chunks: int):
        n = self.flat_param.numel()
        per = -(-n // max(int(chunks), 1))
        per = -(-per // 1024) * 1024  # whole 4-KiB pieces
        return [(a, min(a + per, n)) for a in range(0, n, per)]

    def step_chunked(self, chunks: int = 4, zero_grad: bool = True, reduce: bool = False):
        """The same update as ``step(nan_scan=False)`` applied range by range (``ghr_adam_step_range``).  With
        ``reduce`` (more than one rank) every range's gradient all-reduce is started up front and the range is updated
        as soon as its sum has arrived, so all but the last chunk of the Adam pass runs under the remaining
        communication.  The skip-on-NaN decision must be known before the first range is touched: the flag the fused
        backward maintains is exact for this rank's gradients and is OR-ed over the ranks first (4 bytes).  Only valid
        when every gradient of the step came through the fused renderer's direct backward (as with ``nan_scan=False``);
        a NaN that only appears in the cross-rank sum (+inf on one rank, -inf on another) is not caught."""
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        guard = 2 if self.nan_guard else 0
        self._direct_backwards = 0
        self._acc_event = None
        ranges = self._chunk_ranges(chunks)
        works = None
        if reduce and dist.is_initialized() and dist.get_world_size() > 1:
            flag = self.state_dev[1:2]
            flag_work = dist.all_reduce(flag, op=dist.ReduceOp.MAX, async_op=True)
            works = [dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in ranges]
            flag_work.wait()
        n = self.flat_param.numel()
        with torch.cuda.device(self.flat_param.device):
            for i, (a, b) in enumerate(ranges):
                if works is not None:
                    works[i].wait()  # orders the current stream behind this chunk's collective
                _lib.check(_lib.lib().ghr_adam_step_range(
                    _stream(), n, a, b - a, _ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.exp_avg),
                    _ptr(self.exp_avg_sq), _ptr(self.state_dev), len(self.param_groups), self._ends, lrs,
                    self.betas[0], self.betas[1], self.eps, guard, int(zero_grad), int(i == len(ranges) - 1)))

    def state_dict(self):
        return {"flat_param": self.flat_param, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "state": self.state_dev, "lrs": [g["lr"] for g in self.param_groups],
                "names": [g.get("name") for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.flat_param.copy_(sd["flat_param"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.state_dev.copy_(sd["state"])
        for g, lr in zip(self.param_groups, sd["lrs"]):
            g["lr"] = lr
