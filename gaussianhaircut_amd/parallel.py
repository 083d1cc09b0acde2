"""View-sharded data parallelism (new design; the reference is single-GPU, batch = 1 view).

One process per GPU (``torchrun``), a full replica of the Gaussian parameters and Adam state on every rank, the
views of a global step dealt round-robin (view k -> rank k mod G), local gradient accumulation, then ONE
``all_reduce(SUM)`` over a single flat fp32 bucket that aliases every leaf ``.grad`` -- Gaussian gradients only, never
images or workspaces -- followed by the identical Adam step on every rank (replicas stay bit-identical).
Backend ``nccl`` is RCCL on ROCm (xGMI inside a node); ``gloo`` is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_views(views: List, rank: int, world: int) -> List:
    """View k -> rank k mod G."""
    return views[rank::world]


class FlatGradBucket:
    """One contiguous fp32 buffer whose slices ARE the parameters' ``.grad`` tensors.

    autograd accumulates in place into an existing ``.grad``, so after ``backward()`` the bucket already holds the
    flattened gradient -- no pack/unpack copies around the collective.  ``zero()`` replaces
    ``optimizer.zero_grad(set_to_none=True)`` (which would drop the aliasing)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)
            off += k

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, average_over: Optional[int] = None, async_op: bool = False):
        """SUM across ranks (then divide by ``average_over`` if given)."""
        work = None
        if dist.is_initialized() and dist.get_world_size() > 1:
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if average_over and average_over != 1 and not async_op:
            self.flat.div_(average_over)
        return work

    def has_nan(self) -> torch.Tensor:
        return torch.isnan(self.flat).any()


def all_reduce_densification_stats(xyz_gradient_accum, denom, max_radii2D):
    """SUM / SUM / MAX of the densification statistics (gaussian_model.py:739-741, train_gaussians.py:163)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM)
        dist.all_reduce(denom, op=dist.ReduceOp.SUM)
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)


def param_checksum(params: Iterable[torch.Tensor]) -> float:
    """Cheap replica-consistency check: every rank must report the same value."""
    s = torch.zeros((), dtype=torch.float64, device=next(iter(params)).device)
    for p in params:
        s = s + p.detach().double().sum()
    return float(s.item())
