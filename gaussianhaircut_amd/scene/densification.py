"""Densify / clone / split / prune with optimizer-state surgery.

Mirror of the reference's ``src/scene/gaussian_model.py:560-741`` (``reset_opacity``, ``replace_tensor_to_optimizer``,
``_prune_optimizer``, ``prune_points``, ``cat_tensors_to_optimizer``, ``densification_postfix``, ``densify_and_split``,
``densify_and_clone``, ``densify_and_prune``, ``add_densification_stats``), device-agnostic, for both optimizers:
``torch.optim.Adam`` (the reference's per-parameter state surgery) and ``optim.FusedAdam`` (the flat buffers are re-laid
once per operation).  ``densify_and_split`` draws its samples from an optional ``generator`` so that data-parallel ranks
 split identically; without one it uses the global RNG exactly like the reference.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from ..utils.general_utils import build_rotation, inverse_sigmoid


RELAY_KERNEL = os.environ.get("GHR_DENSIFY_RELAY_KERNEL", "1") != "0"  # the event's re-lay as one HIP pass

class DensificationMixin:
    # ------------------------------------------------------------------ optimizer surgery
    def _is_fused(self):
        from ..optim import FusedAdam
        return isinstance(self.optimizer, FusedAdam)

    def _assign(self, t):
        self._xyz = t["xyz"]
        self._features_dc = t["f_dc"]
        self._features_rest = t["f_rest"]
        self._opacity = t["opacity"]
        self._label = t["label"]
        self._scaling = t["scaling"]
        self._rotation = t["rotation"]

    def replace_tensor_to_optimizer(self, tensor, name):
        if self._is_fused():
            return self.optimizer.replace(tensor, name)
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                stored = self.optimizer.state.get(group["params"][0], None)
                if stored is not None:
                    stored["exp_avg"] = torch.zeros_like(tensor)
                    stored["exp_avg_sq"] = torch.zeros_like(tensor)
                    del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(tensor.requires_grad_(True))
                if stored is not None:
                    self.optimizer.state[group["params"][0]] = stored
                out[group["name"]] = group["params"][0]
        return out

    def reset_opacity(self):
        """gaussian_model.py:515-518."""
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    def _prune_optimizer(self, mask):
        if self._is_fused():
            return self.optimizer.prune(mask)
        out = {}
        for group in self.optimizer.param_groups:
            stored = self.optimizer.state.get(group["params"][0], None)
            if stored is not None:
                stored["exp_avg"] = stored["exp_avg"][mask]
                stored["exp_avg_sq"] = stored["exp_avg_sq"][mask]
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
                self.optimizer.state[group["params"][0]] = stored
            else:
                group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
            out[group["name"]] = group["params"][0]
        return out

    def prune_points(self, mask):
        """gaussian_model.py:614-632: drop the rows where ``mask`` is True."""
        valid = ~mask
        t = self._prune_optimizer(valid)
        self._assign(t)
        self._orient_conf = t["orient_conf"] if "orient_conf" in t else torch.zeros_like(self._xyz[:, :1])
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid]
        self.denom = self.denom[valid]
        if len(self.max_radii2D):
            self.max_radii2D = self.max_radii2D[valid]

    def cat_tensors_to_optimizer(self, tensors_dict):
        if self._is_fused():
            return self.optimizer.extend(tensors_dict)
        out = {}
        for group in self.optimizer.param_groups:
            assert len(group["params"]) == 1
            ext = tensors_dict[group["name"]]
            stored = self.optimizer.state.get(group["params"][0], None)
            if stored is not None:
                stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(ext)), dim=0)
                stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
                self.optimizer.state[group["params"][0]] = stored
            else:
                group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            out[group["name"]] = group["params"][0]
        return out

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_orient_confs,
                              new_labels, new_scaling, new_rotation):
        """gaussian_model.py:656-678."""
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "orient_conf": new_orient_confs, "label": new_labels, "scaling": new_scaling, "rotation": new_rotation}
        t = self.cat_tensors_to_optimizer(d)
        self._assign(t)
        self._orient_conf = t["orient_conf"] if "orient_conf" in t else torch.zeros_like(self._label)
        P, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)

    # ------------------------------------------------------------------ densification
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        """gaussian_model.py:680-708."""
        n_init = self.get_xyz.shape[0]
        dev = self.get_xyz.device
        padded = torch.zeros((n_init,), device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = torch.where(padded >= grad_threshold, True, False)
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        means = torch.zeros((stds.size(0), 3), device=dev)
        samples = torch.normal(mean=means, std=stds, generator=generator)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        new_rotation = self._rotation[sel].repeat(N, 1)
        new_features_dc = self._features_dc[sel].repeat(N, 1, 1)
        new_features_rest = self._features_rest[sel].repeat(N, 1, 1)
        new_opacity = self._opacity[sel].repeat(N, 1)
        new_orient_conf = self._orient_conf[sel].repeat(N, 1)
        new_label = self._label[sel].repeat(N, 1)
        self.densification_postfix(new_xyz, new_features_dc, new_features_rest, new_opacity, new_orient_conf,
                                   new_label, new_scaling, new_rotation)
        prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool)))
        self.prune_points(prune_filter)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """gaussian_model.py:710-725."""
        sel = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel], self._opacity[sel],
                                   self._orient_conf[sel], self._label[sel], self._scaling[sel], self._rotation[sel])

    @torch.no_grad()
    def _densify_and_prune_onepass(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """The same event with ONE re-lay of the optimizer's flat buffers (FusedAdam): the reference's sequence -- clone,
        postfix, split, postfix, prune the split sources, prune by opacity / size (gaussian_model.py:680-741) -- re-lays every
        buffer four times and gathers by boolean mask some sixty times (each a device-to-host round trip for the row count).
        Here the three decisions are taken on per-row scalars, the surviving rows are described by (source row, kind) and every
        group is gathered once by index: three host round trips in all.  Same rows, same order, same values, same moments, same
        random numbers (the split samples are drawn with the same shape from the same generator) as the stepwise path --
        tests/test_gpu_fused.py compares the two bit for bit."""
        dev = self.get_xyz.device
        P0 = self.get_xyz.shape[0]
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        scal = self.get_scaling
        smax = torch.max(scal, dim=1).values
        limit = self.percent_dense * extent
        sel_c = torch.logical_and(torch.norm(grads, dim=-1) >= max_grad, smax <= limit)          # densify_and_clone
        # densify_and_split looks at the model AFTER the clones were appended, with the gradients padded by zeros: a clone
        # (small by selection) can never be split, so the selection lives on the original rows
        sel_s = torch.logical_and(grads.squeeze(-1) >= max_grad, smax > limit)
        idx_c = sel_c.nonzero(as_tuple=True)[0]
        idx_s = sel_s.nonzero(as_tuple=True)[0]
        n_c, n_s, N = int(idx_c.numel()), int(idx_s.numel()), 2
        if max_grad <= 0 and n_c:  # (zero-padded gradients would pass a non-positive threshold: leave that to the stepwise path)
            return None
        # the split children (gaussian_model.py:688-697)
        stds = scal[idx_s].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds, generator=generator)
        rots = build_rotation(self._rotation[idx_s]).repeat(N, 1, 1)
        child_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[idx_s].repeat(N, 1)
        child_scaling = self.scaling_inverse_activation(scal[idx_s].repeat(N, 1) / (0.8 * N))
        # rows after clone + split, before any pruning: the originals, the clones, the children
        src = torch.cat([torch.arange(P0, device=dev), idx_c, idx_s.repeat(N)])
        P2 = P0 + n_c + N * n_s
        keep = torch.ones(P2, dtype=torch.bool, device=dev)
        keep[idx_s] = False                                                                       # the split sources go
        # the final prune looks at the model after all that: opacity and largest scale of every row (the children's from
        # their new raw scaling, through the activation, exactly as get_scaling would)
        prune = self.opacity_activation(self._opacity[src]).squeeze(-1) < min_opacity
        if max_screen_size:
            row_smax = smax[src]
            if n_s:
                row_smax[P0 + n_c:] = torch.max(self.scaling_activation(child_scaling), dim=1).values
            # (max_radii2D was reset by the densification_postfix calls of the stepwise sequence: `big_points_vs` is empty)
            prune = torch.logical_or(prune, row_smax > 0.1 * extent)
        keep &= ~prune
        rows = keep.nonzero(as_tuple=True)[0]
        take = src[rows]
        fresh = rows >= P0                       # clones and children start with zero moments
        child = rows - (P0 + n_c)                # >= 0 for the children: their place in child_xyz / child_scaling
        is_child = child >= 0
        o = self.optimizer
        if RELAY_KERNEL and hasattr(o, "relay_rows"):
            # one HIP pass writes the new flat buffers (optim.FusedAdam.relay_rows / ghr_adam_relay_rows)
            t = o.relay_rows(take, fresh, child if n_s else None,
                             {"xyz": child_xyz, "scaling": child_scaling} if n_s else None)
            self._assign(t)
            self._orient_conf = t["orient_conf"] if "orient_conf" in t else torch.zeros_like(self._label)
            P = self.get_xyz.shape[0]
            self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
            self.denom = torch.zeros((P, 1), device=dev)
            self.max_radii2D = torch.zeros((P,), device=dev)
            return P
        o.sync_moments()
        ps, ms, vs = [], [], []
        off = 0
        for g in o.param_groups:
            p = g["params"][0]
            k = p.numel()
            m = o.exp_avg[off:off + k].view(p.shape)
            v = o.exp_avg_sq[off:off + k].view(p.shape)
            off += k
            pn = p.data.index_select(0, take)
            bshape = (-1,) + (1,) * (p.dim() - 1)
            if g["name"] in ("xyz", "scaling") and n_s:  # (selects, not masked assignments: those read the row count back)
                new = (child_xyz if g["name"] == "xyz" else child_scaling).index_select(0, child.clamp_min(0))
                pn = torch.where(is_child.view(bshape), new, pn)
            zero = torch.zeros((), dtype=pn.dtype, device=dev)
            mn = torch.where(fresh.view(bshape), zero, m.index_select(0, take))
            vn = torch.where(fresh.view(bshape), zero, v.index_select(0, take))
            ps.append(pn); ms.append(mn); vs.append(vn)
        t = o._rebuild(ps, ms, vs)
        self._assign(t)
        self._orient_conf = t["orient_conf"] if "orient_conf" in t else torch.zeros_like(self._label)
        P = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)
        return P

    ONEPASS_DENSIFY = True

    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """gaussian_model.py:727-741 (the reference calls it under ``torch.no_grad()``, train_gaussians.py:146)."""
        if self.ONEPASS_DENSIFY and self._is_fused() and all(len(g["params"]) == 1 for g in self.optimizer.param_groups):
            if self._densify_and_prune_onepass(max_grad, min_opacity, extent, max_screen_size, generator) is not None:
                return
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        prune_mask = self.get_opacity.squeeze() < min_opacity
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_vs), big_ws)
        self.prune_points(prune_mask)

    @torch.no_grad()
    def update_max_radii(self, radii, visibility_filter):
        """train_gaussians.py:162-163."""
        self.max_radii2D[visibility_filter] = torch.max(self.max_radii2D[visibility_filter],
                                                        radii[visibility_filter].to(self.max_radii2D.dtype))
