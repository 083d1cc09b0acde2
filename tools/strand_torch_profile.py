"""torch.profiler view of the strand-stage iteration of tools/strandstep.py: which PyTorch operators (and which source lines)
still run beside the library's kernels (round 6)."""
import os
import sys
from types import SimpleNamespace

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd.gaussian_renderer import render_hair  # noqa: E402
from gaussianhaircut_amd.scene.cameras import ring_cameras  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands  # noqa: E402
from gaussianhaircut_amd.trainer import strand_training_step  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["cfg3"]
    bg = syn.background(dev)
    S, n_seg, n_head = 30_000, 99, 100_000
    head = syn.make_model(spec, dev)
    with torch.no_grad():
        head._label[:n_head] = -4.0
        head._label[n_head:] = 4.0
    head.precompute_head()
    g = torch.Generator().manual_seed(9)
    unit = torch.nn.functional.normalize
    origins = unit(torch.randn(S, 1, 3, generator=g), dim=-1)
    dirs = torch.randn(S, n_seg, 3, generator=g) * 0.003 + unit(torch.randn(S, 1, 3, generator=g), dim=-1) * 0.01
    feats = torch.randn(S * n_seg, 16, 3, generator=g) * 0.1
    hair = GaussianModelStrands(3).create_from_strands(origins.to(dev), dirs.to(dev), feats.to(dev))
    cam = ring_cameras(1, spec.W, spec.H, device=dev)[0]
    opt = OptimizationParams()
    opt.lambda_dorient, opt.lambda_dmask = 0.1, 0.1
    pipe = SimpleNamespace(debug=False, fused_projection=True)
    with torch.no_grad():
        hair.initialize_gaussians_hair()
        p = render_hair(cam, head, hair, pipe, bg)
        cam.original_image, cam.original_mask = p["render"].clamp(0, 1).detach(), p["mask"].clamp(0, 1).detach()
        cam.original_orient_angle = p["orient_angle"].detach()
        cam.original_orient_conf = torch.ones_like(p["orient_conf"]).detach()
        hair._dirs.mul_(1.02)
    hair.training_setup(opt)
    for i in range(6):
        strand_training_step(head, hair, [cam], bg, opt, i + 1, pipe=pipe)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for i in range(4):
            strand_training_step(head, hair, [cam], bg, opt, i + 7, pipe=pipe)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
    seen = {}
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::exp", "aten::zero_", "aten::fill_", "aten::mul",
                      "aten::clone", "aten::contiguous", "aten::zeros", "aten::ones_like", "aten::sum", "aten::div"):
            stack = [f for f in (e.stack or []) if "gaussianhaircut_amd" in f or "tools/" in f][:4]
            key = (e.name, str(e.input_shapes)[:80], tuple(stack))
            seen[key] = seen.get(key, 0) + 1
    for (name, shapes, stack), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print("OP %-14s x%-3d %s\n      %s" % (name, n, shapes, "\n      ".join(stack) if stack else "(autograd engine / no python frame)"))


if __name__ == "__main__":
    main()
