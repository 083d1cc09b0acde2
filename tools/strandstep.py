"""The strand-stage iteration of bench.py's `strand_stage` block, alone, for rocprofv3 --kernel-trace --stats (round 6)."""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd.gaussian_renderer import render_hair  # noqa: E402
from gaussianhaircut_amd.scene.cameras import ring_cameras  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands  # noqa: E402
from gaussianhaircut_amd.trainer import strand_training_step  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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
    losses = []
    for i in range(4):
        losses.append(strand_training_step(head, hair, [cam], bg, opt, i + 1, pipe=pipe))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        losses.append(strand_training_step(head, hair, [cam], bg, opt, 5 + i, pipe=pipe))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("STRAND loss %.5f -> %.5f over %d iterations (%d strand Gaussians), optimizer step %d, skipped-step flag %d" % (
        float(losses[0]), float(losses[-1]), len(losses), S * n_seg, int(hair.optimizer.state_dev[0]),
        int(hair.optimizer.state_dev[1])))
    print("STRAND %.3f ms per iteration (host issue %.3f)" % (1e3 * (time.perf_counter() - t0) / K, 1e3 * (t1 - t0) / K))


if __name__ == "__main__":
    main()
