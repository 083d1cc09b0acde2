"""A short stage-1 run end to end on one GPU (round 6): the loop of src/train_gaussians.py:96-181 -- a camera per iteration,
render, the four losses, backward, the per-iteration densification statistics (inside k_project_bwd), densify_and_prune at its
interval (one HIP re-lay), an opacity reset, the SH degree going up, Adam (inside the last backward) -- on the cfg3 strand model
against a perturbed copy of itself, with the schedule compressed (densify every 100 iterations from 100, opacity reset at 250,
SH degree up every 100) so that 400 iterations see every kind of event.

    python tools/mini_train.py [iterations] > profiles/r06_mini_train.txt
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd.scene.cameras import ring_cameras  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams  # noqa: E402
from gaussianhaircut_amd.trainer import densification_step, make_ground_truth, training_step  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS[os.environ.get("MINI_CFG", "cfg3")]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    opt.densify_from_iter, opt.densification_interval, opt.opacity_reset_interval = 100, 100, 250
    opt.densify_until_iter = iters
    bg = syn.background(dev)
    model = syn.make_model(spec, dev)
    model.active_sh_degree = 0
    cams = ring_cameras(16, spec.W, spec.H, device=dev)
    with torch.no_grad():
        gt = syn.make_model(spec, dev)
        g = torch.Generator(device="cpu").manual_seed(202)
        gt._xyz.add_((0.002 * torch.randn(gt._xyz.shape, generator=g)).to(dev))
        gt._features_dc.add_((0.1 * torch.randn(gt._features_dc.shape, generator=g)).to(dev))
        make_ground_truth(gt, cams, bg)
        del gt
    model.training_setup(opt)
    gen = torch.Generator(device=dev).manual_seed(7)
    order = torch.randperm(16, generator=torch.Generator().manual_seed(1)).tolist()
    print("MINI %s: %d Gaussians, %d iterations, 16 cameras" % (spec.name, model.get_xyz.shape[0], iters))
    window, t_win, events = [], time.perf_counter(), 0.0
    first = None
    for it in range(1, iters + 1):
        cam = cams[order[it % 16]]
        if it % 100 == 0:
            model.oneupSHdegree()
        loss = training_step(model, [cam], bg, opt, it, densify_stats=True)
        window.append(loss)
        P0 = model.get_xyz.shape[0]
        t0 = time.perf_counter()
        changed = densification_step(model, None, opt, it, 2.5, generator=gen, stats_done=True)
        if changed or it % opt.opacity_reset_interval == 0:
            torch.cuda.synchronize()
            events += time.perf_counter() - t0
            print("MINI iteration %4d: %s %d -> %d Gaussians (%.2f ms)" % (
                it, "densify_and_prune" if changed else "opacity reset", P0, model.get_xyz.shape[0],
                1e3 * (time.perf_counter() - t0)))
        if it % 50 == 0:
            torch.cuda.synchronize()
            mean = float(torch.stack(window).mean())
            first = mean if first is None else first
            dt = time.perf_counter() - t_win
            print("MINI iteration %4d: mean loss of the last 50 %.5f, %d Gaussians, SH degree %d, %.3f ms per iteration (events "
                  "included), optimizer step %d, skipped-step flag %d" % (
                      it, mean, model.get_xyz.shape[0], model.active_sh_degree, 1e3 * dt / 50,
                      int(model.optimizer.state_dev[0]), int(model.optimizer.state_dev[1])))
            window, t_win = [], time.perf_counter()
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(p).all()) for p in model.leaf_parameters())
    print("MINI done: loss %.5f -> %.5f, parameters finite: %s, %d Gaussians, %.1f MB allocated, events %.1f ms in all" % (
        first, mean, finite, model.get_xyz.shape[0], torch.cuda.memory_allocated() / 1e6, 1e3 * events))


if __name__ == "__main__":
    main()
