"""Cost of one densification event at BASELINE size: the reference's densify_and_prune
(src/scene/gaussian_model.py:596-741: clone + split + prune with optimizer-state surgery) on the 500k strand model with
FusedAdam, thresholds chosen so that the model roughly doubles -- the "2M after densify / clone" road of BASELINE configs[4].

    python tools/densify_bench.py [cfg] > profiles/r06_densify.txt

It happens every opt.densification_interval = 100 iterations (arguments/__init__.py): its cost per iteration is 1 % of what
is printed here."""
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
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS[cfg]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    bg = syn.background(dev)
    model = syn.make_model(spec, dev)
    cams = ring_cameras(4, spec.W, spec.H, device=dev)
    with torch.no_grad():
        gt = syn.make_model(spec, dev)
        g = torch.Generator(device="cpu").manual_seed(202)
        gt._xyz.add_((0.002 * torch.randn(gt._xyz.shape, generator=g)).to(dev))
        gt._features_dc.add_((0.05 * torch.randn(gt._features_dc.shape, generator=g)).to(dev))
        make_ground_truth(gt, cams, bg)
        del gt
    model.training_setup(opt)
    # statistics of a few iterations, kept by the projection backward itself
    for it in range(8):
        training_step(model, [cams[it % 4]], bg, opt, it + 1, densify_stats=True)
    torch.cuda.synchronize()
    grads = (model.xyz_gradient_accum / model.denom.clamp_min(1)).reshape(-1)
    seen = model.denom.reshape(-1) > 0
    # threshold = median gradient of the Gaussians seen: about half of them are cloned or split
    thr = float(grads[seen].median())
    print("DENSIFY %s: P = %d, seen %d, threshold (median viewspace gradient) %.3e" % (spec.name, model.get_xyz.shape[0], int(seen.sum()), thr))
    for rep in range(3):
        P0 = model.get_xyz.shape[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            model.densify_and_prune(thr, 0.005, 2.5, 20, generator=torch.Generator(device=dev).manual_seed(rep))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        P1 = model.get_xyz.shape[0]
        print("DENSIFY event %d: %d -> %d Gaussians in %.2f ms (wall, synchronised) = %.3f ms per iteration at the reference's "
              "interval of 100" % (rep, P0, P1, 1e3 * dt, 10 * dt))
        # a few training steps on the resized model (buffers re-laid, capacity guesses re-learnt), then new statistics
        for it in range(6):
            training_step(model, [cams[it % 4]], bg, opt, 100 * (rep + 1) + it, densify_stats=it > 0,
                          defer_counts=it > 0)
        torch.cuda.synchronize()
        grads = (model.xyz_gradient_accum / model.denom.clamp_min(1)).reshape(-1)
        seen = model.denom.reshape(-1) > 0
        thr = float(grads[seen].quantile(0.8)) if int(seen.sum()) else thr
    t0 = time.perf_counter()
    for it in range(20):
        training_step(model, [cams[it % 4]], bg, opt, 1000 + it, densify_stats=True)
    torch.cuda.synchronize()
    print("DENSIFY after: %d Gaussians, %.4f ms per gradient step with the statistics kept in-kernel" %
          (model.get_xyz.shape[0], 1e3 * (time.perf_counter() - t0) / 20))


if __name__ == "__main__":
    main()
