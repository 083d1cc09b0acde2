"""Where the HOST's time per training step goes (round 6): cProfile over steps of a workload small enough that the GPU idles
(cfg1), so that wall time = host time.    python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd.scene.cameras import ring_cameras  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams  # noqa: E402
from gaussianhaircut_amd.trainer import make_ground_truth, training_step  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["cfg1"]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    bg = syn.background(dev)
    model = syn.make_model(spec, dev)
    pool = ring_cameras(4, spec.W, spec.H, device=dev)
    with torch.no_grad():
        gt = syn.make_model(spec, dev)
        gt._features_dc.add_(0.2)
        make_ground_truth(gt, pool, bg)
    model.training_setup(opt)
    for i in range(50):
        training_step(model, [pool[i % 4]], bg, opt, i + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        training_step(model, [pool[i % 4]], bg, opt, 51 + i)
    torch.cuda.synchronize()
    print("HOST %.4f ms per step (cfg1: the GPU idles)" % (1e3 * (time.perf_counter() - t0) / K))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(K):
        training_step(model, [pool[i % 4]], bg, opt, 51 + K + i)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    for line in s.getvalue().splitlines():
        if line.strip():
            print("HOST", line[:200])


if __name__ == "__main__":
    main()
