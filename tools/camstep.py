"""One mode of the configs[2] step, alone, for rocprofv3 --kernel-trace --stats (round 6):
    python tools/camstep.py plain|leaf|residual|fixed [steps]
plain = bench.py's headline (16 cycling cameras), fixed = camera 0 only, leaf = the five camera tensors are leaves that
require grad, residual = scene.cameras.TrainableCamera + torch.optim.Adam over the camera parameters."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd.scene.cameras import TrainableCamera, ring_cameras  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams  # noqa: E402
from gaussianhaircut_amd.trainer import make_ground_truth, training_step  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402

LEAVES = ("world_view_transform", "full_proj_transform", "camera_center", "FoVx", "FoVy")


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS[os.environ.get("CAMSTEP_CFG", "cfg3")]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    bg = syn.background(dev)
    model = syn.make_model(spec, dev)
    pool = ring_cameras(16, spec.W, spec.H, device=dev, cls=TrainableCamera if mode == "residual" else None)
    with torch.no_grad():
        gt = syn.make_model(spec, dev)
        g = torch.Generator(device="cpu").manual_seed(202)
        gt._xyz.add_((0.002 * torch.randn(gt._xyz.shape, generator=g)).to(dev))
        make_ground_truth(gt, pool, bg)
        del gt
    model.training_setup(opt)
    after = None
    if mode == "fixed":
        pool = pool[:1]
    if mode == "leaf":
        lp = []
        for c in pool:
            c2 = copy.copy(c)
            for n in LEAVES:
                setattr(c2, n, getattr(c, n).detach().clone().requires_grad_(True))
            lp.append(c2)
        pool = lp

        def after(c):
            for n in LEAVES:
                getattr(c, n).grad = None
    if mode == "residual":
        cam_opt = torch.optim.Adam([{"params": [c._rotation_res for c in pool], "lr": 0.001},
                                    {"params": [c._translation_res for c in pool], "lr": 0.0016},
                                    {"params": [c._fov_res for c in pool], "lr": 0.001}], lr=0.0, eps=1e-15)

        def after(c):
            cam_opt.step()
            cam_opt.zero_grad(set_to_none=True)
    for i in range(5):
        training_step(model, [pool[i % len(pool)]], bg, opt, i + 1)
        if after:
            after(pool[i % len(pool)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for i in range(K):
        h0 = time.perf_counter()
        training_step(model, [pool[(5 + i) % len(pool)]], bg, opt, 6 + i)
        if after:
            after(pool[(5 + i) % len(pool)])
        host += time.perf_counter() - h0
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("CAMSTEP %s: %.4f ms per step over %d steps (host issue time %.4f ms per step)" % (mode, 1e3 * dt / K, K, 1e3 * t_issue / K))


if __name__ == "__main__":
    main()
