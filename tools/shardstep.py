"""The 4-views-per-rank step of BASELINE configs[3] on ONE rank sent through the collective branch (backend nccl = RCCL,
GHR_FORCE_COLLECTIVES=1): what a rank of the 8-GPU run executes, minus the wire.  A/B of the SH-gradient forms (round 6):
    GHR_FACTORED_SH_REDUCE=0   every view's backward read-modify-writes 192 B of SH gradients per Gaussian
    GHR_FACTORED_SH_REDUCE=1   views leave dL/d(rgb) tables; folded before the sums (GHR_SH_MAX_VIEWS=0) or gathered (=16)
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd import optim  # noqa: E402
from gaussianhaircut_amd.scene.cameras import ring_cameras  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams  # noqa: E402
from gaussianhaircut_amd.trainer import make_ground_truth, training_step  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1",
                      GHR_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if "GHR_SH_MAX_VIEWS" in os.environ:
        optim.FACTORED_SH_MAX_VIEWS = int(os.environ["GHR_SH_MAX_VIEWS"])
    spec = syn.CONFIGS["cfg3"]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    model, gt = syn.make_model(spec, dev), syn.make_model(spec, dev)
    with torch.no_grad():
        gt._features_dc.add_(0.25)
    cams = ring_cameras(8, spec.W, spec.H, device=dev)
    bg = syn.background(dev)
    make_ground_truth(gt, cams, bg)
    del gt
    model.training_setup(opt)
    folds = []
    orig = model.optimizer._rebuild_sh_from_views
    model.optimizer._rebuild_sh_from_views = lambda g, **k: (folds.append(model.optimizer._views["gather"]), orig(g, **k))[1]
    for i in range(6):
        training_step(model, [cams[(i * V + j) % 8] for j in range(V)], bg, opt, i + 1, global_views=V)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        training_step(model, [cams[(i * V + j) % 8] for j in range(V)], bg, opt, 7 + i, global_views=V)
    t_issue = time.perf_counter()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / K
    print("SHARDSTEP host issue %.4f ms per step (the Python side of the step; the step is host-bound when this equals the total)" %
          (1e3 * (t_issue - t0) / K))
    if os.environ.get("SHARDSTEP_PROFILE"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(K):
            training_step(model, [cams[(i * V + j) % 8] for j in range(V)], bg, opt, 7 + K + i, global_views=V)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(45)
    print("SHARDSTEP views %d  GHR_FACTORED_SH_REDUCE=%s max_views %d  %s: %.4f ms per step" % (
        V, os.environ.get("GHR_FACTORED_SH_REDUCE", "1"), optim.FACTORED_SH_MAX_VIEWS,
        "SH tables %s" % ("gathered" if folds[-1] else "folded on the rank") if folds else "SH gradients accumulated in place", ms))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
