#!/usr/bin/env python
"""bench.py -- the measured hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A *step* is one global gradient step of the reference's stage-1 loop shape (src/train_gaussians.py:96-181, no
densification) over the whole view batch: per rank ``views_per_gpu`` views (default 4 = the per-GPU shard
of BASELINE.json configs[3], "32 views sharded 4-per-GPU on 8 GPUs"; the same 4 views per GPU at every N, so the series
is weak scaling) of the 500k strand-aligned model at 1920x1080 -- render() (fused HIP
projection + rasterizer forward), the four stage-1 losses incl. the orientation term (lambda_dorient = 0.1 as in
run.sh:112-115), backward (HIP loss / rasterizer / projection backward), one flat all-reduce of the
Gaussian gradients when N > 1 (RCCL), Adam.  Weak scaling: per-GPU work is fixed as N grows.  At N = 1 the line also
carries ``single_view_step``: the same loop with ONE view per step (BASELINE.json configs[2]).

One JSON line on rank 0:  value = Gaussians rasterized per second over the whole job = N * views_per_gpu * P / t_step
(inputs resident in HBM; P = Gaussians of the model, every one of them goes through projection, cull and -- if
visible -- the rasterizer, forward and backward, each step).
  roofline     : k_render_bwd (the dominant kernel), algorithmic bytes / HIP-event duration recorded around the kernel
                 inside the timed steps on the launch stream, against 8 TB/s HBM.
  cpu_baseline : the CPU oracle (oracle/ghr_oracle.c, OpenMP) on ONE view of the same workload, rasterizer fwd+bwd only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg5", "tiny"])
    ap.add_argument("--views-per-gpu", type=int, default=4,
                    help="views per GPU per global step; 4 = the per-GPU shard of BASELINE.json configs[3] (32 views on 8 GPUs)")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the views of a step alternate on (default: trainer's choice, 2; 1 = one stream, "
                         "the setting per-kernel rocprof averages should be taken with)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-op-only", action="store_true")
    args = ap.parse_args()

    from gaussianhaircut_amd import _lib
    from gaussianhaircut_amd import diff_gaussian_rasterization as dgr
    from gaussianhaircut_amd.parallel import FlatGradBucket, init_distributed
    from gaussianhaircut_amd.scene.cameras import ring_cameras
    from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
    from gaussianhaircut_amd.trainer import make_ground_truth, training_step
    from gaussianhaircut_amd.utils import synthetic as syn

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    # GHR_BENCH_BACKEND=gloo + GHR_BENCH_SHARE_GPU=1: functional check of the N > 1 path on a box with fewer GPUs than
    # ranks (ranks share devices, the gradient all-reduce goes through gloo); never a performance number
    share = os.environ.get("GHR_BENCH_SHARE_GPU") == "1"
    rank, world = init_distributed(backend=os.environ.get("GHR_BENCH_BACKEND") or None)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if share:
        local_rank %= torch.cuda.device_count()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = _lib.lib()

    spec = syn.CONFIGS[args.workload]
    V = args.views_per_gpu
    global_views = V * world
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1  # the reference's stage-1 command line (run.sh:112-115)

    # ---- model replica (identical on every rank: CPU-seeded), per-rank views, synthetic ground truth ----------------
    model = syn.make_model(spec, dev)
    all_cams = ring_cameras(global_views, spec.W, spec.H, device=dev)  # camera 0 == the SURVEY front camera
    cams = all_cams[rank::world][:V]
    bg = syn.background(dev)
    with torch.no_grad():
        gt = syn.make_model(spec, dev)
        g = torch.Generator(device="cpu").manual_seed(202)
        gt._xyz.add_((0.002 * torch.randn(gt._xyz.shape, generator=g)).to(dev))
        gt._features_dc.add_((0.05 * torch.randn(gt._features_dc.shape, generator=g)).to(dev))
        make_ground_truth(gt, cams, bg)
        del gt
    model.training_setup(opt)  # FusedAdam on ROCm: flat params / grads / moments
    bucket = None
    torch.cuda.synchronize()

    K, Wm = args.steps, args.warmup
    from gaussianhaircut_amd import trainer as _tr
    from gaussianhaircut_amd.gaussian_renderer import render as _render
    L.ghr_set_profile_events(None, None, None, None)

    def step(it, timed):
        return training_step(model, cams, bg, opt, it, bucket=bucket, global_views=global_views, streams=args.streams)

    for i in range(Wm):
        step(i + 1, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(Wm + i + 1, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # ---- roofline leg: HIP events around the two render kernels, recorded by the library on the launch stream.  In the
    # timed steps above the kernels of two views share the GPU (two streams), so a kernel's wall time there is not a
    # property of the kernel; here the same view (this rank's first camera, same parameters) runs alone:
    # render + loss + backward, K passes, no optimizer step (the accumulated gradients are dropped afterwards).
    n_ev = max(K, 10)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_ev)]
    for quad in evs:
        for e in quad:
            e.record()  # materialise the hipEvent_t
    torch.cuda.synchronize()
    V1 = 1.0 / global_views
    for q in [None, None] + evs:
        if q is not None:
            L.ghr_set_profile_events(*[ctypes.c_void_p(e.cuda_event) for e in q])
        pkg = _render(cams[0], model, _tr.PIPE, bg)
        loss = _tr.view_loss(pkg, cams[0], opt, scale=V1)
        loss.backward()
    L.ghr_set_profile_events(None, None, None, None)
    torch.cuda.synchronize()
    model.optimizer.flat_grad.zero_()
    model.optimizer.state_dev[1:2].zero_()
    model.optimizer._direct_backwards = 0
    torch.cuda.synchronize()
    replicas_identical = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # identical reduced gradients + identical Adam => the replicas must still be bit-identical
        bits = model.optimizer.flat_param.view(torch.int32).to(torch.int64)
        ck = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()])
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))

    stats = dict(dgr.LAST_STATS)
    P_model = spec.P
    ms_per_step = 1e3 * elapsed / K
    value = world * V * P_model / (elapsed / K)

    fwd_ms = [q[0].elapsed_time(q[1]) for q in evs]
    bwd_ms = [q[2].elapsed_time(q[3]) for q in evs]
    fwd_avg, bwd_avg = sum(fwd_ms) / len(fwd_ms), sum(bwd_ms) / len(bwd_ms)

    R, Pv = stats.get("num_rendered", 0), stats.get("P", 0)
    N_pix = spec.W * spec.H
    T_tiles = ((spec.W + 15) // 16) * ((spec.H + 15) // 16)
    # K8's share of B_bwd = per instance 68 B read + 64 B gradient payload, per pixel 48 B, ranges 8 B/tile
    bytes_bwd_kernel = 132 * R + 48 * N_pix + 8 * T_tiles
    bytes_fwd_kernel = 68 * R + 48 * N_pix + 8 * T_tiles
    achieved = bytes_bwd_kernel / (bwd_avg * 1e-3) / 1e9 if bwd_avg > 0 else 0.0
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_k_render_bwd.json")
    if os.path.exists(pmc_file) and args.workload == "cfg3":  # the committed PMC passes were taken on this workload
        try:
            traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": "k_render_bwd", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_bwd_kernel, "avg_kernel_ms": round(bwd_avg, 4),
                "note": "VALU/LDS-bound gradient walk; algorithmic bytes per SURVEY.md 8(d); kernel duration from HIP "
                        "events over %d solo passes of one view (in the timed steps two views share the GPU)" % n_ev}

    out = {
        "metric": "gaussians_rasterized_per_sec_fwd_bwd_1080p", "value": round(value, 1), "unit": "Gaussians/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d Gaussians (%s), %d view(s)/GPU/step at %dx%d, render + 4 stage-1 losses (L1, SSIM, mask, orientation 0.1) + backward%s + Adam" %
                   (spec.name, P_model, spec.kind, V, spec.W, spec.H, "+RCCL grad all-reduce" if world > 1 else ""),
                   "views_per_gpu": V, "global_views": global_views, "parallelism": "view-dp%d" % world,
                   "P_rasterized_per_view": Pv, "num_rendered_per_view": R},
        "grad_steps_per_sec": round(K / elapsed, 3),
        "kernels_ms": {"k_render_fwd": round(fwd_avg, 4), "k_render_bwd": round(bwd_avg, 4),
                       "k_render_fwd_hbm_frac": round(bytes_fwd_kernel / (fwd_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                       if fwd_avg > 0 else None},
        "roofline": roofline,
    }
    if replicas_identical is not None:
        out["replicas_identical"] = replicas_identical

    if rank == 0 and world == 1:
        # BASELINE.json configs[2]: the same stage-1 step with ONE view per gradient step
        K1 = max(10, K)
        for i in range(3):
            training_step(model, cams[:1], bg, opt, Wm + K + i + 1, global_views=1, streams=args.streams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(K1):
            training_step(model, cams[:1], bg, opt, Wm + K + 3 + i + 1, global_views=1, streams=args.streams)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / K1
        out["single_view_step"] = {"workload": "BASELINE configs[2]: 1 view per gradient step", "steps": K1,
                                   "ms_per_step": round(1e3 * dt1, 4), "gaussians_per_sec": round(P_model / dt1, 1),
                                   "grad_steps_per_sec": round(1.0 / dt1, 2)}
        if not args.no_op_only:
            out["op_only"] = op_only_bench(dev)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, model, cams[0])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def op_only_bench(dev, cfg="cfg2", iters=50, warm=10):  # 10 warm-up + 50 timed, median and p10 / p90
    """Rasterizer op alone -- the product's GaussianRasterizer autograd op (mode A: conic + colours precomputed, what
    render() hands it), forward and backward: BASELINE.json configs[1], Gaussians / (t_fwd + t_bwd)."""
    from gaussianhaircut_amd import diff_gaussian_rasterization as dgr
    from gaussianhaircut_amd.utils import synthetic as syn
    spec = syn.CONFIGS[cfg]
    ri = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in syn.raster_inputs(spec).items()}
    dL = (syn.grad_image(spec, 101) * (spec.H * spec.W)).to(dev).contiguous()
    rs = dgr.GaussianRasterizationSettings(image_height=spec.H, image_width=spec.W, tanfovx=ri["tanfovx"],
                                           tanfovy=ri["tanfovy"], bg=ri["bg"], scale_modifier=1.0,
                                           viewmatrix=ri["viewmatrix"], projmatrix=ri["projmatrix"], sh_degree=3,
                                           campos=ri["campos"], prefiltered=True, debug=False)
    rast = dgr.GaussianRasterizer(rs)
    leaves = {k: ri[k].clone().requires_grad_(True) for k in ("means3D", "means2D", "colors", "opacities", "conic")}
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf, tb = [], []
    for i in range(iters + warm):
        for t in leaves.values():
            t.grad = None
        e[0].record()
        color, radii = rast(means3D=leaves["means3D"], means2D=leaves["means2D"], shs=None,
                            colors_precomp=leaves["colors"], opacities=leaves["opacities"], cov3D_precomp=ri["cov3D"],
                            conic_precomp=leaves["conic"])
        e[1].record()
        torch.autograd.backward(color, grad_tensors=dL)
        e[2].record()
        torch.cuda.synchronize()
        if i >= warm:
            tf.append(e[0].elapsed_time(e[1]))
            tb.append(e[1].elapsed_time(e[2]))
    tf.sort(), tb.sort()
    mf, mb = tf[len(tf) // 2], tb[len(tb) // 2]
    pct = lambda v, q: round(v[min(len(v) - 1, int(q * len(v)))], 4)
    return {"workload": spec.name, "P": ri["P"], "num_rendered": dgr.LAST_STATS["num_rendered"], "fwd_ms": round(mf, 4),
            "bwd_ms": round(mb, 4), "gaussians_per_sec_fwd_bwd": round(ri["P"] / ((mf + mb) * 1e-3), 1),
            "fwd_ms_p10_p90": [pct(tf, 0.1), pct(tf, 0.9)], "bwd_ms_p10_p90": [pct(tb, 0.1), pct(tb, 0.9)],
            "iters": iters, "warmup": warm,
            "note": "GaussianRasterizer op (autograd, workspace allocation and the num_rendered read included)"}


def cpu_baseline(spec, model, cam):
    """The oracle (CPU restatement of the reference's CUDA semantics; 'port') on one view of the same workload."""
    import oracle
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests import helpers as hp
    cpu_model_inputs = syn.raster_inputs(spec, "cpu")
    dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)
    best = None
    for _ in range(3):  # bounded sample: three passes over one view (~1 s each on a 128-core host), best pass reported
        t0 = time.perf_counter()
        out_o, radii_o, st = hp.oracle_forward(oracle, cpu_model_inputs, "A")
        t1 = time.perf_counter()
        hp.oracle_backward(oracle, st, cpu_model_inputs, dL, "A")
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[2] - best[0]:
            best = (t0, t1, t2)
    t0, t1, t2 = best
    return {"value": round(spec.P / (t2 - t0), 1), "unit": "Gaussians/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "1 view of %s, best of 3 passes, rasterizer fwd (%.2f s) + bwd (%.2f s) only; projection/loss/Adam not included" %
                      (spec.name, t1 - t0, t2 - t1)}


if __name__ == "__main__":
    main()
