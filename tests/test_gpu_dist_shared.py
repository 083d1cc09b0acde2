"""`-m gpu`: the N > 1 path bench.py takes -- trainer.training_step -> FusedAdam.step_chunked(reduce=True): two view
streams per rank, chunked asynchronous gradient all-reduces, each chunk's Adam update as its sum arrives, the NaN flag
OR-ed over the ranks first -- run by TWO ranks that share cuda:0, with gloo carrying the collectives (the builder's and
the driver's test boxes have one GPU; RCCL itself is exercised by the driver's multi-GPU bench).  SURVEY 8(e):
  * the all-reduced flat gradient equals the 1-rank accumulation of the same 8 views to 1e-5 of its largest entry,
  * the replicas stay bit-identical over several steps (also with the SH bands above the active degree left out of
    the all-reduce),
  * a non-finite gradient on ONE rank skips the step on BOTH (parameters, moments and step counter untouched).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

VIEWS, STEPS = 8, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(dev, sh_degree):
    from gaussianhaircut_amd.scene.cameras import ring_cameras
    from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
    from gaussianhaircut_amd.trainer import make_ground_truth
    from gaussianhaircut_amd.utils import synthetic as syn
    spec = syn.CONFIGS["tiny_strands"]
    opt = OptimizationParams()
    opt.lambda_dorient = 0.1
    model, gt = syn.make_model(spec, dev), syn.make_model(spec, dev)
    model.active_sh_degree = sh_degree
    with torch.no_grad():
        gt._features_dc.add_(0.25)
        gt._xyz.add_(0.003 * torch.randn(gt._xyz.shape, generator=torch.Generator().manual_seed(5)).to(dev))
    cams = ring_cameras(VIEWS, spec.W, spec.H, device=dev)
    bg = syn.background(dev)
    make_ground_truth(gt, cams, bg)
    model.training_setup(opt)
    return model, cams, bg, opt


def _capture_reduced_gradient(model, store):
    """step / step_chunked with the gradient zeroing taken out, so the (reduced) flat gradient can be copied first"""
    o = model.optimizer
    orig_c, orig_s = o.step_chunked, o.step

    def chunked(chunks=4, zero_grad=True, reduce=False, shard=None):
        # (replicated update: with the sharded one a rank only ever holds ITS slices of the reduced gradient)
        orig_c(chunks=chunks, zero_grad=False, reduce=reduce, shard=False)
        store.append(o.flat_grad.detach().clone())
        o.flat_grad.zero_()

    def step(zero_grad=True, nan_scan=True):
        o.fold_own_views()  # (a multi-view step on one rank keeps its SH gradients as per-view tables until the update)
        store.append(o.flat_grad.detach().clone())
        orig_s(zero_grad=zero_grad, nan_scan=nan_scan)

    o.step_chunked, o.step = chunked, step


def _worker(rank, world, port, q, sh_degree, poison_rank, shard=None, factored=True):
    import torch.distributed as dist
    from gaussianhaircut_amd.parallel import param_checksum, shard_views
    from gaussianhaircut_amd.trainer import training_step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, cams, bg, opt = _scene(dev, sh_degree)
    mine = shard_views(cams, rank, world)
    from gaussianhaircut_amd import optim as _optim
    _optim.FACTORED_SH_REDUCE = bool(factored)  # SH gradients as per-view factors (all-gather) or as sums (all-reduce)
    rebuilds = []
    _orig_rebuild = model.optimizer._rebuild_sh_from_views
    model.optimizer._rebuild_sh_from_views = lambda g, **k: (rebuilds.append(tuple(g.shape)), _orig_rebuild(g, **k))[1]
    grads = []
    if shard is None:
        _capture_reduced_gradient(model, grads)
    else:  # ZeRO-1 against the replicated update: the trainer's own call, with the optimizer sharded or not
        from gaussianhaircut_amd import _lib, optim
        optim.SHARD_ADAM = bool(shard)
        optim.SHARD_WITH_GATHERED_VIEWS = bool(shard)  # (by default the 13 floats left to sum next to gathered views are not sharded)
        _lib.lib().ghr_set_deterministic(1)  # two RUNS are compared bit for bit: the gradient walk's atomics must be ordered
        calls = []
        orig = model.optimizer.step_chunked
        model.optimizer.step_chunked = lambda **kw: (calls.append(kw), orig(**kw))[1]
    for it in range(STEPS):
        if shard is not None and it == 1:
            model.active_sh_degree = 2  # the SH-band plan changes: the sharded ranges move, their moments are synced first
        training_step(model, mine, bg, opt, it + 1, global_views=VIEWS)
    torch.cuda.synchronize()
    out = dict(rank=rank, checksum=param_checksum(model.leaf_parameters()),
               grad0=grads[0].cpu().numpy() if grads else None,
               step=int(model.optimizer.state_dev[0]), chunked=model.optimizer.flat_param.numel(), rebuilds=rebuilds)
    if shard is not None:
        o = model.optimizer
        out["stale"] = o._moment_shards is not None
        if o.moments_stale():  # a lone checkpoint call must fail loudly, not start the collective (optim.StaleMomentsError)
            from gaussianhaircut_amd.optim import StaleMomentsError
            try:
                o.state_dict()
                out["lone_state_dict"] = "returned"
            except StaleMomentsError:
                out["lone_state_dict"] = "raised"
        o.sync_moments()
        out.update(params=o.flat_param.cpu().numpy(), m=o.exp_avg.cpu().numpy(), v=o.exp_avg_sq.cpu().numpy(),
                   n_calls=len(calls), state=model.optimizer.state_dict()["state"][0]["exp_avg"].cpu().numpy())
    if poison_rank is not None:
        before = model.optimizer.flat_param.detach().clone()
        m_before = model.optimizer.exp_avg.detach().clone()
        if rank == poison_rank:  # this rank's ground truth makes its loss -- and all its gradients -- NaN
            mine[0].original_image = mine[0].original_image.clone()
            mine[0].original_image[:, :, mine[0].original_image.shape[2] // 2] = float("nan")  # a column through the hair
        training_step(model, mine, bg, opt, STEPS + 1, global_views=VIEWS)
        torch.cuda.synchronize()
        out["skipped"] = bool(torch.equal(model.optimizer.flat_param, before) and
                              torch.equal(model.optimizer.exp_avg, m_before) and
                              int(model.optimizer.state_dev[0]) == out["step"] and int(model.optimizer.state_dev[1]) == 0
                              # (the trainer's own step leaves the gradient buffer UNDEFINED -- zero_grad="defer" --; the
                              # capturing wrapper of the other tests zero-fills it)
                              and (shard is not None or float(model.optimizer.flat_grad.abs().sum()) == 0.0))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(sh_degree, poison_rank=None, shard=None, factored=True):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sh_degree, poison_rank, shard, factored)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(q, procs, 2)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _collect(q, procs, n, timeout=900):
    """The workers' results; fails as soon as one of them has died (its peers would wait in a collective until the timeout)."""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError("worker exit codes %s after %.0f s" % ([p.exitcode for p in procs], time.time() - t0))
    return sorted(out, key=lambda d: d["rank"])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("sh_degree,factored", [(3, True), (1, True), (3, False), (1, False)])
def test_two_ranks_on_one_gpu_through_step_chunked_reduce(sh_degree, factored):
    """``factored``: the SH gradients travel as per-view dL/d(rgb) tables (all-gather, 12 B per Gaussian and view) and are
    rebuilt on every rank (ABI 19, optim.FusedAdam.begin_factored_views) instead of being summed by the all-reduce."""
    from gaussianhaircut_amd.trainer import training_step
    res = _run_two_ranks(sh_degree, poison_rank=1 if sh_degree == 3 else None, factored=factored)
    assert res[0]["checksum"] == res[1]["checksum"], "replicas diverged"
    assert res[0]["step"] == res[1]["step"] == STEPS
    for r in res:  # 8 views on 2 ranks: four slots per rank, gathered rank-major, one rebuild per step (the poisoned one too)
        n = STEPS + (1 if sh_degree == 3 else 0)
        assert len(r["rebuilds"]) == (n if factored else 0) and all(s[0] == 8 for s in r["rebuilds"]), r["rebuilds"]
    np.testing.assert_array_equal(res[0]["grad0"], res[1]["grad0"])
    if sh_degree == 3:
        assert res[0]["skipped"] and res[1]["skipped"], "a non-finite gradient on one rank must skip the step on both"
    # one rank accumulating the same 8 views
    dev = torch.device("cuda:0")
    model, cams, bg, opt = _scene(dev, sh_degree)
    grads = []
    _capture_reduced_gradient(model, grads)
    training_step(model, cams, bg, opt, 1, global_views=VIEWS, fuse_adam=False)  # (the capture wraps optimizer.step())
    torch.cuda.synchronize()
    ref = grads[0].cpu().numpy()
    got = res[0]["grad0"]
    scale = np.abs(ref).max()
    assert scale > 0 and np.abs(got - ref).max() <= 1e-5 * scale, (np.abs(got - ref).max(), scale)
    if sh_degree < 3:  # the bands that were left out of the all-reduce are zero on the reference as well
        P = model.get_xyz.shape[0]
        rest = ref[6 * P: 51 * P].reshape(P, 15, 3)
        assert np.abs(rest[:, (sh_degree + 1) ** 2 - 1:]).max() == 0.0


@pytest.mark.timeout(1500)
def test_sharded_adam_equals_the_replicated_update_bit_for_bit_on_two_ranks():
    """ZeRO-1 (FusedAdam.step_chunked(shard=True): reduce-scatter, Adam on this rank's slice, all-gather of the parameters)
    against the replicated update (all-reduce, every rank updates everything) through trainer.training_step on two ranks:
    after three steps -- the second with a different active SH degree, so the sharded ranges move and the stale moments are
    synced in between -- the parameters of every rank are the same BITS in both runs, and so are the moments once
    ``sync_moments()`` has gathered them (a skipped step on a non-finite gradient included)."""
    runs = {}
    for shard in (True, False):
        res = _run_two_ranks(1, poison_rank=0, shard=shard)
        assert res[0]["checksum"] == res[1]["checksum"], "replicas diverged"
        assert res[0]["step"] == res[1]["step"] == STEPS and res[0]["n_calls"] == STEPS
        assert res[0]["skipped"] and res[1]["skipped"]
        for k in ("params", "m", "v", "state"):
            np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)
        assert res[0]["stale"] == shard   # sharded: the other rank's slices of the moments were stale until synced
        if shard:
            assert res[0]["lone_state_dict"] == "raised" and res[1]["lone_state_dict"] == "raised"
        runs[shard] = res[0]
    for k in ("params", "m", "v", "state"):
        np.testing.assert_array_equal(runs[True][k], runs[False][k], err_msg="sharded vs replicated: " + k)
    assert np.abs(runs[True]["m"]).max() > 0


def _nccl_worker(port, q):
    """ONE rank on backend "nccl" (= RCCL on ROCm), GHR_FORCE_COLLECTIVES=1: the collective branch of step_chunked runs
    through RCCL -- communicator init on the GPU, async work handles whose wait() is a STREAM dependency (not a host
    block as with gloo), collectives on RCCL's own stream ordered behind the current stream at call time, the `packed`
    temporaries of the SH-band plan alive until consumed."""
    import torch.distributed as dist
    from gaussianhaircut_amd.trainer import training_step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from gaussianhaircut_amd import _lib
    _lib.lib().ghr_set_deterministic(1)  # two RUNS are compared bit for bit: the gradient walk's atomics must be ordered
    out = {}
    from gaussianhaircut_amd import optim as _optim
    for deg in (3, 1):
        runs = []
        # degree 3: the four views' dL/d(rgb) tables are GATHERED (one rank: the gather is the identity) and the SH ranges are
        # not reduced; degree 1: the threshold is set to 0, so the rank only FOLDS its own four tables before the usual sums
        _optim.FACTORED_SH_MAX_VIEWS = 16 if deg == 3 else 0
        for force in ("1", "0"):
            os.environ["GHR_FORCE_COLLECTIVES"] = force
            model, cams, bg, opt = _scene(dev, deg)
            calls = []
            orig = model.optimizer.step_chunked
            model.optimizer.step_chunked = lambda **kw: (calls.append(kw), orig(**kw))[1]
            folds = []
            o_ = model.optimizer
            orig_fold = o_._rebuild_sh_from_views
            o_._rebuild_sh_from_views = lambda g, o_=o_, orig_fold=orig_fold, folds=folds, **k: (folds.append(o_._views["gather"]), orig_fold(g, **k))[1]
            for it in range(STEPS):
                training_step(model, cams[:4], bg, opt, it + 1, global_views=4)
            torch.cuda.synchronize()
            runs.append((model.optimizer.flat_param.detach().clone(), model.optimizer.exp_avg_sq.detach().clone(),
                         int(model.optimizer.state_dev[0]), len(calls), [c.get("reduce") for c in calls]))
            if force == "1":
                out["folds_%d" % deg] = list(folds)
            if force == "1" and deg == 3:  # a non-finite gradient skips the step through the collective branch too
                before = model.optimizer.flat_param.detach().clone()
                cams[0].original_image = cams[0].original_image.clone()
                cams[0].original_image[:, :, cams[0].original_image.shape[2] // 2] = float("nan")
                training_step(model, cams[:4], bg, opt, STEPS + 1, global_views=4)
                torch.cuda.synchronize()
                out["skipped"] = bool(torch.equal(model.optimizer.flat_param, before) and
                                      int(model.optimizer.state_dev[0]) == STEPS and int(model.optimizer.state_dev[1]) == 0)
        (p1, v1, s1, n1, r1), (p0, v0, s0, n0, r0) = runs
        out[deg] = dict(params_equal=bool(torch.equal(p1, p0)), v_equal=bool(torch.equal(v1, v0)), steps=(s1, s0),
                        max_diff=float((p1 - p0).abs().max()), n_diff=int((p1 != p0).sum()),
                        first_diff=int((p1 != p0).nonzero()[0]) if bool((p1 != p0).any()) else -1, n=int(p1.numel()),
                        chunked_calls=(n1, n0), reduce_flags=r1, finite=bool(torch.isfinite(p1).all()),
                        moved=float((p1 - _scene(dev, deg)[0].optimizer.flat_param).abs().max()))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_one_rank_on_rccl_through_the_collective_branch_is_bit_identical_to_the_local_step():
    """VERDICT r2 next #6a: the N > 1 step had only ever run over gloo.  Here RCCL itself carries it (one rank -- the box
    has one GPU): all-reduce of one rank = identity, so parameters and moments after 3 steps must equal, bit for bit,
    the run that takes the local branch; with active_sh_degree 1 the packed SH-band plan is on the wire."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=800)
    p.join(120)
    assert p.exitcode == 0
    for deg in (3, 1):
        r = res[deg]
        assert r["chunked_calls"][0] == STEPS and all(r["reduce_flags"]), r   # the collective branch was the one taken
        assert r["chunked_calls"][1] == 0, r                                     # ... and the local one otherwise
        assert r["params_equal"] and r["v_equal"] and r["steps"] == (STEPS, STEPS) and r["finite"], "deg %d: %s" % (deg, sorted(r.items()))
        assert r["moved"] > 0
        # every step of the collective branch rebuilt the SH gradients from the views' tables: gathered (3) / folded locally (1)
        if os.environ.get("GHR_FACTORED_SH_REDUCE", "1") != "0":
            assert res["folds_%d" % deg] == [deg == 3] * STEPS, res["folds_%d" % deg]
    assert res["skipped"]


def _empty_rank_worker(rank, world, port, q):
    """rank 1 holds NO views; between the first and the second step both ranks reset the opacities (optimizer surgery:
    the replaced group is marked `skip next step` until fresh gradients arrive)"""
    import torch.distributed as dist
    from gaussianhaircut_amd.parallel import param_checksum
    from gaussianhaircut_amd.trainer import training_step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, cams, bg, opt = _scene(dev, 3)
    mine = cams[:4] if rank == 0 else []
    training_step(model, mine, bg, opt, 1, global_views=4, views_per_rank=4)  # (rank 0 holds all four: say so)
    model.reset_opacity()
    before = model.optimizer.flat_param.detach().clone()
    training_step(model, mine, bg, opt, 2, global_views=4, views_per_rank=4)
    torch.cuda.synchronize()
    o = model.optimizer
    q.put(dict(rank=rank, checksum=param_checksum(model.leaf_parameters()), step=int(o.state_dev[0]),
               skipped=int(o.state_dev[1]), moved=float((o.flat_param - before).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_a_rank_without_views_stays_a_bit_identical_replica_across_optimizer_surgery():
    """a rank that holds no views in a step still takes part in the collectives and in the update; the
    `parameters replaced since the last backward` marks that optimizer surgery leaves are cleared by training_step on
    EVERY rank (not only where a backward ran), otherwise that rank's step would be a no-op and the replicas diverge."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(q, procs, 2, timeout=800)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0]["checksum"] == res[1]["checksum"], "replicas diverged"
    assert res[0]["step"] == res[1]["step"] == 2 and res[0]["skipped"] == res[1]["skipped"] == 0
    assert res[0]["moved"] > 0 and res[0]["moved"] == res[1]["moved"]
