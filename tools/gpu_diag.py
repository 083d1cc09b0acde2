"""Non-asserting stage-by-stage GPU diagnostic (prints mismatch statistics vs the oracle).  Used on the GPU box to
get maximum information out of one gpurun call: python tools/gpu_diag.py [cfg ...]"""
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402
from tests import helpers as hp  # noqa: E402
from tests.gpu_helpers import GpuRun, to_dev  # noqa: E402


def diag(cfg, mode):
    spec = syn.CONFIGS[cfg]
    dev = torch.device("cuda:0")
    ri = syn.raster_inputs(spec)
    t = time.time()
    out_o, radii_o, st_o = hp.oracle_forward(oracle, ri, mode)
    t_of = time.time() - t
    run = GpuRun(to_dev(ri, dev), mode, debug=True)
    ins = run.inspect()
    print("== %s %s P=%d R_gpu=%d R_oracle=%d oracle_fwd %.2fs" % (cfg, mode, ri["P"], run.R, st_o.num_rendered, t_of))
    r = run.radii.cpu().numpy()
    print("   radii mismatches:", int((r != radii_o).sum()))
    vis = radii_o > 0
    print("   depth bits mismatches:", int((ins["depths"][vis].view(np.uint32) != st_o.depths[vis].view(np.uint32)).sum()))
    print("   xy bits mismatches:", int((ins["rec"][vis, 0:2].view(np.uint32) != st_o.xy[vis].view(np.uint32)).sum()))
    print("   conic/opacity bits mismatches:", int((ins["rec"][vis, 2:6].view(np.uint32) != st_o.conic_opacity[vis].view(np.uint32)).sum()))
    ts = ins["tile_start"]
    rng = np.stack([ts[:-1], ts[1:]], 1).astype(np.uint32)
    rng[ts[:-1] == ts[1:]] = 0
    print("   ranges mismatches:", int((rng != st_o.ranges).sum()))
    if run.R == st_o.num_rendered:
        print("   point_list mismatches:", int((ins["point_list"] != st_o.point_list).sum()))
    frag = st_o.fragile.reshape(-1).astype(bool)
    ok = ~frag
    print("   fragile px:", int(frag.sum()), " n_contrib mismatches (non-fragile):", int((ins["n_contrib"][ok] != st_o.n_contrib[ok]).sum()))
    print("   final_T max err:", float(np.abs(ins["final_T"][ok] - st_o.final_T[ok]).max()))
    got = run.out.cpu().numpy().reshape(10, -1)
    ref = out_o.reshape(10, -1)
    err = np.abs(got - ref)
    print("   image max err non-fragile per channel:", np.round(err[:, ok].max(axis=1), 7).tolist())
    print("   image max err fragile:", float(err[:, frag].max()) if frag.any() else 0.0, " nan:", int(np.isnan(got).sum()))
    dL = syn.grad_image(spec, 101).numpy() * (spec.H * spec.W)
    dL[:, st_o.fragile.astype(bool)] = 0
    t = time.time()
    ref_g = hp.oracle_backward(oracle, st_o, ri, dL, mode)
    t_ob = time.time() - t
    got_g = run.backward(torch.from_numpy(dL))
    for k in ref_g:
        a, b = got_g[k].reshape(-1), ref_g[k].reshape(-1)
        sc = np.abs(b).max() if b.size else 0
        bad = ~hp.grad_close(a, b)
        print("   grad %-14s max|ref| %.3e  max abs err %.3e  rel-to-max %.2e  bad %d/%d nan %d" %
              (k, sc, float(np.abs(a - b).max()) if a.size else 0, float(np.abs(a - b).max() / (sc + 1e-30)) if a.size else 0,
               int(bad.sum()), bad.size, int(np.isnan(a).sum())))
    print("   oracle bwd %.2fs" % t_ob)


if __name__ == "__main__":
    cases = [("tiny", "A"), ("tiny", "B_sr"), ("tiny_strands", "A"), ("cfg1", "A"), ("cfg2", "A")]
    if len(sys.argv) > 1:
        cases = [(c, "A") for c in sys.argv[1:]]
    for c, m in cases:
        try:
            diag(c, m)
        except Exception:
            traceback.print_exc()
