"""Phase profile of k_project / k_project_bwd inside the single-view step (a build of
tools/experiments/r05_k_project_and_bwd_phase_profile.patch with -DGHR_K8_PROF [-DGHR_PROF_K1], selected by GHR_LIB_PATH):
runs bench.py's step a few times, then reads the per-wave cycle counters the LAST launch left."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from gaussianhaircut_amd import _lib  # noqa: E402

P_MODEL = 500000


def main():
    sys.argv = ["bench.py", "--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--no-op-only", "--streams", "1",
                "--shard-views", "0"]
    bench.main()
    L = _lib.lib()
    assert hasattr(L, "ghr_debug_prof"), "not a -DGHR_K8_PROF build"
    n_waves = 65536  # (k_project: slot 4 b + wave; k_project_bwd, one wave per workgroup: slot 4 b; unused slots stay 0)
    buf = np.zeros((65536, 8), np.uint64)
    rc = L.ghr_debug_prof(ctypes.c_void_p(buf.ctypes.data), 65536, 0)
    assert rc == 0, rc
    v = buf[:n_waves].astype(np.float64)
    used = v[:, 6] > 0
    tot = v[used].sum(axis=0)
    names = os.environ.get("GHR_PROF_NAMES", "p0,p1,p2,p3,p4,p5,total,p7").split(",")
    print("PHASES " + "  ".join("%s=%.1f%%" % (nm, 100 * tot[i] / tot[6]) for i, nm in enumerate(names) if i != 6))
    c = v[used, 6]
    print("PHASES waves=%d shader cycles per wave: mean %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" %
          (used.sum(), c.mean(), np.percentile(c, 10), np.percentile(c, 50), np.percentile(c, 90), c.max()))
    for i, nm in enumerate(names):
        if i != 6:
            print("PHASES  %-28s mean %7.0f  p50 %7.0f  p90 %7.0f" % (nm, v[used, i].mean(), np.percentile(v[used, i], 50),
                                                                      np.percentile(v[used, i], 90)))


if __name__ == "__main__":
    main()
