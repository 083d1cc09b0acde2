"""CPU measurement: what the lists of k_render_bwd_cells hold -- chunk fill, share of half-empty
tail chunks, false-positive rate of K7's conservative cell masks -- over the ORACLE's sorted tile lists of one view.

    python tools/cellstats/run.py cfg3 [cfg2 ...]  > profiles/r03_cellstats.txt

Measurement tool only (imports oracle/, like the tests): never on the product path.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    so = os.path.join(HERE, "_build", "libcellstats.so")
    src = os.path.join(HERE, "cellstats.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O2", "-std=c++17",
                        "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", so, src], check=True)
    return ctypes.CDLL(so)


NAMES = ["pairs", "hits_cons", "hits_exact", "hits_cons_uncut", "chunks_cons", "chunks_exact", "chunks_cons_band",
         "chunks_exact_band", "chunks_cons_tile", "chunks_exact_tile", "chunks8_exact_8x4", "cells_cons", "cells_exact",
         "one_chunk_cells_cons", "one_chunk_cells_exact", "fp_margin", "fp_saturated", "half_tail_cons", "half_tail_exact",
         "pair_steps64_per_cell", "pair_steps64_per_tile"]


def main():
    L = build()
    for cfg in sys.argv[1:] or ["cfg3"]:
        spec = syn.CONFIGS[cfg]
        ri = syn.raster_inputs(spec)
        n = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in ri.items()}
        out, radii, st = oracle.rasterize_forward(n["bg"], n["means3D"], n["colors"], n["opacities"], n["viewmatrix"],
                                                  n["projmatrix"], n["tanfovx"], n["tanfovy"], spec.H, spec.W,
                                                  cov3D_precomp=n["cov3D"], conic_precomp=n["conic"])
        res = np.zeros(32, np.float64)
        hc, he, hp = np.zeros(130, np.uint64), np.zeros(130, np.uint64), np.zeros(17, np.uint64)
        p = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        ranges = np.ascontiguousarray(st.ranges, np.uint32)
        L.cellstats(ctypes.c_int(spec.W), ctypes.c_int(spec.H), p(ranges), p(st.point_list), p(st.xy),
                    p(st.conic_opacity), p(st.n_contrib), p(res), p(hc), p(he), p(hp), ctypes.c_void_p(0))
        r = dict(zip(NAMES, res))
        R = st.num_rendered
        print("== %s  P=%d  R=%d  tiles with work=%d" % (spec.name, st.P, R, int((ranges[:, 1] > ranges[:, 0]).sum())))
        for k in NAMES:
            print("  %-24s %12.0f" % (k, r[k]))
        print("  false positives of the stored masks: %.1f %% of the hits K8 walks (margins %.1f %%, every pixel already "
              "finished %.1f %%)" % (100 * (1 - r["hits_exact"] / r["hits_cons"]), 100 * r["fp_margin"] / r["hits_cons"],
                                   100 * r["fp_saturated"] / r["hits_cons"]))
        print("  chunk fill (hits / 16 chunks): stored masks %.3f, exact masks %.3f" %
              (r["hits_cons"] / 16 / r["chunks_cons"], r["hits_exact"] / 16 / r["chunks_exact"]))
        print("  pair slots used (pairs / 256 chunks): stored %.3f, exact %.3f; pixels per exact hit %.2f" %
              (r["pairs"] / 256 / r["chunks_cons"], r["pairs"] / 256 / r["chunks_exact"], r["pairs"] / r["hits_exact"]))
        print("  chunks per working cell: stored %.2f (%.1f %% of the cells have one chunk), exact %.2f (%.1f %%)" %
              (r["chunks_cons"] / r["cells_cons"], 100 * r["one_chunk_cells_cons"] / r["cells_cons"],
               r["chunks_exact"] / r["cells_exact"], 100 * r["one_chunk_cells_exact"] / r["cells_exact"]))
        print("  tail chunks at most half full: stored %.1f %% of all chunks, exact %.1f %%" %
              (100 * r["half_tail_cons"] / r["chunks_cons"], 100 * r["half_tail_exact"] / r["chunks_exact"]))
        print("  chunks relative to today's (%.0f): exact masks %.3f | exact + tails packed per band %.3f | per tile %.3f |"
              " stored + per band %.3f | 8x4 regions, 8-entry chunks (same pair slots) %.3f" %
              (r["chunks_cons"], r["chunks_exact"] / r["chunks_cons"], r["chunks_exact_band"] / r["chunks_cons"],
               r["chunks_exact_tile"] / r["chunks_cons"], r["chunks_cons_band"] / r["chunks_cons"],
               r["chunks8_exact_8x4"] / r["chunks_cons"]))
        print("  lanes bound to contributing pairs (DESIGN.md 11): wave steps of 64 pairs packed per cell %.0f (fill %.3f), per "
              "tile %.0f; today's chunks hold 256 pair slots each: %.0f wave steps of 64 slots -> %.2fx / %.2fx fewer" %
              (r["pair_steps64_per_cell"], r["pairs"] / 64 / r["pair_steps64_per_cell"], r["pair_steps64_per_tile"],
               4 * r["chunks_cons"], 4 * r["chunks_cons"] / r["pair_steps64_per_cell"],
               4 * r["chunks_cons"] / r["pair_steps64_per_tile"]))
        cum = np.cumsum(he[1:].astype(np.float64)) / max(he[1:].sum(), 1)
        print("  hits per working cell (exact): p50 %d p90 %d p99 %d" % tuple(1 + int(np.searchsorted(cum, q)) for q in (.5, .9, .99)))
        print("  histogram hits/cell exact (0..32):", " ".join(str(int(x)) for x in he[:33]))
        print("  histogram hits/cell stored (0..32):", " ".join(str(int(x)) for x in hc[:33]))
        print("  pixels per exact hit (1..16):", " ".join(str(int(x)) for x in hp[1:]))


if __name__ == "__main__":
    main()
