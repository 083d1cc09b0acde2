"""CPU model of K8's launch schedule (tools/cellstats): how much of the kernel is packing loss -- workgroups of 4 waves
holding their CU slot until the slowest wave is done, and the tail of the launch -- and what a heavy-tiles-first order
would recover.  Per-cell hits come from the oracle's lists (stored masks); costs in cycles are the measured phase times of
k_render_bwd_cells (profiles/r03a): per tile prologue, per cell overhead, per chunk.

    python tools/cellstats/schedule_sim.py cfg3
"""
import ctypes
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402
from tools.cellstats.run import build  # noqa: E402

PROLOGUE, PER_CELL, PER_CHUNK = 6000.0, 7700.0, 3300.0   # cycles (5 waves per SIMD resident), profiles/r03a
SLOTS = 256 * 5                                           # workgroups resident on the chip


def tile_durations(hits):
    """hits [T,16] -> duration of each tile's workgroup: 4 waves draw cells as they finish (index order)."""
    T = hits.shape[0]
    out = np.zeros(T)
    for t in range(T):
        h = hits[t]
        if h.sum() == 0:
            continue
        waves = [0.0, 0.0, 0.0, 0.0]
        for c in range(16):
            if h[c]:
                i = int(np.argmin(waves))
                waves[i] += PER_CELL + PER_CHUNK * ((int(h[c]) + 15) // 16)
        out[t] = PROLOGUE + max(waves)
    return out


def makespan(durs, order):
    heap = [0.0] * SLOTS
    heapq.heapify(heap)
    end = 0.0
    for t in order:
        if durs[t] == 0:
            continue
        s = heapq.heappop(heap)
        e = s + durs[t]
        end = max(end, e)
        heapq.heappush(heap, e)
    return end


def xcd_order(T):
    run = 8
    grid = (T + run - 1) // run
    grid = (grid + 7) // 8 * 8 * run
    o = []
    for b in range(grid):
        xcd, k = b & 7, b >> 3
        j, off = k // run, k % run
        t = (j * 8 + xcd) * run + off
        if t < T:
            o.append(t)
    return o


def main():
    L = build()
    for cfg in sys.argv[1:] or ["cfg3"]:
        spec = syn.CONFIGS[cfg]
        ri = syn.raster_inputs(spec)
        n = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in ri.items()}
        out, radii, st = oracle.rasterize_forward(n["bg"], n["means3D"], n["colors"], n["opacities"], n["viewmatrix"],
                                                  n["projmatrix"], n["tanfovx"], n["tanfovy"], spec.H, spec.W,
                                                  cov3D_precomp=n["cov3D"], conic_precomp=n["conic"])
        T = st.ranges.shape[0]
        res = np.zeros(32, np.float64)
        hc, he, hp = np.zeros(130, np.uint64), np.zeros(130, np.uint64), np.zeros(17, np.uint64)
        hits = np.zeros((T, 16), np.uint16)
        p = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        ranges = np.ascontiguousarray(st.ranges, np.uint32)
        L.cellstats(ctypes.c_int(spec.W), ctypes.c_int(spec.H), p(ranges), p(st.point_list), p(st.xy),
                    p(st.conic_opacity), p(st.n_contrib), p(res), p(hc), p(he), p(hp), p(hits))
        d = tile_durations(hits)
        work = d.sum()
        ideal = work / SLOTS
        raster = makespan(d, xcd_order(T))
        lpt = makespan(d, list(np.argsort(-d)))
        # waves instead of workgroups as the unit that holds a slot (what 1-wave workgroups would give)
        wave_work = 0.0
        for t in range(T):
            h = hits[t]
            if h.sum():
                wave_work += 4 * PROLOGUE + sum(PER_CELL + PER_CHUNK * ((int(x) + 15) // 16) for x in h if x)
        print("== %s: %d working tiles; workgroup cycles: sum %.3g, mean %.0f, p99 %.0f, max %.0f" %
              (spec.name, (d > 0).sum(), work, d[d > 0].mean(), np.percentile(d[d > 0], 99), d.max()))
        print("   makespan on %d workgroup slots (cycles): perfect packing %.0f | launch order (XCD-interleaved raster) "
              "%.0f (%.2fx) | heaviest tile first %.0f (%.2fx)" % (SLOTS, ideal, raster, raster / ideal, lpt, lpt / ideal))
        print("   slot-cycles held idle inside workgroups (waves waiting for their tile's slowest wave): %.1f %%" %
              (100 * (1 - wave_work / (4 * work))))


if __name__ == "__main__":
    main()
