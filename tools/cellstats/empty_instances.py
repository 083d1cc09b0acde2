"""CPU measurement (round 6): how many (Gaussian, tile) instances of a view touch NO pixel of their tile -- the square tile
rect of the reference's K1 (3 sigma of the LONGER axis, auxiliary.h:46-56) around a thin strand segment holds tiles the needle
never reaches.  Such an instance is scattered, sorted, given a zeroed gradient line and gathered in the projection backward for
nothing.  Criterion per pixel as forward.cu:351-360 (power <= 0, alpha >= 1/255), early termination ignored.

    python tools/cellstats/empty_instances.py cfg3 [cfg5 ...]

Measurement tool only (imports oracle/, like the tests): never on the product path.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    for cfg in sys.argv[1:] or ["cfg3"]:
        spec = syn.CONFIGS[cfg]
        ri = syn.raster_inputs(spec)
        n = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in ri.items()}
        _, radii, st = oracle.rasterize_forward(n["bg"], n["means3D"], n["colors"], n["opacities"], n["viewmatrix"],
                                                n["projmatrix"], n["tanfovx"], n["tanfovy"], spec.H, spec.W,
                                                cov3D_precomp=n["cov3D"], conic_precomp=n["conic"])
        gx = (spec.W + 15) // 16
        ranges = np.asarray(st.ranges, np.int64)
        pl = np.asarray(st.point_list, np.int64)
        xy, co = np.asarray(st.xy, np.float32), np.asarray(st.conic_opacity, np.float32)
        px, py = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32))
        empty = tot = 0
        hist = np.zeros(18, np.int64)   # cells (4x4 pixels) hit per instance: 0 .. 16
        per_gauss_tiles, per_gauss_live = np.zeros(st.P, np.int64), np.zeros(st.P, np.int64)
        for t in np.nonzero(ranges[:, 1] > ranges[:, 0])[0]:
            ids = pl[ranges[t, 0]:ranges[t, 1]]
            x0, y0 = np.float32(16 * (t % gx)), np.float32(16 * (t // gx))
            dx = xy[ids, 0][:, None, None] - (x0 + px)[None]
            dy = xy[ids, 1][:, None, None] - (y0 + py)[None]
            a, b, c, o = (co[ids, k][:, None, None] for k in range(4))
            power = np.float32(-0.5) * (a * dx * dx + c * dy * dy) - b * dx * dy
            inside = ((y0 + py) < spec.H)[None] & ((x0 + px) < spec.W)[None]
            hit = (power <= 0) & (np.minimum(np.float32(0.99), o * np.exp(power)) >= np.float32(1.0 / 255.0)) & inside
            cells = hit.reshape(-1, 4, 4, 4, 4).any(axis=(2, 4)).reshape(-1, 16).sum(axis=1)
            hist += np.bincount(cells, minlength=18)[:18]
            live = cells > 0
            empty += int((~live).sum())
            tot += ids.size
            np.add.at(per_gauss_tiles, ids, 1)
            np.add.at(per_gauss_live, ids, live.astype(np.int64))
        vis = per_gauss_tiles > 0
        print("== %s  P=%d  R=%d: instances that touch no pixel of their tile %d = %.1f %%; tiles per visible Gaussian %.2f, of "
              "which touched %.2f" % (spec.name, st.P, tot, empty, 100.0 * empty / max(tot, 1),
                                      per_gauss_tiles[vis].mean(), per_gauss_live[vis].mean()))
        print("   cells (4x4 pixels) hit per instance, 0 .. 16: %s" % " ".join(str(int(v)) for v in hist[:17]))


if __name__ == "__main__":
    main()
