"""Strand-stage (render_hair) step timing on the GPU box: fused segmented projection vs the generic PyTorch path.
    python tools/bench_hair.py [n_strands] [n_head]"""
import sys
import time
from types import SimpleNamespace

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianhaircut_amd.gaussian_renderer import render_hair  # noqa: E402
from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands  # noqa: E402
from gaussianhaircut_amd.utils import synthetic as syn  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 5051
    n_head = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["cfg3"]
    head = syn.make_model(spec, dev)
    with torch.no_grad():
        head._label[:n_head] = -4.0
        head._label[n_head:] = 4.0
    head.precompute_head()
    g = torch.Generator().manual_seed(9)
    n_seg = 99
    origins = torch.nn.functional.normalize(torch.randn(S, 1, 3, generator=g), dim=-1)
    dirs = torch.randn(S, n_seg, 3, generator=g) * 0.003 + torch.nn.functional.normalize(torch.randn(S, 1, 3, generator=g), dim=-1) * 0.01
    feats = torch.randn(S * n_seg, 16, 3, generator=g) * 0.1
    hair = GaussianModelStrands(3).create_from_strands(origins.to(dev), dirs.to(dev), feats.to(dev))
    cam, bg = syn.make_view(spec, dev), syn.background(dev)
    w = torch.randn(6, spec.H, spec.W, device=dev)
    for name, fused in (("fused", True), ("generic", False)):
        pipe = SimpleNamespace(debug=False, fused_projection=fused)
        ts = []
        for it in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hair.initialize_gaussians_hair()
            pkg = render_hair(cam, head, hair, pipe, bg)
            full = torch.cat([pkg["render"], pkg["mask"], pkg["orient_conf"]], dim=0)
            (full * w).sum().backward()
            hair._dirs.grad = None
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        print("HAIR %s: head %d + strands %d x %d = %d Gaussians, 1080p, initialize+render_hair+backward: median %.2f ms" %
              (name, n_head, S, n_seg, n_head + S * n_seg, 1e3 * ts[len(ts) // 2]))


if __name__ == "__main__":
    main()
