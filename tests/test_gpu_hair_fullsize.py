"""`-m gpu`: render_hair() AT SIZE -- the strand stage of the reference (src/gaussian_renderer/__init__.py:116-214;
src/train_strands.py:49: 30 000 strands x 99 segments over a frozen head) -- against the ORACLE CHAIN.

Scene: 100 000 frozen head Gaussians (blobs; the rows of a 200k model labelled "head") + 6061 strands x 99 segments =
600 039 strand Gaussians (src/scene/gaussian_model_strands.py:435-452), 1920x1080, SH degree 3.

    fused path (GPU):   k_project in its explicit mode on TWO segments of one rasterizer state (head rows 0 .. n_head,
                        padding up to a multiple of 256, strand rows from row0) -> binning -> K7 -> loss -> K8 ->
                        k_project_bwd -> autograd through initialize_gaussians_hair() to `_dirs`
    oracle chain (CPU): the same render_hair() host code with the PyTorch projection (mode A_sr of the op: conic AND
                        scales / rotations given) around oracle.rasterize_forward / backward

What is compared, element by element (no quantiles): K1 state of both segments for every Gaussian the two chains
rasterize identically (depth key bits, pixel mean bits, conic / opacity to 5e-6, strands 1e-5) with the decisions that differ COUNTED
(<= 1e-4 P); the padding rows between the segments are culled (radius 0, no instances); the sorted tile lists exactly;
n_contrib exactly and the image to 1e-4 off the named pixels (oracle-fragile + tiles of a flipped Gaussian); the gradients
w.r.t. `_dirs`, SH (dc, rest) and `orient_conf` per row for a seeded random dL/dout (leg A of the render() test).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gaussianhaircut_amd.utils import synthetic as syn
from tests import helpers as hp

pytestmark = pytest.mark.gpu

FUSED = SimpleNamespace(debug=False, fused_projection=True)
GENERIC = SimpleNamespace(debug=False, fused_projection=False)
N_HEAD, STRANDS, SEG = 100_000, 6061, 99


def _scene(dev, cam="front"):
    from gaussianhaircut_amd.scene.gaussian_model_strands import GaussianModelStrands
    spec = syn.WorkloadSpec("hair_head_200k", 2 * N_HEAD, 1920, 1080, 21, "random", np.log(0.008))
    head = syn.make_model(spec, dev)
    with torch.no_grad():
        head._label[:N_HEAD] = -4.0   # sigmoid < 0.5 -> head
        head._label[N_HEAD:] = 4.0    # hair-labelled free Gaussians are dropped in the strand stage
    head.precompute_head()
    g = torch.Generator().manual_seed(9)
    origins = torch.nn.functional.normalize(torch.randn(STRANDS, 1, 3, generator=g), dim=-1)
    dirs = (torch.randn(STRANDS, SEG, 3, generator=g) * 0.003 +
            torch.nn.functional.normalize(torch.randn(STRANDS, 1, 3, generator=g), dim=-1) * 0.01)
    # multiples of 2^-16: the polyline's running sums (torch.cumsum associates differently on the CPU and on the GPU)
    # are then exact in fp32 on both sides, so the two chains see bit-identical strand Gaussian centres
    origins = torch.round(origins * 65536.0) / 65536.0
    dirs = torch.round(dirs * 65536.0) / 65536.0
    feats = torch.randn(STRANDS * SEG, 16, 3, generator=g) * 0.1
    feats[:, 0] += 0.4
    conf = 0.3 * torch.randn(STRANDS * SEG, 1, generator=g)
    hair = GaussianModelStrands(3).create_from_strands(origins.to(dev), dirs.to(dev), feats.to(dev),
                                                       orient_conf_log=conf.to(dev))
    hair.active_sh_degree = 3
    hair.initialize_gaussians_hair()
    return spec, head, hair, syn.make_view(spec, dev, cam)


def _rects(xy, rad, W, H):
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def tile(v, g):
        return np.clip(np.trunc(v / np.float32(16)).astype(np.int64), 0, g)
    r = rad.astype(np.float32)
    out = np.stack([tile(xy[:, 0] - r, gx), tile(xy[:, 1] - r, gy),      # auxiliary.h:46-56, same operation order
                    tile(xy[:, 0] + r + np.float32(16) - np.float32(1), gx),
                    tile(xy[:, 1] + r + np.float32(16) - np.float32(1), gy)], axis=1)
    out[rad == 0] = 0
    return out


@pytest.mark.parametrize("cam", ["front", "ring13roll"])   # scene/cameras.py:parity_camera (R = I / a rolled ring view)
def test_render_hair_fused_vs_oracle_chain_at_strand_stage_size(oracle_mod, cam):
    from gaussianhaircut_amd.gaussian_renderer import render_hair
    from tests import oracle_backend as ob
    from tests.gpu_helpers import inspect_fused
    from tests.test_gpu_fused_fullsize import _assert_rows
    dev = torch.device("cuda:0")
    spec, head_c, hair_c, cam_c = _scene("cpu", cam)
    _, head_g, hair_g, cam_g = _scene(dev, cam)
    W, H = spec.W, spec.H
    n_head, n_hair = int(head_c.mask_precomp.sum()), hair_c.get_xyz.shape[0]
    assert n_head == N_HEAD and n_hair == STRANDS * SEG
    row0 = (n_head + 255) // 256 * 256
    rows = row0 + n_hair

    with ob.oracle_rasterizer():
        pc = render_hair(cam_c, head_c, hair_c, GENERIC, syn.background("cpu"))
    st = ob.LAST["state"]
    keep = torch.cat([head_c.filter_points(cam_c)[head_c.mask_precomp], hair_c.filter_points(cam_c)]).numpy()
    idx = np.nonzero(keep)[0]                                   # oracle row j -> Gaussian idx[j] of [head | hair]
    to_ws = lambda i: np.where(i < n_head, i, i - n_head + row0)  # noqa: E731  [head | hair] index -> workspace row

    pg = render_hair(cam_g, head_g, hair_g, FUSED, syn.background(dev))
    torch.cuda.synchronize()
    from gaussianhaircut_amd.diff_gaussian_rasterization import LAST_STATS
    R = int(LAST_STATS["num_rendered"])
    assert R > 1_000_000, R
    ins = inspect_fused(pg.renders_packed, rows, W, H, R)

    # ---- K1 state of both segments; padding rows culled
    # (round 6: the per-row arrays the caller sees -- radii, NDC means -- are written WITHOUT the padding between the segments,
    # through displaced base pointers, include/ghr.h; the workspace rows themselves keep it: the padding rects are culled)
    radii_ws = ins["radii"]
    assert radii_ws.shape[0] == rows and (ins["rects"][n_head:row0, :2] == 0).all(), "padding rows not culled"
    P = n_head + n_hair
    radii_g = radii_ws[:P]
    assert np.array_equal(radii_g, pg["radii"].cpu().numpy())
    assert (radii_g[n_head:] > 0).mean() > 0.3, "strand radii missing from the compact array"
    radii_c = pc["radii"].numpy()
    rect_c = np.zeros((P, 4), np.int64)
    rect_c[idx] = _rects(st.xy.astype(np.float32), st.radii.astype(np.int64), W, H)
    rg = np.concatenate([ins["rects"][:n_head], ins["rects"][row0:]])
    rect_g = np.stack([rg[:, 0] & 0xffff, rg[:, 1] & 0xffff, rg[:, 0] >> 16, rg[:, 1] >> 16], axis=1).astype(np.int64)
    rect_g[radii_g == 0] = 0
    flipped = (radii_c != radii_g) | (rect_c != rect_g).any(axis=1)
    n_flip = int(flipped.sum())
    assert n_flip <= max(2, int(1e-4 * P)), "K1 decisions differ for %d of %d Gaussians" % (n_flip, P)
    same = (radii_c > 0) & ~flipped
    pos = np.full(P, -1, np.int64)
    pos[idx] = np.arange(idx.size)
    j = pos[same]
    assert (j >= 0).all()
    ws = to_ws(np.nonzero(same)[0])
    for seg, sel in (("head", np.nonzero(same)[0] < n_head), ("hair", np.nonzero(same)[0] >= n_head)):
        assert sel.any(), seg
        np.testing.assert_array_equal(ins["depths"][ws[sel]].view(np.uint32), st.depths[j[sel]].view(np.uint32), seg)
        np.testing.assert_array_equal(ins["rec"][ws[sel], 0:2].view(np.uint32), st.xy[j[sel]].view(np.uint32), seg)
        co_g, co_c = ins["rec"][ws[sel], 2:6], st.conic_opacity[j[sel]]
        rel = np.abs(co_g - co_c) / (np.abs(co_c).max(axis=1, keepdims=True) + 1e-30)
        # (hair: scaling = |dir| / 2 and the parallel-transport quaternion are computed by torch on the CPU resp. the GPU
        # before either projection sees them, and a needle's conic amplifies their last-bit differences: measured 5.9e-6)
        assert rel.max() < (1e-5 if seg == "hair" else 5e-6), (seg, rel.max())

    # ---- sorted tile lists (workspace rows on our side, [head | hair] kept indices on the oracle's)
    pl_c = to_ws(idx[st.point_list.astype(np.int64)])
    pl_g = ins["point_list"].astype(np.int64)
    fl_ws = np.zeros(rows, bool)
    fl_ws[to_ws(np.nonzero(flipped)[0])] = True
    np.testing.assert_array_equal(pl_g[~fl_ws[pl_g]], pl_c[~fl_ws[pl_c]])
    if n_flip == 0:
        assert R == st.num_rendered
        ts = ins["tile_start"]
        r = np.stack([ts[:-1], ts[1:]], axis=1).astype(np.uint32)
        r[ts[:-1] == ts[1:]] = 0
        np.testing.assert_array_equal(r, st.ranges)

    # ---- pixels left out, by name
    mask = st.fragile.reshape(H, W).astype(bool).copy()
    assert mask.mean() < 5e-3, mask.mean()
    for g_ in np.nonzero(flipped)[0]:
        for rc in (rect_c[g_], rect_g[g_]):
            mask[16 * rc[1]:16 * rc[3], 16 * rc[0]:16 * rc[2]] = True
    assert mask.mean() < 1e-2, mask.mean()
    ok = ~mask.reshape(-1)
    if n_flip == 0:
        np.testing.assert_array_equal(ins["n_contrib"][ok], st.n_contrib[ok])
    img_c = pc.renders_packed.detach().numpy().reshape(10, -1)[:, ok]
    img_g = pg.renders_packed.detach().cpu().numpy().reshape(10, -1)[:, ok]
    # The 2D-direction channels (5..7) composite SIGNED features of magnitude up to focal / z ~ 370: a pixel under a few
    # hundred strands is a small sum of large terms, and 1e-4 of the RESULT is not a meaningful bar for two fp32 chains
    # whose per-Gaussian inputs already differ in the last bits.  The bar is 1e-4 of the composited magnitude
    # M = sum_i w_i |f_i| (the oracle's own weights, features replaced by their absolute values; = the result itself for
    # the non-negative channels), never below the image criterion of tests/helpers.py.
    M, _, _, _ = oracle_mod.render_forward(st.ranges, st.point_list, st.xy, np.abs(ob.LAST["colors"]), st.conic_opacity,
                                           np.zeros(10, np.float32), H, W)
    M = M.reshape(10, -1)[:, ok]
    close = np.abs(img_g - img_c) <= 1e-4 * np.maximum(np.maximum(1.0, np.abs(img_c)), M)
    assert hp.image_close(img_g[[0, 1, 2, 3, 4, 8, 9]], img_c[[0, 1, 2, 3, 4, 8, 9]]).all()  # plain criterion elsewhere
    if not close.all():
        ch, px = np.nonzero(~close)
        worst = np.argmax(np.abs(img_g - img_c)[ch, px])
        pxs = np.nonzero(ok)[0][px]
        raise AssertionError("%d px-channels off (per channel %s), max err %g; worst: channel %d pixel %d got %g ref %g "
                             "n_contrib %d final_T %g" % ((~close).sum(), np.bincount(ch, minlength=10).tolist(),
                                                          np.abs(img_g - img_c).max(), ch[worst], pxs[worst],
                                                          img_g[ch[worst], px[worst]], img_c[ch[worst], px[worst]],
                                                          st.n_contrib[pxs[worst]], st.final_T[pxs[worst]]))
    vs_c, vs_g = pc["viewspace_points"].detach().numpy(), pg["viewspace_points"].detach().cpu().numpy()
    assert np.abs(vs_g[keep][:, :2] - vs_c[keep][:, :2]).max() < 1e-5

    # ---- gradients to the strand parameters (through initialize_gaussians_hair) for a seeded dL/dout
    w = syn.grad_image(spec, 101) * (H * W)
    w[:, torch.from_numpy(mask)] = 0.0
    (pc.renders_packed * w).sum().backward()
    (pg.renders_packed * w.to(dev)).sum().backward()
    for n in ("_dirs", "_features_dc", "_features_rest", "_orient_conf"):
        a, b = getattr(hair_g, n).grad.cpu().numpy(), getattr(hair_c, n).grad.numpy()
        a, b = a.reshape(-1, a.shape[-1]) if n == "_dirs" else a, b.reshape(-1, b.shape[-1]) if n == "_dirs" else b
        assert np.abs(b).max() > 0, n
        _assert_rows("hair " + n, a, b)
    # the densification signal of the strand rows; the frozen head's rows stay exactly 0 on the fused path
    vg_g, vg_c = pg["viewspace_points"].grad.cpu().numpy(), pc["viewspace_points"].grad.numpy()
    assert np.abs(vg_g[:n_head]).max() == 0
    _assert_rows("hair viewspace", vg_g[n_head:, :2], vg_c[n_head:, :2])
    print("hair fullsize: rows %d R %d flips %d masked %.4f img err %.2e" %
          (rows, R, n_flip, mask.mean(), np.abs(img_g - img_c).max()), flush=True)
