"""Pins for the oracle (the reference has no tests or golden vectors): closed-form micro-cases."""
import math

import numpy as np
import torch

from gaussianhaircut_amd.scene.cameras import make_camera
from gaussianhaircut_amd.utils import synthetic as syn


def _cam(W, H):
    cam = make_camera(W, H)
    return cam, math.tan(float(cam.FoVx) / 2), math.tan(float(cam.FoVy) / 2)


def _fwd(oracle, W, H, xyz, scales, opac, colors, bg=None, rot=None):
    cam, tx, ty = _cam(W, H)
    P = xyz.shape[0]
    rot = rot if rot is not None else np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1))
    bg = np.zeros(10, np.float32) if bg is None else bg
    return oracle.rasterize_forward(bg, xyz, colors, opac, cam.world_view_transform.numpy(),
                                    cam.full_proj_transform.numpy(), tx, ty, H, W, scales=scales, rotations=rot), cam


def test_single_isotropic_gaussian_closed_form(oracle_mod):
    W = H = 65
    o, s = 0.8, 0.1
    (out, radii, st), cam = _fwd(oracle_mod, W, H, np.zeros((1, 3), np.float32), np.full((1, 3), s, np.float32),
                                 np.array([o], np.float32), np.ones((1, 10), np.float32))
    focal = H / (2 * math.tan(float(cam.FoVy) / 2))
    var = (s * focal / 4.0) ** 2 + 0.3
    assert st.xy[0].tolist() == [32.0, 32.0]  # ndc2Pix(0, 65) = ((0+1)*65-1)/2
    np.testing.assert_allclose(st.conic_opacity[0], [1 / var, 0, 1 / var, o], rtol=2e-6, atol=1e-7)
    assert radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    assert st.depths[0] == 4.0
    ys, xs = np.mgrid[0:H, 0:W]
    alpha = np.minimum(0.99, o * np.exp(-0.5 * ((xs - 32.0) ** 2 + (ys - 32.0) ** 2) / var))
    alpha[alpha < 1 / 255] = 0
    inrect = (np.abs(xs - 32) <= radii[0]) & (np.abs(ys - 32) <= radii[0])
    assert np.abs(out[0][inrect] - alpha[inrect]).max() < 1e-6
    assert np.abs(st.final_T.reshape(H, W)[inrect] - (1 - alpha[inrect])).max() < 1e-6


def test_two_gaussians_front_to_back_and_background(oracle_mod):
    W = H = 33
    xyz = np.array([[0, 0, 1.0], [0, 0, 0.0]], np.float32)  # index 0 is FARTHER (view z 5) than index 1 (view z 4)
    colors = np.zeros((2, 10), np.float32)
    colors[0, 0], colors[1, 0] = 1.0, 0.25
    bg = np.full(10, 0.5, np.float32)
    (out, radii, st), cam = _fwd(oracle_mod, W, H, xyz, np.full((2, 3), 0.3, np.float32),
                                 np.array([0.9, 0.6], np.float32), colors, bg=bg)
    tile0 = st.point_list[st.ranges[0, 0]:st.ranges[0, 1]]
    assert tile0.tolist() == [1, 0]  # sorted by depth, nearer first
    c = 16  # centre pixel
    a_near = 0.6 * math.exp(0.0)
    a_far = 0.9 * math.exp(0.0)
    expect = 0.25 * a_near + 1.0 * a_far * (1 - a_near) + (1 - a_near) * (1 - a_far) * 0.5
    assert abs(out[0, c, c] - expect) < 1e-6
    assert st.n_contrib.reshape(H, W)[c, c] == 2


def test_saturation_stops_blending_and_n_contrib(oracle_mod):
    """T(1-alpha) < 1e-4 => the entry is NOT blended and the pixel is done (forward.cu:372-377)."""
    W = H = 17
    P = 6
    xyz = np.zeros((P, 3), np.float32)
    xyz[:, 2] = np.arange(P) * 0.1
    colors = np.zeros((P, 10), np.float32)
    colors[:, 0] = 1
    (out, radii, st), cam = _fwd(oracle_mod, W, H, xyz, np.full((P, 3), 0.5, np.float32),
                                 np.full(P, 0.95, np.float32), colors)
    # alpha = .95 at the centre: T = .05, .0025, 1.25e-4; the FOURTH would give 6.25e-6 < 1e-4 -> not blended, done
    assert st.n_contrib.reshape(H, W)[8, 8] == 3
    assert abs(st.final_T.reshape(H, W)[8, 8] - 1.25e-4) < 1e-9
    assert abs(out[0, 8, 8] - 0.95 * (1 + 0.05 + 0.0025)) < 1e-6


def test_near_plane_cull_and_offscreen(oracle_mod):
    W, H = 64, 48
    xyz = np.array([[0, 0, -3.875], [0, 0, -3.75], [50, 0, 0], [0, 0, -10]], np.float32)
    (out, radii, st), cam = _fwd(oracle_mod, W, H, xyz, np.full((4, 3), 0.01, np.float32),
                                 np.full(4, 0.5, np.float32), np.ones((4, 10), np.float32))
    assert radii[0] == 0 and radii[1] > 0 and radii[2] == 0 and radii[3] == 0  # z<=0.2, ok, empty rect, behind
    assert st.tiles_touched[0] == 0 and st.tiles_touched[2] == 0
    vis = oracle_mod.mark_visible(xyz, cam.world_view_transform.numpy(), cam.full_proj_transform.numpy())
    assert vis.tolist() == [False, True, True, False]


def test_key_order_ties_by_index_and_ranges(oracle_mod):
    W = H = 32
    P = 40
    xyz = np.zeros((P, 3), np.float32)
    xyz[:, 2] = (np.arange(P) % 3) * 0.5  # 3 depths, many ties
    (out, radii, st), cam = _fwd(oracle_mod, W, H, xyz, np.full((P, 3), 0.05, np.float32),
                                 np.full(P, 0.1, np.float32), np.ones((P, 10), np.float32))
    for t in range(st.ranges.shape[0]):
        ids = st.point_list[st.ranges[t, 0]:st.ranges[t, 1]]
        d = st.depths[ids]
        assert (np.diff(d) >= 0).all()
        for dv in np.unique(d):
            assert (np.diff(ids[d == dv]) > 0).all()  # stable: equal depth keeps ascending Gaussian index
    assert st.ranges[:, 1].max() == st.num_rendered


def test_filter_points_agrees_with_oracle_cull():
    """The reference's own Python restatement of K1's cull (gaussian_model.py:143-228), mirrored in our GaussianModel,
    selects exactly the Gaussians the oracle keeps (radii > 0)."""
    import oracle
    for cfg in ("tiny", "ragged", "tiny_strands"):
        spec = syn.CONFIGS[cfg]
        model = syn.make_model(spec)
        cam = syn.make_view(spec)
        with torch.no_grad():
            conic = model.get_conic(cam)
            model.get_mean_2d(cam)
            keep = model.filter_points(cam).numpy()
        d, radii, xy, co, cov, tiles = oracle.preprocess(model.get_xyz.detach().numpy(),
                                                         model.get_opacity.detach().numpy(),
                                                         cam.world_view_transform.numpy(),
                                                         cam.full_proj_transform.numpy(),
                                                         math.tan(float(cam.FoVx) / 2), math.tan(float(cam.FoVy) / 2),
                                                         spec.H, spec.W, conic_precomp=conic.numpy())
        assert (keep == (radii > 0)).mean() > 0.999  # differently-rounded maths may flip a borderline Gaussian
