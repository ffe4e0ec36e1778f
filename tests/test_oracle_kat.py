"""Analytic known-answer tests for the CPU oracle.  The reference holds no golden
vectors for the rasterizer, so these closed-form cases are what pins the restatement."""
import math

import numpy as np
import pytest

C0 = 0.28209479177387814
W = H = 64
FOV = math.radians(60.0)
TAN = math.tan(FOV / 2)
FOCAL = W / (2 * TAN)


def camera():
    """Camera at the origin looking down +z, identity view matrix."""
    from pegasus_amd import graphics as G
    wvt, proj, full, center = G.camera_matrices(np.eye(3), np.zeros(3), FOV, FOV)
    return dict(width=W, height=H, tanfovx=TAN, tanfovy=TAN, viewmatrix=wvt, projmatrix=full, campos=center)


def splat(x, y, z, sigma_px, opacity, rgb):
    """One isotropic Gaussian whose screen-space sigma is sigma_px at depth z, colour via SH degree 0."""
    s = sigma_px * z / FOCAL
    dc = (np.asarray(rgb, np.float64) - 0.5) / C0
    return dict(mean=(x, y, z), scale=(s, s, s), rot=(1, 0, 0, 0), op=opacity, dc=dc)


def run(oracle, splats, bg=(0, 0, 0), **kw):
    n = len(splats)
    shs = np.zeros((n, 16, 3), np.float32)
    shs[:, 0] = [s["dc"] for s in splats]
    cam = camera()
    cam.update(kw)
    return oracle.forward(
        np.array([s["mean"] for s in splats], np.float32), np.array([s["op"] for s in splats], np.float32),
        scales=np.array([s["scale"] for s in splats], np.float32),
        rotations=np.array([s["rot"] for s in splats], np.float32), shs=shs, sh_degree=0,
        bg=np.asarray(bg, np.float32), **cam)


def pixel_of(x, y, z):
    """Pixel coordinate the point projects to: ((ndc+1)*S-1)/2 with ndc = x/(z*tan)."""
    return ((x / (z * TAN) + 1) * W - 1) / 2, ((y / (z * TAN) + 1) * H - 1) / 2


def world_of(px, py, z):
    return ((2 * px + 1) / W - 1) * z * TAN, ((2 * py + 1) / H - 1) * z * TAN


def test_single_gaussian_centre_pixel(oracle):
    x, y = world_of(20, 30, 2.0)
    r = run(oracle, [splat(x, y, 2.0, 3.0, 0.6, (0.9, 0.5, 0.1))], bg=(0.2, 0.2, 0.2))
    assert r["xy"][0] == pytest.approx([20, 30], abs=1e-3)
    a = 0.6                                                  # exp(0) = 1 at the centre
    np.testing.assert_allclose(r["color"][:, 30, 20], np.array([0.9, 0.5, 0.1]) * a + 0.2 * (1 - a), atol=2e-6)
    assert r["out_depth"][0, 30, 20] == pytest.approx(2.0 * a, abs=1e-6)     # sum T alpha z, no normalisation
    assert r["final_T"][30, 20] == pytest.approx(1 - a, abs=1e-6)
    assert r["n_contrib"][30, 20] == 1
    # a far corner sees only background
    np.testing.assert_allclose(r["color"][:, 0, 63], 0.2, atol=1e-7)
    assert r["out_depth"][0, 0, 63] == 0 and r["final_T"][0, 63] == 1


def test_gaussian_falloff_and_radius(oracle):
    sigma = 2.5
    x, y = world_of(32, 32, 1.5)
    r = run(oracle, [splat(x, y, 1.5, sigma, 0.9, (1, 1, 1))])
    var = sigma * sigma + 0.3                                  # low-pass dilation
    for d in (1, 2, 4):
        expect = 0.9 * math.exp(-0.5 * d * d / var)
        assert r["color"][0, 32, 32 + d] == pytest.approx(expect, rel=2e-4)
        assert r["color"][0, 32 - d, 32] == pytest.approx(expect, rel=2e-4)
    assert r["radii"][0] == math.ceil(3 * math.sqrt(var))
    cx, cy, cz, op = r["conic_opacity"][0]
    assert (cx, cz) == pytest.approx((1 / var, 1 / var), rel=1e-4) and abs(cy) < 1e-4 and op == np.float32(0.9)


def test_two_gaussians_depth_order(oracle):
    xa, ya = world_of(10, 10, 1.0)
    xb, yb = world_of(10, 10, 3.0)
    near = splat(xa, ya, 1.0, 2.0, 0.5, (1, 0, 0))
    far = splat(xb, yb, 3.0, 2.0, 0.8, (0, 0, 1))
    for order in ([near, far], [far, near]):                   # input order must not matter
        r = run(oracle, order)
        c = r["color"][:, 10, 10]
        np.testing.assert_allclose(c, [0.5, 0, 0.8 * 0.5], atol=2e-6)       # c1 a1 + c2 a2 (1 - a1)
        assert r["out_depth"][0, 10, 10] == pytest.approx(1.0 * 0.5 + 3.0 * 0.8 * 0.5, abs=2e-6)
        assert r["final_T"][10, 10] == pytest.approx(0.5 * 0.2, abs=1e-6)


def test_equal_depth_resolved_by_index(oracle):
    x, y = world_of(40, 40, 2.0)
    a = splat(x, y, 2.0, 2.0, 0.5, (1, 0, 0))
    b = splat(x, y, 2.0, 2.0, 0.5, (0, 1, 0))
    r = run(oracle, [a, b])
    np.testing.assert_allclose(r["color"][:, 40, 40], [0.5, 0.25, 0], atol=2e-6)   # index 0 is in front
    r = run(oracle, [b, a])
    np.testing.assert_allclose(r["color"][:, 40, 40], [0.25, 0.5, 0], atol=2e-6)


def test_alpha_thresholds(oracle):
    x, y = world_of(5, 5, 1.0)
    # alpha below 1/255 everywhere: the pixel is untouched
    r = run(oracle, [splat(x, y, 1.0, 2.0, 0.0039, (1, 1, 1))], bg=(0.3, 0.3, 0.3))
    assert r["n_contrib"][5, 5] == 0 and r["color"][0, 5, 5] == np.float32(0.3)
    r = run(oracle, [splat(x, y, 1.0, 2.0, 0.004, (1, 1, 1))], bg=(0.3, 0.3, 0.3))
    assert r["n_contrib"][5, 5] == 1
    # opacity 1.0 is clamped to alpha 0.99
    r = run(oracle, [splat(x, y, 1.0, 2.0, 1.0, (1, 1, 1))])
    assert r["color"][0, 5, 5] == pytest.approx(0.99, abs=1e-7) and r["final_T"][5, 5] == pytest.approx(0.01, abs=1e-7)


def test_near_plane_cull(oracle):
    r = run(oracle, [splat(0, 0, 0.2, 2.0, 0.9, (1, 1, 1)), splat(0, 0, 0.2001, 2.0, 0.9, (1, 1, 1))])
    assert r["radii"][0] == 0 and r["tiles_touched"][0] == 0        # z <= 0.2 culled
    assert r["radii"][1] > 0


def test_tile_rectangle_on_corner(oracle):
    """A splat centred exactly between four tiles with a radius < 16 touches exactly those 4 tiles."""
    x, y = world_of(15.5, 15.5, 2.0)
    r = run(oracle, [splat(x, y, 2.0, 1.0, 0.9, (1, 1, 1))])
    assert r["radii"][0] == math.ceil(3 * math.sqrt(1.3)) == 4
    assert r["tiles_touched"][0] == 4
    assert sorted((r["keys_sorted"] >> np.uint64(32)).tolist()) == [0, 1, 4, 5]       # 64/16 = 4 tiles per row


def test_offscreen_splat_has_zero_radius(oracle):
    x, y = world_of(500, 10, 2.0)
    r = run(oracle, [splat(x, y, 2.0, 2.0, 0.9, (1, 1, 1))])
    assert r["radii"][0] == 0 and r["num_instances"] == 0


def test_sh_degree0_colour_and_clamp(oracle):
    x, y = world_of(8, 8, 1.0)
    r = run(oracle, [splat(x, y, 1.0, 2.0, 0.5, (0.25, 0.75, 0.5))])
    np.testing.assert_allclose(r["rgb"][0], [0.25, 0.75, 0.5], atol=1e-6)       # C0*dc + 0.5
    s = splat(x, y, 1.0, 2.0, 0.5, (0, 0, 0))
    s["dc"] = np.array([-5.0, 0.0, 5.0])
    r = run(oracle, [s])
    np.testing.assert_allclose(r["rgb"][0], [0.0, 0.5, 0.5 + 5 * C0], atol=1e-6)  # clamped at 0 from below only


def test_sh_degree1_direction(oracle):
    """Band 1: colour = C0 dc + 0.5 - C1 y sh1 + C1 z sh2 - C1 x sh3 with (x,y,z) the unit view direction."""
    C1 = 0.4886025119029199
    cam = camera()
    pos = np.array([[0.3, -0.2, 2.0]], np.float32)
    d = pos[0] / np.linalg.norm(pos[0])
    shs = np.zeros((1, 16, 3), np.float32)
    shs[0, 1] = [1, 0, 0]; shs[0, 2] = [0, 1, 0]; shs[0, 3] = [0, 0, 1]
    r = oracle.forward(pos, np.array([0.5], np.float32), scales=np.full((1, 3), 0.01, np.float32),
                       rotations=np.array([[1, 0, 0, 0]], np.float32), shs=shs, sh_degree=1,
                       bg=np.zeros(3, np.float32), stage="preprocess", **cam)
    np.testing.assert_allclose(r["rgb"][0], [0.5 - C1 * d[1], 0.5 + C1 * d[2], 0.5 - C1 * d[0]], atol=1e-6)


def test_early_termination_keeps_T_above_threshold(oracle):
    x, y = world_of(12, 12, 1.0)
    layers = [splat(*world_of(12, 12, 1.0 + 0.01 * k), 1.0 + 0.01 * k, 2.0, 0.9, (1, 1, 1)) for k in range(100)]
    r = run(oracle, layers)
    # T after k blends is 0.1^k; the 5th would give 1e-5 < 1e-4 and is NOT blended
    assert r["n_contrib"][12, 12] == 4
    assert r["final_T"][12, 12] == pytest.approx(1e-4, rel=1e-4) and r["final_T"][12, 12] >= 1e-4 * (1 - 1e-4)


def test_empty_scene_stays_zero(oracle):
    cam = camera()
    r = oracle.forward(np.zeros((0, 3), np.float32), np.zeros(0, np.float32), scales=np.zeros((0, 3), np.float32),
                       rotations=np.zeros((0, 4), np.float32), shs=np.zeros((0, 16, 3), np.float32), sh_degree=3,
                       bg=np.ones(3, np.float32), **cam)
    assert not r["color"].any() and not r["out_depth"].any()      # not the background: zeros


def test_anisotropic_rotated_covariance(oracle):
    """cov3D = R S^2 R^T and its EWA projection, against numpy float64."""
    cam = camera()
    q = np.array([0.8, 0.2, -0.4, 0.4]); q /= np.linalg.norm(q)
    s = np.array([0.05, 0.01, 0.03])
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    S3 = R @ np.diag(s * s) @ R.T
    pos = np.array([0.2, 0.1, 2.5])
    J = np.array([[FOCAL / pos[2], 0, -FOCAL * pos[0] / pos[2] ** 2], [0, FOCAL / pos[2], -FOCAL * pos[1] / pos[2] ** 2]])
    S2 = J @ S3 @ J.T + 0.3 * np.eye(2)
    conic = np.linalg.inv(S2)
    r = oracle.forward(pos[None].astype(np.float32), np.array([0.7], np.float32), scales=s[None].astype(np.float32),
                       rotations=q[None].astype(np.float32), shs=np.zeros((1, 16, 3), np.float32), sh_degree=0,
                       bg=np.zeros(3, np.float32), stage="preprocess", **cam)
    np.testing.assert_allclose(r["cov3d"][0], S3[np.triu_indices(3)], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(r["conic_opacity"][0, :3], [conic[0, 0], conic[0, 1], conic[1, 1]], rtol=2e-4)
    lam = np.linalg.eigvalsh(S2).max()
    assert r["radii"][0] == math.ceil(3 * math.sqrt(lam))


def test_masks_and_quantisation_match_numpy(oracle):
    """The reference's host ops: np.linalg.norm(img - c, axis=2) <= 0.1 (src/gs/render.py:60-63,89-93) and
    (img*255).astype(uint8), (depth*1000).astype(uint16) (pegasus.py:347,355)."""
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1.2, size=(3, 40, 50)).astype(np.float32)
    colors = rng.uniform(0, 1, size=(4, 3)).astype(np.float32)
    img[:, :10, :10] = colors[2][:, None, None] + 0.03
    depth = rng.uniform(0, 60, size=(40, 50)).astype(np.float32)
    m = oracle.color_masks(img, colors, 0.1)
    hwc = img.transpose(1, 2, 0)
    for k in range(4):
        np.testing.assert_array_equal(m[k].astype(bool), np.linalg.norm(hwc - colors[k], axis=2) <= 0.1)
    rgb8, mm = oracle.quantize(img, depth)
    np.testing.assert_array_equal(rgb8, (hwc * 255).astype(np.uint8))
    np.testing.assert_array_equal(mm, (depth * 1000).astype(np.uint16))


def test_tie_index_breaks_exact_depth_ties(oracle):
    """Two coincident-depth Gaussians overlapping a pixel: the smaller tie_index is blended first, whatever the
    storage order (PgrScene.tie_index / PgrOracleIn.tie_index)."""
    import numpy as np
    from pegasus_amd import scenes
    _, views = scenes.scene_c1(n=16)
    v = views[0]
    mk = lambda order: dict(
        means3d=np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]], np.float32)[order],
        opacities=np.array([0.6, 0.6], np.float32)[order],
        scales=np.full((2, 3), 0.05, np.float32), rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (2, 1)),
        colors_precomp=np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], np.float32)[order])
    a = oracle.forward(**mk([0, 1]), **v.raster_kwargs(), num_threads=1)
    b = oracle.forward(**mk([1, 0]), **v.raster_kwargs(), num_threads=1)                       # stored swapped
    c = oracle.forward(**mk([1, 0]), **v.raster_kwargs(), num_threads=1, tie_index=[1, 0])     # ... with the caller's order
    cy, cx = v.height // 2, v.width // 2
    assert a["color"][0, cy, cx] > a["color"][1, cy, cx] and b["color"][1, cy, cx] > b["color"][0, cy, cx]
    np.testing.assert_array_equal(a["color"], c["color"])
    np.testing.assert_array_equal(c["gauss_sorted"][:4] ^ 1, a["gauss_sorted"][:4])   # positions differ, order by tie index


def test_oracle_agrees_with_independent_dense_float64_renderer(oracle):
    """A second, independently written restatement (oracle/dense_ref.py: numpy float64, no tiles, no lists, every
    Gaussian against every pixel, global depth order) renders the C1-style cube to the same image as the tiled fp32
    oracle -- the tile machinery (rectangles, binning, per-tile sort, early-out) cannot have changed the picture."""
    import numpy as np
    from oracle.dense_ref import dense_forward
    from pegasus_amd import scenes
    cloud, views = scenes.scene_c1(seed=21, n=1500)
    v = scenes.make_view(views[0].R_c2w.T, views[0].t_w2c, 96, 80, fovx=views[0].fovx, fovy=views[0].fovy)
    act = cloud.activated()
    kw = v.raster_kwargs((0.1, 0.3, 0.5))
    o = oracle.forward(**act, sh_degree=3, **kw, num_threads=4)
    c64, d64 = dense_forward(sh_degree=3, **{k: np.asarray(a, np.float64) for k, a in act.items()}, **kw)
    assert np.abs(c64 - o["color"]).max() < 5e-5
    assert np.abs(d64 - o["out_depth"][0]).max() < 5e-5
    assert (o["radii"] > 0).sum() > 500


def test_sh_colour_model_against_first_principles_basis(oracle):
    """The colour model's basis pinned from first principles: oracle/dense_ref.py derives the real spherical harmonics
    from associated Legendre recurrences and factorial normalisation (Condon-Shortley phase kept -- the 3DGS paper's
    convention), sharing no constant with the product's table (pegasus_amd/sh_utils.py) or the C oracle's.  All three
    must agree: a wrong sign or constant in one of them cannot hide."""
    import numpy as np
    from oracle.dense_ref import real_sh_basis
    from pegasus_amd import graphics as G, scenes
    from pegasus_amd.sh_utils import sh_basis, C0, C1
    rng = np.random.default_rng(5)
    dirs = rng.normal(size=(200, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = np.concatenate([dirs, np.eye(3), -np.eye(3)])
    for deg in range(4):
        mine = np.stack([real_sh_basis(deg, d) for d in dirs])
        assert mine.shape[1] == (deg + 1) ** 2
        np.testing.assert_allclose(mine, sh_basis(deg, dirs), atol=1e-14)
    # closed forms of the first two bands, and orthonormality of the whole basis on the sphere (Monte-Carlo free: Lebedev-
    # like check through the addition theorem  sum_m Y_lm(d)^2 = (2l+1)/(4 pi))
    b = real_sh_basis(3, dirs[0])
    assert abs(b[0] - 0.5 / np.sqrt(np.pi)) < 1e-15 and abs(C0 - 0.5 / np.sqrt(np.pi)) < 1e-15
    np.testing.assert_allclose(b[1:4], np.sqrt(3 / (4 * np.pi)) * np.array([-dirs[0][1], dirs[0][2], -dirs[0][0]]), atol=1e-15)
    assert abs(C1 - np.sqrt(3 / (4 * np.pi))) < 1e-15
    for l, sl in ((0, slice(0, 1)), (1, slice(1, 4)), (2, slice(4, 9)), (3, slice(9, 16))):
        for d in dirs[:20]:
            assert abs((real_sh_basis(3, d)[sl] ** 2).sum() - (2 * l + 1) / (4 * np.pi)) < 1e-13
    # the C oracle's per-Gaussian colour = max(basis . coefficients + 0.5, 0) with THIS basis
    n = 64
    R, t = G.look_at_opencv((0.0, 0.0, -3.0), (0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0))
    v = scenes.make_view(R, t, 64, 64, fovx=1.0, fovy=1.0)
    means = rng.uniform(-0.8, 0.8, size=(n, 3)).astype(np.float32)
    shs = rng.normal(0, 0.4, size=(n, 16, 3)).astype(np.float32)
    o = oracle.forward(means, np.full(n, 0.5, np.float32), scales=np.full((n, 3), 0.05, np.float32),
                       rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)), shs=shs, sh_degree=3,
                       **v.raster_kwargs(), stage="preprocess")
    seen = o["radii"] > 0
    assert seen.sum() > 40
    for i in np.nonzero(seen)[0]:
        d = means[i].astype(np.float64) - v.camera_center.astype(np.float64)
        want = np.maximum(real_sh_basis(3, d / np.linalg.norm(d)) @ shs[i].astype(np.float64) + 0.5, 0.0)
        np.testing.assert_allclose(o["rgb"][i], want, atol=2e-6)


def test_depth_mode_switch_known_answers(oracle):
    """The depth rule is a switch (SURVEY.md section 8a "Depth variant": unverified, "keep it a one-line switch").
    mode 0: sum T alpha z.  mode 1: that sum over 1 - T_final (= the blended weights' total): one splat -> exactly its z;
    two coincident splats at z1 < z2 -> (a1 z1 + a2 (1 - a1) z2) / (a1 + a2 (1 - a1)); untouched pixels -> 0 in both."""
    x, y = world_of(20, 30, 2.0)
    one = [splat(x, y, 2.0, 3.0, 0.6, (0.9, 0.5, 0.1))]
    r0, r1 = run(oracle, one, depth_mode=0), run(oracle, one, depth_mode=1)
    assert r0["out_depth"][0, 30, 20] == pytest.approx(2.0 * 0.6, abs=1e-6)
    assert r1["out_depth"][0, 30, 20] == pytest.approx(2.0, abs=1e-5)
    assert r1["out_depth"][0, 0, 63] == 0.0 and r0["out_depth"][0, 0, 63] == 0.0          # nothing blended there
    np.testing.assert_array_equal(r0["color"], r1["color"])                               # only the depth image differs
    np.testing.assert_array_equal(r0["final_T"], r1["final_T"])
    np.testing.assert_array_equal(r0["n_contrib"], r1["n_contrib"])
    xa, ya = world_of(20, 30, 1.5)
    xb, yb = world_of(20, 30, 3.0)
    two = [splat(xb, yb, 3.0, 3.0, 0.8, (0.1, 0.9, 0.1)), splat(xa, ya, 1.5, 3.0, 0.5, (0.9, 0.1, 0.1))]
    a1, a2 = 0.5, 0.8
    r0, r1 = run(oracle, two, depth_mode=0), run(oracle, two, depth_mode=1)
    d = a1 * 1.5 + a2 * (1 - a1) * 3.0
    assert r0["out_depth"][0, 30, 20] == pytest.approx(d, abs=2e-6)
    assert r1["out_depth"][0, 30, 20] == pytest.approx(d / (a1 + a2 * (1 - a1)), abs=1e-5)
    # everywhere: normalised = un-normalised / (1 - T), exactly as the expression says
    T = r0["final_T"]
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.where(1.0 - T > 0, r0["out_depth"][0] / (np.float32(1.0) - T), np.float32(0.0)).astype(np.float32)
    np.testing.assert_array_equal(r1["out_depth"][0], want)
