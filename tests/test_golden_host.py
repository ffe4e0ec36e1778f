"""Host-side maths against golden vectors captured from the parts of the reference that import
(tests/golden/make_golden_from_reference.py)."""
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


def test_hemisphere_views_match_bop_sampler():
    from pegasus_amd import graphics as G
    g = np.load(GOLD / "bop_fibonacci_views.npz")
    for n in (128, 1024, 33):
        views = G.hemisphere_views(n, float(g[f"radius_{n}"]))
        R = np.stack([v[0] for v in views]); t = np.stack([v[1] for v in views])
        assert R.shape == g[f"R_{n}"].shape
        np.testing.assert_allclose(R, g[f"R_{n}"], atol=1e-12)
        np.testing.assert_allclose(t, g[f"t_{n}"], atol=1e-12)
    assert len(G.hemisphere_views(128, 0.45)) == 65


def test_camera_matrices_conventions():
    from pegasus_amd import graphics as G
    rng = np.random.default_rng(1)
    from scipy.spatial.transform import Rotation as Rot
    R_c2w = Rot.random(random_state=3).as_matrix()
    t = rng.normal(size=3)
    wvt, proj, full, center = G.camera_matrices(R_c2w, t, 0.9, 0.7)
    W2C = wvt.T
    np.testing.assert_allclose(W2C[:3, :3], R_c2w.T, atol=1e-6)           # T_w2c[:3,:3] = cam.R.T (pegasus_working.py:464-466)
    np.testing.assert_allclose(W2C[:3, 3], t, atol=1e-6)
    np.testing.assert_allclose(center, -R_c2w @ t, atol=1e-5)             # camera centre in world coordinates
    P = proj.T
    assert P[3, 2] == 1 and P[0, 0] == np.float32(1 / np.tan(0.45)) and P[1, 1] == np.float32(1 / np.tan(0.35))
    np.testing.assert_allclose(full, wvt @ proj, atol=1e-6)
    assert G.focal2fov(G.fov2focal(0.8, 640), 640) == np.float64(0.8).item() or abs(G.focal2fov(G.fov2focal(0.8, 640), 640) - 0.8) < 1e-12


def test_semantic_colour_table_matches_reference():
    from pegasus_amd.masks import generate_colors
    g = np.load(GOLD / "graphic_utils.npz")
    for n in (1, 3, 6, 8, 21):
        np.testing.assert_array_equal(generate_colors(n), g[f"colors_{n}"])


def test_pose_sequence_fixture_schema():
    tq = np.load(GOLD / "simulation_steps_body1_first200.npz")["t_q_xyzw"]
    assert tq.shape == (200, 7)
    np.testing.assert_allclose(np.linalg.norm(tq[:, 3:], axis=1), 1.0, atol=1e-6)
