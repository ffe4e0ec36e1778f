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


def test_scene_gt_info_matches_the_bop_toolkit_golden():
    """bop_pose.gt_info_from_masks against entries computed with the reference's own bop_toolkit_lib.misc.calc_2d_bbox and
    the counting rules of its scripts/calc_gt_info.py:139-172 (tests/golden/make_golden_from_reference.py gt_info): random
    silhouettes and visible masks incl. absent and fully occluded objects, a depth image with holes."""
    import torch
    from pegasus_amd import bop_pose
    g = np.load(GOLD / "bop_gt_info.npz")
    n, H, W = (int(v) for v in g["shape"])
    unpack = lambda a: np.unpackbits(a, axis=-1)[..., :W].astype(bool)
    sil, vis, valid = unpack(g["sil"]), unpack(g["vis"]), unpack(g["depth_valid"])
    # as n images of one object each, and as one batch of 2 images x n/2 objects sharing the per-image depth mask
    one = bop_pose.gt_info_from_masks(torch.from_numpy(vis[:, None]), torch.from_numpy(sil[:, None]), torch.from_numpy(valid))
    for k in ("px_count_all", "px_count_valid", "px_count_visib", "bbox_obj", "bbox_visib"):
        np.testing.assert_array_equal(one[k][:, 0], g[k], err_msg=k)
    np.testing.assert_allclose(one["visib_fract"][:, 0], g["visib_fract"], rtol=0, atol=1e-15)
    e = bop_pose.scene_gt_info_entry(one, 3)
    assert e == [{"px_count_all": int(g["px_count_all"][3]), "px_count_valid": int(g["px_count_valid"][3]),
                  "px_count_visib": int(g["px_count_visib"][3]), "visib_fract": float(g["visib_fract"][3]),
                  "bbox_obj": g["bbox_obj"][3].tolist(), "bbox_visib": g["bbox_visib"][3].tolist()}]
    assert g["bbox_visib"][4].tolist() == [-1, -1, -1, -1] and one["bbox_obj"][5, 0].tolist() == [-1, -1, -1, -1]
    numpy_in = bop_pose.gt_info_from_masks(vis[:, None], sil[:, None])             # numpy inputs, all depth valid
    np.testing.assert_array_equal(numpy_in["px_count_valid"][:, 0], g["px_count_all"])


def test_projected_points_match_the_bop_toolkit_golden():
    """bop_pose.project_points (scene_gt's projected_points / projected_center: P = K T[:3] on homogeneous points, the
    reference's writer at /root/reference/src/tools/pegasus_working.py:547-563) against bop_toolkit_lib.misc.project_pts."""
    from pegasus_amd import bop_pose
    g = np.load(GOLD / "bop_gt_info.npz")
    for K, R, t, pts, uv in zip(g["proj_K"], g["proj_R"], g["proj_t"], g["proj_pts"], g["proj_uv"]):
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        np.testing.assert_allclose(bop_pose.project_points(K, T, pts), uv, rtol=0, atol=1e-10)
