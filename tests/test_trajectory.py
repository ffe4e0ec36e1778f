"""Dynamic-sequence poses (BASELINE.json configs[4]): the absolute per-step composition of pegasus_amd/trajectory.py
against the reference's recursion of consecutive deltas (/root/reference/src/gs/pegasus_setup.py:160-193), on the committed
fixture of the reference's own trajectory (tests/golden/simulation_steps_body1_first200.npz)."""
import numpy as np
import pytest

from pegasus_amd import trajectory as TJ


def test_fixture_is_the_recorded_drop():
    traj = TJ.load_fixture()
    assert traj.shape == (200, 7)
    np.testing.assert_allclose(np.linalg.norm(traj[:, 3:], axis=1), 1.0, atol=1e-6)
    assert traj[0, 2] > 0.5 and traj[-1, 2] < 0.07           # body 1 falls half a metre and comes to rest


@pytest.mark.parametrize("step", [0, 1, 17, 60, 199])
def test_absolute_composition_equals_accumulated_deltas(step):
    traj = TJ.load_fixture()
    rng = np.random.default_rng(step)
    pts = rng.normal(0, 0.05, (500, 3)) + np.array([0.3, -0.2, 0.1])
    c = pts.mean(0)
    ref = TJ.accumulate_deltas(traj, step, c, pts)           # the reference's loop, literally
    T = TJ.absolute_pose(traj, step)
    mine = (T[:3, :3] @ (pts - c).T).T + c + T[:3, 3]
    np.testing.assert_allclose(mine, ref, atol=1e-9)


def test_sequence_tables_layout_and_rest():
    from scipy.spatial.transform import Rotation
    traj = TJ.load_fixture()
    centers = [np.array([0.1 * k, -0.05 * k, 0.06]) for k in range(3)]
    tables, motions = TJ.sequence_poses(traj, centers, 200, phase=5)
    assert tables.shape == (200, 3, 20) and tables.dtype == np.float32 and len(motions) == 200
    # the last step is the resting scene: identity for every object
    np.testing.assert_allclose(tables[199, :, 0:9].reshape(3, 3, 3), np.broadcast_to(np.eye(3), (3, 3, 3)), atol=1e-7)
    np.testing.assert_allclose(tables[199, :, 9:12], 0, atol=1e-9)
    for k in range(3):
        np.testing.assert_allclose(tables[0, k, 12:15], centers[k], atol=1e-7)
        # object k runs 5 k steps ahead of object 0
        T = TJ.relative_to_rest(traj, min(199, 10 + 5 * k))
        np.testing.assert_allclose(tables[10, k, 0:9].reshape(3, 3), T[:3, :3], atol=1e-6)
        np.testing.assert_allclose(tables[10, k, 9:12], T[:3, 3], atol=1e-7)
        q = tables[10, k, 15:19]                                                     # (w, x, y, z) of the same rotation
        Rq = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        np.testing.assert_allclose(Rq, T[:3, :3], atol=1e-6)
        # world-space motion of the merged object: moves its centre by t and nothing else
        m = motions[10][k + 1]
        np.testing.assert_allclose(m @ np.append(centers[k], 1.0), np.append(centers[k] + T[:3, 3], 1.0), atol=1e-9)
    assert tables[0, 0, 11] > 0.45                                                   # starts half a metre above its rest
