"""Object poses of a dynamic sequence, composed ABSOLUTELY per time step from a simulated trajectory.

The reference simulates the whole drop first and stores it (``src/engine/simulation_steps.json``: per body and step a
position ``t`` and a quaternion ``q`` in scipy's x,y,z,w order; /root/reference/src/engine/physical_simulation.py:163-168),
then renders step after step, moving every object by the DELTA between consecutive steps
(/root/reference/src/gs/pegasus_setup.py:160-193):

    step 0 :  x <- R(q_0) (x - mean) + mean + t_0                         (dynamic_object_pose)
    step s :  x <- R(q_s q_{s-1}^-1) (x - mean') + mean' + (t_s - t_{s-1})   (update_object_pose)

A rotation about the cloud's current mean leaves that mean where it is, so after s steps the accumulated motion is

    x_s = R(q_s) (x - mean) + mean + t_s

-- the step-s sample applied to the canonical object about its own centre.  That closed form is what this module
produces: time steps become independent (they shard across GPUs) and no fp32 drift accumulates over
a 200-step sequence.  ``tests/test_trajectory.py`` checks it against the reference's delta recursion.

For BASELINE.json configs[4] the synthetic C5 scene holds its objects AT REST (the end of a drop); the sequence replays
the recorded fall of body 1 (``tests/golden/simulation_steps_body1_first200.npz``, captured from the reference's fixture
by tests/golden/make_golden_from_reference.py) relative to its final sample, for every object at its own place and with
its own phase offset ("replicated with per-object offsets").
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np

FIXTURE = Path(__file__).resolve().parents[1] / "tests" / "golden" / "simulation_steps_body1_first200.npz"


def load_fixture(path=None) -> np.ndarray:
    """[S, 7] float64: t (3), q (x, y, z, w) of body 1 for steps 0 .. S-1."""
    with np.load(str(path or FIXTURE)) as d:
        return np.asarray(d["t_q_xyzw"], dtype=np.float64)


def _rot(q_xyzw) -> np.ndarray:
    from scipy.spatial.transform import Rotation
    return Rotation.from_quat(np.asarray(q_xyzw, dtype=np.float64)).as_matrix()


def absolute_pose(traj: np.ndarray, step: int) -> np.ndarray:
    """4x4 [R(q_s) | t_s]: the accumulated motion of the reference's recursion at ``step`` (applied about the centre)."""
    T = np.eye(4)
    T[:3, :3] = _rot(traj[step, 3:7])
    T[:3, 3] = traj[step, 0:3]
    return T


def relative_to_rest(traj: np.ndarray, step: int, rest: int = -1) -> np.ndarray:
    """Motion that takes the object from its pose at ``rest`` (default: the last sample) to its pose at ``step``,
    about the object's centre: R = R(q_s) R(q_rest)^T, t = t_s - t_rest."""
    T = np.eye(4)
    T[:3, :3] = _rot(traj[step, 3:7]) @ _rot(traj[rest, 3:7]).T
    T[:3, 3] = traj[step, 0:3] - traj[rest, 0:3]
    return T


def accumulate_deltas(traj: np.ndarray, step: int, center, points) -> np.ndarray:
    """The reference's recursion, literally (float64): initial pose, then ``step`` delta updates about the moving
    mean.  Test helper for the closed form above; ``points`` [n,3] with mean ``center``."""
    from scipy.spatial.transform import Rotation
    x = np.asarray(points, dtype=np.float64)
    c = np.asarray(center, dtype=np.float64)
    x = (Rotation.from_quat(traj[0, 3:7]).as_matrix() @ (x - c).T).T + c + traj[0, 0:3]
    for s in range(1, step + 1):
        m = x.mean(0)
        q_delta = Rotation.from_quat(traj[s, 3:7]) * Rotation.from_quat(traj[s - 1, 3:7]).inv()
        x = (q_delta.as_matrix() @ (x - m).T).T + m + (traj[s, 0:3] - traj[s - 1, 0:3])
    return x


def sequence_poses(traj: np.ndarray, centers: Sequence, n_steps: int, phase: int = 5, first_step: int = 0
                   ) -> Tuple[np.ndarray, List[Dict[int, np.ndarray]]]:
    """Pose rows of ``n_steps`` time steps for K objects resting at ``centers`` (their cloud means in the merged scene).

    Object k (1-based) at time step s shows trajectory sample min(S-1, first_step + s + phase * (k-1)) relative to
    the trajectory's last sample: every object falls onto its own resting place, object k running ``phase`` steps ahead
    of object k-1.  Returns (tables [n_steps, K, 20] float32 for PgrPosedObjects / FrameRenderer ``poses``,
    world_motion: per step {obj_id: 4x4 world-space motion of the merged object, C T C^-1}) -- multiply the latter onto an
    object's rest placement to get its model-to-world pose for the BOP records."""
    from .compose import pose_table
    S, K = traj.shape[0], len(centers)
    cache: Dict[int, np.ndarray] = {}
    tables = np.zeros((n_steps, K, 20), np.float32)
    motions: List[Dict[int, np.ndarray]] = []
    for s in range(n_steps):
        pairs, mot = [], {}
        for k in range(K):
            idx = min(S - 1, first_step + s + phase * k)
            T = cache.get(idx)
            if T is None:
                T = cache[idx] = relative_to_rest(traj, idx)
            c = np.asarray(centers[k], dtype=np.float64)
            pairs.append((T, c))
            Cp, Cm = np.eye(4), np.eye(4)
            Cp[:3, 3], Cm[:3, 3] = c, -c
            mot[k + 1] = Cp @ T @ Cm
        tables[s] = pose_table(pairs)
        motions.append(mot)
    return tables, motions
