"""Camera-trajectory interpolation: SLERP for the rotation, linear for the position -- what
PegasusSetup.create_camera_trajectory feeds the renderer (/root/reference/src/gs/pegasus_setup.py:85-143 via
/root/reference/src/utility/pose_interpolation.py:87-107).  Pinned by tests/golden/pose_interpolation.npz."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot


def slerp(q1, q2, alpha: float, dot_threshold: float = 0.9995):
    q1, q2 = np.asarray(q1, dtype=np.float64), np.asarray(q2, dtype=np.float64)
    dot = float(q1 @ q2)
    if dot < 0:
        q1, dot = -q1, -dot
    if dot > dot_threshold:             # nearly parallel: normalised lerp
        res = q1 + alpha * (q2 - q1)
        return res / np.linalg.norm(res)
    theta_0 = np.arccos(dot)
    theta = theta_0 * alpha
    s2 = np.sin(theta) / np.sin(theta_0)
    return (np.cos(theta) - dot * s2) * q1 + s2 * q2


def interpolate_pose(t, t1, pose1, t2, pose2) -> np.ndarray:
    """4x4 poses at times t1, t2 -> 4x4 float32 pose at t in [t1, t2]."""
    assert t1 <= t <= t2
    r = (float(t) - float(t1)) / (float(t2) - float(t1))
    q1 = Rot.from_matrix(np.asarray(pose1)[:3, :3]).as_quat()
    q2 = Rot.from_matrix(np.asarray(pose2)[:3, :3]).as_quat()
    out = np.eye(4, dtype=np.float32)
    out[:3, :3] = Rot.from_quat(slerp(q1, q2, r)).as_matrix()
    out[:3, 3] = np.asarray(pose1)[:3, 3] + r * (np.asarray(pose2)[:3, 3] - np.asarray(pose1)[:3, 3])
    return out
