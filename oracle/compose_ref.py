"""numpy float64 restatement of PEGASUS's object posing (TEST INFRASTRUCTURE; see oracle/pgr_oracle.h).

Follows /root/reference/src/gs/gaussian_model.py:
  :482-497  apply_rotation_on_xyz / apply_translation_on_xyz / apply_transformation_on_xyz
            x' = R (x - mean(x)) + mean(x), then + t
  :499-505  apply_rotation_on_splats: q' = quat(R @ build_rotation(normalise(q)))   (w,x,y,z)
  :507-546  apply_rotation_on_sh: bands 1..3 of _features_rest multiplied by the band's rotation matrix.
            The reference takes the matrices from e3nn (absent here); the defining property
            f'(d) = f(R^T d) is what tests/test_compose.py checks instead.
"""
import numpy as np


def quat_to_matrix(q):
    w, x, y, z = (np.asarray(q, np.float64) / np.linalg.norm(q)).tolist()
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def compose_object_ref(xyz, rot, T):
    xyz = np.asarray(xyz, np.float64)
    R, t = np.asarray(T, np.float64)[:3, :3], np.asarray(T, np.float64)[:3, 3]
    mean = xyz.mean(0)
    out_xyz = (R @ (xyz - mean).T).T + mean + t
    out_R = np.stack([R @ quat_to_matrix(q) for q in np.asarray(rot, np.float64)])
    return out_xyz, out_R
