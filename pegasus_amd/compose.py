"""Scene composition on the device: environment + posed objects -> one merged, activated point cloud that
the rasterizer consumes directly.

What the reference does per frame (/root/reference/pegasus.py:255-264,387-390): deepcopy the environment,
``merge_gaussians`` every object (6 x torch.vstack each), and in dynamic mode apply the step's delta pose
through host-side scipy/e3nn (gaussian_model.py:482-546).  Here the merged buffers are allocated ONCE
(environment rows are written once), pose-independent object attributes (opacity, scale, f_dc) are
written once, and a pose update is one ``pgr_compose_object`` launch per object that writes position,
orientation and rotated SH straight into the object's rows.  Poses are ABSOLUTE (T_k applied to the canonical
object), which is mathematically what the reference's accumulated deltas produce and keeps time steps
independent.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np
import torch

from . import _lib
from .sh_rotation import sh_rotation_matrices


def make_pose(T, center) -> _lib.PgrObjectPose:
    """PgrObjectPose from a 4x4 rigid transform and the rotation centre (the object's cloud mean)."""
    from scipy.spatial.transform import Rotation as Rot
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    R = T[:3, :3]
    q_xyzw = Rot.from_matrix(R).as_quat()
    D1, D2, D3 = sh_rotation_matrices(R)
    f = lambda a: np.asarray(a, dtype=np.float32).reshape(-1)
    p = _lib.PgrObjectPose()
    p.R[:] = f(R); p.t[:] = f(T[:3, 3]); p.center[:] = f(center)
    p.q[:] = f([q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]])
    p.D1[:] = f(D1); p.D2[:] = f(D2); p.D3[:] = f(D3)
    return p


def pose_table(poses) -> np.ndarray:
    """[K, 20] float32 rows (R[9] row-major, t[3], center[3], q[4] (w,x,y,z), 0) for the posed-objects path
    (PgrPosedObjects / oracle ``poses``) from K PgrObjectPose structs (``make_pose``) or (T, center) pairs.
    (T, center) pairs skip the SH band matrices, which this path does not use."""
    from scipy.spatial.transform import Rotation as Rot
    rows = np.zeros((len(poses), _lib.PGR_POSE_STRIDE), dtype=np.float32)
    for k, p in enumerate(poses):
        if isinstance(p, _lib.PgrObjectPose):
            rows[k, 0:9], rows[k, 9:12], rows[k, 12:15], rows[k, 15:19] = list(p.R), list(p.t), list(p.center), list(p.q)
        else:
            T, center = np.asarray(p[0], dtype=np.float64).reshape(4, 4), np.asarray(p[1], dtype=np.float64)
            q = Rot.from_matrix(T[:3, :3]).as_quat()
            rows[k, 0:9], rows[k, 9:12], rows[k, 12:15] = T[:3, :3].reshape(-1), T[:3, 3], center
            rows[k, 15:19] = (q[3], q[0], q[1], q[2])
    return rows


def compose_object(xyz, rot, f_rest, pose: _lib.PgrObjectPose, out_xyz, out_rot, out_rest, out_rest_stride=None):
    """Thin wrapper over pgr_compose_object for torch device tensors (f_rest [n,R,3] contiguous)."""
    L = _lib.lib()
    if xyz.device.type != "cuda":
        raise RuntimeError("compose_object needs HIP device tensors; there is no CPU path")
    n = int(xyz.shape[0])
    n_rest = int(f_rest.shape[1]) if f_rest is not None else 0
    in_stride = 3 * n_rest
    if out_rest_stride is None:
        out_rest_stride = in_stride
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(xyz.device):
        _lib.check(L.pgr_compose_object(n, p(xyz), p(rot), p(f_rest), n_rest, in_stride, C.byref(pose), p(out_xyz),
                                        p(out_rot), p(out_rest), int(out_rest_stride),
                                        C.c_void_p(torch.cuda.current_stream(xyz.device).cuda_stream)),
                   "pgr_compose_object")


class SceneComposer:
    """Merged scene = environment rows, then object 1 rows, object 2 rows, ...  (the order PEGASUS merges in)."""

    def __init__(self, environment, objects: Sequence, device="cuda:0"):
        """``environment`` and each object: a cloud with raw GaussianModel parameters as numpy/torch arrays
        (attributes xyz [n,3], features_dc [n,1,3], features_rest [n,15,3], opacity [n,1], scaling [n,3],
        rotation [n,4]) -- e.g. pegasus_amd.scenes.SplatCloud or pegasus_amd.gaussian_model.GaussianModel."""
        self.device = torch.device(device)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a
                                      ).detach().to(self.device, torch.float32).contiguous()
        g = lambda c, *names: next(getattr(c, k) for k in names if hasattr(c, k))
        clouds = [environment, *objects]
        sizes = [int(g(c, "xyz", "_xyz").shape[0]) for c in clouds]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        N = int(self.offsets[-1])
        self.n_env, self.n = sizes[0], N
        dev = self.device
        self.means3d = torch.empty((N, 3), device=dev)
        self.opacities = torch.empty((N,), device=dev)
        self.scales = torch.empty((N, 3), device=dev)
        self.rotations = torch.empty((N, 4), device=dev)
        self.shs = torch.empty((N, 16, 3), device=dev)
        self.object_id = torch.zeros((N,), dtype=torch.int32, device=dev)
        self.canon: List[dict] = []
        for k, c in enumerate(clouds):
            s = slice(int(self.offsets[k]), int(self.offsets[k + 1]))
            xyz, rot = t(g(c, "xyz", "_xyz")), t(g(c, "rotation", "_rotation"))
            f_dc, f_rest = t(g(c, "features_dc", "_features_dc")), t(g(c, "features_rest", "_features_rest"))
            # activations = the reference's getters (gaussian_model.py:105-128), applied once
            self.opacities[s] = torch.sigmoid(t(g(c, "opacity", "_opacity")).reshape(-1))
            self.scales[s] = torch.exp(t(g(c, "scaling", "_scaling")))
            self.shs[s, 0:1] = f_dc
            self.object_id[s] = k
            if k == 0:
                self.means3d[s] = xyz
                self.rotations[s] = torch.nn.functional.normalize(rot)
                self.shs[s, 1:] = f_rest
            else:
                self.canon.append(dict(xyz=xyz, rot=rot, f_rest=f_rest, center=xyz.mean(0).cpu().numpy(), slice=s))
        self.set_poses([np.eye(4)] * len(objects))

    def set_poses(self, poses: Sequence):
        """poses[k]: 4x4 rigid transform of object k+1 (x' = R (x - mean) + mean + t, as
        apply_transformation_on_xyz does)."""
        if len(poses) != len(self.canon):
            raise ValueError("one pose per object")
        for T, c in zip(poses, self.canon):
            s = c["slice"]
            pose = make_pose(T, c["center"])
            # rotated SH go straight into columns 1..15 of the merged [N,16,3] feature rows (stride 48 floats)
            out_rest = self.shs[s.start:s.stop].view(-1)[3:]
            compose_object(c["xyz"], c["rot"], c["f_rest"], pose, self.means3d[s], self.rotations[s], out_rest,
                           out_rest_stride=48)

    def tensors(self):
        return dict(means3d=self.means3d, opacities=self.opacities, scales=self.scales, rotations=self.rotations,
                    shs=self.shs)
