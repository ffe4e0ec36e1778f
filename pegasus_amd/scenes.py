"""Synthetic merged point clouds and camera sets for BASELINE.json's configs.

The real datasets are 50 GB downloads (/root/reference/README.md:8) that are not available, so
every workload is generated from ``numpy.random.default_rng(seed)`` in fp32.  Layout follows
what PEGASUS hands to the rasterizer: one merged cloud, environment Gaussians first and then
each object in turn (/root/reference/pegasus.py:255-264, src/gs/gaussian_model.py:584-591),
with the raw (pre-activation) parameter tensors of ``GaussianModel``
(/root/reference/src/gs/gaussian_model.py:277-288).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import graphics as G


@dataclass
class SplatCloud:
    """Raw GaussianModel parameters (numpy fp32) + the object id of every Gaussian (0 = environment)."""
    xyz: np.ndarray            # [N,3]
    features_dc: np.ndarray    # [N,1,3]
    features_rest: np.ndarray  # [N,15,3]
    opacity: np.ndarray        # [N,1] logits
    scaling: np.ndarray        # [N,3] log-scales
    rotation: np.ndarray       # [N,4] (w,x,y,z), not normalised
    object_id: np.ndarray = field(default=None)  # [N] int32

    @property
    def n(self) -> int:
        return int(self.xyz.shape[0])

    # activations = the reference's getters (gaussian_model.py:105-128), in numpy fp32
    def activated(self):
        scales = np.exp(self.scaling).astype(np.float32)
        nrm = np.sqrt((self.rotation.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32))
        rots = (self.rotation / np.maximum(nrm, np.float32(1e-12))).astype(np.float32)
        opac = (1.0 / (1.0 + np.exp(-self.opacity.astype(np.float32)))).astype(np.float32)
        shs = np.concatenate([self.features_dc, self.features_rest], axis=1).astype(np.float32)
        return dict(means3d=np.ascontiguousarray(self.xyz, dtype=np.float32), opacities=opac.reshape(-1),
                    scales=scales, rotations=rots, shs=np.ascontiguousarray(shs))

    @staticmethod
    def concat(parts):
        cat = lambda k: np.concatenate([getattr(p, k) for p in parts], axis=0)
        return SplatCloud(cat("xyz"), cat("features_dc"), cat("features_rest"), cat("opacity"),
                          cat("scaling"), cat("rotation"), cat("object_id"))


@dataclass
class View:
    width: int
    height: int
    fovx: float
    fovy: float
    R_c2w: np.ndarray
    t_w2c: np.ndarray
    world_view_transform: np.ndarray
    full_proj_transform: np.ndarray
    camera_center: np.ndarray

    @property
    def tanfovx(self):
        return math.tan(self.fovx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.fovy * 0.5)

    def raster_kwargs(self, bg=(0.0, 0.0, 0.0)):
        return dict(width=self.width, height=self.height, tanfovx=self.tanfovx, tanfovy=self.tanfovy,
                    viewmatrix=self.world_view_transform, projmatrix=self.full_proj_transform,
                    campos=self.camera_center, bg=np.asarray(bg, np.float32))


def make_view(R_w2c, t_w2c, width, height, fx=None, fy=None, fovx=None, fovy=None) -> View:
    if fovx is None:
        fovx = G.focal2fov(fx, width)
    if fovy is None:
        fovy = G.focal2fov(fy, height)
    R_c2w = np.asarray(R_w2c, dtype=np.float64).T
    wvt, _, full, center = G.camera_matrices(R_c2w, t_w2c, fovx, fovy)
    return View(width, height, fovx, fovy, R_c2w, np.asarray(t_w2c, np.float64), wvt, full, center)


# ------------------------------------------------------------------------------------------
def _common_appearance(rng, n):
    opacity = rng.normal(2.0, 1.5, size=(n, 1)).astype(np.float32)
    f_dc = rng.uniform(-1.5, 1.5, size=(n, 1, 3)).astype(np.float32)
    f_rest = rng.normal(0.0, 0.1, size=(n, 15, 3)).astype(np.float32)
    rot = rng.normal(size=(n, 4)).astype(np.float32)
    return opacity, f_dc, f_rest, rot


def _box_surface(rng, n, dims):
    """n points uniform on the surface of an axis-aligned box centred at the origin; returns
    (xyz, normal_axis)."""
    dx, dy, dz = dims
    areas = np.array([dy * dz, dy * dz, dx * dz, dx * dz, dx * dy, dx * dy])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    uvw = rng.uniform(-0.5, 0.5, size=(n, 3))
    axis = face // 2
    sign = np.where(face % 2 == 0, -0.5, 0.5)
    uvw[np.arange(n), axis] = sign
    return (uvw * np.asarray(dims)).astype(np.float32), axis


def box_object(rng, n, dims, log_scale_mu, log_scale_sigma, normal_factor, object_id) -> SplatCloud:
    xyz, axis = _box_surface(rng, n, dims)
    scaling = rng.normal(log_scale_mu, log_scale_sigma, size=(n, 3)).astype(np.float32)
    scaling[np.arange(n), axis] += np.float32(math.log(normal_factor))
    opacity, f_dc, f_rest, rot = _common_appearance(rng, n)
    return SplatCloud(xyz, f_dc, f_rest, opacity, scaling, rot, np.full(n, object_id, np.int32))


def ground_plane(rng, n, size, log_scale_mu, log_scale_sigma, normal_factor, height_noise) -> SplatCloud:
    xyz = np.empty((n, 3), np.float32)
    xyz[:, :2] = rng.uniform(-0.5 * size, 0.5 * size, size=(n, 2))
    xyz[:, 2] = rng.uniform(-height_noise, height_noise, size=n)
    scaling = rng.normal(log_scale_mu, log_scale_sigma, size=(n, 3)).astype(np.float32)
    scaling[:, 2] += np.float32(math.log(normal_factor))
    opacity, f_dc, f_rest, rot = _common_appearance(rng, n)
    # splats lie in the plane: keep rotations near identity so the thin axis stays the normal
    rot = (np.array([1.0, 0, 0, 0], np.float32) + 0.05 * rot).astype(np.float32)
    return SplatCloud(xyz, f_dc, f_rest, opacity, scaling, rot, np.zeros(n, np.int32))


def rigid_transform(cloud: SplatCloud, Rm: np.ndarray, t: np.ndarray) -> SplatCloud:
    """x' = R x + t, q' = q_R * q (SH left unrotated: appearance only, synthetic data)."""
    from scipy.spatial.transform import Rotation as Rot
    xyz = (cloud.xyz.astype(np.float64) @ Rm.T + t).astype(np.float32)
    qR = Rot.from_matrix(Rm)
    q = cloud.rotation.astype(np.float64)
    q_xyzw = np.concatenate([q[:, 1:], q[:, :1]], axis=1)
    nrm = np.linalg.norm(q_xyzw, axis=1, keepdims=True)
    comp = (qR * Rot.from_quat(q_xyzw / nrm)).as_quat() * nrm
    rot = np.concatenate([comp[:, 3:], comp[:, :3]], axis=1).astype(np.float32)
    return SplatCloud(xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling, rot,
                      cloud.object_id)


# ------------------------------------------------------------------------------------------
def scene_c1(seed: int = 1, n: int = 10_000):
    """C1: 10 k Gaussians on the faces of a unit cube, one 256x256 view from distance 3."""
    rng = np.random.default_rng(seed)
    cloud = box_object(rng, n, (1.0, 1.0, 1.0), math.log(0.02), 0.3, 0.1, object_id=1)
    fov = math.radians(50.0)
    R, t = G.look_at_opencv((0.0, 0.0, -3.0), (0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0))
    return cloud, [make_view(R, t, 256, 256, fovx=fov, fovy=fov)]


def object_views(n_views: int, radius: float, width=800, height=800, focal=1000.0, target=(0, 0, 0),
                 min_views_on_sphere=None):
    """First ``n_views`` cameras of the Fibonacci upper hemisphere, looking at ``target``."""
    n_sphere = min_views_on_sphere or 2 * n_views
    views = G.hemisphere_views(n_sphere, radius)
    while len(views) < n_views:
        n_sphere *= 2
        views = G.hemisphere_views(n_sphere, radius)
    tgt = np.asarray(target, np.float64)
    out = []
    for R, t in views[:n_views]:
        eye = -R.T @ t + tgt
        out.append(make_view(R, -R @ eye, width, height, fx=focal, fy=focal))
    return out


def scene_c2(seed: int = 2, n: int = 150_000, n_views: int = 64, width=800, height=800):
    """C2: one cracker-box-sized object (0.06 x 0.16 x 0.21 m), 64 hemisphere views at 0.45 m."""
    rng = np.random.default_rng(seed)
    cloud = box_object(rng, n, (0.06, 0.16, 0.21), math.log(0.002), 0.4, 0.15, object_id=1)
    return cloud, object_views(n_views, 0.45, width, height, 1000.0 * width / 800.0, min_views_on_sphere=128)


CAMERA_SETS = {
    # "512 (C3) / 4096 (C4) views from the fibonacci sampler" -- the first n_views of the BOP
    # toolkit's Fibonacci upper hemisphere (elevation 0 .. 90 degrees, grazing views included)
    "fibonacci": None,
    # rounds 1-4 rendered this subset: directions below ~8.6 degrees of elevation (eye_dir.z <= 0.15, 15 % of the set)
    # skipped and back-filled by repeating directions at other radii.  Kept for A/B against the earlier rounds' numbers.
    "fibonacci_above_9deg": 0.15,
}


def merged_scene(seed: int, n_env: int, n_objects: int, n_per_object: int, n_views: int,
                 width=800, height=800, plane_size=2.0, camera_set: str = "fibonacci"):
    """C3/C4/C5 construction: ground-plane environment + objects resting on it."""
    min_z = CAMERA_SETS[camera_set]
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    parts = [ground_plane(rng, n_env, plane_size, math.log(0.004), 0.5, 0.1, 0.01)]
    poses = []
    for k in range(n_objects):
        dims = np.array([0.06, 0.16, 0.21]) * rng.uniform(0.7, 1.3, size=3)
        obj = box_object(rng, n_per_object, dims, math.log(0.002), 0.4, 0.15, object_id=k + 1)
        r = 0.5 * math.sqrt(rng.uniform())
        phi = rng.uniform(0, 2 * math.pi)
        yaw = rng.uniform(0, 2 * math.pi)
        tilt = Rot.from_euler("xy", rng.normal(0, 0.05, size=2))
        # lying on one of its faces: choose which axis points up
        up_axis = int(rng.integers(3))
        lay = [Rot.from_euler("y", 90, degrees=True), Rot.from_euler("x", 90, degrees=True), Rot.identity()][up_axis]
        Rm = (Rot.from_euler("z", yaw) * tilt * lay).as_matrix()
        rest_h = 0.5 * dims[up_axis] + 0.012
        t = np.array([r * math.cos(phi), r * math.sin(phi), rest_h])
        poses.append((Rm, t))
        parts.append(rigid_transform(obj, Rm, t))
    cloud = SplatCloud.concat(parts)
    # cameras: Fibonacci hemisphere, radius cycling over 0.8..1.2 m, looking at the scene centre
    base = G.hemisphere_views(2 * n_views + 2, 1.0)
    views = []
    k = 0
    target = np.array([0.0, 0.0, 0.05])
    while len(views) < n_views:
        R, t = base[k % len(base)]
        eye_dir = -R.T @ t
        if min_z is None or eye_dir[2] > min_z:
            radius = 0.8 + 0.4 * ((len(views) * 0.6180339887498949) % 1.0)
            eye = eye_dir * radius + target
            views.append(make_view(R, -R @ eye, width, height, fx=1000.0 * width / 800.0,
                                   fy=1000.0 * height / 800.0))
        k += 1
        if k > 8 * (n_views + len(base)):
            raise RuntimeError("camera sampler did not converge")
    return cloud, views, poses


def scene_c3(seed: int = 3, n_views: int = 512, scale: float = 1.0, width=800, height=800, camera_set="fibonacci"):
    """C3/C4: 1.36 M environment + 8 x 80 k objects = 2.0 M Gaussians (scale < 1 shrinks all counts)."""
    cloud, views, _ = merged_scene(seed, int(1_360_000 * scale), 8, int(80_000 * scale), n_views, width, height,
                                   camera_set=camera_set)
    return cloud, views


def scene_c5(seed: int = 5, n_views: int = 200, scale: float = 1.0, width=800, height=800, camera_set="fibonacci"):
    """C5: 3.4 M environment + 20 x 80 k objects = 5.0 M Gaussians."""
    cloud, views, _ = merged_scene(seed, int(3_400_000 * scale), 20, int(80_000 * scale), n_views, width, height,
                                   camera_set=camera_set)
    return cloud, views
