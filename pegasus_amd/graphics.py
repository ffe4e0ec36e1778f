"""Camera / projection maths for the rasterizer boundary (host side, numpy).

These are the build's own versions of the helpers PEGASUS imports from the missing
``gaussian-splatting-pegasus`` submodule (``utils.graphics_utils``; import sites
/root/reference/src/gs/pegasus_setup.py:10-15, /root/reference/pegasus.py:20-23).  Their
behaviour is fixed by SURVEY.md section 8a "Camera matrices" and by how the reference uses them:
/root/reference/src/gs/pegasus_setup.py:97,119-122 (R is camera-to-world, focal2fov) and
/root/reference/src/tools/pegasus_working.py:349-356,464-466 (fov2focal, T_w2c[:3,:3] = cam.R.T).
"""
from __future__ import annotations

import math

import numpy as np

ZNEAR = 0.01
ZFAR = 100.0


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0) -> np.ndarray:
    """World-to-view 4x4 (float32).  ``R`` is the CAMERA-TO-WORLD rotation, ``t`` the
    world-to-camera translation (COLMAP convention, as PEGASUS passes them)."""
    w2c = np.eye(4, dtype=np.float64)
    w2c[:3, :3] = np.asarray(R, dtype=np.float64).T
    w2c[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def getWorld2View(R, t) -> np.ndarray:
    return getWorld2View2(R, t)


def getProjectionMatrix(znear: float, zfar: float, fovX: float, fovY: float) -> np.ndarray:
    """Symmetric-frustum perspective matrix with z mapped to [0,1] and w = +z (float32)."""
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 1.0 / math.tan(fovX / 2.0)
    P[1, 1] = 1.0 / math.tan(fovY / 2.0)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_matrices(R_c2w, t_w2c, fovx: float, fovy: float, znear: float = ZNEAR, zfar: float = ZFAR):
    """Returns (world_view_transform, projection_matrix, full_proj_transform, camera_center) in the
    TRANSPOSED (row-vector) storage the rasterizer surface expects, all float32."""
    wvt = getWorld2View2(R_c2w, t_w2c).T.copy()
    proj = getProjectionMatrix(znear, zfar, fovx, fovy).T.copy()
    full = (wvt @ proj).astype(np.float32)
    center = np.linalg.inv(wvt)[3, :3].astype(np.float32)
    return wvt, proj, full, center


def look_at_opencv(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """World-to-camera rotation R_w2c (rows = camera x right, y down, z forward) and translation
    t = -R_w2c @ eye for a camera at ``eye`` looking at ``target`` (OpenCV axes)."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    side = np.cross(fwd, np.asarray(up, dtype=np.float64))
    if not np.any(side):
        side = np.array([1.0, 0.0, 0.0])
    side /= np.linalg.norm(side)
    upv = np.cross(side, fwd)
    R = np.stack([side, -upv, fwd])
    return R, -R @ eye


def fibonacci_sphere(n_pts: int, radius: float = 1.0) -> np.ndarray:
    """Odd number of near-equidistant points on a sphere (Fibonacci lattice), south to north."""
    if n_pts % 2 != 1:
        raise ValueError("n_pts must be odd")
    half = n_pts // 2
    golden = (math.sqrt(5.0) + 1.0) / 2.0
    ga = 2.0 * math.pi * (golden - 1.0)
    pts = []
    for i in range(-half, half + 1):
        lat = math.asin((2 * i) / float(2 * half + 1))
        lon = (ga * i) % (2.0 * math.pi)
        s = math.cos(lat) * radius
        pts.append([math.cos(lon) * s, math.sin(lon) * s, math.tan(lat) * s])
    return np.asarray(pts, dtype=np.float64)


def hemisphere_views(min_n_views: int, radius: float, elev_range=(0.0, 0.5 * math.pi)):
    """Look-at views from the upper part of a Fibonacci sphere; the same view set as the BOP
    toolkit's ``sample_views(min_n_views, radius, elev_range=..., mode='fibonacci')``
    (/root/reference/submodules/bop_toolkit/bop_toolkit_lib/view_sampler.py:168-237), pinned by
    tests/golden/bop_fibonacci_views.npz.  Returns a list of (R_w2c[3,3], t_w2c[3])."""
    n = min_n_views if min_n_views % 2 == 1 else min_n_views + 1
    views = []
    for pt in fibonacci_sphere(n, radius):
        a = np.linalg.norm(pt)
        b = np.linalg.norm([pt[0], pt[1], 0.0])
        elev = math.acos(b / a)
        if pt[2] < 0:
            elev = -elev
        if not (elev_range[0] <= elev <= elev_range[1]):
            continue
        R, t = look_at_opencv(pt, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0))
        views.append((R, t))
    return views
