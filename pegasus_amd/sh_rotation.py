"""Rotation of real spherical-harmonics coefficients (bands 1-3) under a rigid rotation of the object.

Reference: /root/reference/src/gs/gaussian_model.py:507-546 builds Wigner-D matrices with e3nn
(``o3.wigner_D(l, alpha, -beta, gamma)`` after the axis permutation P = [[0,0,1],[1,0,0],[0,1,0]]) and applies
them per band to ``_features_rest``.  e3nn is not available here, and the matrices depend on the basis
convention, so they are derived directly from the rasterizer's own basis (pegasus_amd.sh_utils.sh_basis):

    a splat with coefficients c radiates  f(d) = sum_m c_m Y_m(d).  After rotating the object by R the same
    radiance must leave in direction R d:  f'(d) = f(R^T d).  Band l is closed under rotation, so
    c' = D_l(R) c  with  D_l = pinv(B) @ B_rot,  B[k,m] = Y_lm(d_k),  B_rot[k,m] = Y_lm(R^T d_k)

over a fixed well-conditioned set of sample directions.  Exact up to fp64 round-off (tests check
|f'(d) - f(R^T d)| < 1e-12 on random directions).
"""
from __future__ import annotations

import numpy as np

from .graphics import fibonacci_sphere
from .sh_utils import sh_basis

_DIRS = fibonacci_sphere(61, 1.0)
_DIRS = _DIRS / np.linalg.norm(_DIRS, axis=1, keepdims=True)
_BAND = {1: slice(1, 4), 2: slice(4, 9), 3: slice(9, 16)}
_B = sh_basis(3, _DIRS)
_PINV = {l: np.linalg.pinv(_B[:, s]) for l, s in _BAND.items()}


def sh_rotation_matrices(R) -> tuple:
    """(D1 [3,3], D2 [5,5], D3 [7,7]) float64 such that c'_l = D_l @ c_l rotates band l with the object."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    Brot = sh_basis(3, _DIRS @ R)          # rows: Y(R^T d_k)  since (R^T d)^T = d^T R
    return tuple(_PINV[l] @ Brot[:, s] for l, s in _BAND.items())


def rotate_sh_rest(f_rest: np.ndarray, R) -> np.ndarray:
    """f_rest [N,15,3] (coefficient-major, RGB-minor) -> rotated copy (numpy reference of the HIP kernel)."""
    D1, D2, D3 = sh_rotation_matrices(R)
    out = np.array(f_rest, dtype=np.float64, copy=True)
    out[:, 0:3] = np.einsum("ij,njc->nic", D1, f_rest[:, 0:3])
    out[:, 3:8] = np.einsum("ij,njc->nic", D2, f_rest[:, 3:8])
    out[:, 8:15] = np.einsum("ij,njc->nic", D3, f_rest[:, 8:15])
    return out
