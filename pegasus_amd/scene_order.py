"""Spatially coherent storage order for a resident scene.

The rasterizer's results do not depend on the order Gaussians are stored in (lists are ordered by depth, ties by
index), but its memory behaviour does: with neighbours in space adjacent in memory a binning chunk touches few
tiles (long runs per tile, few global reservations) and a tile's gathers of 48-B splat records fall into shared
cache lines.  ``spatial_order`` returns a permutation that sorts every group (environment, each object) by the
Morton code of its positions; groups keep their relative order, so ``index < n_env  <=>  environment`` (the
fused semantic pass relies on it) still holds.
"""
from __future__ import annotations

import numpy as np


def _part1by2(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton_codes(xyz: np.ndarray, bits: int = 16) -> np.ndarray:
    """63-bit-safe Morton codes of positions quantised to ``bits`` per axis over their bounding box."""
    xyz = np.asarray(xyz, dtype=np.float64)
    lo, hi = xyz.min(axis=0), xyz.max(axis=0)
    span = np.maximum(hi - lo, 1e-12)
    q = np.minimum(((xyz - lo) / span * (2 ** bits)).astype(np.int64), 2 ** bits - 1)
    return _part1by2(q[:, 0]) | (_part1by2(q[:, 1]) << np.uint64(1)) | (_part1by2(q[:, 2]) << np.uint64(2))


def spatial_order(xyz: np.ndarray, object_id: np.ndarray | None = None, bits: int = 16) -> np.ndarray:
    """Permutation (int64 [N]): stable by group id (object_id, 0 = environment first), Morton order inside a group."""
    n = int(np.asarray(xyz).shape[0])
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    codes = morton_codes(xyz, bits)
    if object_id is None:
        return np.argsort(codes, kind="stable")
    return np.lexsort((codes, np.asarray(object_id).astype(np.int64)))
