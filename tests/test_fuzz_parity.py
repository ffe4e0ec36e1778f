"""Randomised parity: many small scenes with random geometry statistics, image sizes (ragged included), cameras, SH
degrees, scale modifiers and a pinch of hostile values -- every one through the whole HIP pipeline and the oracle:
bit-exact radii / tile rectangles / per-tile lists / n_contrib, images to 1e-4.  The fixed-size parity scenes exercise
each code path on purpose; this one looks for the combination nobody thought of (tie-heavy lists, one-pixel images,
everything culled, splats larger than the image, clustered depths that defeat the bucket sort)."""
import math

import numpy as np
import pytest

from pegasus_amd import graphics as G, scenes

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 6000, 20000]))
    W, H = int(rng.choice([1, 2, 16, 17, 33, 100, 160, 257])), int(rng.choice([1, 3, 16, 31, 48, 112, 130]))
    cloud, _ = scenes.scene_c1(seed=100 + seed, n=max(n, 6))
    act = {k: a[:n].copy() for k, a in cloud.activated().items()}
    style = seed % 6
    if style == 1:                                   # a tight depth cluster: buckets pile up, merge-sort fallback
        act["means3d"][:, 2] = np.float32(0.25) + rng.normal(0, 1e-4, n).astype(np.float32)
    elif style == 2:                                 # exact depth ties in bulk
        act["means3d"][:, 2] = np.float32(rng.choice([0.0, 0.125, 0.5]))
    elif style == 3:                                 # huge splats (screen-filling), few of them visible
        act["scales"] *= np.float32(40.0)
    elif style == 4:                                 # dust: everything below one pixel
        act["scales"] *= np.float32(0.01)
    elif style == 5:                                 # hostile sprinkle
        pick = lambda f: rng.random(n) < f
        act["means3d"][pick(0.02)] = np.nan
        act["opacities"][pick(0.05)] = 0.0
        act["opacities"][pick(0.05)] = 1.0
        act["rotations"][pick(0.05)] *= np.float32(9.0)
        act["scales"][pick(0.03), 0] = 0.0
    fov = math.radians(float(rng.uniform(20, 110)))
    eye = rng.normal(0, 1.0, 3) * float(rng.choice([0.3, 1.5, 4.0]))
    if np.linalg.norm(eye) < 1e-3:
        eye = np.array([0.0, 0.0, -2.0])
    R, t = G.look_at_opencv(tuple(eye), tuple(rng.normal(0, 0.2, 3)), up=(0, -1, 0))
    view = scenes.make_view(R, t, W, H, fovx=fov, fovy=2 * math.atan(math.tan(fov / 2) * H / W))
    return act, view, int(rng.integers(0, 4)), float(rng.choice([1.0, 1.0, 0.3, 2.5])), tuple(rng.random(3).astype(np.float32))


@pytest.mark.parametrize("block", range(6))
def test_random_scenes_match_the_oracle(oracle, gpu_device, block):
    from helpers import gpu_forward
    for seed in range(block * 8, block * 8 + 8):
        act, view, deg, mod, bg = _case(seed)
        o = oracle.forward(**act, sh_degree=deg, **view.raster_kwargs(bg), num_threads=4, scale_modifier=mod, cull_mode=1)
        g = gpu_forward(act, view, sh_degree=deg, bg=bg, device=str(gpu_device), scale_modifier=mod)
        tag = f"seed {seed}: n {act['means3d'].shape[0]}, {view.width}x{view.height}, degree {deg}, modifier {mod}"
        np.testing.assert_array_equal(g["radii"], o["radii"], err_msg=tag)
        np.testing.assert_array_equal(g["tiles_touched"], o["tiles_touched"], err_msg=tag)
        assert g["num_instances"] == o["num_instances"], tag
        np.testing.assert_array_equal(g["gauss_sorted"], o["gauss_sorted"], err_msg=tag)
        amb = o["ambig"].astype(bool)
        fin = np.isfinite(o["color"]).all(axis=0) & np.isfinite(o["out_depth"][0])
        ok = ~amb & fin
        np.testing.assert_array_equal(g["n_contrib"][ok], o["n_contrib"][ok], err_msg=tag)
        scale = np.maximum(1.0, np.abs(o["color"]))
        assert (np.abs(g["color"] - o["color"]) / scale)[:, ok].max(initial=0) <= 1e-4, tag
        assert (np.abs(g["out_depth"][0] - o["out_depth"][0]) / np.maximum(1.0, np.abs(o["out_depth"][0])))[ok].max(initial=0) <= 1e-4, tag
        assert np.array_equal(np.isfinite(g["color"]).all(axis=0)[~amb], np.isfinite(o["color"]).all(axis=0)[~amb]), tag
