"""The tight-list mode (cull_mode=1) that the HIP binning implements must not change a single bit of any
image: dropped (Gaussian, tile) instances are exactly those that every pixel of the tile would have skipped.
CPU only -- oracle against oracle."""
import math

import numpy as np
import pytest

from pegasus_amd import scenes


def _both(oracle, cloud, view, bg=(0.1, 0.2, 0.3), **kw):
    act = cloud.activated()
    a = oracle.forward(**act, sh_degree=3, **view.raster_kwargs(bg), num_threads=8, cull_mode=0, **kw)
    b = oracle.forward(**act, sh_degree=3, **view.raster_kwargs(bg), num_threads=8, cull_mode=1, **kw)
    return a, b


def _assert_bit_identical_images(a, b):
    for k in ("color", "out_depth", "final_T"):
        np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32), err_msg=k)
    np.testing.assert_array_equal(a["radii"], b["radii"])
    np.testing.assert_array_equal(a["tiles_touched"], b["tiles_touched"])


@pytest.mark.parametrize("maker", [
    lambda: scenes.scene_c1(),
    lambda: scenes.scene_c2(n=40_000, n_views=2, width=400, height=400),
    lambda: scenes.scene_c3(scale=0.03, n_views=2, width=400, height=304),
])
def test_tight_lists_preserve_every_pixel(oracle, maker):
    cloud, views = maker()
    for v in views:
        a, b = _both(oracle, cloud, v)
        _assert_bit_identical_images(a, b)
        assert b["num_instances"] < a["num_instances"]
        # tight lists are sub-sequences of the reference lists, tile by tile, in the same order
        ka, kb = a["keys_sorted"], b["keys_sorted"]
        pa = np.stack([ka, a["gauss_sorted"].astype(np.uint64)], 1)
        pb = np.stack([kb, b["gauss_sorted"].astype(np.uint64)], 1)
        sa = {tuple(x) for x in pa.tolist()}
        assert all(tuple(x) in sa for x in pb.tolist())
        assert np.all(np.diff(kb.astype(np.int64) >> 32) >= 0)


def test_tight_lists_with_anisotropic_and_faint_splats(oracle):
    """Needle-like, large and nearly transparent splats stress the edge-minimum bound and the opacity bound."""
    rng = np.random.default_rng(5)
    cloud, views = scenes.scene_c1(seed=21, n=6000)
    cloud.scaling[:, 0] += np.float32(math.log(8.0))           # needles
    cloud.scaling[::7] += np.float32(math.log(5.0))            # a few very large ones
    cloud.opacity[:] = rng.normal(-3.0, 2.5, size=cloud.opacity.shape).astype(np.float32)
    a, b = _both(oracle, cloud, views[0])
    _assert_bit_identical_images(a, b)
    assert b["num_instances"] < a["num_instances"]


def test_predicate_known_answers(oracle):
    W = H = 64
    iso = lambda s2: (1.0 / s2, 0.0, 1.0 / s2)      # conic of an isotropic splat with variance s2
    # centre inside the tile: always kept (unless opacity below 1/255)
    assert oracle.tile_may_contribute((20.0, 20.0), (*iso(4.0), 0.9), 1, 1, W, H)
    assert not oracle.tile_may_contribute((20.0, 20.0), (*iso(4.0), 0.003), 1, 1, W, H)
    # sigma = 2 px, opacity 0.9: alpha >= 1/255 needs d^2 <= 2*4*ln(229.5) = 43.5 -> d <= 6.6 px
    assert oracle.tile_may_contribute((21.0, 8.0), (*iso(4.0), 0.9), 0, 0, W, H)      # 6 px right of x = 15
    assert not oracle.tile_may_contribute((24.0, 8.0), (*iso(4.0), 0.9), 0, 0, W, H)  # 9 px away
    # diagonal neighbour: distance to the corner pixel (15,15) counts
    assert oracle.tile_may_contribute((19.0, 19.0), (*iso(4.0), 0.9), 0, 0, W, H)     # 5.7 px
    assert not oracle.tile_may_contribute((21.0, 21.0), (*iso(4.0), 0.9), 0, 0, W, H) # 8.5 px
    # degenerate conic: no claim, keep
    assert oracle.tile_may_contribute((100.0, 100.0), (0.0, 0.0, 0.0, 0.9), 0, 0, W, H)
