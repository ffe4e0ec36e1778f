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
        # ... and against the oracle WITHOUT the tight-list predicate (reference-style lists: every tile of the 3-sigma
        # rectangle): same radii, same images, and every pixel's last blended Gaussian although the list layouts differ
        o0 = oracle.forward(**act, sh_degree=deg, **view.raster_kwargs(bg), num_threads=4, scale_modifier=mod, cull_mode=0)
        np.testing.assert_array_equal(g["radii"], o0["radii"], err_msg=tag)
        assert o0["num_instances"] >= g["num_instances"], tag
        amb0 = o0["ambig"].astype(bool)
        ok0 = ~amb0 & np.isfinite(o0["color"]).all(axis=0) & np.isfinite(o0["out_depth"][0])
        assert (np.abs(g["color"] - o0["color"]) / np.maximum(1.0, np.abs(o0["color"])))[:, ok0].max(initial=0) <= 1e-4, tag
        from test_gpu_parity import _last_blended
        lg = _last_blended(g["gauss_sorted"], g["ranges"], g["n_contrib"], view.width, view.height)
        l0 = _last_blended(o0["gauss_sorted"], o0["ranges"], o0["n_contrib"], view.width, view.height)
        np.testing.assert_array_equal(lg[ok0], l0[ok0], err_msg=tag)


@pytest.mark.parametrize("seed", range(6))
def test_random_merged_scenes_fused_frames_equal_separate_passes(gpu_device, seed):
    """FrameRenderer on random small merged scenes (random object count, densities, image size, background, objects in
    front of / behind / inside the environment's splats): the fused frames (semantic image as a second accumulator of the
    scene's walk, Morton layout, block culling) equal two separate passes, bit for bit -- asynchronous slots included."""
    import torch
    from pegasus_amd import frames as F
    rng = np.random.default_rng(300 + seed)
    k_obj = int(rng.integers(1, 6))
    n_env, n_obj = int(rng.choice([500, 4000, 30000])), int(rng.choice([200, 2000, 9000]))
    W, H = int(rng.choice([96, 200, 333])), int(rng.choice([64, 150, 240]))
    cloud, views, _ = scenes.merged_scene(40 + seed, n_env, k_obj, n_obj, 5, W, H, plane_size=float(rng.choice([0.6, 2.0])))
    act = cloud.activated()
    if seed % 2:                                  # translucent environment: semantic and scene walks diverge late
        act["opacities"][cloud.object_id == 0] *= np.float32(0.15)
    if seed % 3 == 0:                             # fat object splats
        act["scales"][cloud.object_id > 0] *= np.float32(6.0)
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device, bg=tuple(rng.random(3).astype(np.float32)))
    specs = [fr.view_spec(v) for v in views]
    ref = {k: v.clone() for k, v in fr.render_batch(specs).items()}
    torch.cuda.synchronize()
    fa, fb = fr.alloc_frames(3, H, W), fr.alloc_frames(2, H, W)
    ha = fr.render_frames_async(specs[:3], fa, slot=0)
    hb = fr.render_frames_async(specs[3:], fb, slot=1)
    ha.wait(); hb.wait()
    for k in ("color", "depth", "seg", "seg_depth", "masks"):
        got = torch.cat([fa[k], fb[k]])
        assert torch.equal(got, ref[k]), (seed, k)
    assert int(ref["masks"].sum()) > 0


def test_fused_semantic_with_aux_outputs(gpu_device):
    """The AUX + FUSED compositor instance (n_contrib / final_T together with the semantic image): scene outputs equal the
    plain AUX render, semantic outputs equal the fused frames path -- including quarters that take the plain loop because
    their tile holds no object entry."""
    import torch
    from pegasus_amd import frames as F, rasterizer as R
    cloud, views, _ = scenes.merged_scene(77, 20000, 3, 3000, 3, 200, 150, plane_size=2.0)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device, bg=(0.2, 0.1, 0.3))
    specs = [fr.view_spec(v) for v in views]
    kw = dict(shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3, tie_index=fr.tie_index, want_radii=False)
    plain = R.forward_views(fr.means3d, fr.opacities, specs, want_aux=True, **kw)
    plain = [{k: v.clone() for k, v in r.items() if v is not None} for r in plain]
    both = R.forward_views(fr.means3d, fr.opacities, specs, want_aux=True, semantic=fr.semantic, **kw)
    frames = fr.render_frames(specs)
    torch.cuda.synchronize()
    for i in range(len(specs)):
        for k in ("color", "depth", "final_T", "n_contrib"):
            assert torch.equal(both[i][k], plain[i][k]), (i, k)
        assert torch.equal(both[i]["sem_color"], frames["seg"][i]) and torch.equal(both[i]["sem_depth"], frames["seg_depth"][i])
    assert float(frames["seg"].std()) > 0


@pytest.mark.parametrize("seed", range(6))
def test_random_merged_scenes_layered_silhouettes_equal_per_object_passes(gpu_device, seed):
    """The layered silhouette pass (per-(tile, object) lists, masks from the compositor's epilogue, empty lists pre-filled) on
    the same random merged scenes: bit-equal to one single-object pass + mask launch per object -- any object count, object
    splats fat enough to cover many tiles, translucent objects (edge quarters that never saturate), random backgrounds."""
    import torch
    from pegasus_amd import frames as F
    rng = np.random.default_rng(900 + seed)
    k_obj = int(rng.integers(1, 10))
    n_env, n_obj = int(rng.choice([300, 3000])), int(rng.choice([50, 700, 5000]))
    W, H = int(rng.choice([96, 200, 333])), int(rng.choice([64, 150, 240]))
    cloud, views, _ = scenes.merged_scene(140 + seed, n_env, k_obj, n_obj, 4, W, H, plane_size=float(rng.choice([0.6, 2.0])))
    act = cloud.activated()
    if seed % 2:
        act["opacities"][cloud.object_id > 0] *= np.float32(0.2)
    if seed % 3 == 0:
        act["scales"][cloud.object_id > 0] *= np.float32(5.0)
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device, bg=tuple(rng.random(3).astype(np.float32)))
    specs = [fr.view_spec(v) for v in views]
    a = fr.render_silhouettes(specs).clone()
    b = fr.render_silhouettes_per_object(specs).clone()
    torch.cuda.synchronize()
    assert a.shape == (4, k_obj, H, W) and torch.equal(a, b), seed
