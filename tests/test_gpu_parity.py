"""GPU parity: the HIP path (through the drop-in surface and the C ABI) against the CPU oracle on the
same seeded inputs.  Integer/index stages bit-exact; images within BASELINE.json's 1e-4."""
import math

import numpy as np
import pytest

from pegasus_amd import scenes

pytestmark = pytest.mark.gpu


def _run_both(oracle, cloud, view, gpu_device, sh_degree=3, bg=(0.0, 0.0, 0.0), **kw):
    from helpers import gpu_forward
    act = cloud.activated()
    # cull_mode=1: the oracle lists only the instances that can contribute, exactly as the HIP binning does;
    # tests/test_oracle_tight_lists.py proves that its images are bit-identical to the reference-list mode.
    o = oracle.forward(**act, sh_degree=sh_degree, **view.raster_kwargs(bg), num_threads=8,
                       scale_modifier=kw.get("scale_modifier", 1.0), cull_mode=1)
    g = gpu_forward(act, view, sh_degree=sh_degree, bg=bg, device=str(gpu_device), **kw)
    return g, o


def _assert_same_ranges(gr, orr):
    """Same [start,end) for every non-empty tile; an empty tile may carry any start == end."""
    ge, oe = gr[:, 0] == gr[:, 1], orr[:, 0] == orr[:, 1]
    np.testing.assert_array_equal(ge, oe)
    np.testing.assert_array_equal(gr[~ge], orr[~oe])


def _check_all(g, o):
    from helpers import assert_images_match, assert_preprocess_bit_exact
    assert_preprocess_bit_exact(g, o)
    assert g["num_instances"] == o["num_instances"]
    # per-tile lists: same ranges, same Gaussian order, hence the same (tile, depth) key sequence
    _assert_same_ranges(g["ranges"], o["ranges"])
    np.testing.assert_array_equal(g["gauss_sorted"], o["gauss_sorted"])
    lens = (g["ranges"][:, 1] - g["ranges"][:, 0]).astype(np.int64)
    tile_of = np.repeat(np.arange(lens.size, dtype=np.uint64), lens)
    keys = (tile_of << np.uint64(32)) | g["depth"][g["gauss_sorted"]].view(np.uint32).astype(np.uint64)
    np.testing.assert_array_equal(keys, o["keys_sorted"])
    assert_images_match(g, o)


def test_c1_cube_256(oracle, gpu_device):
    cloud, views = scenes.scene_c1()
    g, o = _run_both(oracle, cloud, views[0], gpu_device)
    _check_all(g, o)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_degrees_and_background(oracle, gpu_device, deg):
    cloud, views = scenes.scene_c1(seed=10 + deg, n=3000)
    g, o = _run_both(oracle, cloud, views[0], gpu_device, sh_degree=deg, bg=(0.25, 0.5, 1.0))
    _check_all(g, o)


def test_c2_object_800(oracle, gpu_device):
    cloud, views = scenes.scene_c2(n=150_000, n_views=3)
    for v in views:
        g, o = _run_both(oracle, cloud, v, gpu_device)
        _check_all(g, o)


def test_c3_merged_scene_scaled(oracle, gpu_device):
    cloud, views = scenes.scene_c3(scale=0.1, n_views=2)
    for v in views:
        g, o = _run_both(oracle, cloud, v, gpu_device)
        _check_all(g, o)


def test_ragged_image_size(oracle, gpu_device):
    """Width/height not multiples of 16 (PEGASUS default is 640x480; also an odd size)."""
    cloud, _ = scenes.scene_c1(seed=5, n=5000)
    fov = math.radians(50)
    from pegasus_amd import graphics as G
    R, t = G.look_at_opencv((0.3, -0.2, -3.0), (0, 0, 0), up=(0, -1, 0))
    for (w, h) in ((640, 480), (333, 211)):
        v = scenes.make_view(R, t, w, h, fovx=fov, fovy=fov * h / w)
        g, o = _run_both(oracle, cloud, v, gpu_device)
        _check_all(g, o)


def test_scale_modifier(oracle, gpu_device):
    cloud, views = scenes.scene_c1(seed=6, n=2000)
    g, o = _run_both(oracle, cloud, views[0], gpu_device, scale_modifier=1.7)
    _check_all(g, o)


def test_all_culled_gives_background(oracle, gpu_device):
    """Every Gaussian behind the near plane: image = bg, depth 0, radii 0."""
    cloud, views = scenes.scene_c1(seed=7, n=500)
    cloud.xyz[:, 2] -= 10.0   # camera sits at z=-3 looking +z
    g, o = _run_both(oracle, cloud, views[0], gpu_device, bg=(0.1, 0.2, 0.3))
    assert (g["radii"] == 0).all() and g["num_instances"] == 0
    np.testing.assert_array_equal(g["color"], o["color"])
    assert np.allclose(g["color"][1], 0.2) and (g["out_depth"] == 0).all()


def test_empty_scene_stays_zero(gpu_device):
    import torch
    from pegasus_amd import diff_gaussian_rasterization as dgr
    dev = gpu_device
    s = dgr.GaussianRasterizationSettings(64, 64, 0.5, 0.5, torch.ones(3, device=dev), 1.0,
                                          torch.eye(4, device=dev), torch.eye(4, device=dev), 0,
                                          torch.zeros(3, device=dev), False, False)
    color, radii, depth = dgr.GaussianRasterizer(s)(
        torch.zeros((0, 3), device=dev), None, torch.zeros((0, 1), device=dev),
        shs=torch.zeros((0, 16, 3), device=dev), scales=torch.zeros((0, 3), device=dev),
        rotations=torch.zeros((0, 4), device=dev))
    assert color.shape == (3, 64, 64) and depth.shape == (1, 64, 64) and radii.shape == (0,)
    assert float(color.abs().max()) == 0.0 and float(depth.abs().max()) == 0.0


def test_precomputed_colour_and_cov(oracle, gpu_device):
    from helpers import gpu_forward, assert_images_match
    cloud, views = scenes.scene_c1(seed=8, n=4000)
    act = cloud.activated()
    v = views[0]
    base = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), stage="preprocess")
    colors = np.random.default_rng(0).uniform(0, 1, size=(cloud.n, 3)).astype(np.float32)
    kw = dict(means3d=act["means3d"], opacities=act["opacities"], cov3d_precomp=base["cov3d"],
              colors_precomp=colors)
    o = oracle.forward(**kw, **v.raster_kwargs(), num_threads=4, cull_mode=1)
    g = gpu_forward(act, v, colors_precomp=colors, cov3d_precomp=base["cov3d"], device=str(gpu_device))
    np.testing.assert_array_equal(g["radii"], o["radii"])
    np.testing.assert_array_equal(g["gauss_sorted"], o["gauss_sorted"])
    assert_images_match(g, o)


def test_mark_visible(oracle, gpu_device):
    import torch
    from pegasus_amd import diff_gaussian_rasterization as dgr
    cloud, views = scenes.scene_c1(seed=9, n=4096 + 17)
    cloud.xyz[::3, 2] -= 3.5
    v = views[0]
    dev = gpu_device
    s = dgr.GaussianRasterizationSettings(v.height, v.width, v.tanfovx, v.tanfovy, torch.zeros(3, device=dev), 1.0,
                                          torch.from_numpy(v.world_view_transform).to(dev),
                                          torch.from_numpy(v.full_proj_transform).to(dev), 3,
                                          torch.from_numpy(v.camera_center).to(dev), False, False)
    vis = dgr.GaussianRasterizer(s).markVisible(torch.from_numpy(cloud.xyz).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(vis, oracle.mark_visible(cloud.xyz, v.world_view_transform))
    assert 0 < vis.sum() < vis.size


def test_masks_and_quantisation(oracle, gpu_device):
    import ctypes as C
    import torch
    from pegasus_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    H, W, K = 120, 200, 5
    colors = rng.uniform(0, 1, size=(K, 3)).astype(np.float32)
    img = rng.uniform(0, 1.3, size=(3, H, W)).astype(np.float32)
    for k in range(K):   # paint patches near each colour, some just inside / outside the 0.1 ball
        img[:, 10 * k:10 * k + 8, :50] = colors[k][:, None, None] + rng.normal(0, 0.05, size=(3, 8, 50))
    depth = rng.uniform(0, 70, size=(H, W)).astype(np.float32)
    dev = gpu_device
    timg, tdepth, tcol = (torch.from_numpy(a).to(dev) for a in (img, depth, colors))
    masks = torch.zeros((K, H, W), dtype=torch.uint8, device=dev)
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    mm = torch.zeros((H, W), dtype=torch.int16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.pgr_color_masks(P(timg), 1, W, H, P(tcol), K, 0.1, P(masks), st))
    _lib.check(L.pgr_quantize_frame(P(timg), P(tdepth), W, H, P(rgb8), P(mm), st))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(masks.cpu().numpy(), oracle.color_masks(img, colors, 0.1))
    o_rgb, o_mm = oracle.quantize(img, depth)
    np.testing.assert_array_equal(rgb8.cpu().numpy(), o_rgb)
    np.testing.assert_array_equal(mm.cpu().numpy().view(np.uint16), o_mm)
    # and against numpy's own casts, which is what the reference executes (pegasus.py:347,355)
    np.testing.assert_array_equal(o_rgb, (img.transpose(1, 2, 0) * 255).astype(np.uint8))
    np.testing.assert_array_equal(o_mm, (depth * 1000).astype(np.uint16))


def test_batch_matches_single_view_and_oracle(oracle, gpu_device):
    """pgr_forward_batch: every view of a batch equals the single-view result bit for bit, and the oracle."""
    import torch
    from helpers import assert_images_match, fetch_workspace, gpu_forward
    from pegasus_amd import rasterizer
    cloud, views = scenes.scene_c3(scale=0.05, n_views=5)
    act = cloud.activated()
    dev = gpu_device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bgs = [(0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.2, 0.4, 0.6), (0.0, 0.0, 0.0), (0.5, 0.5, 0.5)]
    specs = [rasterizer.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.asarray(bg, np.float32)),
                                 t(v.world_view_transform), t(v.full_proj_transform), t(v.camera_center))
             for v, bg in zip(views, bgs)]
    res = rasterizer.forward_views(t(act["means3d"]), t(act["opacities"]), specs, shs=t(act["shs"]),
                                   scales=t(act["scales"]), rotations=t(act["rotations"]), sh_degree=3,
                                   want_radii=True, want_aux=True)
    torch.cuda.synchronize()
    info = rasterizer.last_forward_info()
    batch = []
    for k, (r, v) in enumerate(zip(res, views)):
        d = dict(color=r["color"].cpu().numpy(), out_depth=r["depth"].cpu().numpy(), radii=r["radii"].cpu().numpy(),
                 final_T=r["final_T"].cpu().numpy(), n_contrib=r["n_contrib"].cpu().numpy().astype(np.uint32),
                 num_instances=info["num_instances"][k])
        d.update(fetch_workspace(k, cloud.n, v.width, v.height))
        batch.append(d)
    # same batch without radii / aux outputs (the throughput configuration): identical frames
    res2 = rasterizer.forward_views(t(act["means3d"]), t(act["opacities"]), specs, shs=t(act["shs"]),
                                    scales=t(act["scales"]), rotations=t(act["rotations"]), sh_degree=3,
                                    want_radii=False)
    torch.cuda.synchronize()
    for k in range(len(views)):
        assert res2[k]["radii"] is None
        np.testing.assert_array_equal(res2[k]["color"].cpu().numpy(), batch[k]["color"])
        np.testing.assert_array_equal(res2[k]["depth"].cpu().numpy(), batch[k]["out_depth"])
    for k, (v, bg) in enumerate(zip(views, bgs)):
        single = gpu_forward(act, v, bg=bg, device=str(dev))
        for key in ("color", "out_depth", "radii", "final_T", "n_contrib", "gauss_sorted"):
            np.testing.assert_array_equal(batch[k][key], single[key], err_msg=f"view {k} {key}")
        o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(bg), num_threads=8, cull_mode=1)
        np.testing.assert_array_equal(batch[k]["gauss_sorted"], o["gauss_sorted"])
        assert_images_match(batch[k], o)


@pytest.mark.parametrize("n,spread", [(16000, 0.02), (90000, 0.05), (40000, 0.004)])
def test_long_tile_lists(oracle, gpu_device, n, spread):
    """Lists beyond the small LDS sort tier (4096), beyond the large tier (16384) and with many equal
    depths: a dense clump of small, faint splats in front of the camera."""
    rng = np.random.default_rng(n)
    cloud, views = scenes.scene_c1(seed=3, n=n)
    cloud.xyz[:] = rng.normal(0, spread, size=(n, 3)).astype(np.float32)
    cloud.xyz[: n // 3, 2] = np.float32(0.125)            # thousands of exactly equal depths -> index tie-break
    cloud.scaling[:] = np.log(0.004).astype(np.float32)
    cloud.opacity[:] = rng.normal(-4.0, 0.5, size=(n, 1)).astype(np.float32)   # faint: no early saturation
    g, o = _run_both(oracle, cloud, views[0], gpu_device)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    assert lens.max() > (16384 if n >= 40000 else 4096), lens.max()
    _check_all(g, o)


def test_frame_renderer_rgb_depth_masks(oracle, gpu_device):
    """FrameRenderer = PEGASUS's per-frame outputs: scene RGB+depth, object-only semantic image, K masks."""
    import torch
    from helpers import assert_images_match
    from pegasus_amd import frames as F, masks as M
    cloud, views = scenes.scene_c3(scale=0.04, n_views=3, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                         cloud.object_id, sh_degree=3, device=gpu_device)
    assert fr.K == 8 and fr.n_env == int((cloud.object_id == 0).sum())
    out = fr.render_batch([fr.view_spec(v) for v in views])
    torch.cuda.synchronize()
    n_env = fr.n_env
    sem_shs = fr.sem_shs.cpu().numpy()
    assert fr.order is not None               # resident copy is in Morton order per object: same for the oracle
    act = {k: np.ascontiguousarray(a[fr.order]) for k, a in act.items()}
    for i, v in enumerate(views):
        o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, tie_index=fr.order)
        g = dict(color=out["color"][i].cpu().numpy(), out_depth=out["depth"][i].cpu().numpy())
        amb = o["ambig"].astype(bool)
        assert np.abs(g["color"] - o["color"])[:, ~amb].max() <= 1e-4
        assert np.abs(g["out_depth"] - o["out_depth"])[:, ~amb].max() <= 1e-4
        so = oracle.forward(act["means3d"][n_env:], act["opacities"][n_env:], scales=act["scales"][n_env:],
                            rotations=act["rotations"][n_env:], shs=sem_shs, sh_degree=0, **v.raster_kwargs(),
                            num_threads=8, tie_index=fr.order[n_env:] - n_env)
        samb = so["ambig"].astype(bool)
        seg = out["seg"][i].cpu().numpy()
        assert np.abs(seg - so["color"])[:, ~samb].max() <= 1e-4
        # masks: exact wherever the colour distance is not within 1e-4 of the 0.1 threshold
        om = oracle.color_masks(so["color"], fr.colors_np, 0.1)
        gm = out["masks"][i].cpu().numpy()
        dist = np.stack([np.linalg.norm(so["color"].transpose(1, 2, 0) - c, axis=2) for c in fr.colors_np])
        decided = (np.abs(dist - 0.1) > 2e-4) & ~samb[None]
        np.testing.assert_array_equal(gm[decided], om[decided])
        assert decided.mean() > 0.999
        assert gm.sum() > 0      # the objects are visible
        # and the device mask kernel is bit-exact against the oracle on the SAME image
        np.testing.assert_array_equal(M.color_masks(out["seg"][i], fr.colors).cpu().numpy(),
                                      oracle.color_masks(seg, fr.colors_np, 0.1))


def test_async_frame_pipeline_matches_sync(gpu_device):
    """render_batch_async (two streams, two workspaces, no host sync until wait) == render_batch, bit for bit,
    including when the first attempt overflows its instance capacity."""
    import torch
    from pegasus_amd import frames as F, rasterizer
    cloud, views = scenes.scene_c3(scale=0.04, n_views=6, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                         cloud.object_id, sh_degree=3, device=gpu_device)
    specs = [fr.view_spec(v) for v in views]
    ref = {k: v.clone() for k, v in fr.render_batch(specs).items()}
    torch.cuda.synchronize()
    fa, fb = fr.alloc_frames(3, 240, 320), fr.alloc_frames(3, 240, 320)
    h0 = fr.render_batch_async(specs[:3], fa, slot=0)
    h1 = fr.render_batch_async(specs[3:], fb, slot=1)
    h0.wait(); h1.wait()
    for k in ("color", "depth", "seg", "masks"):
        assert torch.equal(fa[k], ref[k][:3]), k
        assert torch.equal(fb[k], ref[k][3:]), k
    # force an overflow on the async path: shrink the capacity hint, the handle must transparently re-render
    rasterizer.reset_capacity(1000)
    fc = fr.alloc_frames(3, 240, 320)
    fr.render_batch_async(specs[:3], fc, slot=0).wait()
    torch.cuda.synchronize()
    for k in ("color", "depth", "seg", "masks"):
        assert torch.equal(fc[k], ref[k][:3]), k


def test_fused_semantic_pass_is_bit_identical(gpu_device):
    """The semantic image composited from the SCENE's lists (environment entries dropped on the fly) equals the
    image of a separate objects-only render bit for bit -- with and without an instance overflow on the way."""
    import torch
    from pegasus_amd import frames as F, rasterizer
    cloud, views = scenes.scene_c3(scale=0.04, n_views=5, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                         cloud.object_id, sh_degree=3, device=gpu_device, bg=(0.1, 0.0, 0.2))
    specs = [fr.view_spec(v) for v in views]
    ref = {k: v.clone() for k, v in fr.render_batch(specs).items()}          # two separate passes
    torch.cuda.synchronize()
    f = fr.alloc_frames(5, 240, 320)
    fr.render_frames_async(specs, f, slot=0).wait()
    for k in ("color", "depth", "seg", "seg_depth", "masks"):
        assert torch.equal(f[k], ref[k]), k
    assert int(f["masks"].sum()) > 0
    rasterizer.reset_capacity(2000)
    f2 = fr.alloc_frames(5, 240, 320)
    fr.render_frames_async(specs, f2, slot=1).wait()
    torch.cuda.synchronize()
    for k in ("color", "depth", "seg", "masks"):
        assert torch.equal(f2[k], ref[k]), k


def test_count_walk_verdicts_every_mode(oracle, gpu_device, monkeypatch):
    """The scatter walk takes the count walk's verdicts (one bit per candidate tile) where a group has a region and
    evaluates the predicate itself where it has none; the lists must not depend on which: (a) groups with more
    candidates than a region holds (huge splats) next to ordinary ones, (b) the same with the verdicts switched off,
    (c) an instance capacity so small that only some of the groups get a region."""
    import torch
    from pegasus_amd import rasterizer
    # (a) 20 k splats blown up 6x at 512x512: thousands of candidates per 64-Gaussian group for most groups, few for some
    cloud, views = scenes.scene_c1(seed=21, n=20_000)
    cloud.scaling[:10_000] += np.float32(np.log(6.0))
    v = scenes.make_view(views[0].R_c2w.T, views[0].t_w2c, 512, 512, fovx=views[0].fovx, fovy=views[0].fovy)
    g, o = _run_both(oracle, cloud, v, gpu_device)
    assert o["num_instances"] > 500_000
    _check_all(g, o)
    # (b) verdicts off: the very same lists
    monkeypatch.setenv("PGR_BIN_RECORDS", "0")
    g0, _ = _run_both(oracle, cloud, v, gpu_device)
    monkeypatch.delenv("PGR_BIN_RECORDS")
    for k in ("gauss_sorted", "ranges", "color", "n_contrib"):
        np.testing.assert_array_equal(g0[k], g[k], err_msg=k)
    # (c) 200 k one-tile splats, most of them outside the view: fewer instances than Gaussians, and a capacity that just
    # holds them -> regions for about a third of the groups only
    rng = np.random.default_rng(5)
    n = 200_000
    cloud2 = scenes.box_object(rng, n, (8.0, 8.0, 1.0), math.log(0.004), 0.4, 0.1, object_id=1)
    g2, o2 = _run_both(oracle, cloud2, views[0], gpu_device)
    need = int(o2["num_instances"])
    assert 1000 < need < n // 3, need
    dev = torch.device(gpu_device)
    rasterizer.set_capacity_hint((dev, n, views[0].width, views[0].height), need + 64)
    rasterizer.drop_async_workspaces()
    g3, _ = _run_both(oracle, cloud2, views[0], gpu_device)
    assert rasterizer.last_forward_info()["used_max_instances"] == need + 64
    assert (need + 64) * 12 // 768 < (n + 63) // 64        # fewer regions than groups
    _check_all(g2, o2)
    _check_all(g3, o2)


def test_large_image_many_tiles(oracle, gpu_device):
    """More than 16384 tiles: the binning kernels sweep the tile space in several LDS passes."""
    cloud, _ = scenes.scene_c1(seed=12, n=6000)
    cloud.scaling[:] += np.float32(np.log(1.5))
    from pegasus_amd import graphics as G
    R, t = G.look_at_opencv((0.2, -0.1, -2.2), (0, 0, 0), up=(0, -1, 0))
    w, h = 2405, 1817          # 151 x 114 = 17214 tiles, ragged on both axes
    v = scenes.make_view(R, t, w, h, fovx=math.radians(60), fovy=math.radians(60) * h / w)
    g, o = _run_both(oracle, cloud, v, gpu_device)
    assert o["ranges"].shape[0] > 16384 and o["num_instances"] > 50_000
    _check_all(g, o)


def test_single_gaussian_and_two_pixel_image(oracle, gpu_device):
    cloud, views = scenes.scene_c1(seed=13, n=1)
    cloud.xyz[:] = 0.0
    g, o = _run_both(oracle, cloud, views[0], gpu_device, bg=(0.3, 0.6, 0.9))
    _check_all(g, o)
    assert g["num_instances"] >= 1
    from pegasus_amd import graphics as G
    R, t = G.look_at_opencv((0.0, 0.0, -3.0), (0, 0, 0), up=(0, -1, 0))
    v = scenes.make_view(R, t, 2, 1, fovx=math.radians(50), fovy=math.radians(25))
    cloud2, _ = scenes.scene_c1(seed=14, n=300)
    g, o = _run_both(oracle, cloud2, v, gpu_device)
    _check_all(g, o)


@pytest.mark.gpu
def test_spatial_order_does_not_change_frames(gpu_device):
    """FrameRenderer stores the scene in Morton order per object (scene_order.py); lists are depth-ordered and exact
    depth ties are broken by the caller's index, so the frames are those of the input order, bit for bit."""
    import torch
    from pegasus_amd import scenes
    from pegasus_amd.frames import FrameRenderer
    cloud, views = scenes.scene_c3(scale=0.02, n_views=2)
    a = cloud.activated()
    out = []
    for so in (False, True):
        fr = FrameRenderer(a["means3d"], a["opacities"], a["scales"], a["rotations"], a["shs"], cloud.object_id,
                           device=gpu_device, spatial_order=so)
        assert (fr.order is not None) == so
        f = fr.render_frames([fr.view_spec(v) for v in views])
        out.append({k: v.clone() for k, v in f.items()})
    perm = fr.order
    assert sorted(perm.tolist()) == list(range(cloud.n)) and np.all(np.diff(cloud.object_id[perm]) >= 0)
    for k in ("color", "depth", "seg", "masks"):      # exact depth ties keep the caller's order (PgrScene.tie_index)
        assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("n,spread", [(120000, 0.05), (300000, 0.08)])
def test_very_long_lists_depth_partitioned(oracle, gpu_device, n, spread):
    """Lists of 20-100 k entries with smoothly spread depths: the sort partitions them by depth through the alt
    buffer and sorts every < 16384-key segment in LDS (tilebin.hip.h partition_sort_long)."""
    rng = np.random.default_rng(n)
    cloud, views = scenes.scene_c1(seed=4, n=n)
    cloud.xyz[:] = rng.normal(0, spread, size=(n, 3)).astype(np.float32)
    cloud.scaling[:] = np.log(0.004).astype(np.float32)
    cloud.opacity[:] = rng.normal(-4.0, 0.5, size=(n, 1)).astype(np.float32)   # faint: no early saturation
    g, o = _run_both(oracle, cloud, views[0], gpu_device)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    assert lens.max() > 2 * 16384, lens.max()
    _check_all(g, o)


@pytest.mark.parametrize("n,spread,equal", [(16000, 0.02, True), (90000, 0.05, True), (120000, 0.05, False)])
def test_tie_index_order_on_long_lists(oracle, gpu_device, n, spread, equal):
    """PgrScene.tie_index: exact depth ties (thousands of them here) follow an arbitrary caller permutation through
    every sort path (bucket sort with 12-byte entries, merge-sort fallback, depth-partitioned and L2-merged lists);
    lists bit-exact against the oracle with the same tie_index."""
    import torch
    from helpers import fetch_workspace
    from pegasus_amd import rasterizer as R
    rng = np.random.default_rng(n + 1)
    cloud, views = scenes.scene_c1(seed=3, n=n)
    cloud.xyz[:] = rng.normal(0, spread, size=(n, 3)).astype(np.float32)
    if equal:
        cloud.xyz[: n // 3, 2] = np.float32(0.125)
    cloud.scaling[:] = np.log(0.004).astype(np.float32)
    cloud.opacity[:] = rng.normal(-4.0, 0.5, size=(n, 1)).astype(np.float32)
    tie = rng.permutation(n).astype(np.int32)
    act, v = cloud.activated(), views[0]
    o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1, tie_index=tie)
    o_pos = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
    if equal:
        assert not np.array_equal(o["gauss_sorted"], o_pos["gauss_sorted"])      # the tie order does matter here
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device, dt)
    spec = R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.zeros(3, np.float32)), t(v.world_view_transform),
                      t(v.full_proj_transform), t(v.camera_center))
    res = R.forward_views(t(act["means3d"]), t(act["opacities"]), [spec], shs=t(act["shs"]), scales=t(act["scales"]),
                          rotations=t(act["rotations"]), sh_degree=3, want_radii=True, want_aux=True,
                          tie_index=t(tie, torch.int32))
    torch.cuda.synchronize()
    w = fetch_workspace(0, n, v.width, v.height)
    np.testing.assert_array_equal(w["gauss_sorted"], o["gauss_sorted"])
    amb = o["ambig"].astype(bool)
    assert np.abs(res[0]["color"].cpu().numpy() - o["color"])[:, ~amb].max() <= 1e-4


@pytest.mark.parametrize("n,pairs", [(9000, 300), (12000, 40), (15800, 1500)])
def test_windowed_sort_with_scattered_depth_ties(oracle, gpu_device, n, pairs):
    """The windowed sort (8193..15 872 keys, tilebin.hip.h window_sort_tile) in a three-view batch with `pairs` groups of two
    or three Gaussians at exactly equal depth spread over the list (a few keys per bucket share their depth: the in-bucket
    tie-break by the caller's tie_index runs, the list is NOT rejected) -- positions and tie indices both shuffled."""
    import torch
    from helpers import fetch_workspace
    from pegasus_amd import rasterizer as R
    rng = np.random.default_rng(n + pairs)
    cloud, views = scenes.scene_c1(seed=9, n=n)
    cloud.xyz[:, 0] = np.float32(0.0929) + rng.normal(0, 5e-4, n).astype(np.float32)
    cloud.xyz[:, 1] = np.float32(0.0929) + rng.normal(0, 5e-4, n).astype(np.float32)
    cloud.xyz[:, 2] = rng.normal(0, 0.2, n).astype(np.float32)
    src = rng.choice(n, 3 * pairs, replace=False)
    cloud.xyz[src[pairs:2 * pairs], 2] = cloud.xyz[src[:pairs], 2]
    cloud.xyz[src[2 * pairs::2], 2] = cloud.xyz[src[:pairs:2][: len(src[2 * pairs::2])], 2]
    cloud.xyz[src, 0] = np.float32(0.0929); cloud.xyz[src, 1] = np.float32(0.0929)            # same ray: same view depth
    cloud.scaling[:] = np.log(0.0005).astype(np.float32)
    cloud.opacity[:] = rng.normal(-3.0, 0.5, size=(n, 1)).astype(np.float32)
    tie = rng.permutation(n).astype(np.int32)
    act, v = cloud.activated(), views[0]
    o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1, tie_index=tie)
    o_pos = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    assert lens.max() == n, lens.max()
    assert not np.array_equal(o["gauss_sorted"], o_pos["gauss_sorted"])                      # the ties are real
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device, dt)
    spec = R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.zeros(3, np.float32)), t(v.world_view_transform),
                      t(v.full_proj_transform), t(v.camera_center))
    for tie_arg, want in ((t(tie, torch.int32), o), (None, o_pos)):
        R.forward_views(t(act["means3d"]), t(act["opacities"]), [spec] * 3, shs=t(act["shs"]), scales=t(act["scales"]),
                        rotations=t(act["rotations"]), sh_degree=3, want_radii=True, want_aux=True, tie_index=tie_arg)
        torch.cuda.synchronize()
        for k in range(3):
            w = fetch_workspace(k, n, v.width, v.height)
            np.testing.assert_array_equal(w["gauss_sorted"], want["gauss_sorted"], err_msg=f"view {k}")


@pytest.mark.parametrize("degree,stride", [(0, 16), (1, 16), (2, 16), (3, 16), (1, 4), (2, 9), (3, 17), (0, 1)])
def test_split_sh_layout_equals_the_concatenated_one(oracle, gpu_device, degree, stride):
    """PgrScene::shs_rest: the SH coefficients as the model stores them (_features_dc [N,1,3] + _features_rest [N,K-1,3],
    /root/reference/src/gs/gaussian_model.py:118-121) instead of get_features' torch.cat -- per-Gaussian colours bit-equal to
    the concatenated layout's and to the oracle's, for every active degree and for strides that are not 16 (12-byte loads at
    4-byte alignment); one view through the early-status call and a batch of three through the synchronous one."""
    import torch
    from helpers import fetch_workspace
    from pegasus_amd import rasterizer as R
    cloud, (view,) = scenes.scene_c1(seed=21, n=6000)
    act = cloud.activated()
    rng = np.random.default_rng(5)
    shs = rng.normal(0, 0.4, size=(cloud.n, stride, 3)).astype(np.float32)
    shs[:, 0] += 0.8
    o = oracle.forward(**{**act, "shs": shs}, sh_degree=degree, **view.raster_kwargs(), num_threads=4, cull_mode=1)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu_device)
    T = {k: tt(a) for k, a in act.items() if k != "shs"}
    full, dc, rest = tt(shs), tt(shs[:, :1]), tt(shs[:, 1:])
    spec = R.ViewSpec(view.height, view.width, view.tanfovx, view.tanfovy, tt(np.zeros(3)), tt(view.world_view_transform),
                      tt(view.full_proj_transform), tt(view.camera_center))
    common = dict(scales=T["scales"], rotations=T["rotations"], sh_degree=degree, want_radii=True)
    a = R.forward_views(T["means3d"], T["opacities"], [spec] * 3, shs=full, **common)
    torch.cuda.synchronize()
    a = [{k: r[k].clone() for k in ("color", "depth", "radii")} for r in a]
    b = R.forward_views(T["means3d"], T["opacities"], [spec] * 3, shs=dc, shs_rest=rest, **common)
    torch.cuda.synchronize()
    w = fetch_workspace(1, cloud.n, view.width, view.height)
    m = o["radii"] > 0
    rec = w["rgb4"][m][:, :3]
    np.testing.assert_array_equal(rec.view(np.uint32), o["rgb"][m].view(np.uint32))
    for ra, rb in zip(a, b):
        for k in ("color", "depth", "radii"):
            assert torch.equal(ra[k], rb[k]), k
    pb = R.forward_views(T["means3d"], T["opacities"], [spec], shs=dc, shs_rest=rest, async_slot=("split-sh", 0),
                         early_status=True, **common)
    one = pb.wait()[0]
    for k in ("color", "depth", "radii"):
        assert torch.equal(one[k], a[0][k]), k


def test_every_tile_of_a_small_image_holds_a_long_list(oracle, gpu_device):
    """A 64x64 image (16 tiles) whose every tile lists 14-35 k pinpoint splats, three views per call: 36 of the batch's 48
    lists go through the split pre-pass (the launch that carries its workgroups is sized by the number of tiles -- here a
    small one -- not by a constant) and their segments through the 512 x 16 kernel, the 12 corner lists through the
    windowed sort -- lists bit-exact against the oracle."""
    import torch
    from helpers import fetch_workspace
    from pegasus_amd import graphics as G, rasterizer as R
    rng = np.random.default_rng(77)
    n = 320_000
    cloud, _ = scenes.scene_c1(seed=9, n=n)
    cloud.xyz[:, 0] = rng.uniform(-0.9, 0.9, n).astype(np.float32)
    cloud.xyz[:, 1] = rng.uniform(-0.9, 0.9, n).astype(np.float32)
    cloud.xyz[:, 2] = rng.normal(0, 0.3, n).astype(np.float32)
    cloud.scaling[:] = np.log(0.0005).astype(np.float32)
    cloud.opacity[:] = rng.normal(-3.0, 0.5, size=(n, 1)).astype(np.float32)
    Rm, t = G.look_at_opencv((0.0, 0.0, -3.0), (0, 0, 0), up=(0, -1, 0))
    v = scenes.make_view(Rm, t, 64, 64, fovx=math.radians(40), fovy=math.radians(40))
    act = cloud.activated()
    o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    assert lens.size == 16 and lens.min() > 8192 and (lens > 15872).sum() == 12, lens
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu_device)
    spec = R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, tt(np.zeros(3)), tt(v.world_view_transform),
                      tt(v.full_proj_transform), tt(v.camera_center))
    T = {k: tt(a) for k, a in act.items()}
    R.forward_views(T["means3d"], T["opacities"], [spec] * 3, shs=T["shs"], scales=T["scales"], rotations=T["rotations"],
                    sh_degree=3, want_radii=True, want_aux=True)
    torch.cuda.synchronize()
    for k in range(3):
        w = fetch_workspace(k, n, v.width, v.height)
        np.testing.assert_array_equal(w["gauss_sorted"], o["gauss_sorted"], err_msg=f"view {k}")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hostile_inputs_do_not_fault_and_match_oracle(oracle, gpu_device, seed):
    """NaN / Inf positions, huge and vanishing scales, opacity exactly 0 and 1, zero and unnormalised quaternions,
    splats behind / on the near plane: no fault, no hang, the same radii and lists as the oracle, equal images
    wherever the oracle's are finite."""
    from helpers import gpu_forward
    rng = np.random.default_rng(900 + seed)
    n = 3000
    cloud, views = scenes.scene_c1(seed=30 + seed, n=n)
    v = scenes.make_view(views[0].R_c2w.T, views[0].t_w2c, 160, 112, fovx=views[0].fovx, fovy=views[0].fovy)
    act = cloud.activated()
    pick = lambda frac: rng.random(n) < frac
    act["means3d"][pick(0.01)] = np.nan
    act["means3d"][pick(0.01), 0] = np.inf
    act["means3d"][pick(0.01), 2] = -np.inf
    act["means3d"][pick(0.02), 2] = np.float32(-3.0 + 0.2)            # exactly on the near plane (camera at z = -3)
    act["scales"][pick(0.02)] = np.float32(1e3)
    act["scales"][pick(0.02)] = np.float32(1e-8)
    act["scales"][pick(0.01), 1] = 0.0
    act["opacities"][pick(0.03)] = 0.0
    act["opacities"][pick(0.03)] = 1.0
    act["rotations"][pick(0.01)] = 0.0
    act["rotations"][pick(0.02)] *= np.float32(37.0)
    act["shs"][pick(0.01)] = np.float32(1e6)
    o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs((0.3, 0.2, 0.1)), num_threads=8, cull_mode=1)
    g = gpu_forward(act, v, sh_degree=3, bg=(0.3, 0.2, 0.1), device=str(gpu_device))
    np.testing.assert_array_equal(g["radii"], o["radii"])
    assert g["num_instances"] == o["num_instances"]
    np.testing.assert_array_equal(g["gauss_sorted"], o["gauss_sorted"])
    ok = np.isfinite(o["color"]).all(axis=0) & np.isfinite(o["out_depth"][0]) & ~o["ambig"].astype(bool)
    assert ok.mean() > 0.5
    scale = np.maximum(1.0, np.abs(o["color"]))
    assert (np.abs(g["color"] - o["color"]) / scale)[:, ok].max() <= 1e-4
    assert np.array_equal(np.isfinite(g["color"]).all(axis=0)[~o["ambig"].astype(bool)],
                          np.isfinite(o["color"]).all(axis=0)[~o["ambig"].astype(bool)])


def test_frame_renderer_silhouette_masks(oracle, gpu_device):
    """FrameRenderer.render_silhouettes: object k alone in its semantic colour, thresholded (/root/reference/src/gs/
    render.py:36-65), against the oracle's single-object renders; a silhouette contains the object's visible mask."""
    import torch
    from pegasus_amd import frames as F
    from pegasus_amd.sh_utils import RGB2SH
    cloud, views = scenes.scene_c3(scale=0.03, n_views=3, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device)
    specs = [fr.view_spec(v) for v in views]
    sil = fr.render_silhouettes(specs).cpu().numpy()
    vis = fr.render_frames(specs)["masks"].cpu().numpy()
    assert sil.shape == vis.shape == (3, fr.K, 240, 320)
    # occluded parts are silhouette only (the rest: edge pixels whose colour sits at the threshold in one of the two renders)
    assert ((sil | vis) == sil).mean() > 0.999 and sil.sum() > vis.sum()
    oid = cloud.object_id
    for k in (1, fr.K):
        m = oid == k
        shs = np.zeros((int(m.sum()), 16, 3), np.float32)
        shs[:, 0] = RGB2SH(fr.colors_np[k - 1])
        for vi, v in enumerate(views):
            o = oracle.forward(act["means3d"][m], act["opacities"][m], scales=act["scales"][m], rotations=act["rotations"][m],
                               shs=shs, sh_degree=0, num_threads=8, **v.raster_kwargs())
            dist = np.linalg.norm(o["color"].transpose(1, 2, 0) - fr.colors_np[k - 1], axis=2)
            decided = (np.abs(dist - 0.1) > 2e-4) & ~o["ambig"].astype(bool)
            np.testing.assert_array_equal(sil[vi, k - 1][decided], (dist <= 0.1)[decided].astype(np.uint8))


def test_pack_frames_matches_numpy_casts_and_bit_planes(gpu_device):
    """pgr_pack_frames (what leaves the GPU per finished batch): the reference's two numpy casts
    (/root/reference/pegasus.py:347,355) for a whole batch, and K masks as bit planes; K > 8 spills into a second byte."""
    import torch
    from pegasus_amd import masks as M
    rng = np.random.default_rng(11)
    for (B, K, H, W) in ((3, 8, 37, 53), (2, 11, 16, 16), (1, 1, 5, 7)):
        color = rng.uniform(-0.2, 1.4, size=(B, 3, H, W)).astype(np.float32)
        depth = rng.uniform(0, 70, size=(B, 1, H, W)).astype(np.float32)
        masks = (rng.uniform(size=(B, K, H, W)) < 0.3).astype(np.uint8)
        t = lambda a: torch.from_numpy(a).to(gpu_device)
        out = M.pack_frames(t(color), t(depth), t(masks))
        torch.cuda.synchronize()
        with np.errstate(invalid="ignore"):
            want_rgb = (color.transpose(0, 2, 3, 1) * 255).astype(np.int64).astype(np.uint8)      # wraps like the kernel
        np.testing.assert_array_equal(out["rgb"].cpu().numpy(), want_rgb)
        np.testing.assert_array_equal(out["depth_mm"].cpu().numpy().view(np.uint16), (depth[:, 0] * 1000).astype(np.uint16))
        bits = out["mask_bits"].cpu().numpy()
        assert bits.shape == (B, H, W, (K + 7) // 8)
        for m in range(K):
            np.testing.assert_array_equal((bits[..., m // 8] >> (m % 8)) & 1, masks[:, m])
        np.testing.assert_array_equal(M.unpack_mask_bits(out["mask_bits"], K).cpu().numpy(), masks)
        # in-range colours: exactly numpy's uint8 cast
        inr = (color >= 0) & (color * 255 < 256)
        np.testing.assert_array_equal(out["rgb"].cpu().numpy()[inr.transpose(0, 2, 3, 1)],
                                      (color.transpose(0, 2, 3, 1) * 255).astype(np.uint8)[inr.transpose(0, 2, 3, 1)])
        only = M.pack_frames(depth=t(depth))
        assert set(only) == {"depth_mm"} and torch.equal(only["depth_mm"], out["depth_mm"])


def _last_blended(gauss_sorted, ranges, n_contrib, W, H):
    """Gaussian id of the last entry every pixel blended (or -1): list-position bookkeeping made comparable between the
    tight lists of the HIP path and reference-style lists."""
    gy, gx = (H + 15) // 16, (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 16) * gx + xs // 16
    pos = ranges[tile, 0].astype(np.int64) + n_contrib.astype(np.int64) - 1
    out = np.full((H, W), -1, np.int64)
    m = n_contrib > 0
    out[m] = gauss_sorted[pos[m]]
    return out


def test_c2_against_reference_style_lists(oracle, gpu_device):
    """The HIP path against the oracle run WITHOUT the tight-list predicate (cull_mode=0: every tile of the 3-sigma
    rectangle listed, as the reference's rasterizer emits them): radii equal, images within 1e-4, and every pixel's last
    blended Gaussian identical although the two list layouts differ."""
    from helpers import gpu_forward
    cloud, views = scenes.scene_c2(n=150_000, n_views=2)
    act = cloud.activated()
    v = views[1]
    g = gpu_forward(act, v, sh_degree=3, device=str(gpu_device))
    o0 = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=0)
    assert o0["num_instances"] > g["num_instances"]                  # reference-style lists are longer
    np.testing.assert_array_equal(g["radii"], o0["radii"])
    np.testing.assert_array_equal(g["tiles_touched"], o0["tiles_touched"])
    amb = o0["ambig"].astype(bool)
    assert amb.mean() <= 5e-4
    assert np.abs(g["color"] - o0["color"])[:, ~amb].max() <= 1e-4
    assert np.abs(g["out_depth"] - o0["out_depth"])[:, ~amb].max() <= 1e-4
    assert np.abs(g["final_T"] - o0["final_T"])[~amb].max() <= 1e-4
    lg = _last_blended(g["gauss_sorted"], g["ranges"], g["n_contrib"], v.width, v.height)
    lo = _last_blended(o0["gauss_sorted"], o0["ranges"], o0["n_contrib"], v.width, v.height)
    np.testing.assert_array_equal(lg[~amb], lo[~amb])
    assert (lg >= 0).mean() > 0.02


def test_hip_agrees_with_independent_dense_float64_renderer(gpu_device):
    """The HIP path against the second, independently written restatement directly (oracle/dense_ref.py: numpy float64, no
    tiles, no lists, every Gaussian against every pixel in global depth order, its own SH basis derived from Legendre
    recurrences) -- not through the C oracle: an error the oracle and the kernels shared (they were developed together)
    would show here."""
    from helpers import gpu_forward
    from oracle.dense_ref import dense_forward
    for seed, deg, bg in ((21, 3, (0.1, 0.3, 0.5)), (22, 1, (0.0, 0.0, 0.0))):
        cloud, views = scenes.scene_c1(seed=seed, n=1500)
        v = scenes.make_view(views[0].R_c2w.T, views[0].t_w2c, 96, 80, fovx=views[0].fovx, fovy=views[0].fovy)
        act = cloud.activated()
        g = gpu_forward(act, v, sh_degree=deg, bg=bg, device=str(gpu_device), fetch_intermediates=False)
        c64, d64 = dense_forward(sh_degree=deg, **{k: np.asarray(a, np.float64) for k, a in act.items()}, **v.raster_kwargs(bg))
        # float64 vs float32 arithmetic: differences are rounding only, except where a threshold decision flips
        dc, dd = np.abs(g["color"] - c64), np.abs(g["out_depth"][0] - d64)
        assert np.quantile(dc, 0.999) < 5e-5 and np.quantile(dd, 0.999) < 5e-5, (seed, dc.max(), dd.max())
        assert (dc > 1e-3).mean() < 1e-3 and (g["radii"] > 0).sum() > 500


def test_layered_silhouettes_equal_per_object_passes(gpu_device):
    """render_silhouettes (ONE layered pass: per-(tile, object) lists, masks from the compositor's epilogue) is bit-equal to
    K single-object passes + mask launches (round 2's form) -- static and posed, image sizes that are and are not a
    multiple of the tile size, a background close to one of the semantic colours, objects that straddle binning chunks."""
    import torch
    from pegasus_amd import compose, frames as F
    for (w, h, bg, scale) in ((320, 240, (0.0, 0.0, 0.0), 0.03), (333, 250, (0.0, 0.0, 0.0), 0.05)):
        cloud, views = scenes.scene_c3(scale=scale, n_views=4, width=w, height=h)
        act = cloud.activated()
        fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                             sh_degree=3, device=gpu_device, bg=bg)
        specs = [fr.view_spec(v) for v in views]
        a = fr.render_silhouettes(specs).clone()
        b = fr.render_silhouettes_per_object(specs).clone()
        torch.cuda.synchronize()
        assert a.shape == b.shape == (4, fr.K, h, w) and torch.equal(a, b)
        assert int(a.sum()) > 0 and all(int(a[:, k].sum()) > 0 for k in range(fr.K))
        # posed: every object moved per view
        rng = np.random.default_rng(5)
        oid = cloud.object_id
        centers = [act["means3d"][oid == k].astype(np.float64).mean(0) for k in range(1, fr.K + 1)]
        from scipy.spatial.transform import Rotation as Rot

        def rigid():
            T = np.eye(4)
            T[:3, :3] = Rot.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix()
            T[:3, 3] = rng.normal(0, 0.03, 3)
            return T
        poses = np.stack([compose.pose_table([(rigid(), centers[k]) for k in range(fr.K)]) for _ in views])
        ap = fr.render_silhouettes(specs, poses=poses).clone()
        bp = fr.render_silhouettes_per_object(specs, poses=poses).clone()
        torch.cuda.synchronize()
        assert torch.equal(ap, bp) and not torch.equal(ap, a)
    # a background that IS object 3's colour: empty pixels of layer 3 are inside its mask, in both forms
    cloud, views = scenes.scene_c3(scale=0.03, n_views=2, width=160, height=128)
    act = cloud.activated()
    from pegasus_amd import masks as M
    c3 = tuple(float(x) for x in M.generate_colors(8)[2])
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device, bg=c3)
    specs = [fr.view_spec(v) for v in views]
    a, b = fr.render_silhouettes(specs).clone(), fr.render_silhouettes_per_object(specs).clone()
    assert torch.equal(a, b) and float(a[:, 2].float().mean()) > 0.5 and float(a[:, 0].float().mean()) < 0.5


def test_layered_silhouettes_after_an_instance_overflow(gpu_device):
    import torch
    from pegasus_amd import frames as F, rasterizer
    cloud, views = scenes.scene_c3(scale=0.03, n_views=3, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device)
    specs = [fr.view_spec(v) for v in views]
    ref = fr.render_silhouettes(specs).clone()
    rasterizer.reset_capacity(1500)
    again = fr.render_silhouettes(specs).clone()
    torch.cuda.synchronize()
    assert torch.equal(ref, again)


def test_depth_mode_switch_matches_oracle(oracle, gpu_device):
    """PgrDepthMode (SURVEY.md 8a "keep it a one-line switch"): 0 = sum T alpha z (default), 1 = that sum over 1 - T_final.
    Both against the oracle's same switch; colour, final_T and n_contrib do not depend on it; the semantic depth follows."""
    import torch
    from pegasus_amd import frames as F, rasterizer as R
    cloud, views = scenes.scene_c3(scale=0.03, n_views=2, width=200, height=150)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device, bg=(0.2, 0.1, 0.0), spatial_order=False)
    res = {}
    for mode in (0, 1):
        specs = [fr.view_spec(v, depth_mode=mode) for v in views]
        res[mode] = R.forward_views(fr.means3d, fr.opacities, specs, shs=fr.shs, scales=fr.scales, rotations=fr.rotations,
                                    sh_degree=3, want_aux=True, semantic=fr.semantic)
        torch.cuda.synchronize()
        for i, v in enumerate(views):
            o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs((0.2, 0.1, 0.0)), num_threads=8, cull_mode=1, depth_mode=mode)
            ok = ~o["ambig"].astype(bool)
            g = res[mode][i]
            assert np.abs(g["depth"].cpu().numpy()[0] - o["out_depth"][0])[ok].max() <= 1e-4
            assert np.abs(g["color"].cpu().numpy() - o["color"])[:, ok].max() <= 1e-4
            np.testing.assert_array_equal(g["n_contrib"].cpu().numpy().astype(np.uint32)[ok], o["n_contrib"][ok])
    for i in range(len(views)):
        a, b = res[0][i], res[1][i]
        assert torch.equal(a["color"], b["color"]) and torch.equal(a["final_T"], b["final_T"]) and torch.equal(a["sem_color"], b["sem_color"])
        T = a["final_T"]
        want = torch.where(1.0 - T > 0, a["depth"][0] / (1.0 - T), torch.zeros_like(T))
        assert torch.equal(b["depth"][0], want)                       # the normalised image IS the default one over 1 - T
        assert float((b["depth"][0] - a["depth"][0]).abs().max()) > 1e-3
        assert not torch.equal(a["sem_depth"], b["sem_depth"])


def test_pack_records_is_pack_frames_in_one_buffer(gpu_device):
    """pgr_pack_records: the three images of pgr_pack_frames back to back in ONE record per frame (what one gather / one
    writer task moves), into a caller-provided buffer with a larger row stride; record_views() slices it back, zero-copy."""
    import torch
    from pegasus_amd import masks as M
    rng = np.random.default_rng(12)
    for (B, K, H, W) in ((3, 8, 37, 53), (2, 11, 16, 16), (1, 1, 5, 7), (2, 0, 9, 9)):
        t = lambda a: torch.from_numpy(a).to(gpu_device)
        color = t(rng.uniform(-0.2, 1.4, size=(B, 3, H, W)).astype(np.float32))
        depth = t(rng.uniform(0, 70, size=(B, 1, H, W)).astype(np.float32))
        masks = t((rng.uniform(size=(B, K, H, W)) < 0.3).astype(np.uint8)) if K else None
        want = M.pack_frames(color, depth, masks)
        lay = M.record_layout(H, W, K)
        rec = M.pack_records(color, depth, masks)
        assert rec.shape == (B, lay["bytes"]) and rec.dtype == torch.uint8
        big = torch.full((B + 1, lay["bytes"] + 32), 7, dtype=torch.uint8, device=gpu_device)
        assert M.pack_records(color, depth, masks, out=big) is big
        torch.cuda.synchronize()
        for r in (rec, big[:B]):
            v = M.record_views(r, H, W, K)
            assert torch.equal(v["rgb"], want["rgb"]) and torch.equal(v["depth_mm"], want["depth_mm"])
            if K:
                assert torch.equal(v["mask_bits"], want["mask_bits"])
            assert v["rgb"].data_ptr() == r.data_ptr()                    # views, not copies
        assert int((big[B] != 7).sum()) == 0 and int((big[:, lay["bytes"]:] != 7).sum()) == 0   # nothing written outside
        # a rank-major gathered buffer [world, B, bytes] slices the same way
        stacked = torch.stack([rec, rec])
        assert torch.equal(M.record_views(stacked, H, W, K)["depth_mm"][1], want["depth_mm"])


def test_scene_prepare_constants_match_the_per_call_path(gpu_device):
    """pgr_scene_prepare: the inverse tie permutation and the object-id byte table, computed once, give the frames the
    per-call path gives (which rebuilds them in the workspace every batch)."""
    import torch
    from pegasus_amd import frames as F, rasterizer as R
    cloud, views = scenes.scene_c3(scale=0.03, n_views=3, width=256, height=192)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device)
    assert fr.tie_inv is not None and fr.semantic["object_id_u8"] is not None
    inv = fr.tie_inv.long()
    assert torch.equal(inv[fr.tie_index.long()], torch.arange(fr.n, device=gpu_device))
    assert torch.equal(fr.semantic["object_id_u8"].int(), fr.semantic["object_id"][fr.n_env:])
    specs = [fr.view_spec(v) for v in views]
    a = {k: v.clone() for k, v in fr.render_frames(specs).items()}
    sem = {k: v for k, v in fr.semantic.items() if k != "object_id_u8"}
    f = fr.alloc_frames(3, 192, 256)
    outs = [dict(color=f["color"][i], depth=f["depth"][i], radii=None, sem_color=f["seg"][i], sem_depth=f["seg_depth"][i],
                 sem_masks=f["masks"][i]) for i in range(3)]
    R.forward_views(fr.means3d, fr.opacities, specs, shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3,
                    want_radii=False, outputs=outs, semantic=sem, tie_index=fr.tie_index)       # no tie_inv, no byte table
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], f[k]), k


@pytest.mark.parametrize("n_views,env_switch", [(1, None), (2, None), (5, None), (5, "PGR_BIN_RECORDS"), (5, "PGR_BLOCK_CULL")])
def test_layered_call_many_small_layers_and_unlayered_gaussians(gpu_device, monkeypatch, n_views, env_switch):
    """pgr_forward_layers_async through the C ABI on a scene the FrameRenderer never builds: Gaussians that belong to NO
    layer in front (layer id 0: dropped), then 37 small layers, so that binning chunks and 64-Gaussian groups straddle
    several layers; one- and two-view calls (their own launch shapes) and the A/B switches.  Reference: every layer rendered
    alone through the plain batch call + pgr_color_masks."""
    import torch
    from pegasus_amd import masks as M, rasterizer as R
    if env_switch:
        monkeypatch.setenv(env_switch, "0")
    rng = np.random.default_rng(21)
    K, per, n0 = 37, 333, 1500
    n = n0 + K * per
    xyz = np.concatenate([rng.normal(0, 0.25, (n0, 3)), np.concatenate([rng.normal(rng.uniform(-0.3, 0.3, 3), 0.05, (per, 3)) for _ in range(K)])]).astype(np.float32)
    layer = np.concatenate([np.zeros(n0, np.int32), np.repeat(np.arange(1, K + 1, dtype=np.int32), per)])
    scales = np.exp(rng.normal(np.log(0.01), 0.4, (n, 3))).astype(np.float32)
    rots = rng.normal(size=(n, 4)).astype(np.float32); rots /= np.linalg.norm(rots, axis=1, keepdims=True)
    opac = (1 / (1 + np.exp(-rng.normal(1.0, 1.5, n)))).astype(np.float32)
    colors_np = M.generate_colors(K)
    from pegasus_amd.sh_utils import RGB2SH
    shs = RGB2SH(colors_np[np.maximum(layer, 1) - 1]).astype(np.float32).reshape(n, 1, 3)
    _, views = scenes.scene_c3(scale=0.001, n_views=n_views, width=200, height=136)
    dev = gpu_device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    T = dict(xyz=t(xyz), op=t(opac), sc=t(scales), rot=t(rots), shs=t(shs), layer=t(layer), colors=t(colors_np))
    bg = t(np.array([0.05, 0.0, 0.1], np.float32))
    specs = [R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, bg, t(v.world_view_transform.astype(np.float32)),
                        t(v.full_proj_transform.astype(np.float32)), t(v.camera_center.astype(np.float32))) for v in views]
    H, W = views[0].height, views[0].width
    out = torch.full((n_views, K, H, W), 9, dtype=torch.uint8, device=dev)
    h = R.forward_views(T["xyz"], T["op"], specs, shs=T["shs"], scales=T["sc"], rotations=T["rot"], sh_degree=0, want_radii=False,
                        outputs=[dict(radii=None, sem_masks=out[i]) for i in range(n_views)], async_slot=("layered-test", n_views),
                        layers=dict(layer_id=T["layer"], n_layers=K, mask_colors=T["colors"], mask_threshold=0.1))
    h.wait()
    torch.cuda.synchronize()
    assert int((out > 1).sum()) == 0                                   # every pixel of every plane was written
    for k in range(1, K + 1):
        s = slice(n0 + (k - 1) * per, n0 + k * per)
        res = R.forward_views(T["xyz"][s], T["op"][s], specs, shs=T["shs"][s], scales=T["sc"][s], rotations=T["rot"][s],
                              sh_degree=0, want_radii=False)
        want = torch.stack([M.color_masks(r["color"], T["colors"][k - 1:k], 0.1)[0] for r in res])
        assert torch.equal(out[:, k - 1], want), f"layer {k}"
    assert int(out.sum()) > 0


def test_empty_scene_layered_and_fused_masks_stay_zero(gpu_device):
    """N = 0 (SURVEY.md 8a "Edge cases": outputs stay zero-filled, no background) for the two outputs round 4 added: the
    layered call's mask planes and the fused call's sem_masks."""
    import torch
    from pegasus_amd import rasterizer as R
    dev = gpu_device
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
    spec = R.ViewSpec(48, 64, 0.5, 0.5, torch.ones(3, device=dev), torch.eye(4, device=dev), torch.eye(4, device=dev), z(3))
    colors = torch.tensor([[1.0, 1.0, 1.0], [0.2, 0.4, 0.6]], device=dev)        # colour 0 IS the background: still zeros
    planes = torch.full((2, 48, 64), 7, dtype=torch.uint8, device=dev)
    R.forward_views(z(0, 3), z(0, 1), [spec], shs=z(0, 1, 3), scales=z(0, 3), rotations=z(0, 4), sh_degree=0, want_radii=False,
                    outputs=[dict(radii=None, sem_masks=planes)], async_slot=("empty-layered", 0),
                    layers=dict(layer_id=z(0, dt=torch.int32), n_layers=2, mask_colors=colors, mask_threshold=0.1)).wait()
    torch.cuda.synchronize()
    assert int(planes.sum()) == 0
    out = dict(color=torch.full((3, 48, 64), 5.0, device=dev), depth=torch.full((1, 48, 64), 5.0, device=dev), radii=None,
               sem_color=torch.full((3, 48, 64), 5.0, device=dev), sem_depth=torch.full((1, 48, 64), 5.0, device=dev),
               sem_masks=torch.full((2, 48, 64), 7, dtype=torch.uint8, device=dev))
    sem = dict(object_id=z(0, dt=torch.int32), colors=colors, n_env=0, k=2, mask_colors=colors, mask_threshold=0.1)
    R.forward_views(z(0, 3), z(0, 1), [spec], shs=z(0, 1, 3), scales=z(0, 3), rotations=z(0, 4), sh_degree=0, want_radii=False,
                    outputs=[out], semantic=sem)
    torch.cuda.synchronize()
    assert all(float(out[k].float().abs().max()) == 0.0 for k in ("color", "depth", "sem_color", "sem_depth", "sem_masks"))


@pytest.mark.parametrize("size", [(320, 240), (333, 250)])
def test_frame_records_from_the_compositor_equal_pack_records(gpu_device, size):
    """PgrOutputs::record: the frame record written by the compositor's epilogue (uint8 RGB | uint16 depth mm | mask bit planes)
    is bit for bit what pgr_pack_records makes of the frame's float images and masks -- fused frames (K masks), raster-only
    frames (no mask section), a non-zero background, an image size that is not a multiple of the tile, the depth-mode switch,
    and after an instance overflow re-rendered the batch."""
    import torch
    from pegasus_amd import frames as F, masks as M, rasterizer
    w, h = size
    cloud, views = scenes.scene_c3(scale=0.03, n_views=4, width=w, height=h)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device, bg=(0.3, 0.05, 0.6))
    for masks, mode in ((True, 0), (False, 0), (True, 1)):
        specs = [fr.view_spec(v, depth_mode=mode) for v in views]
        f = fr.alloc_frames(4, h, w, masks=masks, records=True)
        f["records"].fill_(0xAB)
        fr.render_frames_async(specs, f, masks=masks, slot=0).wait()
        want = M.pack_records(f["color"], f["depth"], f["masks"] if masks else None)
        torch.cuda.synchronize()
        lay = M.record_layout(h, w, fr.K if masks else 0)
        assert f["records"].shape == want.shape == (4, lay["bytes"])
        got, ref = M.record_views(f["records"], h, w, fr.K if masks else 0), M.record_views(want, h, w, fr.K if masks else 0)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (masks, mode, k)
        g = fr.render_frames(specs, fr.alloc_frames(4, h, w, masks=masks, records=True), masks=masks)      # the blocking form too
        assert all(torch.equal(M.record_views(g["records"], h, w, fr.K if masks else 0)[k], ref[k]) for k in ref)
        # RECORDS-ONLY views (round 6: PgrOutputs color = depth = sem_* = NULL, what a rank of a view-sharded job asks for): the
        # same record, no image written; and records + the semantic image alone (the dataset writer's frame set)
        for images in (False, "seg"):
            ro = fr.alloc_frames(4, h, w, masks=masks, records=True, images=images)
            assert "color" not in ro and "masks" not in ro and ("seg" in ro) == (images == "seg" and masks)
            ro["records"].fill_(0xCD)
            fr.render_frames_async(specs, ro, masks=masks, slot=0).wait()
            torch.cuda.synchronize()
            got = M.record_views(ro["records"], h, w, fr.K if masks else 0)
            for k in ref:
                assert torch.equal(got[k], ref[k]), ("records only", images, masks, mode, k)
            if "seg" in ro:
                assert torch.equal(ro["seg"], f["seg"])
    rasterizer.reset_capacity(1500)
    specs = [fr.view_spec(v) for v in views]
    f = fr.alloc_frames(4, h, w, records=True)
    fr.render_frames_async(specs, f, slot=1).wait()
    torch.cuda.synchronize()
    want = M.pack_records(f["color"], f["depth"], f["masks"])
    ref, got = M.record_views(want, h, w, fr.K), M.record_views(f["records"], h, w, fr.K)
    assert all(torch.equal(got[k], ref[k]) for k in ref)
    rasterizer.reset_capacity(1500)                        # the overflow re-render of a records-only batch
    ro = fr.alloc_frames(4, h, w, records=True, images=False)
    fr.render_frames_async(specs, ro, slot=1).wait()
    torch.cuda.synchronize()
    got = M.record_views(ro["records"], h, w, fr.K)
    assert all(torch.equal(got[k], ref[k]) for k in ref)


@pytest.mark.parametrize("size", [(1, 1), (2, 3), (17, 3), (15, 33), (64, 1)])
def test_tiny_and_ragged_images_through_the_round4_outputs(gpu_device, size):
    """One-pixel, one-row, sub-tile and odd-sized images (record sections that start at odd byte counts, quarters that lie
    partly or wholly outside the image) through the outputs round 4 added: layered silhouettes == per-object passes, records
    from the epilogue == pgr_pack_records, fused masks == separate pass."""
    import torch
    from pegasus_amd import frames as F, masks as M
    W, H = size
    for k_obj in (1, 3):
        cloud, views, _ = scenes.merged_scene(500 + W + H, 400, k_obj, 300, 3, W, H, plane_size=0.6)
        act = cloud.activated()
        fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                             sh_degree=3, device=gpu_device, bg=(0.1, 0.2, 0.3))
        specs = [fr.view_spec(v) for v in views]
        assert torch.equal(fr.render_silhouettes(specs).clone(), fr.render_silhouettes_per_object(specs).clone())
        f = fr.alloc_frames(3, H, W, records=True)
        fr.render_frames_async(specs, f, slot=0).wait()
        want = M.pack_records(f["color"], f["depth"], f["masks"])
        ref = fr.render_batch(specs)
        torch.cuda.synchronize()
        got_v, want_v = M.record_views(f["records"], H, W, fr.K), M.record_views(want, H, W, fr.K)
        assert all(torch.equal(got_v[k], want_v[k]) for k in ("rgb", "depth_mm", "mask_bits"))
        assert all(torch.equal(f[k], ref[k]) for k in ("color", "depth", "seg", "masks"))


@pytest.mark.parametrize("ties", [True, False])
@pytest.mark.parametrize("n", [1, 2, 512, 513, 1024, 1025, 2048, 2049, 4096, 4097, 8192, 8193, 7936, 7937, 12000, 15872, 15873,
                               16000, 16001, 16384, 16385, 20481, 33000, 70000])
def test_sort_tier_boundaries(oracle, gpu_device, n, ties):
    """ONE tile list of exactly n keys, n on both sides of every capacity edge of the per-tile sort (256 x 2 / 4 / 8 keys,
    512 x 8, 512 x 16 over 3584 buckets, the windowed sort of 8193..15 872 keys -- one window up to 7936 positions, two
    beyond; with `ties` its reject path into the open-ended kernel's queue --, beyond 15 872 keys the split pre-pass whose depth
    segments the 512 x 16 kernel sorts from the segment queue (with `ties`: a piled-up coarse bucket, the list falls back to the
    open-ended kernel), the 16 000-key LDS image of the open-ended kernel, 16 384 = its register capacity):
    n pinpoint splats inside one tile -- with a third of them at exactly equal depth (`ties`: the pile-up rejects the bucket
    sort, so every tier's merge-sort fallback and the index tie-break run) or all depths distinct (the bucket sort itself at
    exactly its capacity).  As a one-view call (every long list goes to the open-ended tier's kernel) and as a three-view
    batch (one kernel per tier): the list bit-exact against the oracle."""
    import torch
    from helpers import fetch_workspace
    from pegasus_amd import rasterizer as R
    rng = np.random.default_rng(n)
    cloud, views = scenes.scene_c1(seed=9, n=n)
    cloud.xyz[:, 0] = np.float32(0.0929) + rng.normal(0, 5e-4, n).astype(np.float32)       # 8.5 px off the image centre:
    cloud.xyz[:, 1] = np.float32(0.0929) + rng.normal(0, 5e-4, n).astype(np.float32)       # the middle of a tile
    cloud.xyz[:, 2] = rng.normal(0, 0.2, n).astype(np.float32)
    if ties:
        cloud.xyz[: n // 3, 2] = np.float32(0.125)                                         # exact depth ties
    cloud.scaling[:] = np.log(0.0005).astype(np.float32)
    cloud.opacity[:] = rng.normal(-3.0, 0.5, size=(n, 1)).astype(np.float32)
    act, v = cloud.activated(), views[0]
    o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    assert lens.max() == n and (lens > 0).sum() == 1, (lens.max(), (lens > 0).sum())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu_device)
    spec = R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.zeros(3)), t(v.world_view_transform),
                      t(v.full_proj_transform), t(v.camera_center))
    T = {k: t(a) for k, a in act.items()}
    for n_views in (1, 3):
        res = R.forward_views(T["means3d"], T["opacities"], [spec] * n_views, shs=T["shs"], scales=T["scales"],
                              rotations=T["rotations"], sh_degree=3, want_radii=True, want_aux=True)
        torch.cuda.synchronize()
        for k in range(n_views):
            w = fetch_workspace(k, n, v.width, v.height)
            np.testing.assert_array_equal(w["gauss_sorted"], o["gauss_sorted"], err_msg=f"{n_views} views, view {k}")
            np.testing.assert_array_equal(res[k]["n_contrib"].cpu().numpy().astype(np.uint32)[~o["ambig"].astype(bool)],
                                          o["n_contrib"][~o["ambig"].astype(bool)])
