"""PEGASUS's render path through the compat module names: Camera -> render() -> the four wrappers of
src/gs/render.py, on the GPU, against the oracle."""
import math
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_render_facade_and_mask_wrappers(oracle, gpu_device):
    sys.path.insert(0, str(ROOT / "compat"))
    try:
        import torch
        from argparse import ArgumentParser
        from arguments import PipelineParams                  # noqa: the names PEGASUS imports
        from gaussian_renderer import render, GaussianModel
        from scene.cameras import Camera
        from utils.sh_utils import RGB2SH
        from pegasus_amd import masks, render as prender, scenes
        from helpers import assert_images_match

        dev = gpu_device
        # the reference's render loop runs under torch.no_grad() (/root/reference/pegasus.py:248); outside it render()
        # hands back graph-attached tensors, exactly like upstream (viewspace_points requires grad)
        grad_was = torch.is_grad_enabled()
        torch.set_grad_enabled(False)
        rng = np.random.default_rng(2)
        env_c = scenes.ground_plane(rng, 15000, 1.0, np.log(0.006), 0.4, 0.1, 0.005)
        obj_c = [scenes.rigid_transform(scenes.box_object(rng, 5000, (0.1, 0.12, 0.16), np.log(0.004), 0.3, 0.2, k + 1),
                                        np.eye(3), np.array([0.15 * k - 0.08, 0.05 * k, 0.09])) for k in range(2)]
        mk = lambda c: GaussianModel.from_arrays(c.xyz, c.features_dc, c.features_rest, c.opacity, c.scaling,
                                                 c.rotation, device=dev)
        env, objs = mk(env_c), {k + 1: mk(c) for k, c in enumerate(obj_c)}
        colors = torch.from_numpy(masks.generate_colors(len(objs))).to(dev)
        prender.assign_semantic_colors(objs, colors)

        v = scenes.scene_c3(scale=0.001, n_views=1, width=320, height=240, camera_set="fibonacci_above_9deg")[1][0]
        H, W = v.height, v.width
        cam = Camera(colmap_id=0, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=torch.empty((3, H, W)),
                     gt_alpha_mask=None, image_name="0", uid=0, data_device=str(dev))
        np.testing.assert_allclose(cam.world_view_transform.cpu().numpy(), v.world_view_transform, atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.cpu().numpy(), v.full_proj_transform, atol=1e-5)
        pipe = PipelineParams(ArgumentParser())
        bg = torch.zeros(3, device=dev)

        # full scene = environment + objects in their own colours (pegasus.py:255-264)
        import copy
        scene = copy.deepcopy(env)
        for o in objs.values():
            scene.merge_gaussians(o)
        pkg = render(cam, scene, pipe, bg)
        assert set(pkg) == {"render", "depth", "viewspace_points", "visibility_filter", "radii"}
        merged = scenes.SplatCloud.concat([env_c, *obj_c])
        act = merged.activated()
        # the facade applies torch's activations on the device; feed the oracle those exact values
        act = dict(means3d=scene.get_xyz.cpu().numpy(), opacities=scene.get_opacity.cpu().numpy().reshape(-1),
                   scales=scene.get_scaling.cpu().numpy(), rotations=scene.get_rotation.cpu().numpy(),
                   shs=scene.get_features.cpu().numpy())
        kw = dict(width=W, height=H, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
                  viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
                  campos=cam.camera_center.cpu().numpy(), bg=np.zeros(3, np.float32))
        o = oracle.forward(**act, sh_degree=3, num_threads=8, cull_mode=1, **kw)
        amb = o["ambig"].astype(bool)
        assert np.abs(pkg["render"].cpu().numpy() - o["color"])[:, ~amb].max() <= 1e-4
        assert np.abs(pkg["depth"].cpu().numpy() - o["out_depth"])[:, ~amb].max() <= 1e-4
        np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), o["radii"])
        np.testing.assert_array_equal(pkg["visibility_filter"].cpu().numpy(), o["radii"] > 0)

        rgb, depth = prender.render_rgb_and_depth(cam, scene, pipe, bg)
        assert tuple(rgb.shape) == (H, W, 3) and tuple(depth.shape) == (H, W, 1) and rgb.device.type == "cpu"

        # masks: wrappers == oracle semantic render + colour-distance rule (render.py:68-97, 36-65)
        vis, seg = prender.render_visib_mask(cam, env, objs, colors, H, W, pipe, bg)
        assert vis.shape == (H, W, 2) and vis.dtype == np.float64 and tuple(seg.shape) == (H, W, 3)
        n_env = env_c.n
        sem_shs = np.zeros((merged.n - n_env, 16, 3), np.float32)
        oid = merged.object_id[n_env:]
        sem_shs[:, 0] = RGB2SH(colors.cpu().numpy()[oid - 1])
        so = oracle.forward(act["means3d"][n_env:], act["opacities"][n_env:], scales=act["scales"][n_env:],
                            rotations=act["rotations"][n_env:], shs=sem_shs, sh_degree=3, num_threads=8, **kw)
        assert np.abs(seg.numpy().transpose(2, 0, 1) - so["color"])[:, ~so["ambig"].astype(bool)].max() <= 1e-4
        ref_masks = oracle.color_masks(so["color"], colors.cpu().numpy(), 0.1)
        dist = np.stack([np.linalg.norm(so["color"].transpose(1, 2, 0) - c, axis=2) for c in colors.cpu().numpy()])
        decided = (np.abs(dist - 0.1) > 2e-4) & ~so["ambig"].astype(bool)[None]
        np.testing.assert_array_equal(vis.transpose(2, 0, 1)[decided], ref_masks[decided].astype(np.float64))
        assert vis.sum() > 100

        sil = prender.render_silhouette_mask(cam, objs, env, W, H, colors, pipe, bg)
        assert sil.shape == (H, W, 2) and sil.dtype == np.float64
        # the layered call (all objects in one pass) against the reference's form: one render per object, bit for bit
        np.testing.assert_array_equal(sil, prender._render_silhouette_mask_per_object(cam, objs, env, W, H, colors, pipe, bg))
        white = torch.ones(3, device=dev)
        np.testing.assert_array_equal(prender.render_silhouette_mask(cam, objs, env, W, H, colors, pipe, white),
                                      prender._render_silhouette_mask_per_object(cam, objs, env, W, H, colors, pipe, white))
        assert (sil.sum(axis=(0, 1)) >= vis.sum(axis=(0, 1)) * 0.98).all()   # a silhouette covers the visible part
        sem = prender.render_semanticsegmentation_mask(cam, env, objs, colors, H, W, pipe, bg, False)
        assert sem.dtype == np.uint8 and sem.shape == (H, W, 3)
        np.testing.assert_array_equal(sem, (np.ascontiguousarray(seg.numpy()) * 255).astype("uint8"))
        torch.set_grad_enabled(grad_was)
        pkg = render(cam, scene, pipe, bg)                       # outside no_grad: upstream's training-style contract
        assert pkg["render"].requires_grad and pkg["viewspace_points"].requires_grad
    finally:
        sys.path.remove(str(ROOT / "compat"))
        try:
            torch.set_grad_enabled(True)
        except NameError:
            pass


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("uniform", 20000), ("clustered", 30000), ("plane", 10000), ("tiny", 3), ("one", 1),
                                    ("duplicates", 5000)])
def test_distCUDA2_matches_kdtree(gpu_device, kind, n):
    """simple_knn._C.distCUDA2 (pgr_knn_mean_dist2): mean squared distance to the 3 nearest neighbours, against
    scipy's exact k-d tree in float64."""
    import sys
    import torch
    from pathlib import Path
    from scipy.spatial import cKDTree
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "compat"))
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    if kind == "uniform":
        pts = rng.uniform(-1, 1, size=(n, 3))
    elif kind == "clustered":
        centres = rng.uniform(-5, 5, size=(20, 3))
        pts = centres[rng.integers(20, size=n)] + rng.normal(0, 0.01, size=(n, 3))
        pts[:50] = rng.uniform(-50, 50, size=(50, 3))                        # far outliers: many empty shells
    elif kind == "plane":
        pts = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), np.zeros((n, 1))], axis=1)   # degenerate axis
    elif kind == "duplicates":
        pts = np.repeat(rng.uniform(-1, 1, size=(n // 5, 3)), 5, axis=0)    # 4 coincident neighbours each
    else:
        pts = rng.uniform(-1, 1, size=(n, 3))
    pts = pts.astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy()
    assert got.shape == (n,)
    if n < 4:
        assert (got > 1e30).all()            # fewer than 3 neighbours: FLT_MAX terms, as upstream
        return
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-12)


@pytest.mark.gpu
def test_generate_writes_bop_layout(gpu_device, tmp_path):
    """python -m pegasus_amd.generate: frames of a (small) dynamic scene land on disk in the reference's BOP layout;
    the PNGs decode to the quantised frames and the JSON records cover every frame and object."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    from pegasus_amd.dataset_writer import decode_png
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, "-m", "pegasus_amd.generate", "--out", str(tmp_path), "--frames", "6", "--batch", "4",
                        "--scale", "0.01", "--size", "160", "--dynamic"], cwd=str(root), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    scene = tmp_path / "train" / "000000"
    gt, cam = json.loads((scene / "scene_gt.json").read_text()), json.loads((scene / "scene_camera.json").read_text())
    assert sorted(gt) == [str(i) for i in range(6)] and sorted(cam) == sorted(gt)
    assert len(gt["0"]) == 8 and len(gt["0"][0]["cam_R_m2c"]) == 9 and len(cam["0"]["cam_K"]) == 9
    assert gt["0"][0]["cam_R_m2c"] != gt["5"][0]["cam_R_m2c"]                      # the objects move
    rgb = decode_png((scene / "rgb" / "000003.png").read_bytes())
    depth = decode_png((scene / "depth" / "000003.png").read_bytes())
    masks = [decode_png((scene / "mask_visib" / f"000003_{k:06d}.png").read_bytes()) for k in range(8)]
    assert rgb.shape == (160, 160, 3) and rgb.dtype == np.uint8 and rgb.max() > 0
    assert depth.shape == (160, 160) and depth.dtype == np.uint16 and 200 < depth[depth > 0].mean() < 3000   # millimetres
    assert all(set(np.unique(m)) <= {0, 255} for m in masks) and sum(int(m.sum()) for m in masks) > 0
    # the reference writer's other two image kinds (pegasus_working.py:423-434): the uint8 semantic image and the silhouettes
    sem = decode_png((scene / "sem_mask" / "000003.png").read_bytes())
    sil = [decode_png((scene / "mask" / f"000003_{k:06d}.png").read_bytes()) for k in range(8)]
    assert sem.shape == (160, 160, 3) and sem.dtype == np.uint8 and sem.max() > 0
    assert all(set(np.unique(m)) <= {0, 255} for m in sil)
    # a silhouette holds its object's visible mask (up to threshold pixels at blended edges), and occluded parts beside it
    assert sum(int((s_ >= v_).sum()) for s_, v_ in zip(sil, masks)) > 0.999 * 8 * 160 * 160
    assert sum(int(m.sum()) for m in sil) >= sum(int(m.sum()) for m in masks)
    # where exactly one object is visible the semantic image shows that object's colour
    e0 = gt["3"][0]
    assert set(e0) >= {"cam_R_m2c", "cam_t_m2c", "T_w2c", "T_m2w", "obj_id", "bullet_obj_id", "3d_bounding_box_model_coord",
                       "3d_bounding_center", "projected_center", "projected_points"}
    assert np.array(e0["projected_points"]).shape == (8, 2) and len(e0["T_w2c"]) == 16
    # scene_gt_info.json (BOP format, bop_datasets_format.md:116-129) from the same masks: counts and boxes agree with the PNGs
    info = json.loads((scene / "scene_gt_info.json").read_text())
    assert sorted(info) == sorted(gt) and len(info["3"]) == 8
    for k in range(8):
        e = info["3"][k]
        assert e["px_count_visib"] == int((masks[k] == 255).sum()) and e["px_count_all"] == int((sil[k] == 255).sum())
        assert e["px_count_valid"] == int(((sil[k] == 255) & (depth != 0)).sum())
        if e["px_count_visib"]:
            ys, xs = np.nonzero(masks[k])
            assert e["bbox_visib"] == [int(xs.min()), int(ys.min()), int(xs.max() - xs.min()), int(ys.max() - ys.min())]
            ys, xs = np.nonzero(sil[k])
            assert e["bbox_obj"] == [int(xs.min()), int(ys.min()), int(xs.max() - xs.min()), int(ys.max() - ys.min())]
            assert abs(e["visib_fract"] - e["px_count_visib"] / e["px_count_all"]) < 1e-12
        else:
            assert e["bbox_visib"] == [-1, -1, -1, -1] and e["bbox_obj"] == [-1, -1, -1, -1]
    assert sum(e["px_count_visib"] for e in info["3"]) > 0


@pytest.mark.gpu
def test_render_facade_activation_cache_follows_the_model(gpu_device):
    """render() keeps the activated parameters of an unchanged model between calls (inference); an in-place edit or a
    re-assignment of a raw tensor must show in the very next frame."""
    import sys
    from argparse import ArgumentParser
    import torch
    root = str(Path(__file__).resolve().parents[1])
    for p in (root, root + "/compat"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gaussian_renderer import render, GaussianModel
    from scene.cameras import Camera
    from arguments import PipelineParams
    from pegasus_amd import scenes
    dev = gpu_device
    cloud, views = scenes.scene_c1(seed=9, n=4000)
    v = views[0]
    with torch.no_grad():
        pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling,
                                       cloud.rotation, device=dev)
        cam = Camera(colmap_id=0, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
                     image_height=v.height, gt_alpha_mask=None, image_name="0", uid=0, data_device=str(dev))
        pipe = PipelineParams(ArgumentParser())
        bg = torch.zeros(3, device=dev)
        a = render(cam, pc, pipe, bg)["render"].clone()
        b = render(cam, pc, pipe, bg)["render"].clone()          # second call: cached activations
        assert torch.equal(a, b)
        pc._opacity.sub_(2.0)                                      # in-place edit: version counter moves
        c = render(cam, pc, pipe, bg)["render"].clone()
        assert not torch.equal(a, c)
        pc._features_dc = pc._features_dc.clone() * 0.5            # re-assignment
        d = render(cam, pc, pipe, bg)["render"].clone()
        assert not torch.equal(c, d)
        fresh = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc * 0.5, cloud.features_rest, cloud.opacity - 2.0,
                                          cloud.scaling, cloud.rotation, device=dev)
        e = render(cam, fresh, pipe, bg)["render"]
        assert torch.equal(d, e)                                   # what an uncached model with the same values renders


@pytest.mark.gpu
def test_semantic_wrappers_follow_objects_cameras_and_backgrounds(gpu_device):
    """render_visib_mask / render_semanticsegmentation_mask share one objects-only scene and one render per (scene, camera,
    background) (round 3).  Whatever is kept must be dropped the moment an object moves, is re-assigned or re-coloured, the
    camera changes or the background changes -- every result below is compared with the same call on a cold cache."""
    import copy
    import sys
    from argparse import ArgumentParser
    import torch
    root = str(Path(__file__).resolve().parents[1])
    for p in (root, root + "/compat"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gaussian_renderer import GaussianModel
    from scene.cameras import Camera
    from arguments import PipelineParams
    from pegasus_amd import masks, render as prender, scenes
    dev = gpu_device
    rng = np.random.default_rng(4)
    env_c = scenes.ground_plane(rng, 8000, 1.0, np.log(0.006), 0.4, 0.1, 0.005)
    obj_c = [scenes.rigid_transform(scenes.box_object(rng, 4000, (0.1, 0.12, 0.16), np.log(0.004), 0.3, 0.2, k + 1),
                                    np.eye(3), np.array([0.15 * k - 0.08, 0.05 * k, 0.09])) for k in range(2)]
    mk = lambda c: GaussianModel.from_arrays(c.xyz, c.features_dc, c.features_rest, c.opacity, c.scaling, c.rotation, device=dev)
    # (cameras that look DOWN on the hand-placed objects: the default set starts at elevation 0)
    views = scenes.scene_c3(scale=0.001, n_views=2, width=320, height=240, camera_set="fibonacci_above_9deg")[1]
    cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
                   image_height=v.height, gt_alpha_mask=None, image_name=str(i), uid=i, data_device=str(dev))
            for i, v in enumerate(views)]
    H, W = views[0].height, views[0].width
    pipe = PipelineParams(ArgumentParser())
    with torch.no_grad():
        env, objs = mk(env_c), {k + 1: mk(c) for k, c in enumerate(obj_c)}
        colors = torch.from_numpy(masks.generate_colors(len(objs))).to(dev)
        prender.assign_semantic_colors(objs, colors)
        bg = torch.zeros(3, device=dev)

        def both(cam, background, cold):
            if cold:
                prender.reset_cache()
            vis, seg = prender.render_visib_mask(cam, env, objs, colors, H, W, pipe, background)
            sem = prender.render_semanticsegmentation_mask(cam, env, objs, colors, H, W, pipe, background, False)
            return vis.copy(), seg.clone(), sem.copy()

        def same(a, b):
            return np.array_equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2])

        first = both(cams[0], bg, cold=True)
        assert first[0].sum() > 100
        np.testing.assert_array_equal(first[2], (np.ascontiguousarray(first[1].numpy()) * 255).astype("uint8"))
        assert same(both(cams[0], bg, cold=False), first)                    # warm == cold
        other_cam = both(cams[1], bg, cold=False)                           # another camera on the warm scene
        assert same(other_cam, both(cams[1], bg, cold=True)) and not same(other_cam, first)
        white = torch.ones(3, device=dev)                                   # another background tensor
        w_warm = both(cams[0], white, cold=False)
        assert same(w_warm, both(cams[0], white, cold=True)) and not torch.equal(w_warm[1], first[1])
        bg.fill_(0.25)                                                      # the SAME background tensor, edited in place
        e_warm = both(cams[0], bg, cold=False)
        assert same(e_warm, both(cams[0], bg, cold=True)) and not torch.equal(e_warm[1], first[1])
        bg.zero_()
        objs[1].apply_translation_on_xyz(torch.tensor([0.05, 0.0, 0.02], device=dev))       # an object moves (re-assignment)
        moved = both(cams[0], bg, cold=False)
        assert same(moved, both(cams[0], bg, cold=True)) and not np.array_equal(moved[0], first[0])
        objs[2]._xyz.add_(torch.tensor([0.0, -0.04, 0.0], device=dev))                       # ... and one in place
        moved2 = both(cams[0], bg, cold=False)
        assert same(moved2, both(cams[0], bg, cold=True)) and not np.array_equal(moved2[0], moved[0])
        objs[1]._opacity = objs[1]._opacity - 4.0                                            # nearly transparent now
        faint = both(cams[0], bg, cold=False)
        assert same(faint, both(cams[0], bg, cold=True)) and faint[0][..., 0].sum() < moved2[0][..., 0].sum()
        swapped = torch.flip(colors, dims=[0]).contiguous()                                  # objects re-coloured
        prender.assign_semantic_colors(objs, swapped)
        recol = both(cams[0], bg, cold=False)
        assert same(recol, both(cams[0], bg, cold=True)) and not torch.equal(recol[1], faint[1])
        # the objects keep the paint, as the reference's callers expect (render.py:51-52)
        assert torch.equal(objs[2]._features_dc, objs[2]._features_dc_semantics.expand_as(objs[2]._features_dc))
        # Two pose updates between semantic renders (masks asked for every 2nd step).  apply_transformation /
        # apply_translation replace _xyz with a fresh tensor per update, so ids and allocator blocks are recycled: a key of
        # (id, version, data_ptr) could match tensors that merely reuse the addresses of two updates ago.  The slot keeps
        # its keyed tensors alive, so it cannot (round-3 advisor finding).  The cold reference leaves the warm slot as it was.
        for step in range(8):
            for _ in range(2):
                objs[1].apply_translation_on_xyz(torch.tensor([0.004, 0.0, 0.001], device=dev))
            warm = both(cams[0], bg, cold=False)
            saved = prender.swap_cache(None)
            cold = both(cams[0], bg, cold=False)           # (the slot is empty: a cold render)
            prender.swap_cache(saved)
            assert same(warm, cold), step


@pytest.mark.gpu
def test_generate_view_sharded_over_two_ranks_writes_the_same_dataset(gpu_device, tmp_path):
    """python -m pegasus_amd.generate under torchrun: frame f -> rank f mod 2, every rank renders and WRITES its own frames
    under their global numbers, only the pose records travel (gloo, Python objects).  The dataset equals the one a single
    process writes: every PNG byte for byte, both JSON files."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    flags = ["--frames", "7", "--batch", "3", "--scale", "0.01", "--size", "128", "--dynamic"]
    env = {k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one, two = tmp_path / "one", tmp_path / "two"
    r = subprocess.run([sys.executable, "-m", "pegasus_amd.generate", "--out", str(one), *flags], cwd=str(root), env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "pegasus_amd.generate", "--out", str(two), *flags], cwd=str(root), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    files = sorted(p.relative_to(one) for p in one.rglob("*") if p.is_file())
    assert files == sorted(p.relative_to(two) for p in two.rglob("*") if p.is_file()) and len(files) == 7 * (3 + 16) + 3
    for f in files:
        if f.suffix == ".png":
            assert (one / f).read_bytes() == (two / f).read_bytes(), f
        else:
            assert json.loads((one / f).read_text()) == json.loads((two / f).read_text()), f


@pytest.mark.gpu
def test_render_follows_cameras_with_transposed_matrices_and_in_place_edits(gpu_device):
    """render() keeps contiguous copies of a camera's matrices while the camera's tensors are unchanged (upstream's Camera
    hands out `.transpose(0, 1)` VIEWS, whose contiguous copy cost two kernels per call).  The copies must follow the
    camera: an upstream-style camera (transposed views) renders like this build's own Camera, an in-place edit of its matrices
    (through the view or through its base tensor) or a re-assignment is seen by the next call, and the shared zero
    `viewspace_points` stays zero."""
    import types
    import torch
    from argparse import ArgumentParser
    sys.path.insert(0, str(ROOT / "compat"))
    from arguments import PipelineParams
    from pegasus_amd import gaussian_renderer as GR, graphics as G, scenes
    from pegasus_amd.cameras import Camera
    from pegasus_amd.gaussian_model import GaussianModel
    dev = gpu_device
    cloud, views = scenes.scene_c3(scale=0.01, n_views=3, width=320, height=240, camera_set="fibonacci_above_9deg")
    pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling, cloud.rotation,
                                   device=dev)
    pipe = PipelineParams(ArgumentParser())
    bg = torch.zeros(3, device=dev)

    def ours(v):
        return Camera(colmap_id=0, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
                      image_height=v.height, gt_alpha_mask=None, image_name="0", uid=0, data_device=str(dev))

    def upstream_style(v):
        """What the reference's scene.cameras.Camera computes (/root/reference/src/gs/pegasus_setup.py:130-140 builds it):
        transposed VIEWS on the device, full_proj by bmm, the centre as a row of the (column-major) inverse."""
        c = types.SimpleNamespace(image_width=v.width, image_height=v.height, FoVx=v.fovx, FoVy=v.fovy)
        c.base_view = torch.tensor(G.getWorld2View2(v.R_c2w, v.t_w2c)).to(dev)
        c.world_view_transform = c.base_view.transpose(0, 1)
        c.projection_matrix = torch.tensor(G.getProjectionMatrix(G.ZNEAR, G.ZFAR, v.fovx, v.fovy)).to(dev).transpose(0, 1)
        c.full_proj_transform = (c.world_view_transform.unsqueeze(0).bmm(c.projection_matrix.unsqueeze(0))).squeeze(0)
        c.camera_center = c.world_view_transform.inverse()[3, :3]
        return c

    with torch.no_grad():
        ref = [GR.render(upstream_style(v), pc, pipe, bg) for v in views]                    # fresh cameras: cold copies
        own = GR.render(ours(views[0]), pc, pipe, bg)
        # (this build's Camera multiplies the same matrices in contiguous layout: the products may differ in the last bit)
        assert float((own["render"] - ref[0]["render"]).abs().max()) < 1e-4 and int((own["radii"] != ref[0]["radii"]).sum()) <= 2
        cam = upstream_style(views[0])
        assert not cam.world_view_transform.is_contiguous()
        a = GR.render(cam, pc, pipe, bg)
        assert torch.equal(a["render"], ref[0]["render"]) and torch.equal(a["radii"], ref[0]["radii"])
        assert torch.equal(GR.render(cam, pc, pipe, bg)["render"], ref[0]["render"])           # warm: the kept copies
        # the camera moves IN PLACE: through the base tensor of the view, through the views themselves
        other = upstream_style(views[1])
        cam.base_view.copy_(other.base_view)
        cam.full_proj_transform.copy_(other.full_proj_transform)
        cam.camera_center.copy_(other.camera_center)
        b = GR.render(cam, pc, pipe, bg)
        assert torch.equal(b["render"], ref[1]["render"]) and not torch.equal(b["render"], ref[0]["render"])
        # ... and by re-assignment
        third = upstream_style(views[2])
        cam.world_view_transform, cam.full_proj_transform, cam.camera_center = (third.world_view_transform, third.full_proj_transform,
                                                                                third.camera_center)
        c = GR.render(cam, pc, pipe, bg)
        assert torch.equal(c["render"], ref[2]["render"])
        assert torch.equal(c["visibility_filter"], c["radii"] > 0) and int(c["visibility_filter"].sum()) > 100
        assert float(c["viewspace_points"].abs().max()) == 0.0 and c["viewspace_points"].shape == pc.get_xyz.shape
        assert c["viewspace_points"] is a["viewspace_points"]                                   # the shared zero probe


@pytest.mark.gpu
def test_render_after_an_instance_overflow_redoes_the_queued_visibility_filter(gpu_device):
    """render() queues its visibility filter (radii > 0) behind the compositor BEFORE it reads the batch's status; if that
    status says "instance overflow" the view is rendered again and the filter must be taken from the NEW radii (the
    `_was_redone` branch of diff_gaussian_rasterization.rasterize_gaussians): a capacity hint far below the view's need
    forces the branch; image, radii and visibility_filter must equal the roomy first call's."""
    import sys
    from argparse import ArgumentParser
    import torch
    root = str(Path(__file__).resolve().parents[1])
    for p in (root, root + "/compat"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gaussian_renderer import render, GaussianModel
    from scene.cameras import Camera
    from arguments import PipelineParams
    from pegasus_amd import rasterizer, scenes
    dev = torch.device(gpu_device)
    cloud, views = scenes.scene_c3(scale=0.02, n_views=2, width=320, height=240)
    v = views[0]
    with torch.no_grad():
        pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling,
                                       cloud.rotation, device=dev)
        cam = Camera(colmap_id=0, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
                     image_height=v.height, gt_alpha_mask=None, image_name="0", uid=0, data_device=str(dev))
        pipe = PipelineParams(ArgumentParser())
        bg = torch.zeros(3, device=dev)
        first = {k: t.clone() for k, t in render(cam, pc, pipe, bg).items() if torch.is_tensor(t)}
        need = int(rasterizer.last_forward_info()["num_instances"][0])
        assert need > 4000
        key = (dev, cloud.n, v.width, v.height)                      # (capacity_hints()' key of this scene shape)
        rasterizer.set_capacity_hint(key, 1024)                    # far below the need: the next call overflows and is redone
        rasterizer.drop_async_workspaces()
        again = render(cam, pc, pipe, bg)
        info = rasterizer.last_forward_info()
        assert int(info["used_max_instances"]) >= need > 1024      # the capacity did grow: the branch ran
        for k in ("render", "depth", "radii", "visibility_filter"):
            assert torch.equal(again[k], first[k]), k
        assert int(again["visibility_filter"].sum()) == int((first["radii"] > 0).sum()) > 0


@pytest.mark.gpu
def test_render_returns_behind_the_tile_scan_with_stream_ordered_outputs(gpu_device):
    """render() waits for the call's STATUS WORDS only (pgr_forward_posed_early_status: final behind the tile scan) and returns
    while scatter, sort and compositor still run.  What the caller does next is ordered behind them by the stream: a clone
    queued straight after the return, a second render() into the SAME workspace, and an in-place edit of the model between two
    calls must all see complete images -- equal to the synchronous batch call's."""
    import sys
    from argparse import ArgumentParser
    import torch
    root = str(Path(__file__).resolve().parents[1])
    for p in (root, root + "/compat"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gaussian_renderer import render, GaussianModel
    from scene.cameras import Camera
    from arguments import PipelineParams
    from pegasus_amd import rasterizer as R, scenes
    dev = torch.device(gpu_device)
    cloud, views = scenes.scene_c3(scale=0.05, n_views=3, width=400, height=304)
    with torch.no_grad():
        pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling,
                                       cloud.rotation, device=dev)
        cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
                       image_height=v.height, gt_alpha_mask=None, image_name=str(i), uid=i, data_device=str(dev))
                for i, v in enumerate(views)]
        pipe = PipelineParams(ArgumentParser())
        bg = torch.tensor([0.1, 0.2, 0.3], device=dev)

        def sync_reference(cam):
            spec = R.ViewSpec(int(cam.image_height), int(cam.image_width), math.tan(0.5 * cam.FoVx), math.tan(0.5 * cam.FoVy),
                              bg, cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
            r = R.forward_views(pc.get_xyz, pc.get_opacity, [spec], shs=pc.get_features, scales=pc.get_scaling,
                                rotations=pc.get_rotation, sh_degree=pc.active_sh_degree, want_radii=True)[0]
            torch.cuda.synchronize()
            return {k: r[k].clone() for k in ("color", "depth", "radii")}

        want = [sync_reference(c) for c in cams]
        for rep in range(3):                       # back to back, nothing synchronises in between
            got = []
            for c in cams:
                out = render(c, pc, pipe, bg)
                got.append((out["render"].clone(), out["depth"].clone(), out["radii"].clone(), out["visibility_filter"].clone()))
            for g, w in zip(got, want):
                assert torch.equal(g[0], w["color"]) and torch.equal(g[1], w["depth"]) and torch.equal(g[2], w["radii"])
                assert torch.equal(g[3], w["radii"] > 0)
        info = R.last_forward_info()
        assert int(info["num_instances"][0]) > 0 and info["n_views"] == 1
        # (nine renders of the unchanged model: the first three read the SH coefficients as stored -- _features_dc +
        #  _features_rest, PgrScene::shs_rest --, from the fourth on the kept concatenation: all equal the reference above)
        from pegasus_amd import gaussian_renderer as GR
        assert "get_features" in GR._SLOTS[pc] and GR._SLOTS[pc]["split_renders"][1] == GR.SPLIT_RENDERS
        # an in-place edit of the model right behind a call lands AFTER that call's preprocess in stream order
        before = render(cams[0], pc, pipe, bg)["render"]
        pc._xyz[:, 2] += 0.05
        moved = render(cams[0], pc, pipe, bg)["render"].clone()
        pc._xyz[:, 2] -= 0.05
        assert torch.equal(before, want[0]["color"])
        assert not torch.equal(moved, want[0]["color"])
