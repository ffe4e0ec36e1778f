"""Edges of the C-ABI contract that the parity scenes never reach: a view whose instance total passes 2^32, and a
semantic descriptor that is rejected before anything is enqueued.  (The reference extension sizes its buffers on the
host after a device->host read of the instance total -- SURVEY.md section 2a -- and simply fails to allocate; here the
total is computed on the device, so the device must not wrap.)"""
import numpy as np
import pytest

from pegasus_amd import scenes


def test_capacity_growth_gives_up_beyond_the_per_view_limit():
    from pegasus_amd import rasterizer as R
    assert R.grown_capacity(1_000_000, 1.25) == 1_251_024
    assert R.grown_capacity(R.MAX_INSTANCES, 1.6) == R.MAX_INSTANCES           # clamped, still renderable
    with pytest.raises(RuntimeError, match="per-view limit"):
        R.grown_capacity(R.MAX_INSTANCES + 1, 1.25)
    with pytest.raises(RuntimeError, match=">= 4294967295"):                       # the device's saturated count
        R.grown_capacity(0xFFFFFFFF, 1.25)


@pytest.mark.gpu
def test_instance_total_just_beyond_32_bits_reports_overflow(gpu_device):
    """1.72 M opaque splats that each cover all 2500 tiles of an 800x800 view: 4.3e9 instances = 2^32 + 5 M.  A 32-bit
    running total would wrap to 5 M -- below the 10 M capacity -- and the scatter pass would write through wrapped
    ranges.  The device total is 64-bit: overflow is flagged, nothing is scattered, the host gives up cleanly, and the
    library keeps working afterwards."""
    import torch
    from pegasus_amd import rasterizer as R
    n, W, H = 1_720_000, 800, 800
    assert n * ((W + 15) // 16) * ((H + 15) // 16) - (1 << 32) < 6 * n      # wrapped total < initial capacity
    dev = gpu_device
    g = torch.Generator(device="cpu").manual_seed(0)
    means = (torch.rand((n, 3), generator=g) - 0.5).mul_(0.2).to(dev)
    op = torch.full((n,), 0.99, device=dev)
    scales = torch.full((n, 3), 100.0, device=dev)
    rot = torch.zeros((n, 4), device=dev); rot[:, 0] = 1.0
    col = torch.rand((n, 3), generator=g).to(dev)
    _, views = scenes.scene_c1()
    v = scenes.make_view(views[0].R_c2w.T, views[0].t_w2c, W, H, fovx=views[0].fovx, fovy=views[0].fovy)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    spec = R.ViewSpec(H, W, v.tanfovx, v.tanfovy, t(np.zeros(3)), t(v.world_view_transform), t(v.full_proj_transform),
                      t(v.camera_center))
    R.set_capacity_hint((dev, n, W, H), None)
    with pytest.raises(RuntimeError, match="per-view limit"):
        R.forward_views(means, op, [spec], colors_precomp=col, scales=scales, rotations=rot, want_radii=False)
    torch.cuda.synchronize()
    # still alive: a small well-behaved subset renders, and to the same image twice
    k = 20_000
    small = torch.full((k, 3), 0.02, device=dev)
    a = R.forward_views(means[:k], op[:k], [spec], colors_precomp=col[:k], scales=small, rotations=rot[:k])[0]["color"].clone()
    b = R.forward_views(means[:k], op[:k], [spec], colors_precomp=col[:k], scales=small, rotations=rot[:k])[0]["color"]
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(a, b) and float(a.max()) > 0


@pytest.mark.gpu
def test_semantic_descriptor_is_checked_before_any_enqueue(gpu_device):
    """NULL colours / object ids, or a view without a sem_color target, are PGR_ERR_INVALID_ARGUMENT up front -- in
    round 1 they were found at stage 4, with the tables copy and three stages already on the stream."""
    import torch
    from pegasus_amd import rasterizer as R
    dev = gpu_device
    cloud, views = scenes.scene_c1(n=2000)
    act = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in cloud.activated().items()}
    v = views[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    spec = R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.zeros(3)), t(v.world_view_transform),
                      t(v.full_proj_transform), t(v.camera_center))
    oid = torch.zeros(cloud.n, dtype=torch.int32, device=dev); oid[1000:] = 1
    good = dict(object_id=oid, colors=torch.tensor([[0.2, 0.7, 0.4]], device=dev), n_env=1000, k=1)
    kw = dict(shs=act["shs"], scales=act["scales"], rotations=act["rotations"], sh_degree=3, want_radii=False)
    ok = R.forward_views(act["means3d"], act["opacities"], [spec], semantic=good, **kw)[0]
    assert float(ok["sem_color"].abs().max()) > 0
    for bad in (dict(good, colors=None), dict(good, object_id=None), dict(good, k=0), dict(good, n_env=-1)):
        with pytest.raises(ValueError, match="invalid argument"):
            R.forward_views(act["means3d"], act["opacities"], [spec], semantic=bad, **kw)
    outs = [dict(color=torch.empty((3, v.height, v.width), device=dev), depth=torch.empty((1, v.height, v.width), device=dev),
                 radii=None)]                                   # no sem_color target
    with pytest.raises(ValueError, match="invalid argument"):
        R.forward_views(act["means3d"], act["opacities"], [spec], semantic=good, outputs=outs, **kw)
    # records-only views (round 6): no image at all is fine WITH a record -- but the semantic image can only be left out when
    # the descriptor names the colours its mask planes are thresholded against; depth without colour is no shape at all
    from pegasus_amd import masks as M
    rec = lambda k: torch.empty((M.record_layout(v.height, v.width, k)["bytes"],), dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError, match="invalid argument"):
        R.forward_views(act["means3d"], act["opacities"], [spec], outputs=[dict(radii=None)], **kw)                # nothing to write
    with pytest.raises(ValueError, match="invalid argument"):
        R.forward_views(act["means3d"], act["opacities"], [spec], semantic=good, outputs=[dict(radii=None, record=rec(0))], **kw)
    with pytest.raises(ValueError, match="invalid argument"):
        R.forward_views(act["means3d"], act["opacities"], [spec],
                        outputs=[dict(radii=None, record=rec(0), depth=torch.empty((1, v.height, v.width), device=dev))], **kw)
    r0 = dict(radii=None, record=rec(0))
    R.forward_views(act["means3d"], act["opacities"], [spec], outputs=[r0], **kw)                                  # raster-only record
    withm = dict(good, mask_colors=torch.tensor([[0.2, 0.7, 0.4]], device=dev), mask_threshold=0.1)
    r1 = dict(radii=None, record=rec(1))
    R.forward_views(act["means3d"], act["opacities"], [spec], semantic=withm, outputs=[r1], **kw)
    torch.cuda.synchronize()
    want = M.pack_records(ok["color"][None], ok["depth"][None])
    assert torch.equal(M.record_views(r0["record"], v.height, v.width, 0)["rgb"], M.record_views(want[0], v.height, v.width, 0)["rgb"])
    assert torch.equal(M.record_views(r1["record"], v.height, v.width, 1)["rgb"], M.record_views(want[0], v.height, v.width, 0)["rgb"])
    again = R.forward_views(act["means3d"], act["opacities"], [spec], semantic=good, **kw)[0]
    torch.cuda.synchronize()
    assert torch.equal(again["sem_color"], ok["sem_color"])


@pytest.mark.gpu
def test_batch_in_which_only_some_views_overflow(gpu_device):
    """A batch whose instance capacity is enough for some of its views and not for others: the overflowed views are left out
    of the sort queues and the compositor (nothing reads past the buffers), the others render, the host sees the flags, and
    the transparent re-render ends with the frames a roomy first attempt gives."""
    import numpy as np
    import torch
    from pegasus_amd import frames as F, rasterizer, scenes
    cloud, views = scenes.scene_c3(scale=0.04, n_views=4, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                         cloud.object_id, sh_degree=3, device=gpu_device)
    specs = [fr.view_spec(v) for v in views]
    ref = {k: v.clone() for k, v in fr.render_frames(specs).items()}
    torch.cuda.synchronize()
    need = []
    for sp in specs:                                       # every view alone: its own instance count
        rasterizer.forward_views(fr.means3d, fr.opacities, [sp], shs=fr.shs, scales=fr.scales, rotations=fr.rotations,
                                 sh_degree=fr.sh_degree, tie_index=fr.tie_index, want_radii=False)
        need.append(int(rasterizer.last_forward_info()["num_instances"][0]))
    need.sort()
    assert need[0] < need[-1], need
    between = (need[0] + need[-1]) // 2                    # room for the lightest view, not for the heaviest
    rasterizer.reset_capacity(between)
    out = fr.alloc_frames(len(specs), 240, 320)
    fr.render_frames_async(specs, out, slot=0).wait()
    torch.cuda.synchronize()
    for k in ("color", "depth", "seg", "masks"):
        assert torch.equal(out[k], ref[k]), k


@pytest.mark.gpu
def test_early_status_reports_every_view_of_a_batch(gpu_device):
    """pgr_forward_posed_early_status with FOUR views: the tile scan of every view stores that view's instance count and
    overflow flag into the pinned host scratch, the event behind the scan releases the host, and the status it reads is the
    batch's -- the counts of every view, and PGR_ERR_INSTANCE_OVERFLOW (re-render at the grown capacity) when only some views
    do not fit.  Images equal the synchronous call's."""
    import torch
    from pegasus_amd import frames as F, rasterizer as R, scenes
    cloud, views = scenes.scene_c3(scale=0.04, n_views=4, width=320, height=240)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                         cloud.object_id, sh_degree=3, device=gpu_device)
    specs = [fr.view_spec(v) for v in views]
    kw = dict(shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=fr.sh_degree, tie_index=fr.tie_index, want_radii=False)
    ref = R.forward_views(fr.means3d, fr.opacities, specs, **kw)
    torch.cuda.synchronize()
    need = [int(x) for x in R.last_forward_info()["num_instances"]]
    ref = [{k: r[k].clone() for k in ("color", "depth")} for r in ref]
    assert min(need) < max(need)
    pb = R.forward_views(fr.means3d, fr.opacities, specs, async_slot=("early-batch", 0), early_status=True, **kw)
    got = pb.wait()
    assert pb.num_instances == need and not getattr(pb, "_was_redone", False)
    for g, r in zip(got, ref):
        assert torch.equal(g["color"], r["color"]) and torch.equal(g["depth"], r["depth"])
    # room for the lightest view only: the flags of the others arrive with the early status, the batch is rendered again
    key = (torch.device(gpu_device), int(fr.means3d.shape[0]), 320, 240)
    R.set_capacity_hint(key, (min(need) + max(need)) // 2)
    R.drop_async_workspaces()
    pb = R.forward_views(fr.means3d, fr.opacities, specs, async_slot=("early-batch", 0), early_status=True, **kw)
    got = pb.wait()
    torch.cuda.synchronize()
    assert getattr(pb, "_was_redone", False) and pb.num_instances == need
    for g, r in zip(got, ref):
        assert torch.equal(g["color"], r["color"]) and torch.equal(g["depth"], r["depth"])
