"""Hierarchical culling (pegasus_amd/csrc/blockcull.hip.h): a 64-Gaussian block is skipped in a view only if every
one of its Gaussians would be culled by the exact per-Gaussian code.  Checked directly (no visible Gaussian ever sits
in a block declared invisible -- on coherent, incoherent and hostile scenes) and end to end (radii, per-tile lists and
images bit-identical with the test switched off)."""
import os

import numpy as np
import pytest

from pegasus_amd import scenes

pytestmark = pytest.mark.gpu


def _specs(views, dev):
    import torch
    from pegasus_amd import rasterizer as R
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    return [R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.zeros(3)), t(v.world_view_transform),
                       t(v.full_proj_transform), t(v.camera_center)) for v in views]


def _render(act_t, specs, **kw):
    import torch
    from pegasus_amd import rasterizer as R
    res = R.forward_views(act_t["means3d"], act_t["opacities"], specs, shs=act_t["shs"], scales=act_t["scales"],
                          rotations=act_t["rotations"], sh_degree=3, want_radii=True, want_aux=True, **kw)
    torch.cuda.synchronize()
    return res


def _with_cull(flag, fn):
    old = os.environ.get("PGR_BLOCK_CULL")
    os.environ["PGR_BLOCK_CULL"] = "1" if flag else "0"
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("PGR_BLOCK_CULL", None)
        else:
            os.environ["PGR_BLOCK_CULL"] = old


@pytest.mark.parametrize("layout", ["morton", "input", "shuffled"])
def test_no_visible_gaussian_in_an_invisible_block(gpu_device, layout):
    """C3 at 5 % scale, 40 cameras: radii from the exact path (test off) against pgr_block_visibility's bits."""
    import torch
    from pegasus_amd import rasterizer as R
    from pegasus_amd.scene_order import spatial_order
    cloud, views = scenes.scene_c3(scale=0.05, n_views=40, width=400, height=304)
    act = cloud.activated()
    n = cloud.n
    if layout == "morton":
        perm = spatial_order(act["means3d"], cloud.object_id)
    elif layout == "shuffled":
        perm = np.random.default_rng(0).permutation(n)
    else:
        perm = np.arange(n)
    act_t = {k: torch.from_numpy(np.ascontiguousarray(a[perm])).to(gpu_device) for k, a in act.items()}
    specs = _specs(views, gpu_device)
    vis = R.block_visibility(act_t["means3d"], specs, scales=act_t["scales"], rotations=act_t["rotations"])
    assert vis.shape == ((n + 63) // 64, len(specs))
    res = _with_cull(False, lambda: _render(act_t, specs))
    pad = (-n) % 64
    for v, r in enumerate(res):
        seen = torch.nn.functional.pad(r["radii"] > 0, (0, pad)).reshape(-1, 64).any(dim=1)
        assert not bool((seen & ~vis[:, v]).any()), f"view {v}: a listed Gaussian sits in a culled block"
    culled = 1.0 - float(vis.float().mean())
    if layout == "morton":
        assert culled > 0.25, culled          # the point of the exercise: a good share of the plane is off-screen
    if layout == "shuffled":
        assert culled < 0.05, culled          # incoherent order: boxes span the scene, nothing to cull, nothing wrong


@pytest.mark.parametrize("seed", [0, 1])
def test_hostile_blocks_are_never_culled_wrongly(gpu_device, seed):
    """NaN / Inf positions, huge, vanishing and zero scales, unnormalised and zero quaternions, cameras inside the
    cloud and looking away from it."""
    import torch
    from pegasus_amd import graphics as G, rasterizer as R
    rng = np.random.default_rng(50 + seed)
    n = 20000
    cloud, views = scenes.scene_c1(seed=60 + seed, n=n)
    act = cloud.activated()
    act["means3d"] = (act["means3d"] * rng.choice([0.2, 1.0, 30.0], size=(n, 1))).astype(np.float32)
    order = np.argsort(act["means3d"][:, 0] + 10 * np.round(act["means3d"][:, 1]))    # some coherence, some not
    act = {k: a[order] for k, a in act.items()}
    pick = lambda frac: rng.random(n) < frac
    act["means3d"][pick(0.002)] = np.nan
    act["means3d"][pick(0.002), 1] = np.inf
    act["scales"][pick(0.01)] = np.float32(1e3)
    act["scales"][pick(0.01)] = np.float32(1e-8)
    act["scales"][pick(0.01), 2] = 0.0
    act["rotations"][pick(0.01)] = 0.0
    act["rotations"][pick(0.03)] *= np.float32(37.0)
    act_t = {k: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device) for k, a in act.items()}
    cams = []
    for k in range(24):
        eye = rng.normal(0, [3.0, 3.0, 3.0][k % 3], size=3)
        at = rng.normal(0, 2.0, size=3) if k % 4 else eye + rng.normal(0, 1.0, size=3)      # some look away
        Rm, t = G.look_at_opencv(tuple(eye), tuple(at), up=(0, -1, 0))
        cams.append(scenes.make_view(Rm, t, 320, 208, fovx=np.radians(60), fovy=np.radians(41)))
    specs = _specs(cams, gpu_device)
    vis = R.block_visibility(act_t["means3d"], specs, scales=act_t["scales"], rotations=act_t["rotations"])
    off = _with_cull(False, lambda: _render(act_t, specs))
    on = _with_cull(True, lambda: _render(act_t, specs))
    pad = (-n) % 64
    for v in range(len(specs)):
        seen = torch.nn.functional.pad(off[v]["radii"] > 0, (0, pad)).reshape(-1, 64).any(dim=1)
        assert not bool((seen & ~vis[:, v]).any()), v
        assert torch.equal(on[v]["radii"], off[v]["radii"])
        assert torch.equal(on[v]["n_contrib"], off[v]["n_contrib"])
        a, b = on[v]["color"], off[v]["color"]
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert 0.02 < 1.0 - float(vis.float().mean()) < 0.98


def test_frames_are_bit_identical_with_and_without_block_culling(gpu_device):
    """FrameRenderer (Morton layout, fused semantic image, masks) and the posed path: same bits either way; posed
    objects' blocks are never culled."""
    import torch
    from pegasus_amd import frames as F
    from pegasus_amd.compose import pose_table
    cloud, views = scenes.scene_c3(scale=0.04, n_views=6, width=480, height=352)
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                         sh_degree=3, device=gpu_device)
    specs = [fr.view_spec(v) for v in views]
    keys = ("color", "depth", "seg", "seg_depth", "masks")
    snap = lambda f: {k: f[k].clone() for k in keys}
    on = _with_cull(True, lambda: snap(fr.render_frames(specs)))
    off = _with_cull(False, lambda: snap(fr.render_frames(specs)))
    for k in keys:
        assert torch.equal(on[k], off[k]), k
    assert float(on["masks"].sum()) > 0
    # dynamic: every object flies far out of its rest box (a culled rest box must not hide it)
    Rz = lambda a: np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    poses = np.zeros((len(specs), fr.K, 20), np.float32)
    oid = cloud.object_id
    centers = [act["means3d"][oid == k].mean(0) for k in range(1, fr.K + 1)]
    for s in range(len(specs)):
        pairs = []
        for k in range(fr.K):
            T = Rz(0.4 * s + k)
            T[:3, 3] = [0.6 * np.sin(s + k), 0.6 * np.cos(2 * s + k), 0.05 * s]
            pairs.append((T, centers[k]))
        poses[s] = pose_table(pairs)
    on = _with_cull(True, lambda: snap(fr.render_frames(specs, poses=poses)))
    off = _with_cull(False, lambda: snap(fr.render_frames(specs, poses=poses)))
    for k in keys:
        assert torch.equal(on[k], off[k]), ("posed", k)


def test_more_than_64_views_per_batch(gpu_device):
    """The visibility words of a batch are built 64 views per ballot: 70 views cross a ballot and a word boundary."""
    import torch
    from pegasus_amd import rasterizer as R
    from pegasus_amd.scene_order import spatial_order
    cloud, views = scenes.scene_c3(scale=0.01, n_views=70, width=160, height=112)
    act = cloud.activated()
    perm = spatial_order(act["means3d"], cloud.object_id)
    act_t = {k: torch.from_numpy(np.ascontiguousarray(a[perm])).to(gpu_device) for k, a in act.items()}
    specs = _specs(views, gpu_device)
    on = _with_cull(True, lambda: _render(act_t, specs))
    off = _with_cull(False, lambda: _render(act_t, specs))
    vis = R.block_visibility(act_t["means3d"], specs, scales=act_t["scales"], rotations=act_t["rotations"])
    assert vis.shape[1] == 70 and 0.05 < 1.0 - float(vis.float().mean()) < 0.95
    for v in range(70):
        assert torch.equal(on[v]["radii"], off[v]["radii"]), v
        assert torch.equal(on[v]["n_contrib"], off[v]["n_contrib"]), v
        assert torch.equal(on[v]["color"], off[v]["color"]), v
