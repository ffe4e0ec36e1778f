"""Scene composition: SH band rotation (CPU) and the HIP pose kernel (GPU)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot


def test_sh_rotation_defining_property():
    """Rotating the coefficients with the object == evaluating the original radiance at the inverse-rotated
    direction: f'(d) = f(R^T d); the matrices are orthogonal and compose like the rotations."""
    from pegasus_amd.sh_rotation import rotate_sh_rest, sh_rotation_matrices
    from pegasus_amd.sh_utils import sh_basis
    rng = np.random.default_rng(0)
    for seed in range(4):
        R = Rot.random(random_state=seed).as_matrix()
        c = rng.normal(size=(7, 16, 3))
        cr = np.concatenate([c[:, :1], rotate_sh_rest(c[:, 1:], R)], 1)
        d = rng.normal(size=(500, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        lhs = np.einsum("km,nmc->nkc", sh_basis(3, d), cr)
        rhs = np.einsum("km,nmc->nkc", sh_basis(3, d @ R), c)
        assert np.abs(lhs - rhs).max() < 1e-12
        for D in sh_rotation_matrices(R):
            assert np.abs(D @ D.T - np.eye(D.shape[0])).max() < 1e-12
    R1, R2 = Rot.random(random_state=10).as_matrix(), Rot.random(random_state=11).as_matrix()
    for A, B, Cm in zip(sh_rotation_matrices(R2 @ R1), sh_rotation_matrices(R2), sh_rotation_matrices(R1)):
        assert np.abs(A - B @ Cm).max() < 1e-12
    for D in sh_rotation_matrices(np.eye(3)):
        assert np.abs(D - np.eye(D.shape[0])).max() < 1e-12


def test_band1_is_the_rotation_in_yzx_order():
    """Band 1 of the 3DGS basis is (-y, z, -x) * C1: D1 must be R conjugated by that signed permutation."""
    from pegasus_amd.sh_rotation import sh_rotation_matrices
    R = Rot.random(random_state=5).as_matrix()
    S = np.array([[0, -1, 0], [0, 0, 1], [-1, 0, 0]], float)     # (x,y,z) -> (-y, z, -x)
    np.testing.assert_allclose(sh_rotation_matrices(R)[0], S @ R @ S.T, atol=1e-12)


@pytest.mark.gpu
def test_compose_kernel_matches_reference(gpu_device):
    import torch
    from oracle.compose_ref import compose_object_ref, quat_to_matrix
    from pegasus_amd import compose
    from pegasus_amd.sh_rotation import rotate_sh_rest
    rng = np.random.default_rng(1)
    n = 5000
    xyz = rng.normal(size=(n, 3)).astype(np.float32)
    rot = rng.normal(size=(n, 4)).astype(np.float32) * rng.uniform(0.2, 3, size=(n, 1)).astype(np.float32)
    f_rest = rng.normal(size=(n, 15, 3)).astype(np.float32)
    T = np.eye(4); T[:3, :3] = Rot.random(random_state=2).as_matrix(); T[:3, 3] = [0.3, -0.2, 0.7]
    dev = gpu_device
    t = lambda a: torch.from_numpy(a).to(dev)
    pose = compose.make_pose(T, xyz.astype(np.float64).mean(0))
    o_xyz, o_rot, o_rest = torch.empty((n, 3), device=dev), torch.empty((n, 4), device=dev), torch.empty((n, 15, 3), device=dev)
    compose.compose_object(t(xyz), t(rot), t(f_rest), pose, o_xyz, o_rot, o_rest)
    torch.cuda.synchronize()
    ref_xyz, ref_R = compose_object_ref(xyz, rot, T)
    np.testing.assert_allclose(o_xyz.cpu().numpy(), ref_xyz, atol=2e-6)
    got_R = np.stack([quat_to_matrix(q) for q in o_rot.cpu().numpy().astype(np.float64)])
    np.testing.assert_allclose(got_R, ref_R, atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(o_rot.cpu().numpy(), axis=1), 1.0, atol=1e-6)
    np.testing.assert_allclose(o_rest.cpu().numpy(), rotate_sh_rest(f_rest, T[:3, :3]), atol=1e-5)


@pytest.mark.gpu
def test_scene_composer_renders_like_a_host_merged_cloud(oracle, gpu_device):
    """SceneComposer(env, objects).set_poses -> rasterize == oracle on a cloud merged and posed on the host."""
    import torch
    from helpers import assert_images_match
    from oracle.compose_ref import compose_object_ref
    from pegasus_amd import compose, rasterizer, scenes
    from pegasus_amd.sh_rotation import rotate_sh_rest
    rng = np.random.default_rng(4)
    env = scenes.ground_plane(rng, 20000, 1.0, np.log(0.006), 0.4, 0.1, 0.005)
    objs = [scenes.box_object(rng, 6000, (0.1, 0.15, 0.2), np.log(0.004), 0.3, 0.2, k + 1) for k in range(2)]
    poses = []
    for k in range(2):
        T = np.eye(4); T[:3, :3] = Rot.random(random_state=20 + k).as_matrix(); T[:3, 3] = [0.2 * k - 0.1, 0.1, 0.12]
        poses.append(T)
    comp = compose.SceneComposer(env, objs, device=gpu_device)
    comp.set_poses(poses)
    # host-side merge (float64 maths, then the same fp32 activations)
    posed = []
    for o, T in zip(objs, poses):
        x, Rm = compose_object_ref(o.xyz, o.rotation, T)
        q = Rot.from_matrix(Rm).as_quat()
        posed.append(scenes.SplatCloud(x.astype(np.float32), o.features_dc,
                                       rotate_sh_rest(o.features_rest, T[:3, :3]).astype(np.float32), o.opacity,
                                       o.scaling, np.concatenate([q[:, 3:], q[:, :3]], 1).astype(np.float32),
                                       o.object_id))
    merged = scenes.SplatCloud.concat([env, *posed])
    act = merged.activated()
    _, views = scenes.scene_c3(scale=0.001, n_views=2, width=320, height=240, camera_set="fibonacci_above_9deg")
    for v in views:
        spec = rasterizer.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, torch.zeros(3, device=gpu_device),
                                   *(torch.from_numpy(a).to(gpu_device) for a in
                                     (v.world_view_transform, v.full_proj_transform, v.camera_center)))
        r = rasterizer.forward_views(comp.means3d, comp.opacities, [spec], shs=comp.shs, scales=comp.scales,
                                     rotations=comp.rotations, sh_degree=3)[0]
        torch.cuda.synchronize()
        o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8)
        # inputs differ by fp32-vs-fp64 posing (~1e-6), so compare images at 1e-3 and require visible content
        assert np.abs(r["color"].cpu().numpy() - o["color"]).mean() < 1e-4
        assert np.quantile(np.abs(r["color"].cpu().numpy() - o["color"]), 0.999) < 5e-3
        assert (o["color"] > 0.05).mean() > 0.2


def _posed_case(seed=3):
    """Small merged scene + K poses; returns (cloud, view, poses list of (T, center), numpy-composed activated scene)."""
    from scipy.spatial.transform import Rotation as Rot
    from pegasus_amd import scenes
    from pegasus_amd.sh_rotation import sh_rotation_matrices
    cloud, views = scenes.scene_c3(seed=seed, scale=0.01, n_views=1, width=320, height=240)
    oid = cloud.object_id
    K = int(oid.max())
    rng = np.random.default_rng(seed)
    act = cloud.activated()
    comp = {k: v.copy() for k, v in act.items()}
    poses = []
    for k in range(1, K + 1):
        sel = np.nonzero(oid == k)[0]
        R = Rot.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix()
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.normal(0, 0.03, 3)
        center = act["means3d"][sel].mean(0)
        poses.append((T, center))
        # composed copy, float64 then rounded: x' = R (x - c) + c + t, q' = q_R (x) q, SH bands rotated
        x = act["means3d"][sel].astype(np.float64)
        comp["means3d"][sel] = ((R @ (x - center).T).T + center + T[:3, 3]).astype(np.float32)
        qR = Rot.from_matrix(R).as_quat()          # x,y,z,w
        q = act["rotations"][sel].astype(np.float64)
        qq = Rot.from_quat(np.concatenate([q[:, 1:], q[:, :1]], axis=1))
        prod = (Rot.from_quat(qR) * qq).as_quat()
        comp["rotations"][sel] = np.concatenate([prod[:, 3:], prod[:, :3]], axis=1).astype(np.float32)
        D1, D2, D3 = sh_rotation_matrices(R)
        sh = act["shs"][sel].astype(np.float64)    # [n,16,3]
        sh[:, 1:4] = np.einsum("rk,nkc->nrc", D1, sh[:, 1:4])
        sh[:, 4:9] = np.einsum("rk,nkc->nrc", D2, sh[:, 4:9])
        sh[:, 9:16] = np.einsum("rk,nkc->nrc", D3, sh[:, 9:16])
        comp["shs"][sel] = sh.astype(np.float32)
    return cloud, views[0], poses, act, comp


def test_posed_objects_equal_composed_scene(oracle):
    """Posing objects inside the preprocess (oracle `object_id` + `poses`) renders the image of the explicitly
    composed scene: positions / orientations by the same formula, colour = unrotated SH in the object's frame."""
    from pegasus_amd.compose import pose_table
    cloud, v, poses, act, comp = _posed_case()
    table = pose_table(poses)
    a = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, object_id=cloud.object_id, poses=table)
    b = oracle.forward(**comp, sh_degree=3, **v.raster_kwargs(), num_threads=8)
    moved = np.abs(b["color"] - oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8)["color"]).max()
    assert moved > 0.05                                   # the poses do change the picture
    diff = np.abs(a["color"] - b["color"])
    assert np.percentile(diff, 99.5) < 2e-3 and (diff > 2e-2).mean() < 2e-3, (diff.max(), (diff > 2e-2).mean())
    assert np.abs(a["out_depth"] - b["out_depth"]).mean() < 1e-3
    assert (a["radii"] > 0).sum() == pytest.approx((b["radii"] > 0).sum(), rel=0.01)


@pytest.mark.gpu
def test_posed_batch_matches_oracle(oracle, gpu_device):
    """A batch of TIME STEPS (one camera each, its own object poses) through pgr_forward_posed_async == the oracle's
    posed preprocess per step: radii and n_contrib bit-exact, images within 1e-4."""
    import torch
    from pegasus_amd import rasterizer as R
    from pegasus_amd.compose import pose_table
    from pegasus_amd.frames import FrameRenderer
    cloud, v, poses0, act, _ = _posed_case(seed=5)
    _, _, poses1, _, _ = _posed_case(seed=6)
    steps = [pose_table(poses0), pose_table(poses1), pose_table(poses0[::-1])]
    dev = gpu_device
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    spec = R.ViewSpec(v.height, v.width, v.tanfovx, v.tanfovy, t(np.zeros(3, np.float32)), t(v.world_view_transform),
                      t(v.full_proj_transform), t(v.camera_center))
    posed = dict(object_id=t(cloud.object_id, torch.int32), poses=t(np.stack(steps)))
    res = R.forward_views(t(act["means3d"]), t(act["opacities"]), [spec] * 3, shs=t(act["shs"]), scales=t(act["scales"]),
                          rotations=t(act["rotations"]), sh_degree=3, want_radii=True, want_aux=True, posed=posed)
    torch.cuda.synchronize()
    for r, table in zip(res, steps):
        o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1,
                           object_id=cloud.object_id, poses=table)
        np.testing.assert_array_equal(r["radii"].cpu().numpy(), o["radii"])
        amb = o["ambig"].astype(bool)
        np.testing.assert_array_equal(r["n_contrib"].cpu().numpy()[~amb], o["n_contrib"][~amb])
        assert np.abs(r["color"].cpu().numpy() - o["color"])[:, ~amb].max() <= 1e-4
        assert np.abs(r["depth"].cpu().numpy() - o["out_depth"])[:, ~amb].max() <= 1e-4
    assert not torch.equal(res[0]["color"], res[1]["color"])
    # the frame path: same steps through FrameRenderer (Morton-ordered resident copy), fused semantic image included
    fr = FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                       device=dev)
    f = fr.render_frames([fr.view_spec(v)] * 3, poses=np.stack(steps))
    for i, r in enumerate(res):
        assert (f["color"][i] - r["color"]).abs().max().item() < 1e-6
    assert f["masks"].sum().item() > 0
