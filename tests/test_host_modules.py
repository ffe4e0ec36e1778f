"""Host-side modules either side of the render path: PLY I/O, BOP pose records, pose interpolation golden."""
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"


def test_ply_roundtrip(tmp_path):
    from pegasus_amd.ply_io import read_ply_vertices, write_ply_vertices
    rng = np.random.default_rng(0)
    cols = {k: rng.normal(size=17).astype(np.float32) for k in ["x", "y", "z", "opacity", "f_dc_0", "rot_3"]}
    write_ply_vertices(tmp_path / "a" / "pc.ply", cols)
    v = read_ply_vertices(tmp_path / "a" / "pc.ply")
    assert v.dtype.names == tuple(cols)
    for k in cols:
        np.testing.assert_array_equal(v[k], cols[k])
    # ascii variant
    txt = "ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty uchar r\nend_header\n1.5 7\n-2 255\n"
    (tmp_path / "b.ply").write_text(txt)
    v = read_ply_vertices(tmp_path / "b.ply")
    assert v["x"].tolist() == [1.5, -2.0] and v["r"].tolist() == [7, 255]


def test_gaussian_model_ply_layout(tmp_path):
    """f_rest is stored channel-major in the file and coefficient-major in memory (gaussian_model.py:252-268)."""
    from pegasus_amd.gaussian_model import GaussianModel
    rng = np.random.default_rng(1)
    n = 11
    m = GaussianModel.from_arrays(rng.normal(size=(n, 3)), rng.normal(size=(n, 1, 3)), rng.normal(size=(n, 15, 3)),
                                  rng.normal(size=(n, 1)), rng.normal(size=(n, 3)), rng.normal(size=(n, 4)),
                                  device="cpu")
    m.save_ply(tmp_path / "pc.ply")
    from pegasus_amd.ply_io import read_ply_vertices
    v = read_ply_vertices(tmp_path / "pc.ply")
    assert len(v.dtype.names) == 62
    # f_rest_k = channel (k // 15), coefficient (k % 15)
    np.testing.assert_array_equal(v["f_rest_16"], m._features_rest[:, 1, 1].numpy())
    m2 = GaussianModel(3, device="cpu")
    m2.load_ply(tmp_path / "pc.ply")
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        np.testing.assert_array_equal(getattr(m2, k).numpy(), getattr(m, k).numpy())
    assert m2.get_features.shape == (n, 16, 3) and m2.active_sh_degree == 3
    np.testing.assert_allclose(m2.get_opacity.numpy(), 1 / (1 + np.exp(-m._opacity.numpy())), rtol=1e-6)
    cov = m2.get_covariance().numpy()
    assert cov.shape == (n, 6) and (cov[:, [0, 3, 5]] > 0).all()


def test_bop_pose_record():
    from pegasus_amd import bop_pose
    from scipy.spatial.transform import Rotation as Rot
    R_c2w = Rot.random(random_state=1).as_matrix()
    t = np.array([0.1, -0.2, 1.5])
    T_m2w = np.eye(4); T_m2w[:3, :3] = Rot.random(random_state=2).as_matrix(); T_m2w[:3, 3] = [0.3, 0.1, 0.05]
    e = bop_pose.scene_gt_entry(R_c2w, t, {7: T_m2w})[0]
    assert e["obj_id"] == 7
    Rm2c = np.array(e["cam_R_m2c"]).reshape(3, 3)
    np.testing.assert_allclose(Rm2c, R_c2w.T @ T_m2w[:3, :3], atol=1e-12)
    np.testing.assert_allclose(e["cam_t_m2c"], R_c2w.T @ T_m2w[:3, 3] + t, atol=1e-12)
    # a model point lands where the camera sees it
    p = np.array([0.02, -0.03, 0.04])
    np.testing.assert_allclose(Rm2c @ p + e["cam_t_m2c"], R_c2w.T @ (T_m2w[:3, :3] @ p + T_m2w[:3, 3]) + t, atol=1e-12)
    K = np.array(bop_pose.scene_camera_entry(0.8, 0.6, 640, 480)["cam_K"]).reshape(3, 3)
    assert K[0, 2] == 320 and K[1, 2] == 240 and abs(K[0, 0] - 320 / np.tan(0.4)) < 1e-9
    assert bop_pose.scene_gt_entry(R_c2w, t, {7: T_m2w}, translation_scale=1000.0)[0]["cam_t_m2c"][2] == e["cam_t_m2c"][2] * 1000
    # the other fields of the reference's record (/root/reference/src/tools/pegasus_working.py:565-576), against a numpy
    # restatement of its lines 463-466, 547-563
    assert e["bullet_obj_id"] == 7 and "projected_points" not in e
    T_w2c = np.eye(4); T_w2c[:3, :3] = R_c2w.T; T_w2c[:3, 3] = t
    np.testing.assert_allclose(np.array(e["T_w2c"]).reshape(4, 4), T_w2c, atol=1e-15)
    np.testing.assert_allclose(np.array(e["T_m2w"]).reshape(4, 4), T_m2w, atol=1e-15)
    corners = np.array([[x, y, z] for x in (-0.05, 0.05) for y in (-0.03, 0.03) for z in (-0.1, 0.1)])
    center = np.array([0.001, -0.002, 0.003])
    b = bop_pose.scene_gt_entry(R_c2w, t, {7: T_m2w}, K=K, boxes={7: (corners, center)}, dataset_ids={7: 21})[0]
    assert b["obj_id"] == 21 and b["bullet_obj_id"] == 7
    np.testing.assert_array_equal(np.array(b["3d_bounding_box_model_coord"]), corners)
    np.testing.assert_array_equal(np.array(b["3d_bounding_center"]), center)
    T = T_w2c @ T_m2w
    P = K @ T[:3]
    hom = (P @ np.concatenate([corners, np.ones((8, 1))], 1).T).T
    np.testing.assert_allclose(np.array(b["projected_points"]), hom[:, :2] / hom[:, 2:], atol=1e-12)
    hc = P @ np.append(center, 1.0)
    np.testing.assert_allclose(np.array(b["projected_center"]), [hc[:2] / hc[2]], atol=1e-12)
    assert np.array(b["projected_points"]).shape == (8, 2) and np.array(b["projected_center"]).shape == (1, 2)
    with pytest.raises(ValueError):
        bop_pose.scene_gt_entry(R_c2w, t, {7: T_m2w}, boxes={7: (corners, center)})
    # batch form: boxes reach every frame's records, K comes from the view's FoV
    from types import SimpleNamespace
    V = SimpleNamespace(R_c2w=R_c2w, t_w2c=t, fovx=0.8, fovy=0.6, width=640, height=480)
    gt, cam = bop_pose.batch_pose_records([V, V], {7: T_m2w}, boxes={7: (corners, center)})
    assert gt["1"][0]["projected_points"] == b["projected_points"] and len(cam["0"]["cam_K"]) == 9


def test_pose_interpolation_matches_reference_golden():
    from pegasus_amd.pose_interpolation import interpolate_pose
    g = np.load(GOLD / "pose_interpolation.npz")
    for p1, p2, t, out in zip(g["pose1"], g["pose2"], g["t"], g["out"]):
        np.testing.assert_allclose(interpolate_pose(float(t), 0.0, p1, 1.0, p2), out, atol=1e-6)


def test_spatial_order_is_group_preserving_morton_permutation():
    """scene_order.spatial_order: a permutation, environment first / objects in id order, Morton-coherent inside a
    group (neighbours in the order are neighbours in space)."""
    from pegasus_amd.scene_order import morton_codes, spatial_order
    rng = np.random.default_rng(0)
    n = 20000
    xyz = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    oid = np.sort(rng.integers(0, 4, size=n)).astype(np.int32)
    perm = spatial_order(xyz, oid)
    assert sorted(perm.tolist()) == list(range(n))
    assert (np.diff(oid[perm]) >= 0).all()
    codes = morton_codes(xyz)                        # one bounding box for the whole cloud, as spatial_order uses
    for k in range(4):
        sel = perm[oid[perm] == k]
        assert (np.diff(codes[sel].astype(np.int64)) >= 0).all()
    # locality: mean distance between consecutive points drops by an order of magnitude against the input order
    d_in = np.linalg.norm(np.diff(xyz[oid == 1], axis=0), axis=1).mean()
    d_mo = np.linalg.norm(np.diff(xyz[perm][oid[perm] == 1], axis=0), axis=1).mean()
    assert d_mo < 0.2 * d_in
    assert spatial_order(np.zeros((0, 3), np.float32)).shape == (0,)
    assert sorted(spatial_order(xyz[:100]).tolist()) == list(range(100))


def test_pose_table_rows():
    """compose.pose_table: [K,20] rows R|t|center|q(wxyz)|0 from (T, center) pairs; q is R as a unit quaternion."""
    from scipy.spatial.transform import Rotation as Rot
    from pegasus_amd.compose import pose_table
    rng = np.random.default_rng(1)
    pairs = []
    for _ in range(3):
        T = np.eye(4)
        T[:3, :3] = Rot.from_rotvec(rng.normal(0, 1, 3)).as_matrix()
        T[:3, 3] = rng.normal(0, 1, 3)
        pairs.append((T, rng.normal(0, 1, 3)))
    rows = pose_table(pairs)
    assert rows.shape == (3, 20) and rows.dtype == np.float32
    for r, (T, c) in zip(rows, pairs):
        np.testing.assert_allclose(r[0:9].reshape(3, 3), T[:3, :3], atol=1e-6)
        np.testing.assert_allclose(r[9:12], T[:3, 3], atol=1e-6)
        np.testing.assert_allclose(r[12:15], c, atol=1e-6)
        w, x, y, z = r[15:19]
        np.testing.assert_allclose(Rot.from_quat([x, y, z, w]).as_matrix(), T[:3, :3], atol=1e-5)
        assert r[19] == 0.0


def test_png_round_trip():
    """dataset_writer.encode_png: valid PNG (signature, CRCs via decode) for the three image kinds the dataset holds."""
    from pegasus_amd.dataset_writer import decode_png, encode_png
    rng = np.random.default_rng(0)
    for a in (rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8), rng.integers(0, 256, size=(20, 31), dtype=np.uint8),
              rng.integers(0, 65536, size=(16, 45)).astype(np.uint16)):
        png = encode_png(a)
        assert png[:8] == bytes([0x89]) + b"PNG\r\n\x1a\n"
        np.testing.assert_array_equal(decode_png(png), a)
    import pytest
    with pytest.raises(ValueError):
        encode_png(np.zeros((4, 4), np.float32))


def test_scene_directory_round_trip(tmp_path):
    """compat `from scene import Scene`: a model directory (point_cloud/iteration_N/point_cloud.ply + cameras.json, the
    layout upstream's training writes and /root/reference/src/gs/gs_object_rotation.py:79 opens) written through
    Scene.save / camera_to_JSON and read back: same Gaussians, same camera matrices."""
    import json
    import sys
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "compat"))
    try:
        from scene import Scene, GaussianModel
        from scene.cameras import Camera
    finally:
        sys.path.pop(0)
    from pegasus_amd import scenes
    from pegasus_amd.scene import camera_to_JSON
    cloud, _ = scenes.scene_c1(n=500)
    _, views = scenes.scene_c2(n=10, n_views=5, width=320, height=240)
    gm = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling,
                                   cloud.rotation, device="cpu")
    cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, gt_alpha_mask=None,
                   image_name=f"im{i}", uid=i, data_device="cpu", image_width=v.width, image_height=v.height)
            for i, v in enumerate(views)]
    args = type("Args", (), dict(model_path=str(tmp_path), data_device="cpu"))()
    (tmp_path / "cameras.json").write_text(json.dumps([camera_to_JSON(i, c) for i, c in enumerate(cams)]))
    Scene(args, gm, shuffle=False).save(7000)
    Scene(args, gm, shuffle=False).save(30000)
    gm2 = GaussianModel(3, device="cpu")
    sc = Scene(args, gm2, load_iteration=-1, shuffle=False)
    assert sc.loaded_iter == 30000 and len(sc.getTrainCameras()) == 5 and sc.getTestCameras() == []
    for a, b in ((gm._xyz, gm2._xyz), (gm._features_rest, gm2._features_rest), (gm._rotation, gm2._rotation),
                 (gm._opacity, gm2._opacity), (gm._scaling, gm2._scaling), (gm._features_dc, gm2._features_dc)):
        assert torch.equal(a, b)
    for c, d in zip(cams, sc.getTrainCameras()):
        assert (d.image_width, d.image_height, d.image_name) == (c.image_width, c.image_height, c.image_name)
        np.testing.assert_allclose(d.world_view_transform.numpy(), c.world_view_transform.numpy(), atol=2e-6)
        np.testing.assert_allclose(d.full_proj_transform.numpy(), c.full_proj_transform.numpy(), atol=2e-5)
        np.testing.assert_allclose(d.camera_center.numpy(), c.camera_center.numpy(), atol=2e-6)
    with pytest.raises(ValueError):
        Camera(colmap_id=0, R=views[0].R_c2w, T=views[0].t_w2c, FoVx=1.0, FoVy=1.0, image=None, gt_alpha_mask=None,
               image_name="x", uid=0, data_device="cpu")


def test_kept_activations_follow_the_model_on_cpu():
    """gaussian_renderer keeps a model's activated parameters between render() calls (inference only).  The key must miss
    on every edit the version counter sees, survive tensors without a version counter, and never reuse a slot after the
    raw tensor's storage was swapped (advisor findings of round 2)."""
    import numpy as np
    import torch
    from pegasus_amd import gaussian_renderer as GR
    from pegasus_amd.gaussian_model import GaussianModel
    rng = np.random.default_rng(0)
    n = 50
    mk = lambda: GaussianModel.from_arrays(rng.normal(size=(n, 3)), rng.normal(size=(n, 1, 3)), rng.normal(size=(n, 15, 3)),
                                           rng.normal(size=(n, 1)), rng.normal(size=(n, 3)), rng.normal(size=(n, 4)),
                                           device="cpu")
    with torch.no_grad():
        pc = mk()
        a = GR.kept_activation(pc, "get_opacity")
        assert GR.kept_activation(pc, "get_opacity") is a                          # unchanged model: the kept tensor itself
        pc._opacity.add_(1.0)                                            # in-place edit: version counter moves
        b = GR.kept_activation(pc, "get_opacity")
        assert b is not a and torch.equal(b, pc.get_opacity)
        pc._opacity.data.add_(1.0)                                       # bypasses the version counter: documented blind spot
        assert GR.kept_activation(pc, "get_opacity") is b
        GR.invalidate_activations(pc)                                    # ... with this remedy
        assert torch.equal(GR.kept_activation(pc, "get_opacity"), pc.get_opacity)
        # storage swap A -> B -> A': the slot keeps A's storage alive, so A' cannot reappear at A's address under the same key
        seen = GR.kept_activation(pc, "get_scaling")
        first_ptr = pc._scaling.data_ptr()
        for _ in range(4):
            pc._scaling.data = pc._scaling.data.clone() * 0.5
            cur = GR.kept_activation(pc, "get_scaling")
            assert torch.equal(cur, pc.get_scaling)
        assert pc._scaling.data_ptr() != first_ptr or torch.equal(seen, pc.get_scaling)
        f = GR.kept_activation(pc, "get_features")
        pc._features_rest = pc._features_rest.clone()                   # re-assignment of one of the two sources
        assert GR.kept_activation(pc, "get_features") is not f
    with torch.inference_mode():
        pi = mk()                                                        # inference tensors: no version counter
        out = GR.kept_activation(pi, "get_rotation")
        assert torch.equal(out, pi.get_rotation)
    with torch.enable_grad():
        assert GR.kept_activation(pc, "get_opacity") is not GR.kept_activation(pc, "get_opacity")   # autograd on: never kept


def test_merge_gaussians_appends_in_place_and_equals_vstack():
    """GaussianModel.merge_gaussians is the reference's six vstacks (/root/reference/src/gs/gaussian_model.py:584-591) in
    result; a deep-copied scene (what pegasus.py:255-264 composes every frame) takes the merged rows into spare capacity
    instead of re-copying the whole scene per object.  Aliases (copy.copy), masked and re-assigned attributes must fall back."""
    import copy
    import numpy as np
    import torch
    from pegasus_amd.gaussian_model import GaussianModel
    rng = np.random.default_rng(0)
    mk = lambda n: GaussianModel.from_arrays(rng.normal(size=(n, 3)), rng.normal(size=(n, 1, 3)), rng.normal(size=(n, 15, 3)),
                                             rng.normal(size=(n, 1)), rng.normal(size=(n, 3)), rng.normal(size=(n, 4)), device="cpu")
    rows = GaussianModel._ROW_ATTRS
    with torch.no_grad():
        env, a, b = mk(100), mk(30), mk(50)
        env.meta_info = {"name": ["env"]}
        env._features_dc_color = env._features_dc.clone()
        scene = copy.deepcopy(env)
        assert scene.meta_info == env.meta_info and scene.meta_info is not env.meta_info
        assert torch.equal(scene._features_dc_color, env._features_dc_color)
        assert scene._features_dc_color.data_ptr() != env._features_dc_color.data_ptr()
        assert all(torch.equal(getattr(scene, k), getattr(env, k)) and getattr(scene, k).data_ptr() != getattr(env, k).data_ptr()
                   for k in rows)
        want = {k: torch.vstack((getattr(env, k), getattr(a, k), getattr(b, k))) for k in rows}
        p0 = scene._xyz.data_ptr()
        scene.merge_gaussians(a)
        scene.merge_gaussians(b)
        assert all(torch.equal(getattr(scene, k), want[k]) for k in rows)
        assert scene._xyz.data_ptr() == p0                                   # appended, not re-copied
        assert env._xyz.shape[0] == 100 and a._xyz.shape[0] == 30            # sources untouched
        twin = copy.copy(scene)                                              # two models over the same buffers
        scene.merge_gaussians(a)
        twin.merge_gaussians(b)
        assert torch.equal(scene._xyz, torch.vstack((want["_xyz"], a._xyz)))
        assert torch.equal(twin._xyz, torch.vstack((want["_xyz"], b._xyz)))
        keep = torch.ones(scene._xyz.shape[0], dtype=bool)
        keep[:50] = False
        scene.mask_points(keep)                                              # re-assigned attributes: fresh buffer on merge
        before = scene._opacity.clone()
        scene.merge_gaussians(b)
        assert torch.equal(scene._opacity, torch.vstack((before, b._opacity)))
        big = mk(5)                                                          # growth beyond the spare capacity
        for _ in range(6):
            big.merge_gaussians(a)
        assert big._xyz.shape[0] == 5 + 6 * 30 and torch.equal(big._rotation[-30:], a._rotation)
    p = torch.nn.Parameter(torch.zeros(4, 3))                                # autograd on + a parameter: plain vstack
    m = mk(4)
    m._xyz = p
    m.merge_gaussians(a)
    assert m._xyz.requires_grad and m._xyz.shape[0] == 34


def test_network_gui_round_trip_over_loopback():
    """The viewer socket of the reference's GUI branch (/root/reference/pegasus.py:85,249-279): a loopback client sends one
    camera request and reads the answer -- init / try_connect / receive / send as the loop calls them, `conn` as the module
    attribute the loop tests and resets."""
    import json
    import socket
    import struct
    import sys
    from pathlib import Path
    import numpy as np
    import torch
    root = str(Path(__file__).resolve().parents[1])
    for p in (root, root + "/compat"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gaussian_renderer import network_gui
    network_gui.device = "cpu"
    network_gui.conn = None
    network_gui.init("127.0.0.1", 0)                       # port 0: any free port
    network_gui.try_connect()
    assert network_gui.conn is None                        # nobody there yet; the loop just carries on
    client = socket.create_connection(("127.0.0.1", network_gui.port))
    for _ in range(100):
        network_gui.try_connect()
        if network_gui.conn is not None:
            break
    assert network_gui.conn is not None
    rng = np.random.default_rng(0)
    view, proj = rng.normal(size=(4, 4)), rng.normal(size=(4, 4))
    req = dict(resolution_x=64, resolution_y=48, train=0, fov_y=0.7, fov_x=0.9, z_near=0.01, z_far=100.0, shs_python=0,
               rot_scale_python=1, keep_alive=1, scaling_modifier=0.5, view_matrix=view.reshape(-1).tolist(),
               view_projection_matrix=proj.reshape(-1).tolist())
    body = json.dumps(req).encode()
    client.sendall(struct.pack("<i", len(body)) + body)
    cam, train, shs_py, cov_py, keep, mod = network_gui.receive()
    assert (cam.image_width, cam.image_height, train, shs_py, cov_py, keep, mod) == (64, 48, False, False, True, True, 0.5)
    want_view = view.astype(np.float32).copy(); want_view[:, 1] *= -1; want_view[:, 2] *= -1
    want_proj = proj.astype(np.float32).copy(); want_proj[:, 1] *= -1
    np.testing.assert_array_equal(cam.world_view_transform.numpy(), want_view)
    np.testing.assert_array_equal(cam.full_proj_transform.numpy(), want_proj)
    np.testing.assert_allclose(cam.camera_center.numpy(), np.linalg.inv(want_view.astype(np.float64))[3, :3], rtol=1e-4, atol=1e-5)
    image = bytes(range(256)) * 36                         # 64 x 48 x 3 bytes
    network_gui.send(image, "/data/scene")
    got = b""
    while len(got) < len(image) + 4 + 11:
        got += client.recv(65536)
    assert got[:len(image)] == image
    assert struct.unpack("<i", got[len(image):len(image) + 4])[0] == 11 and got[len(image) + 4:] == b"/data/scene"
    # a request without a resolution carries no camera; the answer is the check string alone
    req.update(resolution_x=0, resolution_y=0)
    body = json.dumps(req).encode()
    client.sendall(struct.pack("<i", len(body)) + body)
    assert network_gui.receive()[0] is None
    network_gui.send(None, "x")
    assert client.recv(16) == struct.pack("<i", 1) + b"x"
    client.close()
    network_gui.conn = None                                # what the loop does after a dropped connection


def test_record_views_slice_any_record_tensor_without_copying():
    """masks.record_views on host memory (what rank 0 / a writer holds after the gather): [world, B, bytes] and [B, world, bytes]
    (the transposed global view) slice into the three images as views of the same storage."""
    import torch
    from pegasus_amd import masks as M
    H, W, K = 6, 10, 11
    lay = M.record_layout(H, W, K)
    P, J = H * W, (K + 7) // 8
    world, B = 3, 2
    buf = torch.zeros((world, B, lay["bytes"]), dtype=torch.uint8)
    rng = np.random.default_rng(0)
    rgb = torch.from_numpy(rng.integers(0, 256, (world, B, H, W, 3), dtype=np.uint8))
    depth = torch.from_numpy(rng.integers(-2**15, 2**15, (world, B, H, W), dtype=np.int16))
    bits = torch.from_numpy(rng.integers(0, 256, (world, B, H, W, J), dtype=np.uint8))
    buf[..., :3 * P] = rgb.reshape(world, B, -1)
    buf[..., lay["off_depth"]:lay["off_depth"] + 2 * P] = depth.reshape(world, B, -1).view(torch.uint8)
    buf[..., lay["off_masks"]:lay["off_masks"] + J * P] = bits.reshape(world, B, -1)
    for t, perm in ((buf, lambda x: x), (buf.transpose(0, 1), lambda x: x.transpose(0, 1))):
        v = M.record_views(t, H, W, K)
        assert torch.equal(v["rgb"], perm(rgb)) and torch.equal(v["depth_mm"], perm(depth)) and torch.equal(v["mask_bits"], perm(bits))
        assert v["rgb"].data_ptr() == buf.data_ptr() and v["depth_mm"].dtype == torch.int16
    # masks unpack to [.., K, H, W] planes
    planes = M.unpack_mask_bits(M.record_views(buf[0], H, W, K)["mask_bits"], K)
    assert planes.shape == (B, K, H, W)
    for m in range(K):
        assert torch.equal(planes[:, m], (bits[0, ..., m // 8] >> (m % 8)) & 1)


def test_capacity_and_cache_reset_hooks_are_public():
    """The host-side module state (learned instance capacities, cached asynchronous workspaces, the kept objects-only scene)
    is reachable through public functions only (round-4 verdict: tests reached into rasterizer._WS)."""
    from pegasus_amd import rasterizer as R, render as RW
    before = R.capacity_hints()
    try:
        R.reset_capacity()
        assert R.capacity_hints() == {}
        R.set_capacity_hint(("cpu", 10, 4, 4), 1234)
        R.set_capacity_hint(("cpu", 20, 4, 4), 99)
        assert R.capacity_hints() == {("cpu", 10, 4, 4): 1234, ("cpu", 20, 4, 4): 99}
        R.reset_capacity(500, free_workspaces=False)               # every learned capacity set to 500
        assert set(R.capacity_hints().values()) == {500}
        R.set_capacity_hint(("cpu", 10, 4, 4), None)
        assert list(R.capacity_hints()) == [("cpu", 20, 4, 4)]
        snapshot = R.capacity_hints()
        snapshot.clear()                                            # a copy: the module's table is untouched
        assert len(R.capacity_hints()) == 1
        R.drop_async_workspaces()                                   # nothing cached on this host: a no-op
    finally:
        R.reset_capacity()
        for k, v in before.items():
            R.set_capacity_hint(k, v)
    kept = RW.swap_cache({"key": "k", "scene": object()})
    assert RW.swap_cache(kept)["key"] == "k"
    RW.reset_cache()
    assert RW.swap_cache(None) == {}
    with pytest.raises(RuntimeError, match="per-view limit"):      # the bounded retries end in this error, never in a recursion
        R.grown_capacity(R.MAX_INSTANCES + 1, 1.6)
