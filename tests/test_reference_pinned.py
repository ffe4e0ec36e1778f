"""Everything of the reference that EXECUTES in the build container, as committed outputs (tests/golden/*.npz, written by
tests/golden/make_golden_from_reference.py) against this build:

  gs_model.npz        /root/reference/src/gs/gaussian_model.py  getters :105-128, merge_gaussians :584-591, mask_points
                      :598-623, apply_translation/rotation/transformation_on_xyz + apply_rotation_on_splats :482-505
  render_masks.npz    /root/reference/src/gs/render.py:36-129 (the three mask wrappers over a stand-in render())
  pose_recursion.npz  /root/reference/src/gs/pegasus_setup.py:160-208 (dynamic_object_pose, update_object_pose)
  submodule_import_manifest.json   every name the reference imports from its absent gaussian-splatting submodule

Host rows under ``-m "not gpu"`` (incl. the oracle's colour-mask / cast / pose restatements, which the GPU parity tests lean
on); the same goldens against the HIP kernels (pgr_color_masks, pgr_pack_frames, pgr_pack_records, pgr_compose_object)
under ``-m gpu``.  The rasterizer itself (rows a5-a10) has no golden: its source is absent from the reference."""
import copy
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
RAW = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _model(g, prefix, device="cpu"):
    from pegasus_amd.gaussian_model import GaussianModel
    m = GaussianModel(3, device=device)
    for k in RAW:
        setattr(m, k, torch.from_numpy(g[prefix + k]).to(device))
    m.active_sh_degree = 3
    return m


def _same_rotation(q_mine, q_ref, atol):
    """Unit quaternions (w, x, y, z) of the same rotations: equal up to the sign of each row (q and -q are one rotation;
    the reference's sign is whatever scipy's from_matrix picks)."""
    q_mine, q_ref = np.asarray(q_mine, np.float64), np.asarray(q_ref, np.float64)
    s = np.sign(np.sum(q_mine * q_ref, axis=1, keepdims=True))
    np.testing.assert_allclose(q_mine * s, q_ref, atol=atol)


# ---------------------------------------------------------------------------------------------------- manifest

def test_every_name_the_reference_imports_from_its_absent_submodule_resolves():
    manifest = json.loads((GOLD / "submodule_import_manifest.json").read_text())
    assert "train" in manifest and "diff_gaussian_rasterization" not in manifest   # (only gaussian_renderer imports that one)
    lines = ["import importlib"]
    for module, names in manifest.items():
        lines.append(f"m = importlib.import_module({module!r})")
        for n in names:
            if n:
                lines.append(f"assert hasattr(m, {n!r}), ({module!r}, {n!r})")
    lines += ["import diff_gaussian_rasterization as d",
              "assert d.GaussianRasterizationSettings._fields[0] == 'image_height' and len(d.GaussianRasterizationSettings._fields) == 12",
              "from train import training",
              "try:\n    training(None, None, None, [], [], [], None, -1)\nexcept NotImplementedError as e:\n    assert 'out of' in str(e) or 'not part' in str(e)\nelse:\n    raise SystemExit('train.training did not raise')",
              "print('ok')"]
    env = dict(os.environ, PYTHONPATH=f"{ROOT}{os.pathsep}{ROOT / 'compat'}")
    r = subprocess.run([sys.executable, "-c", "\n".join(lines)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------- GaussianModel (host)

def test_gaussian_model_getters_merge_mask_and_xyz_poses_equal_the_reference_class():
    from pegasus_amd.gaussian_model import build_rotation
    g = np.load(GOLD / "gs_model.npz")
    a, b = _model(g, "a"), _model(g, "b")
    for name in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features"):
        np.testing.assert_array_equal(getattr(a, name).numpy(), g["a." + name], err_msg=name)     # same torch ops: bit-equal
    assert tuple(a.get_features.shape) == (257, 16, 3) and tuple(a.get_opacity.shape) == (257, 1)
    np.testing.assert_array_equal(a.get_covariance().numpy(), g["a.get_covariance_1"])
    np.testing.assert_array_equal(a.get_covariance(1.7).numpy(), g["a.get_covariance_1.7"])
    # merge: once into a deep copy (the frame loop's order, pegasus.py:255-264: in-place append), once into a plain model
    for m in (copy.deepcopy(a), _model(g, "a")):
        m.merge_gaussians(b)
        for k in RAW:
            np.testing.assert_array_equal(getattr(m, k).numpy(), g["merged" + k], err_msg=k)
    m.max_radii2D = torch.arange(257 + 64, dtype=torch.float32)
    m.mask_points(torch.from_numpy(g["mask"]))
    for k in RAW:
        np.testing.assert_array_equal(getattr(m, k).numpy(), g["masked" + k], err_msg=k)
    np.testing.assert_array_equal(m.max_radii2D.numpy(), g["masked.max_radii2D"])
    R, t, T = (torch.from_numpy(g[k]) for k in ("R", "t", "T"))
    m = _model(g, "a"); m.apply_translation_on_xyz(t); np.testing.assert_array_equal(m._xyz.numpy(), g["translated._xyz"])
    m = _model(g, "a"); m.apply_rotation_on_xyz(R); np.testing.assert_array_equal(m._xyz.numpy(), g["rotated_about_mean._xyz"])
    m = _model(g, "a"); m.apply_rotation_on_xyz(R, origin=True)
    np.testing.assert_array_equal(m._xyz.numpy(), g["rotated_about_origin._xyz"])
    m = _model(g, "a"); m.apply_transformation_on_xyz(T); np.testing.assert_array_equal(m._xyz.numpy(), g["transformed._xyz"])
    # the splat quaternions the reference gets through scipy: R @ build_rotation(q), as a quaternion, sign free
    from scipy.spatial.transform import Rotation as Rot
    Rq = (R @ build_rotation(torch.from_numpy(g["a_rotation"]))).numpy()
    q = np.roll(Rot.from_matrix(Rq).as_quat(), 1, axis=-1)
    _same_rotation(q, g["splats_rotated._rotation"], 1e-6)


def test_oracle_pose_restatement_equals_the_reference_class():
    """oracle/compose_ref.py (what tests/test_compose.py holds the HIP composition kernel to) against the reference's own
    xyz and quaternions."""
    from scipy.spatial.transform import Rotation as Rot
    from oracle.compose_ref import compose_object_ref
    g = np.load(GOLD / "gs_model.npz")
    xyz, Rm = compose_object_ref(g["a_xyz"], g["a_rotation"], g["T"])
    np.testing.assert_allclose(xyz, g["transformed._xyz"], atol=2e-6)
    _same_rotation(np.roll(Rot.from_matrix(Rm).as_quat(), 1, axis=-1), g["splats_rotated._rotation"], 1e-6)
    T2 = np.eye(4); T2[:3, :3] = g["R2"]
    _, Rm2 = compose_object_ref(g["a_xyz"], g["splats_rotated._rotation"], T2)
    _same_rotation(np.roll(Rot.from_matrix(Rm2).as_quat(), 1, axis=-1), g["splats_rotated_twice._rotation"], 1e-6)


# ---------------------------------------------------------------------------------------------------- mask wrappers (host)

def _objects(g, device="cpu"):
    from pegasus_amd.render import assign_semantic_colors
    objects = {int(i): _model(g, f"obj{int(i)}", device) for i in g["object_ids"]}
    assign_semantic_colors(objects, g["colors"])
    return objects


def test_oracle_colour_masks_and_casts_equal_the_reference_wrappers(oracle):
    """The oracle's a11 / a12 restatements (pgr_oracle_color_masks, pgr_oracle_quantize) against what
    /root/reference/src/gs/render.py returned for the same images: threshold-sphere pixels, values beyond 1.0."""
    g = np.load(GOLD / "render_masks.npz")
    H, W, K = (int(v) for v in g["shape"])
    m = oracle.color_masks(g["visib.image"], g["colors"], 0.1)                       # [K,H,W] uint8
    assert g["visib.masks"].dtype == np.float64 and g["visib.masks"].shape == (H, W, K)
    np.testing.assert_array_equal(m.transpose(1, 2, 0).astype(np.float64), g["visib.masks"])
    np.testing.assert_array_equal(g["visib.seg_image"], g["visib.image"].transpose(1, 2, 0))
    rgb, _ = oracle.quantize(g["semseg.image"], np.zeros((H, W), np.float32))
    assert g["semseg.out"].dtype == np.uint8 and (g["semseg.image"] > 1.0).any()
    np.testing.assert_array_equal(rgb, g["semseg.out"])
    # silhouettes: object i's own image against ITS colour only; ids missing from the dictionary stay empty
    ids = [int(i) for i in g["object_ids"]]
    want = np.zeros((H, W, K))
    for j, oid in enumerate(ids):
        want[..., oid - 1] = oracle.color_masks(g["silhouette.images"][j], g["colors"][oid - 1:oid], 0.1)[0]
    np.testing.assert_array_equal(want, g["silhouette.masks"])
    assert not g["silhouette.masks"][..., [2, 4]].any() and g["visib.masks"][..., [2, 4]].any()


def test_semantic_scene_composition_equals_what_the_reference_hands_to_render():
    """The scene of the two semantic renders (render.py:68-84,100-113: deepcopy, paint, merge, mask the environment out) and
    of every silhouette render (:36-57): this build composes them without the environment copy; the rows must be the
    reference's, in its order, painted with the semantic SH colour."""
    from pegasus_amd import render as RW
    g = np.load(GOLD / "render_masks.npz")
    env, objects = _model(g, "env"), _objects(g)
    RW.reset_cache()
    scene, _ = RW._semantic_scene(env, objects)
    for k in RAW:
        np.testing.assert_array_equal(getattr(scene, k).numpy(), g["visib.scene" + k], err_msg=k)
        np.testing.assert_array_equal(getattr(scene, k).numpy(), g["semseg.scene" + k], err_msg=k)
    assert scene.active_sh_degree == env.active_sh_degree and int(env._xyz.shape[0]) == 50      # the environment is untouched
    black = RW._without_environment(copy.deepcopy(env), env._xyz.shape[0])
    for j, oid in enumerate(int(i) for i in g["object_ids"]):
        s = copy.deepcopy(black)
        s.merge_gaussians(gaussian=objects[oid])
        for k in RAW:
            np.testing.assert_array_equal(getattr(s, k).numpy(), g[f"silhouette.scene{j}" + k], err_msg=k)
    RW.reset_cache()


# ---------------------------------------------------------------------------------------------------- pose recursion (host)

def test_absolute_poses_equal_the_reference_recursion_on_its_own_class():
    """pegasus_amd/trajectory.py composes step s absolutely; the reference accumulates fp32 deltas on its GaussianModel
    (pegasus_setup.py:160-208).  Golden = the reference's xyz / quaternions after steps 0, 1, 2, 17, 60, 120, 199 of the
    committed trajectory.  fp32 drift of 199 accumulated updates bounds the tolerance (measured 3e-6)."""
    from scipy.spatial.transform import Rotation as Rot
    from pegasus_amd import trajectory as TJ
    from oracle.compose_ref import compose_object_ref
    g = np.load(GOLD / "pose_recursion.npz")
    traj = TJ.load_fixture()
    xyz0, rot0 = g["canonical_xyz"], g["canonical_rotation"]
    np.testing.assert_allclose(g["R_init"], Rot.from_quat(traj[0, 3:]).as_matrix(), atol=1e-6)
    np.testing.assert_allclose(g["t_init"], traj[0, :3], atol=1e-6)
    for i, s in enumerate(int(v) for v in g["steps"]):
        T = TJ.absolute_pose(traj, s)
        xyz, Rm = compose_object_ref(xyz0, rot0, T)
        np.testing.assert_allclose(xyz, g["xyz"][i], atol=2e-5, err_msg=f"step {s}")
        _same_rotation(np.roll(Rot.from_matrix(Rm).as_quat(), 1, axis=-1), g["rotation"][i], 2e-5)
        # and the float64 restatement of the recursion used by tests/test_trajectory.py
        ref = TJ.accumulate_deltas(traj, s, xyz0.astype(np.float64).mean(0), xyz0)
        np.testing.assert_allclose(ref, g["xyz"][i], atol=2e-5, err_msg=f"step {s}")


# ---------------------------------------------------------------------------------------------------- on the GPU

@pytest.mark.gpu
def test_gpu_gaussian_model_equals_the_reference_class(gpu_device):
    g = np.load(GOLD / "gs_model.npz")
    a, b = _model(g, "a", gpu_device), _model(g, "b", gpu_device)
    for name in ("get_xyz", "get_features"):
        np.testing.assert_array_equal(getattr(a, name).cpu().numpy(), g["a." + name])
    for name in ("get_scaling", "get_rotation", "get_opacity"):                      # device exp / sigmoid / rsqrt: 1 ulp
        np.testing.assert_allclose(getattr(a, name).cpu().numpy(), g["a." + name], rtol=3e-7, atol=1e-9, err_msg=name)
    # (off-diagonal terms are differences of products: absolute error of the largest entry)
    np.testing.assert_allclose(a.get_covariance(1.7).cpu().numpy(), g["a.get_covariance_1.7"], rtol=2e-6,
                               atol=2e-6 * float(np.abs(g["a.get_covariance_1.7"]).max()))
    m = copy.deepcopy(a)
    m.merge_gaussians(b)
    for k in RAW:
        np.testing.assert_array_equal(getattr(m, k).cpu().numpy(), g["merged" + k], err_msg=k)
    m.mask_points(torch.from_numpy(g["mask"]).to(gpu_device))
    for k in RAW:
        np.testing.assert_array_equal(getattr(m, k).cpu().numpy(), g["masked" + k], err_msg=k)
    R, t, T = (torch.from_numpy(g[k]).to(gpu_device) for k in ("R", "t", "T"))
    m = _model(g, "a", gpu_device); m.apply_translation_on_xyz(t)
    np.testing.assert_array_equal(m._xyz.cpu().numpy(), g["translated._xyz"])
    m = _model(g, "a", gpu_device); m.apply_rotation_on_xyz(R)
    np.testing.assert_allclose(m._xyz.cpu().numpy(), g["rotated_about_mean._xyz"], atol=1e-6)
    m = _model(g, "a", gpu_device); m.apply_rotation_on_xyz(R, origin=True)
    np.testing.assert_allclose(m._xyz.cpu().numpy(), g["rotated_about_origin._xyz"], atol=1e-6)
    m = _model(g, "a", gpu_device); m.apply_transformation_on_xyz(T)
    np.testing.assert_allclose(m._xyz.cpu().numpy(), g["transformed._xyz"], atol=1e-6)
    # pgr_compose_object: orientation only, twice in a row with two matrices (R as tensor, then as ndarray)
    m = _model(g, "a", gpu_device); m.apply_rotation_on_splats(R)
    _same_rotation(m._rotation.cpu().numpy(), g["splats_rotated._rotation"], 1e-6)
    m.apply_rotation_on_splats(np.asarray(g["R2"], np.float64))
    _same_rotation(m._rotation.cpu().numpy(), g["splats_rotated_twice._rotation"], 2e-6)
    # one fused launch for xyz + orientation (+ SH)
    m = _model(g, "a", gpu_device); m.apply_transformation(T)
    np.testing.assert_allclose(m._xyz.cpu().numpy(), g["transformed._xyz"], atol=1e-6)
    _same_rotation(m._rotation.cpu().numpy(), g["splats_rotated._rotation"], 1e-6)


@pytest.mark.gpu
def test_gpu_mask_wrappers_equal_the_reference_wrappers(gpu_device, monkeypatch):
    """pegasus_amd/render.py's wrappers with ``render`` replaced by the golden's images: masks by pgr_color_masks, the uint8
    cast by pgr_pack_frames, and the frame record of pgr_pack_records -- bit for bit what /root/reference/src/gs/render.py
    returned (float64 [H,W,K] masks, float32 HWC image, uint8 HWC semantic image)."""
    from pegasus_amd import masks as M
    from pegasus_amd import render as RW
    g = np.load(GOLD / "render_masks.npz")
    H, W, K = (int(v) for v in g["shape"])
    env, objects = _model(g, "env", gpu_device), _objects(g, gpu_device)
    color_set = torch.from_numpy(g["colors"]).to(gpu_device)
    queue, seen = [], []

    def fake_render(cam, scene, pipe, bg):
        seen.append(int(scene._xyz.shape[0]))
        return {"render": torch.from_numpy(queue.pop(0).copy()).to(gpu_device)}

    monkeypatch.setattr(RW, "render", fake_render)
    RW.reset_cache()
    cam = type("Cam", (), dict(world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4), camera_center=torch.zeros(3),
                               image_height=H, image_width=W, FoVx=0.8, FoVy=0.6))()
    bg = torch.zeros(3, device=gpu_device)
    with torch.no_grad():
        queue.append(g["visib.image"])
        masks, seg_image = RW.render_visib_mask(cam, env, objects, color_set, H, W, None, bg)
        assert masks.dtype == np.float64 and masks.shape == (H, W, K)
        np.testing.assert_array_equal(masks, g["visib.masks"])
        np.testing.assert_array_equal(np.asarray(seg_image), g["visib.seg_image"])
        RW.reset_cache()
        queue.append(g["semseg.image"])
        sem = RW.render_semanticsegmentation_mask(cam, env, objects, color_set, H, W, None, bg, False)
        assert sem.dtype == np.uint8
        np.testing.assert_array_equal(sem, g["semseg.out"])
        RW.reset_cache()
        queue.extend(list(g["silhouette.images"]))
        sil = RW._render_silhouette_mask_per_object(cam, objects, env, W, H, color_set, None, bg)
        assert sil.dtype == np.float64
        np.testing.assert_array_equal(sil, g["silhouette.masks"])
    n_obj = [int(o._xyz.shape[0]) for o in objects.values()]
    assert seen == [sum(n_obj), sum(n_obj)] + n_obj and not queue
    RW.reset_cache()
    # the frame record (what leaves the GPU at N > 1): RGB bytes and mask bits of the same images
    img = torch.from_numpy(g["visib.image"]).to(gpu_device)
    m = M.color_masks(img, color_set, 0.1)
    packed = M.pack_frames(color=img[None], depth=torch.zeros((1, 1, H, W), device=gpu_device), masks=m[None])
    bits = packed["mask_bits"][0].cpu().numpy().reshape(H, W, -1)
    want_bits = np.packbits(g["visib.masks"].astype(np.uint8), axis=-1, bitorder="little")
    np.testing.assert_array_equal(bits, want_bits)
    np.testing.assert_array_equal(packed["rgb"][0].cpu().numpy(),
                                  (np.ascontiguousarray(g["visib.image"].transpose(1, 2, 0)) * 255).astype("uint8"))


@pytest.mark.gpu
def test_gpu_pose_recursion_equals_the_reference_recursion(gpu_device):
    """The reference's own update loop (pegasus_setup.py:160-208: per step apply_transformation_on_xyz,
    apply_rotation_on_splats, apply_rotation_on_sh with the delta pose) driven through THIS build's GaussianModel on the
    device, and the absolute pose through pgr_compose_object, against the reference's xyz / quaternions."""
    from scipy.spatial.transform import Rotation as Rot
    from pegasus_amd import trajectory as TJ
    g = np.load(GOLD / "pose_recursion.npz")
    traj = TJ.load_fixture()
    steps = [int(v) for v in g["steps"]]
    obj = _model(g, "canonical", gpu_device)

    def apply(R, t):                                                    # apply_transformation_on_gs, literally
        T = torch.eye(4, dtype=torch.float32, device=gpu_device)
        T[:3, :3] = R
        T[:3, 3] = t
        obj.apply_transformation_on_xyz(T=T)
        obj.apply_rotation_on_splats(R=R)
        obj.apply_rotation_on_sh(R=R)

    f32 = lambda a: torch.from_numpy(np.asarray(a)).type(torch.float32).to(gpu_device)
    apply(f32(Rot.from_quat(traj[0, 3:]).as_matrix()), f32(traj[0, :3]))
    for s in range(0, max(steps) + 1):
        if s > 0:
            q_delta = Rot.from_quat(traj[s, 3:]) * Rot.from_quat(traj[s - 1, 3:]).inv()
            apply(f32(q_delta.as_matrix()), f32(traj[s, :3] - traj[s - 1, :3]))
        if s in steps:
            i = steps.index(s)
            np.testing.assert_allclose(obj._xyz.cpu().numpy(), g["xyz"][i], atol=3e-5, err_msg=f"step {s}")
            _same_rotation(obj._rotation.cpu().numpy() / np.linalg.norm(obj._rotation.cpu().numpy(), axis=1, keepdims=True),
                           g["rotation"][i], 3e-5)
    for i, s in enumerate(steps):                                       # absolute composition: one launch per step
        m = _model(g, "canonical", gpu_device)
        m.apply_transformation(TJ.absolute_pose(traj, s))
        np.testing.assert_allclose(m._xyz.cpu().numpy(), g["xyz"][i], atol=2e-5, err_msg=f"step {s}")
        _same_rotation(m._rotation.cpu().numpy(), g["rotation"][i], 2e-5)
