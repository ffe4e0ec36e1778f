"""Generates tests/golden/*.npz from the parts of /root/reference that import in THIS container.

Run here only (``python tests/golden/make_golden_from_reference.py``); /root/reference does not
exist on the GPU box and nothing at test/bench time reads it.  Only inputs and the reference's
OUTPUTS are stored -- no reference source text.

What imports: bop_toolkit view_sampler (with imageio/png stubbed),
src.utility.pose_interpolation, src.utility.graphic_utils (``.to('cuda')`` patched to identity
for generate_colors), bop_toolkit_lib.misc (calc_2d_bbox: the bounding boxes of scene_gt_info.json).
The rasterizer itself is absent, so no rasterizer golden can be produced.

Round 6 adds the reference modules that import once their ABSENT third-party dependencies are replaced by empty
stand-in modules (plyfile, open3d, e3nn, sphecerix, cv2, imageio, colmap_wrapper, src.dataset.*: none of them is on the
code paths executed here) and the names of the absent gaussian-splatting submodule are resolved through compat/:
  gs_model        src/gs/gaussian_model.py  GaussianModel getters, merge_gaussians, mask_points, pose methods
  render_masks    src/gs/render.py:36-129   the three mask wrappers over a deterministic stand-in for render()
  pose_recursion  src/gs/pegasus_setup.py:160-208  dynamic_object_pose + update_object_pose over the committed fixture
  manifest        every name the reference imports from the absent submodule
``.to('cuda')`` is redirected to the CPU while the reference code runs (there is no GPU in the build container).

``python tests/golden/make_golden_from_reference.py [section ...]`` regenerates only the named sections
(views, gt_info, gs_model, render_masks, pose_recursion, manifest); default: all.
"""
import json
import math
import os
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def main():
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    assert REF.exists(), "reference checkout not present"
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "submodules" / "bop_toolkit"))
    os.environ.setdefault("PEGASUS_PATH", "/tmp")
    os.environ.setdefault("BOP_PATH", "/tmp")
    for name in ("imageio", "png", "cv2"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)

    if want("gt_info"):
        gt_info_golden()
    if want("manifest"):
        manifest_golden()
    if want("gs_model"):
        gs_model_golden()
    if want("render_masks"):
        render_masks_golden()
    if want("pose_recursion"):
        pose_recursion_golden()
    if only and only <= {"gt_info", "manifest", "gs_model", "render_masks", "pose_recursion"}:
        return
    # 1. BOP fibonacci hemisphere views (config-2/3 camera sets)
    from bop_toolkit_lib import view_sampler
    cases = {}
    for (n, radius) in ((128, 0.45), (1024, 1.0), (33, 0.8)):
        views, _ = view_sampler.sample_views(n, radius=radius, elev_range=(0, 0.5 * math.pi), mode="fibonacci")
        cases[f"R_{n}"] = np.stack([v["R"] for v in views])
        cases[f"t_{n}"] = np.stack([v["t"].reshape(3) for v in views])
        cases[f"radius_{n}"] = np.float64(radius)
    np.savez_compressed(OUT / "bop_fibonacci_views.npz", **cases)

    # 2. pose interpolation (camera trajectory maths, pegasus_setup.py:85-143)
    from src.utility import pose_interpolation as pi
    rng = np.random.default_rng(11)
    from scipy.spatial.transform import Rotation as Rot
    p1, p2, ts, outs = [], [], [], []
    for _ in range(16):
        a, b = np.eye(4), np.eye(4)
        a[:3, :3] = Rot.random(random_state=int(rng.integers(1 << 30))).as_matrix()
        b[:3, :3] = Rot.random(random_state=int(rng.integers(1 << 30))).as_matrix()
        a[:3, 3] = rng.normal(size=3)
        b[:3, 3] = rng.normal(size=3)
        t = float(rng.uniform(0, 1))
        p1.append(a); p2.append(b); ts.append(t)
        outs.append(pi.interpolate_pose(t, 0.0, a, 1.0, b))
    np.savez_compressed(OUT / "pose_interpolation.npz", pose1=np.stack(p1), pose2=np.stack(p2),
                        t=np.asarray(ts), out=np.stack(outs))

    # 3. graphic_utils: quaternion <-> matrix, semantic colour table
    import torch
    from src.utility import graphic_utils as gu
    q = rng.normal(size=(32, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Rm = np.stack([gu.qvec2rotmat(v) for v in q])
    qb = np.stack([gu.rotmat2qvec(m) for m in Rm])
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self
    try:
        colors = {f"colors_{n}": gu.generate_colors(n).numpy() for n in (1, 3, 6, 8, 21)}
    finally:
        torch.Tensor.to = orig_to
    np.savez_compressed(OUT / "graphic_utils.npz", q=q, R=Rm, q_back=qb, **colors)

    # 4. first 200 steps of body 1 of the reference's pose-sequence fixture (dynamic config)
    sim = json.loads((REF / "src" / "engine" / "simulation_steps.json").read_text())
    meta = {"keys": sorted(sim.keys())}
    traj = sim["trajectory"] if "trajectory" in sim else sim
    bodies = sorted(traj.keys(), key=lambda s: int(s))
    meta["bodies"] = bodies
    body = traj[bodies[-1]]
    steps = sorted(body.keys(), key=lambda s: int(s))[:200]
    tq = np.asarray([[*body[s]["t"], *body[s]["q"]] for s in steps], dtype=np.float64)
    np.savez_compressed(OUT / "simulation_steps_body1_first200.npz", t_q_xyzw=tq)
    (OUT / "simulation_steps_meta.json").write_text(json.dumps(meta))
    print("golden fixtures written to", OUT)


def gt_info_golden():
    """scene_gt_info.json entries (submodules/bop_toolkit/docs/bop_datasets_format.md:116-129) as
    submodules/bop_toolkit/scripts/calc_gt_info.py:139-172 derives them from an object's silhouette mask, its visible mask and
    the depth image -- the bounding boxes through the reference's own bop_toolkit_lib.misc.calc_2d_bbox."""
    from bop_toolkit_lib import misc
    rng = np.random.default_rng(23)
    H, W, n = 37, 53, 24
    sil = np.zeros((n, H, W), bool)
    vis = np.zeros((n, H, W), bool)
    depth_valid = rng.random((n, H, W)) > 0.1
    for i in range(n):
        if i % 6 == 5:
            continue                                   # an object that is not in the image at all
        y0, x0 = int(rng.integers(0, H - 4)), int(rng.integers(0, W - 4))
        y1, x1 = int(rng.integers(y0 + 1, H + 1)), int(rng.integers(x0 + 1, W + 1))
        blob = rng.random((y1 - y0, x1 - x0)) > 0.35
        sil[i, y0:y1, x0:x1] = blob
        if i % 6 != 4:                                 # (i % 6 == 4: fully occluded -> no visible pixel)
            vis[i] = sil[i] & (rng.random((H, W)) > 0.4)
    out = dict(px_count_all=[], px_count_valid=[], px_count_visib=[], visib_fract=[], bbox_obj=[], bbox_visib=[])
    for i in range(n):
        px_all, px_valid, px_visib = int(sil[i].sum()), int((depth_valid[i] & sil[i]).sum()), int(vis[i].sum())
        bbox = bbox_visib = [-1, -1, -1, -1]
        if px_visib > 0:
            ys, xs = sil[i].nonzero()
            bbox = misc.calc_2d_bbox(xs, ys, (W, H))
            ys, xs = vis[i].nonzero()
            bbox_visib = misc.calc_2d_bbox(xs, ys, (W, H))
        out["px_count_all"].append(px_all); out["px_count_valid"].append(px_valid); out["px_count_visib"].append(px_visib)
        out["visib_fract"].append(px_visib / float(px_all) if px_all > 0 else 0.0)
        out["bbox_obj"].append([int(e) for e in bbox]); out["bbox_visib"].append([int(e) for e in bbox_visib])
    # projection of model points into the image (scene_gt's projected_points / projected_center): the toolkit's project_pts
    from scipy.spatial.transform import Rotation as Rot
    proj = dict(K=[], R=[], t=[], pts=[], uv=[])
    for i in range(12):
        K = np.array([[rng.uniform(400, 1200), 0, rng.uniform(200, 600)], [0, rng.uniform(400, 1200), rng.uniform(200, 600)], [0, 0, 1.0]])
        R = Rot.random(random_state=int(rng.integers(1 << 30))).as_matrix()
        t = np.array([[rng.normal(0, 0.2)], [rng.normal(0, 0.2)], [rng.uniform(0.5, 3.0)]])
        pts = rng.normal(0, 0.1, (9, 3))
        proj["K"].append(K); proj["R"].append(R); proj["t"].append(t[:, 0]); proj["pts"].append(pts)
        proj["uv"].append(misc.project_pts(pts, K, R, t))
    np.savez_compressed(OUT / "bop_gt_info.npz", sil=np.packbits(sil, axis=-1), vis=np.packbits(vis, axis=-1),
                        depth_valid=np.packbits(depth_valid, axis=-1), shape=np.asarray([n, H, W]),
                        **{k: np.asarray(v) for k, v in out.items()}, **{"proj_" + k: np.asarray(v) for k, v in proj.items()})
    print("bop_gt_info.npz written")


# ---------------------------------------------------------------------------------------------------------------
# round 6: the reference's own GaussianModel, mask wrappers and pose recursion, executed here

ABSENT_THIRD_PARTY = ("plyfile", "open3d", "e3nn", "sphecerix", "cv2", "imageio", "colmap_wrapper", "src.dataset",
                      "src.dataset.dataset_envs", "src.dataset.ycb_objects", "src.dataset.cup_noodle_dataset")
SUBMODULE_TOPLEVEL = ("arguments", "scene", "gaussian_renderer", "utils", "train", "simple_knn", "diff_gaussian_rasterization")


class _Anything(types.ModuleType):
    """Stand-in for an absent third-party module: every attribute is another stand-in (never called on these paths)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        raise RuntimeError(f"stand-in for the absent module {self.__name__} was called")


def _reference_env():
    """sys.modules / sys.path such that the reference's src.gs modules import: stand-ins for absent third-party modules,
    compat/ for the names of the absent gaussian-splatting submodule."""
    repo = OUT.parents[1]
    for pth in (str(repo), str(repo / "compat")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    for name in ABSENT_THIRD_PARTY:
        top = name.split(".")[0]
        if top == "src":
            sys.modules.setdefault(name, _Anything(name))
            continue
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = _Anything(name)
            for sub in ("o3", "dataloader"):
                sys.modules.setdefault(name + "." + sub, _Anything(name + "." + sub))


class _cuda_is_cpu:
    """``x.to('cuda')`` / ``device='cuda'`` land on the CPU while the reference code runs."""

    def __enter__(self):
        import torch
        self.torch = torch
        self.saved = {k: getattr(torch, k) for k in ("asarray", "eye", "ones", "zeros")}
        self.to = torch.Tensor.to

        def fix(v):
            return "cpu" if (isinstance(v, str) and v.startswith("cuda")) or \
                (isinstance(v, torch.device) and v.type == "cuda") else v

        def to(t, *a, **k):
            return self.to(t, *[fix(v) for v in a], **{kk: fix(v) for kk, v in k.items()})

        def wrap(f):
            return lambda *a, **k: f(*a, **{kk: fix(v) for kk, v in k.items()})

        torch.Tensor.to = to
        for k, f in self.saved.items():
            setattr(torch, k, wrap(f))
        return self

    def __exit__(self, *exc):
        self.torch.Tensor.to = self.to
        for k, f in self.saved.items():
            setattr(self.torch, k, f)


def manifest_golden():
    """Every name the reference imports from the ABSENT gaussian-splatting submodule (SURVEY.md section 8b "Companion
    names"), read from the import statements of every python file of the reference outside submodules/ -- names and the
    places that import them only, by ast."""
    import ast
    entries = {}
    files = sorted(str(f.relative_to(REF)) for f in REF.rglob("*.py") if "submodules" not in f.relative_to(REF).parts)
    for rel in files:
        try:
            tree = ast.parse((REF / rel).read_text())
        except SyntaxError:
            continue
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.level == 0 and \
                    node.module.split(".")[0] in SUBMODULE_TOPLEVEL:
                for a in node.names:
                    entries.setdefault(node.module, {}).setdefault(a.name, []).append(f"{rel}:{node.lineno}")
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name.split(".")[0] in SUBMODULE_TOPLEVEL:
                        entries.setdefault(a.name, {}).setdefault("", []).append(f"{rel}:{node.lineno}")
    out = {m: {n: sorted(set(w)) for n, w in sorted(names.items())} for m, names in sorted(entries.items())}
    (OUT / "submodule_import_manifest.json").write_text(json.dumps(out, indent=1) + "\n")
    print("submodule_import_manifest.json:", sum(len(v) for v in out.values()), "names in", len(out), "modules")


def _raw_model(rng, n, cls, sh_degree=3):
    import torch
    m = cls(sh_degree)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    m._xyz = t(rng.normal(0, 0.3, (n, 3)) + np.array([0.4, -0.1, 0.2]))
    m._features_dc = t(rng.uniform(-1.5, 1.5, (n, 1, 3)))
    m._features_rest = t(rng.normal(0, 0.1, (n, 15, 3)))
    m._opacity = t(rng.normal(2, 1.5, (n, 1)))
    m._scaling = t(rng.normal(np.log(0.02), 0.3, (n, 3)))
    m._rotation = t(rng.normal(0, 1, (n, 4)) * rng.uniform(0.2, 3.0, (n, 1)))       # NOT normalised: the getter does that
    m.active_sh_degree = sh_degree
    return m


_RAW = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _raw(m, prefix):
    return {prefix + k: getattr(m, k).detach().cpu().numpy() for k in _RAW}


def gs_model_golden():
    """/root/reference/src/gs/gaussian_model.py executed on seeded CPU tensors: the six getters (:105-128), merge_gaussians
    (:584-591), mask_points (:598-623), apply_translation_on_xyz / apply_rotation_on_xyz (both origins) /
    apply_transformation_on_xyz / apply_rotation_on_splats (:482-505).  get_covariance and apply_rotation_on_splats call
    build_scaling_rotation / strip_symmetric / build_rotation of the ABSENT utils.general_utils: compat/'s versions stand
    in (recorded in the file as `compat_names`).  apply_rotation_on_sh needs e3nn (absent): not pinned."""
    import copy
    import torch
    from scipy.spatial.transform import Rotation as Rot
    _reference_env()
    with _cuda_is_cpu():
        from src.gs.gaussian_model import GaussianModel as RefModel
        rng = np.random.default_rng(61)
        out = {}
        a = _raw_model(rng, 257, RefModel)
        b = _raw_model(rng, 64, RefModel)
        out.update(_raw(a, "a")); out.update(_raw(b, "b"))
        for g in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features"):
            out["a." + g] = getattr(a, g).numpy()
        out["a.get_covariance_1"] = a.get_covariance().numpy()
        out["a.get_covariance_1.7"] = a.get_covariance(1.7).numpy()
        m = copy.deepcopy(a)
        m.merge_gaussians(b)
        out.update(_raw(m, "merged"))
        mask = torch.from_numpy(rng.random(257 + 64) > 0.4)
        m.max_radii2D = torch.arange(257 + 64, dtype=torch.float32)
        m.mask_points(mask)
        out["mask"] = mask.numpy()
        out.update(_raw(m, "masked"))
        out["masked.max_radii2D"] = m.max_radii2D.numpy()
        R = torch.from_numpy(Rot.random(random_state=7).as_matrix()).float()
        t = torch.tensor([0.25, -0.5, 0.125])
        T = torch.eye(4); T[:3, :3] = R; T[:3, 3] = t
        out["R"], out["t"], out["T"] = R.numpy(), t.numpy(), T.numpy()
        m = copy.deepcopy(a); m.apply_translation_on_xyz(t); out["translated._xyz"] = m._xyz.numpy()
        m = copy.deepcopy(a); m.apply_rotation_on_xyz(R); out["rotated_about_mean._xyz"] = m._xyz.numpy()
        m = copy.deepcopy(a); m.apply_rotation_on_xyz(R, origin=True); out["rotated_about_origin._xyz"] = m._xyz.numpy()
        m = copy.deepcopy(a); m.apply_transformation_on_xyz(T); out["transformed._xyz"] = m._xyz.numpy()
        m = copy.deepcopy(a); m.apply_rotation_on_splats(R); out["splats_rotated._rotation"] = m._rotation.numpy()
        # twice in a row with two different matrices (a pose update after the initial pose, pegasus_setup.py:160-193)
        R2 = torch.from_numpy(Rot.random(random_state=8).as_matrix()).float()
        m.apply_rotation_on_splats(R2)
        out["R2"] = R2.numpy(); out["splats_rotated_twice._rotation"] = m._rotation.numpy()
    out["compat_names"] = np.asarray(["utils.general_utils.build_rotation", "utils.general_utils.build_scaling_rotation",
                                      "utils.general_utils.strip_symmetric"])
    np.savez_compressed(OUT / "gs_model.npz", **out)
    print("gs_model.npz written:", len(out), "arrays")


def _mask_images(rng, colors, H, W):
    """[3,H,W] float32: pixels ON a semantic colour, AT the 0.1 sphere around one (+- 1e-6, +- 1e-7 and the fp32 neighbours
    of the decision), black background, free values up to 1.9 (beyond the uint8 range after * 255)."""
    K = colors.shape[0]
    img = np.zeros((H, W, 3), np.float32)
    kind = rng.integers(0, 5, (H, W))
    for y in range(H):
        for x in range(W):
            k = int(rng.integers(0, K))
            c = colors[k].astype(np.float64)
            if kind[y, x] == 0:
                img[y, x] = colors[k]
            elif kind[y, x] == 1:
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                r = 0.1 + rng.choice([-1e-6, -1e-7, -2e-8, 0.0, 2e-8, 1e-7, 1e-6])
                img[y, x] = (c + r * d).astype(np.float32)
            elif kind[y, x] == 2:
                d = np.zeros(3); d[int(rng.integers(0, 3))] = rng.choice([-1.0, 1.0])      # along one channel
                r = np.float64(np.float32(0.1)) + rng.choice([-1e-7, -1e-8, 0.0, 1e-8, 1e-7])
                img[y, x] = (c + r * d).astype(np.float32)
            elif kind[y, x] == 3:
                img[y, x] = rng.uniform(0, 1.9, 3).astype(np.float32)
            # kind 4: background
    return np.ascontiguousarray(img.transpose(2, 0, 1))


class _Obj:
    pass


def render_masks_golden():
    """/root/reference/src/gs/render.py:36-129 -- render_silhouette_mask, render_visib_mask,
    render_semanticsegmentation_mask -- with the reference's GaussianModel as scene container and ``render`` replaced by a
    stand-in that returns prepared images and records the scene it was handed.  Object ids {1, 2, 4, 6} of a 6-colour set:
    ids 3 and 5 are missing from the dictionary."""
    import copy
    import torch
    _reference_env()
    with _cuda_is_cpu():
        from src.gs.gaussian_model import GaussianModel as RefModel
        from src.utility import graphic_utils as gu
        calls = []
        queue = []

        def render(cam, scene, pipe, bg):
            calls.append({k: getattr(scene, k).detach().clone().numpy() for k in _RAW})
            img = queue.pop(0)
            return {"render": torch.from_numpy(img.copy()), "depth": torch.zeros((1,) + img.shape[1:])}

        fake = types.ModuleType("gaussian_renderer")
        fake.render, fake.network_gui = render, types.ModuleType("network_gui")
        saved = sys.modules.get("gaussian_renderer")
        sys.modules["gaussian_renderer"] = fake
        sys.modules.pop("src.gs.render", None)
        try:
            from src.gs import render as ref_render
            from utils.sh_utils import RGB2SH
            rng = np.random.default_rng(62)
            H, W, K = 24, 40, 6
            color_set = gu.generate_colors(K)
            colors = color_set.numpy()
            env = _raw_model(rng, 50, RefModel)
            objects = {}
            for oid, n in ((1, 5), (2, 9), (4, 3), (6, 7)):
                o = _raw_model(rng, n, RefModel)
                o._features_dc_semantics = RGB2SH(color_set[oid - 1])                    # pegasus.py:223-232
                o._features_rest_semantics = torch.asarray([0, 0, 0])
                objects[oid] = o
            out = {"colors": colors, "object_ids": np.asarray(list(objects)), "shape": np.asarray([H, W, K])}
            out.update(_raw(env, "env"))
            for oid, o in objects.items():
                out.update(_raw(o, f"obj{oid}"))
            cam = _Obj()
            bg = torch.zeros(3)
            # visib
            vis_img = _mask_images(rng, colors, H, W)
            queue.append(vis_img)
            masks, seg_image = ref_render.render_visib_mask(cam, env, objects, color_set, H, W, None, bg)
            out["visib.image"], out["visib.masks"], out["visib.seg_image"] = vis_img, masks, np.asarray(seg_image)
            out.update({"visib.scene" + k: v for k, v in calls.pop().items()})
            # semantic segmentation (uint8 cast, values beyond 1.0 included)
            sem_img = _mask_images(rng, colors, H, W)
            queue.append(sem_img)
            sem = ref_render.render_semanticsegmentation_mask(cam, env, objects, color_set, H, W, None, bg, False)
            out["semseg.image"], out["semseg.out"] = sem_img, sem
            out.update({"semseg.scene" + k: v for k, v in calls.pop().items()})
            # silhouettes: one render per object of the dictionary
            sil_imgs = [_mask_images(rng, colors, H, W) for _ in objects]
            queue.extend(sil_imgs)
            sil = ref_render.render_silhouette_mask(cam, objects, env, W, H, color_set, None, bg)
            out["silhouette.images"], out["silhouette.masks"] = np.stack(sil_imgs), sil
            for i, c in enumerate(calls):
                out.update({f"silhouette.scene{i}" + k: v for k, v in c.items()})
            assert not queue
        finally:
            sys.modules.pop("src.gs.render", None)
            if saved is not None:
                sys.modules["gaussian_renderer"] = saved
            else:
                sys.modules.pop("gaussian_renderer", None)
    np.savez_compressed(OUT / "render_masks.npz", **out)
    print("render_masks.npz written:", {k: (v.dtype, v.shape) for k, v in out.items() if k.endswith(("masks", ".out"))})


def pose_recursion_golden():
    """/root/reference/src/gs/pegasus_setup.py:160-208 -- dynamic_object_pose (step 0) then update_object_pose for steps
    1..S-1 -- on the reference's GaussianModel, trajectory = the committed fixture (body 1, first 200 steps).  The xyz and
    the splat quaternions after selected steps are stored.  apply_rotation_on_sh needs e3nn (absent): replaced by a no-op,
    SH rotation is not pinned by this file."""
    import torch
    _reference_env()
    with _cuda_is_cpu():
        from src.gs.gaussian_model import GaussianModel as RefModel
        from src.gs import pegasus_setup as ps
        tq = np.load(OUT / "simulation_steps_body1_first200.npz")["t_q_xyzw"]
        S = tq.shape[0]
        setup = object.__new__(ps.PegasusSetup)
        setup.object_trajectory = {"1": {str(s): {"t": tq[s, :3].tolist(), "q": tq[s, 3:].tolist()} for s in range(S)}}
        rng = np.random.default_rng(63)
        obj = _raw_model(rng, 96, RefModel)
        obj.apply_rotation_on_sh = lambda R: None
        out = _raw(obj, "canonical")
        objs = setup.dynamic_object_pose({1: obj})
        keep = (0, 1, 2, 17, 60, 120, 199)
        out["steps"] = np.asarray(keep)
        xyz, rot = [], []
        for s in range(S):
            if s > 0:
                objs = setup.update_object_pose(objs, s)
            if s in keep:
                xyz.append(objs[1]._xyz.numpy().copy()); rot.append(objs[1]._rotation.numpy().copy())
        out["xyz"], out["rotation"] = np.stack(xyz), np.stack(rot)
        out["R_init"], out["t_init"] = obj.R_init.numpy(), obj.t_init.numpy()
        out["last_transformation_matrix"] = obj.transformation_matrix.numpy()
    np.savez_compressed(OUT / "pose_recursion.npz", **out)
    print("pose_recursion.npz written")


if __name__ == "__main__":
    main()
