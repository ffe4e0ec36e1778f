"""Generates tests/golden/*.npz from the parts of /root/reference that import in THIS container.

Run here only (``python tests/golden/make_golden_from_reference.py``); /root/reference does not
exist on the GPU box and nothing at test/bench time reads it.  Only inputs and the reference's
OUTPUTS are stored -- no reference source text.

What imports: bop_toolkit view_sampler (with imageio/png stubbed),
src.utility.pose_interpolation, src.utility.graphic_utils (``.to('cuda')`` patched to identity
for generate_colors), bop_toolkit_lib.misc (calc_2d_bbox: the bounding boxes of scene_gt_info.json).
The rasterizer itself is absent, so no rasterizer golden can be produced.

``python tests/golden/make_golden_from_reference.py [section ...]`` regenerates only the named sections
(views, interpolation, graphic_utils, trajectory, gt_info); default: all.
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
    if only and only <= {"gt_info"}:
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


if __name__ == "__main__":
    main()
