"""Generates tests/golden/*.npz from the parts of /root/reference that import in THIS container.

Run here only (``python tests/golden/make_golden_from_reference.py``); /root/reference does not
exist on the GPU box and nothing at test/bench time reads it.  Only inputs and the reference's
OUTPUTS are stored -- no reference source text.

What imports: bop_toolkit view_sampler (with imageio/png stubbed),
src.utility.pose_interpolation, src.utility.graphic_utils (``.to('cuda')`` patched to identity
for generate_colors).  The rasterizer itself is absent, so no rasterizer golden can be produced.
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


if __name__ == "__main__":
    main()
