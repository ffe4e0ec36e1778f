"""Stage times of ONE call against the number of views in it (HIP events at the stage boundaries, pgr_forward_batch_profiled):
where a call stops being bound by the critical path of its longest lists / chunks and becomes throughput.
    python scripts/batch_size_curve.py [c3|c5]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import _lib, frames as F, rasterizer as R

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
cloud, views, label = bench.build_workload(wl, 1.0, 512 if wl == "c3" else 200)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
pick = [views[(k * (len(views) - 1)) // 31] for k in range(32)]          # 32 cameras spread over the set (ordered by elevation)
print(f"# {label}: stage milliseconds of one raster-only call (RGB + depth) by the number of views in it; median of 7 calls; cameras spread over the set")
print(f"{'views':>5s} " + " ".join(f"{n:>12s}" for n in _lib.STAGE_NAMES) + f" {'sum':>9s} {'sum/view':>9s} {'composite/view':>15s}")
for nv in (1, 2, 4, 8, 16, 32):
    specs = [fr.view_spec(v) for v in pick[:: 32 // nv][:nv]]
    rows = []
    for _ in range(9):
        ms = []
        R.forward_views(fr.means3d, fr.opacities, specs, shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3,
                        want_radii=False, stage_ms=ms, tie_index=fr.tie_index, tie_inv=fr.tie_inv)
        rows.append(ms)
    m = np.median(np.asarray(rows[2:]), axis=0)
    print(f"{nv:5d} " + " ".join(f"{x:12.4f}" for x in m) + f" {m.sum():9.4f} {m.sum() / nv:9.4f} {m[4] / nv:15.4f}")
