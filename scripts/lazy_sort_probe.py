"""Upper bounds of a lazy tile sort, MEASURED (build_variants/lib_lazy.so = -DPGR_LAZY_PROBE; PGR_LIB points at it).
The probe makes the bucket sort treat every key whose bucket starts behind P percent of its list as "back":
  mode 1: back keys are not ranked inside their bucket (the verdict's first form: stop at coarse depth buckets)
  mode 2: back keys are neither parked, ranked nor written (what an ideal front/back split would skip)
The lists behind the cut are wrong in both modes -- only the SORT STAGE's time means anything here (HIP events of
pgr_forward_batch_profiled through FrameRenderer.render_frames(stage_ms=...)).  python scripts/lazy_sort_probe.py [c3|c5]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import _lib, frames as F

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
L = _lib.lib()
handle = C.CDLL(str(_lib.LIB_PATH))
B = 32
cloud, views, label = bench.build_workload(wl, 1.0, 4 * B)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views]
frames = fr.alloc_frames(B, views[0].height, views[0].width)


def sort_ms():
    rows = []
    for rep in range(3):
        for b in range(4):
            ms = []
            fr.render_frames(specs[b * B:(b + 1) * B], frames, stage_ms=ms)
            rows.append(ms)
    return np.median(np.asarray(rows), axis=0)


handle.pgr_debug_set_lazy(0, 0)
base = sort_ms()
print(f"{label}: stage ms per view, normal sort: " + "  ".join(f"{n} {m / B:.4f}" for n, m in zip(_lib.STAGE_NAMES, base)))
for mode, what in ((1, "back keys not ranked (coarse buckets only)"), (2, "back keys not parked / ranked / written")):
    for pct in (75, 50, 25):
        handle.pgr_debug_set_lazy(mode, pct)
        m = sort_ms()
        print(f"  mode {mode} ({what}), front = {pct:2d} % of every list: tile_sort {m[3] / B:.4f} ms per view ({m[3] / base[3]:.3f} of normal)")
handle.pgr_debug_set_lazy(0, 0)
