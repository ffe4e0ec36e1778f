import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "..")
import bench
from pegasus_amd import _lib, frames as F
L = _lib.lib(); handle = C.CDLL(str(_lib.LIB_PATH))
B = 16
cloud, views, label = bench.build_workload("c3", 1.0, B)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:B]]
fr.render_frames(specs, None, masks=False)
cap = 1 << 18
buf = np.zeros((cap, 12), np.uint64)
handle.pgr_debug_sort_timing(buf.ctypes.data_as(C.c_void_p), cap)
fr.render_frames(specs, None, masks=False)
n = handle.pgr_debug_sort_timing(buf.ctypes.data_as(C.c_void_p), cap)
rec = buf[:n].astype(np.float64)
for tier in (0, 1, 2):
    r = rec[rec[:, 9] == tier]
    l0 = r[:, 0]
    print("tier", tier, "load+zero percentiles", np.percentile(l0, [5, 25, 50, 75, 95, 99]).round())
    t0 = r[:, 11] - r[:, 11].min()
    order = np.argsort(t0)
    k = len(r) // 8
    for q in range(8):
        sel = order[q * k:(q + 1) * k]
        print(f"   start-time octile {q}: start {t0[sel].mean():.0f}  load+zero {l0[sel].mean():.0f}  total {r[sel, :9].sum(1).mean():.0f}  keys {r[sel, 10].mean():.0f}")
    c = np.corrcoef(r[:, 10], l0)[0, 1]
    print("   corr(keys, load+zero)", round(c, 3))
