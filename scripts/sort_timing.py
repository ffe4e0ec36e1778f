"""Debug: where a sort workgroup's time goes (a -DPGR_SORT_TIMING build copied over csrc/libpegasus_raster.so).
Cycles are s_memtime ticks of wave 0 of each workgroup; one record per list."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "..")
import bench
from pegasus_amd import _lib, frames as F

L = _lib.lib()
handle = C.CDLL(str(_lib.LIB_PATH))
workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = 16
cloud, views, label = bench.build_workload(workload, 1.0, B)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                     sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:B]]
fr.render_frames(specs, None, masks=False)
cap = 1 << 18
buf = np.zeros((cap, 12), np.uint64)
handle.pgr_debug_sort_timing(buf.ctypes.data_as(C.c_void_p), cap)
fr.render_frames(specs, None, masks=False)
n = handle.pgr_debug_sort_timing(buf.ctypes.data_as(C.c_void_p), cap)
rec = buf[:n].astype(np.float64)
names = ["load+zero", "min/max", "hist atomics", "totals", "scan", "scatter keys", "rank", "output", "obj marker"]
print(label, "records", n)
for tier, tname in enumerate(["256 thr (<=2048)", "512 thr (<=4096)", "1024 thr x 8 (<=8192; segments of partitioned lists too)", "1024 thr x 16 (longer)"]):
    r = rec[rec[:, 9] == tier]
    if not len(r):
        continue
    span = (r[:, 11].max() - r[:, 11].min() + r[:, :9].sum(1).max())
    print(f"  {tname}: lists/view {len(r) / B:.0f}  keys/list {r[:, 10].mean():.0f}  cycles/list {r[:, :9].sum(1).mean():.0f}"
          f"  (first start to last end: {span:.0f} cycles)")
    print("     " + "  ".join(f"{nm} {r[:, k].mean():.0f}" for k, nm in enumerate(names)))
