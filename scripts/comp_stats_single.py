"""Debug (-DPGR_COMP_STATS build): the compositor's walk of ONE view launched alone -- the drop-in path's launch shape."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import _lib, frames as F, rasterizer as R

L = _lib.lib()
handle = C.CDLL(str(_lib.LIB_PATH))
cloud, views, label = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "c3", 1.0, 8)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, device="cuda:0",
                     spatial_order=False)
out = (C.c_ulonglong * 32)()
for v in views[:4]:
    R.forward_views(fr.means3d, fr.opacities, [fr.view_spec(v)], shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3)
    handle.pgr_debug_comp_stats(out, 1)
    o = list(out)
    print(f"{label}: waves {o[5]}  batches {o[6]}  walked {o[0] / 1e6:.2f} M  live {o[1] / 1e6:.2f} M  evaluated {o[2] / 1e6:.2f} M | longest wave "
          f"{o[15]} clocks (100 MHz counter: {o[15] / 100:.1f} us) with {o[16]} batches; waves with > 16 / 32 / 64 / 96 batches: {o[17]} / {o[18]} / {o[19]} / {o[20]}; "
          f"mean wave {o[21] / max(o[5], 1) / 100:.2f} us")
