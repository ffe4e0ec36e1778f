"""Debug: per-wave work and cycles of the compositor backward (a -DPGR_BWD_STATS build copied over
csrc/libpegasus_raster.so): is the kernel the sum of its waves or the length of its longest one?"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pegasus_amd import _lib, diff_gaussian_rasterization as dgr, scenes

dev = torch.device("cuda:0")
cloud, views = scenes.scene_c3(scale=float(sys.argv[1]) if len(sys.argv) > 1 else 1.0, n_views=2)
a = cloud.activated()
tt = lambda arr, rg=True: torch.from_numpy(np.ascontiguousarray(arr)).to(dev).requires_grad_(rg)
means, op, sc, rot, shs = tt(a["means3d"]), tt(a["opacities"].reshape(-1, 1)), tt(a["scales"]), tt(a["rotations"]), tt(a["shs"])
v = views[0]
s = dgr.GaussianRasterizationSettings(v.height, v.width, v.tanfovx, v.tanfovy, torch.zeros(3, device=dev), 1.0,
                                      tt(v.world_view_transform, False), tt(v.full_proj_transform, False), 3,
                                      tt(v.camera_center, False), False, False)
for _ in range(2):
    color, radii, depth = dgr.GaussianRasterizer(s)(means, torch.zeros_like(means, requires_grad=True), op, shs=shs, scales=sc, rotations=rot)
    (color.sum() + depth.sum()).backward()
torch.cuda.synchronize()
handle = C.CDLL(str(_lib.LIB_PATH))
n = 4 * ((v.width + 15) // 16) * ((v.height + 15) // 16)
buf = np.zeros((n, 4), np.uint64)
handle.pgr_debug_bwd_stats(buf.ctypes.data_as(C.c_void_p), n)
b = buf.astype(np.float64)
cyc = b[:, 3]
print(f"waves {n}  sum of wave cycles {cyc.sum() / 1e6:.1f} M  longest wave {cyc.max() / 1e6:.3f} M cycles  mean {cyc.mean() / 1e3:.1f} k")
print(f"  sum / (1024 SIMDs x 8 waves) = {cyc.sum() / 8192 / 1e6:.3f} M cycles if perfectly packed")
print(f"  entries parked {b[:, 1].sum() / 1e6:.2f} M  with a valid pixel {b[:, 2].sum() / 1e6:.2f} M  cycles per valid entry (own wave) {cyc.sum() / max(b[:, 2].sum(), 1):.0f}")
i = int(cyc.argmax())
print(f"  longest wave: n_used {b[i, 0]:.0f}  parked {b[i, 1]:.0f}  valid {b[i, 2]:.0f}  cycles/valid {b[i, 3] / max(b[i, 2], 1):.0f}")
q = np.percentile(cyc, [50, 90, 99, 99.9])
print("  cycles percentiles 50/90/99/99.9:", (q / 1e3).round(1), "k")
