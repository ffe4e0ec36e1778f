"""Per-tile list length distribution of a workload (which sort tier handles how many keys)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from helpers import fetch_workspace
from pegasus_amd import frames as F, rasterizer as R

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_total = 512 if workload == "c3" else (200 if workload == "c5" else 64)
cloud, views, label = bench.build_workload(workload, 1.0, n_total)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
edges = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 1 << 30]
tot_l = np.zeros(len(edges) - 1); tot_k = np.zeros(len(edges) - 1)
nv = 16
for v in [views[(k * (len(views) - 1)) // (nv - 1)] for k in range(nv)]:      # spread over the camera set (ordered by elevation)
    R.forward_views(fr.means3d, fr.opacities, [fr.view_spec(v)], shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3, want_radii=True)
    torch.cuda.synchronize()
    w = fetch_workspace(0, cloud.n, v.width, v.height)
    lens = (w["ranges"][:, 1].astype(np.int64) - w["ranges"][:, 0])
    for b in range(len(edges) - 1):
        m = (lens > edges[b]) & (lens <= edges[b + 1])
        tot_l[b] += m.sum(); tot_k[b] += lens[m].sum()
print(label, "per view:")
for b in range(len(edges) - 1):
    print(f"  ({edges[b]:6d}, {edges[b+1]:10d}]  lists {tot_l[b]/nv:8.1f}  keys {tot_k[b]/nv/1e6:7.3f} M  ({tot_k[b]/tot_k.sum():.1%})")
print("  max list", int(lens.max()))
