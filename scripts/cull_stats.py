import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from pegasus_amd import frames as F, rasterizer as R
cloud, views, label = bench.build_workload("c3", 1.0, 64)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:32]]
vis = R.block_visibility(fr.means3d, specs, scales=fr.scales, rotations=fr.rotations)
print("blocks", vis.shape, "culled fraction", 1 - float(vis.float().mean()))
res = R.forward_views(fr.means3d, fr.opacities, specs[:4], shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3, want_radii=True)
for v, r in enumerate(res):
    seen = (r["radii"] > 0)
    blk = torch.nn.functional.pad(seen, (0, (-seen.numel()) % 64)).reshape(-1, 64)
    print("view", v, "V/N", float(seen.float().mean()), "blocks with any visible", float(blk.any(1).float().mean()), "vis bit set", float(vis[:, v].float().mean()))
