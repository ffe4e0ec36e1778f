"""The layered silhouette pass alone (for kernel traces): 3 + 10 batches of 32 C3 views."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from pegasus_amd import frames as F
cloud, views, label = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "c3", 1.0, 64)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:32]]
out = torch.empty((32, fr.K, 800, 800), dtype=torch.uint8, device="cuda:0")
for _ in range(3): fr.render_silhouettes(specs, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fr.render_silhouettes(specs, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"{label}: layered silhouettes, 32 views: {dt*1e3:.2f} ms per batch = {dt/32*1e3:.4f} ms per view")
