"""Time of the silhouette data point ('seg_sil') on C3: the layered pass (one call for all K objects) against round 2's
K single-object passes, per batch of 32 views; the two results must be bit-equal."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import frames as F
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
cloud, views, label = bench.build_workload(wl, 1.0, 64)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:32]]
outs = {}
only = sys.argv[2] if len(sys.argv) > 2 else ""       # "layered": the one-pass form alone (for a kernel trace of it)
for name, fn in (("layered (one pass)", fr.render_silhouettes), ("per object (K passes)", fr.render_silhouettes_per_object)):
    if only and not name.startswith(only):
        continue
    out = torch.empty((32, fr.K, 800, 800), dtype=torch.uint8, device="cuda:0")
    for _ in range(3): fn(specs, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn(specs, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    outs[name] = out
    print(f"{label}: silhouettes of {fr.K} objects, 32 views, {name}: {dt*1e3:.2f} ms per batch = {dt/32*1e3:.4f} ms per view")
if len(outs) == 2:
    a, b = outs.values()
    print("bit-equal:", bool(torch.equal(a, b)), " mask pixels:", int(a.sum()))
