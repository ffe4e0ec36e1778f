"""Host-side cost of enqueuing one batch (render_frames_async returns before the GPU finishes)."""
import cProfile
import pstats
import sys
import time
import torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import frames as F

cloud, views, label = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "c2", 1.0, 16)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                     sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:16]]
fa, fb = fr.alloc_frames(16, 800, 800), fr.alloc_frames(16, 800, 800)
for i in range(3):
    fr.render_frames_async(specs, fa, slot=0).wait()
torch.cuda.synchronize()
ts = []
for i in range(60):
    t0 = time.perf_counter()
    h = fr.render_frames_async(specs, fa if i % 2 == 0 else fb, slot=i % 2)
    t1 = time.perf_counter()
    h.wait()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
print(label, "enqueue ms:", [round(a * 1e3, 2) for a, _ in ts], "wait ms:", [round(b * 1e3, 2) for _, b in ts])
import gc
print('gc counts', gc.get_count(), 'spikes at', [i for i, (a, b) in enumerate(ts) if a + b > 3e-3])
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    fr.render_frames_async(specs, fa, slot=0).wait()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
