import sys, time, torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import frames as F, rasterizer as R
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
cloud, views, label = bench.build_workload(wl, 1.0, 64 if wl == "c2" else 512)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views]
B = 32
fa, fb = fr.alloc_frames(B, 800, 800), fr.alloc_frames(B, 800, 800)
def bv(i): return [specs[(i * B + k) % len(specs)] for k in range(B)]
def run(first, n, log=None):
    pend = None
    for i in range(first, first + n):
        t0 = time.perf_counter()
        h = fr.render_frames_async(bv(i), fa if i % 2 == 0 else fb, slot=i % 2)
        t1 = time.perf_counter()
        if pend is not None: pend.wait()
        t2 = time.perf_counter()
        if log is not None: log.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
        pend = h
    pend.wait()
run(0, 4); torch.cuda.synchronize()
log = []
t0 = time.perf_counter(); run(4, 12, log); torch.cuda.synchronize(); t1 = time.perf_counter()
print(wl, "ms per step", (t1 - t0) / 12 * 1e3, "hint", R.capacity_hints())
print("(enqueue ms, wait ms) per step:", log)
