import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import torch, bench
from pegasus_amd import frames as F, rasterizer as R
cloud, views, label = bench.build_workload("c2", 1.0, 64)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views]
B = 32
fa, fb = fr.alloc_frames(B, 800, 800), fr.alloc_frames(B, 800, 800)
def run(n):
    pend = None
    for i in range(n):
        h = fr.render_frames_async(specs[(i % 2) * B:(i % 2) * B + B], fa if i % 2 == 0 else fb, slot=i % 2)
        if pend is not None: pend.wait()
        pend = h
    pend.wait()
run(4); torch.cuda.synchronize()
t0 = time.perf_counter(); run(16); torch.cuda.synchronize(); t1 = time.perf_counter()
print("async pipeline: ms per batch", (t1 - t0) / 16 * 1e3, "K", fr.K)
# enqueue cost alone
t0 = time.perf_counter(); h = fr.render_frames_async(specs[:B], fa, slot=0); t1 = time.perf_counter(); h.wait(); t2 = time.perf_counter()
print("enqueue ms", (t1 - t0) * 1e3, "wait ms", (t2 - t1) * 1e3, "num_instances max", max(R.last_forward_info().get("num_instances", [0])))
ms = []
fr.render_frames(specs[:B], fa, stage_ms=ms); print("stage ms per batch", [round(x, 3) for x in ms], "sum", sum(ms))
pr = cProfile.Profile(); pr.enable(); run(8); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
