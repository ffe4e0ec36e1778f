"""Records-only frame sets (PgrOutputs color = depth = sem_* = NULL with `record` set, round 6) against full frame sets with
records beside the images: frames/s of the pipelined batch path over the 512 C3 cameras, interleaved on one box.
    python scripts/records_only_ab.py [full|records|both] [repeats]
Under `rocprofv3 --pmc WRITE_SIZE -- python3 scripts/records_only_ab.py records 1` the compositor's WRITE_SIZE per launch is
that of the records-only form (profiles/r06_records_only.txt)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from pegasus_amd import frames as F

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cloud, views, label = bench.build_workload("c3", 1.0, 512)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, sh_degree=3,
                     device="cuda:0")
specs = [fr.view_spec(v) for v in views]
B, H, W, SLOTS = 32, views[0].height, views[0].width, 3
sets = {"full": [fr.alloc_frames(B, H, W, records=True) for _ in range(SLOTS)],
        "records": [fr.alloc_frames(B, H, W, records=True, images=False) for _ in range(SLOTS)]}


def run(kind):
    pending = []
    for i in range(len(specs) // B):
        pending.append(fr.render_frames_async(specs[i * B:(i + 1) * B], sets[kind][i % SLOTS], slot=i % SLOTS))
        while len(pending) >= SLOTS:
            pending.pop(0).wait()
    while pending:
        pending.pop(0).wait()
    torch.cuda.synchronize()


kinds = ["full", "records"] if mode == "both" else [mode]
for k in kinds:
    run(k)                                    # workspaces reach their size
for rep in range(repeats):
    for k in kinds:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(k)
        dt = time.perf_counter() - t0
        print(f"{k:8s} {len(specs) / dt:8.1f} frames/s  ({label}; bytes written per frame by the epilogue: "
              f"{'3.84 MB record' if k == 'records' else '25.6 MB images and mask planes + 3.84 MB record'})")
a, b = sets["full"][0]["records"], sets["records"][0]["records"]
if mode == "both":
    from pegasus_amd import masks as M
    ga, gb = M.record_views(a, H, W, fr.K), M.record_views(b, H, W, fr.K)
    print("records equal:", all(bool(torch.equal(ga[k], gb[k])) for k in ga))
