"""Soak test: every one of the 512 C3 frames rendered several times (pipelined, alternating slots) must hash to the
same bits each time -- catches races (wave-level LDS hand-offs, atomics in binning, sort paths) that single runs miss."""
import sys
import time
import torch
sys.path.insert(0, ".")
import bench
from pegasus_amd import frames as F

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 4
workload = sys.argv[2] if len(sys.argv) > 2 else "c3"
cloud, views, label = bench.build_workload(workload, 1.0, 512 if workload == "c3" else 128)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                     sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views]
B = 32
H, W = views[0].height, views[0].width
fa, fb = fr.alloc_frames(B, H, W), fr.alloc_frames(B, H, W)


def digest(f):
    parts = []
    for k in ("color", "depth", "seg", "masks"):
        t = f[k]
        x = t.view(torch.int32) if t.dtype == torch.float32 else t.to(torch.int32)
        w = torch.arange(1, x[0].numel() + 1, device=x.device, dtype=torch.int64).view(x[0].shape)
        parts.append((x.to(torch.int64) * w).flatten(1).sum(dim=1))          # order-sensitive checksum per frame
    return torch.stack(parts, dim=1)


ref = None
t0 = time.perf_counter()
for rep in range(repeats):
    sums = []
    pending = None
    for i in range(len(specs) // B):
        f = fa if i % 2 == 0 else fb
        h = fr.render_frames_async(specs[i * B:(i + 1) * B], f, slot=i % 2)
        if pending is not None:
            pending[0].wait()
            sums.append(digest(pending[1]))
        pending = (h, f)
    pending[0].wait()
    sums.append(digest(pending[1]))
    cur = torch.cat(sums).cpu()
    if ref is None:
        ref = cur
    else:
        bad = (cur != ref).any(dim=1).nonzero().flatten().tolist()
        print(f"repeat {rep}: {len(bad)} of {cur.shape[0]} frames differ", bad[:10])
        assert not bad
print(f"{label}: {repeats} x {ref.shape[0]} frames bit-identical across repeats ({time.perf_counter() - t0:.1f} s)")
