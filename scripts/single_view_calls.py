"""N gaussian_renderer.render() calls on the C3 scene, one camera each -- the drop-in path of an unchanged PEGASUS loop
(/root/reference/pegasus.py:254-271).  Run under `rocprofv3 --kernel-trace` by scripts/single_view_trace.sh; prints the
un-profiled per-call time when run on its own.  python scripts/single_view_calls.py [calls] [workload] [concatenated]"""
import sys
import time
from argparse import ArgumentParser
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "compat"))
import bench
from pegasus_amd import gaussian_renderer as GR
from pegasus_amd.cameras import Camera
from pegasus_amd.gaussian_model import GaussianModel
from arguments import PipelineParams

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 32
workload = sys.argv[2] if len(sys.argv) > 2 else "c3"
dev = "cuda:0"
cloud, views, label = bench.build_workload(workload, 1.0, max(n_calls, 8))
pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling, cloud.rotation, device=dev)
cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width, image_height=v.height,
               gt_alpha_mask=None, image_name=str(i), uid=i, data_device=dev) for i, v in enumerate(views[:n_calls])]
if len(sys.argv) > 3 and sys.argv[3] == "concatenated":      # A/B: get_features' torch.cat (kept between calls) instead of the stored layout
    GR._colour = lambda pc_, pipe_, cam_, override: dict(shs=GR.kept_activation(pc_, "get_features"))
pipe = PipelineParams(ArgumentParser())
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for c in cams[:4]:
        GR.render(c, pc, pipe, bg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in cams:
        GR.render(c, pc, pipe, bg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(cams)
print(f"{label}: render() {dt * 1e3:.3f} ms per call ({1 / dt:.0f} views/s), {len(cams)} calls")
