"""Host-side cost of one gaussian_renderer.render() call (cProfile over N calls on the C3 scene)."""
import cProfile
import io
import pstats
import sys
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

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
cloud, views, label = bench.build_workload("c3", 1.0, n_calls)
pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling, cloud.rotation, device=dev)
cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width, image_height=v.height,
               gt_alpha_mask=None, image_name=str(i), uid=i, data_device=dev) for i, v in enumerate(views[:n_calls])]
pipe = PipelineParams(ArgumentParser())
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for c in cams[:4]:
        GR.render(c, pc, pipe, bg)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for c in cams:
        GR.render(c, pc, pipe, bg)
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
