"""Host-side profile of consecutive single-view gaussian_renderer.render() calls on the C3 scene (cProfile, top entries by
cumulative and own time) next to the wall time per call: where the microseconds between two calls' kernels go."""
import cProfile
import pstats
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

dev = torch.device("cuda:0")
cloud, views, label = bench.build_workload("c3", 1.0, 64)
pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling, cloud.rotation, device=dev)
cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width, image_height=v.height,
               gt_alpha_mask=None, image_name=str(i), uid=i, data_device=dev) for i, v in enumerate(views[:48])]
pipe = PipelineParams(ArgumentParser())
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for c in cams[:8]:
        GR.render(c, pc, pipe, bg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in cams:
        GR.render(c, pc, pipe, bg)
    torch.cuda.synchronize()
    print(f"{label}: render() {(time.perf_counter() - t0) / len(cams) * 1e3:.4f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for c in cams:
        GR.render(c, pc, pipe, bg)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
st.sort_stats("cumulative").print_stats(14)
