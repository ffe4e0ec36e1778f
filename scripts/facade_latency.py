"""Per-call cost of the drop-in `gaussian_renderer.render()` -- what PEGASUS's own loops call, one view at a time, under
torch.no_grad() (pegasus.py:248,271) -- on the 2 M-Gaussian scene at 800x800: with the host fetching every image (the
reference's `.cpu()` idiom, src/gs/render.py:19) and with the images left on the device."""
import sys
import time
from argparse import ArgumentParser
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "compat")
import bench
from pegasus_amd import gaussian_renderer as GR
from pegasus_amd.gaussian_model import GaussianModel
from pegasus_amd.cameras import Camera
from arguments import PipelineParams

dev = "cuda:0"
cloud, views, label = bench.build_workload("c3", 1.0, 64)
pc = GaussianModel.from_arrays(cloud.xyz, cloud.features_dc, cloud.features_rest, cloud.opacity, cloud.scaling, cloud.rotation,
                               device=dev)
cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
               image_height=v.height, gt_alpha_mask=None, image_name=str(i), uid=i, data_device=dev) for i, v in enumerate(views)]
pipe = PipelineParams(ArgumentParser())
bg = torch.zeros(3, device=dev)
torch.set_grad_enabled(False)
for c in cams[:4]:
    GR.render(c, pc, pipe, bg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in cams:
    pkg = GR.render(c, pc, pipe, bg)
torch.cuda.synchronize()
t1 = time.perf_counter()
for c in cams:
    pkg = GR.render(c, pc, pipe, bg)
    img = pkg["render"].cpu()
t2 = time.perf_counter()
n = len(cams)
print(f"{label}: render() per call {1e3 * (t1 - t0) / n:.2f} ms ({n / (t1 - t0):.0f} views/s) images left on the device; "
      f"{1e3 * (t2 - t1) / n:.2f} ms ({n / (t2 - t1):.0f} views/s) with .cpu() of the colour image per call")

# where the per-call time goes
from pegasus_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
import math


def timed(fn, reps=64):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


t_act = timed(lambda i: (pc.get_xyz, pc.get_opacity, pc.get_scaling, pc.get_rotation, pc.get_features))
xyz, op, sc, rot, shs = pc.get_xyz, pc.get_opacity, pc.get_scaling, pc.get_rotation, pc.get_features


def raster(i):
    c = cams[i % n]
    s = GaussianRasterizationSettings(int(c.image_height), int(c.image_width), math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), bg, 1.0,
                                      c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False)
    return GaussianRasterizer(s)(means3D=xyz, means2D=None, shs=shs, colors_precomp=None, opacities=op, scales=sc, rotations=rot,
                                 cov3D_precomp=None)


t_ras = timed(raster)
print(f"  activations + feature cat {t_act:.2f} ms; GaussianRasterizer alone {t_ras:.2f} ms")
