#!/usr/bin/env python3
"""bench.py -- rendered views/sec (RGB + depth + mask) on the 2 M-Gaussian merged scene @800x800.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` ; for N>1 it is launched through
``python -m torch.distributed.run`` with one rank per GPU.  One "step" = one pass of the rasterizer hot
path over one camera of the synthetic merged scene (BASELINE.json configs[2]: environment + 8 objects,
2.0 M Gaussians, 800x800).  Views shard across ranks with no data-path collective (weak scaling: every
rank renders K views of its own shard of the camera list); rank 0 prints ONE JSON line.

The JSON line also carries
  roofline      -- the dominant kernel's algorithmic bytes / its average duration measured live with
                   HIP events on the launch stream (pgr_forward_profiled), against 8 TB/s HBM peak
  cpu_baseline  -- the CPU oracle ("port": there is no reference CPU rasterizer) timed on this box's
                   host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink Gaussian counts (debug only; INVALID as a result)")
    ap.add_argument("--views", type=int, default=64, help="distinct cameras cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--profile-steps", type=int, default=8, help="steps measured per-stage with HIP events")
    return ap.parse_args()


def build_workload(name, scale, n_views):
    from pegasus_amd import scenes
    if name == "c1":
        cloud, views = scenes.scene_c1()
        label = "C1 10k-Gaussian cube, 256x256"
    elif name == "c2":
        cloud, views = scenes.scene_c2(n=int(150_000 * scale), n_views=n_views)
        label = "C2 single object 150k Gaussians, 800x800 hemisphere views"
    elif name == "c5":
        cloud, views = scenes.scene_c5(scale=scale, n_views=n_views)
        label = "C5 5M-Gaussian scene, 800x800"
    else:
        cloud, views = scenes.scene_c3(scale=scale, n_views=n_views)
        label = "C3 merged env + 8 objects, 2M Gaussians, 800x800"
    if scale != 1.0:
        label += f" [scale={scale}: NOT the baseline config]"
    return cloud, views, label


def algorithmic_bytes(N, V, I, P):
    """SURVEY.md section 8d / BASELINE.md section 3:  B = 16 N + 272 V + 88 I + 16 P  bytes per view, and the
    per-stage split it is the sum of."""
    per_stage = {
        "preprocess": 12 * N + 224 * V + 4 * N + 48 * V,
        "scan": 0,
        "emit": 12 * I,
        "sort": 24 * I,
        "ranges": 8 * I,
        "composite": 44 * I + 16 * P,
    }
    return 16 * N + 272 * V + 88 * I + 16 * P, per_stage


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from pegasus_amd import _lib
    from pegasus_amd import diff_gaussian_rasterization as dgr
    L = _lib.lib()

    # every rank builds the same scene (replicated: 472 MB at 2 M Gaussians) and takes views rank::world
    n_views_total = max(args.views * world, world)
    cloud, views, label = build_workload(args.workload, args.scale, n_views_total)
    my_views = views[rank::world] or views
    act = cloud.activated()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    means, opac, scales, rots, shs = (t(act[k]) for k in ("means3d", "opacities", "scales", "rotations", "shs"))
    bg = torch.zeros(3, device=dev)
    settings = []
    for v in my_views:
        settings.append(dgr.GaussianRasterizationSettings(
            v.height, v.width, v.tanfovx, v.tanfovy, bg, 1.0, t(v.world_view_transform), t(v.full_proj_transform),
            3, t(v.camera_center), False, False))
    W, H = my_views[0].width, my_views[0].height
    P = W * H

    def step(i, want_aux=False):
        s = settings[i % len(settings)]
        return dgr.rasterize_gaussians(means, None, shs, None, opac, scales, rots, None, s, want_aux=want_aux)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- per-view statistics and per-stage HIP-event timing (outside the timed region) ----
    stage_ms = np.zeros((0, _lib.PGR_NUM_STAGES))
    stats = []
    if rank == 0:
        rows = []
        for i in range(max(1, min(args.profile_steps, len(settings)))):
            s = settings[i]
            color = torch.empty((3, H, W), device=dev)
            depth = torch.empty((1, H, W), device=dev)
            radii = torch.empty((cloud.n,), dtype=torch.int32, device=dev)
            ncontrib = torch.empty((H, W), dtype=torch.int32, device=dev)
            scene = _lib.PgrScene(n=cloud.n, means3d=means.data_ptr(), opacities=opac.data_ptr(),
                                  scales=scales.data_ptr(), rotations=rots.data_ptr(), cov3d_precomp=None,
                                  shs=shs.data_ptr(), colors_precomp=None, sh_degree=3, sh_stride=16,
                                  scale_modifier=1.0)
            cam = _lib.PgrCamera(W, H, float(s.tanfovx), float(s.tanfovy), s.viewmatrix.data_ptr(),
                                 s.projmatrix.data_ptr(), s.campos.data_ptr(), s.bg.data_ptr())
            outs = _lib.PgrOutputs(color.data_ptr(), depth.data_ptr(), radii.data_ptr(), None, ncontrib.data_ptr())
            info = dgr.last_forward_info()
            ws = info["workspace"]
            ms = (C.c_float * _lib.PGR_NUM_STAGES)()
            need = C.c_int64(0)
            _lib.check(L.pgr_forward_profiled(C.byref(scene), C.byref(cam), C.byref(outs), C.c_void_p(ws.data_ptr()),
                                              ws.numel(), info["used_max_instances"], C.byref(need),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream), ms),
                       "pgr_forward_profiled")
            rows.append(list(ms))
            V = int((radii > 0).sum().item())
            stats.append(dict(N=cloud.n, V=V, I=int(need.value), evals=int(ncontrib.sum(dtype=torch.int64).item())))
        stage_ms = np.asarray(rows)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_views = args.steps * world
    value = total_views / elapsed
    N = cloud.n
    V = float(np.mean([s["V"] for s in stats]))
    I = float(np.mean([s["I"] for s in stats]))
    evals = float(np.mean([s["evals"] for s in stats]))
    B_view, per_stage_bytes = algorithmic_bytes(N, V, I, P)
    mean_ms = stage_ms.mean(axis=0)
    dom = int(np.argmax(mean_ms))
    dom_name = _lib.STAGE_NAMES[dom]
    dom_bytes = per_stage_bytes[dom_name]
    achieved = dom_bytes / (mean_ms[dom] * 1e-3) / 1e9 if mean_ms[dom] > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "kernel_ms": round(float(mean_ms[dom]), 4), "algorithmic_bytes_per_launch": int(dom_bytes),
        "stage_ms": {k: round(float(m), 4) for k, m in zip(_lib.STAGE_NAMES, mean_ms)},
        "whole_path": {"bytes_per_view": int(B_view), "achieved": round(B_view * value / 1e9, 2),
                       "frac": round(B_view * value / 1e9 / HBM_PEAK_GBS, 5)},
        "composite_evals_per_s": round(evals / (mean_ms[5] * 1e-3), 1) if mean_ms[5] > 0 else None,
        "N": N, "V": round(V), "I": round(I), "P": P,
    }

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        cores = os.cpu_count() or 1
        n_done, t_cpu = 0, 0.0
        while n_done < len(my_views) and (n_done == 0 or t_cpu + t_cpu / n_done < args.cpu_budget_s):
            v = my_views[n_done]
            t1 = time.perf_counter()
            oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=cores, want_binning=False)
            t_cpu += time.perf_counter() - t1
            n_done += 1
        cpu = {"value": round(n_done / t_cpu, 4), "unit": "views/s", "cores": cores, "kind": "port",
               "sample": f"first {n_done} view(s) of the same scene and cameras, oracle/pgr_oracle.c with OpenMP "
                         f"({cores} threads); no reference CPU rasterizer exists"}

    line = {
        "metric": "rendered views/sec (RGB+depth) on 2M-Gaussian scene @800x800",
        "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": label, "gaussians": N, "width": W, "height": H, "views_per_step": 1,
                   "distinct_views": len(my_views), "outputs": "color[3,H,W] f32 + depth[1,H,W] f32 + radii",
                   "parallelism": f"view-shard x{world}"},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
