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
import math
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
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink Gaussian counts (debug only; INVALID as a result)")
    ap.add_argument("--views", type=int, default=512, help="distinct cameras cycled through (configs[2]: 512 views)")
    ap.add_argument("--batch", type=int, default=32,
                    help="views per step (one batch call); 16 default steps x 32 = the 512 views of configs[2], each once")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-semantic", action="store_true",
                    help="semantic image by a second full pass over the objects (default: fused into the scene pass)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking render_batch per step (no pipelining)")
    ap.add_argument("--serialize-slots", action="store_true",
                    help="A/B: make the two pipeline slots execute one after the other on the GPU (round-1 behaviour)")
    ap.add_argument("--input-order", action="store_true",
                    help="keep the scene in its input order (default: one-time Morton layout per object, outside the timed region)")
    ap.add_argument("--dynamic", action="store_true",
                    help="dynamic sequence: every frame is a TIME STEP with its own object poses (posed inside the "
                         "preprocess) and one camera, plus its BOP pose records")
    ap.add_argument("--raster-only", action="store_true", help="time only the full-scene RGB+depth pass (R), no masks")
    ap.add_argument("--slots", type=int, default=3, help="batches in flight (pipeline slots: workspaces, frame sets)")
    ap.add_argument("--streams", type=int, default=0, help="streams the slots share round-robin (0 = one per slot)")
    ap.add_argument("--step-log", action="store_true", help="per-step (enqueue, wait) host milliseconds on stderr")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also gather every batch's finished frames to rank 0 inside the timed region, in the "
                         "reference's on-disk precision (uint8 RGB, uint16 depth, uint8 masks; pegasus.py:347,355) -- the "
                         "one RCCL exchange of the path (SURVEY.md section 8e); asynchronous, overlapping the next batch")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="batches measured per-stage with HIP events (0 = the same batches as the timed steps)")
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
        "bin_count": 0,
        "bin_scatter": 12 * I,
        "tile_sort": 24 * I + 8 * I,
        "composite": 44 * I + 16 * P,      # + 16 P for the fused semantic image, not counted (SURVEY's figure)
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

    from pegasus_amd import _lib, frames as F, rasterizer
    _lib.lib()

    # every rank builds the same scene (replicated: 472 MB at 2 M Gaussians) and takes views rank::world
    B = max(1, args.batch)
    n_views_total = max(args.views, B) * world
    cloud, views, label = build_workload(args.workload, args.scale, n_views_total)
    my_views = views[rank::world] or views
    act = cloud.activated()
    fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                         cloud.object_id, sh_degree=3, device=dev, spatial_order=not args.input_order)
    fr.serialize_slots = args.serialize_slots
    fr.n_streams = args.streams or None
    specs = [fr.view_spec(v) for v in my_views]
    W, H = my_views[0].width, my_views[0].height
    P = W * H
    with_masks = not args.raster_only and fr.K > 0
    frames = fr.alloc_frames(B, H, W, masks=with_masks)      # frame buffers, reused (two sets: 2-deep pipeline)
    n_slots = max(1, args.slots)
    frame_sets = [frames] + [fr.alloc_frames(B, H, W, masks=with_masks) for _ in range(n_slots - 1)]

    def batch_views(i):
        return [specs[(i * B + k) % len(specs)] for k in range(B)]

    # dynamic sequence (BASELINE.json configs[4]): the whole trajectory exists before rendering starts (the reference
    # simulates first, /root/reference/pegasus.py:216 then :247); time step s moves object k rigidly about its centre
    pose_seq = m2w_seq = None
    if args.dynamic:
        from scipy.spatial.transform import Rotation as Rot
        from pegasus_amd.compose import pose_table
        from pegasus_amd import bop_pose
        oid = cloud.object_id
        centers = [act["means3d"][oid == k].mean(0) for k in range(1, fr.K + 1)]
        S = (2 * (args.warmup + args.steps) + max(1, args.profile_steps) + 1) * B
        pose_seq = np.zeros((S, fr.K, 20), np.float32)
        m2w_seq = []
        for s_i in range(S):
            pairs, m2w = [], {}
            for k in range(fr.K):
                T = np.eye(4)
                # bounded, periodic motion: the objects spin, rock, slide within a few centimetres and bounce -- the
                # workload stays the same however long the sequence runs
                T[:3, :3] = Rot.from_euler("zx", [0.015 * s_i * (1 + 0.1 * k), 0.3 * math.sin(0.013 * s_i + k)]).as_matrix()
                T[:3, 3] = [0.04 * math.sin(0.02 * s_i + k), 0.04 * math.cos(0.017 * s_i + 2 * k),
                            0.02 * abs(math.sin(0.05 * s_i + k))]
                pairs.append((T, centers[k]))
                C4 = np.eye(4); C4[:3, 3] = centers[k]
                Ci = np.eye(4); Ci[:3, 3] = -centers[k]
                m2w[k + 1] = C4 @ T @ Ci            # placement of the (already merged) object at time s
            pose_seq[s_i] = pose_table(pairs)
            m2w_seq.append(m2w)

    def batch_poses(i):
        return None if pose_seq is None else pose_seq[(i * B) % len(pose_seq):(i * B) % len(pose_seq) + B]

    def batch_records(i):
        if pose_seq is None:
            return None
        s0 = (i * B) % len(pose_seq)
        vs = [my_views[(i * B + k) % len(my_views)] for k in range(B)]
        return bop_pose.batch_pose_records(vs, m2w_seq[s0:s0 + B])

    def step(i, **kw):
        if args.separate_semantic:
            return fr.render_batch(batch_views(i), frames, masks=with_masks, **kw)
        kw.pop("sem_stage_ms", None)
        if pose_seq is not None and kw.get("stage_ms") is None:   # (the profiling entry point has no posed variant:
            kw.pop("stage_ms", None)                              #  stage times are taken on the unposed scene)
            return fr.render_frames(batch_views(i), frames, masks=with_masks, poses=batch_poses(i), **kw)
        return fr.render_frames(batch_views(i), frames, masks=with_masks, **kw)

    gather_state = {"inflight": None, "bytes": 0}

    def gather_finished(fr_set):
        """Quantise a finished batch on the GPU and start its gather to rank 0 (grouped send/recv on RCCL: the peers
        stream over their own xGMI links); the previous batch's gather is completed first, so one is in flight."""
        if not (args.gather and world > 1):
            return
        from pegasus_amd import masks as M
        from pegasus_amd import view_shard as VS
        if gather_state["inflight"] is not None:
            finish, works = gather_state["inflight"]
            for w_ in works:
                w_.wait()
            finish()
        q = [M.quantize_frame(fr_set["color"][k], fr_set["depth"][k, 0]) for k in range(B)]
        local = {"rgb": torch.stack([a for a, _ in q]), "depth_mm": torch.stack([b for _, b in q])}
        if "masks" in fr_set:
            local["masks"] = fr_set["masks"][:B]
        gather_state["bytes"] = sum(t.numel() * t.element_size() for t in local.values())
        gather_state["inflight"] = VS.gather_frames(local, B * world, dst=0, async_op=True)

    def run_steps(first, count):
        """`count` steps as an `n_slots`-deep software pipeline: batch i is enqueued (scene pass and semantic pass on two
        streams, no host sync) while batch i-1 finishes; every batch's overflow status is checked."""
        if args.sync_steps:
            for i in range(first, first + count):
                gather_finished(step(i))
            return
        pending = []                                  # at most n_slots - 1 older batches in flight
        for i in range(first, first + count):
            render = fr.render_batch_async if args.separate_semantic else fr.render_frames_async
            if pose_seq is not None:
                h = fr.render_frames_async(batch_views(i), frame_sets[i % n_slots], masks=with_masks,
                                           slot=i % n_slots, poses=batch_poses(i))
                batch_records(i)                  # BOP scene_gt / scene_camera entries of the batch (host, overlapped)
            else:
                t_e = time.perf_counter()
                h = render(batch_views(i), frame_sets[i % n_slots], masks=with_masks, slot=i % n_slots)
            t_w = time.perf_counter()
            pending.append(h)
            while len(pending) >= n_slots:
                gather_finished(pending.pop(0).wait())
            if args.step_log and pose_seq is None:
                print(f"step {i}: enqueue {(t_w - t_e) * 1e3:.2f} ms, wait {(time.perf_counter() - t_w) * 1e3:.2f} ms", file=sys.stderr)
        while pending:
            gather_finished(pending.pop(0).wait())
        if gather_state["inflight"] is not None:
            finish, works = gather_state["inflight"]
            for w_ in works:
                w_.wait()
            finish()
            gather_state["inflight"] = None

    # set-up, not warm-up: let the workspaces reach their size.  The instance capacity is a hint that grows when a batch
    # comes close to it (a multi-GB reallocation, tens of ms, once per scene and pipeline slot); small scenes start below
    # their need (C2: 1 M for views that list 1.4 M) and would otherwise pay that inside the timed region.
    for _ in range(3):
        before = dict(rasterizer._WS.capacity_hint)
        run_steps(0, max(2, n_slots))                 # every slot: its stream, workspace and frame set exist after this
        torch.cuda.synchronize()
        if dict(rasterizer._WS.capacity_hint) == before:
            break
    run_steps(0, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- per-view statistics and per-stage HIP-event timing of whole batches (outside the timed region) ----
    stage_ms = np.zeros((0, _lib.PGR_NUM_STAGES))
    sem_ms = np.zeros((0, _lib.PGR_NUM_STAGES))
    stats = []
    raster_only_fps = None
    if rank == 0:
        for i in range(args.profile_steps if args.profile_steps > 0 else min(4, args.steps)):   # N, V, I, evaluations
            res = rasterizer.forward_views(fr.means3d, fr.opacities, batch_views(args.warmup + i), shs=fr.shs, scales=fr.scales,
                                           rotations=fr.rotations, sh_degree=3, want_radii=True, want_aux=True)
            info = rasterizer.last_forward_info()
            for k, r in enumerate(res):
                stats.append(dict(V=int((r["radii"] > 0).sum().item()), I=info["num_instances"][k],
                                  evals=int(r["n_contrib"].sum(dtype=torch.int64).item())))
            del res
        rows, srows = [], []
        prof = (range(args.warmup, args.warmup + args.steps) if args.profile_steps <= 0
                else range(max(1, args.profile_steps)))
        for i in prof:
            ms, sms = [], []
            step(i, stage_ms=ms, sem_stage_ms=sms)
            rows.append(ms)
            srows.append(sms if sms else [0.0] * _lib.PGR_NUM_STAGES)
        stage_ms, sem_ms = np.asarray(rows), np.asarray(srows)
        # R: raster-only rate (one full-scene RGB+depth forward per view), for the record next to F
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            fr.render_batch(batch_views(i), frames, masks=False)
        torch.cuda.synchronize()
        raster_only_fps = args.steps * B / (time.perf_counter() - t1)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_views = args.steps * B * world
    value = total_views / elapsed
    N = cloud.n
    V = float(np.mean([s["V"] for s in stats]))
    I = float(np.mean([s["I"] for s in stats]))
    evals = float(np.mean([s["evals"] for s in stats]))
    B_view, per_stage_bytes = algorithmic_bytes(N, V, I, P)
    mean_ms = stage_ms.mean(axis=0)          # per batch of B views
    dom = int(np.argmax(mean_ms))
    dom_name = _lib.STAGE_NAMES[dom]
    launches = 1 if dom_name == "composite" else B      # the compositor covers the whole batch in one launch
    dom_bytes = per_stage_bytes[dom_name] * B / launches
    dom_ms = float(mean_ms[dom]) / launches
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "kernel_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(dom_bytes),
        "launches_per_step": launches,
        "stage_ms_per_view": {k: round(float(m) / B, 4) for k, m in zip(_lib.STAGE_NAMES, mean_ms)},
        "whole_path": {"bytes_per_view": int(B_view), "achieved": round(B_view * value / world / 1e9, 2),
                       "frac": round(B_view * value / world / 1e9 / HBM_PEAK_GBS, 5)},
        "composite_evals_per_s": round(evals * B / (mean_ms[4] * 1e-3), 1) if mean_ms[4] > 0 else None,
        "semantic": "separate objects-only pass" if args.separate_semantic else
                    "fused: second accumulator in the scene's compositing walk (stage composite)",
        "raster_only_views_per_s": round(raster_only_fps, 2) if raster_only_fps else None,
        "N": N, "V": round(V), "I": round(I), "P": P,
    }

    # HBM-side traffic of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    # (separate passes, gfx950 unit = KiB; see profiles/README.md), committed as profiles/pmc_traffic.json
    try:
        pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        k = pmc["kernels"].get(dom_name)
        if k and pmc.get("workload") == args.workload and pmc.get("batch") == B:
            roofline["traffic"] = int(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])
            roofline["traffic_source"] = pmc["source"]
    except (OSError, ValueError, KeyError):
        pass
    # the compositor is not bound by either roofline the schema names (HBM / MFMA): report beside them how busy the unit
    # that does bound it is -- VALU issue slots from the SQ counters (profiles/pmc_busy.json, scripts/pmc_busy.py)
    try:
        busy = json.loads((ROOT / "profiles" / "pmc_busy.json").read_text())
        k = next((v for name, v in busy["kernels"].items() if dom_name == "composite" and "composite_quarter_kernel<false, false>" in name), None)
        if k:
            roofline["valu_issue"] = {"busy_frac": k["valu_busy"], "lds_busy_frac": k["lds_busy"],
                                      "kernel": "composite_quarter_kernel<false, false> (raster-only twin of the fused kernel)",
                                      "definition": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles); SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles)",
                                      "source": busy["source"]}
    except (OSError, ValueError, KeyError):
        pass

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        cores = os.cpu_count() or 1
        n_done, t_cpu = 0, 0.0
        n_env = fr.n_env
        sem_shs = fr.sem_shs.cpu().numpy() if with_masks else None
        while n_done < len(my_views) and (n_done == 0 or t_cpu + t_cpu / n_done < args.cpu_budget_s):
            v = my_views[n_done]
            t1 = time.perf_counter()
            oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=cores, want_binning=False)
            if with_masks:
                seg = oracle.forward(act["means3d"][n_env:], act["opacities"][n_env:], scales=act["scales"][n_env:],
                                     rotations=act["rotations"][n_env:], shs=sem_shs, sh_degree=0,
                                     **v.raster_kwargs(), num_threads=cores, want_binning=False)
                oracle.color_masks(seg["color"], fr.colors_np, 0.1)
            t_cpu += time.perf_counter() - t1
            n_done += 1
        try:
            cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except (OSError, StopIteration):
            cpu_model = "unknown"
        cpu = {"value": round(n_done / t_cpu, 4), "unit": "frames/s" if with_masks else "views/s", "cores": cores,
               "kind": "port", "cpu_model": cpu_model,
               "sample": f"first {n_done} frame(s) of the same scene and cameras, oracle/pgr_oracle.c with OpenMP "
                         f"({cores} threads), reference-style lists; no reference CPU rasterizer exists"}

    line = {
        "metric": (f"rendered views/sec (RGB+depth+mask) on {N / 1e6:.2g}M-Gaussian scene @{W}x{H}" if with_masks else
                   f"rendered views/sec (RGB+depth, raster only) on {N / 1e6:.2g}M-Gaussian scene @{W}x{H}"),
        "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": label, "gaussians": N, "width": W, "height": H, "views_per_step": B,
                   "distinct_views": len(my_views), "objects": fr.K,
                   "sequence": ("dynamic: every frame is a time step with its own object poses (posed inside the "
                                "preprocess) + BOP pose records" if args.dynamic else "static scene, camera batches"),
                   "scene_layout": ("input order" if fr.order is None else
                                    "Morton order per object (one-time, at scene load, outside the timed region)"),
                   "outputs": ("color[3,H,W] f32 + depth[1,H,W] f32 + semantic image[3,H,W] f32 + masks[K,H,W] u8"
                               if with_masks else "color[3,H,W] f32 + depth[1,H,W] f32"),
                   "parallelism": f"view-shard x{world}",
                   "gather": (f"every batch's quantised frames gathered to rank 0 inside the timed region "
                              f"({gather_state['bytes'] / 1e6:.0f} MB per rank and batch)" if args.gather and world > 1 else
                              "off (frames stay on the rank that rendered them)")},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
