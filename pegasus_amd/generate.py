"""End-to-end example of the batch path: a synthetic merged scene -> frames -> BOP-layout dataset on disk.

    python -m pegasus_amd.generate --out /tmp/pegasus_ds --frames 64 [--workload c3 --scale 0.1] [--dynamic]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m pegasus_amd.generate --out ... --frames 4096

Under torchrun the frames shard over the ranks (frame f -> rank f mod world; scene replicated, one GPU
per rank) and EVERY rank is a writer: it renders its frames, encodes and writes them under their global frame numbers, and
only the pose records travel (a gather of Python objects to rank 0, which writes the two JSON files).  No frame crosses a
link: the PNG encoders of all ranks run in parallel, which is what bounds a real run.

What PEGASUS's generate_dataset loop does per frame (/root/reference/pegasus.py:247-395: merge, 3+K renders, numpy
masks, PNG + JSON writers), in batches of cameras on the GPU; PNG encoding is the only per-frame host work and runs on
a thread pool beside the renderer (the reference starts a writer thread per frame, pegasus.py:346-358)."""
from __future__ import annotations

import argparse
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5"])
    ap.add_argument("--scale", type=float, default=0.1, help="fraction of the workload's Gaussian counts")
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--dynamic", action="store_true", help="every frame a time step with its own object poses")
    ap.add_argument("--writers", type=int, default=None, help="PNG encoder threads (default: host cores, at most 32)")
    args = ap.parse_args()

    import os
    import torch
    from . import bop_pose, scenes
    from .dataset_writer import BopSceneWriter
    from .frames import FrameRenderer
    if not torch.cuda.is_available():
        raise SystemExit("pegasus_amd.generate needs a HIP device")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()}"
    torch.cuda.set_device(device)            # every default-device allocation and the synchronize() below: THIS rank's GPU
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")          # only Python objects (pose records) travel: no frame crosses a link
    maker = dict(c2=lambda: scenes.scene_c2(n=int(150_000 * args.scale), n_views=args.frames, width=args.size, height=args.size),
                 c3=lambda: scenes.scene_c3(scale=args.scale, n_views=args.frames, width=args.size, height=args.size),
                 c5=lambda: scenes.scene_c5(scale=args.scale, n_views=args.frames, width=args.size, height=args.size))
    cloud, views = maker[args.workload]()
    act = cloud.activated()
    fr = FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                       device=device)
    centers = [act["means3d"][cloud.object_id == k].mean(0) for k in range(1, fr.K + 1)]
    # model-space bounding boxes for the scene_gt records (the reference takes the mesh's minimal oriented box from open3d;
    # here: the axis-aligned box of the object's Gaussians in its rest frame, corners in binary order z-fastest)
    boxes = {}
    for k in range(1, fr.K + 1):
        p = act["means3d"][cloud.object_id == k].astype(np.float64)
        lo, hi = p.min(0), p.max(0)
        boxes[k] = (np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]), p.mean(0))
    writers = args.writers
    if writers is None and world > 1:        # N ranks share the host's cores: their encoder pools must not add up to N x 32
        writers = max(1, min(32, (os.cpu_count() or 1) // world))
    w = BopSceneWriter(args.out, workers=writers)
    t0 = time.perf_counter()
    t_gpu = 0.0
    mine = list(range(rank, args.frames, world))          # frame f -> rank f mod world
    seq_tables = seq_motion = frame_set = None
    for b0 in range(0, len(mine), args.batch):
        ids = mine[b0:b0 + args.batch]
        vs = [views[f] for f in ids]
        poses, m2w = None, [{k + 1: np.eye(4) for k in range(fr.K)}] * len(vs)
        if args.dynamic and fr.K:
            # time step s of the recorded drop (the reference's trajectory fixture, pegasus_amd/trajectory.py)
            from . import trajectory as TJ
            if seq_tables is None:
                seq_tables, seq_motion = TJ.sequence_poses(TJ.load_fixture(), centers, 200)
            idx = [f % 200 for f in ids]
            poses, m2w = seq_tables[idx], [seq_motion[i] for i in idx]
        t1 = time.perf_counter()
        specs = [fr.view_spec(v) for v in vs]
        # records-only frames: the compositor's epilogue writes what the writers take (uint8 RGB, uint16 mm, mask bits) and
        # the semantic image; no fp32 colour / depth image, no mask plane (3.84 + 7.68 MB per frame instead of 29.4)
        H, W = int(specs[0].image_height), int(specs[0].image_width)
        if frame_set is None or frame_set["records"].shape[0] != len(vs):
            frame_set = fr.alloc_frames(len(vs), H, W, records=True, images="seg")
        frames = fr.render_frames(specs, frames=frame_set, poses=poses)
        sil = fr.render_silhouettes(specs, poses=poses) if fr.K else None          # 'seg_sil': one layered pass for all objects
        torch.cuda.synchronize(device)
        t_gpu += time.perf_counter() - t1
        gt, cam = bop_pose.batch_pose_records(vs, m2w, boxes=boxes)
        w.add_batch(frames, gt, cam, n=len(vs), silhouettes=sil, frame_ids=ids, record_shape=(H, W, fr.K))
    scene = w.close(write_json=False)
    if world > 1:
        import torch.distributed as dist
        records = [None] * world if rank == 0 else None
        dist.gather_object((w.scene_gt, w.scene_camera, w.scene_gt_info), records, dst=0)
        if rank == 0:
            w.merge_records(records[1:])
    if rank == 0:
        w.write_records()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"{args.frames} frames ({cloud.n} Gaussians, {fr.K} objects) on {world} rank(s) -> {scene}: rank 0 rendered "
              f"{w.n_frames} of them in {t_gpu * 1e3:.1f} ms, total {dt:.2f} s ({args.frames / dt:.1f} frames/s with PNG "
              f"encoding on {w.workers} host thread(s) per rank)")


if __name__ == "__main__":
    main()
