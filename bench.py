#!/usr/bin/env python3
"""bench.py -- rendered views/sec (RGB + depth + mask) on the 2 M-Gaussian merged scene @800x800.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.

* N = 1: this process renders.
* N > 1 and no torchrun environment: this process is only a LAUNCHER.  Before anything touches a GPU it starts
  ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`` as a
  CHILD process and exits with its code (it never re-execs itself), after checking that N HIP devices are visible.
* N > 1 under an outer torchrun (WORLD_SIZE set; how the driver launches it): one rank per GPU over RCCL; WORLD_SIZE must
  equal --gpus or the run refuses to print a mislabelled line.

One "step" = one batch of B = 32 cameras of the synthetic merged scene (BASELINE.json configs[2]: environment + 8 objects,
2.0 M Gaussians, 800x800) through the rasterizer hot path.  Views shard across ranks (view v -> rank v mod N, scene
replicated: SURVEY.md section 8e), weak scaling: every rank renders K batches of its own shard.  For N > 1 the path's ONE
collective -- the gather of every finished batch to rank 0, in the reference's on-disk precision -- runs inside the timed
region by default (asynchronous, one gather in flight beside the next batch); the render-only rate is timed in a second
K-step pass and reported beside it.  Rank 0 prints ONE JSON line.

The JSON line also carries
  roofline      -- the dominant kernel's algorithmic bytes / its average duration measured live with HIP events on the
                   launch stream (pgr_forward_batch_profiled), against 8 TB/s HBM peak; `valu` prices the compositor's
                   pixel-Gaussian evaluations against the chip's VALU lane rate
  cpu_baseline  -- the CPU oracle ("port": there is no reference CPU rasterizer) timed on this box's host cores on a
                   bounded sample of the same workload (rank 0, N = 1 only)
  drop_in       -- PEGASUS's own per-camera loop (one render() per view through the reference's four wrappers) on a
                   bounded sample (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_LANE_OPS = 78.6e12   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (SURVEY.md section 8d "secondary ceiling")
# Vector-issue ceiling of a kernel (round 6; rounds 3-5 divided by a flat 4 cycles per instruction, which is the SQ counter's
# convention, not a measured cost): cycles_needed = sum over the SQ instruction classes of (instructions of the class per
# launch, counters) x (mean issue cost of the class in the kernel's ISA, measured per instruction kind with 8 waves per SIMD:
# scripts/microbench/valu_classes.hip) / 1024 SIMDs -- scripts/issue_model.py -> profiles/issue_model.json,
# profiles/r06_issue_model.txt.  frac = cycles_needed / (live kernel duration x the shader clock read DURING the run by
# pgr_clock_probe): 1.0 = the vector ports never idle.
N_SIMD = 1024
LANE_OPS_PER_EVAL = 20         # VALU lane-ops of one pixel-Gaussian evaluation (same section)
SEQUENCE_STEPS = 200           # BASELINE.json configs[4]: "Dynamic 200-step physics sequence"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink Gaussian counts (debug only; INVALID as a result)")
    ap.add_argument("--views", type=int, default=512, help="distinct cameras cycled through (configs[2]: 512 views)")
    ap.add_argument("--camera-set", default="fibonacci", choices=["fibonacci", "fibonacci_above_9deg"],
                    help="c3 / c5 cameras: fibonacci (default) = the first --views directions of the BOP toolkit's Fibonacci upper "
                         "hemisphere, elevation 0..90 degrees, as SURVEY.md section 8d names them; fibonacci_above_9deg = the "
                         "subset rounds 1-4 rendered (directions below ~8.6 degrees skipped and back-filled), for A/B only")
    ap.add_argument("--width", type=int, default=800, help="image width (c3 / c5; 800 = the baseline config)")
    ap.add_argument("--height", type=int, default=800, help="image height (c3 / c5; 800 = the baseline config)")
    ap.add_argument("--objects", type=int, default=0,
                    help="objects of the merged scene (c3 / c5; 0 = the config's own 8 / 20).  --width 640 --height 480 --objects 6 "
                         "--data-points all is the reference's own default run (/root/reference/pegasus.py:486-503): reported as a "
                         "side workload, never as the headline")
    ap.add_argument("--batch", type=int, default=32,
                    help="views per step (one batch call); 16 default steps x 32 = the 512 views of configs[2], each once")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the per-camera render() sample (drop_in)")
    ap.add_argument("--facade", action="store_true",
                    help="only the drop-in path: frames/s of PEGASUS's per-camera loop through the four render wrappers "
                         "and the latency of one render() call, on a larger sample")
    ap.add_argument("--separate-semantic", action="store_true",
                    help="semantic image by a second full pass over the objects (default: fused into the scene pass)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking render_batch per step (no pipelining)")
    ap.add_argument("--serialize-slots", action="store_true",
                    help="A/B: make the pipeline slots execute one after the other on the GPU (round-1 behaviour)")
    ap.add_argument("--input-order", action="store_true",
                    help="keep the scene in its input order (default: one-time Morton layout per object, outside the timed region)")
    ap.add_argument("--dynamic", action="store_true",
                    help="dynamic sequence (BASELINE.json configs[4]): every frame is a TIME STEP of the recorded drop "
                         "(tests/golden/simulation_steps_body1_first200.npz, the reference's simulation_steps.json body 1) "
                         "with its own object poses, posed inside the preprocess, one camera per step, + BOP pose records")
    ap.add_argument("--raster-only", action="store_true", help="time only the full-scene RGB+depth pass (R), no masks")
    ap.add_argument("--slots", type=int, default=3, help="batches in flight (pipeline slots: workspaces, frame sets)")
    ap.add_argument("--streams", type=int, default=0, help="streams the slots share round-robin (0 = one per slot)")
    ap.add_argument("--step-log", action="store_true", help="per-step (enqueue, wait) host milliseconds on stderr")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="N > 1 (default there): gather every batch's finished frames to rank 0 inside the timed region, in "
                         "the reference's on-disk precision (uint8 RGB, uint16 depth, one mask byte per pixel; "
                         "pegasus.py:347,355) -- the one RCCL exchange of the path (SURVEY.md section 8e)")
    ap.add_argument("--no-gather", dest="gather", action="store_false", help="N > 1: frames stay on the rank that rendered them")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl = RCCL (the measured configuration).  gloo stages the gathered "
                         "frames through host memory: rehearsals and CPU tests only, flagged in the JSON line")
    ap.add_argument("--share-devices", action="store_true",
                    help="REHEARSAL: allow more ranks than HIP devices (rank r uses device r mod count; needs --backend gloo, "
                         "RCCL refuses two ranks on one GPU); the JSON line is flagged and is not a result")
    ap.add_argument("--force-dist", action="store_true",
                    help="REHEARSAL: initialise the process group and run the gather path even with ONE rank (under torchrun "
                         "--nproc-per-node 1): RCCL initialisation, the gather of uint8 / 16-bit-as-bytes frames and the "
                         "collectives of the timing protocol on the real backend, without a second GPU")
    ap.add_argument("--stub-renderer", action="store_true",
                    help="TEST ONLY: exercise launch / sharding / gather / timing with a deterministic CPU frame source "
                         "(no rasterizer, gloo); the JSON line is flagged invalid")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--cpu-only", action="store_true",
                    help="BASELINE.json configs[0] (plumbing, no GPU): time ONLY the CPU oracle on the chosen workload's cameras "
                         "(default there: --workload c1, the 10 k-Gaussian cube at 256x256) and print a line flagged cpu_only; "
                         "the product path is not involved and needs no HIP device")
    ap.add_argument("--full-outputs", action="store_true",
                    help="N > 1 with --records epilogue: keep writing the fp32 images and mask planes beside the records "
                         "(rounds 4-5; default since round 6: records-only frame sets, PgrOutputs color = NULL)")
    ap.add_argument("--records", default="epilogue", choices=["epilogue", "pack"],
                    help="N > 1: where the gathered frame records come from -- epilogue (default on RCCL): the compositor writes "
                         "them straight into the gather's send buffers (PgrOutputs::record); pack: pgr_pack_records after the "
                         "batch (one-rank RCCL A/B on one box: 5751 / 5717 vs 5617 / 5606 frames/s with the gather)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed repeats of the K steps inside this one invocation (BASELINE.md section 4: >= 5 repeats, "
                         "median + min): value / ms_per_step are the MEDIAN repeat, min / max / every repeat ride along")
    ap.add_argument("--data-points", default="default", choices=["default", "all"],
                    help="default = ['rgb','depth','seg_vis'] (+ the semantic image the masks come from); all adds 'seg_sil', "
                         "the K per-object silhouette masks (/root/reference/pegasus.py:491, src/gs/render.py:36-65), as one "
                         "layered batch pass per step inside the timed region")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="batches measured per-stage with HIP events (0 = the same batches as the timed steps)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------------------------
# launcher

def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices() -> int:
    """HIP devices this process could use.  torch.cuda.device_count() does not initialise the GPU on this image, so the
    launcher may call it and still spawn children afterwards."""
    import torch
    return int(torch.cuda.device_count())


def self_launch(args, argv) -> int:
    """``python bench.py --gpus N`` without a torchrun environment: start the N ranks as a child process tree."""
    n = int(args.gpus)
    if not args.stub_renderer:
        have = visible_devices()
        if have < n and not args.share_devices:
            print(f"bench.py --gpus {n}: only {have} HIP device(s) visible on this node; refusing to print a line that is "
                  f"not an {n}-GPU measurement (rehearsal on fewer devices: --share-devices --backend gloo)", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), *argv]
    env = dict(os.environ)
    env["PGR_BENCH_LAUNCHER"] = "self"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    print("bench.py launcher:", " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------------------------------------------------
# workload

CAMERA_SET_TEXT = {
    "fibonacci": "first {n} directions of the BOP toolkit's Fibonacci upper hemisphere (sample_views mode='fibonacci', elevation "
                 "0..90 degrees, grazing views included; golden: tests/golden/bop_fibonacci_views.npz), radius cycling 0.8..1.2 m, "
                 "looking at the scene centre",
    "fibonacci_above_9deg": "NOT the stated camera set: Fibonacci hemisphere directions with eye_dir.z > 0.15 (elevation above "
                            "~8.6 degrees) only, the skipped ones back-filled by repeating directions at other radii (what rounds "
                            "1-4 rendered; A/B only)",
}


def build_workload(name, scale, n_views, with_poses=False, camera_set="fibonacci", width=800, height=800, objects=0):
    from pegasus_amd import scenes
    rest = None
    if name == "c1":
        cloud, views = scenes.scene_c1()
        label = "C1 10k-Gaussian cube, 256x256"
    elif name == "c2":
        cloud, views = scenes.scene_c2(n=int(150_000 * scale), n_views=n_views)
        label = "C2 single object 150k Gaussians, 800x800 hemisphere views"
    elif name == "c5":
        cloud, views, rest = scenes.merged_scene(5, int(3_400_000 * scale), objects or 20, int(80_000 * scale), n_views,
                                                 width, height, camera_set=camera_set)
        label = f"C5 5M-Gaussian scene (3.4M environment + {objects or 20} objects), {width}x{height}"
    else:
        cloud, views, rest = scenes.merged_scene(3, int(1_360_000 * scale), objects or 8, int(80_000 * scale), n_views,
                                                 width, height, camera_set=camera_set)
        label = f"C3 merged env + {objects or 8} objects, {cloud.n / 1e6:.3g}M Gaussians, {width}x{height}"
    if scale != 1.0:
        label += f" [scale={scale}: NOT the baseline config]"
    if name in ("c3", "c5") and ((width, height) != (800, 800) or objects not in (0, 8 if name == "c3" else 20)):
        label += " [image size / object count differ: NOT the baseline config]"
    if with_poses:
        return cloud, views, label, rest
    return cloud, views, label


def algorithmic_bytes(N, V, I, P):
    """SURVEY.md section 8d / BASELINE.md section 3:  B = 16 N + 272 V + 88 I + 16 P  bytes per view, and the
    per-stage split it is the sum of."""
    per_stage = {
        "preprocess": 12 * N + 224 * V + 4 * N + 48 * V,
        "bin_count": 0,
        "bin_scatter": 12 * I,
        "tile_sort": 24 * I + 8 * I,
        "composite": 44 * I + 16 * P,      # (the fused semantic image adds 16 P of writes: reported beside it)
    }
    return 16 * N + 272 * V + 88 * I + 16 * P, per_stage


class RealEngine:
    """The product path: FrameRenderer over the resident scene, `n_slots` batches in flight."""
    stub = False

    def __init__(self, args, rank, world, dev):
        import torch
        from pegasus_amd import _lib, frames as F
        _lib.lib()
        self.args, self.rank, self.world, self.dev, self.torch = args, rank, world, dev, torch
        B = self.B = max(1, args.batch)
        n_views_total = max(args.views, B) * world
        cloud, views, label, rest = build_workload(args.workload, args.scale, n_views_total, with_poses=True,
                                                   camera_set=args.camera_set, width=args.width, height=args.height,
                                                   objects=args.objects)
        self.cloud, self.views, self.label = cloud, views, label
        self.act = act = cloud.activated()
        fr = self.fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"],
                                       cloud.object_id, sh_degree=3, device=dev, spatial_order=not args.input_order)
        fr.serialize_slots = args.serialize_slots
        fr.n_streams = args.streams or None
        self.specs = [fr.view_spec(v) for v in views]
        self.W, self.H = views[0].width, views[0].height
        self.with_masks = not args.raster_only and fr.K > 0
        self.n_slots = max(1, args.slots)
        self.frame_sets = [fr.alloc_frames(B, self.H, self.W, masks=self.with_masks) for _ in range(self.n_slots)]
        self.frames = self.frame_sets[0]
        self.with_sil = args.data_points == "all" and self.with_masks
        if self.with_sil:
            for f in self.frame_sets:
                f["sil"] = torch.empty((B, fr.K, self.H, self.W), dtype=torch.uint8, device=dev)
        # dynamic sequence: the whole trajectory exists before rendering starts (the reference simulates first,
        # /root/reference/pegasus.py:216 then :247); time step s -> rank s mod world, poses composed absolutely per step
        self.pose_seq = self.m2w_seq = None
        if args.dynamic:
            if fr.K == 0 or rest is None:
                raise SystemExit("--dynamic needs a scene with objects (c3, c5)")
            from pegasus_amd import trajectory as TJ
            oid = cloud.object_id
            centers = [act["means3d"][oid == k].astype(np.float64).mean(0) for k in range(1, fr.K + 1)]
            self.pose_seq, motions = TJ.sequence_poses(TJ.load_fixture(), centers, SEQUENCE_STEPS)
            rest_m2w = []
            for Rm, t in rest:
                M = np.eye(4); M[:3, :3] = Rm; M[:3, 3] = t
                rest_m2w.append(M)
            self.m2w_seq = [{k: mot[k] @ rest_m2w[k - 1] for k in mot} for mot in motions]

    # frame j of this rank is global frame j * world + rank (view v -> rank v mod world; step s -> rank s mod world)
    def frame_ids(self, i):
        return [((i * self.B + k) * self.world + self.rank) for k in range(self.B)]

    def batch_views(self, i):
        return [self.specs[g % len(self.specs)] for g in self.frame_ids(i)]

    def batch_poses(self, i):
        return None if self.pose_seq is None else self.pose_seq[[g % SEQUENCE_STEPS for g in self.frame_ids(i)]]

    def batch_records(self, i):
        if self.pose_seq is None:
            return None
        from pegasus_amd import bop_pose
        ids = self.frame_ids(i)
        return bop_pose.batch_pose_records([self.views[g % len(self.views)] for g in ids],
                                           [self.m2w_seq[g % SEQUENCE_STEPS] for g in ids])

    def enqueue(self, i, slot, records=None):
        """``records``: a uint8 [B, record_bytes] tensor (a FrameGather send buffer) the compositor's epilogue fills with the
        batch's frame records -- what leaves the GPU, written while the frames are made."""
        fr, a = self.fr, self.args
        if a.separate_semantic:
            return fr.render_batch_async(self.batch_views(i), self.frame_sets[slot], masks=self.with_masks, slot=slot)
        h = fr.render_frames_async(self.batch_views(i), self.frame_sets[slot], masks=self.with_masks, slot=slot,
                                   poses=self.batch_poses(i), records=records)
        if self.with_sil:                     # 'seg_sil': all K silhouettes of the batch, one layered pass on its own stream
            _, hs = fr.render_silhouettes(self.batch_views(i), out=self.frame_sets[slot]["sil"], poses=self.batch_poses(i),
                                          slot=slot, wait=False)
            frames_h = h

            class _Both:
                def wait(_self):
                    f = frames_h.wait()
                    hs.wait()
                    return f
            h = _Both()
        self.batch_records(i)                 # BOP scene_gt / scene_camera entries of the batch (host, overlapped)
        return h

    def step_blocking(self, i, **kw):
        fr, a = self.fr, self.args
        if a.separate_semantic:
            return fr.render_batch(self.batch_views(i), self.frames, masks=self.with_masks, **kw)
        kw.pop("sem_stage_ms", None)
        poses = self.batch_poses(i)
        if self.with_sil and kw.get("stage_ms") is None:
            fr.render_silhouettes(self.batch_views(i), out=self.frames["sil"], poses=poses)
        if poses is not None and kw.get("stage_ms") is None:      # (the profiling entry point has no posed variant:
            kw.pop("stage_ms", None)                              #  stage times are taken on the unposed scene)
            return fr.render_frames(self.batch_views(i), self.frames, masks=self.with_masks, poses=poses, **kw)
        return fr.render_frames(self.batch_views(i), self.frames, masks=self.with_masks, **kw)

    def record_bytes(self):
        from pegasus_amd import masks as M
        return M.record_layout(self.H, self.W, self.fr.K if self.with_masks else 0)["bytes"]

    def set_records_only(self):
        """N > 1 with direct records: a rank's only product is the frame record in the gather's send buffer, so its frame
        sets name no image (PgrOutputs color = depth = sem_* = NULL): the compositor writes 3.84 MB per 800x800 frame, not
        29.4 MB of fp32 / mask planes that nothing reads."""
        self.frames = self.frame_sets[0]           # (one full set stays for the untimed side passes: stage profile, raster-only rate)
        self.frame_sets = [dict() for _ in range(self.n_slots)]
        self.records_only = True

    def full_frame_set(self):
        return self.fr.alloc_frames(self.B, self.H, self.W, masks=self.with_masks)

    def pack(self, fr_set, out):
        """What leaves the GPU for a finished batch: ONE uint8 record per frame, written straight into `out` (a
        preallocated [B, record_bytes] device tensor -- the frame set is re-rendered while the records travel)."""
        from pegasus_amd import masks as M
        return M.pack_records(color=fr_set["color"][:self.B], depth=fr_set["depth"][:self.B],
                              masks=fr_set["masks"][:self.B] if "masks" in fr_set else None, out=out)

    def sync(self):
        self.torch.cuda.synchronize()

    def settle(self, run_steps):
        """Set-up, not warm-up: let the workspaces reach their size.  The instance capacity is a hint that grows when a
        batch comes close to it (a multi-GB reallocation, tens of ms, once per scene and pipeline slot); small scenes start
        below their need and would otherwise pay that inside the timed region."""
        from pegasus_amd import rasterizer
        # over the SAME batches the warm-up and the timed region run: a later batch with more instances than the first few
        # (other cameras, other time steps of a dynamic sequence) would otherwise grow the workspace inside a timed repeat
        # (round 4: one of five repeats of the dynamic runs took 8.3 instead of 5.4 ms per step)
        n = max(2, self.n_slots, self.args.warmup + self.args.steps)
        for _ in range(3):
            before = rasterizer.capacity_hints()
            run_steps(0, n, False)                        # every slot: its stream, workspace and frame set exist after this
            self.sync()
            if rasterizer.capacity_hints() == before:
                break


class ClockProbe:
    """The shader clock WHILE a measured run is resident: one wave on a side stream reads its s_memtime (shader cycles) and
    s_memrealtime (100 MHz) counters around a sleep loop (pgr_clock_probe); two stamps on the MEASURED stream, before and
    after the run, tell whether the probe really ran beside it -- a stream that happens to share its hardware queue with
    the measured one starts its kernel only when that queue drains, and then reads an idle chip's clock."""

    def __init__(self, dev):
        import torch
        self.torch, self.dev = torch, dev
        self.ticks = torch.zeros((3, 4), dtype=torch.int64, device=dev)      # rows: probe, stamp before, stamp after
        self.stream = None

    def _launch(self, row, spin_us, stream):
        import ctypes as C
        from pegasus_amd import _lib
        return _lib.lib().pgr_clock_probe(C.c_void_p(self.ticks[row].data_ptr()), int(spin_us), C.c_void_p(stream.cuda_stream))

    def start(self, spin_us):
        cur = self.torch.cuda.current_stream(self.dev)
        self.stream = self.torch.cuda.Stream(self.dev)                         # a new stream per attempt: another queue
        self.ticks.zero_()
        self.stream.wait_stream(cur)
        self._launch(1, 0, cur)                                                # stamp: the measured stream gets here
        self._launch(0, spin_us, self.stream)

    def stop(self):
        """-> (MHz, fraction of the measured run the probe was resident for) or (None, 0.0)"""
        if self.stream is None:
            return None, 0.0
        self._launch(2, 0, self.torch.cuda.current_stream(self.dev))
        self.torch.cuda.synchronize(self.dev)
        self.stream = None
        (c, r, p0, p1), (_, _, m0, _), (_, _, _, m1) = (tuple(int(v) for v in row) for row in self.ticks.tolist())
        if r <= 0 or m1 <= m0:
            return None, 0.0
        overlap = max(0, min(p1, m1) - max(p0, m0)) / float(m1 - m0)
        return round(c / r * 100.0, 1), round(overlap, 3)


class StubEngine:
    """TEST ONLY (--stub-renderer): a deterministic CPU frame source with the engine's interface, so that the launcher,
    the sharding, the asynchronous gather, its check and the timing protocol run on 2 gloo ranks without a GPU."""
    stub = True

    def __init__(self, args, rank, world, dev):
        import torch
        self.args, self.rank, self.world, self.torch = args, rank, world, torch
        self.B = max(1, args.batch)
        self.n_slots = max(1, args.slots)
        self.label = "STUB frame source (no rasterizer): launch-path test only"
        self.W, self.H, self.with_masks, self.pose_seq = 5, 4, True, None

    def frame_ids(self, i):
        return [((i * self.B + k) * self.world + self.rank) for k in range(self.B)]

    @staticmethod
    def frames_of(ids, torch):
        g = torch.tensor(ids, dtype=torch.int64)
        rgb = (g.view(-1, 1, 1, 1) % 251 + torch.arange(3).view(1, 1, 1, 3)).to(torch.uint8).expand(-1, 4, 5, 3).contiguous()
        depth = ((g * 7) % 30000).to(torch.int16).view(-1, 1, 1).expand(-1, 4, 5).contiguous()
        bits = (g % 256).to(torch.uint8).view(-1, 1, 1, 1).expand(-1, 4, 5, 1).contiguous()
        return {"rgb": rgb, "depth_mm": depth, "mask_bits": bits}

    def enqueue(self, i, slot, records=None):
        ids = self.frame_ids(i)

        class _H:
            def wait(_self):
                return ids
        return _H()

    def step_blocking(self, i, **kw):
        return self.frame_ids(i)

    def record_bytes(self):
        from pegasus_amd import masks as M
        return M.record_layout(self.H, self.W, 8)["bytes"]

    @classmethod
    def records_of(cls, ids, torch, H=4, W=5):
        """The stub's frames as records in the product's layout (host-only layout query of the library)."""
        from pegasus_amd import masks as M
        lay = M.record_layout(H, W, 8)
        f = cls.frames_of(ids, torch)
        rec = torch.zeros((len(ids), lay["bytes"]), dtype=torch.uint8)
        rec[:, :3 * H * W] = f["rgb"].reshape(len(ids), -1)
        rec[:, lay["off_depth"]:lay["off_depth"] + 2 * H * W] = f["depth_mm"].reshape(len(ids), -1).view(torch.uint8)
        rec[:, lay["off_masks"]:lay["off_masks"] + H * W] = f["mask_bits"].reshape(len(ids), -1)
        return rec

    def pack(self, ids, out):
        out.copy_(self.records_of(ids, self.torch, self.H, self.W))
        return out

    def sync(self):
        pass

    def settle(self, run_steps):
        run_steps(0, 1, False)


# --------------------------------------------------------------------------------------------------------------------
# worker

def _checksums(local):
    """Per-tensor byte sums (int64) of a packed batch -- what the gather check compares across ranks."""
    import torch
    return {k: int(t.contiguous().view(torch.uint8).to(torch.int64).sum().item()) for k, t in local.items()}


_REAL_STDOUT = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Libraries under us write there too -- RCCL prints a version banner at
    communicator creation, gloo its "Rank 0 is connected to ..." lines (both seen in this round's rehearsals, straight to
    file descriptor 1, unordered against Python's buffered stream) -- so for the rest of the process descriptor 1 IS stderr,
    and the line goes to a private duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line: dict):
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def run_worker(args):
    import torch
    import torch.distributed as dist
    protect_stdout()

    env_world = os.environ.get("WORLD_SIZE")
    world = int(env_world) if env_world else 1
    if args.gpus is None:
        args.gpus = world
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}; refusing to print a mislabelled line")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "gloo" if args.stub_renderer else args.backend
    use_dist = world > 1 or args.force_dist
    rehearsal = args.stub_renderer or args.share_devices or args.force_dist or (world > 1 and backend != "nccl")
    dev = torch.device("cpu")
    if not args.stub_renderer:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev and not args.share_devices:
            raise SystemExit(f"bench.py: rank {rank} has no device of its own ({n_dev} visible, local rank {local_rank})")
        if args.share_devices and backend == "nccl" and world > n_dev:
            raise SystemExit("--share-devices needs --backend gloo (RCCL refuses two ranks on one GPU)")
        torch.cuda.set_device(local_rank % n_dev)
        dev = torch.device("cuda", local_rank % n_dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()              # the size actually initialised is what the line reports
    devices = [None]
    me = {"rank": rank, "device": str(dev)}
    if dev.type == "cuda":
        p = torch.cuda.get_device_properties(dev)
        me.update(name=p.name, total_memory_gb=round(p.total_memory / 2**30, 1), pci_bus_id=getattr(p, "pci_bus_id", None))
    if use_dist:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    else:
        devices = [me]

    gather_on = use_dist and (args.gather if args.gather is not None else True)
    eng = (StubEngine if args.stub_renderer else RealEngine)(args, rank, world, dev)
    if args.facade:
        if world != 1 or eng.stub:
            raise SystemExit("--facade measures the single-process drop-in path: run it with --gpus 1")
        d = drop_in_numbers(eng, n_frames=16, n_render_calls=128)
        d_dyn = drop_in_numbers(eng, n_frames=16, n_render_calls=8, dynamic=True)
        d["dynamic"] = {k: d_dyn[k] for k in ("mode", "frames_per_s", "ms_per_frame", "frames_per_s_all_data_points", "ms_per_part",
                                                "passes_ms_per_frame")}
        mode = "DYNAMIC scene: every object re-posed between frames through the reference's three pose calls, the objects-only " \
               "semantic scene rebuilt every frame" if args.dynamic else \
               "static scene: the objects-only semantic scene and its render are shared by the two semantic wrappers and kept " \
               "while no object moves"
        emit({"metric": "drop-in frames/sec through PEGASUS's per-camera loop (RGB+depth+visible masks+semantic "
                        f"mask, one render() per data point; {mode}) on {eng.cloud.n / 1e6:.2g}M-Gaussian scene @{eng.W}x{eng.H}",
              "value": d["dynamic"]["frames_per_s"] if args.dynamic else d["frames_per_s"], "unit": "frames/s", "n_gpus": 1,
              "higher_is_better": True, "dtype": "f32", "data": "synthetic",
              "config": {"workload": eng.label, "objects": eng.fr.K, "camera_set": args.camera_set},
              "drop_in": d})
        return 0
    B, n_slots = eng.B, eng.n_slots
    # ---- the one exchange of the path (N > 1): per batch ONE gather of ONE uint8 record per frame, between buffers that
    # exist before the first batch (pegasus_amd/view_shard.py FrameGather).  Rank r's local frame i is global frame
    # i * world + r, so the root's rank-major receive buffer is the global order under a transposed view: nothing is
    # reordered, copied or allocated per batch.
    fg = stage = None
    gstat = {"host_s": 0.0, "wire_s": 0.0, "batches": 0}
    # DIRECT records (RCCL, the measured configuration): the send buffers of the gather ARE where the compositor's epilogue
    # writes every frame's record (PgrOutputs::record) -- no pack pass re-reads the images.  A ring of n_slots + 2 buffers:
    # batch i renders into buffer i mod R, its gather starts when the batch is done and has two more batches of time before
    # the buffer is rendered into again.
    direct = False
    ring = collections.deque()                     # ring indices of the batches in flight, oldest first
    seq = {"enq": 0}
    if gather_on:
        from pegasus_amd import view_shard as VS
        host_wire = backend == "gloo"              # gloo moves host memory: rehearsals and CPU tests only
        direct = (args.records == "epilogue" and not host_wire and not eng.stub and not args.sync_steps
                  and not args.separate_semantic)
        fg = VS.FrameGather(cap=B, record_bytes=eng.record_bytes(), device="cpu" if host_wire else dev, dst=0,
                            depth=n_slots + 2 if direct else 2, pin_memory=host_wire and not eng.stub)
        if direct and not args.full_outputs and not eng.with_sil:
            eng.set_records_only()
        if host_wire and not eng.stub:             # device-side staging of the rehearsal: pack on the GPU, copy to pinned
            stage = [torch.empty((B, eng.record_bytes()), dtype=torch.uint8, device=dev) for _ in range(2)]

    def enqueue(i, slot):
        if not direct:
            return eng.enqueue(i, slot)
        r = seq["enq"] % fg.depth
        seq["enq"] += 1
        fg.finish(r)                               # the gather that read this buffer depth batches ago (RCCL: a stream dependency)
        ring.append(r)
        return eng.enqueue(i, slot, records=fg.send_buffer(r))

    def gather_finished(token, gather):
        """Pack a finished batch into this slot's send buffer and start its gather to rank 0 (grouped send/recv on RCCL:
        the peers stream over their own xGMI links).  Two slots: one gather is in flight beside the batch being packed."""
        if direct:
            r = ring.popleft()
            if gather:
                t0 = time.perf_counter()
                fg.start(r)                        # the records are in the send buffer already: the compositor wrote them
                gstat["batches"] += 1
                gstat["host_s"] += time.perf_counter() - t0
            return
        if not gather:
            return
        t0 = time.perf_counter()
        slot = gstat["batches"] & 1
        fg.finish(slot)                            # the gather that last used this slot's buffers (two batches ago)
        t1 = time.perf_counter()
        wire = t1 - t0                             # host time spent waiting for bytes to move (RCCL: none, the wait is a stream dependency)
        if stage is not None:
            eng.pack(token, stage[slot])
            t2 = time.perf_counter()
            fg.send_buffer(slot).copy_(stage[slot], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            wire += time.perf_counter() - t2       # the rehearsal's device -> pinned host leg belongs to its wire
        else:
            eng.pack(token, fg.send_buffer(slot))
        fg.start(slot)
        gstat["batches"] += 1
        gstat["wire_s"] += wire
        gstat["host_s"] += time.perf_counter() - t0 - wire

    def finish_inflight():
        if fg is not None:
            t0 = time.perf_counter()
            fg.finish_all()
            gstat["wire_s"] += time.perf_counter() - t0

    def run_steps(first, count, gather):
        """`count` steps as an `n_slots`-deep software pipeline: batch i is enqueued (no host sync) while batch i-1
        finishes; every batch's overflow status is checked in wait()."""
        if args.sync_steps:
            for i in range(first, first + count):
                gather_finished(eng.step_blocking(i), gather)
            finish_inflight()
            return
        pending = []                                  # at most n_slots - 1 older batches in flight
        for i in range(first, first + count):
            t_e = time.perf_counter()
            h = enqueue(i, i % n_slots)
            t_w = time.perf_counter()
            pending.append(h)
            while len(pending) >= n_slots:
                gather_finished(pending.pop(0).wait(), gather)
            if args.step_log:
                print(f"step {i}: enqueue {(t_w - t_e) * 1e3:.2f} ms, wait {(time.perf_counter() - t_w) * 1e3:.2f} ms", file=sys.stderr)
        while pending:
            gather_finished(pending.pop(0).wait(), gather)
        finish_inflight()

    def timed(first, count, gather):
        """EXACTLY `count` steps bracketed by barrier + synchronize on both sides.  Returns (MAX over ranks, every
        rank's own elapsed seconds between the two barriers' inner synchronisations)."""
        eng.sync()
        if use_dist:
            dist.barrier()
        eng.sync()
        t0 = time.perf_counter()
        run_steps(first, count, gather)
        eng.sync()
        own = time.perf_counter() - t0             # this rank's own work, before it waits for the others
        if use_dist:
            dist.barrier()
        eng.sync()
        el = time.perf_counter() - t0
        per_rank = [own]
        if use_dist:
            te = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
            per_rank = [None] * world
            dist.all_gather_object(per_rank, own)
        return el, per_rank

    probe = ClockProbe(dev) if (not eng.stub and world == 1) else None
    clock = {}
    eng.probe, eng.clock = probe, clock
    eng.settle(run_steps)
    run_steps(0, args.warmup, gather_on)
    # The scene, 512+ view specs and the workspaces are thousands of long-lived Python objects; the per-batch host work of a
    # dynamic run (pose tables, BOP records: dicts of lists) allocates enough containers to trigger full collections, each of
    # which walks all of them -- one repeat in five took 8 instead of 5.4 ms per step.  Everything alive now is set-up:
    # move it out of the collector's sight.
    import gc
    gc.collect()
    gc.freeze()
    repeats = max(1, args.repeats)
    gstat.update(host_s=0.0, wire_s=0.0, batches=0)
    runs = [timed(args.warmup, args.steps, gather_on) for _ in range(repeats)]
    gather_host_s, gather_wire_s = gstat["host_s"] / repeats, gstat["wire_s"] / repeats
    order = sorted(range(repeats), key=lambda r: runs[r][0])
    med = order[(repeats - 1) // 2]                # the median repeat (lower middle for an even count)
    elapsed, per_rank = runs[med]
    elapsed_render_only = timed(args.warmup, args.steps, False)[0] if gather_on else None

    # ---- gather check (N > 1, outside the timed region): one more batch through the same pack + gather; rank 0 compares
    # the byte sums of what it RECEIVED from every rank with the sums those ranks computed on what they SENT
    gather_check = None
    if use_dist and gather_on:
        i_chk = args.warmup + args.steps
        records_ok = True
        if direct:                                    # one batch through the same path: render into send buffer 0, gather it
            fg.finish_all()
            seq["enq"] = 0
            token = enqueue(i_chk, 0).wait()
            if getattr(eng, "records_only", False):   # the same batch once more with every image, for the pack kernel to read
                token = eng.fr.render_frames(eng.batch_views(i_chk), eng.full_frame_set(), masks=eng.with_masks,
                                             poses=eng.batch_poses(i_chk))
            eng.sync()
            # the compositor's records against the pack kernel's on the same frames (sender side, every rank)
            # (section by section: neither writer touches the 16-byte alignment padding between the sections)
            from pegasus_amd import masks as M_
            k_rec = eng.fr.K if eng.with_masks else 0
            got = M_.record_views(fg.send_buffer(0), eng.H, eng.W, k_rec)
            want = M_.record_views(eng.pack(token, torch.zeros_like(fg.send_buffer(0))), eng.H, eng.W, k_rec)
            records_ok = all(bool(torch.equal(got[kk], want[kk])) for kk in want)
            gather_finished(token, True)
        else:
            token = eng.step_blocking(i_chk)
            eng.sync()
            gstat["batches"] = 0
            gather_finished(token, True)
        recv = fg.finish(0)
        eng.sync()
        sums = [None] * world
        dist.all_gather_object(sums, (int(fg.send_buffer(0).to(torch.int64).sum().item()), records_ok))
        records_ok = all(s_[1] for s_ in sums)
        sums = [s_[0] for s_ in sums]
        if rank == 0:
            ok = all(int(recv[r].to(torch.int64).sum().item()) == sums[r] for r in range(world))
            if eng.stub:                              # the stub's frames are a function of their global id: check content too,
                glob = fg.global_view(0)              # through the (rank, i) -> global id rule g = i * world + r
                ids = [((i_chk * B + k) * world + r) for k in range(B) for r in range(world)]
                want = StubEngine.records_of(ids, torch, eng.H, eng.W).view(B, world, -1)
                ok = ok and torch.equal(glob, want)
            gather_check = "ok" if (ok and records_ok) else "MISMATCH"
        flag = torch.tensor([0 if (rank != 0 or gather_check == "ok") else 1], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag)
        if int(flag.item()):
            raise SystemExit("bench.py: gathered frames differ from what the ranks sent (gather check failed)")

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return 0

    total_views = args.steps * B * world
    value = total_views / elapsed
    all_el = [r[0] for r in runs]
    line = {
        "metric": None, "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # >= 5 repeats, median + min.  value / ms_per_step are the MEDIAN repeat of `steps` steps
        "repeats": repeats, "value_min": round(total_views / max(all_el), 3), "value_max": round(total_views / min(all_el), 3),
        "ms_per_step_all": [round(e / args.steps * 1e3, 4) for e in all_el],
        # every rank's own time for the median repeat (before the closing barrier): a straggler shows here
        "per_rank_s": {"min": round(min(per_rank), 6), "max": round(max(per_rank), 6), "rank0": round(per_rank[0], 6),
                       "all": [round(x, 6) for x in per_rank]},
    }
    launcher = ("self: python bench.py --gpus N started torch.distributed.run as a child process"
                if os.environ.get("PGR_BENCH_LAUNCHER") == "self" else
                ("external torchrun (WORLD_SIZE in the environment)" if env_world else "single process"))
    dist_info = {"world_size": world, "backend": ("nccl (RCCL)" if backend == "nccl" else backend) if use_dist else None,
                 "launcher": launcher, "devices": devices}
    gather_info = {"mode": "off (frames stay on the rank that rendered them)"}
    if gather_on:
        gather_info = {
            "mode": "ONE collective per batch inside the timed region: every rank's [B, record] uint8 send buffer gathered into "
                    "rank 0's preallocated [world, B, record] buffer (rank-major = global order under a transposed view: frame "
                    "g = i * world + r); two buffer slots, one gather in flight; nothing allocated or reordered per batch",
            "payload": "one record per frame: uint8 RGB [H,W,3] | uint16 depth mm [H,W] | K masks as bit planes (ceil(K/8) bytes per pixel)",
            "records": ("written by the compositor's epilogue straight into the gather's send buffers (PgrOutputs::record; a ring of "
                        f"{fg.depth} buffers), no pack pass" + ("; RECORDS-ONLY views: no fp32 image or mask plane is written"
                                                                if getattr(eng, "records_only", False) else "")
                        if direct else "pgr_pack_records into the send buffer after the batch"),
            "bytes_per_rank_and_batch": fg.bytes_per_rank_and_batch,
            "views_per_s_with_gather": round(value, 3),
            "views_per_s_render_only": round(total_views / elapsed_render_only, 3),
            "inbound_to_root_gb_per_s": round(fg.bytes_per_rank_and_batch * (world - 1) * args.steps / elapsed / 1e9, 2),
            # rank 0's own per-batch overhead: host time to enqueue the pack kernel and start the collective (there is no
            # assembly step: nothing is reordered or copied on arrival), mean over the repeats; and, apart from it, the host
            # time spent WAITING for bytes to move -- zero on RCCL (completion is a stream dependency), the whole transfer on
            # the gloo rehearsal (device -> pinned host -> loopback TCP)
            "rank0_gather_host_ms_per_step": round(gather_host_s / args.steps * 1e3, 4),
            "rank0_gather_host_frac_of_step": round(gather_host_s / elapsed, 4),
            "rank0_wait_for_wire_ms_per_step": round(gather_wire_s / args.steps * 1e3, 4),
            "check": gather_check,
        }
    if eng.stub:
        line.update(metric="STUB launch-path test (no rasterizer) -- INVALID as a measurement", stub=True,
                    config={"workload": eng.label, "views_per_step": B, "parallelism": f"view-shard x{world}",
                            "distributed": dist_info, "gather": gather_info})
        emit(line)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    _finish_real_line(args, eng, line, value, world, dist_info, gather_info, rehearsal)
    emit(line)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _finish_real_line(args, eng, line, value, world, dist_info, gather_info, rehearsal):
    """Rank 0, product path: per-view statistics, per-stage HIP-event timing of whole batches, roofline, CPU baseline and
    the drop-in sample -- all outside the timed region."""
    import torch
    from pegasus_amd import _lib, rasterizer
    fr, B, W, H = eng.fr, eng.B, eng.W, eng.H
    P = W * H
    with_masks = eng.with_masks
    stats = []
    # N, V, I, evaluations: four batches spread EVENLY over the timed steps (the camera set is ordered by elevation: the first
    # batches alone are the grazing views)
    n_stat = args.profile_steps if args.profile_steps > 0 else min(4, args.steps)
    for i in [(j * args.steps) // n_stat for j in range(n_stat)]:
        res = rasterizer.forward_views(fr.means3d, fr.opacities, eng.batch_views(args.warmup + i), shs=fr.shs, scales=fr.scales,
                                       rotations=fr.rotations, sh_degree=3, want_radii=True, want_aux=True)
        info = rasterizer.last_forward_info()
        for k, r in enumerate(res):
            stats.append(dict(V=int((r["radii"] > 0).sum().item()), I=info["num_instances"][k],
                              evals=int(r["n_contrib"].sum(dtype=torch.int64).item())))
        del res
    prof = (range(args.warmup, args.warmup + args.steps) if args.profile_steps <= 0 else range(max(1, args.profile_steps)))
    probe, clock = getattr(eng, "probe", None), getattr(eng, "clock", {})
    # the stage-profile run, with the clock probe resident beside it (NOT in the timed region: a resident kernel holds one of
    # the process's hardware queues, and the pipeline slot whose stream shares it waits behind it -- measured: 6.1 k -> 5.3 k
    # frames/s with a 20 ms probe per repeat).  Repeated on another probe stream if the probe did not overlap the run.
    for attempt in range(4 if probe is not None else 1):
        rows, srows = [], []
        if probe is not None:
            probe.start(min(900000, int(1.3 * len(prof) * B / max(value / world, 1.0) * 1e6) + 2000))    # a little longer than the run
        for i in prof:
            ms, sms = [], []
            eng.step_blocking(i, stage_ms=ms, sem_stage_ms=sms)
            rows.append(ms)
            srows.append(sms if sms else [0.0] * _lib.PGR_NUM_STAGES)
        if probe is None:
            break
        mhz, overlap = probe.stop()
        clock.setdefault("attempts", []).append({"mhz": mhz, "overlap": overlap})
        if mhz and overlap >= 0.8:
            clock["stage_profile"], clock["overlap"] = mhz, overlap
            break
    stage_ms = np.asarray(rows)
    # R: raster-only rate (one full-scene RGB+depth forward per view), for the record next to F
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        fr.render_batch(eng.batch_views(i), eng.frames, masks=False)
    torch.cuda.synchronize()
    raster_only_fps = args.steps * B / (time.perf_counter() - t1)

    N = eng.cloud.n
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
    evals_per_s = evals * B / (mean_ms[4] * 1e-3) if mean_ms[4] > 0 else None
    # `bound` starts as "hbm" (the formula's bytes against the HBM peak) and is replaced below by what the counters show for
    # this kernel when they are on file: the compositor and the binning walks are vector-ISSUE bound, not bandwidth bound
    roofline = {
        "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "hbm": {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5)},
        "hbm_frac": round(achieved / HBM_PEAK_GBS, 5),
        "kernel_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(dom_bytes),
        "launches_per_step": launches,
        "stage_ms_per_view": {k: round(float(m) / B, 4) for k, m in zip(_lib.STAGE_NAMES, mean_ms)},
        "whole_path": {"bytes_per_view": int(B_view), "achieved": round(B_view * value / world / 1e9, 2),
                       "frac": round(B_view * value / world / 1e9 / HBM_PEAK_GBS, 5)},
        # the compositor is VALU-bound: SURVEY.md section 8d's secondary ceiling, evaluations from the n_contrib sums
        "valu": (None if not evals_per_s else
                 {"evals_per_s": round(evals_per_s, 1), "lane_ops_per_eval": LANE_OPS_PER_EVAL, "peak": VALU_PEAK_LANE_OPS,
                  "unit": "lane-op/s", "frac": round(evals_per_s * LANE_OPS_PER_EVAL / VALU_PEAK_LANE_OPS, 4),
                  "definition": "sum of n_contrib over the batch (pixel-Gaussian evaluations up to each pixel's last blended "
                                "entry) / composite stage time x 20 lane-ops, against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"}),
        "composite_evals_per_s": round(evals_per_s, 1) if evals_per_s else None,
        "semantic": "separate objects-only pass" if args.separate_semantic else
                    "fused: second accumulator in the scene's compositing walk (stage composite; +16 P bytes of image writes "
                    "per view that SURVEY's 44 I + 16 P does not count)",
        "raster_only_views_per_s": round(raster_only_fps, 2),
        "N": N, "V": round(V), "I": round(I), "P": P,
    }
    # HBM-side traffic and unit utilisation of the dominant kernel: rocprofv3 --pmc passes of this same command (separate
    # passes per counter group, gfx950 unit and FETCH_SIZE corrections: profiles/README.md), written by
    # scripts/pmc_profile.sh -> scripts/pmc_report.py as ONE file, so the numbers here and the committed text summary
    # cannot diverge
    try:
        # (profiles/pmc.json = the C3 profile; other workloads: profiles/pmc_<workload>.json when one is on file)
        pmc_file = ROOT / "profiles" / ("pmc.json" if args.workload == "c3" else f"pmc_{args.workload}.json")
        pmc = json.loads(pmc_file.read_text())
        kern = pmc["stage_kernel"].get(dom_name)
        k = pmc["kernels"].get(kern) if kern else None
        fused_default = with_masks and not args.separate_semantic
        if (k and pmc.get("workload") == args.workload and pmc.get("batch") == B and pmc.get("fused") == fused_default
                and (W, H) == (800, 800) and not args.objects and args.scale == 1.0 and not args.dynamic):
            roofline["traffic"] = int(k["traffic_bytes_per_launch"])
            roofline["traffic_detail"] = {kk: k[kk] for kk in ("kernel", "dispatches", "fetch_size_kib_per_launch",
                                                               "write_size_kib_per_launch") if kk in k}
            roofline["traffic_detail"]["kernel"] = kern
            roofline["traffic_source"] = pmc["source"]
            # the whole path's HBM-side traffic (counters) beside the formula's bytes: every kernel one frames batch launches
            per_batch = sum(kk["traffic_bytes_per_launch"]        # (each of them is launched once per batch)
                            for name, kk in pmc["kernels"].items()
                            if "traffic_bytes_per_launch" in kk and (not name.startswith("composite_quarter_kernel") or name == kern))
            roofline["whole_path"]["traffic_bytes_per_view"] = int(per_batch / B)
            roofline["whole_path"]["traffic_frac"] = round(per_batch / B * value / world / 1e9 / HBM_PEAK_GBS, 5)
            if "valu_busy" in k:
                roofline["valu_issue"] = {"busy_frac": k["valu_busy"], "lds_busy_frac": k.get("lds_busy"), "kernel": kern,
                                          "definition": pmc.get("busy_definition")}
                hbm_live = k["traffic_bytes_per_launch"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_ms > 0 else 0.0
                roofline["bound_by_counters"] = {"valu_issue_busy": round(float(k["valu_busy"]), 4),
                                                 "lds_busy": round(float(k.get("lds_busy") or 0.0), 4), "hbm": round(hbm_live, 4)}
                # The issue model (profiles/issue_model.json, scripts/issue_model.py): cycles this kernel's vector instructions
                # NEED at the measured per-kind issue costs, against the cycles it TOOK = live duration x the shader clock read
                # during the stage-profile run.  Only with counters of THESE kernels (the source hash the loaded library was built from).
                lib_sha = _lib.lib().pgr_version().decode().rsplit(" ", 1)[-1]      # hash of the sources the library was built from
                im = json.loads((ROOT / "profiles" / "issue_model.json").read_text())
                mk = im["kernels"].get(args.workload, {}).get(kern)
                same_build = pmc.get("library_sha16") == lib_sha and im.get("library_sha16", {}).get(args.workload) == lib_sha
                clock_mhz = clock.get("stage_profile")
                if mk and same_build and clock_mhz and dom_ms > 0 and float(k["valu_busy"]) >= max(float(k.get("lds_busy") or 0.0), hbm_live):
                    taken = dom_ms * 1e-3 * clock_mhz * 1e6
                    insts = float(mk["valu_insts"])
                    roofline.update(
                        bound="valu_issue", achieved=round(insts / (dom_ms * 1e-3) / 1e9, 3),
                        peak=round(N_SIMD * clock_mhz * 1e6 / mk["cycles_per_inst"] / 1e9, 1), unit="G wave-inst/s",
                        frac=round(mk["cycles_needed"] / taken, 4),
                        frac_definition="SIMD cycles this kernel's vector instructions need (per SQ class: counter count x mean "
                                        "measured issue cost of the class in the kernel's ISA; profiles/r06_issue_model.txt) / "
                                        "cycles it took (live HIP-event duration x clock_mhz); peak = 1024 SIMDs x clock / "
                                        "cycles_per_inst_model")
                    roofline["issue_model"] = {
                        "kernel": kern, "clock_mhz": round(clock_mhz, 1), "clock_probe_overlap": clock.get("overlap"),
                        "clock_probe_attempts": clock.get("attempts"),
                        "clock_source": "pgr_clock_probe: one wave's s_memtime / s_memrealtime ticks on a side stream during the stage-profile "
                                        "run; clock_probe_overlap = share of that run the probe was resident for (time stamps on both streams)",
                        "cycles_per_inst_model": mk["cycles_per_inst"], "valu_insts_per_launch": insts,
                        "cycles_needed_per_simd": mk["cycles_needed"], "cycles_taken": round(taken),
                        "frac_all_cheapest_kind": round(mk["frac_cheapest"] * mk["kernel_cycles"] / taken, 4),
                        "frac_all_dearest_kind": round(mk["frac_dearest"] * mk["kernel_cycles"] / taken, 4),
                        "frac_under_profiler": mk["frac"], "salu_per_cu_cycle": mk["salu_frac"], "class_cost_cycles": mk["class_cost"],
                        # every kernel of the path by the same model, under the profiler's clock (its own cycles)
                        "path_under_profiler": {name: kk["frac"] for name, kk in im["kernels"].get(args.workload, {}).items()
                                                if not name.startswith("composite_quarter_kernel") or name == kern},
                        "source": "scripts/issue_model.py: profiles/r06_valu_classes.txt (costs), r06_valu_classes_pmc.txt (classes), "
                                  + pmc_file.name + " (class counters), hipcc --save-temps ISA (split inside a class)",
                        # what the model is NOT: a promise that 1 - frac is there to be had by trimming vector instructions
                        "model_check": "a build with 9 % fewer priced vector cycles per valid entry but two scalar instructions and a "
                                       "branch more (blend under an exec mask) ran no faster; capped at 6 / 5 / 4 waves per SIMD the kernel "
                                       "takes +3 / +8 / +18 % (the flat part of its occupancy curve: an issue limit, not latency).  The limit "
                                       "is the SIMD's issue as a whole -- this vector figure AND salu_per_cu_cycle -- not the vector port "
                                       "alone (profiles/r06_blend_exec_ab.txt)"}
                elif not same_build:
                    roofline["bound_note"] = (f"counter files on record describe another build of the library (pmc {pmc.get('library_sha16')}, "
                                              f"issue model {im.get('library_sha16', {}).get(args.workload)}, loaded {lib_sha}): `bound` stays the "
                                              "formula's HBM figure; re-run scripts/r06_issue_model.sh + scripts/issue_model.py")
    except (OSError, ValueError, KeyError):
        pass
    if "bound_by_counters" not in roofline and "bound_note" not in roofline:
        roofline["bound_note"] = ("no counter profile on file for this workload / shape: `bound` is the formula's HBM figure only; on the "
                                  "profiled configs (profiles/pmc*.json) this kernel is vector-issue bound, not bandwidth bound")

    # the two side legs must not be able to lose the headline: a failure in one of them is recorded in its place
    def side_leg(fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as e:                              # noqa: BLE001 (reported in the line, with its type)
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {"error": f"{type(e).__name__}: {e}"[:500]}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = side_leg(_cpu_baseline, args, eng)
    drop_in = None
    if world == 1 and not args.no_drop_in and with_masks and not args.dynamic and args.workload in ("c3", "c5"):
        drop_in = side_leg(drop_in_numbers, eng, n_frames=6, n_render_calls=48)
        dyn = side_leg(drop_in_numbers, eng, n_frames=6, n_render_calls=4, dynamic=True)
        if isinstance(drop_in, dict) and "error" not in drop_in:
            drop_in["dynamic"] = ({k: dyn[k] for k in ("mode", "frames_per_s", "ms_per_frame", "frames_per_s_all_data_points",
                                                       "ms_per_part", "passes_ms_per_frame")} if "error" not in dyn else dyn)

    N_label = f"{N / 1e6:.2g}M"
    line.update(
        metric=(f"rendered views/sec (RGB+depth+mask) on {N_label}-Gaussian scene @{W}x{H}" if with_masks else
                f"rendered views/sec (RGB+depth, raster only) on {N_label}-Gaussian scene @{W}x{H}"),
        config={"workload": eng.label, "gaussians": N, "width": W, "height": H, "views_per_step": B,
                "distinct_views": len(eng.views) // world, "objects": fr.K,
                "camera_set": (args.camera_set + ": " + CAMERA_SET_TEXT[args.camera_set].format(n=len(eng.views))
                               if args.workload in ("c3", "c5") else "the workload's own views"),
                "sequence": (f"dynamic: every frame is a time step of the reference's recorded drop (simulation_steps.json "
                             f"body 1, steps 0..{SEQUENCE_STEPS - 1}, per-object phase offsets; poses composed absolutely and "
                             f"applied inside the preprocess) + BOP pose records" if args.dynamic else
                             "static scene, camera batches"),
                "scene_layout": ("input order" if fr.order is None else
                                 "Morton order per object (one-time, at scene load, outside the timed region)"),
                "outputs": (("color[3,H,W] f32 + depth[1,H,W] f32 + semantic image[3,H,W] f32 + masks[K,H,W] u8"
                             + (" + silhouette masks[K,H,W] u8" if eng.with_sil else ""))
                            if with_masks else "color[3,H,W] f32 + depth[1,H,W] f32"),
                "data_points": (["rgb", "depth", "seg_vis", "sem_seg"] + (["seg_sil"] if eng.with_sil else [])) if with_masks
                               else ["rgb", "depth"],
                "parallelism": f"view-shard x{world}", "distributed": dist_info, "gather": gather_info},
        roofline=roofline, cpu_baseline=cpu, drop_in=drop_in)
    if rehearsal:
        line["rehearsal"] = ("NOT a result: ranks share devices, the gather runs on gloo through host memory, or a one-rank "
                             "process group was forced; only the launch, sharding, collective and gather logic is exercised")


def _cpu_baseline(args, eng):
    import oracle
    oracle.build()
    fr, act = eng.fr, eng.act
    # the port is timed at the thread count that suits it best on this host, not at "all of them": its parallel regions
    # are short, and on the GPU box's 256 hardware threads a frame takes 1.0 s where 32 threads take 0.15 s
    n_cpu = os.cpu_count() or 1
    trial = {}
    for c in sorted({c for c in (8, 16, 32, 64, 128, n_cpu) if c <= n_cpu}):
        t1 = time.perf_counter()
        oracle.forward(**act, sh_degree=3, **eng.views[0].raster_kwargs(), num_threads=c, want_binning=False)
        trial[c] = time.perf_counter() - t1
    cores = min(trial, key=trial.get)

    def frame_on(threads, v, step):
        """one FRAME of the metric on the CPU: scene pass + objects-only semantic pass + K masks (or the scene pass alone)"""
        posed = {} if eng.pose_seq is None else dict(object_id=oid, poses=eng.pose_seq[step % SEQUENCE_STEPS])
        t1 = time.perf_counter()
        oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=threads, want_binning=False, **posed)
        if with_masks:
            posed_o = {} if eng.pose_seq is None else dict(object_id=oid[n_env:], poses=eng.pose_seq[step % SEQUENCE_STEPS])
            seg = oracle.forward(act["means3d"][n_env:], act["opacities"][n_env:], scales=act["scales"][n_env:],
                                 rotations=act["rotations"][n_env:], shs=sem_shs, sh_degree=0,
                                 **v.raster_kwargs(), num_threads=threads, want_binning=False, **posed_o)
            oracle.color_masks(seg["color"], fr.colors_np, 0.1)
        return time.perf_counter() - t1
    n_done, t_cpu = 0, 0.0
    n_env = fr.n_env
    with_masks = eng.with_masks
    sem_shs = fr.sem_shs.cpu().numpy() if with_masks else None
    oid = eng.cloud.object_id
    while n_done < len(eng.views) and (n_done == 0 or t_cpu + t_cpu / n_done < args.cpu_budget_s):
        t_cpu += frame_on(cores, eng.views[n_done], n_done)
        n_done += 1
    # SURVEY.md section 8d asks for (i) 1 thread and (ii) all host cores beside it: one frame each (a 1-thread frame of
    # the 2 M-Gaussian scene takes seconds, the bound on this leg's run time)
    t_one = frame_on(1, eng.views[0], 0)
    t_all = frame_on(n_cpu, eng.views[0], 0)
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        cpu_model = "unknown"
    return {"value": round(n_done / t_cpu, 4), "unit": "frames/s" if with_masks else "views/s", "cores": cores,
            "kind": "port", "cpu_model": cpu_model,
            "one_thread": {"value": round(1.0 / t_one, 4), "cores": 1, "sample": "1 frame"},
            "all_cores": {"value": round(1.0 / t_all, 4), "cores": n_cpu, "sample": "1 frame"},
            "sample": f"first {n_done} frame(s) of the same scene and cameras, oracle/pgr_oracle.c with OpenMP, "
                      f"reference-style lists; {cores} threads = the fastest of a one-view trial at "
                      f"{{{', '.join(f'{c}: {t:.2f} s' for c, t in trial.items())}}} on this host's {n_cpu} hardware threads; "
                      f"no reference CPU rasterizer exists"}


# --------------------------------------------------------------------------------------------------------------------
# the drop-in path: what unchanged PEGASUS calls

def drop_in_numbers(eng, n_frames=4, n_render_calls=48, dynamic=False):
    """PEGASUS's own per-camera loop on this scene (/root/reference/pegasus.py:254-358): per frame one deepcopy + merge of
    the scene, then render_rgb_and_depth, render_visib_mask and render_semanticsegmentation_mask through
    pegasus_amd/render.py's wrappers (same names and arguments as /root/reference/src/gs/render.py) -- and the latency of a
    single gaussian_renderer.render() call over the merged model."""
    import copy
    import torch
    from argparse import ArgumentParser
    sys.path.insert(0, str(ROOT / "compat"))
    from pegasus_amd import render as RW
    from pegasus_amd import gaussian_renderer as GR
    from pegasus_amd.cameras import Camera
    from pegasus_amd.gaussian_model import GaussianModel
    from pegasus_amd import masks as M
    from arguments import PipelineParams
    dev, cloud = eng.dev, eng.cloud
    oid = cloud.object_id
    K = int(oid.max())

    def model(sel):
        return GaussianModel.from_arrays(cloud.xyz[sel], cloud.features_dc[sel], cloud.features_rest[sel], cloud.opacity[sel],
                                         cloud.scaling[sel], cloud.rotation[sel], device=dev)
    env = model(oid == 0)
    objects = {k: model(oid == k) for k in range(1, K + 1)}
    colors = M.generate_colors(K)
    RW.assign_semantic_colors(objects, colors)
    color_set = torch.as_tensor(colors, device=dev)
    cams = [Camera(colmap_id=i, R=v.R_c2w, T=v.t_w2c, FoVx=v.fovx, FoVy=v.fovy, image=None, image_width=v.width,
                   image_height=v.height, gt_alpha_mask=None, image_name=str(i), uid=i, data_device=dev)
            for i, v in enumerate(eng.views[:max(n_frames + 1, n_render_calls)])]
    pipe = PipelineParams(ArgumentParser())
    bg = torch.zeros(3, device=dev)
    H, W = eng.H, eng.W

    parts = {"compose": 0.0, "render_rgb_and_depth": 0.0, "render_visib_mask": 0.0, "render_semanticsegmentation_mask": 0.0,
             "render_silhouette_mask": 0.0, "update_object_pose": 0.0}

    # DYNAMIC mode of the unchanged loop (/root/reference/pegasus.py:387-390 -> src/gs/pegasus_setup.py:178-226): after every
    # frame each object receives the DELTA between two samples of its recorded trajectory through the reference's three
    # calls (apply_transformation_on_xyz, apply_rotation_on_splats, apply_rotation_on_sh) -- which replaces its tensors, so
    # the kept objects-only semantic scene of pegasus_amd/render.py is rebuilt every frame
    traj = None
    if dynamic:
        from scipy.spatial.transform import Rotation
        from pegasus_amd import trajectory as TJ
        traj = TJ.load_fixture()
    step = {"i": 0}

    def update_object_pose():
        i = step["i"] = step["i"] + 1
        for k, obj in objects.items():
            a, b = min(len(traj) - 1, i + 5 * (k - 1)), min(len(traj) - 1, i - 1 + 5 * (k - 1))
            t_delta = torch.from_numpy(traj[a, 0:3] - traj[b, 0:3]).type(torch.float32).to(dev)
            q_delta = Rotation.from_quat(traj[a, 3:7]) * Rotation.from_quat(traj[b, 3:7]).inv()
            Rm = torch.from_numpy(q_delta.as_matrix()).type(torch.float32).to(dev)
            T = torch.eye(4, dtype=torch.float32, device=dev)
            T[:3, :3] = Rm
            T[:3, 3] = t_delta
            obj.apply_transformation_on_xyz(T=T)                     # pegasus_setup.py:195-208
            obj.apply_rotation_on_splats(R=Rm)
            obj.apply_rotation_on_sh(R=Rm)

    def lap(name, t_prev):
        torch.cuda.synchronize()
        now = time.perf_counter()
        parts[name] += now - t_prev
        return now

    def frame(cam, clock=False, silhouettes=False):
        t = time.perf_counter()
        scene = copy.deepcopy(env)                                   # pegasus.py:255-264
        for obj in objects.values():
            obj._features_dc = copy.deepcopy(obj._features_dc_color)
            obj._features_rest = copy.deepcopy(obj._features_rest_color)
            scene.merge_gaussians(gaussian=obj)
        t = lap("compose", t) if clock else t
        rgb, depth = RW.render_rgb_and_depth(cam, scene, pipe, bg)
        t = lap("render_rgb_and_depth", t) if clock else t
        masks, seg = RW.render_visib_mask(cam, env, objects, color_set, H, W, pipe, bg)
        t = lap("render_visib_mask", t) if clock else t
        sem = RW.render_semanticsegmentation_mask(cam, env, objects, color_set, H, W, pipe, bg, False)
        t = lap("render_semanticsegmentation_mask", t) if clock else t
        if silhouettes or clock:                                     # 'seg_sil' (pegasus.py:491's fifth data point)
            sil = RW.render_silhouette_mask(cam, objects, env, W, H, color_set, pipe, bg)
            t = lap("render_silhouette_mask", t) if clock else t
        if traj is not None:
            update_object_pose()
            t = lap("update_object_pose", t) if clock else t
        return scene

    with torch.no_grad():
        for _ in range(3):                                           # warm-up: allocator blocks of every size, pinned host
            scene = frame(cams[0])                                   # buffers, first-use costs of the pose path
        # a host-bound loop on a shared box: the median of three passes (one pass alone has come out a third low -- a
        # neighbour on the host's cores -- where the passes before and after it agreed to 3 %)
        def timed_pass(**kw):
            nonlocal scene
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in cams[1:n_frames + 1]:
                scene = frame(c, **kw)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n_frames
        passes = sorted(timed_pass() for _ in range(3))
        t_frame = passes[1]
        frame(cams[0], silhouettes=True)
        passes_all = sorted(timed_pass(silhouettes=True) for _ in range(3))      # all five default data points: + the K silhouettes
        t_frame_all = passes_all[1]
        for c in cams[1:n_frames + 1]:                               # once more with a synchronisation after every part
            frame(c, clock=True)
        for c in cams[:4]:
            GR.render(c, scene, pipe, bg)
        calls = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in cams[:n_render_calls]:
                GR.render(c, scene, pipe, bg)
            torch.cuda.synchronize()
            calls.append((time.perf_counter() - t0) / min(n_render_calls, len(cams)))
        t_call = sorted(calls)[1]
    return {"mode": "dynamic" if dynamic else "static",
            "frames_per_s": round(1.0 / t_frame, 2), "ms_per_frame": round(t_frame * 1e3, 3),
            "frames_per_s_all_data_points": round(1.0 / t_frame_all, 2),
            "render_call_ms": round(t_call * 1e3, 4), "render_calls_per_s": round(1.0 / t_call, 1),
            "ms_per_part": {k: round(v / n_frames * 1e3, 3) for k, v in parts.items()},
            "frame": "deepcopy + merge of the scene, render_rgb_and_depth, render_visib_mask (K masks to the host as float64, "
                     "as the reference returns them) and render_semanticsegmentation_mask -- the ['rgb','seg_vis','sem_seg'] "
                     "data points of /root/reference/pegasus.py:254-358, one camera per frame (frames_per_s_all_data_points: + "
                     "render_silhouette_mask, 'seg_sil': all K objects in one layered call, K float64 masks to the host).  STATIC scene: the two semantic "
                     "wrappers share one objects-only scene and one render per camera, kept while no object moves.  DYNAMIC "
                     "(mode = dynamic): after every frame each object is moved by the delta of its recorded trajectory through the "
                     "reference's three pose calls (pegasus.py:387-390, pegasus_setup.py:178-208), so the objects-only scene is "
                     "rebuilt every frame",
            "sample": f"median of three passes of {n_frames} frames / of {min(n_render_calls, len(cams))} render() calls; same scene and "
                      "cameras as the batch path",
            "passes_ms_per_frame": [round(t * 1e3, 3) for t in passes]}


def run_cpu_only(args):
    """configs[0]: the CPU oracle alone (the checker timed as the reported CPU baseline -- the only role in which bench.py may
    run anything under oracle/)."""
    import oracle
    oracle.build()
    cloud, views, label = build_workload(args.workload, args.scale, max(1, min(args.views, 8)), camera_set=args.camera_set,
                                         width=args.width, height=args.height, objects=args.objects)
    act = cloud.activated()
    n_cpu = os.cpu_count() or 1
    rows = {}
    for threads in sorted({1, min(8, n_cpu), n_cpu}):
        oracle.forward(**act, sh_degree=3, **views[0].raster_kwargs(), num_threads=threads, want_binning=False)     # warm
        t0, n = time.perf_counter(), 0
        while n < 200 and (n < 3 or time.perf_counter() - t0 < args.cpu_budget_s / 3):
            oracle.forward(**act, sh_degree=3, **views[n % len(views)].raster_kwargs(), num_threads=threads, want_binning=False)
            n += 1
        rows[threads] = n / (time.perf_counter() - t0)
    best = max(rows, key=rows.get)
    emit({"metric": f"CPU oracle views/sec (RGB+depth) on {cloud.n}-Gaussian scene @{views[0].width}x{views[0].height} (no GPU)",
          "value": round(rows[best], 3), "unit": "views/s", "n_gpus": 0, "higher_is_better": True, "dtype": "f32",
          "data": "synthetic", "cpu_only": True, "config": {"workload": label},
          "cpu_baseline": {"value": round(rows[best], 3), "unit": "views/s", "cores": best, "kind": "port",
                           "per_thread_count": {str(k): round(v, 3) for k, v in rows.items()},
                           "sample": "oracle/pgr_oracle.c (reference-style lists) on the workload's first cameras"}})
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.cpu_only:
        if "--workload" not in " ".join(argv):
            args.workload = "c1"
        return run_cpu_only(args)
    if (args.gpus or 1) > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    if args.facade:
        args.no_cpu_baseline = True
    return run_worker(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
