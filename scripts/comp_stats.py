"""Debug: compositor work statistics from a -DPGR_COMP_STATS (-DPGR_SORT_STATS) build of the library, loaded through PGR_LIB
(hipcc <pegasus_amd/build.py FLAGS> -DPGR_COMP_STATS -DPGR_SORT_STATS -o build_variants/lib_stats.so pegasus_raster.hip;
PGR_LIB=$PWD/build_variants/lib_stats.so python scripts/comp_stats.py c3 frames -- the product .so is never overwritten)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "..")
import bench
from pegasus_amd import _lib, frames as F

L = _lib.lib()
handle = C.CDLL(str(_lib.LIB_PATH))
workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = 16
cloud, views, label = bench.build_workload(workload, 1.0, B)
act = cloud.activated()
fr = F.FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id,
                     sh_degree=3, device="cuda:0")
specs = [fr.view_spec(v) for v in views[:B]]
fr.render_frames(specs, None, masks=False)
out = (C.c_ulonglong * 32)()
if hasattr(handle, "pgr_debug_sort_stats"):
    handle.pgr_debug_sort_stats(out, 1)
    fr.render_frames(specs, None, masks=False)
    handle.pgr_debug_sort_stats(out, 1)
    lists, keys, sq, rl, rk = [out[i] / B for i in range(5)]
    print(f"{label}: bucket sort per view: lists {lists:.0f}  keys {keys/1e6:.2f} M  sum k^2 / keys {sq/max(keys,1):.2f}  "
          f"rejected lists {rl:.1f} ({rk/max(keys,1):.2%} of keys)")
if not hasattr(handle, "pgr_debug_comp_stats"):
    raise SystemExit(0)
handle.pgr_debug_comp_stats(out, 1)
masks = len(sys.argv) > 2 and sys.argv[2] == "frames"
fr.render_frames(specs, None, masks=masks)
handle.pgr_debug_comp_stats(out, 1)
walk, live, ev, alive, blend, waves, batches = [out[i] / B for i in range(7)]
print(f"fused semantic: {masks}; semantic wave-entries with their own blend {(out[7] & 0xffffffff) / B / 1e6:.2f} M, "
      f"riding the scene blend (quarter still object-only) {(out[7] >> 32) / B / 1e6:.2f} M; pairs evaluated ~{live / 2e6:.2f} M")
print(f"{label}: per view: waves {waves:.0f}  batches {batches:.0f}  entries walked {walk/1e6:.2f} M  live after skip {live/1e6:.2f} M "
      f"({live/walk:.2%})  wave-entries evaluated {ev/1e6:.2f} M  pixel-entries: alive {alive/1e6:.1f} M "
      f"({alive/(ev*64):.2%} of lanes)  blended {blend/1e6:.1f} M ({blend/(ev*64):.2%})  live/batch {live/batches:.1f}")
if masks:
    fw, tb, tw, tg, tl, tin, tend = [out[i] / B for i in range(8, 15)]
    print(f"fused quarters {fw:.0f} per view; {tin:.0f} enter the semantic tail (scene pixels saturated, objects-only walk continues), "
          f"{tend:.0f} walk to the tile's last object entry; tail: {tb:.0f} batches, {tw/1e6:.3f} M entries walked, "
          f"{tg/1e6:.3f} M object records gathered, {tl/1e6:.3f} M live after the skip test")
    names = ("PLAIN", "RIDE", "GENERAL")
    mb, me = [out[22 + m] / B for m in range(3)], [out[25 + m] / B for m in range(3)]
    print("fused quarters, batches with parked entries per pair-loop mode: " +
          ", ".join(f"{n} {b:.0f} batches / {e / 1e6:.3f} M entries" for n, b, e in zip(names, mb, me)))
