"""Workload statistics of one view on the GPU path (list lengths, early-out depth, tail)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from pegasus_amd import scenes
from helpers import gpu_forward

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cloud, views = scenes.scene_c3(scale=scale, n_views=4)
act = cloud.activated()
for v in views[:3]:
    g = gpu_forward(act, v)
    rng = g["ranges"].astype(np.int64)
    ln = rng[:, 1] - rng[:, 0]
    H, W = v.height, v.width
    nc = g["n_contrib"].astype(np.int64)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tile_max = nc.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    # per wave (4 rows of 16) max
    wave_max = nc.reshape(gy, 4, 4, gx, 16).max(axis=(2, 4))
    print(f"N={cloud.n} V={(g['radii']>0).sum()} I={g['num_instances']}")
    print("  list len: mean %.0f  p50 %d p90 %d p99 %d max %d" % (ln.mean(), *np.percentile(ln, [50, 90, 99]), ln.max()))
    print("  n_contrib per pixel: mean %.0f p50 %d p99 %d max %d" % (nc.mean(), *np.percentile(nc, [50, 99]), nc.max()))
    print("  tile max n_contrib: mean %.0f p50 %d p90 %d p99 %d max %d ; sum(256*tile_max)=%.3g ; sum(n_contrib)=%.3g ; sum(256*len)=%.3g" % (
        tile_max.mean(), *np.percentile(tile_max, [50, 90, 99]), tile_max.max(), 256.0 * tile_max.sum(), nc.sum(), 256.0 * ln.sum()))
    print("  sum(64*wave_max)=%.3g" % (64.0 * wave_max.sum()))
    r = g["radii"][g["radii"] > 0]
    tt = g["tiles_touched"][g["tiles_touched"] > 0]
    print("  radii: mean %.1f p50 %d p99 %d max %d; tiles_touched mean %.2f p99 %d max %d" % (r.mean(), *np.percentile(r, [50, 99]), r.max(), tt.mean(), np.percentile(tt, 99), tt.max()))
    op = act["opacities"][g["radii"] > 0]
    print("  opacity of visible: mean %.2f  frac<0.1 %.3f" % (op.mean(), (op < 0.1).mean()))
