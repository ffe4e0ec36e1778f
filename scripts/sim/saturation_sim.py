"""Offline (CPU, oracle data): how much of every tile's sorted list lies BEHIND the depth at which the tile's last pixel
saturates ("stop sorting and scattering what is never composited"), and what per-quarter reach bits in the
index words would let the compositor skip.

Per tile of one full-size view: alpha of every (entry, pixel) pair in float64, the stop rule as an inclusive running product
(before its stop a pixel blends every hit, so its stop entry is the first whose running product of (1 - alpha) over the hits
falls below 1e-4), then
  need(pixel)   = entries the walk visits for that pixel: stop index + 1, or the whole list if it never saturates
  need(tile)    = max over the tile's pixels      -> keys of the list the tile ever consumes (a lazy sort's lower bound)
  need(quarter) = max over the quarter's pixels   -> what a quarter wave walks (in 64-entry batches)
and per quarter, inside its walked prefix: entries that reach the quarter at all (STATIC: alpha >= 1/255 at some pixel of
the 8x8 block -- what a reach bit computed once per instance could say) and entries that reach a still-alive pixel when
their batch is staged (DYNAMIC: what today's alive-box skip test approximates).

    python scripts/sim/saturation_sim.py c3|c5 [view index] [scale]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle
from pegasus_amd import scenes

oracle.build()
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
vi = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cloud, views = (scenes.scene_c5 if wl == "c5" else scenes.scene_c3)(scale=scale, n_views=8)
act = cloud.activated()
v = views[vi]
o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
xy, co, gs, rg = o["xy"].astype(np.float64), o["conic_opacity"].astype(np.float64), o["gauss_sorted"], o["ranges"]
W, H = v.width, v.height
gx, gy = (W + 15) // 16, (H + 15) // 16
lens = (rg[:, 1] - rg[:, 0]).astype(np.int64)
I = int(lens.sum())
ly, lx = np.divmod(np.arange(256), 16)
quarter_of = (ly // 8) * 2 + lx // 8
tiers = [(1, 2048), (2049, 4096), (4097, 8192), (8193, 1 << 30)]
acc = dict(need_tile=0, need_tile64=0, need_q=0, need_q64=0, static=0, dynamic=0, batches_now=0, batches_static=0, nonsat_tiles=0,
           walked_entries=0)
tier_keys = [0] * 4
tier_need = [0] * 4
front = {256: 0, 512: 0, 1024: 0, 2048: 0}      # keys a "front F keys first" split would sort; tiles needing the back part
front_back_tiles = {k: 0 for k in front}
front_back_keys = {k: 0 for k in front}
for t in np.nonzero(lens)[0]:
    ids = gs[rg[t, 0]:rg[t, 1]]
    n = len(ids)
    ty, tx = divmod(int(t), gx)
    px = (tx * 16 + lx).astype(np.float64)
    py = (ty * 16 + ly).astype(np.float64)
    inside = (px < W) & (py < H)
    dx = xy[ids, 0][:, None] - px[None, :]
    dy = xy[ids, 1][:, None] - py[None, :]
    A, B, C, op = (co[ids, k][:, None] for k in range(4))
    power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
    alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 0)))
    hit = (power <= 0) & (alpha >= 1 / 255) & inside[None, :]
    cp = np.cumprod(np.where(hit, 1.0 - alpha, 1.0), axis=0)
    sat = cp < 1e-4
    anysat = sat.any(0)
    stop = np.where(anysat, sat.argmax(0), n - 1)            # index of the last entry the pixel's walk visits
    need_px = np.where(inside, stop + 1, 0)
    need_t = int(need_px.max())
    acc["need_tile"] += need_t
    acc["need_tile64"] += min(n, (need_t + 63) // 64 * 64)
    acc["nonsat_tiles"] += int((~anysat & inside).any())
    k = next(i for i, (a, b) in enumerate(tiers) if a <= n <= b)
    tier_keys[k] += n
    tier_need[k] += need_t
    for F in front:
        if n <= F:
            front[F] += n
        else:
            front[F] += F
            if need_t > F:
                front_back_tiles[F] += 1
                front_back_keys[F] += n - F
    # alive state of every pixel when entry i is visited: alive_at[i, p] = i <= stop[p]
    for q in range(4):
        m = (quarter_of == q) & inside
        if not m.any():
            continue
        need_q = int(need_px[m].max())
        acc["need_q"] += need_q
        nb = (need_q + 63) // 64
        acc["need_q64"] += min(n, nb * 64)
        acc["batches_now"] += nb
        hq = hit[:need_q][:, m]
        static = hq.any(1)
        acc["static"] += int(static.sum())
        acc["batches_static"] += (int(static.sum()) + 63) // 64
        # dynamic: reaches a pixel that is still alive when the entry's 64-batch is staged
        stop_q = stop[m]
        batch_start = (np.arange(need_q) // 64) * 64
        alive_at_batch = batch_start[:, None] <= stop_q[None, :]
        acc["dynamic"] += int((hq & alive_at_batch).any(1).sum())
        acc["walked_entries"] += need_q
print(f"{wl} view {vi}: N {cloud.n}  I {I}  non-empty tiles {int((lens > 0).sum())}  longest list {int(lens.max())}")
print(f"tile-level need (exact)            {acc['need_tile']:9d}  {acc['need_tile'] / I:.3f} of I   -> keys behind their tile's saturation: {1 - acc['need_tile'] / I:.3f}")
print(f"tile-level need, 64-entry batches  {acc['need_tile64']:9d}  {acc['need_tile64'] / I:.3f} of I")
print(f"tiles with a never-saturating pixel {acc['nonsat_tiles']} of {int((lens > 0).sum())}")
for (a, b), kk, nn in zip(tiers, tier_keys, tier_need):
    if kk:
        print(f"  lists of {a:5d}..{b if b < 1 << 30 else 'inf':>5}: keys {kk:9d} ({kk / I:.3f} of I), needed {nn / kk:.3f} of them")
for F in front:
    print(f"  front-{F:4d} split: first pass sorts {front[F] / I:.3f} of I; {front_back_tiles[F]} tiles need their back part "
          f"({front_back_keys[F] / I:.3f} of I sorted in a second pass) -> {1 - (front[F] + front_back_keys[F]) / I:.3f} of the keys never ranked")
print(f"quarter walks: exact {acc['need_q'] / (4 * I):.3f} of 4 I, in 64-batches {acc['need_q64'] / (4 * I):.3f}; batches {acc['batches_now']}")
w = acc["walked_entries"]
print(f"inside the walked prefixes: STATIC reach {acc['static'] / w:.3f}, DYNAMIC reach (alive pixels at batch start) {acc['dynamic'] / w:.3f}")
print(f"64-entry batches: now {acc['batches_now']}, of statically reaching entries {acc['batches_static']} ({acc['batches_static'] / acc['batches_now']:.3f})")
