"""Offline estimate (CPU, oracle data) of what finer skip granularity could save in the compositor's pair loop.

For a sample of tiles of one C3 view: walk every 8x8 quarter's list exactly as the kernel does (64-entry batches, per-pixel
alpha / transmittance / stop rule in float64), and count per batch
  parked   entries that can reach a still-alive pixel of the quarter (exact form of the skip test)
  ... the same per 8x4 half and per 4x4 block, where each lane group would walk its OWN compacted list and the wave's trip
  count is the longest of its groups' lists.
Output: sum over batches of  parked(quarter)  vs  max(parked(top), parked(bottom))  vs  max over four 4x4 blocks."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle
from pegasus_amd import scenes

oracle.build()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cloud, views = scenes.scene_c3(scale=scale, n_views=8)
act = cloud.activated()
v = views[3]
o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
xy, co, gs, rg = o["xy"].astype(np.float64), o["conic_opacity"].astype(np.float64), o["gauss_sorted"], o["ranges"]
W, H = v.width, v.height
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(np.nonzero(rg[:, 1] > rg[:, 0])[0], size=min(n_tiles, int((rg[:, 1] > rg[:, 0]).sum())), replace=False)
tot = dict(walked=0, parked=0, evaluated=0, halves=0, blocks=0, rows2=0, halves_eval=0, batches=0)
for t in tiles:
    ids = gs[rg[t, 0]:rg[t, 1]]
    ty, tx = divmod(int(t), gx)
    for q in range(4):
        qx0, qy0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        px = (qx0 + np.arange(64) % 8).astype(np.float64)
        py = (qy0 + np.arange(64) // 8).astype(np.float64)
        inside = (px < W) & (py < H)
        dx = xy[ids, 0][:, None] - px[None, :]
        dy = xy[ids, 1][:, None] - py[None, :]
        A, B, C, op = (co[ids, k][:, None] for k in range(4))
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 0)))
        hit = (power <= 0) & (alpha >= 1 / 255) & inside[None, :]
        # sequential transmittance with the stop rule
        T = np.ones(64)
        alive = inside.copy()
        n = len(ids)
        alive_at = np.zeros((n, 64), bool)
        for i in range(n):
            alive_at[i] = alive
            if not alive.any():
                n = i
                break
            val = alive & hit[i]
            tt = T * (1 - alpha[i])
            stop = val & (tt < 1e-4)
            alive = alive & ~stop
            bl = val & ~stop
            T = np.where(bl, tt, T)
        tot["walked"] += ((n + 63) // 64) * 64 if n else 0
        for b0 in range(0, n, 64):
            sl = slice(b0, min(n, b0 + 64))
            a0 = alive_at[b0]                                  # alive pixels when the batch is staged
            reach0 = hit[sl] & a0[None, :]                     # what the (exact) skip test sees
            parked = reach0.any(1)
            tot["parked"] += int(parked.sum())
            tot["evaluated"] += int((hit[sl] & alive_at[sl]).any(1).sum())
            top, bot = reach0[:, :32].any(1), reach0[:, 32:].any(1)
            tot["halves"] += max(int(top.sum()), int(bot.sum()))
            e = hit[sl] & alive_at[sl]
            tot["halves_eval"] += max(int(e[:, :32].any(1).sum()), int(e[:, 32:].any(1).sum()))
            r = reach0.reshape(-1, 8, 8)
            blocks = [r[:, y0:y0 + 4, x0:x0 + 4].any((1, 2)).sum() for y0 in (0, 4) for x0 in (0, 4)]
            tot["blocks"] += int(max(blocks))
            rows2 = [r[:, y0:y0 + 2, :].any((1, 2)).sum() for y0 in (0, 2, 4, 6)]
            tot["rows2"] += int(max(rows2))
            tot["batches"] += 1
print(f"tiles {len(tiles)}  batches {tot['batches']}  entries walked {tot['walked']}")
print(f"parked per quarter (now)            {tot['parked']:9d}  1.000")
for k, name in (("halves", "two 8x4 halves, own lists"), ("blocks", "four 4x4 blocks, own lists"), ("rows2", "four 8x2 row pairs, own lists")):
    print(f"{name:35s} {tot[k]:9d}  {tot[k] / tot['parked']:.3f}")
print(f"evaluated (valid != 0) now          {tot['evaluated']:9d}; two halves {tot['halves_eval']} ({tot['halves_eval'] / max(1, tot['evaluated']):.3f})")
