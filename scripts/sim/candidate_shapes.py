"""How many candidate tiles would the binning walks enumerate per view if a splat's candidates were a PARALLELOGRAM of
tiles (a fixed number of tiles per tile row, shifted along the ellipse's centre line) instead of the axis-aligned box of
its alpha >= 1/255 ellipse?  CPU only, on oracle data:   python scripts/sim/candidate_shapes.py [c3|c5] [view]"""
import math
import sys

import numpy as np

sys.path.insert(0, ".")
import oracle
from pegasus_amd import scenes

oracle.build()
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
vi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cloud, views = (scenes.scene_c5 if wl == "c5" else scenes.scene_c3)(n_views=max(4, vi + 1))
v = views[vi]
act = cloud.activated()
o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=32, cull_mode=1)
listed = int(o["num_instances"])
xy, co, radii = o["xy"], o["conic_opacity"], o["radii"]
vis = radii > 0
x, y = xy[vis, 0].astype(np.float64), xy[vis, 1].astype(np.float64)
A, B, C, op = (co[vis, k].astype(np.float64) for k in range(4))
rad = radii[vis].astype(np.float64)
T = 16
gx, gy = (v.width + T - 1) // T, (v.height + T - 1) // T
ok = (op >= 1 / 255) & (A > 0) & (C > 0) & (A * C - B * B > 0)
x, y, A, B, C, op, rad = (a[ok] for a in (x, y, A, B, C, op, rad))
tau = 2.0 * np.log(255.0 * op) * 1.001 + 0.02            # (the kernel's bound is slightly looser still)
det = A * C - B * B
cxx, cyy = C / det, A / det                                  # covariance diagonal
ex, ey = np.sqrt(tau * cxx) + 1.0, np.sqrt(tau * cyy) + 1.0
# 3-sigma rectangle (reference) and the ellipse box, in tiles
r_minx = np.clip(np.floor((x - rad) / T), 0, gx); r_maxx = np.clip(np.floor((x + rad + T - 1) / T), 0, gx)
r_miny = np.clip(np.floor((y - rad) / T), 0, gy); r_maxy = np.clip(np.floor((y + rad + T - 1) / T), 0, gy)
b_minx = np.maximum(r_minx, np.floor((x - ex - (T - 1)) / T)); b_maxx = np.minimum(r_maxx, np.floor((x + ex) / T) + 1)
b_miny = np.maximum(r_miny, np.floor((y - ey - (T - 1)) / T)); b_maxy = np.minimum(r_maxy, np.floor((y + ey) / T) + 1)
bw, bh = np.maximum(b_maxx - b_minx, 0), np.maximum(b_maxy - b_miny, 0)
box = bw * bh
# parallelogram by rows: at height dy the ellipse spans  -B/A dy +- sqrt((tau - det/A dy^2)/A);  over a row's pixel-centre
# span [16 r, 16 r + 15] the centre line moves by |B/A| * 15 and the half width is at most hw = sqrt(tau / A)
def para(A, B, det, tau, bw_, bh_):
    hw = np.sqrt(tau / A) + 1.0
    shift = np.abs(B / A) * (T - 1)
    per_row = np.floor((2 * hw + shift + (T - 1)) / T) + 1          # tiles whose span meets an interval of that length
    return np.minimum(per_row, bw_) * bh_
rows = para(A, B, det, tau, bw, bh)
cols = para(C, B, det, tau, bh, bw)
best = np.minimum(rows, cols)
# exact per-row intervals (the ellipse cut by each row band): the floor of any row-wise scheme
print(f"{wl} view {vi}: {int(vis.sum())} visible Gaussians, {listed} listed instances")
for name, a in (("3-sigma rectangle (reference lists)", (r_maxx - r_minx) * (r_maxy - r_miny)), ("ellipse box (today)", box),
                ("parallelogram by rows", rows), ("parallelogram by columns", cols), ("the better of the two", best)):
    print(f"  {name:38s} {a.sum() / 1e6:7.2f} M candidates   listed / candidates = {listed / a.sum():.2f}")
big = box > 16
print(f"  splats with a box of more than 16 tiles: {big.mean():.3f} of the splats, {box[big].sum() / box.sum():.2f} of today's candidates, "
      f"{best[big].sum() / box[big].sum():.2f} of those left by the parallelogram")

# exact per-row intervals: the ellipse {q <= tau} cut by each tile row's pixel-centre band -- the floor of any row-wise scheme
gi = np.repeat(np.arange(len(x)), bh.astype(np.int64))                      # one entry per (splat, tile row of its box)
first = np.cumsum(bh.astype(np.int64)) - bh.astype(np.int64)
row = b_miny[gi] + (np.arange(len(gi)) - first[gi])
d0, d1 = row * T - y[gi], row * T + (T - 1) - y[gi]                         # band in splat-centred coordinates
Ag, Bg, Cg, tg = A[gi], B[gi], C[gi], tau[gi]
k = Cg - Bg * Bg / Ag                                                        # q = A (dx + B/A dy)^2 + k dy^2
def edge(dy):                                                                # x range of the ellipse at height dy (nan: none)
    h2 = (tg - k * dy * dy) / Ag
    h = np.sqrt(np.where(h2 >= 0, h2, np.nan))
    return -Bg / Ag * dy - h, -Bg / Ag * dy + h
lo0, hi0 = edge(d0); lo1, hi1 = edge(d1)
# the rightmost / leftmost points of the whole ellipse sit at dy = -+ B/A sqrt(tau / k') ...: take them when inside the band
exx = np.sqrt(tg * Cg / det[gi])                                             # global half extent in x
dy_r = -Bg / Cg * exx                                                        # height of the rightmost point (dq/dy = 0 there)
lo = np.fmin(lo0, lo1); hi = np.fmax(hi0, hi1)
hi = np.where((dy_r >= d0) & (dy_r <= d1), exx, hi)
lo = np.where((-dy_r >= d0) & (-dy_r <= d1), -exx, lo)
# a band that contains the centre line's crossing but neither edge point: covered by the cases above or empty
ok_row = np.isfinite(lo) & np.isfinite(hi)
xl = np.floor((x[gi] + lo - 1.0 - (T - 1)) / T); xr = np.floor((x[gi] + hi + 1.0) / T) + 1
wrow = np.where(ok_row, np.clip(np.minimum(xr, b_maxx[gi]) - np.maximum(xl, b_minx[gi]), 0, None), 0)
print(f"  {'exact interval per tile row (+-1 px)':38s} {wrow.sum() / 1e6:7.2f} M candidates   listed / candidates = {listed / wrow.sum():.2f}"
      f"   ({len(gi) / 1e6:.2f} M (splat, row) pairs to compute)")
# the same restricted to splats of at most 4 tile rows (and 15 columns), whose per-row (offset, width) pairs fit one 32-bit word
small = (bh <= 4) & (bw <= 15)
per_splat = np.bincount(gi, weights=wrow, minlength=len(x))
mixed = np.where(small, per_splat, box)
print(f"  {'row intervals for boxes of <= 4 rows only':38s} {mixed.sum() / 1e6:7.2f} M candidates   listed / candidates = {listed / mixed.sum():.2f}"
      f"   ({small.mean():.3f} of the splats, {box[small].sum() / box.sum():.2f} of today's candidates)")
for hh in (1, 2, 3, 4):
    m = bh == hh
    print(f"    boxes of {hh} rows: {m.mean():.3f} of the splats, today {box[m].sum() / 1e6:.2f} M -> {per_splat[m].sum() / 1e6:.2f} M")
