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
