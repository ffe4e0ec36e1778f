"""Per-tile-row candidate intervals for splats of 2..4 tile rows (exact per-row extents of the ellipse {q <= tau2}) in numpy
fp32, checked for conservativeness against the oracle's LISTED instances (every listed (Gaussian, tile) must lie inside the
intervals) and counted: 5.78 M -> 4.53 M candidates per C3 view.  Round 6 built the cheaper parallelogram form of this into the
preprocess and both binning walks (lists bit-exact, 4.59 M candidates) and measured a net loss -- profiles/r06_rows_ab.txt; the
product enumerates the box.   python scripts/sim/row_codes.py [c3|c5|fuzz] [n]"""
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle

f32 = np.float32
T = 16


def row_intervals(x, y, A, B, C, op, minx, miny, maxx, maxy):
    """All arrays fp32 / int32 per splat.  Returns (use_rows bool[n], off int[n,4], wid int[n,4])."""
    with np.errstate(all="ignore"):
        w, h = maxx - minx, maxy - miny
        t = f32(255.0) * op
        bits = t.view(np.uint32)
        e = ((bits >> 23) & 0xff).astype(np.int32) - 127
        m = ((bits & 0x007fffff) | 0x3f800000).view(np.float32)
        tau = f32(1.3862944) * (e.astype(f32) + (m - f32(1.0)) + f32(0.0861))
        # an upper bound of every threshold the predicate can apply to a tile of the box
        DX = np.maximum(np.abs(f32(T) * minx.astype(f32) - x), np.abs(f32(T) * maxx.astype(f32) - x))
        DY = np.maximum(np.abs(f32(T) * miny.astype(f32) - y), np.abs(f32(T) * maxy.astype(f32) - y))
        M = A * DX * DX + f32(2.0) * np.abs(B) * DX * DY + C * DY * DY
        tau2 = f32(1.001) * (tau + f32(0.00001) * M + f32(0.01)) + f32(0.001)
        rA = f32(1.0) / A
        bA = B * rA                                        # centre line: dx = -bA dy
        k = C - B * bA                                     # q = A (dx + bA dy)^2 + k dy^2
        det = A * k
        exg = np.sqrt(tau2 * C / det)                      # half extents of {q <= tau2}
        eyg = np.sqrt(tau2 * rA * A / k) if False else np.sqrt(tau2 / k)
        dyR = -(B / C) * exg                               # height of the rightmost point; the leftmost sits at -dyR
        use = (h >= 2) & (h <= 4) & (w <= 15) & (w >= 1) & (A > 0) & (C > 0) & (k > 0) & np.isfinite(exg) & np.isfinite(eyg) \
            & np.isfinite(tau2) & np.isfinite(dyR) & (op >= f32(1.0 / 255.0))
        off = np.zeros((len(x), 4), np.int32); wid = np.zeros((len(x), 4), np.int32)

        def edge(dy):
            h2 = np.maximum((tau2 - k * dy * dy) * rA, f32(0.0))
            hw = np.sqrt(h2)
            c = -(bA * dy)
            return c - hw, c + hw

        INF = f32(np.inf)
        for r in range(4):
            ty = miny + r
            lo_b = np.where(r == 0, -INF, f32(T) * ty.astype(f32) - f32(0.5) - y)                   # continuous band of the row's pixel
            hi_b = np.where(r == h - 1, INF, f32(T) * (ty + 1).astype(f32) - f32(0.5) - y)         # centres, open at the box's ends
            empty = (lo_b > eyg) | (hi_b < -eyg) | (r >= h)
            e0 = np.maximum(lo_b, -eyg); e1 = np.minimum(hi_b, eyg)
            g0, f0 = edge(e0); g1, f1 = edge(e1)
            hi = np.where((dyR >= e0) & (dyR <= e1), exg, np.where(dyR < e0, f0, f1))
            lo = np.where((-dyR >= e0) & (-dyR <= e1), -exg, np.where(-dyR < e0, g0, g1))
            hi = hi + f32(0.001) * np.abs(hi) + f32(1.0)
            lo = lo - f32(0.001) * np.abs(lo) - f32(1.0)
            t_lo = np.floor((x + lo - f32(T - 1)) * f32(1.0 / T)).astype(np.int64)               # first tile whose span reaches lo
            t_hi = np.floor((x + hi) * f32(1.0 / T)).astype(np.int64)
            t_lo = np.clip(t_lo, minx, maxx); t_hi = np.clip(t_hi + 1, minx, maxx)
            ok = ~empty & np.isfinite(lo) & np.isfinite(hi)
            bad = ~empty & ~ok
            use &= ~bad
            off[:, r] = np.where(ok, t_lo - minx, 0)
            wid[:, r] = np.where(ok, np.maximum(t_hi - t_lo, 0), 0)
        return use, off, wid


def check(o, width, height, label):
    xy, co, radii = o["xy"], o["conic_opacity"], o["radii"]
    gx, gy = (width + T - 1) // T, (height + T - 1) // T
    n = len(radii)
    # the candidate rectangle as the preprocess builds it (3-sigma rectangle clipped to the ellipse box)
    x, y = xy[:, 0].astype(f32), xy[:, 1].astype(f32)
    A, B, C, op = (co[:, k].astype(f32) for k in range(4))
    rad = radii.astype(np.int64)
    with np.errstate(all="ignore"):
        rminx = np.clip(((x - rad) / T).astype(np.int64), 0, gx); rmaxx = np.clip(((x + rad + T - 1) / T).astype(np.int64), 0, gx)
        rminy = np.clip(((y - rad) / T).astype(np.int64), 0, gy); rmaxy = np.clip(((y + rad + T - 1) / T).astype(np.int64), 0, gy)
    vis = (radii > 0) & (rmaxx > rminx) & (rmaxy > rminy)
    # listed instances from the oracle's tight lists
    ranges, gs = o["ranges"], o["gauss_sorted"]
    tiles = np.repeat(np.arange(len(ranges)), (ranges[:, 1] - ranges[:, 0]).astype(np.int64))
    g = gs[: len(tiles)].astype(np.int64)
    tx, ty = tiles % gx, tiles // gx
    # the kernel's candidate rectangle (preprocess.hip.h candidate_rect): the 3-sigma rectangle clipped to the ellipse's box
    with np.errstate(all="ignore"):
        tt = f32(255.0) * op
        bits = tt.view(np.uint32)
        ee = (((bits >> 23) & 0xff).astype(np.int32) - 127).astype(f32)
        mm = ((bits & 0x007fffff) | 0x3f800000).view(np.float32)
        tau = f32(1.3862944) * (ee + (mm - f32(1.0)) + f32(0.0861))
        D = rad.astype(f32) + f32(2 * T)
        Mx = (A + f32(2.0) * np.abs(B) + C) * D * D
        tau2 = f32(1.001) * (tau + f32(0.00001) * Mx + f32(0.01)) + f32(0.001)
        det = A * C - B * B
        cxx, cyy = C / det, A / det                         # (the preprocess has the covariance itself; its inverse here)
        ex = np.sqrt(tau2 * cxx) * f32(1.001) + f32(1.0); ey = np.sqrt(tau2 * cyy) * f32(1.001) + f32(1.0)
        okc = (A > 0) & (C > 0) & (ex < 1e9) & (ey < 1e9)
        cminx = np.where(okc, np.maximum(rminx, np.floor((x - ex - f32(T - 1)) / f32(T)).astype(np.int64)), rminx)
        cminy = np.where(okc, np.maximum(rminy, np.floor((y - ey - f32(T - 1)) / f32(T)).astype(np.int64)), rminy)
        cmaxx = np.where(okc, np.minimum(rmaxx, np.floor((x + ex) / f32(T)).astype(np.int64) + 1), rmaxx)
        cmaxy = np.where(okc, np.minimum(rmaxy, np.floor((y + ey) / f32(T)).astype(np.int64) + 1), rmaxy)
    vis &= (cmaxx > cminx) & (cmaxy > cminy) & (op >= f32(1 / 255))
    rminx, rminy, rmaxx, rmaxy = cminx, cminy, cmaxx, cmaxy
    use, off, wid = row_intervals(x, y, A, B, C, op, rminx.astype(np.int32), rminy.astype(np.int32), rmaxx.astype(np.int32), rmaxy.astype(np.int32))
    r = ty - rminy[g]
    inside_box = (tx >= rminx[g]) & (tx < rmaxx[g]) & (r >= 0) & (r < (rmaxy - rminy)[g])
    assert inside_box.all(), "listed instance outside the candidate rectangle?"
    rr = np.clip(r, 0, 3)
    in_row = (tx - rminx[g] >= off[g, rr]) & (tx - rminx[g] < off[g, rr] + wid[g, rr])
    viol = use[g] & ~in_row
    box = ((rmaxx - rminx) * (rmaxy - rminy))[vis].sum()
    rows = np.where(use, wid.sum(1), (rmaxx - rminx) * (rmaxy - rminy))[vis].sum()
    print(f"{label}: {len(g)} listed instances, {int(viol.sum())} outside their row intervals; candidates box {box / 1e6:.3f} M -> "
          f"rows {rows / 1e6:.3f} M ({use[vis].mean():.3f} of the visible splats in row mode)")
    return int(viol.sum())


if __name__ == "__main__":
    oracle.build()
    what = sys.argv[1] if len(sys.argv) > 1 else "c3"
    bad = 0
    if what == "fuzz":
        import test_fuzz_parity as TF
        for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
            act, view, deg, mod, bg = TF._case(seed)
            o = oracle.forward(**act, sh_degree=deg, **view.raster_kwargs(bg), num_threads=8, scale_modifier=mod, cull_mode=1)
            bad += check(o, view.width, view.height, f"seed {seed}")
    else:
        from pegasus_amd import scenes
        cloud, views = (scenes.scene_c5 if what == "c5" else scenes.scene_c3)(n_views=4)
        for vi in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
            v = views[vi]
            o = oracle.forward(**cloud.activated(), sh_degree=3, **v.raster_kwargs(), num_threads=32, cull_mode=1)
            bad += check(o, v.width, v.height, f"{what} view {vi}")
    print("violations:", bad)
