"""Dense float64 forward of the rasterizer for FINITE-DIFFERENCE gradient checks (TEST INFRASTRUCTURE).

No tiles, no lists: every Gaussian is evaluated at every pixel of the tiles its 3-sigma rectangle touches, in
(depth, index) order -- the same image function as oracle/pgr_oracle.c, in float64, for scenes of a few dozen
Gaussians.  Differences to the fp32 oracle are rounding only, except at the discontinuities of the model
(alpha thresholds, the T < 1e-4 stop, ceil() of the radius), which the gradient tests stay away from."""
import math

import numpy as np


def real_sh_basis(deg, d):
    """Real spherical harmonics Y_l^m(d), l = 0..deg, m = -l..l, of a unit direction d -- derived here from first
    principles in float64 (associated Legendre functions + the orthonormalisation factor from factorials), NOT from the
    product's constant table (pegasus_amd/sh_utils.py) or the oracle's (oracle/pgr_oracle.c): a wrong constant or sign in
    either shows up as a disagreement with this function (tests/test_oracle_kat.py).

        Y_l^0  = K_l^0 P_l^0(cos t)
        Y_l^m  = sqrt(2) K_l^m  cos(m p)  P_l^m(cos t)      m > 0
        Y_l^-m = sqrt(2) K_l^m  sin(m p)  P_l^m(cos t)      m > 0
        K_l^m  = sqrt((2 l + 1) / (4 pi) * (l - m)! / (l + m)!)

    with P_l^m INCLUDING the Condon-Shortley phase (-1)^m -- the convention of the 3DGS paper's colour model
    (band 1 = (-y, z, -x) * sqrt(3 / 4 pi)), cited at /root/reference/README.md:285-299."""
    x, y, z = (float(v) for v in d)
    ct = max(-1.0, min(1.0, z))
    st = math.sqrt(max(0.0, 1.0 - ct * ct))
    phi = math.atan2(y, x)
    # P_m^m = (-1)^m (2m-1)!! sin^m ; P_{m+1}^m = x (2m+1) P_m^m ; (l-m) P_l^m = x (2l-1) P_{l-1}^m - (l+m-1) P_{l-2}^m
    P = {}
    for m in range(deg + 1):
        pmm = 1.0
        for k in range(1, m + 1):
            pmm *= -(2 * k - 1) * st
        P[(m, m)] = pmm
        if m + 1 <= deg:
            P[(m + 1, m)] = ct * (2 * m + 1) * pmm
        for l in range(m + 2, deg + 1):
            P[(l, m)] = (ct * (2 * l - 1) * P[(l - 1, m)] - (l + m - 1) * P[(l - 2, m)]) / (l - m)
    out = []
    for l in range(deg + 1):
        for m in range(-l, l + 1):
            a = abs(m)
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - a) / math.factorial(l + a))
            if m == 0:
                out.append(K * P[(l, 0)])
            elif m > 0:
                out.append(math.sqrt(2.0) * K * math.cos(a * phi) * P[(l, a)])
            else:
                out.append(math.sqrt(2.0) * K * math.sin(a * phi) * P[(l, a)])
    return np.asarray(out, dtype=np.float64)


def quat_R(q):
    r, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                     [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                     [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])


def dense_forward(means3d, opacities, scales, rotations, shs, sh_degree, width, height, tanfovx, tanfovy, viewmatrix,
                  projmatrix, campos, bg, scale_modifier=1.0):
    f = lambda a: np.asarray(a, dtype=np.float64)
    means3d, opacities, scales, rotations, shs = f(means3d), f(opacities).reshape(-1), f(scales), f(rotations), f(shs)
    vm, pm, campos, bg = f(viewmatrix).reshape(4, 4), f(projmatrix).reshape(4, 4), f(campos), f(bg)
    W, H = width, height
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    n = means3d.shape[0]
    recs = []
    for i in range(n):
        p = np.append(means3d[i], 1.0)
        t = p @ vm[:, :3]                 # row-vector convention: vm is stored transposed
        if t[2] <= 0.2:
            continue
        h = p @ pm
        ndc = h[:2] / (h[3] + 1e-7)
        R = quat_R(rotations[i])
        M = R * (scale_modifier * scales[i])[None, :]
        S3 = M @ M.T
        cx = min(1.3 * tanfovx, max(-1.3 * tanfovx, t[0] / t[2])) * t[2]
        cy = min(1.3 * tanfovy, max(-1.3 * tanfovy, t[1] / t[2])) * t[2]
        J = np.array([[fx / t[2], 0, -fx * cx / t[2] ** 2], [0, fy / t[2], -fy * cy / t[2] ** 2]])
        Wm = vm[:3, :3].T                # world -> view rotation
        T = J @ Wm
        cov = T @ S3 @ T.T + 0.3 * np.eye(2)
        det = cov[0, 0] * cov[1, 1] - cov[0, 1] ** 2
        if det == 0:
            continue
        conic = np.array([cov[1, 1], -cov[0, 1], cov[0, 0]]) / det
        mid = 0.5 * (cov[0, 0] + cov[1, 1])
        lam = mid + math.sqrt(max(0.1, mid * mid - det))
        radius = math.ceil(3 * math.sqrt(lam))
        pix = np.array([((ndc[0] + 1) * W - 1) * 0.5, ((ndc[1] + 1) * H - 1) * 0.5])
        ct = lambda v, hi: 0 if not v > 0 else (hi if v >= hi else int(v))
        minx, miny = ct((pix[0] - radius) / 16, gx), ct((pix[1] - radius) / 16, gy)
        maxx, maxy = ct((pix[0] + radius + 15) / 16, gx), ct((pix[1] + radius + 15) / 16, gy)
        if maxx <= minx or maxy <= miny:
            continue
        d = means3d[i] - campos
        d = d / np.linalg.norm(d)
        b = real_sh_basis(sh_degree, d)
        rgb = np.maximum(b @ shs[i, :b.shape[0]] + 0.5, 0.0)
        recs.append((t[2], i, pix, conic, opacities[i], rgb, (minx, miny, maxx, maxy)))
    recs.sort(key=lambda r: (np.float32(r[0]), r[1]))
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    T = np.ones((H, W)); C = np.zeros((3, H, W)); D = np.zeros((H, W)); done = np.zeros((H, W), bool)
    for z, i, pix, conic, op, rgb, (minx, miny, maxx, maxy) in recs:
        inrect = (xs >= 16 * minx) & (xs < 16 * maxx) & (ys >= 16 * miny) & (ys < 16 * maxy)
        dx, dy = pix[0] - xs, pix[1] - ys
        power = -0.5 * (conic[0] * dx * dx + conic[2] * dy * dy) - conic[1] * dx * dy
        alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 0)))
        valid = inrect & ~done & (power <= 0) & (alpha >= 1 / 255)
        test_T = T * (1 - alpha)
        stop = valid & (test_T < 1e-4)
        done |= stop
        blend = valid & ~stop
        w = np.where(blend, alpha * T, 0.0)
        C += rgb[:, None, None] * w
        D += z * w
        T = np.where(blend, test_T, T)
    return C + T * bg[:, None, None], D
