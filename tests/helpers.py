"""Shared helpers for the parity tests: run the HIP path through the drop-in surface and fetch
its intermediates through the C ABI (pgr_workspace_view)."""
import ctypes as C

import numpy as np
import torch

from pegasus_amd import _lib
from pegasus_amd import diff_gaussian_rasterization as dgr


def gpu_forward(act: dict, view, sh_degree=3, bg=(0.0, 0.0, 0.0), device="cuda:0", scale_modifier=1.0,
                colors_precomp=None, cov3d_precomp=None, fetch_intermediates=True):
    """act: activated numpy arrays (means3d, opacities, scales, rotations, shs).  Returns numpy dict with the
    same keys the oracle binding uses."""
    dev = torch.device(device)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    settings = dgr.GaussianRasterizationSettings(
        image_height=view.height, image_width=view.width, tanfovx=view.tanfovx, tanfovy=view.tanfovy,
        bg=t(np.asarray(bg, np.float32)), scale_modifier=scale_modifier,
        viewmatrix=t(view.world_view_transform), projmatrix=t(view.full_proj_transform), sh_degree=sh_degree,
        campos=t(view.camera_center), prefiltered=False, debug=False)
    means = t(act["means3d"])
    shs = None if colors_precomp is not None else t(act["shs"])
    color, radii, depth, final_T, n_contrib = dgr.rasterize_gaussians(
        means, None, shs, t(colors_precomp), t(act["opacities"]),
        None if cov3d_precomp is not None else t(act["scales"]),
        None if cov3d_precomp is not None else t(act["rotations"]), t(cov3d_precomp), settings, want_aux=True)
    torch.cuda.synchronize()
    r = dict(color=color.cpu().numpy(), out_depth=depth.cpu().numpy(), radii=radii.cpu().numpy(),
             final_T=final_T.cpu().numpy(), n_contrib=n_contrib.cpu().numpy().astype(np.uint32))
    info = dgr.last_forward_info()
    r["num_instances"] = info["num_instances"][0]
    n = means.shape[0]
    if fetch_intermediates and n > 0:
        r.update(fetch_workspace(0, n, view.width, view.height))
    return r


def fetch_workspace(view_index, n, width, height):
    """Copies one view's intermediate arrays out of the last forward's workspace (via pgr_workspace_view)."""
    from pegasus_amd import rasterizer
    info = rasterizer.last_forward_info()
    ws = info["workspace"]
    v = rasterizer.workspace_view(view_index)
    I = info["num_instances"][view_index]
    tiles = ((width + 15) // 16) * ((height + 15) // 16)
    base = ws.data_ptr()

    def grab(ptr, count, dtype):
        nbytes = count * np.dtype(dtype).itemsize
        off = ptr - base
        return ws[off:off + nbytes].cpu().numpy().view(dtype).copy()
    r = {}
    rec = grab(v["splats"], 12 * n, np.float32).reshape(n, 12)
    r["xy"] = np.ascontiguousarray(rec[:, 0:2])
    r["conic_opacity"] = np.ascontiguousarray(rec[:, 2:6])
    r["rgb4"] = np.ascontiguousarray(rec[:, 8:12])        # record: x, y, A, B, C, opacity, B/C, B/A, r, g, b, depth
    r["depth"] = np.ascontiguousarray(rec[:, 11])
    rects = grab(v["rects"], 4 * n, np.uint16).reshape(n, 4).astype(np.int32)
    r["rects"] = rects
    r["tiles_touched"] = (rects[:, 2] - rects[:, 0]) * (rects[:, 3] - rects[:, 1])
    r["gauss_sorted"] = grab(v["gauss_sorted"], I, np.uint32)
    r["ranges"] = grab(v["ranges"], 2 * tiles, np.uint32).reshape(tiles, 2)
    return r


def assert_preprocess_bit_exact(g, o):
    """Per-Gaussian stage: bit-exact for survivors; culled entries are don't-care except radii / tiles."""
    np.testing.assert_array_equal(g["radii"], o["radii"])
    np.testing.assert_array_equal(g["tiles_touched"], o["tiles_touched"])
    m = o["radii"] > 0
    for k in ("xy", "depth", "conic_opacity"):
        np.testing.assert_array_equal(g[k][m].view(np.uint32), o[k][m].view(np.uint32), err_msg=k)
    np.testing.assert_array_equal(g["rgb4"][m][:, :3].view(np.uint32), o["rgb"][m].view(np.uint32), err_msg="rgb")


def assert_images_match(g, o, tol=1e-4, max_ambig_frac=5e-4):
    """1e-4 is BASELINE.json's float tolerance.  Pixels the oracle flags as ambiguous (a threshold
    decision within rounding distance of flipping, because exp() is not bit-identical) are excluded
    and must be rare."""
    amb = o["ambig"].astype(bool)
    assert amb.mean() <= max_ambig_frac, f"too many ambiguous pixels: {amb.mean()}"
    ok = ~amb
    dc = np.abs(g["color"] - o["color"])[:, ok]
    dd = np.abs(g["out_depth"][0] - o["out_depth"][0])[ok]
    dt = np.abs(g["final_T"] - o["final_T"])[ok]
    assert dc.max(initial=0) <= tol, f"colour max err {dc.max()}"
    assert dd.max(initial=0) <= tol, f"depth max err {dd.max()}"
    assert dt.max(initial=0) <= tol, f"final_T max err {dt.max()}"
    np.testing.assert_array_equal(g["n_contrib"][ok], o["n_contrib"][ok])
    # The excluded pixels are not unconstrained: a flipped threshold decision adds or drops ONE entry -- alpha >= 1/255 with
    # alpha at the threshold (weight <= 1/255), or the entry at which T (1 - alpha) crosses 1e-4 (weight alpha T <= 0.0099) --
    # so even there the images may differ by at most ~0.01 x the entry's colour / depth.
    if amb.any():
        cmax = max(1.0, float(np.abs(o["color"]).max()))
        zmax = max(1.0, float(np.abs(o["out_depth"]).max()))
        ac = np.abs(g["color"] - o["color"])[:, amb].max()
        ad = np.abs(g["out_depth"][0] - o["out_depth"][0])[amb].max()
        at = np.abs(g["final_T"] - o["final_T"])[amb].max()
        assert ac <= 0.02 * cmax and ad <= 0.02 * zmax and at <= 0.02, f"flagged pixels differ by more than one entry: {ac} {ad} {at}"
