"""``gaussian_renderer.render`` counterpart: the facade every PEGASUS render path goes through
(/root/reference/src/gs/render.py:16,57,86,118; /root/reference/pegasus.py:271;
/root/reference/src/gs/gs_viewer.py:66; /root/reference/src/visualization/object_visualization.py:623).

Contract fixed by its callers: returns a dict with "render" [3,H,W], "depth" [1,H,W],
"viewspace_points", "visibility_filter" (radii > 0), "radii" (src/gs/render.py:16-17,
src/gs/dev/gs_manipulation_static.py:311-315)."""
from __future__ import annotations

import math
import weakref

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from .sh_utils import eval_sh


# Activated parameters of a model, kept between calls while the model does not change.  PEGASUS renders one scene from
# many cameras, one `render()` call each (pegasus.py:254-271, src/gs/render.py); the getters behind `pc.get_opacity`,
# `get_scaling`, `get_rotation` and `get_features` are four elementwise kernels and a 192-byte-per-Gaussian `cat`
#  -- 0.36 ms of a 0.93 ms call on the 2 M-Gaussian scene.  Keyed on the identity and version counter of
# the raw tensors: any in-place edit or re-assignment (pose application, merge, mask) misses.  Inference only.
_ACT_CACHE = weakref.WeakKeyDictionary()
_RAW = {"get_opacity": ("_opacity",), "get_scaling": ("_scaling",), "get_rotation": ("_rotation",),
        "get_features": ("_features_dc", "_features_rest")}


def _activated(pc, getter: str):
    raw = [getattr(pc, a, None) for a in _RAW[getter]]
    if torch.is_grad_enabled() or any(not isinstance(t, torch.Tensor) for t in raw):
        return getattr(pc, getter)
    key = tuple((id(t), t._version, t.data_ptr(), tuple(t.shape)) for t in raw)
    try:
        slot = _ACT_CACHE.setdefault(pc, {})
    except TypeError:                      # a model class that cannot be weakly referenced: no cache
        return getattr(pc, getter)
    hit = slot.get(getter)
    if hit is not None and hit[0] == key:
        return hit[1]
    value = getattr(pc, getter)
    slot[getter] = (key, value, raw)       # `raw` keeps the keyed tensors alive: their ids and addresses cannot be reused
    return value


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene ``pc`` (a GaussianModel) from ``viewpoint_camera``.  ``bg_color`` must be on the GPU."""
    # upstream's idiom: a zero tensor whose .grad receives the screen-space (NDC-scaled) gradient of the 2D means, which
    # training-style callers read for densification (render_pkg["viewspace_points"].grad)
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True,
                                          device=pc.get_xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False,
        debug=getattr(pipe, "debug", False))
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = pc.get_xyz
    means2D = screenspace_points
    opacity = _activated(pc, "get_opacity")

    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = _activated(pc, "get_scaling")
        rotations = _activated(pc, "get_rotation")

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = _activated(pc, "get_features")
    else:
        colors_precomp = override_color

    rendered_image, radii, depth = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)

    return {"render": rendered_image, "depth": depth, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}
