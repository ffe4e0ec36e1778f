"""``gaussian_renderer.render``: the one function every PEGASUS render path goes through
(/root/reference/src/gs/render.py:16,57,86,118; /root/reference/pegasus.py:271;
/root/reference/src/gs/gs_viewer.py:66; /root/reference/src/visualization/object_visualization.py:623).

Its contract is fixed by those callers: ``render(camera, model, pipe, bg, scaling_modifier=1.0, override_color=None)``
returns a dict with "render" [3,H,W], "depth" [1,H,W], "viewspace_points", "visibility_filter" (radii > 0) and "radii"
(/root/reference/src/gs/render.py:16-17, /root/reference/src/gs/dev/gs_manipulation_static.py:311-315).  The body below
is this build's own: three small input builders (view, geometry, colour) in front of the drop-in rasterizer, with the
model's activated parameters kept between calls of an unchanged model.
"""
from __future__ import annotations

import math
import weakref

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians
from .sh_utils import sh_basis

# ---- activated parameters of a model, kept while the model does not change ------------------------------------------------
# PEGASUS renders one scene from many cameras with one render() call each (pegasus.py:254-271); the getters behind
# get_opacity / get_scaling / get_rotation / get_features are four elementwise kernels and a 192-byte-per-Gaussian
# concatenation: 0.36 ms of a 0.93 ms call on the 2 M-Gaussian scene.  Inference only.
#
# A slot is valid while every raw tensor behind the getter is the same object (id), has the same autograd version counter
# and the same storage address and shape.  The slot holds an alias of each raw tensor's storage, so neither the object id
# nor the address can be recycled for something else while the slot exists.  What the key CANNOT see is a write that
# bypasses the version counter (`param.data.add_(..)`, raw pointer writes from another library): after such an edit call
# ``invalidate_activations(model)``.  Tensors without a version counter (created under torch.inference_mode()) are simply
# not cached.
SPLIT_RENDERS = 4        # the render of an unchanged model from which on its SH coefficients are concatenated (before: read as stored)
_SLOTS = weakref.WeakKeyDictionary()
_ZERO_PROBES = {}        # no-grad "viewspace_points": one zero tensor per (shape, dtype, device)
_SOURCES = {"get_opacity": ("_opacity",), "get_scaling": ("_scaling",), "get_rotation": ("_rotation",),
            "get_features": ("_features_dc", "_features_rest")}


def invalidate_activations(pc=None) -> None:
    """Forget the kept activations of ``pc`` (or of every model).  Needed only after edits the version counter does not
    see, e.g. writes through ``tensor.data``."""
    if pc is None:
        _SLOTS.clear()
    else:
        _SLOTS.pop(pc, None)


def kept_activation(pc, getter: str):
    """``getattr(pc, getter)`` (get_opacity / get_scaling / get_rotation / get_features), kept between calls while the raw
    tensors behind it are unchanged (inference only; see the comment above)."""
    sources = [getattr(pc, name, None) for name in _SOURCES[getter]]
    if torch.is_grad_enabled() or not all(isinstance(t, torch.Tensor) for t in sources):
        return getattr(pc, getter)
    try:
        key = tuple((id(t), t._version, t.data_ptr(), tuple(t.shape)) for t in sources)
        per_model = _SLOTS.setdefault(pc, {})
    except (RuntimeError, TypeError):      # inference tensors have no version counter; some classes take no weak reference
        return getattr(pc, getter)
    slot = per_model.get(getter)
    if slot is None or slot[0] != key:
        slot = per_model[getter] = (key, getattr(pc, getter), sources, [t.detach() for t in sources])
    return slot[1]


# ---- the three input groups of a rasterizer call ---------------------------------------------------------------------------

def _camera_matrices(cam):
    """(viewmatrix, projmatrix, campos) as the kernels read them: fp32, contiguous.  The reference's Camera keeps
    ``world_view_transform`` as a TRANSPOSED VIEW (`torch.tensor(..).transpose(0, 1).cuda()`), so every render() paid a
    copy kernel + launch for it (two per call in the round-4 trace); the contiguous copies are kept on the camera while the
    three tensors are the same objects with the same version counters (inference only, like the model's activations)."""
    src = (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
    if torch.is_grad_enabled():
        return src
    try:
        key = tuple((id(t), t._version, t.data_ptr(), tuple(t.shape), t.stride()) for t in src)
        slot = getattr(cam, "_pgr_matrices", None)
        if slot is None or slot[0] != key:
            slot = (key, tuple(t.detach().float().contiguous() for t in src), src)
            cam._pgr_matrices = slot
        return slot[1]
    except (RuntimeError, AttributeError, TypeError):      # inference tensors, objects that take no attribute
        return src


def _view(cam, pc, pipe, bg, scaling_modifier) -> GaussianRasterizationSettings:
    vm, pm, cp = _camera_matrices(cam)
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(0.5 * cam.FoVx), tanfovy=math.tan(0.5 * cam.FoVy),
        bg=bg, scale_modifier=scaling_modifier, viewmatrix=vm, projmatrix=pm, campos=cp,
        sh_degree=pc.active_sh_degree, prefiltered=False, debug=bool(getattr(pipe, "debug", False)))


def _geometry(pc, pipe, scaling_modifier) -> dict:
    """Either the model's own 3D covariances (pipe.compute_cov3D_python) or scale + rotation for the kernel to combine."""
    if getattr(pipe, "compute_cov3D_python", False):
        return dict(cov3D_precomp=pc.get_covariance(scaling_modifier))
    return dict(scales=kept_activation(pc, "get_scaling"), rotations=kept_activation(pc, "get_rotation"))


def _colour(pc, pipe, cam, override_color) -> dict:
    """Precomputed per-Gaussian colours (an override, or SH evaluated here when pipe.convert_SHs_python asks for it), else
    the SH coefficients for the kernel to evaluate."""
    if override_color is not None:
        return dict(colors_precomp=override_color)
    if not getattr(pipe, "convert_SHs_python", False):
        # inference: the coefficients where the model stores them (PgrScene::shs_rest) -- get_features is a torch.cat of
        # _features_dc and _features_rest (/root/reference/src/gs/gaussian_model.py:118-121), 768 MB moved per render() of a
        # freshly merged 2 M-Gaussian scene, which PEGASUS's loop builds anew for every frame (pegasus.py:255-264)
        # A model that KEEPS being rendered unchanged (a viewer, a camera sweep over one scene) does get the concatenation,
        # once, at its SPLIT_RENDERS-th render: rows of 192 bytes sit in their cache lines, the stored rows of 180 + 12 bytes
        # straddle them -- 9 us of a 0.44 ms call on the 2 M-Gaussian scene (same box: preprocess 92 vs 83 us) -- and the
        # 0.25 ms copy pays for itself after some thirty calls.
        if not torch.is_grad_enabled():
            dc, rest = getattr(pc, "_features_dc", None), getattr(pc, "_features_rest", None)
            if (isinstance(dc, torch.Tensor) and isinstance(rest, torch.Tensor) and dc.dim() == 3 and rest.dim() == 3 and
                    dc.shape[1] == 1 and dc.shape[2] == 3 and rest.shape[2] == 3 and rest.shape[0] == dc.shape[0] and
                    dc.dtype == torch.float32 and rest.dtype == torch.float32 and dc.is_cuda and rest.device == dc.device and
                    dc.is_contiguous() and rest.is_contiguous()):
                try:
                    key = tuple((id(t), t._version, t.data_ptr(), tuple(t.shape)) for t in (dc, rest))
                    per_model = _SLOTS.setdefault(pc, {})
                except (RuntimeError, TypeError):      # no version counter / no weak reference: nothing to count with
                    return dict(shs=dc, shs_rest=rest)
                kept = per_model.get("get_features")
                if kept is not None and kept[0] == key:
                    return dict(shs=kept[1])                       # the concatenation exists: use it
                seen = per_model.get("split_renders")
                count = seen[1] + 1 if seen is not None and seen[0] == key else 1
                per_model["split_renders"] = (key, count, [dc.detach(), rest.detach()])     # (aliases keep ids / addresses ours)
                if count < SPLIT_RENDERS:
                    return dict(shs=dc, shs_rest=rest)
        return dict(shs=kept_activation(pc, "get_features"))
    coeff = pc.get_features                                        # [N, (max_deg+1)^2, 3]
    towards = torch.nn.functional.normalize(pc.get_xyz - cam.camera_center.reshape(1, 3), dim=1)
    basis = sh_basis(pc.active_sh_degree, towards)                 # [N, (deg+1)^2]
    rgb = torch.einsum("nk,nkc->nc", basis, coeff[:, :basis.shape[1]])
    return dict(colors_precomp=(rgb + 0.5).clamp_min(0.0))


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene ``pc`` (a GaussianModel) from ``viewpoint_camera``.  ``bg_color`` must be on the GPU."""
    xyz = pc.get_xyz
    # "viewspace_points": a zero tensor shaped like the means whose .grad receives the screen-space (NDC-scaled) gradient
    # of the 2D means -- training-style callers read it for densification.  A leaf that asks for a gradient only when
    # autograd is on; the render loops run under torch.no_grad() (pegasus.py:248) and get plain zeros -- ONE zero tensor
    # per shape and device, not a 24 MB fill per call (nothing writes it without autograd).
    if torch.is_grad_enabled():
        probe = torch.zeros_like(xyz, requires_grad=True)
    else:
        pkey = (tuple(xyz.shape), xyz.dtype, xyz.device)
        probe = _ZERO_PROBES.get(pkey)
        if probe is None:
            if len(_ZERO_PROBES) >= 4:                             # PEGASUS alternates between the merged scene and the objects-only
                _ZERO_PROBES.pop(next(iter(_ZERO_PROBES)))         # scene: a few shapes stay, the oldest goes
            probe = _ZERO_PROBES[pkey] = torch.zeros_like(xyz)
    settings = _view(viewpoint_camera, pc, pipe, bg_color, scaling_modifier)
    inputs = dict(means3D=xyz, means2D=probe, opacities=kept_activation(pc, "get_opacity"),
                  **_colour(pc, pipe, viewpoint_camera, override_color), **_geometry(pc, pipe, scaling_modifier))
    if torch.is_grad_enabled():
        image, radii, depth = GaussianRasterizer(raster_settings=settings)(**inputs)
    else:
        # the render loops (torch.no_grad(), pegasus.py:248): straight to the forward, without building an nn.Module per call
        # and going through its __call__ machinery (0.571 -> 0.553 ms per call on the 2 M-Gaussian scene: a single view
        # leaves the GPU idle between its kernels, so host microseconds are wall-clock microseconds)
        # "visibility_filter" is queued behind the compositor before the host waits for the call's status
        extra = {}
        image, radii, depth = rasterize_gaussians(
            inputs["means3D"], inputs["means2D"], inputs.get("shs"), inputs.get("colors_precomp"), inputs["opacities"],
            inputs.get("scales"), inputs.get("rotations"), inputs.get("cov3D_precomp"), settings,
            after_enqueue=lambda r: extra.__setitem__("vis", r["radii"] > 0), sh_rest=inputs.get("shs_rest"))
        return {"render": image, "depth": depth, "viewspace_points": probe, "visibility_filter": extra["vis"], "radii": radii}
    return {"render": image, "depth": depth, "viewspace_points": probe, "visibility_filter": radii > 0, "radii": radii}
