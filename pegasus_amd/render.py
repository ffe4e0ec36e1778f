"""PEGASUS's four render wrappers (/root/reference/src/gs/render.py), same names, arguments and return
types, on top of this build's ``render()``.  The mask arithmetic runs on the device (pgr_color_masks) and only
the finished masks are copied to the host, where the reference computes them with numpy on full float images."""
from __future__ import annotations

import copy

import numpy as np
import torch

from . import masks as M
from .gaussian_renderer import render
from .sh_utils import RGB2SH


def _to_host(t: torch.Tensor) -> torch.Tensor:
    """``t.cpu()`` through page-locked memory: the reference fetches every image with .cpu() (render.py:19-20), and a
    pageable destination halves the PCIe rate (10 MB of RGB + depth per frame, 41 MB of float64 masks).  The pinned block
    comes from torch's caching host allocator and returns to it when the caller drops the tensor."""
    if t.device.type != "cuda" or t.requires_grad:
        return t.cpu()
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t)
    return out


def render_rgb_and_depth(cam, gs_scene, pipe_settings, bg, debug=False):
    render_pkg = render(cam, gs_scene, pipe_settings, bg)
    rgb_image = _to_host(render_pkg["render"]).permute((1, 2, 0))
    depth_image = _to_host(render_pkg["depth"]).permute((1, 2, 0))
    return rgb_image, depth_image


def _restore_semantics(obj):
    obj._features_dc[:] = obj._features_dc_semantics
    obj._features_rest[:, :] = obj._features_rest_semantics


def _without_environment(scene, n_env):
    mask_env = torch.ones(scene._xyz.shape[0], dtype=bool, device=scene._xyz.device)
    mask_env[:n_env] = False
    scene.mask_points(mask_env)
    return scene


def render_silhouette_mask(cam, gs_object_list, gs_env, width, height, color_set, pipe_settings, bg):
    """Every object rendered ALONE in its semantic colour over an empty environment and thresholded (render.py:36-65: one
    deepcopy + merge + render per object).  Here: ONE layered rasterizer call for all objects (pgr_forward_layers_async:
    per-(tile, object) lists, the compositor's epilogue thresholds layer k against colour k) over the objects-only scene
    the other two semantic wrappers share -- bit-equal to the per-object form below, which stays as the path for the
    settings the layered call does not take (python-side covariances / SH, autograd, object ids that do not ascend)."""
    ids = list(gs_object_list.keys())
    plain = (not torch.is_grad_enabled() and not getattr(pipe_settings, "compute_cov3D_python", False)
             and not getattr(pipe_settings, "convert_SHs_python", False) and ids == sorted(ids) and len(ids) > 0
             and min(ids) >= 1 and max(ids) <= int(color_set.shape[0]))
    if not plain:
        return _render_silhouette_mask_per_object(cam, gs_object_list, gs_env, width, height, color_set, pipe_settings, bg)
    import math
    from . import rasterizer as R
    from .gaussian_renderer import kept_activation as _kept
    scene, key = _semantic_scene(gs_env, gs_object_list)
    dev = scene._xyz.device
    K = int(color_set.shape[0])
    layer_id = _kept_scene.get("layer_id") if key is not None else None
    if layer_id is None:
        layer_id = torch.cat([torch.full((int(gs_object_list[i]._xyz.shape[0]),), int(i), dtype=torch.int32, device=dev)
                              for i in ids])
        if key is not None:
            _kept_scene["layer_id"] = layer_id
    spec = R.ViewSpec(int(cam.image_height), int(cam.image_width), math.tan(0.5 * cam.FoVx), math.tan(0.5 * cam.FoVy), bg,
                      cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
    out = torch.empty((K, int(cam.image_height), int(cam.image_width)), dtype=torch.uint8, device=dev)
    colors = torch.as_tensor(color_set, device=dev).float().contiguous()
    h = R.forward_views(scene.get_xyz, _kept(scene, "get_opacity"), [spec], shs=_kept(scene, "get_features"),
                        scales=_kept(scene, "get_scaling"), rotations=_kept(scene, "get_rotation"),
                        sh_degree=int(scene.active_sh_degree), want_radii=False, outputs=[dict(radii=None, sem_masks=out)],
                        async_slot=("drop-in-silhouette", 0),
                        layers=dict(layer_id=layer_id, n_layers=K, mask_colors=colors, mask_threshold=M.MASK_THRESHOLD))
    h.wait()
    return _to_host(out.permute(1, 2, 0).to(torch.float64)).numpy()


def _render_silhouette_mask_per_object(cam, gs_object_list, gs_env, width, height, color_set, pipe_settings, bg):
    """One single-object render per object over an EMPTY environment (render.py:36-65), as the reference does it."""
    black = _without_environment(copy.deepcopy(gs_env), gs_env._xyz.shape[0])
    mask_silhouette = np.zeros((height, width, color_set.shape[0]))
    for gs_object_id, current in gs_object_list.items():
        scene = copy.deepcopy(black)
        _restore_semantics(current)
        scene.merge_gaussians(gaussian=current)
        img = render(cam, scene, pipe_settings, bg)["render"]
        m = M.color_masks(img, color_set[gs_object_id - 1].reshape(1, 3), M.MASK_THRESHOLD)
        mask_silhouette[m[0].cpu().numpy().astype(bool), gs_object_id - 1] = 1
    return mask_silhouette


_GEOMETRY = ("_xyz", "_opacity", "_scaling", "_rotation")
_ALL = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
_kept_scene = {}       # the objects-only scene of the last call and, per camera, its render


def reset_cache():
    """Forgets the kept objects-only scene and its last render (the next semantic wrapper call rebuilds both)."""
    _kept_scene.clear()


def swap_cache(state=None):
    """Replaces the kept objects-only scene by ``state`` (None: empty) and returns what was kept before -- lets a caller
    run a cold render beside a warm slot and put the slot back."""
    before = dict(_kept_scene)
    _kept_scene.clear()
    if state:
        _kept_scene.update(state)
    return before


def _fingerprint(tensors):
    """Identity of a set of tensors as far as torch can tell without reading them: object, version counter, storage,
    shape.  None (= never equal) for tensors without a version counter (torch.inference_mode)."""
    try:
        return tuple((id(t), t._version, t.data_ptr(), tuple(t.shape)) for t in tensors)
    except RuntimeError:
        return None


def _semantic_scene(gs_environment, gs_object_list):
    """The scene the reference composes for its two semantic renders (/root/reference/src/gs/render.py:68-84,100-113):
    deepcopy the environment, paint every object in its semantic colour (IN PLACE -- the caller's objects keep the paint,
    as in the reference), merge them, then mask the environment rows out again.  What is left is the objects alone in
    dictionary order, so it is built as exactly that: one concatenation per attribute, no 2 M-row copy that is thrown away.
    The scene of an unchanged object list is kept between calls (both semantic wrappers ask for the same one per frame)."""
    objects = list(gs_object_list.values())
    for current in objects:
        _restore_semantics(current)
    # the tensors the key is made of.  The slot KEEPS them (below): a key built from id() / data_ptr() is only meaningful
    # while those objects are alive -- apply_transformation replaces _xyz / _rotation with fresh tensors on every pose
    # update, and freed ids and allocator blocks are handed out again, so that after two pose updates the old key could
    # match tensors that merely reuse the addresses (round-3 advisor finding: stale poses in the masks)
    keyed = [getattr(o, k) for o in objects for k in _GEOMETRY] + \
            [t for o in objects for t in (o._features_dc_semantics, o._features_rest_semantics) if isinstance(t, torch.Tensor)]
    key = _fingerprint(keyed)
    key = None if key is None else (key, tuple(id(o) for o in objects), id(gs_environment),
                                    getattr(gs_environment, "active_sh_degree", None))
    if key is not None and _kept_scene.get("key") == key:
        return _kept_scene["scene"], key
    scene = copy.copy(gs_environment)                   # the environment's settings (SH degrees, ...), none of its rows
    for k in _ALL:
        parts = [getattr(o, k) for o in objects]
        setattr(scene, k, torch.cat(parts, 0) if parts else getattr(gs_environment, k)[:0])
    _kept_scene.clear()
    _kept_scene.update(key=key, scene=scene, renders={}, keep=(objects, keyed))
    return scene, key


def _semantic_image(cam, gs_environment, gs_object_list, pipe_settings, bg):
    """Objects-only semantic render of ``cam``.  The reference renders this same image twice per frame, once in
    render_visib_mask and once in render_semanticsegmentation_mask; the second request of an unchanged (scene, camera,
    background) gets the first one's image."""
    scene, key = _semantic_scene(gs_environment, gs_object_list)
    cam_key = None
    if key is not None:
        cam_key = _fingerprint([cam.world_view_transform, cam.full_proj_transform, cam.camera_center, bg])
        cam_key = None if cam_key is None else (cam_key, id(cam), int(cam.image_height), int(cam.image_width),
                                                float(cam.FoVx), float(cam.FoVy), torch.is_grad_enabled(),
                                                bool(getattr(pipe_settings, "convert_SHs_python", False)),
                                                bool(getattr(pipe_settings, "compute_cov3D_python", False)))
        hit = _kept_scene["renders"].get("last")
        if cam_key is not None and hit is not None and hit[0] == cam_key:
            return hit[1]
    img = render(cam, scene, pipe_settings, bg)["render"]
    if cam_key is not None and not torch.is_grad_enabled():
        _kept_scene["renders"]["last"] = (cam_key, img, (cam, bg))
    return img


def render_visib_mask(cam, gs_environment, gs_object_list, color_set, height, width, pipe_settings, bg):
    """All objects in their semantic colours, environment removed; K masks by colour distance (render.py:68-97)."""
    img = _semantic_image(cam, gs_environment, gs_object_list, pipe_settings, bg)
    m = M.color_masks(img, color_set, M.MASK_THRESHOLD)
    # float64 [H,W,K] on the host, as the reference returns it (np.zeros default dtype); widened on the device: the host
    # cast of 41 MB costs more than moving it
    individual = _to_host(m.permute(1, 2, 0).to(torch.float64)).numpy()
    return individual, _to_host(img).permute((1, 2, 0))


def render_semanticsegmentation_mask(cam, gs_environment, gs_object_list, color_set, height, width, pipe_settings,
                                     bg, debug):
    img = _semantic_image(cam, gs_environment, gs_object_list, pipe_settings, bg)
    rgb8 = M.pack_frames(color=img[None])["rgb"][0]            # (img * 255).astype(uint8) on the device, HWC
    return _to_host(rgb8).numpy()


def assign_semantic_colors(gaussians_object_list, semantic_colors):
    """What PEGASUS.init_start_position does to every object (pegasus.py:223-232)."""
    for obj_id, obj in gaussians_object_list.items():
        obj._features_dc_color = copy.deepcopy(obj._features_dc)
        obj._features_rest_color = copy.deepcopy(obj._features_rest)
        c = torch.as_tensor(semantic_colors[obj_id - 1]).to(obj._features_dc.device)
        obj._features_dc_semantics = RGB2SH(c)
        obj._features_rest_semantics = torch.asarray([0, 0, 0], device=obj._features_dc.device)
