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


def render_rgb_and_depth(cam, gs_scene, pipe_settings, bg, debug=False):
    render_pkg = render(cam, gs_scene, pipe_settings, bg)
    rgb_image = render_pkg["render"].cpu().permute((1, 2, 0))
    depth_image = render_pkg["depth"].cpu().permute((1, 2, 0))
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
    """One single-object render per object over an EMPTY environment (render.py:36-65)."""
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


def _semantic_scene(gs_environment, gs_object_list):
    scene = copy.deepcopy(gs_environment)
    for current in gs_object_list.values():
        _restore_semantics(current)
        scene.merge_gaussians(gaussian=current)
    return _without_environment(scene, gs_environment._xyz.shape[0])


def render_visib_mask(cam, gs_environment, gs_object_list, color_set, height, width, pipe_settings, bg):
    """All objects in their semantic colours, environment removed; K masks by colour distance (render.py:68-97)."""
    img = render(cam, _semantic_scene(gs_environment, gs_object_list), pipe_settings, bg)["render"]
    m = M.color_masks(img, color_set, M.MASK_THRESHOLD)
    individual = m.permute(1, 2, 0).cpu().numpy().astype(np.float64)
    return individual, img.cpu().permute((1, 2, 0))


def render_semanticsegmentation_mask(cam, gs_environment, gs_object_list, color_set, height, width, pipe_settings,
                                     bg, debug):
    img = render(cam, _semantic_scene(gs_environment, gs_object_list), pipe_settings, bg)["render"]
    return (np.ascontiguousarray(img.cpu().permute((1, 2, 0))) * 255).astype("uint8")


def assign_semantic_colors(gaussians_object_list, semantic_colors):
    """What PEGASUS.init_start_position does to every object (pegasus.py:223-232)."""
    for obj_id, obj in gaussians_object_list.items():
        obj._features_dc_color = copy.deepcopy(obj._features_dc)
        obj._features_rest_color = copy.deepcopy(obj._features_rest)
        c = torch.as_tensor(semantic_colors[obj_id - 1]).to(obj._features_dc.device)
        obj._features_dc_semantics = RGB2SH(c)
        obj._features_rest_semantics = torch.asarray([0, 0, 0], device=obj._features_dc.device)
