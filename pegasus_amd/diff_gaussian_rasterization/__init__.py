"""Drop-in for the ``diff_gaussian_rasterization`` extension PEGASUS renders through.

Same surface as the module the reference installs from its (absent) submodule
``depth-diff-gaussian-rasterization`` (/root/reference/setup.sh:19; consumers
/root/reference/src/gs/render.py:16-17,57-58,86-87,118-119 via ``gaussian_renderer.render``):

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None)
        -> (color[3,H,W], radii[N] int32, depth[1,H,W])
    GaussianRasterizer.markVisible(positions) -> bool[N]

The compute is libpegasus_raster.so (hand-written HIP for gfx950) reached through its C ABI
(include/pegasus_raster.h); tensors are passed as raw device pointers, work is enqueued on
torch's current stream, and memory comes from torch's caching allocator.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import nn

from .. import _lib, rasterizer
from ..rasterizer import dev_f32 as _dev_f32, _ptr

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "last_forward_info"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def last_forward_info() -> dict:
    return rasterizer.last_forward_info()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings: GaussianRasterizationSettings, want_aux: bool = False):
    """Forward rasterization of one view.  Returns (color, radii, depth) -- plus (final_T, n_contrib)
    when ``want_aux``."""
    rs = raster_settings
    view = rasterizer.ViewSpec(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg, rs.viewmatrix,
                               rs.projmatrix, rs.campos)
    r = rasterizer.forward_views(means3D, opacities, [view], shs=sh, colors_precomp=colors_precomp, scales=scales,
                                 rotations=rotations, cov3D_precomp=cov3Ds_precomp, sh_degree=rs.sh_degree,
                                 scale_modifier=rs.scale_modifier, want_radii=True, want_aux=want_aux)[0]
    if want_aux:
        return r["color"], r["radii"], r["depth"], r["final_T"], r["n_contrib"]
    return r["color"], r["radii"], r["depth"]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool[N]: Gaussian centre is in front of the near plane of this view."""
        L = _lib.lib()
        with torch.no_grad():
            device = positions.device
            if device.type != "cuda":
                raise RuntimeError("markVisible needs a HIP device tensor")
            pos = _dev_f32(positions, device)
            n = int(positions.shape[0])
            present = torch.zeros((n,), dtype=torch.uint8, device=device)
            if n:
                view = _dev_f32(self.raster_settings.viewmatrix, device)
                with torch.cuda.device(device):
                    _lib.check(L.pgr_mark_visible(n, _ptr(pos), _ptr(view), _ptr(present),
                                                  C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                               "pgr_mark_visible")
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        with torch.no_grad():
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, self.raster_settings)
