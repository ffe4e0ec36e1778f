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

from .. import _lib

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


class _Workspace:
    """Per-device scratch owned by torch's allocator; grown on demand, reused across calls."""

    def __init__(self):
        self.buf: dict = {}
        self.capacity_hint: dict = {}

    def get(self, device, nbytes: int) -> torch.Tensor:
        t = self.buf.get(device)
        if t is None or t.numel() < nbytes:
            t = None
            self.buf[device] = None
            t = torch.empty(int(nbytes * 1.1) + 256, dtype=torch.uint8, device=device)
            self.buf[device] = t
        return t


_WS = _Workspace()
_LAST_INFO: dict = {}


def last_forward_info() -> dict:
    """{'num_instances', 'max_instances', 'n', 'workspace', ...} of the most recent forward (tests, bench)."""
    return dict(_LAST_INFO)


def _dev_f32(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings: GaussianRasterizationSettings, want_aux: bool = False):
    """Forward rasterization of one view.  Returns (color, radii, depth) -- plus (final_T, n_contrib)
    when ``want_aux``."""
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("GaussianRasterizer needs tensors on a HIP device (torch device 'cuda'); "
                           "there is no CPU path")
    rs = raster_settings
    H, W = int(rs.image_height), int(rs.image_width)
    n = int(means3D.shape[0])

    means3D = _dev_f32(means3D, device)
    opacities = _dev_f32(opacities, device)
    sh = _dev_f32(sh, device)
    colors_precomp = _dev_f32(colors_precomp, device)
    scales = _dev_f32(scales, device)
    rotations = _dev_f32(rotations, device)
    cov3Ds_precomp = _dev_f32(cov3Ds_precomp, device)
    bg = _dev_f32(rs.bg, device)
    view = _dev_f32(rs.viewmatrix, device)
    proj = _dev_f32(rs.projmatrix, device)
    campos = _dev_f32(rs.campos, device)

    color = torch.empty((3, H, W), dtype=torch.float32, device=device)
    depth = torch.empty((1, H, W), dtype=torch.float32, device=device)
    radii = torch.empty((n,), dtype=torch.int32, device=device)
    final_T = torch.empty((H, W), dtype=torch.float32, device=device) if want_aux else None
    n_contrib = torch.empty((H, W), dtype=torch.int32, device=device) if want_aux else None

    scene = _lib.PgrScene(
        n=n, means3d=_ptr(means3D), opacities=_ptr(opacities), scales=_ptr(scales), rotations=_ptr(rotations),
        cov3d_precomp=_ptr(cov3Ds_precomp), shs=_ptr(sh), colors_precomp=_ptr(colors_precomp),
        sh_degree=int(rs.sh_degree), sh_stride=int(sh.shape[1]) if sh is not None else 0,
        scale_modifier=float(rs.scale_modifier))
    cam = _lib.PgrCamera(image_width=W, image_height=H, tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
                         viewmatrix=_ptr(view), projmatrix=_ptr(proj), campos=_ptr(campos), bg=_ptr(bg))
    outs = _lib.PgrOutputs(color=_ptr(color), depth=_ptr(depth), radii=_ptr(radii), final_T=_ptr(final_T),
                           n_contrib=_ptr(n_contrib))

    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    key = (device, n, W, H)
    max_inst = _WS.capacity_hint.get(key, max(1 << 20, 8 * n))
    need = C.c_int64(0)
    with torch.cuda.device(device):
        for _attempt in range(3):
            nbytes = L.pgr_workspace_bytes(n, W, H, max_inst)
            if nbytes == 0:
                raise ValueError("pgr_workspace_bytes: invalid sizes")
            ws = _WS.get(device, nbytes)
            status = L.pgr_forward(C.byref(scene), C.byref(cam), C.byref(outs), C.c_void_p(ws.data_ptr()),
                                   ws.numel(), max_inst, C.byref(need), stream)
            if status != _lib.PGR_ERR_INSTANCE_OVERFLOW:
                break
            max_inst = int(need.value * 1.25) + 1024   # grow once to what this view needs, then retry
        _lib.check(status, "pgr_forward")
    used_max_inst = max_inst
    if need.value > 0.8 * max_inst:
        max_inst = int(need.value * 1.5) + 1024
    _WS.capacity_hint[key] = max_inst
    _LAST_INFO.update(num_instances=int(need.value), max_instances=int(max_inst),
                      used_max_instances=int(used_max_inst), n=n, width=W, height=H,
                      workspace=ws, workspace_bytes=int(ws.numel()))
    if want_aux:
        return color, radii, depth, final_T, n_contrib
    return color, radii, depth


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
