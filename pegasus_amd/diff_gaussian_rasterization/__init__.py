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
from typing import NamedTuple

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
                        raster_settings: GaussianRasterizationSettings, want_aux: bool = False, after_enqueue=None,
                        sh_rest=None):
    """Forward rasterization of one view.  Returns (color, radii, depth) -- plus (final_T, n_contrib)
    when ``want_aux``.

    The call is ENQUEUED without a host round trip (pgr_forward_posed_early_status: tables through pinned memory), then
    ``after_enqueue(result_dict)`` runs -- work that only needs the outputs in stream order, e.g. render()'s visibility
    filter, is queued behind the compositor -- and then the host waits for the call's STATUS WORDS only, which are final
    behind the tile scan (a third into the call: the one thing the host has to decide is whether the instance capacity
    held).  The function returns while scatter, sort and compositor still run; the tensors it returns are complete in
    stream order, like the result of any torch operation.  Until round 6 it waited for the end of the call, and the GPU
    idled for the host code between two render() calls (78 us of a 0.50 ms call, profiles/r06_single_view_timeline.txt).
    ``after_enqueue`` runs again if an instance overflow re-rendered the view.
    ``sh_rest``: with it, ``sh`` is the model's _features_dc [N,1,3] and ``sh_rest`` its _features_rest [N,K-1,3]
    (PgrScene::shs_rest) -- the coefficients where they are stored, instead of get_features' concatenation."""
    rs = raster_settings
    view = rasterizer.ViewSpec(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg, rs.viewmatrix,
                               rs.projmatrix, rs.campos)
    pb = rasterizer.forward_views(means3D, opacities, [view], shs=sh, colors_precomp=colors_precomp, scales=scales,
                                  rotations=rotations, cov3D_precomp=cov3Ds_precomp, sh_degree=rs.sh_degree,
                                  scale_modifier=rs.scale_modifier, want_radii=True, want_aux=want_aux,
                                  async_slot=("single-view", 0), early_status=True, shs_rest=sh_rest)
    if isinstance(pb, list):            # (an empty view list cannot happen here; an empty SCENE still returns a handle)
        r = pb[0]
    else:
        pb.record_info = True
        if after_enqueue is not None:
            after_enqueue(pb.results[0])
        r = pb.wait()[0]
        if after_enqueue is not None and getattr(pb, "_was_redone", False):
            after_enqueue(r)
    if want_aux:
        return r["color"], r["radii"], r["depth"], r["final_T"], r["n_contrib"]
    return r["color"], r["radii"], r["depth"]


_TENSOR_KEYS = ("means3D", "sh", "colors", "op", "scales", "rot", "cov", "bg", "view", "proj", "campos")


class _RasterizeGaussians(torch.autograd.Function):
    """Differentiable single-view rasterization (training path).  The forward keeps its own workspace (the
    backward walks the same per-tile lists), so it does not share the pooled scratch of the no-grad path.

    Every tensor the backward re-reads goes through ``ctx.save_for_backward``: an in-place update of means, scales,
    rotations, opacities or SH between forward and backward trips autograd's version-counter check instead of
    silently pairing new parameter values with the forward's lists, final_T and n_contrib."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        L = _lib.lib()
        rs = raster_settings
        device = means3D.device
        if device.type != "cuda":
            raise RuntimeError("GaussianRasterizer needs tensors on a HIP device (torch device 'cuda'); "
                               "there is no CPU path")
        H, W, n = int(rs.image_height), int(rs.image_width), int(means3D.shape[0])
        t = {k: _dev_f32(v, device) for k, v in dict(means3D=means3D, sh=sh, colors=colors_precomp, op=opacities,
                                                     scales=scales, rot=rotations, cov=cov3Ds_precomp, bg=rs.bg,
                                                     view=rs.viewmatrix, proj=rs.projmatrix, campos=rs.campos).items()}
        scene = _lib.PgrScene(n=n, means3d=_ptr(t["means3D"]), opacities=_ptr(t["op"]), scales=_ptr(t["scales"]),
                              rotations=_ptr(t["rot"]), cov3d_precomp=_ptr(t["cov"]), shs=_ptr(t["sh"]),
                              colors_precomp=_ptr(t["colors"]), sh_degree=int(rs.sh_degree),
                              sh_stride=int(t["sh"].shape[1]) if t["sh"] is not None else 0,
                              scale_modifier=float(rs.scale_modifier))
        cam = _lib.PgrCamera(image_width=W, image_height=H, tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
                             viewmatrix=_ptr(t["view"]), projmatrix=_ptr(t["proj"]), campos=_ptr(t["campos"]),
                             bg=_ptr(t["bg"]))
        color = torch.empty((3, H, W), dtype=torch.float32, device=device)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=device)
        radii = torch.empty((n,), dtype=torch.int32, device=device)
        final_T = torch.empty((H, W), dtype=torch.float32, device=device)
        n_contrib = torch.empty((H, W), dtype=torch.int32, device=device)
        outs = _lib.PgrOutputs(color=_ptr(color), depth=_ptr(depth), radii=_ptr(radii), final_T=_ptr(final_T),
                               n_contrib=_ptr(n_contrib))
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        max_inst = max(1 << 18, 4 * n)
        need = C.c_int64(0)
        with torch.cuda.device(device):
            for _attempt in range(3):
                ws = torch.empty(L.pgr_workspace_bytes(n, W, H, max_inst), dtype=torch.uint8, device=device)
                status = L.pgr_forward(C.byref(scene), C.byref(cam), C.byref(outs), C.c_void_p(ws.data_ptr()),
                                       ws.numel(), max_inst, C.byref(need), stream)
                if status != _lib.PGR_ERR_INSTANCE_OVERFLOW:
                    break
                max_inst = rasterizer.grown_capacity(need.value, 1.25)
            _lib.check(status, "pgr_forward")
        ctx.rs, ctx.max_inst, ctx.n = rs, max_inst, n
        # non-tensor state stays on ctx; tensors (inputs as the kernels read them + the forward's own buffers) are saved
        ctx.present = tuple(k for k in _TENSOR_KEYS if t[k] is not None)
        ctx.op_shape = tuple(opacities.shape)
        ctx.save_for_backward(*(t[k] for k in ctx.present), ws, radii, final_T, n_contrib)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_depth):
        L = _lib.lib()
        rs, n = ctx.rs, ctx.n
        saved = ctx.saved_tensors          # raises if an input was modified in place since the forward
        k = len(ctx.present)
        t = dict.fromkeys(_TENSOR_KEYS)
        t.update(zip(ctx.present, saved[:k]))
        ws, radii, final_T, n_contrib = saved[k:]
        device = ws.device
        H, W = int(rs.image_height), int(rs.image_width)
        # preprocess_backward_kernel writes every element of every gradient it is handed (zeros for culled Gaussians): no
        # zero fill (384 MB for the SH gradient of a 2 M-Gaussian scene) -- except for an empty scene, where nothing runs
        alloc = torch.empty if n > 0 else torch.zeros
        z = lambda *shape: alloc(shape, dtype=torch.float32, device=device)
        g = dict(means2d=z(n, 3), means3d=z(n, 3), opacities=z(n, 1))
        if t["sh"] is not None:
            g["shs"] = z(*t["sh"].shape)
        else:
            g["colors"] = z(n, 3)
        if t["cov"] is not None:
            g["cov3d"] = z(n, 6)
        else:
            g["scales"], g["rotations"] = z(n, 3), z(n, 4)
        grads = _lib.PgrGradOutputs(**{k: _ptr(v) for k, v in g.items()})
        scene = _lib.PgrScene(n=n, means3d=_ptr(t["means3D"]), opacities=_ptr(t["op"]), scales=_ptr(t["scales"]),
                              rotations=_ptr(t["rot"]), cov3d_precomp=_ptr(t["cov"]), shs=_ptr(t["sh"]),
                              colors_precomp=_ptr(t["colors"]), sh_degree=int(rs.sh_degree),
                              sh_stride=int(t["sh"].shape[1]) if t["sh"] is not None else 0,
                              scale_modifier=float(rs.scale_modifier))
        cam = _lib.PgrCamera(image_width=W, image_height=H, tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
                             viewmatrix=_ptr(t["view"]), projmatrix=_ptr(t["proj"]), campos=_ptr(t["campos"]),
                             bg=_ptr(t["bg"]))
        gc = grad_color.contiguous().float()
        gd = None if grad_depth is None else grad_depth.contiguous().float()
        rows = torch.empty((n, 12), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _lib.check(L.pgr_backward(C.byref(scene), C.byref(cam), _ptr(gc), _ptr(gd), _ptr(final_T), _ptr(n_contrib),
                                      _ptr(radii), C.c_void_p(ws.data_ptr()), ws.numel(), ctx.max_inst,
                                      C.byref(grads), _ptr(rows),
                                      C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "pgr_backward")
        # means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings
        return (g["means3d"], g["means2d"], g.get("shs"), g.get("colors"), g["opacities"].view(ctx.op_shape),
                g.get("scales"), g.get("rotations"), g.get("cov3d"), None)


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
        inputs = (means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
        if torch.is_grad_enabled() and any(x is not None and x.requires_grad for x in inputs):
            if means2D is None:
                means2D = torch.zeros_like(means3D, requires_grad=False)
            return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                             cov3D_precomp, self.raster_settings)
        with torch.no_grad():
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, self.raster_settings)
