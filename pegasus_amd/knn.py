"""Mean squared distance to the 3 nearest neighbours (the reference's `simple_knn._C.distCUDA2`,
/root/reference/src/gs/gaussian_model.py:25,147) on the HIP library's grid search (`pgr_knn_mean_dist2`)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def dist2(points: torch.Tensor) -> torch.Tensor:
    """points [N,3] float32 on a HIP device -> [N] float32: mean of the squared distances to the 3 nearest other
    points.  There is no CPU path."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 needs a tensor on a HIP device (torch device 'cuda'); there is no CPU path")
    L = _lib.lib()
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points must be [N,3]")
    n = int(pts.shape[0])
    out = torch.empty((n,), dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    with torch.cuda.device(pts.device):
        ws = torch.empty((L.pgr_knn_workspace_bytes(n),), dtype=torch.uint8, device=pts.device)
        _lib.check(L.pgr_knn_mean_dist2(n, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(ws.data_ptr()), ws.numel(),
                                        C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)),
                   "pgr_knn_mean_dist2")
    return out


distCUDA2 = dist2
