"""Per-object masks and frame quantisation on the device.

Reference behaviour: /root/reference/src/gs/render.py:60-63,89-93 (``np.linalg.norm(img - c, axis=2) <= 0.1``
per semantic colour, on the host) and /root/reference/pegasus.py:347,355 (uint8 / uint16-mm casts).
Here both run as HIP kernels on the frames where they were rendered (pgr_color_masks,
pgr_quantize_frame), so only the final small integers ever cross PCIe.
"""
from __future__ import annotations

import colorsys
import ctypes as C

import numpy as np
import torch

from . import _lib

MASK_THRESHOLD = 0.1


def generate_colors(n: int, mode: str = "bgr") -> np.ndarray:
    """Semantic colour table: hue i/n, lightness 0.6, saturation 0.7, returned BGR by default, float32 [n,3].
    Same values as /root/reference/src/utility/graphic_utils.py:40-60 (pinned by tests/golden/graphic_utils.npz)."""
    colors = []
    for i in range(n):
        r, g, b = colorsys.hls_to_rgb(i / n, 0.6, 0.7)
        if mode == "bgr":
            colors.append((b, g, r))
        elif mode == "rgb":
            colors.append((r, g, b))
        else:
            raise ValueError("Color mode {} is not supported".format(mode))
    return np.asarray(colors, dtype=np.float32).reshape(n, 3)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def color_masks(img_chw: torch.Tensor, colors: torch.Tensor, threshold: float = MASK_THRESHOLD,
                out: torch.Tensor = None) -> torch.Tensor:
    """uint8 [K,H,W] (or [B,K,H,W] for a [B,3,H,W] batch, one launch): 1 where the rendered colour is within
    ``threshold`` (L2) of semantic colour k."""
    L = _lib.lib()
    if img_chw.device.type != "cuda":
        raise RuntimeError("color_masks needs a HIP device tensor; there is no CPU path")
    img = img_chw.contiguous().float()
    colors = colors.to(img.device).contiguous().float().reshape(-1, 3)
    batched = img.dim() == 4
    nb = img.shape[0] if batched else 1
    H, W = img.shape[-2:]
    K = colors.shape[0]
    if out is None:
        out = torch.empty(((nb, K, H, W) if batched else (K, H, W)), dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(L.pgr_color_masks(C.c_void_p(img.data_ptr()), nb, W, H, C.c_void_p(colors.data_ptr()), K,
                                     float(threshold), C.c_void_p(out.data_ptr()), _stream(img.device)),
                   "pgr_color_masks")
    return out


def quantize_frame(img_chw: torch.Tensor, depth: torch.Tensor):
    """(uint8 [H,W,3], uint16-as-int16-storage [H,W]) exactly as the reference's numpy casts."""
    L = _lib.lib()
    if img_chw.device.type != "cuda":
        raise RuntimeError("quantize_frame needs HIP device tensors; there is no CPU path")
    img = img_chw.contiguous().float()
    d = depth.contiguous().float()
    _, H, W = img.shape
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=img.device)
    mm = torch.empty((H, W), dtype=torch.int16, device=img.device)   # bit pattern of uint16 millimetres
    with torch.cuda.device(img.device):
        _lib.check(L.pgr_quantize_frame(C.c_void_p(img.data_ptr()), C.c_void_p(d.data_ptr()), W, H,
                                        C.c_void_p(rgb.data_ptr()), C.c_void_p(mm.data_ptr()), _stream(img.device)),
                   "pgr_quantize_frame")
    return rgb, mm


def pack_frames(color: torch.Tensor = None, depth: torch.Tensor = None, masks: torch.Tensor = None) -> dict:
    """One launch for a whole finished batch (pgr_pack_frames): ``color`` [B,3,H,W] -> "rgb" uint8 [B,H,W,3],
    ``depth`` [B,1,H,W] -> "depth_mm" int16 storage of uint16 millimetres [B,H,W] (both exactly the reference's numpy
    casts, /root/reference/pegasus.py:347,355), ``masks`` uint8 [B,K,H,W] -> "mask_bits" uint8 [B,H,W,ceil(K/8)] with
    mask m in bit m % 8 of byte m // 8.  The outputs are NEW tensors: a frame set may be re-rendered while they travel."""
    L = _lib.lib()
    ref = next(t for t in (color, depth, masks) if t is not None)
    if ref.device.type != "cuda":
        raise RuntimeError("pack_frames needs HIP device tensors; there is no CPU path")
    dev = ref.device
    B, (H, W) = int(ref.shape[0]), ref.shape[-2:]
    out, ptr = {}, lambda t: None if t is None else C.c_void_p(t.data_ptr())
    if color is not None:
        color = color.contiguous().float()
        out["rgb"] = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    if depth is not None:
        depth = depth.contiguous().float()
        out["depth_mm"] = torch.empty((B, H, W), dtype=torch.int16, device=dev)
    K = 0
    if masks is not None:
        masks = masks.contiguous()
        if masks.dtype != torch.uint8:
            raise ValueError("masks must be uint8 [B,K,H,W] (pgr_color_masks output)")
        K = int(masks.shape[1])
        out["mask_bits"] = torch.empty((B, H, W, (K + 7) // 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.pgr_pack_frames(ptr(color), ptr(depth), ptr(masks), B, K, W, H, ptr(out.get("rgb")),
                                     ptr(out.get("depth_mm")), ptr(out.get("mask_bits")), _stream(dev)),
                   "pgr_pack_frames")
    return out


def record_layout(height: int, width: int, k: int) -> dict:
    """Byte offsets of the three sections of a frame record and its size (pgr_frame_record_layout; host only)."""
    lay = _lib.PgrRecordLayout()
    _lib.check(_lib.lib().pgr_frame_record_layout(int(width), int(height), int(k), C.byref(lay)), "pgr_frame_record_layout")
    return dict(off_rgb=int(lay.off_rgb), off_depth=int(lay.off_depth), off_masks=int(lay.off_masks), bytes=int(lay.bytes))


def pack_records(color: torch.Tensor = None, depth: torch.Tensor = None, masks: torch.Tensor = None,
                 out: torch.Tensor = None) -> torch.Tensor:
    """One launch, ONE uint8 record per frame (pgr_pack_records): [B, record_layout(H, W, K)["bytes"]] holding the frame's
    uint8 RGB (HWC), uint16 depth millimetres and the K masks as bit planes back to back -- the unit the gather to the root
    rank and the disk writers move.  ``out``: a preallocated [>= B, bytes] uint8 tensor (e.g. a FrameGather send buffer)
    is filled in place; record_views() slices a record tensor back into the three images without copying."""
    L = _lib.lib()
    ref = next(t for t in (color, depth, masks) if t is not None)
    if ref.device.type != "cuda":
        raise RuntimeError("pack_records needs HIP device tensors; there is no CPU path")
    dev = ref.device
    B, (H, W) = int(ref.shape[0]), ref.shape[-2:]
    K = 0
    if color is not None:
        color = color.contiguous().float()
    if depth is not None:
        depth = depth.contiguous().float()
    if masks is not None:
        masks = masks.contiguous()
        if masks.dtype != torch.uint8:
            raise ValueError("masks must be uint8 [B,K,H,W]")
        K = int(masks.shape[1])
    nbytes = record_layout(H, W, K)["bytes"]
    if out is None:
        out = torch.empty((B, nbytes), dtype=torch.uint8, device=dev)
    if out.dtype != torch.uint8 or out.dim() != 2 or out.shape[0] < B or out.shape[1] < nbytes or out.stride(1) != 1 \
            or out.stride(0) % 16 or out.device != dev:
        raise ValueError(f"out must be a uint8 [>= {B}, >= {nbytes}] device tensor with a 16-byte-aligned row stride")
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(L.pgr_pack_records(ptr(color), ptr(depth), ptr(masks), B, K, W, H, ptr(out), int(out.stride(0)),
                                      _stream(dev)), "pgr_pack_records")
    return out


def record_views(records: torch.Tensor, height: int, width: int, k: int) -> dict:
    """Zero-copy views of a [..., bytes] uint8 record tensor (any device, any leading shape, last dimension contiguous):
    "rgb" uint8 [..., H, W, 3], "depth_mm" int16 storage of uint16 millimetres [..., H, W], "mask_bits" uint8
    [..., H, W, ceil(k/8)]."""
    lay = record_layout(height, width, k)
    P, J = height * width, (k + 7) // 8
    lead = tuple(records.shape[:-1])
    out = {"rgb": records[..., :3 * P].unflatten(-1, (height, width, 3)),
           "depth_mm": records[..., lay["off_depth"]:lay["off_depth"] + 2 * P].view(torch.int16).unflatten(-1, (height, width))}
    if J:
        out["mask_bits"] = records[..., lay["off_masks"]:lay["off_masks"] + J * P].unflatten(-1, (height, width, J))
    assert out["rgb"].shape[:-3] == lead
    return out


def unpack_mask_bits(mask_bits: torch.Tensor, k: int) -> torch.Tensor:
    """Inverse of pack_frames' mask packing: uint8 [B,H,W,ceil(k/8)] -> uint8 [B,k,H,W] (any device; host writers)."""
    planes = [(mask_bits[..., m // 8] >> (m % 8)) & 1 for m in range(k)]
    return torch.stack(planes, 1) if planes else mask_bits.new_zeros((mask_bits.shape[0], 0) + tuple(mask_bits.shape[1:3]))
