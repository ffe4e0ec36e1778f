"""Host side of the rasterizer: turns torch tensors into the C-ABI structs of
include/pegasus_raster.h, manages the torch-owned workspace, and handles instance-capacity growth.

``forward_views`` renders a batch of views of one scene through ``pgr_forward_batch``; the
drop-in ``GaussianRasterizer`` (pegasus_amd.diff_gaussian_rasterization) is the n_views == 1 case.
There is no CPU fallback anywhere in this module.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib


@dataclass
class ViewSpec:
    """What GaussianRasterizationSettings says about one view (tensors stay on the device)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    campos: torch.Tensor
    depth_mode: int = 0          # PgrDepthMode: 0 = sum T alpha z (default), 1 = normalised by 1 - T_final


class _Workspace:
    """Per-device scratch from torch's caching allocator; grown on demand, reused across calls."""

    def __init__(self):
        self.buf = {}
        self.capacity_hint = {}

    def get(self, device, nbytes: int, slot=0) -> torch.Tensor:
        key = (device, slot)
        t = self.buf.get(key)
        if t is None or t.numel() < nbytes:
            self.buf[key] = None
            del t
            t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
            self.buf[key] = t
        return t

    def pinned(self, slot, nbytes: int) -> torch.Tensor:
        key = ("pinned", slot)
        t = self.buf.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes) + 64, dtype=torch.uint8).pin_memory()
            self.buf[key] = t
        return t

    def status_event(self, device, slot) -> torch.cuda.Event:
        """One event per asynchronous slot for pgr_forward_posed_early_status.  torch creates the HIP event at the first
        record(); after that ``cuda_event`` is the handle the library records again on every call."""
        key = ("status-event", device, slot)
        ev = self.buf.get(key)
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            self.buf[key] = ev
        return ev


_WS = _Workspace()
_LAST_INFO: dict = {}

MAX_INSTANCES = 0x7FFFFFFF      # per view: list positions are 32-bit (include/pegasus_raster.h)


def grown_capacity(need: int, factor: float) -> int:
    """Instance capacity for a retry after PGR_ERR_INSTANCE_OVERFLOW.  The device reports the count saturated at
    2^32 - 1; a view that needs more than MAX_INSTANCES list entries cannot be rendered and the retry gives up."""
    if need > MAX_INSTANCES:
        raise RuntimeError(f"a view lists {'>= ' if need >= 0xFFFFFFFF else ''}{need} (Gaussian, tile) instances; the "
                           f"rasterizer's per-view limit is {MAX_INSTANCES}")
    return min(MAX_INSTANCES, int(need * factor) + 1024)


def reset_capacity(hint: Optional[int] = None, *, free_workspaces: bool = True) -> None:
    """Public reset hook of the module's instance-capacity bookkeeping.

    ``hint=None`` forgets every learned capacity (the next call of each scene shape starts from its default);
    ``hint=k`` sets every capacity learned so far to ``k`` list entries per view (tests force overflows that way).
    ``free_workspaces`` also drops the cached asynchronous workspaces, so the next batch really is allocated at the new
    capacity.  The library itself keeps no state: this is host-side bookkeeping of the torch-owned scratch."""
    if hint is None:
        _WS.capacity_hint.clear()
    else:
        for key in list(_WS.capacity_hint):
            _WS.capacity_hint[key] = int(hint)
    if free_workspaces:
        drop_async_workspaces()


def drop_async_workspaces() -> None:
    """Releases the cached DEVICE workspaces of the asynchronous slots -- keys (device, ("async", slot)) -- so that the next
    batch of each slot allocates afresh.  The page-locked host scratch of a slot (("pinned", slot): a few KiB of tables and
    status words, independent of the instance capacity) stays: re-pinning memory per reset costs more than it frees."""
    for key in [k for k in _WS.buf if isinstance(k, tuple) and len(k) == 2 and isinstance(k[1], tuple) and len(k[1]) == 2
                and k[1][0] == "async"]:
        _WS.buf.pop(key)


def capacity_hints() -> dict:
    """Learned per-view instance capacities, keyed by (device, n, width, height[, "layers", n_layers])."""
    return dict(_WS.capacity_hint)


def set_capacity_hint(key, hint: Optional[int]) -> None:
    """Sets (or with ``None`` forgets) the learned capacity of one scene shape (key as in capacity_hints())."""
    if hint is None:
        _WS.capacity_hint.pop(key, None)
    else:
        _WS.capacity_hint[key] = int(hint)


def last_forward_info() -> dict:
    """Bookkeeping of the most recent forward: num_instances per view, capacities, workspace tensor."""
    return dict(_LAST_INFO)


def dev_f32(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class PendingBatch:
    """Handle of an asynchronous forward_views call: ``wait()`` blocks until the batch has finished on its
    stream, checks the overflow flags (re-rendering synchronously with a larger capacity if needed) and
    returns the list of result dicts."""

    def __init__(self, results, event, scratch, nv, key, max_inst, redo):
        self.results, self._event, self._scratch, self._nv = results, event, scratch, nv
        self._key, self._max_inst, self._redo = key, max_inst, redo
        self.num_instances = None
        self.record_info = False     # True: wait() fills last_forward_info() like a synchronous call (single-view drop-in)

    def wait(self):
        if self._event is not None:
            self._event.synchronize()
            need = (C.c_int64 * self._nv)()
            status = _lib.lib().pgr_batch_status(C.c_void_p(self._scratch.data_ptr()), self._nv, need)
            self.num_instances = [int(x) for x in need]
            peak = max(self.num_instances)
            if peak > 0.8 * self._max_inst and peak <= MAX_INSTANCES:
                _WS.capacity_hint[self._key] = max(_WS.capacity_hint.get(self._key, 0), grown_capacity(peak, 1.6))   # few, large steps:
                # every growth reallocates the multi-GB workspace (tens of ms)
            self._event = None
            if self.record_info and status != _lib.PGR_ERR_INSTANCE_OVERFLOW:
                ws = self._keep[1]
                _LAST_INFO.clear()
                _LAST_INFO.update(num_instances=self.num_instances, max_instances=int(self._max_inst),
                                  used_max_instances=int(self._max_inst), n=int(self._key[1]), width=int(self._key[2]),
                                  height=int(self._key[3]), n_views=self._nv, workspace=ws, workspace_bytes=int(ws.numel()))
            if status == _lib.PGR_ERR_INSTANCE_OVERFLOW:
                self._was_redone = True
                _WS.capacity_hint[self._key] = max(_WS.capacity_hint.get(self._key, 0), grown_capacity(peak, 1.6))   # raises beyond MAX_INSTANCES
                self.results = self._redo()          # synchronous path grows the workspace and retries
            else:
                _lib.check(status, "pgr_forward_batch_async")
            self._redo = None
        return self.results


def forward_views(means3D, opacities, views: Sequence[ViewSpec], *, shs=None, colors_precomp=None, scales=None,
                  rotations=None, cov3D_precomp=None, sh_degree=0, scale_modifier=1.0, want_radii=True,
                  want_aux=False, stage_ms: Optional[list] = None, outputs: Optional[list] = None,
                  async_slot=None, semantic: Optional[dict] = None, posed: Optional[dict] = None, tie_index=None,
                  tie_inv=None, layers: Optional[dict] = None, early_status: bool = False, shs_rest=None):
    """Renders ``len(views)`` views of one scene.  Returns a list of dicts with keys
    color[3,H,W], depth[1,H,W], radii[N] (or None), and final_T / n_contrib when ``want_aux``.

    ``stage_ms``: pass an empty list to use the profiling entry point; it receives the per-stage
    milliseconds (whole batch) measured with HIP events on the launch stream.
    ``outputs``: optional pre-allocated list of dicts (same keys) to render into.
    ``async_slot``: not None -> enqueue on torch's CURRENT stream without synchronising and return a
    PendingBatch; the slot names the workspace / pinned scratch to use (one batch in flight per slot).
    ``shs_rest``: the SH coefficients as the model stores them -- ``shs`` = _features_dc [N,1,3], ``shs_rest`` = _features_rest
    [N,K-1,3] (PgrScene::shs_rest) -- instead of their concatenation; results are bit-identical.
    ``early_status`` (with ``async_slot``, not layered): PendingBatch.wait() returns as soon as the call's status words are
    final -- behind the tile scan, pgr_forward_posed_early_status -- instead of at the end of the call; the outputs are
    complete in stream order (whatever the caller queues on the current stream, or fetches with .cpu(), comes after them).
    ``semantic``: dict(object_id int32[N], colors float32[K,3], n_env, k) -> the fused objects-only semantic
    render is written to r["sem_color"] (and r["sem_depth"]) of every view (pgr_forward_frames_async).
    ``semantic`` may also carry ``mask_colors`` float32[K,3] (+ ``mask_threshold``): every output dict with a ``sem_masks``
    uint8[K,H,W] tensor then receives the K colour-distance masks of the semantic image from the compositor's epilogue
    (bit for bit what color_masks() computes from ``sem_color``), and ``object_id_u8`` (scene_prepare()).
    An output dict with a ``record`` uint8[record_layout(H, W, K)["bytes"]] tensor (masks.record_layout; K = semantic["k"] when
    the descriptor carries mask_colors, else 0) also receives the view's frame record from the compositor's epilogue: bit
    for bit masks.pack_records of that view's color / depth / sem_masks.
    ``tie_index``: int32[N] permutation -- exact depth ties are broken by it instead of the position (PgrScene.tie_index);
    ``tie_inv``: its inverse from scene_prepare() (otherwise rebuilt per call).
    ``layers``: dict(layer_id int32[N], n_layers, mask_colors float32[n_layers,3], mask_threshold) -> LAYERED call
    (pgr_forward_layers_async, asynchronous only): Gaussian i is composited into image layer_id[i] alone and every
    output dict's ``sem_masks`` uint8[n_layers,H,W] receives the layers' masks (silhouettes); no colour image.
    ``posed``: dict(object_id int32[N], poses float32[len(views), K, 20]) -> dynamic scene: view i places object k by
    poses[i, k-1] inside the preprocess (pgr_forward_posed_async; pegasus_amd.compose.pose_table builds the rows).
    """
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("the rasterizer needs tensors on a HIP device (torch device 'cuda'); there is no CPU path")
    nv = len(views)
    if nv == 0:
        return []
    H, W = int(views[0].image_height), int(views[0].image_width)
    n = int(means3D.shape[0])
    means3D_in, opacities_in = means3D, opacities      # (an empty scene converts to None = NULL pointers: re-entrant calls get the originals)
    means3D = dev_f32(means3D, device)
    opacities = dev_f32(opacities, device)
    shs = dev_f32(shs, device)
    shs_rest = dev_f32(shs_rest, device)
    if shs_rest is not None:
        if shs is None or shs.dim() != 3 or shs.shape[1] != 1 or shs_rest.dim() != 3 or shs_rest.shape[0] != shs.shape[0]:
            raise ValueError("shs_rest goes with shs = [N,1,3] (the first coefficient) and is [N,K-1,3]")
        if shs_rest.shape[1] == 0:
            shs_rest = None                        # nothing but the first coefficient: the plain layout with stride 1
    colors_precomp = dev_f32(colors_precomp, device)
    scales = dev_f32(scales, device)
    rotations = dev_f32(rotations, device)
    cov3D_precomp = dev_f32(cov3D_precomp, device)

    scene = _lib.PgrScene(
        n=n, means3d=_ptr(means3D), opacities=_ptr(opacities), scales=_ptr(scales), rotations=_ptr(rotations),
        cov3d_precomp=_ptr(cov3D_precomp), shs=_ptr(shs), colors_precomp=_ptr(colors_precomp),
        sh_degree=int(sh_degree),
        sh_stride=(int(shs.shape[1]) + (int(shs_rest.shape[1]) if shs_rest is not None else 0)) if shs is not None else 0,
        scale_modifier=float(scale_modifier), tie_index=_ptr(tie_index), tie_inv=_ptr(tie_inv) if tie_index is not None else None,
        shs_rest=_ptr(shs_rest))

    cams = (_lib.PgrCamera * nv)()
    outs = (_lib.PgrOutputs * nv)()
    keep = []
    results: List[dict] = []
    for i, v in enumerate(views):
        if int(v.image_height) != H or int(v.image_width) != W:
            raise ValueError("all views of a batch must share the image size")
        bg, vm, pm, cp = (dev_f32(t, device) for t in (v.bg, v.viewmatrix, v.projmatrix, v.campos))
        keep.append((bg, vm, pm, cp))
        cams[i] = _lib.PgrCamera(image_width=W, image_height=H, tanfovx=float(v.tanfovx), tanfovy=float(v.tanfovy),
                                 viewmatrix=_ptr(vm), projmatrix=_ptr(pm), campos=_ptr(cp), bg=_ptr(bg),
                                 depth_mode=int(getattr(v, "depth_mode", 0)))
        if outputs is not None:
            r = outputs[i]
        else:
            r = dict(color=torch.empty((3, H, W), dtype=torch.float32, device=device),
                     depth=torch.empty((1, H, W), dtype=torch.float32, device=device),
                     radii=torch.empty((n,), dtype=torch.int32, device=device) if want_radii else None)
            if want_aux:
                r["final_T"] = torch.empty((H, W), dtype=torch.float32, device=device)
                r["n_contrib"] = torch.empty((H, W), dtype=torch.int32, device=device)
            if semantic is not None:
                r["sem_color"] = torch.empty((3, H, W), dtype=torch.float32, device=device)
                r["sem_depth"] = torch.empty((1, H, W), dtype=torch.float32, device=device)
        outs[i] = _lib.PgrOutputs(color=_ptr(r.get("color")), depth=_ptr(r.get("depth")), radii=_ptr(r.get("radii")),
                                  final_T=_ptr(r.get("final_T")), n_contrib=_ptr(r.get("n_contrib")),
                                  sem_color=_ptr(r.get("sem_color")) if semantic is not None else None,
                                  sem_depth=_ptr(r.get("sem_depth")) if semantic is not None else None,
                                  sem_masks=_ptr(r.get("sem_masks")) if (semantic is not None or layers is not None) else None,
                                  record=_ptr(r.get("record")) if layers is None else None)
        results.append(r)

    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    key = (device, n, W, H)
    max_inst = _WS.capacity_hint.get(key, max(1 << 20, 6 * n))
    sem_struct = posed_struct = None
    if posed is not None:
        if stage_ms is not None:
            raise ValueError("the profiling entry point does not take posed objects")
        poses = dev_f32(posed["poses"], device)
        if poses.dim() != 3 or poses.shape[0] != nv or poses.shape[2] != _lib.PGR_POSE_STRIDE:
            raise ValueError("posed['poses'] must be [n_views, K, 20]")
        keep.append(poses)
        posed_struct = _lib.PgrPosedObjects(object_id=_ptr(posed["object_id"]), poses=_ptr(poses),
                                            k_objects=int(poses.shape[1]))
    if semantic is not None:
        sem_struct = _lib.PgrSemantic(object_id=_ptr(semantic["object_id"]), colors=_ptr(semantic["colors"]),
                                      n_env=int(semantic["n_env"]), k_objects=int(semantic["k"]),
                                      object_id_u8=_ptr(semantic.get("object_id_u8")),
                                      mask_colors=_ptr(semantic.get("mask_colors")),
                                      mask_threshold=float(semantic.get("mask_threshold", 0.1)))
    layers_struct = None
    if layers is not None:
        if async_slot is None or semantic is not None or stage_ms is not None:
            raise ValueError("a layered call is asynchronous (async_slot) and takes no semantic descriptor")
        layers_struct = _lib.PgrLayers(layer_id=_ptr(layers["layer_id"]), n_layers=int(layers["n_layers"]),
                                       mask_colors=_ptr(layers["mask_colors"]),
                                       mask_threshold=float(layers.get("mask_threshold", 0.1)))
        key = (device, n, W, H, "layers", int(layers["n_layers"]))
        max_inst = _WS.capacity_hint.get(key, max(1 << 20, 6 * n))
    if semantic is not None or posed is not None:
        if async_slot is None and stage_ms is None:      # synchronous fused call: enqueue asynchronously, wait, retry on overflow
            kw = dict(shs=shs, shs_rest=shs_rest, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                      cov3D_precomp=cov3D_precomp, sh_degree=sh_degree, scale_modifier=scale_modifier,
                      want_radii=want_radii, want_aux=want_aux, outputs=results, semantic=semantic, posed=posed,
                      tie_index=tie_index, tie_inv=tie_inv)
            for _attempt in range(3):
                pb = forward_views(means3D_in, opacities_in, views, async_slot="sync-fused", **kw)
                pb._redo = None
                pb._event.synchronize()
                need = (C.c_int64 * nv)()
                status = L.pgr_batch_status(C.c_void_p(pb._scratch.data_ptr()), nv, need)
                if status != _lib.PGR_ERR_INSTANCE_OVERFLOW:
                    _lib.check(status, "pgr_forward_frames_async")
                    _LAST_INFO.clear()
                    _LAST_INFO.update(num_instances=[int(x) for x in need], max_instances=int(pb._max_inst),
                                      used_max_instances=int(pb._max_inst), n=n, width=W, height=H, n_views=nv,
                                      workspace=pb._keep[1], workspace_bytes=int(pb._keep[1].numel()))
                    return results
                _WS.capacity_hint[key] = grown_capacity(max(need), 1.6)
            raise RuntimeError("instance capacity did not converge")
    if async_slot is not None:
        with torch.cuda.device(device):
            if layers_struct is not None:
                nbytes = L.pgr_layers_workspace_bytes(n, W, H, max_inst, nv, layers_struct.n_layers)
            else:
                nbytes = L.pgr_batch_workspace_bytes(n, W, H, max_inst, nv)
            ws = _WS.get(device, nbytes, slot=("async", async_slot))
            sb = L.pgr_host_scratch_bytes(nv)
            scratch = _WS.pinned(async_slot, sb)
            if layers_struct is not None:
                _lib.check(L.pgr_forward_layers_async(C.byref(scene), C.byref(layers_struct),
                                                      C.byref(posed_struct) if posed_struct else None, nv,
                                                      cams, outs, C.c_void_p(ws.data_ptr()), ws.numel(), max_inst,
                                                      C.c_void_p(scratch.data_ptr()), scratch.numel(), stream),
                           "pgr_forward_layers_async")
            elif early_status:
                ev = _WS.status_event(device, async_slot)
                _lib.check(L.pgr_forward_posed_early_status(C.byref(scene), C.byref(sem_struct) if sem_struct else None,
                                                            C.byref(posed_struct) if posed_struct else None, nv,
                                                            cams, outs, C.c_void_p(ws.data_ptr()), ws.numel(), max_inst,
                                                            C.c_void_p(scratch.data_ptr()), scratch.numel(), stream,
                                                            C.c_void_p(ev.cuda_event)),
                           "pgr_forward_posed_early_status")
            else:
                _lib.check(L.pgr_forward_posed_async(C.byref(scene), C.byref(sem_struct) if sem_struct else None,
                                                     C.byref(posed_struct) if posed_struct else None, nv,
                                                     cams, outs, C.c_void_p(ws.data_ptr()), ws.numel(), max_inst,
                                                     C.c_void_p(scratch.data_ptr()), scratch.numel(), stream),
                           "pgr_forward_posed_async")
            if not (early_status and layers_struct is None):
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
        kw = dict(shs=shs, shs_rest=shs_rest, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                  cov3D_precomp=cov3D_precomp, sh_degree=sh_degree, scale_modifier=scale_modifier,
                  want_radii=want_radii, want_aux=want_aux, outputs=results, semantic=semantic, posed=posed,
                  tie_index=tie_index, tie_inv=tie_inv)
        if layers is not None:          # a layered call is asynchronous only: the retry after an overflow is one too
            def redo():
                # bounded like the fused path: every attempt runs at the capacity the device asked for (a view that needs
                # more than MAX_INSTANCES raises from grown_capacity instead of overflowing for ever)
                for _attempt in range(3):
                    pb2 = forward_views(means3D_in, opacities_in, views, async_slot=async_slot, layers=layers, **kw)
                    pb2._redo = None
                    pb2._event.synchronize()
                    need2 = (C.c_int64 * nv)()
                    status2 = L.pgr_batch_status(C.c_void_p(pb2._scratch.data_ptr()), nv, need2)
                    if status2 != _lib.PGR_ERR_INSTANCE_OVERFLOW:
                        _lib.check(status2, "pgr_forward_layers_async")
                        return pb2.results
                    _WS.capacity_hint[key] = grown_capacity(max(need2), 1.6)
                raise RuntimeError("instance capacity did not converge")
        else:
            redo = lambda: forward_views(means3D_in, opacities_in, views, **kw)
        pb = PendingBatch(results, ev, scratch, nv, key, max_inst, redo)
        pb._keep = (keep, ws, cams, outs, scene)
        return pb
    need = (C.c_int64 * nv)()
    ms = (C.c_float * _lib.PGR_NUM_STAGES)()
    with torch.cuda.device(device):
        for _attempt in range(3):
            nbytes = L.pgr_batch_workspace_bytes(n, W, H, max_inst, nv)
            if nbytes == 0:
                raise ValueError("pgr_batch_workspace_bytes: invalid sizes")
            ws = _WS.get(device, nbytes)
            if stage_ms is not None:
                status = L.pgr_forward_batch_profiled(C.byref(scene), C.byref(sem_struct) if sem_struct else None,
                                                      nv, cams, outs, C.c_void_p(ws.data_ptr()), ws.numel(),
                                                      max_inst, need, stream, ms)
            else:
                status = L.pgr_forward_batch(C.byref(scene), nv, cams, outs, C.c_void_p(ws.data_ptr()), ws.numel(),
                                             max_inst, need, stream)
            if status != _lib.PGR_ERR_INSTANCE_OVERFLOW:
                break
            max_inst = grown_capacity(max(need), 1.25)   # grow to what the largest view needs, then retry
        _lib.check(status, "pgr_forward_batch")
    used_max_inst = max_inst
    peak = max(need) if nv else 0
    if peak > 0.8 * max_inst:
        max_inst = grown_capacity(peak, 1.6)
    _WS.capacity_hint[key] = max_inst
    if stage_ms is not None:
        stage_ms[:] = list(ms)
    _LAST_INFO.clear()
    _LAST_INFO.update(num_instances=[int(x) for x in need], max_instances=int(max_inst),
                      used_max_instances=int(used_max_inst), n=n, width=W, height=H, n_views=nv,
                      workspace=ws, workspace_bytes=int(ws.numel()))
    del keep
    return results


def workspace_view(view_index: int = 0) -> dict:
    """Device pointers (as ints) of view ``view_index`` inside the last forward's workspace."""
    L = _lib.lib()
    info = _LAST_INFO
    ws = info["workspace"]
    v = _lib.PgrWorkspaceView()
    _lib.check(L.pgr_workspace_view(C.c_void_p(ws.data_ptr()), ws.numel(), info["n"], info["width"], info["height"],
                                    info["used_max_instances"], info["n_views"], view_index, C.byref(v)),
               "pgr_workspace_view")
    return {k: getattr(v, k) for k, _ in _lib.PgrWorkspaceView._fields_}


def block_visibility(means3D, views: Sequence[ViewSpec], *, scales=None, rotations=None, cov3D_precomp=None,
                     scale_modifier=1.0) -> torch.Tensor:
    """bool[ceil(n / 64), len(views)]: False = none of the 64 Gaussians of that block can have a non-zero radius in
    that view (pgr_block_visibility: the conservative test the batch calls use internally to skip whole waves)."""
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("block_visibility needs tensors on a HIP device")
    n, nv = int(means3D.shape[0]), len(views)
    means3D, scales, rotations, cov3D_precomp = (dev_f32(t, device) for t in (means3D, scales, rotations, cov3D_precomp))
    ones = torch.ones((max(n, 1), 3), dtype=torch.float32, device=device)       # opacities / colours are not looked at
    scene = _lib.PgrScene(n=n, means3d=_ptr(means3D), opacities=_ptr(ones), scales=_ptr(scales), rotations=_ptr(rotations),
                          cov3d_precomp=_ptr(cov3D_precomp), shs=None, colors_precomp=_ptr(ones), sh_degree=0, sh_stride=0,
                          scale_modifier=float(scale_modifier), tie_index=None)
    cams = (_lib.PgrCamera * nv)()
    keep = []
    for i, v in enumerate(views):
        bg, vm, pm, cp = (dev_f32(t, device) for t in (v.bg, v.viewmatrix, v.projmatrix, v.campos))
        keep.append((bg, vm, pm, cp))
        cams[i] = _lib.PgrCamera(image_width=int(v.image_width), image_height=int(v.image_height), tanfovx=float(v.tanfovx),
                                 tanfovy=float(v.tanfovy), viewmatrix=_ptr(vm), projmatrix=_ptr(pm), campos=_ptr(cp), bg=_ptr(bg))
    groups, words = (n + 63) // 64, (nv + 31) // 32
    out = torch.zeros((groups, words), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        ws = torch.empty(L.pgr_block_visibility_workspace_bytes(n, nv) + 256, dtype=torch.uint8, device=device)
        _lib.check(L.pgr_block_visibility(C.byref(scene), nv, cams, C.c_void_p(ws.data_ptr()), ws.numel(),
                                          C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                   "pgr_block_visibility")
        torch.cuda.current_stream(device).synchronize()
    bits = (out.unsqueeze(2) >> torch.arange(32, device=device, dtype=torch.int32)) & 1
    return bits.reshape(groups, words * 32)[:, :nv].bool()


def scene_prepare(n: int, tie_index: Optional[torch.Tensor] = None, semantic: Optional[dict] = None) -> dict:
    """Per-SCENE constants of the batch calls, computed once (pgr_scene_prepare): ``tie_inv`` (inverse permutation of
    ``tie_index``) and ``object_id_u8`` (the object Gaussians' ids as bytes).  Returns a dict with those two device
    tensors (None where the input is absent) and the cache tensor that owns their memory ("cache": keep it alive)."""
    L = _lib.lib()
    ref = tie_index if tie_index is not None else (semantic["object_id"] if semantic is not None else None)
    if ref is None or n == 0:
        return dict(tie_inv=None, object_id_u8=None, cache=None)
    device = ref.device
    if device.type != "cuda":
        raise RuntimeError("scene_prepare needs tensors on a HIP device")
    scene = _lib.PgrScene(n=int(n), tie_index=_ptr(tie_index))
    sem = None
    if semantic is not None:
        sem = _lib.PgrSemantic(object_id=_ptr(semantic["object_id"]), colors=_ptr(semantic["colors"]),
                               n_env=int(semantic["n_env"]), k_objects=int(semantic["k"]))
    with torch.cuda.device(device):
        cache = torch.empty(int(L.pgr_scene_cache_bytes(int(n))) + 256, dtype=torch.uint8, device=device)
        p_inv, p_u8 = C.c_void_p(), C.c_void_p()
        _lib.check(L.pgr_scene_prepare(C.byref(scene), C.byref(sem) if sem is not None else None,
                                       C.c_void_p(cache.data_ptr()), cache.numel(), C.byref(p_inv), C.byref(p_u8),
                                       C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "pgr_scene_prepare")

    def view(ptr, nbytes, dtype):
        if not ptr.value:
            return None
        off = ptr.value - cache.data_ptr()
        return cache[off:off + nbytes].view(dtype)
    n_obj = int(n) - int(semantic["n_env"]) if semantic is not None else 0
    return dict(tie_inv=view(p_inv, 4 * int(n), torch.int32), object_id_u8=view(p_u8, n_obj, torch.uint8), cache=cache)
