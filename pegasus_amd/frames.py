"""Batched frame production: what one iteration of PEGASUS's per-frame loop emits
(/root/reference/pegasus.py:254-358 with data points ['rgb','depth','seg_vis'], pegasus.py:491):

    rgb + depth   : the merged scene (environment + objects) rendered with its own SH colours
    seg_vis       : the objects alone, each painted with its semantic colour (f_dc = RGB2SH(c_k), f_rest = 0,
                    /root/reference/pegasus.py:230-232), thresholded into K masks (||img - c_k|| <= 0.1,
                    /root/reference/src/gs/render.py:68-97)

The reference does this one camera at a time with a deepcopy + 6 vstack per frame and host-side numpy
masks; here a batch of cameras is ONE pipeline pass over resident device tensors (scene image, objects-only semantic image
from the same per-tile lists, the K masks from the compositor's epilogue), and nothing leaves the GPU until the caller
asks for it.  'seg_sil' (every object's silhouette, /root/reference/src/gs/render.py:36-65) is one more pass for all K
objects at once (render_silhouettes: a layered call).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import masks as M
from . import rasterizer as R
from .sh_utils import RGB2SH


class FrameRenderer:
    def __init__(self, means3d, opacities, scales, rotations, shs, object_id, sh_degree=3, device="cuda:0",
                 bg=(0.0, 0.0, 0.0), color_mode="bgr", spatial_order=True):
        """All arrays are ACTIVATED values (numpy or torch), environment Gaussians first (object_id 0) and
        each object's Gaussians after (object_id k = 1..K), as PEGASUS merges them.

        spatial_order: one-time scene layout step (pegasus_amd/scene_order.py): the resident copy is stored in
        Morton order inside the environment and inside every object.  Frames do not depend on the storage order
        (per-tile lists are depth-ordered; only EXACT depth ties fall back to the index), the memory behaviour of
        binning and gathering does.  ``self.order[i]`` = caller's index of resident Gaussian i (None: unchanged)."""
        self.device = torch.device(device)
        self.order = None
        _oid = (object_id.detach().cpu().numpy() if torch.is_tensor(object_id) else np.asarray(object_id)).astype(np.int64)
        if _oid.size and (np.diff(_oid) < 0).any():
            raise ValueError("object_id must be non-decreasing: environment Gaussians first (0), then object 1, 2, ...")
        if spatial_order and len(means3d) > 1:
            from .scene_order import spatial_order as _spatial_order
            host = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
            perm = _spatial_order(host(means3d), host(object_id))
            pick = lambda a: (a[torch.as_tensor(perm, device=a.device)] if torch.is_tensor(a) else np.asarray(a)[perm])
            means3d, opacities, scales, rotations, shs, object_id = (
                pick(a) for a in (means3d, opacities, scales, rotations, shs, object_id))
            self.order = perm
        # the resident copy keeps the CALLER's tie order: exact depth ties are broken by the caller's index, so the
        # frames are bit-identical to rendering the arrays as passed (PgrScene.tie_index)
        self.tie_index = None if self.order is None else torch.from_numpy(self.order.astype(np.int32)).to(self.device)
        t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a
                                                        ).to(self.device, dt).contiguous()
        self.means3d, self.opacities, self.scales, self.rotations, self.shs = (
            t(means3d), t(opacities), t(scales), t(rotations), t(shs))
        oid = (object_id.detach().cpu().numpy() if torch.is_tensor(object_id) else np.asarray(object_id)).astype(np.int64)
        self.sh_degree = int(sh_degree)
        self.bg = t(np.asarray(bg, np.float32))
        self.n = int(self.means3d.shape[0])
        self.K = int(oid.max()) if oid.size else 0
        obj = np.nonzero(oid > 0)[0]
        if obj.size and not np.all(np.diff(obj) == 1):
            raise ValueError("object Gaussians must follow the environment Gaussians contiguously")
        self.n_env = int(obj[0]) if obj.size else self.n
        # semantic-colour cloud of the objects alone (environment masked out, render.py:81-83)
        self.colors_np = M.generate_colors(max(self.K, 1), color_mode)[: self.K]
        self.colors = t(self.colors_np) if self.K else None
        if self.K:
            # fused semantic pass: object ids + the rgb value each object's Gaussians carry, evaluated like the
            # preprocess kernel evaluates SH degree 0: max(C0 * dc + 0.5, 0) in fp32
            dc_k = RGB2SH(self.colors_np).astype(np.float32)
            carried = np.maximum(np.float32(0.28209479177387814) * dc_k + np.float32(0.5), np.float32(0.0))
            self.semantic = dict(object_id=torch.from_numpy(oid.astype(np.int32)).to(self.device),
                                 colors=t(carried.astype(np.float32)), n_env=self.n_env, k=self.K,
                                 # the masks come out of the compositor's epilogue (bit for bit masks.color_masks of seg)
                                 mask_colors=self.colors, mask_threshold=M.MASK_THRESHOLD)
            dc = RGB2SH(self.colors_np[oid[self.n_env:] - 1]).astype(np.float32)       # [n_obj,3]
            self.sem_shs = t(dc.reshape(-1, 1, 3))
            s = slice(self.n_env, self.n)
            self.obj = dict(means3d=self.means3d[s], opacities=self.opacities[s], scales=self.scales[s],
                            rotations=self.rotations[s])
            # tie order of the objects-only cloud: the caller's order of the object Gaussians among themselves
            self.obj_tie_index = None if self.tie_index is None else (self.tie_index[s] - self.n_env).contiguous()
        # per-scene constants of the batch calls, once (pgr_scene_prepare): the inverse tie permutation, the object ids as bytes
        self._prep = R.scene_prepare(self.n, self.tie_index, self.semantic if self.K else None)
        self.tie_inv = self._prep["tie_inv"]
        self.obj_tie_inv = None
        if self.K:
            self.semantic["object_id_u8"] = self._prep["object_id_u8"]
            # the objects-only cloud (separate semantic pass, silhouettes): its own inverse tie permutation; layer = object id
            self._sil_layer_id = self.semantic["object_id"][self.n_env:].contiguous()
            self._sil_prep = R.scene_prepare(self.n - self.n_env, self.obj_tie_index)
            self.obj_tie_inv = self._sil_prep["tie_inv"]

    def view_spec(self, view, depth_mode: int = 0) -> R.ViewSpec:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        return R.ViewSpec(view.height, view.width, view.tanfovx, view.tanfovy, self.bg,
                          t(view.world_view_transform), t(view.full_proj_transform), t(view.camera_center),
                          depth_mode=int(depth_mode))

    def alloc_frames(self, batch: int, height: int, width: int, masks: bool = True, records: bool = False,
                     images: bool = True):
        """``records``: also a uint8 [batch, record bytes] tensor the compositor's epilogue fills with every frame's RECORD
        (uint8 RGB | uint16 depth mm | the K masks as bit planes: masks.record_layout / record_views) -- what leaves the GPU
        for a finished frame, without a pass that re-reads the images.  ``images=False`` (with records): a RECORDS-ONLY frame
        set -- no fp32 images, no mask planes; the compositor writes 3.84 MB per 800x800 frame instead of 29.4 MB (a rank of
        a view-sharded job: nothing there reads the planes).  ``images="seg"``: the records plus the semantic image (a
        dataset writer: its sem_mask PNG is the one product the record does not carry)."""
        dev = self.device
        if images is not True:               # False, or "seg": the record + the semantic image (the writers' sem_mask PNG)
            if not records:
                raise ValueError("a frame set without images needs records")
            f = dict(records=torch.empty((batch, self.record_bytes(height, width, masks)), dtype=torch.uint8, device=dev))
            if images == "seg" and masks and self.K:
                f["seg"] = torch.empty((batch, 3, height, width), device=dev)
            return f
        f = dict(color=torch.empty((batch, 3, height, width), device=dev),
                 depth=torch.empty((batch, 1, height, width), device=dev))
        if masks and self.K:
            f["seg"] = torch.empty((batch, 3, height, width), device=dev)
            f["seg_depth"] = torch.empty((batch, 1, height, width), device=dev)
            f["masks"] = torch.empty((batch, self.K, height, width), dtype=torch.uint8, device=dev)
        if records:
            f["records"] = torch.empty((batch, self.record_bytes(height, width, masks)), dtype=torch.uint8, device=dev)
        return f

    @staticmethod
    def _outs(frames: dict, B: int, fused: bool):
        """Per-view output dicts of a frame set; a records-only set (alloc_frames(images=False)) names no image."""
        if "color" not in frames:
            return [dict(radii=None, **({"sem_color": frames["seg"][i]} if fused and "seg" in frames else {})) for i in range(B)]
        outs = [dict(color=frames["color"][i], depth=frames["depth"][i], radii=None) for i in range(B)]
        if fused:
            for i in range(B):
                outs[i]["sem_color"], outs[i]["sem_depth"] = frames["seg"][i], frames["seg_depth"][i]
                outs[i]["sem_masks"] = frames["masks"][i]
        return outs

    def record_bytes(self, height: int, width: int, masks: bool = True) -> int:
        return M.record_layout(height, width, self.K if masks else 0)["bytes"]

    def _posed(self, poses, B):
        """poses [B, K, 20] (pegasus_amd.compose.pose_table per time step) -> the posed-objects argument."""
        if poses is None:
            return None
        if self.K == 0:
            raise ValueError("the scene has no objects to pose")
        t = torch.as_tensor(np.asarray(poses, dtype=np.float32) if not torch.is_tensor(poses) else poses)
        t = t.to(self.device, torch.float32).contiguous()
        if tuple(t.shape) != (B, self.K, 20):
            raise ValueError(f"poses must be [{B}, {self.K}, 20]")
        return dict(object_id=self.semantic["object_id"], poses=t)

    def render_frames_async(self, specs: Sequence[R.ViewSpec], frames: dict, masks: bool = True, slot: int = 0,
                            poses=None, records: torch.Tensor = None):
        """The fast path: ONE batch call renders the scene (color, depth) and -- from the same per-tile lists --
        the objects-only semantic image (seg) and its K masks (compositor epilogue); enqueued on side stream ``slot`` (2 slots =
        two batches in flight on two streams; the streams are independent of each other and of the caller's stream
        after their start, so the batches overlap on the GPU).  Returns a ``wait()``-able handle; nothing
        synchronises the host.  Keep at most one batch in flight per slot.
        ``poses`` [B, K, 20]: dynamic scene -- frame i shows object k at poses[i, k-1] (time steps as a batch).
        ``records`` (default: frames["records"] if the frame set has them): uint8 [>= B, record bytes] that receives the
        frames' records from the compositor's epilogue -- e.g. a FrameGather send buffer."""
        B = len(specs)
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if records is None:
            records = frames.get("records")
        if not hasattr(self, "_slot_streams"):
            self._slot_streams = {}
        # `slot` selects the workspace (and the caller's frame set); slots share `n_streams` streams round-robin when that
        # is set: two batches queued behind each other on a stream keep it from draining when one of them completes
        n_streams = getattr(self, "n_streams", None)
        skey = slot if not n_streams else slot % int(n_streams)
        st = self._slot_streams.get(skey)
        if st is None:
            st = self._slot_streams[skey] = torch.cuda.Stream(dev)
        fused = masks and self.K > 0
        outs = self._outs(frames, B, fused)
        if records is not None:
            if records.shape[1] < self.record_bytes(int(specs[0].image_height), int(specs[0].image_width), fused) or records.stride(0) % 16:
                raise ValueError("records: rows of at least record_bytes(H, W, masks) bytes with a 16-byte-aligned stride")
            for i in range(B):
                outs[i]["record"] = records[i]
        st.wait_stream(cur)
        posed = self._posed(poses, B)
        if posed is not None:
            # allocated on the caller's stream, read on the slot stream: tell the caching allocator, so that a caller who drops
            # its pose tensor right after this call cannot have the memory recycled under the running batch.  (The view
            # specs' matrices and every other argument are referenced by the returned handle until wait().)
            posed["poses"].record_stream(st)
        with torch.cuda.stream(st):
            h = R.forward_views(self.means3d, self.opacities, specs, shs=self.shs, scales=self.scales, tie_index=self.tie_index,
                                tie_inv=self.tie_inv, rotations=self.rotations, sh_degree=self.sh_degree, want_radii=False,
                                outputs=outs, async_slot=("frames", slot), semantic=self.semantic if fused else None,
                                posed=posed)
            ev = torch.cuda.Event()
            ev.record(st)
        # The caller's stream does NOT wait here: the two slots' streams then really overlap on the GPU (batch i+1's
        # HBM-bound preprocess / LDS-bound sort beside batch i's VALU-bound compositor).  Consumers are ordered by
        # wait() (host-side event synchronisation) or by making their own stream wait on ``pending.event``.
        if getattr(self, "serialize_slots", False):      # A/B switch for measurements: round 1's behaviour
            cur.wait_stream(st)
        renderer = self

        class _Pending:
            event = ev
            _args = (specs, posed)       # keeps the caller's device tensors alive while the batch runs

            def wait(self_inner):
                h.wait()           # (a batch re-rendered after an instance overflow rewrites its masks as well)
                ev.synchronize()
                return frames
        return _Pending()

    def render_batch_async(self, specs: Sequence[R.ViewSpec], frames: dict, masks: bool = True, slot: int = 0):
        """Enqueues the scene pass and the semantic pass of one batch on two side streams (two workspaces) so
        their kernels overlap, chains the mask kernels behind the semantic pass, and returns a ``wait()``-able
        handle without synchronising the host.  ``slot`` selects the workspace pair: keep at most one batch in
        flight per slot (two slots + two frame sets = a 2-deep pipeline)."""
        B = len(specs)
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if not hasattr(self, "_streams"):
            self._streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        s_scene, s_sem = self._streams
        outs = [dict(color=frames["color"][i], depth=frames["depth"][i], radii=None) for i in range(B)]
        s_scene.wait_stream(cur)
        with torch.cuda.stream(s_scene):
            h1 = R.forward_views(self.means3d, self.opacities, specs, shs=self.shs, scales=self.scales, tie_index=self.tie_index,
                                 tie_inv=self.tie_inv, rotations=self.rotations, sh_degree=self.sh_degree, want_radii=False,
                                 outputs=outs, async_slot=("scene", slot))
        h2 = None
        if masks and self.K:
            souts = [dict(color=frames["seg"][i], depth=frames["seg_depth"][i], radii=None) for i in range(B)]
            s_sem.wait_stream(cur)
            with torch.cuda.stream(s_sem):
                h2 = R.forward_views(self.obj["means3d"], self.obj["opacities"], specs, shs=self.sem_shs, tie_index=self.obj_tie_index,
                                     tie_inv=self.obj_tie_inv, scales=self.obj["scales"], rotations=self.obj["rotations"], sh_degree=0,
                                     want_radii=False, outputs=souts, async_slot=("sem", slot))
                M.color_masks(frames["seg"][:B], self.colors, M.MASK_THRESHOLD, out=frames["masks"][:B])
                ev_masks = torch.cuda.Event()
                ev_masks.record(s_sem)
        # no cur.wait_stream here either (see render_frames_async): wait() orders the consumers
        renderer = self

        class _Pending:
            def wait(self_inner):
                h1.wait()
                if h2 is not None:
                    before = h2.results
                    h2.wait()
                    ev_masks.synchronize()
                    if h2.results is not before:      # semantic pass was re-rendered after an overflow: redo masks
                        M.color_masks(frames["seg"][:B], renderer.colors, M.MASK_THRESHOLD, out=frames["masks"][:B])
                        torch.cuda.current_stream(dev).synchronize()
                return frames
        return _Pending()

    def render_frames(self, specs: Sequence[R.ViewSpec], frames: dict = None, masks: bool = True,
                      stage_ms: list = None, poses=None):
        """Blocking form of render_frames_async on the current stream (fused semantic pass); ``stage_ms``
        receives the HIP-event stage times of the whole batch."""
        B = len(specs)
        H, W = int(specs[0].image_height), int(specs[0].image_width)
        if frames is None:
            frames = self.alloc_frames(B, H, W, masks)
        fused = masks and self.K > 0
        outs = self._outs(frames, B, fused)
        if "records" in frames:
            for i in range(B):
                outs[i]["record"] = frames["records"][i]
        R.forward_views(self.means3d, self.opacities, specs, shs=self.shs, scales=self.scales, tie_index=self.tie_index,
                        tie_inv=self.tie_inv, rotations=self.rotations, sh_degree=self.sh_degree, want_radii=False,
                        outputs=outs, stage_ms=stage_ms, semantic=self.semantic if fused else None,
                        posed=self._posed(poses, B))
        return frames

    def render_silhouettes(self, specs: Sequence[R.ViewSpec], out: torch.Tensor = None, poses=None, slot: int = 0,
                           wait: bool = True):
        """Silhouette masks [B, K, H, W] uint8: object k rendered ALONE in its semantic colour and thresholded, i.e. its
        full outline whatever occludes it in the scene (/root/reference/src/gs/render.py:36-65 does this with one
        deepcopy + merge + render per object and camera).  ONE layered batch call for all K objects (round 4;
        pgr_forward_layers_async): the objects-only cloud is binned into per-(tile, object) lists -- the view is K stacked
        copies of the tile grid -- and the compositor's epilogue thresholds layer k against colour k, so no image is
        written or read back.  Bit-identical to K single-object passes (render_silhouettes_per_object, kept as the
        checker).  ``poses`` as in render_frames; ``wait=False`` returns (out, pending) with the batch still in flight
        on side stream ``slot``."""
        B = len(specs)
        H, W = int(specs[0].image_height), int(specs[0].image_width)
        if out is None:
            out = torch.empty((B, max(self.K, 1), H, W), dtype=torch.uint8, device=self.device)
        if self.K == 0:
            return out[:, :0] if wait else (out[:, :0], None)
        posed = None
        if poses is not None:
            posed = dict(object_id=self._sil_layer_id, poses=self._posed(poses, B)["poses"])
        layers = dict(layer_id=self._sil_layer_id, n_layers=self.K, mask_colors=self.colors,
                      mask_threshold=M.MASK_THRESHOLD)
        outs = [dict(radii=None, sem_masks=out[i]) for i in range(B)]
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if not hasattr(self, "_sil_streams"):
            self._sil_streams = {}
        st = self._sil_streams.get(slot)
        if st is None:
            st = self._sil_streams[slot] = torch.cuda.Stream(dev)
        st.wait_stream(cur)
        out.record_stream(st)                  # written on the slot's stream: the allocator must not recycle it under the pass
        if posed is not None:
            posed["poses"].record_stream(st)
        with torch.cuda.stream(st):
            h = R.forward_views(self.obj["means3d"], self.obj["opacities"], specs, shs=self.sem_shs, scales=self.obj["scales"],
                                rotations=self.obj["rotations"], sh_degree=0, want_radii=False, posed=posed,
                                tie_index=self.obj_tie_index, tie_inv=self.obj_tie_inv, outputs=outs,
                                async_slot=("silhouette", slot), layers=layers)
        h._args = (specs, posed, layers)
        if not wait:
            return out[:, :self.K], h
        h.wait()
        return out[:, :self.K]

    def render_silhouettes_per_object(self, specs: Sequence[R.ViewSpec], out: torch.Tensor = None, poses=None) -> torch.Tensor:
        """Round 2's form of render_silhouettes, kept as its checker: one batch call per object over its slice of the
        resident scene and one mask launch each (0.22 ms per view for 8 objects at full size)."""
        B = len(specs)
        H, W = int(specs[0].image_height), int(specs[0].image_width)
        if out is None:
            out = torch.empty((B, max(self.K, 1), H, W), dtype=torch.uint8, device=self.device)
        if self.K == 0:
            return out[:, :0]
        if not hasattr(self, "_obj_slices"):
            oid = self.semantic["object_id"].cpu().numpy()
            self._obj_slices = [(int(np.searchsorted(oid, k, "left")), int(np.searchsorted(oid, k, "right")))
                                for k in range(1, self.K + 1)]
        posed_all = self._posed(poses, B)
        if getattr(self, "_sil_img", None) is None or self._sil_img[0].shape != (B, 3, H, W):
            self._sil_img = [torch.empty((B, 3, H, W), device=self.device) for _ in range(2)]
        pending = [None, None]
        for k, (a, b) in enumerate(self._obj_slices):
            if b <= a:
                out[:, k] = 0
                continue
            s = slice(a, b)
            posed = None
            if posed_all is not None:
                posed = dict(object_id=torch.ones(b - a, dtype=torch.int32, device=self.device),
                             poses=posed_all["poses"][:, k:k + 1].contiguous())
            tie = None if self.tie_index is None else self._dense_rank(self.tie_index[s])
            slot = k % 2
            if pending[slot] is not None:
                pending[slot].wait()
            img = self._sil_img[slot]
            outs = [dict(color=img[i], depth=self._sil_depth(B, H, W)[i], radii=None) for i in range(B)]
            pending[slot] = R.forward_views(self.means3d[s], self.opacities[s], specs,
                                            shs=self.sem_shs[a - self.n_env:b - self.n_env], scales=self.scales[s],
                                            rotations=self.rotations[s], sh_degree=0, want_radii=False, posed=posed,
                                            tie_index=tie, outputs=outs, async_slot=("silhouette1", slot))
            out[:, k:k + 1] = M.color_masks(img, self.colors[k:k + 1], M.MASK_THRESHOLD)      # (kernel output is contiguous)
        redo = False
        for p in pending:
            if p is not None:
                p.wait()
                redo = redo or getattr(p, "_was_redone", False)
        if redo:            # an instance overflow re-rendered a pass after its masks were taken: start over (sized now)
            return self.render_silhouettes_per_object(specs, out, poses)
        return out[:, :self.K]

    def _sil_depth(self, B, H, W):
        if getattr(self, "_sil_d", None) is None or self._sil_d.shape != (B, 1, H, W):
            self._sil_d = torch.empty((B, 1, H, W), device=self.device)      # depth of the object passes: not kept
        return self._sil_d

    @staticmethod
    def _dense_rank(t: torch.Tensor) -> torch.Tensor:
        """A slice of a permutation -> the permutation of 0..len-1 with the same order (what PgrScene.tie_index wants)."""
        order = torch.argsort(t)
        rank = torch.empty_like(order)
        rank[order] = torch.arange(t.numel(), device=t.device)
        return rank.to(torch.int32)

    def render_batch(self, specs: Sequence[R.ViewSpec], frames: dict = None, masks: bool = True,
                     stage_ms: list = None, sem_stage_ms: list = None):
        """Renders len(specs) frames into ``frames`` (allocated if None).  Returns the dict of batched
        tensors: color [B,3,H,W], depth [B,1,H,W], and with masks: seg [B,3,H,W], masks [B,K,H,W] uint8."""
        B = len(specs)
        H, W = int(specs[0].image_height), int(specs[0].image_width)
        if frames is None:
            frames = self.alloc_frames(B, H, W, masks)
        outs = [dict(color=frames["color"][i], depth=frames["depth"][i], radii=None) for i in range(B)]
        R.forward_views(self.means3d, self.opacities, specs, shs=self.shs, scales=self.scales, tie_index=self.tie_index,
                        tie_inv=self.tie_inv, rotations=self.rotations, sh_degree=self.sh_degree, want_radii=False,
                        outputs=outs, stage_ms=stage_ms)
        if masks and self.K:
            souts = [dict(color=frames["seg"][i], depth=frames["seg_depth"][i], radii=None) for i in range(B)]
            R.forward_views(self.obj["means3d"], self.obj["opacities"], specs, shs=self.sem_shs, tie_index=self.obj_tie_index,
                            tie_inv=self.obj_tie_inv, scales=self.obj["scales"], rotations=self.obj["rotations"], sh_degree=0,
                            want_radii=False, outputs=souts, stage_ms=sem_stage_ms)
            M.color_masks(frames["seg"][:B], self.colors, M.MASK_THRESHOLD, out=frames["masks"][:B])
        return frames
