"""BOP-layout writer for the batch path ("mask/depth writers").

Layout and naming follow the reference's writer (/root/reference/src/visualization/object_visualization.py:373-391,
458-480; /root/reference/pegasus.py:347-358):

    <out>/train/<scene:06d>/rgb/<frame:06d>.png              8-bit RGB,  (img * 255).astype(uint8)
    <out>/train/<scene:06d>/depth/<frame:06d>.png            16-bit grey, (depth * 1000).astype(uint16)  (millimetres)
    <out>/train/<scene:06d>/mask_visib/<frame:06d>_<obj:06d>.png   8-bit 0/255      visible masks ('seg_vis')
    <out>/train/<scene:06d>/mask/<frame:06d>_<obj:06d>.png         8-bit 0/255      silhouettes   ('seg_sil')
    <out>/train/<scene:06d>/sem_mask/<frame:06d>.png               8-bit RGB, uint8 of the semantic image ('sem_seg')
    <out>/train/<scene:06d>/scene_gt.json, scene_camera.json      pegasus_amd.bop_pose records
    <out>/train/<scene:06d>/scene_gt_info.json                    per object and frame: pixel counts, visible fraction, 2D
                                                                  boxes (BOP format, bop_datasets_format.md:116-129) -- written
                                                                  when a batch brings both visible masks and silhouettes

(the five image kinds of the reference's write_training_data, /root/reference/src/tools/pegasus_working.py:412-439, for the
data points ['rgb','depth','seg_vis','seg_sil','sem_seg'] of /root/reference/pegasus.py:491)

The reference encodes PNGs with imageio (absent here); PNG is deflate + CRC, so zlib is all it takes.  Quantisation
runs on the GPU (pgr_quantize_frame); only the 8/16-bit images cross PCIe.
"""
from __future__ import annotations

import json
import struct
import zlib
from pathlib import Path

import numpy as np


def encode_png(a: np.ndarray, level: int = 1) -> bytes:
    """uint8 [H,W] / [H,W,3] or uint16 [H,W] -> PNG bytes (colour type 0 or 2, no interlace, filter 0)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint8 and a.ndim == 2:
        depth, ctype, rows = 8, 0, a
    elif a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
        depth, ctype, rows = 8, 2, a.reshape(a.shape[0], -1)
    elif a.dtype == np.uint16 and a.ndim == 2:
        depth, ctype, rows = 16, 0, a.astype(">u2").view(np.uint8).reshape(a.shape[0], -1)     # big-endian samples
    else:
        raise ValueError("encode_png takes uint8 [H,W], uint8 [H,W,3] or uint16 [H,W]")
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()                  # filter type 0 per row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b""))


def decode_png(data: bytes) -> np.ndarray:
    """Inverse of encode_png for the files this module writes (filter 0 only): used by the tests."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = 3 if ctype == 2 else 1
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch * depth // 8)
    assert not raw[:, 0].any()
    body = raw[:, 1:]
    if depth == 16:
        return body.reshape(h, w, 2).copy().view(">u2").reshape(h, w).astype(np.uint16)
    return body.reshape(h, w, 3).copy() if ch == 3 else body.copy()


class BopSceneWriter:
    """Collects the frames of one scene; `add_batch` takes FrameRenderer output (device tensors) + pose records.

    PNG encoding is the only per-frame host work of the batch path and, on one thread, 100x slower than rendering
    (2.9 s for 64 frames that render in 26 ms).  Like the reference, which starts a thread per frame for its writers
    (/root/reference/pegasus.py:346-358), the encoder runs beside the renderer: `add_batch` quantises on the GPU, copies
    the 8/16-bit images to the host once per batch and hands one task per PNG to a thread pool (zlib releases the GIL);
    `close()` waits for them.  At most `max_pending` batches are in flight (bounded host memory)."""

    def __init__(self, out_dir, scene_id: int = 0, png_level: int = 1, workers: int = None, max_pending: int = 4):
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.scene = Path(out_dir) / "train" / f"{scene_id:06d}"
        for d in ("rgb", "depth", "mask_visib", "mask", "sem_mask"):
            (self.scene / d).mkdir(parents=True, exist_ok=True)
        self.scene_gt, self.scene_camera, self.scene_gt_info = {}, {}, {}
        self.n_frames, self.level = 0, png_level
        self.workers = max(1, min(32, os.cpu_count() or 1) if workers is None else int(workers))
        self._pool = ThreadPoolExecutor(max_workers=self.workers) if self.workers > 1 else None
        self._pending, self._max_pending = [], max(1, int(max_pending))

    def _write(self, path, image):
        path.write_bytes(encode_png(image, self.level))

    def _submit(self, path, image, batch_futures):
        if self._pool is None:
            self._write(path, image)
        else:
            batch_futures.append(self._pool.submit(self._write, path, image))

    def add_batch(self, frames: dict, scene_gt: dict, scene_camera: dict, n: int = None, silhouettes=None, frame_ids=None,
                  record_shape=None):
        """``frames``: FrameRenderer output (color, depth, and with masks: seg, masks); ``silhouettes``: uint8 [B,K,H,W]
        of FrameRenderer.render_silhouettes (or frames["sil"]) -> the mask/ directory.  ``frame_ids``: the frames' numbers in
        the dataset (default: consecutive) -- a view-sharded run gives every rank's writer the GLOBAL ids of its frames."""
        from . import masks as M
        masks_dev = frames.get("masks")
        if "color" in frames:
            n = frames["color"].shape[0] if n is None else n
            # GPU: uint8 HWC / uint16 millimetres for the whole batch in one launch (pgr_pack_frames), then ONE
            # device->host copy per kind and batch
            packed = M.pack_frames(color=frames["color"][:n], depth=frames["depth"][:n])
            depth_mm_dev = packed["depth_mm"]
            rgb8 = packed["rgb"].cpu().numpy()
            mm = depth_mm_dev.cpu().numpy().view(np.uint16)
        else:
            # a records-only frame set (FrameRenderer.alloc_frames(images=False | "seg")): the compositor's epilogue wrote the
            # casts already -- the record IS what the writers take
            if record_shape is None:
                raise ValueError("a frame set without images needs record_shape=(H, W, K)")
            n = frames["records"].shape[0] if n is None else n
            H, W, K = (int(v) for v in record_shape)
            rv = M.record_views(frames["records"][:n], H, W, K)
            depth_mm_dev = rv["depth_mm"]
            rgb8 = rv["rgb"].cpu().numpy()
            mm = depth_mm_dev.cpu().numpy().view(np.uint16)
            masks_dev = M.unpack_mask_bits(rv["mask_bits"], K) if K else None
        mk = (masks_dev[:n] * 255).cpu().numpy() if masks_dev is not None else None
        sem8 = M.pack_frames(color=frames["seg"][:n])["rgb"].cpu().numpy() if "seg" in frames else None
        if silhouettes is None:
            silhouettes = frames.get("sil")
        sil = (silhouettes[:n] * 255).cpu().numpy() if silhouettes is not None else None
        info = None
        if silhouettes is not None and masks_dev is not None:
            # BOP's scene_gt_info from the masks the batch already holds, counted on the GPU (a few hundred numbers per batch)
            from . import bop_pose
            info = bop_pose.gt_info_from_masks(masks_dev[:n], silhouettes[:n], depth_mm_dev != 0)
        futures = []
        for i in range(n):
            fid = self.n_frames if frame_ids is None else int(frame_ids[i])
            self._submit(self.scene / "rgb" / f"{fid:06d}.png", rgb8[i], futures)
            self._submit(self.scene / "depth" / f"{fid:06d}.png", mm[i], futures)
            if mk is not None:
                for k in range(mk.shape[1]):
                    self._submit(self.scene / "mask_visib" / f"{fid:06d}_{k:06d}.png", mk[i, k], futures)
            if sem8 is not None:
                self._submit(self.scene / "sem_mask" / f"{fid:06d}.png", sem8[i], futures)
            if sil is not None:
                for k in range(sil.shape[1]):
                    self._submit(self.scene / "mask" / f"{fid:06d}_{k:06d}.png", sil[i, k], futures)
            self.scene_gt[str(fid)] = scene_gt[str(i)]
            self.scene_camera[str(fid)] = scene_camera[str(i)]
            if info is not None:
                from . import bop_pose
                self.scene_gt_info[str(fid)] = bop_pose.scene_gt_info_entry(info, i)
            self.n_frames += 1
        if futures:
            self._pending.append(futures)
            while len(self._pending) > self._max_pending:
                for f in self._pending.pop(0):
                    f.result()                       # re-raises a writer's exception here

    def close(self, write_json: bool = True):
        """Waits for the PNG tasks; writes scene_gt.json / scene_camera.json unless ``write_json`` is False (a view-sharded
        run merges the ranks' records first: ``merge_records``)."""
        for batch in self._pending:
            for f in batch:
                f.result()
        self._pending = []
        if self._pool is not None:
            self._pool.shutdown(wait=True)
        if write_json:
            self.write_records()
        return self.scene

    def merge_records(self, others):
        """Adds the (scene_gt, scene_camera) dict pairs of other ranks' writers (disjoint frame ids)."""
        for rec in others:
            self.scene_gt.update(rec[0])
            self.scene_camera.update(rec[1])
            if len(rec) > 2:
                self.scene_gt_info.update(rec[2])

    def write_records(self):
        order = lambda d: {k: d[k] for k in sorted(d, key=int)}
        (self.scene / "scene_gt.json").write_text(json.dumps(order(self.scene_gt)))
        (self.scene / "scene_camera.json").write_text(json.dumps(order(self.scene_camera)))
        if self.scene_gt_info:
            (self.scene / "scene_gt_info.json").write_text(json.dumps(order(self.scene_gt_info)))
