"""View-sharded multi-GPU rendering: one process per GPU, scene replicated, views dealt round-robin,
finished frames gathered to one rank.

Views are independent given the merged scene, so there is NO collective on the data path of a frame; the
only communication is the gather of finished frames, done with ``torch.distributed.gather`` (on the
"nccl" backend = RCCL: grouped send/recv straight into the root's receive tensors, so the 7 peers of an
8-GPU node stream to the root over their own xGMI links concurrently).  Backend-agnostic: the same code runs
on gloo in the CPU tests.

Shape of the exchange (round 4): ONE collective per batch, carrying ONE uint8 record per frame (what
``pgr_pack_records`` writes: uint8 RGB, uint16 depth millimetres, the K masks as bit planes), into buffers that exist
before the first batch.  With view v -> rank v mod world and equal-sized chunks the root's rank-major receive buffer
``recv[r, i]`` IS the global order under a transposed view -- frame g = i * world + r -- so nothing is reordered, copied
or allocated on the root per batch (``FrameGather``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Global indices owned by ``rank``: rank, rank+world, ...  (view v -> rank v mod world)."""
    return list(range(rank, n_items, world))


def max_local(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


def global_id(rank: int, i: int, world: int) -> int:
    """The (rank, i) -> global frame rule of a gathered batch: local frame i of rank r is global frame i * world + r."""
    return i * world + rank


class FrameGather:
    """Per-batch gather of fixed-size frame records to ``dst`` with everything preallocated.

    ``send_buffer(slot)`` is this rank's uint8 [cap, record_bytes] staging tensor (the pack kernel writes straight into
    it); ``start(slot)`` issues the one collective of the batch; ``finish(slot)`` completes it and returns, on ``dst``, the
    rank-major receive tensor [world, cap, record_bytes] (elsewhere None).  ``depth`` slots: one gather in flight beside
    the batch being packed.  ``global_view(slot)`` is the same memory as [cap, world, record_bytes]: row-major order of
    its first two axes is the global frame order; ranks that owned fewer than ``cap`` frames of a batch leave their tail
    rows undefined.
    """

    def __init__(self, cap: int, record_bytes: int, device, dst: int = 0, group=None, depth: int = 2,
                 pin_memory: bool = False):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.cap, self.record_bytes, self.depth = int(cap), int(record_bytes), int(depth)
        kw = dict(dtype=torch.uint8, device=device)
        if pin_memory and torch.device(device).type == "cpu":
            kw["pin_memory"] = True
        self.send = [torch.zeros((self.cap, self.record_bytes), **kw) for _ in range(self.depth)]      # (zeroed once: the 16-byte padding between a record's sections is never written)
        self.recv = ([torch.empty((self.world, self.cap, self.record_bytes), **kw) for _ in range(self.depth)]
                     if self.rank == dst else None)
        # the per-rank receive views handed to the collective: made once (unbind returns views of recv)
        self._recv_lists = [list(r.unbind(0)) for r in self.recv] if self.recv is not None else None
        self._work = [None] * self.depth
        self.batches = 0

    @property
    def bytes_per_rank_and_batch(self) -> int:
        return self.cap * self.record_bytes

    def send_buffer(self, slot: int) -> torch.Tensor:
        return self.send[slot % self.depth]

    def start(self, slot: int):
        s = slot % self.depth
        if self._work[s] is not None:
            raise RuntimeError("FrameGather: slot still in flight (finish() it first)")
        self._work[s] = dist.gather(self.send[s], self._recv_lists[s] if self._recv_lists is not None else None,
                                    dst=self.dst, group=self.group, async_op=True)
        self.batches += 1

    def finish(self, slot: int) -> Optional[torch.Tensor]:
        s = slot % self.depth
        w = self._work[s]
        if w is not None:
            w.wait()
            self._work[s] = None
        return self.recv[s] if self.recv is not None else None

    def finish_all(self):
        for s in range(self.depth):
            self.finish(s)

    def global_view(self, slot: int) -> Optional[torch.Tensor]:
        """[cap, world, record_bytes] view of the receive buffer: [i, r] = global frame i * world + r (no copy)."""
        return None if self.recv is None else self.recv[slot % self.depth].transpose(0, 1)


def _as_bytes(t: torch.Tensor) -> torch.Tensor:
    """[n, ...] tensor -> uint8 [n, bytes per item] view (neither RCCL nor gloo carries 16-bit integers; everything
    travels as bytes and is viewed back on arrival)."""
    t = t.contiguous()
    return t.view(torch.uint8).reshape(t.shape[0], -1) if t.shape[0] else t.new_zeros((0, _item_bytes(t)), dtype=torch.uint8)


def _item_bytes(t: torch.Tensor) -> int:
    n = 1
    for d in t.shape[1:]:
        n *= int(d)
    return n * t.element_size()


def gather_frames(local: Dict[str, torch.Tensor], n_items: int, dst: int = 0, group=None,
                  async_op: bool = False):
    """Generic form for callers without preallocated records: ``local[k]`` holds this rank's frames [n_local, ...] in the
    order of shard_indices().  All tensors of the dict travel as ONE uint8 record per frame in ONE gather.  On ``dst``
    returns {k: tensor [n_items, ...]} in GLOBAL order (one strided copy per tensor kind out of the rank-major receive
    buffer), elsewhere None.  With async_op returns (finish, works): call finish() after waiting."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cap = max_local(n_items, world)
    keys = list(local.keys())
    meta = {k: (tuple(local[k].shape[1:]), local[k].dtype, _item_bytes(local[k])) for k in keys}
    rec_bytes = sum(m[2] for m in meta.values())
    first = local[keys[0]]
    record = torch.zeros((cap, rec_bytes), dtype=torch.uint8, device=first.device)
    off = 0
    for k in keys:
        b = _as_bytes(local[k])
        record[: b.shape[0], off:off + meta[k][2]] = b
        off += meta[k][2]
    recv = torch.empty((world, cap, rec_bytes), dtype=torch.uint8, device=first.device) if rank == dst else None
    w = dist.gather(record, list(recv.unbind(0)) if recv is not None else None, dst=dst, group=group, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        # [cap, world, bytes]: row-major order of the first two axes is the global order; one strided copy per kind
        glob = recv.transpose(0, 1)
        out, off = {}, 0
        for k in keys:
            shp, dt, nb = meta[k]
            flat = glob[:, :, off:off + nb].reshape(cap * world, nb)[:n_items].contiguous()
            out[k] = flat.view(dt).reshape((n_items,) + shp)
            off += nb
        return out

    if async_op:
        return finish, [w]
    return finish()


def render_sharded(n_views: int, render_batch: Callable[[List[int]], Dict[str, torch.Tensor]], batch: int,
                   gather: bool = True, dst: int = 0, group=None) -> Optional[Dict[str, torch.Tensor]]:
    """Every rank renders its shard of ``n_views`` in batches of ``batch`` global view indices handed to
    ``render_batch`` (which returns a dict of [len(indices), ...] tensors); returns the assembled frames on ``dst``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_indices(n_views, rank, world)
    cap = max_local(n_views, world)
    pieces: List[Dict[str, torch.Tensor]] = []
    for s in range(0, cap, batch):
        idx = mine[s:s + batch]
        if idx:
            pieces.append({k: v.clone() for k, v in render_batch(idx).items()})
    local = {k: torch.cat([p[k] for p in pieces], 0) for k in pieces[0]} if pieces else None
    if world == 1 or not gather:
        return local if rank == dst else None
    # a rank that owns no view (n_views < world) still takes part in the gather: learn the frame layout
    meta = {k: (tuple(v.shape[1:]), v.dtype, str(v.device)) for k, v in local.items()} if local else None
    metas = [None] * world
    dist.all_gather_object(metas, meta, group=group)
    ref = next((m for m in metas if m), None)
    if ref is None:
        return None
    if local is None:
        dev = ref[next(iter(ref))][2]
        dev = torch.device(dev) if dev.startswith("cpu") else torch.device("cuda", torch.cuda.current_device())
        local = {k: torch.empty((0,) + shp, dtype=dt, device=dev) for k, (shp, dt, _) in ref.items()}
    return gather_frames(local, n_views, dst=dst, group=group)
