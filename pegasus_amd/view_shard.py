"""View-sharded multi-GPU rendering: one process per GPU, scene replicated, views dealt round-robin,
finished frames gathered to one rank.

Views are independent given the merged scene, so there is NO collective on the data path of a frame; the
only communication is the final gather of finished frames, done with ``torch.distributed.gather`` (on the
"nccl" backend = RCCL: grouped send/recv, so the 7 peers of an 8-GPU node stream to the root over their
own xGMI links concurrently).  Backend-agnostic: the same code runs on gloo in the CPU tests.
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


def gather_frames(local: Dict[str, torch.Tensor], n_items: int, dst: int = 0, group=None,
                  async_op: bool = False):
    """``local[k]`` holds this rank's frames [n_local, ...] in the order of shard_indices().  On ``dst``
    returns {k: tensor [n_items, ...]} in GLOBAL order, elsewhere None.  With async_op returns
    (finish, works): call finish() after waiting to assemble."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cap = max_local(n_items, world)
    works, staging, wire_dtype = [], {}, {}
    for k, t in local.items():
        # neither RCCL/NCCL nor gloo carry 16-bit integers (the uint16-millimetre depth images travel as int16 storage):
        # such tensors go over the wire as bytes and are viewed back on arrival
        wire_dtype[k] = t.dtype
        if t.dtype in (torch.int16, getattr(torch, "uint16", torch.int16)):
            t = t.contiguous().view(torch.uint8)
        pad = t
        if t.shape[0] < cap:   # ranks with one view fewer pad so every rank contributes equal-sized chunks
            pad = torch.cat([t, t.new_zeros((cap - t.shape[0],) + tuple(t.shape[1:]))], 0)
        pad = pad.contiguous()
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        w = dist.gather(pad, bufs, dst=dst, group=group, async_op=async_op)
        staging[k] = bufs
        if async_op:
            works.append(w)

    def finish():
        if rank != dst:
            return None
        out = {}
        for k, bufs in staging.items():
            full = torch.empty((n_items,) + tuple(bufs[0].shape[1:]), dtype=bufs[0].dtype, device=bufs[0].device)
            for r in range(world):
                idx = shard_indices(n_items, r, world)
                if idx:
                    full[idx] = bufs[r][: len(idx)]
            out[k] = full if full.dtype == wire_dtype[k] else full.view(wire_dtype[k])
        return out

    if async_op:
        return finish, works
    return finish()


def render_sharded(n_views: int, render_batch: Callable[[List[int]], Dict[str, torch.Tensor]], batch: int,
                   gather: bool = True, dst: int = 0, group=None) -> Optional[Dict[str, torch.Tensor]]:
    """Every rank renders its shard of ``n_views`` in batches of ``batch`` global view indices handed to
    ``render_batch`` (which returns a dict of [len(indices), ...] tensors); the gather of batch b is issued
    asynchronously and overlaps the rendering of batch b+1.  Returns the assembled frames on ``dst``."""
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
