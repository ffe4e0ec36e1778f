"""The N>1 path on CPU: world_size-2 gloo processes, view sharding + frame gather (no rendering: the frame
'renderer' is a deterministic stub, because the rasterizer has no CPU path by design)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_frames(indices, h=6, w=5):
    idx = torch.tensor(indices, dtype=torch.float32).view(-1, 1, 1, 1)
    color = idx + torch.arange(3, dtype=torch.float32).view(1, 3, 1, 1) * 0.25 + torch.zeros(1, 3, h, w)
    masks = (torch.tensor(indices).view(-1, 1, 1, 1) % 3 == torch.arange(2).view(1, 2, 1, 1)).to(torch.uint8) \
        + torch.zeros(1, 2, h, w, dtype=torch.uint8)
    return dict(color=color, masks=masks)


def _worker(rank, world, port, n_views, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pegasus_amd import view_shard as vs
        calls = []

        def render_batch(indices):
            calls.append(list(indices))
            return _stub_frames(indices)
        out = vs.render_sharded(n_views, render_batch, batch, gather=True, dst=0)
        mine = vs.shard_indices(n_views, rank, world)
        assert [i for c in calls for i in c] == mine
        assert all(len(c) <= batch for c in calls)
        if rank == 0:
            expect = _stub_frames(list(range(n_views)))
            ok = all(torch.equal(out[k], expect[k]) for k in expect)
            q.put(("ok" if ok else "mismatch", rank))
        else:
            q.put(("ok" if out is None else "non-root got frames", rank))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [("ok", r) for r in range(world)]


@pytest.mark.parametrize("world,n_views,batch", [(2, 7, 2), (2, 8, 4), (2, 1, 3), (4, 10, 2), (4, 3, 2), (8, 21, 3), (8, 64, 4)])
def test_shard_and_gather(world, n_views, batch):
    """world 2 / 4 / 8, n_views divisible by the world size and not, fewer views than ranks."""
    _spawn(_worker, world, n_views, batch)


def _frame_gather_worker(rank, world, port, q):
    """FrameGather: preallocated send / receive buffers, one gather per batch, two slots in flight, and the
    (rank, i) -> global id rule g = i * world + r of the rank-major receive buffer."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pegasus_amd import view_shard as vs
        cap, rec = 3, 48
        fg = vs.FrameGather(cap, rec, "cpu", dst=0, depth=2)
        ptrs = [t.data_ptr() for t in fg.send] + ([t.data_ptr() for t in fg.recv] if rank == 0 else [])
        ok = True

        def record(g):                                   # a frame's record is a function of its global id
            return ((torch.arange(rec, dtype=torch.int64) * 7 + g * 13) % 251).to(torch.uint8)
        expect = {}
        for step in range(5):
            slot = step & 1
            got = fg.finish(slot)                        # the batch that used this slot two steps ago
            if step >= 2 and rank == 0:
                glob = fg.global_view(slot)              # [cap, world, rec]: [i, r] = global frame i * world + r
                want = expect.pop(step - 2)
                ok = ok and got is fg.recv[slot] and torch.equal(glob.reshape(cap * world, rec), want)
            buf = fg.send_buffer(slot)
            for i in range(cap):
                buf[i] = record(step * 1000 + vs.global_id(rank, i, world))
            expect[step] = torch.stack([record(step * 1000 + g) for g in range(cap * world)])
            fg.start(slot)
        fg.finish_all()
        if rank == 0:
            for step in (3, 4):
                ok = ok and torch.equal(fg.global_view(step & 1).reshape(cap * world, rec), expect[step])
        # nothing was allocated per batch: the buffers are the ones made in the constructor
        ok = ok and ptrs == [t.data_ptr() for t in fg.send] + ([t.data_ptr() for t in fg.recv] if rank == 0 else [])
        ok = ok and fg.batches == 5 and (fg.recv is None) == (rank != 0)
        q.put(("ok" if ok else "mismatch", rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_frame_gather_preallocated_rank_major(world):
    _spawn(_frame_gather_worker, world)


def test_shard_indices_partition():
    from pegasus_amd import view_shard as vs
    for n in (0, 1, 7, 4096):
        for world in (1, 2, 8):
            parts = [vs.shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) == (vs.max_local(n, world) if n else 0)


def _async_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pegasus_amd import view_shard as vs
        B = 3                                   # views per rank and batch (bench.py --gather: one gather per batch)
        ok = True
        inflight = None
        for step in range(3):
            mine = [step * 100 + rank + world * k for k in range(B)]       # global order inside the batch: rank::world
            local = {"rgb": torch.tensor(mine, dtype=torch.uint8).view(-1, 1, 1, 1).expand(B, 4, 5, 3).contiguous(),
                     "depth_mm": (torch.tensor(mine, dtype=torch.int16) * 7 - 300).view(-1, 1, 1).expand(B, 4, 5).contiguous()}
            if inflight is not None:
                finish, works, expect_step = inflight
                for w in works:
                    w.wait()
                got = finish()
                if rank == 0:
                    want = torch.tensor([expect_step * 100 + i for i in range(B * world)])
                    ok = ok and torch.equal(got["rgb"][:, 0, 0, 0].long(), want % 256) \
                        and torch.equal(got["depth_mm"][:, 0, 0].long(), want * 7 - 300) \
                        and got["rgb"].dtype == torch.uint8 and got["depth_mm"].dtype == torch.int16
                else:
                    ok = ok and got is None
            finish, works = vs.gather_frames(local, B * world, dst=0, async_op=True)
            inflight = (finish, works, step)
        for w in inflight[1]:
            w.wait()
        inflight[0]()
        q.put(("ok" if ok else "mismatch", rank))
    finally:
        dist.destroy_process_group()


def test_async_gather_of_quantised_batches():
    """The generic dict form: uint8 / int16 frames as ONE record per frame in ONE gather, asynchronous."""
    _spawn(_async_worker, 2)


def _full_size_worker(rank, world, port, q):
    """FrameGather at the REAL sizes of an 8-rank run: 32 records of 3.84 MB per rank and batch (800x800, K <= 8:
    masks.record_layout's layout restated below), rank 0's receive buffer world x B x record = 0.98 GB.  Exercises the
    buffer arithmetic (offsets beyond 2^31 bytes, the transposed global view) that the 48-byte records above cannot."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pegasus_amd import view_shard as vs
        P = 800 * 800
        a16 = lambda v: (v + 15) // 16 * 16
        rec = a16(3 * P) + a16(2 * P) + a16(1 * P)             # rgb u8 | depth u16 | one mask byte per pixel
        assert rec == 3_840_000
        cap = 32
        fg = vs.FrameGather(cap, rec, "cpu", dst=0, depth=1)
        assert fg.bytes_per_rank_and_batch == cap * rec == 122_880_000
        buf = fg.send_buffer(0)
        # a record is a function of its global id: a ramp whose phase is the id, and the id itself at both ends
        ramp = torch.arange(rec, dtype=torch.int32)
        for i in range(cap):
            g = vs.global_id(rank, i, world)
            buf[i] = ((ramp + g * 97) % 251).to(torch.uint8)
            buf[i, 0], buf[i, -1] = g % 256, g // 256
        fg.start(0)
        got = fg.finish(0)
        ok = True
        if rank == 0:
            assert got.numel() == world * cap * rec == 983_040_000 and got.numel() * 1 > 2**29
            glob = fg.global_view(0)                             # [cap, world, rec]: [i, r] = frame i * world + r
            assert tuple(glob.shape) == (cap, world, rec)
            ids = glob[:, :, 0].to(torch.int64) + 256 * glob[:, :, -1].to(torch.int64)
            ok = ok and torch.equal(ids.reshape(-1), torch.arange(cap * world))
            for g in (0, 1, world, cap * world - 1, cap * world // 2 + 3):       # whole records, first / last / middle
                want = ((ramp + g * 97) % 251).to(torch.uint8)
                want[0], want[-1] = g % 256, g // 256
                ok = ok and torch.equal(glob[g // world, g % world], want)
        q.put(("ok" if ok else "mismatch", rank))
    finally:
        dist.destroy_process_group()


def test_frame_gather_world8_at_the_real_record_size():
    """world 8 x 32 frames x 3.84 MB = 0.98 GB into rank 0 (round-4 verdict item 8)."""
    _spawn(_full_size_worker, 8)
