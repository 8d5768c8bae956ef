"""Multi-rank framebuffer path on CPU: two gloo ranks each produce their tile shard (here with the
oracle standing in for the GPU kernels), gather to rank 0 and de-interleave; the result must be the
single-process frame bytewise."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, tile, out_path):
    sys.path.insert(0, ROOT)
    import oracle
    from raytracing_amd import host
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = host.Scene.load(os.path.join(ROOT, "tests", "golden", "data", "window_fixed.qdmg"))
    s.settings.want_aa = 0
    s.settings.frame_width, s.settings.frame_height = 200, 120     # ragged vs the 32-pixel tiles
    W, H = 200, 120
    tpr = sharding.tiles_per_rank_padded(W, H, tile, world)
    shard = torch.zeros((tpr, tile, tile, 3), dtype=torch.float32)
    for k, (x0, y0, x1, y1) in enumerate(sharding.owned_tile_rects(W, H, tile, rank, world)):
        rgb, _, _ = oracle.render_rect(s, x0, y0, x1, y1, threads=1)
        shard[k, : y1 - y0, : x1 - x0] = torch.from_numpy(rgb)
    gathered = [torch.zeros_like(shard) for _ in range(world)] if rank == 0 else None
    dist.gather(shard, gathered, dst=0)
    if rank == 0:
        frame = sharding.untile(torch.stack(gathered).numpy(), W, H, tile, world)
        full, _, _ = oracle.render_rect(s, 0, 0, W, H, threads=2)
        np.save(out_path, np.array([np.array_equal(frame, full), frame.shape == (H, W, 3)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reproduces_the_frame(tmp_path):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), 32, out), nprocs=2, join=True)
    assert np.load(out).all()


def _order_worker(rank, world, port, F, n_frames, out_path):
    """Two ranks drive frames through sharding.SlotSchedule (what FrameRenderer.submit_frame walks) with one gloo group per
    slot standing in for the slot's RCCL communicator: each frame's shard is gathered on its slot's group, asynchronously
    (several gathers in flight, as on the GPU streams), rank 1 deliberately lagging.  Rank 0 checks that every frame
    arrived whole and that both ranks issued the same (frame, slot) sequence."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    groups = [dist.new_group(list(range(world))) for _ in range(F)]          # created in slot order on every rank
    sched = sharding.SlotSchedule(F)
    pending, ok = {}, True
    for n in range(n_frames):
        k = sched.next()
        if k in pending:                                   # the slot's previous gather must be done before its buffers are reused
            work, got, frame_no = pending.pop(k)
            work.wait()
            if rank == 0:
                ok &= all(int(got[r][0]) == frame_no * 10 + r for r in range(world))
        if rank == 1 and n % 2 == 0:
            import time; time.sleep(0.01)
        shard = torch.full((4,), float(n * 10 + rank))
        got = [torch.zeros(4) for _ in range(world)] if rank == 0 else None
        pending[k] = (dist.gather(shard, got, dst=0, group=groups[k], async_op=True), got, n)
    for k, (work, got, frame_no) in pending.items():
        work.wait()
        if rank == 0:
            ok &= all(int(got[r][0]) == frame_no * 10 + r for r in range(world))
    logs = [None] * world
    dist.all_gather_object(logs, sched.log)
    if rank == 0:
        np.save(out_path, np.array([ok, all(l == logs[0] for l in logs), logs[0] == [(n, n % F) for n in range(n_frames)]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_issue_the_gathers_of_frames_in_flight_in_one_order(tmp_path):
    out = str(tmp_path / "order.npy")
    mp.spawn(_order_worker, args=(2, _free_port(), 3, 11, out), nprocs=2, join=True)
    assert np.load(out).all()


def test_tile_ownership_is_a_partition():
    for (W, H, tile, world) in [(3840, 2160, 32, 8), (992, 744, 32, 3), (100, 60, 16, 4), (17, 9, 8, 2)]:
        seen = np.zeros((H, W), np.int32)
        total = 0
        for r in range(world):
            rects = sharding.owned_tile_rects(W, H, tile, r, world)
            assert len(rects) == sharding.tiles_owned(W, H, tile, r, world) <= sharding.tiles_per_rank_padded(W, H, tile, world)
            total += len(rects)
            for (x0, y0, x1, y1) in rects:
                seen[y0:y1, x0:x1] += 1
        assert (seen == 1).all()
        tx, ty = sharding.tile_grid(W, H, tile)
        assert total == tx * ty


def test_aa_slices_cover_the_list_in_rank_order():
    for n in (0, 1, 7, 8, 9, 1000, 12345):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                per, first, count = sharding.aa_slice(n, r, world)
                assert first == min(n, r * per) and 0 <= count <= per
                if count:
                    assert first == nxt
                    nxt = first + count
            assert nxt == n
            assert per * world >= n
