"""The multi-rank frame path with the real kernels on a one-GPU box: two ranks, both on cuda:0, run FrameRenderer with
frames in flight -- trace on per-slot HIP streams, gather to rank 0, de-interleave -- and every frame must be the
frame a lone qd_render gives.  RCCL refuses two ranks on one GPU, and gloo's device-tensor gather hangs on this
ROCm build whenever a kernel is still in flight on the stream (a plain matmul is enough), so the ranks use
FrameRenderer(collectives="host"): shard -> host, gloo gather of host tensors, -> device.  Slots, streams, working sets,
submit order and the de-interleave of a two-rank frame are the product's.  bench.py drives the same calls, one rank per
GPU, over RCCL; where two GPUs are visible the last test here does too.  (A device without peers in its process must be
bound by qd_init per rank: local_device = rank.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUM_ORDER_TOL = 2e-6       # two device renders of the same pixel: float additions in a different order


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, flight, out_path, aa=False, rccl=False):
    import datetime
    sys.path.insert(0, ROOT)
    log = open(f"{out_path}.rank{rank}.log", "w")

    def note(msg):
        log.write(msg + "\n"); log.flush()

    import torch
    import torch.distributed as dist
    from raytracing_amd import host, renderer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    note("init")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    note("group up")
    s = host.Scene.load(os.path.join(ROOT, "tests", "golden", "data", "kdtree_teapot_lowres_env.qdmg"))
    if aa:
        s = host.Scene.load(os.path.join(ROOT, "tests", "golden", "data", "hw9", "nonconvex.qdmg"))
        s.settings.want_aa, s.settings.want_adaptive_aa, s.settings.show_aa = 1, 1, 0
    else:
        s.settings.want_aa = 0
    # collectives="host": the exchange staged through host memory over gloo (RCCL refuses two ranks per device)
    # rccl: one GPU per rank and the library's own exchange (only where two GPUs are visible)
    fr = renderer.FrameRenderer(s, tile=32, rank=rank, world=world, local_device=rank if rccl else 0, frames_in_flight=flight,
                                collectives="rccl" if rccl else "host")
    note("scene on device")
    if aa:
        frame = fr.render_frame()
        fr.check()
        note("aa frame complete")
        if rank == 0:
            ref, _ = fr.dscene.render_frame()            # qd_render + qd_antialias on this rank alone
            first, _, _ = fr.dscene.render()
            got = frame.cpu().numpy()
            np.save(out_path, np.array([float(np.abs(got - ref).max()), float(np.abs(got - first).max() > 1e-3)]))
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        fr.dscene.close()
        return
    slots = [fr.submit_frame() for _ in range(2 * flight + 1)][-flight:]
    note("submitted")
    frames = [fr.wait(k) for k in slots]
    fr.check()
    note("frames complete")
    if rank == 0:
        ref, _, _ = fr.dscene.render()
        worst = max(float(np.abs(f.cpu().numpy() - ref).max()) for f in frames)
        np.save(out_path, np.array([worst, float(sorted(slots) == list(range(flight)))]))
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    fr.dscene.close()


def _run_two(tmp_path, flight, aa, rccl=False):
    out = str(tmp_path / "res.npy")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, flight, out, aa, rccl)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=100)
    logs = " | ".join(open(f"{out}.rank{r}.log").read().replace("\n", ",") if os.path.exists(f"{out}.rank{r}.log") else "-" for r in range(2))
    stuck = [p for p in procs if p.is_alive()]
    for p in stuck:
        p.kill()
        p.join()
    assert not stuck, f"ranks did not finish; progress: {logs}"
    assert all(p.exitcode == 0 for p in procs), f"exit codes {[p.exitcode for p in procs]}; progress: {logs}"
    return np.load(out)


@pytest.mark.parametrize("flight", [1, 3])
def test_two_ranks_with_frames_in_flight(tmp_path, flight):
    worst, slots_ok = _run_two(tmp_path, flight, False)
    assert slots_ok == 1.0
    assert worst <= SUM_ORDER_TOL, worst


def test_two_ranks_share_the_aa_taps(tmp_path):
    """wantAA on two ranks: frame gathered, sent back out, both ranks mark, each traces half of the listed pixels' taps,
    rank 0 applies the replace rule (qd_aa_begin_device / qd_aa_trace_part_async / qd_aa_end_device) == one rank's
    qd_render + qd_antialias."""
    worst, changed = _run_two(tmp_path, 1, True)
    assert changed == 1.0
    assert worst <= SUM_ORDER_TOL, worst


def _two_gpus():
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.parametrize("flight,aa", [(1, False), (3, False), (1, True)])
def test_two_ranks_over_rccl_when_two_gpus_are_visible(tmp_path, flight, aa):
    """The same two-rank frames and AA pass with one GPU per rank and the library's RCCL exchange (one communicator per slot
    stream).  Skipped on the one-GPU boxes this build has had so far: multi-rank parity over RCCL is unpinned until it runs."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    worst, flag = _run_two(tmp_path, flight, aa, rccl=True)
    assert flag == 1.0
    assert worst <= SUM_ORDER_TOL, worst
