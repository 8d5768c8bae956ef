"""Frame renderer: one process per GPU, tiles sharded over ranks, framebuffer gathered over RCCL.

torch is plumbing here (device buffers, the current HIP stream, torch.distributed); every pixel is
produced by the HIP kernels behind the C-ABI (raytracing_amd.device).
"""
import ctypes as C

import torch

from . import abi, device, sharding


class _Slot:
    """Buffers of one frame in flight."""

    def __init__(self, fr, stream):
        dev, tile, world = fr.dev, fr.tile, fr.world
        self.stream = stream                                   # None: whatever stream is current at the call
        self.shard = torch.zeros((fr.tpr, tile, tile, 3), dtype=torch.float32, device=dev)
        self.stats = torch.zeros(8, dtype=torch.int64, device=dev)   # qd_stats: 7 counters + kernel_ms slot
        self.frame = torch.zeros((fr.H, fr.W, 3), dtype=torch.float32, device=dev) if fr.rank == 0 else None
        self.gathered = (torch.zeros((world, fr.tpr, tile, tile, 3), dtype=torch.float32, device=dev)
                         if (fr.rank == 0 and fr.exchange) else None)


class FrameRenderer:
    """frames_in_flight = F > 1: submit_frame() rotates over F slots, each with its own HIP stream, buffers and
    (inside the library, qd_scene_bind_stream) working set, so that consecutive frames overlap on the GPU: a frame's
    last ray generations keep only a few waves busy and the next frame's pixels fill the rest of the chip."""

    def __init__(self, scene, tile=32, rank=0, world=1, local_device=0, frames_in_flight=1, collectives="rccl",
                 force_collectives=False, dscene=None, comm=None, streams=None):
        """collectives: "rccl" = the library's own exchange step (qd_comm_*: grouped ncclSend / ncclRecv to rank 0 behind
        the C-ABI; the unique id travels over torch.distributed's process group), "torch" = torch.distributed.gather /
        broadcast, "host" = the same exchange staged through host memory over a gloo group (a rehearsal of N ranks that SHARE
        one GPU, where RCCL refuses two ranks per device: every collective first drains the stream, so nothing overlaps),
        "none" = no exchange at all (rehearsals that time one rank's shard on a lone GPU).  force_collectives
        runs the exchange even at world 1 (a one-rank communicator: tests).  dscene / comm: an existing DeviceScene of this
        scene / an existing device.Comm to use instead of making new ones, streams: the HIP streams
        of the slots (a second renderer for another frame size takes all three from the first: same working sets, same hardware queues)."""
        if not 1 <= frames_in_flight <= 7:
            raise ValueError("frames_in_flight must be 1..7 (libqdtrace keeps 8 working sets per scene, one of them the default stream's)")
        if collectives not in ("rccl", "torch", "host", "none"):
            raise ValueError("collectives must be 'rccl', 'torch', 'host' or 'none'")
        self.collectives = collectives
        self.scene, self.tile, self.rank, self.world = scene, tile, rank, world
        self.dev = torch.device("cuda", local_device)
        torch.cuda.set_device(self.dev)
        device.init(local_device)
        self._own_dscene, self._own_comm = dscene is None, False
        self.dscene = dscene if dscene is not None else device.DeviceScene(scene)
        st = scene.settings
        self.W, self.H = st.frame_width, st.frame_height
        self.tpr = sharding.tiles_per_rank_padded(self.W, self.H, tile, world)
        self.F = frames_in_flight
        self.exchange = (world > 1 or force_collectives) and collectives != "none"
        # One RCCL communicator PER SLOT: the gathers of frames in flight run on F different streams at once, and the operations
        # of one communicator must not be in flight from several streams together (sharding.SlotSchedule).  `comm` (given or
        # made here) is a list of F communicators, created in slot order on every rank; self.comm = slot 0's, which also
        # carries the AA pass's broadcast / gather on the caller's stream.
        self.comms = None
        if self.exchange and collectives == "rccl":
            if comm is not None:
                self.comms = list(comm) if isinstance(comm, (list, tuple)) else [comm]
                if len(self.comms) != self.F:
                    raise ValueError("comm: one communicator per frame in flight")
            else:
                self.comms = [device.Comm(rank, world, exchange=self._exchange_id if world > 1 else None) for _ in range(self.F)]
                self._own_comm = True
        self.comm = self.comms[0] if self.comms else None
        if streams is not None and len(streams) != self.F:
            raise ValueError("streams: one per frame in flight")
        self.slots = [_Slot(self, (streams[i] if streams is not None else torch.cuda.Stream(self.dev)) if self.F > 1 else None) for i in range(self.F)]
        for sl in self.slots:
            if sl.stream is not None:
                self.dscene.bind_stream(sl.stream.cuda_stream)
        self.schedule = sharding.SlotSchedule(self.F)
        s0 = self.slots[0]                                      # the single-frame interface works on slot 0
        self.shard, self.stats, self.frame, self.gathered = s0.shard, s0.stats, s0.frame, s0.gathered

    def render_shard(self, slot=0):
        """Enqueue this rank's tiles on torch's current stream (no host sync)."""
        sl = self.slots[slot]
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.dscene.render_tiles_async(sl.shard.data_ptr(), self.tile, self.rank, self.world,
                                       stats_dev_ptr=sl.stats.data_ptr(), stream=stream)

    def _exchange_id(self, uid):
        """rank 0's 128-byte RCCL unique id -> every rank, over the torch.distributed group the launcher set up."""
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        t = torch.zeros(128, dtype=torch.uint8, device=self.dev if on_gpu else "cpu")
        if uid is not None:
            t.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    # The collectives, in one place (tests on a one-GPU box swap these two for host-staged gloo).
    def _coll_gather(self, src, dst, slot=0):
        """src of every rank -> dst[r] on rank 0 (dst is None elsewhere).  Each peer's block goes straight to the
        root over that peer's own xGMI link, not around a ring.  On the communicator of `slot`."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        if self.comms is not None:
            self.comms[slot].gather_async(src.data_ptr(), dst.data_ptr() if self.rank == 0 else 0, src.numel(), stream=stream)
            return
        import torch.distributed as dist
        if self.collectives == "host":
            torch.cuda.current_stream(self.dev).synchronize()
            mine = src.cpu()
            got = [torch.zeros_like(mine) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(mine, got, dst=0)
            if self.rank == 0:
                dst.copy_(torch.stack(got))
            return
        dist.gather(src, list(dst.unbind(0)) if self.rank == 0 else None, dst=0)

    def _coll_broadcast(self, t):
        if self.comm is not None:
            self.comm.broadcast_async(t.data_ptr(), t.numel(), stream=torch.cuda.current_stream(self.dev).cuda_stream)
            return
        import torch.distributed as dist
        if self.collectives == "host":
            torch.cuda.current_stream(self.dev).synchronize()
            h = t.cpu()
            dist.broadcast(h, src=0)
            t.copy_(h)
            return
        dist.broadcast(t, src=0)

    def _gather_shards(self, sl, slot=0):
        self._coll_gather(sl.shard, sl.gathered, slot)

    def gather(self, slot=0):
        """The one exchange step: compacted shards -> rank 0, then de-interleave."""
        sl = self.slots[slot]
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        if self.exchange:
            self._gather_shards(sl, slot)
            if self.rank != 0:
                return None
            src = sl.gathered
        else:
            if self.world > 1:
                raise RuntimeError("FrameRenderer(collectives='none') has one rank's shard only: there is no frame to assemble")
            src = sl.shard
        device.untile_async(src.data_ptr(), sl.frame.data_ptr(), self.W, self.H, self.tile, self.world, self.tpr,
                            stream=stream)
        return sl.frame

    def submit_frame(self):
        """Enqueue one whole frame (trace + gather + de-interleave) on the next slot's stream and return the slot index;
        no host sync.  Every rank must submit frames in the same order (the gathers are collectives).  The frame of
        slot k is complete after wait(k); it is overwritten by the F-th submit after this one."""
        k = self.schedule.next()
        sl = self.slots[k]
        if sl.stream is None:
            self.render_shard(k)
            self.gather(k)
        else:
            sl.stream.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(sl.stream):
                self.render_shard(k)
                self.gather(k)
        return k

    def settle(self, frames=device.SETTLE_FRAMES):
        """Untimed set-up: lets the library time both scheduling forms on this scene's real frames (qd_set_pipeline 2) so
        that the frames that follow run in the faster one.  The library times frames that ran alone on the GPU, so the
        frames here are rendered one at a time (submit, wait); each form needs two of them (the first loads its kernels).
        Every rank must call it, with the same count (the frames are gathered)."""
        for _ in range(frames):
            k = self.submit_frame()
            self.wait(k)
            torch.cuda.synchronize(self.dev)
            self.dscene.check()

    def wait(self, slot=None):
        """Host-wait for one slot's stream (or all of them) and return that slot's frame (rank 0) or None."""
        for k, sl in enumerate(self.slots):
            if slot is None or k == slot:
                (sl.stream or torch.cuda.current_stream(self.dev)).synchronize()
        return self.slots[slot or 0].frame

    def render_frame(self, antialias=None):
        """RenderScene: every rank traces its tiles, rank 0 gathers; if settings.wantAA (or antialias=True) the AA pass
        follows on the assembled frame (src/main.cpp:106-153 needs the whole first pass): the frame goes back out to all
        ranks, every rank marks (same frame, same list), traces the taps of its slice of the list, rank 0 gathers the
        taps and applies the replace rule."""
        self.render_shard()
        frame = self.gather()
        want = self.scene.settings.want_aa if antialias is None else antialias
        if not want:
            return frame
        cur = torch.cuda.current_stream(self.dev)
        if not self.exchange:
            cur.synchronize()
            self.dscene.antialias_device(frame.data_ptr())
            return frame
        if frame is None:
            if getattr(self, "_aa_frame", None) is None:
                self._aa_frame = torch.zeros((self.H, self.W, 3), dtype=torch.float32, device=self.dev)
            frame = self._aa_frame
        self._coll_broadcast(frame)
        cur.synchronize()
        n = self.dscene.aa_begin(frame.data_ptr())          # identical on every rank: a function of the frame alone
        per, first, count = sharding.aa_slice(n, self.rank, self.world)
        taps = torch.zeros((max(per, 1), 4, 3), dtype=torch.float32, device=self.dev)
        self.dscene.aa_trace_part(first, count, taps.data_ptr(), stream=cur.cuda_stream)
        all_taps = torch.zeros((self.world,) + tuple(taps.shape), dtype=torch.float32, device=self.dev) if self.rank == 0 else None
        self._coll_gather(taps, all_taps)
        cur.synchronize()
        if self.rank == 0:
            self.dscene.aa_end(all_taps.data_ptr(), frame.data_ptr())     # rank-major blocks of `per` pixels = list order
            return frame
        self.dscene.aa_end()
        return None

    def stats_dict(self):
        v = self.stats.cpu().tolist()
        names = [n for n, _ in abi.QdStats._fields_]
        return dict(zip(names[:7], v[:7]))

    def check(self):
        self.dscene.check()

    def close(self):
        """Waits for this renderer's streams, then frees what it made itself (the RCCL communicator, the device scene)."""
        torch.cuda.synchronize(self.dev)
        if self._own_comm and self.comms:
            for c in self.comms:
                c.close()
        self.comm = self.comms = None
        if self._own_dscene and self.dscene is not None:
            self.dscene.close()
        self.dscene = None

    def quantize_bgr8(self):
        """8-bit BGR bottom-up rows of rank 0's frame, as Bitmap::SaveBMP lays them out (src/bitmap.cpp:209-217)."""
        row_bytes = (self.W * 3 + 3) // 4 * 4
        out = torch.zeros((self.H, row_bytes), dtype=torch.uint8, device=self.dev)
        device.quantize_bgr8_async(self.frame.data_ptr(), out.data_ptr(), self.W, self.H, row_bytes,
                                   stream=torch.cuda.current_stream(self.dev).cuda_stream)
        return out
