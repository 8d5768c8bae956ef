"""Frame renderer: one process per GPU, tiles sharded over ranks, framebuffer gathered over RCCL.

torch is plumbing here (device buffers, the current HIP stream, torch.distributed); every pixel is
produced by the HIP kernels behind the C-ABI (raytracing_amd.device).
"""
import ctypes as C

import torch

from . import abi, device, sharding


class FrameRenderer:
    def __init__(self, scene, tile=32, rank=0, world=1, local_device=0):
        self.scene, self.tile, self.rank, self.world = scene, tile, rank, world
        self.dev = torch.device("cuda", local_device)
        torch.cuda.set_device(self.dev)
        device.init(local_device)
        self.dscene = device.DeviceScene(scene)
        st = scene.settings
        self.W, self.H = st.frame_width, st.frame_height
        self.tpr = sharding.tiles_per_rank_padded(self.W, self.H, tile, world)
        self.shard = torch.zeros((self.tpr, tile, tile, 3), dtype=torch.float32, device=self.dev)
        self.stats = torch.zeros(8, dtype=torch.int64, device=self.dev)   # qd_stats: 7 counters + kernel_ms slot
        self.frame = torch.zeros((self.H, self.W, 3), dtype=torch.float32, device=self.dev) if rank == 0 else None
        self.gathered = (torch.zeros((world, self.tpr, tile, tile, 3), dtype=torch.float32, device=self.dev)
                         if (rank == 0 and world > 1) else None)

    def render_shard(self):
        """Enqueue this rank's tiles on torch's current stream (no host sync)."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.dscene.render_tiles_async(self.shard.data_ptr(), self.tile, self.rank, self.world,
                                       stats_dev_ptr=self.stats.data_ptr(), stream=stream)

    def gather(self):
        """The one exchange step: compacted shards -> rank 0 (direct peer->root, not a ring), then de-interleave."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        if self.world > 1:
            import torch.distributed as dist
            if self.rank == 0:
                dist.gather(self.shard, list(self.gathered.unbind(0)), dst=0)
                src = self.gathered
            else:
                dist.gather(self.shard, None, dst=0)
                return None
        else:
            src = self.shard
        device.untile_async(src.data_ptr(), self.frame.data_ptr(), self.W, self.H, self.tile, self.world, self.tpr,
                            stream=stream)
        return self.frame

    def render_frame(self, antialias=None):
        """RenderScene: every rank traces its tiles, rank 0 gathers; if settings.wantAA (or antialias=True) rank 0 then
        runs the AA pass on the assembled frame (src/main.cpp:106-153 needs the whole first pass)."""
        self.render_shard()
        frame = self.gather()
        want = self.scene.settings.want_aa if antialias is None else antialias
        if want and frame is not None:
            torch.cuda.current_stream(self.dev).synchronize()
            self.dscene.antialias_device(frame.data_ptr())
        return frame

    def stats_dict(self):
        v = self.stats.cpu().tolist()
        names = [n for n, _ in abi.QdStats._fields_]
        return dict(zip(names[:7], v[:7]))

    def check(self):
        self.dscene.check()

    def quantize_bgr8(self):
        """8-bit BGR bottom-up rows of rank 0's frame, as Bitmap::SaveBMP lays them out (src/bitmap.cpp:209-217)."""
        row_bytes = (self.W * 3 + 3) // 4 * 4
        out = torch.zeros((self.H, row_bytes), dtype=torch.uint8, device=self.dev)
        device.quantize_bgr8_async(self.frame.data_ptr(), out.data_ptr(), self.W, self.H, row_bytes,
                                   stream=torch.cuda.current_stream(self.dev).cuda_stream)
        return out
