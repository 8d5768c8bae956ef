"""Single-GPU rehearsal of the strong-scaling bench: times the trace kernel of ONE rank's share of the 4K frame
for world sizes 1, 2, 4, 8 (rank 0's tiles only; no collective).  Kernel-side scaling efficiency = T(1) / (N * T(N))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import host, renderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
scene = host.Scene.synthetic(n, 42, 3840, 2160, 8, os.path.join(ROOT, "tests/golden/data/env/forest"))
base = None
worlds = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 2, 4, 8)
for world in worlds:
    for rank in sorted({0, world - 1}):
        fr = renderer.FrameRenderer(scene, tile=32, rank=rank, world=world, local_device=0, collectives="none")
        for _ in range(2):
            fr.render_shard()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
        for a, b in ev:
            a.record(); fr.render_shard(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        rays = fr.stats_dict()["rays"]
        if base is None:
            base = ms
        print(f"world {world} rank {rank}: {ms:7.3f} ms  rays {rays}  kernel-side efficiency {base / (world * ms):.3f}", flush=True)
        fr.dscene.close()
        del fr
