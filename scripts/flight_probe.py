"""Single-GPU rehearsal of strong scaling with frames in flight: the trace kernel of rank 0's share of the 4K frame for
world sizes 1, 2, 4, 8 (no collective), F = 1..4 frames in flight on separate streams / working sets.
usage: flight_probe.py [n] [worlds] [Fmax]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"   # as bench.py does, before the HIP runtime is loaded
import torch
from raytracing_amd import host, renderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
worlds = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 8)
FMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = 12
scene = host.Scene.synthetic(n, 42, 3840, 2160, 8, os.path.join(ROOT, "tests/golden/data/env/forest"))
base = {}
for world in worlds:
    fr = renderer.FrameRenderer(scene, tile=32, rank=0, world=world, local_device=0, frames_in_flight=max(2, FMAX), collectives="none")
    for F in range(1, FMAX + 1):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                with torch.cuda.stream(fr.slots[i % F].stream):
                    fr.render_shard(i % F)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K * 1e3
        base.setdefault(F, dt)
        print(f"world {world} frames in flight {F}: {dt:7.3f} ms/frame   kernel-side efficiency vs world {worlds[0]} at the same F: {base[F] * worlds[0] / (world * dt):.3f}", flush=True)
    fr.check()
    fr.dscene.close()
    del fr
