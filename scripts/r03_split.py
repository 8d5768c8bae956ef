"""Probe: the latency of a LONE frame when it is enqueued as K concurrent parts (tile-interleaved shards, one stream and
working set each) instead of one launch sequence.  No library change: qd_render_tiles_async with phase k of stride K.
usage: r03_split.py <synth|scene.qdmg> [width height]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
from raytracing_amd import device, host, sharding
DATA = os.path.join(ROOT, "tests", "golden", "data")
name = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 3 else None
H = int(sys.argv[3]) if len(sys.argv) > 3 else None
device.init(0)
if name == "synth":
    sc = host.Scene.synthetic(2240, seed=42, width=W or 3840, height=H or 2160, max_trace_depth=8, env_folder=os.path.join(DATA, "env", "forest"))
else:
    sc = host.Scene.load(os.path.join(DATA, name))
    if W:
        sc.settings.frame_width, sc.settings.frame_height = W, H
sc.settings.want_aa = 0
W, H = sc.settings.frame_width, sc.settings.frame_height
ds = device.DeviceScene(sc)
dev = torch.device("cuda:0")
TILE, REPS = 32, 8
for form in (0, 1):
    device.set_pipeline(form)
    line = []
    for K in (1, 2, 3, 4):
        tpr = sharding.tiles_per_rank_padded(W, H, TILE, K)
        bufs = [torch.zeros((tpr, TILE, TILE, 3), dtype=torch.float32, device=dev) for _ in range(K)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(REPS + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                ds.render_tiles_async(bufs[k].data_ptr(), TILE, k, K, stream=streams[k].cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if rep >= 2:
                best = min(best, dt)
        ds.check()
        line.append("K=%d %.2f ms" % (K, best))
        del bufs, streams
    print(name, W, H, "form", form, " | ".join(line), flush=True)
ds.close()
