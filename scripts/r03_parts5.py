"""Probe under rocprofv3 --kernel-trace: a burst of frames in flight whose first frame is enqueued in parts (or not: argv[1] = parts)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
from raytracing_amd import device, host, renderer
DATA = os.path.join(ROOT, "tests", "golden", "data")
parts = int(sys.argv[1])
sc = host.Scene.load(os.path.join(DATA, "lecture7_fixed.qdmg"))
sc.settings.frame_width, sc.settings.frame_height, sc.settings.want_aa = 3840, 2160, 0
device.init(0)
device.set_pipeline(1)
device.set_frame_parts(parts)
fr = renderer.FrameRenderer(sc, tile=32, rank=0, world=1, local_device=0, frames_in_flight=4, collectives="rccl")
for i in range(8):
    k = fr.submit_frame(); fr.wait(k); torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.synchronize()
    time.sleep(0.05)
    t = time.perf_counter(); seen = []
    for i in range(16):
        fr.submit_frame(); seen.append(fr.dscene.frame_parts())
    th = time.perf_counter()
    torch.cuda.synchronize()
    print("burst: host enqueue %.2f ms, total %.2f ms = %.2f ms/frame, parts %s" % ((th - t) * 1e3, (time.perf_counter() - t) * 1e3, (time.perf_counter() - t) * 1e3 / 16, seen), flush=True)
fr.close()
