"""Probe: which frames of FrameRenderer's settle / in-flight loops are enqueued in parts, and what the library's own lone-frame figure says."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
from raytracing_amd import device, host, renderer
DATA = os.path.join(ROOT, "tests", "golden", "data")
sc = host.Scene.load(os.path.join(DATA, "lecture7_fixed.qdmg"))
sc.settings.frame_width, sc.settings.frame_height, sc.settings.want_aa = 3840, 2160, 0
fr = renderer.FrameRenderer(sc, tile=32, rank=0, world=1, local_device=0, frames_in_flight=4, collectives="rccl")
for i in range(8):
    t = time.perf_counter()
    k = fr.submit_frame(); parts = fr.dscene.frame_parts()
    fr.wait(k); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) * 1e3
    print("settle-like frame", i, "slot", k, "parts", parts, "wall %.2f ms" % dt, fr.dscene.schedule_info(), flush=True)
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter(); seen = []
    for i in range(40):
        fr.submit_frame(); seen.append(fr.dscene.frame_parts())
    torch.cuda.synchronize()
    print("in flight: %.2f ms/frame, parts per frame %s" % ((time.perf_counter() - t) * 1e3 / 40, seen), flush=True)
device.set_frame_parts(1)
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(40):
        fr.submit_frame()
    torch.cuda.synchronize()
    print("in flight, parts off: %.2f ms/frame" % ((time.perf_counter() - t) * 1e3 / 40), flush=True)
fr.close()
