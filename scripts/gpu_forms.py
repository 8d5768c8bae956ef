"""Lone-frame kernel time of both scheduling forms for every fixture scene at its own size (data for the first-frame prior
of the adaptive choice).  usage: python scripts/gpu_forms.py [WxH]"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from raytracing_amd import device, host

device.init(0)
size = sys.argv[1] if len(sys.argv) > 1 else None
for path in sorted(glob.glob(os.path.join(ROOT, "tests/golden/data/*.qdmg")) + glob.glob(os.path.join(ROOT, "tests/golden/data/*/*.qdmg"))):
    s = host.Scene.load(path)
    s.settings.want_aa = 0
    if size:
        s.settings.frame_width, s.settings.frame_height = map(int, size.split("x"))
    ds = device.DeviceScene(s)
    ms = []
    for form in (0, 1):
        device.set_pipeline(form)
        t = []
        for _ in range(4):
            _, _, st = ds.render()
            t.append(st["kernel_ms"] if isinstance(st, dict) else st.kernel_ms)
        ms.append(min(t[1:]))
    device.set_pipeline(2)
    ds.close()
    ds = device.DeviceScene(s)                      # a fresh handle: what a caller that renders one image gets
    ds.render()
    first = ds.schedule_info()["form"]
    nodes = ds.kd_info()["nodes"]
    print(f"{os.path.relpath(path, ROOT + '/tests/golden/data'):42s} {s.settings.frame_width}x{s.settings.frame_height}  form0 {ms[0]:8.3f} ms  form1 {ms[1]:8.3f} ms  -> {'wavefront' if ms[1] < ms[0] else 'one-kernel'}  {ms[0] / ms[1]:.2f}  first frame in form {first}  kd nodes {nodes}", flush=True)
    ds.close()
