"""Diagnostic: render one frame of the bench scene through qd_render (prints [qd profile] shares with the QD_PROFILE build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, host
arg = sys.argv[1] if len(sys.argv) > 1 else "2240"
if arg.endswith(".qdmg"):                      # a fixture scene instead of the synthetic mesh
    s = host.Scene.load(os.path.join(ROOT, "tests/golden/data", arg))
    s.settings.want_aa = 0
else:
    s = host.Scene.synthetic(int(arg), 42, 3840, 2160, 8, os.path.join(ROOT, "tests/golden/data/env/forest"))
if len(sys.argv) > 3:
    device.use_library(os.path.join(ROOT, sys.argv[3]))       # a QD_PROFILE build under build_variants/
device.init(0)
device.set_pipeline(0)
ds = device.DeviceScene(s)
rect = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 0, s.settings.frame_width, s.settings.frame_height]
for i in range(2):
    rgb, _, st = ds.render(*rect)
    print({k: st[k] for k in ("rays", "kernel_ms")}, flush=True)
