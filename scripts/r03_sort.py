"""Experiment (profiles/r03_experiments.md section 8): the wavefront form with every secondary generation ordered for
coherence before its walk (QD_WF_SORT, raytracing_amd/csrc/device/wfsort_lab.h).  One process per mode: the library
reads the switch once.  Prints parity against the one-kernel form; the library prints the device times when the scene closes.
usage: QD_WF_SORT=<1|2|3> python scripts/r03_sort.py <synth|scene.qdmg> [width height] [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, host
DATA = os.path.join(ROOT, "tests", "golden", "data")
name = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 3 else None
H = int(sys.argv[3]) if len(sys.argv) > 3 else None
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 6
device.init(0)
if name == "synth":
    sc = host.Scene.synthetic(2240, seed=42, width=W or 3840, height=H or 2160, max_trace_depth=8, env_folder=os.path.join(DATA, "env", "forest"))
else:
    sc = host.Scene.load(os.path.join(DATA, name))
    if W:
        sc.settings.frame_width, sc.settings.frame_height = W, H
sc.settings.want_aa = 0
ds = device.DeviceScene(sc)
device.set_pipeline(0)
ref, _, sref = ds.render()
device.set_pipeline(1)
for _ in range(frames):
    got, _, st = ds.render()
print(name, "QD_WF_SORT", os.environ.get("QD_WF_SORT"), "rays", st["rays"], "same counts", st["rays"] == sref["rays"] and st["shadow_rays"] == sref["shadow_rays"],
      "max colour diff %.3g" % np.abs(got - ref).max(), flush=True)
ds.close()
