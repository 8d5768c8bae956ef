"""Development aid: device-built KD tree (qd_set_kd_build 1) against the host-built one (2) -- sizes, build times, hits."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, host
DATA = os.path.join(ROOT, "tests", "golden", "data")
device.init(0)
which = sys.argv[1:] or ["kdtree_fixed.qdmg", "hw9/dragon.qdmg", "synth200", "synth2240"]
for name in which:
    if name.startswith("synth"):
        sc = host.Scene.synthetic(int(name[5:]), seed=42, width=640, height=360, max_trace_depth=4, env_folder=os.path.join(DATA, "env", "forest"))
    else:
        sc = host.Scene.load(os.path.join(DATA, name))
    out = {}
    for mode in (1, 2):
        device.set_kd_build(mode)
        t = time.perf_counter(); ds = device.DeviceScene(sc); dt = time.perf_counter() - t
        rgb, hits, st = ds.render(hits=True)
        out[mode] = (ds.kd_info(), ds.setup_info(), dt, rgb, hits)
        print(name, "mode", mode, ds.kd_info(), {k: round(v, 4) for k, v in ds.setup_info().items()}, "create %.3f s" % dt, "rays", st["rays"], flush=True)
        ds.close()
    a, b = out[1], out[2]
    print("   same sizes:", a[0] == b[0], " hit node/tri equal:", bool((a[4][0] == b[4][0]).all() and (a[4][1] == b[4][1]).all()),
          " dist bits equal:", bool((a[4][2].view(np.int64) == b[4][2].view(np.int64)).all()), " max colour diff %.3g" % np.abs(a[3] - b[3]).max(), flush=True)
