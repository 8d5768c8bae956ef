"""PCIe-inclusive rate of the host-buffer entry point: wall time of qd_render for the whole 4K frame of the bench scene
(device allocation + trace + D2H copy of 99.5 MB into a pageable numpy buffer) next to the kernel time it reports."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device, host
s = host.Scene.synthetic(2240, 42, 3840, 2160, 8, os.path.join(ROOT, "tests/golden/data/env/forest"))
device.init(0)
ds = device.DeviceScene(s)
import ctypes as C
import numpy as np
from raytracing_amd import abi
L = device.lib()
W, H = 3840, 2160
for label, buf in (("contiguous", np.ones((H, W, 3), np.float32)), ("pitch 4096 px", np.ones((H, 4096, 3), np.float32))):
    pitch = buf.shape[1] * 3
    for i in range(3):
        st = abi.QdStats()
        t = time.perf_counter()
        rc = L.qd_render(ds._h, C.byref(s.camera), C.byref(s.settings), 0, 0, W, H, buf.ctypes.data, pitch, None, None, None, C.byref(st))
        dt = time.perf_counter() - t
        assert rc == 0
        print(f"qd_render 3840x2160 into a touched host buffer ({label}): wall {dt * 1e3:.2f} ms, kernel {st.kernel_ms:.2f} ms, "
              f"{st.rays / dt / 1e6:.0f} Mrays/s host-buffer inclusive", flush=True)
