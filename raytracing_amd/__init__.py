"""raytracing_amd: MI355X-native per-pixel trace path for the "Quad Damage" ray tracer.

Layout: csrc/host (scene front end, C++), csrc/device (HIP kernels + C-ABI), host.py / device.py
(ctypes faces of the two libraries).  The CPU oracle lives outside the package, under oracle/.
"""
from . import abi, host  # noqa: F401
