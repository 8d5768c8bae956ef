"""CPU oracle (checker only): ctypes face of oracle/liboracle.so.

TEST INFRASTRUCTURE.  Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg only; nothing under raytracing_amd/ imports this package.
"""
import ctypes as C
import os

import numpy as np

from raytracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("raytrace_calls", "depth_killed", "shadow_queries", "mesh_queries",
                                          "bbox_tests", "tri_tests", "kd_nodes_visited")]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["rays"] = d["raytrace_calls"] - d["depth_killed"] + d["shadow_queries"]
        return d


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m raytracing_amd.build oracle`")
        L = C.CDLL(path)
        P = C.POINTER
        L.orc_render_rect.restype = C.c_int
        L.orc_render_rect.argtypes = [P(abi.QdSceneDesc), P(abi.QdCamera), P(abi.QdSettings), C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, P(OrcStats)]
        L.orc_render_pixels.restype = C.c_int
        L.orc_render_pixels.argtypes = [P(abi.QdSceneDesc), P(abi.QdCamera), P(abi.QdSettings), C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(OrcStats)]
        L.orc_render_frame.restype = C.c_int
        L.orc_render_frame.argtypes = [P(abi.QdSceneDesc), C.c_void_p, C.c_int, P(OrcStats)]
        L.orc_convert_to_8bit.restype = C.c_uint
        L.orc_convert_to_8bit.argtypes = [C.c_double]
        L.orc_to_rgb32.restype = None
        L.orc_to_rgb32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_camera_rays.restype = C.c_int
        L.orc_camera_rays.argtypes = [P(abi.QdSceneDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_trace_rays.restype = C.c_int
        L.orc_trace_rays.argtypes = [P(abi.QdSceneDesc), C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_random01_bulk.restype = None
        L.orc_random01_bulk.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pixel_paths.restype = None
        L.orc_pixel_paths.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_unit_funcs.restype = C.c_int
        L.orc_unit_funcs.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


REF_RECORD = np.dtype([("node", "<i4"), ("pad", "<i4"), ("dist", "<f8"), ("ip", "<f8", 3), ("normal", "<f8", 3),
                       ("u", "<f8"), ("v", "<f8"), ("rgb", "<f4", 3), ("padf", "<f4")])


def trace_rays(scene, rays):
    """Raytrace's node loop + bump + Shade for explicit rays (n x 6 f64): records in oracle/ref_driver.cpp's layout."""
    scene.ensure_kd()          # the reference-shaped trees: Mesh::Intersect's recursive walk (src/mesh.cpp:58-92) needs them
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    out = np.zeros(len(rays), REF_RECORD)
    if lib().orc_trace_rays(scene.desc_ptr, len(rays), rays.ctypes.data, out.ctypes.data) != 0:
        raise RuntimeError("orc_trace_rays failed")
    return out


def render_rect(scene, x0, y0, x1, y1, camera=None, settings=None, hits=False, rng_mode=1, threads=None):
    """Trace the pixel rectangle; returns (rgb[h,w,3] f32, hits or None, stats dict)."""
    scene.ensure_kd()          # the reference-shaped trees: Mesh::Intersect's recursive walk (src/mesh.cpp:58-92) needs them
    threads = threads or min(8, os.cpu_count() or 1)
    w, h = x1 - x0, y1 - y0
    rgb = np.zeros((h, w, 3), np.float32)
    hn = np.full((h, w), -2, np.int32) if hits else None
    ht = np.full((h, w), -2, np.int32) if hits else None
    hd = np.zeros((h, w), np.float64) if hits else None
    st = OrcStats()
    rc = lib().orc_render_rect(scene.desc_ptr, C.byref(camera) if camera is not None else None,
                               C.byref(settings) if settings is not None else None, x0, y0, x1, y1,
                               rgb.ctypes.data, w * 3, hn.ctypes.data if hits else None, ht.ctypes.data if hits else None,
                               hd.ctypes.data if hits else None, rng_mode, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render_rect failed")
    return rgb, ((hn, ht, hd) if hits else None), st.as_dict()


def render_pixels(scene, xy, camera=None, settings=None, hits=False, rng_mode=1, threads=None):
    scene.ensure_kd()          # the reference-shaped trees: Mesh::Intersect's recursive walk (src/mesh.cpp:58-92) needs them
    threads = threads or min(8, os.cpu_count() or 1)
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    n = xy.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    hn = np.full(n, -2, np.int32) if hits else None
    ht = np.full(n, -2, np.int32) if hits else None
    hd = np.zeros(n, np.float64) if hits else None
    st = OrcStats()
    rc = lib().orc_render_pixels(scene.desc_ptr, C.byref(camera) if camera is not None else None,
                                 C.byref(settings) if settings is not None else None, n, xy.ctypes.data, rgb.ctypes.data,
                                 hn.ctypes.data if hits else None, ht.ctypes.data if hits else None,
                                 hd.ctypes.data if hits else None, rng_mode, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render_pixels failed")
    return rgb, ((hn, ht, hd) if hits else None), st.as_dict()


def render_frame(scene, rng_mode=0):
    """The reference's Render(): simple pass + AA pass, bucket order, single thread."""
    scene.ensure_kd()          # the reference-shaped trees: Mesh::Intersect's recursive walk (src/mesh.cpp:58-92) needs them
    s = scene.settings
    rgb = np.zeros((s.frame_height, s.frame_width, 3), np.float32)
    st = OrcStats()
    if lib().orc_render_frame(scene.desc_ptr, rgb.ctypes.data, rng_mode, C.byref(st)) != 0:
        raise RuntimeError("orc_render_frame failed")
    return rgb, st.as_dict()


def to_rgb32(rgb, srgb=False, shifts=(16, 8, 0)):
    """Color::toRGB32 / toSRGB32 (src/color.cpp:16-36) for every pixel of float32[..., 3] -> uint32[...]."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros(rgb.shape[:-1], np.uint32)
    lib().orc_to_rgb32(rgb.ctypes.data, out.size, 1 if srgb else 0, shifts[0], shifts[1], shifts[2], out.ctypes.data)
    return out
