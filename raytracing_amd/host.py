"""Python face of the host front end (libqdhost.so): scene loading -> flat qd_scene_desc."""
import ctypes as C
import os

import numpy as np

from . import abi

_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "lib", "libqdhost.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m raytracing_amd.build host`")
        L = C.CDLL(path)
        L.qdh_load_scene.restype = C.c_void_p
        L.qdh_load_scene.argtypes = [C.c_char_p]
        L.qdh_synth_scene.restype = C.c_void_p
        L.qdh_synth_scene.argtypes = [C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.qdh_scene_desc.restype = C.POINTER(abi.QdSceneDesc)
        L.qdh_scene_desc.argtypes = [C.c_void_p]
        L.qdh_free_scene.argtypes = [C.c_void_p]
        L.qdh_scene_build_kd.restype = C.c_int
        L.qdh_scene_build_kd.argtypes = [C.c_void_p]
        L.qdh_set_eager_kd.restype = None
        L.qdh_set_eager_kd.argtypes = [C.c_int]
        L.qdh_last_error.restype = C.c_char_p
        L.qdh_camera_corners.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.POINTER(abi.QdCamera)]
        L.qdh_bucket_list.restype = C.c_int
        L.qdh_bucket_list.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]
        L.qdh_save_bmp.restype = C.c_int
        L.qdh_save_bmp.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_size_t]
        L.qdh_save_exr.restype = C.c_int
        L.qdh_save_exr.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_size_t]
        L.qdh_load_image.restype = C.POINTER(C.c_float)
        L.qdh_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.qdh_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


class SceneError(RuntimeError):
    pass


class _PrepHooks(C.Structure):
    _fields_ = [("heightfield", C.c_void_p), ("highmap", C.c_void_p), ("differentiate", C.c_void_p)]


def prep_on_device(enable=True):
    """Scenes loaded afterwards get their heightfield tables and differentiated bump maps from the device library's
    kernels (qdh_set_prep_hooks with qd_prep_*; the device must be initialised) instead of the CPU code; same bits."""
    L = lib()
    L.qdh_set_prep_hooks.argtypes = [C.POINTER(_PrepHooks)]
    L.qdh_set_prep_hooks.restype = None
    if not enable:
        L.qdh_set_prep_hooks(None)
        return
    from . import device
    D = device.lib()
    hooks = _PrepHooks(*[C.cast(getattr(D, n), C.c_void_p).value for n in ("qd_prep_heightfield", "qd_prep_highmap", "qd_prep_differentiate")])
    L.qdh_set_prep_hooks(C.byref(hooks))


class Scene:
    """A loaded scene: owns the flat description handed to qd_scene_create / the oracle."""

    def __init__(self, handle):
        self._h = handle
        self.desc_ptr = lib().qdh_scene_desc(handle)
        self.desc = self.desc_ptr.contents

    @classmethod
    def load(cls, path):
        h = lib().qdh_load_scene(os.fsencode(path))
        if not h:
            raise SceneError(lib().qdh_last_error().decode())
        return cls(h)

    @classmethod
    def synthetic(cls, n, seed=42, width=3840, height=2160, max_trace_depth=8, env_folder=None):
        h = lib().qdh_synth_scene(n, seed, width, height, max_trace_depth,
                                  os.fsencode(env_folder) if env_folder else None)
        if not h:
            raise SceneError(lib().qdh_last_error().decode())
        return cls(h)

    @property
    def settings(self):
        return self.desc.settings

    @property
    def camera(self):
        return self.desc.camera

    def ensure_kd(self):
        """The reference-shaped KD trees of the meshes (qdh_scene_build_kd): built on first request.  The device library
        builds its own tree from the triangles and does not need them; the CPU oracle and qd_set_kd_build(0) do."""
        if lib().qdh_scene_build_kd(self._h) != 0:
            raise SceneError(lib().qdh_last_error().decode())
        return self

    def kd_stats(self, mesh=0):
        self.ensure_kd()
        m = self.desc.meshes[mesh]
        leaves = m.kd_leaves
        return {"triangles": m.tri_count, "max_depth": m.kd_max_depth, "leaves": leaves,
                "avg_depth": (m.kd_depth_sum / leaves) if leaves else 0.0,
                "avg_tris": (m.kd_leaf_tris / leaves) if leaves else 0.0}

    def close(self):
        if self._h:
            lib().qdh_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bucket_list(width, height):
    n = lib().qdh_bucket_list(width, height, None, 0)
    arr = (C.c_int32 * (4 * n))()
    lib().qdh_bucket_list(width, height, arr, n)
    return np.ctypeslib.as_array(arr).reshape(n, 4).copy()


def save_bmp(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    if lib().qdh_save_bmp(os.fsencode(path), rgb.ctypes.data_as(C.POINTER(C.c_float)), w, h, w * 3) != 0:
        raise RuntimeError(lib().qdh_last_error().decode())


def save_exr(path, rgb):
    """Bitmap::SaveEXR (src/bitmap.cpp:254-277): RGBA half, alpha 1, uncompressed scan lines."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    if lib().qdh_save_exr(os.fsencode(path), rgb.ctypes.data_as(C.POINTER(C.c_float)), w, h, w * 3) != 0:
        raise RuntimeError(lib().qdh_last_error().decode())


def load_image(path):
    w, h = C.c_uint32(), C.c_uint32()
    p = lib().qdh_load_image(os.fsencode(path), C.byref(w), C.byref(h))
    if not p:
        raise RuntimeError(lib().qdh_last_error().decode())
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    lib().qdh_free(p)
    return out


def camera_corners(position, yaw, pitch, roll, fov, aspect):
    cam = abi.QdCamera()
    pos = (C.c_double * 3)(*position)
    lib().qdh_camera_corners(pos, yaw, pitch, roll, fov, aspect, C.byref(cam))
    return cam
