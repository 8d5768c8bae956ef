"""Python face of the HIP trace library (libqdtrace.so).  No fallback: a missing library or GPU is an error."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_OVERRIDE = None          # development only: scripts/ A/B another build of the same kernels through use_library()
_INIT_DEVICE = None
_KD_BUILD, _KD_REFINE_LEAF = 1, 2       # what qd_set_kd_build / qd_set_kd_refine were last given (DeviceScene: does the library need the caller's tree?)
SETTLE_FRAMES = 6            # lone frames after which the adaptive choice of the scheduling form has both figures (2 per form + 2 to average)


class DeviceError(RuntimeError):
    pass


def _torch_first():
    """A process that uses both this library and PyTorch must load PyTorch's HIP runtime first: the wheel bundles its own
    libamdhip64 under the same soname as /opt/rocm's, the dynamic linker keeps whichever came first for both, and torch on
    the system runtime reports "No HIP GPUs are available".  (The library itself needs no torch: a C / C++ host links
    /opt/rocm's runtime and nothing else.)"""
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


def lib():
    global _LIB
    if _LIB is None:
        path = _LIB_OVERRIDE or os.path.join(_HERE, "lib", "libqdtrace.so")
        if not os.path.exists(path):
            raise DeviceError(f"{path} is missing: the HIP extension was not built (python -m raytracing_amd.build device)")
        _torch_first()
        L = C.CDLL(path)
        P = C.POINTER
        L.qd_last_error.restype = C.c_char_p
        L.qd_init.argtypes = [C.c_int]
        L.qd_set_counting.argtypes = [C.c_int]
        L.qd_scene_create.argtypes = [P(abi.QdSceneDesc), P(C.c_void_p)]
        L.qd_scene_destroy.argtypes = [C.c_void_p]
        L.qd_scene_check.argtypes = [C.c_void_p]
        L.qd_render.argtypes = [C.c_void_p, P(abi.QdCamera), P(abi.QdSettings), C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, P(abi.QdStats)]
        L.qd_frame_begin.argtypes = [C.c_void_p, P(abi.QdCamera), P(abi.QdSettings), C.c_void_p]
        L.qd_frame_read_rect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.qd_frame_end.argtypes = [C.c_void_p, P(abi.QdStats)]
        L.qd_antialias.argtypes = [C.c_void_p, P(abi.QdCamera), P(abi.QdSettings), C.c_void_p, C.c_size_t, P(abi.QdStats)]
        L.qd_antialias_device.argtypes = [C.c_void_p, P(abi.QdCamera), P(abi.QdSettings), C.c_void_p, P(abi.QdStats)]
        L.qd_aa_begin_device.argtypes = [C.c_void_p, P(abi.QdSettings), C.c_void_p, P(C.c_uint32)]
        L.qd_aa_trace_part_async.argtypes = [C.c_void_p, P(abi.QdCamera), P(abi.QdSettings), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.qd_aa_end_device.argtypes = [C.c_void_p, P(abi.QdSettings), C.c_void_p, C.c_void_p]
        L.qd_render_tiles_async.argtypes = [C.c_void_p, P(abi.QdCamera), P(abi.QdSettings), C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.qd_render_tiles_noalloc_async.argtypes = L.qd_render_tiles_async.argtypes
        L.qd_scene_bind_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.qd_untile_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.qd_quantize_bgr8_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        L.qd_pack_rgb32_async.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.qd_set_kd_refine.argtypes = [C.c_int, C.c_double]
        L.qd_set_pipeline.argtypes = [C.c_int]
        L.qd_set_kd_build.argtypes = [C.c_int]
        L.qd_set_wavefront_budget.argtypes = [C.c_uint64, C.c_uint32]
        L.qd_scene_wavefront_info.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_uint32), P(C.c_int)]
        L.qd_set_frame_parts.argtypes = [C.c_int]
        L.qd_scene_frame_parts.argtypes = [C.c_void_p, P(C.c_uint32)]
        L.qd_scene_schedule_info.argtypes = [C.c_void_p, P(C.c_int), P(C.c_double), P(C.c_double)]
        L.qd_prep_heightfield.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.qd_prep_highmap.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.qd_prep_differentiate.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.qd_scene_setup_info.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double)]
        L.qd_scene_kd_info.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
        L.qd_comm_unique_id.argtypes = [C.c_void_p]
        L.qd_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_void_p)]
        L.qd_comm_destroy.argtypes = [C.c_void_p]
        L.qd_comm_gather_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.qd_comm_broadcast_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.qd_gather_frame_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        for name in abi.QDTRACE_SYMBOLS:
            if name not in ("qd_last_error", "qd_shutdown", "qd_scene_destroy"):
                getattr(L, name).restype = C.c_int
        _LIB = L
    return _LIB


def use_library(path):
    """Development aid for scripts/ (A/B of kernel builds): load `path` instead of the in-tree library.  Must be called
    before the first use; the product never calls it and no environment variable reaches it."""
    global _LIB_OVERRIDE
    if _LIB is not None:
        raise DeviceError("use_library: the device library is already loaded")
    _LIB_OVERRIDE = path


def _check(rc, what):
    if rc != 0:
        raise DeviceError(f"{what}: {lib().qd_last_error().decode()}")


def init(device=0):
    global _INIT_DEVICE
    _check(lib().qd_init(device), "qd_init")
    _INIT_DEVICE = device


def set_counting(on):
    _check(lib().qd_set_counting(1 if on else 0), "qd_set_counting")


def set_kd_build(mode):
    """Device KD tree of scenes created afterwards: 1 = whole tree by the surface-area heuristic, built on the device (default);
    2 = the same tree built by host threads; 0 = the caller's planes with refined leaves."""
    global _KD_BUILD
    _check(lib().qd_set_kd_build(mode), "qd_set_kd_build")
    _KD_BUILD = mode


def set_pipeline(pipeline):
    """0 = one persistent kernel with per-lane ray queues, 1 = wavefront pipeline (generations of rays in HBM queues),
    2 = the faster of the two, measured per scene (library default)."""
    _check(lib().qd_set_pipeline(pipeline), "qd_set_pipeline")


def set_wavefront_budget(queue_bytes=0, max_entries=0):
    """Bytes the ray queues of the wavefront form may take per working set (0 = 16 GB: frames beyond it are rendered in bands)
    and, for tests, a cap on a queue's entries that the band sizes ignore (bands then overflow and are rendered again in place)."""
    _check(lib().qd_set_wavefront_budget(queue_bytes, max_entries), "qd_set_wavefront_budget")


def set_frame_parts(parts=0):
    """qd_set_frame_parts: concurrent parts a lone wavefront frame is enqueued as (0 = default 4, 1 = off)."""
    _check(lib().qd_set_frame_parts(parts), "qd_set_frame_parts")


def set_kd_refine(max_leaf_tris=2, step_cost=0.6):
    """Leaf refinement of the device copy of KD trees for scenes created afterwards (0 = caller's tree unchanged)."""
    global _KD_REFINE_LEAF
    _check(lib().qd_set_kd_refine(max_leaf_tris, step_cost), "qd_set_kd_refine")
    _KD_REFINE_LEAF = max_leaf_tris


class DeviceScene:
    """Scene buffers resident in HBM (qd_scene_create)."""

    def __init__(self, scene):
        if _INIT_DEVICE is None:
            init(0)
        self.host_scene = scene
        if (_KD_BUILD == 0 or _KD_REFINE_LEAF <= 0) and hasattr(scene, "ensure_kd"):
            scene.ensure_kd()          # these modes start from the caller's (reference-shaped) tree, which the front end builds on request
        h = C.c_void_p()
        _check(lib().qd_scene_create(scene.desc_ptr, C.byref(h)), "qd_scene_create")
        self._h = h

    def render(self, x0=0, y0=0, x1=None, y1=None, camera=None, settings=None, hits=False):
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        x1 = st.frame_width if x1 is None else x1
        y1 = st.frame_height if y1 is None else y1
        w, h = x1 - x0, y1 - y0
        rgb = np.zeros((h, w, 3), np.float32)
        hn = np.full((h, w), -2, np.int32) if hits else None
        ht = np.full((h, w), -2, np.int32) if hits else None
        hd = np.zeros((h, w), np.float64) if hits else None
        stats = abi.QdStats()
        _check(lib().qd_render(self._h, C.byref(cam), C.byref(st), x0, y0, x1, y1, rgb.ctypes.data, w * 3,
                               hn.ctypes.data if hits else None, ht.ctypes.data if hits else None,
                               hd.ctypes.data if hits else None, C.byref(stats)), "qd_render")
        return rgb, ((hn, ht, hd) if hits else None), stats.as_dict()

    def antialias(self, rgb, camera=None, settings=None):
        """RenderScene's AA pass on a finished first-pass frame (float32[H, W, 3], C-contiguous); in place."""
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        if rgb.dtype != np.float32 or not rgb.flags["C_CONTIGUOUS"] or rgb.shape != (st.frame_height, st.frame_width, 3):
            raise DeviceError("antialias: rgb must be C-contiguous float32[frame_height, frame_width, 3]")
        stats = abi.QdStats()
        _check(lib().qd_antialias(self._h, C.byref(cam), C.byref(st), rgb.ctypes.data, st.frame_width * 3, C.byref(stats)), "qd_antialias")
        return stats.as_dict()

    def antialias_device(self, rgb_dev_ptr, camera=None, settings=None):
        """The AA pass on a contiguous W x H x 3 float frame in device memory, in place (synchronous)."""
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        stats = abi.QdStats()
        _check(lib().qd_antialias_device(self._h, C.byref(cam), C.byref(st), C.c_void_p(rgb_dev_ptr), C.byref(stats)), "qd_antialias_device")
        return stats.as_dict()

    def aa_begin(self, rgb_dev_ptr, settings=None):
        """Mark + list on the assembled first-pass frame (device); returns the number of pixels whose taps are to be traced."""
        st = settings if settings is not None else self.host_scene.settings
        n = C.c_uint32()
        _check(lib().qd_aa_begin_device(self._h, C.byref(st), C.c_void_p(rgb_dev_ptr), C.byref(n)), "qd_aa_begin_device")
        return int(n.value)

    def aa_trace_part(self, first_pixel, n_pixels, taps_dev_ptr, stats_dev_ptr=None, stream=None, camera=None, settings=None):
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        _check(lib().qd_aa_trace_part_async(self._h, C.byref(cam), C.byref(st), first_pixel, n_pixels, C.c_void_p(taps_dev_ptr or 0),
                                            C.c_void_p(stats_dev_ptr or 0), C.c_void_p(stream or 0)), "qd_aa_trace_part_async")

    def aa_end(self, taps_dev_ptr=None, rgb_dev_ptr=None, settings=None):
        st = settings if settings is not None else self.host_scene.settings
        _check(lib().qd_aa_end_device(self._h, C.byref(st), C.c_void_p(taps_dev_ptr or 0), C.c_void_p(rgb_dev_ptr or 0)), "qd_aa_end_device")

    def render_frame(self, camera=None, settings=None):
        """RenderScene (src/main.cpp:106-161): first pass, then the AA pass when settings.wantAA is set."""
        st = settings if settings is not None else self.host_scene.settings
        rgb, _, stats = self.render(camera=camera, settings=settings)
        if st.want_aa:
            aa = self.antialias(rgb, camera=camera, settings=settings)
            for k in ("rays", "primary_rays", "shadow_rays", "depth_killed", "kd_nodes_visited", "tris_tested", "mesh_queries", "kernel_ms"):
                stats[k] += aa[k]
        return rgb, stats

    def frame_begin(self, camera=None, settings=None, stream=None):
        """Enqueue the whole first pass of the frame (qd_frame_begin); rectangles are then read with frame_read_rect."""
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        _check(lib().qd_frame_begin(self._h, C.byref(cam), C.byref(st), C.c_void_p(stream or 0)), "qd_frame_begin")

    def frame_read_rect(self, x0, y0, x1, y1, out=None):
        """Rectangle [x0,x1) x [y0,y1) of the frame begun last, as float32[h, w, 3] (`out`: a view to fill, last axis contiguous)."""
        w, h = x1 - x0, y1 - y0
        if out is None:
            out = np.empty((h, w, 3), np.float32)
        if out.dtype != np.float32 or out.shape != (h, w, 3) or out.strides[2] != 4 or out.strides[1] != 12 or out.strides[0] % 4:
            raise DeviceError("frame_read_rect: out must be float32[h, w, 3] with contiguous pixels")
        _check(lib().qd_frame_read_rect(self._h, x0, y0, x1, y1, out.ctypes.data, out.strides[0] // 4), "qd_frame_read_rect")
        return out

    def frame_end(self):
        stats = abi.QdStats()
        _check(lib().qd_frame_end(self._h, C.byref(stats)), "qd_frame_end")
        return stats.as_dict()

    def render_buckets(self, on_bucket=None, camera=None, settings=None, launch_per_bucket=False):
        """SimpleRender (src/main.cpp:85-104): the frame bucket by bucket in GetBucketList order (src/sdl.cpp:253-276);
        after each bucket `on_bucket((x0, y0, x1, y1), frame)` plays DisplayVFBRect -- returning False stops the render
        there, as the reference does when the user closes the window (src/main.cpp:99-100), and the call returns
        (frame, False) with the buckets finished so far.  The frame is traced by ONE launch ahead of the loop
        (qd_frame_begin) and the loop reads finished buckets; launch_per_bucket=True traces each bucket by a qd_render
        call of its own instead (the reference's literal shape: a 48 x 48 launch leaves most of the GPU empty)."""
        from . import host
        st = settings if settings is not None else self.host_scene.settings
        W, H = st.frame_width, st.frame_height
        frame = np.zeros((H, W, 3), np.float32)
        if not launch_per_bucket:
            self.frame_begin(camera=camera, settings=settings)
        try:
            for (x0, y0, x1, y1) in host.bucket_list(W, H):
                x0, y0, x1, y1 = int(x0), int(y0), int(x1), int(y1)
                if launch_per_bucket:
                    frame[y0:y1, x0:x1] = self.render(x0, y0, x1, y1, camera=camera, settings=settings)[0]
                else:
                    self.frame_read_rect(x0, y0, x1, y1, out=frame[y0:y1, x0:x1])
                if on_bucket is not None and on_bucket((x0, y0, x1, y1), frame) is False:
                    return frame, False
            return frame, True
        finally:
            if not launch_per_bucket:
                self.frame_end()

    def render_tiles_async(self, rgb_dev_ptr, tile, phase, stride, stats_dev_ptr=None, stream=None,
                           camera=None, settings=None):
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        _check(lib().qd_render_tiles_async(self._h, C.byref(cam), C.byref(st), tile, phase, stride,
                                           C.c_void_p(rgb_dev_ptr), C.c_void_p(stats_dev_ptr or 0),
                                           C.c_void_p(stream or 0)), "qd_render_tiles_async")

    def render_tiles_noalloc_async(self, rgb_dev_ptr, tile, phase, stride, stats_dev_ptr=None, stream=None,
                                   camera=None, settings=None):
        """qd_render_tiles_async for callers that cannot allocate: working memory is the library's static device array."""
        st = settings if settings is not None else self.host_scene.settings
        cam = camera if camera is not None else self.host_scene.camera
        _check(lib().qd_render_tiles_noalloc_async(self._h, C.byref(cam), C.byref(st), tile, phase, stride,
                                                   C.c_void_p(rgb_dev_ptr), C.c_void_p(stats_dev_ptr or 0),
                                                   C.c_void_p(stream or 0)), "qd_render_tiles_noalloc_async")

    def bind_stream(self, stream, max_trace_depth=None):
        """Reserve a working set of the scene for launches on `stream` (frames in flight: one stream per frame)."""
        d = self.host_scene.settings.max_trace_depth if max_trace_depth is None else max_trace_depth
        _check(lib().qd_scene_bind_stream(self._h, d, C.c_void_p(stream or 0)), "qd_scene_bind_stream")

    def check(self):
        _check(lib().qd_scene_check(self._h), "qd_scene_check")

    def schedule_info(self):
        """The adaptive choice of the scheduling form: form of the last launch, measured ns per pixel of each form."""
        f, a, b = C.c_int(), C.c_double(), C.c_double()
        _check(lib().qd_scene_schedule_info(self._h, C.byref(f), C.byref(a), C.byref(b)), "qd_scene_schedule_info")
        return {"form": int(f.value), "ns_per_pixel": [float(a.value), float(b.value)]}

    def frame_parts(self):
        """Concurrent parts the scene's latest launch was enqueued as (qd_scene_frame_parts)."""
        n = C.c_uint32(0)
        _check(lib().qd_scene_frame_parts(self._h, C.byref(n)), "qd_scene_frame_parts")
        return int(n.value)

    def wavefront_info(self, width=None, height=None):
        """Bands a frame of this size takes in the wavefront form, and whether a band ever outgrew its queues."""
        st = self.host_scene.settings
        b, o = C.c_uint32(), C.c_int()
        _check(lib().qd_scene_wavefront_info(self._h, width or st.frame_width, height or st.frame_height, C.byref(b), C.byref(o)), "qd_scene_wavefront_info")
        return {"bands": int(b.value), "overflowed": bool(o.value)}

    def setup_info(self):
        """Seconds qd_scene_create spent building the device KD trees / uploading the scene."""
        a, b = C.c_double(), C.c_double()
        _check(lib().qd_scene_setup_info(self._h, C.byref(a), C.byref(b)), "qd_scene_setup_info")
        return {"kd_build_s": float(a.value), "upload_s": float(b.value)}

    def kd_info(self):
        v = [C.c_uint64() for _ in range(4)]
        _check(lib().qd_scene_kd_info(self._h, *[C.byref(x) for x in v]), "qd_scene_kd_info")
        return dict(zip(("nodes", "leaf_refs", "leaves", "max_leaf"), (int(x.value) for x in v)))

    def close(self):
        if self._h:
            lib().qd_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def untile_async(src_ptr, dst_ptr, width, height, tile, stride, tiles_per_rank_padded, stream=None):
    _check(lib().qd_untile_async(C.c_void_p(src_ptr), C.c_void_p(dst_ptr), width, height, tile, stride,
                                 tiles_per_rank_padded, C.c_void_p(stream or 0)), "qd_untile_async")


def quantize_bgr8_async(rgb_ptr, out_ptr, width, height, row_bytes, stream=None):
    _check(lib().qd_quantize_bgr8_async(C.c_void_p(rgb_ptr), C.c_void_p(out_ptr), width, height, row_bytes,
                                        C.c_void_p(stream or 0)), "qd_quantize_bgr8_async")


def tiles_owned(width, height, tile, phase, stride):
    n_tiles = ((width + tile - 1) // tile) * ((height + tile - 1) // tile)
    return (n_tiles - phase + stride - 1) // stride if n_tiles > phase else 0


def pack_rgb32_async(rgb_ptr, pitch_floats, out_ptr, width, height, use_srgb=False, shifts=(16, 8, 0), stream=None):
    """Color::toRGB32 / toSRGB32 per pixel of a device image (what DisplayVFBRect shows)."""
    _check(lib().qd_pack_rgb32_async(C.c_void_p(rgb_ptr), pitch_floats, C.c_void_p(out_ptr), width, height, 1 if use_srgb else 0,
                                     shifts[0], shifts[1], shifts[2], C.c_void_p(stream or 0)), "qd_pack_rgb32_async")


class Comm:
    """RCCL communicator of this process' GPU (qd_comm_create): one process per GPU.  `exchange(id_bytes_or_None)` is how
    the 128-byte unique id travels from rank 0 to the others: it is called with rank 0's id on rank 0 and with None
    elsewhere, and returns the id on every rank (torch.distributed, MPI, a file -- the library does not care)."""

    def __init__(self, rank, world, exchange=None):
        if _INIT_DEVICE is None:
            raise DeviceError("Comm: device.init(local_device) first")
        self.rank, self.world = rank, world
        uid = None
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            _check(lib().qd_comm_unique_id(buf), "qd_comm_unique_id")
            uid = bytes(buf)
        if world > 1:
            if exchange is None:
                raise DeviceError("Comm: world > 1 needs an `exchange` callable for the unique id")
            uid = exchange(uid)
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
            raise DeviceError("Comm: the exchanged unique id must be 128 bytes")
        h = C.c_void_p()
        idbuf = (C.c_uint8 * 128).from_buffer_copy(bytes(uid))
        _check(lib().qd_comm_create(idbuf, rank, world, C.byref(h)), "qd_comm_create")
        self._h = h

    def gather_async(self, src_ptr, dst_ptr, count, stream=None):
        _check(lib().qd_comm_gather_async(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr or 0), count, C.c_void_p(stream or 0)), "qd_comm_gather_async")

    def broadcast_async(self, buf_ptr, count, stream=None):
        _check(lib().qd_comm_broadcast_async(self._h, C.c_void_p(buf_ptr), count, C.c_void_p(stream or 0)), "qd_comm_broadcast_async")

    def gather_frame_async(self, shard_ptr, gathered_ptr, frame_ptr, width, height, tile, stream=None):
        _check(lib().qd_gather_frame_async(self._h, C.c_void_p(shard_ptr), C.c_void_p(gathered_ptr or 0), C.c_void_p(frame_ptr or 0),
                                           width, height, tile, C.c_void_p(stream or 0)), "qd_gather_frame_async")

    def close(self):
        if self._h:
            lib().qd_comm_destroy(self._h)
            self._h = None
