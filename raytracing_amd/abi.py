"""ctypes mirror of include/qdtrace.h and include/qdhost.h (plain C structs, no torch types)."""
import ctypes as C

c_double3 = C.c_double * 3
c_float3 = C.c_float * 3


class QdGeom(C.Structure):
    _fields_ = [("type", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("p", C.c_double * 4)]


class QdNode(C.Structure):
    _fields_ = [("geom", C.c_int32), ("shader", C.c_int32), ("bump_tex", C.c_int32), ("shadow_transparency", C.c_float),
                ("M", C.c_double * 9), ("Minv", C.c_double * 9), ("MinvT", C.c_double * 9), ("offset", c_double3)]


class QdKdNode(C.Structure):
    _fields_ = [("split", C.c_double), ("meta", C.c_uint32), ("count", C.c_uint32)]


class QdMesh(C.Structure):
    _fields_ = [("tri_first", C.c_uint32), ("tri_count", C.c_uint32), ("kd_root", C.c_int32), ("flags", C.c_uint32),
                ("bbox_min", c_double3), ("bbox_max", c_double3),
                ("kd_max_depth", C.c_uint32), ("kd_leaves", C.c_uint32),
                ("kd_depth_sum", C.c_uint64), ("kd_leaf_tris", C.c_uint64)]


class QdHeightfield(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("use_optimization", C.c_int32), ("max_k", C.c_int32),
                ("bbox_min", c_double3), ("bbox_max", c_double3),
                ("heights_off", C.c_uint64), ("maxheights_off", C.c_uint64), ("highmap_off", C.c_uint64),
                ("normals_off", C.c_uint64)]


class QdShader(C.Structure):
    _fields_ = [("type", C.c_int32), ("tex", C.c_int32), ("color", c_float3), ("i0", C.c_int32), ("i1", C.c_int32),
                ("pad_", C.c_int32), ("p0", C.c_double), ("p1", C.c_double)]


class QdLayer(C.Structure):
    _fields_ = [("shader", C.c_int32), ("tex", C.c_int32), ("blend", c_float3)]


class QdTexture(C.Structure):
    _fields_ = [("type", C.c_int32), ("bitmap", C.c_int32), ("color1", c_float3), ("color2", c_float3),
                ("bilinear", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("pad_", C.c_int32),
                ("scaling", C.c_double), ("p0", C.c_double)]


class QdProcLine(C.Structure):
    _fields_ = [("color_u", c_float3), ("color_v", c_float3), ("freq_u", C.c_double), ("freq_v", C.c_double)]


class QdBitmap(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texel_off", C.c_uint64)]


class QdLight(C.Structure):
    _fields_ = [("pos", c_double3), ("intensity", C.c_double)]


class QdCamera(C.Structure):
    _fields_ = [("top_left", c_double3), ("top_right", c_double3), ("bottom_left", c_double3), ("position", c_double3)]


class QdSettings(C.Structure):
    _fields_ = [("frame_width", C.c_int32), ("frame_height", C.c_int32), ("ambient", c_float3),
                ("max_trace_depth", C.c_uint32), ("want_aa", C.c_int32), ("want_adaptive_aa", C.c_int32),
                ("show_aa", C.c_int32), ("use_srgb", C.c_int32), ("aa_debug_color", c_float3), ("pad_", C.c_int32),
                ("aa_threshold", C.c_double)]


def _arr(name, typ, count_type=C.c_uint32):
    return [("n_" + name, count_type), (name, C.POINTER(typ))]


class QdSceneDesc(C.Structure):
    _fields_ = ([("settings", QdSettings), ("camera", QdCamera)]
                + _arr("nodes", QdNode) + _arr("geoms", QdGeom) + _arr("meshes", QdMesh)
                + _arr("hfields", QdHeightfield) + _arr("shaders", QdShader) + _arr("layers", QdLayer)
                + _arr("textures", QdTexture) + _arr("proc_lines", QdProcLine) + _arr("bitmaps", QdBitmap)
                + _arr("lights", QdLight)
                + [("n_tris", C.c_uint64), ("tri_verts", C.POINTER(C.c_double)),
                   ("tri_gnormal", C.POINTER(C.c_double)), ("tri_shade", C.POINTER(C.c_double))]
                + _arr("kd_nodes", QdKdNode, C.c_uint64) + _arr("leaf_tris", C.c_uint32, C.c_uint64)
                + _arr("texels", C.c_float, C.c_uint64) + _arr("hf_f32", C.c_float, C.c_uint64)
                + _arr("hf_f64", C.c_double, C.c_uint64)
                + [("env_present", C.c_int32), ("env_bilinear", C.c_int32), ("env_faces", C.c_int32 * 6)])


class QdStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("depth_killed", C.c_uint64), ("kd_nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("mesh_queries", C.c_uint64), ("kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# Every symbol include/qdtrace.h declares (checked by the CPU test-suite against the built library).
QDTRACE_SYMBOLS = ["qd_init", "qd_set_counting", "qd_scene_check", "qd_scene_setup_info", "qd_scene_create", "qd_render", "qd_frame_begin", "qd_frame_read_rect", "qd_frame_end", "qd_antialias", "qd_antialias_device", "qd_aa_begin_device", "qd_aa_trace_part_async", "qd_aa_end_device", "qd_render_tiles_async", "qd_render_tiles_noalloc_async", "qd_scene_bind_stream", "qd_untile_async",
                   "qd_quantize_bgr8_async", "qd_pack_rgb32_async", "qd_set_kd_refine", "qd_set_kd_build", "qd_set_pipeline", "qd_set_wavefront_budget", "qd_scene_wavefront_info", "qd_set_frame_parts", "qd_scene_frame_parts", "qd_scene_schedule_info", "qd_scene_kd_info", "qd_scene_destroy", "qd_shutdown", "qd_last_error",
                   "qd_comm_unique_id", "qd_comm_create", "qd_comm_destroy", "qd_comm_gather_async", "qd_comm_broadcast_async", "qd_gather_frame_async",
                   "qd_prep_heightfield", "qd_prep_highmap", "qd_prep_differentiate"]
QDHOST_SYMBOLS = ["qdh_load_scene", "qdh_synth_scene", "qdh_scene_desc", "qdh_camera_corners", "qdh_bucket_list",
                  "qdh_save_bmp", "qdh_save_exr", "qdh_load_image", "qdh_free", "qdh_free_scene", "qdh_last_error", "qdh_math_probe", "qdh_set_prep_hooks", "qdh_prep_hook_calls",
                  "qdh_scene_build_kd", "qdh_set_eager_kd"]
