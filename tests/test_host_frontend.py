"""Host front end (loader, OBJ/KD build, images, camera, buckets) against the reference's recorded outputs."""
import json
import math
import os

import numpy as np
import pytest

from raytracing_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))


@pytest.mark.parametrize("name", sorted(KATS["kd_stats"]))
def test_kd_build_statistics_match_the_reference_printout(load_scene, name):
    # the reference prints these at src/mesh.cpp:187,211-213 ("%lf" -> 6 decimals)
    exp = KATS["kd_stats"][name]
    got = load_scene(name).kd_stats(0)
    assert got["triangles"] == exp["triangles"]
    assert got["max_depth"] == exp["max_depth"]
    assert got["leaves"] == exp["leaves"]
    assert round(got["avg_depth"], 6) == pytest.approx(exp["avg_depth"], abs=5e-7)
    assert round(got["avg_tris"], 6) == pytest.approx(exp["avg_tris"], abs=5e-7)


def test_small_meshes_get_no_tree(load_scene):
    s = load_scene("hw9/nonconvex.qdmg")          # 44 triangles < 50 (src/mesh.cpp:200)
    assert s.desc.n_meshes == 1 and s.desc.meshes[0].kd_root == -1 and s.desc.n_kd_nodes == 0
    assert s.desc.meshes[0].tri_count == 44


def test_mesh_bbox_includes_the_dummy_origin_vertex(load_scene):
    m = load_scene("kdtree_fixed.qdmg").desc.meshes[0]
    for a in range(3):
        assert m.bbox_min[a] <= 0.0 <= m.bbox_max[a]


def test_kd_tree_is_well_formed(load_scene):
    d = load_scene("kdtree_teapot_lowres_noenv.qdmg").ensure_kd().desc
    nodes = np.ctypeslib.as_array(C_cast(d.kd_nodes, d.n_kd_nodes * 4, "u4"), shape=(d.n_kd_nodes * 4,)).reshape(-1, 4)
    meta, count = nodes[:, 2], nodes[:, 3]
    axis, idx = meta & 3, meta >> 2
    leaves = axis == 3
    assert leaves.sum() == 379
    assert (idx[~leaves] + 1 < d.n_kd_nodes).all()
    assert (idx[leaves] + count[leaves] <= d.n_leaf_tris).all()
    assert count[leaves].sum() == d.n_leaf_tris
    # every triangle is referenced by at least one leaf
    refs = np.ctypeslib.as_array(d.leaf_tris, shape=(d.n_leaf_tris,))
    assert set(refs.tolist()) == set(range(992))


def C_cast(ptr, n, kind):
    import ctypes
    t = {"u4": ctypes.c_uint32}[kind]
    return ctypes.cast(ptr, ctypes.POINTER(t))


def test_scene_inventory_and_parser_quirks(load_scene):
    s = load_scene("lecture7_fixed.qdmg")
    d = s.desc
    assert d.n_nodes == 5 and d.n_lights == 1 and d.env_present == 1 and d.env_bilinear == 1
    # "height -0.01" is NOT a recognised Plane key: the plane stays at y = 0 (src/geometry.cpp:43-47)
    plane = d.geoms[d.nodes[0].geom]
    assert plane.type == 0 and plane.p[0] == 0.0 and plane.p[1] == 200.0
    # Reflection defaults (src/shading.h:97) and the glossy override
    refl = [d.shaders[i] for i in range(d.n_shaders) if d.shaders[i].type == 5]
    assert refl[0].p0 == 0.99 and refl[0].p1 == 1.0 and refl[0].i0 == 32
    assert refl[1].p0 == 0.95 and refl[1].p1 == 0.97 and refl[1].i0 == 25
    # translate overwrites, scale accumulates; order as written (src/transform.cpp:18-42)
    sphere = d.nodes[1]
    assert list(sphere.offset) == [100.0, 50.0, 60.0] and sphere.M[0] == 50.0 and sphere.Minv[0] == pytest.approx(1 / 50.0)
    # BitmapTexture stores the inverse scaling (src/texture.cpp:203-204)
    tex = [d.textures[i] for i in range(d.n_textures) if d.textures[i].type == 4][0]
    assert tex.scaling == 1.0 / 100


def test_layered_parsing_with_and_without_blend_texture(load_scene):
    d = load_scene("window_fixed.qdmg").desc
    lay = [d.shaders[i] for i in range(d.n_shaders) if d.shaders[i].type == 7][0]
    assert lay.i1 == 2
    l0, l1 = d.layers[lay.i0], d.layers[lay.i0 + 1]
    assert d.shaders[l0.shader].type == 6 and l0.tex == -1 and list(l0.blend) == [1, 1, 1]
    assert d.shaders[l1.shader].type == 5 and d.textures[l1.tex].type == 5 and d.textures[l1.tex].p0 == 1.6


def test_unknown_class_is_an_error(tmp_path):
    # the shipped scenes' "Reflection" class is unknown to the reference parser (src/scene.cpp:502-506)
    p = tmp_path / "bad.qdmg"
    p.write_text("Camera c {\n position (0,0,0)\n}\nReflection r {\n}\n")
    with pytest.raises(host.SceneError, match="Unknown object class `Reflection'"):
        host.Scene.load(str(p))


def test_missing_required_property_and_file(tmp_path):
    p = tmp_path / "a.qdmg"
    p.write_text("Camera c {\n fov 90\n}\n")
    with pytest.raises(host.SceneError, match="Required property `position'"):
        host.Scene.load(str(p))
    p.write_text("Camera c {\n position (0,0,0)\n}\nMesh m {\n file \"nope.obj\"\n}\n")
    with pytest.raises(host.SceneError, match="Required file not found"):
        host.Scene.load(str(p))


def test_comments_quotes_bools_and_shaderless_nodes(tmp_path):
    p = tmp_path / "a.qdmg"
    p.write_text("""// header
GlobalSettings {
  frameWidth 64   # trailing comment
  wantAA off
  wantAdaptiveAA 0
  useSRGB yes
}
/* block
Sphere hidden {
}
*/
Camera c {
  position (1, 2, 3)
}
Sphere s {
  radius 2
}
Const k {
}
Node with_shader {
  geometry s
  shader k
}
Node without_shader {
  geometry s
}
""")
    s = host.Scene.load(str(p))
    st = s.settings
    assert st.frame_width == 64 and st.frame_height == 480 and st.want_aa == 0 and st.want_adaptive_aa == 0 and st.use_srgb == 1
    assert s.desc.n_geoms == 1 and s.desc.n_nodes == 1      # shader-less node dropped (src/scene.cpp:603-610)
    assert list(s.desc.shaders[0].color) == [0.5, 0.5, 0.5]  # Const default colour GRAY
    assert st.max_trace_depth == 4


def test_camera_corners_follow_begin_frame():
    cam = host.camera_corners((0, 0, 0), 0, 0, 0, 90, 4 / 3)
    scale = math.tan(math.radians(45)) / math.sqrt((4 / 3) ** 2 + 1)
    assert cam.top_left[0] == pytest.approx(-4 / 3 * scale * 1.5) and cam.top_left[1] == pytest.approx(scale * 1.5)
    assert cam.top_left[2] == 1.0 and cam.top_right[0] == -cam.top_left[0] and cam.bottom_left[1] == -cam.top_left[1]


def test_bucket_list_is_serpentine_and_clipped():
    b = host.bucket_list(100, 100)      # 3x3 buckets of 48 (src/sdl.cpp:253-276)
    assert b.shape == (9, 4)
    assert b[0].tolist() == [0, 0, 48, 48] and b[2].tolist() == [96, 0, 100, 48]
    assert b[3].tolist() == [96, 48, 100, 96] and b[5].tolist() == [0, 48, 48, 96]
    assert b[8].tolist() == [96, 96, 100, 100]


def test_exr_piz_decode_is_sane_and_bmp_roundtrip(tmp_path, data_dir):
    sky = host.load_image(os.path.join(data_dir, "env", "forest", "posy.exr"))
    assert sky.shape == (256, 256, 3) and np.isfinite(sky).all() and sky.min() > 0 and sky.max() > 10   # HDR sky
    ground = host.load_image(os.path.join(data_dir, "env", "forest", "negy.exr"))
    assert ground.mean() < sky.mean()
    # 24-bit BMP write -> read: quantisation floor(x*255+0.5), bottom-up rows, BGR (src/bitmap.cpp:181-220)
    rng = np.random.default_rng(1)
    img = rng.random((5, 7, 3), dtype=np.float32) * 1.2 - 0.1
    path = str(tmp_path / "t.bmp")
    host.save_bmp(path, img)
    back = host.load_image(path)
    exp = np.floor(np.clip(img.astype(np.float64), 0, 1) * 255 + 0.5) / np.float32(255)
    assert np.array_equal(back, exp.astype(np.float32))
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + 5 * 24      # rows padded to 4 bytes


def test_exr_write_is_rgba_half_and_reads_back(tmp_path, data_dir):
    """Bitmap::SaveEXR (src/bitmap.cpp:254-277): RGBA half, alpha 1.  The writer emits uncompressed scan lines; the pixels
    must be the floats rounded to half the IEEE way (what OpenEXR's half(float) does; numpy's float16 is the same rule),
    read back by this repo's reader (itself pinned on the reference's PIZ fixtures), and an HDR fixture must survive a
    round trip exactly (its values are halves already).  Parity with OpenEXR's own reader is unpinned (no OpenEXR here)."""
    import struct
    rng = np.random.default_rng(7)
    img = (rng.standard_normal((9, 13, 3)) * np.array([1e-7, 1.0, 3e4])).astype(np.float32)
    img[0, 0] = [0.0, -0.0, 1.0]
    img[0, 1] = [65504.0, 65519.9, 65520.0]          # largest half; rounds down to it; rounds up to infinity
    img[0, 2] = [6.1035156e-05, 5.9604645e-08, 2.9802322e-08]   # smallest normal; smallest subnormal; half of it (ties to even: 0)
    img[0, 3] = [2049.0, 2051.0, 1.0009765625 + 2 ** -12]       # ties to even in both directions
    path = str(tmp_path / "t.exr")
    host.save_exr(path, img)
    with np.errstate(over="ignore"):
        exp = img.astype(np.float16).astype(np.float32)
    back = host.load_image(path)
    assert back.shape == img.shape and np.array_equal(back.view(np.uint32), exp.view(np.uint32))
    raw = open(path, "rb").read()
    assert struct.unpack("<ii", raw[:8]) == (20000630, 2)
    assert b"channels\0chlist\0" in raw and b"compression\0compression\0\x01\0\0\0\0" in raw
    assert len(raw) == raw.index(b"screenWindowWidth") + len(b"screenWindowWidth\0float\0") + 8 + 1 + 9 * 8 + 9 * (8 + 13 * 8)
    first_chunk = struct.unpack("<Q", raw[len(raw) - 9 * (8 + 13 * 8) - 9 * 8:][:8])[0]
    assert struct.unpack("<ii", raw[first_chunk:first_chunk + 8]) == (0, 13 * 8)
    assert raw[first_chunk + 8:first_chunk + 10] == struct.pack("<e", 1.0)       # the A plane comes first
    sky = host.load_image(os.path.join(data_dir, "env", "forest", "posy.exr"))
    host.save_exr(path, sky)
    assert np.array_equal(host.load_image(path), sky)
    with pytest.raises(RuntimeError):
        host.save_exr(str(tmp_path / "no_such_dir" / "t.exr"), img)


def test_heightfield_tables(load_scene):
    d = load_scene("heightfield_fixed.qdmg").desc
    h = d.hfields[0]
    assert (h.width, h.height, h.use_optimization, h.max_k) == (512, 512, 1, 9)
    f32 = np.ctypeslib.as_array(d.hf_f32, shape=(d.n_hf_f32,))
    heights = f32[h.heights_off:h.heights_off + 512 * 512].reshape(512, 512)
    maxh = f32[h.maxheights_off:h.maxheights_off + 512 * 512].reshape(512, 512)
    hm = f32[h.highmap_off:h.highmap_off + 512 * 512 * 16].reshape(512, 512, 16)
    assert h.bbox_min[1] == heights.min() and h.bbox_max[1] == heights.max() and h.bbox_max[0] == 512
    ref = heights.copy()
    ref[:, :-1] = np.maximum(ref[:, :-1], heights[:, 1:]); ref[:-1] = np.maximum(ref[:-1], np.maximum(heights[1:], np.pad(heights[1:, 1:], ((0, 0), (0, 1)), constant_values=-1)))
    assert np.array_equal(maxh, ref)
    assert (hm[:, :, 0] >= heights).all() and (hm[:, :, 8] >= hm[:, :, 0]).all()


def test_synthetic_scene_is_reproducible():
    a = host.Scene.synthetic(16, seed=42, width=64, height=36, max_trace_depth=3)
    b = host.Scene.synthetic(16, seed=42, width=64, height=36, max_trace_depth=3)
    c = host.Scene.synthetic(16, seed=43, width=64, height=36, max_trace_depth=3)
    va = np.ctypeslib.as_array(a.desc.tri_verts, shape=(a.desc.n_tris * 9,))
    vb = np.ctypeslib.as_array(b.desc.tri_verts, shape=(b.desc.n_tris * 9,))
    vc = np.ctypeslib.as_array(c.desc.tri_verts, shape=(c.desc.n_tris * 9,))
    assert a.desc.n_tris == 2 * 16 * 16 and np.array_equal(va, vb) and not np.array_equal(va, vc)
    assert a.settings.max_trace_depth == 3
    # the reference-shaped tree is built on request (the device builds its own from the triangles)
    assert a.desc.meshes[0].kd_root == -1 and a.desc.meshes[0].flags & 4 and a.desc.n_kd_nodes == 0
    assert a.ensure_kd().desc.meshes[0].kd_root >= 0 and a.desc.n_kd_nodes > 0 and a.desc.n_leaf_tris >= a.desc.n_tris


@pytest.mark.parametrize("n", [96, 2240])
def test_synthetic_scene_vertex_buffer_is_the_committed_one(data_dir, synthetic_sha, n):
    """SURVEY.md 8(d) input 5: the generator is reproducible from (n, seed) alone; the SHA-256 of the vertex buffer
    of the bench scene (n = 2240, 10 035 200 triangles) and of the miniature the parity tests use is committed."""
    from conftest import synthetic_vertex_sha256
    s = host.Scene.synthetic(n, seed=42, width=3840, height=2160, max_trace_depth=8,
                             env_folder=os.path.join(data_dir, "env", "forest"))
    try:
        assert s.desc.n_tris == synthetic_sha[f"n{n}"]["triangles"] == 2 * n * n
        assert synthetic_vertex_sha256(s) == synthetic_sha[f"n{n}"]["sha256"]
    finally:
        s.close()


def test_truncated_or_malformed_exr_is_an_error_not_a_crash(data_dir, tmp_path):
    """The EXR reader bounds every header read by the attribute it is in: cutting a file anywhere in its header, or
    declaring attributes shorter than their fixed layout, must come back as an error."""
    src = open(os.path.join(data_dir, "env", "forest", "posx.exr"), "rb").read()
    good = host.load_image(os.path.join(data_dir, "env", "forest", "posx.exr"))
    assert good.shape == (256, 256, 3)
    at = src.index(b"channels\0chlist\0")
    size_at = at + len(b"channels\0chlist\0")
    cases = [src[:n] for n in (4, 9, at + 5, size_at + 2, size_at + 4 + 3, size_at + 4 + 20, 400)]
    cut = bytearray(src); cut[size_at:size_at + 4] = (5).to_bytes(4, "little")      # channel list shorter than one entry
    cases.append(bytes(cut))
    dwat = src.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0")
    bad = bytearray(src); bad[dwat:dwat + 4] = (3).to_bytes(4, "little")             # dataWindow of 3 bytes
    cases.append(bytes(bad))
    for i, blob in enumerate(cases):
        p = tmp_path / f"bad{i}.exr"
        p.write_bytes(blob)
        with pytest.raises(RuntimeError):
            host.load_image(str(p))
