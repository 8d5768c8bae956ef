"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle on the same scenes.

Bar (BASELINE.json north_star): primary-ray hit node / triangle IDs bit-exact except documented
near-tie rays (< 0.01 % of pixels); per-channel colour error <= 1e-4 absolute (pixels whose
secondary rays took a different discrete branch are counted with the near-ties, same 0.01 % budget).
"""
import os

import numpy as np
import pytest

import oracle
from raytracing_amd import device, host, sharding

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
COLOUR_TOL = 1e-4
NEAR_TIE_BUDGET = 1e-4     # 0.01 % of pixels
SUM_ORDER_TOL = 2e-6       # two device renders of the same pixel: float additions in a different order

SCENES = ["hw9/nonconvex.qdmg", "hw9/axe_test.qdmg", "hw9/testmesh.qdmg", "hw9/dragon.qdmg", "window_fixed.qdmg", "bumpmap_fixed.qdmg", "lecture7_fixed.qdmg",
          "heightfield_fixed.qdmg", "kdtree_teapot_lowres_env.qdmg", "kdtree_teapot_hires_sah_noenv.qdmg",
          "kdtree_fixed.qdmg", "all_classes.qdmg", "ref_pin.qdmg", "csg_generic.qdmg"]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    device.init(0)
    yield


def compare(s, ds, x0, y0, x1, y1):
    rgb, (hn, ht, hd), st = ds.render(x0, y0, x1, y1, hits=True)
    orgb, (ohn, oht, ohd), ost = oracle.render_rect(s, x0, y0, x1, y1, hits=True, rng_mode=1, threads=16)
    n = hn.size
    id_mismatch = int(((hn != ohn) | (ht != oht)).sum())
    dist_mismatch = int((hd.view(np.uint64) != ohd.view(np.uint64)).sum())
    err = np.abs(rgb.astype(np.float64) - orgb.astype(np.float64)).max(axis=2)
    bad = int((~(err <= COLOUR_TOL)).sum())      # NaN counts as bad
    return dict(n=n, id_mismatch=id_mismatch, dist_mismatch=dist_mismatch, bad_colour=bad, max_err=float(np.nanmax(err)),
                gpu=st, cpu=ost, rgb=rgb, orgb=orgb)


@pytest.mark.parametrize("name", SCENES)
def test_scene_matches_oracle(load_scene, name):
    s = load_scene(name)
    s.settings.want_aa = 0
    W, H = s.settings.frame_width, s.settings.frame_height
    ds = device.DeviceScene(s)
    try:
        r = compare(s, ds, 0, 0, W, H)
    finally:
        ds.close()
    budget = max(1, int(r["n"] * NEAR_TIE_BUDGET))
    assert r["id_mismatch"] <= budget, r["id_mismatch"]
    assert r["dist_mismatch"] <= budget, r["dist_mismatch"]
    assert r["bad_colour"] <= budget, (r["bad_colour"], r["max_err"])
    # the ray population is the reference's: same number of traced radiance + shadow rays (near-ties aside)
    assert abs(r["gpu"]["rays"] - r["cpu"]["rays"]) <= max(16, r["cpu"]["rays"] * 1e-5)
    assert r["gpu"]["depth_killed"] == pytest.approx(r["cpu"]["depth_killed"], abs=16)
    assert r["gpu"]["primary_rays"] == r["n"]


def test_synthetic_mesh_scene_matches_oracle(data_dir):
    s = host.Scene.synthetic(96, seed=42, width=320, height=180, max_trace_depth=8,
                             env_folder=os.path.join(data_dir, "env", "forest"))
    ds = device.DeviceScene(s)
    try:
        r = compare(s, ds, 0, 0, 320, 180)
    finally:
        ds.close()
    assert r["id_mismatch"] <= 5 and r["dist_mismatch"] <= 5 and r["bad_colour"] <= 5, (r["id_mismatch"], r["bad_colour"], r["max_err"])
    assert r["gpu"]["rays"] > 2 * r["n"]      # glass: secondary + shadow rays are really traced


@pytest.mark.parametrize("kd_build", [1, 0])
def test_full_size_bench_scene_matches_oracle(data_dir, synthetic_sha, kd_build):
    """BASELINE configs[4] at FULL size (10 035 200 triangles, 3840x2160, maxTraceDepth 8, forest cubemap): the very scene
    bench.py times.  The whole frame is rendered on the device with primary hits; a strided 1/64 pixel subset
    (SURVEY.md 8(d) input 5) is compared with the oracle -- hit node / triangle / bits of the distance inside the 0.01 %
    near-tie budget, colour <= 1e-4 -- and the ray population (radiance, shadow, depth-killed) on four rectangles.
    Exercises what the miniature cannot: 28-bit record indices, multi-GB leaf_verts offsets, deep stacks -- with the device
    tree built from the triangles (kd_build 1, the default: 20 M nodes) and with the caller's midpoint planes kept and
    refined (kd_build 0: 46 M nodes, 34 M leaf references)."""
    from conftest import synthetic_vertex_sha256
    s = host.Scene.synthetic(2240, seed=42, width=3840, height=2160, max_trace_depth=8,
                             env_folder=os.path.join(data_dir, "env", "forest"))
    assert s.desc.n_tris == 10035200
    assert synthetic_vertex_sha256(s) == synthetic_sha["n2240"]["sha256"]
    W, H = 3840, 2160
    device.set_kd_build(kd_build)
    try:
        ds = device.DeviceScene(s)
    finally:
        device.set_kd_build(1)
    try:
        info = ds.kd_info()
        if kd_build == 0:
            assert info["nodes"] > 30_000_000 and info["leaf_refs"] * 72 > 2**31      # really the big tree
        else:
            assert 10_000_000 < info["nodes"] < 30_000_000 and info["leaf_refs"] >= 10_035_200
        rgb, (hn, ht, hd), st = ds.render(hits=True)
        assert st["primary_rays"] == W * H and st["rays"] > 2 * W * H
        if os.environ.get("QD_PIPELINE") in (None, "", "2"):
            assert ds.schedule_info()["form"] == 0           # a deep tree: the first frame runs in the one-kernel form
        ys, xs = np.meshgrid(np.arange(5, H, 8), np.arange(3, W, 8), indexing="ij")
        xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
        assert len(xy) == W * H // 64
        orgb, (ohn, oht, ohd), ost = oracle.render_pixels(s, xy, hits=True, rng_mode=1, threads=16)
        g = lambda a: a[xy[:, 1], xy[:, 0]]
        budget = max(1, int(len(xy) * NEAR_TIE_BUDGET))
        id_mismatch = int(((g(hn) != ohn) | (g(ht) != oht)).sum())
        dist_mismatch = int((g(hd).view(np.uint64) != ohd.view(np.uint64)).sum())
        err = np.abs(g(rgb).astype(np.float64) - orgb.astype(np.float64)).max(axis=1)
        bad = int((~(err <= COLOUR_TOL)).sum())
        print(f"full-size subset: {len(xy)} px, id mismatches {id_mismatch}, distance mismatches {dist_mismatch}, "
              f"bad colours {bad}, max err {float(np.nanmax(err)):.3e}, mesh-hit pixels {int((g(ht) >= 0).sum())}")
        assert int((g(ht) >= 0).sum()) > len(xy) // 4            # the mesh fills a good part of the frame
        assert id_mismatch <= budget and dist_mismatch <= budget, (id_mismatch, dist_mismatch)
        assert bad <= budget, (bad, float(np.nanmax(err)))
        # ray population on rectangles spread over sky, mesh horizon, mesh interior and floor
        for (x0, y0) in [(64, 32), (1800, 700), (3500, 1100), (900, 1900)]:
            x1, y1 = x0 + 256, y0 + 64
            part, _, pst = ds.render(x0, y0, x1, y1)
            opart, _, opst = oracle.render_rect(s, x0, y0, x1, y1, rng_mode=1, threads=16)
            assert np.abs(part - rgb[y0:y1, x0:x1]).max() <= SUM_ORDER_TOL
            assert int((~(np.abs(part.astype(np.float64) - opart).max(axis=2) <= COLOUR_TOL)).sum()) <= 2
            assert abs(pst["rays"] - opst["rays"]) <= 8 and abs(pst["shadow_rays"] - opst["shadow_queries"]) <= 8, (pst, opst)
            assert abs(pst["depth_killed"] - opst["depth_killed"]) <= 8
    finally:
        ds.close()
        s.close()


def test_ragged_rectangles_and_pitch(load_scene):
    s = load_scene("window_fixed.qdmg")
    s.settings.want_aa = 0
    ds = device.DeviceScene(s)
    try:
        full, _, _ = ds.render()
        for (x0, y0, x1, y1) in [(0, 0, 1, 1), (5, 7, 6, 20), (633, 471, 640, 480), (100, 100, 173, 131)]:
            part, hits, st = ds.render(x0, y0, x1, y1, hits=True)
            # a pixel's colour is the sum of its ray tasks' contributions; tasks that were shared between lanes of a wave
            # are added in a different order, so two renders agree to float rounding, not to the bit
            assert np.abs(part - full[y0:y1, x0:x1]).max() <= SUM_ORDER_TOL
            assert st["primary_rays"] == (x1 - x0) * (y1 - y0)
        with pytest.raises(device.DeviceError):
            ds.render(10, 10, 10, 20)            # empty rectangle is rejected, like a bad bucket
        with pytest.raises(device.DeviceError):
            ds.render(0, 0, 641, 480)
    finally:
        ds.close()


def test_tile_sharded_frame_equals_single_rectangle(load_scene):
    """render(N shards) == render(1 GPU) up to the order of float additions: pixels are independent."""
    import torch
    s = load_scene("heightfield_fixed.qdmg")
    s.settings.want_aa = 0
    W, H = s.settings.frame_width, s.settings.frame_height
    ds = device.DeviceScene(s)
    try:
        full, _, _ = ds.render()
        tile = 32
        for world in (1, 3):
            tpr = sharding.tiles_per_rank_padded(W, H, tile, world)
            gathered = torch.zeros((world, tpr, tile, tile, 3), dtype=torch.float32, device="cuda:0")
            for r in range(world):
                ds.render_tiles_async(gathered[r].data_ptr(), tile, r, world)
            frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            device.untile_async(gathered.data_ptr(), frame.data_ptr(), W, H, tile, world, tpr)
            torch.cuda.synchronize()
            ds.check()
            got = frame.cpu().numpy()
            assert np.array_equal(sharding.untile(gathered.cpu().numpy(), W, H, tile, world), got)   # de-interleave is exact
            assert np.abs(got - full).max() <= SUM_ORDER_TOL
    finally:
        ds.close()


def test_4k_heightfield_properties(load_scene):
    """BASELINE config 4 at 3840x2160: too big for a full oracle frame in seconds, so check
    size-independent properties: a strided pixel sample against the oracle, run-to-run agreement."""
    s = load_scene("heightfield_fixed.qdmg")
    st = s.settings
    old = (st.frame_width, st.frame_height, st.want_aa)
    st.frame_width, st.frame_height, st.want_aa = 3840, 2160, 0
    ds = device.DeviceScene(s)
    try:
        a, _, sa = ds.render()
        b, _, sb = ds.render()
        # colours are summed with float atomics in arrival order: two renders agree to rounding, not to the bit
        assert np.abs(a - b).max() <= SUM_ORDER_TOL and sa["rays"] == sb["rays"]
        ys, xs = np.meshgrid(np.arange(3, 2160, 29), np.arange(5, 3840, 31), indexing="ij")
        xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
        orgb, _, _ = oracle.render_pixels(s, xy, rng_mode=1, threads=16)
        err = np.abs(a[xy[:, 1], xy[:, 0]].astype(np.float64) - orgb).max(axis=1)
        assert (~(err <= COLOUR_TOL)).sum() <= max(1, int(len(xy) * 2 * NEAR_TIE_BUDGET)), err.max()
    finally:
        ds.close()
        st.frame_width, st.frame_height, st.want_aa = old


def test_counting_build_tallies_heightfield_steps(load_scene):
    """bench.py prices the heightfield workload on DDA steps and cells tested; the counting build reports
    them in kd_nodes_visited / tris_tested, and counting changes nothing else."""
    s = load_scene("heightfield_fixed.qdmg")
    old = _aa_settings(s, want_aa=0)
    ds = device.DeviceScene(s)
    try:
        plain, _, st0 = ds.render(200, 300, 456, 428)
        device.set_counting(True)
        try:
            counted, _, st1 = ds.render(200, 300, 456, 428)
        finally:
            device.set_counting(False)
    finally:
        ds.close()
        _aa_settings(s, **old)
    assert st0["kd_nodes_visited"] == 0 and st0["tris_tested"] == 0
    assert st1["rays"] == st0["rays"] and np.abs(counted - plain).max() <= SUM_ORDER_TOL
    assert st1["kd_nodes_visited"] > st1["rays"] and 0 < st1["tris_tested"] < st1["kd_nodes_visited"]


def test_quantize_kernel_matches_convert_to_8bit():
    import torch
    rng = np.random.default_rng(3)
    img = (rng.random((37, 53, 3), dtype=np.float32) * 1.4 - 0.2)
    img[0, 0] = [0.5, 1.0, 0.0]
    t = torch.from_numpy(img).cuda()
    row_bytes = (53 * 3 + 3) // 4 * 4
    out = torch.zeros((37, row_bytes), dtype=torch.uint8, device="cuda:0")
    device.quantize_bgr8_async(t.data_ptr(), out.data_ptr(), 53, 37, row_bytes)
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :53 * 3].reshape(37, 53, 3)
    exp = np.floor(np.clip(img.astype(np.float64), 0, 1) * 255 + 0.5).astype(np.uint8)[::-1, :, ::-1]   # bottom-up, BGR
    assert np.array_equal(got, exp)
    assert oracle.lib().orc_convert_to_8bit(0.5) == got[36, 0, 2]


def test_pack_rgb32_kernel_matches_to_rgb32():
    """qd_pack_rgb32_async == Color::toRGB32 / toSRGB32 per pixel (what DisplayVFBRect shows), bit for bit."""
    import torch
    rng = np.random.default_rng(5)
    img = (rng.random((41, 67, 3), dtype=np.float32) * 1.3 - 0.15)
    img[0, :8, 0] = [0.0, 0.0031308, 0.0031309, 0.5, 1.0, 0.2, -1.0, 4.0]
    t = torch.from_numpy(img).cuda()
    out = torch.zeros((41, 67), dtype=torch.int32, device="cuda:0")
    for srgb in (False, True):
        for shifts in ((16, 8, 0), (0, 8, 16)):
            device.pack_rgb32_async(t.data_ptr(), 67 * 3, out.data_ptr(), 67, 41, use_srgb=srgb, shifts=shifts)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.to_rgb32(img, srgb=srgb, shifts=shifts)), (srgb, shifts)


@pytest.mark.parametrize("launch_per_bucket", [False, True])
def test_bucket_by_bucket_render_and_cancel(load_scene, launch_per_bucket):
    """SimpleRender's loop (src/main.cpp:85-104): buckets in GetBucketList order, a display hook per bucket whose False
    stops the render; the finished frame equals the single-rectangle frame.  Both shapes of the loop: one launch ahead of
    it and rectangle reads (qd_frame_begin / qd_frame_read_rect), or a qd_render call per bucket."""
    s = load_scene("kdtree_teapot_lowres_env.qdmg")
    old = _aa_settings(s, want_aa=0)
    ds = device.DeviceScene(s)
    try:
        full, _, _ = ds.render()
        seen = []
        frame, done = ds.render_buckets(lambda r, f: seen.append(r), launch_per_bucket=launch_per_bucket)
        assert done and seen == [tuple(int(v) for v in b) for b in host.bucket_list(s.settings.frame_width, s.settings.frame_height)]
        assert np.abs(frame - full).max() <= SUM_ORDER_TOL
        part, done = ds.render_buckets(lambda r, f: len(seen) < 0 or r != seen[2], launch_per_bucket=launch_per_bucket)
        assert not done
        x0, y0, x1, y1 = seen[2]
        assert np.abs(part[y0:y1, x0:x1] - full[y0:y1, x0:x1]).max() <= SUM_ORDER_TOL
        x0, y0, x1, y1 = seen[3]
        assert not part[y0:y1, x0:x1].any()                     # never started
    finally:
        ds.close()
        _aa_settings(s, **old)


def test_frame_api_errors_and_bucket_loop_cost(load_scene):
    """qd_frame_*: a read without a frame in flight and a rectangle outside the frame are refused; the bucket loop over a
    frame begun ahead costs at most 1.3x the single-rectangle call (kdtree_fixed, 475 buckets; best of three each)."""
    import time
    s = load_scene("kdtree_fixed.qdmg")
    old = _aa_settings(s, want_aa=0)
    ds = device.DeviceScene(s)
    try:
        with pytest.raises(device.DeviceError, match="no frame in flight"):
            ds.frame_read_rect(0, 0, 8, 8)
        W, H = s.settings.frame_width, s.settings.frame_height
        ds.frame_begin()
        with pytest.raises(device.DeviceError, match="bad rectangle"):
            ds.frame_read_rect(0, 0, W + 1, 8)
        part, _, pst = ds.render(16, 16, 80, 64)                  # a rectangle call between begin and end leaves the frame alone
        assert pst["primary_rays"] == 64 * 48
        inside = ds.frame_read_rect(16, 16, 80, 64)
        assert np.abs(inside - part).max() <= SUM_ORDER_TOL
        st = ds.frame_end()
        assert st["primary_rays"] == W * H
        import ctypes as C
        L, cam, stt = device.lib(), s.camera, s.settings
        buckets = [tuple(int(v) for v in b) for b in host.bucket_list(W, H)]
        full = np.zeros((H, W, 3), np.float32)
        frame = np.zeros((H, W, 3), np.float32)
        stats = device.abi.QdStats()

        def single_call():
            assert L.qd_render(ds._h, C.byref(cam), C.byref(stt), 0, 0, W, H, full.ctypes.data, W * 3, None, None, None, C.byref(stats)) == 0

        def bucket_loop():                                       # the loop of INTEGRATION.md, through the C ABI
            assert L.qd_frame_begin(ds._h, C.byref(cam), C.byref(stt), None) == 0
            base = frame.ctypes.data
            for (x0, y0, x1, y1) in buckets:
                assert L.qd_frame_read_rect(ds._h, x0, y0, x1, y1, C.c_void_p(base + (y0 * W + x0) * 12), W * 3) == 0
            assert L.qd_frame_end(ds._h, C.byref(stats)) == 0

        for _ in range(3):                                       # both forms of the schedule measured, staging allocated
            single_call(); bucket_loop()
        single, loop = [], []
        for _ in range(3):
            t0 = time.perf_counter(); single_call(); single.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); bucket_loop(); loop.append(time.perf_counter() - t0)
            assert np.abs(frame - full).max() <= SUM_ORDER_TOL
        print(f"single rectangle {min(single) * 1e3:.1f} ms, bucket loop {min(loop) * 1e3:.1f} ms")
        assert min(loop) <= 1.3 * min(single), (single, loop)
    finally:
        ds.close()
        _aa_settings(s, **old)


def test_device_prep_tables_equal_the_host_tables(tmp_path):
    """heightfield blur / max-heights / normals / high map and Bitmap::Differentiate as HIP kernels
    (qd_prep_*, handed to the front end with qdh_set_prep_hooks) give the CPU code's tables bit for bit: the blurred
    512 x 512 field of heightfield_fixed (15 x 15 gaussian), the same file without blur and without the high map, and
    bumpmap_fixed's differentiated bump map."""
    device.init(0)
    plain = tmp_path / "hf_plain.qdmg"
    src = open(os.path.join(DATA, "heightfield_fixed.qdmg")).read()
    assert "blur 2.5" in src and "useOptimization true" in src
    plain.write_text(src.replace("blur 2.5", "blur 0").replace("useOptimization true", "useOptimization false"))
    for sub in ("texture", "env"):                      # file names in a scene are relative to the scene file
        os.symlink(os.path.join(DATA, sub), tmp_path / sub)

    def tables(path):
        s = host.Scene.load(path)
        d = s.desc
        out = {"hf_f32": np.ctypeslib.as_array(d.hf_f32, shape=(d.n_hf_f32,)).copy() if d.n_hf_f32 else np.zeros(0, np.float32),
               "hf_f64": np.ctypeslib.as_array(d.hf_f64, shape=(d.n_hf_f64,)).copy() if d.n_hf_f64 else np.zeros(0),
               "texels": np.ctypeslib.as_array(d.texels, shape=(d.n_texels * 3,)).copy()}
        if d.n_hfields:
            h = d.hfields[0]
            out["hf_head"] = np.array([h.width, h.height, h.use_optimization, h.max_k, *h.bbox_min, *h.bbox_max], np.float64)
        return out

    for path in (os.path.join(DATA, "heightfield_fixed.qdmg"), str(plain), os.path.join(DATA, "bumpmap_fixed.qdmg")):
        host.prep_on_device(False)
        cpu = tables(path)
        host.prep_on_device(True)
        calls = host.lib().qdh_prep_hook_calls()
        try:
            gpu = tables(path)
        finally:
            host.prep_on_device(False)
        assert host.lib().qdh_prep_hook_calls() - calls == {"heightfield_fixed.qdmg": 2, "hf_plain.qdmg": 1, "bumpmap_fixed.qdmg": 1}[os.path.basename(path)]
        assert cpu.keys() == gpu.keys()
        for k in cpu:
            assert cpu[k].shape == gpu[k].shape, (path, k)
            assert np.array_equal(cpu[k].view(np.uint32 if cpu[k].dtype == np.float32 else np.uint64),
                                  gpu[k].view(np.uint32 if gpu[k].dtype == np.float32 else np.uint64)), (path, k)
    # the hooks were really used: a direct call with a bad argument fails loudly
    assert device.lib().qd_prep_highmap(0, 0, None, 1, None) != 0


def test_nested_layered_matches_oracle_and_too_deep_is_rejected(load_scene, tmp_path):
    """Layered inside Layered (three levels, glossy + Fresnel inside): the device's explicit-stack walk against the
    oracle's recursion, path ids of the inner layers included (the glossy samples would differ otherwise).  Nesting
    beyond the device's stack is refused at scene creation, loudly."""
    s = load_scene("nested_layered.qdmg")
    ds = device.DeviceScene(s)
    try:
        r = compare(s, ds, 0, 0, s.settings.frame_width, s.settings.frame_height)
    finally:
        ds.close()
    assert r["id_mismatch"] == 0 and r["dist_mismatch"] == 0
    assert r["bad_colour"] == 0, (r["bad_colour"], r["max_err"])
    assert r["gpu"]["rays"] == r["cpu"]["rays"]
    p = tmp_path / "deep.qdmg"
    layers = "".join(f"Layered l{i} {{\n layer l{i - 1} (1,1,1)\n}}\n" for i in range(1, 7))
    p.write_text("Camera c {\n position (0,0,-5)\n}\nSphere s {\n}\nConst l0 {\n}\n" + layers + "Node n {\n geometry s\n shader l6\n}\n")
    with pytest.raises(device.DeviceError, match="nested deeper"):
        device.DeviceScene(host.Scene.load(str(p)))


# ---------------------------------------------------------------- anti-aliasing pass (src/main.cpp:58-83, 106-153)
def _aa_settings(s, **kw):
    """load_scene caches scene objects across tests: set every AA field explicitly, hand back the old values."""
    old = {k: getattr(s.settings, k) for k in ("want_aa", "want_adaptive_aa", "show_aa", "aa_threshold")}
    for k, v in kw.items():
        setattr(s.settings, k, v)
    return old


def _aa_compare(s):
    ds = device.DeviceScene(s)
    try:
        rgb, st = ds.render_frame()
    finally:
        ds.close()
    orgb, ost = oracle.render_frame(s, rng_mode=1)
    err = np.abs(rgb.astype(np.float64) - orgb.astype(np.float64)).max(axis=2)
    return rgb, orgb, err, st, ost


def test_adaptive_aa_frame_matches_oracle(load_scene):
    s = load_scene("hw9/nonconvex.qdmg")          # the reference scene that ships with wantAA on
    old = _aa_settings(s, want_aa=1, want_adaptive_aa=1, show_aa=0)
    try:
        rgb, orgb, err, st, ost = _aa_compare(s)
    finally:
        _aa_settings(s, **old)
    n = err.size
    assert int((~(err <= COLOUR_TOL)).sum()) <= max(1, int(n * NEAR_TIE_BUDGET)), float(np.nanmax(err))
    assert st["primary_rays"] > n                 # the second pass really traced extra camera rays


def test_aa_mask_is_the_reference_mask(load_scene):
    s = load_scene("hw9/nonconvex.qdmg")
    old = _aa_settings(s, want_aa=1, want_adaptive_aa=1, show_aa=1)   # marked pixels are painted with aaDebugColor: compares the mask itself
    try:
        rgb, orgb, err, _, _ = _aa_compare(s)
    finally:
        _aa_settings(s, **old)
    dbg = np.array(list(s.settings.aa_debug_color), np.float32)
    mask_gpu = (rgb == dbg).all(axis=2)
    mask_cpu = (orgb == dbg).all(axis=2)
    assert mask_cpu.sum() > 100
    assert int((mask_gpu != mask_cpu).sum()) <= 2     # a colour within rounding of the threshold may flip a pixel


def test_full_aa_and_bucket_border_dependency(load_scene):
    s = load_scene("window_fixed.qdmg")
    old = _aa_settings(s, want_aa=1, want_adaptive_aa=1, show_aa=0, aa_threshold=0.02)   # low threshold: long chains of marked pixels across bucket borders
    try:
        rgb, orgb, err, st, ost = _aa_compare(s)
        n = err.size
        assert int((~(err <= COLOUR_TOL)).sum()) <= max(1, int(n * NEAR_TIE_BUDGET)), float(np.nanmax(err))
        s.settings.want_adaptive_aa = 0              # every pixel gets its four extra rays
        rgb, orgb, err, st, ost = _aa_compare(s)
        assert int((~(err <= COLOUR_TOL)).sum()) <= max(1, int(n * NEAR_TIE_BUDGET)), float(np.nanmax(err))
        assert st["primary_rays"] == 5 * n
    finally:
        _aa_settings(s, **old)


# ---------------------------------------------------------------- device KD tree options
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kdtree_fixed.qdmg", "hw9/dragon.qdmg", "kdtree_teapot_hires_sah_noenv.qdmg", "synth:330"])
def test_device_built_tree_equals_the_host_built_tree(load_scene, data_dir, name):
    """qd_set_kd_build(1), the default, builds the device KD tree with HIP kernels (kdbuild.h), a level
    per round; qd_set_kd_build(2) builds it with host threads (kdrefine.h) by the same rules.  Below 524 288 triangles
    (where the host builder starts to bin a sample) both must give the SAME tree: node, leaf and reference counts and
    the largest leaf equal, node visits and triangle tests of a counted frame equal, hits equal to the bit.  The scenes
    cover binned planes (dragon: 100 k triangles), the exact sweep with one wave and with eight lanes per node, a mesh
    whose triangles overlap heavily (teapot_hires) and the description without a caller's tree (synth: QD_MESH_WANT_KD)."""
    if name.startswith("synth:"):
        s = host.Scene.synthetic(int(name[6:]), seed=42, width=640, height=360, max_trace_depth=4, env_folder=os.path.join(data_dir, "env", "forest"))
        assert s.desc.meshes[0].kd_root == -1 and s.desc.n_kd_nodes == 0          # no tree from the front end: the device makes its own
    else:
        s = load_scene(name)
    old = _aa_settings(s, want_aa=0)
    out = {}
    try:
        for mode in (1, 2):
            device.set_kd_build(mode)
            ds = device.DeviceScene(s)
            try:
                device.set_counting(True)
                rgb, hits, st = ds.render(hits=True)
                out[mode] = (ds.kd_info(), st, hits, rgb, ds.setup_info())
            finally:
                device.set_counting(False)
                ds.close()
    finally:
        device.set_kd_build(1)
        _aa_settings(s, **old)
    (ia, sa, ha, ra, ta), (ib, sb, hb, rb, tb) = out[1], out[2]
    print(f"{name}: device-built {ia} in {ta['kd_build_s'] * 1e3:.1f} ms, host-built {ib} in {tb['kd_build_s'] * 1e3:.1f} ms")
    assert ia == ib and ia["nodes"] > 1000
    assert sa["kd_nodes_visited"] == sb["kd_nodes_visited"] and sa["tris_tested"] == sb["tris_tested"] and sa["rays"] == sb["rays"]
    assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1]) and np.array_equal(ha[2].view(np.uint64), hb[2].view(np.uint64))
    assert np.abs(ra - rb).max() <= 5e-5          # float additions in a different order (a 25-sample glossy floor sums 26 terms per pixel)


@pytest.mark.gpu
def test_callers_tree_modes_need_the_callers_tree(data_dir):
    """qd_set_kd_build(0) and qd_set_kd_refine(0, .) start from the caller's tree; a description without one (libqdhost
    builds its reference-shaped tree on request only) is refused with a message that says what to call."""
    s = host.Scene.synthetic(24, seed=42, width=64, height=36, max_trace_depth=2)
    assert s.desc.meshes[0].kd_root == -1
    import ctypes as C
    device.init(0)
    L = device.lib()
    h = C.c_void_p()
    try:
        assert L.qd_set_kd_build(0) == 0
        assert L.qd_scene_create(s.desc_ptr, C.byref(h)) != 0 and b"qdh_scene_build_kd" in L.qd_last_error()
        s.ensure_kd()
        assert L.qd_scene_create(s.desc_ptr, C.byref(h)) == 0
        L.qd_scene_destroy(h)
    finally:
        L.qd_set_kd_build(1)


def test_callers_tree_refined_tree_and_rebuilt_tree_give_the_same_hits(load_scene):
    """qd_set_kd_refine(0, .) walks the caller's leaves (up to 19 triangles each); qd_set_kd_build(0) keeps the caller's
    planes and splits the leaves further; qd_set_kd_build(1), the default, builds the device tree from the triangles by
    the surface-area heuristic: same closest hits bit for bit, far fewer triangle tests."""
    s = load_scene("kdtree_fixed.qdmg")
    s.settings.want_aa = 0
    out = {}
    try:
        for name, (max_leaf, build) in {"caller": (0, 0), "refined": (2, 0), "rebuilt": (2, 1)}.items():
            device.set_kd_refine(max_leaf, 0.6)
            device.set_kd_build(build)
            ds = device.DeviceScene(s)
            try:
                device.set_counting(True)
                rgb, (hn, ht, hd), st = ds.render(400, 300, 800, 600, hits=True)
                out[name] = (rgb, hn, ht, hd, st, ds.kd_info())
            finally:
                device.set_counting(False)
                ds.close()
    finally:
        device.set_kd_refine(2, 0.6)
        device.set_kd_build(1)
    rgb0, hn0, ht0, hd0, st0, info0 = out["caller"]
    assert (ht0 >= 0).sum() > 1000                       # the mesh is really in view
    for name in ("refined", "rebuilt"):
        rgb, hn, ht, hd, st, info = out[name]
        assert np.array_equal(hn0, hn) and np.array_equal(ht0, ht), name
        assert np.array_equal(hd0.view(np.uint64), hd.view(np.uint64)), name
        assert np.abs(rgb0 - rgb).max() <= SUM_ORDER_TOL, name
        assert st["tris_tested"] * 1.5 < st0["tris_tested"], (name, st["tris_tested"], st0["tris_tested"])
    assert info0["leaf_refs"] == round(s.kd_stats(0)["avg_tris"] * s.kd_stats(0)["leaves"]) and info0["max_leaf"] <= 20
    assert out["refined"][5]["nodes"] > info0["nodes"]
    # the reference's midpoint builder duplicates straddling triangles down to depth 65 (83 references per triangle here);
    # the heuristic stops where a split no longer pays
    assert out["rebuilt"][5]["leaf_refs"] < info0["leaf_refs"] // 4
    assert out["rebuilt"][4]["kd_nodes_visited"] < out["refined"][4]["kd_nodes_visited"]


def test_frame_renderer_runs_the_aa_pass_on_the_gathered_frame(load_scene):
    """FrameRenderer.render_frame with wantAA: tiles -> untile -> qd_antialias_device == qd_render + qd_antialias."""
    from raytracing_amd import renderer
    s = load_scene("hw9/nonconvex.qdmg")
    old = _aa_settings(s, want_aa=1, want_adaptive_aa=1, show_aa=0)
    try:
        fr = renderer.FrameRenderer(s, tile=32, rank=0, world=1, local_device=0)
        try:
            frame = fr.render_frame().cpu().numpy()
            fr.check()
        finally:
            fr.dscene.close()
        ds = device.DeviceScene(s)
        try:
            ref, _ = ds.render_frame()
        finally:
            ds.close()
    finally:
        _aa_settings(s, **old)
    assert np.abs(frame - ref).max() <= SUM_ORDER_TOL
    first, _, _ = device.DeviceScene(s).render()
    assert np.abs(frame - first).max() > 1e-3        # the pass changed something


def test_first_frame_runs_in_the_form_the_prior_names(load_scene):
    """Before anything is measured (a caller that renders one image) the form comes from the scene: wavefront, unless the
    tree is deep (asserted in test_full_size_bench_scene_matches_oracle).  A 25-sample glossy floor, whose queues outgrow
    the memory budget at 4K, is rendered in bands there (test_wavefront_form_renders_a_big_glossy_frame_in_bands) and stays
    on the form.  Either way the frame is the one a pinned form gives."""
    if os.environ.get("QD_PIPELINE") not in (None, "", "2"):
        pytest.skip("the scheduling form is pinned for this run")
    for name, size, want in [("kdtree_teapot_lowres_env.qdmg", None, 1), ("hw9/dragon.qdmg", None, 1), ("hw9/dragon.qdmg", (3840, 2160), 1)]:
        s = load_scene(name)
        st = s.settings
        old = (st.frame_width, st.frame_height, st.want_aa)
        st.want_aa = 0
        if size:
            st.frame_width, st.frame_height = size
        try:
            ds = device.DeviceScene(s)
            try:
                rgb, _, _ = ds.render()
                assert ds.schedule_info()["form"] == want, name
                if size:
                    continue
                device.set_pipeline(1 - want)
                try:
                    ref, _, _ = ds.render()
                finally:
                    device.set_pipeline(2)
                assert np.abs(rgb - ref).max() <= COLOUR_TOL, name          # (25 glossy samples per pixel: sums of ~30 terms in two orders)
            finally:
                ds.close()
        finally:
            st.frame_width, st.frame_height, st.want_aa = old


def test_wavefront_form_renders_a_big_glossy_frame_in_bands(load_scene):
    """A 25-sample glossy floor at 4K asks for 28 queue entries per pixel, 41 GB.  The scene's first wavefront frame is tried
    in one band with what fits 16 GB (most frames whose shader graph asks for that much fill a fraction of it), outgrows it
    and is rendered again in place; from then on the form renders the frame in bands sized by the shader graph
    (qd_scene_wavefront_info), every band with its own loop over the generations.  Both frames are the one-kernel form's;
    nothing is reported, the scene stays eligible for the form."""
    s = load_scene("hw9/dragon.qdmg")
    st = s.settings
    old = (st.frame_width, st.frame_height, st.want_aa)
    st.frame_width, st.frame_height, st.want_aa = 3840, 2160, 0
    try:
        ds = device.DeviceScene(s)
        try:
            assert ds.wavefront_info()["bands"] == 1
            device.set_pipeline(1)
            first, _, s1 = ds.render()                       # outgrows the one band; rendered again on the stream
            assert ds.schedule_info()["form"] == 1
            info = ds.wavefront_info()
            assert info["bands"] >= 2 and not info["overflowed"]
            a, _, sa = ds.render()                           # in bands
            assert ds.schedule_info()["form"] == 1
            device.set_pipeline(0)
            b, _, sb = ds.render()
            ds.check()
            assert not ds.wavefront_info()["overflowed"]
            for got in (s1, sa):
                assert got["rays"] == sb["rays"] and got["primary_rays"] == 3840 * 2160 and got["shadow_rays"] == sb["shadow_rays"]
            assert np.abs(first - b).max() <= COLOUR_TOL
            assert np.abs(a - b).max() <= COLOUR_TOL
        finally:
            device.set_pipeline(2)
            ds.close()
    finally:
        st.frame_width, st.frame_height, st.want_aa = old


def test_big_glossy_frames_in_flight_outgrow_their_first_queues(load_scene):
    """The same 4K glossy scene through the asynchronous entry point with two frames in flight on a fresh handle: the first frame
    starts alone and is enqueued in four parts, the second whole, both with the optimistic queues of a scene's first frames,
    which they outgrow -- every part and the whole frame take the in-stream fallback, on their own streams, side by side.  All
    frames equal the one-kernel form's; afterwards the scene renders in bands and is still eligible for the form."""
    from raytracing_amd import renderer
    s = load_scene("hw9/dragon.qdmg")
    st = s.settings
    old = (st.frame_width, st.frame_height, st.want_aa)
    st.frame_width, st.frame_height, st.want_aa = 3840, 2160, 0
    try:
        device.set_pipeline(0)
        ds = device.DeviceScene(s)
        try:
            ref, _, _ = ds.render()
        finally:
            ds.close()
        device.set_pipeline(1)
        fr = renderer.FrameRenderer(s, tile=32, rank=0, world=1, local_device=0, frames_in_flight=2)
        try:
            assert fr.dscene.wavefront_info()["bands"] == 1
            slots = [fr.submit_frame() for _ in range(2)]
            frames = [fr.wait(k).cpu().numpy() for k in slots]
            fr.check()
            for f in frames:
                assert np.abs(f - ref).max() <= COLOUR_TOL
            info = fr.dscene.wavefront_info()
            assert info["bands"] >= 2 and not info["overflowed"]
            k = fr.submit_frame()                                # sized by the shader graph now: bands (inside four parts), no overflow
            assert np.abs(fr.wait(k).cpu().numpy() - ref).max() <= COLOUR_TOL
            fr.check()
            assert not fr.dscene.wavefront_info()["overflowed"] and fr.dscene.schedule_info()["form"] == 1
        finally:
            fr.close()
    finally:
        device.set_pipeline(2)
        st.frame_width, st.frame_height, st.want_aa = old


@pytest.mark.parametrize("name", ["kdtree_fixed.qdmg", "lecture7_fixed.qdmg"])
def test_wavefront_queues_cannot_fail_a_call(load_scene, name):
    """No call fails over the wavefront queues.  With the queues cut to 20 000 entries (qd_set_wavefront_budget's test switch;
    a generation of these frames has hundreds of thousands of rays) every band of the wavefront form overflows: the launch that follows each band on
    the stream renders it again in the one-kernel form, so the synchronous call, the asynchronous frames and the AA pass all
    return complete frames equal to the one-kernel form's, with its ray counts; nothing is reported by qd_scene_check, and
    the scene is kept off the wavefront form afterwards.  With a budget of 2 MB the frame is cut into dozens of bands of
    some 2 500 pixels (a band that lies on the glass teapot may well outgrow its four entries per pixel and take the
    fallback; a frame-sized band averages over sky and floor): same frame, same ray counts."""
    from raytracing_amd import renderer
    s = load_scene(name)
    old = _aa_settings(s, want_aa=0)
    W, H = s.settings.frame_width, s.settings.frame_height
    try:
        ds = device.DeviceScene(s)
        try:
            device.set_pipeline(0)
            ref, _, sref = ds.render()
        finally:
            ds.close()
        # many small bands
        device.set_pipeline(1)
        device.set_wavefront_budget(2 << 20, 0)
        ds = device.DeviceScene(s)
        try:
            assert ds.wavefront_info()["bands"] > 8
            a, _, sa = ds.render()
            ds.check()
            assert ds.schedule_info()["form"] == 1
            assert sa["rays"] == sref["rays"] and sa["depth_killed"] == sref["depth_killed"]
            assert np.abs(a - ref).max() <= COLOUR_TOL
        finally:
            ds.close()
        # every band overflows and is rendered again in place
        device.set_wavefront_budget(0, 20000)
        ds = device.DeviceScene(s)
        try:
            b, _, sb = ds.render(0, 0, W, H)
            ds.check()                                           # nothing to report
            assert ds.wavefront_info()["overflowed"]
            assert sb["rays"] == sref["rays"] and sb["primary_rays"] == W * H and sb["shadow_rays"] == sref["shadow_rays"]
            assert np.abs(b - ref).max() <= COLOUR_TOL
        finally:
            ds.close()
        # asynchronous frames in flight + the AA pass under the same squeeze, library's own choice of form
        device.set_pipeline(2)
        fr = renderer.FrameRenderer(s, tile=32, rank=0, world=1, local_device=0, frames_in_flight=2)
        try:
            fr.settle()
            for _ in range(3):
                k = fr.submit_frame()
            frame = fr.wait(k).cpu().numpy()
            fr.check()
            assert np.abs(frame - ref).max() <= COLOUR_TOL
            assert fr.dscene.wavefront_info()["overflowed"] and fr.dscene.schedule_info()["form"] == 0     # demoted after its first try
            aa = ref.copy()
            fr.dscene.antialias(aa)
            fr.check()
        finally:
            fr.close()
        device.set_wavefront_budget(0, 0)
        ds = device.DeviceScene(s)
        try:
            aa_ref = ref.copy()
            ds.antialias(aa_ref)
        finally:
            ds.close()
        assert np.abs(aa - aa_ref).max() <= COLOUR_TOL
    finally:
        device.set_pipeline(2)
        device.set_wavefront_budget(0, 0)
        _aa_settings(s, **old)


def test_frame_begun_ahead_survives_an_overflow_and_a_render_in_between(load_scene):
    """qd_frame_begin in the wavefront form with queues that every band outgrows (the test switch of
    qd_set_wavefront_budget), then a qd_render on the same stream -- same working set, same error word -- before the first
    read: the frame read back is the complete frame (the band was rendered again in place, on the stream), the rectangle
    call is right too, and nothing is reported."""
    s = load_scene("kdtree_fixed.qdmg")
    old = _aa_settings(s, want_aa=0)
    W, H = s.settings.frame_width, s.settings.frame_height
    try:
        device.set_pipeline(0)
        ds = device.DeviceScene(s)
        try:
            ref, _, sref = ds.render()
        finally:
            ds.close()
        device.set_pipeline(1)
        device.set_wavefront_budget(0, 20000)
        ds = device.DeviceScene(s)
        try:
            ds.frame_begin()
            part, _, pst = ds.render(560, 400, 688, 496)             # on the teapot: overflows its queues as well
            frame = ds.frame_read_rect(0, 0, W, H)
            st = ds.frame_end()
            ds.check()
            assert ds.wavefront_info()["overflowed"]
            assert st["rays"] == sref["rays"] and st["primary_rays"] == W * H
            assert np.abs(frame - ref).max() <= COLOUR_TOL
            assert pst["primary_rays"] == 128 * 96 and np.abs(part - ref[400:496, 560:688]).max() <= COLOUR_TOL
        finally:
            ds.close()
    finally:
        device.set_pipeline(2)
        device.set_wavefront_budget(0, 0)
        _aa_settings(s, **old)


def test_lone_wavefront_frame_is_enqueued_in_parts(load_scene):
    """A frame of the wavefront form that starts on a drained scene is enqueued as four concurrent parts (8 x 8 units k,
    k + 4, ... each on a stream and working set of the library's own; qd_set_frame_parts): same frame, hit planes and
    ray counts as the whole-frame launch and as the one-kernel form.  A frame enqueued while another one is in flight stays
    whole; so do the frame after it, small frames and the one-kernel form.  With queues that every band outgrows, every part takes its in-stream
    fallback and the frame is still complete."""
    import torch
    s = load_scene("kdtree_fixed.qdmg")
    old = _aa_settings(s, want_aa=0)
    W, H = s.settings.frame_width, s.settings.frame_height
    try:
        ds = device.DeviceScene(s)
        try:
            device.set_pipeline(0)
            ref, href, sref = ds.render(hits=True)
            assert ds.frame_parts() == 1
            device.set_pipeline(1)
            device.set_frame_parts(1)
            whole, hwhole, swhole = ds.render(hits=True)
            assert ds.frame_parts() == 1
            device.set_frame_parts(0)
            for _ in range(2):                                   # twice: the parts' streams and sets are reused
                a, ha, sa = ds.render(hits=True)
                assert ds.frame_parts() == 4 and ds.schedule_info()["form"] == 1
                for k in ("rays", "primary_rays", "shadow_rays", "depth_killed"):
                    assert sa[k] == sref[k] == swhole[k], k
                assert np.abs(a - ref).max() <= COLOUR_TOL and np.abs(a - whole).max() <= COLOUR_TOL
                for x, y in zip(ha[:2], href[:2]):
                    assert (x == y).all()
                assert (ha[2].view(np.int64) == href[2].view(np.int64)).all()
            part, _, pst = ds.render(560, 400, 688, 496)         # a small rectangle stays whole
            assert ds.frame_parts() == 1 and np.abs(part - ref[400:496, 560:688]).max() <= COLOUR_TOL
            # asynchronous frames: the first starts alone (parts), the second is enqueued behind it (whole)
            tile, dev = 32, torch.device("cuda:0")
            from raytracing_amd import sharding
            tpr = sharding.tiles_per_rank_padded(W, H, tile, 1)
            bufs = [torch.zeros((tpr, tile, tile, 3), dtype=torch.float32, device=dev) for _ in range(2)]
            frames = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            seen = []
            for k in range(2):
                ds.render_tiles_async(bufs[k].data_ptr(), tile, 0, 1, stream=streams[k].cuda_stream)
                seen.append(ds.frame_parts())
                device.untile_async(bufs[k].data_ptr(), frames[k].data_ptr(), W, H, tile, 1, tpr, stream=streams[k].cuda_stream)
            torch.cuda.synchronize()
            ds.check()
            assert seen == [4, 1], seen
            for f in frames:
                assert np.abs(f.cpu().numpy() - ref).max() <= COLOUR_TOL
            ds.render()                                          # the first frame after one that was enqueued behind another stays whole
            assert ds.frame_parts() == 1                         # (a caller with frames in flight gains nothing from parts)
            # every part overflows (the test switch): four fallbacks side by side
            device.set_wavefront_budget(0, 20000)
            b, _, sb = ds.render()
            ds.check()
            assert ds.frame_parts() == 4
            assert sb["rays"] == sref["rays"] and sb["shadow_rays"] == sref["shadow_rays"]
            assert np.abs(b - ref).max() <= COLOUR_TOL
        finally:
            ds.close()
    finally:
        device.set_pipeline(2)
        device.set_frame_parts(0)
        device.set_wavefront_budget(0, 0)
        _aa_settings(s, **old)


def test_aa_pass_of_a_large_frame_in_parts(load_scene):
    """The AA pass of a frame whose list of marked pixels is long enough for frame parts (non-adaptive AA marks every pixel:
    1.08 M pixels x 4 taps = 67 500 units of 64 samples): first pass and tap pass are each enqueued as four concurrent parts
    in the wavefront form; the frame and the ray counts are those of whole launches."""
    s = load_scene("kdtree_fixed.qdmg")
    old = _aa_settings(s, want_aa=1, want_adaptive_aa=0, show_aa=0)
    try:
        device.set_pipeline(1)
        out = {}
        for parts in (1, 0):
            device.set_frame_parts(parts)
            ds = device.DeviceScene(s)
            try:
                first, _, _ = ds.render()
                frame = first.copy()
                aa = ds.antialias(frame)
                out[parts] = (frame, aa, ds.frame_parts())
                ds.check()
            finally:
                ds.close()
        assert out[1][2] == 1 and out[0][2] == 4                       # the tap pass itself ran whole / in parts
        assert np.abs(out[0][0] - out[1][0]).max() <= SUM_ORDER_TOL
        for k in ("rays", "primary_rays", "shadow_rays", "depth_killed"):
            assert out[0][1][k] == out[1][1][k], k
        assert out[0][1]["primary_rays"] == 4 * 1200 * 900
    finally:
        device.set_pipeline(2)
        device.set_frame_parts(0)
        _aa_settings(s, **old)


def test_adaptive_choice_times_both_forms_and_reports_setup(load_scene):
    """qd_set_pipeline(2): after FrameRenderer.settle() both scheduling forms have a measured ns-per-pixel figure (lone
    frames, two per form); the frames that follow -- in flight, untimed -- run in the faster one (kdtree_fixed: the forms
    are 2x apart), and they are the lone render's whichever form traced them.  qd_scene_setup_info reports the set-up times."""
    from raytracing_amd import renderer
    if os.environ.get("QD_PIPELINE") not in (None, "", "2"):
        pytest.skip("the scheduling form is pinned for this run")
    s = load_scene("kdtree_fixed.qdmg")
    old = _aa_settings(s, want_aa=0)
    try:
        fr = renderer.FrameRenderer(s, tile=32, rank=0, world=1, local_device=0, frames_in_flight=2)
        try:
            info = fr.dscene.setup_info()
            assert info["kd_build_s"] >= 0. and info["upload_s"] > 0.
            fr.settle()
            sched = fr.dscene.schedule_info()
            assert sched["ns_per_pixel"][0] > 0. and sched["ns_per_pixel"][1] > 0.
            best = 0 if sched["ns_per_pixel"][0] <= sched["ns_per_pixel"][1] else 1
            assert max(sched["ns_per_pixel"]) > 1.3 * min(sched["ns_per_pixel"])
            for _ in range(9):                                         # frames in flight: not timed, all in the better form
                k = fr.submit_frame()
                assert fr.dscene.schedule_info()["form"] == best
            frame = fr.wait(k).cpu().numpy()
            fr.check()
            device.set_pipeline(1 - best)
            try:
                ref, _, _ = fr.dscene.render()                         # the same frame, alone, in the other form
            finally:
                device.set_pipeline(2)
            assert np.abs(frame - ref).max() <= SUM_ORDER_TOL
        finally:
            fr.dscene.close()
    finally:
        _aa_settings(s, **old)


def test_frames_in_flight_give_the_single_frame(load_scene):
    """Frames on different HIP streams use separate working sets of the scene (qd_scene_bind_stream) and overlap on the
    GPU; every one of them must be the frame a lone render gives.  Ten streams on one scene also exercise the take-over
    of the working set used longest ago (the library keeps eight)."""
    import torch
    from raytracing_amd import renderer
    s = load_scene("kdtree_teapot_lowres_env.qdmg")
    old = _aa_settings(s, want_aa=0)
    try:
        W, H = s.settings.frame_width, s.settings.frame_height
        fr = renderer.FrameRenderer(s, tile=32, rank=0, world=1, local_device=0, frames_in_flight=3)
        try:
            ref, _, ref_stats = fr.dscene.render()
            frames = []
            for i in range(7):
                k = fr.submit_frame()
                if i >= 4:                                  # the last three submits: one per slot, still in flight together
                    frames.append(k)
            got = [fr.wait(k).cpu().numpy() for k in frames]
            fr.check()
            assert sorted(frames) == [0, 1, 2]
            for g in got:
                assert np.abs(g - ref).max() <= SUM_ORDER_TOL
            assert fr.stats_dict()["rays"] == ref_stats["rays"]
            # ten streams, one shard buffer each, enqueued back to back
            tile, tpr = 32, sharding.tiles_per_rank_padded(W, H, 32, 1)
            streams = [torch.cuda.Stream() for _ in range(10)]
            shards = [torch.zeros((tpr, tile, tile, 3), dtype=torch.float32, device="cuda:0") for _ in streams]
            torch.cuda.synchronize()                    # torch fills them on ITS stream: not ordered with the ten streams below
            for st_, sh in zip(streams, shards):
                fr.dscene.render_tiles_async(sh.data_ptr(), tile, 0, 1, stream=st_.cuda_stream)
            torch.cuda.synchronize()
            fr.check()
            for sh in shards:
                img = sharding.untile(sh.cpu().numpy()[None], W, H, tile, 1)
                assert np.abs(img - ref).max() <= SUM_ORDER_TOL
        finally:
            fr.dscene.close()
    finally:
        _aa_settings(s, **old)


def test_aa_in_steps_equals_the_single_call(load_scene):
    """qd_aa_begin_device + qd_aa_trace_part_async over three uneven slices + qd_aa_end_device == qd_antialias_device."""
    import torch
    s = load_scene("hw9/nonconvex.qdmg")
    old = _aa_settings(s, want_aa=1, want_adaptive_aa=1, show_aa=0)
    ds = device.DeviceScene(s)
    try:
        first, _, _ = ds.render()
        whole = torch.from_numpy(first).cuda()
        ds.antialias_device(whole.data_ptr())
        stepped = torch.from_numpy(first).cuda()
        n = ds.aa_begin(stepped.data_ptr())
        assert n > 100
        taps = torch.zeros((n, 4, 3), dtype=torch.float32, device="cuda:0")
        cuts = [0, n // 5, n // 5, n // 2 + 3, n]                 # an empty slice among them
        for a, b in zip(cuts[:-1], cuts[1:]):
            ds.aa_trace_part(a, b - a, taps[a:].data_ptr() if b > a else 0)
        torch.cuda.synchronize()
        ds.aa_end(taps.data_ptr(), stepped.data_ptr())
        ds.check()
        assert np.abs(stepped.cpu().numpy() - whole.cpu().numpy()).max() <= SUM_ORDER_TOL
        assert np.abs(whole.cpu().numpy() - first).max() > 1e-3
        with pytest.raises(device.DeviceError, match="no plan"):
            ds.aa_trace_part(0, 1, taps.data_ptr())              # the plan ended with aa_end
    finally:
        ds.close()
        _aa_settings(s, **old)


@pytest.mark.parametrize("name", ["synthetic", "kdtree_fixed.qdmg", "lecture7_fixed.qdmg", "heightfield_fixed.qdmg", "bumpmap_fixed.qdmg"])
def test_specialised_kernel_copy_equals_the_all_code_copy(load_scene, data_dir, name):
    """launch() picks a kernel copy by what the scene holds (heightfield / CSG / other primitives / rich textures compiled
    out when absent); the counting build always runs the copy with everything.  Same source, so same hits and colours."""
    if name == "synthetic":
        s = host.Scene.synthetic(64, seed=7, width=256, height=160, max_trace_depth=6, env_folder=os.path.join(data_dir, "env", "forest"))
        old = None
    else:
        s = load_scene(name)
        old = _aa_settings(s, want_aa=0)
    W, H = s.settings.frame_width, s.settings.frame_height
    rect = (W // 4, H // 4, W // 4 + min(256, W // 2), H // 4 + min(160, H // 2))
    ds = device.DeviceScene(s)
    try:
        rgb0, (hn0, ht0, hd0), st0 = ds.render(*rect, hits=True)
        device.set_counting(True)
        try:
            rgb1, (hn1, ht1, hd1), st1 = ds.render(*rect, hits=True)
        finally:
            device.set_counting(False)
    finally:
        ds.close()
        if old is not None:
            _aa_settings(s, **old)
    assert np.array_equal(hn0, hn1) and np.array_equal(ht0, ht1) and np.array_equal(hd0.view(np.uint64), hd1.view(np.uint64))
    # float additions land in a different order in the two runs; a glossy pixel sums 25+ samples of magnitude > 1, so the
    # rounding of the sum scales with the pixel's value
    assert (np.abs(rgb0 - rgb1) <= 5 * SUM_ORDER_TOL * np.maximum(1.0, np.abs(rgb1))).all(), float(np.abs(rgb0 - rgb1).max())
    assert st0["rays"] == st1["rays"] and st0["shadow_rays"] == st1["shadow_rays"]


def test_rccl_exchange_step_runs_at_world_one(load_scene):
    """The library's own exchange step (qd_comm_create -> ncclCommInitRank, qd_gather_frame_async -> grouped ncclSend /
    ncclRecv + de-interleave, qd_comm_broadcast_async) as far as one GPU allows: a one-rank communicator, frames in flight
    on side streams, the AA pass through broadcast + tap gather.  Every frame must be the frame of a lone qd_render."""
    from raytracing_amd import renderer
    s = load_scene("hw9/nonconvex.qdmg")
    old = _aa_settings(s, want_aa=0)
    try:
        fr = renderer.FrameRenderer(s, tile=32, rank=0, world=1, local_device=0, frames_in_flight=3, collectives="rccl",
                                    force_collectives=True)
        try:
            assert fr.comm is not None and fr.exchange and len(fr.comms) == 3        # one communicator per slot stream
            ref, _, _ = fr.dscene.render()
            slots = [fr.submit_frame() for _ in range(5)][-3:]
            for k in slots:
                assert np.abs(fr.wait(k).cpu().numpy() - ref).max() <= SUM_ORDER_TOL
            fr.check()
            # AA pass over the collectives: frame broadcast, taps traced by "every" rank, taps gathered, replace rule on rank 0
            _aa_settings(s, want_aa=1, want_adaptive_aa=1, show_aa=0)
            frame = fr.render_frame().cpu().numpy()
            fr.check()
            ref_aa, _ = fr.dscene.render_frame()
            assert np.abs(frame - ref_aa).max() <= SUM_ORDER_TOL
            assert np.abs(frame - ref).max() > 1e-3
        finally:
            fr.close()
    finally:
        _aa_settings(s, **old)


def test_rccl_gather_through_the_c_abi_directly():
    """qd_comm_gather_async / qd_gather_frame_async called like a C host would: raw device pointers, one rank."""
    import torch
    device.init(0)
    comm = device.Comm(0, 1)
    try:
        W, H, tile = 100, 70, 32
        tpr = sharding.tiles_per_rank_padded(W, H, tile, 1)
        rng = np.random.default_rng(1)
        shard = torch.from_numpy(rng.random((tpr, tile, tile, 3), dtype=np.float32)).cuda()
        gathered = torch.zeros((1, tpr, tile, tile, 3), dtype=torch.float32, device="cuda:0")
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        comm.gather_frame_async(shard.data_ptr(), gathered.data_ptr(), frame.data_ptr(), W, H, tile, stream=st.cuda_stream)
        st.synchronize()
        assert np.array_equal(gathered.cpu().numpy()[0], shard.cpu().numpy())
        assert np.array_equal(frame.cpu().numpy(), sharding.untile(shard.cpu().numpy()[None], W, H, tile, 1))
        buf = torch.arange(1000, dtype=torch.float32, device="cuda:0")
        comm.broadcast_async(buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), np.arange(1000, dtype=np.float32))
    finally:
        comm.close()
