// C-ABI of the host front end (include/qdhost.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "image.h"
#include "qdhost.h"
#include "scene_impl.h"
#include "vecmath.h"

using namespace qdh;

struct qdh_scene { SceneImpl impl; };

static thread_local std::string g_err;

extern "C" const char* qdh_last_error(void) { return g_err.c_str(); }

extern "C" qdh_scene* qdh_load_scene(const char* path) {
    qdh_scene* s = new qdh_scene;
    std::string err;
    if (!s->impl.load(path, err)) { g_err = err; delete s; return nullptr; }
    g_err.clear();
    return s;
}

extern "C" void qdh_set_prep_hooks(const qdh_prep_hooks* hooks) {
    qdh::g_prep_hooks.heightfield = hooks ? hooks->heightfield : nullptr;
    qdh::g_prep_hooks.highmap = hooks ? hooks->highmap : nullptr;
    qdh::g_differentiate_hook = hooks ? hooks->differentiate : nullptr;
}

extern "C" int qdh_prep_hook_calls(void) { return qdh::g_prep_hook_calls + qdh::g_differentiate_hook_calls; }

extern "C" qd_scene_desc* qdh_scene_desc(qdh_scene* s) { return s ? &s->impl.desc : nullptr; }

extern "C" void qdh_set_eager_kd(int on) { qdh::g_eager_kd = on != 0; }

extern "C" int qdh_scene_build_kd(qdh_scene* s) {
    if (!s) { g_err = "qdh_scene_build_kd: null scene"; return -1; }
    s->impl.build_kd();
    return 0;
}
extern "C" void qdh_free_scene(qdh_scene* s) { delete s; }
extern "C" void qdh_free(void* p) { free(p); }

// Camera::BeginFrame (src/camera.cpp:6-24)
extern "C" void qdh_camera_corners(const double position[3], double yaw, double pitch, double roll,
                                   double fov, double aspect, qd_camera* out) {
    const double wanted_angle = to_radians(fov / 2.);
    const double wanted_length = std::tan(wanted_angle);
    const double actual_length = std::sqrt(aspect * aspect + 1.);
    const double scale = wanted_length / actual_length;
    const double x2d = -aspect * scale * 1.5;
    const double y2d = scale * 1.5;
    const M3 rot = rot_z(to_radians(roll)) * rot_x(to_radians(pitch)) * rot_y(to_radians(yaw));
    const V3 pos(position[0], position[1], position[2]);
    const V3 tl = V3(x2d, y2d, 1.) * rot + pos;
    const V3 tr = V3(-x2d, y2d, 1.) * rot + pos;
    const V3 bl = V3(x2d, -y2d, 1.) * rot + pos;
    const V3* src[4] = {&tl, &tr, &bl, &pos};
    double* dst[4] = {out->top_left, out->top_right, out->bottom_left, out->position};
    for (int i = 0; i < 4; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
}

// GetBucketList (src/sdl.cpp:253-276): 48x48 buckets, serpentine rows, clipped to the frame
extern "C" int qdh_bucket_list(int w, int h, int32_t* rects, int cap) {
    const int bs = 48;
    const int bw = (w - 1) / bs + 1, bh = (h - 1) / bs + 1;
    int n = 0;
    auto push = [&](int x, int y) {
        if (rects && n < cap) {
            rects[n * 4 + 0] = x * bs; rects[n * 4 + 1] = y * bs;
            rects[n * 4 + 2] = std::min((x + 1) * bs, w); rects[n * 4 + 3] = std::min((y + 1) * bs, h);
        }
        ++n;
    };
    for (int y = 0; y < bh; ++y) {
        if (y % 2 == 0) for (int x = 0; x < bw; ++x) push(x, y);
        else for (int x = bw - 1; x >= 0; --x) push(x, y);
    }
    return n;
}

extern "C" int qdh_save_bmp(const char* path, const float* rgb, int w, int h, size_t pitch) {
    std::string err;
    if (!save_bmp(path, rgb, w, h, pitch, err)) { g_err = err; return -1; }
    return 0;
}

extern "C" int qdh_save_exr(const char* path, const float* rgb, int w, int h, size_t pitch) {
    std::string err;
    if (!path || !rgb) { g_err = "qdh_save_exr: null argument"; return -1; }
    if (!save_exr(path, rgb, w, h, pitch, err)) { g_err = err; return -1; }
    return 0;
}

extern "C" float* qdh_load_image(const char* path, uint32_t* w, uint32_t* h) {
    Image img; std::string err;
    if (!load_image(path, img, err)) { g_err = err; return nullptr; }
    float* out = (float*)malloc(img.rgb.size() * sizeof(float));
    memcpy(out, img.rgb.data(), img.rgb.size() * sizeof(float));
    *w = img.w; *h = img.h;
    return out;
}

// Test hook: the host front end's matrix / transform arithmetic on one 16-f64 record {a, b, c, d, s, 0, 0, 0}, laid out as
// words [23..74] of oracle/ref_driver.cpp's function-level record (what the reference's src/matrix.cpp and
// src/transform.cpp compute for the same input): Determinant, Inverse, Transpose of the matrix with rows a, b, c;
// RotationAroundX(s) * RotationAroundY(a.x) * RotationAroundZ(a.y); d * M; a ^ b; a * b; |a|; two zeros; then
// Transform{Scale(a), Rotate(b * 90), Translate(c)}: Point(d), UndoPoint(d), Direction(d), UndoDirection(d), Normal(d).
extern "C" void qdh_math_probe(const double in[16], double out[52]) {
    const V3 a(in[0], in[1], in[2]), b(in[3], in[4], in[5]), c(in[6], in[7], in[8]), d(in[9], in[10], in[11]);
    const double s = in[12];
    int k = 0;
    auto putv = [&](const V3& v) { out[k++] = v.x; out[k++] = v.y; out[k++] = v.z; };
    auto putm = [&](const M3& m) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[k++] = m.m[i][j]; };
    M3 m(0.);
    for (int j = 0; j < 3; ++j) { m.m[0][j] = a[j]; m.m[1][j] = b[j]; m.m[2][j] = c[j]; }
    out[k++] = determinant(m);
    putm(inverse(m));
    putm(transpose(m));
    putm(rot_x(s) * rot_y(in[0]) * rot_z(in[1]));
    putv(d * m);
    putv(cross(a, b));
    out[k++] = dot(a, b);
    out[k++] = length(a);
    out[k++] = 0.; out[k++] = 0.;
    Transform t;
    t.scale(a.x, a.y, a.z);
    t.rotate(b.x * 90, b.y * 90, b.z * 90);
    t.translate(c);
    putv(d * t.m + t.offset);            // Transform::Point, src/transform.cpp:44-47
    putv((d - t.offset) * t.inv);        // UndoPoint :49-52
    putv(d * t.m);                       // Direction :54-57
    putv(d * t.inv);                     // UndoDirection :59-62
    putv(d * t.invT);                    // Normal :64-67
}

// ---------------------------------------------------------------------------
// Synthetic benchmark scene: a displaced n x n grid (2*n*n triangles) of "glass"
// over a checker floor.  Everything is derived from raw mt19937(seed) words so the
// scene is reproducible from (n, seed) alone.
extern "C" qdh_scene* qdh_synth_scene(int n, unsigned seed, int width, int height, int max_trace_depth,
                                      const char* env_folder) {
    if (n < 2 || n > 8192) { g_err = "qdh_synth_scene: n out of range"; return nullptr; }
    qdh_scene* s = new qdh_scene;
    SceneImpl& S = s->impl;
    std::mt19937 rng(seed);
    auto u01 = [&]() { return (double)rng() / 4294967296.0; };
    const int OCT = 6;
    double amp[OCT], fx[OCT], fz[OCT], px[OCT], pz[OCT];
    for (int k = 0; k < OCT; ++k) {
        const double sc = std::ldexp(1.0, k);
        amp[k] = 0.12 * (0.5 + u01()) / sc;
        fx[k] = (1.5 + 2.0 * u01()) * sc;
        fz[k] = (1.5 + 2.0 * u01()) * sc;
        px[k] = 2.0 * kPi * u01();
        pz[k] = 2.0 * kPi * u01();
    }
    std::unique_ptr<MeshData> m(new MeshData);
    m->faceted = false; m->backcull = false; m->use_kd = true; m->use_sah = false;
    const size_t nv = (size_t)(n + 1) * (n + 1);
    m->verts.assign(nv + 1, V3()); m->normals.assign(nv + 1, V3()); m->uvs.assign(nv + 1, V3());
    parallel_for((size_t)n + 1, [&](size_t j0, size_t j1) {
    for (int j = (int)j0; j < (int)j1; ++j)
        for (int i = 0; i <= n; ++i) {
            const size_t at = 1 + (size_t)j * (n + 1) + i;
            const double x = -1.0 + 2.0 * i / n, z = -1.0 + 2.0 * j / n;
            double h = 0, dhx = 0, dhz = 0;
            for (int k = 0; k < OCT; ++k) {
                const double sx = std::sin(fx[k] * x + px[k]), cx = std::cos(fx[k] * x + px[k]);
                const double sz = std::sin(fz[k] * z + pz[k]), cz = std::cos(fz[k] * z + pz[k]);
                h += amp[k] * sx * sz;
                dhx += amp[k] * fx[k] * cx * sz;
                dhz += amp[k] * fz[k] * sx * cz;
            }
            m->verts[at] = V3(x, h, z);
            m->normals[at] = normalized(V3(-dhx, 1.0, -dhz));
            m->uvs[at] = V3((double)i, (double)j, 0.0);
        }
    }, 256);
    m->tris.resize((size_t)2 * n * n);
    auto vid = [&](int i, int j) { return 1 + j * (n + 1) + i; };
    parallel_for((size_t)n, [&](size_t j0, size_t j1) {
    for (int j = (int)j0; j < (int)j1; ++j)
        for (int i = 0; i < n; ++i) {
            const int a = vid(i, j), b = vid(i + 1, j), c = vid(i + 1, j + 1), d = vid(i, j + 1);
            MeshTri t1{}, t2{};
            // counter-clockwise seen from +y so geometric normals point up
            const int q1[3] = {a, d, b}, q2[3] = {b, d, c};
            for (int k = 0; k < 3; ++k) { t1.v[k] = t1.n[k] = t1.t[k] = q1[k]; t2.v[k] = t2.n[k] = t2.t[k] = q2[k]; }
            m->tris[((size_t)j * n + i) * 2] = t1; m->tris[((size_t)j * n + i) * 2 + 1] = t2;
        }
    }, 256);
    m->generate_triangle_data();
    S.mesh_src.push_back(std::move(m));

    // geometries: 0 = mesh, 1 = floor plane
    qd_geom gm{}; gm.type = QD_GEOM_MESH; gm.a = 0; S.geoms.push_back(gm);
    qd_geom gp{}; gp.type = QD_GEOM_PLANE; gp.p[0] = 0.; gp.p[1] = 1e99; S.geoms.push_back(gp);

    // textures: 0 = checker, 1 = fresnel
    qd_texture chk{}; chk.type = QD_TEX_CHECKER; chk.bitmap = -1; chk.scaling = 0.5;
    chk.color1[0] = 0.9f; chk.color1[1] = 0.85f; chk.color1[2] = 0.7f;
    chk.color2[0] = 0.15f; chk.color2[1] = 0.2f; chk.color2[2] = 0.3f;
    S.textures.push_back(chk);
    qd_texture fr{}; fr.type = QD_TEX_FRESNEL; fr.bitmap = -1; fr.scaling = 1.; fr.p0 = 1.5; S.textures.push_back(fr);

    // shaders: 0 = floor lambert, 1 = refr, 2 = refl, 3 = glass (layered)
    qd_shader lam{}; lam.type = QD_SH_LAMBERT; lam.tex = 0; lam.color[0] = lam.color[1] = lam.color[2] = 1.f; S.shaders.push_back(lam);
    qd_shader refr{}; refr.type = QD_SH_REFR; refr.tex = -1; refr.p0 = 1.5; refr.p1 = 0.95; S.shaders.push_back(refr);
    qd_shader refl{}; refl.type = QD_SH_REFL; refl.tex = -1; refl.p0 = 0.9; refl.p1 = 1.; refl.i0 = 32; S.shaders.push_back(refl);
    qd_shader glass{}; glass.type = QD_SH_LAYERED; glass.tex = -1; glass.i0 = 0; glass.i1 = 2; S.shaders.push_back(glass);
    qd_layer l0{}; l0.shader = 1; l0.tex = -1; l0.blend[0] = l0.blend[1] = l0.blend[2] = 1.f; S.layers.push_back(l0);
    qd_layer l1{}; l1.shader = 2; l1.tex = 1; l1.blend[0] = l1.blend[1] = l1.blend[2] = 1.f; S.layers.push_back(l1);

    auto make_node = [&](int geom, int shader, double sx, double sy, double sz, const V3& off, float shadow_t) {
        qd_node nd{}; nd.geom = geom; nd.shader = shader; nd.bump_tex = -1; nd.shadow_transparency = shadow_t;
        M3 M(sx); M.m[1][1] = sy; M.m[2][2] = sz;
        M = M3(1.) * M;
        const M3 inv = inverse(M), invT = transpose(inv);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { nd.M[i * 3 + j] = M.m[i][j]; nd.Minv[i * 3 + j] = inv.m[i][j]; nd.MinvT[i * 3 + j] = invT.m[i][j]; }
        nd.offset[0] = off.x; nd.offset[1] = off.y; nd.offset[2] = off.z;
        S.nodes.push_back(nd);
    };
    make_node(1, 0, 1, 1, 1, V3(0, -40, 0), 0.f);          // floor
    make_node(0, 3, 100, 100, 100, V3(0, 0, 0), 0.6f);     // glass sheet

    qd_light L{}; L.pos[0] = 60; L.pos[1] = 300; L.pos[2] = -80; L.intensity = 120000; S.lights.push_back(L);

    qd_settings& st = S.desc.settings;
    st.frame_width = width; st.frame_height = height;
    st.ambient[0] = st.ambient[1] = st.ambient[2] = 0.25f;
    st.want_aa = 0; st.want_adaptive_aa = 0; st.max_trace_depth = (uint32_t)max_trace_depth;
    // slightly off-axis on purpose: a camera in the x = 0 plane would shoot a whole pixel column exactly along
    // shared triangle edges and KD split planes (exact-tie rays)
    const double cam_pos[3] = {7.3, 70, -140};
    qdh_camera_corners(cam_pos, 3., -25., 0., 90., 1.7778, &S.desc.camera);

    if (env_folder && env_folder[0]) {
        std::string err;
        if (!S.load_env(env_folder, true, err)) { g_err = err; delete s; return nullptr; }
    }
    S.finalize();
    g_err.clear();
    return s;
}
