// .qdmg scene loader and flattener: text scene -> qd_scene_desc.
//
// Grammar and property semantics follow the reference's DefaultSceneParser
// (src/scene.cpp:391-612), property getters (:131-313), the class table (:871-918),
// GlobalSettings (:851-869) and each element's FillProperties:
//   Camera src/camera.cpp:43-52, Light src/light.cpp:3-7, Plane/Sphere/Cube/Csg/RegularPolygon/Node
//   src/geometry.cpp:43-47,:101-105,:178-182,:250-257,:312-317,:341-348, Mesh src/mesh.cpp:170-183,
//   Heightfield src/heightfield.cpp:102-125, shaders src/shading.cpp:36-40,:85-91,:140-145,:200-205,
//   :240-244,:277-338,:346-349, textures src/texture.cpp:16-21,:61-66,:84-113,:120-123,:201-210,
//   :251-260,:298-301, CubemapEnvironment src/environment.cpp:7-37,:110-121.
// Transform lines are applied in file order (src/scene.cpp:286-313, src/transform.cpp:18-42).
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "heightfield.h"
#include "image.h"
#include "mesh.h"
#include "qdhost.h"
#include "scene_impl.h"
#include "vecmath.h"

namespace qdh {

namespace {

struct ParseError { std::string msg; };

[[noreturn]] void fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw ParseError{buf};
}

enum class Kind { Settings, Camera, Environment, Light, Geometry, Texture, Shader, Node };

struct PropLine { int line; std::string name, value; bool recognized = false; };

struct Block {
    Kind kind;
    std::string cls, name;
    int line_begin = 0, line_end = 0;
    std::vector<PropLine> lines;
    int index = -1;   // index within its kind's list

    PropLine* find(const char* n) {
        for (PropLine& l : lines)
            if (l.name == n) { l.recognized = true; return &l; }   // first match wins (src/scene.cpp:108-123)
        return nullptr;
    }
    void required(const char* n) { if (!find(n)) fail("line %d: Required property `%s' not defined", line_end, n); }

    static void strip_braces_commas(std::string& s) { for (char& c : s) if (c == ',' || c == '(' || c == ')') c = ' '; }

    bool get_int(const char* n, int* v, int lo = INT32_MIN, int hi = INT32_MAX) {
        PropLine* p = find(n); if (!p) return false;
        int x; if (1 != sscanf(p->value.c_str(), "%d", &x)) fail("line %d: Invalid integer", p->line);
        if (x < lo || x > hi) fail("line %d: Value outside the allowed bounds (%d .. %d)", p->line, lo, hi);
        *v = x; return true;
    }
    bool get_unsigned(const char* n, unsigned* v) {
        PropLine* p = find(n); if (!p) return false;
        unsigned x; if (1 != sscanf(p->value.c_str(), "%u", &x)) fail("line %d: Invalid unsigned", p->line);
        *v = x; return true;
    }
    bool get_bool(const char* n, bool* v) {
        PropLine* p = find(n); if (!p) return false;
        *v = p->value != "off" && p->value != "false" && p->value != "0";   // src/scene.cpp:152-157
        return true;
    }
    bool get_float(const char* n, float* v, float lo = -1e17f, float hi = 1e17f) {
        PropLine* p = find(n); if (!p) return false;
        float x; if (1 != sscanf(p->value.c_str(), "%f", &x)) fail("line %d: Invalid float", p->line);
        if (x < lo || x > hi) fail("line %d: Value outside the allowed bounds (%f .. %f)", p->line, lo, hi);
        *v = x; return true;
    }
    bool get_double(const char* n, double* v, double lo = -1e120, double hi = 1e120) {
        PropLine* p = find(n); if (!p) return false;
        double x; if (1 != sscanf(p->value.c_str(), "%lf", &x)) fail("line %d: Invalid double", p->line);
        if (x < lo || x > hi) fail("line %d: Value outside the allowed bounds (%f .. %f)", p->line, lo, hi);
        *v = x; return true;
    }
    bool get_color(const char* n, float v[3]) {
        PropLine* p = find(n); if (!p) return false;
        strip_braces_commas(p->value);
        float r, g, b;
        if (3 != sscanf(p->value.c_str(), "%f%f%f", &r, &g, &b)) fail("line %d: Invalid color", p->line);
        v[0] = r; v[1] = g; v[2] = b; return true;
    }
    bool get_vector(const char* n, V3* v) {
        PropLine* p = find(n); if (!p) return false;
        strip_braces_commas(p->value);
        double x, y, z;
        if (3 != sscanf(p->value.c_str(), "%lf%lf%lf", &x, &y, &z)) fail("line %d: Invalid vector", p->line);
        *v = V3(x, y, z); return true;
    }
};

void get3doubles(int line, std::string s, double& a, double& b, double& c) {
    Block::strip_braces_commas(s);
    if (3 != sscanf(s.c_str(), "%lf%lf%lf", &a, &b, &c)) fail("line %d: Expected three double values", line);
}

Kind kind_of_class(const std::string& c, bool& known) {
    known = true;
    if (c == "GlobalSettings") return Kind::Settings;
    static const char* geoms[] = {"Plane", "RegularPolygon", "Sphere", "Cube", "CsgPlus", "CsgAnd", "CsgMinus", "Mesh", "Heightfield"};
    for (const char* g : geoms) if (c == g) return Kind::Geometry;
    if (c == "Node") return Kind::Node;
    static const char* shaders[] = {"Lambert", "Phong", "BlinnPhong", "OrenNayar", "Refl", "Refr", "Layered", "Const"};
    for (const char* s : shaders) if (c == s) return Kind::Shader;
    static const char* texs[] = {"ConstantColorTexture", "CheckerTexture", "MandelbrotTexture", "ProceduralTexture", "BitmapTexture",
                                 "Fresnel", "BumpTexture", "ConstColorShader", "Bumps"};
    for (const char* t : texs) if (c == t) return Kind::Texture;
    if (c == "CubemapEnvironment") return Kind::Environment;
    if (c == "Camera") return Kind::Camera;
    if (c == "Light") return Kind::Light;
    known = false;
    return Kind::Settings;
}

void strip_ws(std::string& s) {
    size_t e = s.size();
    while (e > 0 && isspace((unsigned char)s[e - 1])) --e;
    s.resize(e);
    size_t b = 0;
    while (b < s.size() && isspace((unsigned char)s[b])) ++b;
    s.erase(0, b);
}

std::vector<std::string> tokenize(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0, l = s.size();
    while (i < l) {
        while (i < l && isspace((unsigned char)s[i])) i++;
        if (i >= l) break;
        size_t j = i;
        while (j < l && !isspace((unsigned char)s[j])) j++;
        out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

struct Loader {
    SceneImpl& S;
    std::string root_dir;
    std::vector<Block> blocks;
    std::vector<Block*> geoms, shaders, textures, nodes, lights;
    Block* camera = nullptr;
    Block* env = nullptr;

    // camera parameters
    V3 cam_pos; double cam_aspect = 4. / 3., cam_fov = 90., cam_yaw = 0, cam_pitch = 0, cam_roll = 0;

    explicit Loader(SceneImpl& s) : S(s) {}

    std::string resolve(const std::string& rel, int line) {
        const std::string full = root_dir + rel;
        if (!file_exists(full)) fail("line %d: Required file not found (%s)", line, rel.c_str());
        return full;
    }

    int find_named(const std::vector<Block*>& list, const std::string& name) {
        for (Block* b : list) if (b->name == name) return b->index;
        return -1;
    }

    void read_file(const char* filename) {
        FILE* f = fopen(filename, "rt");
        if (!f) fail("Cannot open scene file `%s'!", filename);
        std::string fn(filename);
        size_t i = fn.size();
        while (i > 0 && fn[i - 1] != '/' && fn[i - 1] != '\\') --i;
        root_dir = fn.substr(0, i);

        char raw[1024];
        bool commented = false;
        int cur_line = 0;
        bool in_block = false;
        std::vector<std::unique_ptr<Block>> owned;
        std::string err;
        while (fgets(raw, sizeof raw, f)) {
            ++cur_line;
            if (commented) { if (raw[0] == '*' && raw[1] == '/') commented = false; continue; }
            std::string line(raw);
            size_t cpos = std::string::npos;
            const size_t c1 = line.find("//"), c2 = line.find('#');
            if (c1 != std::string::npos) cpos = c1;
            if (c2 != std::string::npos && (cpos == std::string::npos || c2 < cpos)) cpos = c2;
            if (cpos != std::string::npos) line.resize(cpos);
            strip_ws(line);
            if (line.empty()) continue;
            if (line[0] == '/' && line.size() > 1 && line[1] == '*') { commented = true; continue; }
            if (line.find("randfloat") != std::string::npos || line.find("randint") != std::string::npos) {
                err = "line " + std::to_string(cur_line) + ": randfloat/randint substitution is not supported by this loader";
                break;
            }
            std::vector<std::string> tok = tokenize(line);
            if (!in_block) {
                if (tok.size() == 1) { err = "Unexpected token `" + tok[0] + "' on line " + std::to_string(cur_line); break; }
                if (tok.size() > 3) { err = "Unexpected content on line " + std::to_string(cur_line) + "!"; break; }
                if (tok.back() != "{") { err = "An object definition should end with a `{' (on line " + std::to_string(cur_line) + ")"; break; }
                bool known = false;
                const Kind k = kind_of_class(tok[0], known);
                if (!known) { err = "Unknown object class `" + tok[0] + "' on line " + std::to_string(cur_line); break; }
                owned.emplace_back(new Block);
                Block& b = *owned.back();
                b.kind = k; b.cls = tok[0]; b.name = tok[1];   // singletons end up named "{" like the reference
                b.line_begin = cur_line;
                in_block = true;
            } else {
                Block& b = *owned.back();
                if (tok.size() == 1) {
                    if (tok[0] == "}") { b.line_end = cur_line; in_block = false; }
                    else { err = "Unexpected token in object definition on line " + std::to_string(cur_line) + ": `" + tok[0] + "'"; break; }
                } else {
                    size_t p = tok[0].size();
                    while (p < line.size() && isspace((unsigned char)line[p])) ++p;
                    std::string value = line.substr(p);
                    if (value.size() >= 2 && value.front() == '"' && value.back() == '"') value = value.substr(1, value.size() - 2);
                    if (value.size() > 255) value.resize(255);
                    b.lines.push_back(PropLine{cur_line, tok[0], value, false});
                }
            }
        }
        fclose(f);
        if (!err.empty()) fail("%s", err.c_str());
        if (in_block) fail("Unfinished object definition at EOF!");
        for (auto& up : owned) blocks.push_back(std::move(*up));
        for (Block& b : blocks) {
            switch (b.kind) {
                case Kind::Geometry: b.index = (int)geoms.size(); geoms.push_back(&b); break;
                case Kind::Shader: b.index = (int)shaders.size(); shaders.push_back(&b); break;
                case Kind::Texture: b.index = (int)textures.size(); textures.push_back(&b); break;
                case Kind::Node: b.index = (int)nodes.size(); nodes.push_back(&b); break;
                case Kind::Light: b.index = (int)lights.size(); lights.push_back(&b); break;
                case Kind::Camera: camera = &b; break;         // last one wins
                case Kind::Environment: env = &b; break;
                case Kind::Settings: break;
            }
        }
    }

    // ---- per-kind property filling ----------------------------------------

    void fill_settings(Block& b) {
        qd_settings& s = S.desc.settings;
        b.get_int("frameWidth", &s.frame_width);
        b.get_int("frameHeight", &s.frame_height);
        b.get_color("ambientLight", s.ambient);
        bool v;
        if (b.get_bool("wantAA", &v)) s.want_aa = v;
        if (b.get_bool("wantAdaptiveAA", &v)) s.want_adaptive_aa = v;
        b.get_double("aaThreshold", &s.aa_threshold);
        b.get_unsigned("maxTraceDepth", &s.max_trace_depth);
        if (b.get_bool("dbg", &v)) {}
        if (b.get_bool("showAA", &v)) s.show_aa = v;
        b.get_color("aaDebugColor", s.aa_debug_color);
        if (b.get_bool("useSRGB", &v)) s.use_srgb = v;
    }

    void fill_camera(Block& b) {
        b.required("position");
        b.get_vector("position", &cam_pos);
        b.get_double("aspectRatio", &cam_aspect, 1e-6);
        b.get_double("fov", &cam_fov, 0.0001, 179);
        b.get_double("yaw", &cam_yaw);
        b.get_double("pitch", &cam_pitch, -90, 90);
        b.get_double("roll", &cam_roll);
    }

    int add_bitmap(Image&& img) {
        qd_bitmap bm; bm.width = img.w; bm.height = img.h; bm.texel_off = S.texels.size() / 3;
        S.texels.insert(S.texels.end(), img.rgb.begin(), img.rgb.end());
        S.bitmaps.push_back(bm);
        return (int)S.bitmaps.size() - 1;
    }

    void fill_env(Block& b) {
        bool bil = false;
        b.get_bool("useBilinearFiltering", &bil);
        b.required("folder");
        PropLine* p = b.find("folder");
        const std::string folder = resolve(p->value, p->line);
        load_env_folder(folder, bil);
    }

    void load_env_folder(const std::string& folder, bool bilinear) {
        static const char* prefixes[2] = {"neg", "pos"};
        static const char* axes[3] = {"x", "y", "z"};
        static const char* suffixes[2] = {".bmp", ".exr"};
        int n = 0;
        for (const char* pre : prefixes)
            for (const char* ax : axes) {
                Image img; std::string err, last;
                for (const char* suf : suffixes) {
                    const std::string fn = folder + "/" + pre + ax + suf;
                    if (file_exists(fn) && load_image(fn.c_str(), img, err)) break;
                    if (!err.empty()) last = err;
                }
                if (!img.ok()) fail("CubemapEnvironment: Could not load maps from `%s' (%s)", folder.c_str(), last.c_str());
                S.desc.env_faces[n++] = add_bitmap(std::move(img));
            }
        S.desc.env_present = 1;
        S.desc.env_bilinear = bilinear ? 1 : 0;
    }

    void fill_light(Block& b) {
        qd_light l{};
        V3 pos; b.get_vector("pos", &pos);
        l.pos[0] = pos.x; l.pos[1] = pos.y; l.pos[2] = pos.z;
        l.intensity = 0;   // the reference leaves it uninitialised when absent (src/light.h:9)
        b.get_double("intensity", &l.intensity);
        S.lights.push_back(l);
    }

    void fill_geometry(Block& b) {
        qd_geom g{};
        if (b.cls == "Plane") {
            g.type = QD_GEOM_PLANE; g.p[0] = 0.; g.p[1] = 1e99;
            b.get_double("y", &g.p[0]);            // note: "height" is NOT a recognised key
            b.get_double("limit", &g.p[1]);
        } else if (b.cls == "Sphere") {
            g.type = QD_GEOM_SPHERE; V3 c; double r = 1.;
            b.get_vector("center", &c); b.get_double("radius", &r, 0.);
            g.p[0] = c.x; g.p[1] = c.y; g.p[2] = c.z; g.p[3] = r;
        } else if (b.cls == "Cube") {
            g.type = QD_GEOM_CUBE; V3 c; double h = .5;
            b.get_vector("center", &c); b.get_double("halfSide", &h);
            g.p[0] = c.x; g.p[1] = c.y; g.p[2] = c.z; g.p[3] = h;
        } else if (b.cls == "RegularPolygon") {
            g.type = QD_GEOM_REGPOLY; V3 c; double r = 1.; unsigned sides = 3;
            b.get_vector("center", &c); b.get_double("radius", &r, 0.); b.get_unsigned("sides", &sides);
            g.p[0] = c.x; g.p[1] = c.y; g.p[2] = c.z; g.p[3] = r; g.a = (int)sides;
        } else if (b.cls == "CsgPlus" || b.cls == "CsgAnd" || b.cls == "CsgMinus") {
            g.type = QD_GEOM_CSG;
            g.c = b.cls == "CsgAnd" ? QD_CSG_AND : (b.cls == "CsgPlus" ? QD_CSG_PLUS : QD_CSG_MINUS);
            b.required("left"); b.required("right");
            PropLine* l = b.find("left"); PropLine* r = b.find("right");
            g.a = find_named(geoms, l->value); g.b = find_named(geoms, r->value);
            if (g.a < 0) fail("line %d: Geometry not defined", l->line);
            if (g.b < 0) fail("line %d: Geometry not defined", r->line);
        } else if (b.cls == "Mesh") {
            std::unique_ptr<MeshData> m(new MeshData);
            b.get_bool("faceted", &m->faceted);
            b.get_bool("backCulling", &m->backcull);
            b.get_bool("useKDTree", &m->use_kd);
            b.get_bool("useSAH", &m->use_sah);
            b.required("file");
            PropLine* p = b.find("file");
            const std::string fn = resolve(p->value, p->line);
            std::string err;
            if (!m->load_obj(fn.c_str(), err)) fail("line %d: %s", b.line_end, err.c_str());
            g.type = QD_GEOM_MESH; g.a = (int)S.mesh_src.size();
            S.mesh_src.push_back(std::move(m));
        } else if (b.cls == "Heightfield") {
            std::unique_ptr<HeightfieldData> h(new HeightfieldData);
            b.required("file");
            PropLine* p = b.find("file");
            const std::string fn = resolve(p->value, p->line);
            Image bmp; std::string err;
            if (!load_image(fn.c_str(), bmp, err)) fail("line %d: %s", p->line, err.c_str());
            double blur = 0.;
            b.get_double("blur", &blur, 0, 1000);
            { std::string perr; if (!h->prepare(bmp, blur, perr)) fail("line %d: %s", p->line, perr.c_str()); }
            b.get_bool("useOptimization", &h->use_optimization);
            g.type = QD_GEOM_HEIGHTFIELD; g.a = (int)S.hf_src.size();
            S.hf_src.push_back(std::move(h));
        }
        S.geoms.push_back(g);
    }

    void fill_texture(Block& b) {
        qd_texture t{};
        t.bitmap = -1; t.scaling = 1.;
        auto load_bitmap_helper = [&](bool diff) {
            double sc = 1.; b.get_double("scaling", &sc);
            t.scaling = 1. / sc;                             // stored inverted (src/texture.cpp:203-204)
            bool bil = false; b.get_bool("useBilinearFiltering", &bil); t.bilinear = bil;
            b.required("file");
            PropLine* p = b.find("file");
            const std::string fn = resolve(p->value, p->line);
            Image img; std::string err;
            load_image(fn.c_str(), img, err);               // a failed load leaves an empty (black) bitmap
            if (diff && img.ok()) differentiate(img);        // BumpmapTexture::BeginRender
            t.bitmap = add_bitmap(std::move(img));
        };
        if (b.cls == "ConstantColorTexture" || b.cls == "ConstColorShader") {
            t.type = QD_TEX_CONST; b.get_color("color", t.color1);
        } else if (b.cls == "CheckerTexture") {
            t.type = QD_TEX_CHECKER;
            b.get_color("color1", t.color1); b.get_color("color2", t.color2); b.get_double("scaling", &t.scaling);
        } else if (b.cls == "MandelbrotTexture") {
            t.type = QD_TEX_MANDELBROT;
            b.get_color("color1", t.color1); b.get_color("color2", t.color2); b.get_double("scaling", &t.scaling);
        } else if (b.cls == "ProceduralTexture") {
            t.type = QD_TEX_PROCEDURAL; t.i0 = (int)S.proc_lines.size();
            for (PropLine& l : b.lines) {
                l.recognized = true;
                if (l.name != "element") continue;
                std::string v = l.value; Block::strip_braces_commas(v);
                qd_proc_line pl{};
                if (8 != sscanf(v.c_str(), "%f%f%f %lf %f%f%f %lf", &pl.color_u[0], &pl.color_u[1], &pl.color_u[2], &pl.freq_u,
                                &pl.color_v[0], &pl.color_v[1], &pl.color_v[2], &pl.freq_v))
                    fail("line %d: Expected a procedural in the format <colorU> <freqU> <colorV> <freqV>", l.line);
                S.proc_lines.push_back(pl);
            }
            t.i1 = (int)S.proc_lines.size() - t.i0;
        } else if (b.cls == "BitmapTexture") {
            t.type = QD_TEX_BITMAP; load_bitmap_helper(false);
        } else if (b.cls == "Fresnel") {
            t.type = QD_TEX_FRESNEL; t.p0 = 1.33; b.get_double("ior", &t.p0, 1e-6, 10.);
        } else if (b.cls == "BumpTexture") {
            t.type = QD_TEX_BUMPMAP; load_bitmap_helper(true);
            t.p0 = 1.; b.get_double("strength", &t.p0);
        } else if (b.cls == "Bumps") {
            t.type = QD_TEX_BUMPS; float s = 0.f; b.get_float("strength", &s); t.p0 = s;
        }
        S.textures.push_back(t);
    }

    void fill_shader(Block& b) {
        qd_shader& s = S.shaders[b.index];
        auto tex_prop = [&](const char* n, int* out) {
            PropLine* p = b.find(n); if (!p) return;
            const int t = find_named(textures, p->value);
            if (t < 0) fail("line %d: Texture not defined", p->line);
            *out = t;
        };
        if (b.cls == "Const") {
            b.get_color("color", s.color);
        } else if (b.cls == "Lambert") {
            b.get_color("color", s.color); tex_prop("texture", &s.tex);
        } else if (b.cls == "Phong" || b.cls == "BlinnPhong") {
            b.get_color("color", s.color); tex_prop("texture", &s.tex);
            b.get_double("specularMultiplier", &s.p0); b.get_double("specularExponent", &s.p1);
        } else if (b.cls == "OrenNayar") {
            b.get_color("color", s.color); tex_prop("texture", &s.tex); b.get_double("sigma", &s.p0);
        } else if (b.cls == "Refl") {
            b.get_double("multiplier", &s.p0); b.get_double("glossiness", &s.p1, 0., 1.); b.get_int("numSamples", &s.i0, 1);
        } else if (b.cls == "Refr") {
            b.get_double("multiplier", &s.p1); b.get_double("ior", &s.p0, 1e-6, 10.);
        } else if (b.cls == "Layered") {
            s.i0 = (int)S.layers.size();
            int count = 0;
            for (PropLine& l : b.lines) {
                l.recognized = true;
                if (l.name != "layer") continue;
                // "layer <shader> (r, g, b) [<texture>|nullptr]"  (src/shading.cpp:277-338)
                std::vector<std::string> tok = tokenize(l.value);
                bool bad = tok.size() < 2;
                std::string shader_name, tex_name, rest;
                if (!bad) {
                    shader_name = tok[0];
                    rest = l.value.substr(l.value.find(tok[0]) + tok[0].size());
                    std::string tr = rest; strip_ws(tr);
                    if (tr.empty()) bad = true;
                    else if (tr.back() != ')') {
                        tex_name = tok.back();
                        const size_t cut = rest.rfind(tex_name);
                        rest = rest.substr(0, cut);
                    }
                }
                auto strip_punct = [](std::string& x) { std::string o; for (char c : x) if (!isspace((unsigned char)c) && c != ',') o.push_back(c); x = o; };
                strip_punct(shader_name); strip_punct(tex_name);
                if (tex_name == "nullptr") tex_name.clear();
                int sh = -1, tx = -1;
                if (!bad) { sh = find_named(shaders, shader_name); bad = sh < 0; }
                if (!bad && !tex_name.empty()) { tx = find_named(textures, tex_name); bad = tx < 0; }
                if (bad) fail("line %d: Expected a line like `layer <shader>, <color>[, <texture>]'", l.line);
                double x, y, z; get3doubles(l.line, rest, x, y, z);
                if (count == 32) continue;   // AddLayer ignores layers beyond 32 (src/shading.cpp:251-258)
                qd_layer ly{}; ly.shader = sh; ly.tex = tx; ly.blend[0] = (float)x; ly.blend[1] = (float)y; ly.blend[2] = (float)z;
                S.layers.push_back(ly); ++count;
            }
            s.i1 = count;
        }
    }

    void init_shader_defaults() {
        S.shaders.resize(shaders.size());
        for (Block* b : shaders) {
            qd_shader s{}; s.tex = -1;
            if (b->cls == "Const") { s.type = QD_SH_CONST; s.color[0] = s.color[1] = s.color[2] = 0.5f; }
            else if (b->cls == "Lambert") s.type = QD_SH_LAMBERT;
            else if (b->cls == "Phong") { s.type = QD_SH_PHONG; s.p0 = 10.; s.p1 = 0.4; }
            else if (b->cls == "BlinnPhong") { s.type = QD_SH_BLINNPHONG; s.p0 = 10.; s.p1 = 0.4; }
            else if (b->cls == "OrenNayar") { s.type = QD_SH_ORENNAYAR; s.p0 = 0.; }
            else if (b->cls == "Refl") { s.type = QD_SH_REFL; s.p0 = 0.99; s.p1 = 1.; s.i0 = 32; }
            else if (b->cls == "Refr") { s.type = QD_SH_REFR; s.p0 = 1.33; s.p1 = 0.99; }
            else if (b->cls == "Layered") s.type = QD_SH_LAYERED;
            S.shaders[b->index] = s;
        }
    }

    // returns false when the node has no shader (dropped from the render list, src/scene.cpp:603-610)
    bool fill_node(Block& b, qd_node& n) {
        n = qd_node{};
        n.geom = -1; n.shader = -1; n.bump_tex = -1; n.shadow_transparency = 0.f;
        if (PropLine* p = b.find("geometry")) {
            n.geom = find_named(geoms, p->value);
            if (n.geom < 0) fail("line %d: Geometry not defined", p->line);
        }
        if (PropLine* p = b.find("shader")) {
            n.shader = find_named(shaders, p->value);
            if (n.shader < 0) fail("line %d: Shader not defined", p->line);
        }
        Transform T;
        for (PropLine& l : b.lines) {
            double x, y, z;
            if (l.name == "scale") { l.recognized = true; get3doubles(l.line, l.value, x, y, z); T.scale(x, y, z); }
            else if (l.name == "rotate") { l.recognized = true; get3doubles(l.line, l.value, x, y, z); T.rotate(x, y, z); }
            else if (l.name == "translate") { l.recognized = true; get3doubles(l.line, l.value, x, y, z); T.translate(V3(x, y, z)); }
        }
        if (PropLine* p = b.find("bump")) {
            n.bump_tex = find_named(textures, p->value);
            if (n.bump_tex < 0) fail("line %d: Texture not defined", p->line);
        }
        b.get_float("shadowTransparency", &n.shadow_transparency);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                n.M[i * 3 + j] = T.m.m[i][j]; n.Minv[i * 3 + j] = T.inv.m[i][j]; n.MinvT[i * 3 + j] = T.invT.m[i][j];
            }
        n.offset[0] = T.offset.x; n.offset[1] = T.offset.y; n.offset[2] = T.offset.z;
        return n.shader >= 0;
    }

    void run(const char* filename) {
        read_file(filename);
        // fill order by element type: settings, camera, environment, light, geometry, texture, shader, node
        for (Block& b : blocks) if (b.kind == Kind::Settings) fill_settings(b);
        if (!camera) fail("scene has no Camera");
        fill_camera(*camera);   // with several Camera blocks the last one is the scene camera (src/scene.cpp:516)
        if (env) fill_env(*env);
        for (Block* b : lights) fill_light(*b);
        for (Block* b : geoms) fill_geometry(*b);
        for (Block* b : textures) fill_texture(*b);
        init_shader_defaults();
        for (Block* b : shaders) fill_shader(*b);
        for (Block* b : nodes) {
            qd_node n;
            if (fill_node(*b, n)) {
                if (n.geom < 0) fail("node `%s' has no geometry", b->name.c_str());
                S.nodes.push_back(n);
            }
        }
        for (Block& b : blocks)
            for (PropLine& l : b.lines)
                if (!l.recognized) S.warnings += std::string(filename) + ":" + std::to_string(l.line) + ": Warning: the property `" + l.name + "' isn't recognized!\n";
        qdh_camera_corners(&cam_pos.x, cam_yaw, cam_pitch, cam_roll, cam_fov, cam_aspect, &S.desc.camera);
    }
};

}  // namespace

SceneImpl::SceneImpl() {
    memset(&desc, 0, sizeof desc);
    qd_settings& s = desc.settings;   // defaults: src/scene.h:230-248
    s.frame_width = 640; s.frame_height = 480;
    s.want_aa = 1; s.want_adaptive_aa = 1; s.aa_threshold = 0.1;
    s.max_trace_depth = 4;
    s.aa_debug_color[0] = 1.f;
    for (int i = 0; i < 6; ++i) desc.env_faces[i] = -1;
}

bool g_eager_kd = false;

// Scene::BeginRender part: mesh bbox, triangle arrays, highmap; then publish flat arrays.  The KD trees of the meshes
// (Mesh::BuildKD) are built by build_kd(), on request.
void SceneImpl::finalize() {
    meshes.clear(); tri_verts.clear(); tri_gnormal.clear(); tri_shade.clear(); kd_nodes.clear(); leaf_tris.clear();
    kd_built = false;
    size_t total = 0;
    for (auto& mp : mesh_src) total += mp->tris.size();
    tri_verts.resize(total * 9); tri_gnormal.resize(total * 3); tri_shade.resize(total * 21);
    size_t first = 0;
    for (auto& mp : mesh_src) {
        MeshData& m = *mp;
        m.compute_bounds();
        qd_mesh qm{};
        qm.tri_first = (uint32_t)first;
        qm.tri_count = (uint32_t)m.tris.size();
        qm.flags = (m.faceted ? QD_MESH_FACETED : 0) | (m.backcull ? QD_MESH_BACKCULL : 0) | (m.wants_kd() ? QD_MESH_WANT_KD : 0);
        qm.bbox_min[0] = m.bmin.x; qm.bbox_min[1] = m.bmin.y; qm.bbox_min[2] = m.bmin.z;
        qm.bbox_max[0] = m.bmax.x; qm.bbox_max[1] = m.bmax.y; qm.bbox_max[2] = m.bmax.z;
        qm.kd_root = -1;
        const size_t n = m.tris.size();
        auto fill = [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const MeshTri& t = m.tris[i];
                const V3& A = m.verts[t.v[0]]; const V3& B = m.verts[t.v[1]]; const V3& C = m.verts[t.v[2]];
                const V3 ab = B - A, ac = C - A;
                double* tv = &tri_verts[(first + i) * 9];
                tv[0] = A.x; tv[1] = A.y; tv[2] = A.z; tv[3] = ab.x; tv[4] = ab.y; tv[5] = ab.z; tv[6] = ac.x; tv[7] = ac.y; tv[8] = ac.z;
                double* gn = &tri_gnormal[(first + i) * 3];
                gn[0] = t.gnormal.x; gn[1] = t.gnormal.y; gn[2] = t.gnormal.z;
                double* sh = &tri_shade[(first + i) * 21];
                for (int k = 0; k < 3; ++k) {
                    const V3& nn = m.normals[t.n[k]];
                    sh[k * 3 + 0] = nn.x; sh[k * 3 + 1] = nn.y; sh[k * 3 + 2] = nn.z;
                    const V3& uv = m.uvs[t.t[k]];
                    sh[9 + k * 2 + 0] = uv.x; sh[9 + k * 2 + 1] = uv.y;
                }
                sh[15] = t.dNdx.x; sh[16] = t.dNdx.y; sh[17] = t.dNdx.z;
                sh[18] = t.dNdy.x; sh[19] = t.dNdy.y; sh[20] = t.dNdy.z;
            }
        };
        parallel_for(n, fill);
        first += n;
        meshes.push_back(qm);
    }
    if (g_eager_kd) build_kd_arrays();

    hfields.clear(); hf_f32.clear(); hf_f64.clear();
    for (auto& hp : hf_src) {
        HeightfieldData& h = *hp;
        if (h.use_optimization && h.highmap.empty()) { std::string perr; if (!h.build_highmap(perr)) throw ParseError{perr}; }
        qd_heightfield q{};
        q.width = h.width; q.height = h.height; q.use_optimization = h.use_optimization; q.max_k = h.max_k;
        q.bbox_min[0] = h.bmin.x; q.bbox_min[1] = h.bmin.y; q.bbox_min[2] = h.bmin.z;
        q.bbox_max[0] = h.bmax.x; q.bbox_max[1] = h.bmax.y; q.bbox_max[2] = h.bmax.z;
        q.heights_off = hf_f32.size(); hf_f32.insert(hf_f32.end(), h.heights.begin(), h.heights.end());
        q.maxheights_off = hf_f32.size(); hf_f32.insert(hf_f32.end(), h.maxheights.begin(), h.maxheights.end());
        q.highmap_off = hf_f32.size(); hf_f32.insert(hf_f32.end(), h.highmap.begin(), h.highmap.end());
        q.normals_off = hf_f64.size();
        for (const V3& n : h.normals) { hf_f64.push_back(n.x); hf_f64.push_back(n.y); hf_f64.push_back(n.z); }
        hfields.push_back(q);
    }
    publish();
}

// The reference-shaped trees (midpoint / "SAH" planes, 20 triangles per leaf): node and leaf arrays of all meshes, rebased.
void SceneImpl::build_kd_arrays() {
    kd_nodes.clear(); leaf_tris.clear();
    for (size_t mi = 0; mi < mesh_src.size(); ++mi) {
        MeshData& m = *mesh_src[mi];
        qd_mesh& qm = meshes[mi];
        m.build_kd();
        qm.kd_max_depth = m.kd_max_depth; qm.kd_leaves = m.kd_leaves;
        qm.kd_depth_sum = m.kd_depth_sum; qm.kd_leaf_tris = m.kd_leaf_tri_sum;
        if (m.kd.empty()) qm.kd_root = -1;
        else {
            // rebase mesh-local node / leaf indices into the global arrays
            const uint32_t node_base = (uint32_t)kd_nodes.size();
            const uint32_t leaf_base = (uint32_t)leaf_tris.size();
            qm.kd_root = (int32_t)node_base;
            kd_nodes.reserve(kd_nodes.size() + m.kd.size());
            for (qd_kdnode n : m.kd) {
                const uint32_t axis = n.meta & 3u, idx = n.meta >> 2;
                n.meta = ((idx + (axis == 3 ? leaf_base : node_base)) << 2) | axis;
                kd_nodes.push_back(n);
            }
            leaf_tris.insert(leaf_tris.end(), m.leaf_tris.begin(), m.leaf_tris.end());
        }
        // the flat arrays now hold everything the trace path needs; drop the builder's copies
        std::vector<qd_kdnode>().swap(m.kd);
        std::vector<uint32_t>().swap(m.leaf_tris);
    }
    kd_built = true;
}

void SceneImpl::build_kd() {
    if (kd_built) return;
    build_kd_arrays();
    publish();
}

void SceneImpl::publish() {
    desc.n_nodes = (uint32_t)nodes.size(); desc.nodes = nodes.data();
    desc.n_geoms = (uint32_t)geoms.size(); desc.geoms = geoms.data();
    desc.n_meshes = (uint32_t)meshes.size(); desc.meshes = meshes.data();
    desc.n_hfields = (uint32_t)hfields.size(); desc.hfields = hfields.data();
    desc.n_shaders = (uint32_t)shaders.size(); desc.shaders = shaders.data();
    desc.n_layers = (uint32_t)layers.size(); desc.layers = layers.data();
    desc.n_textures = (uint32_t)textures.size(); desc.textures = textures.data();
    desc.n_proc_lines = (uint32_t)proc_lines.size(); desc.proc_lines = proc_lines.data();
    desc.n_bitmaps = (uint32_t)bitmaps.size(); desc.bitmaps = bitmaps.data();
    desc.n_lights = (uint32_t)lights.size(); desc.lights = lights.data();
    desc.n_tris = tri_verts.size() / 9;
    desc.tri_verts = tri_verts.data(); desc.tri_gnormal = tri_gnormal.data(); desc.tri_shade = tri_shade.data();
    desc.n_kd_nodes = kd_nodes.size(); desc.kd_nodes = kd_nodes.data();
    desc.n_leaf_tris = leaf_tris.size(); desc.leaf_tris = leaf_tris.data();
    desc.n_texels = texels.size() / 3; desc.texels = texels.data();
    desc.n_hf_f32 = hf_f32.size(); desc.hf_f32 = hf_f32.data();
    desc.n_hf_f64 = hf_f64.size(); desc.hf_f64 = hf_f64.data();
}

bool SceneImpl::load(const char* path, std::string& err) {
    try {
        Loader L(*this);
        L.run(path);
        finalize();
        return true;
    } catch (const ParseError& e) {
        err = e.msg;
        return false;
    }
}

bool SceneImpl::load_env(const std::string& folder, bool bilinear, std::string& err) {
    try {
        Loader L(*this);
        L.load_env_folder(folder, bilinear);
        return true;
    } catch (const ParseError& e) {
        err = e.msg;
        return false;
    }
}

}  // namespace qdh
