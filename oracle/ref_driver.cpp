/*
 * ref_driver.cpp -- drives the UNMODIFIED reference sources to produce golden vectors.
 *
 * TEST INFRASTRUCTURE, build-container only (/root/reference does not exist on the GPU box).  This file is ours;
 * everything it calls is the reference's own object code, compiled by oracle/ref_build.sh from the sources where they
 * lie under /root/reference/src, with no stand-in headers or libraries: the sixteen translation units that include
 * neither SDL nor OpenEXR (vector matrix color colors transform geometry utils KDTree bbox camera light shading
 * shadinghelper texture environment scene).  main.cpp, mesh.cpp, heightfield.cpp, random_generator.cpp, bitmap.cpp and
 * sdl.cpp need SDL / OpenEXR headers the image lacks and are NOT built; symbols they would define (Raytrace, Mesh,
 * Heightfield, Bitmap, GetFrameWidth ...) stay unresolved in the link (--unresolved-symbols=ignore-all) and are never
 * reached: the scenes driven here hold no mesh, heightfield, bitmap, environment, reflection or refraction.
 *
 *   ref_driver scene <file.qdmg> <rays.bin> <out.bin>
 *       The reference's parser cannot run here (DefaultSceneParser::Parse calls GetRandomGen() of random_generator.cpp on
 *       entry, src/scene.cpp:393, and its ParsedBlockImpl is private to scene.cpp), so this harness reads the .qdmg blocks
 *       itself and hands each one to the element's OWN FillProperties() through the reference's public plugin interfaces
 *       ParsedBlock / SceneParser (src/scene.h:111-191): every member of every scene element is set by reference code.
 *       Elements are created and filled in the reference's order (src/scene.cpp:561-610), then Scene::BeginRender /
 *       BeginFrame run.  For every ray {start[3], dir[3]} (f64) the node loop, bump and Shade of Raytrace (src/main.cpp:23-49) are
 *       performed through the public Node::Intersect / Texture::ModifyNormal / Shader::Shade; one 96-byte record per ray:
 *       i32 node (-1 = miss), i32 0, f64 distance, f64 ip[3], f64 normal[3] (as Node::Intersect returned it, before
 *       the bump), f64 u, v, f32 rgb[3] (Shade's result; 0 for a miss), f32 0.
 *   ref_driver funcs <in.bin> <out.bin>
 *       function-level vectors; per 16-f64 input record {a[3], b[3], c[3], d[3], s, 0, 0, 0}:
 *       BBox{min = min(a,b), max = max(a,b)}: IsInside(c), TestIntersect(ray c -> normalize(d)), ClosestIntersection(same),
 *       Split(axis = record index % 3, where = midpoint) -> left.max[axis], right.min[axis];
 *       Reflect(d, n), Refract(dn, n, s), Faceforward(dn, n), FresnelRatio(dn, n, s), OrthonormalSystem(n) with
 *       n = normalize(a - b), dn = normalize(d); Inverse / Determinant / Transpose of the matrix with rows a, b, c;
 *       Transform{Scale(a), Rotate(b * 90), Translate(c)}: Point(d), UndoPoint(d), Direction(d), UndoDirection(d), Normal(d).
 *       Output: 80 f64 per record (layout in emit_funcs below).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <string>

#include "bbox.h"
#include "camera.h"
#include "geometry.h"
#include "light.h"
#include "matrix.h"
#include "scene.h"
#include "shading.h"
#include "texture.h"
#include "transform.h"
#include "utils.h"
#include "vector.h"

static std::vector<double> read_f64(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<double> v((size_t)n / 8);
    if (fread(v.data(), 8, v.size(), f) != v.size()) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

// ---- harness: .qdmg blocks -> SceneElement::FillProperties through the reference's ParsedBlock / SceneParser interfaces
namespace harness {

struct Prop { std::string name, value; int line; };
struct Block { std::string cls, name; std::vector<Prop> props; SceneElement* el = nullptr; int line = 0; };

static std::string trim(const std::string& t) {
    size_t a = 0, b = t.size();
    while (a < b && isspace((unsigned char)t[a])) ++a;
    while (b > a && isspace((unsigned char)t[b - 1])) --b;
    return t.substr(a, b - a);
}

struct Parser : public SceneParser {
    std::vector<Block> blocks;
    template <class T> T* find(const char* name, ElementType type) {
        for (Block& b : blocks)
            if (b.el && b.el->GetElementType() == type && b.name == name)
                if (T* t = dynamic_cast<T*>(b.el)) return t;
        return nullptr;
    }
    Shader* FindShaderByName(const char* name) override { return find<Shader>(name, ElementType::SHADER); }
    Texture* FindTextureByName(const char* name) override { return find<Texture>(name, ElementType::TEXTURE); }
    Geometry* FindGeometryByName(const char* name) override { return find<Geometry>(name, ElementType::GEOMETRY); }
    Node* FindNodeByName(const char* name) override { return find<Node>(name, ElementType::NODE); }
    bool ResolveFullPath(char*) override { return false; }      // no file-backed assets in the scenes driven here
};

struct PB : public ParsedBlock {
    Block& b; Parser& p;
    PB(Block& b_, Parser& p_) : b(b_), p(p_) {}
    const Prop* get(const char* name) { for (const Prop& q : b.props) if (q.name == name) return &q; return nullptr; }
    [[noreturn]] void bad(const Prop* q, const char* what) { fprintf(stderr, "line %d: %s\n", q ? q->line : b.line, what); exit(3); }
    template <class T> bool num(const char* name, const char* fmt, T* value, double lo, double hi) {
        const Prop* q = get(name); if (!q) return false;
        T x; if (1 != sscanf(q->value.c_str(), fmt, &x)) bad(q, "invalid number");
        if ((double)x < lo || (double)x > hi) bad(q, "value outside the allowed bounds");
        *value = x; return true;
    }
    bool GetIntProp(const char* n, int* v, int lo, int hi) override { return num(n, "%d", v, lo, hi); }
    bool GetUnsignedProp(const char* n, unsigned* v, unsigned hi) override { return num(n, "%u", v, 0, hi); }
    bool GetFloatProp(const char* n, float* v, float lo, float hi) override { return num(n, "%f", v, lo, hi); }
    bool GetDoubleProp(const char* n, double* v, double lo, double hi) override { return num(n, "%lf", v, lo, hi); }
    bool GetBoolProp(const char* n, bool* v) override {
        const Prop* q = get(n); if (!q) return false;
        *v = !(q->value == "off" || q->value == "false" || q->value == "0"); return true;
    }
    bool triple(const char* n, double out[3]) {
        const Prop* q = get(n); if (!q) return false;
        char buf[512]; strncpy(buf, q->value.c_str(), sizeof buf - 1); buf[sizeof buf - 1] = 0;
        Get3Doubles(q->line, buf, out[0], out[1], out[2]);             // the reference's own helper (src/scene.cpp)
        return true;
    }
    bool GetColorProp(const char* n, Color* v, float lo, float hi) override {
        const Prop* q = get(n); if (!q) return false;
        char buf[512]; strncpy(buf, q->value.c_str(), sizeof buf - 1); buf[sizeof buf - 1] = 0;
        StripBracesAndCommas(buf);                                      // the reference's own helper
        Color c; if (3 != sscanf(buf, "%f%f%f", &c.r, &c.g, &c.b)) bad(q, "invalid color");
        if (c.r < lo || c.r > hi || c.g < lo || c.g > hi || c.b < lo || c.b > hi) bad(q, "color outside the allowed bounds");
        *v = c; return true;
    }
    bool GetVectorProp(const char* n, Vector* v) override { double t[3]; if (!triple(n, t)) return false; *v = Vector(t[0], t[1], t[2]); return true; }
    bool GetGeometryProp(const char* n, Geometry** v) override {
        const Prop* q = get(n); if (!q) return false;
        if (!(*v = p.FindGeometryByName(q->value.c_str()))) bad(q, "geometry not defined"); return true;
    }
    bool GetIntersectableProp(const char* n, Intersectable** v) override {
        const Prop* q = get(n); if (!q) return false;
        if (Geometry* g = p.FindGeometryByName(q->value.c_str())) { *v = g; return true; }
        Node* nd = p.FindNodeByName(q->value.c_str()); if (!nd) bad(q, "intersectable not defined");
        *v = static_cast<Intersectable*>(nd); return true;
    }
    bool GetShaderProp(const char* n, Shader** v) override {
        const Prop* q = get(n); if (!q) return false;
        if (!(*v = p.FindShaderByName(q->value.c_str()))) bad(q, "shader not defined"); return true;
    }
    bool GetTextureProp(const char* n, Texture** v) override {
        const Prop* q = get(n); if (!q) return false;
        if (!(*v = p.FindTextureByName(q->value.c_str()))) bad(q, "texture not defined"); return true;
    }
    bool GetNodeProp(const char* n, Node** v) override {
        const Prop* q = get(n); if (!q) return false;
        if (!(*v = p.FindNodeByName(q->value.c_str()))) bad(q, "node not defined"); return true;
    }
    bool GetStringProp(const char* n, char* v) override { const Prop* q = get(n); if (!q) return false; strcpy(v, q->value.c_str()); return true; }
    bool GetFilenameProp(const char* n, char*) override { const Prop* q = get(n); if (!q) return false; bad(q, "file-backed assets are not supported by this harness"); }
    bool GetBitmapFileProp(const char* n, Bitmap&) override { const Prop* q = get(n); if (!q) return false; bad(q, "bitmaps are not supported by this harness"); }
    void GetTransformProp(Transform& T) override {                     // scale / rotate / translate lines in file order
        for (const Prop& q : b.props) {
            const int kind = q.name == "scale" ? 0 : q.name == "rotate" ? 1 : q.name == "translate" ? 2 : -1;
            if (kind < 0) continue;
            char buf[512]; strncpy(buf, q.value.c_str(), sizeof buf - 1); buf[sizeof buf - 1] = 0;
            double x, y, z; Get3Doubles(q.line, buf, x, y, z);
            if (kind == 0) T.Scale(x, y, z); else if (kind == 1) T.Rotate(x, y, z); else T.Translate(Vector(x, y, z));
        }
    }
    void RequiredProp(const char* n) override { if (!get(n)) bad(nullptr, "required property missing"); }
    void SignalError(const char* m) override { bad(nullptr, m); }
    void SignalWarning(const char* m) override { fprintf(stderr, "warning: %s\n", m); }
    int GetBlockLines() override { return (int)b.props.size(); }
    void GetBlockLine(int i, int& line, char head[], char tail[]) override { line = b.props[i].line; strcpy(head, b.props[i].name.c_str()); strcpy(tail, b.props[i].value.c_str()); }
    SceneParser& GetParser() override { return p; }
};

// the class table of src/scene.cpp:871-918 restricted to what links without mesh / heightfield / bitmap / environment
static SceneElement* create(const std::string& c) {
    if (c == "GlobalSettings") return &scene.settings;
    if (c == "Plane") return new Plane;
    if (c == "RegularPolygon") return new RegularPolygon;
    if (c == "Sphere") return new Sphere;
    if (c == "Cube") return new Cube;
    if (c == "CsgPlus") return new CsgPlus;
    if (c == "CsgAnd") return new CsgAnd;
    if (c == "CsgMinus") return new CsgMinus;
    if (c == "Node") return new Node;
    if (c == "Lambert") return new Lambert;
    if (c == "Phong") return new Phong;
    if (c == "BlinnPhong") return new BlinnPhong;
    if (c == "OrenNayar") return new OrenNayar;
    if (c == "Layered") return new Layered;
    if (c == "Const") return new ConstColorShader;
    if (c == "ConstantColorTexture" || c == "ConstColorShader") return new ConstantColorTexture;
    if (c == "CheckerTexture") return new CheckerTexture;
    if (c == "MandelbrotTexture") return new MandelbrotTexture;
    if (c == "ProceduralTexture") return new ProceduralTexture;
    if (c == "Fresnel") return new Fresnel;
    if (c == "Bumps") return new BumpsTexture;
    if (c == "Camera") return new Camera;
    if (c == "Light") return new Light;
    return nullptr;
}

static bool load(const char* path) {
    FILE* f = fopen(path, "rt");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return false; }
    static Parser parser;
    char buf[1024];
    int line = 0;
    Block* cur = nullptr;
    while (fgets(buf, sizeof buf, f)) {
        ++line;
        std::string t(buf);
        const size_t c1 = t.find("//"), c2 = t.find('#');
        t = trim(t.substr(0, std::min(c1, c2)));
        if (t.empty()) continue;
        if (!cur) {
            if (t.back() != '{') { fprintf(stderr, "line %d: expected a block\n", line); return false; }
            t = trim(t.substr(0, t.size() - 1));
            Block b; b.line = line;
            const size_t sp = t.find_first_of(" \t");
            b.cls = t.substr(0, sp); b.name = sp == std::string::npos ? "" : trim(t.substr(sp));
            parser.blocks.push_back(b); cur = &parser.blocks.back();
        } else if (t == "}") cur = nullptr;
        else {
            const size_t sp = t.find_first_of(" \t");
            Prop q; q.line = line; q.name = t.substr(0, sp); q.value = sp == std::string::npos ? "" : trim(t.substr(sp));
            cur->props.push_back(q);
        }
    }
    fclose(f);
    for (Block& b : parser.blocks) {
        b.el = create(b.cls);
        if (!b.el) { fprintf(stderr, "line %d: class %s is not available to this harness\n", b.line, b.cls.c_str()); return false; }
        strncpy(b.el->name, b.name.c_str(), sizeof b.el->name - 1);
        switch (b.el->GetElementType()) {                               // what the reference's parser does per new element
            case ElementType::GEOMETRY: scene.geometries.push_back(dynamic_cast<Geometry*>(b.el)); break;
            case ElementType::SHADER: scene.shaders.push_back(dynamic_cast<Shader*>(b.el)); break;
            case ElementType::NODE: scene.nodes.push_back(dynamic_cast<Node*>(b.el)); break;
            case ElementType::TEXTURE: scene.textures.push_back(dynamic_cast<Texture*>(b.el)); break;
            case ElementType::CAMERA: scene.camera = dynamic_cast<Camera*>(b.el); break;
            case ElementType::LIGHT: scene.lights.push_back(dynamic_cast<Light*>(b.el)); break;
            default: break;
        }
    }
    const ElementType order[] = {ElementType::SETTINGS, ElementType::CAMERA, ElementType::ENVIRONMENT, ElementType::LIGHT,
                                 ElementType::GEOMETRY, ElementType::TEXTURE, ElementType::SHADER, ElementType::NODE};   // src/scene.cpp:561-571
    for (ElementType et : order)
        for (Block& b : parser.blocks)
            if (b.el->GetElementType() == et) {
                PB pb(b, parser);
                try { b.el->FillProperties(pb); }
                catch (SyntaxError e) { fprintf(stderr, "line %d: %s\n", e.line, e.msg); return false; }
                catch (FileNotFoundError e) { fprintf(stderr, "line %d: file not found %s\n", e.line, e.filename); return false; }
            }
    for (size_t i = scene.nodes.size(); i > 0; --i)                      // shader-less nodes leave the list, src/scene.cpp:603-610
        if (!scene.nodes[i - 1]->shader) { scene.superNodes.push_back(scene.nodes[i - 1]); scene.nodes.erase(scene.nodes.begin() + i - 1); }
    return scene.camera != nullptr;
}

}  // namespace harness

static int run_scene(const char* qdmg, const char* rays_path, const char* out_path) {
    if (!harness::load(qdmg)) { fprintf(stderr, "scene load failed\n"); return 1; }
    scene.BeginRender();
    scene.BeginFrame();
    const std::vector<double> rays = read_f64(rays_path);
    const size_t n = rays.size() / 6;
    FILE* out = fopen(out_path, "wb");
    if (!out) return 2;
    for (size_t i = 0; i < n; ++i) {
        Ray ray;
        ray.start = Vector(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
        ray.dir = Vector(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        int closest = -1;
        double closestDist = INF;
        IntersectionInfo closestInfo = IntersectionInfo();
        for (size_t k = 0; k < scene.nodes.size(); ++k) {           // src/main.cpp:26-38
            IntersectionInfo info = IntersectionInfo();
            if (!scene.nodes[k]->Intersect(ray, info)) continue;
            if (closestDist <= info.distance) continue;
            closestDist = info.distance; closest = (int)k; closestInfo = info;
        }
        struct { int node, pad; double dist, ip[3], n[3], u, v; float rgb[3], padf; } rec;
        memset(&rec, 0, sizeof rec);
        static_assert(sizeof rec == 96, "record layout");
        rec.node = closest;
        if (closest >= 0) {
            const Node* nd = scene.nodes[closest];
            rec.dist = closestInfo.distance;
            rec.ip[0] = closestInfo.ip.x; rec.ip[1] = closestInfo.ip.y; rec.ip[2] = closestInfo.ip.z;
            rec.n[0] = closestInfo.normal.x; rec.n[1] = closestInfo.normal.y; rec.n[2] = closestInfo.normal.z;
            rec.u = closestInfo.u; rec.v = closestInfo.v;
            closestInfo.rayDir = ray.dir;                            // src/main.cpp:44-49
            if (nd->bump) nd->bump->ModifyNormal(closestInfo);
            const Color c = nd->shader->Shade(ray, closestInfo);
            rec.rgb[0] = c.r; rec.rgb[1] = c.g; rec.rgb[2] = c.b;
        }
        fwrite(&rec, sizeof rec, 1, out);
    }
    fclose(out);
    printf("ref_driver: %zu rays, %zu nodes\n", n, scene.nodes.size());
    return 0;
}

static void put(std::vector<double>& o, const Vector& v) { o.push_back(v.x); o.push_back(v.y); o.push_back(v.z); }
static void put(std::vector<double>& o, const Matrix& m) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.push_back(m.m[i][j]); }

static int run_funcs(const char* in_path, const char* out_path) {
    const std::vector<double> in = read_f64(in_path);
    const size_t n = in.size() / 16;
    std::vector<double> o;
    o.reserve(n * 80);
    for (size_t i = 0; i < n; ++i) {
        const double* r = &in[i * 16];
        const Vector a(r[0], r[1], r[2]), b(r[3], r[4], r[5]), c(r[6], r[7], r[8]), d(r[9], r[10], r[11]);
        const double s = r[12];
        const size_t base = o.size();
        // --- BBox (src/bbox.cpp)                                                    [0..5]
        BBox box; box.MakeEmpty(); box.Add(a); box.Add(b);
        Ray ray; ray.start = c; ray.dir = d; ray.dir.Normalize();
        o.push_back(box.IsInside(c) ? 1. : 0.);
        o.push_back(box.TestIntersect(ray) ? 1. : 0.);
        o.push_back(box.ClosestIntersection(ray));
        const Axis axis = static_cast<Axis>(i % 3);
        BBox left, right;
        box.Split(axis, (box.GetMin(axis) + box.GetMax(axis)) / 2, left, right);
        o.push_back(left.GetMax(axis)); o.push_back(right.GetMin(axis));
        o.push_back(box.GetArea());
        // --- vector helpers (src/vector.cpp:56-90, src/utils.cpp:49-61)             [6..22]
        Vector nrm = a - b; nrm.Normalize();
        Vector dn = d; dn.Normalize();
        put(o, Reflect(d, nrm));
        put(o, Refract(dn, nrm, s));
        put(o, Faceforward(dn, nrm));
        o.push_back(FresnelRatio(dn, nrm, s));
        Vector r1, r2; OrthonormalSystem(nrm, r1, r2);
        put(o, r1); put(o, r2);
        o.push_back(0.);
        // --- matrices (src/matrix.cpp)                                              [23..59]
        Matrix m; for (int k = 0; k < 3; ++k) { m.m[0][k] = a.v[k]; m.m[1][k] = b.v[k]; m.m[2][k] = c.v[k]; }
        o.push_back(Determinant(m));
        put(o, Inverse(m));
        put(o, Transpose(m));
        put(o, RotationAroundX(s) * RotationAroundY(r[0]) * RotationAroundZ(r[1]));
        put(o, d * m);
        put(o, a ^ b);
        o.push_back(a * b);
        o.push_back(a.Length());
        o.push_back(0.); o.push_back(0.);
        // --- Transform (src/transform.cpp)                                          [60..74]
        Transform t;
        t.Scale(a.x, a.y, a.z);
        t.Rotate(b.x * 90, b.y * 90, b.z * 90);
        t.Translate(c);
        put(o, t.Point(d)); put(o, t.UndoPoint(d)); put(o, t.Direction(d)); put(o, t.UndoDirection(d)); put(o, t.Normal(d));
        while (o.size() - base < 80) o.push_back(0.);
        if (o.size() - base != 80) { fprintf(stderr, "layout error: %zu\n", o.size() - base); return 3; }
    }
    FILE* out = fopen(out_path, "wb");
    if (!out) return 2;
    fwrite(o.data(), 8, o.size(), out);
    fclose(out);
    printf("ref_driver: %zu function records\n", n);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 5 && !strcmp(argv[1], "scene")) return run_scene(argv[2], argv[3], argv[4]);
    if (argc == 4 && !strcmp(argv[1], "funcs")) return run_funcs(argv[2], argv[3]);
    fprintf(stderr, "usage: ref_driver scene <file.qdmg> <rays.bin> <out.bin> | ref_driver funcs <in.bin> <out.bin>\n");
    return 64;
}
