#include "mesh.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace qdh {

namespace {

const double kInf = 1e99;                 // src/constants.h:11
const unsigned kMaxTreeDepth = 64;        // src/constants.h:13
const unsigned kTrianglesPerLeaf = 20;    // src/constants.h:14
const double kCostTraversal = 0.3;        // src/constants.h:15
const double kCostIntersect = 1.;         // src/constants.h:16

// whitespace tokeniser (src/utils.cpp:95-109)
void tokenize(const char* s, std::vector<std::string>& out) {
    out.clear();
    size_t i = 0, l = strlen(s);
    while (i < l) {
        while (i < l && isspace((unsigned char)s[i])) i++;
        if (i >= l) break;
        size_t j = i;
        while (j < l && !isspace((unsigned char)s[j])) j++;
        out.emplace_back(s + i, j - i);
        i = j;
    }
}

double to_double(const std::string& s) {      // sscanf("%lf") semantics, 0 on failure
    if (s.empty()) return 0;
    char* end = nullptr;
    const double v = strtod(s.c_str(), &end);
    return end == s.c_str() ? 0 : v;
}

int to_int(const char* b, const char* e) {    // sscanf("%d") semantics, 0 on failure
    if (b == e) return 0;
    std::string t(b, e);
    char* end = nullptr;
    const long v = strtol(t.c_str(), &end, 10);
    return end == t.c_str() ? 0 : (int)v;
}

// "3", "3/4", "3//5", "3/4/5" -> vertex / uv / normal, missing = 0 (src/mesh.cpp:434-442)
void parse_trio(const std::string& s, int& v, int& t, int& n) {
    const char* p = s.c_str();
    const char* e = p + s.size();
    const char* parts[3][2] = {{e, e}, {e, e}, {e, e}};
    int k = 0;
    const char* b = p;
    for (const char* q = p; q <= e && k < 3; ++q)
        if (q == e || *q == '/') { parts[k][0] = b; parts[k][1] = q; ++k; b = q + 1; }
    v = to_int(parts[0][0], parts[0][1]);
    t = k >= 2 ? to_int(parts[1][0], parts[1][1]) : 0;
    n = k >= 3 ? to_int(parts[2][0], parts[2][1]) : 0;
}

// solve x*a + y*b = c in 2-D by Cramer (src/mesh.cpp:454-463)
void solve2d(const V3& a, const V3& b, const V3& c, double& x, double& y) {
    const double m00 = a.x, m01 = b.x, m10 = a.y, m11 = b.y;
    const double h0 = c.x, h1 = c.y;
    const double dcr = m00 * m11 - m10 * m01;
    x = (h0 * m11 - h1 * m01) / dcr;
    y = (m00 * h1 - m10 * h0) / dcr;
}

struct Box { V3 mn, mx; };

double box_area(const Box& b) {   // really 2 * volume (src/bbox.cpp:205-210)
    double r = 2. * (b.mx.x - b.mn.x) * (b.mx.y - b.mn.y) * (b.mx.z - b.mn.z);
    return std::abs(r);
}

struct Builder {
    MeshData& m;
    // per-triangle extent along each axis: PartitionTriangle (src/mesh.cpp:412-432) only asks
    // "is any vertex < pos" and "is any vertex > pos", i.e. min < pos and max > pos.
    std::vector<double> tmin[3], tmax[3];

    explicit Builder(MeshData& mesh) : m(mesh) {
        const size_t n = m.tris.size();
        for (int a = 0; a < 3; ++a) { tmin[a].resize(n); tmax[a].resize(n); }
        for (size_t i = 0; i < n; ++i) {
            const MeshTri& t = m.tris[i];
            for (int a = 0; a < 3; ++a) {
                const double v0 = m.verts[t.v[0]][a], v1 = m.verts[t.v[1]][a], v2 = m.verts[t.v[2]][a];
                tmin[a][i] = std::min(std::min(v0, v1), v2);
                tmax[a][i] = std::max(std::max(v0, v1), v2);
            }
        }
    }

    // 0 = before, 1 = after, 2 = both
    inline int side(uint32_t t, int axis, double pos) const {
        const bool before = tmin[axis][t] < pos, after = tmax[axis][t] > pos;
        if (before) return after ? 2 : 0;
        return after ? 1 : 2;
    }

    void make_leaf(uint32_t node, const std::vector<uint32_t>& list, unsigned depth) {
        m.kd_leaves++;
        m.kd_max_depth = std::max(m.kd_max_depth, depth);
        m.kd_depth_sum += depth;
        m.kd_leaf_tri_sum += list.size();
        m.kd[node].split = 0;
        m.kd[node].meta = ((uint32_t)m.leaf_tris.size() << 2) | 3u;
        m.kd[node].count = (uint32_t)list.size();
        m.leaf_tris.insert(m.leaf_tris.end(), list.begin(), list.end());
    }

    bool find_split(const Box& box, const std::vector<uint32_t>& list, unsigned depth, double& pos, int& axis) const {
        if (!m.use_sah) {
            axis = (int)(depth % 3);
            pos = (box.mn[axis] + box.mx[axis]) / 2;
            return true;
        }
        // "SAH" variant (src/mesh.cpp:272-287, :309-372); the axis picker compares max.z - min.y on purpose
        const double xd = std::abs(box.mx.x - box.mn.x), yd = std::abs(box.mx.y - box.mn.y), zd = std::abs(box.mx.z - box.mn.y);
        if (xd > yd) axis = xd > zd ? 0 : 2;
        else axis = yd > zd ? 1 : 2;
        double best = box_area(box) * (unsigned)list.size() * kCostIntersect;
        bool found = false;
        const double left = box.mn[axis], right = box.mx[axis];
        const double step = (right - left) / 10.;
        for (double sp = left; sp < right; sp += step) {
            Box lb = box, rb = box;
            lb.mx[axis] = sp; rb.mn[axis] = sp;
            unsigned nl = 0, nr = 0;
            for (uint32_t t : list) {
                const int s = side(t, axis, sp);
                if (s != 1) ++nl;
                if (s != 0) ++nr;
            }
            const double cost = kCostTraversal + box_area(lb) * nl * kCostIntersect + box_area(rb) * nr * kCostIntersect;
            if (cost < best) { found = true; best = cost; pos = sp; }
            if (!(step > 0)) break;   // degenerate box: the reference would spin; one candidate is all there is
        }
        return found;
    }

    void build(uint32_t node, const Box& box, const std::vector<uint32_t>& list, unsigned depth) {
        if (depth > kMaxTreeDepth || list.size() < kTrianglesPerLeaf) { make_leaf(node, list, depth); return; }
        double pos = 0; int axis = 0;
        if (!find_split(box, list, depth, pos, axis)) { make_leaf(node, list, depth); return; }
        Box lb = box, rb = box;
        lb.mx[axis] = pos; rb.mn[axis] = pos;
        std::vector<uint32_t> l, r;
        for (uint32_t t : list) {
            const int s = side(t, axis, pos);
            if (s != 1) l.push_back(t);
            if (s != 0) r.push_back(t);
        }
        const uint32_t child = (uint32_t)m.kd.size();
        m.kd.push_back(qd_kdnode{0, 0, 0});
        m.kd.push_back(qd_kdnode{0, 0, 0});
        m.kd[node].split = pos;
        m.kd[node].meta = (child << 2) | (uint32_t)axis;
        m.kd[node].count = 0;
        build(child, lb, l, depth + 1);
        { std::vector<uint32_t>().swap(l); }
        build(child + 1, rb, r, depth + 1);
    }
};

}  // namespace

bool MeshData::load_obj(const char* path, std::string& err) {
    FILE* f = fopen(path, "rt");
    if (!f) { err = std::string("Could not parse OBJ file: ") + path; return false; }
    verts.assign(1, V3(0, 0, 0));
    uvs.assign(1, V3(0, 0, 0));
    normals.assign(1, V3(0, 0, 0));
    tris.clear();
    std::vector<char> line(10000);
    std::vector<std::string> tok;
    bool ok = true;
    while (fgets(line.data(), (int)line.size(), f)) {
        tokenize(line.data(), tok);
        if (tok.empty()) continue;
        if (tok[0] == "v" && tok.size() >= 4) verts.emplace_back(to_double(tok[1]), to_double(tok[2]), to_double(tok[3]));
        else if (tok[0] == "vn" && tok.size() >= 4) normals.emplace_back(to_double(tok[1]), to_double(tok[2]), to_double(tok[3]));
        else if (tok[0] == "vt" && tok.size() >= 3) uvs.emplace_back(to_double(tok[1]), to_double(tok[2]), 0);
        else if (tok[0] == "f" && tok.size() >= 4) {
            for (size_t i = 0; i + 3 < tok.size(); ++i) {   // fan (1, 2+i, 3+i)
                MeshTri t;
                parse_trio(tok[1], t.v[0], t.t[0], t.n[0]);
                parse_trio(tok[2 + i], t.v[1], t.t[1], t.n[1]);
                parse_trio(tok[3 + i], t.v[2], t.t[2], t.n[2]);
                tris.push_back(t);
            }
        }
    }
    fclose(f);
    for (const MeshTri& t : tris)
        for (int k = 0; k < 3; ++k)
            if (t.v[k] < 0 || (size_t)t.v[k] >= verts.size() || t.t[k] < 0 || (size_t)t.t[k] >= uvs.size() ||
                t.n[k] < 0 || (size_t)t.n[k] >= normals.size())
                ok = false;
    if (!ok) { err = std::string("OBJ index out of range in ") + path; return false; }
    generate_triangle_data();
    return true;
}

void MeshData::generate_triangle_data() {
    parallel_for(tris.size(), [&](size_t b0, size_t e0) {
    for (size_t i = b0; i < e0; ++i) {
        MeshTri& t = tris[i];
        const V3& a = verts[t.v[0]]; const V3& b = verts[t.v[1]]; const V3& c = verts[t.v[2]];
        const V3 ab = b - a, ac = c - a;
        t.gnormal = normalized(cross(ab, ac));
        const V3& ta = uvs[t.t[0]]; const V3& tb = uvs[t.t[1]]; const V3& tc = uvs[t.t[2]];
        const V3 tab = tb - ta, tac = tc - ta;
        double px, py, qx, qy;
        solve2d(tab, tac, V3(1, 0, 0), px, qx);
        solve2d(tab, tac, V3(0, 1, 0), py, qy);
        t.dNdx = normalized(px * ab + qx * ac);
        t.dNdy = normalized(py * ab + qy * ac);
    }
    });
}

void MeshData::compute_bounds() {
    bmin = V3(kInf, kInf, kInf);
    bmax = V3(-kInf, -kInf, -kInf);
    for (const V3& v : verts) {      // includes the dummy vertex 0
        bmin.x = std::min(bmin.x, v.x); bmin.y = std::min(bmin.y, v.y); bmin.z = std::min(bmin.z, v.z);
        bmax.x = std::max(bmax.x, v.x); bmax.y = std::max(bmax.y, v.y); bmax.z = std::max(bmax.z, v.z);
    }
}

void MeshData::build_kd() {
    kd.clear(); leaf_tris.clear();
    kd_max_depth = kd_leaves = 0; kd_depth_sum = kd_leaf_tri_sum = 0;
    if (!wants_kd()) return;
    Builder b(*this);
    std::vector<uint32_t> all(tris.size());
    for (size_t i = 0; i < all.size(); ++i) all[i] = (uint32_t)i;
    kd.push_back(qd_kdnode{0, 0, 0});
    b.build(0, Box{bmin, bmax}, all, 0);
}

}  // namespace qdh
