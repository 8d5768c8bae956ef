// CPU check of the device KD construction (raytracing_amd/csrc/device/kdrefine.h): builds the records for a scene and
// walks them on the host with the same rules as the kernel's mesh_query, against a brute-force closest hit.
// usage: kdrefine_check <scene.qdmg | synth:N> <rays> <max_leaf> <step_cost> [rebuild 0|1] [brute 1|0] [cam 0|1]
//        prints "rays R mismatches M ..."; synth:N = the synthetic bench mesh with 2*N*N triangles; brute 0 skips the brute-force
//        check (tree statistics only); cam 1 shoots the rays from the bench camera's position instead of from around the box
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "kdrefine.h"
#include "qdhost.h"

struct V { double x, y, z; };
static V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Mesh::Intersect(ray, triangle), src/mesh.cpp:111-139, without back-face culling
static bool tri_test(const double* t, V o, V d, double& best) {
    const V A{t[0], t[1], t[2]}, AB{t[3], t[4], t[5]}, AC{t[6], t[7], t[8]}, md{-d.x, -d.y, -d.z};
    const V N = cross(AB, AC);
    const double dcr = dot(N, md);
    if (std::fabs(dcr) < 1e-12) return false;
    const V H = sub(o, A);
    const double gamma = dot(N, H) / dcr;
    if (gamma < 0 || gamma > best) return false;
    const double l2 = dot(cross(H, AC), md) / dcr, l3 = dot(cross(AB, H), md) / dcr;
    if (l2 < 0 || l3 < 0 || l2 + l3 > 1) return false;
    best = gamma;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    qdh_scene* hs = std::strncmp(argv[1], "synth:", 6) == 0 ? qdh_synth_scene(std::atoi(argv[1] + 6), 42, 640, 360, 8, nullptr) : qdh_load_scene(argv[1]);
    const bool rebuild = argc > 5 && std::atoi(argv[5]) != 0, do_brute = !(argc > 6 && std::atoi(argv[6]) == 0), cam = argc > 7 && std::atoi(argv[7]) != 0;
    if (!hs) { std::printf("load failed: %s\n", qdh_last_error()); return 2; }
    if (qdh_scene_build_kd(hs) != 0) { std::printf("kd build failed: %s\n", qdh_last_error()); return 2; }     // the caller's tree (refine() wants a root even when it rebuilds)
    const qd_scene_desc* d = qdh_scene_desc(hs);
    qd_refine::Options opt; opt.max_leaf = std::atoi(argv[3]); opt.cost_step = std::atof(argv[4]); opt.rebuild = rebuild;
    qd_refine::Result R;
    const std::string err = qd_refine::refine(d, opt, R);
    if (!err.empty()) { std::printf("refine failed: %s\n", err.c_str()); return 2; }
    const int n_rays = std::atoi(argv[2]);
    long mismatches = 0, rays = 0, hits = 0;
    unsigned long long visited = 0, tested = 0;
    for (uint32_t mi = 0; mi < d->n_meshes; ++mi) {
        const qd_mesh& m = d->meshes[mi];
        if (m.kd_root < 0) continue;
        std::vector<qd_refine::Record> recs;
        const uint32_t root = qd_refine::pack(R.trees[mi], recs, 0);
        const std::vector<uint32_t>& refs = R.trees[mi].refs;
        // every triangle must be referenced at least once
        std::vector<char> seen(m.tri_count, 0);
        for (uint32_t r : refs) { if (r >= m.tri_count) { std::printf("reference out of range\n"); return 1; } seen[r] = 1; }
        for (uint32_t t = 0; t < m.tri_count; ++t) if (!seen[t]) { std::printf("triangle %u lost\n", t); return 1; }
        std::mt19937_64 rng(1234 + mi);
        std::uniform_real_distribution<double> U(0., 1.);
        const double* tv = d->tri_verts + (size_t)m.tri_first * 9;
        for (int r = 0; r < n_rays; ++r, ++rays) {
            V lo{m.bbox_min[0], m.bbox_min[1], m.bbox_min[2]}, hi{m.bbox_max[0], m.bbox_max[1], m.bbox_max[2]};
            const V ext = sub(hi, lo);
            V o{lo.x + ext.x * (U(rng) * 3 - 1), lo.y + ext.y * (U(rng) * 3 - 1), lo.z + ext.z * (U(rng) * 3 - 1)};
            if (r % 4 == 0) {   // some origins exactly on a vertex coordinate (split planes pass through those)
                const double* t = tv + (size_t)(rng() % m.tri_count) * 9;
                o.x = t[0];
            }
            if (cam) o = V{0.073, 0.7, -1.4};             // the bench camera in the mesh's object space (node scale 100)
            V p{lo.x + ext.x * U(rng), lo.y + ext.y * U(rng), lo.z + ext.z * U(rng)};
            V dd = sub(p, o);
            const double len = std::sqrt(dot(dd, dd));
            const V dir{dd.x * (1. / len), dd.y * (1. / len), dd.z * (1. / len)};
            if (r % 7 == 0) { const_cast<double&>(dir.y) = 0.; }
            double brute = 1e99;
            if (do_brute) for (uint32_t t = 0; t < m.tri_count; ++t) tri_test(tv + (size_t)t * 9, o, dir, brute);
            // ---- the kernel's walk (dgeom.h mesh_query) on the host
            double t0 = 0., t1 = 1e99, best = 1e99;
            const double oc[3] = {o.x, o.y, o.z}, dc[3] = {dir.x, dir.y, dir.z};
            double inv[3];
            bool miss = false;
            for (int a = 0; a < 3; ++a) {
                inv[a] = 1. / dc[a];
                const double bl = m.bbox_min[a] - 1e-6, bh = m.bbox_max[a] + 1e-6;
                if (dc[a] == 0.) { if (oc[a] < bl || oc[a] > bh) miss = true; continue; }
                double ta = (bl - oc[a]) * inv[a], tb = (bh - oc[a]) * inv[a];
                if (ta > tb) std::swap(ta, tb);
                t0 = std::fmax(t0, ta); t1 = std::fmin(t1, tb);
            }
            if (!miss && t0 <= t1) {
                struct E { uint32_t node; double t; };
                std::vector<E> stack;
                uint32_t node = root;
                double tmin = t0, tmax = t1;
                bool found = false;
                auto choose = [&](double split, uint32_t axis, uint32_t tok_left, uint32_t tok_right) -> uint32_t {
                    ++visited;
                    const double oa = oc[axis], t = (split - oa) * inv[axis];
                    const uint32_t near_s = (oa > split || (oa == split && dc[axis] > 0)) ? 1u : 0u;
                    const double slack = 1e-7 + 1e-9 * std::fabs(t);
                    if (t > tmax + slack || t <= 0) return near_s;
                    if (t < tmin - slack) return near_s ^ 1u;
                    stack.push_back(E{near_s ? tok_left : tok_right, tmax});
                    tmax = (t == t) ? std::fmin(tmax, t + slack) : tmax;
                    return near_s;
                };
                for (;;) {
                    const uint32_t rec = node & 0x3fffffffu, sub_ = node >> 30;
                    const qd_refine::Record& rc = recs[rec];
                    const uint32_t meta = rc.w[6];
                    uint32_t first = 0, count = 0;
                    bool leaf = false;
                    auto f64 = [&](int at) { double v; std::memcpy(&v, &rc.w[at], 8); return v; };
                    if ((meta & 3u) == 3u) { leaf = true; first = rc.w[0]; count = rc.w[1]; ++visited; }
                    else {
                        const uint32_t c = sub_ ? sub_ - 1u : choose(f64(0), meta & 3u, rec | (1u << 30), rec | (2u << 30));
                        const uint32_t ac = (meta >> (2u + 2u * c)) & 3u;
                        if (ac == 3u) { leaf = true; first = rc.w[2 + 2 * c]; count = rc.w[3 + 2 * c]; ++visited; }
                        else { const uint32_t cb = ((meta >> 6) << 2) + 2u * c; node = cb + choose(f64(2 + 2 * (int)c), ac, cb, cb + 1u); }
                    }
                    if (!leaf) continue;
                    for (uint32_t i = 0; i < count; ++i) { ++tested; if (tri_test(tv + (size_t)refs[first + i] * 9, o, dir, best)) found = true; }
                    if (found && best <= tmax - (2e-7 + 2e-9 * std::fabs(tmax))) break;
                    if (stack.empty()) break;
                    node = stack.back().node; tmin = tmax; tmax = stack.back().t; stack.pop_back();
                }
            }
            if (brute < 1e98) ++hits;
            if (do_brute && best != brute) { if (++mismatches <= 5) std::printf("ray %d: walk %.17g brute %.17g\n", r, best, brute); }
        }
    }
    std::printf("rays %ld hits %ld mismatches %ld nodes/ray %.1f tris/ray %.1f tree nodes %llu refs %llu max_leaf %llu\n", rays, hits, mismatches,
                rays ? (double)visited / rays : 0., rays ? (double)tested / rays : 0., (unsigned long long)R.n_nodes, (unsigned long long)R.n_refs,
                (unsigned long long)R.max_leaf);
    qdh_free_scene(hs);
    return mismatches ? 1 : 0;
}
