// Host mesh preparation: OBJ reader, per-triangle data, bounding box and the
// KD-tree builder (midpoint and the reference's "SAH" variant), producing the flat
// arrays of qd_scene_desc.  Follows src/mesh.cpp:99-104 (bbox), :198-432 (KD build),
// :434-540 (OBJ + triangle data).
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "qdtrace.h"
#include "vecmath.h"

namespace qdh {

// fn(begin, end) over [0, n) in contiguous slices on the host cores (at most 16); small ranges run in place
template <class Fn> void parallel_for(size_t n, Fn&& fn, size_t serial_below = 200000) {
    const unsigned hw = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    if (n < serial_below || hw == 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < hw; ++t) pool.emplace_back([&, t]() { fn(n * t / hw, n * (t + 1) / hw); });
    for (auto& th : pool) th.join();
}

// std::vector whose resize() leaves trivially constructible elements untouched: the arrays of a 10 M triangle mesh are
// gigabytes, and a zero fill on one thread (every page touched there first) costs more than computing their contents
// on all cores.
template <class T> struct default_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_alloc<U>; };
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using raw_vector = std::vector<T, default_init_alloc<T>>;

struct P3 {                      // three doubles without a constructor (V3 zero-fills)
    double x, y, z;
    P3& operator=(const V3& v) { x = v.x; y = v.y; z = v.z; return *this; }
};
struct MeshTri { int v[3]; int n[3]; int t[3]; P3 gnormal, dNdx, dNdy; };

struct MeshData {
    std::vector<V3> verts, normals, uvs;   // index 0 is the dummy (0,0,0) entry, OBJ indices are 1-based
    raw_vector<MeshTri> tris;
    bool faceted = true, backcull = true, use_kd = true, use_sah = false;
    V3 bmin, bmax;

    // KD result (mesh-local)
    std::vector<qd_kdnode> kd;         // empty when no tree is built
    std::vector<uint32_t> leaf_tris;
    uint32_t kd_max_depth = 0, kd_leaves = 0;
    uint64_t kd_depth_sum = 0, kd_leaf_tri_sum = 0;

    bool load_obj(const char* path, std::string& err);
    void generate_triangle_data();      // Mesh::GenerateTrianglesData
    bool wants_kd() const { return use_kd && tris.size() >= 50; }     // src/mesh.cpp:200
    void compute_bounds();              // Mesh::ComputeBoundingGeometry, src/mesh.cpp:99-104
    void build_kd();                    // Mesh::ComputeKDRoot / BuildKD, src/mesh.cpp:198-432 (needs compute_bounds)
    void begin_render() { compute_bounds(); build_kd(); }      // Mesh::BeginRender, src/mesh.cpp:185-190
};

}  // namespace qdh
