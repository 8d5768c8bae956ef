// Host mesh preparation: OBJ reader, per-triangle data, bounding box and the
// KD-tree builder (midpoint and the reference's "SAH" variant), producing the flat
// arrays of qd_scene_desc.  Follows src/mesh.cpp:99-104 (bbox), :198-432 (KD build),
// :434-540 (OBJ + triangle data).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "qdtrace.h"
#include "vecmath.h"

namespace qdh {

struct MeshTri { int v[3]; int n[3]; int t[3]; V3 gnormal, dNdx, dNdy; };

struct MeshData {
    std::vector<V3> verts, normals, uvs;   // index 0 is the dummy (0,0,0) entry, OBJ indices are 1-based
    std::vector<MeshTri> tris;
    bool faceted = true, backcull = true, use_kd = true, use_sah = false;
    V3 bmin, bmax;

    // KD result (mesh-local)
    std::vector<qd_kdnode> kd;         // empty when no tree is built
    std::vector<uint32_t> leaf_tris;
    uint32_t kd_max_depth = 0, kd_leaves = 0;
    uint64_t kd_depth_sum = 0, kd_leaf_tri_sum = 0;

    bool load_obj(const char* path, std::string& err);
    void generate_triangle_data();      // Mesh::GenerateTrianglesData
    void begin_render();                // bbox + KD (Mesh::BeginRender)
};

}  // namespace qdh
