// Owner of every array a qd_scene_desc points at.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "heightfield.h"
#include "mesh.h"
#include "qdtrace.h"

namespace qdh {

struct SceneImpl {
    qd_scene_desc desc;
    std::vector<qd_node> nodes;
    std::vector<qd_geom> geoms;
    std::vector<qd_mesh> meshes;
    std::vector<qd_heightfield> hfields;
    std::vector<qd_shader> shaders;
    std::vector<qd_layer> layers;
    std::vector<qd_texture> textures;
    std::vector<qd_proc_line> proc_lines;
    std::vector<qd_bitmap> bitmaps;
    std::vector<qd_light> lights;
    raw_vector<double> tri_verts, tri_gnormal, tri_shade;       // filled on all cores by finalize()
    std::vector<qd_kdnode> kd_nodes;
    std::vector<uint32_t> leaf_tris;
    std::vector<float> texels;
    std::vector<float> hf_f32;
    std::vector<double> hf_f64;

    std::vector<std::unique_ptr<MeshData>> mesh_src;       // geometry index a -> mesh
    std::vector<std::unique_ptr<HeightfieldData>> hf_src;
    std::string warnings;

    SceneImpl();
    bool load(const char* path, std::string& err);
    bool load_env(const std::string& folder, bool bilinear, std::string& err);
    bool kd_built = false;
    void finalize();   // BeginRender: bbox + highmap (+ the KD trees when g_eager_kd), then publish()
    void build_kd();   // the reference-shaped KD trees of the meshes (Mesh::BuildKD), on request: the device builds its own tree
                       // from the triangles, so only the CPU oracle and qd_set_kd_build(0) read these
    void publish();    // point desc at the vectors
    void build_kd_arrays();
};

extern bool g_eager_kd;     // qdh_set_eager_kd: build the KD trees inside qdh_load_scene, as Scene::BeginRender does

}  // namespace qdh
