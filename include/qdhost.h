/*
 * qdhost.h -- C-ABI of the host front end (libqdhost.so, plain C++17, no GPU).
 *
 * This is the part of the reference that "stays on the host" (SURVEY.md rows
 * M5, C1, F2 and the asset loaders): it reads a .qdmg scene and its assets and
 * produces the flat qd_scene_desc that qd_scene_create() uploads.
 *   .qdmg grammar        : src/scene.cpp:391-612 (DefaultSceneParser::Parse), class table :871-918
 *   OBJ + triangle data  : src/mesh.cpp:434-540
 *   mesh bbox + KD build : src/mesh.cpp:99-104, :198-432
 *   BMP reader / writer  : src/bitmap.cpp:101-220
 *   EXR (scan-line PIZ)  : src/bitmap.cpp:222-252 (the reference calls OpenEXR 2.0.1; this is a from-scratch reader)
 *   heightfield prep     : src/heightfield.cpp:102-214, :260-290
 *   bump differentiate   : src/bitmap.cpp:312-326
 *   camera corners       : src/camera.cpp:6-24
 */
#ifndef QDHOST_H
#define QDHOST_H

#include "qdtrace.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qdh_scene qdh_scene;  /* owns every array referenced by its qd_scene_desc */

/* Replaces Scene::ParseScene + Scene::BeginRender + Camera::BeginFrame
 * (src/scene.cpp:803-821, src/camera.cpp:6-24). Returns NULL on error (see qdh_last_error),
 * like the reference's `false` + message on stderr (src/scene.cpp:579-592). */
qdh_scene* qdh_load_scene(const char* qdmg_path);

/* Synthetic benchmark scene (BASELINE.json configs[4], SURVEY.md 8(d) input 5):
 * displaced-grid glass mesh of 2*n*n triangles over a checker floor, one light,
 * optional cubemap folder (NULL = no environment). */
qdh_scene* qdh_synth_scene(int n, unsigned seed, int width, int height, int max_trace_depth,
                           const char* env_folder);

/* The flat description (valid until qdh_free_scene). Mutable on purpose: callers may
 * override settings (frame size, AA flags, depth) before rendering. */
qd_scene_desc* qdh_scene_desc(qdh_scene* s);

/* The reference-shaped KD trees of the scene's meshes (Mesh::BuildKD, src/mesh.cpp:198-432: midpoint or "SAH" planes, 20
 * triangles per leaf) are built ON REQUEST: libqdtrace builds the device tree from the triangles (qd_set_kd_build 1), so
 * a load that only feeds the GPU does not pay for a tree nobody walks (2 s for 10 M triangles); such meshes carry
 * QD_MESH_WANT_KD and kd_root -1 until then.  qdh_scene_build_kd builds them (once) and re-points the description's
 * kd_nodes / leaf_tris / mesh statistics: what the CPU oracle walks, what qd_set_kd_build(0) refines, what the
 * reference prints at src/mesh.cpp:211-213.  qdh_set_eager_kd(1): scenes loaded afterwards build them inside the load,
 * as Scene::BeginRender does (src/scene.cpp:809-821). */
int  qdh_scene_build_kd(qdh_scene* s);
void qdh_set_eager_kd(int on);

/* Recompute camera corners after changing camera parameters (Camera::BeginFrame). */
void qdh_camera_corners(const double position[3], double yaw, double pitch, double roll,
                        double fov, double aspect, qd_camera* out);

/* Serpentine 48x48 bucket list (GetBucketList, src/sdl.cpp:253-276): writes up to cap rects
 * as x0,y0,x1,y1 and returns the number of buckets. */
int qdh_bucket_list(int width, int height, int32_t* rects, int cap);

/* Bitmap::SaveBMP (src/bitmap.cpp:181-220) on a float RGB image, incl. ConvertTo8Bit quantisation. */
int qdh_save_bmp(const char* path, const float* rgb, int width, int height, size_t row_pitch_floats);

/* Bitmap::SaveEXR (src/bitmap.cpp:254-277): the float RGB image as RGBA half, alpha 1.  Written as an uncompressed
 * scan-line OpenEXR file (the reference's Imf::RgbaOutputFile compresses with PIZ; the decoded pixels are the same). */
int qdh_save_exr(const char* path, const float* rgb, int width, int height, size_t row_pitch_floats);

/* Bitmap::LoadImage (BMP / EXR) exposed for tests: returns malloc'd float RGB (free with qdh_free). */
float* qdh_load_image(const char* path, uint32_t* width, uint32_t* height);
void   qdh_free(void* p);

/* Test hook: the front end's matrix / Transform arithmetic (src/matrix.cpp:10-133, src/transform.cpp:5-73) on one
 * 16-double record, 52 doubles out; checked bit for bit against the reference's own object code (tests/golden/ref/). */
void qdh_math_probe(const double in[16], double out[52]);

/* Scene preparation on the device.  The table passes of BeginRender -- heightfield blur / max-heights /
 * normals / high map (src/heightfield.cpp:102-214, 260-290) and Bitmap::Differentiate (src/bitmap.cpp:312-326) -- run on
 * the CPU by default.  A caller that has the device library hands its three entry points over (they have exactly these
 * signatures: qd_prep_heightfield, qd_prep_highmap, qd_prep_differentiate in qdtrace.h) and scenes loaded afterwards
 * get their tables from HIP kernels; the tables are bit-identical either way.  A hook that returns non-zero fails the
 * load.  NULL (or a NULL member) = CPU code.  Process-wide; set it before loading scenes. */
typedef struct qdh_prep_hooks {
    int (*heightfield)(uint32_t width, uint32_t height, const float* rgb, const float* gauss, int r,
                       float* heights, float* maxheights, double* normals);
    int (*highmap)(uint32_t width, uint32_t height, const float* heights, int max_k, float* highmap);
    int (*differentiate)(uint32_t width, uint32_t height, float* rgb);
} qdh_prep_hooks;
void qdh_set_prep_hooks(const qdh_prep_hooks* hooks);
int qdh_prep_hook_calls(void);      /* table passes that went through a hook so far (for tests) */

void qdh_free_scene(qdh_scene* s);
const char* qdh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* QDHOST_H */
