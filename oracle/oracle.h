/*
 * oracle.h -- C interface of the CPU oracle (oracle/liboracle.so).
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "qdtrace.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_stats {
    uint64_t raytrace_calls;    /* every Raytrace() entry, src/main.cpp:18                       */
    uint64_t depth_killed;      /* ... of which cut by depth > maxTraceDepth, src/main.cpp:20     */
    uint64_t shadow_queries;    /* GetShadowTransparency calls, src/shadinghelper.cpp:20          */
    uint64_t mesh_queries;      /* nbIntersections, src/mesh.cpp:38                               */
    uint64_t bbox_tests;        /* nbBBoxIntersections, src/mesh.cpp:81                           */
    uint64_t tri_tests;         /* nbTriIntersections, src/mesh.cpp:116                           */
    uint64_t kd_nodes_visited;  /* recursion entries of Mesh::Intersect(KDTreeNode*, ...)         */
} orc_stats;

/* rng_mode 0: libc rand() in call order (reference-faithful, single-threaded);
 * rng_mode 1: path-hashed stream (what the device path uses for glossy reflection). */
int orc_render_rect(const qd_scene_desc* S, const qd_camera* cam, const qd_settings* st,
                    int x0, int y0, int x1, int y1, float* rgb, size_t row_pitch_floats,
                    int32_t* hit_node, int32_t* hit_tri, double* hit_dist,
                    int rng_mode, int threads, orc_stats* stats);

int orc_render_pixels(const qd_scene_desc* S, const qd_camera* cam, const qd_settings* st,
                      size_t n, const int32_t* xy, float* rgb,
                      int32_t* hit_node, int32_t* hit_tri, double* hit_dist,
                      int rng_mode, int threads, orc_stats* stats);

/* SimpleRender + AARender in the reference's bucket order, one thread; rgb is W*H*3 floats. */
int orc_render_frame(const qd_scene_desc* S, float* rgb, int rng_mode, orc_stats* stats);

unsigned orc_convert_to_8bit(double x);

/* Reference-pinned vectors (tests/golden/ref/, oracle/ref_driver.cpp): camera rays of a rectangle (6 f64 per pixel),
 * Raytrace's node loop + bump + Shade for explicit rays (96-byte records), function-level BBox / vector helpers. */
int orc_camera_rays(const qd_scene_desc* S, int x0, int y0, int x1, int y1, double* out);
int orc_trace_rays(const qd_scene_desc* S, size_t n, const double* rays, void* out_records);
int orc_unit_funcs(size_t n, const double* in, double* out);

/* rng_mode 1's stream, for the statistical tests of the glossy sampler (tests/test_glossy_statistics.py). */
void orc_random01_bulk(size_t n, const uint32_t* path, const uint32_t* sample, const uint32_t* draw, double* out);
void orc_pixel_paths(size_t n, const int32_t* xy, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
