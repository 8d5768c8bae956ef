/*
 * qdtrace.h -- C-ABI boundary of the MI355X trace path ("Quad Damage" per-pixel
 * trace path: ray generation -> closest hit (KD-tree mesh / analytic / CSG /
 * heightfield) -> shading loop with shadow + secondary rays -> float RGB).
 *
 * The reference has no FFI layer; the seam this header replaces is the pixel
 * loop of SimpleRender():
 *     vfb[y][x] = Raytrace(scene.camera->GetScreenRay(x, y));
 *     (reference: src/main.cpp:85-104, Raytrace src/main.cpp:18-56,
 *      Camera::GetScreenRay src/camera.cpp:26-41)
 * Inputs are the immutable scene *after* Scene::BeginRender() (src/scene.cpp:809-821)
 * and the camera *after* Camera::BeginFrame() (src/camera.cpp:6-24), flattened
 * into the plain arrays of qd_scene_desc; the output is the caller-owned
 * float RGB framebuffer (reference: `Color vfb[1920][1920]`, src/main.cpp:15).
 *
 * Everything here is plain C: pointers, sizes, PODs.  No torch / HIP types.
 * All multi-byte values are little-endian host order; all geometry is FP64,
 * all colours FP32 (reference: src/vector.h:7-16, src/color.h:29-33).
 */
#ifndef QDTRACE_H
#define QDTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------
 * Scene description (host memory, caller-owned; qd_scene_create copies it).
 * -------------------------------------------------------------------------- */

/* Geometry kinds (reference: class table src/scene.cpp:875-885). */
enum {
    QD_GEOM_PLANE       = 0, /* src/geometry.cpp:6-28    p = {y, limit}               */
    QD_GEOM_SPHERE      = 1, /* src/geometry.cpp:49-87   p = {cx, cy, cz, radius}     */
    QD_GEOM_CUBE        = 2, /* src/geometry.cpp:107-162 p = {cx, cy, cz, halfSide}   */
    QD_GEOM_CSG         = 3, /* src/geometry.cpp:184-242 a = left geom, b = right geom, c = op */
    QD_GEOM_REGPOLY     = 4, /* src/geometry.cpp:259-296 p = {cx, cy, cz, radius}, a = sides */
    QD_GEOM_MESH        = 5, /* src/mesh.cpp:30-168      a = mesh index               */
    QD_GEOM_HEIGHTFIELD = 6  /* src/heightfield.cpp:20-77 a = heightfield index       */
};
enum { QD_CSG_AND = 0, QD_CSG_PLUS = 1, QD_CSG_MINUS = 2 }; /* src/geometry.h:124-140 */

typedef struct qd_geom {
    int32_t type;
    int32_t a, b, c;
    double  p[4];
} qd_geom;

/* One scene node: geometry + shader + transform (src/geometry.h:142-156,
 * src/transform.h:29-33).  Matrices are row-major 3x3, applied as row-vector * M
 * (src/matrix.cpp:10-17). */
typedef struct qd_node {
    int32_t geom;                 /* index into geoms[]                        */
    int32_t shader;               /* index into shaders[]                      */
    int32_t bump_tex;             /* index into textures[] or -1               */
    float   shadow_transparency;  /* src/geometry.h:148                        */
    double  M[9];                 /* m_Transform                               */
    double  Minv[9];              /* m_InverseTransform                        */
    double  MinvT[9];             /* m_TransposedInverse                       */
    double  offset[3];            /* m_Offset                                  */
} qd_node;

/* KD-tree node, 16 bytes (reference KDTreeNode is 24 B, src/KDTree.h:8-18).
 * meta = (index << 2) | axis; axis 0..2 = inner node (index = left child,
 * right child = index + 1, mirroring `new KDTreeNode[2]`, src/KDTree.cpp:23);
 * axis 3 = leaf (index = first entry in leaf_tris[], count = entries). */
typedef struct qd_kdnode {
    double   split;
    uint32_t meta;
    uint32_t count;
} qd_kdnode;

enum { QD_MESH_FACETED = 1, QD_MESH_BACKCULL = 2,
       QD_MESH_WANT_KD = 4   /* the mesh is to be walked through a KD tree (useKDTree and >= 50 triangles, src/mesh.cpp:200) although
                              * the description carries none (kd_root < 0): qd_scene_create builds the device tree from the triangles */ };

typedef struct qd_mesh {
    uint32_t tri_first;   /* first triangle of this mesh in the global triangle arrays  */
    uint32_t tri_count;
    int32_t  kd_root;     /* index into kd_nodes[] or -1 (no tree: < 50 tris or useKDTree false, src/mesh.cpp:200) */
    uint32_t flags;       /* QD_MESH_*  (src/mesh.h:48-49)                               */
    double   bbox_min[3]; /* includes the dummy vertex (0,0,0), src/mesh.cpp:516,99-104  */
    double   bbox_max[3];
    /* KD build statistics, the values the reference prints at src/mesh.cpp:211-213 */
    uint32_t kd_max_depth;
    uint32_t kd_leaves;
    uint64_t kd_depth_sum;
    uint64_t kd_leaf_tris;
} qd_mesh;

/* Heightfield (src/heightfield.h:19-45).  Arrays are row-major [z * width + x]. */
typedef struct qd_heightfield {
    uint32_t width, height;
    int32_t  use_optimization;
    int32_t  max_k;                /* src/heightfield.cpp:262 */
    double   bbox_min[3], bbox_max[3];
    uint64_t heights_off;          /* offsets (in elements) into hf_f32[] */
    uint64_t maxheights_off;
    uint64_t highmap_off;          /* 16 floats per cell; valid iff use_optimization */
    uint64_t normals_off;          /* offset (in doubles) into hf_f64[], 3 per cell   */
} qd_heightfield;

/* Shaders (src/shading.h, class table src/scene.cpp:888-895). */
enum {
    QD_SH_CONST = 0, QD_SH_LAMBERT = 1, QD_SH_PHONG = 2, QD_SH_BLINNPHONG = 3,
    QD_SH_ORENNAYAR = 4, QD_SH_REFL = 5, QD_SH_REFR = 6, QD_SH_LAYERED = 7
};
typedef struct qd_shader {
    int32_t type;
    int32_t tex;        /* diffuse texture index or -1                              */
    float   color[3];
    int32_t i0;         /* REFL: numSamples; LAYERED: first layer in layers[]       */
    int32_t i1;         /* LAYERED: number of layers                                */
    int32_t pad_;
    double  p0;         /* PHONG: specularMultiplier; ORENNAYAR: sigma; REFL: multiplier; REFR: ior */
    double  p1;         /* PHONG: specularExponent;  REFL: glossiness; REFR: multiplier             */
} qd_shader;

typedef struct qd_layer {  /* src/shading.h:131-136 */
    int32_t shader;
    int32_t tex;           /* blend texture or -1 */
    float   blend[3];
} qd_layer;

/* Textures (src/texture.h, class table src/scene.cpp:898-906). */
enum {
    QD_TEX_CONST = 0, QD_TEX_CHECKER = 1, QD_TEX_MANDELBROT = 2, QD_TEX_PROCEDURAL = 3,
    QD_TEX_BITMAP = 4, QD_TEX_FRESNEL = 5, QD_TEX_BUMPMAP = 6, QD_TEX_BUMPS = 7
};
typedef struct qd_texture {
    int32_t type;
    int32_t bitmap;      /* BITMAP/BUMPMAP: index into bitmaps[]                      */
    float   color1[3];
    float   color2[3];
    int32_t bilinear;
    int32_t i0, i1;      /* PROCEDURAL: first line / line count in proc_lines[]       */
    int32_t pad_;
    double  scaling;     /* CHECKER/MANDELBROT: scaling; BITMAP/BUMPMAP: 1/scaling (stored inverted, src/texture.cpp:203-204) */
    double  p0;          /* FRESNEL: ior; BUMPMAP: strength; BUMPS: strength (float value) */
} qd_texture;

typedef struct qd_proc_line { /* src/texture.h:69-76 */
    float  color_u[3]; float color_v[3];
    double freq_u, freq_v;
} qd_proc_line;

typedef struct qd_bitmap {
    uint32_t width, height;
    uint64_t texel_off;  /* offset in float3 texels into texels[] (row-major x + y*w, src/bitmap.cpp:49) */
} qd_bitmap;

typedef struct qd_light { double pos[3]; double intensity; } qd_light; /* src/light.h:6-13 */

typedef struct qd_camera {      /* Camera after BeginFrame, src/camera.cpp:6-24 */
    double top_left[3], top_right[3], bottom_left[3], position[3];
} qd_camera;

typedef struct qd_settings {    /* src/scene.h:228-252 */
    int32_t  frame_width, frame_height;
    float    ambient[3];
    uint32_t max_trace_depth;
    int32_t  want_aa, want_adaptive_aa, show_aa, use_srgb;
    float    aa_debug_color[3];
    int32_t  pad_;
    double   aa_threshold;
} qd_settings;

typedef struct qd_scene_desc {
    qd_settings settings;
    qd_camera   camera;

    uint32_t n_nodes;      const qd_node*        nodes;     /* file order, shader-less nodes removed (src/scene.cpp:603-610) */
    uint32_t n_geoms;      const qd_geom*        geoms;
    uint32_t n_meshes;     const qd_mesh*        meshes;
    uint32_t n_hfields;    const qd_heightfield* hfields;
    uint32_t n_shaders;    const qd_shader*      shaders;
    uint32_t n_layers;     const qd_layer*       layers;
    uint32_t n_textures;   const qd_texture*     textures;
    uint32_t n_proc_lines; const qd_proc_line*   proc_lines;
    uint32_t n_bitmaps;    const qd_bitmap*      bitmaps;
    uint32_t n_lights;     const qd_light*       lights;

    /* Triangles of all meshes, concatenated (index = mesh.tri_first + local index).
     * tri_verts: 9 doubles per triangle: A, AB = B - A, AC = C - A (the operands
     *            the reference forms per test at src/mesh.cpp:118-125).
     * tri_gnormal: 3 doubles (MeshTriangle::geometryNormal, src/mesh.cpp:477-478).
     * tri_shade: 21 doubles: nA,nB,nC (9), uvA,uvB,uvC (x,y each: 6), dNdx, dNdy (6). */
    uint64_t n_tris;
    const double* tri_verts;
    const double* tri_gnormal;
    const double* tri_shade;

    uint64_t n_kd_nodes;   const qd_kdnode* kd_nodes;
    uint64_t n_leaf_tris;  const uint32_t*  leaf_tris;  /* mesh-local triangle indices */

    uint64_t n_texels;     const float*  texels;        /* 3 floats per texel */
    uint64_t n_hf_f32;     const float*  hf_f32;
    uint64_t n_hf_f64;     const double* hf_f64;

    int32_t env_present;     /* CubemapEnvironment (src/environment.h:30-50)            */
    int32_t env_bilinear;
    int32_t env_faces[6];    /* bitmap index per face, order negx,negy,negz,posx,posy,posz */
} qd_scene_desc;

/* ----------------------------------------------------------------------------
 * Device library entry points (libqdtrace.so, HIP / gfx950).
 * -------------------------------------------------------------------------- */

typedef struct qd_scene qd_scene;   /* opaque: scene buffers resident in HBM of the current device */

typedef struct qd_stats {
    uint64_t rays;            /* traced Raytrace calls (depth test passed, src/main.cpp:20) + shadow queries (src/shadinghelper.cpp:20) */
    uint64_t primary_rays;
    uint64_t shadow_rays;
    uint64_t depth_killed;    /* Raytrace calls cut by ray.depth > maxTraceDepth       */
    uint64_t kd_nodes_visited;/* KD nodes touched (inner + leaf); plus, for heightfields, DDA steps (src/heightfield.cpp:33) */
    uint64_t tris_tested;     /* triangle tests after back-face culling (nbTriIntersections, src/mesh.cpp:116);
                               * plus, for heightfields, cells whose two triangles were tested (src/heightfield.cpp:50-58) */
    uint64_t mesh_queries;    /* ray/mesh queries that pass the mesh bbox (nbIntersections, src/mesh.cpp:38)     */
    double   kernel_ms;       /* device time of the trace kernel(s), HIP events         */
} qd_stats;

/* Threads.  qd_last_error() is per thread.  Every call that takes a qd_scene* locks the handle for its duration, so calls
 * on one handle from several threads are safe and serialise on the host (the asynchronous ones only enqueue, the GPU work
 * of different streams still overlaps); different handles do not contend.  What a handle holds ONE of -- the AA plan
 * (qd_aa_begin_device .. qd_aa_end_device), the frame of qd_frame_begin, the staging of the synchronous calls -- belongs
 * to one logical caller at a time.  Each working set (stream) has its own device error word: a synchronous call clears
 * and reads only its own, qd_scene_check reads them all.  qd_init and the qd_set_* switches are process-wide and are
 * meant to be called before the threads start. */

/* Replaces: process start-up (the reference has none for a device). 0 on success, <0 on error.
 * device = HIP device ordinal to bind this process to (one process per GPU). */
int qd_init(int device);

/* Replaces: Scene::BeginRender() fan-out as far as the render path is concerned
 * (src/scene.cpp:809-821): uploads the flattened scene to HBM. The caller keeps
 * ownership of every pointer in *desc. */
int qd_scene_create(const qd_scene_desc* desc, qd_scene** out);

/* Replaces: the double loop of SimpleRender() (src/main.cpp:90-97) restricted to the
 * pixel rectangle [x0,x1) x [y0,y1) of a frame_width x frame_height frame.
 *   camera/settings : as Scene::BeginFrame() leaves them (src/scene.cpp:823-835)
 *   rgb             : caller-owned HOST buffer, pixel (x,y) at rgb[(y-y0)*row_pitch_floats + (x-x0)*3]
 *   hit_node/hit_tri/hit_dist : optional (may be NULL) primary-ray closest hit per pixel,
 *                     packed (y-y0)*(x1-x0)+(x-x0): node index (-1 = sky), mesh-local
 *                     triangle index (-1 for non-mesh), world-space distance.
 *   stats           : optional.
 * Returns 0, or <0 and sets qd_last_error().  Device staging for the rectangle, the hit planes and the timing events
 * belongs to the scene handle and is reused from call to call (grown when a larger rectangle comes). */
int qd_render(qd_scene* scene, const qd_camera* camera, const qd_settings* settings,
              int x0, int y0, int x1, int y1,
              float* rgb, size_t row_pitch_floats,
              int32_t* hit_node, int32_t* hit_tri, double* hit_dist,
              qd_stats* stats);

/* Replaces: RenderScene's bucket loop (src/main.cpp:85-104: for each 48 x 48 bucket of GetBucketList, src/sdl.cpp:253-276,
 * trace its pixels, then DisplayVFBRect(bucket)) without a launch per bucket.  qd_frame_begin enqueues the WHOLE first
 * pass of the frame on `hip_stream` (NULL = default stream) into a buffer the scene handle keeps, and returns at once.
 * qd_frame_read_rect copies the finished rectangle [x0,x1) x [y0,y1) to the caller's HOST buffer (pixel (x,y) at
 * rgb[(y-y0)*row_pitch_floats + (x-x0)*3]): the first read after qd_frame_begin waits for the frame and fetches it to
 * a pinned host copy in one transfer, every read is then a copy of rows on the host -- so the reference's loop keeps its
 * shape (read a bucket, display it) at the cost of one frame-sized launch.  qd_frame_end returns the frame's counters
 * (optional) and ends the plan; a new qd_frame_begin also ends it.  Values are those of qd_render(0,0,W,H).
 * One frame at a time per scene handle; errors of the launch (ray queue overflow) surface at the first read. */
int qd_frame_begin(qd_scene* scene, const qd_camera* camera, const qd_settings* settings, void* hip_stream);
int qd_frame_read_rect(qd_scene* scene, int x0, int y0, int x1, int y1, float* rgb, size_t row_pitch_floats);
int qd_frame_end(qd_scene* scene, qd_stats* stats);

/* Replaces: the anti-aliasing pass of RenderScene (src/main.cpp:106-153, IsTooDifferent :58-83), which
 * the reference runs after SimpleRender when settings.wantAA is set.
 *   rgb : caller-owned HOST buffer holding the finished first pass of the WHOLE frame (what qd_render wrote
 *         for 0,0,W,H), pixel (x,y) at rgb[y*row_pitch_floats + x*3]; replaced in place by the anti-aliased frame.
 * Semantics kept: 48x48 buckets in display order, a bucket's pixels are marked before any of them is
 * replaced, marked pixels become (first + 4 more rays at the fixed offsets) / 5, pixels of earlier buckets
 * are seen already replaced, neighbours outside the frame read as black and coordinates >= 1920 are skipped
 * (the static vfb), showAA paints aaDebugColor, wantAdaptiveAA false marks everything.
 * The caller decides whether to call it (settings.wantAA); the function itself does not look at want_aa. */
int qd_antialias(qd_scene* scene, const qd_camera* camera, const qd_settings* settings,
                 float* rgb, size_t row_pitch_floats, qd_stats* stats);

/* Same pass on a frame that is already in DEVICE memory (contiguous frame_width x frame_height x 3 floats, e.g.
 * the frame qd_untile_async assembled on rank 0), in place.  Synchronous: the mark / replace rule needs two small
 * read-backs.  Work queued on other streams for the same buffer must have finished. */
int qd_antialias_device(qd_scene* scene, const qd_camera* camera, const qd_settings* settings,
                        float* rgb_dev, qd_stats* stats);

/* The same pass in three steps, so that the taps -- the only expensive part -- can be traced by several GPUs
 * (the pass needs the whole first-pass frame, so it runs after the gather).  Every rank holding the
 * assembled first-pass frame calls qd_aa_begin_device (mark + list; the list depends on the frame only, so all ranks
 * get the same one; *n_listed = pixels whose four taps are to be traced, 0 with showAA or when nothing is marked),
 * then traces ITS slice of the list with qd_aa_trace_part_async (taps_dev receives n_pixels * 4 * 3 floats, tap-major
 * per pixel; enqueue only), the slices are gathered, and the rank that keeps the frame calls qd_aa_end_device with all
 * n_listed * 12 floats (combine, bucket-ordered mark / replace rule, frame updated in place).  A rank that does not keep
 * the frame ends its plan with qd_aa_end_device(scene, settings, NULL, NULL).  qd_antialias_device is exactly
 * begin + one part covering the whole list + end.  Replaces: src/main.cpp:106-153 as above. */
int qd_aa_begin_device(qd_scene* scene, const qd_settings* settings, const float* rgb_dev, uint32_t* n_listed);
int qd_aa_trace_part_async(qd_scene* scene, const qd_camera* camera, const qd_settings* settings,
                           uint32_t first_pixel, uint32_t n_pixels, float* taps_dev, qd_stats* stats_dev, void* hip_stream);
int qd_aa_end_device(qd_scene* scene, const qd_settings* settings, const float* taps_dev, float* rgb_dev);

/* Same as qd_render but every output pointer is DEVICE memory and the call only
 * enqueues work on `hip_stream` (a hipStream_t passed as void*; NULL = default
 * stream).  Pixels are selected by tile ownership instead of a rectangle:
 * the frame is cut into tile x tile squares numbered row-major; this call renders
 * the tiles t with t % tile_stride == tile_phase (multi-GPU sharding,
 * one process per GPU) and writes them *compacted*: owned tile number k (in
 * increasing t) occupies rgb_dev[k*tile*tile*3 ...], row-major inside the tile.
 * Used by bench.py / the RCCL gather path.  stats_dev: optional device qd_stats.
 * Frames in flight: a scene keeps up to 8 working sets (ray queues, task lists, control words),
 * each bound to the stream that used it last.  Calls on ONE stream share a set and run in order;
 * calls on DIFFERENT streams use different sets and may overlap on the GPU (the thin tail of one
 * frame runs beside the start of the next) -- give each such call its own rgb_dev / stats_dev.  A ninth
 * stream takes over the set used longest ago and first waits, on the device, for that set's last launch. */
int qd_render_tiles_async(qd_scene* scene, const qd_camera* camera, const qd_settings* settings,
                          int tile, int tile_phase, int tile_stride,
                          float* rgb_dev, qd_stats* stats_dev, void* hip_stream);

/* No counterpart in the reference.  For callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  Same arguments, pixel selection and output layout as qd_render_tiles_async,
 * but the call allocates nothing (no hipMalloc / hipFree, no events): the ray queues and control words live in a
 * file-scope __device__ array of 96 MiB inside libqdtrace.so (each GPU the library runs on holds its copy until the
 * process ends; qd_init looks its address up once per device).  The frame is one launch of the trace kernel on as many
 * lanes as 96 MiB hold at 88 B x (queue depth of this scene at settings->max_trace_depth) per lane.
 * Limits: fails with a clear error when that is less than one wave per CU (queue depth > ~69 on 256 CUs), never
 * falls back to allocating.  Every call on one GPU shares that GPU's array, which keeps the last call's contents (the
 * call clears the control words it needs zeroed): calls that OVERLAP on one GPU -- different streams, or another
 * process thread -- race; calls on one stream run in order and are safe.  Outputs equal qd_render_tiles_async's up to
 * the order of float additions (<= 2e-6). */
int qd_render_tiles_noalloc_async(qd_scene* scene, const qd_camera* camera, const qd_settings* settings,
                                  int tile, int tile_phase, int tile_stride,
                                  float* rgb_dev, qd_stats* stats_dev, void* hip_stream);

/* Replaces: nothing in the reference (it renders one frame at a time on one thread, src/main.cpp:163-168).
 * Binds a working set of the scene to `hip_stream` and sizes it for max_trace_depth now, so that later
 * qd_render_tiles_async calls on that stream allocate nothing (see "Frames in flight" above). */
int qd_scene_bind_stream(qd_scene* scene, int max_trace_depth, void* hip_stream);

/* De-interleave kernel for the gathered framebuffer: src_dev holds, for each rank r in
 * [0, tile_stride), that rank's compacted tiles (each rank's block padded to
 * tiles_per_rank_padded tiles); writes the row-major W x H x 3 image to dst_dev. */
int qd_untile_async(const float* src_dev, float* dst_dev, int width, int height,
                    int tile, int tile_stride, int tiles_per_rank_padded, void* hip_stream);

/* ----------------------------------------------------------------------------
 * Multi-GPU (one process per GPU; RCCL over xGMI).  Replaces: nothing in the reference (one render thread, one
 * process, src/sdl.cpp:194-233); SURVEY.md 8(e): the image shards by tiles, scene buffers are replicated, the
 * framebuffer is gathered to rank 0.  A C / C++ host does:
 *   rank 0: qd_comm_unique_id(id), hands the 128 bytes to the other ranks (MPI, a socket, a file ...);
 *   every rank: qd_init(local device), qd_comm_create(id, rank, world, &comm), qd_scene_create(...);
 *   per frame, every rank: qd_render_tiles_async(scene, ..., tile, rank, world, shard_dev, ...) then
 *                          qd_gather_frame_async(comm, shard_dev, gathered_dev, frame_dev, W, H, tile, stream);
 * rank 0's frame_dev holds the frame once the stream has drained.  qd_init(device) binds a process to ONE GPU
 * (SURVEY's sketch had qd_init(n_devices) for a single process driving all GPUs; one process per GPU is this build's
 * model, so the device ordinal is the argument and the world is formed by qd_comm_create).
 * -------------------------------------------------------------------------- */
typedef struct qd_comm qd_comm;
int qd_comm_unique_id(uint8_t id_out[128]);                                      /* ncclGetUniqueId */
int qd_comm_create(const uint8_t id[128], int rank, int world, qd_comm** out);   /* ncclCommInitRank on the qd_init device; collective over all ranks */
int qd_comm_destroy(qd_comm* comm);
/* `count` floats of every rank -> dst_dev[r * count] on rank 0 (dst_dev ignored elsewhere): grouped ncclSend / ncclRecv,
 * each peer straight to the root.  Rank 0 may produce its block in place (src_dev == dst_dev).  Enqueue only. */
int qd_comm_gather_async(qd_comm* comm, const float* src_dev, float* dst_dev, size_t count, void* hip_stream);
/* rank 0's buffer -> every rank (the AA pass hands the assembled first-pass frame back out).  Enqueue only. */
int qd_comm_broadcast_async(qd_comm* comm, float* buf_dev, size_t count, void* hip_stream);
/* The exchange step of a frame: every rank's compacted shard (tiles_per_rank_padded = ceil(n_tiles / world) tiles of
 * tile*tile*3 floats) -> gathered_dev on rank 0 (world such blocks), then the de-interleave kernel writes frame_dev
 * (width*height*3 floats, row-major).  gathered_dev / frame_dev are ignored on the other ranks.  Enqueue only. */
int qd_gather_frame_async(qd_comm* comm, const float* shard_dev, float* gathered_dev, float* frame_dev,
                          int width, int height, int tile, void* hip_stream);

/* --------------------------------------------------------------------------
 * Scene preparation on the device.  Host pointers in, host pointers out (once per scene; the kernels
 * are in csrc/device/prep.h); every table is bit-identical to the one libqdhost computes on the CPU
 * (tests/test_gpu_parity.py::test_device_prep_tables_equal_the_host_tables).  libqdhost calls them when they are given to
 * it with qdh_set_prep_hooks (include/qdhost.h).
 *
 * qd_prep_heightfield  Replaces: Heightfield::FillProperties' grey / gaussian blur pass and PopulateMaxHeights /
 *   PopulateNormals (src/heightfield.cpp:102-125, 170-214).  rgb: width*height*3 floats of the loaded bitmap; gauss: r*r
 *   floats, gauss[y*r + x] = exp(-(x^2+y^2)/(2 blur^2)) / (2 pi blur^2) as floats, r = min(128, nearestInt(3 blur)), or
 *   r = 0 for no blur (the weights need libm's exp and come from the caller).  Out: heights, maxheights (width*height
 *   floats each), normals (3 doubles per cell).
 * qd_prep_highmap      Replaces: Heightfield::BuildHighMap (src/heightfield.cpp:260-290).  highmap: 16 floats per cell,
 *   levels 0 .. max_k-1 filled, the rest zero.
 * qd_prep_differentiate Replaces: Bitmap::Differentiate (src/bitmap.cpp:312-326), in place on width*height*3 floats. */
int qd_prep_heightfield(uint32_t width, uint32_t height, const float* rgb, const float* gauss, int r,
                        float* heights, float* maxheights, double* normals);
int qd_prep_highmap(uint32_t width, uint32_t height, const float* heights, int max_k, float* highmap);
int qd_prep_differentiate(uint32_t width, uint32_t height, float* rgb);

/* Replaces: Color::toRGB32 / ConvertTo8Bit (src/color.cpp:16-24, src/utils.h:22,34):
 * clamp, floor(x*255+0.5); rgb float image (device) -> packed 8-bit B,G,R rows
 * bottom-up as Bitmap::SaveBMP writes them (src/bitmap.cpp:209-217). */
int qd_quantize_bgr8_async(const float* rgb_dev, uint8_t* bgr_dev, int width, int height,
                           size_t row_bytes, void* hip_stream);

/* Replaces: Color::toRGB32 / Color::toSRGB32 (src/color.cpp:16-36) as DisplayVFB / DisplayVFBRect apply them
 * to every pixel of the frame or of a finished bucket (src/sdl.cpp:38-52, 302-322; useSRGB from GlobalSettings,
 * src/main.cpp:99): float rgb image (device, row pitch in floats) -> one packed 32-bit word per pixel,
 * (b << blue_shift) | (g << green_shift) | (r << red_shift), channels through ConvertTo8Bit or, with use_srgb,
 * ConvertTo8Bit_sRGB (src/utils.h:34-35).  The reference's defaults are shifts 16 / 8 / 0. */
int qd_pack_rgb32_async(const float* rgb_dev, size_t row_pitch_floats, uint32_t* out_dev, int width, int height,
                        int use_srgb, int red_shift, int green_shift, int blue_shift, void* hip_stream);

/* Counting build of the kernels: when on, renders also fill kd_nodes_visited / tris_tested /
 * mesh_queries (the reference's debug counters nbBBoxIntersections / nbTriIntersections /
 * nbIntersections, src/mesh.cpp:20-22); off by default because the counters cost registers. */
int qd_set_counting(int on);

/* Replaces: nothing in the reference (its Raytrace recursion has one form).  Selects how a frame is scheduled on the GPU:
 * 0 = one persistent kernel, every lane with its own LIFO queue of ray tasks (memory bounded by queue depth x lanes);
 * 1 = wavefront pipeline: generations of rays in HBM queues, a KD walk kernel per mesh node that refills idle lanes inside
 *     its loop and a full-width shading kernel per generation.  Queues hold max(4, tasks per hit + 1) entries per pixel
 *     within 16 GB per working set.  A frame that asks for more (25 glossy samples at 4K) is first tried in one band with
 *     what fits (at least 8 entries per pixel; most such frames use a fraction of what they may ask for), and once a
 *     frame has outgrown that the scene's frames are rendered in bands of whole 8 x 8 units, one after the other on the
 *     stream.  No call fails over the queues: a generation that outgrows them sets a device word, and the launch that
 *     follows every band on the stream -- the band once more in form 0, which returns at once when the word is clear --
 *     renders that band again in place; a scene whose bands still overflow (glass that keeps doubling) then stays on form 0
 *     (qd_scene_wavefront_info tells).  The frame is complete and correct either way;
 * 2 = (default) the faster of the two for the scene at hand: each form is timed with HIP events on frames of the scene that
 *     ran alone on the GPU (enqueued when the scene's earlier work had finished, finished before the next launch; read
 *     without waiting), a form's first such frame is dropped, the faster form renders, and a caller that waits for its
 *     frames sends every 64th to the other form.  A caller that keeps several frames in flight renders a few frames one
 *     at a time first (FrameRenderer.settle(): 6); frames that overlap are never timed.  Until there are figures the
 *     scene renders in form 1, or in form 0 if its device tree has more than 4 M nodes.
 *     Which one wins depends on the frame: form 1 where rays have very different fates (the reference's kdtree_test: 2x),
 *     form 0 where all pixels carry similar work (the 10 M triangle terrain: 1.3x).
 * Same arithmetic in both forms, same results up to the order of float additions.  Applies to calls made after it. */
int qd_set_pipeline(int pipeline);
/* What the adaptive choice has measured for a scene: the form its most recent launch ran in (0 / 1) and the device time per
 * pixel slot of each form in nanoseconds (0 = not measured yet); also reads the latest frame's events if that frame ran
 * alone and has finished.  Any pointer may be NULL. */
int qd_scene_schedule_info(const qd_scene* scene, int* last_form, double* ns_per_pixel_form0, double* ns_per_pixel_form1);

/* Replaces: nothing in the reference.  Memory the ray queues of form 1 may take per working set (default 16 GB; 0 = default):
 * a frame beyond the budget is rendered in bands (with the default budget only after a first frame of the scene outgrew one
 * band, see qd_set_pipeline; an explicit budget is a hard limit from the first frame).  max_entries, for tests (0 = off): a cap
 * on the entries of a queue that the band sizes do not take into account, so that bands overflow and the in-stream fallback
 * runs.  qd_scene_wavefront_info: the bands the NEXT width x height frame of this scene takes, and whether a band sized by
 * the shader graph ever outgrew its queues (it was rendered again in form 0 on the spot, and the scene stays on form 0). */
int qd_set_wavefront_budget(uint64_t queue_bytes, uint32_t max_entries);
int qd_scene_wavefront_info(const qd_scene* scene, int width, int height, uint32_t* bands, int* overflowed);

/* Replaces: nothing in the reference.  A frame of form 1 that is enqueued while nothing else of the scene is in flight -- the
 * only kind a caller that renders one image, or waits for every frame, ever produces -- is enqueued as `parts` (default 4,
 * the maximum; 1 = off; 0 = default) concurrent launch sequences: part k renders the 8 x 8 pixel units k, k + parts, ... on
 * a stream and working set the library owns (part 0 on the caller's), forked from and joined to the caller's stream with
 * events.  For the caller it stays one enqueue-only launch on its stream; results are the same up to the order of float
 * additions.  One part's generation tails run under the other parts' bulk, as with frames in flight: kdtree_test 10.0 ->
 * 8.8 ms, lecture7 at 3840 x 2160 8.1 -> 5.8 ms.  Frames below 1024 x 768 pixels, frames of form 0, frames enqueued while
 * others are in flight and the first frame after such a frame (a caller with frames in flight gains nothing from parts)
 * stay whole.  qd_scene_frame_parts: how many parts the scene's latest launch was enqueued as. */
int qd_set_frame_parts(int parts);
int qd_scene_frame_parts(const qd_scene* scene, uint32_t* last_parts);

/* Replaces: the leaf criterion of Mesh::BuildKD (src/mesh.cpp:218, TRIANGLES_PER_LEAF = 20 in
 * src/constants.h:14) for the DEVICE copy of the tree only.  qd_scene_create keeps the caller's
 * tree and splits each of its leaves further with a surface-area heuristic until a leaf holds at most
 * `max_leaf_tris` triangles or a split costs more than it saves (`step_cost` = cost of one descent
 * step in triangle tests).  max_leaf_tris = 0 uses the caller's tree unchanged.  Default (2, 0.6).
 * The closest hit does not depend on the grouping; only exact distance ties between two different
 * triangles may resolve differently.  Applies to scenes created after the call. */
int qd_set_kd_refine(int max_leaf_tris, double step_cost);

/* Replaces: Mesh::BuildKD (src/mesh.cpp:198-432: planes at the midpoint of the box on axis depth % 3, or by the 10-step
 * "SAH" scan; 20 triangles per leaf) for the DEVICE copy of the tree only.
 *   mode 1 (default): the device tree is built ON THE DEVICE from the triangles (HIP kernels, csrc/device/kdbuild.h: a
 *           level of the tree per round; 0.2 s for 10 M triangles) by a surface-area heuristic over clipped triangle
 *           bounds (64 bins per axis for nodes above 192 triangles, exact sweep below), triangles classified against a
 *           plane as PartitionTriangle does (src/mesh.cpp:412-432); the caller's tree is not consulted and need not
 *           exist (a mesh with QD_MESH_WANT_KD and kd_root -1 gets its tree here);
 *   mode 2: the same tree (the same planes, node for node, below 524 288 triangles) built by host threads (kdrefine.h:
 *           1.2 s for 10 M triangles on 16 cores) -- kept as the check of mode 1;
 *   mode 0: the caller's planes are kept and only its leaves are split further (qd_set_kd_refine); needs the caller's tree.
 * On the 10 M triangle bench mesh modes 1 / 2 visit 28 % fewer nodes and test 39 % fewer triangles per ray than mode 0, in a
 * tree of 20 M instead of 46 M nodes.  The closest hit does not depend on the tree.  With qd_set_kd_refine(0, .) the
 * caller's tree is used as it is, whatever the mode.  Applies to scenes created after the call. */
int qd_set_kd_build(int mode);

/* Size of the device tree of a scene (after refinement): nodes, leaf references, leaves, largest leaf.
 * Any pointer may be NULL.  The reference prints the same figures for its own tree at src/mesh.cpp:211-213. */
int qd_scene_kd_info(const qd_scene* scene, uint64_t* nodes, uint64_t* leaf_refs, uint64_t* leaves, uint64_t* max_leaf);
/* Where qd_scene_create spent its time (seconds): building the device KD trees (kdbuild.h on the device, or kdrefine.h on host threads) and uploading the
 * scene + gathering the leaf vertices on the device.  Replaces: the "KD-tree built in %.2lfs" report of src/mesh.cpp:209. */
int qd_scene_setup_info(const qd_scene* scene, double* kd_build_seconds, double* upload_seconds);

/* Replaces: nothing in the reference (its render loop cannot fail).  Returns <0 if an asynchronous launch on this scene raised
 * a device error word since the last check (the per-lane ray queue of form 0 overflowed -- its depth is derived from the shader
 * graph, so this says the description lied about it; synchronises the device).  The queues of form 1 never raise it. */
int qd_scene_check(qd_scene* scene);

void qd_scene_destroy(qd_scene* scene);
void qd_shutdown(void);
const char* qd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* QDTRACE_H */
