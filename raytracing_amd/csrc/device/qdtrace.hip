// libqdtrace.so -- HIP (gfx950 / MI355X) trace path behind the C-ABI of include/qdtrace.h.
//
// One persistent kernel replaces the reference's recursive Raytrace() (src/main.cpp:18-56):
// every lane owns one pixel at a time and runs an explicit, bounded LIFO queue of ray tasks
// (radiance rays and shadow rays) that lives in an HBM workspace laid out lane-interleaved
// (word-major, lanes contiguous) so that a wave's queue traffic coalesces.  Colour is
// accumulated as sum(weight * leaf colour): the reference's shading tree is linear in its leaves
// (SURVEY.md section 3C), so recursion unrolls into weights.  Waves pull 8x8-pixel units from
// eight work queues (one per XCD label, with stealing) so that each XCD's private L2 sees one
// contiguous image region and therefore one region of the KD-tree.
#include <hipcub/hipcub.hpp>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstddef>
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "dshade.h"
#include "qdtrace.h"
#include "kdrefine.h"
#include "kdbuild.h"

using namespace qd;

#ifdef QD_NO_OUTLINE
#define QD_OUTLINE __forceinline__
#else
#define QD_OUTLINE __attribute__((noinline))
#endif

namespace {

// ------------------------------------------------------------------------------------------
// Ray task queue entry: 10 x 8-byte words per slot.
//   0..2 origin, 3..5 direction, 6 = (w.r, w.g), 7 = (w.b, flags|depth),
//   radiance ray : 8 = path id (low 32 bits)
//   shadow ray   : 8 = squared distance to the light, 9 = intensity / |ip - light|^2
constexpr int kEntryWords = 11;
#ifndef QD_UNIT_SIDE
#define QD_UNIT_SIDE 8
#endif
constexpr int kUnitSide = QD_UNIT_SIDE;                  // work unit = kUnitSide x kUnitSide pixels, handed to one wave
constexpr uint32_t kUnitPx = kUnitSide * kUnitSide;
#ifndef QD_BUSY_LANES
#define QD_BUSY_LANES 48
#endif
#ifndef QD_BUSY_WINDOWS
#define QD_BUSY_WINDOWS 3
#endif
constexpr uint32_t kBusyWindows = QD_BUSY_WINDOWS;
constexpr uint32_t kBusyLanes = QD_BUSY_LANES;            // tasks per round that count as "all lanes busy" (see 1b in the kernel)
constexpr uint32_t kCtlStride = 32;                      // control words sit 128 bytes apart: one cache line each, so the waves'
                                                         // atomics on one counter do not queue up behind those on its neighbours
constexpr uint32_t kFlagShadow = 0x80000000u, kFlagPrimary = 0x40000000u, kDepthMask = 0x00ffffffu;

struct Params {
    qd_camera cam;
    int W, H;
    float ambient[3];
    uint32_t max_depth;
    int mode;                     // 0 = rectangle, 1 = interleaved tiles (compacted output), 2 = anti-aliasing taps of listed pixels
    const uint32_t* aa_pixels;    // mode 2: x | y << 16 per listed pixel; sample j = tap 1 + j % 4 of pixel j / 4
    uint32_t n_samples;           // mode 2: 4 * listed pixels
    int x0, y0, rw, rh;           // rectangle
    int tile, tiles_x, phase, stride;
    uint32_t n_units;             // 8x8 pixel units to render
    uint32_t unit0;               // first unit of this launch (a frame rendered in bands or parts: wavefront form, see launch_wavefront)
    uint32_t unit_step;           // the launch's i-th unit is unit0 + i * unit_step (0 reads as 1; > 1: one of the interleaved parts of a lone frame)
    const uint32_t* cond;         // not null: the launch does nothing unless *cond != 0 (the in-stream fallback of a wavefront band)
    uint32_t units_x;             // rectangle mode: units per row
    uint32_t* queue_ctr;          // 8 work-queue counters, kCtlStride words apart (zeroed before launch)
    uint64_t* stack;              // ray task workspace
    uint32_t stack_cap;           // slots per lane
    uint32_t n_lanes;
    float* rgb; size_t pitch;     // output (pitch in floats, rectangle mode); zeroed by launch(), the kernel adds into it
    size_t rgb_floats;            // floats of `rgb` this launch may write
    int32_t* hit_node; int32_t* hit_tri; double* hit_dist;
    unsigned long long* stats;    // 8 counters, see qd_stats
    int* error_flag;
    // passes (see launch()): pass 0 hands out pixels, later passes hand out the ray tasks the pass before left behind
    int source;                          // 0 = pixel slots, 1 = task list
    uint32_t grace;                      // rounds a wave may still run once the pass's work is all handed out; then it leaves
                                         // its queued tasks to the next pass (0xffffffff = run to the end)
    uint32_t* pass_done;                 // set by the first wave that finds nothing left to hand out; one cache line further:
                                         // the number of waves of this pass that have left
    const uint64_t* in_list; const uint32_t* in_count;     // source 1: tasks, word-major (word w of task i at [w * list_cap + i]);
    uint64_t* out_list; uint32_t* out_count; size_t list_cap;   // eight regions of list_cap / 8 tasks with a count each
    // wavefront pipeline (wavefront.h): two ray queues (generations alternate), hit records, control words, deep KD stack levels
    uint64_t* wf_ray[2]; uint64_t* wf_res; uint32_t* wf_ctl; uint2* wf_deep; uint32_t wf_cap;
    uint32_t wf_limit;                   // entries of a queue that appends may fill (= wf_cap; smaller only under qd_set_wavefront_budget's test switch)
    uint32_t* wf_over;                   // set by an append that finds its queue full (the band is then rendered again, see wf_band_end_kernel)
    unsigned long long* user_stats;      // wavefront form: the caller's counters; `stats` is then the band's own block
    uint32_t* host_flag;                 // pinned host word: "a wavefront band of this scene overflowed"
};

struct Task {
    DV o, d;
    Col w;
    uint32_t flags_depth;
    uint32_t path;
    double aux0, aux1;
    uint32_t pixel;               // output slot the ray's colour is added to (a task may be traced by any lane of the wave)
};

// The kernels take the scene and the launch parameters by value; a function that is NOT inlined into a kernel takes them
// by reference.  A reference to the kernel's by-value parameter would make the compiler copy the whole struct to scratch
// memory in every wave's prologue (0.6 KB per lane) and read its fields back lane by lane; instead the kernel hands out a
// reference to the argument as it lies in the kernel-argument segment (ordinary constant memory: explicit arguments start
// at offset 0, each at its natural alignment), and the callee reads it through uniform_const: address made wave-uniform,
// constant address space -> scalar loads (dmath.h).  kernarg_scene / kernarg_params are for code that is inlined into a
// KERNEL: in a function behind a real call the compiler resolves the kernel-argument pointer to null.
constexpr size_t kParamsKernarg = (sizeof(DScene) + alignof(Params) - 1) / alignof(Params) * alignof(Params);
QD_DEV const DScene& kernarg_scene() {
    return *(const DScene*)(const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
}
QD_DEV const Params& kernarg_params() {
    return *(const Params*)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kParamsKernarg);
}

// What the two accessors above rely on, checked: both structs are plain bytes, and Params lies behind DScene where a struct of
// the two would put it.  qd_init also launches kernarg_selftest_kernel once per device (sentinel values in both arguments, read
// back through the accessors): a compiler or code-object change that moves the explicit arguments fails qd_init instead of
// handing the callees garbage scene pointers.
static_assert(std::is_trivially_copyable<DScene>::value && std::is_trivially_copyable<Params>::value, "kernel arguments are copied as bytes");
struct KernargLayout { DScene s; Params p; };
static_assert(offsetof(KernargLayout, p) == kParamsKernarg, "Params must follow DScene at its natural alignment in the kernel-argument segment");
__global__ void kernarg_selftest_kernel(const DScene S, const Params P, uint32_t* out) {
    const DScene& s2 = kernarg_scene();
    const Params& p2 = kernarg_params();
    bool ok = s2.n_nodes == S.n_nodes && s2.n_lights == S.n_lights && s2.nodes == S.nodes && s2.env_faces[5] == S.env_faces[5];
    ok = ok && p2.W == P.W && p2.H == P.H && p2.n_units == P.n_units && p2.rgb == P.rgb && p2.wf_cap == P.wf_cap && p2.max_depth == P.max_depth;
    if (threadIdx.x == 0) out[0] = ok ? 0x600dbeefu : 0xbadu;
}

QD_DEV uint64_t pack2f(float a, float b) { return (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32); }
QD_DEV uint64_t pack_fu(float a, uint32_t b) { return (uint64_t)__float_as_uint(a) | ((uint64_t)b << 32); }

struct LaneQueue {
    uint64_t* base; size_t stride; uint32_t cap; int sp; int* err;
    QD_DEV void push(const Task& t) {
        if ((uint32_t)sp >= cap) { *err = 1; return; }
        uint64_t* p = base + (size_t)sp * kEntryWords * stride;
        p[0 * stride] = __double_as_longlong(t.o.x); p[1 * stride] = __double_as_longlong(t.o.y); p[2 * stride] = __double_as_longlong(t.o.z);
        p[3 * stride] = __double_as_longlong(t.d.x); p[4 * stride] = __double_as_longlong(t.d.y); p[5 * stride] = __double_as_longlong(t.d.z);
        p[6 * stride] = pack2f(t.w.r, t.w.g);
        p[7 * stride] = pack_fu(t.w.b, t.flags_depth);
        if (t.flags_depth & kFlagShadow) { p[8 * stride] = __double_as_longlong(t.aux0); p[9 * stride] = __double_as_longlong(t.aux1); }
        else p[8 * stride] = t.path;
        p[10 * stride] = t.pixel;
        ++sp;
    }
    // entry `idx` of the queue that starts at `b` (any lane's queue of this wave: same layout, base shifted by the lane)
    static QD_DEV Task load(const uint64_t* b, size_t stride, int idx) {
        const uint64_t* p = b + (size_t)idx * kEntryWords * stride;
        Task t;
        t.o = mk(__longlong_as_double(p[0 * stride]), __longlong_as_double(p[1 * stride]), __longlong_as_double(p[2 * stride]));
        t.d = mk(__longlong_as_double(p[3 * stride]), __longlong_as_double(p[4 * stride]), __longlong_as_double(p[5 * stride]));
        const uint64_t a = p[6 * stride], w = p[7 * stride];
        t.w = mkc(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)w));
        t.flags_depth = (uint32_t)(w >> 32);
        if (t.flags_depth & kFlagShadow) { t.aux0 = __longlong_as_double(p[8 * stride]); t.aux1 = __longlong_as_double(p[9 * stride]); t.path = 0; }
        else { t.path = (uint32_t)p[8 * stride]; t.aux0 = t.aux1 = 0; }
        t.pixel = (uint32_t)p[10 * stride];
        return t;
    }
    QD_DEV Task pop() { --sp; return load(base, stride, sp); }
};

// ------------------------------------------------------------------------------------------
// One ray task against the node list, one code path for both kinds of ray (so that lanes carrying
// shadow rays and lanes carrying radiance rays run the same traversal instructions):
//  * radiance ray: closest node in file order, strict '<' (src/main.cpp:23-38).  Only the winner's
//    minimal record (node, distance, triangle, barycentrics) is carried through the search; its full
//    hit record is rebuilt once at the end (analytic geometry is simply intersected again, which is
//    deterministic; mesh / heightfield records are functions of the stored distance).
//  * shadow ray: product of shadowTransparency over the nodes whose closest hit is nearer than the light
//    (ShadingHelper::GetShadowTransparency, src/shadinghelper.cpp:20-40).  Once the product is exactly 0
//    no later factor can change it, so the node loop stops there.
// GEOM: which geometry code this copy of the node loop contains besides meshes and planes (kGeomHf: heightfield DDA,
// kGeomCsg: CSG walk, kGeomPrim: sphere / cube / regular polygon).  qd_scene_create derives the mask from the scene and
// launch() picks the kernel: what a scene does not hold is compiled out -- smaller live ranges, fewer spills around the
// mesh query (the 10 M triangle scene, meshes + a plane: +9.5 %).
// kGeomCsgDeep: a CSG node whose operands are not all analytic primitives (another CSG, a mesh, a heightfield): csg_generic.
constexpr int kGeomHf = 1, kGeomCsg = 2, kGeomPrim = 4, kGeomCsgDeep = 8, kGeomAll = 15;
template <bool STATS, int GEOM, int LDS_LEVELS>
QD_DEV int trace_nodes_body(const DScene& S, const DV& o, const DV& d, bool shadow, double target,
                            float& transp, double& best_dw, GHit& hit, Counters& cnt) {
    int best_node = -1;
    best_dw = kInf; transp = 1.f;
    double w_dist = 0, w_l2 = 0, w_l3 = 0;
    int w_tri = -1;
    const uint32_t n_nodes = S.n_nodes;
    uint32_t i = 0;
    bool final_pass = false;
    for (;;) {
        if (!final_pass && i == n_nodes) {
            if (shadow || best_node < 0) break;
            final_pass = true; i = (uint32_t)best_node;
        }
        const unsigned long long ta0 = QD_STAMP();
        const qd_node& n = S.nodes[i];
        const ObjRay r = to_object(n, o, d);
        const qd_geom g = S.geoms[n.geom];
        double dist = 0, c_l2 = 0, c_l3 = 0;
        int c_tri = -1;
        bool got;
        const unsigned long long ta1 = QD_STAMP();
        cnt.t_a += ta1 - ta0;
        if (g.type == QD_GEOM_MESH) {
            if (final_pass) {
                hit.dist = w_dist; hit.ip = r.o + r.d * w_dist; hit.tri = w_tri; hit.mesh = g.a; hit.l2 = w_l2; hit.l3 = w_l3;
                break;
            }
            const MHit mh = mesh_query<STATS, LDS_LEVELS>(mesh_data(S), g.a, r.o, r.d, shadow, target, r.len);
            got = mh.tri >= 0;
            if (got) { dist = mh.dist; c_tri = mh.tri; c_l2 = mh.l2; c_l3 = mh.l3; }
            if (STATS) { cnt.kd_nodes += mh.n_nodes; cnt.tris += mh.n_tris; cnt.mesh_queries += mh.n_nodes ? 1u : 0u; }
#ifdef QD_PROFILE
            cnt.t_desc += mh.n_nodes; cnt.t_leaf += mh.n_tris;
#endif
            cnt.t_b += QD_STAMP() - ta1;
        } else if ((GEOM & kGeomHf) && g.type == QD_GEOM_HEIGHTFIELD) {
            if (final_pass) { hf_finish(S, g.a, r.o, r.d, w_dist, hit); break; }
            got = hf_intersect<STATS>(S, g.a, r.o, r.d, dist, cnt.kd_nodes, cnt.tris);
        } else {
            GHit h;
            if ((GEOM & kGeomCsgDeep) && g.type == QD_GEOM_CSG) {
                Counters again = cnt;                       // the winner is intersected a second time: do not count that walk twice
                got = csg_generic<STATS, kMaxCsgNesting - 1>(S, n.geom, r.o, r.d, h, final_pass ? again : cnt);
            } else got = analytic_intersect<(GEOM & kGeomCsg) != 0, (GEOM & kGeomPrim) != 0>(S, g, r.o, r.d, h);
            cnt.t_c += QD_STAMP() - ta1;
            if (final_pass) { hit = h; break; }
            dist = h.dist;
        }
        const unsigned long long ta2 = QD_STAMP();
        if (got) {
            const double dw = dist / r.len;
            if (shadow) {
                if (sqr(dw) < target) {
                    transp *= n.shadow_transparency;
                    if (transp == 0.f) break;
                }
            } else if (!(best_dw <= dw)) {
                best_dw = dw; best_node = (int)i; w_dist = dist; w_tri = c_tri; w_l2 = c_l2; w_l3 = c_l3;
            }
        }
        cnt.t_d += QD_STAMP() - ta2;
        ++i;
    }
    return best_node;
}

// behind a real call (the kernel copies that run 4 waves per SIMD: the traversal's register allocation stays small)
template <bool STATS, int GEOM>
__device__ QD_OUTLINE int trace_nodes(const DScene& S_, const DV& o, const DV& d, bool shadow, double target,
                                      float& transp, double& best_dw, GHit& hit, Counters& cnt) {
    return trace_nodes_body<STATS, GEOM, kLdsStack>(uniform_const(S_), o, d, shadow, target, transp, best_dw, hit, cnt);
}

// World-space IntersectionInfo of the winning hit (src/geometry.cpp:331-338).
template <bool RICH>
QD_DEV void assemble_info(const DScene& S, const qd_node& n, GHit& h, double dw, const DV& ray_dir, Info& info) {
    const bool want_dnd = RICH && n.bump_tex >= 0;
    DV dndx = mk(0, 0, 0), dndy = mk(0, 0, 0);
    if (h.tri >= 0) mesh_finish(S, h, dndx, dndy, want_dnd);
    else if (h.dnd == 1) { dndx = mk(1, 0, 0); dndy = mk(0, 0, 1); }
    info.normal = normalize(mul_mat(h.n, n.MinvT));
    if (want_dnd) {
        info.dNdx = normalize(mul_mat(dndx, n.M));
        info.dNdy = normalize(mul_mat(dndy, n.M));
    } else { info.dNdx = dndx; info.dNdy = dndy; }
    info.ip = mul_mat(h.ip, n.M) + ld3(n.offset);
    info.distance = dw;
    info.u = h.u; info.v = h.v;
    info.rayDir = ray_dir;
}

// One non-layered shader at a hit: adds its direct colour to acc and queues its rays
// (Lambert/Phong/BlinnPhong/OrenNayar/Const/Reflection/Refraction::Shade, src/shading.cpp).
constexpr int kMaxLayerNesting = 4;      // Layered inside Layered ...: levels below the node's own shader (qd_scene_create checks)
template <bool RICH, class Queue>
__device__ QD_OUTLINE void eval_nested_layered(const DScene& S, const Params& P, const qd_shader& top, Col w, uint32_t slot, const Task& ray,
                                               uint32_t depth, const Info& info, Col& acc, Queue& q);

// NESTED: a Layered shader met here (a layer that is itself Layered) goes to eval_nested_layered, which walks the inner
// layers with an explicit stack and calls the NESTED = false copy for the leaves -- no recursion on the device.
// Queue: where child rays go -- the lane's LIFO queue (one-kernel form) or the next generation's queue (wavefront form).
// The body is inlined into the wavefront shading kernel (its own kernel, 2 waves per SIMD: +10 .. +30 % there) and sits
// behind a real call in the one-kernel form (eval_leaf_shader below: the traversal's register allocation stays small).
// S_mem / P_mem: the scene and the parameters as they lie in memory, for the one call made from here (see kernarg_scene).
template <bool NESTED, bool RICH, class Queue>
QD_DEV void eval_leaf_shader_body(const DScene& S, const Params& P, const DScene& S_mem, const Params& P_mem, int si, Col w, uint32_t slot,
                                  const Task& ray, uint32_t depth, const Info& info, Col& acc, Queue& q) {
    const qd_shader sh = S.shaders[si];
    switch (sh.type) {
        case QD_SH_LAYERED:
            if (NESTED) {
                // a rare path behind a real call: it works on copies, so that the caller's ray, hit record, colour and queue
                // state do not have their addresses taken (they would live in scratch memory for the whole kernel)
                const Task ray2 = ray; const Info info2 = info; Col acc2 = acc; Queue q2 = q; const qd_shader sh2 = sh;
                eval_nested_layered<RICH, Queue>(S_mem, P_mem, sh2, w, slot, ray2, depth, info2, acc2, q2);
                acc = acc2; q = q2;
            }
            break;
        case QD_SH_CONST: acc = acc + w * ldc(sh.color); break;
        case QD_SH_LAMBERT: case QD_SH_PHONG: case QD_SH_BLINNPHONG: case QD_SH_ORENNAYAR: {
            const Col diffuse = sh.tex >= 0 ? tex_sample<RICH>(S, sh.tex, info) : ldc(sh.color);
            acc = acc + w * (diffuse * ldc(P.ambient));
            for (uint32_t li = 0; li < S.n_lights; ++li) {
                const qd_light L = S.lights[li];
                const DV lpos = ld3(L.pos);
                Col cl;
                if (sh.type == QD_SH_ORENNAYAR) {                                  // :102-126
                    const double sigma2 = sqr(sh.p0);
                    const double VdotN = dot(-ray.d, info.normal);
                    const DV lightDir = normalize(lpos - info.ip);
                    const double LdotV = dot(lightDir, -ray.d);
                    const double LdotN = dot(lightDir, info.normal);
                    const double sv = LdotV - LdotN * VdotN;
                    const double t = (sv <= 0. ? 1. : smax(LdotN, VdotN));
                    const double a = 1 - 0.5 * sigma2 / (sigma2 + 0.33);
                    const double b = 0.45 * sigma2 / (sigma2 + 0.09);
                    cl = diffuse * (float)(LdotN * (a + b * sv / t));
                } else {
                    const DV lightDir = normalize(info.ip - lpos);                 // :17-19, :59-61
                    const DV normal = faceforward(lightDir, info.normal);
                    const double lambert = dot(normal, -lightDir);
                    cl = diffuse * (float)lambert;
                    if (sh.type == QD_SH_PHONG) {                                  // :42-50
                        const DV refl = reflect(lightDir, normal);
                        const double cg = dot(-ray.d, refl);
                        const double spec = cg > 0. ? pow(cg, sh.p1) : 0.;
                        cl = cl + mkc(1.f, 1.f, 1.f) * (float)spec * (float)sh.p0;
                    } else if (sh.type == QD_SH_BLINNPHONG) {                      // :93-100
                        const DV ld = normalize(lpos - info.ip);
                        const DV half = normalize(ld - ray.d);
                        const double spec = pow(smax(dot(info.normal, half), 0.), sh.p1);
                        cl = cl + mkc(1.f, 1.f, 1.f) * (float)spec * (float)sh.p0;
                    }
                }
                // shadow ray towards the light (ShadingHelper::GetLightContribution, src/shadinghelper.cpp:11-18)
                Task s;
                s.o = info.ip + info.normal * 1e-6;
                s.d = normalize(lpos - s.o);
                s.aux0 = len_sqr(lpos - s.o);
                s.aux1 = L.intensity / len_sqr(info.ip - lpos);
                s.w = w * cl;
                s.flags_depth = kFlagShadow | depth;
                s.path = 0;
                s.pixel = ray.pixel;
                q.push(s);
            }
            break;
        }
        case QD_SH_REFL: {                                                         // :155-198
            const DV n = faceforward(ray.d, info.normal);
            Task r;
            r.o = info.ip + n * 0.000001;
            r.flags_depth = depth + 1;
            r.aux0 = r.aux1 = 0;
            r.pixel = ray.pixel;
            if (sh.p1 == 1) {
                r.d = reflect(ray.d, n);
                r.w = w * (float)sh.p0;
                r.path = path_child(ray.path, slot, 0);
                q.push(r);
            } else {
                const int count = depth > 0 ? 2 : sh.i0;
                const Col wi = (w * (float)sh.p0) / (float)count;
                DV a, b;
                orthonormal_system(n, a, b);
                const double k = tan((1 - sh.p1) * kPi / 2);
                for (int i = 0; i < count; ++i) {
                    const double angle = random01(ray.path, slot * 64u + (uint32_t)i, 0) * 2 * kPi;   // src/utils.cpp:63-69
                    const double radius = dsqrt(random01(ray.path, slot * 64u + (uint32_t)i, 1));
                    double x = cos(angle) * radius, y = sin(angle) * radius;
                    x *= k; y *= k;
                    const DV mn = n + a * x + b * y;
                    r.d = reflect(ray.d, mn);
                    r.w = wi;
                    r.path = path_child(ray.path, slot, (uint32_t)i + 1);
                    q.push(r);
                }
            }
            break;
        }
        case QD_SH_REFR: {                                                         // :213-238
            DV refr;
            if (dot(ray.d, info.normal) < 0.) refr = refract(ray.d, info.normal, 1 / sh.p0);
            else refr = refract(ray.d, -info.normal, sh.p0);
            if (len_sqr(refr) == 0) { acc = acc + w * mkc(1, 0, 0); break; }       // total internal reflection: red
            Task r;
            r.o = info.ip - faceforward(ray.d, info.normal) * 0.000001;
            r.d = refr;
            r.w = w * (float)sh.p1;
            r.flags_depth = depth + 1;
            r.path = path_child(ray.path, slot, 0);
            r.aux0 = r.aux1 = 0;
            r.pixel = ray.pixel;
            q.push(r);
            break;
        }
        default: break;
    }
}

template <bool NESTED, bool RICH, class Queue>
__device__ QD_OUTLINE void eval_leaf_shader(const DScene& S_, const Params& P_, int si, Col w, uint32_t slot, const Task& ray, uint32_t depth,
                             const Info& info, Col& acc, Queue& q) {
    const DScene& S = uniform_const(S_); const Params& P = uniform_const(P_);
    eval_leaf_shader_body<NESTED, RICH, Queue>(S, P, S_, P_, si, w, slot, ray, depth, info, acc, q);
}

// Layered::Shade (src/shading.cpp:260-275) for a Layered that is a layer of another one: the same suffix-product walk as
// in the kernel, one stack frame per nesting level.  A layer's path-id slot is parent_slot * 32 + i + 1 (the oracle's rule).
template <bool RICH, class Queue>
__device__ QD_OUTLINE void eval_nested_layered(const DScene& S_, const Params& P_, const qd_shader& top, Col w, uint32_t slot, const Task& ray,
                                               uint32_t depth, const Info& info, Col& acc, Queue& q) {
    const DScene& S = uniform_const(S_); const Params& P = uniform_const(P_);
    struct Frame { int i0, i; uint32_t slot; Col w, suffix; };
    Frame st[kMaxLayerNesting];
    int sp = 0;
    st[0] = Frame{top.i0, top.i1 - 1, slot, w, mkc(1.f, 1.f, 1.f)};
    while (sp >= 0) {
        Frame& f = st[sp];
        if (f.i < 0) { --sp; continue; }
        const qd_layer l = S.layers[f.i0 + f.i];
        Col blend = ldc(l.blend);
        if (l.tex >= 0) blend = blend * tex_sample<RICH>(S, l.tex, info);
        const Col wl = f.w * (blend * f.suffix);
        const uint32_t child_slot = f.slot * 32u + (uint32_t)f.i + 1u;
        f.suffix = f.suffix * (mkc(1.f, 1.f, 1.f) - blend);
        --f.i;
        const qd_shader inner = S.shaders[l.shader];
        if (inner.type == QD_SH_LAYERED) {
            if (sp + 1 < kMaxLayerNesting) { ++sp; st[sp] = Frame{inner.i0, inner.i1 - 1, child_slot, wl, mkc(1.f, 1.f, 1.f)}; }
        } else {
            eval_leaf_shader<false, RICH, Queue>(S, P, l.shader, wl, child_slot, ray, depth, info, acc, q);
        }
    }
}

// Hands out `want` consecutive pixel slots (slot = unit * kUnitPx + pixel inside the unit, so consecutive slots are
// neighbours on screen) from the eight XCD-labelled queues, the caller's own queue first.  Returns the number granted
// (0 = every queue is exhausted) and the first slot.
QD_DEV uint32_t fetch_slots(const Params& P, uint32_t home, uint32_t want, uint32_t& first) {
    const uint32_t n_slots = P.n_units * kUnitPx;
    const uint32_t per = ((n_slots + 7u) / 8u + kUnitPx - 1u) / kUnitPx * kUnitPx;      // pixels: whole units per queue
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t qi = (home + i) & 7u;
        uint32_t lo, size;
        if (P.source == 0) {
            lo = qi * per;
            if (lo >= n_slots) continue;
            size = min(per, n_slots - lo);
        } else {                                                                        // tasks: queue qi = region qi of the list
            lo = qi * (uint32_t)(P.list_cap / 8);
            size = min(__hip_atomic_load(&P.in_count[qi * kCtlStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (uint32_t)(P.list_cap / 8));
        }
        if (__hip_atomic_load(&P.queue_ctr[qi * kCtlStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= size) continue;
        const uint32_t v = atomicAdd(&P.queue_ctr[qi * kCtlStride], want);
        if (v < size) { first = lo + v; return min(want, size - v); }
    }
    return 0u;
}

QD_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#ifndef QD_WAVES_PER_EU
#define QD_WAVES_PER_EU 4
#endif
constexpr uint32_t kWavesPerCU = 4 * QD_WAVES_PER_EU;

// position of the r-th (0-based) set bit of m
QD_DEV uint32_t nth_set_bit(unsigned long long m, uint32_t r) {
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t w = 32; w >= 1; w >>= 1) {
        const uint32_t c = (uint32_t)__popcll(m & (((1ull << w) - 1ull) << pos));
        if (r >= c) { r -= c; pos += w; }
    }
    return pos;
}

// vfb[y][x] += colour.  The reference assigns Raytrace()'s return value once (src/main.cpp:96); here the value is the
// sum of the contributions of the pixel's ray tasks, which any lane of the wave may have traced, so they are added
// with hardware float atomics into a buffer zeroed before the launch.  Tasks of one pixel handled by one lane arrive
// in program order; only pixels whose tasks were shared between lanes see a different order of additions.
QD_DEV void add_px(const Params& P, uint32_t pixel, const Col& c) {
    if (c.r == 0.f && c.g == 0.f && c.b == 0.f) return;
    float* o = P.rgb + (P.mode == 0 ? (size_t)(pixel / (uint32_t)P.rw) * P.pitch + (size_t)(pixel % (uint32_t)P.rw) * 3 : (size_t)pixel * 3);
    unsafeAtomicAdd(o, c.r); unsafeAtomicAdd(o + 1, c.g); unsafeAtomicAdd(o + 2, c.b);
}

struct PixelSlot { int x, y; uint32_t out_idx; bool valid; uint32_t tap; };

// the 8 x 8 unit that pixel slot `slot` of the launch lies in (see Params::unit0 / unit_step)
QD_DEV uint32_t unit_of_slot(const Params& P, uint32_t slot) { return P.unit0 + (slot / kUnitPx) * (P.unit_step ? P.unit_step : 1u); }

QD_DEV PixelSlot map_pixel(const Params& P, uint32_t unit, uint32_t p) {
    PixelSlot s;
    s.tap = 0;
    const int lx8 = (int)(p % (uint32_t)kUnitSide), ly8 = (int)(p / (uint32_t)kUnitSide);
    if (P.mode == 2) {
        const uint32_t j = unit * kUnitPx + p;
        s.valid = j < P.n_samples;
        const uint32_t xy = s.valid ? P.aa_pixels[j >> 2] : 0u;
        s.x = (int)(xy & 0xffffu); s.y = (int)(xy >> 16);
        s.tap = 1u + (j & 3u);
        s.out_idx = j;
    } else if (P.mode == 0) {
        const int ux = unit % P.units_x, uy = unit / P.units_x;
        const int px = ux * kUnitSide + lx8, py = uy * kUnitSide + ly8;
        s.x = P.x0 + px; s.y = P.y0 + py;
        s.valid = px < P.rw && py < P.rh;
        s.out_idx = (uint32_t)py * (uint32_t)P.rw + (uint32_t)px;
    } else {
        const int sub_per_side = P.tile / kUnitSide, spt = sub_per_side * sub_per_side;
        const int k = unit / spt, sb = unit % spt;
        const int t = P.phase + k * P.stride;
        const int tx = t % P.tiles_x, ty = t / P.tiles_x;
        const int lx = (sb % sub_per_side) * kUnitSide + lx8, ly = (sb / sub_per_side) * kUnitSide + ly8;
        s.x = tx * P.tile + lx; s.y = ty * P.tile + ly;
        s.valid = s.x < P.W && s.y < P.H;
        s.out_idx = (uint32_t)k * (uint32_t)(P.tile * P.tile) + (uint32_t)(ly * P.tile + lx);
    }
    return s;
}

// Persistent lanes: a lane that finishes its pixel immediately takes the next pixel of the wave's
// current 8x8 unit (or of a freshly fetched unit), so no lane waits for the slowest pixel of a tile.
// RICH: bitmap / Mandelbrot / procedural textures or bump maps present (see tex_sample); GEOM: see trace_nodes.
// FLAT: the node loop and the shaders are inlined into the kernel: the hit record and the IntersectionInfo stay in registers
// instead of crossing two calls through scratch memory.  It runs 3 waves per SIMD (168 registers, 12 far-child stack levels in
// LDS).  4 waves per SIMD (128 registers, 8 levels in LDS so that 16 waves' LDS fits a CU) measure 2 - 3 % faster in flight (10.62
// against 10.97 ms per frame) but spill: HBM traffic per frame 21.7 -> 33.2 GB (writes 5.0 -> 9.7 GB, PMC), 1.9 x the algorithmic
// bytes instead of 1.26 x -- not worth it; 2 waves are 26 % slower.  Pays for the copy without heightfield / CSG / extra-primitive / texture code (meshes + planes:
// +4.5 % on the 10 M triangle scene); the copies with more code in them are 20 - 50 % slower that way and keep the calls.
// It contains no call at all -- not even the rare one for a Layered shader inside a layer, which alone costs 6 % -- so
// launch() picks it only for scenes without such nesting.
#ifndef QD_FLAT_WAVES
#define QD_FLAT_WAVES 3
#endif
#ifndef QD_FLAT_LDS_STACK
#define QD_FLAT_LDS_STACK 12
#endif
constexpr int kFlatWavesPerEU = QD_FLAT_WAVES, kFlatLdsStack = QD_FLAT_LDS_STACK;
// the per-lane ray queues are allocated for n_CU x 4 x QD_WAVES_PER_EU waves: a copy of the kernel must never start more
static_assert(QD_FLAT_WAVES >= 1 && QD_FLAT_WAVES <= QD_WAVES_PER_EU, "the call-free kernel copy cannot run more waves per SIMD than the queues are allocated for");
template <bool STATS, int GEOM, bool RICH, bool FLAT = false>
__global__ __launch_bounds__(64, FLAT ? kFlatWavesPerEU : QD_WAVES_PER_EU) void trace_kernel(const DScene S, const Params P) {
    if (P.cond && __hip_atomic_load(P.cond, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;      // a fallback launch with nothing to fall back for
    const uint32_t lane = threadIdx.x;
    const size_t gl = (size_t)blockIdx.x * 64 + lane;
    LaneQueue q;
    q.base = P.stack + gl; q.stride = P.n_lanes; q.cap = P.stack_cap; q.sp = 0; q.err = P.error_flag;
    Counters cnt; cnt.kd_nodes = cnt.tris = cnt.mesh_queries = 0; cnt.t_desc = cnt.t_leaf = cnt.t_a = cnt.t_b = cnt.t_c = cnt.t_d = 0;
    uint32_t n_rays = 0, n_primary = 0, n_shadow = 0, n_killed = 0;
    unsigned long long t_trace = 0, t_shade = 0, t_sched = 0;
    const unsigned long long t_begin = QD_STAMP();

    bool pool_empty = false;
    uint32_t rounds_left = P.grace;            // counts down once the pass's work is all handed out
    uint32_t win_tasks = 0;                    // tasks this wave traced during those rounds
    uint32_t busy_windows = 0;
    unsigned long long t_pool_empty = 0, dbg_rounds = 0, dbg_tasks = 0, dbg_rounds_after = 0, dbg_tasks_after = 0;   // QD_PROFILE diagnostics
    const uint64_t* wave_base = P.stack + (size_t)blockIdx.x * 64;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;

    for (;;) {
        const unsigned long long tp0 = QD_STAMP();
        // 1. lanes whose queue is empty take the next pixels of the frame: one atomic per wave and round, pixel
        //    granularity, so a region of heavy pixels (hundreds of rays each behind glass) is spread over many waves
        unsigned long long idle = __ballot(q.sp == 0);
        while (idle != 0ull && !pool_empty) {
            uint32_t first = 0, got = 0;
            if (lane == 0) got = fetch_slots(P, blockIdx.x & 7u, (uint32_t)__popcll(idle), first);
            got = __shfl(got, 0, 64); first = __shfl(first, 0, 64);
            if (got == 0u) {
                pool_empty = true; t_pool_empty = QD_STAMP();
                if (lane == 0) __hip_atomic_store(P.pass_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            const uint32_t rank = __popcll(idle & lanes_below);
            if (q.sp == 0 && rank < got && P.source != 0) {            // a task the pass before left behind
                const uint64_t* e = P.in_list + (first + rank);
                for (int w = 0; w < kEntryWords; ++w) q.base[(size_t)w * q.stride] = e[(size_t)w * P.list_cap];
                q.sp = 1;
            } else if (q.sp == 0 && rank < got) {
                const uint32_t slot = first + rank;
                const PixelSlot px = map_pixel(P, unit_of_slot(P, slot), slot % kUnitPx);
                if (px.valid) {
                    // Camera::GetScreenRay, src/camera.cpp:26-41 (integer pixel coordinates, no +0.5)
                    const DV TL = ld3(P.cam.top_left), TR = ld3(P.cam.top_right), BL = ld3(P.cam.bottom_left), pos = ld3(P.cam.position);
                    // anti-aliasing taps: kernel offsets of src/main.cpp:108-114
                    const double kx = (px.tap == 1u || px.tap == 4u) ? 0.6 : (px.tap == 3u ? 0.3 : 0.0);
                    const double ky = (px.tap == 2u || px.tap == 4u) ? 0.6 : (px.tap == 3u ? 0.3 : 0.0);
                    const DV through = TL + (TR - TL) * (((double)px.x + kx) / P.W) + (BL - TL) * (((double)px.y + ky) / P.H);
                    Task t;
                    t.o = pos; t.d = normalize(through - pos);
                    t.w = mkc(1.f, 1.f, 1.f);
                    t.flags_depth = kFlagPrimary;
                    t.path = pixel_path(px.x, px.y, px.tap);
                    t.aux0 = t.aux1 = 0;
                    t.pixel = px.out_idx;
                    q.push(t);
                    ++n_primary;
                }
            }
            idle = __ballot(q.sp == 0);
        }
        // 1b. end of the pass.  Work is handed out by pixels, but a pixel on a glass object is a tree of up to thousands of
        //     rays and all of them stay in the wave that took the pixel: once the pixels are gone a few overloaded waves
        //     would run on alone (reference scene kdtree_test: one wave 650 rounds, the median wave 9).  So every wave
        //     -- busy ones included, they look at the flag each round -- runs `grace` more rounds, then appends its
        //     queued tasks to a list and leaves; the next pass (a new launch of this kernel) deals that list out over
        //     all the waves, and so on with ever shallower trees.  The last pass runs to the end.
        if (P.grace != 0xffffffffu) {
            if (!pool_empty) {
                uint32_t done = 0;
                if (lane == 0) done = __hip_atomic_load(P.pass_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__shfl(done, 0, 64) != 0u) pool_empty = true;
            }
            if (pool_empty) {
                if (rounds_left == 0u) {
                    uint32_t incl = (uint32_t)q.sp;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
                    const uint32_t total = __shfl(incl, 63, 64);
                    // Handing tasks over only pays for a wave that is short of lanes, not of work: one that has kept (nearly)
                    // all its lanes busy since the pixels ran out.  A wave whose rounds are thin is walking the last
                    // generations of its ray trees -- a chain of dependencies that another wave could not shorten -- and a
                    // new pass would only make the tasks wait for this one to end.  It looks again after `grace` rounds.
                    // ... and only if there is somebody to take them: three quarters of the pass's waves have left already.
                    uint32_t gone = 0;
                    if (lane == 0) gone = __hip_atomic_load(P.pass_done + kCtlStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gone = __shfl(gone, 0, 64);
                    // ... and stays that way: kBusyWindows windows in a row (a wave that is merely finishing the last few
                    // rounds of its trees is faster alone than through a new pass)
                    busy_windows = (win_tasks >= kBusyLanes * P.grace && gone * 4u >= gridDim.x * 3u) ? busy_windows + 1u : 0u;
                    if (busy_windows < kBusyWindows) { rounds_left = P.grace; win_tasks = 0; goto keep_running; }
                    // one region of the list per XCD label (eight counters, a cache line each); a region holds the full
                    // queues of all the waves with that label, so it cannot overflow
                    const uint32_t region = blockIdx.x & 7u;
                    uint32_t at = 0;
                    if (lane == 0 && total != 0u) at = atomicAdd(&P.out_count[region * kCtlStride], total);
                    at = __shfl(at, 0, 64) + region * (uint32_t)(P.list_cap / 8) + incl - (uint32_t)q.sp;
                    for (int e = 0; e < q.sp; ++e)
                        for (int w = 0; w < kEntryWords; ++w)
                            P.out_list[(size_t)w * P.list_cap + at + e] = q.base[((size_t)e * kEntryWords + w) * q.stride];
                    q.sp = 0;
                    break;
                }
                --rounds_left;
            }
        }
        keep_running:
        // 2. every lane with work takes its newest task
        bool have = q.sp > 0;
#ifndef QD_TASK_IN_SCRATCH
        // the task of the round lives in LDS (88 B per lane next to the 4 KB KD stack: 9.5 KB per wave, 16 waves per CU fit
        // in 160 KB): it is handed to trace_nodes and the shaders by reference, and a stack copy would be written to
        // scratch and read back every round (+3.3 % on the 10 M triangle scene)
        __shared__ Task lds_task[64];
        Task& t = lds_task[lane];
#else
        Task t;
#endif
        if (have) t = q.pop();
        // 3. lanes left without work (no pixels left to hand out) take queued tasks of other lanes of the wave: the ray
        //    tree of one heavy pixel (hundreds of rays behind glass) spreads over the wave instead of running on one lane
        const unsigned long long m_need = __ballot(!have);
        if (m_need != 0ull) {
            // every queued task of the wave is on offer: the r-th lane in need takes the r-th task counted over the
            // lanes' queues from their tops down (inclusive scan of the queue depths, then a search in it)
            const uint32_t mine = (uint32_t)q.sp;
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
            const uint32_t total = __shfl(incl, 63, 64);
            if (total != 0u) {
                const uint32_t n_need = (uint32_t)__popcll(m_need);
                const uint32_t n_take = n_need < total ? n_need : total;
                const uint32_t r = (uint32_t)__popcll(m_need & lanes_below);
                const bool takes = !have && r < n_take;
                uint32_t lo = 0, hi = 63;                      // smallest lane whose inclusive count exceeds r
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t v = __shfl(incl, (int)mid, 64);
                    if (v > r) hi = mid; else lo = mid + 1;
                }
                const uint32_t src = takes ? lo : lane;
                const uint32_t src_incl = __shfl(incl, (int)src, 64), src_sp = __shfl(mine, (int)src, 64);
                if (takes) {
                    const uint32_t k = r - (src_incl - src_sp);                   // k-th from the donor's top
                    t = LaneQueue::load(wave_base + src, q.stride, (int)(src_sp - 1u - k));
                    have = true;
                }
                const uint32_t excl = incl - mine;
                const uint32_t gives = n_take > excl ? (n_take - excl < mine ? n_take - excl : mine) : 0u;
                q.sp -= (int)gives;
            }
        }
        if (__ballot(have) == 0ull) break;         // pool empty and no task left in this wave
        if (pool_empty) win_tasks += (uint32_t)__popcll(__ballot(have));
#ifdef QD_PROFILE
        { const unsigned long long n = (unsigned long long)__popcll(__ballot(have));
          dbg_rounds += 1; dbg_tasks += n; if (pool_empty) { dbg_rounds_after += 1; dbg_tasks_after += n; } }
#endif
        if (!have) continue;

        Col acc = mkc(0, 0, 0);                    // this task's own colour
        do {
        const bool shadow = (t.flags_depth & kFlagShadow) != 0;
        const uint32_t depth = t.flags_depth & kDepthMask;
        if (!shadow && depth > P.max_depth) { ++n_killed; break; }             // src/main.cpp:20-21
        if (shadow) ++n_shadow; else ++n_rays;
        GHit h; double dw; float transp;
        const unsigned long long tp1 = QD_STAMP();
        t_sched += tp1 - tp0;
        int node;
        if constexpr (FLAT) node = trace_nodes_body<STATS, GEOM, kFlatLdsStack>(S, t.o, t.d, shadow, t.aux0, transp, dw, h, cnt);
        else node = trace_nodes<STATS, GEOM>(kernarg_scene(), t.o, t.d, shadow, t.aux0, transp, dw, h, cnt);
        const unsigned long long tp2 = QD_STAMP();
        t_trace += tp2 - tp1;
        if (shadow) {
            acc = acc + t.w * (float)((double)transp * t.aux1);
            break;
        }
        if ((t.flags_depth & kFlagPrimary) && P.hit_node) {
            P.hit_node[t.pixel] = node;
            P.hit_tri[t.pixel] = node >= 0 ? h.tri : -1;
            P.hit_dist[t.pixel] = node >= 0 ? dw : 0.;
        }
        if (node < 0) {
            if (S.env_present) acc = acc + t.w * env_lookup(S, t.d);
            break;
        }
        const qd_node& n = S.nodes[node];
        Info info;
        assemble_info<RICH>(S, n, h, dw, t.d, info);
        if (RICH && n.bump_tex >= 0) tex_modify_normal(S, n.bump_tex, info);
        const qd_shader top = S.shaders[n.shader];
        if (top.type == QD_SH_LAYERED) {
            // Layered::Shade (src/shading.cpp:260-275): result = b*layer + (1-b)*result, i.e. layer i carries
            // b_i * prod_{j>i}(1-b_j); walk the layers from the last to the first keeping that suffix product.
            Col suffix = mkc(1.f, 1.f, 1.f);
            for (int i = top.i1 - 1; i >= 0; --i) {
                const qd_layer l = S.layers[top.i0 + i];
                Col blend = ldc(l.blend);
                if (l.tex >= 0) blend = blend * tex_sample<RICH>(S, l.tex, info);
                if constexpr (FLAT) eval_leaf_shader_body<false, RICH>(S, P, kernarg_scene(), kernarg_params(), l.shader, t.w * (blend * suffix), (uint32_t)i + 1, t, depth, info, acc, q);
                else eval_leaf_shader<true, RICH>(kernarg_scene(), kernarg_params(), l.shader, t.w * (blend * suffix), (uint32_t)i + 1, t, depth, info, acc, q);
                suffix = suffix * (mkc(1.f, 1.f, 1.f) - blend);
            }
        } else {
            if constexpr (FLAT) eval_leaf_shader_body<false, RICH>(S, P, kernarg_scene(), kernarg_params(), n.shader, t.w, 0, t, depth, info, acc, q);
            else eval_leaf_shader<true, RICH>(kernarg_scene(), kernarg_params(), n.shader, t.w, 0, t, depth, info, acc, q);
        }
        t_shade += QD_STAMP() - tp2;
        } while (false);
        add_px(P, t.pixel, acc);
    }
    if (lane == 0 && P.grace != 0xffffffffu) atomicAdd(P.pass_done + kCtlStride, 1u);     // this wave has left the pass
#ifdef QD_PROFILE
    if (P.stats) { atomicAdd(&P.stats[12], cnt.t_desc); atomicAdd(&P.stats[13], cnt.t_leaf); }   // lane / wave iterations of the KD loop
    if (P.stats && lane == 0) {   // diagnostic build: wave-cycle shares go to their own words, never into outputs
        const unsigned long long t_end = QD_STAMP();
        { unsigned long long* w = P.stats + 24 + 6 * (size_t)(blockIdx.x & 4095u);
          w[0] = t_end - t_begin; w[1] = t_pool_empty ? t_pool_empty - t_begin : 0ull; w[2] = dbg_rounds; w[3] = dbg_tasks; w[4] = dbg_rounds_after; w[5] = dbg_tasks_after; }
        atomicMin(&P.stats[16], t_begin); atomicMax(&P.stats[17], t_end); atomicMax(&P.stats[18], t_pool_empty);
        atomicMin(&P.stats[19], t_pool_empty ? t_pool_empty : ~0ull); atomicAdd(&P.stats[20], t_end - t_begin);
        atomicAdd(&P.stats[8], t_end - t_begin);
        atomicAdd(&P.stats[9], t_sched); atomicAdd(&P.stats[10], t_trace); atomicAdd(&P.stats[11], t_shade);
        atomicAdd(&P.stats[14], cnt.t_a); atomicAdd(&P.stats[15], cnt.t_b); atomicAdd(&P.stats[7], cnt.t_c); atomicAdd(&P.stats[6], cnt.t_d);
    }
#endif

    if (P.stats) {
        const uint32_t r = wave_sum(n_rays), p = wave_sum(n_primary), s = wave_sum(n_shadow), k = wave_sum(n_killed);
        uint32_t a = 0, b = 0, c = 0;
        if (STATS) { a = wave_sum(cnt.kd_nodes); b = wave_sum(cnt.tris); c = wave_sum(cnt.mesh_queries); }
        if (lane == 0) {
            if (r + s) atomicAdd(&P.stats[0], (unsigned long long)r + s);
            if (p) atomicAdd(&P.stats[1], (unsigned long long)p);
            if (s) atomicAdd(&P.stats[2], (unsigned long long)s);
            if (k) atomicAdd(&P.stats[3], (unsigned long long)k);
            if (STATS) {
                atomicAdd(&P.stats[4], (unsigned long long)a);
                atomicAdd(&P.stats[5], (unsigned long long)b);
                atomicAdd(&P.stats[6], (unsigned long long)c);
            }
        }
    }
}

#include "wavefront.h"
#include "wfsort_lab.h"

// leaf_verts[ref] = tri_verts[leaf_tris[ref]] (9 doubles each): one pass at scene creation
__global__ void gather_leaf_verts_kernel(const double* __restrict__ tri_verts, const uint32_t* __restrict__ leaf_tris,
                                         double* __restrict__ leaf_verts, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const size_t ref = i / 9, w = i % 9;
    leaf_verts[i] = tri_verts[(size_t)leaf_tris[ref] * 9 + w];
}

__global__ void untile_kernel(const float* __restrict__ src, float* __restrict__ dst, int W, int H, int tile, int stride, int tpr) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const int tiles_x = (W + tile - 1) / tile;
    const int t = (y / tile) * tiles_x + (x / tile);
    const int r = t % stride, k = t / stride;
    const size_t s = (((size_t)r * tpr + k) * tile * tile + (size_t)(y % tile) * tile + (x % tile)) * 3;
    const size_t d = ((size_t)y * W + x) * 3;
    dst[d] = src[s]; dst[d + 1] = src[s + 1]; dst[d + 2] = src[s + 2];
}

// ConvertTo8Bit (src/utils.h:22,34): floor(clamp(x,0,1)*255 + 0.5); rows bottom-up, B,G,R (src/bitmap.cpp:209-217)
__global__ void quantize_kernel(const float* __restrict__ rgb, uint8_t* __restrict__ out, int W, int H, size_t row_bytes) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const float* p = rgb + ((size_t)y * W + x) * 3;
    uint8_t* o = out + (size_t)(H - 1 - y) * row_bytes + (size_t)x * 3;
    auto q8 = [](float v) { const double c = fmin(fmax((double)v, 0.0), 1.0); return (uint8_t)(int)floor(c * 255.0 + 0.5); };
    o[0] = q8(p[2]); o[1] = q8(p[1]); o[2] = q8(p[0]);
}

// Color::toRGB32 / toSRGB32 (src/color.cpp:16-36) as DisplayVFBRect applies them per pixel (src/sdl.cpp:302-322):
// ConvertTo8Bit or ConvertTo8Bit_sRGB (src/utils.h:34-35) per channel, packed with the surface's channel shifts.
__global__ void pack_rgb32_kernel(const float* __restrict__ rgb, size_t pitch, uint32_t* __restrict__ out, int W, int H, int srgb,
                                  int rs, int gs, int bs) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const float* p = rgb + (size_t)y * pitch + (size_t)x * 3;
    auto q8 = [srgb](float v) {
        double c = (double)v;
        if (srgb) c = c <= 0.0031308 ? c * 12.92 : 1.055 * pow(c, 1 / 2.4) - 0.055;
        c = fmin(fmax(c, 0.0), 1.0);
        return (uint32_t)(int)floor(c * 255.0 + 0.5);
    };
    out[(size_t)y * W + x] = (q8(p[2]) << bs) | (q8(p[1]) << gs) | (q8(p[0]) << rs);
}

// ------------------------------------------------------------------------------------------ anti-aliasing pass
// RenderScene's second pass (src/main.cpp:106-153) and IsTooDifferent (:58-83).  The reference walks the
// buckets in display order; for each bucket it first marks the pixels that differ from a neighbour, then replaces
// every marked pixel by the average of its first-pass colour and four more rays, IN PLACE.  A pixel on a bucket
// border therefore sees neighbours in earlier buckets already anti-aliased.  Here:
//   aa_mark_kernel    : marks with first-pass colours only, and lists every pixel that is marked or that has a
//                       neighbour in an earlier bucket (only those can still become marked) -> taps are traced
//   aa_combine_kernel : ((((first + t1) + t2) + t3) + t4) / 5 in the reference's order of float additions
//   aa_resolve_kernel : the mark / replace rule with "earlier bucket -> current value, else first-pass value";
//                       iterated to its fixed point, which is the sequential result because the dependency
//                       (earlier buckets only) has no cycles.
QD_DEV int aa_bucket_order(int x, int y, int W) {          // GetBucketList, src/sdl.cpp:253-276: 48 x 48, serpentine rows
    const int bw = (W - 1) / 48 + 1, bx = x / 48, by = y / 48;
    return by * bw + ((by & 1) ? bw - 1 - bx : bx);
}
constexpr int kVfb = 1920;                                  // VFB_MAX_SIZE: neighbours beyond it are skipped, inside it but
                                                            // outside the frame they read as black (static vfb, src/main.cpp:15)
// cur == nullptr: every neighbour is read from `first`
QD_DEV bool aa_too_different(const float* first, const float* cur, int x, int y, int W, int H, double threshold) {
    const int dx[8] = {0, 0, -1, 1, -1, -1, 1, 1}, dy[8] = {1, -1, 0, 0, -1, 1, 1, -1};
    const float* mine = first + ((size_t)y * W + x) * 3;
    const int my_order = aa_bucket_order(x, y, W);
    bool out = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned nx = (unsigned)x + dx[i], ny = (unsigned)y + dy[i];
        if ((unsigned)kVfb <= nx || (unsigned)kVfb <= ny) continue;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if ((int)nx < W && (int)ny < H) {
            const bool earlier = cur && aa_bucket_order((int)nx, (int)ny, W) < my_order;
            const float* q = (earlier ? cur : first) + ((size_t)ny * W + nx) * 3;
            o0 = q[0]; o1 = q[1]; o2 = q[2];
        }
        double diff = 0;
        diff += fabsf(mine[0] - o0); diff += fabsf(mine[1] - o1); diff += fabsf(mine[2] - o2);
        if (diff > threshold) out = true;
    }
    return out;
}

__global__ void aa_mark_kernel(const float* __restrict__ first, int W, int H, int adaptive, double threshold,
                               int32_t* __restrict__ slot, uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    bool cand = !adaptive || aa_too_different(first, nullptr, x, y, W, H, threshold);
    if (!cand) {
        const int my_order = aa_bucket_order(x, y, W);
        for (int dy = -1; dy <= 1 && !cand; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = x + dx, ny = y + dy;
                if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                if (aa_bucket_order(nx, ny, W) < my_order) { cand = true; break; }
            }
    }
    int32_t sl = -1;
    if (cand) { sl = (int32_t)atomicAdd(count, 1u); list[sl] = (uint32_t)x | ((uint32_t)y << 16); }
    slot[(size_t)y * W + x] = sl;
}

// The list comes out of aa_mark_kernel in the order the atomics landed; sorted by (y, x) it is a function of the frame
// alone, so that every rank of a multi-GPU pass cuts the same slices.  This re-derives the pixel -> list index map.
__global__ void aa_slot_kernel(const uint32_t* __restrict__ list, uint32_t n, int W, int32_t* __restrict__ slot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t xy = list[i];
    slot[(size_t)(xy >> 16) * W + (xy & 0xffffu)] = (int32_t)i;
}

__global__ void aa_combine_kernel(const float* __restrict__ first, const float* __restrict__ taps, const uint32_t* __restrict__ list,
                                  uint32_t n, int W, float* __restrict__ aa) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t xy = list[j];
    const float* f = first + ((size_t)(xy >> 16) * W + (xy & 0xffffu)) * 3;
    Col r = mkc(f[0], f[1], f[2]);
    for (int t = 0; t < 4; ++t) { const float* q = taps + ((size_t)j * 4 + t) * 3; r = r + mkc(q[0], q[1], q[2]); }
    r = r / (float)double(5);
    aa[(size_t)j * 3] = r.r; aa[(size_t)j * 3 + 1] = r.g; aa[(size_t)j * 3 + 2] = r.b;
}

__global__ void aa_resolve_kernel(const float* __restrict__ first, const float* __restrict__ cur, float* __restrict__ next,
                                  const float* __restrict__ aa, const int32_t* __restrict__ slot, int W, int H, int adaptive, double threshold,
                                  int show, float dbg_r, float dbg_g, float dbg_b, int* __restrict__ changed, int* __restrict__ error_flag) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t i = ((size_t)y * W + x) * 3;
    const bool need = !adaptive || aa_too_different(first, cur, x, y, W, H, threshold);
    float r = first[i], g = first[i + 1], b = first[i + 2];
    if (need) {
        const int32_t sl = slot[(size_t)y * W + x];
        if (show) { r = dbg_r; g = dbg_g; b = dbg_b; }
        else if (sl < 0) *error_flag = 2;                   // cannot happen: the listed set covers every pixel that can be marked
        else { r = aa[(size_t)sl * 3]; g = aa[(size_t)sl * 3 + 1]; b = aa[(size_t)sl * 3 + 2]; }
    }
    if (r != cur[i] || g != cur[i + 1] || b != cur[i + 2]) *changed = 1;
    next[i] = r; next[i + 1] = g; next[i + 2] = b;
}

// ------------------------------------------------------------------------------------------ static scratch
// Working memory of qd_render_tiles_noalloc_async: control words + per-lane ray queues in a file-scope device array, so
// that the call allocates nothing.  One copy per GPU, held until the process ends; every call on a GPU shares it.
// In host code the array's NAME is only a stand-in: its device address is looked up once per device in qd_init
// (hipGetSymbolAddress) and kept in g_scratch_ptr[device]; host code uses that pointer, never the name.
constexpr size_t kScratchBytes = (size_t)96 << 20;        // 96 MiB
constexpr size_t kScratchCtlBytes = 4096;                 // control words of one pass (kPassWords * 4 = 2304 B), then the queues
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;

// ------------------------------------------------------------------------------------------ host side
thread_local std::string g_err;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};     // device address of `scratch` on each device qd_init has bound
int g_device = -1;
int g_num_cu = 0;
bool g_counting = false;
int g_pipeline = 2;           // qd_set_pipeline: 0 = one persistent kernel with per-lane LIFO queues, 1 = wavefront pipeline (wavefront.h), 2 = the faster of the two, measured per scene
int g_refine_leaf = 2;        // qd_set_kd_refine
double g_refine_cost = 0.6;
constexpr uint64_t kWfDefaultBudget = 16ull << 30;
uint64_t g_wf_budget = kWfDefaultBudget;   // qd_set_wavefront_budget: bytes of ray queues per working set (wavefront form); for tests, a cap on the
uint32_t g_wf_max_entries = 0;        // entries of a queue that the band sizes do NOT take into account (0 = none): bands then overflow
int g_kd_build = 1;           // qd_set_kd_build: 0 = the caller's planes, leaves refined; 1 = whole tree by the surface-area heuristic, built on the device (kdbuild.h); 2 = the same by host threads (kdrefine.h)

int set_err(const std::string& m) { g_err = m; return -1; }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

}  // namespace

// Working memory of one frame in flight: the per-lane ray queues, the two task lists handed from pass to pass and the
// control words.  A scene holds up to kMaxWorkSets of them, each bound to the HIP stream that used it last, so frames
// rendered on different streams overlap on the GPU: the tail of one frame (few waves walking the last generations of
// their ray trees) runs beside the start of the next.  Launches on one stream share a set and serialise as before.
constexpr int kMaxWorkSets = 8;              // streams beyond the process's hardware queues (GPU_MAX_HW_QUEUES, 4 by default) share one and serialise
constexpr uint32_t kMaxFrameParts = 4;
constexpr int kErrAa = 8, kErrNoAlloc = 9, kErrFlags = 10;     // error words beyond the kMaxWorkSets per-set ones
struct WorkSet {
    bool bound = false;
    int last_form = 0;                       // scheduling form of the set's last launch
    hipStream_t stream = nullptr;
    uint32_t stack_cap = 0;
    uint64_t* stack = nullptr;
    uint32_t* queue_ctr = nullptr;           // per pass: 8 work-queue counters, 8 list counts, done flag, one cache line each
    uint64_t* lists[2] = {nullptr, nullptr}; // task lists handed from pass to pass (ping-pong), each as large as all the queues
    size_t list_cap = 0;
    hipEvent_t idle = nullptr;               // recorded behind the set's last launch; a stream that takes the set over waits on it
    uint64_t last_use = 0;
    // wavefront pipeline: ray queues of two alternating generations, hit records, control words, deep KD stack levels
    uint64_t* wf_ray[2] = {nullptr, nullptr};
    uint64_t* wf_res = nullptr;
    uint32_t* wf_ctl = nullptr;
    uint2* wf_deep = nullptr;
    size_t wf_cap = 0;
    uint32_t wf_gens = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // around the set's latest frame (adaptive choice of the scheduling form; t1 also tells
                                             // whether the set still has work in flight)
};

// Adaptive choice of the scheduling form (qd_set_pipeline 2).  The figure per form is the latency of a frame that ran ALONE
// on the GPU: a frame that was enqueued when every working set of the scene had drained and that had finished before the
// next one was enqueued.  Such frames are what a caller that waits for each frame produces anyway, and what
// FrameRenderer.settle() produces on purpose before frames are put in flight; frames that overlap others are not timed
// (their durations stretch each other, and block-wise throughput figures turned out to swing by +-30 % with the order in
// which overlapping frames happen to finish).  Lone latency ranks the two forms like in-flight throughput on every
// workload measured except lecture7 at 4K (alone 8.3 vs 9.1 ms, in flight 8.0 vs 7.7 ms: a 4 % loss).
struct FormSample {
    bool pending = false;
    int form = 0;
    uint64_t slots = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // the working set's events (not owned)
};

// State of an AA pass between its three steps (mark / trace the taps / combine + resolve), see aa_begin.
struct AaPlan {
    int W = 0, H = 0;
    float* first = nullptr;        // copy of the first-pass frame
    int32_t* slot = nullptr;       // per pixel: index in the list or -1
    uint32_t* list = nullptr;      // x | y << 16 of every pixel whose taps are traced (+ one word for the count)
    uint32_t n_list = 0;
    bool traced = false;           // false: nothing listed, or showAA (marks are painted, no taps)
};

// Buffers the synchronous entry points stage their data in (qd_render's rectangle and hit planes, the anti-aliasing pass's
// frame copies, lists and flags, the frame of qd_frame_begin): held by the scene handle, grown when a call needs more,
// freed by qd_scene_destroy -- a loop of calls (the reference renders a frame bucket by bucket) allocates once.
enum { kStRgb, kStHitNode, kStHitTri, kStHitDist, kStAaFrame, kStAaTaps, kStAaFirst, kStAaSlot, kStAaList, kStAaSorted, kStAaTmp,
       kStAaA, kStAaB, kStAaAa, kStAaFlags, kStFrame, kStFrameStats, kStCount };
struct Staging {
    void* p[kStCount] = {};
    size_t bytes[kStCount] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;       // around the timed launches of a synchronous call
    hipEvent_t f0 = nullptr, f1 = nullptr;       // around the launches of qd_frame_begin (a qd_render may come between begin and read)
    void* host_frame = nullptr;                  // pinned copy of the qd_frame_begin frame
    size_t host_bytes = 0;
};

// A frame rendered ahead of a loop of rectangle reads (qd_frame_begin / qd_frame_read_rect / qd_frame_end).
struct FramePlan {
    int W = 0, H = 0;
    bool launched = false, fetched = false;
    hipStream_t stream = nullptr;
    Params P;
    uint32_t max_depth = 0;
    float ms = 0.f;
};

struct qd_scene {
    DScene d;
    AaPlan aa;
    Staging stg;
    FramePlan frame;
    std::vector<void*> allocs;
    uint32_t n_lanes = 0;
    WorkSet sets[kMaxWorkSets];
    uint64_t use_clock = 0;
    unsigned long long* stats = nullptr; // 8 counters
    int* error_flag = nullptr;           // kErrFlags device words: one per working set (its launches raise it), the AA rule's, the no-alloc call's
    std::mutex mu;                       // the handle's host state (working sets, plans, staging): one caller at a time
    uint32_t spawn0 = 1, spawn1 = 1;     // tasks queued per hit at depth 0 / deeper
    int geom_mask = 7;                   // kGeomHf | kGeomCsg | kGeomPrim | kGeomCsgDeep as present in the scene: selects the kernel copy
    bool rich = true;                    // bitmap / Mandelbrot / procedural textures or bump maps present
    bool nested_layered = false;         // some Layered shader is a layer of another one (the FLAT kernel copy has no code for that)
    bool small_mesh = false;             // some node's mesh has no KD tree (< 50 triangles): the wavefront shading kernel walks it through a call
    std::vector<int> kd_node_list;       // scene nodes whose geometry is a mesh with a KD tree: one walk launch each (wavefront pipeline)
    // adaptive choice of the scheduling form: nanoseconds per pixel slot measured for each form (0 = not tried yet)
    double form_ns[2] = {0., 0.};
    uint32_t form_samples[2] = {0, 0};   // lone frames timed per form (the first is a warm-up)
    uint32_t form_tries[2] = {0, 0};     // frames spent on a form that has no figure yet
    FormSample sample;                   // the latest frame, if it started alone
    bool serial_caller = false;          // the latest sample was accepted: the caller waits for its frames
    uint64_t form_frames = 0;
    int wf_level = 0;                    // sizing of the wavefront queues (wavefront_plan): 0 = one band, optimistic; 1 = by the shader graph, in bands
    bool wavefront_overflowed = false;   // a band of level 1 outgrew its queues: the scene stays on the one-kernel form
    uint32_t* host_flag = nullptr;       // two pinned host words the device sets when a band of level 0 / 1 overflows (read without waiting), and their device address
    uint32_t* host_flag_dev = nullptr;
    int last_form = 0;                   // form of the most recent launch (qd_scene_schedule_info)
    uint32_t last_parts = 1;             // concurrent parts the most recent launch was enqueued as (launch())
    bool prev_overlapped = false;        // the most recent launch was enqueued while earlier work of the scene was still in flight
    hipStream_t part_stream[kMaxFrameParts - 1] = {};   // the library's own streams and working sets for parts 1.. of a lone wavefront frame
    WorkSet part_sets[kMaxFrameParts - 1];              // (apart from `sets`: callers' streams never take them over, and the other way round)
    hipEvent_t part_fork = nullptr, part_join[kMaxFrameParts - 1] = {};
    uint64_t kd_nodes = 0, kd_refs = 0, kd_leaves = 0, kd_max_leaf = 0;   // device tree after refinement
    double t_kd_build = 0., t_upload = 0.;   // seconds of qd_scene_create spent building the device KD tree / uploading and gathering
};

namespace {

template <class T>
int upload(qd_scene* s, const T* src, size_t n, const T** dst) {
    *dst = nullptr;
    void* p = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T) + 64;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T*)p;
    return 0;
}

constexpr uint32_t kPassWords = 18 * kCtlStride;   // control words per pass, one cache line each: 8 work-queue counters, 8 list counts,
                                                   // done flag, count of waves that have left
#ifndef QD_PASSES
#define QD_PASSES 3
#endif
constexpr int kPasses = QD_PASSES;

void free_set(WorkSet& w) {
    if (w.stack) (void)hipFree(w.stack);
    if (w.queue_ctr) (void)hipFree(w.queue_ctr);
    for (int i = 0; i < 2; ++i) if (w.lists[i]) (void)hipFree(w.lists[i]);
    for (void* p : {(void*)w.wf_ray[0], (void*)w.wf_ray[1], (void*)w.wf_res, (void*)w.wf_ctl, (void*)w.wf_deep}) if (p) (void)hipFree(p);
    if (w.idle) (void)hipEventDestroy(w.idle);
    if (w.t0) (void)hipEventDestroy(w.t0);
    if (w.t1) (void)hipEventDestroy(w.t1);
    w = WorkSet();
}

// The working set for a launch on `stream`, sized for max_depth: the set the stream used before, else a free one, else
// the one used longest ago (the stream then waits for that set's last launch).  Allocates only when a set is new or
// too small; qd_scene_bind_stream does this ahead of time.
int size_set(qd_scene* s, WorkSet* w, hipStream_t stream, uint32_t max_depth);
int acquire_set(qd_scene* s, hipStream_t stream, uint32_t max_depth, WorkSet** out) {
    WorkSet* w = nullptr;
    for (auto& c : s->sets) if (c.bound && c.stream == stream) { w = &c; break; }
    if (!w) for (auto& c : s->sets) if (!c.bound) { w = &c; break; }
    if (!w) {
        w = &s->sets[0];
        for (auto& c : s->sets) if (c.last_use < w->last_use) w = &c;
        HIP_TRY(hipStreamWaitEvent(stream, w->idle, 0));
    }
    if (size_set(s, w, stream, max_depth)) return -1;
    *out = w;
    return 0;
}

// Binds `w` to `stream` and sizes its one-kernel queues for max_depth (allocates only when the set is new or too small).
int size_set(qd_scene* s, WorkSet* w, hipStream_t stream, uint32_t max_depth) {
    // LIFO bound: each processed task removes one slot and adds at most spawn tasks.
    const uint64_t cap64 = (uint64_t)s->spawn0 + (uint64_t)s->spawn1 * max_depth + 2;
    if (cap64 > 4096) return set_err("ray queue bound too large for this scene (samples x depth)");
    const uint32_t cap = (uint32_t)cap64;
    w->bound = true; w->stream = stream; w->last_use = ++s->use_clock;
    if (!w->idle) HIP_TRY(hipEventCreateWithFlags(&w->idle, hipEventDisableTiming));
    if (!w->queue_ctr) HIP_TRY(hipMalloc((void**)&w->queue_ctr, (size_t)kPasses * kPassWords * sizeof(uint32_t)));
    if (!w->stack || cap > w->stack_cap) {
        if (w->stack) { (void)hipFree(w->stack); w->stack = nullptr; }     // hipFree waits for the device: nothing still reads it
        HIP_TRY(hipMalloc((void**)&w->stack, (size_t)cap * kEntryWords * 8 * s->n_lanes));
        w->stack_cap = cap;
        for (int i = 0; i < 2; ++i) {                 // a list can take every queue filled to the brim: it cannot overflow
            if (w->lists[i]) { (void)hipFree(w->lists[i]); w->lists[i] = nullptr; }
            HIP_TRY(hipMalloc((void**)&w->lists[i], (size_t)cap * kEntryWords * 8 * s->n_lanes));
        }
        w->list_cap = (size_t)cap * s->n_lanes;
    }
    return 0;
}

// One frame (or rectangle, or sample list) = kPasses launches of the trace kernel on the stream: pass 0 hands out pixels,
// each later pass hands out the ray tasks the pass before left in its list (see 1b in the kernel), the last pass runs
// to the end.  A pass that finds an empty list is one launch of waves that leave at once (~20 us).
#ifndef QD_GRACE
#define QD_GRACE 4
#endif
#ifndef QD_LATER_DIV
#define QD_LATER_DIV 2
#endif
constexpr uint32_t kLaterPassDiv = QD_LATER_DIV;
constexpr uint32_t kGrace = QD_GRACE;

constexpr uint32_t kWfMaxGens = 67;          // generations a frame may have: maxTraceDepth + 2

// Ray queues of the wavefront form for a launch over n_slots pixel slots: entries per queue (`cap`) and pixel slots per
// band.  A generation holds at most (rays of the one before) x (tasks queued per hit); generation 1 is bounded by
// pixels x spawn0, later ones shrink in every scene measured (rays leave the scene faster than glass doubles them), so the
// shader graph asks for max(4, spawn0 + 1) entries per pixel slot.  Within 16 GB per working set:
//   level 0 (a scene's first frames): ONE band whatever the graph asks for -- what fits the budget, at least 8 entries per
//            pixel slot.  A 25-sample glossy sphere that covers a tenth of a 4K frame asks for 27 per pixel and fills 5; bands
//            would only add generation tails to such a frame (lecture7 at 4K: 6.0 against 7.0 ms);
//   level 1 (after a frame of level 0 outgrew its queues -- it was rendered again in place, see launch_wavefront): by the
//            shader graph, a frame that asks for more than the budget (a 25-sample glossy FLOOR at 4K: 28 entries per pixel,
//            41 GB) in BANDS of whole 8 x 8 units, one after the other on the stream, each with its own loop over the
//            generations and the same queues.  A band that overflows at this level takes the scene off the form.
// An explicit qd_set_wavefront_budget is a hard limit: level 1 from the start.
struct WfPlan { uint64_t cap; uint64_t band_slots; };
WfPlan wavefront_plan(const qd_scene* s, uint64_t n_slots, uint32_t share = 1) {      // share: the launch is one of `share` parts of a frame and gets that fraction of the budget
    const size_t n_kd = std::max<size_t>(1, s->kd_node_list.size());
    const uint64_t entry_bytes = 2ull * kEntryWords * 8 + (uint64_t)kWfResWords * 8 * n_kd;
    const uint64_t budget = std::max<uint64_t>(kUnitPx * 4, std::min<uint64_t>(g_wf_budget / std::max(1u, share) / entry_bytes, (1ull << 31) - 1));
    const uint64_t demand = std::max<uint64_t>(4, (uint64_t)s->spawn0 + 1);
    uint64_t want = n_slots * demand;
    // small launches (a bucket, a rectangle that lies on glass) get up to 32 entries per pixel while that stays below 8 M entries
    want = std::max<uint64_t>(want, std::min<uint64_t>(n_slots * 32, 8u << 20));
    want = std::max<uint64_t>(want, 1u << 16);
    WfPlan p;
    if (s->wf_level == 0 && std::max(budget, n_slots * 8) < (1ull << 31)) {
        p.cap = std::min(want, std::max(budget, n_slots * 8));
        p.band_slots = n_slots;
        return p;
    }
    p.cap = std::min(want, budget);
    p.band_slots = p.cap >= n_slots * demand ? n_slots : std::max<uint64_t>(kUnitPx, p.cap / demand / kUnitPx * kUnitPx);
    return p;
}

// words behind the per-generation control words of a working set: the overflow word (a cache line), then the band's counters
constexpr size_t kWfTail = (size_t)(kWfMaxGens + 1) * kWfGenWords;
constexpr size_t kWfTailWords = kCtlStride + 24 * 2;

// The wavefront form of a frame (wavefront.h): walk(0) shade(0) walk(1) ... walk(D + 1) shade(D + 1) on the stream.
// ... its buffers first (apart from the launches, so that the adaptive choice does not time an allocation):
int reserve_wavefront(qd_scene* s, WorkSet* w, const Params& P, uint32_t max_depth) {
    if (max_depth + 3 > kWfMaxGens) return set_err("maxTraceDepth too large for the wavefront pipeline");
    const size_t n_kd = std::max<size_t>(1, s->kd_node_list.size());
    const uint64_t cap64 = wavefront_plan(s, (uint64_t)P.n_units * kUnitPx, P.unit_step).cap;
    // (no clearing here: a memset on the null stream is not ordered with the caller's stream and could land in the middle of the
    // frame; every band clears the words it uses on its own stream)
    if (!w->wf_ctl) HIP_TRY(hipMalloc((void**)&w->wf_ctl, (kWfTail + kWfTailWords) * sizeof(uint32_t)));
    if (!w->wf_deep) HIP_TRY(hipMalloc((void**)&w->wf_deep, (size_t)(kKdStack - kLdsStack) * s->n_lanes * sizeof(uint2)));
    if (cap64 > w->wf_cap) {
        for (void** p : {(void**)&w->wf_ray[0], (void**)&w->wf_ray[1], (void**)&w->wf_res}) if (*p) { (void)hipFree(*p); *p = nullptr; }
        w->wf_cap = 0;
        for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc((void**)&w->wf_ray[i], (size_t)cap64 * kEntryWords * 8));
        HIP_TRY(hipMalloc((void**)&w->wf_res, (size_t)cap64 * kWfResWords * 8 * n_kd));
        w->wf_cap = (size_t)cap64;
    }
    return 0;
}

int launch_one_kernel(qd_scene* s, WorkSet* w, Params& P, uint32_t max_depth, hipStream_t stream, bool fallback = false);

// The queues cannot be outgrown unnoticed -- and not at the price of a frame either.  An append that finds its queue full
// sets the working set's overflow word and writes nothing.  Behind every band, on the stream: wf_band_end_kernel (no overflow:
// the band's counters go to the caller's; overflow: the band's pixels are zeroed) and ONE launch of the one-kernel form on
// the same band that returns at once unless the word is set (its per-lane queues are bounded by construction).  So the frame
// is complete and correct either way, nothing comes back to the host and no call fails; a band that overflowed cost its time
// twice, and the pinned host word keeps the scene on the one-kernel form from the next launch on.
// err_owner: the working set whose error word the launch reports to (a part of a frame: the set of the caller's stream);
// clear_frame: zero the output first (a part of a frame: launch() has done it once for all parts).
int launch_wavefront(qd_scene* s, WorkSet* w, Params& P, uint32_t max_depth, hipStream_t stream, WorkSet* err_owner = nullptr, bool clear_frame = true) {
    if (reserve_wavefront(s, w, P, max_depth)) return -1;          // (a no-op when launch() has reserved already)
    const uint64_t n_slots = (uint64_t)P.n_units * kUnitPx;
    const WfPlan plan = wavefront_plan(s, n_slots, P.unit_step);
    P.wf_ray[0] = w->wf_ray[0]; P.wf_ray[1] = w->wf_ray[1]; P.wf_res = w->wf_res; P.wf_ctl = w->wf_ctl; P.wf_deep = w->wf_deep;
    P.wf_cap = (uint32_t)w->wf_cap;
    P.wf_limit = g_wf_max_entries ? std::min<uint32_t>(P.wf_cap, std::max<uint32_t>(g_wf_max_entries, 64u)) : P.wf_cap;
    P.error_flag = s->error_flag + ((err_owner ? err_owner : w) - s->sets);
    P.wf_over = w->wf_ctl + kWfTail;
    P.host_flag = s->host_flag_dev + (s->wf_level ? 1 : 0);           // which level's frame it was that overflowed
    unsigned long long* const band_stats = (unsigned long long*)(w->wf_ctl + kWfTail + kCtlStride);
    unsigned long long* const user_stats = P.stats;
    if (clear_frame) HIP_TRY(hipMemsetAsync(P.rgb, 0, P.rgb_floats * sizeof(float), stream));
    using ShadeK = void (*)(const DScene, const Params, int);
    // call-free copies by scene content (see wf_nodes); a scene that needs a call -- a mesh too small for a tree, a Layered
    // shader inside a layer, CSG over CSG / mesh / heightfield -- runs the one copy that holds everything
#ifdef QD_DEV_BUILD      // quick compile for development: mask 0 and the copy with everything only
    static const ShadeK shade_plain[8] = {wf_shade_kernel<false, 0, false, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>,
                                          wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>};
    static const ShadeK shade_rich[8] = {wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>,
                                         wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>, wf_shade_kernel<false, 7, true, false>};
#else
    static const ShadeK shade_plain[8] = {wf_shade_kernel<false, 0, false, false>, wf_shade_kernel<false, 1, false, false>, wf_shade_kernel<false, 2, false, false>, wf_shade_kernel<false, 3, false, false>,
                                          wf_shade_kernel<false, 4, false, false>, wf_shade_kernel<false, 5, false, false>, wf_shade_kernel<false, 6, false, false>, wf_shade_kernel<false, 7, false, false>};
    static const ShadeK shade_rich[8] = {wf_shade_kernel<false, 0, true, false>, wf_shade_kernel<false, 1, true, false>, wf_shade_kernel<false, 2, true, false>, wf_shade_kernel<false, 3, true, false>,
                                         wf_shade_kernel<false, 4, true, false>, wf_shade_kernel<false, 5, true, false>, wf_shade_kernel<false, 6, true, false>, wf_shade_kernel<false, 7, true, false>};
#endif
    const bool needs_calls = (s->geom_mask & kGeomCsgDeep) != 0 || s->nested_layered || s->small_mesh;
    const ShadeK sk = g_counting ? wf_shade_kernel<true, kGeomAll, true, true> : needs_calls ? wf_shade_kernel<false, kGeomAll, true, true>
                                 : (s->rich ? shade_rich : shade_plain)[s->geom_mask & 7];
    const int lab = wf_sort_mode();
    if (lab && !g_lab.e0) { HIP_TRY(hipEventCreate(&g_lab.e0)); HIP_TRY(hipEventCreate(&g_lab.e1)); }
    if (lab) g_lab.frames++;
    const uint32_t max_blocks = s->n_lanes / 64;
    const uint32_t all_units = P.n_units, first_unit = P.unit0, step = P.unit_step ? P.unit_step : 1u;
    const uint32_t band_units = (uint32_t)(plan.band_slots / kUnitPx);
    for (uint32_t u = 0; u < all_units; u += band_units) {
        P.unit0 = first_unit + u * step;
        P.n_units = std::min(band_units, all_units - u);
        const uint64_t band_slots = (uint64_t)P.n_units * kUnitPx;
        const uint32_t grid0 = (uint32_t)std::min<uint64_t>(max_blocks, std::max<uint64_t>(1, (band_slots + 63) / 64));
        HIP_TRY(hipMemsetAsync(w->wf_ctl, 0, (size_t)(max_depth + 3) * kWfGenWords * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(w->wf_ctl + kWfTail, 0, kWfTailWords * sizeof(uint32_t), stream));
        P.stats = band_stats; P.user_stats = user_stats; P.cond = nullptr;
        for (uint32_t g = 0; g <= max_depth + 1; ++g) {
            if (lab) HIP_TRY(hipEventRecord(g_lab.e0, stream));
            for (size_t k = 0; k < s->kd_node_list.size(); ++k) {
                if (g_counting) hipLaunchKernelGGL(wf_walk_kernel<true>, dim3(grid0), dim3(64), 0, stream, s->d, P, (int)g, s->kd_node_list[k], (int)k);
                else hipLaunchKernelGGL(wf_walk_kernel<false>, dim3(grid0), dim3(64), 0, stream, s->d, P, (int)g, s->kd_node_list[k], (int)k);
                HIP_TRY(hipGetLastError());
            }
            hipLaunchKernelGGL(sk, dim3(grid0), dim3(64), 0, stream, s->d, P, (int)g);
            HIP_TRY(hipGetLastError());
            if (lab) {                                                  // experiment (wfsort_lab.h): time the generation, order the next one
                float ms = 0.f;
                HIP_TRY(hipEventRecord(g_lab.e1, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (hipEventElapsedTime(&ms, g_lab.e0, g_lab.e1) == hipSuccess) g_lab.ms_trace += ms;
                if (g <= max_depth) {
                    HIP_TRY(hipEventRecord(g_lab.e0, stream));
                    if (wf_lab_sort(P.wf_ray[(g + 1) & 1], (size_t)P.wf_cap, w->wf_ctl + (size_t)(g + 1) * kWfGenWords, P.wf_limit, stream)) return set_err("QD_WF_SORT: sort failed");
                    HIP_TRY(hipEventRecord(g_lab.e1, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    if (hipEventElapsedTime(&ms, g_lab.e0, g_lab.e1) == hipSuccess) g_lab.ms_sort += ms;
                }
            }
        }
        hipLaunchKernelGGL(wf_band_end_kernel, dim3((unsigned)((band_slots + 63) / 64)), dim3(64), 0, stream, P);
        HIP_TRY(hipGetLastError());
        P.stats = user_stats; P.cond = P.wf_over;                      // the fallback: does nothing unless the band overflowed
        if (launch_one_kernel(s, w, P, max_depth, stream, true)) return -1;
    }
    P.unit0 = first_unit; P.n_units = all_units; P.stats = user_stats; P.cond = nullptr;
    HIP_TRY(hipEventRecord(w->idle, stream));
    return 0;
}



// Adaptive choice (qd_set_pipeline 2, the default): which form is faster depends on the frame -- the wavefront form wins
// where rays have very different fates (a glass object in front of a backdrop: the reference's kdtree_test, 2x) and,
// by a few percent, on the heightfield / CSG scenes; the one-kernel form where every pixel carries similar work behind
// a deep tree (the 10 M triangle terrain, 1.5x).  So each form is timed on real frames of the scene that ran alone (see
// FormSample) and the faster one renders; a caller that waits for its frames also sends every 64th frame to the other
// form, so that a change of camera or settings is noticed.
constexpr uint32_t kFormTries = 4, kFormExploreEvery = 64;

// Settles the pending sample: accepted if its frame has finished by now (at_launch: before the next frame is enqueued).
void read_sample(qd_scene* s, bool at_launch) {
    FormSample& f = s->sample;
    if (!f.pending) return;
    if (hipEventQuery(f.t1) != hipSuccess) { if (at_launch) { f.pending = false; s->serial_caller = false; } return; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, f.t0, f.t1) == hipSuccess && ms > 0.f && f.slots) {
        const double ns = (double)ms * 1e6 / (double)f.slots;
        double& acc = s->form_ns[f.form];
        if (s->form_samples[f.form]++ == 0) {}             // a form's first frame pays for loading its kernels: not a figure
        else acc = acc == 0. ? ns : 0.5 * acc + 0.5 * ns;
        s->serial_caller = true;
    }
    f.pending = false;
}

// The form of a scene's first frames, before anything has been measured (a caller that renders one image never gets
// further).  At their own sizes the wavefront form renders 14 of the 17 fixture scenes faster (1.1x .. 4.5x, kdtree_test
// 2.6x; scripts/gpu_forms.py) and loses where a frame is launch-bound (hw9/axe_test, 0.5 against 1.0 ms) or the tree is
// deep (the 10 M triangle terrain, 46 M records: 0.65x).  So: wavefront, unless the tree is large or the shader graph asks
// for more queue entries per pixel than reserve_wavefront grants for the frame.
int prior_form(const qd_scene* s, uint64_t) {
    if (s->kd_nodes > (4u << 20)) return 0;
    return 1;
}

// The pinned words the device sets when a band overflows: level 0 -> size by the shader graph from now on; level 1 -> off the form.
// (An explicit queue budget or entry limit is a hard limit: level 1 at once.)
void read_overflow_words(qd_scene* s) {
    if (!s->host_flag) return;
    const volatile uint32_t* f = s->host_flag;
    if (s->wf_level == 0 && (f[0] || g_wf_budget != kWfDefaultBudget || g_wf_max_entries)) s->wf_level = 1;
    if (f[1]) s->wavefront_overflowed = true;
}

int choose_form(qd_scene* s, uint64_t n_slots) {
    if (g_pipeline != 2) return g_pipeline;
    if (s->kd_node_list.size() > kWfMaxKdNodes || s->wavefront_overflowed) return 0;
    const int first = prior_form(s, n_slots);
    for (int f : {first, 1 - first})
        if (s->form_ns[f] == 0. && s->form_tries[f] < kFormTries) { s->form_tries[f]++; return f; }
    if (s->form_ns[0] == 0. && s->form_ns[1] == 0.) return first;       // (a caller whose frames always overlap and who never settles)
    if (s->form_ns[0] == 0. || s->form_ns[1] == 0.) return s->form_ns[1] != 0. ? 1 : 0;
    const int best = s->form_ns[1] < s->form_ns[0] ? 1 : 0;
    if (s->serial_caller && ++s->form_frames % kFormExploreEvery == 0) return 1 - best;
    return best;
}

// A LONE frame of the wavefront form -- one enqueued while nothing else of the scene is in flight: what a caller that renders
// one image, or waits for each frame, produces -- is enqueued as kMaxFrameParts concurrent PARTS: part k renders the 8 x 8
// units k, k + 4, k + 8 ... on a stream and working set of its own (part 0 on the caller's), forked from and joined to the
// caller's stream with events, so for the caller it is still one enqueue-only launch on its stream.  Why: a generation's walk
// launch ends in a tail of a few long walks, its shade launch starts cold; with four sequences in flight one part's tails run
// under another part's bulk, which is what frames in flight give a caller who has several frames (scripts/r03_split.py:
// kdtree_test 10.0 -> 8.8 ms, lecture7 at 4K 8.1 -> 5.8 ms, kdtree_test at 4K 54.5 -> 42.8 ms).  The one-kernel form gains nothing
// (12.97 -> 12.81 ms: its waves persist across the frame) and stays one launch sequence; frames that are not alone stay whole too.
// So does the first frame of a burst that follows a burst: a caller who keeps F frames in flight gains nothing from parts (F
// sequences already overlap) and would lose -- seven launch sequences side by side are slower than four (lecture7 at 4K: the
// first four frames of a burst took 34 instead of 24 ms) -- so a frame is only split if the scene's previous frame was not
// enqueued behind another one either (or there was none): the caller looks like one who waits for each frame.
constexpr uint32_t kPartMinUnits = 12288;         // 1024 x 768 pixels: small frames are launch-bound and parts only add launches
uint32_t g_frame_parts = kMaxFrameParts;          // qd_set_frame_parts

int launch(qd_scene* s, Params& P, uint32_t max_depth, hipStream_t stream) {
    WorkSet* w = nullptr;
    if (acquire_set(s, stream, max_depth, &w)) return -1;
    read_sample(s, true);
    read_overflow_words(s);                   // a band outgrew its queues (and was rendered again in place)?
    const int form = choose_form(s, (uint64_t)P.n_units * kUnitPx);
    bool drained = true;                                  // does this frame start on a drained GPU (as far as this scene goes)?
    for (auto& c : s->sets) if (drained && c.t1 && hipEventQuery(c.t1) != hipSuccess) drained = false;
    const bool alone = drained && g_pipeline == 2;        // (only the adaptive choice takes samples)
    if (!w->t0) { HIP_TRY(hipEventCreate(&w->t0)); HIP_TRY(hipEventCreate(&w->t1)); }
    bool wavefront = form == 1 && s->kd_node_list.size() <= kWfMaxKdNodes;

    // parts: their streams, working sets and queues first (nothing of this is timed)
    uint32_t parts = 1;
    WorkSet* pw[kMaxFrameParts] = {w, nullptr, nullptr, nullptr};
    Params pp[kMaxFrameParts];
    if (wavefront && drained && !s->prev_overlapped && g_frame_parts > 1 && P.n_units >= kPartMinUnits && P.unit_step <= 1u && !wf_sort_mode()) {
        parts = std::min<uint32_t>(g_frame_parts, kMaxFrameParts);
        if (!s->part_fork) HIP_TRY(hipEventCreateWithFlags(&s->part_fork, hipEventDisableTiming));
        for (uint32_t k = 1; k < parts; ++k) {
            if (!s->part_stream[k - 1]) HIP_TRY(hipStreamCreateWithFlags(&s->part_stream[k - 1], hipStreamNonBlocking));
            if (!s->part_join[k - 1]) HIP_TRY(hipEventCreateWithFlags(&s->part_join[k - 1], hipEventDisableTiming));
            pw[k] = &s->part_sets[k - 1];
            if (size_set(s, pw[k], s->part_stream[k - 1], max_depth)) return -1;
        }
        // the caller's set keeps queues for a WHOLE frame (part 0 uses a fraction of them): its next frame may be enqueued behind
        // others and stay whole, and growing the queues then would cost a device-wide wait in the middle of the caller's loop
        if (reserve_wavefront(s, w, P, max_depth)) { (void)hipGetLastError(); parts = 1; }
        for (uint32_t k = 0; k < parts && parts > 1; ++k) {
            pp[k] = P;
            pp[k].unit0 = P.unit0 + k; pp[k].unit_step = parts; pp[k].n_units = (P.n_units - k + parts - 1) / parts;
            if (reserve_wavefront(s, pw[k], pp[k], max_depth)) { (void)hipGetLastError(); parts = 1; }   // no room: the frame stays whole
        }
    }
    if (wavefront && parts == 1 && reserve_wavefront(s, w, P, max_depth)) {
        if (g_pipeline != 2) return -1;
        (void)hipGetLastError();                           // no room for its queues: the scene stays on the one-kernel form
        s->wavefront_overflowed = true; wavefront = false;
    }
    HIP_TRY(hipEventRecord(w->t0, stream));
    int rc = 0;
    if (parts > 1) {
        HIP_TRY(hipMemsetAsync(P.rgb, 0, P.rgb_floats * sizeof(float), stream));
        HIP_TRY(hipEventRecord(s->part_fork, stream));
        for (uint32_t k = 1; k < parts; ++k) HIP_TRY(hipStreamWaitEvent(s->part_stream[k - 1], s->part_fork, 0));
        for (uint32_t k = 0; k < parts && !rc; ++k)
            rc = launch_wavefront(s, pw[k], pp[k], max_depth, k ? s->part_stream[k - 1] : stream, w, false);
        for (uint32_t k = 1; k < parts; ++k) {             // (also after a failed enqueue: the caller's stream must not run ahead of what was enqueued)
            if (hipEventRecord(s->part_join[k - 1], s->part_stream[k - 1]) != hipSuccess || hipStreamWaitEvent(stream, s->part_join[k - 1], 0) != hipSuccess)
                return set_err("launch: joining the parts of a frame failed");
        }
    } else rc = wavefront ? launch_wavefront(s, w, P, max_depth, stream) : launch_one_kernel(s, w, P, max_depth, stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(w->t1, stream));
    s->sample.pending = alone;
    s->sample.form = wavefront ? 1 : 0; s->sample.slots = (uint64_t)P.n_units * kUnitPx; s->sample.t0 = w->t0; s->sample.t1 = w->t1;
    s->last_form = w->last_form = wavefront ? 1 : 0;
    s->last_parts = parts;
    s->prev_overlapped = !drained;
    return 0;
}

// fallback: the launch behind a wavefront band (P.cond set): one pass that runs to the end, the frame buffer is left alone
int launch_one_kernel(qd_scene* s, WorkSet* w, Params& P, uint32_t max_depth, hipStream_t stream, bool fallback) {
    P.stack = w->stack; P.stack_cap = w->stack_cap; P.n_lanes = s->n_lanes;
    if (!fallback) P.error_flag = s->error_flag + (w - s->sets);      // (the fallback of a band reports where the band does)
    P.list_cap = w->list_cap;
    HIP_TRY(hipMemsetAsync(w->queue_ctr, 0, (size_t)(fallback ? 1 : kPasses) * kPassWords * sizeof(uint32_t), stream));
    if (!fallback) HIP_TRY(hipMemsetAsync(P.rgb, 0, P.rgb_floats * sizeof(float), stream));       // the kernel adds colours into it
    // the copy for meshes + planes without texture code (FLAT, see trace_kernel) may run fewer waves per SIMD than the others
    const bool flat = !g_counting && (s->geom_mask & 15) == 0 && !s->nested_layered;
    const uint32_t max_blocks = flat ? s->n_lanes / 64 / (uint32_t)QD_WAVES_PER_EU * (uint32_t)kFlatWavesPerEU : s->n_lanes / 64;
    const int n_passes = fallback ? 1 : kPasses;
    for (int pass = 0; pass < n_passes; ++pass) {
        uint32_t* ctl = w->queue_ctr + (size_t)pass * kPassWords;
        P.queue_ctr = ctl; P.out_count = ctl + 8 * kCtlStride; P.pass_done = ctl + 16 * kCtlStride;
        P.source = pass == 0 ? 0 : 1;
        P.in_count = pass == 0 ? nullptr : ctl - kPassWords + 8 * kCtlStride;
        P.in_list = pass == 0 ? nullptr : w->lists[(pass - 1) & 1];
        P.out_list = w->lists[pass & 1];
        P.grace = pass == n_passes - 1 ? 0xffffffffu : kGrace;
        // later passes usually find an empty list; a launch costs ~45 ns per wave (scratch set-up), so they start fewer waves
        const uint32_t grid = pass == 0 ? (P.n_units < max_blocks ? (P.n_units ? P.n_units : 1) : max_blocks) : (max_blocks + kLaterPassDiv - 1) / kLaterPassDiv;
        using Kernel = void (*)(const DScene, const Params);
#ifdef QD_DEV_BUILD
        static const Kernel plain[8] = {trace_kernel<false, 0, false, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>,
                                        trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>};
        static const Kernel rich[8] = {trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>,
                                       trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>, trace_kernel<false, 7, true>};
#else
        static const Kernel plain[8] = {trace_kernel<false, 0, false, true>, trace_kernel<false, 1, false>, trace_kernel<false, 2, false>, trace_kernel<false, 3, false>,
                                        trace_kernel<false, 4, false>, trace_kernel<false, 5, false>, trace_kernel<false, 6, false>, trace_kernel<false, 7, false>};
        static const Kernel rich[8] = {trace_kernel<false, 0, true, true>, trace_kernel<false, 1, true>, trace_kernel<false, 2, true>, trace_kernel<false, 3, true>,
                                       trace_kernel<false, 4, true>, trace_kernel<false, 5, true>, trace_kernel<false, 6, true>, trace_kernel<false, 7, true>};
#endif
        const Kernel kernel = g_counting ? trace_kernel<true, kGeomAll, true>                       // counting: one copy, all code
                            : (s->geom_mask & kGeomCsgDeep) ? trace_kernel<false, kGeomAll, true>    // CSG over CSG / mesh / heightfield: the copy with everything
                            : (!flat && (s->geom_mask & 7) == 0) ? (s->rich ? trace_kernel<false, 0, true> : trace_kernel<false, 0, false>)   // meshes + planes with a nested Layered
                                         : (s->rich ? rich : plain)[s->geom_mask & 7];
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, stream, s->d, P);
        HIP_TRY(hipGetLastError());
    }
    if (!fallback) HIP_TRY(hipEventRecord(w->idle, stream));
    return 0;
}

// Staging buffer `which` of at least `bytes` bytes (see Staging).  Growing one frees the old block, which waits for the device.
int stage(qd_scene* s, int which, size_t bytes, void** out) {
    Staging& g = s->stg;
    if (bytes == 0) bytes = 1;
    if (g.bytes[which] < bytes) {
        if (g.p[which]) { (void)hipFree(g.p[which]); g.p[which] = nullptr; g.bytes[which] = 0; }
        HIP_TRY(hipMalloc(&g.p[which], bytes));
        g.bytes[which] = bytes;
    }
    *out = g.p[which];
    return 0;
}

// What a synchronous entry point does around its launches: counters cleared, events around the launches, wait, the working
// set's error word read (and cleared if raised; it is never cleared ahead of a launch, so a qd_render between qd_frame_begin
// and the frame's first read cannot wipe the frame's error).  `begin_only` stops after the enqueue (qd_frame_begin);
// finish_launch does the rest.
int finish_launch(qd_scene* s, Params& P, uint32_t max_depth, hipStream_t stream, float* ms, bool frame_events);
int timed_launch(qd_scene* s, Params& P, uint32_t max_depth, hipStream_t stream, float* ms, bool begin_only = false, bool frame_events = false) {
    Staging& g = s->stg;
    hipEvent_t& e0 = frame_events ? g.f0 : g.e0;
    hipEvent_t& e1 = frame_events ? g.f1 : g.e1;
    if (!e0) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); }
    WorkSet* w = nullptr;
    if (acquire_set(s, stream, max_depth, &w)) return -1;            // the set launch() is about to use: its error word
    HIP_TRY(hipMemsetAsync(P.stats, 0, 24 * sizeof(unsigned long long), stream));             // the call's own counters (P.stats)
    HIP_TRY(hipMemsetAsync(P.stats + 16, 0xff, sizeof(unsigned long long), stream));
    HIP_TRY(hipMemsetAsync(P.stats + 19, 0xff, sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(e0, stream));
    if (launch(s, P, max_depth, stream)) return -1;
    HIP_TRY(hipEventRecord(e1, stream));
    return begin_only ? 0 : finish_launch(s, P, max_depth, stream, ms, frame_events);
}
int finish_launch(qd_scene* s, Params&, uint32_t, hipStream_t stream, float* ms, bool frame_events) {
    Staging& g = s->stg;
    HIP_TRY(hipEventSynchronize(frame_events ? g.f1 : g.e1));
    if (ms) HIP_TRY(hipEventElapsedTime(ms, frame_events ? g.f0 : g.e0, frame_events ? g.f1 : g.e1));
    WorkSet* w = nullptr;
    for (auto& c : s->sets) if (c.bound && c.stream == stream) { w = &c; break; }
    if (!w) return set_err("the stream's working set was taken over before the call finished (more than 8 streams on one scene)");
    int* flag = s->error_flag + (w - s->sets);
    int err_flag = 0;
    HIP_TRY(hipMemcpy(&err_flag, flag, sizeof(int), hipMemcpyDeviceToHost));
    if (err_flag) {
        (void)hipMemset(flag, 0, sizeof(int));
        return set_err("ray queue overflow on device");
    }
    return 0;
}

void fill_common(Params& P, const qd_camera* cam, const qd_settings* st) {
    memset(&P, 0, sizeof P);
    P.cam = *cam;
    P.W = st->frame_width; P.H = st->frame_height;
    P.ambient[0] = st->ambient[0]; P.ambient[1] = st->ambient[1]; P.ambient[2] = st->ambient[2];
    P.max_depth = st->max_trace_depth;
}

}  // namespace

extern "C" const char* qd_last_error(void) { return g_err.c_str(); }

extern "C" int qd_init(int device) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return set_err("qd_init: no HIP device visible");
    if (device < 0 || device >= n) return set_err("qd_init: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (device < kMaxDevices && !g_scratch_ptr[device]) {    // this device's copy of the static scratch array (device is current)
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return set_err(std::string("qd_init: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    {   // kernarg_scene() / kernarg_params(): do they see what the kernel was given, on this device and code object?
        static bool checked[kMaxDevices] = {};
        if (device < kMaxDevices && !checked[device]) {
            DScene S; Params P;
            memset(&S, 0, sizeof S); memset(&P, 0, sizeof P);
            S.n_nodes = 0x1234567u; S.n_lights = 77u; S.nodes = (const qd_node*)0x1000; S.env_faces[5] = -12345;
            P.W = 4321; P.H = 8765; P.n_units = 0xabcdefu; P.rgb = (float*)0x2000; P.wf_cap = 0x13579bu; P.max_depth = 31u;
            uint32_t* d_out = nullptr; uint32_t h_out = 0;
            HIP_TRY(hipMalloc((void**)&d_out, sizeof(uint32_t)));
            hipLaunchKernelGGL(kernarg_selftest_kernel, dim3(1), dim3(64), 0, nullptr, S, P, d_out);
            const hipError_t e = hipMemcpy(&h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
            (void)hipFree(d_out);
            if (e != hipSuccess) return set_err(std::string("qd_init: kernel-argument self-test: ") + hipGetErrorString(e));
            if (h_out != 0x600dbeefu) return set_err("qd_init: kernel-argument self-test failed: the kernels' explicit arguments are not where kernarg_scene() / kernarg_params() read them");
            checked[device] = true;
        }
    }
    g_device = device;
    g_num_cu = prop.multiProcessorCount;
    return 0;
}

extern "C" void qd_shutdown(void) { g_device = -1; }

extern "C" int qd_set_counting(int on) { g_counting = on != 0; return 0; }

extern "C" int qd_set_kd_build(int mode) {
    if (mode < 0 || mode > 2) return set_err("qd_set_kd_build: 0 (the caller's planes, leaves refined), 1 (whole tree by the surface-area heuristic, built on the device) or 2 (the same, built by host threads)");
    g_kd_build = mode;
    return 0;
}

extern "C" int qd_set_pipeline(int pipeline) {
    if (pipeline < 0 || pipeline > 2) return set_err("qd_set_pipeline: 0 (one kernel, per-lane queues), 1 (wavefront) or 2 (the faster of the two, measured)");
    g_pipeline = pipeline;
    return 0;
}

extern "C" int qd_set_kd_refine(int max_leaf_tris, double step_cost) {
    if (max_leaf_tris < 0 || !(step_cost > 0.)) return set_err("qd_set_kd_refine: max_leaf_tris >= 0 and step_cost > 0 required");
    g_refine_leaf = max_leaf_tris; g_refine_cost = step_cost;
    return 0;
}

extern "C" int qd_scene_kd_info(const qd_scene* s, uint64_t* nodes, uint64_t* leaf_refs, uint64_t* leaves, uint64_t* max_leaf) {
    if (!s) return set_err("qd_scene_kd_info: null scene");
    if (nodes) *nodes = s->kd_nodes;
    if (leaf_refs) *leaf_refs = s->kd_refs;
    if (leaves) *leaves = s->kd_leaves;
    if (max_leaf) *max_leaf = s->kd_max_leaf;
    return 0;
}

extern "C" int qd_scene_setup_info(const qd_scene* s, double* kd_build_seconds, double* upload_seconds) {
    if (!s) return set_err("qd_scene_setup_info: null scene");
    if (kd_build_seconds) *kd_build_seconds = s->t_kd_build;
    if (upload_seconds) *upload_seconds = s->t_upload;
    return 0;
}

extern "C" int qd_scene_schedule_info(const qd_scene* cs, int* last_form, double* ns_per_pixel_form0, double* ns_per_pixel_form1) {
    if (!cs) return set_err("qd_scene_schedule_info: null scene");
    qd_scene* s = const_cast<qd_scene*>(cs);
    std::lock_guard<std::mutex> lock(s->mu);
    read_sample(s, false);                   // the latest frame, if it ran alone and has finished
    if (last_form) *last_form = s->last_form;
    if (ns_per_pixel_form0) *ns_per_pixel_form0 = s->form_ns[0];
    if (ns_per_pixel_form1) *ns_per_pixel_form1 = s->form_ns[1];
    return 0;
}

extern "C" int qd_scene_create(const qd_scene_desc* desc, qd_scene** out) {
    if (!desc || !out) return set_err("qd_scene_create: null argument");
    if (g_device < 0) return set_err("qd_scene_create: qd_init has not been called");
    *out = nullptr;
    // validate indices the kernels rely on (a faulting kernel can take the whole GPU down)
    for (uint32_t i = 0; i < desc->n_geoms; ++i) {
        const qd_geom& g = desc->geoms[i];
        if (g.type == QD_GEOM_CSG) {
            if (g.a < 0 || g.b < 0 || (uint32_t)g.a >= desc->n_geoms || (uint32_t)g.b >= desc->n_geoms) return set_err("qd_scene_create: CSG operand out of range");
        } else if (g.type == QD_GEOM_MESH) {
            if (g.a < 0 || (uint32_t)g.a >= desc->n_meshes) return set_err("qd_scene_create: mesh index out of range");
        } else if (g.type == QD_GEOM_HEIGHTFIELD) {
            if (g.a < 0 || (uint32_t)g.a >= desc->n_hfields) return set_err("qd_scene_create: heightfield index out of range");
        } else if (g.type < 0 || g.type > QD_GEOM_HEIGHTFIELD) return set_err("qd_scene_create: unknown geometry type");
    }
    // CSG trees: at most kMaxCsgNesting levels of CSG nodes (the device unrolls the recursion of CsgOp::Intersect); the depth
    // bound also rejects cycles
    std::function<int(int, int)> csg_depth = [&](int gi, int level) -> int {
        const qd_geom& g = desc->geoms[gi];
        if (g.type != QD_GEOM_CSG) return 0;
        if (level >= kMaxCsgNesting) return kMaxCsgNesting + 1;
        return 1 + std::max(csg_depth(g.a, level + 1), csg_depth(g.b, level + 1));
    };
    for (uint32_t i = 0; i < desc->n_geoms; ++i)
        if (desc->geoms[i].type == QD_GEOM_CSG && csg_depth((int)i, 0) > kMaxCsgNesting)
            return set_err("qd_scene_create: CSG geometries nested deeper than 4 levels (or a cycle)");
    for (uint32_t i = 0; i < desc->n_meshes; ++i) {
        const qd_mesh& m = desc->meshes[i];
        if ((uint64_t)m.tri_first + m.tri_count > desc->n_tris) return set_err("qd_scene_create: mesh triangle range out of bounds");
        if (m.kd_root >= 0 && (uint64_t)m.kd_root >= desc->n_kd_nodes) return set_err("qd_scene_create: kd root out of bounds");
    }
    for (uint64_t i = 0; i < desc->n_kd_nodes; ++i) {
        const qd_kdnode& n = desc->kd_nodes[i];
        const uint64_t idx = n.meta >> 2;
        if ((n.meta & 3u) == 3u) { if (idx + n.count > desc->n_leaf_tris) return set_err("qd_scene_create: kd leaf range out of bounds"); }
        else if (idx + 1 >= desc->n_kd_nodes) return set_err("qd_scene_create: kd child out of bounds");
    }
    {   // every mesh's tree, walked from its root: a node is reached once (no cycles, no child shared between parents -- the
        // refinement's descent would not terminate on those), and a leaf's references index the OWNING mesh's triangles
        std::vector<uint8_t> seen(desc->n_kd_nodes, 0);
        std::vector<uint64_t> todo;
        for (uint32_t mi = 0; mi < desc->n_meshes; ++mi) {
            const qd_mesh& m = desc->meshes[mi];
            if (m.kd_root < 0) continue;
            todo.assign(1, (uint64_t)m.kd_root);
            while (!todo.empty()) {
                const uint64_t i = todo.back(); todo.pop_back();
                if (seen[i]) return set_err("qd_scene_create: kd tree has a cycle or a node with two parents");
                seen[i] = 1;
                const qd_kdnode& n = desc->kd_nodes[i];
                const uint64_t idx = n.meta >> 2;
                if ((n.meta & 3u) == 3u) {
                    for (uint64_t k = 0; k < n.count; ++k)
                        if (desc->leaf_tris[idx + k] >= m.tri_count) return set_err("qd_scene_create: leaf triangle index outside its mesh");
                } else { todo.push_back(idx); todo.push_back(idx + 1); }
            }
        }
    }
    auto tex_ok = [&](int t) { return t >= -1 && t < (int)desc->n_textures; };
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const qd_texture& t = desc->textures[i];
        if ((t.type == QD_TEX_BITMAP || t.type == QD_TEX_BUMPMAP) && (t.bitmap < 0 || (uint32_t)t.bitmap >= desc->n_bitmaps)) return set_err("qd_scene_create: texture bitmap out of range");
        if (t.type == QD_TEX_PROCEDURAL && (t.i0 < 0 || t.i1 < 0 || (uint32_t)(t.i0 + t.i1) > desc->n_proc_lines)) return set_err("qd_scene_create: procedural lines out of range");
    }
    for (uint32_t i = 0; i < desc->n_bitmaps; ++i) {
        const qd_bitmap& b = desc->bitmaps[i];
        if (b.texel_off + (uint64_t)b.width * b.height > desc->n_texels) return set_err("qd_scene_create: bitmap texels out of range");
    }
    uint32_t spawn0 = 1, spawn1 = 1;
    auto leaf_spawn = [&](const qd_shader& sh, bool deep) -> uint32_t {
        switch (sh.type) {
            case QD_SH_LAMBERT: case QD_SH_PHONG: case QD_SH_BLINNPHONG: case QD_SH_ORENNAYAR: return desc->n_lights;
            case QD_SH_REFL: return sh.p1 == 1 ? 1u : (deep ? 2u : (uint32_t)std::max(sh.i0, 1));
            case QD_SH_REFR: return 1u;
            default: return 0u;
        }
    };
    // tasks one hit can queue: the leaves of the shader tree (Layered may nest, kMaxLayerNesting levels below the top)
    std::string spawn_err;
    bool nested_layered = false;             // a Layered shader that is a layer of another one
    std::function<bool(const qd_shader&, int, uint32_t&, uint32_t&)> tree_spawn = [&](const qd_shader& sh, int level, uint32_t& s0, uint32_t& s1) -> bool {
        if (sh.type != QD_SH_LAYERED) { s0 += leaf_spawn(sh, false); s1 += leaf_spawn(sh, true); return true; }
        if (level > kMaxLayerNesting) { spawn_err = "qd_scene_create: Layered shaders nested deeper than 4 levels"; return false; }
        if (level >= 1) nested_layered = true;
        if (sh.i0 < 0 || sh.i1 < 0 || (uint32_t)(sh.i0 + sh.i1) > desc->n_layers) { spawn_err = "qd_scene_create: layer range out of bounds"; return false; }
        for (int l = 0; l < sh.i1; ++l) {
            const qd_layer& ly = desc->layers[sh.i0 + l];
            if (ly.shader < 0 || (uint32_t)ly.shader >= desc->n_shaders || !tex_ok(ly.tex)) { spawn_err = "qd_scene_create: layer reference out of range"; return false; }
            if (!tree_spawn(desc->shaders[ly.shader], level + 1, s0, s1)) return false;
        }
        return true;
    };
    for (uint32_t i = 0; i < desc->n_shaders; ++i) {
        const qd_shader& sh = desc->shaders[i];
        if (!tex_ok(sh.tex)) return set_err("qd_scene_create: shader texture out of range");
        uint32_t s0 = 0, s1 = 0;
        if (!tree_spawn(sh, 0, s0, s1)) return set_err(spawn_err);
        spawn0 = std::max(spawn0, s0); spawn1 = std::max(spawn1, s1);
    }
    for (uint32_t i = 0; i < desc->n_nodes; ++i) {
        const qd_node& n = desc->nodes[i];
        if (n.geom < 0 || (uint32_t)n.geom >= desc->n_geoms || n.shader < 0 || (uint32_t)n.shader >= desc->n_shaders || !tex_ok(n.bump_tex))
            return set_err("qd_scene_create: node reference out of range");
    }
    if (desc->env_present)
        for (int f = 0; f < 6; ++f)
            if (desc->env_faces[f] < 0 || (uint32_t)desc->env_faces[f] >= desc->n_bitmaps) return set_err("qd_scene_create: environment face out of range");
    for (uint32_t i = 0; i < desc->n_hfields; ++i) {
        const qd_heightfield& h = desc->hfields[i];
        const uint64_t cells = (uint64_t)h.width * h.height;
        if (h.heights_off + cells > desc->n_hf_f32 || h.maxheights_off + cells > desc->n_hf_f32 || h.normals_off + cells * 3 > desc->n_hf_f64 ||
            (h.use_optimization && h.highmap_off + cells * 16 > desc->n_hf_f32) || h.max_k > 16)
            return set_err("qd_scene_create: heightfield arrays out of range");
    }

    qd_scene* s = new qd_scene;
    memset(&s->d, 0, sizeof s->d);
    DScene& d = s->d;
    int rc = 0;
    rc |= upload(s, desc->nodes, desc->n_nodes, &d.nodes);
    rc |= upload(s, desc->geoms, desc->n_geoms, &d.geoms);
    std::vector<qd_mesh> meshes_dev(desc->meshes, desc->meshes + desc->n_meshes);
    auto wants_tree = [](const qd_mesh& m) { return m.kd_root >= 0 || (m.flags & QD_MESH_WANT_KD) != 0; };
    rc |= upload(s, desc->tri_verts, (size_t)desc->n_tris * 9, &d.tri_verts);
    size_t n_leaf_refs = 0;
    const bool device_build = g_kd_build == 1 && g_refine_leaf > 0;
    if (!rc && device_build) {
        // the device KD trees, built on the device from the triangles just uploaded (kdbuild.h)
        const auto t_build0 = std::chrono::steady_clock::now();
        qd_kdbuild::Opt opt;
        opt.max_leaf = g_refine_leaf; opt.cost_step = g_refine_cost; opt.max_depth = kKdStack - 2; opt.binned_above = 192;
        qd_kdbuild::Trees T;
        for (size_t mi = 0; mi < meshes_dev.size() && !rc; ++mi) {
            if (!wants_tree(meshes_dev[mi])) continue;
            uint32_t root = 0;
            const std::string err = qd_kdbuild::build_mesh(d.tri_verts, meshes_dev[mi], opt, T, &root);
            if (!err.empty()) { rc = set_err("qd_scene_create: device KD build: " + err); break; }
            meshes_dev[mi].kd_root = (int32_t)root;
        }
        if (!rc) {
            s->kd_nodes = T.n_nodes; s->kd_refs = T.n_refs; s->kd_leaves = T.leaves; s->kd_max_leaf = T.max_leaf;
            n_leaf_refs = (size_t)T.n_refs;
            if (hipDeviceSynchronize() != hipSuccess) rc = set_err("qd_scene_create: device KD build failed");
            void* recs = T.records.release(); void* refs = T.leaf_refs.release();
            if (recs) s->allocs.push_back(recs);
            if (refs) s->allocs.push_back(refs);
            d.kd_nodes = (const qd_kdnode*)recs; d.leaf_tris = (const uint32_t*)refs;
        }
        s->t_kd_build = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count();
    } else if (!rc) {
        // the device KD trees built by the host cores (kdrefine.h): the caller's planes with refined leaves (mode 0), or the
        // whole tree by the same heuristic as the device builder (mode 2)
        for (const qd_mesh& m : meshes_dev)
            if (m.kd_root < 0 && (m.flags & QD_MESH_WANT_KD) && !(g_kd_build == 2 && g_refine_leaf > 0)) {
                qd_scene_destroy(s);
                return set_err("qd_scene_create: qd_set_kd_build(0) / qd_set_kd_refine(0, .) need the caller's KD tree, and this scene description "
                               "has none (libqdhost builds it on request: qdh_scene_build_kd)");
            }
        for (qd_mesh& m : meshes_dev) if (m.kd_root < 0 && (m.flags & QD_MESH_WANT_KD)) m.kd_root = 0;     // mode 2 ignores the caller's tree
        std::vector<qd_refine::Record> kd_records;
        std::vector<uint32_t> leaf_global;       // leaf references as GLOBAL triangle indices
        qd_scene_desc view = *desc;
        view.meshes = meshes_dev.data();
        qd_refine::Options opt;
        opt.max_leaf = g_refine_leaf;
        opt.cost_step = g_refine_cost;
        opt.max_depth = kKdStack - 2;
        opt.rebuild = g_kd_build != 0 && g_refine_leaf > 0;
        qd_refine::Result refined;
        const auto t_build0 = std::chrono::steady_clock::now();
        const std::string err = qd_refine::refine(&view, opt, refined);
        if (!err.empty()) { qd_scene_destroy(s); return set_err("qd_scene_create: " + err); }
        if (refined.n_refs >= (1ull << 32) || refined.n_nodes >= (1ull << 27)) { qd_scene_destroy(s); return set_err("qd_scene_create: KD tree too large for the 28-bit record index"); }
        leaf_global.reserve(refined.n_refs);
        for (size_t mi = 0; mi < meshes_dev.size(); ++mi) {
            const qd_refine::MeshTree& T = refined.trees[mi];
            if (T.nodes.empty()) { meshes_dev[mi].kd_root = -1; continue; }
            meshes_dev[mi].kd_root = (int32_t)qd_refine::pack(T, kd_records, leaf_global.size());
            for (uint32_t t : T.refs) leaf_global.push_back(t + meshes_dev[mi].tri_first);
        }
        s->kd_nodes = refined.n_nodes; s->kd_refs = refined.n_refs; s->kd_leaves = refined.leaves; s->kd_max_leaf = refined.max_leaf;
        s->t_kd_build = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count();
        if (kd_records.size() >= (1u << 28)) { qd_scene_destroy(s); return set_err("qd_scene_create: KD tree too large for the 28-bit record index"); }
        { const qd_refine::Record* recs = nullptr; rc |= upload(s, kd_records.data(), kd_records.size(), &recs); d.kd_nodes = (const qd_kdnode*)recs; }
        rc |= upload(s, leaf_global.data(), leaf_global.size(), &d.leaf_tris);
        n_leaf_refs = leaf_global.size();
    }
    const auto t_upload0 = std::chrono::steady_clock::now();
    rc |= upload(s, meshes_dev.data(), meshes_dev.size(), &d.meshes);
    rc |= upload(s, desc->hfields, desc->n_hfields, &d.hfields);
    rc |= upload(s, desc->shaders, desc->n_shaders, &d.shaders);
    rc |= upload(s, desc->layers, desc->n_layers, &d.layers);
    rc |= upload(s, desc->textures, desc->n_textures, &d.textures);
    rc |= upload(s, desc->proc_lines, desc->n_proc_lines, &d.proc_lines);
    rc |= upload(s, desc->bitmaps, desc->n_bitmaps, &d.bitmaps);
    rc |= upload(s, desc->lights, desc->n_lights, &d.lights);
    rc |= upload(s, desc->tri_gnormal, (size_t)desc->n_tris * 3, &d.tri_gnormal);
    rc |= upload(s, desc->tri_shade, (size_t)desc->n_tris * 21, &d.tri_shade);
    if (!rc) {
        const size_t n_words = n_leaf_refs * 9;
        void* lv = nullptr;
        if (hipMalloc(&lv, (n_words ? n_words : 1) * sizeof(double) + 256) != hipSuccess) rc = set_err("hipMalloc(leaf_verts) failed");
        else {
            s->allocs.push_back(lv);
            d.leaf_verts = (const double*)lv;
            if (n_words) {
                hipLaunchKernelGGL(gather_leaf_verts_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, nullptr,
                                   d.tri_verts, d.leaf_tris, (double*)lv, n_words);
                if (hipDeviceSynchronize() != hipSuccess) rc = set_err("gather_leaf_verts failed");
            }
        }
    }
    {
        std::vector<int> slot(desc->n_nodes, -1);
        for (uint32_t i = 0; i < desc->n_nodes; ++i) {
            const qd_geom& g = desc->geoms[desc->nodes[i].geom];
            if (g.type == QD_GEOM_MESH && meshes_dev[g.a].kd_root >= 0) { slot[i] = (int)s->kd_node_list.size(); s->kd_node_list.push_back((int)i); }
            else if (g.type == QD_GEOM_MESH) s->small_mesh = true;
        }
        rc |= upload(s, slot.data(), slot.size(), &d.node_kd_slot);
    }
    rc |= upload(s, desc->texels, (size_t)desc->n_texels * 3, &d.texels);
    rc |= upload(s, desc->hf_f32, (size_t)desc->n_hf_f32, &d.hf_f32);
    rc |= upload(s, desc->hf_f64, (size_t)desc->n_hf_f64, &d.hf_f64);
    d.n_nodes = desc->n_nodes; d.n_lights = desc->n_lights;
    d.env_present = desc->env_present; d.env_bilinear = desc->env_bilinear;
    for (int f = 0; f < 6; ++f) d.env_faces[f] = desc->env_faces[f];
    s->spawn0 = spawn0; s->spawn1 = spawn1;
    s->nested_layered = nested_layered;
    s->geom_mask = 0;
    for (uint32_t i = 0; i < desc->n_geoms; ++i) {
        const int t = desc->geoms[i].type;
        if (t == QD_GEOM_HEIGHTFIELD) s->geom_mask |= kGeomHf;
        else if (t == QD_GEOM_CSG) {
            s->geom_mask |= kGeomCsg;
            for (int op : {desc->geoms[i].a, desc->geoms[i].b}) {
                const int ot = desc->geoms[op].type;
                if (ot == QD_GEOM_CSG || ot == QD_GEOM_MESH || ot == QD_GEOM_HEIGHTFIELD) s->geom_mask |= kGeomCsgDeep;
            }
        }
        else if (t != QD_GEOM_MESH && t != QD_GEOM_PLANE) s->geom_mask |= kGeomPrim;
    }
    s->rich = false;
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const int t = desc->textures[i].type;
        if (t != QD_TEX_CONST && t != QD_TEX_CHECKER && t != QD_TEX_FRESNEL) s->rich = true;
    }
    for (uint32_t i = 0; i < desc->n_nodes; ++i) if (desc->nodes[i].bump_tex >= 0) s->rich = true;
#ifdef QD_NO_GEOM_MASK
    s->geom_mask = kGeomAll; s->rich = true;
#endif
    s->n_lanes = (uint32_t)(g_num_cu > 0 ? g_num_cu : 256) * kWavesPerCU * 64u;
    if (!rc && hipMalloc((void**)&s->stats, (24 + 6 * 4096) * sizeof(unsigned long long)) != hipSuccess) rc = set_err("hipMalloc(stats) failed");
    if (!rc && hipMalloc((void**)&s->error_flag, kErrFlags * sizeof(int)) != hipSuccess) rc = set_err("hipMalloc(error flags) failed");
    if (!rc && hipMemset(s->error_flag, 0, kErrFlags * sizeof(int)) != hipSuccess) rc = set_err("hipMemset failed");
    if (!rc && (hipHostMalloc((void**)&s->host_flag, 2 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&s->host_flag_dev, s->host_flag, 0) != hipSuccess)) rc = set_err("hipHostMalloc(overflow word) failed");
    if (!rc) { s->host_flag[0] = 0u; s->host_flag[1] = 0u; }
    if (!rc) {                                   // working set of the default stream, so that a first render allocates nothing
        WorkSet* w = nullptr;
        rc = acquire_set(s, nullptr, desc->settings.max_trace_depth, &w);
    }
    if (rc) { qd_scene_destroy(s); return -1; }
    s->t_upload = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_upload0).count();
    *out = s;
    return 0;
}

extern "C" void qd_scene_destroy(qd_scene* s) {
    if (!s) return;
    wf_lab_report();
    for (void* p : s->allocs) (void)hipFree(p);
    for (auto& w : s->sets) free_set(w);
    for (auto& w : s->part_sets) free_set(w);
    for (void* p : s->stg.p) if (p) (void)hipFree(p);
    if (s->stg.host_frame) (void)hipHostFree(s->stg.host_frame);
    if (s->stg.e0) (void)hipEventDestroy(s->stg.e0);
    if (s->stg.e1) (void)hipEventDestroy(s->stg.e1);
    if (s->stg.f0) (void)hipEventDestroy(s->stg.f0);
    if (s->stg.f1) (void)hipEventDestroy(s->stg.f1);
    if (s->stats) (void)hipFree(s->stats);
    if (s->error_flag) (void)hipFree(s->error_flag);
    if (s->host_flag) (void)hipHostFree(s->host_flag);
    for (auto& st : s->part_stream) if (st) (void)hipStreamDestroy(st);
    for (auto& e : s->part_join) if (e) (void)hipEventDestroy(e);
    if (s->part_fork) (void)hipEventDestroy(s->part_fork);
    delete s;
}

extern "C" int qd_render(qd_scene* s, const qd_camera* cam, const qd_settings* st, int x0, int y0, int x1, int y1,
                         float* rgb, size_t pitch, int32_t* hit_node, int32_t* hit_tri, double* hit_dist, qd_stats* stats) {
    if (!s || !cam || !st || !rgb) return set_err("qd_render: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (x1 <= x0 || y1 <= y0 || x0 < 0 || y0 < 0 || x1 > st->frame_width || y1 > st->frame_height) return set_err("qd_render: bad rectangle");
    if ((hit_node || hit_tri || hit_dist) && !(hit_node && hit_tri && hit_dist)) return set_err("qd_render: hit outputs must be all set or all NULL");
    const int rw = x1 - x0, rh = y1 - y0;
    if (pitch < (size_t)rw * 3) return set_err("qd_render: row pitch smaller than the rectangle");
    Params P; fill_common(P, cam, st);
    P.mode = 0; P.x0 = x0; P.y0 = y0; P.rw = rw; P.rh = rh;
    P.units_x = (uint32_t)(rw + kUnitSide - 1) / kUnitSide;
    P.n_units = P.units_x * (uint32_t)((rh + kUnitSide - 1) / kUnitSide);
    const size_t npx = (size_t)rw * rh;
    float* d_rgb = nullptr; int32_t* d_hn = nullptr; int32_t* d_ht = nullptr; double* d_hd = nullptr;
    auto cleanup = []() {};                                  // the staging buffers stay with the scene handle
#define TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return -1; } } while (0)
    if (stage(s, kStRgb, npx * 3 * sizeof(float), (void**)&d_rgb)) return -1;
    if (hit_node) {
        if (stage(s, kStHitNode, npx * sizeof(int32_t), (void**)&d_hn) || stage(s, kStHitTri, npx * sizeof(int32_t), (void**)&d_ht) ||
            stage(s, kStHitDist, npx * sizeof(double), (void**)&d_hd)) return -1;
    }
    P.rgb = d_rgb; P.pitch = (size_t)rw * 3; P.rgb_floats = npx * 3; P.hit_node = d_hn; P.hit_tri = d_ht; P.hit_dist = d_hd;
    P.stats = s->stats;
    float ms = 0;
    if (timed_launch(s, P, st->max_trace_depth, nullptr, &ms)) { set_err(std::string("qd_render: ") + g_err); return -1; }
    if (pitch == (size_t)rw * 3) TRY2(hipMemcpy(rgb, d_rgb, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));      // contiguous rows: one copy
    else TRY2(hipMemcpy2D(rgb, pitch * sizeof(float), d_rgb, (size_t)rw * 3 * sizeof(float), (size_t)rw * 3 * sizeof(float), rh, hipMemcpyDeviceToHost));
    if (hit_node) {
        TRY2(hipMemcpy(hit_node, d_hn, npx * sizeof(int32_t), hipMemcpyDeviceToHost));
        TRY2(hipMemcpy(hit_tri, d_ht, npx * sizeof(int32_t), hipMemcpyDeviceToHost));
        TRY2(hipMemcpy(hit_dist, d_hd, npx * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (stats) {
        unsigned long long h[24];
        TRY2(hipMemcpy(h, s->stats, sizeof h, hipMemcpyDeviceToHost));
#ifdef QD_PROFILE
        { std::vector<unsigned long long> w(6 * 4096);
          (void)hipMemcpy(w.data(), s->stats + 24, w.size() * 8, hipMemcpyDeviceToHost);
          std::vector<int> idx(4096); for (int i = 0; i < 4096; ++i) idx[i] = i;
          std::sort(idx.begin(), idx.end(), [&](int a, int b) { return w[6 * a] > w[6 * b]; });
          for (int k : {0, 1, 16, 256, 1024, 2048, 3072, 4095}) { const int i = idx[k];
              fprintf(stderr, "[qd profile] wave #%d by life: life %llu ticks, pool empty at %llu, %llu rounds / %llu tasks in all, %llu rounds / %llu tasks after\n",
                      k, w[6 * i], w[6 * i + 1], w[6 * i + 2], w[6 * i + 3], w[6 * i + 4], w[6 * i + 5]); }
          unsigned long long tl = 0, tr = 0, tt = 0; for (int i = 0; i < 4096; ++i) { tl += w[6 * i]; tr += w[6 * i + 2]; tt += w[6 * i + 3]; }
          fprintf(stderr, "[qd profile] all waves: life %llu ticks, rounds %llu, tasks %llu\n", tl, tr, tt); }
        fprintf(stderr, "[qd profile] timeline (ticks since first wave start): first pool-empty %llu, last pool-empty %llu, last wave end %llu, mean wave life %llu\n",
                h[19] - h[16], h[18] - h[16], h[17] - h[16], h[20] / 4096);
        fprintf(stderr, "[qd profile] wave-cycles total %llu sched %llu trace %llu shade %llu | kd lane-iters %llu wave-iters %llu | xform %llu mesh %llu analytic %llu bookkeeping %llu\n", h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[7], h[6]);
#endif
        stats->rays = h[0]; stats->primary_rays = h[1]; stats->shadow_rays = h[2]; stats->depth_killed = h[3];
        stats->kd_nodes_visited = h[4]; stats->tris_tested = h[5]; stats->mesh_queries = h[6];
        stats->kernel_ms = ms;
    }
    cleanup();
#undef TRY2
    return 0;
}


// ------------------------------------------------------------------------------------------ frame ahead of a bucket loop
// The reference renders a frame bucket by bucket and shows each bucket as it is done (RenderScene, src/main.cpp:85-104 +
// DisplayVFBRect).  A launch per 48 x 48 bucket leaves the GPU empty (36 waves) and pays a round trip each; so the whole
// frame is enqueued once (qd_frame_begin, returns at once), and the bucket loop reads finished rectangles out of it
// (qd_frame_read_rect: the first read waits for the frame and brings it to a pinned host copy in one transfer, every read
// is then a copy of rows on the host).
extern "C" int qd_frame_begin(qd_scene* s, const qd_camera* cam, const qd_settings* st, void* hip_stream) {
    if (!s || !cam || !st) return set_err("qd_frame_begin: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    const int W = st->frame_width, H = st->frame_height;
    if (W <= 0 || H <= 0) return set_err("qd_frame_begin: frame size out of range");
    FramePlan& f = s->frame;
    if (f.launched && !f.fetched) HIP_TRY(hipEventSynchronize(s->stg.f1));      // an abandoned frame: let it drain before its buffer is reused
    f = FramePlan();
    float* d_frame = nullptr;
    const size_t npx = (size_t)W * H;
    if (stage(s, kStFrame, npx * 3 * sizeof(float), (void**)&d_frame)) return -1;
    if (s->stg.host_bytes < npx * 3 * sizeof(float)) {
        if (s->stg.host_frame) { (void)hipHostFree(s->stg.host_frame); s->stg.host_frame = nullptr; s->stg.host_bytes = 0; }
        HIP_TRY(hipHostMalloc(&s->stg.host_frame, npx * 3 * sizeof(float), hipHostMallocDefault));
        s->stg.host_bytes = npx * 3 * sizeof(float);
    }
    fill_common(f.P, cam, st);
    f.P.mode = 0; f.P.x0 = 0; f.P.y0 = 0; f.P.rw = W; f.P.rh = H;
    f.P.units_x = (uint32_t)(W + kUnitSide - 1) / kUnitSide;
    f.P.n_units = f.P.units_x * (uint32_t)((H + kUnitSide - 1) / kUnitSide);
    unsigned long long* d_stats = nullptr;            // counters of its own: a qd_render may come between begin and end
    if (stage(s, kStFrameStats, (24 + 6 * 4096) * sizeof(unsigned long long), (void**)&d_stats)) return -1;
    f.P.rgb = d_frame; f.P.pitch = (size_t)W * 3; f.P.rgb_floats = npx * 3; f.P.stats = d_stats;
    f.W = W; f.H = H; f.stream = (hipStream_t)hip_stream; f.max_depth = st->max_trace_depth;
    if (timed_launch(s, f.P, f.max_depth, f.stream, nullptr, true, true)) { set_err(std::string("qd_frame_begin: ") + g_err); return -1; }
    f.launched = true;
    return 0;
}

namespace {
int frame_fetch(qd_scene* s) {
    FramePlan& f = s->frame;
    if (!f.launched) return set_err("no frame in flight (qd_frame_begin first)");
    if (f.fetched) return 0;
    if (finish_launch(s, f.P, f.max_depth, f.stream, &f.ms, true)) { f = FramePlan(); return -1; }
    HIP_TRY(hipMemcpy(s->stg.host_frame, f.P.rgb, (size_t)f.W * f.H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    f.fetched = true;
    return 0;
}
}  // namespace

extern "C" int qd_frame_read_rect(qd_scene* s, int x0, int y0, int x1, int y1, float* rgb, size_t pitch) {
    if (!s || !rgb) return set_err("qd_frame_read_rect: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (frame_fetch(s)) { set_err(std::string("qd_frame_read_rect: ") + g_err); return -1; }
    const FramePlan& f = s->frame;
    if (x1 <= x0 || y1 <= y0 || x0 < 0 || y0 < 0 || x1 > f.W || y1 > f.H) return set_err("qd_frame_read_rect: bad rectangle");
    const size_t rw = (size_t)(x1 - x0);
    if (pitch < rw * 3) return set_err("qd_frame_read_rect: row pitch smaller than the rectangle");
    const float* src = (const float*)s->stg.host_frame;
    for (int y = y0; y < y1; ++y) memcpy(rgb + (size_t)(y - y0) * pitch, src + ((size_t)y * f.W + x0) * 3, rw * 3 * sizeof(float));
    return 0;
}

extern "C" int qd_frame_end(qd_scene* s, qd_stats* stats) {
    if (!s) return set_err("qd_frame_end: null scene");
    std::lock_guard<std::mutex> lock(s->mu);
    if (frame_fetch(s)) { set_err(std::string("qd_frame_end: ") + g_err); return -1; }
    if (stats) {
        unsigned long long h[8];
        HIP_TRY(hipMemcpy(h, s->frame.P.stats, sizeof h, hipMemcpyDeviceToHost));
        stats->rays = h[0]; stats->primary_rays = h[1]; stats->shadow_rays = h[2]; stats->depth_killed = h[3];
        stats->kd_nodes_visited = h[4]; stats->tris_tested = h[5]; stats->mesh_queries = h[6];
        stats->kernel_ms = s->frame.ms;
    }
    s->frame = FramePlan();
    return 0;
}


namespace {
#define TRY3(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); aa_free(s); return -1; } } while (0)

void aa_free(qd_scene* s) { s->aa = AaPlan(); }      // the plan's buffers are staging buffers of the scene handle: they stay

// Step 1 of the AA pass: copy the first-pass frame (device, contiguous) into the handle, mark, list.
int aa_begin(qd_scene* s, const qd_settings* st, const float* rgb_dev, uint32_t* n_listed) {
    const int W = st->frame_width, H = st->frame_height;
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return set_err("qd_antialias: frame size out of range");
    aa_free(s);
    AaPlan& a = s->aa;
    const size_t npx = (size_t)W * H;
    uint32_t* d_count = nullptr;
    if (stage(s, kStAaFirst, npx * 3 * sizeof(float), (void**)&a.first) || stage(s, kStAaSlot, npx * sizeof(int32_t), (void**)&a.slot) ||
        stage(s, kStAaList, (npx + 1) * sizeof(uint32_t), (void**)&a.list)) { aa_free(s); return -1; }     // last word of the list: the count
    d_count = a.list + npx;
    TRY3(hipMemcpyAsync(a.first, rgb_dev, npx * 3 * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    TRY3(hipMemsetAsync(d_count, 0, sizeof(uint32_t), nullptr));
    TRY3(hipMemsetAsync(s->error_flag + kErrAa, 0, sizeof(int), nullptr));
    const dim3 blk(64), grd((unsigned)(W + 63) / 64, (unsigned)H);
    hipLaunchKernelGGL(aa_mark_kernel, grd, blk, 0, nullptr, a.first, W, H, st->want_adaptive_aa, st->aa_threshold, a.slot, a.list, d_count);
    TRY3(hipMemcpy(&a.n_list, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (a.n_list > 1) {
        uint32_t* sorted = nullptr; void* tmp = nullptr; size_t tmp_bytes = 0;
        TRY3(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, a.list, sorted, (int)a.n_list));
        if (stage(s, kStAaSorted, (size_t)a.n_list * sizeof(uint32_t), (void**)&sorted) || stage(s, kStAaTmp, tmp_bytes, &tmp)) { aa_free(s); return -1; }
        TRY3(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, a.list, sorted, (int)a.n_list));
        TRY3(hipMemcpyAsync(a.list, sorted, (size_t)a.n_list * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
        hipLaunchKernelGGL(aa_slot_kernel, dim3((a.n_list + 255u) / 256u), dim3(256), 0, nullptr, a.list, a.n_list, W, a.slot);
        TRY3(hipGetLastError());
    }
    a.W = W; a.H = H;
    a.traced = a.n_list && !st->show_aa;
    if (n_listed) *n_listed = a.traced ? a.n_list : 0u;
    return 0;
}

// Step 2: the four taps of listed pixels [first_pixel, first_pixel + n_pixels) -> taps_dev[n_pixels * 4 * 3] (enqueue only).
int aa_trace(qd_scene* s, const qd_camera* cam, const qd_settings* st, uint32_t first_pixel, uint32_t n_pixels, float* taps_dev,
             unsigned long long* stats_dev, hipStream_t stream) {
    const AaPlan& a = s->aa;
    if (!a.first || st->frame_width != a.W || st->frame_height != a.H) return set_err("qd_aa_trace_part_async: no plan for this frame (qd_aa_begin_device first)");
    if (!a.traced || n_pixels == 0) return 0;
    if ((uint64_t)first_pixel + n_pixels > a.n_list) return set_err("qd_aa_trace_part_async: pixel range outside the list");
    Params P; fill_common(P, cam, st);
    P.mode = 2; P.aa_pixels = a.list + first_pixel; P.n_samples = n_pixels * 4u;
    P.n_units = (P.n_samples + kUnitPx - 1u) / kUnitPx;
    P.rgb = taps_dev; P.rgb_floats = (size_t)P.n_samples * 3; P.stats = stats_dev;
    return launch(s, P, st->max_trace_depth, stream);
}

// Step 3: combine first-pass colour and taps, apply the bucket-ordered mark / replace rule, write the frame.
constexpr int kAaBatch = 8;          // iterations of the rule enqueued between two looks at their "changed" flags
int aa_end(qd_scene* s, const qd_settings* st, const float* taps_dev, float* rgb_dev) {
    AaPlan& a = s->aa;
    const int W = a.W, H = a.H;
    if (!a.first || st->frame_width != W || st->frame_height != H) return set_err("qd_aa_end_device: no plan for this frame (qd_aa_begin_device first)");
    if (a.traced && !taps_dev) return set_err("qd_aa_end_device: taps missing");
    const size_t npx = (size_t)W * H;
    float *d_a = nullptr, *d_b = nullptr, *d_aa = nullptr; int* d_changed = nullptr;
    auto cleanup = [&]() { aa_free(s); };
#define TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return -1; } } while (0)
    if (stage(s, kStAaA, npx * 3 * sizeof(float), (void**)&d_a) || stage(s, kStAaB, npx * 3 * sizeof(float), (void**)&d_b) ||
        stage(s, kStAaFlags, kAaBatch * sizeof(int), (void**)&d_changed)) { cleanup(); return -1; }
    if (a.traced) {
        if (stage(s, kStAaAa, (size_t)a.n_list * 3 * sizeof(float), (void**)&d_aa)) { cleanup(); return -1; }
        hipLaunchKernelGGL(aa_combine_kernel, dim3((a.n_list + 255u) / 256u), dim3(256), 0, nullptr, a.first, taps_dev, a.list, a.n_list, W, d_aa);
    }
    // Fixed point of the bucket-ordered rule; iteration k is exact for every pixel whose chain of earlier-bucket
    // neighbours is shorter than k, so the bucket count bounds the loop.  kAaBatch iterations are enqueued back to back,
    // each with a flag of its own, and the flags are read once: an iteration after the fixed point changes nothing, so
    // running on to the end of the batch costs a few launches and saves a round trip per iteration.
    TRY2(hipMemcpyAsync(d_a, a.first, npx * 3 * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    float *cur = d_a, *next = d_b;
    const dim3 blk(64), grd((unsigned)(W + 63) / 64, (unsigned)H);
    const int max_iter = ((W - 1) / 48 + 1) * ((H - 1) / 48 + 1) + 2;
    bool settled = false;
    for (int iter = 0; !settled; iter += kAaBatch) {
        if (iter > max_iter) { cleanup(); return set_err("qd_antialias: mark / replace rule did not settle"); }
        TRY2(hipMemsetAsync(d_changed, 0, kAaBatch * sizeof(int), nullptr));
        for (int k = 0; k < kAaBatch; ++k) {
            hipLaunchKernelGGL(aa_resolve_kernel, grd, blk, 0, nullptr, a.first, cur, next, d_aa, a.slot, W, H, st->want_adaptive_aa, st->aa_threshold,
                               st->show_aa, st->aa_debug_color[0], st->aa_debug_color[1], st->aa_debug_color[2], d_changed + k, s->error_flag + kErrAa);
            std::swap(cur, next);
        }
        int changed[kAaBatch];
        TRY2(hipMemcpy(changed, d_changed, sizeof changed, hipMemcpyDeviceToHost));
        for (int k = 0; k < kAaBatch; ++k) if (!changed[k]) settled = true;
    }
    int rule_flag = 0;                                     // the rule's own word (the sets that traced the taps report through qd_scene_check / their own calls)
    TRY2(hipMemcpy(&rule_flag, s->error_flag + kErrAa, sizeof rule_flag, hipMemcpyDeviceToHost));
    if (rule_flag) {
        (void)hipMemset(s->error_flag + kErrAa, 0, sizeof rule_flag);
        cleanup();
        return set_err("qd_antialias: marked pixel outside the traced set");
    }
    TRY2(hipMemcpy(rgb_dev, cur, npx * 3 * sizeof(float), hipMemcpyDeviceToDevice));
    cleanup();
#undef TRY2
    return 0;
}

// rgb: host buffer with `pitch`, or (rgb_dev) a contiguous W x H x 3 device buffer; in place either way
int antialias_impl(qd_scene* s, const qd_camera* cam, const qd_settings* st, float* rgb, size_t pitch, float* rgb_dev, qd_stats* stats) {
    if (!s || !cam || !st || !(rgb || rgb_dev)) return set_err("qd_antialias: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    const int W = st->frame_width, H = st->frame_height;
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return set_err("qd_antialias: frame size out of range");
    if (!rgb_dev && pitch < (size_t)W * 3) return set_err("qd_antialias: row pitch smaller than the frame");
    const size_t npx = (size_t)W * H;
    float *d_frame = nullptr, *d_taps = nullptr;
    Staging& g = s->stg;
    auto cleanup = [&]() { aa_free(s); };
#define TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return -1; } } while (0)
    float* frame = rgb_dev;
    if (!rgb_dev) {
        if (stage(s, kStAaFrame, npx * 3 * sizeof(float), (void**)&d_frame)) return -1;
        TRY2(hipMemcpy2D(d_frame, (size_t)W * 3 * sizeof(float), rgb, pitch * sizeof(float), (size_t)W * 3 * sizeof(float), H, hipMemcpyHostToDevice));
        frame = d_frame;
    }
    if (!g.e0) { TRY2(hipEventCreate(&g.e0)); TRY2(hipEventCreate(&g.e1)); }
    TRY2(hipMemsetAsync(s->stats, 0, 24 * sizeof(unsigned long long), nullptr));
    TRY2(hipMemsetAsync(s->stats + 16, 0xff, sizeof(unsigned long long), nullptr)); TRY2(hipMemsetAsync(s->stats + 19, 0xff, sizeof(unsigned long long), nullptr));
    uint32_t n_list = 0;
    if (aa_begin(s, st, frame, &n_list)) { cleanup(); return -1; }
    float ms = 0;
    if (n_list) {
        if (stage(s, kStAaTaps, (size_t)n_list * 4 * 3 * sizeof(float), (void**)&d_taps)) { cleanup(); return -1; }
        WorkSet* w0 = nullptr;
        if (acquire_set(s, nullptr, st->max_trace_depth, &w0)) { cleanup(); return -1; }
        int* const flag = s->error_flag + (w0 - s->sets);          // the error word of the set these launches use
        TRY2(hipEventRecord(g.e0, nullptr));
        if (aa_trace(s, cam, st, 0, n_list, d_taps, s->stats, nullptr)) { cleanup(); return -1; }
        TRY2(hipEventRecord(g.e1, nullptr));
        TRY2(hipEventSynchronize(g.e1));
        TRY2(hipEventElapsedTime(&ms, g.e0, g.e1));
        int err_flag = 0;
        TRY2(hipMemcpy(&err_flag, flag, sizeof(int), hipMemcpyDeviceToHost));
        if (err_flag) { (void)hipMemset(flag, 0, sizeof(int)); cleanup(); return set_err("qd_antialias: ray queue overflow on device"); }
    }
    if (aa_end(s, st, d_taps, frame)) { cleanup(); return -1; }
    if (!rgb_dev) TRY2(hipMemcpy2D(rgb, pitch * sizeof(float), frame, (size_t)W * 3 * sizeof(float), (size_t)W * 3 * sizeof(float), H, hipMemcpyDeviceToHost));
    if (stats) {
        unsigned long long h[16];
        TRY2(hipMemcpy(h, s->stats, sizeof h, hipMemcpyDeviceToHost));
        stats->rays = h[0]; stats->primary_rays = h[1]; stats->shadow_rays = h[2]; stats->depth_killed = h[3];
        stats->kd_nodes_visited = h[4]; stats->tris_tested = h[5]; stats->mesh_queries = h[6];
        stats->kernel_ms = ms;
    }
    cleanup();
#undef TRY2
    return 0;
}
#undef TRY3
}  // namespace

extern "C" int qd_aa_begin_device(qd_scene* s, const qd_settings* st, const float* rgb_dev, uint32_t* n_listed) {
    if (!s || !st || !rgb_dev) return set_err("qd_aa_begin_device: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    return aa_begin(s, st, rgb_dev, n_listed);
}

extern "C" int qd_aa_trace_part_async(qd_scene* s, const qd_camera* cam, const qd_settings* st, uint32_t first_pixel, uint32_t n_pixels,
                                      float* taps_dev, qd_stats* stats_dev, void* hip_stream) {
    if (!s || !cam || !st || (!taps_dev && n_pixels)) return set_err("qd_aa_trace_part_async: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (stats_dev) HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(qd_stats), (hipStream_t)hip_stream));
    return aa_trace(s, cam, st, first_pixel, n_pixels, taps_dev, (unsigned long long*)stats_dev, (hipStream_t)hip_stream);
}

extern "C" int qd_aa_end_device(qd_scene* s, const qd_settings* st, const float* taps_dev, float* rgb_dev) {
    if (!s || !st) return set_err("qd_aa_end_device: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!rgb_dev) { HIP_TRY(hipDeviceSynchronize()); aa_free(s); return 0; }     // a rank that only traced a slice drops its plan
    return aa_end(s, st, taps_dev, rgb_dev);
}

extern "C" int qd_antialias(qd_scene* s, const qd_camera* cam, const qd_settings* st, float* rgb, size_t pitch, qd_stats* stats) {
    if (!rgb) return set_err("qd_antialias: null argument");
    return antialias_impl(s, cam, st, rgb, pitch, nullptr, stats);
}

extern "C" int qd_antialias_device(qd_scene* s, const qd_camera* cam, const qd_settings* st, float* rgb_dev, qd_stats* stats) {
    if (!rgb_dev) return set_err("qd_antialias_device: null argument");
    return antialias_impl(s, cam, st, nullptr, 0, rgb_dev, stats);
}

extern "C" int qd_render_tiles_async(qd_scene* s, const qd_camera* cam, const qd_settings* st, int tile, int tile_phase,
                                     int tile_stride, float* rgb_dev, qd_stats* stats_dev, void* hip_stream) {
    if (!s || !cam || !st || !rgb_dev) return set_err("qd_render_tiles_async: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (tile < 8 || tile % 8 || tile_stride < 1 || tile_phase < 0 || tile_phase >= tile_stride) return set_err("qd_render_tiles_async: bad tiling");
    hipStream_t stream = (hipStream_t)hip_stream;
    Params P; fill_common(P, cam, st);
    P.mode = 1; P.tile = tile; P.phase = tile_phase; P.stride = tile_stride;
    P.tiles_x = (st->frame_width + tile - 1) / tile;
    const int tiles_y = (st->frame_height + tile - 1) / tile;
    const int n_tiles = P.tiles_x * tiles_y;
    const int n_owned = n_tiles > tile_phase ? (n_tiles - tile_phase + tile_stride - 1) / tile_stride : 0;
    P.n_units = (uint32_t)n_owned * (uint32_t)((tile / kUnitSide) * (tile / kUnitSide));
    P.rgb = rgb_dev; P.rgb_floats = (size_t)n_owned * tile * tile * 3;
    P.stats = (unsigned long long*)stats_dev;   // first 7 fields of qd_stats are the 64-bit counters
    if (stats_dev) HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(qd_stats), stream));
    if (P.n_units == 0) return 0;
    return launch(s, P, st->max_trace_depth, stream);
}

// The tile-sharded render without a single allocation: queues and control words live in the static `scratch` array
// (through this device's g_scratch_ptr), the frame is ONE launch of the trace kernel (no hand-over lists, so waves keep
// their ray trees to the end) on as many lanes as the scratch holds for this scene's queue depth.
extern "C" int qd_render_tiles_noalloc_async(qd_scene* s, const qd_camera* cam, const qd_settings* st, int tile, int tile_phase,
                                             int tile_stride, float* rgb_dev, qd_stats* stats_dev, void* hip_stream) {
    if (!s || !cam || !st || !rgb_dev) return set_err("qd_render_tiles_noalloc_async: null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (tile < 8 || tile % 8 || tile_stride < 1 || tile_phase < 0 || tile_phase >= tile_stride) return set_err("qd_render_tiles_noalloc_async: bad tiling");
    if (g_device < 0 || g_device >= kMaxDevices || !g_scratch_ptr[g_device]) return set_err("qd_render_tiles_noalloc_async: qd_init has not bound this device's scratch");
    unsigned char* const ptr = g_scratch_ptr[g_device];
    hipStream_t stream = (hipStream_t)hip_stream;
    // queue depth this scene needs at this trace depth (same bound as acquire_set) -> lanes the scratch can hold
    const uint64_t cap64 = (uint64_t)s->spawn0 + (uint64_t)s->spawn1 * st->max_trace_depth + 2;
    const size_t lane_bytes = (size_t)cap64 * kEntryWords * 8;
    const size_t max_lanes = (kScratchBytes - kScratchCtlBytes) / lane_bytes;
    uint32_t waves = (uint32_t)std::min<size_t>(max_lanes / 64, s->n_lanes / 64);
    waves &= ~7u;                                           // whole groups of eight: one wave per XCD label and queue
    const uint32_t min_waves = (uint32_t)(g_num_cu > 0 ? g_num_cu : 256);
    if (cap64 > 4096 || waves < min_waves)
        return set_err("qd_render_tiles_noalloc_async: the ray queues of this scene at this maxTraceDepth (" + std::to_string(cap64) +
                       " entries of 88 B per lane, one wave per CU at least) exceed the 96 MiB static scratch; use qd_render_tiles_async");
    Params P; fill_common(P, cam, st);
    P.mode = 1; P.tile = tile; P.phase = tile_phase; P.stride = tile_stride;
    P.tiles_x = (st->frame_width + tile - 1) / tile;
    const int tiles_y = (st->frame_height + tile - 1) / tile;
    const int n_tiles = P.tiles_x * tiles_y;
    const int n_owned = n_tiles > tile_phase ? (n_tiles - tile_phase + tile_stride - 1) / tile_stride : 0;
    P.n_units = (uint32_t)n_owned * (uint32_t)((tile / kUnitSide) * (tile / kUnitSide));
    P.rgb = rgb_dev; P.rgb_floats = (size_t)n_owned * tile * tile * 3;
    P.stats = (unsigned long long*)stats_dev;
    if (stats_dev) HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(qd_stats), stream));
    if (P.n_units == 0) return 0;
    uint32_t* ctl = (uint32_t*)ptr;
    P.stack = (uint64_t*)(ptr + kScratchCtlBytes); P.stack_cap = (uint32_t)cap64; P.n_lanes = waves * 64u;
    P.error_flag = s->error_flag + kErrNoAlloc;
    P.queue_ctr = ctl; P.out_count = ctl + 8 * kCtlStride; P.pass_done = ctl + 16 * kCtlStride;
    P.source = 0; P.grace = 0xffffffffu;                    // one pass that runs to the end: no task lists
    P.in_count = nullptr; P.in_list = nullptr; P.out_list = nullptr; P.list_cap = 0;
    // the scratch keeps the last call's contents: the control words must start at zero, the queues need not (every lane
    // starts with an empty queue and only reads what it pushed)
    HIP_TRY(hipMemsetAsync(ptr, 0, kScratchCtlBytes, stream));
    HIP_TRY(hipMemsetAsync(P.rgb, 0, P.rgb_floats * sizeof(float), stream));
    using Kernel = void (*)(const DScene, const Params);
    static const Kernel plain[8] = {trace_kernel<false, 0, false>, trace_kernel<false, 1, false>, trace_kernel<false, 2, false>, trace_kernel<false, 3, false>,
                                    trace_kernel<false, 4, false>, trace_kernel<false, 5, false>, trace_kernel<false, 6, false>, trace_kernel<false, 7, false>};
    static const Kernel rich[8] = {trace_kernel<false, 0, true>, trace_kernel<false, 1, true>, trace_kernel<false, 2, true>, trace_kernel<false, 3, true>,
                                   trace_kernel<false, 4, true>, trace_kernel<false, 5, true>, trace_kernel<false, 6, true>, trace_kernel<false, 7, true>};
    const Kernel kernel = g_counting ? trace_kernel<true, kGeomAll, true>
                        : (s->geom_mask & kGeomCsgDeep) ? trace_kernel<false, kGeomAll, true> : (s->rich ? rich : plain)[s->geom_mask & 7];
    const uint32_t grid = P.n_units < waves ? P.n_units : waves;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, stream, s->d, P);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int qd_scene_bind_stream(qd_scene* s, int max_trace_depth, void* hip_stream) {
    if (!s || max_trace_depth < 0) return set_err("qd_scene_bind_stream: bad argument");
    std::lock_guard<std::mutex> lock(s->mu);
    WorkSet* w = nullptr;
    return acquire_set(s, (hipStream_t)hip_stream, (uint32_t)max_trace_depth, &w);
}

extern "C" int qd_untile_async(const float* src_dev, float* dst_dev, int width, int height, int tile, int tile_stride,
                               int tiles_per_rank_padded, void* hip_stream) {
    if (!src_dev || !dst_dev || width <= 0 || height <= 0 || tile < 8 || tile_stride < 1) return set_err("qd_untile_async: bad argument");
    hipLaunchKernelGGL(untile_kernel, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)hip_stream, src_dev, dst_dev, width, height,
                       tile, tile_stride, tiles_per_rank_padded);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int qd_quantize_bgr8_async(const float* rgb_dev, uint8_t* bgr_dev, int width, int height, size_t row_bytes, void* hip_stream) {
    if (!rgb_dev || !bgr_dev || width <= 0 || height <= 0 || row_bytes < (size_t)width * 3) return set_err("qd_quantize_bgr8_async: bad argument");
    hipLaunchKernelGGL(quantize_kernel, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)hip_stream, rgb_dev, bgr_dev, width, height, row_bytes);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int qd_pack_rgb32_async(const float* rgb_dev, size_t row_pitch_floats, uint32_t* out_dev, int width, int height, int use_srgb,
                                   int red_shift, int green_shift, int blue_shift, void* hip_stream) {
    if (!rgb_dev || !out_dev || width <= 0 || height <= 0 || row_pitch_floats < (size_t)width * 3) return set_err("qd_pack_rgb32_async: bad argument");
    for (int sh : {red_shift, green_shift, blue_shift}) if (sh < 0 || sh > 24) return set_err("qd_pack_rgb32_async: channel shift outside 0..24");
    hipLaunchKernelGGL(pack_rgb32_kernel, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)hip_stream, rgb_dev, row_pitch_floats, out_dev,
                       width, height, use_srgb, red_shift, green_shift, blue_shift);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Device error flag of the last asynchronous launches on this scene (ray queue overflow): 0 = fine.
extern "C" int qd_scene_check(qd_scene* s) {
    if (!s) return set_err("qd_scene_check: null scene");
    std::lock_guard<std::mutex> lock(s->mu);
    int f[kErrFlags];
    HIP_TRY(hipMemcpy(f, s->error_flag, sizeof f, hipMemcpyDeviceToHost));
    bool any = false;
    for (int i = 0; i < kErrFlags; ++i) any |= f[i] != 0;
    read_overflow_words(s);
    if (any) {
        (void)hipMemset(s->error_flag, 0, sizeof f);
        return set_err("ray queue overflow on device");
    }
    return 0;
}

extern "C" int qd_set_wavefront_budget(uint64_t queue_bytes, uint32_t max_entries) {
    g_wf_budget = queue_bytes ? queue_bytes : kWfDefaultBudget;
    g_wf_max_entries = max_entries;
    return 0;
}

extern "C" int qd_set_frame_parts(int parts) {
    if (parts < 0 || parts > (int)kMaxFrameParts) return set_err("qd_set_frame_parts: 0 (default) or 1 .. 4");
    g_frame_parts = parts ? (uint32_t)parts : kMaxFrameParts;
    return 0;
}

extern "C" int qd_scene_frame_parts(const qd_scene* s, uint32_t* last_parts) {
    if (!s) return set_err("qd_scene_frame_parts: null scene");
    if (last_parts) *last_parts = s->last_parts;
    return 0;
}

extern "C" int qd_scene_wavefront_info(const qd_scene* cs, int width, int height, uint32_t* bands, int* overflowed) {
    if (!cs || width <= 0 || height <= 0) return set_err("qd_scene_wavefront_info: bad argument");
    qd_scene* s = const_cast<qd_scene*>(cs);
    std::lock_guard<std::mutex> lock(s->mu);
    const uint64_t n_slots = (uint64_t)((width + kUnitSide - 1) / kUnitSide) * (uint64_t)((height + kUnitSide - 1) / kUnitSide) * kUnitPx;
    read_overflow_words(s);
    const WfPlan plan = wavefront_plan(s, n_slots);
    if (bands) *bands = (uint32_t)((n_slots + plan.band_slots - 1) / plan.band_slots);
    if (overflowed) *overflowed = s->wavefront_overflowed ? 1 : 0;
    return 0;
}

#include "qdcomm.h"
#include "prep.h"
