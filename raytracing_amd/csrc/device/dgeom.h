// Device intersection routines: analytic primitives, CSG, heightfield DDA and the
// stack-based KD-tree traversal + triangle test for meshes.
//
// Arithmetic restates the reference function by function (cited below).  Two deliberate
// structural differences from the CPU code, neither of which changes a result:
//  * per-hit shading data (normal, uv, dNdx/dNdy) is NOT produced per candidate but once for
//    the winning hit (the reference recomputes it on every passing triangle, src/mesh.cpp:141-163,
//    only the last write survives);
//  * the mesh KD-tree is walked iteratively with a parametric [tmin,tmax] interval and a private
//    stack instead of the recursive, box-gated walk of src/mesh.cpp:58-92.  The triangle arithmetic,
//    the running-distance reject (`gamma > distance`, equal passes, later triangle wins) and the leaf
//    order are identical, so the result is the same closest hit (measured 0
//    mismatches between the recursive walk and brute force); only exact-tie rays can differ.
#pragma once
#include "dmath.h"
#include "qdtrace.h"

namespace qd {

struct DScene {
    const qd_node* nodes; uint32_t n_nodes; uint32_t n_lights;
    const qd_geom* geoms;
    const qd_mesh* meshes;
    const qd_heightfield* hfields;
    const qd_shader* shaders;
    const qd_layer* layers;
    const qd_texture* textures;
    const qd_proc_line* proc_lines;
    const qd_bitmap* bitmaps;
    const qd_light* lights;
    const double* tri_verts;
    const double* tri_gnormal;
    const double* tri_shade;
    const qd_kdnode* kd_nodes;
    const uint32_t* leaf_tris;    // device copy holds GLOBAL triangle indices (mesh tri_first added)
    const double* leaf_verts;     // 9 doubles per leaf reference, in leaf order: a leaf's triangles are contiguous
    const float* texels;
    const float* hf_f32;
    const double* hf_f64;
    const int* node_kd_slot;      // per scene node: which walk launch of the wavefront pipeline answers it (KD-tree meshes), else -1
    int env_present, env_bilinear;
    int env_faces[6];
};

// Object-space hit record of one geometry query.
struct GHit {
    double dist;
    DV ip, n;
    double u, v;
    double l2, l3;      // mesh barycentrics of the winning triangle
    int tri;            // mesh-local triangle index or -1
    int mesh;           // mesh index when tri >= 0
    int dnd;            // 0: dNdx/dNdy left at zero (sphere, cube, polygon), 1: (1,0,0)/(0,0,1), 2: from the mesh triangle
};

struct Counters {
    uint32_t kd_nodes, tris, mesh_queries;
    unsigned long long t_desc, t_leaf, t_a, t_b, t_c, t_d;   // diagnostic build only (QD_PROFILE): wave cycles per region
};
#ifdef QD_PROFILE
#define QD_STAMP() __builtin_amdgcn_s_memtime()
#else
#define QD_STAMP() 0ull
#endif

constexpr int kKdStack = 72;
#ifndef QD_LDS_STACK
#define QD_LDS_STACK 8
#endif
constexpr int kLdsStack = QD_LDS_STACK;     // stack levels kept in LDS: 8 x 64 lanes x 8 bytes = 4 KB per wave

// KD node as one 16-byte load (hipMalloc'd arrays are 256-byte aligned, nodes are 16 bytes)
struct KdNode { double split; uint32_t meta, count; };
QD_DEV KdNode ld_kdnode(const DScene& S, uint32_t i) {
    const uint4 v = reinterpret_cast<const uint4*>(S.kd_nodes)[i];
    KdNode n;
    n.split = __hiloint2double((int)v.y, (int)v.x);
    n.meta = v.z; n.count = v.w;
    return n;
}   // MAX_TREE_DEPTH 64 -> leaves at depth <= 65 (src/constants.h:13, src/mesh.cpp:218)

// ---------------------------------------------------------------- mesh: triangle test
// The mesh query is compiled as a real function (not inlined into the kernel): the traversal loop then gets
// a register allocation of its own and the kernel's long-lived state is saved once around the call instead of
// being spilled and reloaded inside the loop.  Pointers cross the call as global-address-space pointers so the
// loop uses global_load, not flat_load.
#define QD_GLOBAL __attribute__((address_space(1)))
typedef const QD_GLOBAL double* gp_f64;
typedef const QD_GLOBAL uint32_t* gp_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const QD_GLOBAL u32x4* gp_u32x4;

struct MeshData {            // the arrays a mesh query reads, by value
    const qd_mesh* meshes;
    const qd_kdnode* kd_nodes;
    const double* leaf_verts;
    const uint32_t* leaf_tris;
    const double* tri_verts;
    const double* tri_gnormal;
};
QD_DEV MeshData mesh_data(const DScene& S) { return MeshData{S.meshes, S.kd_nodes, S.leaf_verts, S.leaf_tris, S.tri_verts, S.tri_gnormal}; }

QD_DEV KdNode kd_unpack(const u32x4 v) {
    KdNode n;
    n.split = __hiloint2double((int)v.y, (int)v.x);
    n.meta = v.z; n.count = v.w;
    return n;
}

// component pick without control flow (the ternary on doubles compiles to branches inside the traversal loop)
QD_DEV double pick(const DV& a, uint32_t axis) {
    const bool is0 = axis == 0u, is1 = axis == 1u;
    const int lo = is0 ? __double2loint(a.x) : (is1 ? __double2loint(a.y) : __double2loint(a.z));
    const int hi = is0 ? __double2hiint(a.x) : (is1 ? __double2hiint(a.y) : __double2hiint(a.z));
    return __hiloint2double(hi, lo);
}

// Mesh::Intersect(ray, triangle, info), src/mesh.cpp:111-168.  Returns true when the triangle
// replaces the running hit.  The three quotients are only formed once the sign tests (which are
// exact consequences of `gamma < 0`, `lambda2 < 0`, `lambda3 < 0`) have passed.
struct TriV { DV A, AB, AC; };
template <class P> QD_DEV TriV ld_tri(P tv) { TriV t; t.A = ld3(tv); t.AB = ld3(tv + 3); t.AC = ld3(tv + 6); return t; }

template <bool STATS, class GetGlobal>
QD_DEV bool tri_test(gp_f64 gnormal, uint32_t flags, const TriV& tr, GetGlobal global_index, const DV& o, const DV& d, const DV& md,
                     double& best, double& l2o, double& l3o, uint32_t& n_tests) {
    const DV A = tr.A, AB = tr.AB, AC = tr.AC;
    const DV N = cross(AB, AC);
    const double dcr = dot(N, md);                 // Det(AB, AC, -D)
    if (flags & QD_MESH_BACKCULL) {
        // reference: ray.dir * geometryNormal > 0 with geometryNormal = N * (1/|N|).  dir.N == -dcr exactly;
        // its sign equals the sign of dir.gnormal unless the product is within rounding of zero, where the
        // stored unit normal decides.
        const double mag = fabs(N.x * d.x) + fabs(N.y * d.y) + fabs(N.z * d.z);
        bool cull;
        if (fabs(dcr) > 1e-12 * mag) cull = dcr < 0;
        else cull = dot(d, ld3(gnormal + (size_t)global_index() * 3)) > 0;
        if (cull) return false;
    }
    if (STATS) n_tests++;
    if (fabs(dcr) < 1e-12) return false;
    const DV H = o - A;
    const double gn = dot(N, H);
    const bool dneg = dcr < 0;
    if (gn != 0 && ((gn < 0) != dneg)) return false;          // gamma < 0
    const double l2n = dot(cross(H, AC), md);
    if (l2n != 0 && ((l2n < 0) != dneg)) return false;        // lambda2 < 0
    const double l3n = dot(cross(AB, H), md);
    if (l3n != 0 && ((l3n < 0) != dneg)) return false;        // lambda3 < 0
    const double gamma = gn / dcr;
    if (gamma < 0 || gamma > best) return false;
    const double l2 = l2n / dcr, l3 = l3n / dcr;
    if (l2 < 0 || l3 < 0 || l2 + l3 > 1) return false;
    best = gamma; l2o = l2; l3o = l3;
    return true;
}

struct MHit { double dist, l2, l3; int tri; uint32_t n_nodes, n_tris; };

// Closest hit of one mesh (object space).  Replaces Mesh::Intersect(ray, info) src/mesh.cpp:30-56 and the
// recursive KD walk :58-92.
// any-hit mode (shadow rays): ShadingHelper::GetShadowTransparency (src/shadinghelper.cpp:29-37) only asks whether
// the node's closest hit satisfies Sqr(distance) < targetDistSq.  The closest hit satisfies it iff ANY accepted
// candidate does (x -> Sqr(x / len) is monotone), so the walk stops at the first such candidate and never looks
// beyond the light.
// Returns tri = -1 for a miss.  n_nodes / n_tris are filled in the counting build.
// LDS_LEVELS: far-child stack levels kept in LDS (a kernel that runs fewer waves per CU affords more)
template <bool STATS, int LDS_LEVELS = kLdsStack>
__device__ __attribute__((noinline)) MHit mesh_query(MeshData D, int mesh_idx, DV o, DV d, bool any, double target, double rlen) {
    MHit out;
    out.dist = kInf; out.l2 = 0; out.l3 = 0; out.tri = -1; out.n_nodes = 0; out.n_tris = 0;
    const QD_GLOBAL qd_mesh& m = ((const QD_GLOBAL qd_mesh*)D.meshes)[mesh_idx];
    const gp_f64 gnormal = (gp_f64)D.tri_gnormal;
    const DV md = -d;
    // slab interval of the ray in the mesh box, padded like BBox::IsInside's 1e-6 (src/bbox.cpp:24-31)
    double t0 = 0., t1 = kInf;
    const DV inv = mk(1. / d.x, 1. / d.y, 1. / d.z);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double oa = comp(o, a), da = comp(d, a);
        const double lo = m.bbox_min[a] - 1e-6, hi = m.bbox_max[a] + 1e-6;
        if (da == 0.) { if (oa < lo || oa > hi) return out; continue; }
        double ta = (lo - oa) * comp(inv, a), tb = (hi - oa) * comp(inv, a);
        if (ta > tb) { const double s = ta; ta = tb; tb = s; }
        t0 = fmax(t0, ta); t1 = fmin(t1, tb);
    }
    if (any) t1 = fmin(t1, dsqrt(target) * rlen * (1. + 1e-9) + 1e-7);   // nothing beyond the light matters
    if (t0 > t1) return out;

    const uint32_t flags = m.flags, tri_first = m.tri_first;
    double best = kInf, l2 = 0, l3 = 0;
    int tri = -1;
    uint32_t n_nodes = 0, n_tris = 0;
    if (m.kd_root < 0) {
        const uint32_t n_tri = m.tri_count;
        const gp_f64 tv = (gp_f64)D.tri_verts;
        for (uint32_t i = 0; i < n_tri; ++i) {
            const TriV tr = ld_tri(tv + (size_t)(tri_first + i) * 9);
            if (tri_test<STATS>(gnormal, flags, tr, [&]() { return tri_first + i; }, o, d, md, best, l2, l3, n_tris)) {
                tri = (int)i;
                if (any && sqr(best / rlen) < target) break;
            }
        }
    } else {
        // Traversal as one loop with one trip to memory per iteration.  A lane is either at a node (fetch 16
        // bytes, step down or open a leaf) or inside a leaf (fetch one triangle, test it); both kinds of lanes
        // issue their loads before anybody waits, so lanes at different stages of their walk share every trip
        // instead of waiting for each other at leaf boundaries.  The top of the far-child stack is kept in
        // registers: a pop re-fills it from scratch without anything waiting on that load.
        // Far-child stack: the top entry in registers, the next kLdsStack entries in LDS (8 bytes each: token and the
        // interval end rounded UP to float, which only widens the interval), deeper ones in scratch.  The workgroup
        // is one wave, so the LDS block belongs to this wave alone.
        __shared__ uint2 lds_stack[LDS_LEVELS][64];
        uint2 stk_deep[kKdStack - LDS_LEVELS];
        const uint32_t lane_id = threadIdx.x & 63u;
        auto stack_put = [&](int at, uint32_t tok, double t) {
            float f = (float)t;
            if ((double)f < t) f = __uint_as_float(__float_as_uint(f) + (f >= 0.f ? 1u : 0xffffffffu));   // next float up
            const uint2 e = make_uint2(tok, __float_as_uint(f));
            if (at < LDS_LEVELS) lds_stack[at][lane_id] = e; else stk_deep[at - LDS_LEVELS] = e;
        };
        auto stack_get = [&](int at, uint32_t& tok, double& t) {
            const uint2 e = at < LDS_LEVELS ? lds_stack[at][lane_id] : stk_deep[at - LDS_LEVELS];
            tok = e.x; t = (double)__uint_as_float(e.y);
        };
        int sp = 0;                                   // entries in use; entry sp-1 lives in top_*, the rest in stk_*[0 .. sp-2]
        uint32_t top_node = 0;
        double top_t = 0.;
        // position in the tree: record index (28 bits) | sub << 30; sub 0 = the record's own node, 1 / 2 = only its
        // left / right child (a far child queued from inside a record), see kdrefine.h for the record format
        uint32_t node = (uint32_t)m.kd_root;
        uint32_t ref = 0, left = 0;                   // inside a leaf while left != 0: next leaf reference, references left
        uint32_t hit_ref = 0xffffffffu;
        double tmin = t0, tmax = t1;
        const uint32_t dpos = (d.x > 0 ? 1u : 0u) | (d.y > 0 ? 2u : 0u) | (d.z > 0 ? 4u : 0u);
        const gp_u32x4 kd = (gp_u32x4)D.kd_nodes;
        const gp_f64 leaf_verts = (gp_f64)D.leaf_verts;
        const gp_u32 leaf_tris = (gp_u32)D.leaf_tris;
        TriV tr;
        tr.A = tr.AB = tr.AC = mk(0., 0., 0.);
        bool alive = true;
        bool leaf_done = false;
        // One split plane: returns the side (0 left, 1 right) to continue in and queues the other side's token when
        // the ray's interval straddles the plane.  near side = the side the origin is on (src/mesh.cpp:75-77: right
        // iff start > split); an origin exactly on the plane belongs to the side the ray travels into.  The sign of
        // t is exact.
        auto choose = [&](double split, uint32_t axis, uint32_t tok_left, uint32_t tok_right) -> uint32_t {
            if (STATS) n_nodes++;
            const double oa = pick(o, axis);
            const double t = (split - oa) * pick(inv, axis);
            const uint32_t near_s = (oa > split || (oa == split && ((dpos >> axis) & 1u))) ? 1u : 0u;
            const double slack = 1e-7 + 1e-9 * fabs(t);
            if (t > tmax + slack || t <= 0) return near_s;          // plane beyond the interval, or at/behind the origin
            if (t < tmin - slack) return near_s ^ 1u;               // crossed before the interval starts
            // (also reached when t is NaN: ray inside the plane)
            if (sp <= kKdStack) {
                if (sp > 0) stack_put(sp - 1, top_node, top_t);
                top_node = near_s ? tok_left : tok_right; top_t = tmax; ++sp;
            }
            tmax = (t == t) ? fmin(tmax, t + slack) : tmax;
            return near_s;
        };
        // One loop, one trip to memory per iteration.  A lane is either at a record (fetch 32 bytes, step down two
        // levels or open a leaf) or inside a leaf (fetch one triangle, test it); both kinds of lanes issue their loads
        // before anybody waits, so lanes at different stages of their walk share every trip.
        for (;;) {
            const unsigned long long m_alive = __ballot(alive);
            if (!m_alive) break;
            const bool in_leaf = alive && left != 0u;
#ifdef QD_PROFILE
            {   // diagnostic: lane iterations in n_nodes, wave iterations in n_tris (credited to the first live lane)
                if (alive) n_nodes++;
                if (alive && (unsigned)(__ffsll((long long)m_alive) - 1) == (threadIdx.x & 63u)) n_tris++;
            }
#endif
            leaf_done = false;
            // one trip for both kinds of lanes: everybody issues its loads, then everybody waits once
            const bool at_node = alive && !in_leaf;
            const uint32_t rec = node & 0x3fffffffu, sub = node >> 30;
            u32x4 ra = {0u, 0u, 0u, 0u}, rb = {0u, 0u, 3u, 0u};
            if (in_leaf) tr = ld_tri(leaf_verts + (size_t)ref * 9);
            if (at_node) { ra = kd[2u * rec]; rb = kd[2u * rec + 1u]; }
            asm volatile("" : "+v"(ra), "+v"(rb), "+v"(tr.A.x), "+v"(tr.A.y), "+v"(tr.A.z), "+v"(tr.AB.x), "+v"(tr.AB.y), "+v"(tr.AB.z),
                              "+v"(tr.AC.x), "+v"(tr.AC.y), "+v"(tr.AC.z));
            if (in_leaf) {
                if (tri_test<STATS>(gnormal, flags, tr, [&]() { return leaf_tris[ref]; }, o, d, md, best, l2, l3, n_tris)) {
                    hit_ref = ref;
                    if (any && sqr(best / rlen) < target) alive = false;
                }
                ++ref; --left;
                leaf_done = left == 0u;
            }
            if (at_node) {
                const uint32_t meta = rb.z;
                if ((meta & 3u) == 3u) {                                  // leaf record
                    if (STATS) n_nodes++;
                    ref = ra.x; left = ra.y;
                    leaf_done = left == 0u;
                } else {
                    const uint32_t c = sub ? sub - 1u
                                           : choose(__hiloint2double((int)ra.y, (int)ra.x), meta & 3u, rec | (1u << 30), rec | (2u << 30));
                    const uint32_t ac = (meta >> (2u + 2u * c)) & 3u;
                    const uint32_t lo = c ? rb.x : ra.z, hi = c ? rb.y : ra.w;   // the child's plane, or first / count of a leaf child
                    if (ac == 3u) {
                        if (STATS) n_nodes++;
                        ref = lo; left = hi;
                        leaf_done = left == 0u;
                    } else {
                        const uint32_t cb = ((meta >> 6) << 2) + 2u * c;
                        node = cb + choose(__hiloint2double((int)hi, (int)lo), ac, cb, cb + 1u);
                    }
                }
            }
            if (leaf_done && alive) {
                // front-to-back: a hit inside this leaf's interval cannot be beaten by later leaves
                if (hit_ref != 0xffffffffu && best <= tmax - (2e-7 + 2e-9 * fabs(tmax))) alive = false;
                else if (sp == 0) alive = false;
                else {
                    node = top_node; tmin = tmax; tmax = top_t;
                    --sp;
                    if (sp > 0) stack_get(sp - 1, top_node, top_t);
                }
            }
        }
        if (hit_ref != 0xffffffffu) tri = (int)(leaf_tris[hit_ref] - tri_first);
    }
    out.dist = best; out.tri = tri; out.l2 = l2; out.l3 = l3; out.n_nodes = n_nodes; out.n_tris = n_tris;
    return out;
}

// Shading data of a mesh hit (src/mesh.cpp:141-163), evaluated once for the winner.
QD_DEV void mesh_finish(const DScene& S, GHit& h, DV& dndx, DV& dndy, bool want_dnd) {
    const qd_mesh& m = S.meshes[h.mesh];
    h.dnd = 2;
    const size_t t = (size_t)m.tri_first + (uint32_t)h.tri;
    const double* sh = S.tri_shade + t * 21;
    if (!(m.flags & QD_MESH_FACETED)) {
        const DV nA = ld3(sh), nB = ld3(sh + 3), nC = ld3(sh + 6);
        h.n = normalize(nA + h.l2 * (nB - nA) + h.l3 * (nC - nA));
    } else {
        h.n = ld3(S.tri_gnormal + t * 3);
    }
    h.u = sh[9] + h.l2 * (sh[11] - sh[9]) + h.l3 * (sh[13] - sh[9]);
    h.v = sh[10] + h.l2 * (sh[12] - sh[10]) + h.l3 * (sh[14] - sh[10]);
    if (want_dnd) { dndx = ld3(sh + 15); dndy = ld3(sh + 18); }
}

// ---------------------------------------------------------------- boxes (src/bbox.cpp)
QD_DEV bool box_inside(const double* mn, const double* mx, const DV& p) {
    return is_between(p.x, mn[0], mx[0], 1e-6) && is_between(p.y, mn[1], mx[1], 1e-6) && is_between(p.z, mn[2], mx[2], 1e-6);
}
QD_DEV double box_closest(const double* mn, const double* mx, const DV& start, const DV& dir) {   // :99-137
    if (box_inside(mn, mx, start)) return 0.;
    double best = kInf;
    for (int dim = 0; dim < 3; ++dim) {
        const double d = comp(dir, dim), s = comp(start, dim), lo = mn[dim], hi = mx[dim];
        if (d < 0 && s < lo) continue;
        if (d > 0 && s > hi) continue;
        if (is_zero(d, 1e-9)) continue;
        const double mul = 1. / d;
        const int u = (dim == 0) ? 1 : 0, v = (dim == 2) ? 1 : 2;
        for (int j = 0; j < 2; ++j) {
            const double dist = ((j ? hi : lo) - s) * mul;
            if (dist < 0) continue;
            const double x = comp(start, u) + comp(dir, u) * dist;
            if (is_between(x, mn[u], mx[u], 0)) {
                const double y = comp(start, v) + comp(dir, v) * dist;
                if (is_between(y, mn[v], mx[v], 0)) best = smin(best, dist);
            }
        }
    }
    return best;
}

// IntersectTriangleFast, src/mesh.cpp:542-580
QD_DEV bool tri_fast(const DV& start, const DV& dir, const DV& A, const DV& B, const DV& C, double& dist) {
    const DV AB = B - A, AC = C - A, D = -dir, H = start - A;
    const DV n = cross(AB, AC);
    const double dcr = dot(n, D);
    if (fabs(dcr) < 1e-12) return false;
    const double gamma = dot(n, H) / dcr;
    if (gamma < 0 || gamma > dist) return false;
    const double l2 = dot(cross(H, AC), D) / dcr;
    const double l3 = dot(cross(AB, H), D) / dcr;
    if (l2 < 0 || l3 < 0 || l2 + l3 > 1) return false;
    dist = gamma;
    return true;
}

// ---------------------------------------------------------------- heightfield (src/heightfield.cpp:20-100, :216-245)
// Counting build: n_steps = DDA steps (one maxHeight word + one 64-byte highmap line each), n_cells = cells whose two
// triangles were tested (four heights = 16 B) -- SURVEY.md 8(d)'s units for the heightfield workload.
template <bool STATS>
QD_DEV bool hf_intersect(const DScene& S, int hf, const DV& start, const DV& dir, double& out_dist, uint32_t& n_steps, uint32_t& n_cells) {
    const qd_heightfield& h = S.hfields[hf];
    const float* heights = S.hf_f32 + h.heights_off;
    const float* maxh = S.hf_f32 + h.maxheights_off;
    const float* highmap = S.hf_f32 + h.highmap_off;
    const int W = (int)h.width, H = (int)h.height;
    auto get_height = [&](int x, int y) { x = min(max(x, 0), W - 1); y = min(max(y, 0), H - 1); return heights[(size_t)y * W + x]; };
    const DV step = dir / (dsqrt(sqr(dir.x) + sqr(dir.z)));
    const double dist = box_closest(h.bbox_min, h.bbox_max, start, dir);
    const double mx = 1.0 / dir.x, mz = 1.0 / dir.z;
    DV p = start + dir * (dist + 1e-6);
    while (box_inside(h.bbox_min, h.bbox_max, p)) {
        if (STATS) ++n_steps;
        int x0 = (int)floor(p.x), z0 = (int)floor(p.z);
        if (!(x0 < 0 || x0 >= W || z0 < 0 || z0 >= H) && h.use_optimization) {
            int k = 0;
            while (k < h.max_k && smin(p.y, p.y + dir.y * (1 << k)) > (double)highmap[((size_t)z0 * W + x0) * 16 + k]) ++k;
            --k;
            if (k >= 0) {
                p = p + dir * (1 << k);
                x0 = (int)floor(p.x); z0 = (int)floor(p.z);
            }
        }
        if (x0 < 0 || x0 >= W || z0 < 0 || z0 >= H) break;
        const double lx = dir.x > 0 ? (ceil(p.x) - p.x) * mx : (floor(p.x) - p.x) * mx;
        const double lz = dir.z > 0 ? (ceil(p.z) - p.z) * mz : (floor(p.z) - p.z) * mz;
        const DV p_next = p + step * (smin(lx, lz) + 1e-6);
        if (smin(p.y, p_next.y) < (double)maxh[(size_t)z0 * W + x0]) {
            double closest = kInf;
            if (STATS) ++n_cells;
            const DV A = mk(x0, get_height(x0, z0), z0);
            const DV B = mk(x0 + 1, get_height(x0 + 1, z0), z0);
            const DV C = mk(x0 + 1, get_height(x0 + 1, z0 + 1), z0 + 1);
            const DV D = mk(x0, get_height(x0, z0 + 1), z0 + 1);
            if (tri_fast(start, dir, A, B, D, closest) || tri_fast(start, dir, B, C, D, closest)) {
                out_dist = closest;
                return true;
            }
        }
        p = p_next;
    }
    return false;
}

// The rest of Heightfield::Intersect's outputs (src/heightfield.cpp:61-68) are functions of the hit distance.
QD_DEV void hf_finish(const DScene& S, int hf, const DV& start, const DV& dir, double dist, GHit& out) {
    const qd_heightfield& h = S.hfields[hf];
    const double* normals = S.hf_f64 + h.normals_off;
    const int W = (int)h.width;
    out.dist = dist;
    out.ip = start + dir * dist;
    // GetNormal (:230-245): bilinear with float weights
    const float fx = static_cast<float>(out.ip.x), fy = static_cast<float>(out.ip.z);
    unsigned ux = (unsigned)floorf(fx), uy = (unsigned)floorf(fy);
    const float pw = fx - ux, qw = fy - uy;
    ux = min(ux, h.width - 1); uy = min(uy, h.height - 1);
    const unsigned x1 = min(h.width - 1, ux + 1), y1 = min(h.height - 1, uy + 1);
    auto nrm = [&](unsigned x, unsigned y) { return ld3(normals + ((size_t)y * W + x) * 3); };
    out.n = nrm(ux, uy) * ((1 - pw) * (1 - qw)) + nrm(x1, uy) * ((pw) * (1 - qw)) +
            nrm(ux, y1) * ((1 - pw) * (qw)) + nrm(x1, y1) * ((pw) * (qw));
    out.u = out.ip.x / h.width;
    out.v = out.ip.z / h.height;
    out.tri = -1; out.dnd = 1;
}

// ---------------------------------------------------------------- analytic primitives (src/geometry.cpp)
QD_DEV bool plane_intersect(const qd_geom& g, const DV& start, const DV& dir, GHit& out) {   // :6-28
    const double height = g.p[0], limit = g.p[1];
    if (start.y > height && dir.y >= 0.) return false;
    if (start.y < height && dir.y <= 0.) return false;
    const double sf = (height - start.y) / dir.y;
    const DV ip = start + dir * sf;
    if (fabs(ip.x) > limit || fabs(ip.z) > limit) return false;
    out.ip = ip; out.dist = sf;
    out.n = mk(0., start.y > height ? 1. : -1., 0.);
    out.u = ip.x; out.v = ip.z;
    out.tri = -1; out.dnd = 1;
    return true;
}

QD_DEV bool sphere_intersect(const qd_geom& g, const DV& start, const DV& dir, GHit& out) {  // :49-87
    const DV center = mk(g.p[0], g.p[1], g.p[2]);
    const double R = g.p[3];
    const DV H = start - center;
    const double A = 1;
    const double B = 2 * dot(H, dir);
    const double C = len_sqr(H) - sqr(R);
    const double discr = sqr(B) - 4 * A * C;
    if (discr < 0) return false;
    const double sq = dsqrt(discr);
    const double p1 = (-B - sq) / (2 * A);
    const double p2 = (-B + sq) / (2 * A);
    double p;
    if (p1 > 0) p = p1; else if (p2 > 0) p = p2; else return false;
    out.dist = p;
    out.ip = start + dir * p;
    const DV rel = out.ip - center;
    out.n = normalize(rel);
    out.u = (atan2(rel.z, rel.x) + kPi) / (2 * kPi);
    out.v = -(asin(rel.y / R) + kPi / 2) / kPi;
    out.tri = -1; out.dnd = 0;
    return true;
}

QD_DEV void cube_side(const qd_geom& g, double level, double s, double d, const DV& start, const DV& dir,
                      double nx, double ny, double nz, GHit& out) {                          // :107-148
    const double cx = g.p[0], cy = g.p[1], cz = g.p[2], hs = g.p[3];
    if (s > level && d >= 0) return;
    if (s < level && d <= 0) return;
    const double sf = (level - s) / d;
    const DV ip = start + dir * sf;
    if (ip.y > cy + hs + 1e-6) return;
    if (ip.y < cy - hs - 1e-6) return;
    if (ip.x > cx + hs + 1e-6) return;
    if (ip.x < cx - hs - 1e-6) return;
    if (ip.z > cz + hs + 1e-6) return;
    if (ip.z < cz - hs - 1e-6) return;
    if (out.dist <= sf) return;
    out.ip = ip; out.dist = sf; out.n = mk(nx, ny, nz);
    if (ny == 0) { out.u = ip.x + ip.z; out.v = ip.y; }
    else { out.u = ip.x; out.v = ip.z; }
}

QD_DEV bool cube_intersect(const qd_geom& g, const DV& start, const DV& dir, GHit& out) {    // :150-162
    const double cx = g.p[0], cy = g.p[1], cz = g.p[2], hs = g.p[3];
    out.dist = kInf; out.tri = -1; out.dnd = 0;
    cube_side(g, cx - hs, start.x, dir.x, start, dir, -1, 0, 0, out);
    cube_side(g, cx + hs, start.x, dir.x, start, dir, +1, 0, 0, out);
    cube_side(g, cy - hs, start.y, dir.y, start, dir, 0, -1, 0, out);
    cube_side(g, cy + hs, start.y, dir.y, start, dir, 0, +1, 0, out);
    cube_side(g, cz - hs, start.z, dir.z, start, dir, 0, 0, -1, out);
    cube_side(g, cz + hs, start.z, dir.z, start, dir, 0, 0, +1, out);
    return out.dist < kInf;
}

QD_DEV bool regpoly_intersect(const qd_geom& g, const DV& start, const DV& dir, GHit& out) { // :259-296
    const DV cn = mk(g.p[0], g.p[1], g.p[2]);
    const double R = g.p[3];
    const unsigned sides = (unsigned)g.a;
    if (start.y > cn.y && dir.y >= 0.) return false;
    if (start.y < cn.y && dir.y <= 0.) return false;
    const double sf = (cn.y - start.y) / dir.y;
    const DV ip = start + dir * sf;
    const DV dd = ip - cn;
    double angle = atan2(dd.z, dd.x);
    if (angle < 0) angle += 2 * kPi;
    const double side_angle = 2 * kPi / sides;
    const unsigned k1 = (unsigned)(angle / side_angle);
    const unsigned k2 = (k1 + 1 == sides ? 0 : k1 + 1);
    const DV vk1 = mk(cn.x + cos(k1 * side_angle) * R, cn.y, cn.z + sin(k1 * side_angle) * R);
    const DV vk2 = mk(cn.x + cos(k2 * side_angle) * R, cn.y, cn.z + sin(k2 * side_angle) * R);
    auto cross_scalar = [](const DV& l, const DV& r) { return l.x * r.y + l.y * r.z + l.z * r.x - l.x * r.z - l.y * r.x - l.z * r.y; };
    const int s1 = sign_of(cross_scalar(vk1 - cn, ip - vk1));
    const int s2 = sign_of(cross_scalar(vk2 - vk1, ip - vk2));
    const int s3 = sign_of(cross_scalar(cn - vk2, ip - cn));
    if (!((s1 >= 0 && s2 >= 0 && s3 >= 0) || (s1 <= 0 && s2 <= 0 && s3 <= 0))) return false;
    out.ip = ip; out.dist = sf;
    out.n = mk(0., start.y > cn.y ? 1. : -1., 0.);
    out.u = ip.x - cn.x; out.v = ip.z - cn.z;
    out.tri = -1; out.dnd = 0;
    return true;
}

// IsInside of a non-CSG geometry
QD_DEV bool leaf_inside(const qd_geom& g, const DV& p) {
    switch (g.type) {
        case QD_GEOM_PLANE: return p.y == g.p[0];
        case QD_GEOM_SPHERE: return len_sqr(p - mk(g.p[0], g.p[1], g.p[2])) <= sqr(g.p[3]);
        case QD_GEOM_CUBE: return is_zero(p.x - g.p[0], g.p[3]) & is_zero(p.y - g.p[1], g.p[3]) & is_zero(p.z - g.p[2], g.p[3]);
        case QD_GEOM_REGPOLY: return p.y == g.p[1];
        default: return false;
    }
}

// Plane / Sphere / Cube / RegularPolygon
QD_DEV bool prim_intersect(const qd_geom& g, const DV& start, const DV& dir, GHit& out) {
    switch (g.type) {
        case QD_GEOM_PLANE: return plane_intersect(g, start, dir, out);
        case QD_GEOM_SPHERE: return sphere_intersect(g, start, dir, out);
        case QD_GEOM_CUBE: return cube_intersect(g, start, dir, out);
        case QD_GEOM_REGPOLY: return regpoly_intersect(g, start, dir, out);
        default: return false;
    }
}

QD_DEV bool csg_op(int op, bool a, bool b) { return op == QD_CSG_AND ? (a && b) : (op == QD_CSG_PLUS ? (a || b) : (a && !b)); }

// CsgOp::Intersect, src/geometry.cpp:184-234, for operands that are analytic primitives (the common case: this walk is
// inlined into the node loop).  CSG over CSG / meshes / heightfields: csg_generic below.
QD_DEV bool csg_intersect(const DScene& S, const qd_geom& g, const DV& start, const DV& dir, GHit& out) {
    const qd_geom ga = S.geoms[g.a], gb = S.geoms[g.b];
    GHit li, ri;
    bool inA = leaf_inside(ga, start), inB = leaf_inside(gb, start);
    const bool cur = csg_op(g.c, inA, inB);
    bool lhit = prim_intersect(ga, start, dir, li);
    bool rhit = prim_intersect(gb, start, dir, ri);
    while (lhit || rhit) {
        if (lhit && (!rhit || li.dist <= ri.dist)) {
            inA = !inA;
            if (csg_op(g.c, inA, inB) != cur) { out = li; return true; }
            const double prev = li.dist;
            lhit = prim_intersect(ga, li.ip + dir * 1e-6, dir, li);
            if (lhit) li.dist += prev + 1e-6;
        } else if (rhit && (!lhit || ri.dist <= li.dist)) {
            inB = !inB;
            if (csg_op(g.c, inA, inB) != cur) { out = ri; return true; }
            const double prev = ri.dist;
            rhit = prim_intersect(gb, ri.ip + dir * 1e-6, dir, ri);
            if (rhit) ri.dist += prev + 1e-6;
        }
    }
    return false;
}

// ---------------------------------------------------------------- CSG over any geometry
// CsgOp::Intersect takes any Geometry* as operand (src/geometry.cpp:184-242): a mesh, a heightfield, another CsgOp.
// Scenes that hold such a node run the kernel copy with kGeomCsgDeep; its walk is the same boundary walk as
// csg_intersect above, with operands resolved by these functions.  Device code has no recursion: nesting is unrolled by
// a template depth (kMaxCsgNesting levels of CSG nodes, qd_scene_create refuses deeper trees), each level a real
// function so that the node loop of trace_nodes holds one call site, not the walk.
constexpr int kMaxCsgNesting = 4;

// Mesh::IsInside and Heightfield::IsInside are constant false (src/mesh.cpp:94-97, src/heightfield.h:14)
template <int D> QD_DEV bool geom_inside_t(const DScene& S, int gi, const DV& p) {
    const qd_geom g = S.geoms[gi];
    if (g.type == QD_GEOM_CSG) {
        if constexpr (D > 0) return csg_op(g.c, geom_inside_t<D - 1>(S, g.a, p), geom_inside_t<D - 1>(S, g.b, p));
        else return false;
    }
    return leaf_inside(g, p);
}

// Closest hit of one non-CSG geometry of any kind, as a full record.  A mesh hit carries triangle + barycentrics (its
// normal / uv are derived once for the winner by mesh_finish in assemble_info).
template <bool STATS>
__device__ __attribute__((noinline)) bool leaf_any_intersect(const DScene& S_, int gi, const DV& start, const DV& dir, GHit& out, Counters& cnt) {
    const DScene& S = uniform_const(S_);
    const qd_geom g = S.geoms[gi];
    if (g.type == QD_GEOM_MESH) {
        const MHit mh = mesh_query<STATS>(mesh_data(S), g.a, start, dir, false, 0., 1.);
        if (STATS) { cnt.kd_nodes += mh.n_nodes; cnt.tris += mh.n_tris; cnt.mesh_queries += mh.n_nodes ? 1u : 0u; }
        if (mh.tri < 0) return false;
        out.dist = mh.dist; out.ip = start + dir * mh.dist; out.tri = mh.tri; out.mesh = g.a; out.l2 = mh.l2; out.l3 = mh.l3; out.dnd = 2;
        return true;
    }
    if (g.type == QD_GEOM_HEIGHTFIELD) {
        double dist = 0;
        if (!hf_intersect<STATS>(S, g.a, start, dir, dist, cnt.kd_nodes, cnt.tris)) return false;
        hf_finish(S, g.a, start, dir, dist, out);
        return true;
    }
    return prim_intersect(g, start, dir, out);
}

template <bool STATS, int D>
__device__ __attribute__((noinline)) bool csg_generic(const DScene& S, int gi, const DV& start, const DV& dir, GHit& out, Counters& cnt);

template <bool STATS, int D>
QD_DEV bool geom_any_intersect(const DScene& S, int gi, const DV& start, const DV& dir, GHit& out, Counters& cnt) {
    if (S.geoms[gi].type == QD_GEOM_CSG) {
        if constexpr (D > 0) return csg_generic<STATS, D - 1>(S, gi, start, dir, out, cnt);
        else return false;
    }
    return leaf_any_intersect<STATS>(S, gi, start, dir, out, cnt);
}

// CsgOp::Intersect (src/geometry.cpp:184-234) for the CSG geometry gi whose operands may nest D more levels.
template <bool STATS, int D>
__device__ __attribute__((noinline)) bool csg_generic(const DScene& S_, int gi, const DV& start, const DV& dir, GHit& out, Counters& cnt) {
    const DScene& S = uniform_const(S_);
    const qd_geom g = S.geoms[gi];
    GHit li, ri;
    li.tri = ri.tri = -1;
    bool inA = geom_inside_t<D>(S, g.a, start), inB = geom_inside_t<D>(S, g.b, start);
    const bool cur = csg_op(g.c, inA, inB);
    bool lhit = geom_any_intersect<STATS, D>(S, g.a, start, dir, li, cnt);
    bool rhit = geom_any_intersect<STATS, D>(S, g.b, start, dir, ri, cnt);
    while (lhit || rhit) {
        if (lhit && (!rhit || li.dist <= ri.dist)) {
            inA = !inA;
            if (csg_op(g.c, inA, inB) != cur) { out = li; return true; }
            const double prev = li.dist;
            lhit = geom_any_intersect<STATS, D>(S, g.a, li.ip + dir * 1e-6, dir, li, cnt);
            if (lhit) li.dist += prev + 1e-6;
        } else if (rhit && (!lhit || ri.dist <= li.dist)) {
            inB = !inB;
            if (csg_op(g.c, inA, inB) != cur) { out = ri; return true; }
            const double prev = ri.dist;
            rhit = geom_any_intersect<STATS, D>(S, g.b, ri.ip + dir * 1e-6, dir, ri, cnt);
            if (rhit) ri.dist += prev + 1e-6;
        }
    }
    return false;
}

// Analytic geometry (primitives and CSG of primitives): full hit record.  CSG / PRIMS: whether the scene holds CSG nodes /
// primitives other than planes (see trace_nodes' GEOM mask); what it does not hold is not compiled into the caller.
template <bool CSG, bool PRIMS>
QD_DEV bool analytic_intersect(const DScene& S, const qd_geom& g, const DV& start, const DV& dir, GHit& out) {
    if (CSG && g.type == QD_GEOM_CSG) return csg_intersect(S, g, start, dir, out);
    if (!PRIMS) return plane_intersect(g, start, dir, out);
    return prim_intersect(g, start, dir, out);
}

}  // namespace qd
