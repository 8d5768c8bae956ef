// Wavefront pipeline: the frame as generations of rays in HBM queues instead of per-lane LIFO queues inside one kernel.
//
//   generation g:  wf_walk_kernel   one launch per scene node whose geometry is a KD-tree mesh: every ray of the generation
//                                   (generation 0: made from the pixel slot, no queue) is taken to that node's object space
//                                   -- the node is uniform for the launch, its matrices sit in scalar registers -- clipped
//                                   against the mesh box and walked through the tree; the closest hit (or the first hit
//                                   nearer than the light for a shadow ray) goes to a result record at the ray's index;
//                  wf_shade_kernel  full width over the rays of g, in queue order: Raytrace's node loop (src/main.cpp:23-38)
//                                   with analytic geometry intersected in place and KD meshes answered by the result
//                                   records; a shadow ray adds weight * transparency * light term to its pixel, a radiance
//                                   ray is shaded and its child rays (radiance depth + 1, shadow rays) are appended to
//                                   ray queue g + 1.
//   A frame is walk(0) shade(0) walk(1) ... shade(D) walk(D + 1) shade(D + 1): a fixed sequence of launches, every kernel
//   reads the size of its input from device counters, so nothing comes back to the host (the calls stay enqueue-only).
//
// Why: in the one-kernel form a round lasts as long as its slowest ray (53 % of the lanes busy in the KD loop), shading
// state lives across the traversal (spills, 0.55 KB of scratch written per ray) and each wave keeps its ray trees to
// itself.  Here the walk kernel holds the state of a walk and nothing else, and refills idle lanes from the queue INSIDE
// its loop: a refill is one coalesced 48-byte read and a transform by uniform matrices, so a lane whose ray ends -- a miss
// of the mesh box, a short walk -- takes the next ray while its neighbours are still walking.  Everything else runs
// full width on rays in queue order, which is the order of the pixels (children are appended wave by wave, lane by
// lane), so neighbouring lanes keep walking neighbouring parts of the tree.  Queue traffic is coalesced word-major streams.
// Same arithmetic as the one-kernel form: to_object / analytic_intersect / the triangle test / assemble_info / the shaders
// are the same functions.
#pragma once

// control words per generation, one cache line each: [0] rays queued, [1..8] walk cursors of KD node slot 0 (one per eighth
// of the queue), [9..16] shade cursors, then 8 walk cursors per further KD node slot
constexpr uint32_t kWfMaxKdNodes = 8;            // scene nodes with a KD-tree mesh (more: the one-kernel form renders the scene)
constexpr uint32_t kWfCurShade = 9;
constexpr uint32_t kWfGenWords = (17 + 8 * kWfMaxKdNodes) * kCtlStride;
QD_DEV uint32_t wf_walk_cursor(uint32_t slot) { return slot == 0 ? 1u : 17u + 8u * (slot - 1u); }
constexpr uint32_t kWfNone = 0xffffffffu;
constexpr int kWfResWords = 4;                   // result record of a walk: distance (kInf = miss), triangle, two barycentrics
constexpr uint32_t kWfBlock = 128;               // indices a wave reserves with one atomic
#ifndef QD_WF_WAVES
#define QD_WF_WAVES 4
#endif
#ifndef QD_WF_MIN_ACTIVE
#define QD_WF_MIN_ACTIVE 48      // the KD loop is left for a refill when fewer lanes than this are still walking
#endif

// Appends tasks to the next generation's queue: one atomic per call site and wave (the lanes that reach the call together).
struct WfAppender {
    uint64_t* base; size_t cap; size_t limit; uint32_t* count; uint32_t* over; uint32_t max_depth; uint32_t n_killed;
    unsigned long long lanes_below;
    QD_DEV void push(const Task& t) {
        if (!(t.flags_depth & kFlagShadow) && (t.flags_depth & kDepthMask) > max_depth) { ++n_killed; return; }   // src/main.cpp:20-21
        const unsigned long long m = __ballot(1);
        const uint32_t rank = (uint32_t)__popcll(m & lanes_below);
        uint32_t first = 0;
        if (rank == 0) first = atomicAdd(count, (uint32_t)__popcll(m));
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        const size_t i = (size_t)first + rank;
        if (i >= limit) { *over = 1u; return; }          // the band is rendered again in the one-kernel form (wf_band_end_kernel)
        uint64_t* p = base + i;
        p[0 * cap] = __double_as_longlong(t.o.x); p[1 * cap] = __double_as_longlong(t.o.y); p[2 * cap] = __double_as_longlong(t.o.z);
        p[3 * cap] = __double_as_longlong(t.d.x); p[4 * cap] = __double_as_longlong(t.d.y); p[5 * cap] = __double_as_longlong(t.d.z);
        p[6 * cap] = pack2f(t.w.r, t.w.g);
        p[7 * cap] = pack_fu(t.w.b, t.flags_depth);
        if (t.flags_depth & kFlagShadow) { p[8 * cap] = t.pixel; p[9 * cap] = __double_as_longlong(t.aux0); p[10 * cap] = __double_as_longlong(t.aux1); }
        else p[8 * cap] = (uint64_t)t.path | ((uint64_t)t.pixel << 32);
    }
};

// The primary ray of pixel slot `slot` (Camera::GetScreenRay, src/camera.cpp:26-41); false for a slot outside the frame.
QD_DEV bool wf_primary(const Params& P, uint32_t slot, Task& t) {
    const PixelSlot px = map_pixel(P, unit_of_slot(P, slot), slot % kUnitPx);
    if (!px.valid) return false;
    const DV TL = ld3(P.cam.top_left), TR = ld3(P.cam.top_right), BL = ld3(P.cam.bottom_left), pos = ld3(P.cam.position);
    const double kx = (px.tap == 1u || px.tap == 4u) ? 0.6 : (px.tap == 3u ? 0.3 : 0.0);      // anti-aliasing taps, src/main.cpp:108-114
    const double ky = (px.tap == 2u || px.tap == 4u) ? 0.6 : (px.tap == 3u ? 0.3 : 0.0);
    const DV through = TL + (TR - TL) * (((double)px.x + kx) / P.W) + (BL - TL) * (((double)px.y + ky) / P.H);
    t.o = pos; t.d = normalize(through - pos);
    t.w = mkc(1.f, 1.f, 1.f);
    t.flags_depth = kFlagPrimary;
    t.path = pixel_path(px.x, px.y, px.tap);
    t.aux0 = t.aux1 = 0;
    t.pixel = px.out_idx;
    return true;
}

// Hands out `want` consecutive indices of a generation.  The index range is cut into eight regions with a cursor each
// (a cache line apart); a wave drains the region of its XCD label first (waves are dealt round-robin over the 8 XCDs, so
// blockIdx % 8 names waves that share an L2), then takes from the others.  In generation 0 an index is a pixel slot, so
// every XCD works on one contiguous part of the image, hence on one part of the KD tree (the one-kernel form does the
// same with its eight work queues).  Returns the number granted, 0 = everything is handed out.
QD_DEV uint32_t wf_fetch(uint32_t* cursors, uint32_t n_in, uint32_t home, uint32_t want, uint32_t& first) {
    const uint32_t per = ((n_in + 7u) / 8u + 63u) / 64u * 64u;
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t k = (home + i) & 7u;
        const uint32_t lo = k * per;
        if (lo >= n_in) continue;
        const uint32_t size = min(per, n_in - lo);
        if (__hip_atomic_load(&cursors[k * kCtlStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= size) continue;
        const uint32_t v = atomicAdd(&cursors[k * kCtlStride], want);
        if (v < size) { first = lo + v; return min(want, size - v); }
    }
    return 0u;
}

#ifndef QD_WF_SHADE_WAVES
#define QD_WF_SHADE_WAVES 2
#endif

// One scene node with a KD-tree mesh, all rays of generation `gen`.
template <bool STATS>
__global__ __launch_bounds__(64, QD_WF_WAVES) void wf_walk_kernel(const DScene S, const Params P, const int gen, const int node_idx, const int slot) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    uint32_t* const ctl = P.wf_ctl + (size_t)gen * kWfGenWords;
    const size_t cap = P.wf_cap;
    const uint32_t n_in = gen == 0 ? P.n_units * kUnitPx
                                   : min(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), P.wf_limit);
    uint32_t* const cursors = &ctl[wf_walk_cursor((uint32_t)slot) * kCtlStride];
    const uint64_t* const Q = P.wf_ray[gen & 1];
    uint64_t* const R = P.wf_res + (size_t)slot * kWfResWords * cap;
    // the node and its mesh are the same for every ray of the launch: read once, before anything is stored
    const qd_node nd = S.nodes[node_idx];
    const qd_mesh ms = S.meshes[S.geoms[nd.geom].a];
    const gp_u32x4 kd = (gp_u32x4)S.kd_nodes;
    const gp_f64 leaf_verts = (gp_f64)S.leaf_verts;
    const gp_u32 leaf_tris = (gp_u32)S.leaf_tris;
    const gp_f64 gnormal = (gp_f64)S.tri_gnormal;
    const uint32_t mflags = ms.flags, tri_first = ms.tri_first, kd_root = (uint32_t)ms.kd_root;

    __shared__ uint2 lds_stack[kLdsStack][64];
    uint2* const stk_deep = P.wf_deep + ((size_t)blockIdx.x * 64 + lane);      // levels beyond LDS: HBM, lane-interleaved, rarely reached
    const size_t deep_stride = (size_t)gridDim.x * 64;

    uint32_t ridx = kWfNone;                 // ray being walked (queue index; pixel slot in generation 0)
    DV o = mk(0, 0, 0), d = mk(0, 0, 1), inv = mk(0, 0, 0);
    double tmin = 0, tmax = 0, top_t = 0, best = kInf, l2 = 0, l3 = 0;
    double any_limit = -1.;                  // shadow walk: a candidate nearer than this ends the walk
    uint32_t node = 0, ref = 0, left = 0, hit_ref = kWfNone, top_node = 0, dpos = 0;
    int sp = 0;
    uint32_t c_nodes = 0, c_tris = 0, c_queries = 0;
    uint32_t blk_next = 0, blk_end = 0;      // indices this wave has reserved (uniform)
    bool exhausted = false;

    auto stack_put = [&](int at, uint32_t tok, double t) {
        float f = (float)t;
        if ((double)f < t) f = __uint_as_float(__float_as_uint(f) + (f >= 0.f ? 1u : 0xffffffffu));   // next float up
        const uint2 e = make_uint2(tok, __float_as_uint(f));
        if (at < kLdsStack) lds_stack[at][lane] = e; else stk_deep[(size_t)(at - kLdsStack) * deep_stride] = e;
    };
    auto stack_get = [&](int at, uint32_t& tok, double& t) {
        uint2 e;
        if (at < kLdsStack) e = lds_stack[at][lane]; else e = stk_deep[(size_t)(at - kLdsStack) * deep_stride];
        tok = e.x; t = (double)__uint_as_float(e.y);
    };
    // one split plane of the walk (see mesh_query in dgeom.h: same rule, same slack)
    auto choose = [&](double split, uint32_t axis, uint32_t tok_left, uint32_t tok_right) -> uint32_t {
        if (STATS) c_nodes++;
        const double oa = pick(o, axis);
        const double t = (split - oa) * pick(inv, axis);
        const uint32_t near_s = (oa > split || (oa == split && ((dpos >> axis) & 1u))) ? 1u : 0u;
        const double slack = 1e-7 + 1e-9 * fabs(t);
        if (t > tmax + slack || t <= 0) return near_s;
        if (t < tmin - slack) return near_s ^ 1u;
        if (sp <= kKdStack) {
            if (sp > 0) stack_put(sp - 1, top_node, top_t);
            top_node = near_s ? tok_left : tok_right; top_t = tmax; ++sp;
        }
        tmax = (t == t) ? fmin(tmax, t + slack) : tmax;
        return near_s;
    };

    for (;;) {
        // ---- 1. idle lanes take the next rays: a coalesced read of the world-space ray, the node's transform (uniform), the
        //         slab interval of the mesh box.  A ray that misses the box is answered at once and the lane asks again.
        for (;;) {
            const unsigned long long idle = __ballot(ridx == kWfNone);
            if (idle == 0ull || exhausted) break;
            if (blk_next == blk_end) {
                uint32_t first = 0, got = 0;
                if (lane == 0) got = wf_fetch(cursors, n_in, blockIdx.x & 7u, kWfBlock, first);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got); first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
                if (got == 0u) { exhausted = true; break; }
                blk_next = first; blk_end = first + got;
            }
            const uint32_t take = min((uint32_t)__popcll(idle), blk_end - blk_next);
            const uint32_t rank = (uint32_t)__popcll(idle & lanes_below);
            if (ridx == kWfNone && rank < take) {
                const uint32_t i = blk_next + rank;
                bool ok = true, shadow = false;
                DV wo, wd;
                double target = 0.;
                if (gen == 0) {
                    Task t;
                    ok = wf_primary(P, i, t);
                    wo = t.o; wd = t.d;
                } else {
                    wo = mk(__longlong_as_double((long long)Q[0 * cap + i]), __longlong_as_double((long long)Q[1 * cap + i]), __longlong_as_double((long long)Q[2 * cap + i]));
                    wd = mk(__longlong_as_double((long long)Q[3 * cap + i]), __longlong_as_double((long long)Q[4 * cap + i]), __longlong_as_double((long long)Q[5 * cap + i]));
                    shadow = ((uint32_t)(Q[7 * cap + i] >> 32) & kFlagShadow) != 0;
                    if (shadow) target = __longlong_as_double((long long)Q[9 * cap + i]);
                }
                if (ok) {
                    const ObjRay r = to_object(nd, wo, wd);
                    o = r.o; d = r.d;
                    inv = mk(1. / d.x, 1. / d.y, 1. / d.z);
                    double t0 = 0., t1 = kInf;
                    bool miss = false;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {                       // slab interval of the mesh box, padded like BBox::IsInside
                        const double oa = comp(o, a), da = comp(d, a);
                        const double lo = ms.bbox_min[a] - 1e-6, hi = ms.bbox_max[a] + 1e-6;
                        if (da == 0.) { if (oa < lo || oa > hi) miss = true; continue; }
                        double ta = (lo - oa) * comp(inv, a), tb = (hi - oa) * comp(inv, a);
                        if (ta > tb) { const double s = ta; ta = tb; tb = s; }
                        t0 = fmax(t0, ta); t1 = fmin(t1, tb);
                    }
                    if (shadow) t1 = fmin(t1, dsqrt(target) * r.len * (1. + 1e-9) + 1e-7);      // nothing beyond the light matters
                    if (miss || t0 > t1) R[0 * cap + i] = (uint64_t)__double_as_longlong(kInf);
                    else {
                        if (STATS) c_queries++;
                        // a candidate at distance x ends a shadow walk iff Sqr(x / len) < target (src/shadinghelper.cpp:33);
                        // x < sqrt(target) * len * (1 - 1e-12) implies it, a candidate in the last 1e-12 just walks on
                        // (the shading kernel makes the exact test on the result either way)
                        any_limit = shadow ? dsqrt(target) * r.len * (1. - 1e-12) : -1.;
                        ridx = i;
                        best = kInf; l2 = l3 = 0; hit_ref = kWfNone;
                        node = kd_root; ref = 0; left = 0; sp = 0; top_node = 0; top_t = 0.;
                        tmin = t0; tmax = t1;
                        dpos = (d.x > 0 ? 1u : 0u) | (d.y > 0 ? 2u : 0u) | (d.z > 0 ? 4u : 0u);
                    }
                }
            }
            blk_next += take;
            if ((uint32_t)__popcll(__ballot(ridx != kWfNone)) >= (uint32_t)QD_WF_MIN_ACTIVE) break;
        }
        if (__ballot(ridx != kWfNone) == 0ull) break;                    // generation handed out and nothing left in this wave

        // ---- 2. KD walk: one trip to memory per iteration for every walking lane (record fetch or triangle fetch); left for a
        //         refill when too few lanes are still walking
        TriV tr; tr.A = tr.AB = tr.AC = mk(0., 0., 0.);
        for (;;) {
            const unsigned long long m_walk = __ballot(ridx != kWfNone);
            if (m_walk == 0ull) break;
            if ((uint32_t)__popcll(m_walk) < (uint32_t)QD_WF_MIN_ACTIVE && !exhausted) break;
            const bool walking = ridx != kWfNone;
            const bool in_leaf = walking && left != 0u;
            const bool at_node = walking && !in_leaf;
            bool leaf_done = false;
            const uint32_t rec = node & 0x3fffffffu, sub = node >> 30;
            u32x4 ra = {0u, 0u, 0u, 0u}, rb = {0u, 0u, 3u, 0u};
            if (in_leaf) tr = ld_tri(leaf_verts + (size_t)ref * 9);
            if (at_node) { ra = kd[2u * rec]; rb = kd[2u * rec + 1u]; }
            asm volatile("" : "+v"(ra), "+v"(rb), "+v"(tr.A.x), "+v"(tr.A.y), "+v"(tr.A.z), "+v"(tr.AB.x), "+v"(tr.AB.y), "+v"(tr.AB.z),
                              "+v"(tr.AC.x), "+v"(tr.AC.y), "+v"(tr.AC.z));
            bool walk_over = false;
            if (in_leaf) {
                const DV md = -d;
                if (tri_test<STATS>(gnormal, mflags, tr, [&]() { return leaf_tris[ref]; }, o, d, md, best, l2, l3, c_tris)) {
                    hit_ref = ref;
                    if (best < any_limit) walk_over = true;
                }
                ++ref; --left;
                leaf_done = left == 0u;
            }
            if (at_node) {
                const uint32_t meta = rb.z;
                if ((meta & 3u) == 3u) {
                    if (STATS) c_nodes++;
                    ref = ra.x; left = ra.y;
                    leaf_done = left == 0u;
                } else {
                    const uint32_t c = sub ? sub - 1u
                                           : choose(__hiloint2double((int)ra.y, (int)ra.x), meta & 3u, rec | (1u << 30), rec | (2u << 30));
                    const uint32_t ac = (meta >> (2u + 2u * c)) & 3u;
                    const uint32_t lo = c ? rb.x : ra.z, hi = c ? rb.y : ra.w;
                    if (ac == 3u) {
                        if (STATS) c_nodes++;
                        ref = lo; left = hi;
                        leaf_done = left == 0u;
                    } else {
                        const uint32_t cb = ((meta >> 6) << 2) + 2u * c;
                        node = cb + choose(__hiloint2double((int)hi, (int)lo), ac, cb, cb + 1u);
                    }
                }
            }
            if (leaf_done && !walk_over) {
                if (hit_ref != kWfNone && best <= tmax - (2e-7 + 2e-9 * fabs(tmax))) walk_over = true;
                else if (sp == 0) walk_over = true;
                else {
                    node = top_node; tmin = tmax; tmax = top_t;
                    --sp;
                    if (sp > 0) stack_get(sp - 1, top_node, top_t);
                }
            }
            if (walking && walk_over) {                                  // the answer for this ray and this node
                if (hit_ref != kWfNone) {
                    R[0 * cap + ridx] = (uint64_t)__double_as_longlong(best);
                    R[1 * cap + ridx] = (uint64_t)(leaf_tris[hit_ref] - tri_first);
                    R[2 * cap + ridx] = (uint64_t)__double_as_longlong(l2);
                    R[3 * cap + ridx] = (uint64_t)__double_as_longlong(l3);
                } else R[0 * cap + ridx] = (uint64_t)__double_as_longlong(kInf);
                ridx = kWfNone;
            }
        }
    }

    if (STATS && P.stats) {
        const uint32_t a = wave_sum(c_nodes), b = wave_sum(c_tris), c = wave_sum(c_queries);
        if (lane == 0) {
            if (a) atomicAdd(&P.stats[4], (unsigned long long)a);
            if (b) atomicAdd(&P.stats[5], (unsigned long long)b);
            if (c) atomicAdd(&P.stats[6], (unsigned long long)c);
        }
    }
}

// Raytrace's node loop for one ray of the shading kernel (same rules as trace_nodes): analytic geometry, heightfields and
// small meshes without a tree are intersected here, a KD-tree mesh is answered by the walk kernel's result record.
// CALLS: whether the copy may contain real calls (mesh_query for a mesh too small for a tree, the nested-Layered walk,
// csg_generic); a scene that needs none of them runs a call-free copy (a call site, however rare, costs the kernel's
// register allocation a few percent).
template <bool STATS, int GEOM, bool CALLS>
QD_DEV int wf_nodes(const DScene& S, const Params& P, uint32_t ridx, const DV& o, const DV& d, bool shadow, double target,
                                   float& transp, double& best_dw, GHit& hit, Counters& cnt) {
    int best_node = -1;
    best_dw = kInf; transp = 1.f;
    double w_dist = 0, w_l2 = 0, w_l3 = 0;
    int w_tri = -1;
    const uint32_t n_nodes = S.n_nodes;
    const size_t cap = P.wf_cap;
    uint32_t i = 0;
    bool final_pass = false;
    for (;;) {
        if (!final_pass && i == n_nodes) {
            if (shadow || best_node < 0) break;
            final_pass = true; i = (uint32_t)best_node;
        }
        const qd_node& n = S.nodes[i];
        const ObjRay r = to_object(n, o, d);
        const qd_geom g = S.geoms[n.geom];
        double dist = 0, c_l2 = 0, c_l3 = 0;
        int c_tri = -1;
        bool got;
        if (g.type == QD_GEOM_MESH) {
            if (final_pass) {
                hit.dist = w_dist; hit.ip = r.o + r.d * w_dist; hit.tri = w_tri; hit.mesh = g.a; hit.l2 = w_l2; hit.l3 = w_l3;
                break;
            }
            const int slot = S.node_kd_slot[i];
            if (slot >= 0) {
                const uint64_t* R = P.wf_res + (size_t)slot * kWfResWords * cap;
                dist = __longlong_as_double((long long)R[0 * cap + ridx]);
                got = dist < kInf;
                if (got) { c_tri = (int)(uint32_t)R[1 * cap + ridx]; c_l2 = __longlong_as_double((long long)R[2 * cap + ridx]); c_l3 = __longlong_as_double((long long)R[3 * cap + ridx]); }
            } else if constexpr (CALLS) {                                // a mesh without a tree (< 50 triangles, src/mesh.cpp:200): all its triangles
                const MHit mh = mesh_query<STATS>(mesh_data(S), g.a, r.o, r.d, shadow, target, r.len);
                got = mh.tri >= 0;
                if (got) { dist = mh.dist; c_tri = mh.tri; c_l2 = mh.l2; c_l3 = mh.l3; }
                if (STATS) { cnt.kd_nodes += mh.n_nodes; cnt.tris += mh.n_tris; cnt.mesh_queries += mh.n_nodes ? 1u : 0u; }
            } else got = false;                                          // (launch() never picks this copy for such a scene)
        } else if ((GEOM & kGeomHf) && g.type == QD_GEOM_HEIGHTFIELD) {
            if (final_pass) { hf_finish(S, g.a, r.o, r.d, w_dist, hit); break; }
            got = hf_intersect<STATS>(S, g.a, r.o, r.d, dist, cnt.kd_nodes, cnt.tris);
        } else {
            GHit h;
            if ((GEOM & kGeomCsgDeep) && g.type == QD_GEOM_CSG) {
                Counters again = cnt;
                got = csg_generic<STATS, kMaxCsgNesting - 1>(kernarg_scene(), n.geom, r.o, r.d, h, final_pass ? again : cnt);
            } else got = analytic_intersect<(GEOM & kGeomCsg) != 0, (GEOM & kGeomPrim) != 0>(S, g, r.o, r.d, h);
            if (final_pass) { hit = h; break; }
            dist = h.dist;
        }
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
        ++i;
    }
    return best_node;
}

// The rays of generation `gen`, full width and in queue order: the node loop, then Raytrace from
// `closestInfo.rayDir = ray.dir` on (src/main.cpp:40-55) or the shadow term (src/shadinghelper.cpp:11-18).
template <bool STATS, int GEOM, bool RICH, bool CALLS>
__global__ __launch_bounds__(64, QD_WF_SHADE_WAVES) void wf_shade_kernel(const DScene S, const Params P, const int gen) {
    const uint32_t lane = threadIdx.x;
    uint32_t* const ctl = P.wf_ctl + (size_t)gen * kWfGenWords;
    uint32_t* const ctl_next = ctl + kWfGenWords;
    const size_t cap = P.wf_cap;
    const uint32_t n_in = gen == 0 ? P.n_units * kUnitPx
                                   : min(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), P.wf_limit);
    const uint64_t* const Q = P.wf_ray[gen & 1];
    WfAppender q;
    q.base = P.wf_ray[(gen + 1) & 1]; q.cap = cap; q.limit = P.wf_limit; q.count = &ctl_next[0]; q.over = P.wf_over; q.max_depth = P.max_depth; q.n_killed = 0;
    q.lanes_below = (1ull << lane) - 1ull;
    Counters cnt; cnt.kd_nodes = cnt.tris = cnt.mesh_queries = 0; cnt.t_desc = cnt.t_leaf = cnt.t_a = cnt.t_b = cnt.t_c = cnt.t_d = 0;
    uint32_t n_rays = 0, n_shadow = 0, n_primary = 0;
    for (;;) {
        uint32_t first = 0, got = 0;
        if (lane == 0) got = wf_fetch(&ctl[kWfCurShade * kCtlStride], n_in, blockIdx.x & 7u, 64u, first);
        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got); first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (got == 0u) break;
        if (lane >= got) continue;
        const uint32_t ridx = first + lane;
        Task t;
        if (gen == 0) { if (!wf_primary(P, ridx, t)) continue; ++n_primary; }
        else {
            t.o = mk(__longlong_as_double((long long)Q[0 * cap + ridx]), __longlong_as_double((long long)Q[1 * cap + ridx]), __longlong_as_double((long long)Q[2 * cap + ridx]));
            t.d = mk(__longlong_as_double((long long)Q[3 * cap + ridx]), __longlong_as_double((long long)Q[4 * cap + ridx]), __longlong_as_double((long long)Q[5 * cap + ridx]));
            const uint64_t a = Q[6 * cap + ridx], w = Q[7 * cap + ridx], pp = Q[8 * cap + ridx];
            t.w = mkc(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)w));
            t.flags_depth = (uint32_t)(w >> 32);
            if (t.flags_depth & kFlagShadow) {
                t.pixel = (uint32_t)pp; t.path = 0;
                t.aux0 = __longlong_as_double((long long)Q[9 * cap + ridx]); t.aux1 = __longlong_as_double((long long)Q[10 * cap + ridx]);
            } else { t.path = (uint32_t)pp; t.pixel = (uint32_t)(pp >> 32); t.aux0 = t.aux1 = 0; }
        }
        const bool shadow = (t.flags_depth & kFlagShadow) != 0;
        const uint32_t depth = t.flags_depth & kDepthMask;
        if (shadow) ++n_shadow; else ++n_rays;
        GHit h; double dw; float transp;
        const int node = wf_nodes<STATS, GEOM, CALLS>(S, P, ridx, t.o, t.d, shadow, t.aux0, transp, dw, h, cnt);
        if (shadow) { add_px(P, t.pixel, t.w * (float)((double)transp * t.aux1)); continue; }
        if ((t.flags_depth & kFlagPrimary) && P.hit_node) {
            P.hit_node[t.pixel] = node;
            P.hit_tri[t.pixel] = node >= 0 ? h.tri : -1;
            P.hit_dist[t.pixel] = node >= 0 ? dw : 0.;
        }
        Col acc = mkc(0, 0, 0);
        if (node < 0) {
            if (S.env_present) acc = t.w * env_lookup(S, t.d);
            add_px(P, t.pixel, acc);
            continue;
        }
        const qd_node& n = S.nodes[node];
        Info info;
        assemble_info<RICH>(S, n, h, dw, t.d, info);
        if (RICH && n.bump_tex >= 0) tex_modify_normal(S, n.bump_tex, info);
        const qd_shader top = S.shaders[n.shader];
        if (top.type == QD_SH_LAYERED) {                                 // Layered::Shade, src/shading.cpp:260-275 (suffix products, see trace_kernel)
            Col suffix = mkc(1.f, 1.f, 1.f);
            for (int i = top.i1 - 1; i >= 0; --i) {
                const qd_layer l = S.layers[top.i0 + i];
                Col blend = ldc(l.blend);
                if (l.tex >= 0) blend = blend * tex_sample<RICH>(S, l.tex, info);
                eval_leaf_shader_body<CALLS, RICH>(S, P, kernarg_scene(), kernarg_params(), l.shader, t.w * (blend * suffix), (uint32_t)i + 1, t, depth, info, acc, q);
                suffix = suffix * (mkc(1.f, 1.f, 1.f) - blend);
            }
        } else {
            eval_leaf_shader_body<CALLS, RICH>(S, P, kernarg_scene(), kernarg_params(), n.shader, t.w, 0, t, depth, info, acc, q);
        }
        add_px(P, t.pixel, acc);
    }
    if (P.stats) {
        const uint32_t r = wave_sum(n_rays), s = wave_sum(n_shadow), p = wave_sum(n_primary), k = wave_sum(q.n_killed);
        uint32_t a = 0, b = 0, c = 0;
        if (STATS) { a = wave_sum(cnt.kd_nodes); b = wave_sum(cnt.tris); c = wave_sum(cnt.mesh_queries); }
        if (lane == 0) {
            if (r + s) atomicAdd(&P.stats[0], (unsigned long long)r + s);
            if (p) atomicAdd(&P.stats[1], (unsigned long long)p);
            if (s) atomicAdd(&P.stats[2], (unsigned long long)s);
            if (k) atomicAdd(&P.stats[3], (unsigned long long)k);
            if (STATS) {
                if (a) atomicAdd(&P.stats[4], (unsigned long long)a);
                if (b) atomicAdd(&P.stats[5], (unsigned long long)b);
                if (c) atomicAdd(&P.stats[6], (unsigned long long)c);
            }
        }
    }
}

// End of a band of the wavefront form (launch_wavefront), one thread per pixel slot of the band.  No append found its queue
// full: the band's counters are added to the caller's.  Otherwise rays are missing from the band: its pixels are zeroed and
// its counters dropped, and the launch that follows on the stream -- the one-kernel form on the same band, which does nothing
// unless *P.wf_over is set -- renders it again; a pinned host word tells the library to keep the scene off this form.
__global__ __launch_bounds__(64) void wf_band_end_kernel(const Params P) {
    const uint32_t over = __hip_atomic_load(P.wf_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (over == 0u) {
        if (blockIdx.x == 0 && threadIdx.x < 8u && P.user_stats && P.stats[threadIdx.x]) atomicAdd(&P.user_stats[threadIdx.x], P.stats[threadIdx.x]);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(P.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    if (slot >= P.n_units * kUnitPx) return;
    const PixelSlot px = map_pixel(P, unit_of_slot(P, slot), slot % kUnitPx);
    if (!px.valid) return;
    float* o = P.rgb + (P.mode == 0 ? (size_t)(px.out_idx / (uint32_t)P.rw) * P.pitch + (size_t)(px.out_idx % (uint32_t)P.rw) * 3 : (size_t)px.out_idx * 3);
    o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
}
