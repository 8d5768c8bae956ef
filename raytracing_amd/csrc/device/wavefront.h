// Wavefront pipeline: the frame as generations of rays in HBM queues instead of per-lane LIFO queues inside one kernel.
//
//   generation g:  wf_trace_kernel  reads ray queue g   (generation 0: rays are made from the pixel slot, no queue),
//                                   walks the node list / KD trees, and
//                                     - shadow ray  : adds weight * transparency * light term to its pixel,
//                                     - radiance ray: appends a hit record (ray index, node, distances, triangle, barycentrics);
//                  wf_shade_kernel  reads the hit records of g, rebuilds the full hit of the winner, runs the shaders and
//                                   appends the child rays (radiance depth + 1, shadow rays) to ray queue g + 1.
//   A frame is trace(0) shade(0) trace(1) ... shade(D) trace(D + 1): a fixed sequence of launches, every kernel reads the
//   size of its input from device counters, so nothing comes back to the host (the calls stay enqueue-only).
//
// Why: in the one-kernel form a round lasts as long as its slowest ray (53 % of the lanes busy in the KD loop), shading
// state lives across the traversal (spills, 0.55 KB of scratch written per ray) and each wave keeps its ray trees to
// itself.  Here the traversal kernel holds traversal state only and refills idle lanes from the queue INSIDE its loop
// (a lane whose ray ends -- a miss of the mesh box, a short walk -- takes the next ray while its neighbours are still
// walking), the shading kernel runs full width on hits, and queue traffic is coalesced word-major streams.
// Same arithmetic as the one-kernel form: to_object / analytic_intersect / the triangle test / assemble_info / the shaders
// are the same functions.
#pragma once

// control words per generation, one cache line each: [0] rays queued, [1..8] trace cursors (one per eighth of the queue),
// [9] hits queued, [10] shade cursor
constexpr uint32_t kWfGenWords = 12 * kCtlStride;
constexpr uint32_t kWfCurTrace = 1, kWfHits = 9, kWfCurShade = 10;
constexpr uint32_t kWfNone = 0xffffffffu;
constexpr int kWfHitWords = 6;
#ifndef QD_WF_WAVES
#define QD_WF_WAVES 3
#endif
#ifndef QD_WF_MIN_ACTIVE
#define QD_WF_MIN_ACTIVE 40      // the KD loop is left for a refill when fewer lanes than this are still walking
#endif

// Appends tasks to the next generation's queue: one atomic per call site and wave (the lanes that reach the call together).
struct WfAppender {
    uint64_t* base; size_t cap; uint32_t* count; int* err; uint32_t max_depth; uint32_t n_killed;
    unsigned long long lanes_below;
    QD_DEV void push(const Task& t) {
        if (!(t.flags_depth & kFlagShadow) && (t.flags_depth & kDepthMask) > max_depth) { ++n_killed; return; }   // src/main.cpp:20-21
        const unsigned long long m = __ballot(1);
        const uint32_t rank = (uint32_t)__popcll(m & lanes_below);
        uint32_t first = 0;
        if (rank == 0) first = atomicAdd(count, (uint32_t)__popcll(m));
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        const size_t i = (size_t)first + rank;
        if (i >= cap) { *err = 1; return; }
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
    const PixelSlot px = map_pixel(P, slot / kUnitPx, slot % kUnitPx);
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

// Register discipline of the traversal kernel: the KD loop runs with the state of the walk only (object-space ray, interval,
// far-child stack top, running hit: ~45 VGPRs besides the triangle under test).  Everything a ray needs only between
// geometries -- its world-space ray, the shadow target, the best node so far and its hit parameters -- lives in LDS, one
// column per lane (ColdState, 104 B per lane next to the 4 KB far-child stack), and is read where the node list needs it.
// With that state in registers the loop needed 217 VGPRs (2 waves per SIMD) or spilled 16 loads + 6 stores per iteration.
struct ColdState {
    double wo[3][64], wd[3][64];          // world-space ray
    double target[64], rlen[64];          // shadow ray: squared distance to the light; |Minv * dir| of the node being walked
    double best_dw[64], w_dist[64], w_l2[64], w_l3[64];
    int best_node[64], w_tri[64];
    float transp[64];
    uint32_t counts[3][64];               // rays, shadow rays, primary rays traced by this lane
};

template <bool STATS, int GEOM>
__global__ __launch_bounds__(64, QD_WF_WAVES) void wf_trace_kernel(const DScene S, const Params P, const int gen) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    uint32_t* const ctl = P.wf_ctl + (size_t)gen * kWfGenWords;
    const size_t cap = P.wf_cap;
    const uint32_t n_in = gen == 0 ? P.n_units * kUnitPx
                                   : min(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (uint32_t)cap);
    const uint64_t* const Q = P.wf_ray[gen & 1];
    uint64_t* const Hq = P.wf_hit;
    const MeshData D = mesh_data(S);
    const uint32_t n_nodes = S.n_nodes;

    __shared__ uint2 lds_stack[kLdsStack][64];
    __shared__ ColdState C;
    uint2* const stk_deep = P.wf_deep + ((size_t)blockIdx.x * 64 + lane);      // levels beyond LDS: HBM, lane-interleaved, rarely reached
    const size_t deep_stride = (size_t)gridDim.x * 64;

    // ---- per-lane state kept in registers
    uint32_t ridx = kWfNone;                 // ray in flight (queue index; pixel slot in generation 0)
    uint32_t ni = 0;                         // node of the node list the ray is at
    bool shadow = false, in_kd = false;
    // KD walk in flight (object space of node ni)
    DV o = mk(0, 0, 0), d = mk(0, 0, 1), inv = mk(0, 0, 0);
    double tmin = 0, tmax = 0, top_t = 0, best = kInf, l2 = 0, l3 = 0;
    double any_limit = 0;                    // shadow walk: a candidate nearer than this ends the walk (conservative form of Sqr(dist / len) < target)
    uint32_t node = 0, ref = 0, left = 0, hit_ref = kWfNone, top_node = 0, dpos_flags = 0;
    int sp = 0;
    uint32_t c_nodes = 0, c_tris = 0, c_queries = 0;
    bool exhausted = false;
    C.counts[0][lane] = 0; C.counts[1][lane] = 0; C.counts[2][lane] = 0;

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
        const uint32_t near_s = (oa > split || (oa == split && ((dpos_flags >> axis) & 1u))) ? 1u : 0u;
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
    // result of the geometry of node ni for this ray (Raytrace's node loop, src/main.cpp:26-38; GetShadowTransparency,
    // src/shadinghelper.cpp:29-37): returns true when the ray needs no further node
    auto node_result = [&](bool got, double dist, int c_tri, double c_l2, double c_l3, double len, float shadow_transparency) -> bool {
        if (!got) return false;
        const double dw = dist / len;
        if (shadow) {
            if (sqr(dw) < C.target[lane]) {
                const float tr_ = C.transp[lane] * shadow_transparency;
                C.transp[lane] = tr_;
                if (tr_ == 0.f) return true;
            }
        } else if (!(C.best_dw[lane] <= dw)) {
            C.best_dw[lane] = dw; C.best_node[lane] = (int)ni; C.w_dist[lane] = dist; C.w_tri[lane] = c_tri; C.w_l2[lane] = c_l2; C.w_l3[lane] = c_l3;
        }
        return false;
    };
    // the ray is done: a shadow ray adds its term to the pixel, a radiance ray leaves a hit record for the shading kernel
    auto finish_ray = [&]() {
        if (shadow) {
            const uint64_t a = Q[6 * cap + ridx], w = Q[7 * cap + ridx];
            const Col wt = mkc(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)w));
            const double aux1 = __longlong_as_double((long long)Q[10 * cap + ridx]);
            add_px(P, (uint32_t)Q[8 * cap + ridx], wt * (float)((double)C.transp[lane] * aux1));
        } else {
            const unsigned long long m = __ballot(1);
            const uint32_t rank = (uint32_t)__popcll(m & lanes_below);
            uint32_t first = 0;
            if (rank == 0) first = atomicAdd(&ctl[kWfHits * kCtlStride], (uint32_t)__popcll(m));
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            const size_t h = (size_t)first + rank;
            if (h >= cap) *P.error_flag = 1;
            else {
                const int bn = C.best_node[lane];
                Hq[0 * cap + h] = (uint64_t)ridx | ((uint64_t)(uint32_t)bn << 32);
                if (bn >= 0) {
                    Hq[1 * cap + h] = (uint64_t)__double_as_longlong(C.best_dw[lane]);
                    Hq[2 * cap + h] = (uint64_t)__double_as_longlong(C.w_dist[lane]);
                    const int wt = C.w_tri[lane];
                    if (wt >= 0) {
                        Hq[3 * cap + h] = (uint64_t)(uint32_t)wt;
                        Hq[4 * cap + h] = (uint64_t)__double_as_longlong(C.w_l2[lane]);
                        Hq[5 * cap + h] = (uint64_t)__double_as_longlong(C.w_l3[lane]);
                    } else Hq[3 * cap + h] = 0xffffffffull;
                }
            }
        }
        ridx = kWfNone;
    };

    for (;;) {
        // ---- 1. idle lanes take the next rays of the generation; lanes between geometries walk on through the node list.
        //         Repeated while that leaves too few lanes for the KD loop (rays that miss a mesh's box end right here).
        for (;;) {
            const unsigned long long idle = __ballot(ridx == kWfNone);
            if (idle != 0ull && !exhausted) {
                uint32_t first = 0, got = 0;
                if (lane == 0) got = wf_fetch(&ctl[kWfCurTrace * kCtlStride], n_in, blockIdx.x & 7u, (uint32_t)__popcll(idle), first);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got); first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
                if (got == 0u) exhausted = true;
                else {
                    const uint32_t rank = (uint32_t)__popcll(idle & lanes_below);
                    if (ridx == kWfNone && rank < got) {
                        const uint32_t i = first + rank;
                        bool ok = true;
                        uint32_t fd = kFlagPrimary;
                        DV wo, wd;
                        if (gen == 0) {
                            Task t;
                            ok = wf_primary(P, i, t);
                            wo = t.o; wd = t.d;
                            if (ok) C.counts[2][lane] += 1u;
                        } else {
                            wo = mk(__longlong_as_double((long long)Q[0 * cap + i]), __longlong_as_double((long long)Q[1 * cap + i]), __longlong_as_double((long long)Q[2 * cap + i]));
                            wd = mk(__longlong_as_double((long long)Q[3 * cap + i]), __longlong_as_double((long long)Q[4 * cap + i]), __longlong_as_double((long long)Q[5 * cap + i]));
                            fd = (uint32_t)(Q[7 * cap + i] >> 32);
                        }
                        if (ok) {
                            ridx = i;
                            shadow = (fd & kFlagShadow) != 0;
                            C.wo[0][lane] = wo.x; C.wo[1][lane] = wo.y; C.wo[2][lane] = wo.z;
                            C.wd[0][lane] = wd.x; C.wd[1][lane] = wd.y; C.wd[2][lane] = wd.z;
                            C.target[lane] = shadow ? __longlong_as_double((long long)Q[9 * cap + i]) : 0.;
                            C.best_node[lane] = -1; C.w_tri[lane] = -1; C.best_dw[lane] = kInf; C.transp[lane] = 1.f;
                            ni = 0; in_kd = false;
                            C.counts[shadow ? 1 : 0][lane] += 1u;
                        }
                    }
                }
            }
            // node list: everything that is not a KD walk is done here, lane by lane, until the lane enters a KD tree or its ray ends
            while (ridx != kWfNone && !in_kd) {
                if (ni == n_nodes) { finish_ray(); break; }
                const qd_node& n = S.nodes[ni];
                const ObjRay r = to_object(n, mk(C.wo[0][lane], C.wo[1][lane], C.wo[2][lane]), mk(C.wd[0][lane], C.wd[1][lane], C.wd[2][lane]));
                const qd_geom g = S.geoms[n.geom];
                bool done = false;
                if (g.type == QD_GEOM_MESH) {
                    const QD_GLOBAL qd_mesh& m = ((const QD_GLOBAL qd_mesh*)D.meshes)[g.a];
                    o = r.o; d = r.d;
                    inv = mk(1. / d.x, 1. / d.y, 1. / d.z);
                    double t0 = 0., t1 = kInf;
                    bool miss = false;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {                       // slab interval of the mesh box, padded like BBox::IsInside
                        const double oa = comp(o, a), da = comp(d, a);
                        const double lo = m.bbox_min[a] - 1e-6, hi = m.bbox_max[a] + 1e-6;
                        if (da == 0.) { if (oa < lo || oa > hi) miss = true; continue; }
                        double ta = (lo - oa) * comp(inv, a), tb = (hi - oa) * comp(inv, a);
                        if (ta > tb) { const double s = ta; ta = tb; tb = s; }
                        t0 = fmax(t0, ta); t1 = fmin(t1, tb);
                    }
                    const double tgt = C.target[lane];
                    if (shadow) t1 = fmin(t1, dsqrt(tgt) * r.len * (1. + 1e-9) + 1e-7);
                    if (miss || t0 > t1) { ++ni; continue; }
                    if (STATS) c_queries++;
                    const uint32_t mflags = m.flags, tri_first = m.tri_first;
                    best = kInf; l2 = l3 = 0; hit_ref = kWfNone;
                    if (m.kd_root < 0) {                                 // small mesh without a tree: all its triangles (src/mesh.cpp:45-52)
                        const uint32_t n_tri = m.tri_count;
                        const gp_f64 tv = (gp_f64)D.tri_verts;
                        const DV md = -d;
                        int tri = -1;
                        for (uint32_t i = 0; i < n_tri; ++i) {
                            const TriV t9 = ld_tri(tv + (size_t)(tri_first + i) * 9);
                            if (tri_test<STATS>((gp_f64)D.tri_gnormal, mflags, t9, [&]() { return tri_first + i; }, o, d, md, best, l2, l3, c_tris)) {
                                tri = (int)i;
                                if (shadow && sqr(best / r.len) < tgt) break;
                            }
                        }
                        done = node_result(tri >= 0, best, tri, l2, l3, r.len, n.shadow_transparency);
                    } else {
                        C.rlen[lane] = r.len;
                        // a candidate at distance x ends a shadow walk iff Sqr(x / len) < target; x < sqrt(target) * len * (1 - 1e-12)
                        // implies it (the exact test is made on the result), a candidate in the last 1e-12 just walks on
                        any_limit = shadow ? dsqrt(tgt) * r.len * (1. - 1e-12) : -1.;
                        node = (uint32_t)m.kd_root; ref = 0; left = 0; sp = 0; top_node = 0; top_t = 0.;
                        tmin = t0; tmax = t1;
                        dpos_flags = (d.x > 0 ? 1u : 0u) | (d.y > 0 ? 2u : 0u) | (d.z > 0 ? 4u : 0u) | (mflags << 3);
                        in_kd = true;
                        break;
                    }
                } else if ((GEOM & kGeomHf) && g.type == QD_GEOM_HEIGHTFIELD) {
                    double dist = 0;
                    const bool got = hf_intersect<STATS>(S, g.a, r.o, r.d, dist, c_nodes, c_tris);
                    done = node_result(got, dist, -1, 0., 0., r.len, n.shadow_transparency);
                } else {
                    GHit h;
                    bool got;
                    if ((GEOM & kGeomCsgDeep) && g.type == QD_GEOM_CSG) {
                        Counters cc; cc.kd_nodes = cc.tris = cc.mesh_queries = 0;
                        got = csg_generic<STATS, kMaxCsgNesting - 1>(S, n.geom, r.o, r.d, h, cc);
                        if (STATS) { c_nodes += cc.kd_nodes; c_tris += cc.tris; c_queries += cc.mesh_queries; }
                    } else got = analytic_intersect<(GEOM & kGeomCsg) != 0, (GEOM & kGeomPrim) != 0>(S, g, r.o, r.d, h);
                    done = node_result(got, h.dist, -1, 0., 0., r.len, n.shadow_transparency);
                }
                if (done) { finish_ray(); break; }
                ++ni;
            }
            const unsigned long long walking = __ballot(in_kd);
            const unsigned long long still_idle = __ballot(ridx == kWfNone);
            if (still_idle == 0ull || exhausted || (uint32_t)__popcll(walking) >= (uint32_t)QD_WF_MIN_ACTIVE) break;
        }
        if (__ballot(ridx != kWfNone) == 0ull) break;                    // generation exhausted and nothing in flight in this wave

        // ---- 2. KD walk: one trip to memory per iteration for every walking lane (record fetch or triangle fetch), left when
        //         too few lanes are still walking and a refill / the node list could give the others something to do
        {
            const gp_u32x4 kd = (gp_u32x4)D.kd_nodes;
            const gp_f64 leaf_verts = (gp_f64)D.leaf_verts;
            const gp_u32 leaf_tris = (gp_u32)D.leaf_tris;
            const gp_f64 gnormal = (gp_f64)D.tri_gnormal;
            TriV tr; tr.A = tr.AB = tr.AC = mk(0., 0., 0.);
            for (;;) {
                const unsigned long long m_walk = __ballot(in_kd);
                if (m_walk == 0ull) break;
                if ((uint32_t)__popcll(m_walk) < (uint32_t)QD_WF_MIN_ACTIVE) {
                    const unsigned long long waiting = __ballot(ridx != kWfNone && !in_kd);      // between geometries
                    if (waiting != 0ull || !exhausted) break;
                }
                const bool in_leaf = in_kd && left != 0u;
                const bool at_node = in_kd && !in_leaf;
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
                    if (tri_test<STATS>(gnormal, dpos_flags >> 3, tr, [&]() { return leaf_tris[ref]; }, o, d, md, best, l2, l3, c_tris)) {
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
                if (in_kd && walk_over) in_kd = false;                   // this mesh is done for this lane; its result is taken below
            }
            // lanes whose walk ended: the result of this mesh, then on through the node list (or the ray is done)
            if (ridx != kWfNone && !in_kd) {
                const qd_node& n = S.nodes[ni];
                const uint32_t tri_first = ((const QD_GLOBAL qd_mesh*)D.meshes)[S.geoms[n.geom].a].tri_first;
                const int tri = hit_ref != kWfNone ? (int)(leaf_tris[hit_ref] - tri_first) : -1;
                const bool done = node_result(tri >= 0, best, tri, l2, l3, C.rlen[lane], n.shadow_transparency);
                if (done) finish_ray(); else ++ni;
            }
        }
    }

    if (P.stats) {
        const uint32_t r = wave_sum(C.counts[0][lane]), s = wave_sum(C.counts[1][lane]), p = wave_sum(C.counts[2][lane]);
        uint32_t a = 0, b = 0, c = 0;
        if (STATS) { a = wave_sum(c_nodes); b = wave_sum(c_tris); c = wave_sum(c_queries); }
        if (lane == 0) {
            if (r + s) atomicAdd(&P.stats[0], (unsigned long long)r + s);
            if (p) atomicAdd(&P.stats[1], (unsigned long long)p);
            if (s) atomicAdd(&P.stats[2], (unsigned long long)s);
            if (STATS) {
                if (a) atomicAdd(&P.stats[4], (unsigned long long)a);
                if (b) atomicAdd(&P.stats[5], (unsigned long long)b);
                if (c) atomicAdd(&P.stats[6], (unsigned long long)c);
            }
        }
    }
}

// Shading of the hits of generation `gen`: Raytrace from `closestInfo.rayDir = ray.dir` on (src/main.cpp:40-55).
template <int GEOM, bool RICH>
__global__ __launch_bounds__(64, QD_WF_WAVES) void wf_shade_kernel(const DScene S, const Params P, const int gen) {
    const uint32_t lane = threadIdx.x;
    uint32_t* const ctl = P.wf_ctl + (size_t)gen * kWfGenWords;
    uint32_t* const ctl_next = ctl + kWfGenWords;
    const size_t cap = P.wf_cap;
    const uint32_t n_hits = min(__hip_atomic_load(&ctl[kWfHits * kCtlStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (uint32_t)cap);
    const uint64_t* const Q = P.wf_ray[gen & 1];
    const uint64_t* const Hq = P.wf_hit;
    WfAppender q;
    q.base = P.wf_ray[(gen + 1) & 1]; q.cap = cap; q.count = &ctl_next[0]; q.err = P.error_flag; q.max_depth = P.max_depth; q.n_killed = 0;
    q.lanes_below = (1ull << lane) - 1ull;
    for (;;) {
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(&ctl[kWfCurShade * kCtlStride], 64u);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (first >= n_hits) break;
        const uint32_t hi = first + lane;
        if (hi >= n_hits) continue;
        const uint64_t h0 = Hq[0 * cap + hi];
        const uint32_t ridx = (uint32_t)h0;
        const int node = (int)(uint32_t)(h0 >> 32);
        Task t;
        if (gen == 0) { if (!wf_primary(P, ridx, t)) continue; }
        else {
            t.o = mk(__longlong_as_double((long long)Q[0 * cap + ridx]), __longlong_as_double((long long)Q[1 * cap + ridx]), __longlong_as_double((long long)Q[2 * cap + ridx]));
            t.d = mk(__longlong_as_double((long long)Q[3 * cap + ridx]), __longlong_as_double((long long)Q[4 * cap + ridx]), __longlong_as_double((long long)Q[5 * cap + ridx]));
            const uint64_t a = Q[6 * cap + ridx], w = Q[7 * cap + ridx], pp = Q[8 * cap + ridx];
            t.w = mkc(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)w));
            t.flags_depth = (uint32_t)(w >> 32);
            t.path = (uint32_t)pp; t.pixel = (uint32_t)(pp >> 32);
            t.aux0 = t.aux1 = 0;
        }
        const uint32_t depth = t.flags_depth & kDepthMask;
        const double dw = node >= 0 ? __longlong_as_double((long long)Hq[1 * cap + hi]) : 0.;
        const int tri = node >= 0 ? (int)(uint32_t)Hq[3 * cap + hi] : -1;
        Col acc = mkc(0, 0, 0);
        if (node < 0) {
            if ((t.flags_depth & kFlagPrimary) && P.hit_node) { P.hit_node[t.pixel] = -1; P.hit_tri[t.pixel] = -1; P.hit_dist[t.pixel] = 0.; }
            if (S.env_present) acc = t.w * env_lookup(S, t.d);
            add_px(P, t.pixel, acc);
            continue;
        }
        // the full hit record of the winner (analytic geometry is intersected again, which is deterministic; mesh and
        // heightfield records are functions of the stored distance)
        const qd_node& n = S.nodes[node];
        const ObjRay r = to_object(n, t.o, t.d);
        const qd_geom g = S.geoms[n.geom];
        const double w_dist = __longlong_as_double((long long)Hq[2 * cap + hi]);
        GHit h;
        if (g.type == QD_GEOM_MESH) {
            h.dist = w_dist; h.ip = r.o + r.d * w_dist; h.tri = tri; h.mesh = g.a;
            h.l2 = __longlong_as_double((long long)Hq[4 * cap + hi]); h.l3 = __longlong_as_double((long long)Hq[5 * cap + hi]);
        } else if ((GEOM & kGeomHf) && g.type == QD_GEOM_HEIGHTFIELD) {
            hf_finish(S, g.a, r.o, r.d, w_dist, h);
        } else if ((GEOM & kGeomCsgDeep) && g.type == QD_GEOM_CSG) {
            Counters cc; cc.kd_nodes = cc.tris = cc.mesh_queries = 0;
            (void)csg_generic<false, kMaxCsgNesting - 1>(S, n.geom, r.o, r.d, h, cc);
        } else {
            (void)analytic_intersect<(GEOM & kGeomCsg) != 0, (GEOM & kGeomPrim) != 0>(S, g, r.o, r.d, h);
        }
        if ((t.flags_depth & kFlagPrimary) && P.hit_node) { P.hit_node[t.pixel] = node; P.hit_tri[t.pixel] = h.tri; P.hit_dist[t.pixel] = dw; }
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
                eval_leaf_shader<true, RICH>(S, P, l.shader, t.w * (blend * suffix), (uint32_t)i + 1, t, depth, info, acc, q);
                suffix = suffix * (mkc(1.f, 1.f, 1.f) - blend);
            }
        } else {
            eval_leaf_shader<true, RICH>(S, P, n.shader, t.w, 0, t, depth, info, acc, q);
        }
        add_px(P, t.pixel, acc);
    }
    if (P.stats) {
        const uint32_t k = wave_sum(q.n_killed);
        if (lane == 0 && k) atomicAdd(&P.stats[3], (unsigned long long)k);
    }
}
