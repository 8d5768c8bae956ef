// The device KD tree, built ON the device (qd_set_kd_build(1), the default).
//
// Replaces Mesh::BuildKD (src/mesh.cpp:198-432) for the device copy of the tree: same rules as the host builder of
// kdrefine.h (surface-area heuristic over triangle bounds clipped to the node box; 64 bins per axis for nodes above 192
// triangles, exact sweep over every bound below; classification of a triangle against a plane as PartitionTriangle,
// src/mesh.cpp:412-432, on its clipped bounds; leaves of at most max_leaf triangles where a split pays), but breadth
// first: a LEVEL of the tree is split per round of kernels, every node of the level at once.
//
//   per level   kb_hist          references of nodes > 192 triangles: 2 x 64 bins per axis (LDS-private per block while a
//                                block's references belong to one node, then flushed with one atomic per non-empty bin)
//               kb_pick_binned   one wave per such node: prefix sums over the bins, cost of the 63 boundaries per axis
//               kb_pick_exact    nodes of 33 .. 192 triangles: one wave each; 3 .. 32 triangles: eight lanes each.  The
//                                clipped bounds go to LDS, every bound is a candidate plane, its counts are exact
//               kb_classify      per reference: left / right flags against the node's plane
//               (scan)           exclusive sums of the flags (hipcub): where every reference lands in the next level,
//                                stable -- children keep their parent's triangle order, nothing depends on atomics
//               kb_emit          per node: the binary node, its two children (box, range of references, size class)
//               kb_scatter       per reference: copied to the left / right child's range, or to the leaf reference array
//   afterwards  kb_pack_*        the binary tree -> 32-byte two-level records (kdrefine.h: Record), level by level, the
//                                four-record groups allocated by a scan
//
// Nothing comes back to the host but a few counters per level (sizes of the next level, to size its launches and buffers).
// The arithmetic of a plane decision is the host builder's, operation for operation (-ffp-contract=off on both sides), so
// for a mesh below 524 288 triangles -- above that the host builder bins a sample -- both build the same tree
// (tests/test_gpu_parity.py::test_device_built_tree_equals_the_host_built_tree).
#pragma once
#include <hipcub/hipcub.hpp>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "kdrefine.h"
#include "qdtrace.h"

namespace qd_kdbuild {

struct Opt { int max_leaf; double cost_step; int max_depth; uint32_t binned_above; };

constexpr uint32_t kClsLeaf = 0, kClsSmall = 1, kClsMid = 2, kClsLarge = 3;
constexpr uint32_t kSmallMax = 32;              // triangles of a node that eight lanes split
constexpr uint32_t kBins = 64;
constexpr uint32_t kHistWords = 3 * 2 * kBins;  // per node: axis x {starts, ends} x bin
constexpr uint32_t kHistChunk = 2048;           // references per block of kb_hist

// A node of the level being split.
struct LNode {
    double lo[3], hi[3];
    double split;
    uint32_t off, cnt;         // its references: [off, off + cnt) of the level's reference arrays
    uint32_t bnode;            // its index in the binary node array
    int32_t axis;              // -1 = becomes a leaf, 0..2 = axis of its plane (pick kernels)
    uint32_t cls;              // kCls*
    uint32_t aux;              // kClsLarge: histogram slot
    uint32_t nl;               // references of the left child (kb_emit)
    uint32_t new_off;          // split: first reference of the left child in the next level; leaf: first leaf reference
    uint32_t s_l, s_r;         // scan values at the node's first reference
    uint32_t child_slot;       // index of the left child in the next level's node list
    uint32_t pad_;
};

// control words (device): [1..3] nodes per size class of the level being emitted, [4] leaves, [5] largest leaf
constexpr int kCtlWords = 8;

__device__ __forceinline__ double kb_half_area(const double* lo, const double* hi) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// bounds of triangle t on axis a, clipped to [blo, bhi] (tri_bounds in kdrefine.h)
__device__ __forceinline__ void kb_clipped(const double* __restrict__ tlo, const double* __restrict__ thi, size_t n_tris, uint32_t t, int a,
                                           double blo, double bhi, double& lo, double& hi) {
    const double l = tlo[(size_t)a * n_tris + t], h = thi[(size_t)a * n_tris + t];
    lo = l < blo ? blo : l;                 // std::max(l, blo)
    hi = bhi < h ? bhi : h;                 // std::min(h, bhi)
    if (hi < lo) hi = lo;
}

// per triangle: min / max of its three vertices per axis, axis-major (tri_verts holds A, B - A, C - A)
__global__ void kb_tri_aabb(const double* __restrict__ tri_verts, uint32_t n, double* __restrict__ tlo, double* __restrict__ thi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double* v = tri_verts + (size_t)t * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double p0 = v[a], p1 = v[a] + v[3 + a], p2 = v[a] + v[6 + a];
        const double m12 = p2 < p1 ? p2 : p1, x12 = p1 < p2 ? p2 : p1;      // std::min(p1, p2), std::max(p1, p2)
        tlo[(size_t)a * n + t] = m12 < p0 ? m12 : p0;                        // std::min(p0, .)
        thi[(size_t)a * n + t] = p0 < x12 ? x12 : p0;                        // std::max(p0, .)
    }
}

__global__ void kb_init_refs(uint32_t* __restrict__ ref_tri, uint32_t* __restrict__ ref_node, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { ref_tri[i] = i; ref_node[i] = 0u; }
}

__device__ __forceinline__ int kb_bin(double x, double L, double inv) {
    int b = (int)((x - L) * inv);
    b = b < 0 ? 0 : b; b = b > (int)kBins - 1 ? (int)kBins - 1 : b;
    return b;
}

// Histograms of the large nodes (binned_plane in kdrefine.h: starts[bin of lo]++, ends[bin of hi]++ per axis).
__global__ __launch_bounds__(256) void kb_hist(const LNode* __restrict__ nodes, const uint32_t* __restrict__ ref_tri, const uint32_t* __restrict__ ref_node,
                                               uint32_t R, const double* __restrict__ tlo, const double* __restrict__ thi, size_t n_tris,
                                               uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[kHistWords];
    const uint32_t first = blockIdx.x * kHistChunk;
    const uint32_t last = min(first + kHistChunk, R) - 1u;
    const uint32_t node_first = ref_node[first], node_last = ref_node[last];
    const bool uniform = node_first == node_last;            // references are sorted by node: the whole chunk is one node's
    if (uniform && nodes[node_first].cls != kClsLarge) return;
    if (uniform) {
        for (uint32_t k = threadIdx.x; k < kHistWords; k += 256) h[k] = 0u;
        __syncthreads();
    }
    for (uint32_t i = first + threadIdx.x; i <= last; i += 256) {
        const uint32_t nd = ref_node[i];
        const LNode& N = nodes[nd];
        if (N.cls != kClsLarge) continue;
        const uint32_t t = ref_tri[i];
        uint32_t* const dst = uniform ? h : hist + (size_t)N.aux * kHistWords;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double L = N.lo[a], W = N.hi[a] - N.lo[a];
            if (!(W > 0.)) continue;
            double lo, hi;
            kb_clipped(tlo, thi, n_tris, t, a, N.lo[a], N.hi[a], lo, hi);
            const double inv = (double)kBins / W;
            atomicAdd(&dst[a * 2 * kBins + kb_bin(lo, L, inv)], 1u);
            atomicAdd(&dst[a * 2 * kBins + kBins + kb_bin(hi, L, inv)], 1u);
        }
    }
    if (uniform) {
        __syncthreads();
        uint32_t* const dst = hist + (size_t)nodes[node_first].aux * kHistWords;
        for (uint32_t k = threadIdx.x; k < kHistWords; k += 256) if (h[k]) atomicAdd(&dst[k], h[k]);
    }
}

// One wave per large node: the plane among the 63 bin boundaries per axis (binned_plane in kdrefine.h, stride 1).
// Sequential rule there: axes 0..2, boundaries 1..63, a candidate replaces the best so far iff its cost is strictly
// smaller, starting from the cost of a leaf (n): i.e. the smallest cost below n, ties to the smallest (axis, boundary).
__global__ __launch_bounds__(64) void kb_pick_binned(LNode* __restrict__ nodes, const uint32_t* __restrict__ list, const uint32_t* __restrict__ hist,
                                                     double cost_step) {
    LNode& N = nodes[list[blockIdx.x]];
    const uint32_t lane = threadIdx.x;
    const double n = (double)N.cnt;
    const double area = kb_half_area(N.lo, N.hi);
    double best_cost = n, best_split = 0.;
    uint32_t best_key = 0xffffffffu;
    const uint32_t* const H = hist + (size_t)N.aux * kHistWords;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double L = N.lo[a], W = N.hi[a] - N.lo[a];
        if (!(W > 0.)) continue;
        const uint32_t st = H[a * 2 * kBins + lane], en = H[a * 2 * kBins + kBins + lane];
        uint32_t is = st, ie = en;                                   // inclusive prefix sums over the bins
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t vs = __shfl_up(is, o, 64), ve = __shfl_up(ie, o, 64);
            if ((int)lane >= o) { is += vs; ie += ve; }
        }
        const uint32_t total = __shfl(ie, 63, 64);
        if (lane >= 1u) {
            const uint32_t below = is - st, ended = ie - en;         // starts / ends in the bins before boundary `lane`
            const int u = (a + 1) % 3, v = (a + 2) % 3;
            const double du = N.hi[u] - N.lo[u], dv = N.hi[v] - N.lo[v];
            const double s = L + W * ((double)lane / (double)kBins);
            const double nl = (double)below * 1.0, nr = (double)(total - ended) * 1.0;
            const double dl = s - N.lo[a], dr = N.hi[a] - s;
            const double al = du * dv + (du + dv) * dl, ar = du * dv + (du + dv) * dr;
            double cost = cost_step + (al * nl + ar * nr) / area;
            if (below == 0u || total == ended) cost *= 0.8;          // cutting off empty space
            if (cost < best_cost) { best_cost = cost; best_key = (uint32_t)a * kBins + lane; best_split = s; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double oc = __shfl_xor(best_cost, o, 64), os = __shfl_xor(best_split, o, 64);
        const uint32_t ok = __shfl_xor(best_key, o, 64);
        if (oc < best_cost || (oc == best_cost && ok < best_key)) { best_cost = oc; best_key = ok; best_split = os; }
    }
    if (lane == 0u) {
        if (best_key == 0xffffffffu) N.axis = -1;
        else { N.axis = (int32_t)(best_key / kBins); N.split = best_split; }
    }
}

// LDS writes of some lanes of a wave read by other lanes of the same wave (a group never spans two waves)
__device__ __forceinline__ void kb_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Nodes of at most NMAX triangles, G lanes each: the exact sweep of Builder::split (kdrefine.h).  Every bound of every
// triangle is a candidate; for a candidate s strictly inside the box, nl = #(lo < s) + #(lo == hi == s) and
// nr = n - #(hi <= s) are what the classification gives; the sequential rule (axes 0..2, candidates ascending, strictly
// smaller cost replaces) picks the smallest cost below n, ties to the smallest axis, then the smallest s.
template <int G, int NMAX>
__global__ __launch_bounds__(64) void kb_pick_exact(LNode* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t n_list,
                                                    const uint32_t* __restrict__ ref_tri, const double* __restrict__ tlo, const double* __restrict__ thi,
                                                    size_t n_tris, double cost_step) {
    constexpr int kGroups = 64 / G;
    constexpr int kStride = 6 * NMAX + 1;                      // doubles per group; odd, so that groups start in different banks
    __shared__ double sb[kGroups * kStride];
    const uint32_t g = threadIdx.x / G, gl = threadIdx.x % G;
    const uint32_t li = blockIdx.x * kGroups + g;
    const bool active = li < n_list;
    const uint32_t ni = active ? list[li] : 0u;
    LNode& N = nodes[ni];
    const uint32_t n = active ? N.cnt : 0u;
    double* const blo = sb + g * kStride;                      // [axis][i]
    double* const bhi = blo + 3 * NMAX;
    for (uint32_t i = gl; i < n; i += G) {
        const uint32_t t = ref_tri[N.off + i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double lo, hi;
            kb_clipped(tlo, thi, n_tris, t, a, N.lo[a], N.hi[a], lo, hi);
            blo[a * NMAX + i] = lo; bhi[a * NMAX + i] = hi;
        }
    }
    kb_wave_sync();
    const double area = kb_half_area(N.lo, N.hi);
    double best_cost = (double)n, best_split = 0.;
    int best_axis = 3;                                         // 3 = none
    uint32_t best_nl = 0, best_nr = 0;
    for (int a = 0; a < 3 && active; ++a) {
        const double bl = N.lo[a], bh = N.hi[a];
        if (!(bh > bl)) continue;
        const int u = (a + 1) % 3, v = (a + 2) % 3;
        const double du = N.hi[u] - N.lo[u], dv = N.hi[v] - N.lo[v];
        const double* const los = blo + a * NMAX;
        const double* const his = bhi + a * NMAX;
        for (uint32_t c = gl; c < 2u * n; c += G) {
            const double s = c < n ? los[c] : his[c - n];
            if (!(s > bl && s < bh)) continue;
            uint32_t pl = 0, ph = 0, fe = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const double lo = los[i], hi = his[i];
                pl += lo < s ? 1u : 0u;
                ph += hi <= s ? 1u : 0u;
                fe += (lo == s && hi == s) ? 1u : 0u;
            }
            const uint32_t nl = pl + fe, nr = n - ph;
            const double dl = s - bl, dr = bh - s;
            const double al = du * dv + (du + dv) * dl, ar = du * dv + (du + dv) * dr;
            double cost = cost_step + (al * nl + ar * nr) / area;
            if (nl == 0u || nr == 0u) cost *= 0.8;             // cutting off empty space
            if (cost < best_cost || (cost == best_cost && a == best_axis && s < best_split)) {
                best_cost = cost; best_axis = a; best_split = s; best_nl = nl; best_nr = nr;
            }
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const double oc = __shfl_xor(best_cost, o, 64), os = __shfl_xor(best_split, o, 64);
        const int oa = __shfl_xor(best_axis, o, 64);
        const uint32_t onl = __shfl_xor(best_nl, o, 64), onr = __shfl_xor(best_nr, o, 64);
        const bool take = oc < best_cost || (oc == best_cost && (oa < best_axis || (oa == best_axis && os < best_split)));
        if (take) { best_cost = oc; best_axis = oa; best_split = os; best_nl = onl; best_nr = onr; }
    }
    if (active && gl == 0u) {
        if (best_axis == 3 || (best_nl == n && best_nr == n)) N.axis = -1;       // no plane pays, or it separates nothing
        else { N.axis = best_axis; N.split = best_split; }
    }
}

// per reference: bit 0 = goes left, bit 1 = goes right (goes_left / goes_right in kdrefine.h); a leaf's references get 0
__global__ void kb_classify(const LNode* __restrict__ nodes, const uint32_t* __restrict__ ref_tri, const uint32_t* __restrict__ ref_node, uint32_t R,
                            const double* __restrict__ tlo, const double* __restrict__ thi, size_t n_tris, uint8_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > R) return;
    if (i == R) { flags[R] = 0; return; }                      // one extra element: the scan's last value is the total
    const LNode& N = nodes[ref_node[i]];
    uint8_t f = 0;
    if (N.axis >= 0) {
        double lo, hi;
        kb_clipped(tlo, thi, n_tris, ref_tri[i], N.axis, N.lo[N.axis], N.hi[N.axis], lo, hi);
        const double s = N.split;
        if (lo < s || (lo == s && hi == s)) f |= 1;
        if (hi > s) f |= 2;
    }
    flags[i] = f;
}

struct FlagToPair {
    __host__ __device__ __forceinline__ uint64_t operator()(uint8_t f) const { return (uint64_t)(f & 1u) | ((uint64_t)((f >> 1) & 1u) << 32); }
};

// per node: low word 1 if it splits, high word its reference count if it becomes a leaf
__global__ void kb_node_flags(const LNode* __restrict__ nodes, uint32_t N, uint64_t* __restrict__ nflag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > N) return;
    if (j == N) { nflag[N] = 0; return; }
    const LNode& n = nodes[j];
    nflag[j] = n.axis >= 0 ? 1ull : ((uint64_t)n.cnt << 32);
}

__device__ __forceinline__ uint32_t kb_class(uint32_t cnt, int depth, const double* lo, const double* hi, const Opt& opt) {
    const double area = kb_half_area(lo, hi);
    if (opt.max_leaf <= 0 || (int)cnt <= opt.max_leaf || depth >= opt.max_depth || !(area > 0.)) return kClsLeaf;
    return cnt > opt.binned_above ? kClsLarge : (cnt > kSmallMax ? kClsMid : kClsSmall);
}

// One atomic per wave: every lane that calls it together (any subset of the wave, same counter) gets a distinct index.
// (A level of the bench mesh has 10 M leaves and 5 M small nodes: one atomic per thread on one counter was 0.12 of the
// builder's 0.20 s.)
__device__ __forceinline__ uint32_t kb_wave_claim(uint32_t* ctr) {
    const unsigned long long m = __ballot(1);
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0u;
    if ((int)lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// per node of the level: its binary node; a split node's children go to the next level's list
__global__ __launch_bounds__(256) void kb_emit(LNode* __restrict__ nodes, uint32_t N, const uint64_t* __restrict__ nscan, const uint64_t* __restrict__ rscan,
                        qd_kdnode* __restrict__ bn, uint32_t bn_count, uint32_t leaf_base, int depth, Opt opt,
                        LNode* __restrict__ next, uint32_t* __restrict__ lists, uint32_t list_cap, uint32_t* __restrict__ ctl) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    LNode n = nodes[j];
    const uint64_t ns = nscan[j];
    if (n.axis < 0) {
        const uint32_t pos = leaf_base + (uint32_t)(ns >> 32);
        qd_kdnode b; b.split = 0.; b.meta = (pos << 2) | 3u; b.count = n.cnt;
        bn[n.bnode] = b;
        nodes[j].new_off = pos;
        (void)kb_wave_claim(&ctl[4]);
        if (n.cnt > __hip_atomic_load(&ctl[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctl[5], n.cnt);   // (rarely true after the first leaves)
        return;
    }
    const uint64_t S = rscan[n.off], E = rscan[n.off + n.cnt];
    const uint32_t s_l = (uint32_t)S, s_r = (uint32_t)(S >> 32);
    const uint32_t nl = (uint32_t)E - s_l, nr = (uint32_t)(E >> 32) - s_r;
    const uint32_t rank = (uint32_t)ns;
    const uint32_t slot = 2u * rank, child_bn = bn_count + slot;
    qd_kdnode b; b.split = n.split; b.meta = (child_bn << 2) | (uint32_t)n.axis; b.count = 0u;
    bn[n.bnode] = b;
    const uint32_t new_off = s_l + s_r;
    nodes[j].nl = nl; nodes[j].new_off = new_off; nodes[j].s_l = s_l; nodes[j].s_r = s_r; nodes[j].child_slot = slot;
#pragma unroll
    for (uint32_t c = 0; c < 2u; ++c) {
        LNode ch;
        for (int a = 0; a < 3; ++a) { ch.lo[a] = n.lo[a]; ch.hi[a] = n.hi[a]; }
        if (c == 0u) ch.hi[n.axis] = n.split; else ch.lo[n.axis] = n.split;
        ch.split = 0.;
        ch.off = new_off + (c ? nl : 0u); ch.cnt = c ? nr : nl;
        ch.bnode = child_bn + c;
        ch.axis = -1;
        ch.cls = kb_class(ch.cnt, depth + 1, ch.lo, ch.hi, opt);
        ch.aux = 0u; ch.nl = 0u; ch.new_off = 0u; ch.s_l = ch.s_r = 0u; ch.child_slot = 0u; ch.pad_ = 0u;
        for (uint32_t k = kClsSmall; k <= kClsLarge; ++k) {          // (per class: the lanes of one claim share a counter)
            if (ch.cls != k) continue;
            const uint32_t at = kb_wave_claim(&ctl[k]);
            lists[(size_t)k * list_cap + at] = slot + c;
            if (k == kClsLarge) ch.aux = at;
        }
        next[slot + c] = ch;
    }
}

// per reference: to its place in the next level (stable: scan positions), or into the leaf reference array
__global__ void kb_scatter(const LNode* __restrict__ nodes, const uint32_t* __restrict__ ref_tri, const uint32_t* __restrict__ ref_node, uint32_t R,
                           const uint8_t* __restrict__ flags, const uint64_t* __restrict__ rscan, uint32_t tri_first,
                           uint32_t* __restrict__ next_tri, uint32_t* __restrict__ next_node, uint32_t* __restrict__ leaf_refs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const LNode& N = nodes[ref_node[i]];
    const uint32_t t = ref_tri[i];
    if (N.axis < 0) { leaf_refs[N.new_off + (i - N.off)] = tri_first + t; return; }
    const uint8_t f = flags[i];
    const uint64_t sc = rscan[i];
    if (f & 1) {
        const uint32_t p = N.new_off + ((uint32_t)sc - N.s_l);
        next_tri[p] = t; next_node[p] = N.child_slot;
    }
    if (f & 2) {
        const uint32_t p = N.new_off + N.nl + ((uint32_t)(sc >> 32) - N.s_r);
        next_tri[p] = t; next_node[p] = N.child_slot + 1u;
    }
}

// ---------------------------------------------------------------- binary nodes -> two-level records (pack() in kdrefine.h)
struct PackItem { uint32_t node, rec; };

__device__ __forceinline__ void kb_leaf_record(uint32_t* r, uint32_t first, uint32_t count) {
    r[0] = first; r[1] = count; r[2] = 0u; r[3] = 0u; r[4] = 0u; r[5] = 0u; r[6] = 3u; r[7] = 0u;
}

// per item: low word 1 if the record needs a group of four for its grandchildren, high word = items it queues
__global__ void kb_pack_count(const qd_kdnode* __restrict__ bn, const PackItem* __restrict__ items, uint32_t n_items, uint64_t* __restrict__ pflag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_items) return;
    if (k == n_items) { pflag[k] = 0; return; }
    const qd_kdnode b = bn[items[k].node];
    uint64_t v = 0;
    if ((b.meta & 3u) != 3u) {
        const uint32_t c0 = b.meta >> 2;
        const uint32_t inner = ((bn[c0].meta & 3u) != 3u ? 1u : 0u) + ((bn[c0 + 1u].meta & 3u) != 3u ? 1u : 0u);
        if (inner) v = 1ull | ((uint64_t)(2u * inner) << 32);
    }
    pflag[k] = v;
}

__global__ void kb_pack_emit(const qd_kdnode* __restrict__ bn, const PackItem* __restrict__ items, uint32_t n_items, const uint64_t* __restrict__ pscan,
                             uint32_t rec_count, uint32_t* __restrict__ recs /* 8 words per record */, PackItem* __restrict__ next_items) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_items) return;
    const PackItem it = items[k];
    const qd_kdnode b = bn[it.node];
    uint32_t r[8];
    if ((b.meta & 3u) == 3u) {
        kb_leaf_record(r, b.meta >> 2, b.count);
    } else {
        const uint64_t ps = pscan[k];
        const uint32_t c0 = b.meta >> 2;
        const qd_kdnode ch0 = bn[c0], ch1 = bn[c0 + 1u];
        const bool in0 = (ch0.meta & 3u) != 3u, in1 = (ch1.meta & 3u) != 3u;
        const uint32_t base = (in0 || in1) ? rec_count + 4u * (uint32_t)ps : 0u;
        uint32_t kid = (uint32_t)(ps >> 32);
        const unsigned long long sb = (unsigned long long)__double_as_longlong(b.split);
        r[0] = (uint32_t)sb; r[1] = (uint32_t)(sb >> 32);
        uint32_t meta = (b.meta & 3u) | ((base >> 2) << 6);
#pragma unroll
        for (uint32_t c = 0; c < 2u; ++c) {
            const qd_kdnode ch = c ? ch1 : ch0;
            const bool inner = c ? in1 : in0;
            meta |= (ch.meta & 3u) << (2u + 2u * c);
            if (!inner) {
                r[2 + 2 * c] = ch.meta >> 2; r[3 + 2 * c] = ch.count;
                if (in0 || in1) {                                  // the two slots of the group nobody will fill
                    kb_leaf_record(recs + (size_t)(base + 2u * c) * 8, 0u, 0u);
                    kb_leaf_record(recs + (size_t)(base + 2u * c + 1u) * 8, 0u, 0u);
                }
            } else {
                const unsigned long long sc = (unsigned long long)__double_as_longlong(ch.split);
                r[2 + 2 * c] = (uint32_t)sc; r[3 + 2 * c] = (uint32_t)(sc >> 32);
                const uint32_t gc = ch.meta >> 2;
                next_items[kid].node = gc; next_items[kid].rec = base + 2u * c; ++kid;
                next_items[kid].node = gc + 1u; next_items[kid].rec = base + 2u * c + 1u; ++kid;
            }
        }
        r[6] = meta; r[7] = 0u;
    }
    uint4* const dst = (uint4*)(recs + (size_t)it.rec * 8);
    dst[0] = make_uint4(r[0], r[1], r[2], r[3]);
    dst[1] = make_uint4(r[4], r[5], r[6], r[7]);
}

__global__ void kb_root_group(uint32_t* __restrict__ recs, uint32_t rec0) {
    if (threadIdx.x < 4u) kb_leaf_record(recs + (size_t)(rec0 + threadIdx.x) * 8, 0u, 0u);
}

// ---------------------------------------------------------------- host driver
// A device buffer that grows (contents kept up to `keep` bytes).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    hipError_t reserve(size_t want, size_t keep = 0) {
        if (want <= bytes) return hipSuccess;
        const size_t grown = std::max(want, bytes + bytes / 2);
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, grown);
        if (e != hipSuccess) return e;
        if (p && keep) { e = hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice); if (e != hipSuccess) { (void)hipFree(q); return e; } }
        if (p) (void)hipFree(p);
        p = q; bytes = grown;
        return hipSuccess;
    }
    void* release() { void* q = p; p = nullptr; bytes = 0; return q; }
    template <class T> T* as() const { return (T*)p; }
};

// The trees of a scene: records and leaf references of all meshes, in two device arrays that grow mesh by mesh.
struct Trees {
    DevBuf records;                 // 32-byte records; a multiple of 4 per tree
    DevBuf leaf_refs;               // GLOBAL triangle indices
    uint64_t n_records = 0, n_refs = 0;
    uint64_t n_nodes = 0, leaves = 0, max_leaf = 0;
};

#define KB_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e_); } while (0)
#ifdef QD_REFINE_TIMING
#define KB_MARK(what) do { (void)hipDeviceSynchronize(); const auto now_ = std::chrono::steady_clock::now(); std::fprintf(stderr, "[kd build, device] %s: %.2f ms\n", what, std::chrono::duration<double>(now_ - mark_).count() * 1e3); mark_ = now_; } while (0)
#else
#define KB_MARK(what) do {} while (0)
#endif

inline unsigned kb_blocks(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// Builds the tree of one mesh (`tri_verts_dev` = the scene's triangle array on the device) and appends it to `out`.
// Returns an empty string and the record index of the root, or an error message.
inline std::string build_mesh(const double* tri_verts_dev, const qd_mesh& m, const Opt& opt, Trees& out, uint32_t* root_record, hipStream_t stream = nullptr) {
#ifdef QD_REFINE_TIMING
    auto mark_ = std::chrono::steady_clock::now();
#endif
    const uint32_t n_tris = m.tri_count;
    if (n_tris == 0u) return "mesh without triangles";
    DevBuf tlo, thi, ref_tri[2], ref_node[2], flags, rscan, nflag, nscan, lnodes[2], lists, hist, bn, ctl, scan_tmp;
    KB_TRY(tlo.reserve((size_t)n_tris * 3 * sizeof(double)));
    KB_TRY(thi.reserve((size_t)n_tris * 3 * sizeof(double)));
    hipLaunchKernelGGL(kb_tri_aabb, dim3(kb_blocks(n_tris, 256)), dim3(256), 0, stream, tri_verts_dev + (size_t)m.tri_first * 9, n_tris, tlo.as<double>(), thi.as<double>());
    size_t ref_cap = (size_t)n_tris + n_tris / 2 + 1024;
    for (int i = 0; i < 2; ++i) { KB_TRY(ref_tri[i].reserve(ref_cap * 4)); KB_TRY(ref_node[i].reserve(ref_cap * 4)); }
    hipLaunchKernelGGL(kb_init_refs, dim3(kb_blocks(n_tris, 256)), dim3(256), 0, stream, ref_tri[0].as<uint32_t>(), ref_node[0].as<uint32_t>(), n_tris);
    KB_TRY(ctl.reserve(kCtlWords * sizeof(uint32_t)));
    KB_TRY(hipMemsetAsync(ctl.p, 0, kCtlWords * sizeof(uint32_t), stream));
    size_t bn_cap = (size_t)n_tris * 3 + 1024;
    KB_TRY(bn.reserve(bn_cap * sizeof(qd_kdnode)));
    uint64_t bn_count = 1;

    // level 0: the root
    LNode root{};
    for (int a = 0; a < 3; ++a) { root.lo[a] = m.bbox_min[a]; root.hi[a] = m.bbox_max[a]; }
    root.off = 0; root.cnt = n_tris; root.bnode = 0; root.axis = -1;
    {
        const double dx = root.hi[0] - root.lo[0], dy = root.hi[1] - root.lo[1], dz = root.hi[2] - root.lo[2];
        const double area = dx * dy + dy * dz + dz * dx;
        if (opt.max_leaf <= 0 || (int)n_tris <= opt.max_leaf || 0 >= opt.max_depth || !(area > 0.)) root.cls = kClsLeaf;
        else root.cls = n_tris > opt.binned_above ? kClsLarge : (n_tris > kSmallMax ? kClsMid : kClsSmall);
    }
    uint32_t N = 1, R = n_tris;
    uint32_t n_cls[4] = {0, 0, 0, 0};
    n_cls[root.cls] = 1;
    size_t node_cap = 1024;
    for (int i = 0; i < 2; ++i) KB_TRY(lnodes[i].reserve(node_cap * sizeof(LNode)));
    KB_TRY(hipMemcpyAsync(lnodes[0].p, &root, sizeof root, hipMemcpyHostToDevice, stream));
    size_t list_cap = 1024;
    KB_TRY(lists.reserve(4 * list_cap * sizeof(uint32_t)));
    KB_TRY(hipMemsetAsync(lists.p, 0, 4 * list_cap * sizeof(uint32_t), stream));        // every class list of level 0 = {node 0}
    const uint64_t leaf_base0 = out.n_refs;
    uint64_t leaf_count = leaf_base0;
    KB_MARK("set-up");

    int cur = 0;
    for (int depth = 0; N != 0u; ++depth) {
        if (depth > opt.max_depth + 1) return "device KD build did not terminate";
        LNode* const nodes = lnodes[cur].as<LNode>();
        const uint32_t* const rt = ref_tri[cur].as<uint32_t>();
        const uint32_t* const rn = ref_node[cur].as<uint32_t>();
        const uint32_t* const L = lists.as<uint32_t>();
        // 1. planes
        if (n_cls[kClsLarge]) {
            KB_TRY(hist.reserve((size_t)n_cls[kClsLarge] * kHistWords * sizeof(uint32_t)));
            KB_TRY(hipMemsetAsync(hist.p, 0, (size_t)n_cls[kClsLarge] * kHistWords * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(kb_hist, dim3(kb_blocks(R, kHistChunk)), dim3(256), 0, stream, nodes, rt, rn, R, tlo.as<double>(), thi.as<double>(), (size_t)n_tris, hist.as<uint32_t>());
            hipLaunchKernelGGL(kb_pick_binned, dim3(n_cls[kClsLarge]), dim3(64), 0, stream, nodes, L + (size_t)kClsLarge * list_cap, hist.as<uint32_t>(), opt.cost_step);
        }
        if (n_cls[kClsMid])
            hipLaunchKernelGGL((kb_pick_exact<64, 192>), dim3(n_cls[kClsMid]), dim3(64), 0, stream, nodes, L + (size_t)kClsMid * list_cap, n_cls[kClsMid], rt,
                               tlo.as<double>(), thi.as<double>(), (size_t)n_tris, opt.cost_step);
        if (n_cls[kClsSmall])
            hipLaunchKernelGGL((kb_pick_exact<8, (int)kSmallMax>), dim3(kb_blocks(n_cls[kClsSmall], 8)), dim3(64), 0, stream, nodes, L + (size_t)kClsSmall * list_cap,
                               n_cls[kClsSmall], rt, tlo.as<double>(), thi.as<double>(), (size_t)n_tris, opt.cost_step);
        // 2. flags and their sums
        KB_TRY(flags.reserve((size_t)R + 1));
        KB_TRY(rscan.reserve(((size_t)R + 1) * sizeof(uint64_t)));
        KB_TRY(nflag.reserve(((size_t)N + 1) * sizeof(uint64_t)));
        KB_TRY(nscan.reserve(((size_t)N + 1) * sizeof(uint64_t)));
        hipLaunchKernelGGL(kb_classify, dim3(kb_blocks((uint64_t)R + 1, 256)), dim3(256), 0, stream, nodes, rt, rn, R, tlo.as<double>(), thi.as<double>(), (size_t)n_tris, flags.as<uint8_t>());
        hipLaunchKernelGGL(kb_node_flags, dim3(kb_blocks((uint64_t)N + 1, 256)), dim3(256), 0, stream, nodes, N, nflag.as<uint64_t>());
        {
            hipcub::TransformInputIterator<uint64_t, FlagToPair, const uint8_t*> in(flags.as<uint8_t>(), FlagToPair());
            size_t need_r = 0, need_n = 0;
            KB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need_r, in, rscan.as<uint64_t>(), (int)(R + 1u), stream));
            KB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need_n, nflag.as<uint64_t>(), nscan.as<uint64_t>(), (int)(N + 1u), stream));
            KB_TRY(scan_tmp.reserve(std::max(need_r, need_n)));
            size_t tmp = scan_tmp.bytes;
            KB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, tmp, in, rscan.as<uint64_t>(), (int)(R + 1u), stream));
            tmp = scan_tmp.bytes;
            KB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, tmp, nflag.as<uint64_t>(), nscan.as<uint64_t>(), (int)(N + 1u), stream));
        }
        uint64_t tot_r = 0, tot_n = 0;
        KB_TRY(hipMemcpyAsync(&tot_r, rscan.as<uint64_t>() + R, sizeof tot_r, hipMemcpyDeviceToHost, stream));
        KB_TRY(hipMemcpyAsync(&tot_n, nscan.as<uint64_t>() + N, sizeof tot_n, hipMemcpyDeviceToHost, stream));
        KB_TRY(hipStreamSynchronize(stream));
        const uint64_t next_R = (tot_r & 0xffffffffull) + (tot_r >> 32);
        const uint64_t next_N = 2 * (tot_n & 0xffffffffull);
        const uint64_t new_leaf_refs = tot_n >> 32;
        if (next_R >= (1ull << 31) || leaf_count + new_leaf_refs >= (1ull << 30) || bn_count + next_N >= (1ull << 30))
            return "KD tree too large for the device builder's 30-bit indices";
        // 3. room for the next level, then emit + scatter
        const int nxt = cur ^ 1;
        if (next_R > ref_cap) {
            ref_cap = (size_t)next_R + next_R / 4;
            KB_TRY(ref_tri[nxt].reserve(ref_cap * 4)); KB_TRY(ref_node[nxt].reserve(ref_cap * 4));
            KB_TRY(ref_tri[cur].reserve(ref_cap * 4, (size_t)R * 4)); KB_TRY(ref_node[cur].reserve(ref_cap * 4, (size_t)R * 4));
        }
        if (next_N > node_cap) {
            node_cap = (size_t)next_N + next_N / 4;
            KB_TRY(lnodes[nxt].reserve(node_cap * sizeof(LNode)));
            KB_TRY(lnodes[cur].reserve(node_cap * sizeof(LNode), (size_t)N * sizeof(LNode)));
        }
        if (next_N > list_cap) {                                           // (the lists of the current level have been consumed)
            list_cap = (size_t)next_N + next_N / 4;
            KB_TRY(lists.reserve(4 * list_cap * sizeof(uint32_t)));
        }
        if (bn_count + next_N > bn_cap) {
            bn_cap = (size_t)(bn_count + next_N) * 2;
            KB_TRY(bn.reserve(bn_cap * sizeof(qd_kdnode), (size_t)bn_count * sizeof(qd_kdnode)));
        }
        KB_TRY(out.leaf_refs.reserve((size_t)(leaf_count + new_leaf_refs + 64) * sizeof(uint32_t), (size_t)leaf_count * sizeof(uint32_t)));
        KB_TRY(hipMemsetAsync(ctl.as<uint32_t>() + 1, 0, 3 * sizeof(uint32_t), stream));
        // (buffers may have moved: take the pointers again)
        hipLaunchKernelGGL(kb_emit, dim3(kb_blocks(N, 256)), dim3(256), 0, stream, lnodes[cur].as<LNode>(), N, nscan.as<uint64_t>(), rscan.as<uint64_t>(),
                           bn.as<qd_kdnode>(), (uint32_t)bn_count, (uint32_t)leaf_count, depth, opt, lnodes[nxt].as<LNode>(), lists.as<uint32_t>(), (uint32_t)list_cap, ctl.as<uint32_t>());
        if (R)
            hipLaunchKernelGGL(kb_scatter, dim3(kb_blocks(R, 256)), dim3(256), 0, stream, lnodes[cur].as<LNode>(), ref_tri[cur].as<uint32_t>(), ref_node[cur].as<uint32_t>(), R,
                               flags.as<uint8_t>(), rscan.as<uint64_t>(), m.tri_first, ref_tri[nxt].as<uint32_t>(), ref_node[nxt].as<uint32_t>(), out.leaf_refs.as<uint32_t>());
        uint32_t h_ctl[kCtlWords];
        KB_TRY(hipMemcpyAsync(h_ctl, ctl.p, sizeof h_ctl, hipMemcpyDeviceToHost, stream));
        KB_TRY(hipStreamSynchronize(stream));
        KB_TRY(hipGetLastError());
        n_cls[kClsSmall] = h_ctl[kClsSmall]; n_cls[kClsMid] = h_ctl[kClsMid]; n_cls[kClsLarge] = h_ctl[kClsLarge];
        bn_count += next_N; leaf_count += new_leaf_refs;
        N = (uint32_t)next_N; R = (uint32_t)next_R; cur = nxt;
#ifdef QD_REFINE_TIMING
        std::fprintf(stderr, "[kd build, device] level %d -> %u nodes (%u / %u / %u small / mid / large), %u references\n", depth + 1, N, n_cls[1], n_cls[2], n_cls[3], R);
#endif
    }
    KB_MARK("levels");
    uint32_t h_ctl[kCtlWords];
    KB_TRY(hipMemcpy(h_ctl, ctl.p, sizeof h_ctl, hipMemcpyDeviceToHost));
    out.leaves += h_ctl[4]; out.max_leaf = std::max<uint64_t>(out.max_leaf, h_ctl[5]);
    out.n_nodes += bn_count;
    out.n_refs = leaf_count;

    // records: level by level over the record tree (a record = a node and its two children)
    DevBuf items[2], pflag, pscan;
    size_t item_cap = 1024;
    for (int i = 0; i < 2; ++i) KB_TRY(items[i].reserve(item_cap * sizeof(PackItem)));
    uint64_t rec_count = out.n_records;
    const uint32_t rec0 = (uint32_t)rec_count;
    KB_TRY(out.records.reserve((size_t)(rec_count + 4 + bn_count) * 32, (size_t)rec_count * 32));   // a first guess: a record per node; grown below
    hipLaunchKernelGGL(kb_root_group, dim3(1), dim3(64), 0, stream, out.records.as<uint32_t>(), rec0);
    rec_count += 4;
    const PackItem first_item{0u, rec0};
    KB_TRY(hipMemcpyAsync(items[0].p, &first_item, sizeof first_item, hipMemcpyHostToDevice, stream));
    uint32_t n_items = 1;
    cur = 0;
    while (n_items) {
        KB_TRY(pflag.reserve(((size_t)n_items + 1) * 8)); KB_TRY(pscan.reserve(((size_t)n_items + 1) * 8));
        hipLaunchKernelGGL(kb_pack_count, dim3(kb_blocks((uint64_t)n_items + 1, 256)), dim3(256), 0, stream, bn.as<qd_kdnode>(), items[cur].as<PackItem>(), n_items, pflag.as<uint64_t>());
        size_t need = 0;
        KB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, pflag.as<uint64_t>(), pscan.as<uint64_t>(), (int)(n_items + 1u), stream));
        KB_TRY(scan_tmp.reserve(need));
        size_t tmp = scan_tmp.bytes;
        KB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, tmp, pflag.as<uint64_t>(), pscan.as<uint64_t>(), (int)(n_items + 1u), stream));
        uint64_t tot = 0;
        KB_TRY(hipMemcpyAsync(&tot, pscan.as<uint64_t>() + n_items, sizeof tot, hipMemcpyDeviceToHost, stream));
        KB_TRY(hipStreamSynchronize(stream));
        const uint64_t groups = tot & 0xffffffffull, kids = tot >> 32;
        if (rec_count + 4 * groups >= (1ull << 28)) return "KD tree too large for the 28-bit record index";
        KB_TRY(out.records.reserve((size_t)(rec_count + 4 * groups) * 32, (size_t)rec_count * 32));
        const int nxt = cur ^ 1;
        if (kids > item_cap) {
            item_cap = (size_t)kids + kids / 4;
            KB_TRY(items[nxt].reserve(item_cap * sizeof(PackItem)));
            KB_TRY(items[cur].reserve(item_cap * sizeof(PackItem), (size_t)n_items * sizeof(PackItem)));
        }
        hipLaunchKernelGGL(kb_pack_emit, dim3(kb_blocks(n_items, 256)), dim3(256), 0, stream, bn.as<qd_kdnode>(), items[cur].as<PackItem>(), n_items, pscan.as<uint64_t>(),
                           (uint32_t)rec_count, out.records.as<uint32_t>(), items[nxt].as<PackItem>());
        rec_count += 4 * groups;
        n_items = (uint32_t)kids; cur = nxt;
    }
    KB_TRY(hipStreamSynchronize(stream));
    KB_TRY(hipGetLastError());
    out.n_records = rec_count;
    *root_record = rec0;
    KB_MARK("records");
    return std::string();
}

}  // namespace qd_kdbuild
