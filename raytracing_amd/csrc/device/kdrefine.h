// Host-side construction of the DEVICE KD tree, run once in qd_scene_create.
//
// Input: the caller's tree (the reference's builder, src/mesh.cpp:216-432, mirrored by libqdhost).
// Output: a tree over the same triangles that the trace kernel walks:
//  1. Every split plane and every leaf list of the caller's tree is kept.
//  2. The reference stops splitting at 20 triangles per leaf (TRIANGLES_PER_LEAF, src/constants.h:14,
//     src/mesh.cpp:218) because a triangle test is cheap next to a recursive box-gated descent on a CPU.  On
//     the device a triangle test is 72 bytes of gathered f64 data and ~100 f64 instructions, a descent step
//     is a fraction of a 16-byte load.  So every leaf is split further with a surface-area heuristic until
//     splitting stops paying (SURVEY.md section 8(f).3 asks for a proper SAH for the large scene).
//  3. The binary tree is packed into 32-byte records that hold two levels each (pack()).
// The closest hit of a ray does not depend on how triangles are grouped; only an exact distance tie between
// two different triangles can resolve differently.
//
// Classification of a triangle against a plane follows the reference's PartitionTriangle
// (src/mesh.cpp:412-432) on the triangle's bounds clipped to the node box: below the plane somewhere -> left,
// above somewhere -> right, lying in the plane -> left.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "qdtrace.h"

namespace qd_refine {

struct Box { double lo[3], hi[3]; };

struct Options {
    int max_leaf = 2;          // leaves with at most this many triangles are never split; 0 = keep the caller's leaves
    double cost_step = 0.6;    // cost of one descent step, in triangle tests
    int max_depth = 70;        // below the device traversal stack (kKdStack)
    bool rebuild = false;      // true: ignore the caller's planes, build the whole tree with the surface-area heuristic
    uint32_t binned_above = 192;   // nodes with more triangles choose their plane from 64 bins per axis, smaller ones by an exact sweep
};

inline double half_area(const Box& b) {
    const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

struct TriBounds { double lo[3], hi[3]; };

inline TriBounds tri_bounds(const double* tri_verts, uint32_t global_tri, const Box& box) {
    const double* v = tri_verts + (size_t)global_tri * 9;      // A, AB, AC
    TriBounds b;
    for (int a = 0; a < 3; ++a) {
        const double p0 = v[a], p1 = v[a] + v[3 + a], p2 = v[a] + v[6 + a];
        b.lo[a] = std::max(std::min(p0, std::min(p1, p2)), box.lo[a]);
        b.hi[a] = std::min(std::max(p0, std::max(p1, p2)), box.hi[a]);
        if (b.hi[a] < b.lo[a]) b.hi[a] = b.lo[a];              // bounds miss the box by rounding: keep a point
    }
    return b;
}
inline bool goes_left(double lo, double hi, double s) { return lo < s || (lo == s && hi == s); }
inline bool goes_right(double hi, double s) { return hi > s; }

// binary node: axis 0..2 inner (split, child = left child, right = child + 1); axis 3 leaf (first, count into refs)
struct BNode { double split; uint32_t child; uint32_t count; uint8_t axis; };

struct SubTree {
    std::vector<BNode> nodes;          // node 0 = subtree root; child indices local
    std::vector<uint32_t> refs;        // mesh-local triangle indices
};

// Plane of a node with many triangles: 64 bins per axis over the clipped triangle bounds (every `stride`-th triangle is
// binned: a sample is enough for the histograms of a node with millions of triangles).  The candidates are the 63 bin
// boundaries per axis; the counts left / right of a boundary are what the exact classification (goes_left / goes_right)
// would give up to the triangles that end exactly on it.
inline void binned_plane(const Box& box, const TriBounds* tb, uint32_t n, uint32_t stride, const Options& opt,
                         double& best_cost, int& best_axis, double& best_split) {
    constexpr int K = 64;
    const double area = half_area(box);
    const double scale = (double)stride;
    for (int a = 0; a < 3; ++a) {
        const double L = box.lo[a], W = box.hi[a] - box.lo[a];
        if (!(W > 0.)) continue;
        uint32_t starts[K] = {0}, ends[K] = {0};
        const double inv = K / W;
        for (uint32_t i = 0; i < n; i += stride) {
            int b0 = (int)((tb[i].lo[a] - L) * inv), b1 = (int)((tb[i].hi[a] - L) * inv);
            b0 = std::min(std::max(b0, 0), K - 1); b1 = std::min(std::max(b1, 0), K - 1);
            starts[b0]++; ends[b1]++;
        }
        const int u = (a + 1) % 3, v = (a + 2) % 3;
        const double du = box.hi[u] - box.lo[u], dv = box.hi[v] - box.lo[v];
        uint32_t below = 0, total = 0;
        for (int b = 0; b < K; ++b) total += ends[b];
        uint32_t ended = 0;
        for (int b = 1; b < K; ++b) {
            below += starts[b - 1]; ended += ends[b - 1];
            const double s = L + W * ((double)b / K);
            const double nl = below * scale, nr = (total - ended) * scale;          // starts before the boundary; ends in a bin at or after it
            const double dl = s - box.lo[a], dr = box.hi[a] - s;
            const double al = du * dv + (du + dv) * dl, ar = du * dv + (du + dv) * dr;
            double cost = opt.cost_step + (al * nl + ar * nr) / area;
            if (below == 0 || total == ended) cost *= 0.8;                           // cutting off empty space
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = s; }
        }
    }
}

class Builder {
public:
    Builder(const double* tri_verts, uint32_t tri_first, const Options& opt) : tv_(tri_verts), first_(tri_first), opt_(opt) {}

    void build(SubTree& out, const Box& box, const std::vector<uint32_t>& tris, int depth) {
        out.nodes.clear(); out.refs.clear();
        out.nodes.push_back(BNode{0., 0u, 0u, 3});
        arena_.assign(tris.begin(), tris.end());
        split(out, 0, box, 0, (uint32_t)tris.size(), depth);
    }

private:
    const double* tv_;
    uint32_t first_;
    Options opt_;
    // Working memory reused from node to node (a 10 M triangle mesh has 20 M nodes: an allocation or two per node was half
    // of the build).  arena_ is a stack of triangle lists: a node's list sits at [off, off + n); its children's lists are
    // pushed behind everything, used by the recursion and popped again.
    std::vector<uint32_t> arena_;
    std::vector<TriBounds> tb_;
    std::vector<double> los_, his_, flat_, cand_;

    void make_leaf(SubTree& out, size_t at, size_t off, uint32_t n) {
        out.nodes[at] = BNode{0., (uint32_t)out.refs.size(), n, 3};
        out.refs.insert(out.refs.end(), arena_.begin() + off, arena_.begin() + off + n);
    }

    void split(SubTree& out, size_t at, const Box& box, size_t off, uint32_t n, int depth) {
        const double area = half_area(box);
        if (opt_.max_leaf <= 0 || (int)n <= opt_.max_leaf || depth >= opt_.max_depth || !(area > 0.)) { make_leaf(out, at, off, n); return; }
        tb_.resize(n);
        for (uint32_t i = 0; i < n; ++i) tb_[i] = tri_bounds(tv_, first_ + arena_[off + i], box);
        double best_cost = (double)n;          // cost of staying a leaf
        int best_axis = -1;
        double best_split = 0.;
        if (n > opt_.binned_above) binned_plane(box, tb_.data(), n, 1, opt_, best_cost, best_axis, best_split);
        for (int a = 0; a < 3 && n <= opt_.binned_above; ++a) {
            if (!(box.hi[a] > box.lo[a])) continue;
            los_.resize(n); his_.resize(n); flat_.clear();
            for (uint32_t i = 0; i < n; ++i) {
                los_[i] = tb_[i].lo[a]; his_[i] = tb_[i].hi[a];
                if (los_[i] == his_[i]) flat_.push_back(los_[i]);
            }
            std::sort(los_.begin(), los_.end()); std::sort(his_.begin(), his_.end());
            std::sort(flat_.begin(), flat_.end());
            cand_.resize(2 * (size_t)n);                                   // candidates = every bound, ascending, once each
            std::merge(los_.begin(), los_.end(), his_.begin(), his_.end(), cand_.begin());
            cand_.erase(std::unique(cand_.begin(), cand_.end()), cand_.end());
            const int u = (a + 1) % 3, v = (a + 2) % 3;
            const double du = box.hi[u] - box.lo[u], dv = box.hi[v] - box.lo[v];
            size_t pl = 0, ph = 0, pf = 0;                                 // sweep: bounds below / at or below the candidate
            for (double s : cand_) {
                while (pl < n && los_[pl] < s) ++pl;                       // triangles that start before s
                while (ph < n && his_[ph] <= s) ++ph;                      // triangles that end at or before s
                while (pf < flat_.size() && flat_[pf] < s) ++pf;
                if (!(s > box.lo[a] && s < box.hi[a])) continue;
                size_t pe = pf;
                while (pe < flat_.size() && flat_[pe] == s) ++pe;          // flat triangles lying in the plane go left
                const uint32_t nl = (uint32_t)(pl + (pe - pf));
                const uint32_t nr = n - (uint32_t)ph;
                const double dl = s - box.lo[a], dr = box.hi[a] - s;
                const double al = du * dv + (du + dv) * dl, ar = du * dv + (du + dv) * dr;
                double cost = opt_.cost_step + (al * nl + ar * nr) / area;
                if (nl == 0 || nr == 0) cost *= 0.8;           // cutting off empty space
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = s; }
            }
        }
        if (best_axis < 0) { make_leaf(out, at, off, n); return; }
        const size_t l0 = arena_.size();
        for (uint32_t i = 0; i < n; ++i) if (goes_left(tb_[i].lo[best_axis], tb_[i].hi[best_axis], best_split)) arena_.push_back(arena_[off + i]);
        const size_t r0 = arena_.size();
        for (uint32_t i = 0; i < n; ++i) if (goes_right(tb_[i].hi[best_axis], best_split)) arena_.push_back(arena_[off + i]);
        const uint32_t nl = (uint32_t)(r0 - l0), nr = (uint32_t)(arena_.size() - r0);
        if (nl == n && nr == n) { arena_.resize(l0); make_leaf(out, at, off, n); return; }
        const size_t c = out.nodes.size();
        out.nodes.push_back(BNode{0., 0u, 0u, 3});
        out.nodes.push_back(BNode{0., 0u, 0u, 3});
        out.nodes[at] = BNode{best_split, (uint32_t)c, 0u, (uint8_t)best_axis};
        Box bl = box, br = box;
        bl.hi[best_axis] = best_split; br.lo[best_axis] = best_split;
        split(out, c, bl, l0, nl, depth + 1);
        split(out, c + 1, br, r0, nr, depth + 1);
        arena_.resize(l0);
    }
};

struct MeshTree {
    std::vector<BNode> nodes;          // node 0 = root
    std::vector<uint32_t> refs;        // mesh-local triangle indices
};

struct Result {
    std::vector<MeshTree> trees;       // per mesh (empty nodes = no tree)
    uint64_t n_nodes = 0, n_refs = 0, leaves = 0, max_leaf = 0;
};

// Returns an empty string, or an error message.
#ifdef QD_REFINE_TIMING
#define QD_REFINE_MARK(what) do { const auto now_ = std::chrono::steady_clock::now(); std::fprintf(stderr, "[kd build] %s: %.3f s\n", what, std::chrono::duration<double>(now_ - mark_).count()); mark_ = now_; } while (0)
#else
#define QD_REFINE_MARK(what) do {} while (0)
#endif
inline std::string refine(const qd_scene_desc* d, const Options& opt, Result& R) {
#ifdef QD_REFINE_TIMING
    auto mark_ = std::chrono::steady_clock::now();
#endif
    R = Result();
    R.trees.resize(d->n_meshes);
    for (uint32_t mi = 0; mi < d->n_meshes; ++mi) {
        const qd_mesh& m = d->meshes[mi];
        if (m.kd_root < 0) continue;
        MeshTree& T = R.trees[mi];
        // 1. the caller's inner nodes, copied; one job per leaf
        struct Job { size_t at; Box box; std::vector<uint32_t> tris; int depth; };
        std::vector<Job> jobs;
        if (opt.rebuild) {
            // whole tree by the surface-area heuristic: the top of the tree is split here, one node after the other (planes from
            // binned, sampled bounds; the classification of every triangle is exact), until a node holds few enough triangles
            // to be one of the parallel jobs below
            const uint32_t job_tris = 32768;
            Box root;
            for (int a = 0; a < 3; ++a) { root.lo[a] = m.bbox_min[a]; root.hi[a] = m.bbox_max[a]; }
            T.nodes.push_back(BNode{0., 0u, 0u, 3});
            std::vector<uint32_t> all(m.tri_count);
            for (uint32_t i = 0; i < m.tri_count; ++i) all[i] = i;
            // Level by level, on the host cores: while a level has fewer nodes than there are threads each node's triangles are
            // classified in parallel slices (kept in order), afterwards the nodes of a level are independent tasks.  Nodes are
            // numbered in job order after each level, so the tree does not depend on thread timing.
            unsigned n_thr = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
            if (m.tri_count < 200000) n_thr = 1;
            struct Cut { int axis = -1; double split = 0.; std::vector<uint32_t> left, right; };
            auto classify = [&](const Job& j, int axis, double split, uint32_t b, uint32_t e, std::vector<uint32_t>& left, std::vector<uint32_t>& right) {
                left.reserve((e - b) / 2 + (e - b) / 8); right.reserve((e - b) / 2 + (e - b) / 8);
                for (uint32_t i = b; i < e; ++i) {
                    const double* v = d->tri_verts + (size_t)(m.tri_first + j.tris[i]) * 9;
                    const double p0 = v[axis], p1 = v[axis] + v[3 + axis], p2 = v[axis] + v[6 + axis];
                    double lo = std::max(std::min(p0, std::min(p1, p2)), j.box.lo[axis]);
                    double hi = std::min(std::max(p0, std::max(p1, p2)), j.box.hi[axis]);
                    if (hi < lo) hi = lo;
                    if (goes_left(lo, hi, split)) left.push_back(j.tris[i]);
                    if (goes_right(hi, split)) right.push_back(j.tris[i]);
                }
            };
            auto cut = [&](const Job& j, Cut& c, unsigned slices) {
                const uint32_t n = (uint32_t)j.tris.size();
                const uint32_t stride = std::max(1u, n / 262144u);
                std::vector<TriBounds> tb((n + stride - 1) / stride);
                for (uint32_t i = 0, k = 0; i < n; i += stride, ++k) tb[k] = tri_bounds(d->tri_verts, m.tri_first + j.tris[i], j.box);
                double best_cost = (double)n;
                binned_plane(j.box, tb.data(), (uint32_t)tb.size(), 1, opt, best_cost, c.axis, c.split);
                if (c.axis < 0) return;       // (histogram counts are those of the sample: any plane found beats a leaf of this size)
                if (slices <= 1) { classify(j, c.axis, c.split, 0, n, c.left, c.right); return; }
                std::vector<std::vector<uint32_t>> L(slices), Rr(slices);
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < slices; ++t)
                    pool.emplace_back([&, t]() { classify(j, c.axis, c.split, (uint32_t)((uint64_t)n * t / slices), (uint32_t)((uint64_t)n * (t + 1) / slices), L[t], Rr[t]); });
                for (auto& t : pool) t.join();
                for (unsigned t = 0; t < slices; ++t) { c.left.insert(c.left.end(), L[t].begin(), L[t].end()); c.right.insert(c.right.end(), Rr[t].begin(), Rr[t].end()); }
            };
            std::vector<Job> level;
            level.push_back(Job{0, root, std::move(all), 0});
            while (!level.empty()) {
                std::vector<Job> work;
                for (Job& j : level) {
                    if (j.tris.size() <= job_tris || j.depth >= opt.max_depth - 24) jobs.push_back(std::move(j));
                    else work.push_back(std::move(j));
                }
                level.clear();
                std::vector<Cut> cuts(work.size());
                if (work.size() < n_thr) for (size_t k = 0; k < work.size(); ++k) cut(work[k], cuts[k], n_thr);
                else {
                    std::atomic<size_t> next_job{0};
                    std::vector<std::thread> pool;
                    for (unsigned t = 0; t < n_thr; ++t)
                        pool.emplace_back([&]() { for (size_t k; (k = next_job.fetch_add(1)) < work.size();) cut(work[k], cuts[k], 1); });
                    for (auto& t : pool) t.join();
                }
                for (size_t k = 0; k < work.size(); ++k) {
                    Job& j = work[k]; Cut& c = cuts[k];
                    const size_t n = j.tris.size();
                    if (c.axis < 0 || (c.left.size() == n && c.right.size() == n)) { jobs.push_back(std::move(j)); continue; }
                    const size_t ch = T.nodes.size();
                    T.nodes.push_back(BNode{0., 0u, 0u, 3});
                    T.nodes.push_back(BNode{0., 0u, 0u, 3});
                    T.nodes[j.at] = BNode{c.split, (uint32_t)ch, 0u, (uint8_t)c.axis};
                    Box bl = j.box, br = j.box;
                    bl.hi[c.axis] = c.split; br.lo[c.axis] = c.split;
                    level.push_back(Job{ch, bl, std::move(c.left), j.depth + 1});
                    level.push_back(Job{ch + 1, br, std::move(c.right), j.depth + 1});
                }
            }
        } else {
            struct Item { uint32_t src; size_t dst; Box box; int depth; };
            std::vector<Item> stack;
            Box root;
            for (int a = 0; a < 3; ++a) { root.lo[a] = m.bbox_min[a]; root.hi[a] = m.bbox_max[a]; }
            T.nodes.push_back(BNode{0., 0u, 0u, 3});
            stack.push_back(Item{(uint32_t)m.kd_root, 0, root, 0});
            while (!stack.empty()) {
                const Item it = stack.back(); stack.pop_back();
                const qd_kdnode n = d->kd_nodes[it.src];
                const uint32_t axis = n.meta & 3u, idx = n.meta >> 2;
                if (axis == 3u) {
                    jobs.push_back(Job{it.dst, it.box, std::vector<uint32_t>(d->leaf_tris + idx, d->leaf_tris + idx + n.count), it.depth});
                    continue;
                }
                if (it.depth >= opt.max_depth) return "KD tree deeper than the device traversal stack allows";
                const size_t c = T.nodes.size();
                T.nodes.push_back(BNode{0., 0u, 0u, 3});
                T.nodes.push_back(BNode{0., 0u, 0u, 3});
                T.nodes[it.dst] = BNode{n.split, (uint32_t)c, 0u, (uint8_t)axis};
                Box bl = it.box, br = it.box;
                bl.hi[axis] = n.split; br.lo[axis] = n.split;
                stack.push_back(Item{idx + 1, c + 1, br, it.depth + 1});
                stack.push_back(Item{idx, c, bl, it.depth + 1});
            }
        }
        QD_REFINE_MARK("top of the tree");
        // 2. one subtree per leaf, built on the host cores
        std::vector<SubTree> subs(jobs.size());
        unsigned n_thr = std::thread::hardware_concurrency();
        n_thr = std::max(1u, std::min(n_thr, 16u));
        if (jobs.size() < 4096 && !opt.rebuild) n_thr = 1;
        if (opt.rebuild && m.tri_count < 20000) n_thr = 1;
        const size_t grain = opt.rebuild ? 1 : 256;
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            Builder bld(d->tri_verts, m.tri_first, opt);
            for (;;) {
                const size_t b = next.fetch_add(grain), e = std::min(jobs.size(), b + grain);
                if (b >= jobs.size()) break;
                for (size_t j = b; j < e; ++j) bld.build(subs[j], jobs[j].box, jobs[j].tris, jobs[j].depth);
            }
        };
        if (n_thr == 1) worker();
        else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(worker);
            for (auto& t : pool) t.join();
        }
        QD_REFINE_MARK("subtrees");
        // stitch: subtree node 0 lands in the slot of the leaf it replaces, the rest is appended in job order (offsets by a
        // prefix sum, copies on the host cores)
        {
            std::vector<size_t> node_off(jobs.size()), ref_off(jobs.size());
            size_t nn = T.nodes.size(), nr = T.refs.size();
            for (size_t j = 0; j < jobs.size(); ++j) { node_off[j] = nn; nn += subs[j].nodes.size() - 1; ref_off[j] = nr; nr += subs[j].refs.size(); }
            T.nodes.resize(nn); T.refs.resize(nr);
            std::atomic<size_t> next_sub{0};
            std::atomic<uint64_t> leaves{0}, max_leaf{0};
            auto stitcher = [&]() {
                uint64_t my_leaves = 0, my_max = 0;
                for (size_t j; (j = next_sub.fetch_add(1)) < jobs.size();) {
                    const SubTree& sub = subs[j];
                    const size_t node_base = node_off[j] - 1;      // local index i >= 1 -> node_base + i
                    const size_t ref_base = ref_off[j];
                    auto fix = [&](BNode n) {
                        if (n.axis == 3) { n.child = (uint32_t)(ref_base + n.child); my_leaves++; my_max = std::max<uint64_t>(my_max, n.count); }
                        else n.child = (uint32_t)(node_base + n.child);
                        return n;
                    };
                    T.nodes[jobs[j].at] = fix(sub.nodes[0]);
                    for (size_t i = 1; i < sub.nodes.size(); ++i) T.nodes[node_base + i] = fix(sub.nodes[i]);
                    std::copy(sub.refs.begin(), sub.refs.end(), T.refs.begin() + ref_base);
                }
                leaves += my_leaves;
                uint64_t seen = max_leaf.load();
                while (seen < my_max && !max_leaf.compare_exchange_weak(seen, my_max)) {}
            };
            if (n_thr == 1) stitcher();
            else {
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(stitcher);
                for (auto& t : pool) t.join();
            }
            R.leaves += leaves.load(); R.max_leaf = std::max<uint64_t>(R.max_leaf, max_leaf.load());
        }
        R.n_nodes += T.nodes.size(); R.n_refs += T.refs.size();
        QD_REFINE_MARK("stitch");
    }
    return std::string();
}

// ------------------------------------------------------------------------------------------------------
// Device records, 32 bytes each, two tree levels per record:
//   inner record  { double s0, s1, s2; uint32 meta; uint32 0 }   meta = (base >> 2) << 6 | a2 << 4 | a1 << 2 | a0
//       a0, s0 : axis / plane of the node itself (a0 in 0..2)
//       a1, s1 : axis / plane of its left child,  a2, s2 : of its right child
//                axis 3 = that child is a leaf; the 8 bytes of its plane then hold { uint32 first, count }
//       base   : first of four consecutive records (one 128-byte line) for the grandchildren:
//                base + 2c + g = child g of inner child c
//   leaf record   { uint32 first, count; ...; uint32 meta = 3 }  (a grandchild, or a root, that is a leaf)
// first/count index the leaf reference arrays.  One 32-byte fetch therefore decides two levels of the walk, and a
// leaf that is the child of a record's node is opened without a fetch of its own.
struct Record { uint32_t w[8]; };

inline Record leaf_record(uint32_t first, uint32_t count) { return Record{{first, count, 0u, 0u, 0u, 0u, 3u, 0u}}; }

// Appends the records of one mesh tree to `out` (whose size is kept a multiple of 4); leaf references are
// shifted by ref_base.  Returns the record index of the root.
inline uint32_t pack(const MeshTree& T, std::vector<Record>& out, size_t ref_base) {
    auto put_f64 = [](Record& r, int at, double v) { std::memcpy(&r.w[at], &v, 8); };
    const size_t root = out.size();
    out.resize(root + 4, leaf_record(0, 0));
    struct Item { uint32_t node; size_t rec; };
    std::vector<Item> stack;
    stack.push_back(Item{0u, root});
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const BNode& n = T.nodes[it.node];
        if (n.axis == 3) { out[it.rec] = leaf_record((uint32_t)(ref_base + n.child), n.count); continue; }
        Record r{};
        put_f64(r, 0, n.split);
        uint32_t meta = n.axis;
        size_t base = 0;
        const bool any_inner = T.nodes[n.child].axis != 3 || T.nodes[n.child + 1].axis != 3;
        if (any_inner) { base = out.size(); out.resize(base + 4, leaf_record(0, 0)); }
        meta |= (uint32_t)((base >> 2) << 6);
        for (uint32_t c = 0; c < 2; ++c) {
            const BNode& ch = T.nodes[n.child + c];
            meta |= (uint32_t)ch.axis << (2 + 2 * c);
            if (ch.axis == 3) { r.w[2 + 2 * c] = (uint32_t)(ref_base + ch.child); r.w[3 + 2 * c] = ch.count; continue; }
            put_f64(r, 2 + 2 * c, ch.split);
            stack.push_back(Item{ch.child + 1, base + 2 * c + 1});
            stack.push_back(Item{ch.child, base + 2 * c});
        }
        r.w[6] = meta;
        out[it.rec] = r;
    }
    return (uint32_t)root;
}

}  // namespace qd_refine
