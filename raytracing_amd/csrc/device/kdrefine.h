// Host-side refinement of the caller's KD tree, run once in qd_scene_create.
//
// The reference stops splitting at 20 triangles per leaf (TRIANGLES_PER_LEAF, src/constants.h:14,
// src/mesh.cpp:218) because a triangle test is cheap next to a recursive box-gated descent on a CPU.
// On the device the balance is the other way round: a triangle test is 72 bytes of gathered f64 data
// and ~100 f64 instructions, a descent step is one 16-byte load.  So every leaf of the caller's tree
// is split further with a surface-area heuristic until splitting stops paying.  The closest hit of a
// ray does not depend on how the triangles are grouped (SURVEY.md section 8(f).3 asks for exactly
// this), only an exact distance tie between two different triangles can resolve differently.
//
// Classification of a triangle against a split plane follows the reference's PartitionTriangle
// (src/mesh.cpp:412-432) on the triangle's bounds clipped to the node box: strictly below the plane
// on some vertex -> left, strictly above -> right, lying in the plane -> left.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "qdtrace.h"

namespace qd_refine {

struct Box { double lo[3], hi[3]; };

struct Options {
    int max_leaf = 2;          // leaves with at most this many triangles are never split
    double cost_step = 0.6;    // cost of one descent step, in triangle tests
    int max_depth = 70;        // below the device traversal stack (kKdStack)
};

inline double half_area(const Box& b) {
    const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

struct TriBounds { double lo[3], hi[3]; };

// subtree under one of the caller's leaves; node 0 is the subtree root, children of node i are at
// nodes[i].meta >> 2 and +1 (local indices), leaves index into refs (local)
struct SubTree {
    std::vector<qd_kdnode> nodes;
    std::vector<uint32_t> refs;
};

class Builder {
public:
    Builder(const double* tri_verts, uint32_t tri_first, const Options& opt) : tv_(tri_verts), first_(tri_first), opt_(opt) {}

    void build(SubTree& out, const Box& box, const uint32_t* tris, uint32_t n, int depth) {
        out.nodes.clear(); out.refs.clear();
        out.nodes.push_back(qd_kdnode{0., 3u, 0u});
        std::vector<uint32_t> list(tris, tris + n);
        split(out, 0, box, list, depth);
    }

private:
    const double* tv_;
    uint32_t first_;
    Options opt_;

    TriBounds bounds(uint32_t t, const Box& box) const {
        const double* v = tv_ + (size_t)(first_ + t) * 9;      // A, AB, AC
        TriBounds b;
        for (int a = 0; a < 3; ++a) {
            const double p0 = v[a], p1 = v[a] + v[3 + a], p2 = v[a] + v[6 + a];
            b.lo[a] = std::max(std::min(p0, std::min(p1, p2)), box.lo[a]);
            b.hi[a] = std::min(std::max(p0, std::max(p1, p2)), box.hi[a]);
            if (b.hi[a] < b.lo[a]) b.hi[a] = b.lo[a];          // bounds miss the box by rounding: keep a point
        }
        return b;
    }

    void make_leaf(SubTree& out, size_t at, const std::vector<uint32_t>& list) {
        out.nodes[at] = qd_kdnode{0., ((uint32_t)out.refs.size() << 2) | 3u, (uint32_t)list.size()};
        out.refs.insert(out.refs.end(), list.begin(), list.end());
    }

    void split(SubTree& out, size_t at, const Box& box, const std::vector<uint32_t>& list, int depth) {
        const uint32_t n = (uint32_t)list.size();
        const double area = half_area(box);
        if ((int)n <= opt_.max_leaf || depth >= opt_.max_depth || !(area > 0.)) { make_leaf(out, at, list); return; }
        std::vector<TriBounds> tb(n);
        for (uint32_t i = 0; i < n; ++i) tb[i] = bounds(list[i], box);
        double best_cost = (double)n;          // cost of staying a leaf
        int best_axis = -1;
        double best_split = 0.;
        std::vector<double> los(n), his(n), flat, cand;
        for (int a = 0; a < 3; ++a) {
            if (!(box.hi[a] > box.lo[a])) continue;
            flat.clear(); cand.clear();
            for (uint32_t i = 0; i < n; ++i) {
                los[i] = tb[i].lo[a]; his[i] = tb[i].hi[a];
                if (los[i] == his[i]) flat.push_back(los[i]);
                cand.push_back(los[i]); cand.push_back(his[i]);
            }
            std::sort(los.begin(), los.end()); std::sort(his.begin(), his.end());
            std::sort(flat.begin(), flat.end()); std::sort(cand.begin(), cand.end());
            cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
            const int u = (a + 1) % 3, v = (a + 2) % 3;
            const double du = box.hi[u] - box.lo[u], dv = box.hi[v] - box.lo[v];
            for (double s : cand) {
                if (!(s > box.lo[a] && s < box.hi[a])) continue;
                const uint32_t below = (uint32_t)(std::lower_bound(los.begin(), los.end(), s) - los.begin());
                const uint32_t in_plane = (uint32_t)(std::upper_bound(flat.begin(), flat.end(), s) - std::lower_bound(flat.begin(), flat.end(), s));
                const uint32_t nl = below + in_plane;
                const uint32_t nr = n - (uint32_t)(std::upper_bound(his.begin(), his.end(), s) - his.begin());
                const double dl = s - box.lo[a], dr = box.hi[a] - s;
                const double al = du * dv + (du + dv) * dl, ar = du * dv + (du + dv) * dr;
                double cost = opt_.cost_step + (al * nl + ar * nr) / area;
                if (nl == 0 || nr == 0) cost *= 0.8;           // cutting off empty space
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = s; }
            }
        }
        if (best_axis < 0) { make_leaf(out, at, list); return; }
        std::vector<uint32_t> left, right;
        for (uint32_t i = 0; i < n; ++i) {
            const double lo = tb[i].lo[best_axis], hi = tb[i].hi[best_axis];
            if (lo < best_split || (lo == best_split && hi == best_split)) left.push_back(list[i]);
            if (hi > best_split) right.push_back(list[i]);
        }
        if (left.size() == n && right.size() == n) { make_leaf(out, at, list); return; }
        const size_t c = out.nodes.size();
        out.nodes.push_back(qd_kdnode{0., 3u, 0u});
        out.nodes.push_back(qd_kdnode{0., 3u, 0u});
        out.nodes[at] = qd_kdnode{best_split, ((uint32_t)c << 2) | (uint32_t)best_axis, 0u};
        Box bl = box, br = box;
        bl.hi[best_axis] = best_split; br.lo[best_axis] = best_split;
        split(out, c, bl, left, depth + 1);
        split(out, c + 1, br, right, depth + 1);
    }
};

struct Result {
    std::vector<qd_kdnode> nodes;      // flat, children adjacent (index, index + 1), per-mesh roots in `roots`
    std::vector<uint32_t> refs;        // mesh-local triangle indices
    std::vector<int32_t> roots;        // per mesh, -1 = no tree
    uint64_t leaves = 0, max_leaf = 0;
    uint32_t max_depth = 0;
};

// Copies every mesh's tree and re-splits its leaves.  `opt.max_leaf <= 0` copies the trees unchanged.
inline void refine(const qd_scene_desc* d, const Options& opt, Result& R) {
    R.nodes.clear(); R.refs.clear(); R.roots.assign(d->n_meshes, -1);
    struct Job { size_t at; Box box; uint32_t first, count; int depth; uint32_t mesh; };
    std::vector<Job> jobs;
    for (uint32_t mi = 0; mi < d->n_meshes; ++mi) {
        const qd_mesh& m = d->meshes[mi];
        if (m.kd_root < 0) continue;
        Box root;
        for (int a = 0; a < 3; ++a) { root.lo[a] = m.bbox_min[a]; root.hi[a] = m.bbox_max[a]; }
        struct Item { uint32_t src; size_t dst; Box box; int depth; };
        std::vector<Item> stack;
        R.roots[mi] = (int32_t)R.nodes.size();
        R.nodes.push_back(qd_kdnode{0., 3u, 0u});
        stack.push_back(Item{(uint32_t)m.kd_root, R.nodes.size() - 1, root, 0});
        while (!stack.empty()) {
            const Item it = stack.back(); stack.pop_back();
            const qd_kdnode n = d->kd_nodes[it.src];
            const uint32_t axis = n.meta & 3u, idx = n.meta >> 2;
            if (axis == 3u) { jobs.push_back(Job{it.dst, it.box, idx, n.count, it.depth, mi}); continue; }
            const size_t c = R.nodes.size();
            R.nodes.push_back(qd_kdnode{0., 3u, 0u});
            R.nodes.push_back(qd_kdnode{0., 3u, 0u});
            R.nodes[it.dst] = qd_kdnode{n.split, ((uint32_t)c << 2) | axis, 0u};
            Box bl = it.box, br = it.box;
            bl.hi[axis] = n.split; br.lo[axis] = n.split;
            stack.push_back(Item{idx + 1, c + 1, br, it.depth + 1});
            stack.push_back(Item{idx, c, bl, it.depth + 1});
        }
    }
    // one subtree per leaf of the caller's tree, built on all host cores
    std::vector<SubTree> subs(jobs.size());
    const bool on = opt.max_leaf > 0;
    unsigned n_thr = std::thread::hardware_concurrency();
    n_thr = std::max(1u, std::min(n_thr, 16u));
    if (jobs.size() < 4096) n_thr = 1;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(256), e = std::min(jobs.size(), b + 256);
            if (b >= jobs.size()) break;
            for (size_t j = b; j < e; ++j) {
                const Job& jb = jobs[j];
                const qd_mesh& m = d->meshes[jb.mesh];
                if (on) {
                    Builder bld(d->tri_verts, m.tri_first, opt);
                    bld.build(subs[j], jb.box, d->leaf_tris + jb.first, jb.count, jb.depth);
                } else {
                    subs[j].nodes.assign(1, qd_kdnode{0., 3u, jb.count});
                    subs[j].refs.assign(d->leaf_tris + jb.first, d->leaf_tris + jb.first + jb.count);
                }
            }
        }
    };
    if (n_thr == 1) worker();
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    // stitch: subtree node 0 lands in the slot of the leaf it replaces, the rest is appended
    size_t n_nodes = R.nodes.size(), n_refs = 0;
    for (const SubTree& s : subs) { n_nodes += s.nodes.size() - 1; n_refs += s.refs.size(); }
    R.nodes.reserve(n_nodes); R.refs.reserve(n_refs);
    for (size_t j = 0; j < jobs.size(); ++j) {
        const SubTree& s = subs[j];
        const size_t node_base = R.nodes.size() - 1;      // local index i >= 1 -> node_base + i
        const size_t ref_base = R.refs.size();
        auto fix = [&](qd_kdnode n) {
            const uint32_t axis = n.meta & 3u, idx = n.meta >> 2;
            if (axis == 3u) {
                n.meta = ((uint32_t)(ref_base + idx) << 2) | 3u;
                R.leaves++; R.max_leaf = std::max<uint64_t>(R.max_leaf, n.count);
            } else n.meta = ((uint32_t)(node_base + idx) << 2) | axis;
            return n;
        };
        R.nodes[jobs[j].at] = fix(s.nodes[0]);
        for (size_t i = 1; i < s.nodes.size(); ++i) R.nodes.push_back(fix(s.nodes[i]));
        R.refs.insert(R.refs.end(), s.refs.begin(), s.refs.end());
    }
}

}  // namespace qd_refine
