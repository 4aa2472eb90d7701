// Host-side build of the FLAT-mode acceleration structure (binned-SAH two-child tree).
//
// This structure has no counterpart in the reference: FLAT mode's result is defined as "nearest
// hit over all flattened nodes, lowest index on ties" (src/ray.rs:87-99 over
// src/flat_scene.rs:71-99), which is independent of how candidates are found, so the build is
// free to cull with any CONSERVATIVE bounds. Callers pad the boxes (see pt_api) so that every hit
// the reference's primitive tests can report lies strictly inside its box.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "pt_scene_view.h"

struct PtBuildBox {
    double lo[3], hi[3];
};

struct PtBvhRef {
    uint32_t child;  // packed reference (pt_scene_view.h)
    PtBuildBox box;
    int depth;      // levels below and including this reference
};

namespace pt_bvh_detail {

inline void grow(PtBuildBox& b, const PtBuildBox& o) {
    for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); }
}
inline PtBuildBox empty_box() {
    PtBuildBox b;
    for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
    return b;
}
inline double half_area(const PtBuildBox& b) {
    double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
    return dx * dy + dy * dz + dz * dx;
}

// f64 -> f32 rounded toward -inf / +inf (boxes may only grow), kept inside [-1e18, 1e18] and finite:
// the device slab test (pt_slab32) multiplies plane coordinates by reciprocals up to 1e18 and must
// not overflow. A coordinate beyond the limit or a NaN (degenerate transform) opens the box to the
// limit on that side, so such a node is simply always visited; rays are assumed to start within
// the same +-1e18.
#define PT_BOX_LIMIT 1e18
inline float round_down(double v) {
    if (!(v > -PT_BOX_LIMIT)) return -(float)PT_BOX_LIMIT;  // also NaN
    if (v > PT_BOX_LIMIT) return (float)PT_BOX_LIMIT;
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float round_up(double v) {
    if (!(v < PT_BOX_LIMIT)) return (float)PT_BOX_LIMIT;  // also NaN
    if (v < -PT_BOX_LIMIT) return -(float)PT_BOX_LIMIT;
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    return f;
}

struct Builder {
    const PtBuildBox* boxes;
    std::vector<uint32_t> order;  // permutation of local indices
    const uint32_t* ids;          // local index -> item id written to the items array
    int max_leaf;
    bool direct = false;  // one item per leaf, named by the leaf reference itself (no entry in the items array)
    std::vector<PtBvhNode>* nodes;
    std::vector<uint32_t>* items;

    PtBvhRef leaf(size_t b, size_t e, const PtBuildBox& box) {
        PtBvhRef r;
        if (direct) {
            r.child = PT_REF_LEAF | ((ids ? ids[order[b]] : order[b]) << 3);
            r.box = box;
            r.depth = 1;
            return r;
        }
        r.child = PT_REF_LEAF | ((uint32_t)items->size() << 3) | (uint32_t)(e - b - 1);
        for (size_t i = b; i < e; i++) items->push_back(ids ? ids[order[i]] : order[i]);
        r.box = box;
        r.depth = 1;
        return r;
    }

    PtBvhRef build(size_t b, size_t e, int level) {
        PtBuildBox box = empty_box(), cbox = empty_box();
        for (size_t i = b; i < e; i++) {
            const PtBuildBox& x = boxes[order[i]];
            grow(box, x);
            for (int k = 0; k < 3; k++) {
                double c = 0.5 * (x.lo[k] + x.hi[k]);
                cbox.lo[k] = std::min(cbox.lo[k], c); cbox.hi[k] = std::max(cbox.hi[k], c);
            }
        }
        size_t n = e - b;
        if (n <= (size_t)max_leaf) return leaf(b, e, box);
        // binned SAH over the three axes
        const int NB = 16;
        double best_cost = INFINITY; int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ax++) {
            double c0 = cbox.lo[ax], c1 = cbox.hi[ax];
            if (!(c1 > c0)) continue;
            double scale = NB / (c1 - c0);
            PtBuildBox bb[NB]; size_t cnt[NB];
            for (int k = 0; k < NB; k++) { bb[k] = empty_box(); cnt[k] = 0; }
            for (size_t i = b; i < e; i++) {
                const PtBuildBox& x = boxes[order[i]];
                int k = (int)((0.5 * (x.lo[ax] + x.hi[ax]) - c0) * scale);
                k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                grow(bb[k], x); cnt[k]++;
            }
            double right_area[NB]; size_t right_cnt[NB];
            PtBuildBox acc = empty_box(); size_t c = 0;
            for (int k = NB - 1; k > 0; k--) { grow(acc, bb[k]); c += cnt[k]; right_area[k] = half_area(acc); right_cnt[k] = c; }
            acc = empty_box(); c = 0;
            for (int k = 0; k < NB - 1; k++) {
                grow(acc, bb[k]); c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                double cost = half_area(acc) * (double)c + right_area[k + 1] * (double)right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
            }
        }
        size_t mid;
        if (best_axis < 0 || level > 32) {
            mid = b + n / 2;  // all centroids coincide (or a degenerate, very deep tree): split by position in the list
        } else {
            double c0 = cbox.lo[best_axis], scale = NB / (cbox.hi[best_axis] - c0);
            auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t i) {
                const PtBuildBox& x = boxes[i];
                int k = (int)((0.5 * (x.lo[best_axis] + x.hi[best_axis]) - c0) * scale);
                k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                return k <= best_bin;
            });
            mid = (size_t)(it - order.begin());
            if (mid == b || mid == e) mid = b + n / 2;
        }
        int32_t idx = (int32_t)nodes->size();
        nodes->push_back(PtBvhNode());
        PtBvhRef l = build(b, mid, level + 1);
        PtBvhRef r = build(mid, e, level + 1);
        PtBvhNode& nd = (*nodes)[idx];
        for (int k = 0; k < 3; k++) {
            nd.lo[k][0] = round_down(l.box.lo[k]); nd.hi[k][0] = round_up(l.box.hi[k]);
            nd.lo[k][1] = round_down(r.box.lo[k]); nd.hi[k][1] = round_up(r.box.hi[k]);
        }
        nd.child0 = l.child; nd.child1 = r.child; nd.pad[0] = nd.pad[1] = 0;
        PtBvhRef out;
        out.child = (uint32_t)idx; out.box = box; out.depth = 1 + std::max(l.depth, r.depth);
        return out;
    }
};

}  // namespace pt_bvh_detail

// Builds a tree over boxes[0..n) and appends its nodes / leaf items to the shared arrays. `ids`
// (optional) maps a local box index to the value stored in the items array.
// `direct` (needs max_leaf == 1): a leaf reference carries its item instead of an index into `items` - one dependent
// load less per leaf visit (the scene-level tree, whose leaves each hold one flattened node).
inline PtBvhRef pt_bvh_build(const PtBuildBox* boxes, const uint32_t* ids, size_t n, int max_leaf,
                             std::vector<PtBvhNode>& nodes, std::vector<uint32_t>& items, bool direct = false) {
    pt_bvh_detail::Builder b;
    b.boxes = boxes; b.ids = ids; b.max_leaf = std::max(1, std::min(max_leaf, 8)); b.nodes = &nodes; b.items = &items;
    b.direct = direct && b.max_leaf == 1 && n < (1u << 28);
    b.order.resize(n);
    for (size_t i = 0; i < n; i++) b.order[i] = (uint32_t)i;
    if (n == 0) {
        PtBvhRef r; r.child = PT_REF_EMPTY; r.box = pt_bvh_detail::empty_box(); r.depth = 1;
        return r;
    }
    return b.build(0, n, 0);
}
