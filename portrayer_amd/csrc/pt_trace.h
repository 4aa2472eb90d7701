// Nearest-hit / any-hit traversal: replaces `scene.root.ray_cast(ray, &mut t_range)`.
//
// FLAT mode reproduces the reference's flat_scene semantics (ray.rs:87-99 fold over
// Vec<FlatSceneNode>, flat_scene.rs:71-99): every candidate is tested in ITS OWN model space over
// [EPSILON, t_best) and the nearest wins, the lowest flat index winning exact ties. That result
// does not depend on the order candidates are visited in, so the kernels walk a bounding-volume
// tree built by pt_scene_upload instead of scanning all nodes, and break exact ties by index.
//
// KD mode reproduces the reference's `kdtree` feature bit for bit (kdtree/node.rs:66-203): the
// reference's own tree (built on the host like kdtree/leaf.rs:89-231), front-to-back with the
// range clipped at every straddled plane, because Cone/Cylinder results depend on the clipped
// range start (quirk Q1, cone.rs:64-76).
//
// The traversal stack lives in LDS: one column of 32-bit words per lane (word i of lane l at
// base[i * stride + l]) so that a wave's pushes and pops hit 64 different banks.
#pragma once

#include "pt_prims.h"
#include "pt_scene_view.h"

#define PT_NO_HIT 0xFFFFFFFFu
#ifndef PT_WALK_POPS_MAX
#define PT_WALK_POPS_MAX (1u << 27)  // watchdog of the wave-uniform walks: more pending subtrees than the largest tree (2^26 nodes) has
#endif
#define PT_BLOCK 256              // threads per block of every traversal kernel = stride of the per-lane LDS columns
#define PT_FRAME_STRIDE PT_BLOCK

// -DPT_DIAG (profiles/diag.sh): one count per WAVEFRONT pass through a place, next to the per-lane counts the
// STATS build keeps anyway; their ratio is the lane occupancy of that place.
#ifdef PT_DIAG
#define PT_WAVE_COUNT(k) do { if (STATS && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) cnt->diag[k]++; } while (0)
#define PT_LANE_COUNT(k) do { if (STATS) cnt->diag[k]++; } while (0)
#else
#define PT_WAVE_COUNT(k) do { } while (0)
#define PT_LANE_COUNT(k) do { } while (0)
#endif
// -DPT_CYCLES (profiles/cycles.sh): shader cycles a wavefront spends in a section, summed over wavefronts into diag[k]
// (only lane 0 of a wave adds; s_memtime itself costs ~10 % of wave cycles, so read the split, not the totals)
#ifdef PT_CYCLES
#define PT_CYC_BEGIN() const unsigned long long pt_cyc0_ = __builtin_readcyclecounter()
#define PT_CYC_END(k) do { if (STATS && (__lane_id() == 0 || __ffsll((long long)__ballot(1)) - 1 == (int)__lane_id())) cnt->diag[k] += __builtin_readcyclecounter() - pt_cyc0_; } while (0)
#else
#define PT_CYC_BEGIN() do { } while (0)
#define PT_CYC_END(k) do { } while (0)
#endif

struct PtHit {
    double t;       // ray parameter of the best hit so far; doubles as the exclusive range end
    uint32_t node;  // flat node index, PT_NO_HIT when nothing was hit
    uint32_t sub;   // analytic: part tag; mesh / triangle: global triangle index
};

// Per-lane traversal stack: word i of the lane at base[i * PT_BLOCK] (a column of the block's LDS area: a wave's
// pushes and pops hit 64 different banks). PtStack is LDS only (pt_cast_kernel); PtStackSpill keeps the first `cap`
// entries in LDS and the rest - only reached by unusually deep excursions - in HBM, so that the LDS part can be
// sized for the occupancy instead of for the worst case of the deepest tree.
struct PtStack {
    uint32_t* base;
    int cap;
    unsigned int* overflow;  // device word set to 1 when a lane runs out of stack (PT_ERR_TRAVERSAL), in every build
};
struct PtStackSpill {
    uint32_t* base;
    int cap;                 // entries in LDS
    int total;               // entries in all
    uint32_t* gbase;         // entry cap + i of the lane at gbase[i * gstride]
    uint32_t gstride;
    unsigned int* overflow;
};
PT_HD int pt_stack_total(const PtStack& s) { return s.cap; }
PT_HD int pt_stack_total(const PtStackSpill& s) { return s.total; }
// Never expected (pt_scene_upload sizes the stack for the deepest walk); recorded unconditionally so that a render
// whose results would be wrong cannot return PT_OK.
template <class Stack>
PT_HD void pt_stack_overflow(const Stack& s) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (s.overflow) atomicOr(s.overflow, 1u);
#else
    (void)s;
#endif
}

PT_HD void pt_push(const PtStack& s, int& sp, uint32_t v) { s.base[sp * PT_BLOCK] = v; sp++; }
PT_HD uint32_t pt_pop(const PtStack& s, int& sp) { sp--; return s.base[sp * PT_BLOCK]; }
// The HBM part is behind a WAVE-UNIFORM branch (one ballot, one scalar branch): written as a per-lane select the compiler
// turned every pop into a flat load of a computed LDS-or-global address plus eight masked-off address instructions.
PT_HD bool pt_any_lane_beyond(int sp, int cap) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(sp >= cap) != 0ull;
#else
    return sp >= cap;
#endif
}
PT_HD void pt_push(const PtStackSpill& s, int& sp, uint32_t v) {
    if (!pt_any_lane_beyond(sp, s.cap)) s.base[sp * PT_BLOCK] = v;
    else if (sp < s.cap) s.base[sp * PT_BLOCK] = v;
    else s.gbase[(size_t)(sp - s.cap) * s.gstride] = v;
    sp++;
}
PT_HD uint32_t pt_pop(const PtStackSpill& s, int& sp) {
    sp--;
    if (!pt_any_lane_beyond(sp, s.cap)) return s.base[sp * PT_BLOCK];
    if (sp < s.cap) return s.base[sp * PT_BLOCK];
    return s.gbase[(size_t)(sp - s.cap) * s.gstride];
}
template <class Stack>
PT_HD void pt_push_f64(const Stack& s, int& sp, double v) {
    union { double d; uint32_t u[2]; } c; c.d = v;
    pt_push(s, sp, c.u[0]); pt_push(s, sp, c.u[1]);
}
template <class Stack>
PT_HD double pt_pop_f64(const Stack& s, int& sp) {
    union { double d; uint32_t u[2]; } c;
    c.u[1] = pt_pop(s, sp); c.u[0] = pt_pop(s, sp);
    return c.d;
}

// Single-precision ray for the tree walk. The slab test evaluates (plane - o) / d as
// fma(plane, inv, -o * inv) with the per-ray constants rounded so that every error makes the overlap
// LONGER: inv = fl32(1 / d); the lower planes use n0 = -(o_hi * inv), the upper planes n1 = -(o_lo * inv)
// (o_hi >= o >= o_lo bound the f64 origin in f32), each then pushed by 2^-22 relative in the direction
// that moves a near plane nearer and a far plane farther (which plane is near depends on the sign of
// inv). An axis the ray is parallel to (|inv| > 1e18, which also keeps every product below FLT_MAX
// for the |coordinates| <= 1e18 that pt_scene_upload guarantees for boxes) is switched off:
// inv = 0, n0 = -inf, n1 = +inf give the interval (-inf, +inf).
struct PtRay32 {
    float ix, iy, iz, n0x, n0y, n0z, n1x, n1y, n1z;
};
PT_HD void pt_ray32_axis(double o, double d, float* inv, float* n0, float* n1) {
    float x = (float)o;
    const float e = 2.4e-7f;  // > 2^-23: covers the rounding of the conversion and of the bound itself
    float ex = fabsf(x) * e + 1e-37f;
    float oh = x + ex, ol = x - ex;
    float i = (float)(1.0 / d);
    if (!(fabsf(i) <= 1e18f)) {  // parallel (or NaN): no constraint from this axis
        *inv = 0.0f; *n0 = -INFINITY; *n1 = INFINITY;
        return;
    }
    float c0 = oh * i, c1 = ol * i;
    float w0 = fabsf(c0) * 4.8e-7f + 1e-37f, w1 = fabsf(c1) * 4.8e-7f + 1e-37f;
    // i > 0: the lower plane is the near one (make it nearer: larger c0), the upper plane the far one (smaller c1); i < 0: the other way round
    c0 = i > 0.0f ? c0 + w0 : c0 - w0;
    c1 = i > 0.0f ? c1 - w1 : c1 + w1;
    *inv = i; *n0 = -c0; *n1 = -c1;
}
PT_HD PtRay32 pt_ray32(const PtRay& r) {
    PtRay32 q;
    pt_ray32_axis(r.o.x, r.d.x, &q.ix, &q.n0x, &q.n1x);
    pt_ray32_axis(r.o.y, r.d.y, &q.iy, &q.n0y, &q.n1y);
    pt_ray32_axis(r.o.z, r.d.z, &q.iz, &q.n0z, &q.n1z);
    return q;
}

// ---- The same idea with the margins folded into the per-ray constants (round 3; the derivation is with pt_slab_pk2 below): per axis
// a pair (i, c) for a box's LOWER plane and one for its UPPER plane, chosen by the direction's sign so that min / max of the two
// fmas are conservative entering / leaving parameters without any widening afterwards.
typedef float pt_f32x2 __attribute__((ext_vector_type(2)));
struct PtRayPk {
    pt_f32x2 a[3], b[3];  // per axis (i, c) for the children's lower planes / upper planes
};
PT_HD float pt_rcp_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// returns 0: axis switched off, 1: positive direction, 2: negative direction
PT_HD int pt_raypk_axis(double o, double d, pt_f32x2* a, pt_f32x2* b) {
    const float i0 = pt_rcp_f32((float)d);
    if (!(fabsf(i0) <= 1e18f)) {
        a->x = 0.0f; a->y = -INFINITY; b->x = 0.0f; b->y = INFINITY;
        return 0;
    }
    const float k = 4.76837158203125e-7f;  // 2^-21
    const float in = i0 - i0 * k, fi = i0 + i0 * k;
    float cn = (float)(-(o * (double)in)), cf = (float)(-(o * (double)fi));
    cn = cn - (fabsf(cn) * 2.4e-7f + 1e-37f);
    cf = cf + (fabsf(cf) * 2.4e-7f + 1e-37f);
    const bool neg = i0 < 0.0f;  // entering through the upper plane
    a->x = neg ? fi : in; a->y = neg ? cf : cn;
    b->x = neg ? in : fi; b->y = neg ? cn : cf;
    return neg ? 2 : 1;
}
PT_HD PtRayPk pt_raypk(const PtRay& r) {  // without the wavefront's view: for walks that always take the per-lane form
    PtRayPk q;
    pt_raypk_axis(r.o.x, r.d.x, &q.a[0], &q.b[0]);
    pt_raypk_axis(r.o.y, r.d.y, &q.a[1], &q.b[1]);
    pt_raypk_axis(r.o.z, r.d.z, &q.a[2], &q.b[2]);
    return q;
}
// The per-lane form of the slab test with those constants, against a segment [tmin, tmax] of the ray (both already rounded outward
// by the caller): the conservative culls of the k-d walks. 6 fmas + 6 min / max + 4 + 1 compare.
PT_HD bool pt_slab_seg_pk(const float* lo, const float* hi, const PtRayPk& q, float tmin, float tmax) {
    const float ax = __builtin_fmaf(lo[0], q.a[0].x, q.a[0].y), bx = __builtin_fmaf(hi[0], q.b[0].x, q.b[0].y);
    const float ay = __builtin_fmaf(lo[1], q.a[1].x, q.a[1].y), by = __builtin_fmaf(hi[1], q.b[1].x, q.b[1].y);
    const float az = __builtin_fmaf(lo[2], q.a[2].x, q.a[2].y), bz = __builtin_fmaf(hi[2], q.b[2].x, q.b[2].y);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return !(tn > tf);
}

// Slab test of one child box against [0, tmax] in f32. Conservative by construction (see PtRay32);
// the final interval is widened by 2^-20 relative (fma + reciprocal roundings are < 2^-22), NaNs
// cannot arise from finite boxes, and the comparison is written so that an unordered result accepts
// the box anyway.
PT_HD bool pt_slab32(const float* lo, const float* hi, const PtRay32& q, float tmax, float* tnear) {
    float x0 = __builtin_fmaf(lo[0], q.ix, q.n0x), x1 = __builtin_fmaf(hi[0], q.ix, q.n1x);
    float y0 = __builtin_fmaf(lo[1], q.iy, q.n0y), y1 = __builtin_fmaf(hi[1], q.iy, q.n1y);
    float z0 = __builtin_fmaf(lo[2], q.iz, q.n0z), z1 = __builtin_fmaf(hi[2], q.iz, q.n1z);
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float w = 9.6e-7f;
    tn = fmaxf(__builtin_fmaf(fabsf(tn), -w, tn), 0.0f);
    tf = fminf(__builtin_fmaf(fabsf(tf), w, tf), tmax);
    *tnear = tn;
    return !(tn > tf);
}

// The same test against a segment [tmin, tmax] of the ray (both already rounded outward by the caller).
PT_HD bool pt_slab32_segment(const float* lo, const float* hi, const PtRay32& q, float tmin, float tmax) {
    float x0 = __builtin_fmaf(lo[0], q.ix, q.n0x), x1 = __builtin_fmaf(hi[0], q.ix, q.n1x);
    float y0 = __builtin_fmaf(lo[1], q.iy, q.n0y), y1 = __builtin_fmaf(hi[1], q.iy, q.n1y);
    float z0 = __builtin_fmaf(lo[2], q.iz, q.n0z), z1 = __builtin_fmaf(hi[2], q.iz, q.n1z);
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float w = 9.6e-7f;
    tn = fmaxf(__builtin_fmaf(fabsf(tn), -w, tn), tmin);
    tf = fminf(__builtin_fmaf(fabsf(tf), w, tf), tmax);
    return !(tn > tf);
}

// The four child boxes of a PtBvh4Node against [0, tmax]: pt_slab32 four times over the node's [axis][child] arrays.
// tn[k] = entry distance of child k, +inf where the box is missed or the slot is unused.
PT_HD void pt_slab32_x4(const PtBvh4Node& n, const PtRay32& q, float tmax, float tn[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float x0 = __builtin_fmaf(n.lo[0][k], q.ix, q.n0x), x1 = __builtin_fmaf(n.hi[0][k], q.ix, q.n1x);
        float y0 = __builtin_fmaf(n.lo[1][k], q.iy, q.n0y), y1 = __builtin_fmaf(n.hi[1][k], q.iy, q.n1y);
        float z0 = __builtin_fmaf(n.lo[2][k], q.iz, q.n0z), z1 = __builtin_fmaf(n.hi[2][k], q.iz, q.n1z);
        float a = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        const float w = 9.6e-7f;
        a = fmaxf(__builtin_fmaf(fabsf(a), -w, a), 0.0f);
        f = fminf(__builtin_fmaf(fabsf(f), w, f), tmax);
        tn[k] = (!(a > f) && n.child[k] != PT_REF_EMPTY) ? a : INFINITY;
    }
}
// Compare-exchange of (distance, reference) pairs: after the five of pt_sort4 the hit children are in order of entry
// distance, misses (+inf) last. Order only decides how soon `tmax` shrinks, never a result.
#define PT_CSWAP(i, j) do { const bool sw_ = t[j] < t[i]; const float tt_ = sw_ ? t[j] : t[i]; t[j] = sw_ ? t[i] : t[j]; t[i] = tt_; \
                            const uint32_t cc_ = sw_ ? c[j] : c[i]; c[j] = sw_ ? c[i] : c[j]; c[i] = cc_; } while (0)
PT_HD void pt_sort4(float t[4], uint32_t c[4]) {
#if defined(PT_BVH4_ORDER) && PT_BVH4_ORDER == 2 // experiment: first and last in place (four exchanges), the middle pair as it falls
    PT_CSWAP(0, 1); PT_CSWAP(2, 3); PT_CSWAP(0, 2); PT_CSWAP(1, 3);
#else
    PT_CSWAP(0, 1); PT_CSWAP(2, 3); PT_CSWAP(0, 2); PT_CSWAP(1, 3); PT_CSWAP(1, 2);
#endif
}

// Generic walk of the build's two-child tree, "while-while": every lane first descends through inner
// nodes until it holds a leaf (or has nothing left), and only then do the lanes test their leaves
// together — the leaf work (f64 primitive tests) is the expensive part and should run with as many
// lanes active as possible. leaf(first, count, sp) tests the items and returns true to stop the walk
// (any-hit). `tmax` is re-read after every leaf so shrinking it culls.
template <bool STATS, class Stack, class Leaf>
PT_HD bool pt_bvh_walk(const PtBvh4Node* nodes, uint32_t root, const PtRay& r, const double& tmax,
                       const Stack& stk, int sp0, Leaf&& leaf, PtCounters* cnt) {
    if (root == PT_REF_EMPTY) return false;
    int sp = sp0;
    const PtRay32 q = pt_ray32(r);
    uint32_t cur = root;
    for (;;) {
        while (!(cur & PT_REF_LEAF)) {
            const PtBvh4Node& n = nodes[cur];
            if (STATS) cnt->n_inner++;
            PT_WAVE_COUNT(4);
            // tmax rounded up: (float) rounds to nearest, one more relative step covers it (inf stays inf)
            float tm = (float)tmax; tm = tm + fabsf(tm) * 2.4e-7f;
            float t[4];
            uint32_t c[4] = {n.child[0], n.child[1], n.child[2], n.child[3]};
            pt_slab32_x4(n, q, tm, t);
            pt_sort4(t, c);
            if (!(t[0] < INFINITY)) {
                if (sp == sp0) return false;
                cur = pt_pop(stk, sp);
                continue;
            }
            const int hits = (t[1] < INFINITY) + (t[2] < INFINITY) + (t[3] < INFINITY);
            if (sp + hits > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; return false; }
            if (t[3] < INFINITY) pt_push(stk, sp, c[3]);
            if (t[2] < INFINITY) pt_push(stk, sp, c[2]);
            if (t[1] < INFINITY) pt_push(stk, sp, c[1]);
            cur = c[0];
        }
        if (STATS) cnt->n_leaf++;
        PT_WAVE_COUNT(5);
        if (leaf((cur & ~PT_REF_LEAF) >> 3, (cur & 7u) + 1u, sp)) return true;
        if (sp == sp0) return false;
        cur = pt_pop(stk, sp);
    }
}

// The same walk over the two-child form (mesh-free scenes: see PtBvhWalker::step).
template <bool STATS, class Stack, class Leaf>
PT_HD bool pt_bvh2_walk(const PtBvhNode* nodes, uint32_t root, const PtRay& r, const double& tmax,
                       const Stack& stk, int sp0, Leaf&& leaf, PtCounters* cnt) {
    if (root == PT_REF_EMPTY) return false;
    int sp = sp0;
    const PtRay32 q = pt_ray32(r);
    uint32_t cur = root;
    for (;;) {
        while (!(cur & PT_REF_LEAF)) {
            const PtBvhNode& n = nodes[cur];
            if (STATS) cnt->n_inner++;
            PT_WAVE_COUNT(4);
            // tmax rounded up: (float) rounds to nearest, one more relative step covers it (inf stays inf)
            float tm = (float)tmax; tm = tm + fabsf(tm) * 2.4e-7f;
            float t0, t1, lo0[3], hi0[3], lo1[3], hi1[3];
            pt_node_get_box(n, 0, lo0, hi0); pt_node_get_box(n, 1, lo1, hi1);
            bool h0 = pt_slab32(lo0, hi0, q, tm, &t0);
            bool h1 = pt_slab32(lo1, hi1, q, tm, &t1);
            uint32_t c0 = n.child0, c1 = n.child1;
            if (h0 && h1) {
                bool swap = t1 < t0;
                if (sp + 1 > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; return false; }
                pt_push(stk, sp, swap ? c0 : c1);
                cur = swap ? c1 : c0;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else {
                if (sp == sp0) return false;
                cur = pt_pop(stk, sp);
            }
        }
        if (STATS) cnt->n_leaf++;
        PT_WAVE_COUNT(5);
        { PT_CYC_BEGIN(); const bool stop_ = leaf((cur & ~PT_REF_LEAF) >> 3, (cur & 7u) + 1u, sp); PT_CYC_END(5); if (stop_) return true; }
        if (sp == sp0) return false;
        cur = pt_pop(stk, sp);
    }
}

// Exclusive range end for a candidate: t == best.t is admitted only when the candidate comes
// before the current winner in flat order (ray.rs:87-99: the first of equal hits wins).
PT_HD double pt_cand_end(const PtHit& best, uint32_t node, uint32_t sub) {
    bool before = best.node != PT_NO_HIT && (node < best.node || (node == best.node && sub < best.sub));
    return before ? pt_next_up(best.t) : best.t;
}

// The same for the hierarchical traversal: between different nodes the one that comes first depth-first wins
// (scene.rs:95-117: own geometry, then the children in order, each a half-open shrinking range).
template <bool HIER>
PT_HD double pt_cand_end_in(const PtSceneView& sc, const PtHit& best, uint32_t node, uint32_t sub) {
    if (!HIER) return pt_cand_end(best, node, sub);
    bool before = best.node != PT_NO_HIT && (node == best.node ? sub < best.sub : sc.dfs_rank[node] < sc.dfs_rank[best.node]);
    return before ? pt_next_up(best.t) : best.t;
}

// World-space ray -> the node's model space. FLAT: the flattened node's composed inverse (flat_scene.rs:74).
// HIER: one SceneNode at a time down the path, each with its own inverse (scene.rs:82) - the same transform in
// exact arithmetic, different roundings.
template <bool HIER>
PT_HD PtRay pt_node_local_ray(const PtSceneView& sc, uint32_t node, const PtRay& ray) {
    if (!HIER) return pt_ray_to_local(sc.inv + 12 * (size_t)node, ray);
    PtRay r = ray;
    const uint32_t* rec = sc.hier_rec + 8 * (size_t)node;  // the node's path in one 32-byte line (pt_api.hip) instead of chain_off -> chain
    const uint32_t len = rec[0] & 255u;
    if (len == 255u) {  // more than seven levels
        for (uint32_t k = sc.chain_off[node]; k < sc.chain_off[node + 1]; k++) r = pt_ray_to_local(sc.g_inv + 12 * (size_t)sc.chain[k], r);
        return r;
    }
    for (uint32_t k = 0; k < len; k++) r = pt_ray_to_local(sc.g_inv + 12 * (size_t)rec[1 + k], r);
    return r;
}

PT_HD double pt_axis(PtVec3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// KDMesh::ray_hit (kdtree/kdmesh.rs:62-74): box test on the tree's root bounds, then the mesh's OWN
// k-d tree of triangles walked exactly like the scene tree (node.rs:33-51 + :66-203): front to back,
// range clipped at straddled planes, the first leaf with a hit wins, and — quirk Q3 — the segment used
// to classify sides ends at start + extent (squared diagonal), so hits farther away than that can be
// missed. Reproduced because results differ from Mesh whenever the quirk bites.
template <bool STATS, class Stack>
PT_HD bool pt_kdmesh_hit(const PtSceneView& sc, const PtMeshInfo& m, const PtRay& local, double start, double end, const Stack& stk,
                         int sp0, double* t_out, uint32_t* tri_out, PtCounters* cnt) {
    if (STATS) cnt->n_bbox++;
    if (!pt_bbox_test_hit(m.kd_bbox_inv, local, start, end)) return false;
    int sp = sp0;
    int32_t cur = m.kd_root;
    const double extent = m.kd_extent;
    const double end0 = end;  // a pending far side's range end = the start of the entry below it, or end0
    const PtRayPk q = pt_raypk(local);
    for (;;) {
        const PtKdNode n = sc.mkd[cur];
        // conservative culls as in pt_trace_kd: a subtree / a triangle whose box the segment [start, end) does not reach reports no hit
        float seg0 = (float)start, seg1 = (float)end;
        seg0 = seg0 - fabsf(seg0) * 2.4e-7f; seg1 = seg1 + fabsf(seg1) * 2.4e-7f;
        if (sc.mkd_box && !pt_slab_seg_pk(n.box, n.box + 3, q, seg0, seg1)) {
            if (STATS) cnt->kd_culled++;
        } else if (n.axis < 0) {  // leaf: [T]::ray_hit (ray.rs:50-63) over the leaf's triangles, in order, strict shrinking end
            if (STATS) cnt->n_leaf++;
            bool found = false;
            double e = end;
            for (int32_t i = 0; i < n.count; i++) {
                uint32_t tri = sc.mkd_items[n.first + i];
                if (sc.mkd_item_box) {
                    const float* ib = sc.mkd_item_box + 6 * (size_t)(n.first + i);
                    if (!pt_slab_seg_pk(ib, ib + 3, q, seg0, seg1)) continue;
                }
                double tt, beta, gamma;
                if (STATS) cnt->n_tri++;
                if (pt_triangle_hit(sc.tri_v + 9 * (size_t)tri, local, start, e, &tt, &beta, &gamma)) { e = tt; *t_out = tt; *tri_out = tri; found = true; }
            }
            if (found) return true;
        } else {
            if (STATS) cnt->n_inner++;
            double t_max = start + extent;
            if (!pt_in_range(start, end, t_max)) t_max = end - PT_EPSILON;
            double t_min = start + PT_EPSILON;
            double o = pt_axis(local.o, n.axis), d = pt_axis(local.d, n.axis);
            bool s = ((o + d * t_min) - n.plane) >= 0.0;
            bool e = ((o + d * t_max) - n.plane) >= 0.0;
            if (s == e) { cur = s ? n.front : n.back; continue; }
            double plane_t = (n.plane - o) / d;
            if (pt_in_range(start, end, plane_t)) {
                if (sp + 3 > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; return false; }
                pt_push(stk, sp, (uint32_t)(s ? n.back : n.front));  // pending far side: (node, start = plane_t)
                pt_push_f64(stk, sp, plane_t);
                cur = s ? n.front : n.back;
                end = plane_t;
                continue;
            }
            if (STATS) cnt->kd_plane_miss++;  // node.rs:146-147 / :177-178 would panic
        }
        if (sp == sp0) return false;
        start = pt_pop_f64(stk, sp);
        cur = (int32_t)pt_pop(stk, sp);
        if (sp == sp0) end = end0;
        else { int peek = sp; end = pt_pop_f64(stk, peek); }
    }
}

// flat_scene.rs:71-99 for one flattened node: transform the ray into model space, dispatch on the
// primitive (primitive.rs:55-62), keep the hit if it beats `best`. Returns true if best changed.
template <bool STATS, bool MESH = true, class Stack = PtStack>
PT_HD bool pt_test_node(const PtSceneView& sc, uint32_t node, const PtRay& ray, double start, PtHit& best, bool any,
                        const Stack& stk, int sp, PtCounters* cnt) {
    const uint32_t* info = sc.info + 4 * (size_t)node;
    uint32_t type = info[0], data = info[1];
    PtRay local = pt_ray_to_local(sc.inv + 12 * (size_t)node, ray);
    if (STATS) cnt->n_analytic++;
    double t;
    uint32_t part = 0;
    bool hit;
    switch (type) {
    case PT_SPHERE: case PT_PLANE: case PT_CUBE: case PT_CYLINDER: case PT_CONE:
        hit = pt_unit_prim_hit(type, local, start, pt_cand_end(best, node, 0), &t, &part);
        break;
    case PT_TRIANGLE: {  // stand-alone triangle, stored after the mesh triangles
        double beta, gamma;
        if (STATS) cnt->n_tri++;
        hit = pt_triangle_hit(sc.tri_v + 9 * (size_t)data, local, start, pt_cand_end(best, node, data), &t, &beta, &gamma);
        part = data;
        break;
    }
    default: {  // PT_MESH / PT_KDMESH: mesh.rs:146-167 (box test, then the nearest triangle)
        if (!MESH) return false;  // PT_MODE_FLAT_NOMESH: the host guarantees there is none
        const PtMeshInfo& m = sc.meshes[data];
        if (type == PT_KDMESH && m.kd_root >= 0) {
            uint32_t tri = 0;
            if (!pt_kdmesh_hit<STATS>(sc, m, local, start, pt_cand_end(best, node, 0), stk, sp, &t, &tri, cnt)) return false;
            best.t = t; best.node = node; best.sub = tri;
            return true;
        }
        if (STATS) cnt->n_bbox++;
        if (!pt_bbox_test_hit(m.bbox_inv, local, start, pt_cand_end(best, node, 0))) return false;
        bool changed = false;
        const double* tri_v = sc.tri_v;
        pt_bvh_walk<STATS>(sc.bvh4, m.blas_root, local, best.t, stk, sp,
            [&](uint32_t first, uint32_t count, int) -> bool {
                for (uint32_t i = 0; i < count; i++) {
                    uint32_t tri = sc.bvh_items[first + i];
                    double tt, beta, gamma;
                    if (STATS) cnt->n_tri++;
                    if (pt_triangle_hit(tri_v + 9 * (size_t)tri, local, start, pt_cand_end(best, node, tri), &tt, &beta, &gamma)) {
                        best.t = tt; best.node = node; best.sub = tri;
                        changed = true;
                        if (any) return true;
                    }
                }
                return false;
            }, cnt);
        return changed;
    }
    }
    if (!hit) return false;
    best.t = t; best.node = node; best.sub = part;
    return true;
}

// FLAT mode for scenes WITHOUT mesh instances: the generic walk with a one-node leaf handler.
// (Measured 4 % faster on big-scene than the two-level loop below with its mesh path compiled out:
// fewer loop-carried registers.)
template <bool STATS, class Stack>
PT_HD bool pt_trace_flat_simple(const PtSceneView& sc, const PtRay& ray, bool any, PtHit& best, const Stack& stk, PtCounters* cnt) {
    best.t = INFINITY; best.node = PT_NO_HIT; best.sub = 0;
    if (sc.n_nodes == 0) return false;
    pt_bvh2_walk<STATS>(sc.bvh, sc.tlas_root, ray, best.t, stk, 0,
        [&](uint32_t first, uint32_t count, int sp) -> bool {
            for (uint32_t i = 0; i < count; i++) {
                uint32_t node = sc.tlas_direct ? first : sc.bvh_items[first + i];
                if (pt_test_node<STATS, false>(sc, node, ray, PT_EPSILON, best, any, stk, sp, cnt) && any) return true;
            }
            return false;
        }, cnt);
    return best.node != PT_NO_HIT;
}

// FLAT mode: nearest hit over [EPSILON, inf) (ray.rs:139-141), or any hit for shadow rays
// (material.rs:174-179 only asks is_none()).
//
// `any` is a per-lane run-time flag, not a template parameter: lanes carrying shadow rays and lanes
// carrying primary / secondary rays walk the tree together in one instruction stream. The scene tree
// and the per-mesh triangle trees are walked by ONE loop: entering a mesh instance pushes a marker,
// switches the lane to the instance's model-space ray (ray.rs:130-135: same t in both spaces) and
// continues in the mesh's tree; popping the marker switches back. Lanes inside a mesh and lanes in
// the scene tree therefore share the inner-node code instead of waiting for each other.
#define PT_REF_MARKER 0xFFFFFFFEu
#define PT_REF_POP 0xFFFFFFFDu     // inside a walk: "take the next pending subtree" (never stored)
// The walk of the build's two-level bounding-volume tree as a resumable machine: begin() takes a ray, every step() is one
// round of the "while-while" loop - down through inner nodes to a leaf, the leaf's candidates, the next pending subtree -
// and returns true once the ray is finished (result in `best`). pt_render_kernel drives it so that a lane whose ray is
// finished can take its NEXT ray while its neighbours are still walking (pt_render_kernel.h); pt_trace_flat() below runs
// it to the end for one ray.
// MESH = false compiles the mesh-instance path out (scenes of analytic primitives and stand-alone
// triangles only): fewer live registers in the hot loop.
template <bool MESH, bool KDMESH = MESH, bool HIER = false>
struct PtBvhWalker {
    PtHit best;
    int sp;
    uint32_t cur;
    PtRay32 q;
    PtRay local;    // model-space ray of the mesh instance being walked
    uint32_t inst;  // flat node index of that instance, PT_NO_HIT while in the scene tree

    PT_HD bool begin(const PtSceneView& sc, const PtRay& ray) {  // false: nothing to walk, the ray is finished (a miss)
        best.t = INFINITY; best.node = PT_NO_HIT; best.sub = 0;
        sp = 0; inst = PT_NO_HIT; cur = sc.tlas_root;
        if (MESH) local = ray;
        if (sc.n_nodes == 0 || sc.tlas_root == PT_REF_EMPTY) return false;
        q = pt_ray32(ray);
        return true;
    }

    template <bool STATS, class Stack>
    PT_HD bool step(const PtSceneView& sc, const PtRay& ray, bool any, const Stack& stk, PtCounters* cnt) {
        // Scenes WITH mesh instances walk the four-child form of the trees: half as many dependent node fetches, which is
        // what a walk through a tree far larger than the caches waits for (big-soup +8 %, mirror +2 %). Mesh-free scenes
        // (their one tree is cache-resident) keep the two-child form: the same instruction count per level pair, and the
        // finer front-to-back order shrinks tmax sooner (big-scene 2.5 % faster than on the four-child form).
        while (MESH && !(cur & PT_REF_LEAF)) {
            const PtBvh4Node& n = sc.bvh4[cur];
            if (STATS) cnt->n_inner++;
            PT_WAVE_COUNT(4);
            float tm = (float)best.t; tm = tm + fabsf(tm) * 2.4e-7f;
            float t[4];
            uint32_t c[4] = {n.child[0], n.child[1], n.child[2], n.child[3]};
            pt_slab32_x4(n, q, tm, t);
            pt_sort4(t, c);
            if (!(t[0] < INFINITY)) {
                cur = PT_REF_EMPTY;  // nothing below: take the next pending subtree
                break;
            }
            const int hits = (t[1] < INFINITY) + (t[2] < INFINITY) + (t[3] < INFINITY);
            if (sp + hits > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; best.node = PT_NO_HIT; return true; }
            if (t[3] < INFINITY) pt_push(stk, sp, c[3]);
            if (t[2] < INFINITY) pt_push(stk, sp, c[2]);
            if (t[1] < INFINITY) pt_push(stk, sp, c[1]);
            cur = c[0];
        }
        while (!MESH && !(cur & PT_REF_LEAF)) {
            const PtBvhNode& n = sc.bvh[cur];
            if (STATS) cnt->n_inner++;
            PT_WAVE_COUNT(4);
            float tm = (float)best.t; tm = tm + fabsf(tm) * 2.4e-7f;
            float t0, t1, lo0[3], hi0[3], lo1[3], hi1[3];
            pt_node_get_box(n, 0, lo0, hi0); pt_node_get_box(n, 1, lo1, hi1);
            bool h0 = pt_slab32(lo0, hi0, q, tm, &t0);
            bool h1 = pt_slab32(lo1, hi1, q, tm, &t1);
            uint32_t c0 = n.child0, c1 = n.child1;
            if (h0 && h1) {
                bool swap = t1 < t0;
                if (sp + 1 > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; best.node = PT_NO_HIT; return true; }
                pt_push(stk, sp, swap ? c0 : c1);
                cur = swap ? c1 : c0;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else {
                cur = PT_REF_EMPTY;  // nothing below: take the next pending subtree
                break;
            }
        }
        if (cur != PT_REF_EMPTY && cur != PT_REF_MARKER) {
            if (STATS) cnt->n_leaf++;
            PT_WAVE_COUNT(5);
            const uint32_t first = (cur & ~PT_REF_LEAF) >> 3, count = (cur & 7u) + 1u;
            bool entered = false;
            for (uint32_t i = 0; i < count && !entered; i++) {
                uint32_t item = ((!MESH || inst == PT_NO_HIT) && sc.tlas_direct) ? first : sc.bvh_items[first + i];
                if (MESH && inst != PT_NO_HIT) {  // a triangle of the current mesh instance (mesh.rs:157-166, triangle.rs:38-80)
                    double tt, beta, gamma;
                    if (STATS) cnt->n_tri++;
                    if (pt_triangle_hit(sc.tri_v + 9 * (size_t)item, local, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, inst, item), &tt, &beta, &gamma)) {
                        best.t = tt; best.node = inst; best.sub = item;
                        if (any) return true;
                    }
                    continue;
                }
                const uint32_t* info = sc.info + 4 * (size_t)item;
                uint32_t type = info[0], data = info[1];
                PtRay lr = pt_node_local_ray<HIER>(sc, item, ray);  // flat_scene.rs:74
                if (STATS) cnt->n_analytic++;
                if (MESH && KDMESH && type == PT_KDMESH && sc.meshes[data].kd_root >= 0) {  // the reference's own triangle tree (quirk Q3)
                    double t; uint32_t tri = 0;
                    if (pt_kdmesh_hit<STATS>(sc, sc.meshes[data], lr, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, item, 0), stk, sp, &t, &tri, cnt)) {
                        best.t = t; best.node = item; best.sub = tri;
                        if (any) return true;
                    }
                } else if (MESH && (type == PT_MESH || type == PT_KDMESH)) {  // mesh.rs:146-155: box test, then the triangles
                    const PtMeshInfo& m = sc.meshes[data];
                    if (STATS) cnt->n_bbox++;
                    if (m.blas_root == PT_REF_EMPTY || !pt_bbox_test_hit(m.bbox_inv, lr, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, item, 0))) continue;
                    if (sp + 2 > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; best.node = PT_NO_HIT; return true; }
                    // remaining items of this leaf (scene leaves hold one node unless PORTRAYER_TLAS_LEAF > 1)
                    if (i + 1 < count) pt_push(stk, sp, PT_REF_LEAF | ((first + i + 1) << 3) | (count - i - 2));
                    pt_push(stk, sp, PT_REF_MARKER);
                    local = lr; q = pt_ray32(lr); inst = item;
                    cur = m.blas_root;
                    entered = true;
                } else {
                    double t; uint32_t part = 0; bool hit;
                    if (type == PT_TRIANGLE) {
                        double beta, gamma;
                        if (STATS) cnt->n_tri++;
                        hit = pt_triangle_hit(sc.tri_v + 9 * (size_t)data, lr, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, item, data), &t, &beta, &gamma);
                        part = data;
                    } else {
                        hit = pt_unit_prim_hit(type, lr, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, item, 0), &t, &part);
                    }
                    if (hit) {
                        best.t = t; best.node = item; best.sub = part;
                        if (any) return true;
                    }
                }
            }
            if (entered) return false;
        }
        // next pending subtree
        for (;;) {
            if (sp == 0) return true;
            cur = pt_pop(stk, sp);
            if (!MESH || cur != PT_REF_MARKER) break;
            q = pt_ray32(ray); inst = PT_NO_HIT;  // leaving the mesh instance: back to the world-space ray
        }
        return false;
    }
};

// FLAT mode: nearest hit over [EPSILON, inf) (ray.rs:139-141), or any hit for shadow rays (material.rs:174-179 only
// asks is_none()), for ONE ray.
template <bool STATS, bool MESH, bool KDMESH = MESH, bool HIER = false, class Stack = PtStack>
PT_HD bool pt_trace_flat(const PtSceneView& sc, const PtRay& ray, bool any, PtHit& best, const Stack& stk, PtCounters* cnt) {
    PtBvhWalker<MESH, KDMESH, HIER> w;
    if (w.begin(sc, ray))
        while (!w.template step<STATS>(sc, ray, any, stk, cnt)) {}
    best = w.best;
    return best.node != PT_NO_HIT;
}

// The reference's k-d walk (kdtree/node.rs:112-202) as a resumable machine like PtBvhWalker. A pending far side is
// (node, start) = 3 words: its range end is the start of the entry below it on the stack (or +inf), because the current
// `end` always equals the plane parameter of the innermost straddled split whose near side is being walked.
//
// MESH = false (scenes of analytic primitives): "while-while" - a step works through splits, culls and pops until the
// lane holds a LEAF, then the lanes run the leaf code (the f32 boxes of the leaf's references and the f64 primitive
// tests, an order of magnitude more instructions than a split step) together.
// MESH = true: a step is ONE node, split or leaf - with mesh instances a leaf can hold a whole mesh-tree walk, and the
// while-while form made every lane wait for it (mirror KD 7.1 -> 6.7 Gray/s, profiles/r02/notes.md).
template <bool MESH>
struct PtKdWalker {
    PtHit best;
    double start, end;
    int sp;
    int32_t cur;
    PtRayPk q;
    static constexpr int32_t NONE = -1;

    PT_HD bool begin(const PtSceneView& sc, const PtRay& ray) {
        best.t = INFINITY; best.node = PT_NO_HIT; best.sub = 0;
        start = PT_EPSILON; end = INFINITY;  // ray.rs:140
        sp = 0; cur = 0;
        q = pt_raypk(ray);
        return true;
    }

    template <bool STATS, class Stack>
    PT_HD bool step(const PtSceneView& sc, const PtRay& ray, bool any, const Stack& stk, PtCounters* cnt) {
        const double extent = sc.kd_extent;
        if (MESH) return step_one_node<STATS>(sc, ray, any, stk, cnt, extent);
        return step_to_leaf<STATS>(sc, ray, any, stk, cnt, extent);
    }

    template <bool STATS, class Stack>
    PT_HD bool step_one_node(const PtSceneView& sc, const PtRay& ray, bool any, const Stack& stk, PtCounters* cnt, const double extent) {
        const PtKdNode n = sc.kd[cur];
        // The ray's segment [start, end), rounded outward, against conservative f32 boxes: first the union of
        // everything below this tree node - a subtree the segment does not reach reports no hit, which is all
        // the reference would find out by walking it - then, in a leaf, each referenced node's own box.
        float seg0 = (float)start, seg1 = (float)end;
        seg0 = seg0 - fabsf(seg0) * 2.4e-7f; seg1 = seg1 + fabsf(seg1) * 2.4e-7f;
        if (sc.kd_box && !pt_slab_seg_pk(n.box, n.box + 3, q, seg0, seg1)) {
            if (STATS) cnt->kd_culled++;
        } else if (n.axis < 0) {  // Leaf: ray.rs:87-99 fold over the leaf's nodes, reference order, strict ends
            if (STATS) cnt->n_leaf++;
            bool found = false;
            // The reference's tree puts a node into every leaf its box touches (6.8 references per node on
            // big-scene) and tests all of them. A node whose (padded, outward-rounded) world box the ray's
            // segment [start, end) does not reach cannot report a hit in that range, so it is skipped before
            // the f64 test: same answers, 60 instead of 2 exact tests per ray saved.
            for (int32_t i = 0; i < n.count; i++) {
                const uint32_t item = sc.kd_items[n.first + i];
                if (sc.node_box) {
                    const float* b = sc.node_box + 6 * (size_t)(n.first + i);
                    if (STATS) cnt->n_bbox++;
                    if (!pt_slab_seg_pk(b, b + 3, q, seg0, seg1)) continue;
                }
                PtHit lb; lb.t = end; lb.node = PT_NO_HIT; lb.sub = 0;
                if (pt_test_node<STATS>(sc, item, ray, start, lb, any, stk, sp, cnt)) {
                    best = lb; end = lb.t; found = true;
                    if (any) return true;
                }
            }
            if (found) return true;  // node.rs:153-157: the first side that hits wins
        } else {
            if (STATS) cnt->n_inner++;
            double t_max = start + extent;                                   // node.rs:118
            if (!pt_in_range(start, end, t_max)) t_max = end - PT_EPSILON;   // node.rs:121
            double t_min = start + PT_EPSILON;                               // node.rs:124
            double o = pt_axis(ray.o, n.axis), d = pt_axis(ray.d, n.axis);
            bool s = ((o + d * t_min) - n.plane) >= 0.0;                     // infinite_plane.rs:27-35
            bool e = ((o + d * t_max) - n.plane) >= 0.0;
            if (s == e) { cur = s ? n.front : n.back; return false; }
            double plane_t = (n.plane - o) / d;                              // node.rs:90-109
            if (pt_in_range(start, end, plane_t)) {
                if (sp + 3 > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; best.node = PT_NO_HIT; return true; }
                pt_push(stk, sp, (uint32_t)(s ? n.back : n.front));
                pt_push_f64(stk, sp, plane_t);
                cur = s ? n.front : n.back;
                end = plane_t;
                return false;
            }
            // node.rs:146-147 / :177-178: the reference panics here; report a miss for this subtree
            if (STATS) cnt->kd_plane_miss++;
        }
        if (sp == 0) { return true; }
        start = pt_pop_f64(stk, sp);
        cur = (int32_t)pt_pop(stk, sp);
        if (sp == 0) end = INFINITY;
        else { int peek = sp; end = pt_pop_f64(stk, peek); }
        return false;
    }

    template <bool STATS, class Stack>
    PT_HD bool step_to_leaf(const PtSceneView& sc, const PtRay& ray, bool any, const Stack& stk, PtCounters* cnt, const double extent) {
        PtKdNode n;
        float seg0 = 0.0f, seg1 = 0.0f;
        bool finished = false;
        for (;;) {  // until this lane holds a leaf whose bounds its segment reaches
            if (cur == NONE) {  // next pending far side
                if (sp == 0) { finished = true; break; }
                start = pt_pop_f64(stk, sp);
                cur = (int32_t)pt_pop(stk, sp);
                if (sp == 0) end = INFINITY;
                else { int peek = sp; end = pt_pop_f64(stk, peek); }
            }
            n = sc.kd[cur];
            PT_WAVE_COUNT(4);
            // The ray's segment [start, end), rounded outward, against conservative f32 boxes: first the union of
            // everything below this tree node - a subtree the segment does not reach reports no hit, which is all
            // the reference would find out by walking it - then, in a leaf, each referenced node's own box.
            seg0 = (float)start; seg1 = (float)end;
            seg0 = seg0 - fabsf(seg0) * 2.4e-7f; seg1 = seg1 + fabsf(seg1) * 2.4e-7f;
            if (sc.kd_box && !pt_slab_seg_pk(n.box, n.box + 3, q, seg0, seg1)) {
                if (STATS) cnt->kd_culled++;
                cur = NONE;
                continue;
            }
            if (n.axis < 0) break;  // a leaf to test
            if (STATS) cnt->n_inner++;
            double t_max = start + extent;                                   // node.rs:118
            if (!pt_in_range(start, end, t_max)) t_max = end - PT_EPSILON;   // node.rs:121
            double t_min = start + PT_EPSILON;                               // node.rs:124
            double o = pt_axis(ray.o, n.axis), d = pt_axis(ray.d, n.axis);
            bool s = ((o + d * t_min) - n.plane) >= 0.0;                     // infinite_plane.rs:27-35
            bool e = ((o + d * t_max) - n.plane) >= 0.0;
            if (s == e) { cur = s ? n.front : n.back; continue; }
            double plane_t = (n.plane - o) / d;                              // node.rs:90-109
            if (pt_in_range(start, end, plane_t)) {
                if (sp + 3 > pt_stack_total(stk)) { pt_stack_overflow(stk); if (STATS) cnt->stack_overflow++; best.node = PT_NO_HIT; return true; }
                pt_push(stk, sp, (uint32_t)(s ? n.back : n.front));
                pt_push_f64(stk, sp, plane_t);
                cur = s ? n.front : n.back;
                end = plane_t;
                continue;
            }
            // node.rs:146-147 / :177-178: the reference panics here; report a miss for this subtree
            if (STATS) cnt->kd_plane_miss++;
            cur = NONE;
        }
        if (finished) return true;
        // Leaf: ray.rs:87-99 fold over the leaf's nodes, reference order, strict ends
        if (STATS) cnt->n_leaf++;
        PT_WAVE_COUNT(5);
        bool found = false;
        // The reference's tree puts a node into every leaf its box touches (6.8 references per node on
        // big-scene) and tests all of them. A node whose (padded, outward-rounded) world box the ray's
        // segment [start, end) does not reach cannot report a hit in that range, so it is skipped before
        // the f64 test: same answers, 60 instead of 2 exact tests per ray saved.
        for (int32_t i = 0; i < n.count; i++) {
            const uint32_t item = sc.kd_items[n.first + i];
            if (sc.node_box) {
                const float* b = sc.node_box + 6 * (size_t)(n.first + i);
                if (STATS) cnt->n_bbox++;
                if (!pt_slab_seg_pk(b, b + 3, q, seg0, seg1)) continue;
            }
            PtHit lb; lb.t = end; lb.node = PT_NO_HIT; lb.sub = 0;
            if (pt_test_node<STATS, MESH>(sc, item, ray, start, lb, any, stk, sp, cnt)) {
                best = lb; end = lb.t; found = true;
                if (any) return true;
                // the segment has shrunk: later references of this leaf are culled against the new end
                seg1 = (float)end; seg1 = seg1 + fabsf(seg1) * 2.4e-7f;
            }
        }
        if (found) return true;  // node.rs:153-157: the first side that hits wins
        cur = NONE;
        return false;
    }
};

template <bool STATS, bool MESH = true, class Stack = PtStack>
PT_HD bool pt_trace_kd(const PtSceneView& sc, const PtRay& ray, bool any, PtHit& best, const Stack& stk, PtCounters* cnt) {
    PtKdWalker<MESH> w;
    if (w.begin(sc, ray))
        while (!w.template step<STATS>(sc, ray, any, stk, cnt)) {}
    best = w.best;
    return best.node != PT_NO_HIT;
}

// ------------------------------------------------------------------------------------------------
// ONE walk per wavefront (scenes of analytic primitives whose hits spawn no rays: PT_MODE_FLAT_NOMESH / HIER_NOMESH, PARK = 0).
//
// The 64 rays of a wavefront pass through one pixel (or a few) and visit nearly the same nodes - the per-lane walk has 61.8 of
// 64 lanes busy per step on big-scene - so the node index and the stack can be WAVE-UNIFORM instead of per lane:
//  * a node is fetched once per wavefront, through the scalar cache into SGPRs (s_load_dwordx16), and its boxes are scalar
//    operands of the 64 slab tests; the stack is one list in the wavefront's LDS columns; no lane keeps a node index, a stack
//    pointer or a stack column;
//  * a child is entered when ANY lane's ray reaches its box - each lane tests with its own nearest hit so far - and the child
//    most of those lanes call nearer is entered first. Every live lane tests what the wavefront visits (a test below a box its
//    ray misses finds nothing and costs no time: the instruction is issued for the others anyway). The counting build also
//    carries, per stack entry, the mask of the lanes whose rays reach the subtree, and counts a lane's steps and tests only
//    there - what a per-lane walk in the same order would count; carrying the masks in the timed build cost 6 %;
//  * in a leaf the flattened node's record (type, inverse transform) is again one scalar fetch for all lanes.
// The result of a ray does not depend on the order of the walk or on which other candidates are tested (FLAT: nearest hit,
// lowest index on exact ties; HIER: nearest hit, first in depth-first order), so this is a change of schedule, not of results.
// Measured on big-scene: 16.6 -> 19.0 Gray/s with the first version (profiles/r02/notes.md). -DPT_NO_PACKET keeps the per-lane walk.
// ------------------------------------------------------------------------------------------------
typedef uint32_t pt_u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t pt_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t pt_u32x4 __attribute__((ext_vector_type(4)));
// A wave-uniform address as the scalar-register pair the s_load instructions need, wherever the compiler chose to compute it
// (an inline-asm "s" operand that arrives in vector registers is a build error, not a copy).
PT_HD const void* pt_uniform_ptr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long a = (unsigned long long)p;
    const unsigned long long lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return (const void*)((hi << 32) | lo);
#else
    return p;
#endif
}
PT_HD pt_u32x16 pt_sload16(const void* p) {
    pt_u32x16 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pt_uniform_ptr(p)) : "memory");
#else
    v = *static_cast<const pt_u32x16*>(p);
#endif
    return v;
}
PT_HD pt_u32x8 pt_sload8(const void* p) {
    pt_u32x8 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pt_uniform_ptr(p)) : "memory");
#else
    v = *static_cast<const pt_u32x8*>(p);
#endif
    return v;
}
PT_HD pt_u32x4 pt_sload4(const void* p) {
    pt_u32x4 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pt_uniform_ptr(p)) : "memory");
#else
    v = *static_cast<const pt_u32x4*>(p);
#endif
    return v;
}
PT_HD double pt_f64_of(uint32_t lo, uint32_t hi) {
    union { double d; uint32_t u[2]; } c;
    c.u[0] = lo; c.u[1] = hi;
    return c.d;
}
PT_HD float pt_f32_of(uint32_t u) {
    union { float f; uint32_t u; } c;
    c.u = u;
    return c.f;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define PT_UNIFORM_U32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define PT_BALLOT(x) __ballot(x)
#define PT_LANE_ID() (threadIdx.x & 63u)
#else
#define PT_UNIFORM_U32(x) ((uint32_t)(x))
#define PT_BALLOT(x) ((x) ? 1ull : 0ull)
#define PT_LANE_ID() 0u
#endif

// 12 doubles (rows 0..2 of a 3x4 matrix) at a wave-uniform address, in one round trip through the scalar cache
PT_HD void pt_sload_mat12(const double* p, double m[12]) {
    pt_u32x16 a;
    pt_u32x8 b;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(pt_uniform_ptr(p)) : "memory");
#else
    a = *reinterpret_cast<const pt_u32x16*>(p);
    b = *reinterpret_cast<const pt_u32x8*>(reinterpret_cast<const char*>(p) + 64);
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; k++) m[8 + k] = pt_f64_of(b[2 * k], b[2 * k + 1]);
}
// ... and the 16 bytes that follow the 96 (PtMeshInfo: the box inverse, then {tri_first, tri_count, blas_root, kd_root}), in the same round trip
PT_HD void pt_sload_mat12_x4(const double* p, double m[12], pt_u32x4& tail) {
    pt_u32x16 a;
    pt_u32x8 b;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x60\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(tail) : "s"(pt_uniform_ptr(p)) : "memory");
#else
    a = *reinterpret_cast<const pt_u32x16*>(p);
    b = *reinterpret_cast<const pt_u32x8*>(reinterpret_cast<const char*>(p) + 64);
    tail = *reinterpret_cast<const pt_u32x4*>(reinterpret_cast<const char*>(p) + 96);
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; k++) m[8 + k] = pt_f64_of(b[2 * k], b[2 * k + 1]);
}
// May a level whose matrices are the identity be skipped for this ray? ((1 x + 0 y) + 0 z) + 0 is x bit for bit unless x is -0 (which
// the additions turn into +0) or something is not finite (0 times infinity). Wave-uniform: true when no lane's ray has such a component.
PT_HD bool pt_ray_identity_safe(const PtRay& r, bool has_ray) {
    auto odd = [](double v) {
        union { double d; uint32_t u[2]; } c; c.d = v;
        return c.u[1] == 0x80000000u || (c.u[1] & 0x7FF00000u) == 0x7FF00000u;  // -0 (or a negative denormal: taken along) / infinity / NaN
    };
    const bool bad = has_ray && (odd(r.o.x) || odd(r.o.y) || odd(r.o.z) || odd(r.d.x) || odd(r.d.y) || odd(r.d.z));
#if defined(__HIP_DEVICE_COMPILE__)
    return !__any(bad);
#else
    return !bad;
#endif
}
// pt_node_local_ray<true> for a wave-uniform node: the path (hier_rec: one scalar fetch) and every SceneNode's inverse come through the
// scalar cache. identity_ok = pt_ray_identity_safe(ray), worked out once per walk: leading levels that are the identity are skipped then.
PT_HD PtRay pt_node_local_ray_rec(const PtSceneView& sc, uint32_t node, const pt_u32x8& rec, const PtRay& ray, bool identity_ok) {
    PtRay r = ray;
    const uint32_t len = rec[0] & 255u;
    if (len == 255u) {  // a path of more than seven levels
        const uint32_t k0 = PT_UNIFORM_U32(sc.chain_off[node]), k1 = PT_UNIFORM_U32(sc.chain_off[node + 1]);
        for (uint32_t k = k0; k < k1; k++) {
            const uint32_t g = PT_UNIFORM_U32(sc.chain[k]);
            double m[12];
            pt_sload_mat12(sc.g_inv + 12 * (size_t)g, m);
            r = pt_ray_to_local(m, r);
        }
        return r;
    }
    bool untouched = identity_ok;  // the ray is still the world ray
    uint32_t ids[7] = {rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7]};
    for (uint32_t k = 0; k < len; k++) {
        if (untouched && ((rec[0] >> (8u + k)) & 1u)) continue;
        untouched = false;
        uint32_t g = ids[0];
#pragma unroll
        for (int j = 1; j < 7; j++) g = k == (uint32_t)j ? ids[j] : g;
        double m[12];
        pt_sload_mat12(sc.g_inv + 12 * (size_t)g, m);
        r = pt_ray_to_local(m, r);
    }
    return r;
}
// The same with the node's OWN level's inverse (sc.own_inv, fetched by the caller in the round trip that brought the path record): when every level above the
// node's own is an identity that may be skipped - every reference scene: primitives under an untransformed root - the own level is all there is to apply, and
// no second, dependent fetch is needed (round 4: the hierarchical semantics' leaf tests took two round trips where flat_scene's take one).
PT_HD PtRay pt_node_local_ray_rec_own(const PtSceneView& sc, uint32_t node, const pt_u32x8& rec, const double* own, const PtRay& ray, bool identity_ok) {
#ifndef PT_HIER_NO_OWN
    const uint32_t len = rec[0] & 255u;
    if (identity_ok && len >= 1u && len <= 7u) {
        const uint32_t above = (1u << (len - 1u)) - 1u;  // the levels before the last
        if (((rec[0] >> 8) & above) == above) {
            if ((rec[0] >> (8u + len - 1u)) & 1u) return ray;  // the own level is an identity too
            return pt_ray_to_local(own, ray);
        }
    }
#endif
    return pt_node_local_ray_rec(sc, node, rec, ray, identity_ok);
}
PT_HD PtRay pt_node_local_ray_uniform(const PtSceneView& sc, uint32_t node, const PtRay& ray, bool identity_ok) {
    pt_u32x8 rec;
    double own[12];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_HIER_NO_OWN)
    pt_u32x16 a;
    pt_u32x8 b;
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx16 %1, %4, 0x0\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(rec), "=&s"(a), "=&s"(b) : "s"(pt_uniform_ptr(sc.hier_rec + 8 * (size_t)node)), "s"(pt_uniform_ptr(sc.own_inv + 12 * (size_t)node)) : "memory");
#pragma unroll
    for (int k = 0; k < 8; k++) own[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; k++) own[8 + k] = pt_f64_of(b[2 * k], b[2 * k + 1]);
#else
    rec = pt_sload8(sc.hier_rec + 8 * (size_t)node);
    for (int k = 0; k < 12; k++) own[k] = sc.own_inv[12 * (size_t)node + k];
#endif
    return pt_node_local_ray_rec_own(sc, node, rec, own, ray, identity_ok);
}

// ------------------------------------------------------------------------------------------------
// The slab test of the wave-uniform walks: both children of a node in one go, the node's planes as SCALAR operands.
//
// What limits these walks is the SCALAR unit - one per CU, shared by its sixteen wavefronts (profiles/r03/notes.md: 5.9e9 scalar
// against 4.9e9 vector instructions per big-scene frame with the first version of this test, which let the scalar unit pick the
// entering / leaving plane of every axis) - so everything that can be per-lane arithmetic is, and the scalar side of a step is
// one load, one block of mask logic and the branches.
//
// Per lane and axis two pairs of f32 constants: A = (i, c) applied to the LOWER planes of the children, B = (i, c) applied to the
// UPPER planes. For a positive direction the ray enters a slab through the lower plane, so A holds the "entering" constants
// (i_n, c_n) and B the "leaving" ones (i_f, c_f); for a negative direction it is the other way round. A plane coordinate P gives
// the parameter fma(P, i, c) ~ (P - o) / d; min / max of the two planes' values are the entering / leaving parameters. The
// constants are rounded so that every error makes the overlap LONGER (the walk may only err towards testing more candidates):
//     i0  = rcp((float)d)                        relative error < 2^-22.4 (conversion 2^-24, v_rcp_f32 one ulp)
//     i_n = i0 (1 - 2^-21),  i_f = i0 (1 + 2^-21)    so |i_n| < |1 / d| < |i_f| by at least 2^-23 relative, whatever i0's error
//     c_n = fl(-o i_n) - margin,  c_f = fl(-o i_f) + margin   (the product in f64 from the f64 origin; margin = 2^-22 relative
//                                                              + 1e-37: more than the conversion's rounding)
// Entering: fma(P, i_n, c_n) = (P - o) i_n - m before its one rounding; for a positive parameter T = (P - o) / d that is at most
// T (1 - 2^-23), which the fma's rounding (2^-24) cannot lift above T; a negative one stays negative (it is clamped to 0 anyway).
// Leaving: fma(P, i_f, c_f) >= T (1 + 2^-23) (1 - 2^-24) >= T for T >= 0; a box with T < 0 lies behind the ray and may be rejected.
// For a box the ray really passes (entering <= leaving, leaving >= 0) the two values come out in that order, so min / max pick
// them correctly; for any other box a wrong order can only turn a rejection into a visit or keep it a rejection.
// An axis the ray is parallel to (|i0| > 1e18 or NaN - which also keeps every product below FLT_MAX for |coordinates| <= 1e18) is
// switched off: A = (0, -inf), B = (0, +inf).
// A step is 6 packed fmas + 20 min / max + 3 compares for both children.
// ------------------------------------------------------------------------------------------------
// Constants of `r` for this lane. *oct (wave-uniform): how the directions of the rays of the lanes in `lanes` relate to the axes -
// bit a set: they all enter slabs of axis a through the UPPER plane (negative direction), clear: through the lower one; PT_OCT_MIXED
// when some axis has rays of both signs among those lanes. Rays through one pixel, or from neighbouring points to one light,
// nearly always share their signs, and then the tree step needs no min / max to tell entering from leaving (pt_slab_pk2).
// Slot `slot` of a mesh leaf (wave-uniform): the triangle's record into scalar registers (dwords 0..15 in `a`, 16 / 17 in b0 / b1), its index returned.
// Default: the record comes from tri_leaf, laid out in slot order - ONE fetch, and a leaf's second triangle sits in the line(s) the first one brought in;
// -DPT_TRI_VIA_ITEMS (and the corner form, -DPT_NO_TRI_EDGES): the slot's index from bvh_items first, then the record it names (two dependent fetches).
PT_HD uint32_t pt_load_leaf_triangle(const PtSceneView& sc, uint32_t slot, pt_u32x16& a, uint32_t& b0, uint32_t& b1);
#ifdef PT_NO_TRI_EDGES  // A/B: the wave-uniform walks test triangles from their corners again
#define PT_TRI_REC(sc) (sc).tri_v
#define PT_TRI_HIT pt_triangle_hit
#else
#define PT_TRI_REC(sc) (sc).tri_e
#define PT_TRI_HIT pt_triangle_hit_e
#endif
// (the same from a base address the caller holds in scalar registers - pt_pin_ptr(sc.tri_leaf) -: read through the scene view it was fetched again for every triangle)
PT_HD uint32_t pt_load_leaf_triangle_at(const double* tri_leaf, uint32_t slot, pt_u32x16& a, uint32_t& b0, uint32_t& b1) {
    const double* rec = tri_leaf + 10 * (size_t)slot;
#if defined(__HIP_DEVICE_COMPILE__)
    pt_u32x4 b;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(pt_uniform_ptr(rec)) : "memory");
    b0 = b[0]; b1 = b[1];
    return b[2];
#else
    a = *reinterpret_cast<const pt_u32x16*>(rec);
    b0 = reinterpret_cast<const uint32_t*>(rec)[16]; b1 = reinterpret_cast<const uint32_t*>(rec)[17];
    return reinterpret_cast<const uint32_t*>(rec)[18];
#endif
}
PT_HD uint32_t pt_load_leaf_triangle(const PtSceneView& sc, uint32_t slot, pt_u32x16& a, uint32_t& b0, uint32_t& b1) {
#if defined(PT_TRI_VIA_ITEMS) || defined(PT_NO_TRI_EDGES)
    const uint32_t tri = PT_UNIFORM_U32(sc.bvh_items[slot]);
    const double* rec = PT_TRI_REC(sc) + 9 * (size_t)tri;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 b;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(pt_uniform_ptr(rec)) : "memory");
    b0 = b[0]; b1 = b[1];
#else
    a = *reinterpret_cast<const pt_u32x16*>(rec);
    b0 = reinterpret_cast<const uint32_t*>(rec)[16]; b1 = reinterpret_cast<const uint32_t*>(rec)[17];
#endif
    return tri;
#else
    const double* rec = sc.tri_leaf + 10 * (size_t)slot;
#if defined(__HIP_DEVICE_COMPILE__)
    pt_u32x4 b;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(pt_uniform_ptr(rec)) : "memory");
    b0 = b[0]; b1 = b[1];
    return b[2];
#else
    a = *reinterpret_cast<const pt_u32x16*>(rec);
    b0 = reinterpret_cast<const uint32_t*>(rec)[16]; b1 = reinterpret_cast<const uint32_t*>(rec)[17];
    return reinterpret_cast<const uint32_t*>(rec)[18];
#endif
#endif
}
#define PT_OCT_MIXED 8
PT_HD PtRayPk pt_raypk(const PtRay& r, bool lanes, int* oct) {
    PtRayPk q;
    const int sx = pt_raypk_axis(r.o.x, r.d.x, &q.a[0], &q.b[0]);
    const int sy = pt_raypk_axis(r.o.y, r.d.y, &q.a[1], &q.b[1]);
    const int sz = pt_raypk_axis(r.o.z, r.d.z, &q.a[2], &q.b[2]);
    const bool nx = PT_BALLOT(lanes && sx == 2) != 0ull, px = PT_BALLOT(lanes && sx == 1) != 0ull;
    const bool ny = PT_BALLOT(lanes && sy == 2) != 0ull, py = PT_BALLOT(lanes && sy == 1) != 0ull;
    const bool nz = PT_BALLOT(lanes && sz == 2) != 0ull, pz = PT_BALLOT(lanes && sz == 1) != 0ull;
    // A lane with a switched-off axis holds A = (0, -inf), B = (0, +inf) there: "entering through the lower plane". Under an octant
    // whose bit for that axis is set pt_slab_pk2 would read B as the entering value (+inf) and reject every box for that lane - its ray
    // would silently miss the scene (level cameras with centre sampling: a whole row / column of pixels). Such a wavefront takes the
    // per-lane form, where min / max sort the pair whatever the octant.
    const bool off = PT_BALLOT(lanes && (sx == 0 || sy == 0 || sz == 0)) != 0ull;
    *oct = (off || (nx && px) || (ny && py) || (nz && pz)) ? PT_OCT_MIXED : ((nx ? 1 : 0) | (ny ? 2 : 0) | (nz ? 4 : 0));
    return q;
}
PT_HD pt_f32x2 pt_pair_f32(uint32_t a, uint32_t b) { pt_f32x2 v; v.x = pt_f32_of(a); v.y = pt_f32_of(b); return v; }
// packed fma with per-lane constants (i, c) broadcast to both halves: (p.x i + c, p.y i + c)
PT_HD pt_f32x2 pt_pk_fma_bcast(pt_f32x2 planes, pt_f32x2 ic) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    pt_f32x2 r;
    // src1 = ic.x for both halves (op_sel 0 / op_sel_hi 0), src2 = ic.y for both halves (op_sel 1 / op_sel_hi 1)
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "s"(planes), "v"(ic));
    return r;
#else
    pt_f32x2 r;
    r.x = __builtin_fmaf(planes.x, ic.x, ic.y); r.y = __builtin_fmaf(planes.y, ic.x, ic.y);
    return r;
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_FCMP_LE(a, b) __builtin_amdgcn_fcmpf((a), (b), 5)  // ordered <=, as a lane mask
#define PT_FCMP_LT(a, b) __builtin_amdgcn_fcmpf((a), (b), 4)
#else
#define PT_FCMP_LE(a, b) (((a) <= (b)) ? 1ull : 0ull)
#define PT_FCMP_LT(a, b) (((a) < (b)) ? 1ull : 0ull)
#endif
// min / max as single instructions: fmaxf / fminf on a value that comes out of inline asm make the compiler canonicalise it first
// (one v_max x, x per operand - twelve extra instructions per tree step)
PT_HD float pt_max2_raw(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return fmaxf(a, b);
#endif
}
PT_HD float pt_min2_raw(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return fminf(a, b);
#endif
}
PT_HD float pt_max3_zero_raw(float a, float b) {  // max(a, b, 0)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    float r; asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return fmaxf(fmaxf(a, b), 0.0f);
#endif
}
PT_HD float pt_max3_raw(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    return fmaxf(fmaxf(a, b), c);
#endif
}
PT_HD float pt_min3_raw(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    return fminf(fminf(a, b), c);
#endif
}
// Both child boxes of node record `v` (PtBvhNode as 16 dwords in scalar registers) against every lane's ray over [0, tm].
// Out: masks of the lanes whose rays reach child 0 / child 1, and of those that reach child 1 strictly before child 0.
// OCT (see pt_raypk): with wave-uniform signs the entering value of an axis is the lower planes' (bit clear) or the upper planes'
// (bit set) - known at compile time, 6 packed fmas + 8 min / max + 3 compares; PT_OCT_MIXED: min / max of the two per lane (+ 12).
// NEAR: the lane's segment starts at t0 > 0 (the k-d semantics' nested mesh walks: a leaf's range [start, end)) - the entering value is
// clamped there instead of at 0, the same instruction with a register in place of the constant.
template <int OCT, bool NEAR = false>
PT_HD void pt_slab_pk2(const pt_u32x16& v, const PtRayPk& q, float tm, unsigned long long* m0, unsigned long long* m1, unsigned long long* one_first, float t0 = 0.0f) {
    const pt_f32x2 ax = pt_pk_fma_bcast(pt_pair_f32(v[0], v[1]), q.a[0]), ay = pt_pk_fma_bcast(pt_pair_f32(v[2], v[3]), q.a[1]), az = pt_pk_fma_bcast(pt_pair_f32(v[4], v[5]), q.a[2]);
    const pt_f32x2 bx = pt_pk_fma_bcast(pt_pair_f32(v[6], v[7]), q.b[0]), by = pt_pk_fma_bcast(pt_pair_f32(v[8], v[9]), q.b[1]), bz = pt_pk_fma_bcast(pt_pair_f32(v[10], v[11]), q.b[2]);
    float tn0, tn1, tf0, tf1;
    if (OCT == PT_OCT_MIXED) {
        if (NEAR) {
            tn0 = pt_max3_raw(pt_max2_raw(pt_min2_raw(ax.x, bx.x), pt_min2_raw(ay.x, by.x)), pt_min2_raw(az.x, bz.x), t0);
            tn1 = pt_max3_raw(pt_max2_raw(pt_min2_raw(ax.y, bx.y), pt_min2_raw(ay.y, by.y)), pt_min2_raw(az.y, bz.y), t0);
        } else {
            tn0 = pt_max3_zero_raw(pt_max2_raw(pt_min2_raw(ax.x, bx.x), pt_min2_raw(ay.x, by.x)), pt_min2_raw(az.x, bz.x));
            tn1 = pt_max3_zero_raw(pt_max2_raw(pt_min2_raw(ax.y, bx.y), pt_min2_raw(ay.y, by.y)), pt_min2_raw(az.y, bz.y));
        }
        tf0 = pt_min3_raw(pt_min2_raw(pt_max2_raw(ax.x, bx.x), pt_max2_raw(ay.x, by.x)), pt_max2_raw(az.x, bz.x), tm);
        tf1 = pt_min3_raw(pt_min2_raw(pt_max2_raw(ax.y, bx.y), pt_max2_raw(ay.y, by.y)), pt_max2_raw(az.y, bz.y), tm);
    } else {
        const pt_f32x2 ex = (OCT & 1) ? bx : ax, lx = (OCT & 1) ? ax : bx;  // entering / leaving values per axis, both children
        const pt_f32x2 ey = (OCT & 2) ? by : ay, ly = (OCT & 2) ? ay : by;
        const pt_f32x2 ez = (OCT & 4) ? bz : az, lz = (OCT & 4) ? az : bz;
        if (NEAR) { tn0 = pt_max3_raw(pt_max2_raw(ex.x, ey.x), ez.x, t0); tn1 = pt_max3_raw(pt_max2_raw(ex.y, ey.y), ez.y, t0); }
        else { tn0 = pt_max3_zero_raw(pt_max2_raw(ex.x, ey.x), ez.x); tn1 = pt_max3_zero_raw(pt_max2_raw(ex.y, ey.y), ez.y); }
        tf0 = pt_min3_raw(pt_min2_raw(lx.x, ly.x), lz.x, tm); tf1 = pt_min3_raw(pt_min2_raw(lx.y, ly.y), lz.y, tm);
    }
    *m0 = PT_FCMP_LE(tn0, tf0);
    *m1 = PT_FCMP_LE(tn1, tf1);
    *one_first = PT_FCMP_LT(tn1, tn0);
}
// The same test handing out the entering parameters as well (pt_descend_mesh's branching form works out which child goes first only where both are reached).
template <int OCT, bool NEAR = false>
PT_HD void pt_slab_pk2_t(const pt_u32x16& v, const PtRayPk& q, float tm, unsigned long long* m0, unsigned long long* m1, float* tn0_out, float* tn1_out, float t0 = 0.0f) {
    const pt_f32x2 ax = pt_pk_fma_bcast(pt_pair_f32(v[0], v[1]), q.a[0]), ay = pt_pk_fma_bcast(pt_pair_f32(v[2], v[3]), q.a[1]), az = pt_pk_fma_bcast(pt_pair_f32(v[4], v[5]), q.a[2]);
    const pt_f32x2 bx = pt_pk_fma_bcast(pt_pair_f32(v[6], v[7]), q.b[0]), by = pt_pk_fma_bcast(pt_pair_f32(v[8], v[9]), q.b[1]), bz = pt_pk_fma_bcast(pt_pair_f32(v[10], v[11]), q.b[2]);
    float tn0, tn1, tf0, tf1;
    if (OCT == PT_OCT_MIXED) {
        if (NEAR) {
            tn0 = pt_max3_raw(pt_max2_raw(pt_min2_raw(ax.x, bx.x), pt_min2_raw(ay.x, by.x)), pt_min2_raw(az.x, bz.x), t0);
            tn1 = pt_max3_raw(pt_max2_raw(pt_min2_raw(ax.y, bx.y), pt_min2_raw(ay.y, by.y)), pt_min2_raw(az.y, bz.y), t0);
        } else {
            tn0 = pt_max3_zero_raw(pt_max2_raw(pt_min2_raw(ax.x, bx.x), pt_min2_raw(ay.x, by.x)), pt_min2_raw(az.x, bz.x));
            tn1 = pt_max3_zero_raw(pt_max2_raw(pt_min2_raw(ax.y, bx.y), pt_min2_raw(ay.y, by.y)), pt_min2_raw(az.y, bz.y));
        }
        tf0 = pt_min3_raw(pt_min2_raw(pt_max2_raw(ax.x, bx.x), pt_max2_raw(ay.x, by.x)), pt_max2_raw(az.x, bz.x), tm);
        tf1 = pt_min3_raw(pt_min2_raw(pt_max2_raw(ax.y, bx.y), pt_max2_raw(ay.y, by.y)), pt_max2_raw(az.y, bz.y), tm);
    } else {
        const pt_f32x2 ex = (OCT & 1) ? bx : ax, lx = (OCT & 1) ? ax : bx;
        const pt_f32x2 ey = (OCT & 2) ? by : ay, ly = (OCT & 2) ? ay : by;
        const pt_f32x2 ez = (OCT & 4) ? bz : az, lz = (OCT & 4) ? az : bz;
        if (NEAR) { tn0 = pt_max3_raw(pt_max2_raw(ex.x, ey.x), ez.x, t0); tn1 = pt_max3_raw(pt_max2_raw(ex.y, ey.y), ez.y, t0); }
        else { tn0 = pt_max3_zero_raw(pt_max2_raw(ex.x, ey.x), ez.x); tn1 = pt_max3_zero_raw(pt_max2_raw(ex.y, ey.y), ez.y); }
        tf0 = pt_min3_raw(pt_min2_raw(lx.x, ly.x), lz.x, tm); tf1 = pt_min3_raw(pt_min2_raw(lx.y, ly.y), lz.y, tm);
    }
    *m0 = PT_FCMP_LE(tn0, tf0);
    *m1 = PT_FCMP_LE(tn1, tf1);
    *tn0_out = tn0; *tn1_out = tn1;
}
// a lane's exclusive range end as the f32 the slab test compares with, rounded up (inf stays inf)
PT_HD float pt_tmax32(double t) { float tm = (float)t; return tm + fabsf(tm) * 2.4e-7f; }

// The scalar side of a step in one block: which children the wavefront enters (code 0: neither, 1: child 0 only, 2: child 1
// only, 3: both), which one first (`near`: the one the majority of the lanes that reach a child enter first; with a single
// child reached that is that child) and which one waits on the stack (`far`, meaningful for code 3).
// m0 / m1: lanes reaching child 0 / 1; one_first: lanes reaching child 1 before child 0; lanes: the lanes that count.
PT_HD uint32_t pt_step_decide(unsigned long long* m0_io, unsigned long long* m1_io, unsigned long long one_first, unsigned long long lanes, uint32_t c0, uint32_t c1,
                              uint32_t* near, uint32_t* far) {
    unsigned long long m0 = *m0_io, m1 = *m1_io;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_PK_ASM)
    uint32_t code, h1, n_sf, n_u, nr, fr;
    unsigned long long t;
    asm volatile(
        "s_and_b64 %[m0], %[m0], %[lanes]\n\t"
        "s_cselect_b32 %[code], 1, 0\n\t"
        "s_and_b64 %[m1], %[m1], %[lanes]\n\t"
        "s_cselect_b32 %[h1], 2, 0\n\t"
        "s_orn2_b64 %[t], %[of], %[m0]\n\t"      // lanes that reach child 1 first: m1 & (one_first | ~m0)
        "s_and_b64 %[t], %[t], %[m1]\n\t"
        "s_bcnt1_i32_b64 %[nsf], %[t]\n\t"
        "s_or_b64 %[t], %[m0], %[m1]\n\t"
        "s_bcnt1_i32_b64 %[nu], %[t]\n\t"
        "s_lshl_b32 %[nsf], %[nsf], 1\n\t"
        "s_cmp_gt_u32 %[nsf], %[nu]\n\t"         // most of them: child 1 first
        "s_cselect_b32 %[nr], %[c1], %[c0]\n\t"
        "s_cselect_b32 %[fr], %[c0], %[c1]\n\t"
        "s_or_b32 %[code], %[code], %[h1]"
        : [code] "=&s"(code), [h1] "=&s"(h1), [t] "=&s"(t), [nsf] "=&s"(n_sf), [nu] "=&s"(n_u), [nr] "=&s"(nr), [fr] "=&s"(fr), [m0] "+s"(m0), [m1] "+s"(m1)
        : [of] "s"(one_first), [lanes] "s"(lanes), [c0] "s"(c0), [c1] "s"(c1)
        : "scc");
    *near = nr; *far = fr; *m0_io = m0; *m1_io = m1;
    return code;
#else
    m0 &= lanes; m1 &= lanes;
    const unsigned long long second_first = m1 & (one_first | ~m0);
    const bool swap = __builtin_popcountll(second_first) * 2 > __builtin_popcountll(m0 | m1);
    *near = swap ? c1 : c0; *far = swap ? c0 : c1; *m0_io = m0; *m1_io = m1;
    return (m0 ? 1u : 0u) | (m1 ? 2u : 0u);
#endif
}
// node record `index` of the tree array at `base` (64-byte records): one scalar load with a 32-bit byte offset
PT_HD pt_u32x16 pt_sload_node(const void* base, uint32_t index) {
    pt_u32x16 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pt_uniform_ptr(base)), "s"(index << 6) : "memory");
#else
    v = static_cast<const pt_u32x16*>(base)[index];
#endif
    return v;
}

// One flattened node against every participating lane's ray; `node` is wave-uniform.
template <bool STATS, bool HIER>
PT_HD bool pt_test_node_uniform(const PtSceneView& sc, uint32_t node, const PtRay& ray, bool identity_ok, PtHit& best, PtCounters* cnt) {
    // the node's record in one round trip through the scalar cache: {type, data, flags, material} and rows 0..2 of its inverse
    pt_u32x4 info;
    pt_u32x16 a;  // doubles 0..7 of the 3x4 inverse
    pt_u32x8 b;   // doubles 8..11
    const void* info_ptr = sc.info + 4 * (size_t)node;
    const void* rec = sc.inv + 12 * (size_t)node;
#if defined(__HIP_DEVICE_COMPILE__)
    pt_u32x8 hrec;  // HIER: the node's path record
    if (HIER) {  // ... the node's path record (hier_rec) and the inverse of its OWN level instead of a composed inverse
        asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx8 %1, %5, 0x0\n\ts_load_dwordx16 %2, %6, 0x0\n\ts_load_dwordx8 %3, %6, 0x40\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(info), "=&s"(hrec), "=&s"(a), "=&s"(b)
                     : "s"(pt_uniform_ptr(info_ptr)), "s"(pt_uniform_ptr(sc.hier_rec + 8 * (size_t)node)), "s"(pt_uniform_ptr(sc.own_inv + 12 * (size_t)node)) : "memory");
    } else {
        asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx16 %1, %4, 0x0\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(info), "=&s"(a), "=&s"(b) : "s"(pt_uniform_ptr(info_ptr)), "s"(pt_uniform_ptr(rec)) : "memory");
    }
#else
    info = *static_cast<const pt_u32x4*>(info_ptr);
    pt_u32x8 hrec;
    if (HIER) {
        hrec = *reinterpret_cast<const pt_u32x8*>(sc.hier_rec + 8 * (size_t)node);
        rec = sc.own_inv + 12 * (size_t)node;
    }
    a = *static_cast<const pt_u32x16*>(rec);
    b = *reinterpret_cast<const pt_u32x8*>(static_cast<const char*>(rec) + 64);
#endif
    const uint32_t type = info[0], data = info[1];
    PtRay local;
    {
        double m[12];
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
#pragma unroll
        for (int k = 0; k < 4; k++) m[8 + k] = pt_f64_of(b[2 * k], b[2 * k + 1]);
        if (HIER) local = pt_node_local_ray_rec_own(sc, node, hrec, m, ray, identity_ok);
        else local = pt_ray_to_local(m, ray);
    }
    if (STATS) cnt->n_analytic++;
    double t;
    uint32_t part = 0;
    bool hit;
    if (type == PT_TRIANGLE) {  // stand-alone triangle, stored after the mesh triangles
        double beta, gamma;
        if (STATS) cnt->n_tri++;
        hit = pt_triangle_hit(sc.tri_v + 9 * (size_t)data, local, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, node, data), &t, &beta, &gamma);
        part = data;
    } else {
        hit = pt_unit_prim_hit(type, local, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, node, 0), &t, &part);
    }
    if (!hit) return false;
    best.t = t; best.node = node; best.sub = part;
    return true;
}

// Down from inner node `cur` to the next leaf: ONE loop with one way out - `cur` becomes a leaf reference, or PT_REF_EMPTY when
// nothing is left (or the stack overflowed: `overflowed`). Every `return` or `break` in here would cost scalar instructions in
// EVERY step (the compiler turns the exits into flags it sets and tests), and the CU's one scalar unit is what these steps wait
// for. `lanes`: the lanes whose rays count; W words per stack entry (3 in the counting build: the node and the mask of the lanes
// that reach it, which `in` follows).
template <bool STATS, int OCT>
PT_HD void pt_descend(const PtBvhNode* bvh, const PtRayPk& q, float tm, unsigned long long lanes, unsigned long long self, uint32_t& cur, int& sp, unsigned long long& in,
                      bool& overflowed, uint32_t* wstack, int words, uint32_t& pops, PtCounters* cnt) {
    constexpr int W = STATS ? 3 : 1;
#if !defined(PT_STEP_ONE_BLOCK) && defined(__HIP_DEVICE_COMPILE__)  // (-DPT_STEP_ONE_BLOCK: round 3's single block for every case, the A/B of profiles/r05/notes.md section 6)
    if (!STATS) {  // (see pt_descend_mesh: the common case - one child reached - in six scalar instructions; vote, push and pop behind branches)
        while (!(cur & PT_REF_LEAF)) {
            const pt_u32x16 v = pt_sload_node(bvh, cur);
            PT_WAVE_COUNT(4);
            unsigned long long m0, m1;
            float tn0, tn1;
            pt_slab_pk2_t<OCT>(v, q, tm, &m0, &m1, &tn0, &tn1);
            uint32_t next, both, f0;
            asm volatile(
                "s_and_b64 %[m0], %[m0], %[lanes]\n\t"
                "s_cselect_b32 %[next], %[c0], %[pop]\n\t"
                "s_cselect_b32 %[f0], 1, 0\n\t"
                "s_and_b64 %[m1], %[m1], %[lanes]\n\t"
                "s_cselect_b32 %[next], %[c1], %[next]\n\t"
                "s_cselect_b32 %[both], %[f0], 0"
                : [next] "=&s"(next), [both] "=&s"(both), [f0] "=&s"(f0), [m0] "+s"(m0), [m1] "+s"(m1)
                : [lanes] "s"(lanes), [c0] "s"(v[12]), [c1] "s"(v[13]), [pop] "s"(PT_REF_POP)
                : "scc");
            if (both) {
                const unsigned long long second_first = m1 & (PT_FCMP_LT(tn1, tn0) | ~m0);
                const bool swap = __builtin_popcountll(second_first) * 2 > __builtin_popcountll(m0 | m1);
                const uint32_t near = swap ? v[13] : v[12], far = swap ? v[12] : v[13];
                if (sp + 1 <= words) { wstack[sp] = far; sp++; next = near; }
                else { overflowed = true; next = PT_REF_EMPTY; }
            } else if (next == PT_REF_POP) {  // neither: the next pending subtree
#ifdef PT_WATCH_MESHFREE_DESCEND
                if (sp > 0 && ++pops <= PT_WALK_POPS_MAX) {
#else
                if (sp > 0) {
#endif
                    sp--;
                    next = PT_UNIFORM_U32(wstack[sp]);
                } else {
                    overflowed = overflowed || sp > 0;
                    next = PT_REF_EMPTY;
                }
            }
            cur = next;
        }
        return;
    }
#endif
    while (!(cur & PT_REF_LEAF)) {
        const pt_u32x16 v = pt_sload_node(bvh, cur);
        PT_WAVE_COUNT(4);
        const unsigned long long mine = STATS ? (lanes & in) : lanes;
        if (STATS && (mine & self)) cnt->n_inner++;
        unsigned long long m0, m1, one_first;
        pt_slab_pk2<OCT>(v, q, tm, &m0, &m1, &one_first);
        uint32_t near, far;
        const uint32_t code = pt_step_decide(&m0, &m1, one_first, mine, v[12], v[13], &near, &far);
        uint32_t next = near;
        if (STATS) in = (code == 3u ? near == v[12] : code == 1u) ? m0 : m1;
        if (code == 3u) {  // both: the other one waits on the stack
            if (sp + W <= words) {
                wstack[sp] = far;
                if (STATS) { const unsigned long long far_mask = far == v[12] ? m0 : m1; wstack[sp + 1] = (uint32_t)far_mask; wstack[sp + 2] = (uint32_t)(far_mask >> 32); }
                sp += W;
            } else {
                overflowed = true; next = PT_REF_EMPTY;
            }
        }
        if (code == 0u) {  // neither: the next pending subtree
            // (the watchdog rides on the pops - a walk that goes on needs them, and they are rare next to the steps: a walk that has taken
            // more pending subtrees than any tree this library accepts has nodes is a defect and ends like a stack overflow)
#ifdef PT_WATCH_MESHFREE_DESCEND  // (measured: the counter inside the mesh-free walk's hand-tuned step costs 3.5 % on big-scene - 15 more spilled scalar
            // registers, profiles/r04/notes.md; that walk keeps the count in its outer loop only, where a leaf has just been tested)
            if (sp > 0 && ++pops <= PT_WALK_POPS_MAX) {
#else
            if (sp > 0) {
#endif
                sp -= W;
                next = PT_UNIFORM_U32(wstack[sp]);
                if (STATS) in = (unsigned long long)PT_UNIFORM_U32(wstack[sp + 1]) | ((unsigned long long)PT_UNIFORM_U32(wstack[sp + 2]) << 32);
            } else {
                overflowed = overflowed || sp > 0;
                next = PT_REF_EMPTY;
            }
        }
        cur = next;
    }
}

// The same inside the two-level walk of scenes with mesh instances: PT_REF_POP when neither child is reached (the caller pops: a
// marker may end a mesh instance), PT_REF_EMPTY when the stack overflowed.
template <bool STATS, int OCT, bool NEAR = false>
PT_HD void pt_descend_mesh(const PtBvhNode* bvh, const PtRayPk& q, float tm, unsigned long long lanes, bool counts, uint32_t& cur, int& sp, uint32_t* wstack, int words,
                           PtCounters* cnt, float t0 = 0.0f) {
#if !defined(PT_STEP_ONE_BLOCK) && defined(__HIP_DEVICE_COMPILE__)  // (-DPT_STEP_ONE_BLOCK: round 3's single block for every case, the A/B of profiles/r05/notes.md section 6)
    {
        // Round 5: the step's scalar side by CASES instead of one straight block of mask arithmetic for all of them. About half of the steps reach one child
        // only (or none); for those the majority vote, the code word and the push logic are never issued - the walk is bound by the scalar unit's issue
        // rate, not by the fetches (profiles/r05/notes.md section 6: two levels per fetch lost 10 %, eight scalar instructions fewer per step won 5 %).
        // Same children, same order, same pushes as the single block (the counting build still runs that one).
        while (!(cur & PT_REF_LEAF)) {
            const pt_u32x16 v = pt_sload_node(bvh, cur);
            PT_WAVE_COUNT(4);
            if (STATS && counts) cnt->n_inner++;
            unsigned long long m0, m1;
            float tn0, tn1;
            pt_slab_pk2_t<OCT, NEAR>(v, q, tm, &m0, &m1, &tn0, &tn1, t0);
            // The scalar side of the common case in six instructions: the masks cut down to the participating lanes, `next` = the one child reached (or
            // PT_REF_POP), `both` = 1 when both are - only then the vote and the push are issued.
            uint32_t next, both, f0;
            asm volatile(
                "s_and_b64 %[m0], %[m0], %[lanes]\n\t"
                "s_cselect_b32 %[next], %[c0], %[pop]\n\t"
                "s_cselect_b32 %[f0], 1, 0\n\t"
                "s_and_b64 %[m1], %[m1], %[lanes]\n\t"
                "s_cselect_b32 %[next], %[c1], %[next]\n\t"
                "s_cselect_b32 %[both], %[f0], 0"
                : [next] "=&s"(next), [both] "=&s"(both), [f0] "=&s"(f0), [m0] "+s"(m0), [m1] "+s"(m1)
                : [lanes] "s"(lanes), [c0] "s"(v[12]), [c1] "s"(v[13]), [pop] "s"(PT_REF_POP)
                : "scc");
            if (both) {  // the one most of the lanes that reach a child enter first (pt_step_decide's vote)
                const unsigned long long second_first = m1 & (PT_FCMP_LT(tn1, tn0) | ~m0);
                const bool swap = __builtin_popcountll(second_first) * 2 > __builtin_popcountll(m0 | m1);
                const uint32_t near = swap ? v[13] : v[12], far = swap ? v[12] : v[13];
#ifdef PT_NO_PUSH_CHECK  // (A/B only: what the overflow test of a push costs)
                wstack[sp] = far; sp++; next = near;
#else
                if (sp + 1 <= words) { wstack[sp] = far; sp++; next = near; }
                else next = PT_REF_EMPTY;
#endif
            }
            cur = next;
        }
        return;
    }
#endif
    while (!(cur & PT_REF_LEAF)) {
        const pt_u32x16 v = pt_sload_node(bvh, cur);
        PT_WAVE_COUNT(4);
        if (STATS && counts) cnt->n_inner++;
        unsigned long long m0, m1, one_first;
        pt_slab_pk2<OCT, NEAR>(v, q, tm, &m0, &m1, &one_first, t0);
        uint32_t near, far;
        const uint32_t code = pt_step_decide(&m0, &m1, one_first, lanes, v[12], v[13], &near, &far);
        uint32_t next = near;
        if (code == 3u) {
            if (sp + 1 <= words) { wstack[sp] = far; sp++; }
            else next = PT_REF_EMPTY;
        }
        if (code == 0u) next = PT_REF_POP;
        cur = next;
    }
}

// wstack: the wavefront's own stack in LDS, `wwords` 32-bit words, linear.
template <bool STATS, bool HIER>
PT_HD void pt_trace_packet(const PtSceneView& sc, const PtRay& ray, bool has_ray, bool any, PtHit& best, uint32_t* wstack, int wwords,
                           unsigned int* overflow, PtCounters* cnt) {
    if (has_ray) { best.t = INFINITY; best.node = PT_NO_HIT; best.sub = 0; }
    if (sc.n_nodes == 0 || sc.tlas_root == PT_REF_EMPTY) return;
    const unsigned long long self = 1ull << PT_LANE_ID();
    bool alive = has_ray;               // the lane still wants candidates (a shadow ray stops at its first hit)
    unsigned long long amask = PT_BALLOT(alive);  // the same as a wave-uniform mask: the slab test's results are masks, never per-lane booleans
    unsigned long long in = ~0ull;      // wave-uniform: lanes whose rays reach the current node's box
    int oct;
    const PtRayPk q = pt_raypk(ray, has_ray, &oct);
    const bool identity_ok = HIER && pt_ray_identity_safe(ray, has_ray);  // (wave-uniform) levels of a path that are the identity may be skipped
    float tm = INFINITY;                 // best.t as the f32 bound of the slab test, rounded up; follows best.t
    uint32_t cur = PT_UNIFORM_U32(sc.tlas_root);
    int sp = 0;                          // words on the stack
    constexpr int W = STATS ? 3 : 1;     // per entry: the node, and in the counting build the mask of the lanes that reach it
    const int words = wwords < W * sc.stack_cap ? wwords : W * sc.stack_cap;  // scene.stack_cap entries, if the LDS region holds them
    uint32_t pops = 0;                   // (watchdog: pending subtrees taken by this walk, see pt_descend)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(pops));       // (kept in a scalar register: left to itself the compiler parks it in a vector register's lane and spills four more of those)
#endif
    for (;;) {
        bool overflowed = false;         // the stack overflowed or the watchdog tripped inside pt_descend
        if (!(cur & PT_REF_LEAF)) {
            const unsigned long long mine = amask;
            switch (STATS ? PT_OCT_MIXED : oct) {  // the counting build always takes the per-lane form (its images are compared with the plain build's)
            case 0: pt_descend<STATS, 0>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 1: pt_descend<STATS, 1>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 2: pt_descend<STATS, 2>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 3: pt_descend<STATS, 3>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 4: pt_descend<STATS, 4>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 5: pt_descend<STATS, 5>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 6: pt_descend<STATS, 6>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            case 7: pt_descend<STATS, 7>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            default: pt_descend<STATS, PT_OCT_MIXED>(sc.bvh, q, tm, mine, self, cur, sp, in, overflowed, wstack, words, pops, cnt); break;
            }
        }
        if (cur == PT_REF_EMPTY) {
            if (overflowed) {
#if defined(__HIP_DEVICE_COMPILE__)
                if (overflow) atomicOr(overflow, pops > PT_WALK_POPS_MAX ? 4u : 1u);  // 4: the watchdog
#endif
                if (STATS) cnt->stack_overflow++;
                if (has_ray) best.node = PT_NO_HIT;
            }
            return;
        }
        const uint32_t first = (cur & ~PT_REF_LEAF) >> 3, count = (cur & 7u) + 1u;
        PT_WAVE_COUNT(5);
        {
        PT_CYC_BEGIN();
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t node = sc.tlas_direct ? first : PT_UNIFORM_U32(sc.bvh_items[first + i]);
            if (alive && (!STATS || (in & self))) {
                if (STATS) cnt->n_leaf++;
                if (pt_test_node_uniform<STATS, HIER>(sc, node, ray, identity_ok, best, cnt)) {
                    tm = pt_tmax32(best.t);
                    if (any) alive = false;
                }
            }
        }
        PT_CYC_END(5);
        }
        amask = PT_BALLOT(alive);
        if (!amask) return;
        if (sp == 0) return;
        sp -= W;
        cur = PT_UNIFORM_U32(wstack[sp]);
        if (STATS) in = (unsigned long long)PT_UNIFORM_U32(wstack[sp + 1]) | ((unsigned long long)PT_UNIFORM_U32(wstack[sp + 2]) << 32);
#ifdef PT_WATCH_MESHFREE_WALK  // Off by default in THIS walk (on in the walks of scenes with meshes and in the k-d walk, where it measures nothing): three scalar
        // instructions per pending subtree cost the headline 1.1 % (c16 / c22), and what they buy is little - a walk of a valid tree ends by itself (a node is
        // entered at most once, pushes are bounded by the stack), and the one hang this code base has seen was not in a walk (profiles/r04/notes.md).
        if (++pops > PT_WALK_POPS_MAX) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (overflow) atomicOr(overflow, 4u);
#endif
            if (STATS) cnt->stack_overflow++;
            if (has_ray) best.node = PT_NO_HIT;
            return;
        }
#endif
    }
}

// A wave-uniform pointer pinned to a scalar register pair (see pt_args_again: what is read through the re-read argument block would
// otherwise be fetched again - one more dependent scalar load - in front of every use inside a loop).
PT_HD const void* pt_pin_ptr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long a = (unsigned long long)pt_uniform_ptr(p);
    asm volatile("" : "+s"(a));
    return (const void*)a;
#else
    return p;
#endif
}
// The walk of ONE mesh instance's triangle tree (round 5), compiled once per octant of the participating lanes' rays: descents, triangle leaves and the
// pops between them stay inside the specialised code until the instance's part of the stack is used up (`sp` back at its value on entry - no marker entry, no
// marker test per pop, no re-dispatch on the octant per descent). Returns 0: instance finished, 1: stack overflow / watchdog, 2: no lane wants candidates any more.
template <bool STATS, bool HIER, int OCT>
PT_HD int pt_walk_instance(const PtSceneView& sc, uint32_t inst, uint32_t root, const PtRay& local, const PtRayPk& q, bool part, bool any, bool& alive, PtHit& best, float& tm,
                           uint32_t* wstack, int& sp, int words, uint32_t& pops, PtCounters* cnt) {
    const int sp0 = sp;
    uint32_t cur = root;
    unsigned long long pmask = PT_BALLOT(alive && part);
    // (the tree's base address pinned to a scalar register pair: read through the re-read argument block it is a value the compiler may fetch again wherever it is
    // short of registers - and it did, in front of every node fetch of every step: one more dependent round trip through the scalar cache per step)
    const PtBvhNode* const bvh = static_cast<const PtBvhNode*>(pt_pin_ptr(sc.bvh));
#if !defined(PT_TRI_VIA_ITEMS) && !defined(PT_NO_TRI_EDGES)
    const double* const tri_leaf = static_cast<const double*>(pt_pin_ptr(sc.tri_leaf));
#endif
    for (;;) {
        if (!(cur & PT_REF_LEAF)) pt_descend_mesh<STATS, OCT>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt);
        if (cur == PT_REF_EMPTY) return 1;
        if (cur != PT_REF_POP) {
            const uint32_t first = (cur & ~PT_REF_LEAF) >> 3, count = (cur & 7u) + 1u;
            PT_WAVE_COUNT(5);
            if (STATS && alive && part) cnt->n_leaf++;
            PT_CYC_BEGIN();
            for (uint32_t i = 0; i < count; i++) {
                pt_u32x16 a;
                uint32_t b0, b1;
                #if !defined(PT_TRI_VIA_ITEMS) && !defined(PT_NO_TRI_EDGES)
                const uint32_t tri = pt_load_leaf_triangle_at(tri_leaf, first + i, a, b0, b1);
#else
                const uint32_t tri = pt_load_leaf_triangle(sc, first + i, a, b0, b1);
#endif
                double tv[9];
#pragma unroll
                for (int k = 0; k < 8; k++) tv[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
                tv[8] = pt_f64_of(b0, b1);
                if (alive && part) {
                    double tt, beta, gamma;
                    if (STATS) cnt->n_tri++;
                    if (PT_TRI_HIT(tv, local, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, inst, tri), &tt, &beta, &gamma)) {
                        best.t = tt; best.node = inst; best.sub = tri;
                        tm = pt_tmax32(tt);
                        if (any) alive = false;
                    }
                }
            }
            PT_CYC_END(5);
            if (!PT_BALLOT(alive)) return 2;
            pmask = PT_BALLOT(alive && part);
        }
        if (sp == sp0) return 0;
#ifndef PT_NO_INSTANCE_WATCHDOG  // (A/B only: what the watchdog costs)
        if (++pops > PT_WALK_POPS_MAX) return 1;
#endif
        sp--;
        cur = PT_UNIFORM_U32(wstack[sp]);
    }
}

// The same for scenes WITH mesh instances (PT_MODE_FLAT, hits spawn no rays): the scene tree and every mesh's triangle tree are
// walked once per wavefront. Entering a mesh instance is wave-uniform too: every lane brings its ray into the instance's space
// with the one inverse transform (scalar operands), the lanes whose rays pass the mesh's exact box test (mesh.rs:146-155) take
// part in its tree, a marker on the stack says where the scene tree continues. Triangle records (nine doubles) arrive through
// the scalar cache like nodes. Two-child nodes throughout (the array the four-child form is made from).
// KDMESH: KDMesh instances keep the reference's own triangle k-d tree (quirk Q3), which every lane walks by itself with its own
// range bookkeeping (pt_kdmesh_hit) on `lane_stk`, a per-lane stack beside the wavefront's. HIER: the hierarchical semantics
// (every SceneNode's own inverse on the way down, ties by depth-first rank).
template <bool STATS, bool KDMESH, bool HIER, class LaneStack>
PT_HD void pt_trace_packet_mesh(const PtSceneView& sc, const PtRay& ray, bool has_ray, bool any, PtHit& best, uint32_t* wstack, int wwords,
                                const LaneStack& lane_stk, unsigned int* overflow, PtCounters* cnt) {
    if (has_ray) { best.t = INFINITY; best.node = PT_NO_HIT; best.sub = 0; }
    if (sc.n_nodes == 0 || sc.tlas_root == PT_REF_EMPTY) return;
    bool alive = has_ray;     // the lane still wants candidates
    bool part = has_ray;      // ... and takes part in the tree being walked (inside a mesh: its ray passed the mesh's box test)
    unsigned long long pmask = PT_BALLOT(alive && part);  // the lanes the slab test's results count for, as a wave-uniform mask
    PtRay local = ray;        // the ray in the space of the tree being walked
    PtRayPk q = pt_raypk(ray);
    int oct = PT_OCT_MIXED;   // wave-uniform: the octant the participating lanes' rays share inside the mesh instance being walked, if they do (sc.mesh_oct)
    const bool identity_ok = HIER && pt_ray_identity_safe(ray, has_ray);  // (wave-uniform) levels of a path that are the identity may be skipped
    float tm = INFINITY;      // best.t as the f32 bound of the slab test (t means the same in every space, ray.rs:130-135)
    uint32_t inst = PT_NO_HIT;  // wave-uniform: flat node of the mesh instance being walked
    uint32_t cur = PT_UNIFORM_U32(sc.tlas_root);
    int sp = 0;
    auto slot = [&](int k) -> uint32_t& { return wstack[k]; };
    const int words = wwords < sc.stack_cap ? wwords : sc.stack_cap;
    uint32_t pops = 0;  // (watchdog: pending subtrees taken by this walk; one that takes more than any tree has nodes ends like a stack overflow)
    // (the scene-level steps' base address: pinned in the instantiations with KDMesh trees, whose walks are mostly scene-level steps - the dielectric workload +1.0 %, its hierarchical form +1.6 %; left to
    // the compiler in the others, where the pair of registers costs the mirror scene 0.4 % - c55)
    const PtBvhNode* const bvh = KDMESH ? static_cast<const PtBvhNode*>(pt_pin_ptr(sc.bvh)) : sc.bvh;
#if defined(PT_PIN_SCENE_PTRS)  // (A/B: the arrays a mesh instance is entered through, held in scalar registers for the whole walk instead of re-read from the argument block at every scene-level leaf)
    const uint32_t* const info_base = static_cast<const uint32_t*>(pt_pin_ptr(sc.info));
    const double* const inv_base = static_cast<const double*>(pt_pin_ptr(sc.inv));
    const PtMeshInfo* const meshes_base = static_cast<const PtMeshInfo*>(pt_pin_ptr(sc.meshes));
#else
    const uint32_t* const info_base = sc.info;
    const double* const inv_base = sc.inv;
    const PtMeshInfo* const meshes_base = sc.meshes;
#endif
    auto overflowed = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        if (overflow) atomicOr(overflow, pops > PT_WALK_POPS_MAX ? 4u : 1u);  // 4: the watchdog
#endif
        if (STATS) cnt->stack_overflow++;
        if (has_ray) best.node = PT_NO_HIT;
    };
    for (;;) {
        // down to the next leaf (pt_descend_mesh): a leaf reference, PT_REF_POP when neither child is reached, PT_REF_EMPTY when the
        // stack overflowed. Always the per-lane form of the slab test: the walks of these scenes are short (5.7 tree steps per ray on
        // macho-cows, 8.6 on the mirror scene) and the octant bookkeeping per ray and per mesh instance cost more than the eight
        // instructions per step it saves (cows -6 %, mirror -3 %; big-scene, mesh-free, +4 %: profiles/r03/notes.md).
        // Round 4 (sc.mesh_oct): INSIDE a mesh instance whose lanes' rays share their direction signs the sorted form runs (`oct`, worked out once per
        // instance entered; the scene-level steps keep the per-lane form): 37 steps per ray on the 1.25 M-triangle soup +6 %, the short walks no longer lose.
        if (!(cur & PT_REF_LEAF)) {
            switch (STATS ? PT_OCT_MIXED : oct) {
            case 0: pt_descend_mesh<STATS, 0>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 1: pt_descend_mesh<STATS, 1>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 2: pt_descend_mesh<STATS, 2>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 3: pt_descend_mesh<STATS, 3>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 4: pt_descend_mesh<STATS, 4>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 5: pt_descend_mesh<STATS, 5>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 6: pt_descend_mesh<STATS, 6>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            case 7: pt_descend_mesh<STATS, 7>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            default: pt_descend_mesh<STATS, PT_OCT_MIXED>(bvh, q, tm, pmask, alive && part, cur, sp, wstack, words, cnt); break;
            }
        }
        if (cur == PT_REF_EMPTY) { overflowed(); return; }
        const bool popped = cur == PT_REF_POP;
        if (!popped) {
            const uint32_t first = (cur & ~PT_REF_LEAF) >> 3, count = (cur & 7u) + 1u;
            PT_WAVE_COUNT(5);
            if (STATS && alive && part) cnt->n_leaf++;
            if (inst != PT_NO_HIT) {  // triangles of the mesh being walked
                PT_CYC_BEGIN();
                for (uint32_t i = 0; i < count; i++) {
                    pt_u32x16 a;
                    uint32_t b0, b1;
                    const uint32_t tri = pt_load_leaf_triangle(sc, first + i, a, b0, b1);
                    double tv[9];
#pragma unroll
                    for (int k = 0; k < 8; k++) tv[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
                    tv[8] = pt_f64_of(b0, b1);
                    if (alive && part) {
                        double tt, beta, gamma;
                        if (STATS) cnt->n_tri++;
                        if (PT_TRI_HIT(tv, local, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, inst, tri), &tt, &beta, &gamma)) {
                            best.t = tt; best.node = inst; best.sub = tri;
                            tm = pt_tmax32(tt);
                            if (any) alive = false;
                        }
                    }
                }
                PT_CYC_END(5);
            } else {
                bool entered = false;
                for (uint32_t i = 0; i < count && !entered; i++) {
                    const uint32_t node = sc.tlas_direct ? first : PT_UNIFORM_U32(sc.bvh_items[first + i]);
                    // (fetches that depend on each other cost a round trip through the scalar cache each - a mesh instance is entered with three: the
                    // node's {type, data, ..}, its inverse, the mesh record's {box inverse, roots} - not with one per field: round 4, c47)
                    const pt_u32x4 info4 = pt_sload4(info_base + 4 * (size_t)node);
                    const uint32_t type = info4[0];
                    if (type == PT_MESH || type == PT_KDMESH) {
                        const uint32_t data = info4[1];
                        const PtMeshInfo* mi = meshes_base + data;
                        PtRay lr;
                        if (HIER) {
                            lr = pt_node_local_ray_uniform(sc, node, ray, identity_ok);
                        } else {
                            double m[12];
                            pt_sload_mat12(inv_base + 12 * (size_t)node, m);
                            lr = pt_ray_to_local(m, ray);
                        }
                        if (STATS && alive) cnt->n_analytic++;
                        double bi[12];
                        pt_u32x4 head;  // {tri_first, tri_count, blas_root, kd_root}
                        pt_sload_mat12_x4(mi->bbox_inv, bi, head);
                        if (KDMESH && type == PT_KDMESH && (int32_t)head[3] >= 0) {  // the reference's own triangle tree (quirk Q3)
                            if (alive) {
                                double t; uint32_t tri = 0;
                                if (pt_kdmesh_hit<STATS>(sc, *mi, lr, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, node, 0), lane_stk, 0, &t, &tri, cnt)) {
                                    best.t = t; best.node = node; best.sub = tri;
                                    tm = pt_tmax32(t);
                                    if (any) alive = false;
                                }
                            }
                            continue;
                        }
                        // mesh.rs:146-155: box test, then the triangles (also a KDMesh without a tree of its own: PORTRAYER_KDMESH_AS_MESH)
                        const uint32_t root = head[2];
                        if (STATS && alive) cnt->n_bbox++;
                        if (root == PT_REF_EMPTY) continue;
                        const bool inside = alive && pt_bbox_test_hit(bi, lr, PT_EPSILON, pt_cand_end_in<HIER>(sc, best, node, 0));
                        const unsigned long long inside_mask = PT_BALLOT(inside);
                        if (!inside_mask) continue;
#ifndef PT_MESH_MARKER_WALK
                        {   // the instance's tree, in the code of its octant (pt_walk_instance); the scene-level walk goes on with this leaf's next node afterwards
                            int io = PT_OCT_MIXED;
#ifdef PT_MESH_ONE_INSTANTIATION  // (A/B: what the eight octant copies of the instance walk cost the scenes with short walks by being in the kernel)
                            const PtRayPk qi = pt_raypk(lr);
#else
                            const PtRayPk qi = (sc.mesh_oct && !STATS) ? pt_raypk(lr, inside, &io) : pt_raypk(lr);
#endif
                            int rc;
                            switch (io) {
                            case 0: rc = pt_walk_instance<STATS, HIER, 0>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 1: rc = pt_walk_instance<STATS, HIER, 1>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 2: rc = pt_walk_instance<STATS, HIER, 2>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 3: rc = pt_walk_instance<STATS, HIER, 3>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 4: rc = pt_walk_instance<STATS, HIER, 4>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 5: rc = pt_walk_instance<STATS, HIER, 5>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 6: rc = pt_walk_instance<STATS, HIER, 6>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            case 7: rc = pt_walk_instance<STATS, HIER, 7>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            default: rc = pt_walk_instance<STATS, HIER, PT_OCT_MIXED>(sc, node, root, lr, qi, inside, any, alive, best, tm, wstack, sp, words, pops, cnt); break;
                            }
                            if (rc == 1) { overflowed(); return; }
                            if (rc == 2) return;
                        }
#else
                        if (sp + 2 > words) { overflowed(); return; }
                        if (i + 1 < count) { slot(sp) = PT_REF_LEAF | ((first + i + 1) << 3) | (count - i - 2); sp++; }  // the rest of this leaf
                        slot(sp) = PT_REF_MARKER; sp++;
                        local = lr; part = inside; inst = node;
                        if (sc.mesh_oct) q = pt_raypk(lr, inside, &oct);
                        else q = pt_raypk(lr);
                        cur = root;
                        entered = true;
#endif
                    } else if (alive) {
                        if (pt_test_node_uniform<STATS, HIER>(sc, node, ray, identity_ok, best, cnt)) {
                            tm = pt_tmax32(best.t);
                            if (any) alive = false;
                        }
                    }
                }
                if (entered) { pmask = PT_BALLOT(alive && part); continue; }
            }
            if (!PT_BALLOT(alive)) return;
            pmask = PT_BALLOT(alive && part);
        }
        // the next pending subtree; a marker ends the walk of a mesh instance
        for (;;) {
            if (sp == 0) return;
            if (++pops > PT_WALK_POPS_MAX) { overflowed(); return; }
            sp--;
            cur = PT_UNIFORM_U32(slot(sp));
            if (cur != PT_REF_MARKER) break;
            inst = PT_NO_HIT; local = ray; part = has_ray;
            q = pt_raypk(ray); oct = PT_OCT_MIXED;
            pmask = PT_BALLOT(alive && part);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ONE walk per wavefront in the reference's k-d tree semantics (round 4; kdtree/node.rs:66-203).
//
// What the reference computes for a ray: the leaves on its path in front-to-back order, each with the range [start, end) the planes
// of the straddled splits above it leave (node.rs:139-186), and the hit of the FIRST leaf that reports one (node.rs:153-157) - a
// leaf's hit being the nearest in ITS range with the leaf's list order breaking exact ties (ray.rs:87-99). The ranges of a ray's
// leaves are disjoint half-open intervals in path order, so "the first leaf with a hit" is "the leaf hit with the smallest t", and a
// leaf's hit depends on nothing but the leaf and its range: the leaves of a ray may be visited in ANY order as long as each is
// tested with the range the reference would give it. (Cone / Cylinder results depend on the range start - quirk Q1, cone.rs:64-76 -,
// which is why the ranges must be the reference's own numbers: every plane parameter below is computed by the reference's
// expression, (plane - o) / d in f64, from the reference's side tests.)
//
// That makes the walk a change of schedule like pt_trace_packet: the NODE and the stack are wave-uniform (a node = 64 bytes through
// the scalar cache, planes and cull boxes as scalar operands), every lane carries its own (start, end), and a split sends each lane's
// range where the reference sends it:
//   * both ends of the lane's classification segment on one side (node.rs:128-137): that child, range unchanged;
//   * straddling (node.rs:139-186): its near side gets [start, plane_t), its far side [plane_t, end);
//   * plane_t outside the range (node.rs:146-147 panics): neither (counted as kd_plane_miss).
// The wavefront enters the child most of its lanes call near first; the other waits on the wavefront's stack. Lanes skip nodes that
// cannot improve their result (start > best.t: every hit below has t >= start).
//
// Where a lane's range lives: (start, end) of the CURRENT node in registers; entering a child replaces at most ONE bound by plane_t,
// and the bound it replaces goes to the lane's slot of that LEVEL (`sav`: one f64 per lane and tree level - LDS for the deepest
// levels, which the walk toggles between, HBM for the top ones). Two bits per level and lane (`codes`) say which bound the slot
// holds: 2 = the END to put back, 3 = the START to put back, 1 = the lane waits for the second child with its range unchanged,
// 0 = nothing. Popping the stack entry (second child of level L) puts back the bounds of the finished levels below L one by one,
// then turns the first child's range into the second's: for a lane with [start, plane_t) the second child gets [plane_t, saved end),
// and the slot now keeps `start` for the way back up; mirror-inverted for a lane that went far side first.
// Nothing here is kept per stack ENTRY and lane (that would be 16 bytes x 64 lanes x depth of LDS per wavefront).
// ------------------------------------------------------------------------------------------------
// ---- Lane predicates as 64-bit MASKS in scalar registers. A compare writes its result there anyway (v_cmp ... -> an SGPR pair), mask
// logic is scalar arithmetic, "any lane?" is a scalar compare with zero, and a select reads the mask back as its condition
// (inverse ballot = a plain copy). Written with `bool`s and __ballot() the compiler materialises every predicate that is not itself a
// compare as 0 / 1 in a vector register and compares that again - two vector instructions per ballot, about sixteen per tree step.
// (The masks cover the lanes that are ACTIVE where the compare executes: use them in wave-uniform control flow only.)
typedef unsigned long long pt_mask;
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_MASK_ALL (~0ull)
#define PT_F64_LT(a, b) __builtin_amdgcn_fcmp((double)(a), (double)(b), 4)   // ordered <
#define PT_F64_LE(a, b) __builtin_amdgcn_fcmp((double)(a), (double)(b), 5)   // ordered <=
#define PT_F64_GE(a, b) __builtin_amdgcn_fcmp((double)(a), (double)(b), 3)   // ordered >=
#define PT_F32_GT(a, b) __builtin_amdgcn_fcmpf((float)(a), (float)(b), 2)    // ordered >
#define PT_U32_EQ(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 32)
#define PT_U32_GE(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 35)
#define PT_U32_LE(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 37)
#define PT_LANES(m) __builtin_amdgcn_inverse_ballot_w64(m)
#define PT_POPC(m) __builtin_popcountll(m)
#else
#define PT_MASK_ALL 1ull
#define PT_F64_LT(a, b) (((double)(a) < (double)(b)) ? 1ull : 0ull)
#define PT_F64_LE(a, b) (((double)(a) <= (double)(b)) ? 1ull : 0ull)
#define PT_F64_GE(a, b) (((double)(a) >= (double)(b)) ? 1ull : 0ull)
#define PT_F32_GT(a, b) (((float)(a) > (float)(b)) ? 1ull : 0ull)
#define PT_U32_EQ(a, b) (((unsigned)(a) == (unsigned)(b)) ? 1ull : 0ull)
#define PT_U32_GE(a, b) (((unsigned)(a) >= (unsigned)(b)) ? 1ull : 0ull)
#define PT_U32_LE(a, b) (((unsigned)(a) <= (unsigned)(b)) ? 1ull : 0ull)
#define PT_LANES(m) ((m) != 0ull)
#define PT_POPC(m) ((int)((m) & 1ull))
#endif
#define PT_MNOT(m) (~(m) & PT_MASK_ALL)
// pt_in_range as a mask: start <= t && t < end (false for NaN)
PT_HD pt_mask pt_in_range_m(double start, double end, double t) { return PT_F64_LE(start, t) & PT_F64_LT(t, end); }
// pt_slab_seg_pk as a mask: the lanes whose segment [tmin, tmax] reaches the box
PT_HD pt_mask pt_slab_seg_pk_m(const float* lo, const float* hi, const PtRayPk& q, float tmin, float tmax) {
    const float ax = __builtin_fmaf(lo[0], q.a[0].x, q.a[0].y), bx = __builtin_fmaf(hi[0], q.b[0].x, q.b[0].y);
    const float ay = __builtin_fmaf(lo[1], q.a[1].x, q.a[1].y), by = __builtin_fmaf(hi[1], q.b[1].x, q.b[1].y);
    const float az = __builtin_fmaf(lo[2], q.a[2].x, q.a[2].y), bz = __builtin_fmaf(hi[2], q.b[2].x, q.b[2].y);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return PT_MNOT(PT_F32_GT(tn, tf));
}
PT_HD pt_mask pt_div_exp_ok_m(double x) {
    union { double d; uint32_t u[2]; } c; c.d = x;
    return PT_U32_LE(((c.u[1] >> 20) & 0x7FFu) - 640u, 767u);
}

// "does any lane ...": the ballot as a 64-bit scalar compared with zero on the SCALAR unit. Written as `if (__ballot(x))` the compiler
// materialises the predicate as 0 / 1 in a vector register and compares that again (two vector instructions per wave-uniform branch).
PT_HD bool pt_any(bool b) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long m = __ballot(b);
    asm volatile("" : "+s"(m));
    return m != 0ull;
#else
    return b;
#endif
}
// 8 dwords at base + byte offset (32 bits): one scalar load, no 64-bit address arithmetic
PT_HD pt_u32x8 pt_sload8_off(const void* base, uint32_t byte_off) {
    pt_u32x8 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(byte_off) : "memory");
#else
    v = *reinterpret_cast<const pt_u32x8*>(static_cast<const char*>(base) + byte_off);
#endif
    return v;
}

PT_HD pt_u32x16 pt_sload16_off(const void* base, uint32_t byte_off) {
    pt_u32x16 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(byte_off) : "memory");
#else
    v = *reinterpret_cast<const pt_u32x16*>(static_cast<const char*>(base) + byte_off);
#endif
    return v;
}

// (pt_rcp_refined / pt_div_exp_ok / pt_div_fast - the short f64 division by a repeated denominator - live in pt_math.h)

// The lanes' replaced range bounds per tree level (see pt_trace_packet_kd). Round 5: only the DEEP levels keep a slot (in LDS); a bound replaced at one of
// the `top_levels` levels nearest the root is not stored at all - it is recomputed when the walk comes back to that level:
//   at any node, a lane's `end` is the plane parameter of the DEEPEST level above it whose code says "straddled, now in the near child" (code 2), or
//   infinity; its `start` is that of the deepest level with code 3 ("straddled, now in the far child"), or epsilon - the codes (two bits per level, in a
//   register) and the planes of the current path determine the range completely. The path is the same for every lane of the wavefront, so the planes of
//   its top levels sit in a small table in the wavefront's LDS (`path`: plane lo, plane hi, axis per level), and a plane parameter is recomputed with the very
//   expression that produced it, (plane - o) / d: the same bits.
// (Round 4 stored the top levels' bounds in the lanes' HBM columns: 16 bytes per lane, level and ray written and read back - 30 GB per big-scene frame.)
struct PtKdSav {
    uint32_t* lds;        // the wavefront's rows for saved bounds: level k >= top_levels in slot k - top_levels; slot j, half h of lane l at lds[(2 j + h) * 64 + l]
    int top_levels;       // levels [0, top_levels) keep no slot: recomputed from `path`
    uint32_t* path;       // the wavefront's table of the current path's top levels: path[3 k] / [3 k + 1] the plane (f64 halves), path[3 k + 2] the axis
};
PT_HD void pt_kd_sav_store(const PtKdSav& s, int level, double v) {  // `level` is wave-uniform and >= s.top_levels
    union { double d; uint32_t u[2]; } c; c.d = v;
    uint32_t* p = s.lds + (size_t)(2 * (level - s.top_levels)) * 64 + PT_LANE_ID();
    p[0] = c.u[0]; p[64] = c.u[1];
}
PT_HD double pt_kd_sav_load(const PtKdSav& s, int level) {
    union { double d; uint32_t u[2]; } c;
    const uint32_t* p = s.lds + (size_t)(2 * (level - s.top_levels)) * 64 + PT_LANE_ID();
    c.u[0] = p[0]; c.u[1] = p[64];
    return c.d;
}
#ifndef PT_KD_WALK_STEPS_MAX
#define PT_KD_WALK_STEPS_MAX (1u << 26)  // more nodes than a k-d tree of 2^26 nodes (the limit of the 32-bit byte offsets) has
#endif
#define PT_KD_WAVE_LEVELS 32  // two bits per level in a 64-bit word per lane. pt_scene_upload REFUSES deeper trees (PT_ERR_SCENE; include/portrayer_hip.h says so):
                              // the render kernels have no other k-d walk (the per-lane PtKdWalker serves pt_test_cast_rays and -DPT_NO_PACKET builds only)

// Mesh::ray_hit below a k-d leaf (mesh.rs:146-167) for the lanes in `part`: the instance's triangle tree walked once per wavefront
// like pt_trace_packet_mesh does, every triangle tested over [start, end of the lane's leaf fold) - nearest triangle, lowest index on
// exact ties (pt_cand_end), which is what the reference's fold over ALL triangles with a shrinking range gives. `wstack`: free words
// of the wavefront's stack. Returns false when they ran out.
template <bool STATS, int OCT, bool BIG>
PT_HD bool pt_packet_mesh_below_kd_oct(const PtSceneView& sc, uint32_t inst, uint32_t root, const PtRay& local, const PtRayPk& q, bool part, double start, bool any, PtHit& lb, bool& found,
                                       uint32_t* wstack, int words, PtCounters* cnt) {
    float tm = pt_tmax32(lb.t);
    // the lane's range starts at `start` (the k-d leaf's): boxes that end before it hold no hit of this leaf - a mesh that spans several k-d leaves
    // is walked once per leaf, each time only where the leaf's range reaches (round 4). Rounded down like the k-d walk's own segment culls.
#ifdef PT_KD_MESH_NO_NEAR
    const float t0 = 0.0f;
#else
    float t0 = (float)start;
    t0 = t0 - fabsf(t0) * 2.4e-7f;
#endif
    unsigned long long pmask = PT_BALLOT(part);
    uint32_t cur = root;
    int sp = 0;
    uint32_t steps = 0;  // (watchdog, as in pt_trace_packet_kd: leaves visited)
    // (see pt_walk_instance; only where the walk is long enough to earn the two fetches back - this function runs once per k-d leaf an instance spans, and pinned for the mirror
    // scene's small meshes the k-d frame was 7 % slower: 41.1 -> 44.3 ms, c62)
    const PtBvhNode* const bvh = BIG ? static_cast<const PtBvhNode*>(pt_pin_ptr(sc.bvh)) : sc.bvh;
#if !defined(PT_TRI_VIA_ITEMS) && !defined(PT_NO_TRI_EDGES)
    const double* const tri_leaf = BIG ? static_cast<const double*>(pt_pin_ptr(sc.tri_leaf)) : sc.tri_leaf;
#endif
    for (;;) {
        if (!(cur & PT_REF_LEAF)) pt_descend_mesh<STATS, OCT, true>(bvh, q, tm, pmask, part, cur, sp, wstack, words, cnt, t0);
        steps++;
        if (cur == PT_REF_EMPTY || steps > PT_KD_WALK_STEPS_MAX) return false;
        if (cur != PT_REF_POP) {
            const uint32_t first = (cur & ~PT_REF_LEAF) >> 3, count = (cur & 7u) + 1u;
            if (STATS && part) cnt->n_leaf++;
            for (uint32_t i = 0; i < count; i++) {
                pt_u32x16 a;
                uint32_t b0, b1;
                #if !defined(PT_TRI_VIA_ITEMS) && !defined(PT_NO_TRI_EDGES)
                const uint32_t tri = pt_load_leaf_triangle_at(tri_leaf, first + i, a, b0, b1);
#else
                const uint32_t tri = pt_load_leaf_triangle(sc, first + i, a, b0, b1);
#endif
                double tv[9];
#pragma unroll
                for (int k = 0; k < 8; k++) tv[k] = pt_f64_of(a[2 * k], a[2 * k + 1]);
                tv[8] = pt_f64_of(b0, b1);
                if (part) {
                    double tt, beta, gamma;
                    if (STATS) cnt->n_tri++;
                    if (PT_TRI_HIT(tv, local, start, pt_cand_end(lb, inst, tri), &tt, &beta, &gamma)) {
                        lb.t = tt; lb.node = inst; lb.sub = tri;
                        found = true;
                        tm = pt_tmax32(tt);
                        if (any) part = false;  // a shadow ray only asks whether anything is in the way (material.rs:174-179)
                    }
                }
            }
            pmask = PT_BALLOT(part);
            if (!pmask) return true;
        }
        if (sp == 0) return true;
        sp--;
        cur = PT_UNIFORM_U32(wstack[sp]);
    }
}

// (round 5: in the code of the octant the participating lanes' rays share inside the instance, like pt_walk_instance - worked out once per call)
template <bool STATS>
PT_HD bool pt_packet_mesh_below_kd(const PtSceneView& sc, uint32_t inst, uint32_t root, uint32_t tri_count, const PtRay& local, bool part, double start, bool any, PtHit& lb, bool& found,
                                   uint32_t* wstack, int words, PtCounters* cnt) {
    // (only for trees deep enough to earn the octant's bookkeeping back: the 1.25 M-triangle soup +5.4 %; the mirror scene's and macho-cows' few-thousand-triangle
    // meshes, walked once per k-d leaf they span, lose 6.6 % / 3.7 % with it - c49)
#ifndef PT_KD_BELOW_OCT
    // ONE instantiation, the per-lane form. The octant instantiations (-DPT_KD_BELOW_OCT: taken by instances of >= 65,536 triangles) make the 1.25 M-triangle soup's k-d frame
    // 8 - 10 % faster - and, by being in the kernel at all, the mirror scene's 6.4 % and macho-cows' 1.7 % slower (registers, code size): c64. The reference's own scenes decide.
    (void)tri_count;
    return pt_packet_mesh_below_kd_oct<STATS, PT_OCT_MIXED, false>(sc, inst, root, local, pt_raypk(local), part, start, any, lb, found, wstack, words, cnt);
#else
    int oct = PT_OCT_MIXED;
    const PtRayPk q = (sc.mesh_oct && !STATS && tri_count >= 65536u) ? pt_raypk(local, part, &oct) : pt_raypk(local);
    switch (oct) {
    case 0: return pt_packet_mesh_below_kd_oct<STATS, 0, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 1: return pt_packet_mesh_below_kd_oct<STATS, 1, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 2: return pt_packet_mesh_below_kd_oct<STATS, 2, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 3: return pt_packet_mesh_below_kd_oct<STATS, 3, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 4: return pt_packet_mesh_below_kd_oct<STATS, 4, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 5: return pt_packet_mesh_below_kd_oct<STATS, 5, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 6: return pt_packet_mesh_below_kd_oct<STATS, 6, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    case 7: return pt_packet_mesh_below_kd_oct<STATS, 7, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    default:
        if (tri_count >= 65536u) return pt_packet_mesh_below_kd_oct<STATS, PT_OCT_MIXED, true>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
        return pt_packet_mesh_below_kd_oct<STATS, PT_OCT_MIXED, false>(sc, inst, root, local, q, part, start, any, lb, found, wstack, words, cnt);
    }
#endif
}

// One split of the k-d walk for the lanes in `mine` (node.rs:112-186): which side the ends of the lane's classification segment are on
// (s, e: Front), who crosses the plane, where (plane_t = (plane - o) / d, node.rs:90-109) and who straddles it inside the range.
PT_HD void pt_kd_split_eval(double o, double d, double y, pt_mask d_ok, double plane, double t_min, double t_max, double start, double end, pt_mask mine,
                            pt_mask* s_out, pt_mask* e_out, pt_mask* cross_out, pt_mask* strad_out, double* plane_t_out) {
    const pt_mask s = PT_F64_GE((o + d * t_min) - plane, 0.0);  // infinite_plane.rs:27-35: Front
    const pt_mask e = PT_F64_GE((o + d * t_max) - plane, 0.0);
    const pt_mask cross = mine & (s ^ e);
    double plane_t = 0.0;
    pt_mask strad = 0ull;
    if (cross) {
        const double n = plane - o;
        const pt_mask fast = d_ok & pt_div_exp_ok_m(n);
        if (cross & PT_MNOT(fast)) plane_t = n / d;   // some crossing lane's operands are outside the short division's window
#ifdef PT_KD_RCP_ON_DEMAND
        else plane_t = pt_div_fast(n, d, pt_rcp_refined(d));
#else
        else plane_t = pt_div_fast(n, d, y);
#endif
        strad = cross & pt_in_range_m(start, end, plane_t);
    }
    *s_out = s; *e_out = e; *cross_out = cross; *strad_out = strad; *plane_t_out = plane_t;
}

// plane_t = (plane - o) / d of pt_kd_split_eval for the lanes in `who`, computed again (a pop at a top level: PtKdSav): the short division where every such
// lane's operands are inside its window, else the hardware's - either way the quotient `/` gives, so the very bits the split produced.
PT_HD double pt_kd_plane_t(double o, double d, double y, pt_mask d_ok, double plane, pt_mask who) {
    const double n = plane - o;
    const pt_mask fast = d_ok & pt_div_exp_ok_m(n);
    if (who & PT_MNOT(fast)) return n / d;
#ifdef PT_KD_RCP_ON_DEMAND
    return pt_div_fast(n, d, pt_rcp_refined(d));
#else
    return pt_div_fast(n, d, y);
#endif
}

// wstack: the wavefront's own stack (linear, `wwords` words); sav: the lanes' saved bounds per level; lane_stk: the lanes' own stacks
// (KDMESH only: KDMesh instances keep the reference's triangle k-d tree, which every lane walks by itself, pt_kdmesh_hit).
template <bool STATS, bool MESH, bool KDMESH, class LaneStack>
PT_HD void pt_trace_packet_kd(const PtSceneView& sc, const PtRay& ray_in, bool has_ray, bool any, PtHit& best, uint32_t* wstack, int wwords, const PtKdSav& sav,
                              const LaneStack& lane_stk, unsigned int* overflow, PtCounters* cnt) {
    // (a copy: picking a component by the split's axis from a ray behind a REFERENCE becomes a load from a selected address, and the
    // caller's ray then lives in scratch memory and is indexed there at every split)
    const PtRay ray = ray_in;
    if (has_ray) { best.t = INFINITY; best.node = PT_NO_HIT; best.sub = 0; }
    // lane predicates are masks in scalar registers (see PT_LANES)
    pt_mask alive = PT_BALLOT(has_ray);   // the lanes that still want candidates (a shadow ray stops at its first hit)
    pt_mask in = alive;                   // the lanes taking part in the subtree of `cur`
    const pt_mask any_m = PT_BALLOT(any);
    double start = PT_EPSILON, end = INFINITY;  // ray.rs:140; the lane's range at the current node
    uint32_t codes_lo = 0, codes_hi = 0;        // two bits per level of the current path (see above): levels 0-15 / 16-31
    const PtRayPk q = pt_raypk(ray);
    const double extent = sc.kd_extent;
    // per axis: the refined reciprocal of the direction component and whether the short division may be used with it (pt_div_fast)
#ifdef PT_KD_RCP_ON_DEMAND  // (A/B: the reciprocal refined at every use - five instructions - instead of six registers held for the whole walk; NaN marks "not computed")
    const double yx = __builtin_nan(""), yy = yx, yz = yx;
#else
    const double yx = pt_rcp_refined(ray.d.x), yy = pt_rcp_refined(ray.d.y), yz = pt_rcp_refined(ray.d.z);
#endif
    const pt_mask okx = pt_div_exp_ok_m(ray.d.x), oky = pt_div_exp_ok_m(ray.d.y), okz = pt_div_exp_ok_m(ray.d.z);
    const void* const kd_base = pt_pin_ptr(sc.kd);
    const void* const ref_base = pt_pin_ptr(sc.kd_ref);
    uint32_t cull = PT_UNIFORM_U32((sc.kd_box != nullptr ? 1u : 0u) | (sc.node_box != nullptr ? 2u : 0u));  // (PORTRAYER_KD_NO_CULL clears both)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_KD_CULL_UNPINNED)
    asm volatile("" : "+s"(cull));  // (a scalar the walk branches on: left alone the two flags were re-materialised through the vector unit - v_cndmask, v_cmp - at every node and every leaf reference)
#endif
    uint32_t cur = 0;                 // wave-uniform: node, its level, words on the stack
    int lev = 0, sp = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // (pinned to scalar registers from the start: the constant 0 otherwise reaches the loop's phi as a copy of some VECTOR register that
    // also holds a 0, and the compiler's SGPR-copy fixing then moves the whole chain - and the scalar load's offset - to the vector side)
    asm volatile("" : "+s"(cur), "+s"(lev), "+s"(sp));
#endif
    // ONE way out of the walk (the test at the very end of the body): an early `return` from inside these loops - with the lanes' divergent
    // code of the failure path behind it - makes the compiler's unified loop exit a join of divergent branches, and every wave-uniform value
    // that passes through it (the node index, the stack pointer, the lane masks) is then taken for divergent: no scalar loads, no SALU block.
    // For the same reason the lanes' own updates below are selects, not branches.
    bool failed = false;   // wave-uniform: the wavefront's stack overflowed, or the watchdog below tripped
    uint32_t steps = 0;    // wave-uniform: nodes visited by this walk. A walk visits a node at most once; one that goes on beyond any tree this
                           // library accepts is a defect (or a corrupted stack) and fails the render (PT_ERR_TRAVERSAL) instead of hanging the GPU
#ifndef PT_KD_SEG_PER_NODE
    float seg0c = 0.0f, seg1c = 0.0f;
    if (!MESH) {
        seg0c = (float)start; seg1c = (float)end;
        seg0c = seg0c - fabsf(seg0c) * 2.4e-7f; seg1c = seg1c + fabsf(seg1c) * 2.4e-7f;
    }
#endif
    for (;;) {
        steps++;
        failed = failed || steps > PT_KD_WALK_STEPS_MAX;
        bool descend = false;  // wave-uniform: go on with `cur` (a child) instead of taking the next pending subtree
        // the lanes this node can still give something: taking part, and no hit yet in front of everything below (every hit
        // below has t >= start, and ranges of different leaves never share a t)
        pt_mask mine = failed ? 0ull : (alive & in & PT_MNOT(PT_F64_LT(best.t, start)));
        if (mine) {
            const pt_u32x16 v = pt_sload16_off(kd_base, cur << 6);
            PT_WAVE_COUNT(4);
            // the lane's segment [start, end), rounded outward, against the conservative f32 bounds of everything below this node:
            // a subtree the segment does not reach reports no hit, which is all the reference would find out by walking it
#ifndef PT_KD_SEG_PER_NODE  // (the f32 segment kept across nodes and converted again only where start / end change - a straddled split, a pop: +0.3 %, c70; -DPT_KD_SEG_PER_NODE: per node as before)
            float seg0 = seg0c, seg1 = seg1c;
            if (MESH) {  // (the instantiations with mesh walks below the leaves lose with the two registers held across the walk - mirror k-d -2.8 %, c71 - and convert per node)
                seg0 = (float)start; seg1 = (float)end;
                seg0 = seg0 - fabsf(seg0) * 2.4e-7f; seg1 = seg1 + fabsf(seg1) * 2.4e-7f;
            }
#else
            float seg0 = (float)start, seg1 = (float)end;
            seg0 = seg0 - fabsf(seg0) * 2.4e-7f; seg1 = seg1 + fabsf(seg1) * 2.4e-7f;
#endif
            if (cull & 1u) {
                const float lo[3] = {pt_f32_of(v[8]), pt_f32_of(v[9]), pt_f32_of(v[10])}, hi[3] = {pt_f32_of(v[11]), pt_f32_of(v[12]), pt_f32_of(v[13])};
                const pt_mask reach = pt_slab_seg_pk_m(lo, hi, q, seg0, seg1);
                if (STATS && PT_LANES(mine & PT_MNOT(reach))) cnt->kd_culled++;
                mine &= reach;
            }
            const int axis = (int)v[2];
            if (axis >= 0) {
                // ---- a split (node.rs:112-202), for the lanes in `mine`
                if (STATS && PT_LANES(mine)) cnt->n_inner++;
                const double plane = pt_f64_of(v[0], v[1]);
                double t_max = start + extent;                                   // node.rs:118
                t_max = PT_LANES(pt_in_range_m(start, end, t_max)) ? t_max : end - PT_EPSILON;   // node.rs:121
                const double t_min = start + PT_EPSILON;                         // node.rs:124
                // (the same arithmetic compiled once per axis, behind a scalar branch: picking o, d and the reciprocal by a run-time axis
                // costs twelve selects per split)
                pt_mask s, e, cross, strad;
                double plane_t;
#ifndef PT_KD_AXIS_SELECT
                if (axis == 0) pt_kd_split_eval(ray.o.x, ray.d.x, yx, okx, plane, t_min, t_max, start, end, mine, &s, &e, &cross, &strad, &plane_t);
                else if (axis == 1) pt_kd_split_eval(ray.o.y, ray.d.y, yy, oky, plane, t_min, t_max, start, end, mine, &s, &e, &cross, &strad, &plane_t);
                else pt_kd_split_eval(ray.o.z, ray.d.z, yz, okz, plane, t_min, t_max, start, end, mine, &s, &e, &cross, &strad, &plane_t);
#else
                pt_kd_split_eval(axis == 0 ? ray.o.x : (axis == 1 ? ray.o.y : ray.o.z), axis == 0 ? ray.d.x : (axis == 1 ? ray.d.y : ray.d.z), axis == 0 ? yx : (axis == 1 ? yy : yz),
                                 axis == 0 ? okx : (axis == 1 ? oky : okz), plane, t_min, t_max, start, end, mine, &s, &e, &cross, &strad, &plane_t);
#endif
                if (STATS && PT_LANES(cross & PT_MNOT(strad))) cnt->kd_plane_miss++;  // node.rs:146-147 / :177-178: the reference panics here; a miss for this subtree
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_KD_SPLIT_GENERAL_ONLY)  // (-DPT_KD_SPLIT_GENERAL_ONLY: every split through the general bookkeeping, the A/B of c62)
                // The common case settled in seven scalar instructions (the pattern of pt_descend's step): no lane's segment crosses the plane and every lane is on one
                // side - that child next, nothing waits, no range changes, code 0. `slow_`: everything else, through the general bookkeeping. ONE place assigns the node index.
                uint32_t c_first, slow_, f_;
                pt_mask t_;
                asm volatile(
                    "s_and_b64 %[t], %[mine], %[s]\n\t"
                    "s_cselect_b32 %[next], %[cf], %[cb]\n\t"
                    "s_cselect_b32 %[f], 1, 0\n\t"
                    "s_andn2_b64 %[t], %[mine], %[s]\n\t"
                    "s_cselect_b32 %[slow], %[f], 0\n\t"
                    "s_cmp_lg_u64 %[cross], 0\n\t"
                    "s_cselect_b32 %[slow], 1, %[slow]"
                    : [next] "=&s"(c_first), [slow] "=&s"(slow_), [f] "=&s"(f_), [t] "=&s"(t_)
                    : [mine] "s"(mine), [s] "s"(s), [cross] "s"(cross), [cf] "s"(v[3]), [cb] "s"(v[4])
                    : "scc");
                pt_mask in_first = mine;
                uint32_t code = 0u;
                descend = mine != 0ull && !failed;
                if (slow_) {
#ifndef PT_KD_SLOW_COMPILER  // (the general case's mask algebra in one block of scalar instructions too - c61's block, which lost when every split issued it, issued only where all of it is needed: +0.9 %, c74; -DPT_KD_SLOW_COMPILER: the compiler's version)
                    pt_mask go_f, go_b, in_second, t0_, t1_;
                    uint32_t c_second, ff_, push_, any_, nf_, nb_;
                    asm volatile(
                        "s_xor_b64 %[t0], %[s], %[e]\n\t"
                        "s_andn2_b64 %[t0], %[mine], %[t0]\n\t"
                        "s_and_b64 %[gf], %[t0], %[s]\n\t"
                        "s_or_b64 %[gf], %[gf], %[strad]\n\t"
                        "s_andn2_b64 %[gb], %[t0], %[s]\n\t"
                        "s_or_b64 %[gb], %[gb], %[strad]\n\t"
                        "s_or_b64 %[t0], %[gf], %[gb]\n\t"
                        "s_cselect_b32 %[any], 1, 0\n\t"
                        "s_and_b64 %[t1], %[t0], %[s]\n\t"
                        "s_bcnt1_i32_b64 %[nf], %[t1]\n\t"
                        "s_andn2_b64 %[t1], %[t0], %[s]\n\t"
                        "s_bcnt1_i32_b64 %[nb], %[t1]\n\t"
                        "s_cmp_ge_u32 %[nf], %[nb]\n\t"
                        "s_cselect_b32 %[ff], 1, 0\n\t"
                        "s_cselect_b32 %[c1], %[cf], %[cb]\n\t"
                        "s_cselect_b32 %[c2], %[cb], %[cf]\n\t"
                        "s_cselect_b64 %[i1], %[gf], %[gb]\n\t"
                        "s_cselect_b64 %[i2], %[gb], %[gf]\n\t"
                        "s_cmp_lg_u64 %[i2], 0\n\t"
                        "s_cselect_b32 %[push], 1, 0"
                        : [gf] "=&s"(go_f), [gb] "=&s"(go_b), [i1] "=&s"(in_first), [i2] "=&s"(in_second), [t0] "=&s"(t0_), [t1] "=&s"(t1_), [c1] "=&s"(c_first), [c2] "=&s"(c_second),
                          [ff] "=&s"(ff_), [push] "=&s"(push_), [any] "=&s"(any_), [nf] "=&s"(nf_), [nb] "=&s"(nb_)
                        : [s] "s"(s), [e] "s"(e), [mine] "s"(mine), [strad] "s"(strad), [cf] "s"(v[3]), [cb] "s"(v[4])
                        : "scc");
                    const bool front_first = ff_ != 0u;
                    const bool push = push_ != 0u;
                    const bool room = sp < wwords;
                    if (room && push) wstack[sp] = (c_second << 5) | (uint32_t)lev;
                    failed = failed || (push && !room);
                    descend = any_ != 0u && !failed;
#else
                    const pt_mask same = mine & PT_MNOT(s ^ e);
                    const pt_mask go_f = (same & s) | strad, go_b = (same & PT_MNOT(s)) | strad;
                    const bool front_first = PT_POPC((go_f | go_b) & s) >= PT_POPC((go_f | go_b) & PT_MNOT(s));
                    c_first = front_first ? v[3] : v[4];
                    const uint32_t c_second = front_first ? v[4] : v[3];
                    in_first = front_first ? go_f : go_b;
                    const pt_mask in_second = front_first ? go_b : go_f;
                    const bool push = in_second != 0ull;
                    const bool room = sp < wwords;
                    if (room && push) wstack[sp] = (c_second << 5) | (uint32_t)lev;
                    failed = failed || (push && !room);
                    descend = (go_f | go_b) != 0ull && !failed;
#endif
                    sp += (push && descend) ? 1 : 0;
                    const pt_mask near_first = front_first ? s : PT_MNOT(s);
                    const pt_mask both = (push && descend) ? strad : 0ull;
                    if (lev < sav.top_levels) {
                        if (push && descend) { sav.path[3 * lev] = v[0]; sav.path[3 * lev + 1] = v[1]; sav.path[3 * lev + 2] = v[2]; }
                    } else if (both) pt_kd_sav_store(sav, lev, PT_LANES(near_first) ? end : start);
                    if (push && descend) {
                        code = PT_LANES(in_second) ? 1u : 0u;
                        if (both) {
                            code = PT_LANES(both & near_first) ? 2u : code;
                            code = PT_LANES(both & PT_MNOT(near_first)) ? 3u : code;
                            end = PT_LANES(both & near_first) ? plane_t : end;
                            start = PT_LANES(both & PT_MNOT(near_first)) ? plane_t : start;
#ifndef PT_KD_SEG_PER_NODE
                            if (!MESH) {
                                seg0c = (float)start; seg1c = (float)end;
                                seg0c = seg0c - fabsf(seg0c) * 2.4e-7f; seg1c = seg1c + fabsf(seg1c) * 2.4e-7f;
                            }
#endif
                        }
                    }
                }
#else
                const pt_mask same = mine & PT_MNOT(s ^ e);
                const pt_mask go_f = (same & s) | strad, go_b = (same & PT_MNOT(s)) | strad;
                // the child most lanes call near goes first (a lane's near side is the side of its range start)
                const bool front_first = PT_POPC((go_f | go_b) & s) >= PT_POPC((go_f | go_b) & PT_MNOT(s));
                const uint32_t c_first = front_first ? v[3] : v[4], c_second = front_first ? v[4] : v[3];
                const pt_mask in_first = front_first ? go_f : go_b, in_second = front_first ? go_b : go_f;
                const bool push = in_second != 0ull;
                const bool room = sp < wwords;
                if (room && push) wstack[sp] = (c_second << 5) | (uint32_t)lev;  // one word: the child and the split's level
                failed = failed || (push && !room);
                descend = (go_f | go_b) != 0ull && !failed;
                sp += (push && descend) ? 1 : 0;
                // the lanes' own bookkeeping, without branches: both children (straddling) - one bound becomes plane_t and goes to the
                // slot, code 2 (the slot holds the end) / 3 (the start); the second child only - code 1; else 0. Lanes whose slot is not
                // read later may write it too.
                const pt_mask near_first = front_first ? s : PT_MNOT(s);
                const pt_mask both = (push && descend) ? strad : 0ull;
                if (lev < sav.top_levels) {  // a top level: no slot - the path table gets the split's plane and axis (every lane writes the same words)
                    if (push && descend) { sav.path[3 * lev] = v[0]; sav.path[3 * lev + 1] = v[1]; sav.path[3 * lev + 2] = v[2]; }
                } else if (both) pt_kd_sav_store(sav, lev, PT_LANES(near_first) ? end : start);
                // (behind wave-uniform branches since round 5: most splits below the top levels push nothing - every lane is on one side - and then there is nothing to select;
                // the walk is bound by instruction issue, big-scene 27.55 -> 27.27 ms, c28)
                uint32_t code = 0u;
                if (push && descend) {
                    code = PT_LANES(in_second) ? 1u : 0u;
                    if (both) {
                        code = PT_LANES(both & near_first) ? 2u : code;
                        code = PT_LANES(both & PT_MNOT(near_first)) ? 3u : code;
                        end = PT_LANES(both & near_first) ? plane_t : end;
                        start = PT_LANES(both & PT_MNOT(near_first)) ? plane_t : start;
                    }
                }
#endif
                if (descend) {
                    const int sh = 2 * (lev & 15);
                    if (lev < 16) codes_lo = (codes_lo & ~(3u << sh)) | (code << sh);
                    else codes_hi = (codes_hi & ~(3u << sh)) | (code << sh);
                    in = in_first; cur = c_first; lev++;
                }
            } else if (mine) {
                // ---- a leaf: ray.rs:87-99 fold over the leaf's nodes in the reference's order with a strictly shrinking end
                if (STATS && PT_LANES(mine)) cnt->n_leaf++;
                PT_WAVE_COUNT(5);
                const uint32_t first = v[5], count = v[6];
                PtHit lb; lb.t = end; lb.node = PT_NO_HIT; lb.sub = 0;
                pt_mask found = 0ull;
                pt_mask want = mine;  // a shadow ray is done with its first hit
                const void* const info_base = pt_pin_ptr(sc.info);
                const void* const inv_base = pt_pin_ptr(sc.inv);
                for (uint32_t i = 0; i < count; i++) {
                    // the reference and its cull box in one scalar fetch: {flattened node, 0, the node's padded world box as 6 f32 rounded outward}
                    const pt_u32x8 ref = pt_sload8_off(ref_base, (first + i) << 5);
                    PT_WAVE_COUNT(6);  // -DPT_DIAG: leaf references looked at, per wavefront (per lane: n_bbox)
                    const uint32_t item = ref[0];
                    pt_mask test = want;
                    if (cull & 2u) {  // a node whose box the lane's segment does not reach cannot report a hit in it
                        const float lo[3] = {pt_f32_of(ref[2]), pt_f32_of(ref[3]), pt_f32_of(ref[4])}, hi[3] = {pt_f32_of(ref[5]), pt_f32_of(ref[6]), pt_f32_of(ref[7])};
                        if (STATS && PT_LANES(want)) cnt->n_bbox++;
                        test &= pt_slab_seg_pk_m(lo, hi, q, seg0, seg1);
                    }
                    if (!test) continue;
                    PT_WAVE_COUNT(7);  // -DPT_DIAG: exact tests, per wavefront (per lane: n_analytic)
                    // the node's record in one round trip through the scalar cache: {type, data, flags, material} and rows 0..2 of its inverse
                    pt_u32x4 info;
                    pt_u32x16 ma;
                    pt_u32x8 mb;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("s_load_dwordx4 %0, %3, %5\n\ts_load_dwordx16 %1, %4, %6\n\ts_load_dwordx8 %2, %4, %7\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(info), "=&s"(ma), "=&s"(mb) : "s"(info_base), "s"(inv_base), "s"(item << 4), "s"(item * 96u), "s"(item * 96u + 64u) : "memory");
#else
                    info = *reinterpret_cast<const pt_u32x4*>(static_cast<const char*>(info_base) + ((size_t)item << 4));
                    ma = *reinterpret_cast<const pt_u32x16*>(static_cast<const char*>(inv_base) + (size_t)item * 96u);
                    mb = *reinterpret_cast<const pt_u32x8*>(static_cast<const char*>(inv_base) + (size_t)item * 96u + 64u);
#endif
                    const uint32_t type = info[0], data = info[1];
                    double mm[12];
#pragma unroll
                    for (int k = 0; k < 8; k++) mm[k] = pt_f64_of(ma[2 * k], ma[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 4; k++) mm[8 + k] = pt_f64_of(mb[2 * k], mb[2 * k + 1]);
                    const PtRay local = pt_ray_to_local(mm, ray);  // flat_scene.rs:74
                    const bool test_l = PT_LANES(test);
                    if (STATS && test_l) cnt->n_analytic++;
                    bool hit = false;
                    if (MESH && (type == PT_MESH || type == PT_KDMESH)) {
                        const PtMeshInfo* mi = sc.meshes + data;
                        double bi[12];
                        pt_u32x4 head;  // {tri_first, tri_count, blas_root, kd_root}: one round trip with the box inverse
                        pt_sload_mat12_x4(mi->bbox_inv, bi, head);
                        if (KDMESH && type == PT_KDMESH && (int32_t)head[3] >= 0) {  // the reference's own triangle tree (quirk Q3), per lane
                            if (test_l) {
                                double t; uint32_t tri = 0;
                                if (pt_kdmesh_hit<STATS>(sc, *mi, local, start, pt_cand_end(lb, item, 0), lane_stk, 0, &t, &tri, cnt)) { lb.t = t; lb.node = item; lb.sub = tri; hit = true; }
                            }
                        } else {  // mesh.rs:146-155: box test, then the triangles (also a KDMesh without a tree of its own: PORTRAYER_KDMESH_AS_MESH)
                            const uint32_t root = head[2];
                            if (STATS && test_l) cnt->n_bbox++;
                            if (root == PT_REF_EMPTY) continue;
                            const bool inside = test_l && pt_bbox_test_hit(bi, local, start, pt_cand_end(lb, item, 0));
                            if (!pt_any(inside)) continue;
                            if (!pt_packet_mesh_below_kd<STATS>(sc, item, root, head[1], local, inside, start, PT_LANES(any_m), lb, hit, wstack + sp, wwords - sp, cnt)) failed = true;
                        }
                    } else if (test_l) {
                        double t; uint32_t part = 0;
                        if (type == PT_TRIANGLE) {  // stand-alone triangle, stored after the mesh triangles
                            double beta, gamma;
                            if (STATS) cnt->n_tri++;
                            hit = pt_triangle_hit(sc.tri_v + 9 * (size_t)data, local, start, pt_cand_end(lb, item, data), &t, &beta, &gamma);
                            part = data;
                        } else {
                            hit = pt_unit_prim_hit(type, local, start, pt_cand_end(lb, item, 0), &t, &part);
                        }
                        if (hit) { lb.t = t; lb.node = item; lb.sub = part; }
                    }
                    const pt_mask hit_m = PT_BALLOT(hit);
                    found |= hit_m;
                    want &= PT_MNOT(hit_m & any_m);
                    // the segment has shrunk: later references of this leaf are culled against the new end
                    if (hit_m) { const float s1 = (float)lb.t; seg1 = PT_LANES(hit_m) ? s1 + fabsf(s1) * 2.4e-7f : seg1; }
                }
                if (found) {  // (start <= lb.t < best.t by the admission test above)
                    const bool f = PT_LANES(found);
                    best.t = f ? lb.t : best.t; best.node = f ? lb.node : best.node; best.sub = f ? lb.sub : best.sub;
                    alive &= PT_MNOT(found & any_m);
                }
            }
        }
        if (descend) continue;
        // ---- the next pending subtree: the second child of the split at level L. (The way out of the walk is tested at the very end of
        // the body and everything before it runs either way - reading slot 0 of an empty stack is harmless -: on a path that leaves the loop
        // early the node index would be undefined, the compiler gives undefined values VECTOR registers, and its SGPR-copy fixing then moves
        // the whole chain of the node index to the vector side, where the scalar load cannot take it.)
        const bool done = failed || alive == 0ull || sp == 0;
        sp -= done ? 0 : 1;
        const uint32_t entry = PT_UNIFORM_U32(wstack[sp]);
        cur = entry >> 5;
        const int L = (int)(entry & 31u);
        if (L >= sav.top_levels) {
            for (int k = lev - 1; k > L; k--) {  // the finished levels below it: put the replaced bounds back
                const uint32_t c = ((k < 16 ? codes_lo : codes_hi) >> (2 * (k & 15))) & 3u;
                const pt_mask c2m = PT_U32_EQ(c, 2u), c3m = PT_U32_EQ(c, 3u);
                if (c2m | c3m) {
                    const double sv = pt_kd_sav_load(sav, k);  // (lanes whose slot holds nothing read it too: not used)
                    end = PT_LANES(c2m) ? sv : end;
                    start = PT_LANES(c3m) ? sv : start;
                }
            }
        }
        {
            const int sh = 2 * (L & 15);
            const uint32_t c = ((L < 16 ? codes_lo : codes_hi) >> sh) & 3u;
            const pt_mask c2m = PT_U32_EQ(c, 2u), c3m = PT_U32_EQ(c, 3u);
            if (L >= sav.top_levels && (c2m | c3m)) {
                // had [start, plane_t) (c == 2, the slot holds the end): now [plane_t, end), the slot keeps the start for the way back up;
                // had [plane_t, end) (c == 3, the slot holds the start): now [start, plane_t), the slot keeps the end
                const double sv = pt_kd_sav_load(sav, L);
                pt_kd_sav_store(sav, L, PT_LANES(c2m) ? start : end);
                const double ns = PT_LANES(c2m) ? end : (PT_LANES(c3m) ? sv : start);
                const double ne = PT_LANES(c2m) ? sv : (PT_LANES(c3m) ? start : end);
                start = ns; end = ne;
            }
            in = PT_MNOT(PT_U32_EQ(c, 0u));
            const uint32_t c2 = PT_LANES(c2m) ? 3u : (PT_LANES(c3m) ? 2u : 0u);
            if (L < 16) codes_lo = (codes_lo & ~(3u << sh)) | (c2 << sh);
            else codes_hi = (codes_hi & ~(3u << sh)) | (c2 << sh);
        }
        if (L < sav.top_levels && !done) {
            // A top level: the range of the second child from the codes of the levels 0 .. L alone (see PtKdSav) - end = the plane parameter of the deepest
            // level with code 2, else infinity; start = that of the deepest level with code 3, else epsilon. Deepest first, until every lane has both.
            pt_mask need_e = in, need_s = in;
            for (int j = L; j >= 0 && (need_e | need_s); j--) {
                const uint32_t cj = ((j < 16 ? codes_lo : codes_hi) >> (2 * (j & 15))) & 3u;
                const pt_mask e_here = PT_U32_EQ(cj, 2u) & need_e, s_here = PT_U32_EQ(cj, 3u) & need_s;
                if (!(e_here | s_here)) continue;
                const uint32_t p0 = PT_UNIFORM_U32(sav.path[3 * j]), p1 = PT_UNIFORM_U32(sav.path[3 * j + 1]), ax = PT_UNIFORM_U32(sav.path[3 * j + 2]);
                const double plane = pt_f64_of(p0, p1);
                double tj;
                if (ax == 0u) tj = pt_kd_plane_t(ray.o.x, ray.d.x, yx, okx, plane, e_here | s_here);
                else if (ax == 1u) tj = pt_kd_plane_t(ray.o.y, ray.d.y, yy, oky, plane, e_here | s_here);
                else tj = pt_kd_plane_t(ray.o.z, ray.d.z, yz, okz, plane, e_here | s_here);
                end = PT_LANES(e_here) ? tj : end;
                start = PT_LANES(s_here) ? tj : start;
                need_e &= PT_MNOT(e_here); need_s &= PT_MNOT(s_here);
            }
            end = PT_LANES(need_e) ? (double)INFINITY : end;
            start = PT_LANES(need_s) ? (double)PT_EPSILON : start;
        }
        lev = L + 1;
#ifndef PT_KD_SEG_PER_NODE
        if (!MESH) {
            seg0c = (float)start; seg1c = (float)end;
            seg0c = seg0c - fabsf(seg0c) * 2.4e-7f; seg1c = seg1c + fabsf(seg1c) * 2.4e-7f;
        }
#endif
        if (done) break;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : : "s"(cur), "s"(lev), "s"(sp));  // (live, and in scalar registers, on the way out: see above)
#endif
    if (failed) {  // never expected (pt_scene_upload sizes the stack); recorded unconditionally so that a render whose results would be wrong cannot return PT_OK
#if defined(__HIP_DEVICE_COMPILE__)
        if (overflow) atomicOr(overflow, steps > PT_KD_WALK_STEPS_MAX ? 4u : 1u);  // 4: the watchdog
#endif
        if (STATS) cnt->stack_overflow++;
        if (has_ray) best.node = PT_NO_HIT;
    }
}

// Traversal of one ray in the semantics of `MODE` (PT_MODE_*).
template <int MODE, bool STATS, class Stack>
PT_HD void pt_trace(const PtSceneView& sc, const PtRay& ray, bool any, PtHit& hit, const Stack& stk, PtCounters* cnt) {
    if (MODE == PT_MODE_KD || MODE == PT_MODE_KD_MESH) pt_trace_kd<STATS, true>(sc, ray, any, hit, stk, cnt);
    else if (MODE == PT_MODE_KD_NOMESH) pt_trace_kd<STATS, false>(sc, ray, any, hit, stk, cnt);
    else if (MODE == PT_MODE_FLAT_NOMESH) pt_trace_flat_simple<STATS>(sc, ray, any, hit, stk, cnt);
    else if (MODE == PT_MODE_FLAT_KDMESH) pt_trace_flat<STATS, true, true>(sc, ray, any, hit, stk, cnt);
    else if (MODE == PT_MODE_HIER) pt_trace_flat<STATS, true, true, true>(sc, ray, any, hit, stk, cnt);
    else if (MODE == PT_MODE_HIER_NOMESH) pt_trace_flat<STATS, false, false, true>(sc, ray, any, hit, stk, cnt);
    else if (MODE == PT_MODE_HIER_MESH) pt_trace_flat<STATS, true, false, true>(sc, ray, any, hit, stk, cnt);
    else pt_trace_flat<STATS, true, false>(sc, ray, any, hit, stk, cnt);
}
