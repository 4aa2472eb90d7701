// Device-side build of mesh triangle trees: see pt_build.h.
//
// Parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for
// Bounding Volume Hierarchy Construction", 2018), bottom-up:
//
//   1. pt_ploc_prepare  per triangle: f64 bounds -> f32 box rounded outward, 63-bit Morton code of the box centre
//   2. rocprim radix sort (key = Morton code, value = triangle) -- library call; sorting is not this path's subject
//   3. repeat until one cluster is left (the clusters stay in Morton order):
//        pt_ploc_nearest  every cluster looks PT_PLOC_RADIUS places to either side for the neighbour whose union
//                         with it has the smallest surface area
//        pt_ploc_merge    clusters that chose each other merge: the lower one becomes the new node (written in the
//                         walk's format: two child boxes + two references), the upper one retires; a node whose
//                         children together hold <= max_leaf triangles becomes a leaf instead
//        rocprim exclusive scan + pt_ploc_compact: close the gaps
//   4. pt_ploc_depth    longest root-to-leaf path (the walk's LDS stack is sized from it)
//
// (A Morton-order tree a la Karras 2012 was measured first: 8.4 ms for 1.25 M triangles but 31 % slower to
// walk than the host's binned-SAH tree, and replacing its upper levels by a host-built SAH tree over
// "treelets" showed the loss sits in the LOWER levels: 2.42 Gray/s with 16-triangle treelets, 1.99 with
// 256, 1.90 without, against 2.74 for the host tree.)
#include "pt_build.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#define PT_BUILD_BLOCK 256
#define PT_PLOC_LEAF 0x80000000u
#define PT_PLOC_NONE 0xFFFFFFFFu
#define PT_PLOC_LIMIT 1e18f  // pt_bvh.h PT_BOX_LIMIT: box coordinates stay finite and within +-1e18
#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 16
#endif

namespace {

__device__ __forceinline__ float pt_box_lo(double v) {
    if (!(v > -1e18)) return -PT_PLOC_LIMIT;  // also NaN
    if (v > 1e18) return PT_PLOC_LIMIT;
    return __double2float_rd(v);
}
__device__ __forceinline__ float pt_box_hi(double v) {
    if (!(v < 1e18)) return PT_PLOC_LIMIT;
    if (v < -1e18) return -PT_PLOC_LIMIT;
    return __double2float_ru(v);
}

__device__ __forceinline__ unsigned long long pt_spread21(unsigned long long x) {  // bit i -> bit 3 i
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ unsigned long long pt_quantise21(double c, double lo, double inv_ext) {
    double u = (c - lo) * inv_ext * 2097152.0;
    if (!(u > 0.0)) return 0ull;
    if (u >= 2097151.0) return 2097151ull;
    return (unsigned long long)u;
}

struct PtBox6 {
    float v[6];  // lo xyz, hi xyz
};

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_ploc_prepare(const double* __restrict__ tri_v, uint32_t tri_first, uint32_t n,
                                                                 double lx, double ly, double lz, double ix, double iy, double iz, double pad,
                                                                 unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                                                 PtBox6* __restrict__ tri_box) {
    uint32_t t = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    if (t >= n) return;
    const double* v = tri_v + 9 * (size_t)(tri_first + t);
    double lo[3], hi[3];
    PtBox6 b;
    for (int k = 0; k < 3; k++) {
        double a = v[k], bb = v[3 + k], c = v[6 + k];
        lo[k] = fmin(fmin(a, bb), c);
        hi[k] = fmax(fmax(a, bb), c);
        b.v[k] = pt_box_lo(lo[k] - pad);
        b.v[3 + k] = pt_box_hi(hi[k] + pad);
    }
    tri_box[t] = b;
    unsigned long long qx = pt_quantise21(0.5 * (lo[0] + hi[0]), lx, ix);
    unsigned long long qy = pt_quantise21(0.5 * (lo[1] + hi[1]), ly, iy);
    unsigned long long qz = pt_quantise21(0.5 * (lo[2] + hi[2]), lz, iz);
    keys[t] = (pt_spread21(qx) << 2) | (pt_spread21(qy) << 1) | pt_spread21(qz);
    vals[t] = t;
}

// the clusters of the first round: the triangles in Morton order
__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_ploc_init(uint32_t n, const uint32_t* __restrict__ vals, const PtBox6* __restrict__ tri_box,
                                                              uint32_t tri_first, uint32_t* __restrict__ items, uint32_t* __restrict__ cid,
                                                              PtBox6* __restrict__ cbox) {
    uint32_t k = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    if (k >= n) return;
    items[k] = tri_first + vals[k];
    cid[k] = k | PT_PLOC_LEAF;
    cbox[k] = tri_box[vals[k]];
}

__device__ __forceinline__ float pt_union_area(const PtBox6& a, const PtBox6& b) {
    float dx = fmaxf(a.v[3], b.v[3]) - fminf(a.v[0], b.v[0]);
    float dy = fmaxf(a.v[4], b.v[4]) - fminf(a.v[1], b.v[1]);
    float dz = fmaxf(a.v[5], b.v[5]) - fminf(a.v[2], b.v[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_ploc_nearest(uint32_t m, int radius, const PtBox6* __restrict__ cbox, uint32_t* __restrict__ nearest) {
    uint32_t i = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    if (i >= m) return;
    const PtBox6 me = cbox[i];
    int lo = (int)i - radius, hi = (int)i + radius;
    if (lo < 0) lo = 0;
    if (hi > (int)m - 1) hi = (int)m - 1;
    float best = INFINITY;
    uint32_t best_j = i;
    for (int j = lo; j <= hi; j++) {
        if (j == (int)i) continue;
        float a = pt_union_area(me, cbox[j]);
        if (a < best || best_j == i) { best = a; best_j = (uint32_t)j; }  // the first of equal areas; any j if every area is NaN / inf
    }
    nearest[i] = best_j;
}

struct PtPlocOut {  // everything pt_ploc_merge writes for the finished tree
    PtBvhNode* nodes;       // d_nodes + node_base
    uint32_t* items;        // d_items + item_base: [0, n) single triangles in Morton order, then max_leaf slots per node
    uint32_t* leaf_count;   // per node: triangles if the node was turned into a leaf, else 0
    uint32_t* node_parent;  // per node
    uint32_t* leaf_parent;  // per sorted triangle
    uint32_t* counter;      // next free node
    uint32_t node_base, item_base, n;
    int max_leaf;
};

// number of triangles if `id` is a leaf of the final tree (a triangle or a collapsed node), else 0; *first = its items
__device__ __forceinline__ uint32_t pt_ploc_leaf_items(const PtPlocOut& o, uint32_t id, uint32_t* first) {
    if (id & PT_PLOC_LEAF) { *first = id & ~PT_PLOC_LEAF; return 1u; }
    uint32_t c = o.leaf_count[id];
    *first = o.n + (uint32_t)o.max_leaf * id;
    return c;
}
__device__ __forceinline__ uint32_t pt_ploc_ref(const PtPlocOut& o, uint32_t id) {
    uint32_t first, c = pt_ploc_leaf_items(o, id, &first);
    return c ? (PT_REF_LEAF | ((o.item_base + first) << 3) | (c - 1u)) : o.node_base + id;
}

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_ploc_merge(uint32_t m, const uint32_t* __restrict__ nearest, const uint32_t* __restrict__ cid,
                                                               const PtBox6* __restrict__ cbox, uint32_t* __restrict__ cid_out,
                                                               PtBox6* __restrict__ cbox_out, uint32_t* __restrict__ keep, PtPlocOut o) {
    uint32_t i = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    if (i >= m) return;
    uint32_t j = nearest[i];
    bool mutual = j != i && nearest[j] == i;
    if (!mutual) { cid_out[i] = cid[i]; cbox_out[i] = cbox[i]; keep[i] = 1u; return; }
    if (i > j) { keep[i] = 0u; return; }
    const uint32_t a = cid[i], b = cid[j];
    const PtBox6 ba = cbox[i], bb = cbox[j];
    const uint32_t node = atomicAdd(o.counter, 1u);
    uint32_t fa, fb, ca = pt_ploc_leaf_items(o, a, &fa), cb = pt_ploc_leaf_items(o, b, &fb);
    if (ca && cb && ca + cb <= (uint32_t)o.max_leaf) {  // a leaf of ca + cb triangles: gather them in the node's own slot
        uint32_t dst = o.n + (uint32_t)o.max_leaf * node;
        for (uint32_t k = 0; k < ca; k++) o.items[dst + k] = o.items[fa + k];
        for (uint32_t k = 0; k < cb; k++) o.items[dst + ca + k] = o.items[fb + k];
        o.leaf_count[node] = ca + cb;
    } else {
        PtBvhNode nd;
        for (int k = 0; k < 3; k++) { nd.lo[k][0] = ba.v[k]; nd.hi[k][0] = ba.v[3 + k]; nd.lo[k][1] = bb.v[k]; nd.hi[k][1] = bb.v[3 + k]; }
        nd.child0 = pt_ploc_ref(o, a);
        nd.child1 = pt_ploc_ref(o, b);
        nd.pad[0] = nd.pad[1] = 0u;
        o.nodes[node] = nd;
        o.leaf_count[node] = 0u;
    }
    if (a & PT_PLOC_LEAF) o.leaf_parent[a & ~PT_PLOC_LEAF] = node; else o.node_parent[a] = node;
    if (b & PT_PLOC_LEAF) o.leaf_parent[b & ~PT_PLOC_LEAF] = node; else o.node_parent[b] = node;
    o.node_parent[node] = PT_PLOC_NONE;
    PtBox6 u;
    for (int k = 0; k < 3; k++) { u.v[k] = fminf(ba.v[k], bb.v[k]); u.v[3 + k] = fmaxf(ba.v[3 + k], bb.v[3 + k]); }
    cid_out[i] = node;
    cbox_out[i] = u;
    keep[i] = 1u;
}

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_ploc_compact(uint32_t m, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                                                                 const uint32_t* __restrict__ cid_in, const PtBox6* __restrict__ cbox_in,
                                                                 uint32_t* __restrict__ cid, PtBox6* __restrict__ cbox, uint32_t* __restrict__ m_out) {
    uint32_t i = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    if (i >= m) return;
    if (keep[i]) { cid[pos[i]] = cid_in[i]; cbox[pos[i]] = cbox_in[i]; }
    if (i == m - 1) *m_out = pos[i] + keep[i];
}

// inner nodes on the path from a triangle to the root, + 1
__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_ploc_depth(uint32_t n, const uint32_t* __restrict__ node_parent, const uint32_t* __restrict__ leaf_parent,
                                                               const uint32_t* __restrict__ leaf_count, int* depth) {
    uint32_t k = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    int d = 0;
    if (k < n) {
        d = 1;
        for (uint32_t p = leaf_parent[k]; p != PT_PLOC_NONE; p = node_parent[p]) d += leaf_count[p] ? 0 : 1;
    }
    for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o));
    if ((threadIdx.x & 63) == 0) atomicMax(depth, d);
}

struct Arena {
    char* base = nullptr;
    size_t used = 0, cap = 0;
    template <class T> T* take(size_t count) {
        used = (used + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + used);
        used += count * sizeof(T);
        return p;
    }
};

}  // namespace

#define PT_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { if (arena.base) hipFree(arena.base); return e_; } } while (0)

hipError_t pt_device_build_mesh_tree(const double* d_tri_v, uint32_t tri_first, uint32_t n, const double lo[3], const double hi[3], double pad, int max_leaf,
                                     PtBvhNode* d_nodes, uint32_t node_base, uint32_t* d_items, uint32_t item_base, hipStream_t stream,
                                     PtDeviceBuildResult* out) {
    Arena arena;
    if (n < 2 || max_leaf < 1 || max_leaf > 8 || n <= (uint32_t)max_leaf) return hipErrorInvalidValue;
    size_t sort_bytes = 0, scan_bytes = 0;
    PT_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (size_t)n, 0u, 63u, stream));
    PT_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    const size_t tmp_bytes = std::max(sort_bytes, scan_bytes);
    // one allocation for every temporary
    size_t need = 8192 + tmp_bytes + (size_t)n * (8 + 8 + 4 + 4 + 24 + 3 * 24 + 3 * 4 + 4 + 4 + 4 + 4 + 4 + 4);
    PT_TRY(hipMalloc((void**)&arena.base, need));
    arena.cap = need;
    auto* keys_in = arena.take<unsigned long long>(n);
    auto* keys = arena.take<unsigned long long>(n);
    auto* vals_in = arena.take<uint32_t>(n);
    auto* vals = arena.take<uint32_t>(n);
    auto* tri_box = arena.take<PtBox6>(n);
    PtBox6* cbox[3] = {arena.take<PtBox6>(n), arena.take<PtBox6>(n), arena.take<PtBox6>(n)};  // current, merged (with gaps), next
    uint32_t* cid[3] = {arena.take<uint32_t>(n), arena.take<uint32_t>(n), arena.take<uint32_t>(n)};
    auto* nearest = arena.take<uint32_t>(n);
    auto* keep = arena.take<uint32_t>(n);
    auto* pos = arena.take<uint32_t>(n);
    auto* leaf_count = arena.take<uint32_t>(n);
    auto* node_parent = arena.take<uint32_t>(n);
    auto* leaf_parent = arena.take<uint32_t>(n);
    auto* scalars = arena.take<uint32_t>(4);  // [0] next node, [1] clusters left, [2] depth
    void* tmp = arena.take<char>(tmp_bytes);
    if (arena.used > arena.cap) { hipFree(arena.base); return hipErrorOutOfMemory; }

    int radius = PT_PLOC_RADIUS;
    if (const char* e = getenv("PORTRAYER_PLOC_RADIUS")) radius = std::min(std::max(atoi(e), 1), 128);

    hipEvent_t e0, e1;
    PT_TRY(hipEventCreate(&e0));
    PT_TRY(hipEventCreate(&e1));
    PT_TRY(hipEventRecord(e0, stream));
    double inv[3];
    for (int k = 0; k < 3; k++) inv[k] = hi[k] > lo[k] ? 1.0 / (hi[k] - lo[k]) : 0.0;
    auto blocks = [](uint32_t count) { return dim3((count + PT_BUILD_BLOCK - 1) / PT_BUILD_BLOCK); };
    PT_TRY(hipMemsetAsync(scalars, 0, 16, stream));
    hipLaunchKernelGGL(pt_ploc_prepare, blocks(n), dim3(PT_BUILD_BLOCK), 0, stream, d_tri_v, tri_first, n, lo[0], lo[1], lo[2], inv[0], inv[1], inv[2], pad,
                       keys_in, vals_in, tri_box);
    PT_TRY(hipGetLastError());
    PT_TRY(rocprim::radix_sort_pairs(tmp, sort_bytes, keys_in, keys, vals_in, vals, (size_t)n, 0u, 63u, stream));
    hipLaunchKernelGGL(pt_ploc_init, blocks(n), dim3(PT_BUILD_BLOCK), 0, stream, n, vals, tri_box, tri_first, d_items + item_base, cid[0], cbox[0]);
    PT_TRY(hipGetLastError());

    PtPlocOut o;
    o.nodes = d_nodes + node_base; o.items = d_items + item_base; o.leaf_count = leaf_count; o.node_parent = node_parent; o.leaf_parent = leaf_parent;
    o.counter = scalars; o.node_base = node_base; o.item_base = item_base; o.n = n; o.max_leaf = max_leaf;
    uint32_t m = n;
    int rounds = 0;
    while (m > 1) {
        if (++rounds > 4096) { hipFree(arena.base); return hipErrorLaunchFailure; }  // every round merges at least the closest pair
        hipLaunchKernelGGL(pt_ploc_nearest, blocks(m), dim3(PT_BUILD_BLOCK), 0, stream, m, radius, cbox[0], nearest);
        hipLaunchKernelGGL(pt_ploc_merge, blocks(m), dim3(PT_BUILD_BLOCK), 0, stream, m, nearest, cid[0], cbox[0], cid[1], cbox[1], keep, o);
        PT_TRY(hipGetLastError());
        PT_TRY(rocprim::exclusive_scan(tmp, scan_bytes, keep, pos, 0u, (size_t)m, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(pt_ploc_compact, blocks(m), dim3(PT_BUILD_BLOCK), 0, stream, m, keep, pos, cid[1], cbox[1], cid[2], cbox[2], scalars + 1);
        PT_TRY(hipGetLastError());
        PT_TRY(hipMemcpyAsync(&m, scalars + 1, 4, hipMemcpyDeviceToHost, stream));
        PT_TRY(hipStreamSynchronize(stream));
        std::swap(cid[0], cid[2]);
        std::swap(cbox[0], cbox[2]);
    }
    hipLaunchKernelGGL(pt_ploc_depth, blocks(n), dim3(PT_BUILD_BLOCK), 0, stream, n, node_parent, leaf_parent, leaf_count, (int*)(scalars + 2));
    PT_TRY(hipGetLastError());
    PT_TRY(hipEventRecord(e1, stream));
    uint32_t h[3] = {0, 0, 0}, root = 0;
    PT_TRY(hipMemcpyAsync(h, scalars, 12, hipMemcpyDeviceToHost, stream));
    PT_TRY(hipMemcpyAsync(&root, cid[0], 4, hipMemcpyDeviceToHost, stream));
    PT_TRY(hipStreamSynchronize(stream));
    float ms = 0.0f;
    PT_TRY(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(arena.base);
    if (h[0] != n - 1 || (root & PT_PLOC_LEAF)) return hipErrorLaunchFailure;  // a binary tree over n leaves has n - 1 nodes
    out->root = node_base + root;  // n > max_leaf: the root is never a leaf
    out->depth = (int)h[2];
    out->ms = ms;
    out->rounds = rounds;
    return hipSuccess;
}
