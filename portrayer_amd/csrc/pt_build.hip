// Device-side build of mesh triangle trees: see pt_build.h.
//
//   1. pt_lbvh_prepare    per triangle: f64 bounds -> f32 box rounded outward, 63-bit Morton code of the box centre
//   2. rocprim radix sort (key = Morton code, value = triangle)  -- the only library call; sorting is not this path's subject
//   3. pt_lbvh_hierarchy  Karras 2012: every inner node finds its range of sorted leaves and its split in O(log n), all in parallel
//   4. pt_lbvh_refit      bottom-up boxes: the second thread to reach a node merges its children's boxes and climbs on
//   5. pt_lbvh_emit       nodes in the walk's format (two child boxes + two references per node); subtrees of <= max_leaf
//                         triangles (contiguous in Morton order) become leaves
//   6. pt_lbvh_treelets   the Morton tree's upper levels split space at the middle regardless of where the triangles are.
//                         They are replaced: the subtrees of <= PT_TREELET triangles whose parents are larger ("treelets",
//                         a few thousand) are handed to the host, which builds a binned-SAH tree over their boxes in a
//                         millisecond or two (pt_bvh.h); its leaves point at the treelets' roots
//   7. pt_lbvh_depth      longest path below a treelet root (the walk's LDS stack is sized from it + the top tree's depth)
#include "pt_build.h"

#include <algorithm>
#include <cstring>
#include <vector>

#include "pt_bvh.h"

#include <rocprim/device/device_radix_sort.hpp>

#define PT_BUILD_BLOCK 256
#define PT_LBVH_LEAF 0x80000000u
#define PT_LBVH_NONE 0xFFFFFFFFu
#ifndef PT_TREELET
#define PT_TREELET 256u  // triangles per treelet (subtrees of the Morton tree kept as they are)
#endif
#define PT_LBVH_LIMIT 1e18f  // pt_bvh.h PT_BOX_LIMIT: box coordinates stay finite and within +-1e18

namespace {

__device__ __forceinline__ float pt_box_lo(double v) {
    if (!(v > -1e18)) return -PT_LBVH_LIMIT;  // also NaN
    if (v > 1e18) return PT_LBVH_LIMIT;
    return __double2float_rd(v);
}
__device__ __forceinline__ float pt_box_hi(double v) {
    if (!(v < 1e18)) return PT_LBVH_LIMIT;
    if (v < -1e18) return -PT_LBVH_LIMIT;
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

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_lbvh_prepare(const double* __restrict__ tri_v, uint32_t tri_first, uint32_t n,
                                                                 double lx, double ly, double lz, double ix, double iy, double iz, double pad,
                                                                 unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                                                 float* __restrict__ leaf_box) {
    uint32_t t = blockIdx.x * PT_BUILD_BLOCK + threadIdx.x;
    if (t >= n) return;
    const double* v = tri_v + 9 * (size_t)(tri_first + t);
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        double a = v[k], b = v[3 + k], c = v[6 + k];
        lo[k] = fmin(fmin(a, b), c);
        hi[k] = fmax(fmax(a, b), c);
        leaf_box[6 * (size_t)t + k] = pt_box_lo(lo[k] - pad);
        leaf_box[6 * (size_t)t + 3 + k] = pt_box_hi(hi[k] + pad);
    }
    unsigned long long qx = pt_quantise21(0.5 * (lo[0] + hi[0]), lx, ix);
    unsigned long long qy = pt_quantise21(0.5 * (lo[1] + hi[1]), ly, iy);
    unsigned long long qz = pt_quantise21(0.5 * (lo[2] + hi[2]), lz, iz);
    keys[t] = (pt_spread21(qx) << 2) | (pt_spread21(qy) << 1) | pt_spread21(qz);
    vals[t] = t;
}

// length of the common prefix of the keys of sorted leaves i and j; equal keys are told apart by their positions
__device__ __forceinline__ int pt_delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012), section 3
__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_lbvh_hierarchy(const unsigned long long* __restrict__ keys, int n, uint32_t* __restrict__ child,
                                                                   uint32_t* __restrict__ range, uint32_t* __restrict__ node_parent,
                                                                   uint32_t* __restrict__ leaf_parent) {
    int i = (int)(blockIdx.x * PT_BUILD_BLOCK + threadIdx.x);
    if (i >= n - 1) return;
    int d = pt_delta(keys, n, i, i + 1) - pt_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = pt_delta(keys, n, i, i - d);
    int lmax = 2;
    while (pt_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (pt_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = pt_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (pt_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int first = i < j ? i : j, last = i < j ? j : i;
    bool left_leaf = first == gamma, right_leaf = last == gamma + 1;
    child[2 * i] = (uint32_t)gamma | (left_leaf ? PT_LBVH_LEAF : 0u);
    child[2 * i + 1] = (uint32_t)(gamma + 1) | (right_leaf ? PT_LBVH_LEAF : 0u);
    range[2 * i] = (uint32_t)first;
    range[2 * i + 1] = (uint32_t)last;
    if (left_leaf) leaf_parent[gamma] = (uint32_t)i; else node_parent[gamma] = (uint32_t)i;
    if (right_leaf) leaf_parent[gamma + 1] = (uint32_t)i; else node_parent[gamma + 1] = (uint32_t)i;
    if (i == 0) node_parent[0] = PT_LBVH_NONE;
}

// another CU may have written the sibling's box a moment ago: read it past this CU's L1
__device__ __forceinline__ float pt_load_coherent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_lbvh_refit(int n, const uint32_t* __restrict__ vals, const float* __restrict__ leaf_box,
                                                               const uint32_t* __restrict__ child, const uint32_t* __restrict__ node_parent,
                                                               const uint32_t* __restrict__ leaf_parent, float* node_box, uint32_t* arrived,
                                                               uint32_t tri_first, uint32_t* __restrict__ items) {
    int k = (int)(blockIdx.x * PT_BUILD_BLOCK + threadIdx.x);
    if (k >= n) return;
    items[k] = tri_first + vals[k];
    float box[6];
    for (int c = 0; c < 6; c++) box[c] = leaf_box[6 * (size_t)vals[k] + c];
    uint32_t me = (uint32_t)k | PT_LBVH_LEAF;
    uint32_t p = leaf_parent[k];
    while (p != PT_LBVH_NONE) {
        // publish nothing yet: the first thread to arrive leaves, the second finds the sibling's box complete
        if (atomicAdd(&arrived[p], 1u) == 0u) return;
        uint32_t c0 = child[2 * p], c1 = child[2 * p + 1];
        uint32_t sib = c0 == me ? c1 : c0;
        float other[6];
        if (sib & PT_LBVH_LEAF) {
            for (int c = 0; c < 6; c++) other[c] = leaf_box[6 * (size_t)vals[sib & ~PT_LBVH_LEAF] + c];
        } else {
            for (int c = 0; c < 6; c++) other[c] = pt_load_coherent(node_box + 6 * (size_t)sib + c);
        }
        for (int c = 0; c < 3; c++) { box[c] = fminf(box[c], other[c]); box[3 + c] = fmaxf(box[3 + c], other[3 + c]); }
        for (int c = 0; c < 6; c++) node_box[6 * (size_t)p + c] = box[c];
        __threadfence();  // the box is visible before the parent's counter moves
        me = p;
        p = node_parent[p];
    }
}

__device__ __forceinline__ void pt_emit_child(uint32_t c, const uint32_t* __restrict__ range, const uint32_t* __restrict__ vals,
                                              const float* __restrict__ leaf_box, const float* __restrict__ node_box, int max_leaf,
                                              uint32_t node_base, uint32_t item_base, uint32_t* ref, float* lo, float* hi) {
    const float* b;
    if (c & PT_LBVH_LEAF) {
        uint32_t k = c & ~PT_LBVH_LEAF;
        *ref = PT_REF_LEAF | ((item_base + k) << 3);  // one triangle
        b = leaf_box + 6 * (size_t)vals[k];
    } else {
        uint32_t first = range[2 * c], count = range[2 * c + 1] - first + 1u;
        *ref = count <= (uint32_t)max_leaf ? (PT_REF_LEAF | ((item_base + first) << 3) | (count - 1u)) : node_base + c;
        b = node_box + 6 * (size_t)c;
    }
    for (int k = 0; k < 3; k++) { lo[k] = b[k]; hi[k] = b[3 + k]; }
}

__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_lbvh_emit(int n, const uint32_t* __restrict__ child, const uint32_t* __restrict__ range,
                                                              const uint32_t* __restrict__ vals, const float* __restrict__ leaf_box,
                                                              const float* __restrict__ node_box, int max_leaf, uint32_t node_base,
                                                              uint32_t item_base, PtBvhNode* __restrict__ nodes) {
    int i = (int)(blockIdx.x * PT_BUILD_BLOCK + threadIdx.x);
    if (i >= n - 1) return;
    PtBvhNode nd;
    memset(&nd, 0, sizeof nd);
    if (range[2 * i + 1] - range[2 * i] + 1u > (uint32_t)max_leaf) {  // smaller subtrees are leaves of their parents: their nodes stay unused
        pt_emit_child(child[2 * i], range, vals, leaf_box, node_box, max_leaf, node_base, item_base, &nd.child0, nd.lo0, nd.hi0);
        pt_emit_child(child[2 * i + 1], range, vals, leaf_box, node_box, max_leaf, node_base, item_base, &nd.child1, nd.lo1, nd.hi1);
    } else {
        nd.child0 = nd.child1 = PT_REF_EMPTY;
    }
    nodes[node_base + (uint32_t)i] = nd;
}

// One thread per inner node and per leaf: report the ones that are treelet roots (subtree of <= limit
// triangles under a parent with more), with their packed reference and box.
struct PtTreelet {
    uint32_t ref;
    float box[6];
};
__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_lbvh_treelets(int n, const uint32_t* __restrict__ range, const uint32_t* __restrict__ node_parent,
                                                                  const uint32_t* __restrict__ leaf_parent, const uint32_t* __restrict__ vals,
                                                                  const float* __restrict__ leaf_box, const float* __restrict__ node_box, uint32_t limit,
                                                                  int max_leaf, uint32_t node_base, uint32_t item_base, uint32_t capacity,
                                                                  uint32_t* __restrict__ count, PtTreelet* __restrict__ out) {
    int g = (int)(blockIdx.x * PT_BUILD_BLOCK + threadIdx.x);
    if (g >= 2 * n - 1) return;
    uint32_t parent, size, ref;
    const float* b;
    if (g < n - 1) {  // inner node g
        parent = node_parent[g];
        uint32_t first = range[2 * g];
        size = range[2 * g + 1] - first + 1u;
        ref = size <= (uint32_t)max_leaf ? (PT_REF_LEAF | ((item_base + first) << 3) | (size - 1u)) : node_base + (uint32_t)g;
        b = node_box + 6 * (size_t)g;
    } else {  // leaf
        uint32_t k = (uint32_t)(g - (n - 1));
        parent = leaf_parent[k];
        size = 1u;
        ref = PT_REF_LEAF | ((item_base + k) << 3);
        b = leaf_box + 6 * (size_t)vals[k];
    }
    if (size > limit || parent == PT_LBVH_NONE) return;
    if (range[2 * parent + 1] - range[2 * parent] + 1u <= limit) return;  // inside a treelet, not its root
    uint32_t slot = atomicAdd(count, 1u);
    if (slot >= capacity) return;
    out[slot].ref = ref;
    for (int c = 0; c < 6; c++) out[slot].box[c] = b[c];
}

// longest chain of inner nodes from a leaf up to (and including) its treelet's root, + 1
__global__ void __launch_bounds__(PT_BUILD_BLOCK) pt_lbvh_depth(int n, const uint32_t* __restrict__ range, const uint32_t* __restrict__ node_parent,
                                                               const uint32_t* __restrict__ leaf_parent, uint32_t limit, int* depth) {
    int k = (int)(blockIdx.x * PT_BUILD_BLOCK + threadIdx.x);
    int d = 0;
    if (k < n) {
        d = 1;
        for (uint32_t p = leaf_parent[k]; p != PT_LBVH_NONE && range[2 * p + 1] - range[2 * p] + 1u <= limit; p = node_parent[p]) d++;
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
    size_t sort_bytes = 0;
    PT_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (size_t)n, 0u, 63u, stream));
    // one allocation for every temporary
    size_t need = 4096 + sort_bytes + (size_t)n * (8 + 8 + 4 + 4 + 24 + 24 + 8 + 8 + 4 + 4 + 4 + sizeof(PtTreelet)) + 256 * 20;
    PT_TRY(hipMalloc((void**)&arena.base, need));
    arena.cap = need;
    auto* keys_in = arena.take<unsigned long long>(n);
    auto* keys = arena.take<unsigned long long>(n);
    auto* vals_in = arena.take<uint32_t>(n);
    auto* vals = arena.take<uint32_t>(n);
    auto* leaf_box = arena.take<float>(6 * (size_t)n);
    auto* node_box = arena.take<float>(6 * (size_t)n);
    auto* child = arena.take<uint32_t>(2 * (size_t)n);
    auto* range = arena.take<uint32_t>(2 * (size_t)n);
    auto* node_parent = arena.take<uint32_t>(n);
    auto* leaf_parent = arena.take<uint32_t>(n);
    auto* arrived = arena.take<uint32_t>(n);
    auto* depth = arena.take<int>(1);
    auto* counter = arena.take<uint32_t>(1);
    const uint32_t top_capacity = n;
    auto* treelets = arena.take<PtTreelet>(top_capacity);
    void* sort_tmp = arena.take<char>(sort_bytes);
    if (arena.used > arena.cap) { hipFree(arena.base); return hipErrorOutOfMemory; }

    hipEvent_t e0, e1;
    PT_TRY(hipEventCreate(&e0));
    PT_TRY(hipEventCreate(&e1));
    PT_TRY(hipEventRecord(e0, stream));
    double inv[3];
    for (int k = 0; k < 3; k++) inv[k] = hi[k] > lo[k] ? 1.0 / (hi[k] - lo[k]) : 0.0;
    const unsigned blocks = (n + PT_BUILD_BLOCK - 1) / PT_BUILD_BLOCK;
    PT_TRY(hipMemsetAsync(arrived, 0, (size_t)n * 4, stream));
    PT_TRY(hipMemsetAsync(depth, 0, 4, stream));
    PT_TRY(hipMemsetAsync(counter, 0, 4, stream));
    hipLaunchKernelGGL(pt_lbvh_prepare, dim3(blocks), dim3(PT_BUILD_BLOCK), 0, stream, d_tri_v, tri_first, n, lo[0], lo[1], lo[2], inv[0], inv[1], inv[2], pad,
                       keys_in, vals_in, leaf_box);
    PT_TRY(hipGetLastError());
    PT_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys, vals_in, vals, (size_t)n, 0u, 63u, stream));
    hipLaunchKernelGGL(pt_lbvh_hierarchy, dim3(blocks), dim3(PT_BUILD_BLOCK), 0, stream, keys, (int)n, child, range, node_parent, leaf_parent);
    PT_TRY(hipGetLastError());
    hipLaunchKernelGGL(pt_lbvh_refit, dim3(blocks), dim3(PT_BUILD_BLOCK), 0, stream, (int)n, vals, leaf_box, child, node_parent, leaf_parent, node_box, arrived,
                       tri_first, d_items + item_base);
    PT_TRY(hipGetLastError());
    hipLaunchKernelGGL(pt_lbvh_emit, dim3(blocks), dim3(PT_BUILD_BLOCK), 0, stream, (int)n, child, range, vals, leaf_box, node_box, max_leaf, node_base, item_base,
                       d_nodes);
    PT_TRY(hipGetLastError());
    // ---- top tree over the treelets
    uint32_t limit = PT_TREELET;
    if (const char* e = getenv("PORTRAYER_TREELET")) limit = (uint32_t)std::max(1, atoi(e));
    if (limit >= n) limit = n - 1;  // the root itself is never a treelet
    hipLaunchKernelGGL(pt_lbvh_treelets, dim3((2 * n - 1 + PT_BUILD_BLOCK - 1) / PT_BUILD_BLOCK), dim3(PT_BUILD_BLOCK), 0, stream, (int)n, range, node_parent,
                       leaf_parent, vals, leaf_box, node_box, limit, max_leaf, node_base, item_base, top_capacity, counter, treelets);
    PT_TRY(hipGetLastError());
    hipLaunchKernelGGL(pt_lbvh_depth, dim3(blocks), dim3(PT_BUILD_BLOCK), 0, stream, (int)n, range, node_parent, leaf_parent, limit, depth);
    PT_TRY(hipGetLastError());
    PT_TRY(hipEventRecord(e1, stream));
    int h_depth = 0;
    uint32_t n_treelets = 0;
    PT_TRY(hipMemcpyAsync(&h_depth, depth, 4, hipMemcpyDeviceToHost, stream));
    PT_TRY(hipMemcpyAsync(&n_treelets, counter, 4, hipMemcpyDeviceToHost, stream));
    PT_TRY(hipStreamSynchronize(stream));
    if (n_treelets < 2 || n_treelets > top_capacity) { hipFree(arena.base); return hipErrorInvalidValue; }  // cannot happen for n > limit >= 1 (capacity = n)
    std::vector<PtTreelet> tl(n_treelets);
    PT_TRY(hipMemcpy(tl.data(), treelets, (size_t)n_treelets * sizeof(PtTreelet), hipMemcpyDeviceToHost));
    std::sort(tl.begin(), tl.end(), [](const PtTreelet& a, const PtTreelet& b) { return a.ref < b.ref; });  // the atomic hands slots out in any order
    std::vector<PtBuildBox> boxes(n_treelets);
    for (uint32_t i = 0; i < n_treelets; i++)
        for (int k = 0; k < 3; k++) { boxes[i].lo[k] = tl[i].box[k]; boxes[i].hi[k] = tl[i].box[3 + k]; }
    std::vector<PtBvhNode> top;
    std::vector<uint32_t> top_items;
    PtBvhRef top_root = pt_bvh_build(boxes.data(), nullptr, n_treelets, 1, top, top_items);
    const uint32_t top_base = node_base + (n - 1);
    auto patch = [&](uint32_t ref) -> uint32_t {  // one-item leaves of the top tree become references to the treelets' roots
        if (ref & PT_REF_LEAF) return tl[top_items[(ref & ~PT_REF_LEAF) >> 3]].ref;
        return top_base + ref;
    };
    for (PtBvhNode& nd : top) { nd.child0 = patch(nd.child0); nd.child1 = patch(nd.child1); }
    PT_TRY(hipMemcpy(d_nodes + top_base, top.data(), top.size() * sizeof(PtBvhNode), hipMemcpyHostToDevice));
    float ms = 0.0f;
    PT_TRY(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(arena.base);
    out->root = patch(top_root.child);
    out->depth = h_depth + top_root.depth;
    out->ms = ms;
    out->n_treelets = n_treelets;
    return hipSuccess;
}
