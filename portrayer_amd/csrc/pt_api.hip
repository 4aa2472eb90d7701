// The C ABI declared in include/portrayer_hip.h: context, scene upload (tree builds), render calls, and the
// small kernels around the render kernel (finishing pass, untile, explicit-ray casts for the parity tests).
// The render kernel itself is in pt_render_kernel.h, instantiated per traversal mode by pt_render_inst.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/portrayer_hip.h"
#include "pt_build.h"
#include "pt_bvh.h"
#include "pt_render_inst.h"
#include "pt_shade.h"

// ------------------------------------------------------------------------------------------------
// Kernels
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(PT_BLOCK) pt_cast_kernel(PtSceneView sc, uint64_t n, const double* o, const double* d, int any,
                                                          double* out_t, int32_t* out_node, int32_t* out_sub, unsigned int* overflow) {
    extern __shared__ uint32_t pt_lds[];
    PtStack stk;
    stk.base = pt_lds + threadIdx.x;
    stk.cap = sc.stack_cap;
    stk.overflow = overflow;
    uint64_t i = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    PtRay r;
    r.o = pt_v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = pt_v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    PtHit hit;
    PtCounters cnt;
    pt_trace<MODE, false>(sc, r, any != 0, hit, stk, &cnt);
    out_t[i] = hit.node == PT_NO_HIT ? INFINITY : hit.t;
    out_node[i] = hit.node == PT_NO_HIT ? -1 : (int32_t)hit.node;
    out_sub[i] = hit.node == PT_NO_HIT ? -1 : (int32_t)hit.sub;
}

__global__ void pt_math_kernel(int op, uint64_t n, const double* a, const double* b, double* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
    case 0: r = sqrt(a[i]); break;
    case 1: r = a[i] / b[i]; break;
    case 2: r = pt_pow(a[i], b[i]); break;  // the kernels' pow (pt_pow.h: glibc's, bit for bit)
    case 3: r = a[i] * b[i] + a[i]; break;  // must NOT be fused (-ffp-contract=off)
    case 4: r = atan2(a[i], b[i]); break;   // sphere.rs:57-58 (texture coordinates)
    case 5: r = acos(a[i]); break;          // sphere.rs:59
    case 6: r = pow(a[i], b[i]); break;     // the device library's pow, for comparison
    case 7:  // the k-d walk's short division (pt_trace.h: pt_div_fast) where its exponent test admits the operands, NaN where it does not
        r = (pt_div_exp_ok(a[i]) && pt_div_exp_ok(b[i])) ? pt_div_fast(a[i], b[i], pt_rcp_refined(b[i])) : __builtin_nan("");
        break;
    default: r = 0.0; break;
    }
    out[i] = r;
}

__global__ void __launch_bounds__(256) pt_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    // four independent 16-byte loads in flight per lane before the first store
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        double2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// the plainest form: one 16-byte element per thread, as many blocks as it takes
__global__ void __launch_bounds__(256) pt_copy1_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// Two-child tree -> four-child tree (PtBvh4Node, pt_scene_view.h). Node i starts with its two children and, while it has
// fewer than four, replaces the inner child with the largest surface area by that child's two children (the child a ray is
// most likely to enter is the one worth opening; taking over both children's children regardless leaves a node with a leaf
// child at three entries). Mesh trees only: on scene-level trees the grandchildren form measured better (transmission-refraction
// 3.32 against 3.72 node visits per ray). PORTRAYER_COLLAPSE=plain | mesh | area.
// One thread per two-child node; entries of the array that no tree uses (the device build reserves n - 1 nodes per mesh
// and may need fewer) hold garbage, are never referenced, and are only kept from reading out of bounds.
// tri_leaf: the edge record of every triangle named by a slot of the items array, in slot order (80 bytes a slot: 9 f64 + the triangle's index), so
// that a mesh leaf's triangles lie side by side and the wave-uniform walks fetch them without going through bvh_items first.
__global__ void __launch_bounds__(256) pt_tri_leaf_kernel(const uint32_t* __restrict__ items, uint32_t n_items, const double* __restrict__ tri_e, uint32_t n_tris,
                                                         double* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t t = items[i];
    double* o = out + 10 * (size_t)i;
    if (t < n_tris) {
        const double* e = tri_e + 9 * (size_t)t;
        for (int k = 0; k < 9; k++) o[k] = e[k];
    } else {
        for (int k = 0; k < 9; k++) o[k] = 0.0;
    }
    union { double d; uint32_t u[2]; } c; c.u[0] = t; c.u[1] = 0u;
    o[9] = c.d;
}

__global__ void __launch_bounds__(256) pt_collapse4_kernel(const PtBvhNode* __restrict__ bvh2, PtBvh4Node* __restrict__ bvh4, uint32_t n, int by_area_mode,
                                                            uint32_t scene_first, uint32_t scene_end) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // by_area_mode 0: never, 1: mesh trees only (nodes outside [scene_first, scene_end)), 2: every tree
    const bool by_area = by_area_mode == 2 || (by_area_mode == 1 && (i < scene_first || i >= scene_end));
    const PtBvhNode a = bvh2[i];
    float lo[4][3], hi[4][3];
    uint32_t child[4];
    int k = 0;
    auto put = [&](const PtBvhNode& nd, int which) {
        pt_node_get_box(nd, which, lo[k], hi[k]);
        child[k] = which ? nd.child1 : nd.child0;
        k++;
    };
    auto inner = [&](uint32_t c) { return c != PT_REF_EMPTY && !(c & PT_REF_LEAF) && c < n; };
    if (by_area) {
        if (a.child0 != PT_REF_EMPTY) put(a, 0);
        if (a.child1 != PT_REF_EMPTY) put(a, 1);
        while (k < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (int j = 0; j < k; j++) {
                if (!inner(child[j])) continue;
                const float dx = hi[j][0] - lo[j][0], dy = hi[j][1] - lo[j][1], dz = hi[j][2] - lo[j][2];
                const float area = dx * dy + dy * dz + dz * dx;
                if (!(area <= best_area)) { best = j; best_area = area; }  // also takes a NaN / infinite area rather than nothing
            }
            if (best < 0) break;
            const PtBvhNode c = bvh2[child[best]];
            const int n_kids = (c.child0 != PT_REF_EMPTY) + (c.child1 != PT_REF_EMPTY);
            if (n_kids == 0) { child[best] = child[k - 1]; for (int ax = 0; ax < 3; ax++) { lo[best][ax] = lo[k - 1][ax]; hi[best][ax] = hi[k - 1][ax]; } k--; continue; }
            // the first child takes the opened entry's place, the second goes to the end
            const int first = c.child0 != PT_REF_EMPTY ? 0 : 1;
            pt_node_get_box(c, first, lo[best], hi[best]);
            child[best] = first ? c.child1 : c.child0;
            if (n_kids == 2) put(c, 1);
        }
    } else {
        auto expand = [&](const PtBvhNode& nd, int which) {
            const uint32_t c0 = which ? nd.child1 : nd.child0;
            if (c0 == PT_REF_EMPTY) return;
            if (!inner(c0)) { put(nd, which); return; }
            const PtBvhNode c = bvh2[c0];
            if (c.child0 != PT_REF_EMPTY) put(c, 0);
            if (c.child1 != PT_REF_EMPTY) put(c, 1);
        };
        expand(a, 0);
        expand(a, 1);
    }
    PtBvh4Node o;
    for (int j = 0; j < 4; j++) {
        const bool used = j < k;
        for (int ax = 0; ax < 3; ax++) { o.lo[ax][j] = used ? lo[j][ax] : (float)PT_BOX_LIMIT; o.hi[ax][j] = used ? hi[j][ax] : -(float)PT_BOX_LIMIT; }
        o.child[j] = used ? child[j] : PT_REF_EMPTY;
    }
    o.pad[0] = o.pad[1] = o.pad[2] = o.pad[3] = 0u;
    bvh4[i] = o;
}

// Second pass of a render: chunk sums -> pixels, a block of PT_BLOCK threads per PT_BLOCK pixel slots. The chunk sums are pixel-major (the
// render kernel's chunk-first lanes write one pixel's sums side by side), so a thread per pixel would read 24 bytes every 24 x n_chunks
// bytes; instead the block first adds them up EIGHT THREADS PER PIXEL - thread j of a pixel holds chunk 8 b + j of the b-th block of eight,
// its seven neighbours' loads next to it, and the pixel's first thread adds the eight values in ascending order (the summation contract: a
// fixed left-to-right association) - leaves the sums in LDS, and then finishes ONE pixel per thread (the three pows with every lane busy).
__global__ void __launch_bounds__(PT_BLOCK) pt_finish_kernel(PtRenderArgs a) {
#ifdef PT_ACCUM_CHUNK_MAJOR
    uint32_t p = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (p < a.n_slots) pt_finish_pixel(a, p, pt_pixel_sum(a, p));
#else
    __shared__ double sums[3 * PT_BLOCK];
    const uint32_t base = blockIdx.x * PT_BLOCK, j = threadIdx.x & 7u;
    for (uint32_t pass = 0; pass < 8u; pass++) {
        const uint32_t local = pass * (PT_BLOCK / 8u) + (threadIdx.x >> 3), p = base + local;
        PtVec3 sum = pt_v3(0.0, 0.0, 0.0);
        for (uint32_t b = 0; b < a.n_chunks; b += 8u) {  // (wave-uniform trip count)
            const uint32_t k = b + j;
            PtVec3 v = pt_v3(0.0, 0.0, 0.0);
            if (p < a.n_slots && k < a.n_chunks) { const double* c = a.accum + 3 * ((size_t)p * a.n_chunks + k); v = pt_v3(c[0], c[1], c[2]); }
            const uint32_t n = a.n_chunks - b < 8u ? a.n_chunks - b : 8u;
            for (uint32_t i = 0; i < n; i++) {  // chunk b + i of this pixel, from thread i of its eight
                const int src = (int)((threadIdx.x & 63u & ~7u) + i);
                const PtVec3 w = pt_v3(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src));
                sum = (b == 0u && i == 0u) ? w : sum + w;
            }
        }
        if (j == 0u) { sums[3 * local] = sum.x; sums[3 * local + 1] = sum.y; sums[3 * local + 2] = sum.z; }
    }
    __syncthreads();
    const uint32_t p = base + threadIdx.x;
    if (p < a.n_slots) pt_finish_pixel(a, p, pt_v3(sums[3 * threadIdx.x], sums[3 * threadIdx.x + 1], sums[3 * threadIdx.x + 2]));
#endif
}

// compact (rank-major, tile-major) -> row-major image
__global__ void pt_untile_kernel(PtRenderArgs a, uint32_t slots_per_rank, const uint8_t* gathered, uint8_t* rgb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total = slots_per_rank * a.tile_ranks;
    if (i >= total) return;
    PtRenderArgs r = a;
    r.tile_rank = i / slots_per_rank;
    uint32_t w = i % slots_per_rank, x, y;
    if (!pt_slot_to_pixel(r, w, &x, &y)) return;
    const uint8_t* s = gathered + 3 * (size_t)i;
    uint8_t* d = rgb + 3 * ((size_t)y * a.width + x);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

// ------------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------------
struct PtBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct pt_context {
    int device = 0;
    int n_cu = 0;
    std::string err;
    PtBuf inv, fwd, nrm, info, tri_v, tri_e, tri_leaf, tri_n, meshes, materials, lights, bvh, bvh4, bvh_items, kd, kd_items;
    PtBuf mat_maps, uv_trans, tex, tex_rgb, srgb_lut, tri_uv, texview, mkd, mkd_items;
    PtBuf node_box, kd_box, mkd_box, mkd_item_box, kd_ref;
    PtBuf g_inv, g_fwd, g_nrm, chain_off, chain, dfs_rank, hier_rec, own_inv;  // PT_TRAVERSE_HIER: the scene graph
    PtBuf bg, rgb, linear, misc;  // (pt_render's host-buffer path; misc: pt_test_cast_rays' overflow flag)
    bool needs_spill = false;  // some material is reflective (recursion frames) or the scene has more than 32 lights
    bool spawns = false;       // some material is reflective: hits spawn rays, so the cost of a pixel varies by orders of magnitude
    bool one_ray = false;      // ... and every reflective material is opaque (no index of refraction): a hit spawns at most one ray, the recursion is a chain
    bool forkable = false;     // ... and no hit draws random numbers after the jitter (no area light, no glossy material): refracted subtrees may be walked by other lanes
    uint32_t launch_seq = 0;   // PtRenderArgs::launch_nonce
    bool four_waves_untextured = false;  // ... or, while the scene has no texture maps: any scene with plain Mesh instances
    bool four_waves_hier = false;        // ... or any scene without KDMesh trees in the hierarchical semantics
    bool four_waves = false;   // traversal-heavy scene without reflective materials, flat_scene / hierarchical semantics: a kernel compiled for more than 3 waves per SIMD
    bool five_waves = false;   // ... mesh-free: 5 waves per SIMD (96 registers)
    bool five_waves_mesh = false;  // ... very many triangles in plain Mesh instances, untextured: 5 waves too
    PtSceneView view;
    bool have_scene = false;
    // Up to PT_SLOTS renders of one context may be in flight on a stream (pt_render_device ... pt_render_finish, oldest first): a frame's
    // events and the page of pinned host memory its overflow flag and counters are copied to - asynchronously, right behind its kernels -
    // belong to its slot, so closing a frame is an event wait and a read of host memory: no blocking copy, and the next frame may already
    // be queued behind it (pt_node: frame k + 1 renders while frame k is gathered).
    // Round 5: a slot also owns the launch's WORK buffers (chunk sums, recursion frames, stack overflow columns, work counters / queues / statistics) and a
    // stream of its own (pt_context_stream), so that two frames may be in flight on two queues at once: the render kernels are persistent - a launch's
    // wavefronts retire one by one over the duration of its longest work items (the TAIL: 0.13 ms of a 1.3 ms share of the headline frame, 11 %;
    // profiles/r05/notes.md section 2) - and the next frame's wavefronts take the places they free instead of waiting for the last one.
    struct Slot {
        PtBuf spill, stack_spill, accum, misc;     // misc: work counter + overflow flag (8 B), PtCounters at +256, the work queues behind them
        hipStream_t stream = nullptr;              // pt_context_stream(ctx, slot): non-blocking, created with the context
        hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the kernels (pt_stats.kernel_ms)
        hipEvent_t copy_done = nullptr;            // behind the copy into `host`
        unsigned char* host = nullptr;   // pinned: [0, 8) work counter + overflow flag, [256, 256 + sizeof(PtCounters)) the counters
        bool pending = false, counted = false;
        uint32_t mode = 0, variant = 0;  // pt_stats.kernel_mode / kernel_variant of the launch
        std::chrono::steady_clock::time_point t_start;
    };
    static constexpr int PT_SLOTS = 2;
    Slot slot[PT_SLOTS];
    int slot_next = 0;     // the slot the next launch takes
    int slot_oldest = 0;   // the oldest launch not yet closed (== slot_next: none, unless every slot is pending)
    uint32_t last_mode = 0, last_variant = 0;
};
#define PT_SLOT_BYTES (256 + sizeof(PtCounters))

static int pt_fail(pt_context* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define PT_HIP(c, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return pt_fail(c, PT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static int pt_reserve(pt_context* c, PtBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return PT_OK;
    if (b.p) { PT_HIP(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    PT_HIP(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return PT_OK;
}
template <class T>
static int pt_upload(pt_context* c, PtBuf& b, const std::vector<T>& v) {
    int rc = pt_reserve(c, b, v.size() * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) PT_HIP(c, hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

extern "C" int pt_abi_version(void) { return PT_ABI_VERSION; }

extern "C" int pt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int pt_context_create(int device, pt_context** out) {
    if (!out) return PT_ERR_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PT_ERR_DEVICE;
    pt_context* c = new pt_context();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return PT_ERR_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return PT_ERR_DEVICE; }
    c->n_cu = prop.multiProcessorCount;
    for (auto& sl : c->slot)
        if (hipEventCreate(&sl.ev0) != hipSuccess || hipEventCreate(&sl.ev1) != hipSuccess || hipEventCreateWithFlags(&sl.copy_done, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess ||
            hipHostMalloc((void**)&sl.host, PT_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) {
            pt_context_destroy(c);
            return PT_ERR_DEVICE;
        }
    memset(&c->view, 0, sizeof c->view);
    *out = c;
    return PT_OK;
}

extern "C" void pt_context_destroy(pt_context* c) {
    if (!c) return;
    hipSetDevice(c->device);
    PtBuf* bufs[] = {&c->inv, &c->fwd, &c->nrm, &c->info, &c->tri_v, &c->tri_e, &c->tri_leaf, &c->tri_n, &c->meshes, &c->materials, &c->lights,
                     &c->bvh, &c->bvh4, &c->bvh_items, &c->kd, &c->kd_items, &c->mat_maps, &c->uv_trans, &c->tex, &c->tex_rgb, &c->srgb_lut, &c->tri_uv, &c->texview, &c->mkd, &c->mkd_items, &c->slot[0].spill, &c->slot[0].stack_spill, &c->slot[0].accum, &c->slot[0].misc, &c->slot[1].spill, &c->slot[1].stack_spill, &c->slot[1].accum, &c->slot[1].misc, &c->bg, &c->rgb, &c->linear, &c->misc, &c->node_box, &c->kd_box, &c->mkd_box, &c->mkd_item_box, &c->kd_ref, &c->g_inv, &c->g_fwd, &c->g_nrm, &c->chain_off, &c->chain, &c->dfs_rank, &c->hier_rec, &c->own_inv};
    for (PtBuf* b : bufs) if (b->p) hipFree(b->p);
    static_assert(pt_context::PT_SLOTS == 2, "the buffer list above names both slots");
    for (auto& sl : c->slot) {
        if (sl.stream) { hipStreamSynchronize(sl.stream); hipStreamDestroy(sl.stream); }
        if (sl.ev0) hipEventDestroy(sl.ev0);
        if (sl.ev1) hipEventDestroy(sl.ev1);
        if (sl.copy_done) hipEventDestroy(sl.copy_done);
        if (sl.host) hipHostFree(sl.host);
    }
    delete c;
}

extern "C" const char* pt_last_error(const pt_context* c) { return c ? c->err.c_str() : "no context"; }

// The context's own stream for the render that takes slot `slot & 1` (the slots are taken in turn: 0, 1, 0, ..): a caller that hands consecutive frames to
// pt_render_device on pt_context_stream(ctx, k & 1) lets frame k + 1 start on the wavefront slots frame k's tail frees (see pt_context::Slot).
// Scenes that park recursion frames in HBM (reflective materials: the chain and interpreter kernels) get ONE stream for both slots: two of their launches at
// once double the frames' working set (277 MB per launch at full occupancy) past what the memory-side cache holds - measured: the mirror scene 18.8 -> 19.6 ms
// per frame with two streams, where big-scene's share of an 8-way split gains 2.4 % and macho-cows 1.5 % (profiles/r05/c12_overlap.txt).
// PORTRAYER_TWO_STREAMS=0 / 1 forces one / two.
extern "C" void* pt_context_stream(pt_context* c, int slot) {
    if (!c) return nullptr;
    bool two = !(c->have_scene && c->needs_spill);
    if (const char* e = getenv("PORTRAYER_TWO_STREAMS")) two = atoi(e) > 0;
    return (void*)c->slot[two ? (slot & 1) : 0].stream;
}
extern "C" int pt_context_next_slot(const pt_context* c) { return c ? c->slot_next : 0; }

// ------------------------------------------------------------------------------------------------
// Scene upload
// ------------------------------------------------------------------------------------------------
static void pt_transform_box(const double* m, const double lo[3], const double hi[3], PtBuildBox* out) {
    *out = pt_bvh_detail::empty_box();
    for (int ix = 0; ix < 2; ix++) for (int iy = 0; iy < 2; iy++) for (int iz = 0; iz < 2; iz++) {
        double x = ix ? hi[0] : lo[0], y = iy ? hi[1] : lo[1], z = iz ? hi[2] : lo[2];
        for (int r = 0; r < 3; r++) {
            double v = m[4 * r] * x + m[4 * r + 1] * y + m[4 * r + 2] * z + m[4 * r + 3];
            out->lo[r] = std::min(out->lo[r], v);
            out->hi[r] = std::max(out->hi[r], v);
        }
    }
}
static void pt_pad_box(PtBuildBox* b, double rel) {
    for (int k = 0; k < 3; k++) {
        double mag = std::max(std::fabs(b->lo[k]), std::fabs(b->hi[k]));
        double pad = rel * std::max(b->hi[k] - b->lo[k], mag) + 1e-300;
        b->lo[k] -= pad; b->hi[k] += pad;
    }
}

extern "C" int pt_scene_upload(pt_context* c, const pt_scene* s, int traverse, const pt_kdtree* kd) {
    if (!c || !s) return PT_ERR_ARGUMENT;
    if (traverse != PT_TRAVERSE_FLAT && traverse != PT_TRAVERSE_KD && traverse != PT_TRAVERSE_HIER)
        return pt_fail(c, PT_ERR_ARGUMENT, "traverse must be PT_TRAVERSE_FLAT, PT_TRAVERSE_KD or PT_TRAVERSE_HIER");
    if (traverse == PT_TRAVERSE_HIER && s->n_nodes &&
        (!s->n_graph_nodes || !s->graph_trans || !s->graph_invtrans || !s->graph_normal_trans || !s->node_chain_off || !s->node_chain || !s->node_dfs_rank))
        return pt_fail(c, PT_ERR_ARGUMENT, "PT_TRAVERSE_HIER needs the scene graph (graph_*, node_chain*, node_dfs_rank)");
    if (traverse == PT_TRAVERSE_KD && !kd) return pt_fail(c, PT_ERR_ARGUMENT, "PT_TRAVERSE_KD needs the host-built k-d tree");
    if (s->n_nodes && (!s->trans || !s->invtrans || !s->normal_trans || !s->prim_type || !s->prim_data || !s->prim_flags || !s->material))
        return pt_fail(c, PT_ERR_ARGUMENT, "null node array");
    PT_HIP(c, hipSetDevice(c->device));
    c->have_scene = false;
    const uint32_t n = s->n_nodes;
    const bool verbose = getenv("PORTRAYER_VERBOSE") != nullptr;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pt_scene_upload] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_begin).count());
        t_begin = now;
    };

    // ---- triangles: every mesh expanded to 72-byte vertex records, stand-alone triangles appended
    std::vector<uint64_t> tri_first(s->n_meshes + 1, 0);
    size_t mesh_tris = s->n_meshes ? (size_t)s->mesh_tri_off[s->n_meshes] : 0;
    size_t total_tris = mesh_tris + s->n_triangles;
    bool any_normals = false;
    for (uint32_t i = 0; i < n; i++) {
        int t = s->prim_type[i];
        if (t < PT_PRIM_SPHERE || t > PT_PRIM_CONE) return pt_fail(c, PT_ERR_ARGUMENT, "unknown primitive type");
        if (s->material[i] < 0 || (uint32_t)s->material[i] >= s->n_materials) return pt_fail(c, PT_ERR_ARGUMENT, "material index out of range");
        if (t == PT_PRIM_MESH || t == PT_PRIM_KDMESH) {
            if (s->prim_data[i] < 0 || (uint32_t)s->prim_data[i] >= s->n_meshes) return pt_fail(c, PT_ERR_ARGUMENT, "mesh index out of range");
            if (s->prim_flags[i] & 1) {
                if (!s->mesh_has_normals || !s->mesh_has_normals[s->prim_data[i]] || !s->mesh_normals)
                    return pt_fail(c, PT_ERR_SCENE, "smooth shading needs a vertex normal per vertex (mesh.rs:135-138)");
                any_normals = true;
            }
        } else if (t == PT_PRIM_TRIANGLE) {
            if (s->prim_data[i] < 0 || (uint32_t)s->prim_data[i] >= s->n_triangles) return pt_fail(c, PT_ERR_ARGUMENT, "triangle index out of range");
            if (s->prim_flags[i] & 1) { if (!s->tri_normals) return pt_fail(c, PT_ERR_SCENE, "triangle normals missing"); any_normals = true; }
        }
    }
    std::vector<double> tri_v(total_tris * 9), tri_n(any_normals ? total_tris * 9 : 0);
    std::vector<PtMeshInfo> meshes(s->n_meshes);
    std::vector<PtBvhNode> bvh;
    std::vector<uint32_t> items;
    std::vector<PtBuildBox> mesh_box(s->n_meshes);
    int max_blas_depth = 0;
    int collapse_mode = 1;  // how two-child trees become four-child trees (pt_collapse4_kernel): 0 grandchildren, 1 mesh trees opened by area, 2 every tree by area
    int blas_leaf = 2;  // triangles per mesh-tree leaf (measured: 2 beats 1, 3 and 4 by 2-5 % on cows / big-soup)
    if (const char* e = getenv("PORTRAYER_BLAS_LEAF")) blas_leaf = std::min(std::max(1, atoi(e)), 8);
    // Where a mesh's triangle tree is built: on the host (binned SAH, pt_bvh.h: the better tree) or on the
    // device (Morton-order tree, pt_build.h: ~100x faster to build). PORTRAYER_BUILD = host | device | auto;
    // auto takes the device for meshes of PORTRAYER_BUILD_MIN (default 65536) triangles or more.
    int build_mode = 2;
    size_t device_build_min = 65536;
    if (const char* e = getenv("PORTRAYER_BUILD")) build_mode = !strcmp(e, "host") ? 0 : (!strcmp(e, "device") ? 1 : 2);
    if (const char* e = getenv("PORTRAYER_BUILD_MIN")) device_build_min = (size_t)std::max(16, atoi(e));
    struct DeviceMesh { uint32_t m, t0, count; double lo[3], hi[3], pad; };
    std::vector<DeviceMesh> device_meshes;
    for (uint32_t m = 0; m < s->n_meshes; m++) {
        uint64_t v0 = s->mesh_vert_off[m], v1 = s->mesh_vert_off[m + 1], t0 = s->mesh_tri_off[m], t1 = s->mesh_tri_off[m + 1];
        if (v1 <= v0) return pt_fail(c, PT_ERR_SCENE, "meshes must have at least one vertex (mesh.rs:71)");
        const double* pos = s->mesh_positions + 3 * v0;
        const double* nrm = (s->mesh_normals && s->mesh_has_normals && s->mesh_has_normals[m]) ? s->mesh_normals + 3 * v0 : nullptr;
        PtBuildBox mb = pt_bvh_detail::empty_box();
        for (uint64_t v = 0; v < v1 - v0; v++)
            for (int k = 0; k < 3; k++) { mb.lo[k] = std::min(mb.lo[k], pos[3 * v + k]); mb.hi[k] = std::max(mb.hi[k], pos[3 * v + k]); }
        double ext = std::max(std::max(mb.hi[0] - mb.lo[0], mb.hi[1] - mb.lo[1]), std::max(mb.hi[2] - mb.lo[2], 1e-30));
        const bool on_device = (build_mode == 1 && t1 - t0 >= 16) || (build_mode == 2 && t1 - t0 >= device_build_min);
        std::vector<PtBuildBox> boxes(on_device ? 0 : t1 - t0);
        std::vector<uint32_t> ids(on_device ? 0 : t1 - t0);
        for (uint64_t t = t0; t < t1; t++) {
            PtBuildBox b = pt_bvh_detail::empty_box();
            for (int corner = 0; corner < 3; corner++) {
                uint32_t vi = s->mesh_indices[3 * t + corner];
                if (vi >= v1 - v0) return pt_fail(c, PT_ERR_ARGUMENT, "mesh index out of range");
                for (int k = 0; k < 3; k++) {
                    double x = pos[3 * (size_t)vi + k];
                    tri_v[9 * t + 3 * corner + k] = x;
                    if (any_normals) tri_n[9 * t + 3 * corner + k] = nrm ? nrm[3 * (size_t)vi + k] : 0.0;
                    b.lo[k] = std::min(b.lo[k], x); b.hi[k] = std::max(b.hi[k], x);
                }
            }
            if (on_device) continue;
            for (int k = 0; k < 3; k++) { b.lo[k] -= 1e-7 * ext; b.hi[k] += 1e-7 * ext; }
            boxes[t - t0] = b;
            ids[t - t0] = (uint32_t)t;
        }
        PtBvhRef ref;
        ref.child = PT_REF_EMPTY; ref.depth = 0;
        if (on_device) {  // built after the triangles are in HBM (below); the root is filled in then
            DeviceMesh dm;
            dm.m = m; dm.t0 = (uint32_t)t0; dm.count = (uint32_t)(t1 - t0);
            for (int k = 0; k < 3; k++) { dm.lo[k] = mb.lo[k]; dm.hi[k] = mb.hi[k]; }
            dm.pad = 1e-7 * ext;
            device_meshes.push_back(dm);
        } else {
            ref = pt_bvh_build(boxes.data(), ids.data(), boxes.size(), blas_leaf, bvh, items);
        }
        max_blas_depth = std::max(max_blas_depth, ref.depth);
        PtMeshInfo& mi = meshes[m];
        for (int r = 0; r < 12; r++) mi.bbox_inv[r] = s->mesh_bounds_invtrans ? s->mesh_bounds_invtrans[16 * (size_t)m + r] : 0.0;
        if (!s->mesh_bounds_invtrans) return pt_fail(c, PT_ERR_ARGUMENT, "mesh_bounds_invtrans missing");
        mi.tri_first = (uint32_t)t0; mi.tri_count = (uint32_t)(t1 - t0);
        mi.blas_root = ref.child; mi.kd_root = -1; mi.kd_extent = 0.0;
        for (int r = 0; r < 12; r++) mi.kd_bbox_inv[r] = 0.0;
        for (int k = 0; k < 3; k++) { mb.lo[k] -= 1e-5 * ext; mb.hi[k] += 1e-5 * ext; }
        mesh_box[m] = mb;
    }
    lap("triangles + mesh trees");
    // ---- KDMesh triangle trees (reference structure)
    std::vector<PtKdNode> mkd;
    std::vector<uint32_t> mkd_items;
    std::vector<float> mkd_box, mkd_item_box;  // conservative f32 bounds per KDMesh tree node / per leaf reference (pt_kdmesh_hit's culls)
    int max_kdm_depth = 0;
    bool any_kdmesh = false;
    if (s->mesh_kd_root && s->n_kdm_nodes) {
        if (!s->kdm_axis || !s->kdm_plane || !s->kdm_front || !s->kdm_back || !s->kdm_first || !s->kdm_count || (s->n_kdm_items && !s->kdm_items) ||
            !s->mesh_kd_bounds || !s->mesh_kd_bounds_invtrans)
            return pt_fail(c, PT_ERR_ARGUMENT, "incomplete KDMesh tree arrays");
        mkd.resize(s->n_kdm_nodes);
        for (uint32_t i = 0; i < s->n_kdm_nodes; i++) {
            PtKdNode& k = mkd[i];
            k.axis = s->kdm_axis[i]; k.plane = s->kdm_plane[i]; k.front = s->kdm_front[i]; k.back = s->kdm_back[i];
            k.first = s->kdm_first[i]; k.count = s->kdm_count[i]; k.pad = 0; k.pad2[0] = k.pad2[1] = 0.0f; for (int r = 0; r < 6; r++) k.box[r] = 0.0f;
            if (k.axis >= 0) {
                if (k.axis > 2 || k.front < 0 || k.back < 0 || (uint32_t)k.front >= s->n_kdm_nodes || (uint32_t)k.back >= s->n_kdm_nodes)
                    return pt_fail(c, PT_ERR_ARGUMENT, "KDMesh tree child out of range");
            } else if (k.first < 0 || k.count < 0 || (uint32_t)(k.first + k.count) > s->n_kdm_items) {
                return pt_fail(c, PT_ERR_ARGUMENT, "KDMesh leaf range out of bounds");
            }
        }
        mkd_items.assign(s->n_kdm_items, 0);
        mkd_box.assign(6 * (size_t)s->n_kdm_nodes, 0.0f);
        mkd_item_box.assign(6 * (size_t)s->n_kdm_items, 0.0f);
        // leaf items are local triangle indices: make them global, mesh by mesh (a leaf belongs to the mesh whose tree reaches it)
        std::vector<int32_t> owner(s->n_kdm_nodes, -1);
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            int32_t root = s->mesh_kd_root[m];
            if (root < 0) continue;
            if ((uint32_t)root >= s->n_kdm_nodes) return pt_fail(c, PT_ERR_ARGUMENT, "KDMesh root out of range");
            std::vector<int32_t> todo{root}, visited;
            const double tri_pad = 1e-7 * std::max(std::max(mesh_box[m].hi[0] - mesh_box[m].lo[0], mesh_box[m].hi[1] - mesh_box[m].lo[1]),
                                                   std::max(mesh_box[m].hi[2] - mesh_box[m].lo[2], 1e-30));
            while (!todo.empty()) {
                int32_t i = todo.back(); todo.pop_back();
                if (owner[i] >= 0) return pt_fail(c, PT_ERR_ARGUMENT, "KDMesh trees must not share nodes");
                owner[i] = (int32_t)m;
                visited.push_back(i);
                if (mkd[i].axis >= 0) { todo.push_back(mkd[i].front); todo.push_back(mkd[i].back); }
                else for (int32_t k = 0; k < mkd[i].count; k++) {
                    int32_t local = s->kdm_items[mkd[i].first + k];
                    if (local < 0 || (uint64_t)local >= s->mesh_tri_off[m + 1] - s->mesh_tri_off[m]) return pt_fail(c, PT_ERR_ARGUMENT, "KDMesh leaf triangle out of range");
                    const uint64_t g = s->mesh_tri_off[m] + (uint64_t)local;
                    mkd_items[mkd[i].first + k] = (uint32_t)g;
                    float* ib = &mkd_item_box[6 * (size_t)(mkd[i].first + k)];  // the triangle's padded box, for the walk's pre-cull
                    for (int r = 0; r < 3; r++) {
                        const double* v = &tri_v[9 * g];
                        ib[r] = pt_bvh_detail::round_down(std::min(v[r], std::min(v[3 + r], v[6 + r])) - tri_pad);
                        ib[3 + r] = pt_bvh_detail::round_up(std::max(v[r], std::max(v[3 + r], v[6 + r])) + tri_pad);
                    }
                }
            }
            for (size_t vi = visited.size(); vi-- > 0;) {  // children were discovered after their parents: bounds bottom-up
                const int32_t i = visited[vi];
                float* b = &mkd_box[6 * (size_t)i];
                for (int r = 0; r < 3; r++) { b[r] = (float)PT_BOX_LIMIT; b[3 + r] = -(float)PT_BOX_LIMIT; }
                auto grow = [&](const float* o) { for (int r = 0; r < 3; r++) { b[r] = std::min(b[r], o[r]); b[3 + r] = std::max(b[3 + r], o[3 + r]); } };
                if (mkd[i].axis >= 0) { grow(&mkd_box[6 * (size_t)mkd[i].front]); grow(&mkd_box[6 * (size_t)mkd[i].back]); }
                else for (int32_t k = 0; k < mkd[i].count; k++) grow(&mkd_item_box[6 * (size_t)(mkd[i].first + k)]);
            }
            PtMeshInfo& mi = meshes[m];
            mi.kd_root = root;
            any_kdmesh = true;
            const double* b = s->mesh_kd_bounds + 6 * (size_t)m;
            double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            mi.kd_extent = (dx * dx + dy * dy) + dz * dz;  // bounding_box.rs:95-99
            for (int r = 0; r < 12; r++) mi.kd_bbox_inv[r] = s->mesh_kd_bounds_invtrans[16 * (size_t)m + r];
            max_kdm_depth = std::max(max_kdm_depth, s->mesh_kd_depth ? std::max(s->mesh_kd_depth[m], 0) : 24);
        }
    }
    for (uint32_t t = 0; t < s->n_triangles; t++)
        for (int k = 0; k < 9; k++) {
            tri_v[9 * (mesh_tris + t) + k] = s->tri_vertices[9 * (size_t)t + k];
            if (any_normals) tri_n[9 * (mesh_tris + t) + k] = s->tri_normals ? s->tri_normals[9 * (size_t)t + k] : 0.0;
        }

    // ---- nodes
    std::vector<double> inv(12 * (size_t)n), fwd(12 * (size_t)n), nrm(9 * (size_t)n);
    std::vector<uint32_t> info(4 * (size_t)n);
    std::vector<double> g_inv, g_fwd, g_nrm;
    std::vector<uint32_t> chain_off, chain, dfs_rank, hier_rec;
    std::vector<double> own_inv;
    if (traverse == PT_TRAVERSE_HIER && n) {
        const uint32_t g = s->n_graph_nodes;
        g_inv.resize(12 * (size_t)g); g_fwd.resize(12 * (size_t)g); g_nrm.resize(9 * (size_t)g);
        for (uint32_t i = 0; i < g; i++) {
            for (int r = 0; r < 12; r++) { g_inv[12 * (size_t)i + r] = s->graph_invtrans[16 * (size_t)i + r]; g_fwd[12 * (size_t)i + r] = s->graph_trans[16 * (size_t)i + r]; }
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) g_nrm[9 * (size_t)i + 3 * r + k] = s->graph_normal_trans[16 * (size_t)i + 4 * r + k];
        }
        chain_off.assign(s->node_chain_off, s->node_chain_off + n + 1);
        if (chain_off[0] != 0) return pt_fail(c, PT_ERR_ARGUMENT, "node_chain_off must start at 0");
        for (uint32_t i = 0; i < n; i++)
            if (chain_off[i + 1] <= chain_off[i]) return pt_fail(c, PT_ERR_ARGUMENT, "every flattened node needs a non-empty path (it contains at least itself)");
        chain.assign(s->node_chain, s->node_chain + chain_off[n]);
        for (uint32_t id : chain) if (id >= g) return pt_fail(c, PT_ERR_ARGUMENT, "node_chain names a graph node out of range");
        dfs_rank.assign(s->node_dfs_rank, s->node_dfs_rank + n);
        // One record per flattened node for the walks (pt_node_local_ray_uniform): its path in one scalar fetch instead of three dependent
        // ones, and which of its levels are the identity (a group without a transform, like every reference scene's root): multiplying a
        // ray by such a level changes no bit unless a component is -0 or not finite, which the walk rules out once per ray.
        auto identity = [&](uint32_t gi) {
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 4; k++) {
                    const double want = r == k ? 1.0 : 0.0;  // (== : a zero of either sign passes)
                    if (g_inv[12 * (size_t)gi + 4 * r + k] != want || g_fwd[12 * (size_t)gi + 4 * r + k] != want) return false;
                    if (k < 3 && g_nrm[9 * (size_t)gi + 3 * r + k] != want) return false;
                }
            return true;
        };
        std::vector<uint8_t> g_ident(g);
        for (uint32_t i = 0; i < g; i++) g_ident[i] = identity(i) ? 1 : 0;
        hier_rec.assign(8 * (size_t)n, 0u);
        own_inv.assign(12 * (size_t)n, 0.0);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t len = chain_off[i + 1] - chain_off[i];
            uint32_t* rec = &hier_rec[8 * (size_t)i];
            for (int r = 0; r < 12; r++) own_inv[12 * (size_t)i + r] = g_inv[12 * (size_t)chain[chain_off[i + 1] - 1] + r];  // the node's own level: the last of its path
            if (len > 7) { rec[0] = 255u; continue; }
            rec[0] = len;
            for (uint32_t k = 0; k < len; k++) {
                const uint32_t gi = chain[chain_off[i] + k];
                rec[1 + k] = gi;
                if (g_ident[gi]) rec[0] |= 1u << (8 + k);
            }
        }
    }
    std::vector<PtBuildBox> node_box(n);
    for (uint32_t i = 0; i < n; i++) {
        for (int r = 0; r < 12; r++) { inv[12 * (size_t)i + r] = s->invtrans[16 * (size_t)i + r]; fwd[12 * (size_t)i + r] = s->trans[16 * (size_t)i + r]; }
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) nrm[9 * (size_t)i + 3 * r + k] = s->normal_trans[16 * (size_t)i + 4 * r + k];
        int t = s->prim_type[i];
        uint32_t data = (uint32_t)s->prim_data[i];
        if (t == PT_PRIM_TRIANGLE) data = (uint32_t)(mesh_tris + data);
        info[4 * (size_t)i] = (uint32_t)t; info[4 * (size_t)i + 1] = data;
        info[4 * (size_t)i + 2] = (uint32_t)s->prim_flags[i]; info[4 * (size_t)i + 3] = (uint32_t)s->material[i];
        // Conservative world-space box: every hit the primitive tests can accept lies inside it
        // (cube.rs:25 / plane.rs:31 accept points up to 1e-5 outside the unit shape; all shapes are
        // padded by 1e-4 model units). Spheres, cylinders and cones get their exact boxes under the
        // affine transform instead of the box of their transformed unit cube.
        const double* M = &s->trans[16 * (size_t)i];
        PtBuildBox& nb = node_box[i];
        auto disc = [&](double yc, double radius, PtBuildBox* b) {  // disc of `radius` in the model xz-plane at height yc
            for (int r = 0; r < 3; r++) {
                double c = M[4 * r + 1] * yc + M[4 * r + 3];
                double e = radius * std::sqrt(M[4 * r] * M[4 * r] + M[4 * r + 2] * M[4 * r + 2]);
                b->lo[r] = std::min(b->lo[r], c - e); b->hi[r] = std::max(b->hi[r], c + e);
            }
        };
        double lo[3], hi[3];
        bool boxed = false;
        switch (t) {
        case PT_PRIM_SPHERE:
            for (int r = 0; r < 3; r++) {
                double e = 1.0001 * std::sqrt(M[4 * r] * M[4 * r] + M[4 * r + 1] * M[4 * r + 1] + M[4 * r + 2] * M[4 * r + 2]);
                nb.lo[r] = M[4 * r + 3] - e; nb.hi[r] = M[4 * r + 3] + e;
            }
            boxed = true; break;
        case PT_PRIM_CYLINDER: nb = pt_bvh_detail::empty_box(); disc(0.5001, 0.5001, &nb); disc(-0.5001, 0.5001, &nb); boxed = true; break;
        case PT_PRIM_CONE: nb = pt_bvh_detail::empty_box(); disc(0.5001, 1e-4, &nb); disc(-0.5001, 0.5001, &nb); boxed = true; break;
        case PT_PRIM_PLANE: lo[0] = lo[2] = -0.5001; hi[0] = hi[2] = 0.5001; lo[1] = -1e-4; hi[1] = 1e-4; break;
        case PT_PRIM_MESH: case PT_PRIM_KDMESH: for (int k = 0; k < 3; k++) { lo[k] = mesh_box[data].lo[k]; hi[k] = mesh_box[data].hi[k]; } break;
        case PT_PRIM_TRIANGLE: {
            const double* v = &tri_v[9 * (size_t)data];
            double ext = 1e-30;
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(v[k], std::min(v[3 + k], v[6 + k])); hi[k] = std::max(v[k], std::max(v[3 + k], v[6 + k]));
                ext = std::max(ext, hi[k] - lo[k]);
            }
            for (int k = 0; k < 3; k++) { lo[k] -= 1e-6 * ext; hi[k] += 1e-6 * ext; }
            break;
        }
        default: lo[0] = lo[1] = lo[2] = -0.5001; hi[0] = hi[1] = hi[2] = 0.5001; break;  // cube
        }
        if (!boxed) pt_transform_box(M, lo, hi, &nb);
        pt_pad_box(&nb, 1e-9);
    }
    int tlas_leaf = 1;  // primitive tests (f64, ~200 instructions) cost far more than a node visit: measured best on big-scene
    if (const char* e = getenv("PORTRAYER_TLAS_LEAF")) tlas_leaf = std::max(1, atoi(e));
    const bool tlas_direct = tlas_leaf == 1 && n < (1u << 28);
    const size_t tlas_first = bvh.size();
    PtBvhRef tlas = pt_bvh_build(node_box.data(), nullptr, n, tlas_leaf, bvh, items, tlas_direct);
    const size_t tlas_end = bvh.size();

    // ---- k-d tree (reference structure, KD mode)
    std::vector<PtKdNode> kdn;
    std::vector<uint32_t> kdi;
    double kd_extent = 0.0;
    int kd_depth = 0;
    if (traverse == PT_TRAVERSE_KD) {
        if (kd->n_nodes == 0 || !kd->axis || !kd->plane || !kd->front || !kd->back || !kd->first || !kd->count || (kd->n_items && !kd->leaf_items))
            return pt_fail(c, PT_ERR_ARGUMENT, "incomplete k-d tree");
        kdn.resize(kd->n_nodes);
        for (uint32_t i = 0; i < kd->n_nodes; i++) {
            PtKdNode& k = kdn[i];
            k.axis = kd->axis[i]; k.plane = kd->plane[i]; k.front = kd->front[i]; k.back = kd->back[i];
            k.first = kd->first[i]; k.count = kd->count[i]; k.pad = 0; k.pad2[0] = k.pad2[1] = 0.0f; for (int r = 0; r < 6; r++) k.box[r] = 0.0f;
            if (k.axis >= 0) {
                if (k.axis > 2 || k.front < 0 || k.back < 0 || (uint32_t)k.front >= kd->n_nodes || (uint32_t)k.back >= kd->n_nodes)
                    return pt_fail(c, PT_ERR_ARGUMENT, "k-d child out of range");
            } else if (k.first < 0 || k.count < 0 || (uint32_t)(k.first + k.count) > kd->n_items) {
                return pt_fail(c, PT_ERR_ARGUMENT, "k-d leaf range out of bounds");
            }
        }
        kdi.resize(kd->n_items);
        for (uint32_t i = 0; i < kd->n_items; i++) {
            if (kd->leaf_items[i] < 0 || (uint32_t)kd->leaf_items[i] >= n) return pt_fail(c, PT_ERR_ARGUMENT, "k-d leaf item out of range");
            kdi[i] = (uint32_t)kd->leaf_items[i];
        }
        double dx = kd->root_max[0] - kd->root_min[0], dy = kd->root_max[1] - kd->root_min[1], dz = kd->root_max[2] - kd->root_min[2];
        kd_extent = (dx * dx + dy * dy) + dz * dz;  // bounding_box.rs:95-99 magnitude_squared
        kd_depth = kd->max_depth < 0 ? 0 : kd->max_depth;
    }

    int rc;
    std::vector<float> node_box32;
    if (traverse == PT_TRAVERSE_KD) {
        node_box32.resize(6 * kdi.size());  // in leaf-item order: the walk reads a leaf's boxes one after the other
        for (size_t j = 0; j < kdi.size(); j++)
            for (int k = 0; k < 3; k++) {
                node_box32[6 * j + k] = pt_bvh_detail::round_down(node_box[kdi[j]].lo[k]);
                node_box32[6 * j + 3 + k] = pt_bvh_detail::round_up(node_box[kdi[j]].hi[k]);
            }
    }
    // the wave-uniform k-d walk (pt_trace_packet_kd) reads a leaf reference and its cull box in ONE scalar fetch: 32 bytes {node, 0, box}
    std::vector<uint32_t> kd_ref32;
    int kd_levels = 0;  // split levels on the deepest path of the tree (root = level 0)
    if (traverse == PT_TRAVERSE_KD) {
        kd_ref32.resize(8 * kdi.size());
        for (size_t j = 0; j < kdi.size(); j++) {
            kd_ref32[8 * j] = kdi[j]; kd_ref32[8 * j + 1] = 0u;
            memcpy(&kd_ref32[8 * j + 2], &node_box32[6 * j], 6 * sizeof(float));
        }
        std::vector<std::pair<uint32_t, int>> todo;
        std::vector<uint8_t> seen(kdn.size(), 0);
        if (!kdn.empty()) todo.push_back({0u, 0});
        while (!todo.empty()) {
            auto [i, lev] = todo.back(); todo.pop_back();
            if (seen[i]) return pt_fail(c, PT_ERR_ARGUMENT, "k-d tree is not a tree");
            seen[i] = 1;
            if (kdn[i].axis < 0) continue;
            kd_levels = std::max(kd_levels, lev + 1);
            todo.push_back({(uint32_t)kdn[i].front, lev + 1}); todo.push_back({(uint32_t)kdn[i].back, lev + 1});
        }
    }
    if (kd_levels > PT_KD_WAVE_LEVELS)  // two bits of per-lane state per level in one 64-bit word (pt_trace_packet_kd); a tree that deep has > 2^32 leaves unless it is a degenerate chain
        return pt_fail(c, PT_ERR_SCENE, "k-d tree deeper than 32 levels (the limit of the k-d walk: include/portrayer_hip.h, pt_kdtree)");
    // what the walk's packed words can address: a stack entry is (node << 5) | level, a node is fetched at byte offset node << 6, a leaf reference at (first + i) << 5
    if (kdn.size() >= ((size_t)1 << 26)) return pt_fail(c, PT_ERR_SCENE, "k-d tree of 2^26 nodes or more");
    if (kd_ref32.size() / 8 >= ((size_t)1 << 27)) return pt_fail(c, PT_ERR_SCENE, "k-d tree with 2^27 leaf references or more");
    std::vector<float> kd_box32;
    if (traverse == PT_TRAVERSE_KD) {  // children follow their parents in the linearised tree (pre-order): one backward sweep
        kd_box32.assign(6 * kdn.size(), 0.0f);
        for (size_t i = kdn.size(); i-- > 0;) {
            float* b = &kd_box32[6 * i];
            const PtKdNode& k = kdn[i];
            for (int r = 0; r < 3; r++) { b[r] = (float)PT_BOX_LIMIT; b[3 + r] = -(float)PT_BOX_LIMIT; }  // empty
            if (k.axis < 0) {
                for (int32_t j = 0; j < k.count; j++)
                    for (int r = 0; r < 3; r++) {
                        b[r] = std::min(b[r], node_box32[6 * (size_t)(k.first + j) + r]);
                        b[3 + r] = std::max(b[3 + r], node_box32[6 * (size_t)(k.first + j) + 3 + r]);
                    }
            } else if ((size_t)k.front > i && (size_t)k.back > i) {
                for (int r = 0; r < 3; r++) {
                    b[r] = std::min(kd_box32[6 * (size_t)k.front + r], kd_box32[6 * (size_t)k.back + r]);
                    b[3 + r] = std::max(kd_box32[6 * (size_t)k.front + 3 + r], kd_box32[6 * (size_t)k.back + 3 + r]);
                }
            } else {  // not in pre-order: no culling at this node
                for (int r = 0; r < 3; r++) { b[r] = -(float)PT_BOX_LIMIT; b[3 + r] = (float)PT_BOX_LIMIT; }
            }
        }
    }
    lap("scene tree");
    if ((rc = pt_upload(c, c->tri_v, tri_v))) return rc;
    {   // the edge form of every triangle (pt_triangle_hit_e): corner a, a - b, a - c
        std::vector<double> tri_e(tri_v.size());
        for (size_t t = 0; t < total_tris; t++) {
            const double* v = &tri_v[9 * t];
            double* e = &tri_e[9 * t];
            e[0] = v[0]; e[1] = v[1]; e[2] = v[2];
            e[3] = v[0] - v[3]; e[4] = v[1] - v[4]; e[5] = v[2] - v[5];
            e[6] = v[0] - v[6]; e[7] = v[1] - v[7]; e[8] = v[2] - v[8];
        }
        if ((rc = pt_upload(c, c->tri_e, tri_e))) return rc;
    }
    {   // tree arrays: the host-built part first, then room for the device-built mesh trees
        size_t n_nodes = bvh.size(), n_items = items.size();
        for (const DeviceMesh& dm : device_meshes) { n_nodes += PT_DEVICE_TREE_NODES(dm.count); n_items += PT_DEVICE_TREE_ITEMS(dm.count, blas_leaf); }
        if (n_items >= (1u << 28) || n_nodes >= (1u << 26)) return pt_fail(c, PT_ERR_SCENE, "too many triangles for the 32-bit tree references (a node is addressed by a 32-bit byte offset: 2^26 nodes)");
        if ((rc = pt_reserve(c, c->bvh, n_nodes * sizeof(PtBvhNode))) || (rc = pt_reserve(c, c->bvh_items, n_items * sizeof(uint32_t)))) return rc;
        if (!bvh.empty()) PT_HIP(c, hipMemcpy(c->bvh.p, bvh.data(), bvh.size() * sizeof(PtBvhNode), hipMemcpyHostToDevice));
        if (!items.empty()) PT_HIP(c, hipMemcpy(c->bvh_items.p, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        lap("upload triangles + host trees");
        uint32_t node_base = (uint32_t)bvh.size(), item_base = (uint32_t)items.size();
        float device_ms = 0.0f;
        int rounds = 0;
        for (const DeviceMesh& dm : device_meshes) {
            PtDeviceBuildResult res;
            PT_HIP(c, pt_device_build_mesh_tree((const double*)c->tri_v.p, dm.t0, dm.count, dm.lo, dm.hi, dm.pad, blas_leaf, (PtBvhNode*)c->bvh.p, node_base,
                                                (uint32_t*)c->bvh_items.p, item_base, nullptr, &res));
            meshes[dm.m].blas_root = res.root;
            max_blas_depth = std::max(max_blas_depth, res.depth);
            node_base += PT_DEVICE_TREE_NODES(dm.count); item_base += PT_DEVICE_TREE_ITEMS(dm.count, blas_leaf);
            device_ms += res.ms; rounds += res.rounds;
        }
        if (verbose && !device_meshes.empty()) fprintf(stderr, "[pt_scene_upload] device tree build: %zu mesh(es), %.2f ms, %d clustering rounds, depth %d\n", device_meshes.size(), device_ms, rounds, max_blas_depth);
        lap("device mesh trees");
        // the walks read the four-child form of every tree (scene tree and mesh trees alike)
        if (const char* e = getenv("PORTRAYER_COLLAPSE")) collapse_mode = strcmp(e, "plain") == 0 ? 0 : (strcmp(e, "area") == 0 ? 2 : 1);
        if ((rc = pt_reserve(c, c->bvh4, std::max<size_t>(n_nodes, 1) * sizeof(PtBvh4Node)))) return rc;
        if (n_nodes) {
            hipLaunchKernelGGL(pt_collapse4_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, nullptr, (const PtBvhNode*)c->bvh.p, (PtBvh4Node*)c->bvh4.p, (uint32_t)n_nodes, collapse_mode, (uint32_t)tlas_first, (uint32_t)tlas_end);
            PT_HIP(c, hipGetLastError());
            PT_HIP(c, hipDeviceSynchronize());
        }
        lap("four-child form");
        if ((rc = pt_reserve(c, c->tri_leaf, std::max<size_t>(n_items, 1) * 80))) return rc;
        if (n_items && total_tris) {
            hipLaunchKernelGGL(pt_tri_leaf_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, nullptr, (const uint32_t*)c->bvh_items.p, (uint32_t)n_items, (const double*)c->tri_e.p,
                               (uint32_t)total_tris, (double*)c->tri_leaf.p);
            PT_HIP(c, hipGetLastError());
            PT_HIP(c, hipDeviceSynchronize());
        }
        lap("triangle records in leaf order");
    }
    for (size_t i = 0; i < kdn.size() && !kd_box32.empty(); i++) for (int r = 0; r < 6; r++) kdn[i].box[r] = kd_box32[6 * i + r];
    for (size_t i = 0; i < mkd.size() && !mkd_box.empty(); i++) for (int r = 0; r < 6; r++) mkd[i].box[r] = mkd_box[6 * i + r];
    if ((rc = pt_upload(c, c->g_inv, g_inv)) || (rc = pt_upload(c, c->g_fwd, g_fwd)) || (rc = pt_upload(c, c->g_nrm, g_nrm)) ||
        (rc = pt_upload(c, c->chain_off, chain_off)) || (rc = pt_upload(c, c->chain, chain)) || (rc = pt_upload(c, c->dfs_rank, dfs_rank)) ||
        (rc = pt_upload(c, c->hier_rec, hier_rec)) || (rc = pt_upload(c, c->own_inv, own_inv)))
        return rc;
    if ((rc = pt_upload(c, c->inv, inv)) || (rc = pt_upload(c, c->fwd, fwd)) || (rc = pt_upload(c, c->nrm, nrm)) ||
        (rc = pt_upload(c, c->info, info)) || (rc = pt_upload(c, c->tri_n, tri_n)) ||
        (rc = pt_upload(c, c->meshes, meshes)) || (rc = pt_upload(c, c->node_box, node_box32)) || (rc = pt_upload(c, c->kd_box, kd_box32)) ||
        (rc = pt_upload(c, c->kd, kdn)) || (rc = pt_upload(c, c->kd_items, kdi)) || (rc = pt_upload(c, c->kd_ref, kd_ref32)) || (rc = pt_upload(c, c->mkd, mkd)) ||
        (rc = pt_upload(c, c->mkd_items, mkd_items)) || (rc = pt_upload(c, c->mkd_box, mkd_box)) || (rc = pt_upload(c, c->mkd_item_box, mkd_item_box)))
        return rc;
    std::vector<double> mats(s->materials, s->materials + 10 * (size_t)s->n_materials);
    c->needs_spill = s->n_lights > PT_LIGHT_ROUND;
    c->spawns = false;
    for (uint32_t m = 0; m < s->n_materials; m++) if (mats[10 * (size_t)m + 7] > 0.0) c->needs_spill = c->spawns = true;  // material.rs:216: reflectivity > 0 spawns children
    // The 4-waves-per-SIMD kernel for scenes where the tree walk outweighs the shading: many scene-level nodes or many instanced
    // triangles, nothing reflective, flat_scene semantics (measured: big-scene +8.7 %, big-soup +6.8 %; macho-cows
    // with its 23 nodes and 17,500 triangles -8 %; reflective scenes and the k-d walk lose, profiles/r02/notes.md)
    {
        uint64_t instanced_tris = 0;
        for (uint32_t i = 0; i < n; i++)
            if (s->prim_type[i] == PT_PRIM_MESH || s->prim_type[i] == PT_PRIM_KDMESH) instanced_tris += s->mesh_tri_off[s->prim_data[i] + 1] - s->mesh_tri_off[s->prim_data[i]];
        // hierarchical semantics (round 3, straight-line kernel): mesh-free scenes with many nodes gain like flat_scene ones (big-scene 25.7 ->
        // 29.4 Gray/s at 4 waves); with mesh instances the walk carries a second ray and spills at 128 registers (macho-cows 16.4 -> 12.0) unless
        // the triangle trees dominate (the 1.25 M-triangle soup: equal)
        // Since the register work of round 3 (arguments re-read, colour parked in LDS, an instantiation of the hierarchical semantics without
        // the KDMesh walker) scenes with plain Mesh instances of any size gain too (c41: macho-cows 21.4 -> 23.4 Gray/s, hierarchical 18.7 -> 19.7);
        // with KDMesh trees the kernels still spill 100+ registers at 128 and stay at 3 waves unless the triangle trees dominate.
        const bool plain_meshes = s->n_meshes > 0 && !any_kdmesh;
        c->four_waves = !c->spawns && ((traverse == PT_TRAVERSE_FLAT && (n >= 256 || instanced_tris >= 65536)) ||
                                       (traverse == PT_TRAVERSE_HIER && ((s->n_meshes == 0 && n >= 256) || instanced_tris >= 65536)));
        c->four_waves_untextured = !c->spawns && plain_meshes && (traverse == PT_TRAVERSE_FLAT || traverse == PT_TRAVERSE_HIER);  // (textured: flat_scene loses 10 % at 4 waves, c45)
        // The hierarchical semantics without KDMesh trees gain at 4 waves whatever the scene (their leaf tests wait for a path record and a matrix per
        // level): macho-cows +5 %, fish (textured) +5 %, normal-mapping (11 textured primitives) +12 %, the mirror scene's chain kernel +9 % (c41, c45).
        c->four_waves_hier = !c->spawns && traverse == PT_TRAVERSE_HIER && !any_kdmesh;
        // mesh-free scenes go one further: 5 waves per SIMD (96 registers, 17 of the kernel's spilled; big-scene 34.4 -> 37.0, hierarchical
        // 29.4 -> 32.5 Gray/s; the k-d walk, 50 spilled, loses and stays at 4)
        c->five_waves = c->four_waves && s->n_meshes == 0 && traverse != PT_TRAVERSE_KD;
        // ... and so do scenes of very many triangles in plain Mesh instances (round 4, c29: their walks wait for node fetches - 3 -> 4 waves was +18 % -; the 1.25 M-triangle
        // scenes +3.7 % / +3.2 %, hierarchical +2.6 %, at 96 registers with 57 spilled; macho-cows, 17,500 triangles, loses 15 % and stays at 4)
        // Round 5 (c47 / c48, after the tree step and the instance walk issue fewer scalar instructions): flat_scene is now FASTER at 4 waves (big-soup 18.96 -> 18.75 ms,
        // big-mesh 19.16 -> 18.86; 18 spilled registers instead of 69) and takes 4; the hierarchical semantics still gain 1 % at 5 and keep them.
        c->five_waves_mesh = !c->spawns && plain_meshes && instanced_tris >= 65536 && traverse == PT_TRAVERSE_HIER;
    }
    std::vector<double> lights(s->lights, s->lights + 15 * (size_t)s->n_lights);
    {   // fork / join of refracted subtrees (pt_shade.h) needs a recursion that draws no random numbers and a dielectric material to be of use
        bool draws = false, dielectric = false;
        for (uint32_t m = 0; m < s->n_materials; m++) {
            if (mats[10 * (size_t)m + 7] > 0.0 && mats[10 * (size_t)m + 8] > 0.0) draws = true;   // glossy reflection (material.rs:221-239)
            if (mats[10 * (size_t)m + 7] > 0.0 && mats[10 * (size_t)m + 9] > 0.0) dielectric = true;
        }
        for (uint32_t l = 0; l < s->n_lights; l++) {
            const double* L = &lights[15 * (size_t)l];
            const bool empty = (L[9] == 0.0 && L[10] == 0.0 && L[11] == 0.0) || (L[12] == 0.0 && L[13] == 0.0 && L[14] == 0.0);  // light.rs:51-53
            if (!empty) draws = true;
        }
        c->forkable = c->spawns && dielectric && !draws && s->n_lights <= PT_LIGHT_ROUND;
        c->one_ray = c->spawns && !dielectric;
    }
    if ((rc = pt_upload(c, c->materials, mats)) || (rc = pt_upload(c, c->lights, lights))) return rc;

    // ---- textures / normal maps (texture.rs)
    bool textured = false;
    for (uint32_t m = 0; m < s->n_materials; m++)
        if ((s->material_texture && s->material_texture[m] >= 0) || (s->material_normal_map && s->material_normal_map[m] >= 0)) textured = true;
    std::vector<int32_t> mat_maps;
    std::vector<double> uv_trans, srgb_lut, tri_uv;
    std::vector<PtTexInfo> texinfo;
    std::vector<uint8_t> tex_rgb;
    if (textured) {
        if (s->n_textures == 0 || !s->texture_size || !s->texture_offset || !s->texture_rgb) return pt_fail(c, PT_ERR_ARGUMENT, "textured material without texture data");
        size_t tex_bytes = 0;
        texinfo.resize(s->n_textures);
        for (uint32_t t = 0; t < s->n_textures; t++) {
            texinfo[t].offset = s->texture_offset[t]; texinfo[t].width = s->texture_size[2 * t]; texinfo[t].height = s->texture_size[2 * t + 1];
            if (texinfo[t].width == 0 || texinfo[t].height == 0) return pt_fail(c, PT_ERR_ARGUMENT, "empty texture");
            tex_bytes = std::max(tex_bytes, (size_t)texinfo[t].offset + 3 * (size_t)texinfo[t].width * texinfo[t].height);
        }
        tex_rgb.assign(s->texture_rgb, s->texture_rgb + tex_bytes);
        mat_maps.resize(2 * (size_t)s->n_materials);
        uv_trans.resize(9 * (size_t)s->n_materials);
        for (uint32_t m = 0; m < s->n_materials; m++) {
            int32_t a = s->material_texture ? s->material_texture[m] : -1, b = s->material_normal_map ? s->material_normal_map[m] : -1;
            if (a >= (int32_t)s->n_textures || b >= (int32_t)s->n_textures) return pt_fail(c, PT_ERR_ARGUMENT, "texture index out of range");
            mat_maps[2 * m] = a < 0 ? -1 : a; mat_maps[2 * m + 1] = b < 0 ? -1 : b;
            for (int k = 0; k < 9; k++) uv_trans[9 * (size_t)m + k] = s->material_uv_trans ? s->material_uv_trans[9 * (size_t)m + k] : (k % 4 == 0 ? 1.0 : 0.0);
        }
        srgb_lut.resize(256);
        for (int k = 0; k < 256; k++) srgb_lut[k] = std::pow((double)k / 255.0, PT_GAMMA);  // texture.rs:167: c.powf(GAMMA), c = byte / 255
        tri_uv.assign(total_tris * 6, 0.0);
        std::vector<uint8_t> tri_has_uv(total_tris, 0);
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            if (!(s->mesh_texcoords && s->mesh_has_texcoords && s->mesh_has_texcoords[m])) continue;
            uint64_t v0 = s->mesh_vert_off[m];
            for (uint64_t t = s->mesh_tri_off[m]; t < s->mesh_tri_off[m + 1]; t++) {
                for (int corner = 0; corner < 3; corner++) {
                    uint64_t vi = v0 + s->mesh_indices[3 * t + corner];
                    tri_uv[6 * t + 2 * corner] = s->mesh_texcoords[2 * vi]; tri_uv[6 * t + 2 * corner + 1] = s->mesh_texcoords[2 * vi + 1];
                }
                tri_has_uv[t] = 1;
            }
        }
        for (uint32_t t = 0; t < s->n_triangles; t++)
            if (s->tri_texcoords && s->tri_has_texcoords && s->tri_has_texcoords[t]) {
                for (int k = 0; k < 6; k++) tri_uv[6 * (mesh_tris + t) + k] = s->tri_texcoords[6 * (size_t)t + k];
                tri_has_uv[mesh_tris + t] = 1;
            }
        // material.rs:133 / :141: the reference panics when a mapped material meets a primitive without texture coordinates
        for (uint32_t i = 0; i < n; i++) {
            int32_t m = s->material[i];
            if (mat_maps[2 * m] < 0 && mat_maps[2 * m + 1] < 0) continue;
            int t = s->prim_type[i];
            bool ok = t == PT_PRIM_SPHERE || t == PT_PRIM_CUBE || t == PT_PRIM_PLANE;
            if (t == PT_PRIM_TRIANGLE) ok = tri_has_uv[mesh_tris + (size_t)s->prim_data[i]];
            if (t == PT_PRIM_MESH || t == PT_PRIM_KDMESH) ok = s->mesh_tri_off[s->prim_data[i] + 1] == s->mesh_tri_off[s->prim_data[i]] || tri_has_uv[s->mesh_tri_off[s->prim_data[i]]];
            if (!ok) return pt_fail(c, PT_ERR_SCENE, "Texture mapping is not supported for this primitive! (material.rs:133,141)");
        }
        if ((rc = pt_upload(c, c->mat_maps, mat_maps)) || (rc = pt_upload(c, c->uv_trans, uv_trans)) || (rc = pt_upload(c, c->tex, texinfo)) ||
            (rc = pt_upload(c, c->tex_rgb, tex_rgb)) || (rc = pt_upload(c, c->srgb_lut, srgb_lut)) || (rc = pt_upload(c, c->tri_uv, tri_uv)))
            return rc;
    }

    PtSceneView& v = c->view;
    memset(&v, 0, sizeof v);
    v.n_nodes = n; v.n_lights = s->n_lights;
    v.inv = (const double*)c->inv.p; v.fwd = (const double*)c->fwd.p; v.nrm = (const double*)c->nrm.p;
    v.info = (const uint32_t*)c->info.p; v.tri_v = (const double*)c->tri_v.p; v.tri_e = (const double*)c->tri_e.p; v.tri_leaf = (const double*)c->tri_leaf.p; v.tri_n = (const double*)c->tri_n.p;
    v.meshes = (const PtMeshInfo*)c->meshes.p; v.materials = (const double*)c->materials.p; v.lights = (const double*)c->lights.p;
    for (int k = 0; k < 3; k++) v.ambient[k] = s->ambient[k];
    v.bvh = (const PtBvhNode*)c->bvh.p; v.bvh4 = (const PtBvh4Node*)c->bvh4.p; v.bvh_items = (const uint32_t*)c->bvh_items.p;
    v.tlas_root = tlas.child; v.tlas_direct = tlas_direct ? 1u : 0u;
    // the octant-sorted slab test inside mesh instances (pt_trace_packet_mesh; round 4, c32: the 1.25 M-triangle scenes +6.0 % / +3.1 %, macho-cows and the
    // mirror scene +0.6 %; PORTRAYER_MESH_OCT=0 walks every triangle tree with the per-lane form again)
    v.mesh_oct = 1u;
    if (const char* e = getenv("PORTRAYER_MESH_OCT")) v.mesh_oct = atoi(e) > 0 ? 1u : 0u;
    v.kd = (const PtKdNode*)c->kd.p; v.kd_items = (const uint32_t*)c->kd_items.p;
    v.kd_extent = kd_extent;
    v.kd_ref = (const uint32_t*)c->kd_ref.p; v.kd_levels = kd_levels;
    v.node_box = traverse == PT_TRAVERSE_KD && !kdi.empty() ? (const float*)c->node_box.p : nullptr;
    v.kd_box = traverse == PT_TRAVERSE_KD ? (const float*)c->kd_box.p : nullptr;
    if (getenv("PORTRAYER_KD_NO_CULL")) v.kd_box = v.node_box = nullptr;  // experiment: the reference's walk as it is
    if (const char* e = getenv("PORTRAYER_KD_CULL")) { const int m = atoi(e); if (!(m & 1)) v.kd_box = nullptr; if (!(m & 2)) v.node_box = nullptr; }  // experiment: bit 0 tree nodes, bit 1 leaf references
    v.mkd = mkd.empty() ? nullptr : (const PtKdNode*)c->mkd.p; v.mkd_items = (const uint32_t*)c->mkd_items.p;  // (null without KDMesh trees: the k-d walk then keeps no LDS rows for lane stacks)
    v.mkd_box = mkd_box.empty() || getenv("PORTRAYER_KD_NO_CULL") ? nullptr : (const float*)c->mkd_box.p;
    v.mkd_item_box = v.mkd_box ? (const float*)c->mkd_item_box.p : nullptr;
    v.mode = traverse == PT_TRAVERSE_KD ? (s->n_meshes == 0 ? PT_MODE_KD_NOMESH : (any_kdmesh ? PT_MODE_KD : PT_MODE_KD_MESH)) : (s->n_meshes == 0 ? PT_MODE_FLAT_NOMESH : (any_kdmesh ? PT_MODE_FLAT_KDMESH : PT_MODE_FLAT));
    if (traverse == PT_TRAVERSE_HIER) {  // the general walker (meshes and KDMesh trees compiled in), or its mesh-free instantiation
        v.mode = s->n_meshes == 0 ? PT_MODE_HIER_NOMESH : (any_kdmesh ? PT_MODE_HIER : PT_MODE_HIER_MESH);
        v.g_inv = (const double*)c->g_inv.p; v.g_fwd = (const double*)c->g_fwd.p; v.g_nrm = (const double*)c->g_nrm.p;
        v.chain_off = (const uint32_t*)c->chain_off.p; v.chain = (const uint32_t*)c->chain.p; v.dfs_rank = (const uint32_t*)c->dfs_rank.p;
        v.hier_rec = (const uint32_t*)c->hier_rec.p; v.own_inv = (const double*)c->own_inv.p;
    }
    // a level of the four-child walk pushes up to three pending children; it covers two levels of the two-child tree in the plain
    // collapse and at least one when nodes are opened by area
    auto wide = [&](int depth2) { return collapse_mode ? 3 * depth2 : 3 * ((depth2 + 1) / 2); };
    int below = std::max(wide(max_blas_depth), 3 * (max_kdm_depth + 1));  // deepest walk under a scene leaf: a mesh tree or a KDMesh tree
    int cap = traverse == PT_TRAVERSE_KD ? 3 * (kd_depth + 1) + below + 2 : wide(tlas.depth) + below + 4;
    v.stack_cap = std::max(cap, 8);
    if (const char* e = getenv("PORTRAYER_STACK_CAP")) v.stack_cap = std::max(1, atoi(e));  // tests: force PT_ERR_TRAVERSAL
    if (textured) {
        v.mat_maps = (const int32_t*)c->mat_maps.p; v.uv_trans = (const double*)c->uv_trans.p; v.tex = (const PtTexInfo*)c->tex.p;
        v.tex_rgb = (const uint8_t*)c->tex_rgb.p; v.srgb_lut = (const double*)c->srgb_lut.p; v.tri_uv = (const double*)c->tri_uv.p;
        std::vector<PtTexView> tv(1);
        tv[0].tex = v.tex; tv[0].tex_rgb = v.tex_rgb; tv[0].uv_trans = v.uv_trans; tv[0].tri_v = v.tri_v; tv[0].tri_uv = v.tri_uv;
        tv[0].mat_maps = v.mat_maps;
        if ((rc = pt_upload(c, c->texview, tv))) return rc;
        v.texview = (const PtTexView*)c->texview.p;
    }
    lap("upload the rest");
    if (v.stack_cap > 4096) return pt_fail(c, PT_ERR_SCENE, "tree too deep for the traversal stack");
    c->have_scene = true;
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------
// Render
// ------------------------------------------------------------------------------------------------
static uint32_t pt_slots_per_rank(const pt_render_params* p) {
    uint32_t rw = p->slice.x1 - p->slice.x0 + 1, rh = p->slice.y1 - p->slice.y0 + 1;
    uint32_t tiles = ((rw + 7) / 8) * ((rh + 7) / 8);
    uint32_t ranks = p->tile_ranks ? p->tile_ranks : 1;
    return ((tiles + ranks - 1) / ranks) * 64;
}

extern "C" uint64_t pt_compact_bytes(const pt_render_params* p) {
    if (!p || p->slice.x1 < p->slice.x0 || p->slice.y1 < p->slice.y0) return 0;
    return 3ull * pt_slots_per_rank(p);
}

static int pt_check_params(pt_context* c, const pt_camera* cam, const pt_render_params* p) {
    if (!c || !cam || !p) return PT_ERR_ARGUMENT;
    if (!c->have_scene) return pt_fail(c, PT_ERR_NO_SCENE, "no scene uploaded");
    if (p->width == 0 || p->height == 0 || p->samples == 0) return pt_fail(c, PT_ERR_ARGUMENT, "width, height and samples must be positive");
    if (p->slice.x0 >= p->width || p->slice.x1 >= p->width || p->slice.y0 >= p->height || p->slice.y1 >= p->height)
        return pt_fail(c, PT_ERR_SLICE, "slice corner outside the image (render.rs:79-90)");
    if (p->tile_ranks == 0 || p->tile_rank >= p->tile_ranks) return pt_fail(c, PT_ERR_ARGUMENT, "tile_rank must be < tile_ranks");
    if (p->sample_mode != PT_SAMPLE_CENTRE && p->sample_mode != PT_SAMPLE_RNG) return pt_fail(c, PT_ERR_ARGUMENT, "bad sample_mode");
    // work items of one launch are indexed in 32 bits (pixel slots x 8-sample chunks): refuse what would wrap
    if (!(p->slice.x1 < p->slice.x0 || p->slice.y1 < p->slice.y0)) {
        uint64_t work = (uint64_t)pt_slots_per_rank(p) * ((p->samples + PT_SAMPLE_CHUNK - 1) / PT_SAMPLE_CHUNK);
        if (work >= 0xFFFFFFFFull - 65536)
            return pt_fail(c, PT_ERR_ARGUMENT, "slice x samples too large for one launch (pixel slots x ceil(samples / 8) must stay below 2^32): render it in slices");
    }
    return PT_OK;
}

// a.run_variant (PT_RUN_*, pt_render_kernel.h) selects the kernel.
static hipError_t pt_dispatch(const PtRenderArgs& a, bool stats, int n_cu, hipStream_t stream, uint32_t* grid, bool launch) {
    const bool tex = a.scene.mat_maps != nullptr;
    switch (a.scene.mode) {
    case PT_MODE_KD: return pt_launch_mode_2(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_FLAT_NOMESH: return pt_launch_mode_3(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_FLAT_KDMESH: return pt_launch_mode_4(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_HIER: return pt_launch_mode_5(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_HIER_NOMESH: return pt_launch_mode_6(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_KD_NOMESH: return pt_launch_mode_7(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_HIER_MESH: return pt_launch_mode_8(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    case PT_MODE_KD_MESH: return pt_launch_mode_9(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    default: return pt_launch_mode_1(a, a.run_variant, stats, tex, n_cu, stream, grid, launch);
    }
}

// The part of the kernel arguments that follows from the render parameters alone: the slice, the rank's tiles, how a wavefront's
// 64 lanes are laid over pixels x chunks x samples, and the number of work items (pt_shade.h: pt_item_lane).
static void pt_fill_work(const pt_render_params* p, PtRenderArgs* a) {
    a->background_rows = p->background_rows;
    a->width = p->width; a->height = p->height;
    a->x0 = p->slice.x0; a->y0 = p->slice.y0; a->x1 = p->slice.x1; a->y1 = p->slice.y1;
    a->samples = p->samples; a->seed = p->seed; a->jitter_mode = p->sample_mode;
    a->tile_rank = p->tile_rank; a->tile_ranks = p->tile_ranks;
    bool empty = p->slice.x1 < p->slice.x0 || p->slice.y1 < p->slice.y0;  // render.rs:60-65: an inverted slice renders nothing
    a->n_slots = empty ? 0 : pt_slots_per_rank(p);
    a->n_chunks = (p->samples + PT_SAMPLE_CHUNK - 1) / PT_SAMPLE_CHUNK;
    // K samples of a pixel run side by side in a wavefront: a whole chunk from SAMPLES = 8 on, else the next power of two
    uint32_t k = PT_SAMPLE_CHUNK;
    if (p->samples < PT_SAMPLE_CHUNK) { k = 1; while (k < p->samples) k *= 2; }
    a->lane_samples = k;
    // C chunks of a pixel side by side: the largest of 8, 4, 2, 1 that leaves at most 1/16 of the chunk slots empty
    // (SAMPLES = 64: 8 chunks -> C = 8, one pixel per wavefront; SAMPLES = 16: C = 2; SAMPLES = 100, 13 chunks: C = 1)
    uint32_t cc = 1;
    if (k == PT_SAMPLE_CHUNK)
        for (uint32_t cand = 8; cand > 1; cand /= 2) {
            uint32_t slots = (a->n_chunks + cand - 1) / cand * cand;
            if ((slots - a->n_chunks) * 16 <= a->n_chunks) { cc = cand; break; }
        }
    if (const char* e = getenv("PORTRAYER_LANE_CHUNKS")) { uint32_t v = (uint32_t)atoi(e); if (k == PT_SAMPLE_CHUNK && (v == 1 || v == 2 || v == 4 || v == 8)) cc = v; }
    a->lane_chunks = cc;
    a->n_items = (a->n_slots / 64) * ((a->n_chunks + cc - 1) / cc) * (k * cc);
    for (a->k_log2 = 0; (1u << a->k_log2) < k; a->k_log2++) { }
    for (a->c_log2 = 0; (1u << a->c_log2) < cc; a->c_log2++) { }
    a->div_groups = pt_fastdiv_make((a->n_chunks + cc - 1) / cc);
    a->div_tiles_x = pt_fastdiv_make(empty ? 1u : (p->slice.x1 - p->slice.x0 + 1u + 7u) / 8u);
}

static int pt_fill_args(pt_context* c, const pt_camera* cam, const pt_render_params* p, PtRenderArgs* a) {
    memset(a, 0, sizeof *a);
    a->scene = c->view;
    for (int k = 0; k < 3; k++) a->cam.eye[k] = cam->eye[k];
    for (int k = 0; k < 12; k++) a->cam.view_to_world[k] = cam->view_to_world[k];
    a->cam.fov_factor = cam->fov_factor; a->cam.aspect = cam->aspect_ratio; a->cam.width = cam->width; a->cam.height = cam->height;
    pt_fill_work(p, a);
    return PT_OK;
}

// Host-side replay of the kernel's work decomposition (no GPU involved): every work item of a launch with these parameters,
// every lane of it, through the same pt_item_lane the kernel uses. Per pixel of the image: how many (lane, item) pairs carry a
// sample of it, the sum of their sample indices, and the sum of the chunk lengths reported by the lanes that add a chunk up.
// Tests check that every sample of every pixel of the rank's tiles is covered exactly once.
extern "C" int pt_test_work_items(const pt_render_params* p, uint32_t* sample_count, uint64_t* sample_index_sum, uint32_t* chunk_length_sum,
                                  uint64_t* n_items, uint32_t* lane_pixels_chunks_samples) {
    if (!p || !sample_count || !sample_index_sum || !chunk_length_sum) return PT_ERR_ARGUMENT;
    if (p->width == 0 || p->height == 0 || p->samples == 0 || p->tile_ranks == 0 || p->tile_rank >= p->tile_ranks) return PT_ERR_ARGUMENT;
    if (p->slice.x0 >= p->width || p->slice.x1 >= p->width || p->slice.y0 >= p->height || p->slice.y1 >= p->height) return PT_ERR_SLICE;
    PtRenderArgs a;
    memset(&a, 0, sizeof a);
    pt_fill_work(p, &a);
    if (n_items) *n_items = a.n_items;
    if (lane_pixels_chunks_samples) {
        lane_pixels_chunks_samples[0] = 64u / (a.lane_samples * a.lane_chunks); lane_pixels_chunks_samples[1] = a.lane_chunks; lane_pixels_chunks_samples[2] = a.lane_samples;
    }
    for (uint32_t w = 0; w < a.n_items; w++)
        for (uint32_t lane = 0; lane < 64; lane++) {
            PtItemLane it, it_k;
            uint32_t x, y, x_k, y_k;
            const bool mine = pt_item_lane(a, w, lane, &it, &x, &y);
            // what the render kernels actually run (shifts + multiply-high divisions) must agree in every field
            const bool mine_k = pt_item_lane_fast(a, w, lane, &it_k, &x_k, &y_k);
            if (mine != mine_k || it.slot != it_k.slot || it.chunk != it_k.chunk || it.sample != it_k.sample || it.first != it_k.first || it.count != it_k.count ||
                (mine && (x != x_k || y != y_k)))
                return PT_ERR_TRAVERSAL;
            if (!mine) continue;
            if (x >= p->width || y >= p->height) return PT_ERR_TRAVERSAL;  // would write outside the image
            const size_t px = (size_t)y * p->width + x;
            sample_count[px]++;
            sample_index_sum[px] += it.sample;
            if (it.first) chunk_length_sum[px] += it.count;
        }
    return PT_OK;
}

static int pt_render_common(pt_context* c, PtRenderArgs& a, bool stats, hipStream_t stream, int slot_index = -1) {
    // LDS per block = traversal stack (as much of it as leaves room for three blocks per CU) + the shaded hit's frame (+ a parked one);
    // deeper stack entries live in HBM (PtStackSpill).
    if (getenv("PORTRAYER_NO_TEX")) a.scene.mat_maps = nullptr;  // experiment: the untextured kernel on a textured scene (wrong picture, timing only)
    const bool tex = a.scene.mat_maps != nullptr;
    // parked recursion frames kept in LDS (pt_shade.h): only scenes that park frames at all have any
    a.park_slots = 1;
    if (const char* e = getenv("PORTRAYER_PARK")) a.park_slots = atoi(e) > 0 ? 1 : 0;  // 0: every parked frame in HBM (tests, measurements)
    if (!c->spawns) a.park_slots = 0;
    // Scenes whose hits spawn rays need the interpreter kernel (3 waves per SIMD); the others run the straight-line kernel at 3 or
    // 4 waves per SIMD. PORTRAYER_INTERP=1 (builds with -DPT_KEEP_INTERP): the interpreter on those too, for A/B runs.
    const bool kd_sem = a.scene.mode == PT_MODE_KD || a.scene.mode == PT_MODE_KD_NOMESH || a.scene.mode == PT_MODE_KD_MESH;
    a.four_waves = (!c->spawns && (c->four_waves || (c->four_waves_untextured && !tex) || c->four_waves_hier)) ? (c->five_waves ? PT_LINE_TOP_WAVES : ((c->five_waves_mesh && !tex) ? PT_MESH_TOP_WAVES : 4)) : 0;
    if (const char* e = getenv("PORTRAYER_WAVES")) {
        const int wv = atoi(e);
        a.four_waves = (!c->spawns && wv >= 4) ? ((wv >= 5 && (a.scene.mode == PT_MODE_FLAT_NOMESH || a.scene.mode == PT_MODE_HIER_NOMESH)) ? PT_LINE_TOP_WAVES : ((wv >= 5 && (a.scene.mode == PT_MODE_FLAT || a.scene.mode == PT_MODE_HIER_MESH)) ? PT_MESH_TOP_WAVES : 4)) : 0;
    }
    if (kd_sem) {
        // The k-d semantics: mesh-free scenes with many nodes take the 4-wave straight-line kernel too (big-scene 35.7 -> 30.7 ms: the per-lane
        // k-d walk waits on its own loads, a fourth wavefront per SIMD hides more of that than the 4 spilled registers cost); with mesh
        // instances the walk needs the registers (167 at 3 waves). PORTRAYER_KD_WAVES=3|4 overrides.
        // Round 4 (one walk per wavefront, pt_trace_packet_kd): mesh-free scenes with many nodes at 5 waves (big-scene 29.2 -> 28.1 ms, c10), scenes with plain
        // Mesh instances (PT_MODE_KD_MESH: no KDMesh walker compiled in) at 4; with KDMesh trees the kernel needs its 168 registers.
        a.four_waves = c->spawns ? 0 : ((a.scene.mode == PT_MODE_KD_NOMESH && a.scene.n_nodes >= 256) ? 5 : (a.scene.mode == PT_MODE_KD_MESH ? 4 : 0));
        if (const char* e = getenv("PORTRAYER_KD_WAVES")) {
            const int wv = atoi(e);
            a.four_waves = (c->spawns || wv < 4 || a.scene.mode == PT_MODE_KD) ? 0 : ((wv >= 5 && a.scene.mode == PT_MODE_KD_NOMESH) ? 5 : 4);
        }
    }
    // Fork / join of refracted subtrees (pt_shade.h) is built, parity-green and OFF by default: it fills the idle lanes and still loses
    // (transmission-refraction 11.4 -> 8.6 Gray/s, profiles/r03/notes.md section 4): the subtrees other lanes walk are other rays, and the
    // one walk per wavefront pays for the union of their paths. PORTRAYER_FORK=1 switches it on for scenes that qualify.
    bool fork = false;
    if (const char* e = getenv("PORTRAYER_FORK")) fork = c->forkable && a.park_slots && atoi(e) > 0;
    if (c->launch_seq == 0) c->launch_seq = (uint32_t)std::chrono::steady_clock::now().time_since_epoch().count() * 2654435761u;  // a different starting point in every context
    a.launch_nonce = ++c->launch_seq;
    bool chain = c->one_ray;
    if (const char* e = getenv("PORTRAYER_CHAIN")) chain = chain && atoi(e) > 0;  // 0: the interpreter kernel also for scenes whose recursion is a chain (A/B runs, tests)
    if (c->spawns) a.run_variant = (chain && a.park_slots) ? PT_RUN_CHAIN : (a.park_slots ? (fork ? PT_RUN_INTERP_FORK : PT_RUN_INTERP_PARK) : PT_RUN_INTERP);
    else if (pt_interpreter_forced()) a.run_variant = a.four_waves ? PT_RUN_INTERP4 : PT_RUN_INTERP;
    else a.run_variant = a.four_waves >= 5 ? PT_RUN_LINE5 : (a.four_waves ? PT_RUN_LINE4 : PT_RUN_LINE3);  // (LINE5: the densest instantiation the mode has)
    if (a.run_variant == PT_RUN_CHAIN) {
        a.park_slots = 0;  // no frame in LDS: the parked colours go straight to the lane's HBM lines
        // 4 waves per SIMD in the flat_scene semantics (mirror scene 25.5 -> 26.8 Gray/s, c34: ten bounces per sample leave a lot of latency to hide),
        // 3 in the hierarchical ones (182 spilled registers at 128: 19.3 -> 14.8) and the k-d ones. PORTRAYER_CHAIN_WAVES=3|4 overrides.
        // (Only where that was measured or the compiler's figures are like the measured case's: untextured, no KDMesh trees - those
        // instantiations spill 57 / 142 registers at 128.)
        const bool flat_sem = (a.scene.mode == PT_MODE_FLAT || a.scene.mode == PT_MODE_FLAT_NOMESH || a.scene.mode == PT_MODE_HIER_MESH) && !tex;  // (HIER_MESH: 22.2 -> 24.2, c41)
        a.four_waves = flat_sem ? 4 : 0;
        if (const char* e = getenv("PORTRAYER_CHAIN_WAVES")) a.four_waves = (atoi(e) == 4 && a.scene.mode != PT_MODE_KD) ? 4 : 0;
        else if (a.scene.mode == PT_MODE_KD_MESH && !tex) a.four_waves = 4;
    }
    size_t block_budget = a.four_waves >= 6 ? 26 * 1024 : (a.four_waves == 5 ? 31 * 1024 : (a.four_waves ? 39 * 1024 : 52 * 1024));  // 3 x 52 KB, 4 x 39 KB, 5 x 31 KB or 6 x 26 KB of the CU's 160 KB
    if (const char* e = getenv("PORTRAYER_LDS_BUDGET_KB")) block_budget = (size_t)std::max(16, std::min(160, atoi(e))) * 1024;  // experiment: 80 = two blocks per CU
    const size_t frame_bytes = (size_t)(PT_LDS_FRAME_F64 + a.park_slots * PT_PARK_F64) * PT_BLOCK * 8;
    int lds_cap = block_budget > frame_bytes ? (int)((block_budget - frame_bytes) / (PT_BLOCK * 4)) : 0;
    lds_cap = std::max(lds_cap, 2);
    if (const char* e = getenv("PORTRAYER_LDS_STACK")) lds_cap = std::max(1, atoi(e));  // experiments / tests of the overflow path
    a.stack_lds_cap = std::min(lds_cap, a.scene.stack_cap);
    // the k-d walk keeps no saved bounds in HBM any more (round 5): its wavefront rows must hold the stack and, where the stack's slack is too small for it, the
    // two rows of the path table (pt_kd_layout, PtKdSav) - also under PORTRAYER_LDS_STACK=1, which then only shrinks the LANES' stacks
    if (kd_sem) a.stack_lds_cap = std::max(a.stack_lds_cap, (a.scene.stack_cap + 63) / 64 + 2);
    a.grid_share = 1;  // (a share of the resident blocks per launch was tried for ranks that share a GPU, round 5 c23: their kernels do not run side by side - 35.5 -> 14.3 Gray/s)
    uint32_t grid = 0;
    PT_HIP(c, pt_dispatch(a, stats, c->n_cu, stream, &grid, false));
    a.n_lanes = grid * PT_BLOCK;
    a.work_div = std::max<uint32_t>(grid * (PT_BLOCK / 64) * 8u, 1u);  // batch = remaining items / (8 x resident wavefronts)
    // How the work items are handed out (pt_render_kernel): batches of consecutive items from one counter, or - scenes whose
    // hits can spawn rays, where an item in the glass costs hundreds of times its neighbour - one item at a time from
    // interleaved queues. transmission-refraction: wavefronts resident 47 % of the launch and 5.2 Gray/s with batches, 11+ without.
    a.batch_max = PT_WORK_BATCH_MAX;
    if (const char* e = getenv("PORTRAYER_BATCH_MAX")) a.batch_max = (uint32_t)std::max(1, atoi(e));
    // The queues also win wherever a wavefront gets few items (a small frame, one GPU's share of a frame: big-scene's 1/8 share +8 %,
    // macho-cows +13 %) and on the mesh-heavy scenes (+3-7 %); very long launches (> 2048 items per resident wavefront: 3840x2160x256)
    // are 1 % better off with batches.
    const uint64_t resident_waves = (uint64_t)grid * (PT_BLOCK / 64);
    const bool long_launch = (uint64_t)a.n_items > 2048ull * std::max<uint64_t>(resident_waves, 1);
    // (round 3's per-lane k-d walk was 1 % better off with batches; the wave-uniform one is not: big-scene +1.1 %, macho-cows +3.2 % with the queues, round 4 c68)
    a.fine_queues = (c->spawns || !long_launch) ? 16 : 0;  // 8 .. 32 queues measured alike, 64 and 4 about 1 % behind
    if (const char* e = getenv("PORTRAYER_FINE_QUEUES")) a.fine_queues = (uint32_t)std::max(0, std::min(PT_FINE_QUEUES, atoi(e)));
    a.item_stride = 1;
    if (const char* e = getenv("PORTRAYER_ITEM_STRIDE")) {  // experiment (batches only): position q -> item (q * stride) mod n; "golden" = 0.618 n
        uint64_t st = strcmp(e, "golden") == 0 ? (uint64_t)((double)a.n_items * 0.6180339887498949) : (uint64_t)atoll(e);
        auto gcd = [](uint64_t x, uint64_t y) { while (y) { uint64_t t = x % y; x = y; y = t; } return x; };
        if (a.n_items > 2 && st != 1) { st = st % a.n_items; if (st < 1) st = 1; while (gcd(st, a.n_items) != 1) st++; a.item_stride = (uint32_t)st; }
    }
    int rc;
    if (slot_index < 0) {  // the host-buffer path (pt_render): one frame at a time
        if (c->slot[c->slot_oldest].pending) return pt_fail(c, PT_ERR_ARGUMENT, "a render is in flight: pt_render_finish first");
        slot_index = c->slot_next;
    }
    pt_context::Slot& sl = c->slot[slot_index];  // the launch's work buffers are its slot's: another frame may be in flight on the other slot's
    const size_t spill_bytes = c->needs_spill ? (size_t)a.n_lanes * PT_SPILL_DEPTHS * PT_SPILL_STRIDE * sizeof(double) : 16;
    if ((rc = pt_reserve(c, sl.spill, spill_bytes))) return rc;
    // + wave_rows: the wavefront's own stack takes LDS rows from the lanes' stacks (pt_wave_rows, pt_render_simple.h): eight, or what a deep tree needs
    const int wave_rows = std::min(std::max(8, (a.scene.stack_cap + 63) / 64), std::max(a.stack_lds_cap, 8));  // (pt_wave_rows: at most this many)
    const int stack_spill_entries = std::max(a.scene.stack_cap - a.stack_lds_cap + wave_rows, 0);
    const size_t stack_column = (size_t)std::max(stack_spill_entries, a.scene.stack_cap);  // everything a lane's own stack can reach (pt_trace_wave gives the lanes fewer LDS rows in the k-d semantics)
    if ((rc = pt_reserve(c, sl.stack_spill, (size_t)a.n_lanes * stack_column * 4))) return rc;  // (the k-d walk's saved bounds needed columns behind this until round 5)
    if ((rc = pt_reserve(c, sl.misc, 256 + sizeof(PtCounters) + PT_FINE_QUEUES * PT_QUEUE_STRIDE * 4))) return rc;
    if ((rc = pt_reserve(c, sl.accum, (size_t)a.n_slots * a.n_chunks * 3 * sizeof(double)))) return rc;
    a.accum = (double*)sl.accum.p;
    a.spill = (double*)sl.spill.p;
    a.stack_spill = (uint32_t*)sl.stack_spill.p;
    a.work_counter = (unsigned int*)sl.misc.p;
    a.overflow_flag = (unsigned int*)sl.misc.p + 1;
    a.counters = (PtCounters*)((char*)sl.misc.p + 256);
    a.work_queues = (unsigned int*)((char*)sl.misc.p + 256 + sizeof(PtCounters));
    c->last_mode = (uint32_t)a.scene.mode;
    c->last_variant = (a.four_waves ? (uint32_t)a.four_waves : 3u) | ((a.run_variant == PT_RUN_LINE3 || a.run_variant == PT_RUN_LINE4 || a.run_variant == PT_RUN_LINE5 || a.run_variant == PT_RUN_CHAIN) ? 0u : PT_KERNEL_INTERPRETER) | (a.run_variant == PT_RUN_CHAIN ? PT_KERNEL_CHAIN : 0u) |
                      ((a.run_variant == PT_RUN_INTERP_PARK || a.run_variant == PT_RUN_INTERP_FORK) ? PT_KERNEL_PARK : 0u) | (a.run_variant == PT_RUN_INTERP_FORK ? PT_KERNEL_FORK : 0u) | (stats ? PT_KERNEL_COUNTING : 0u) | (tex ? PT_KERNEL_TEXTURED : 0u);
    PT_HIP(c, hipMemsetAsync(sl.misc.p, 0, 256 + sizeof(PtCounters) + PT_FINE_QUEUES * PT_QUEUE_STRIDE * 4, stream));
    sl.mode = c->last_mode; sl.variant = c->last_variant; sl.counted = stats;
    PT_HIP(c, hipEventRecord(sl.ev0, stream));
    if (a.n_items) {
        PT_HIP(c, pt_dispatch(a, stats, c->n_cu, stream, &grid, true));
        hipLaunchKernelGGL(pt_finish_kernel, dim3((a.n_slots + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, stream, a);
        PT_HIP(c, hipGetLastError());
    }
    PT_HIP(c, hipEventRecord(sl.ev1, stream));
    // the overflow flag (always) and the counters (counting build) follow the kernels into the slot's pinned page
    PT_HIP(c, hipMemcpyAsync(sl.host, sl.misc.p, stats ? PT_SLOT_BYTES : 8, hipMemcpyDeviceToHost, stream));
    PT_HIP(c, hipEventRecord(sl.copy_done, stream));
    return PT_OK;
}

// Reads the overflow flag (always) and, for the counting build, the counters of a finished launch out of its slot's pinned page (the
// caller has waited for the stream: the copy queued behind the kernels is done). A launch in which any lane ran out of traversal
// stack fails with PT_ERR_TRAVERSAL whether or not the caller asked for statistics.
static int pt_collect_stats(pt_context* c, pt_stats* st, int slot_index) {
    pt_context::Slot& sl = c->slot[slot_index];
    if (st) memset(st, 0, sizeof *st);
    unsigned int head[2];  // work counter, overflow flag
    memcpy(head, sl.host, sizeof head);
    if (st) {
        float ms = 0.f;
        PT_HIP(c, hipEventElapsedTime(&ms, sl.ev0, sl.ev1));
        st->kernel_ms = ms;
        if (sl.counted) {
            PtCounters h;
            memcpy(&h, sl.host + 256, sizeof h);
            st->primary = h.primary; st->shadow = h.shadow; st->reflect = h.reflect; st->refract = h.refract;
            st->depth11_skipped = h.depth11_skipped; st->hits = h.hits; st->n_inner = h.n_inner; st->n_leaf = h.n_leaf;
            st->n_analytic = h.n_analytic; st->n_tri = h.n_tri; st->n_bbox = h.n_bbox; st->kd_plane_miss = h.kd_plane_miss;
            st->stack_overflow = h.stack_overflow;
            for (int k = 0; k < 8; k++) st->diag[k] = h.diag[k];
        }
        if (head[1] && !st->stack_overflow) st->stack_overflow = 1;
        st->kernel_mode = sl.mode; st->kernel_variant = sl.variant;
    }
    if (head[1] & 2u) return pt_fail(c, PT_ERR_TRAVERSAL, "fork / join of refracted subtrees stalled (a lane waited for a colour nobody was computing): results invalid");
    if (head[1] & 4u) return pt_fail(c, PT_ERR_TRAVERSAL, "a tree walk did not end (watchdog): results invalid");
    if (head[1]) return pt_fail(c, PT_ERR_TRAVERSAL, "traversal stack overflow");
    return PT_OK;
}

extern "C" int pt_render(pt_context* c, const pt_camera* cam, const double* background, const pt_render_params* p,
                         uint8_t* rgb, double* linear, pt_stats* stats) {
    int rc = pt_check_params(c, cam, p);
    if (rc) return rc;
    if (!background || !rgb) return pt_fail(c, PT_ERR_ARGUMENT, "background and rgb must not be null");
    PT_HIP(c, hipSetDevice(c->device));
    auto t0 = std::chrono::steady_clock::now();
    PtRenderArgs a;
    pt_fill_args(c, cam, p, &a);
    size_t px = (size_t)p->width * p->height;
    size_t bg_bytes = (p->background_rows ? (size_t)p->height : px) * 3 * sizeof(double);
    if ((rc = pt_reserve(c, c->bg, bg_bytes)) || (rc = pt_reserve(c, c->rgb, px * 3))) return rc;
    if (linear && (rc = pt_reserve(c, c->linear, px * 3 * sizeof(double)))) return rc;
    PT_HIP(c, hipMemcpy(c->bg.p, background, bg_bytes, hipMemcpyHostToDevice));
    // pixels outside the slice / of other ranks keep the caller's bytes (render.rs:135-138)
    PT_HIP(c, hipMemcpy(c->rgb.p, rgb, px * 3, hipMemcpyHostToDevice));
    if (linear) PT_HIP(c, hipMemcpy(c->linear.p, linear, px * 3 * sizeof(double), hipMemcpyHostToDevice));
    a.background = (const double*)c->bg.p;
    a.compact = 0;
    a.rgb = (uint8_t*)c->rgb.p;
    a.linear = linear ? (double*)c->linear.p : nullptr;
    bool counted = p->collect_stats != 0;
    const int slot_index = c->slot_next;
    if ((rc = pt_render_common(c, a, counted, nullptr))) return rc;
    PT_HIP(c, hipDeviceSynchronize());
    PT_HIP(c, hipMemcpy(rgb, c->rgb.p, px * 3, hipMemcpyDeviceToHost));
    if (linear) PT_HIP(c, hipMemcpy(linear, c->linear.p, px * 3 * sizeof(double), hipMemcpyDeviceToHost));
    rc = pt_collect_stats(c, stats, slot_index);
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int pt_render_device(pt_context* c, const pt_camera* cam, const double* d_background, const pt_render_params* p,
                                int compact, void* d_rgb, void* hip_stream) {
    int rc = pt_check_params(c, cam, p);
    if (rc) return rc;
    if (!d_background || !d_rgb) return pt_fail(c, PT_ERR_ARGUMENT, "d_background and d_rgb must not be null");
    PT_HIP(c, hipSetDevice(c->device));
    PtRenderArgs a;
    pt_fill_args(c, cam, p, &a);
    a.background = d_background;
    a.compact = compact ? 1 : 0;
    a.rgb = (uint8_t*)d_rgb;
    a.linear = nullptr;
    const int slot_index = c->slot_next;
    if (c->slot[slot_index].pending) return pt_fail(c, PT_ERR_ARGUMENT, "too many renders in flight on this context: pt_render_finish first");
    c->slot[slot_index].t_start = std::chrono::steady_clock::now();
    if ((rc = pt_render_common(c, a, p->collect_stats != 0, (hipStream_t)hip_stream, slot_index))) return rc;
    c->slot[slot_index].pending = true;
    c->slot_next = (slot_index + 1) % pt_context::PT_SLOTS;
    return PT_OK;
}

extern "C" int pt_render_finish(pt_context* c, pt_stats* stats) {
    if (!c) return PT_ERR_ARGUMENT;
    const int slot_index = c->slot_oldest;
    pt_context::Slot& sl = c->slot[slot_index];
    if (!sl.pending) return pt_fail(c, PT_ERR_ARGUMENT, "no render in flight");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipEventSynchronize(sl.copy_done));  // behind the kernels, ev1 and the copy of the flag / the counters
    sl.pending = false;
    c->slot_oldest = (slot_index + 1) % pt_context::PT_SLOTS;
    int rc = pt_collect_stats(c, stats, slot_index);
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sl.t_start).count();
    return rc;
}

extern "C" int pt_untile_device(pt_context* c, const pt_render_params* p, const void* d_gathered, void* d_rgb, void* hip_stream) {
    if (!c || !p || !d_gathered || !d_rgb) return PT_ERR_ARGUMENT;
    if (p->width == 0 || p->height == 0 || p->tile_ranks == 0) return pt_fail(c, PT_ERR_ARGUMENT, "bad params");
    if (p->slice.x0 >= p->width || p->slice.x1 >= p->width || p->slice.y0 >= p->height || p->slice.y1 >= p->height)
        return pt_fail(c, PT_ERR_SLICE, "slice corner outside the image (render.rs:79-90)");
    if (p->slice.x1 < p->slice.x0 || p->slice.y1 < p->slice.y0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PtRenderArgs a;
    memset(&a, 0, sizeof a);
    a.width = p->width; a.height = p->height;
    a.x0 = p->slice.x0; a.y0 = p->slice.y0; a.x1 = p->slice.x1; a.y1 = p->slice.y1;
    a.tile_ranks = p->tile_ranks;
    uint32_t per = pt_slots_per_rank(p);
    uint32_t total = per * p->tile_ranks;
    hipLaunchKernelGGL(pt_untile_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, a, per, (const uint8_t*)d_gathered, (uint8_t*)d_rgb);
    PT_HIP(c, hipGetLastError());
    return PT_OK;
}

static bool pt_params_to_args(const pt_render_params* p, uint32_t rank, PtRenderArgs* a) {
    if (!p || p->width == 0 || p->height == 0 || p->tile_ranks == 0 || rank >= p->tile_ranks) return false;
    if (p->slice.x0 >= p->width || p->slice.x1 >= p->width || p->slice.y0 >= p->height || p->slice.y1 >= p->height) return false;
    memset(a, 0, sizeof *a);
    a->width = p->width; a->height = p->height;
    a->x0 = p->slice.x0; a->y0 = p->slice.y0; a->x1 = p->slice.x1; a->y1 = p->slice.y1;
    a->tile_rank = rank; a->tile_ranks = p->tile_ranks;
    return true;
}

extern "C" int pt_tile_slot_pixel(const pt_render_params* p, uint32_t rank, uint32_t slot, uint32_t* x, uint32_t* y) {
    PtRenderArgs a;
    if (!x || !y || !pt_params_to_args(p, rank, &a)) return PT_ERR_ARGUMENT;
    if (p->slice.x1 < p->slice.x0 || p->slice.y1 < p->slice.y0 || slot >= pt_slots_per_rank(p)) return 0;
    return pt_slot_to_pixel(a, slot, x, y) ? 1 : 0;
}

extern "C" int pt_untile_host(const pt_render_params* p, const uint8_t* gathered, uint8_t* rgb) {
    PtRenderArgs a;
    if (!gathered || !rgb || !pt_params_to_args(p, 0, &a)) return PT_ERR_ARGUMENT;
    if (p->slice.x1 < p->slice.x0 || p->slice.y1 < p->slice.y0) return PT_OK;
    uint32_t per = pt_slots_per_rank(p);
    for (uint32_t r = 0; r < p->tile_ranks; r++) {
        a.tile_rank = r;
        for (uint32_t w = 0; w < per; w++) {
            uint32_t x, y;
            if (!pt_slot_to_pixel(a, w, &x, &y)) continue;
            const uint8_t* s = gathered + 3 * ((size_t)r * per + w);
            uint8_t* d = rgb + 3 * ((size_t)y * p->width + x);
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        }
    }
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------
// Harness helpers
// ------------------------------------------------------------------------------------------------
extern "C" int pt_device_alloc(pt_context* c, uint64_t bytes, void** out) {
    if (!c || !out) return PT_ERR_ARGUMENT;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMalloc(out, bytes ? bytes : 16));
    return PT_OK;
}
extern "C" int pt_device_free(pt_context* c, void* p) {
    if (!c) return PT_ERR_ARGUMENT;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipFree(p));
    return PT_OK;
}
extern "C" int pt_copy_to_device(pt_context* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return PT_ERR_ARGUMENT;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PT_OK;
}
extern "C" int pt_copy_from_device(pt_context* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return PT_ERR_ARGUMENT;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_synchronize(pt_context* c) {
    if (!c) return PT_ERR_ARGUMENT;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipDeviceSynchronize());
    return PT_OK;
}

extern "C" int pt_measure_copy_bandwidth(pt_context* c, uint64_t bytes, int iters, double* gbps) {
    if (!c || !gbps || bytes < 16) return PT_ERR_ARGUMENT;
    PT_HIP(c, hipSetDevice(c->device));
    void *src = nullptr, *dst = nullptr;
    PT_HIP(c, hipMalloc(&src, bytes));
    PT_HIP(c, hipMalloc(&dst, bytes));
    PT_HIP(c, hipMemset(src, 1, bytes));
    size_t n = bytes / 16;
    double best = 0.0;
    // The box's copy roofline = the best of a few ways to copy: a grid-stride kernel with four loads in flight at four grid
    // sizes, one element per thread, and the runtime's own device-to-device copy.
    const int per_cu[4] = {8, 16, 32, 64};
    for (int i = 0; i < 6 * iters + 1; i++) {
        const int form = i % 6;
        PT_HIP(c, hipEventRecord(c->slot[0].ev0, nullptr));
        if (form < 4) hipLaunchKernelGGL(pt_copy_kernel, dim3(c->n_cu * per_cu[form]), dim3(256), 0, nullptr, (const double2*)src, (double2*)dst, n);
        else if (form == 4) hipLaunchKernelGGL(pt_copy1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const double2*)src, (double2*)dst, n);
        else PT_HIP(c, hipMemcpyAsync(dst, src, n * 16, hipMemcpyDeviceToDevice, nullptr));
        PT_HIP(c, hipEventRecord(c->slot[0].ev1, nullptr));
        PT_HIP(c, hipEventSynchronize(c->slot[0].ev1));
        float ms = 0.f;
        PT_HIP(c, hipEventElapsedTime(&ms, c->slot[0].ev0, c->slot[0].ev1));
        if (i > 0 && ms > 0.f) {
            const double rate = 2.0 * (double)(n * 16) / (ms * 1e-3) / 1e9;
            if (getenv("PORTRAYER_VERBOSE")) fprintf(stderr, "[pt_measure_copy_bandwidth] form %d: %.0f GB/s\n", form, rate);
            best = std::max(best, rate);
        }
    }
    hipFree(src); hipFree(dst);
    *gbps = best;
    return PT_OK;
}

extern "C" int pt_test_cast_rays(pt_context* c, uint64_t n, const double* origins, const double* directions, int any_hit,
                                 double* out_t, int32_t* out_node, int32_t* out_sub) {
    if (!c || !origins || !directions || !out_t || !out_node || !out_sub) return PT_ERR_ARGUMENT;
    if (!c->have_scene) return pt_fail(c, PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    struct Bufs {  // freed on every return path
        void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Bufs() { for (void* q : p) if (q) hipFree(q); }
    } bufs;
    const size_t sizes[5] = {n * 24, n * 24, n * 8, n * 4, n * 4};
    for (int k = 0; k < 5; k++) PT_HIP(c, hipMalloc(&bufs.p[k], sizes[k]));
    double *d_o = (double*)bufs.p[0], *d_d = (double*)bufs.p[1], *d_t = (double*)bufs.p[2];
    int32_t *d_n = (int32_t*)bufs.p[3], *d_s = (int32_t*)bufs.p[4];
    int rc = pt_reserve(c, c->misc, 256 + sizeof(PtCounters));
    if (rc) return rc;
    PT_HIP(c, hipMemset(c->misc.p, 0, 8));
    unsigned int* overflow = (unsigned int*)c->misc.p + 1;
    PT_HIP(c, hipMemcpy(d_o, origins, n * 24, hipMemcpyHostToDevice));
    PT_HIP(c, hipMemcpy(d_d, directions, n * 24, hipMemcpyHostToDevice));
    size_t lds = (size_t)c->view.stack_cap * PT_BLOCK * 4;
    if (lds > 160 * 1024) return pt_fail(c, PT_ERR_SCENE, "pt_test_cast_rays keeps the whole traversal stack in LDS: tree too deep for it");
    dim3 grid((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK));
    auto cast = [&](auto mode) -> hipError_t {
        constexpr int M = decltype(mode)::value;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pt_cast_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(pt_cast_kernel<M>, grid, dim3(PT_BLOCK), lds, nullptr, c->view, n, d_o, d_d, any_hit, d_t, d_n, d_s, overflow);
        return hipSuccess;
    };
    switch (c->view.mode) {
    case PT_MODE_FLAT_NOMESH: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_FLAT_NOMESH>())); break;
    case PT_MODE_KD: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_KD>())); break;
    case PT_MODE_FLAT_KDMESH: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_FLAT_KDMESH>())); break;
    case PT_MODE_HIER: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_HIER>())); break;
    case PT_MODE_HIER_NOMESH: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_HIER_NOMESH>())); break;
    case PT_MODE_KD_NOMESH: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_KD_NOMESH>())); break;
    case PT_MODE_HIER_MESH: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_HIER_MESH>())); break;
    case PT_MODE_KD_MESH: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_KD_MESH>())); break;
    default: PT_HIP(c, cast(std::integral_constant<int, PT_MODE_FLAT>())); break;
    }
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipDeviceSynchronize());
    PT_HIP(c, hipMemcpy(out_t, d_t, n * 8, hipMemcpyDeviceToHost));
    PT_HIP(c, hipMemcpy(out_node, d_n, n * 4, hipMemcpyDeviceToHost));
    PT_HIP(c, hipMemcpy(out_sub, d_s, n * 4, hipMemcpyDeviceToHost));
    unsigned int head[2] = {0, 0};
    PT_HIP(c, hipMemcpy(head, c->misc.p, sizeof head, hipMemcpyDeviceToHost));
    if (head[1] & 4u) return pt_fail(c, PT_ERR_TRAVERSAL, "a tree walk did not end (watchdog): results invalid");
    if (head[1]) return pt_fail(c, PT_ERR_TRAVERSAL, "traversal stack overflow");
    return PT_OK;
}

extern "C" int pt_test_pow_host(uint64_t n, const double* x, const double* y, double* port, double* libm) {
    if (!x || !y || !port || !libm) return PT_ERR_ARGUMENT;
    for (uint64_t i = 0; i < n; i++) { port[i] = pt_pow_glibc(x[i], y[i]); libm[i] = pow(x[i], y[i]); }
    return PT_OK;
}

// The HOST's libm (glibc: what the reference and the oracle call) on explicit inputs, op numbered like pt_test_math: 2 pow, 4 atan2, 5 acos.
// (numpy's vectorised routines are not libm's on every machine: tests compare the device with this.)
extern "C" int pt_test_libm_host(int op, uint64_t n, const double* a, const double* b, double* out) {
    if (!a || !b || !out) return PT_ERR_ARGUMENT;
    for (uint64_t i = 0; i < n; i++) {
        switch (op) {
        case 2: out[i] = pow(a[i], b[i]); break;
        case 4: out[i] = atan2(a[i], b[i]); break;
        case 5: out[i] = acos(a[i]); break;
        default: return PT_ERR_ARGUMENT;
        }
    }
    return PT_OK;
}

extern "C" int pt_test_math(pt_context* c, int op, uint64_t n, const double* a, const double* b, double* out) {
    if (!c || !a || !b || !out) return PT_ERR_ARGUMENT;
    if (n == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    struct Bufs {
        void* p[3] = {nullptr, nullptr, nullptr};
        ~Bufs() { for (void* q : p) if (q) hipFree(q); }
    } bufs;
    for (int k = 0; k < 3; k++) PT_HIP(c, hipMalloc(&bufs.p[k], n * 8));
    double *d_a = (double*)bufs.p[0], *d_b = (double*)bufs.p[1], *d_o = (double*)bufs.p[2];
    PT_HIP(c, hipMemcpy(d_a, a, n * 8, hipMemcpyHostToDevice));
    PT_HIP(c, hipMemcpy(d_b, b, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pt_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, n, d_a, d_b, d_o);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipDeviceSynchronize());
    PT_HIP(c, hipMemcpy(out, d_o, n * 8, hipMemcpyDeviceToHost));
    return PT_OK;
}
