// pt_node_*: the image of one render call tile-partitioned over the GPUs of one node.
//
// One pt_context per GPU, the scene replicated on each; the slice's 8x8 tiles are dealt round-robin to the GPUs
// (tile t belongs to rank t % n - sky-only and geometry-heavy regions of an image balance that way, the reference's
// own rectangular slice API, render.rs:56-66, does not); every GPU renders its tiles into a compact tile-major
// buffer on its own stream; ONE gather over xGMI (RCCL: ncclGather inside a group, one call per rank, single
// process) brings the finished tiles to rank 0, which scatters them into the row-major image (pt_untile_device).
// There is no other communication: pixels are independent, no sum is reduced across GPUs (that would break the
// summation contract, render.rs:36-43).
//
// RCCL is loaded on first use (dlopen) so that single-GPU callers - and processes that already carry another
// copy of RCCL, such as a PyTorch process - never depend on it. Ranks that share a device (tests on a 1-GPU box:
// devices = {0, 0}) are gathered with device-to-device copies instead; RCCL refuses duplicate devices.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../include/portrayer_hip.h"

namespace {
// the few RCCL entry points used, with the types of rccl.h
typedef struct ncclComm* ncclComm_t;
enum { kNcclSuccess = 0, kNcclUint8 = 1 };
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool loaded = false;  // every entry point resolved
    std::string load() {
        if (loaded) return "";
        if (handle) { dlclose(handle); handle = nullptr; }  // an earlier attempt found the library but not every symbol
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return std::string("RCCL not found: ") + dlerror();
        auto sym = [&](const char* n) { return dlsym(handle, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Gather = (decltype(Gather))sym("ncclGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Gather || !GetErrorString) {
            dlclose(handle); handle = nullptr;
            return "RCCL lacks ncclGather / ncclCommInitAll";
        }
        loaded = true;
        return "";
    }
};
Rccl g_rccl;
}  // namespace

struct pt_node {
    std::vector<int> devices;
    std::vector<pt_context*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;       // per rank: its tiles are rendered (and, without RCCL, may be copied)
    std::vector<ncclComm_t> comm;       // empty: ranks share a device, gather by copies
    std::vector<void*> d_bg, d_compact; // per rank
    std::vector<size_t> bg_bytes, compact_bytes;
    void* d_gathered = nullptr; size_t gathered_bytes = 0;  // rank 0
    void* d_full = nullptr; size_t full_bytes = 0;          // rank 0
    std::string err;
    bool have_scene = false;
    size_t bg_ready_bytes = 0;  // size of the background pt_node_upload_background left on every rank
};

static int node_fail(pt_node* n, int code, const std::string& msg) {
    if (n) n->err = msg;
    return code;
}
#define NODE_HIP(n, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return node_fail(n, PT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NODE_CTX(n, r, call)                                                                                \
    do {                                                                                                    \
        int rc_ = (call);                                                                                   \
        if (rc_ != PT_OK) return node_fail(n, rc_, "rank " + std::to_string(r) + ": " + pt_last_error((n)->ctx[r])); \
    } while (0)

static int node_reserve(pt_node* n, int device, void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return PT_OK;
    NODE_HIP(n, hipSetDevice(device));
    if (*p) { NODE_HIP(n, hipFree(*p)); *p = nullptr; *have = 0; }
    NODE_HIP(n, hipMalloc(p, want ? want : 16));
    *have = want;
    return PT_OK;
}

extern "C" int pt_node_create(int n_devices, const int* devices, pt_node** out) {
    if (!out || n_devices <= 0) return PT_ERR_ARGUMENT;
    *out = nullptr;
    int visible = pt_device_count();
    if (visible <= 0) return PT_ERR_DEVICE;
    pt_node* n = new pt_node();
    for (int r = 0; r < n_devices; r++) {
        int d = devices ? devices[r] : r;
        if (d < 0 || d >= visible) { delete n; return PT_ERR_DEVICE; }
        n->devices.push_back(d);
    }
    const size_t ranks = n->devices.size();
    n->ctx.assign(ranks, nullptr); n->stream.assign(ranks, nullptr); n->done.assign(ranks, nullptr);
    n->d_bg.assign(ranks, nullptr); n->d_compact.assign(ranks, nullptr); n->bg_bytes.assign(ranks, 0); n->compact_bytes.assign(ranks, 0);
    auto bail = [&](int code) { pt_node_destroy(n); return code; };
    for (size_t r = 0; r < ranks; r++) {
        if (pt_context_create(n->devices[r], &n->ctx[r]) != PT_OK) return bail(PT_ERR_DEVICE);
        if (hipSetDevice(n->devices[r]) != hipSuccess || hipStreamCreateWithFlags(&n->stream[r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&n->done[r], hipEventDisableTiming) != hipSuccess)
            return bail(PT_ERR_DEVICE);
    }
    std::set<int> distinct(n->devices.begin(), n->devices.end());
    // PORTRAYER_NODE_RCCL=1: also for a single rank, so that a 1-GPU box can run the library loading, the communicator and
    // the gather call as they are (tests); a node of one GPU has nothing to gather and skips RCCL otherwise
    const bool lone_rank_too = getenv("PORTRAYER_NODE_RCCL") != nullptr;
    if ((ranks > 1 || lone_rank_too) && distinct.size() == ranks) {  // one GPU per rank: RCCL over xGMI
        std::string e = g_rccl.load();
        if (!e.empty()) { fprintf(stderr, "pt_node_create: %s\n", e.c_str()); return bail(PT_ERR_DEVICE); }
        n->comm.assign(ranks, nullptr);
        int rc = g_rccl.CommInitAll(n->comm.data(), (int)ranks, n->devices.data());
        if (rc != kNcclSuccess) { fprintf(stderr, "pt_node_create: ncclCommInitAll: %s\n", g_rccl.GetErrorString(rc)); n->comm.clear(); return bail(PT_ERR_DEVICE); }
    }
    *out = n;
    return PT_OK;
}

extern "C" void pt_node_destroy(pt_node* n) {
    if (!n) return;
    for (size_t r = 0; r < n->devices.size(); r++) {
        hipSetDevice(n->devices[r]);
        if (r < n->comm.size() && n->comm[r]) g_rccl.CommDestroy(n->comm[r]);
        if (n->d_bg[r]) hipFree(n->d_bg[r]);
        if (n->d_compact[r]) hipFree(n->d_compact[r]);
        if (n->done[r]) hipEventDestroy(n->done[r]);
        if (n->stream[r]) hipStreamDestroy(n->stream[r]);
        if (n->ctx[r]) pt_context_destroy(n->ctx[r]);
    }
    if (!n->devices.empty()) {
        hipSetDevice(n->devices[0]);
        if (n->d_gathered) hipFree(n->d_gathered);
        if (n->d_full) hipFree(n->d_full);
    }
    delete n;
}

extern "C" const char* pt_node_last_error(const pt_node* n) { return n ? n->err.c_str() : "no node"; }
extern "C" int pt_node_ranks(const pt_node* n) { return n ? (int)n->devices.size() : 0; }
extern "C" int pt_node_uses_rccl(const pt_node* n) { return n && !n->comm.empty() ? 1 : 0; }
extern "C" pt_context* pt_node_context(pt_node* n, int rank) { return (n && rank >= 0 && (size_t)rank < n->ctx.size()) ? n->ctx[rank] : nullptr; }

extern "C" int pt_node_scene_upload(pt_node* n, const pt_scene* scene, int traverse, const pt_kdtree* kd) {
    if (!n || !scene) return PT_ERR_ARGUMENT;
    n->have_scene = false;
    for (size_t r = 0; r < n->ctx.size(); r++) NODE_CTX(n, r, pt_scene_upload(n->ctx[r], scene, traverse, kd));  // replicated: <= ~100 MB even for 1.25 M triangles
    n->have_scene = true;
    return PT_OK;
}

extern "C" int pt_node_device(const pt_node* n, int rank) { return (n && rank >= 0 && (size_t)rank < n->devices.size()) ? n->devices[rank] : -1; }

static int node_check_params(pt_node* n, const pt_render_params* params) {
    if (!n->have_scene) return node_fail(n, PT_ERR_NO_SCENE, "no scene uploaded");
    if (params->tile_ranks != 1 || params->tile_rank != 0) return node_fail(n, PT_ERR_ARGUMENT, "pt_node_render partitions the tiles itself: tile_rank / tile_ranks must be 0 / 1");
    if (params->width == 0 || params->height == 0) return node_fail(n, PT_ERR_ARGUMENT, "width and height must be positive");
    return PT_OK;
}

extern "C" int pt_node_upload_background(pt_node* n, const double* background, const pt_render_params* params, const uint8_t* rgb) {
    if (!n || !background || !params) return PT_ERR_ARGUMENT;
    int rc = node_check_params(n, params);
    if (rc) return rc;
    const uint32_t ranks = (uint32_t)n->ctx.size();
    const size_t px = (size_t)params->width * params->height;
    const size_t bg_bytes = (params->background_rows ? (size_t)params->height : px) * 3 * sizeof(double);
    if ((rc = node_reserve(n, n->devices[0], &n->d_full, &n->full_bytes, px * 3))) return rc;
    NODE_HIP(n, hipSetDevice(n->devices[0]));
    // pixels outside the slice keep the caller's bytes (render.rs:135-138)
    if (rgb) NODE_HIP(n, hipMemcpyAsync(n->d_full, rgb, px * 3, hipMemcpyHostToDevice, n->stream[0]));
    for (uint32_t r = 0; r < ranks; r++) {
        if ((rc = node_reserve(n, n->devices[r], &n->d_bg[r], &n->bg_bytes[r], bg_bytes))) return rc;
        NODE_HIP(n, hipSetDevice(n->devices[r]));
        NODE_HIP(n, hipMemcpyAsync(n->d_bg[r], background, bg_bytes, hipMemcpyHostToDevice, n->stream[r]));
    }
    for (uint32_t r = 0; r < ranks; r++) {
        NODE_HIP(n, hipSetDevice(n->devices[r]));
        NODE_HIP(n, hipStreamSynchronize(n->stream[r]));
    }
    n->bg_ready_bytes = bg_bytes;
    return PT_OK;
}

// Ranks 0 .. launched - 1 have a render in flight and something failed: wait for their streams and close their launches, so
// that no kernel is left running into buffers the caller may free and no context stays flagged as busy.
static void node_abort(pt_node* n, uint32_t launched) {
    for (uint32_t r = 0; r < launched; r++) {
        hipSetDevice(n->devices[r]);
        hipStreamSynchronize(n->stream[r]);
        pt_render_finish(n->ctx[r], nullptr);
    }
}

extern "C" int pt_node_render_resident(pt_node* n, const pt_camera* camera, const pt_render_params* params, pt_stats* stats) {
    if (!n || !camera || !params) return PT_ERR_ARGUMENT;
    int rc = node_check_params(n, params);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t ranks = (uint32_t)n->ctx.size();
    const size_t px = (size_t)params->width * params->height;
    const size_t bg_bytes = (params->background_rows ? (size_t)params->height : px) * 3 * sizeof(double);
    if (n->bg_ready_bytes != bg_bytes || n->full_bytes < px * 3) return node_fail(n, PT_ERR_ARGUMENT, "pt_node_upload_background must come first (same width, height and background_rows)");
    pt_render_params p = *params;
    p.tile_ranks = ranks;
    const size_t per = (size_t)pt_compact_bytes(&p);
    if ((rc = node_reserve(n, n->devices[0], &n->d_gathered, &n->gathered_bytes, per * ranks))) return rc;
    for (uint32_t r = 0; r < ranks; r++)
        if ((rc = node_reserve(n, n->devices[r], &n->d_compact[r], &n->compact_bytes[r], per))) return rc;
    uint32_t launched = 0;
    auto fail_hip = [&](hipError_t e, const char* what) { node_abort(n, launched); return node_fail(n, PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
    hipError_t e;
    for (uint32_t r = 0; r < ranks; r++) {
        if ((e = hipSetDevice(n->devices[r])) != hipSuccess) return fail_hip(e, "hipSetDevice");
        p.tile_rank = r;
        rc = pt_render_device(n->ctx[r], camera, (const double*)n->d_bg[r], &p, 1, n->d_compact[r], n->stream[r]);
        if (rc != PT_OK) { node_abort(n, launched); return node_fail(n, rc, "rank " + std::to_string(r) + ": " + pt_last_error(n->ctx[r])); }
        launched = r + 1;
        if ((e = hipEventRecord(n->done[r], n->stream[r])) != hipSuccess) return fail_hip(e, "hipEventRecord");
    }
    if (per) {
        if (!n->comm.empty()) {  // the frame's ONE collective (the receive buffer only matters on the root)
            int g = g_rccl.GroupStart();
            for (uint32_t r = 0; r < ranks && g == kNcclSuccess; r++)
                g = g_rccl.Gather(n->d_compact[r], r == 0 ? n->d_gathered : n->d_compact[r], per, kNcclUint8, 0, n->comm[r], n->stream[r]);
            int g2 = g_rccl.GroupEnd();
            if (g != kNcclSuccess || g2 != kNcclSuccess) {
                node_abort(n, launched);
                return node_fail(n, PT_ERR_DEVICE, std::string("ncclGather: ") + g_rccl.GetErrorString(g != kNcclSuccess ? g : g2));
            }
        } else {  // ranks sharing a device: copies on rank 0's stream once each rank's tiles are done
            if ((e = hipSetDevice(n->devices[0])) != hipSuccess) return fail_hip(e, "hipSetDevice");
            for (uint32_t r = 0; r < ranks; r++) {
                if (r && (e = hipStreamWaitEvent(n->stream[0], n->done[r], 0)) != hipSuccess) return fail_hip(e, "hipStreamWaitEvent");
                if ((e = hipMemcpyAsync((char*)n->d_gathered + r * per, n->d_compact[r], per, hipMemcpyDeviceToDevice, n->stream[0])) != hipSuccess) return fail_hip(e, "hipMemcpyAsync");
            }
        }
    }
    p.tile_rank = 0;
    rc = pt_untile_device(n->ctx[0], &p, n->d_gathered, n->d_full, n->stream[0]);
    if (rc != PT_OK) { node_abort(n, launched); return node_fail(n, rc, std::string("rank 0: ") + pt_last_error(n->ctx[0])); }
    // the frame is complete when rank 0's stream is (its gather ends when every rank's tiles have arrived); the other ranks'
    // streams end with their send
    for (uint32_t r = 0; r < ranks; r++) {
        if ((e = hipSetDevice(n->devices[r])) != hipSuccess) return fail_hip(e, "hipSetDevice");
        if ((e = hipStreamSynchronize(n->stream[r])) != hipSuccess) return fail_hip(e, "hipStreamSynchronize");
    }
    pt_stats total;
    memset(&total, 0, sizeof total);
    int first_rc = PT_OK;
    std::string first_err;
    for (uint32_t r = 0; r < ranks; r++) {  // every rank is closed, whatever the others report
        pt_stats st;
        rc = pt_render_finish(n->ctx[r], &st);
        if (rc != PT_OK && first_rc == PT_OK) { first_rc = rc; first_err = "rank " + std::to_string(r) + ": " + pt_last_error(n->ctx[r]); }
        total.primary += st.primary; total.shadow += st.shadow; total.reflect += st.reflect; total.refract += st.refract;
        total.depth11_skipped += st.depth11_skipped; total.hits += st.hits; total.n_inner += st.n_inner; total.n_leaf += st.n_leaf;
        total.n_analytic += st.n_analytic; total.n_tri += st.n_tri; total.n_bbox += st.n_bbox; total.kd_plane_miss += st.kd_plane_miss;
        total.stack_overflow += st.stack_overflow;
        if (st.kernel_ms > total.kernel_ms) total.kernel_ms = st.kernel_ms;  // the slowest rank's kernel
        total.kernel_mode = st.kernel_mode; total.kernel_variant = st.kernel_variant;
    }
    if (first_rc != PT_OK) return node_fail(n, first_rc, first_err);
    total.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = total;
    return PT_OK;
}

extern "C" int pt_node_download_image(pt_node* n, const pt_render_params* params, uint8_t* rgb) {
    if (!n || !params || !rgb) return PT_ERR_ARGUMENT;
    const size_t px = (size_t)params->width * params->height;
    if (!n->d_full || n->full_bytes < px * 3) return node_fail(n, PT_ERR_ARGUMENT, "no image of this size on rank 0");
    NODE_HIP(n, hipSetDevice(n->devices[0]));
    NODE_HIP(n, hipMemcpyAsync(rgb, n->d_full, px * 3, hipMemcpyDeviceToHost, n->stream[0]));
    NODE_HIP(n, hipStreamSynchronize(n->stream[0]));
    return PT_OK;
}

extern "C" int pt_node_render(pt_node* n, const pt_camera* camera, const double* background, const pt_render_params* params,
                              uint8_t* rgb, pt_stats* stats) {
    if (!n || !camera || !background || !params || !rgb) return PT_ERR_ARGUMENT;
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if ((rc = pt_node_upload_background(n, background, params, rgb))) return rc;
    if ((rc = pt_node_render_resident(n, camera, params, stats))) return rc;
    if ((rc = pt_node_download_image(n, params, rgb))) return rc;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}
