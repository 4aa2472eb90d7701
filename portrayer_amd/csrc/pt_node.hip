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
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
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

// One thread per rank that makes the rank's HIP calls of a frame (a render launch is a memset, two kernels, three events and a small copy:
// tens of microseconds of host time; eight ranks launched one after the other from one thread made the last GPU start a third of a
// millisecond late, against a share of a frame that takes one). A job is handed over under a mutex; the caller waits for all of them.
struct RankWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, quit = false;
    void start() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                j();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(j); has_job = true; done = false;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return done; });
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

#define PT_NODE_FRAMES 2  // frames in flight: frame k + 1 renders (into the other set of tile buffers) while frame k is gathered and untiled

struct pt_node {
    std::vector<int> devices;
    std::vector<pt_context*> ctx;
    std::vector<hipStream_t> stream;    // per rank: its renders
    std::vector<hipStream_t> gstream;   // per rank: its part of the gather (rank 0: also the untile), so that the next frame's render need not wait for it
    std::vector<hipEvent_t> done[PT_NODE_FRAMES];      // per rank, on `stream`: the frame's tiles are rendered
    std::vector<hipEvent_t> gathered[PT_NODE_FRAMES];  // per rank, on `gstream`: the frame's tile buffer has been sent (rank 0: the image is assembled)
    std::vector<ncclComm_t> comm;       // empty: ranks share a device, gather by copies
    std::vector<void*> d_bg;            // per rank
    std::vector<void*> d_compact[PT_NODE_FRAMES];      // per rank: the tile buffers, one set per frame in flight
    std::vector<size_t> bg_bytes, compact_bytes[PT_NODE_FRAMES];
    void* d_gathered = nullptr; size_t gathered_bytes = 0;  // rank 0
    void* d_full = nullptr; size_t full_bytes = 0;          // rank 0
    std::vector<RankWorker*> worker;    // PORTRAYER_NODE_THREADS=0: none, the calling thread launches every rank
    std::vector<int> rank_rc;           // what the ranks' launches of the frame being begun returned
    std::string err;
    bool have_scene = false;
    size_t bg_ready_bytes = 0;  // size of the background pt_node_upload_background left on every rank
    uint64_t frame_begun = 0, frame_ended = 0;  // frames begun / closed (begun - ended <= PT_NODE_FRAMES)
    uint64_t frame_used[PT_NODE_FRAMES] = {0, 0};  // how often a buffer set has been used (its `gathered` events are valid from the first use on)
    int fail_after_launch = -1;  // PORTRAYER_NODE_FAIL_AFTER_LAUNCH=<rank> (tests): see pt_node_frame_begin
    bool one_stream = false;     // PORTRAYER_NODE_ONE_STREAM=1: see pt_node_frame_begin
    std::vector<double> rank_kernel_ms;    // of the last frame closed: every rank's kernel time (HIP events around its launches)
    double host_ms[5] = {0, 0, 0, 0, 0};   // of the last frame: begin (launches + gather queued), wait (blocked until the image is complete), finish (flags, counters), per-rank launch (the slowest), the ranks' kernel times added up
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
    n->ctx.assign(ranks, nullptr); n->stream.assign(ranks, nullptr); n->gstream.assign(ranks, nullptr);
    n->d_bg.assign(ranks, nullptr); n->bg_bytes.assign(ranks, 0); n->rank_rc.assign(ranks, PT_OK);
    for (int f = 0; f < PT_NODE_FRAMES; f++) {
        n->done[f].assign(ranks, nullptr); n->gathered[f].assign(ranks, nullptr);
        n->d_compact[f].assign(ranks, nullptr); n->compact_bytes[f].assign(ranks, 0);
    }
    auto bail = [&](int code) { pt_node_destroy(n); return code; };
    for (size_t r = 0; r < ranks; r++) {
        if (pt_context_create(n->devices[r], &n->ctx[r]) != PT_OK) return bail(PT_ERR_DEVICE);
        if (hipSetDevice(n->devices[r]) != hipSuccess || hipStreamCreateWithFlags(&n->stream[r], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&n->gstream[r], hipStreamNonBlocking) != hipSuccess)
            return bail(PT_ERR_DEVICE);
        for (int f = 0; f < PT_NODE_FRAMES; f++)
            if (hipEventCreateWithFlags(&n->done[f][r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&n->gathered[f][r], hipEventDisableTiming) != hipSuccess)
                return bail(PT_ERR_DEVICE);
    }
    bool threads = ranks > 1;
    if (const char* e = getenv("PORTRAYER_NODE_THREADS")) threads = atoi(e) > 0;
    if (const char* e = getenv("PORTRAYER_NODE_FAIL_AFTER_LAUNCH")) n->fail_after_launch = atoi(e);
    if (const char* e = getenv("PORTRAYER_NODE_ONE_STREAM")) n->one_stream = atoi(e) > 0;
    if (threads) {
        n->worker.assign(ranks, nullptr);
        for (size_t r = 0; r < ranks; r++) { n->worker[r] = new RankWorker(); n->worker[r]->start(); }
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
    for (RankWorker* w : n->worker) if (w) { w->stop(); delete w; }
    for (size_t r = 0; r < n->devices.size(); r++) {
        hipSetDevice(n->devices[r]);
        if (r < n->stream.size() && n->stream[r]) hipStreamSynchronize(n->stream[r]);
        if (r < n->ctx.size() && n->ctx[r]) for (int k = 0; k < 2; k++) hipStreamSynchronize((hipStream_t)pt_context_stream(n->ctx[r], k));
        if (r < n->gstream.size() && n->gstream[r]) hipStreamSynchronize(n->gstream[r]);
        if (r < n->comm.size() && n->comm[r]) g_rccl.CommDestroy(n->comm[r]);
        if (r < n->d_bg.size() && n->d_bg[r]) hipFree(n->d_bg[r]);
        for (int f = 0; f < PT_NODE_FRAMES; f++) {
            if (r < n->d_compact[f].size() && n->d_compact[f][r]) hipFree(n->d_compact[f][r]);
            if (r < n->done[f].size() && n->done[f][r]) hipEventDestroy(n->done[f][r]);
            if (r < n->gathered[f].size() && n->gathered[f][r]) hipEventDestroy(n->gathered[f][r]);
        }
        if (r < n->stream.size() && n->stream[r]) hipStreamDestroy(n->stream[r]);
        if (r < n->gstream.size() && n->gstream[r]) hipStreamDestroy(n->gstream[r]);
        if (r < n->ctx.size() && n->ctx[r]) pt_context_destroy(n->ctx[r]);
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
    // the scene buffers are read by the renders of open frames (non-blocking streams: the upload's synchronous copies are not ordered against them)
    if (n->frame_begun != n->frame_ended) return node_fail(n, PT_ERR_ARGUMENT, "frames are in flight: pt_node_frame_end first");
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
    // before anything is reallocated: an open frame's untile targets d_full, its renders read d_bg
    if (n->frame_begun != n->frame_ended) return node_fail(n, PT_ERR_ARGUMENT, "frames are in flight: pt_node_frame_end first");
    if ((rc = node_reserve(n, n->devices[0], &n->d_full, &n->full_bytes, px * 3))) return rc;
    NODE_HIP(n, hipSetDevice(n->devices[0]));
    // pixels outside the slice keep the caller's bytes (render.rs:135-138); without a caller's image they are zero, not whatever the buffer held
    if (rgb) NODE_HIP(n, hipMemcpyAsync(n->d_full, rgb, px * 3, hipMemcpyHostToDevice, n->gstream[0]));
    else NODE_HIP(n, hipMemsetAsync(n->d_full, 0, px * 3, n->gstream[0]));
    for (uint32_t r = 0; r < ranks; r++) {
        if ((rc = node_reserve(n, n->devices[r], &n->d_bg[r], &n->bg_bytes[r], bg_bytes))) return rc;
        NODE_HIP(n, hipSetDevice(n->devices[r]));
        NODE_HIP(n, hipMemcpyAsync(n->d_bg[r], background, bg_bytes, hipMemcpyHostToDevice, n->stream[r]));
    }
    // The copies read the CALLER's host buffers, which it may reuse as soon as this call returns: wait for them. (Stream order alone would
    // protect the renders that follow; it does not protect the caller's memory.)
    for (uint32_t r = 0; r < ranks; r++) {
        NODE_HIP(n, hipSetDevice(n->devices[r]));
        NODE_HIP(n, hipStreamSynchronize(n->stream[r]));
    }
    NODE_HIP(n, hipSetDevice(n->devices[0]));
    NODE_HIP(n, hipStreamSynchronize(n->gstream[0]));
    n->bg_ready_bytes = bg_bytes;
    return PT_OK;
}

// Something failed while a frame was being queued: wait for everything in flight and close every rank's open launches, so that no
// kernel is left running into buffers the caller may free and no context stays flagged as busy.
static void node_drain(pt_node* n) {
    for (size_t r = 0; r < n->ctx.size(); r++) {
        hipSetDevice(n->devices[r]);
        hipStreamSynchronize(n->stream[r]);
        for (int k = 0; k < 2; k++) hipStreamSynchronize((hipStream_t)pt_context_stream(n->ctx[r], k));
        hipStreamSynchronize(n->gstream[r]);
        for (int k = 0; k < 4 && pt_render_finish(n->ctx[r], nullptr) != PT_ERR_ARGUMENT; k++) { }  // PT_ERR_ARGUMENT: nothing in flight
    }
    n->frame_ended = n->frame_begun;
}

// A frame, queued: every rank renders its tiles into the frame's buffer set on its own stream (launched by the rank's thread), the ONE
// gather and the untile follow on the ranks' gather streams. Returns without waiting; at most PT_NODE_FRAMES frames may be open.
extern "C" int pt_node_frame_begin(pt_node* n, const pt_camera* camera, const pt_render_params* params) {
    if (!n || !camera || !params) return PT_ERR_ARGUMENT;
    int rc = node_check_params(n, params);
    if (rc) return rc;
    if (n->frame_begun - n->frame_ended >= PT_NODE_FRAMES) return node_fail(n, PT_ERR_ARGUMENT, "too many frames in flight: pt_node_frame_end first");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t ranks = (uint32_t)n->ctx.size();
    const size_t px = (size_t)params->width * params->height;
    const size_t bg_bytes = (params->background_rows ? (size_t)params->height : px) * 3 * sizeof(double);
    if (n->bg_ready_bytes != bg_bytes || n->full_bytes < px * 3) return node_fail(n, PT_ERR_ARGUMENT, "pt_node_upload_background must come first (same width, height and background_rows)");
    const int f = (int)(n->frame_begun % PT_NODE_FRAMES);
    pt_render_params p = *params;
    p.tile_ranks = ranks;
    const size_t per = (size_t)pt_compact_bytes(&p);
    bool grew = n->gathered_bytes < per * ranks;
    for (int g = 0; g < PT_NODE_FRAMES; g++)
        for (uint32_t r = 0; r < ranks; r++) grew = grew || n->compact_bytes[g][r] < per;
    if (grew && n->frame_begun != n->frame_ended) {  // (a buffer an open frame uses is not reallocated under it)
        return node_fail(n, PT_ERR_ARGUMENT, "a larger frame needs new buffers: pt_node_frame_end the open frames first");
    }
    if ((rc = node_reserve(n, n->devices[0], &n->d_gathered, &n->gathered_bytes, per * ranks))) return rc;
    for (int g = 0; g < PT_NODE_FRAMES; g++)  // every buffer set at once: the next frame may be begun while this one is open
        for (uint32_t r = 0; r < ranks; r++)
            if ((rc = node_reserve(n, n->devices[r], &n->d_compact[g][r], &n->compact_bytes[g][r], per))) return rc;
    const bool reuse = n->frame_used[f] > 0;
    // per rank: wait until the buffer set's previous frame has been sent, render, mark, and let the gather stream wait for the mark
    std::vector<double> launch_ms(ranks, 0.0);
    std::vector<std::string> hip_after_launch(ranks);  // a HIP call behind a rank's launch failed: its text (the launch itself is open and must be closed)
    auto launch = [&, f, reuse](uint32_t r) {
        const auto l0 = std::chrono::steady_clock::now();
        int rr = PT_OK;
        hipError_t e = hipSetDevice(n->devices[r]);
        // The frame renders on the stream of the context slot it takes: two open frames are on two queues, and the wavefronts of frame k + 1 start in the
        // places frame k's tail frees (pt_context::Slot, csrc/pt_api.hip; PORTRAYER_NODE_ONE_STREAM=1 keeps every frame on the rank's one stream, for A/B runs)
        hipStream_t rs = n->one_stream ? n->stream[r] : (hipStream_t)pt_context_stream(n->ctx[r], pt_context_next_slot(n->ctx[r]));
        if (e == hipSuccess && reuse) e = hipStreamWaitEvent(rs, n->gathered[f][r], 0);
        if (e == hipSuccess) {
            pt_render_params q = p;
            q.tile_rank = r;
            rr = pt_render_device(n->ctx[r], camera, (const double*)n->d_bg[r], &q, 1, n->d_compact[f][r], rs);
            if (rr == PT_OK) {
                e = hipEventRecord(n->done[f][r], rs);
                if (e == hipSuccess) e = hipStreamWaitEvent(n->gstream[r], n->done[f][r], 0);
                if (e != hipSuccess) { rr = PT_ERR_DEVICE; hip_after_launch[r] = std::string("a HIP call behind its launch failed: ") + hipGetErrorString(e); }  // launched: still to be closed
            }
        } else {
            rr = PT_ERR_DEVICE;
            hip_after_launch[r] = std::string("before its launch: ") + hipGetErrorString(e);
        }
        // fault injection (tests of this error path; read once in pt_node_create): the rank reports a failed HIP call behind a launch that did go out
        if (n->fail_after_launch == (int)r && rr == PT_OK) { rr = PT_ERR_DEVICE; hip_after_launch[r] = "a HIP call behind its launch failed: injected (PORTRAYER_NODE_FAIL_AFTER_LAUNCH)"; }
        n->rank_rc[r] = rr;
        launch_ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - l0).count();
    };
    if (!n->worker.empty()) {
        for (uint32_t r = 0; r < ranks; r++) n->worker[r]->post([&launch, r] { launch(r); });
        for (uint32_t r = 0; r < ranks; r++) n->worker[r]->wait();
    } else {
        for (uint32_t r = 0; r < ranks; r++) launch(r);
    }
    n->frame_begun++;  // (from here on node_drain has something of this frame to close)
    n->frame_used[f]++;
    for (uint32_t r = 0; r < ranks; r++)
        if (n->rank_rc[r] != PT_OK) {
            const std::string why = !hip_after_launch[r].empty() ? hip_after_launch[r] : std::string(pt_last_error(n->ctx[r]));  // (read before node_drain touches the context)
            const int code = n->rank_rc[r];  // always one of the header's negative codes
            node_drain(n);
            return node_fail(n, code, "rank " + std::to_string(r) + ": " + why);
        }
    auto fail_hip = [&](hipError_t e, const char* what) { node_drain(n); return node_fail(n, PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
    hipError_t e;
    if (per) {
        if (!n->comm.empty()) {  // the frame's ONE collective (the receive buffer only matters on the root)
            int g = g_rccl.GroupStart();
            for (uint32_t r = 0; r < ranks && g == kNcclSuccess; r++)
                g = g_rccl.Gather(n->d_compact[f][r], r == 0 ? n->d_gathered : n->d_compact[f][r], per, kNcclUint8, 0, n->comm[r], n->gstream[r]);
            int g2 = g_rccl.GroupEnd();
            if (g != kNcclSuccess || g2 != kNcclSuccess) {
                node_drain(n);
                return node_fail(n, PT_ERR_DEVICE, std::string("ncclGather: ") + g_rccl.GetErrorString(g != kNcclSuccess ? g : g2));
            }
        } else {  // ranks sharing a device: copies on rank 0's gather stream once each rank's tiles are done
            if ((e = hipSetDevice(n->devices[0])) != hipSuccess) return fail_hip(e, "hipSetDevice");
            for (uint32_t r = 0; r < ranks; r++) {
                if (r && (e = hipStreamWaitEvent(n->gstream[0], n->done[f][r], 0)) != hipSuccess) return fail_hip(e, "hipStreamWaitEvent");
                if ((e = hipMemcpyAsync((char*)n->d_gathered + r * per, n->d_compact[f][r], per, hipMemcpyDeviceToDevice, n->gstream[0])) != hipSuccess) return fail_hip(e, "hipMemcpyAsync");
            }
        }
    }
    p.tile_rank = 0;
    rc = pt_untile_device(n->ctx[0], &p, n->d_gathered, n->d_full, n->gstream[0]);
    if (rc != PT_OK) { node_drain(n); return node_fail(n, rc, std::string("rank 0: ") + pt_last_error(n->ctx[0])); }
    // the frame is complete when rank 0's gather stream is (its gather ends when every rank's tiles have arrived); a rank's buffer may be
    // rendered into again when its own gather stream has passed this point
    for (uint32_t r = 0; r < ranks; r++) {
        if ((e = hipSetDevice(n->devices[r])) != hipSuccess) return fail_hip(e, "hipSetDevice");
        if ((e = hipEventRecord(n->gathered[f][r], n->gstream[r])) != hipSuccess) return fail_hip(e, "hipEventRecord");
    }
    n->host_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    n->host_ms[3] = 0.0;
    for (double m : launch_ms) if (m > n->host_ms[3]) n->host_ms[3] = m;
    return PT_OK;
}

// Closes the OLDEST open frame: returns when its image is complete on rank 0 (and its tile buffers are free again).
extern "C" int pt_node_frame_end(pt_node* n, pt_stats* stats) {
    if (!n) return PT_ERR_ARGUMENT;
    if (n->frame_begun == n->frame_ended) return node_fail(n, PT_ERR_ARGUMENT, "no frame in flight");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t ranks = (uint32_t)n->ctx.size();
    const int f = (int)(n->frame_ended % PT_NODE_FRAMES);
    hipError_t e;
    auto fail_hip = [&](hipError_t err, const char* what) { node_drain(n); return node_fail(n, PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(err)); };
    for (uint32_t r = 0; r < ranks; r++) {  // rank 0's event: the image; the others': their sends (already done by then, RCCL's gather is one operation)
        if ((e = hipSetDevice(n->devices[r])) != hipSuccess) return fail_hip(e, "hipSetDevice");
        if ((e = hipEventSynchronize(n->gathered[f][r])) != hipSuccess) return fail_hip(e, "hipEventSynchronize");
    }
    const auto t1 = std::chrono::steady_clock::now();
    pt_stats total;
    memset(&total, 0, sizeof total);
    int first_rc = PT_OK;
    std::string first_err;
    for (uint32_t r = 0; r < ranks; r++) {  // every rank is closed, whatever the others report
        pt_stats st;
        memset(&st, 0, sizeof st);
        int rc = pt_render_finish(n->ctx[r], &st);
        if (rc != PT_OK && first_rc == PT_OK) { first_rc = rc; first_err = "rank " + std::to_string(r) + ": " + pt_last_error(n->ctx[r]); }
        total.primary += st.primary; total.shadow += st.shadow; total.reflect += st.reflect; total.refract += st.refract;
        total.depth11_skipped += st.depth11_skipped; total.hits += st.hits; total.n_inner += st.n_inner; total.n_leaf += st.n_leaf;
        total.n_analytic += st.n_analytic; total.n_tri += st.n_tri; total.n_bbox += st.n_bbox; total.kd_plane_miss += st.kd_plane_miss;
        total.stack_overflow += st.stack_overflow;
        for (int k = 0; k < 8; k++) total.diag[k] += st.diag[k];
        if (n->rank_kernel_ms.size() != ranks) n->rank_kernel_ms.assign(ranks, 0.0);
        n->rank_kernel_ms[r] = st.kernel_ms;
        if (st.kernel_ms > total.kernel_ms) total.kernel_ms = st.kernel_ms;  // the slowest rank's kernel
        total.total_ms += st.kernel_ms;                                     // (moved to host_ms[4] below) the ranks' kernel times added up
        total.kernel_mode = st.kernel_mode; total.kernel_variant = st.kernel_variant;
    }
    n->frame_ended++;
    const auto t2 = std::chrono::steady_clock::now();
    n->host_ms[1] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    n->host_ms[2] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    n->host_ms[4] = total.total_ms;
    if (first_rc != PT_OK) return node_fail(n, first_rc, first_err);
    total.total_ms = 0.0;
    if (stats) *stats = total;
    return PT_OK;
}

extern "C" int pt_node_frames_in_flight(const pt_node* n) { return n ? (int)(n->frame_begun - n->frame_ended) : 0; }

// Host time of the last frame's calls, milliseconds: [0] pt_node_frame_begin as a whole, [1] pt_node_frame_end blocked until the image was
// complete (GPU time, unless frames are pipelined), [2] pt_node_frame_end after that (flags, counters, event times), [3] the slowest
// rank's launch inside begin, [4] the ranks' kernel times (HIP events) added up. [0] + [2] is what the host adds to a frame that is not
// overlapped with another.
extern "C" int pt_node_last_frame_host_ms(const pt_node* n, double out[5]) {
    if (!n || !out) return PT_ERR_ARGUMENT;
    for (int k = 0; k < 5; k++) out[k] = n->host_ms[k];
    return PT_OK;
}

// Every rank's kernel time of the last frame closed (milliseconds, HIP events around the rank's launches). With two frames open the second frame's
// launch shares the GPU with the first one's tail, so its figure covers both: time one frame at a time for a rank's own cost.
extern "C" int pt_node_last_frame_rank_kernel_ms(const pt_node* n, double* out, int n_out) {
    if (!n || !out || n_out < 0) return PT_ERR_ARGUMENT;
    for (int r = 0; r < n_out; r++) out[r] = (size_t)r < n->rank_kernel_ms.size() ? n->rank_kernel_ms[r] : 0.0;
    return PT_OK;
}

extern "C" int pt_node_render_resident(pt_node* n, const pt_camera* camera, const pt_render_params* params, pt_stats* stats) {
    if (!n || !camera || !params) return PT_ERR_ARGUMENT;
    if (n->frame_begun != n->frame_ended) return node_fail(n, PT_ERR_ARGUMENT, "frames are in flight: pt_node_frame_end first");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = pt_node_frame_begin(n, camera, params);
    if (rc) return rc;
    rc = pt_node_frame_end(n, stats);
    if (rc) return rc;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}

extern "C" int pt_node_download_image(pt_node* n, const pt_render_params* params, uint8_t* rgb) {
    if (!n || !params || !rgb) return PT_ERR_ARGUMENT;
    const size_t px = (size_t)params->width * params->height;
    if (!n->d_full || n->full_bytes < px * 3) return node_fail(n, PT_ERR_ARGUMENT, "no image of this size on rank 0");
    if (n->frame_begun != n->frame_ended) return node_fail(n, PT_ERR_ARGUMENT, "frames are in flight (the image on rank 0 is being overwritten): pt_node_frame_end first");
    NODE_HIP(n, hipSetDevice(n->devices[0]));
    NODE_HIP(n, hipMemcpyAsync(rgb, n->d_full, px * 3, hipMemcpyDeviceToHost, n->gstream[0]));
    NODE_HIP(n, hipStreamSynchronize(n->gstream[0]));
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
