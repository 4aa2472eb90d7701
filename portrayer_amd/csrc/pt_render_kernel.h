// The render kernel (render.rs:127-150 and everything below it) and its launcher. Included by
// pt_render_inst.hip, which is compiled once per traversal mode (seven objects built side by side).
//
// Kernel design (CDNA4: 64-wide wavefronts, 256 CUs in 8 XCDs, 160 KB LDS per CU, no matrix cores involved -
// there is no dense contraction on this path):
//  * persistent wavefronts: the grid is what is resident on the chip. A wavefront takes work ITEMS - P neighbouring
//    pixels of an 8x8 tile x C consecutive 8-sample chunks x K samples of a chunk, P C K = 64 (SAMPLES = 64: all 64
//    samples of ONE pixel) - from a private batch it refills with ONE atomicAdd on the launch's counter (guided batch
//    sizes). All 64 lanes start their samples together and the item ends when the last lane has finished: at any
//    time the wavefront walks the trees with rays of ONE kind (primary, shadow to light 0, ...) through one pixel
//    (or a compact window of a few), which is what keeps its lanes in step inside the walk (profiles/r02/notes.md).
//  * one traversal loop per wavefront for all ray kinds: pt_lane_advance() turns whatever the lane traced
//    last into its next ray, so secondary rays re-enter the same loop instead of recursing.
//  * in the flat_scene and hierarchical semantics the trees are walked ONCE PER WAVEFRONT (pt_trace.h: pt_trace_packet,
//    pt_trace_packet_mesh): node index and stack are scalars, nodes and leaf records come through the scalar cache; the
//    k-d tree semantics keep a walk per lane.
//  * LDS: the traversal stack (the wavefront's, or one column per lane, overflowing to HBM beyond the LDS part), per lane the
//    frame of the hit being shaded and - scenes with reflective materials - its youngest parked recursion frame.
//  * the chunk's samples are added in ascending order by the lane of the pixel's first sample, reading its
//    neighbours' finished colours from LDS; pt_finish_kernel adds the chunk sums in order.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "pt_shade.h"

// VAR (template parameter), chosen per scene in pt_scene_upload:
//   0  hits never spawn rays; 3 waves per SIMD (168 VGPRs). The code and lane state for parked recursion frames are compiled out.
//   1  scenes with reflective materials: the youngest parked recursion frame of a lane stays in LDS (PARK); 3 waves per SIMD.
//   2  like 0 with 4 waves per SIMD (128 VGPRs): with the wave-uniform walk the traversal needs few registers of its own, and
//      traversal-heavy scenes gain from the fourth wave (big-scene 19.1 -> 20.8 Gray/s, big-soup 6.9 -> 7.4) where shading-heavy
//      and reflective ones lose (cows -8 %, mirror -7 %, transmission-refraction -21 %; profiles/r02/notes.md).
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 0  // experiments: -DPT_MIN_WAVES=2 / 3 / 4 for every instantiation
#endif
#ifndef PT_INTERP_WAVES
#define PT_INTERP_WAVES 3  // experiments: -DPT_INTERP_WAVES=2 compiles the interpreter kernels (VAR 0, 1, 3) for 2 waves per SIMD (256 VGPRs)
#endif
#define PT_VAR_WAVES(VAR) (PT_MIN_WAVES ? PT_MIN_WAVES : ((VAR) == 2 ? 4 : PT_INTERP_WAVES))  // VAR 3 = VAR 1 + fork / join of refracted subtrees

__device__ __forceinline__ void pt_flush_counters(PtCounters* dst, const PtCounters& c) {
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(&c);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
    for (unsigned i = 0; i < sizeof(PtCounters) / sizeof(unsigned long long); i++)
        if (s[i]) atomicAdd(d + i, s[i]);
}

// LDS of one block: [stack_lds_cap x PT_BLOCK words of traversal stack][hit-frame doubles x PT_BLOCK][park_slots x 12 doubles x PT_BLOCK]
__host__ __device__ inline size_t pt_render_lds_bytes(int stack_lds_cap, bool tex, int park_slots) {
    return (size_t)stack_lds_cap * PT_BLOCK * 4 + (size_t)(PT_LDS_FRAME_F64 + park_slots * PT_PARK_F64) * PT_BLOCK * 8;
}

#include "pt_render_simple.h"

template <int MODE, bool STATS, bool TEX, int VAR>
__global__ void __launch_bounds__(PT_BLOCK, PT_VAR_WAVES(VAR)) pt_render_kernel(PtRenderArgs a0) {
    const PtRenderArgs& a = a0;
    constexpr int PARK = (VAR == 1 || VAR == 3) ? 1 : 0;
    constexpr bool FORK = VAR == 3;  // idle lanes take the refracted subtrees busy lanes offer (pt_shade.h: fork / join)
    extern __shared__ uint32_t pt_lds[];
    const uint32_t lane_global = blockIdx.x * PT_BLOCK + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    PtStackSpill stk;
    stk.base = pt_lds + threadIdx.x;
    stk.cap = a.stack_lds_cap;
    stk.total = a.scene.stack_cap;
    stk.gbase = a.stack_spill + lane_global;
    stk.gstride = a.n_lanes;
    stk.overflow = a.overflow_flag;
    PtFrameRef fr;
    fr.lds = reinterpret_cast<double*>(pt_lds + (size_t)a.stack_lds_cap * PT_BLOCK) + threadIdx.x;
    fr.park = fr.lds + (size_t)PT_LDS_FRAME_F64 * PT_FRAME_STRIDE;
    fr.spill = a.spill + (size_t)lane_global * (PT_SPILL_DEPTHS * PT_SPILL_STRIDE);
    fr.n_lanes = a.n_lanes;

    PtCounters cnt;
    if (STATS) memset(&cnt, 0, sizeof cnt);
    PtLane L;
    L.stage = PT_ST_DONE; L.has_ray = false; L.ray_any = false;
    L.item = 0; L.x = L.y = 0; L.light = L.draw = L.draw0 = L.occluded = 0; L.depth = 0; L.lo = 0;
    L.ray.o = L.ray.d = pt_v3(0.0, 0.0, 0.0);
    L.offer = false; L.base = 0; L.owner = 0; L.fork_seq = 0; L.ticket = 0; L.wait_ticket = 0;
    PtHit hit;
    hit.t = INFINITY; hit.node = PT_NO_HIT; hit.sub = 0;
    unsigned q_next = 0, q_end = 0, q_seen = 0;  // this wavefront's private batch of items (wave-uniform); highest item index seen handed out
    if (a.fine_queues) q_next = blockIdx.x % a.fine_queues;
#ifdef PT_TIMELINE  // profiles/timeline.sh: when wavefronts start and end, and how long the longest item takes (100 MHz ticks)
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_item_max = 0;
#endif

    for (;;) {
        unsigned w;
        if (a.fine_queues == 0u) {
            // Next item of the wavefront's private batch; ONE atomicAdd per batch on the launch's counter (a device-scope
            // atomic round trip stalls the whole wavefront). Guided batch size: large while plenty of work remains, one item
            // at a time near the end, so the launch's tail stays short.
            if (q_next == q_end) {
                unsigned remaining = a.n_items > q_seen ? a.n_items - q_seen : 0u;
                unsigned take = remaining / a.work_div;
                take = take > a.batch_max ? a.batch_max : (take < 1u ? 1u : take);
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(a.work_counter, take);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                q_next = base; q_end = base + take; q_seen = q_end;
            }
            const unsigned q = q_next++;
            if (q >= a.n_items) break;
            w = a.item_stride == 1u ? q : (unsigned)(((unsigned long long)q * a.item_stride) % a.n_items);
        } else {
            // Scenes whose hits spawn rays: an item in the glass costs hundreds of times one beside it, and such items come in
            // large regions. Any batch of consecutive items may then be a batch of heavy ones (one wavefront busy for several
            // launch durations), so items are taken ONE at a time, in image order - every wavefront reaches the heavy region
            // at the same time and the region is shared item by item. To keep 3072 wavefronts off a single counter there are N
            // queues, queue g holding the items g, g + N, g + 2N ...; a wavefront starts at the queue of its block and moves on
            // to the next one when a queue is empty (q_next = its queue, q_end = empty queues seen in a row).
            for (;;) {
                unsigned idx = 0;
                if (lane == 0) idx = atomicAdd(a.work_queues + q_next * PT_QUEUE_STRIDE, 1u);
                idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
                const unsigned long long pos = (unsigned long long)idx * a.fine_queues + q_next;
                if (pos < a.n_items) { w = (unsigned)pos; q_end = 0; break; }
                q_next = q_next + 1u == a.fine_queues ? 0u : q_next + 1u;
                if (++q_end == a.fine_queues) { w = 0xFFFFFFFFu; break; }
            }
            if (w == 0xFFFFFFFFu) break;
        }
#ifdef PT_TIMELINE
        const unsigned long long tl_item = wall_clock64();
#endif
        {
            PtItemLane it0;
            uint32_t x0, y0;
            const bool mine0 = pt_item_lane_fast(a, w, lane, &it0, &x0, &y0);
            L.item = w;
            L.x = x0; L.y = y0;
            L.stage = mine0 ? PT_ST_NEW_SAMPLE : PT_ST_DONE;
        }
        L.has_ray = false;
        unsigned idle_passes = 0;  // FORK: passes in a row in which no lane traced anything (all waiting for mailboxes)
        for (;;) {
            const bool active = L.stage != PT_ST_DONE;
            if (!__any(active)) break;
#ifdef PT_DIAG
            if (STATS) { if (lane == 0) cnt.diag[2]++; if (active) cnt.diag[3]++; }
#endif
#ifdef PT_CYCLES
            const unsigned long long cyc_a = __builtin_readcyclecounter();
#endif
#ifndef PT_NO_ARGS_AGAIN
            // what the interpreter and this pass's walk need of the arguments is fetched now, not kept from the top of the kernel on (pt_render_simple.h).
            // (Round 4 kept this out of <PT_MODE_KD, true, *, 0>, which rendered wrongly with it. Round 5 found why - the compiler's vector register
            // allocator had put a spill store of `best.t` in front of the s_or_b64 exec that re-converges the block, so the lanes that sat out a leaf lost
            // their nearest hit: profiles/r05/notes.md section 1 - and the build now checks and repairs every kernel's assembly for it
            // (tools/check_exec_prologue.py), so no instantiation is special any more.)
            const PtRenderArgs& a = pt_args_again(a0);
#endif
            constexpr bool HIER = MODE == PT_MODE_HIER || MODE == PT_MODE_HIER_NOMESH || MODE == PT_MODE_HIER_MESH;
            uint32_t pre = 0;
#ifdef PT_MAPS_BEFORE
            if (TEX) {
                // Hits whose material has a texture or a normal map: the maps are applied here, outside the state machine (whose
                // registers would otherwise be spilled around the map code on every pass), for the lanes that need them.
                const bool maps = active && L.stage == PT_ST_CLOSEST_DONE && pt_hit_has_maps(a.scene, hit);
                if (__any(maps)) { if (maps) pre = pt_lane_maps<HIER>(a.scene, L.ray, hit, fr); }
            }
#endif
            if (active) pt_lane_advance<STATS, TEX, HIER, PARK, FORK>(a, L, hit, fr, &cnt, pre);
            if (FORK) {
                // Offers and takers, matched by rank: the k-th lane that parked a frame with a refracted ray in this pass writes its
                // thread index to slot k of the wavefront's queue in LDS (the first words of its traversal stack, free between
                // walks); the k-th idle lane reads it and takes the ray out of that lane's parked frame.
                const bool offer = active && L.offer;
                L.offer = false;
                const bool idle = L.stage == PT_ST_DONE;
                const unsigned long long offers = __ballot(offer), idles = __ballot(idle);
                if (offers && idles) {  // wave-uniform
                    const int n_off = __builtin_popcountll(offers), n_idle = __builtin_popcountll(idles);
                    const int n = n_off < n_idle ? n_off : n_idle;
                    const unsigned long long below = (1ull << lane) - 1ull;
                    uint32_t* queue = pt_fork_queue<MODE>(a, pt_lds);
                    const int q = __builtin_popcountll(offers & below), r = __builtin_popcountll(idles & below);
                    const bool given = offer && q < n, takes = idle && r < n;
                    if (given) queue[q] = threadIdx.x;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const uint32_t owner_tid = takes ? queue[r] : threadIdx.x;
                    // the owner's pixel and the depth of its frame (its L.depth is already that of the reflected ray: frame = depth - 1)
                    const uint32_t ox = (uint32_t)__shfl((int)L.x, (int)(owner_tid & 63u)), oy = (uint32_t)__shfl((int)L.y, (int)(owner_tid & 63u));
                    const int od = __shfl(L.depth, (int)(owner_tid & 63u)) - 1;
                    if (takes) {
                        const double* of = fr.park + ((ptrdiff_t)owner_tid - (ptrdiff_t)threadIdx.x);  // the owner's parked frame (LDS column)
                        uint32_t omat, ofs;
                        PtFrameRef::unpack_tag(of[PT_H_TAG * PT_FRAME_STRIDE], &omat, &ofs);
                        L.ray.o = pt_v3(of[(PT_H_P + 0) * PT_FRAME_STRIDE], of[(PT_H_P + 1) * PT_FRAME_STRIDE], of[(PT_H_P + 2) * PT_FRAME_STRIDE]);
                        L.ray.d = pt_v3(of[(PT_H_DIR + 0) * PT_FRAME_STRIDE], of[(PT_H_DIR + 1) * PT_FRAME_STRIDE], of[(PT_H_DIR + 2) * PT_FRAME_STRIDE]);
                        L.x = ox; L.y = oy;
                        L.depth = od + 1; L.base = od + 1; L.lo = od + 1;
                        L.owner = owner_tid | ((uint32_t)od << 16);
                        L.ticket = pt_fork_ticket(a.launch_nonce, ofs >> PT_FS_SEQ_SHIFT, od);
                        L.draw = 2;
                        L.ray_any = false; L.has_ray = true; L.stage = PT_ST_CLOSEST_DONE;
                        if (STATS) { cnt.refract++; cnt.diag[6]++; }
                    }
                    __builtin_amdgcn_wave_barrier();  // the owners' frames were read before any of them is marked
                    if (given) {
                        uint32_t mat, fs;
                        PtFrameRef::unpack_tag(fr.p(0, PT_H_TAG), &mat, &fs);
                        fr.p(0, PT_H_TAG) = PtFrameRef::pack_tag(mat, fs | PT_FS_FORKED);
                    }
                }
            }
#ifdef PT_CYCLES
            const unsigned long long cyc_b = __builtin_readcyclecounter();
#endif
            const bool tracing = L.stage != PT_ST_DONE && L.has_ray;
#ifdef PT_DIAG
            if (STATS) {
                const unsigned long long tm = __ballot(tracing), am = __ballot(tracing && L.ray_any);
                if (lane == 0) { if (tm) cnt.diag[0]++; if (am) cnt.diag[6]++; }
                if (tracing) { cnt.diag[1]++; if (L.ray_any) cnt.diag[7]++; }
            }
#endif
            // One walk per wavefront (pt_trace_wave: pt_trace_packet / pt_trace_packet_mesh, every lane calls it) in the flat_scene and
            // hierarchical semantics; the k-d tree semantics keep the per-lane walk (per-ray ranges and order).
            if (__any(tracing)) { pt_trace_wave<MODE, STATS>(a, L.ray, tracing, L.ray_any, hit, stk, pt_lds, &cnt); idle_passes = 0; }
            else if (FORK && ++idle_passes > (1u << 16)) {
                // Every lane that is not finished waits for a colour no lane is working on: a defect of the fork / join bookkeeping.
                // Reported like a traversal failure (the render returns PT_ERR_TRAVERSAL) rather than hanging the GPU.
                if (lane == 0) atomicOr(a.overflow_flag, 2u);
                L.stage = PT_ST_DONE;
            }
#ifdef PT_CYCLES
            if (STATS && lane == 0) cnt.diag[2] += cyc_b - cyc_a;  // the walk's cycles are counted inside pt_trace_wave
#endif
        }
        // render.rs:36-43 under the summation contract: the chunk's samples in ascending order. A pixel's samples sit
        // in neighbouring lanes; their colours are in the lanes' LDS columns (same wavefront: program order suffices).
        if (FORK) {  // the samples' colours went through HBM: make the lanes' stores visible to the lane that adds them up
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // The lane's place in the item is worked out again (a dozen integer operations on the wave-uniform item index) rather
        // than kept across the loop above, where it would be four registers spilled to scratch memory by every wavefront for
        // every item - 2 GB of HBM writes per frame. The empty asm keeps the compiler from reusing the first computation.
        unsigned w_again = w;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(w_again));
#endif
        PtItemLane it;
        uint32_t x_again, y_again;
        const bool mine = pt_item_lane_fast(a, w_again, lane, &it, &x_again, &y_again);
        if (mine && it.first) {
            PtVec3 sum;
            if (FORK) {  // finished samples wait in the lanes' result slots in HBM (the LDS frames were reused by taken tasks)
                sum = fr.h3(PT_RESULT_DEPTH, PT_H_MAIL);
                for (uint32_t k = 1; k < it.count && k < PT_SAMPLE_CHUNK; k++) {
                    const double* o = fr.spill + (size_t)k * (PT_SPILL_DEPTHS * PT_SPILL_STRIDE) + PT_RESULT_DEPTH * PT_SPILL_STRIDE + PT_H_MAIL;
                    sum = sum + pt_v3(o[0], o[1], o[2]);
                }
            } else {
                sum = fr.l3(PT_L_VALUE);
                for (uint32_t k = 1; k < it.count && k < PT_SAMPLE_CHUNK; k++) {  // (bounded by a constant as well: profiles/r04/notes.md, the hang of round 3's 6-wave build)
                    const double* o = fr.lds + k;
                    sum = sum + pt_v3(o[(PT_L_VALUE + 0) * PT_FRAME_STRIDE], o[(PT_L_VALUE + 1) * PT_FRAME_STRIDE], o[(PT_L_VALUE + 2) * PT_FRAME_STRIDE]);
                }
            }
            double* o = a.accum + 3 * ((size_t)it.slot * a.n_chunks + it.chunk);  // pixel-major: the chunk sums of a pixel side by side (see pt_finish_pixel)
            o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
        }
        __builtin_amdgcn_wave_barrier();
#ifdef PT_TIMELINE
        { const unsigned long long d = wall_clock64() - tl_item; if (d > tl_item_max) tl_item_max = d; }
#endif
    }
#ifdef PT_TIMELINE
    if (STATS && lane == 0) {
        const unsigned long long tl_end = wall_clock64();
        cnt.diag[0] = tl_end - tl_start; cnt.diag[1] = 1;                       // summed: wavefront lifetimes, wavefronts
        atomicMax(&a.counters->diag[2], tl_item_max);                             // longest item
        atomicMax(&a.counters->diag[4], ~tl_start);                               // ~(earliest start)
        atomicMax(&a.counters->diag[5], tl_end);                                  // latest end
        cnt.diag[2] = cnt.diag[4] = cnt.diag[5] = 0;
        // lifetimes in eighths of ... no kernel-wide clock is known here: the host divides
    }
#endif
    if (STATS) pt_flush_counters(a.counters, cnt);
}

// Launch (or, with launch = false, only size) one instantiation. The grid is what is resident: blocks per CU from
// the occupancy query for this kernel with its LDS. The raised LDS limit and the occupancy are properties of (kernel, device):
// kept per device, so that one process can drive several GPUs (pt_node).
#define PT_MAX_DEVICES 64
// KERNEL is a non-type template parameter: every kernel instantiation gets its own `state` (templated on the kernel's TYPE - void(*)(PtRenderArgs)
// for all of them - one array was shared by all kernels of a translation unit, and a kernel inherited the occupancy and the raised LDS limit of
// whichever kernel with the same LDS size had been launched first: ADVICE r03).
template <auto KERNEL>
static hipError_t pt_launch_kernel(size_t lds, const PtRenderArgs& a, int n_cu, hipStream_t stream, uint32_t* grid_out, bool launch) {
    constexpr auto kernel = KERNEL;
    struct PerDevice { size_t lds_allowed = 64 * 1024; size_t occ_lds = ~(size_t)0; int per_cu = 0; };
    static PerDevice state[PT_MAX_DEVICES];  // one per kernel instantiation and device
    static std::mutex state_lock;            // pt_node launches every rank from its own host thread, and ranks may share a device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    PerDevice local;
    PerDevice& st = (dev >= 0 && dev < PT_MAX_DEVICES) ? state[dev] : local;
    int per_cu = 0;
    {
    std::lock_guard<std::mutex> hold(state_lock);  // (held over the two runtime calls as well: they happen once per (kernel, device, LDS size))
    if (lds > st.lds_allowed) {  // gfx950 has 160 KB of LDS per CU; more than 64 KB per block must be asked for
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        st.lds_allowed = lds;
    }
    if (st.occ_lds != lds) {
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, PT_BLOCK, lds);
        if (e != hipSuccess) return e;
        st.per_cu = occ < 1 ? 1 : occ;
        st.occ_lds = lds;
    }
    per_cu = st.per_cu;
    }
    uint32_t want = (a.n_items + (PT_BLOCK / 64) - 1) / (PT_BLOCK / 64);
    if (const char* env = getenv("PORTRAYER_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(env)));  // experiment: fewer resident lanes
    const uint32_t resident = std::max<uint32_t>((uint32_t)(n_cu * per_cu) / std::max<uint32_t>(a.grid_share, 1u), 1u);  // (grid_share: contexts launching side by side on one GPU)
    uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(want, 1u), resident);
    *grid_out = grid;
    if (!launch) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PT_BLOCK), lds, stream, a);
    return hipGetLastError();
}

// Which kernel runs (`variant`, chosen in pt_render_common):
//   PT_RUN_INTERP / PT_RUN_INTERP_PARK   the interpreter kernel (scenes with reflective materials: hits spawn rays), every parked
//                                        recursion frame in HBM / the youngest in LDS; 3 waves per SIMD
//   PT_RUN_INTERP_FORK                   PT_RUN_INTERP_PARK + fork / join: idle lanes take the refracted subtrees busy lanes offer (pt_shade.h)
//   PT_RUN_LINE3 / PT_RUN_LINE4 / _LINE5 the straight-line kernel of pt_render_simple.h (hits spawn nothing), 3 / 4 / 5 waves per SIMD
//   PT_RUN_CHAIN                         the straight-line kernel with a loop over the depth: scenes whose reflective materials are all opaque (3 waves)
//   PT_RUN_INTERP4                       -DPT_KEEP_INTERP builds only: the interpreter at 4 waves per SIMD on a scene without reflective
//                                        materials (what round 2 timed), for A/B runs against the straight-line kernel
template <int MODE, bool STATS, bool TEX>
static hipError_t pt_launch_variant(const PtRenderArgs& a, int variant, bool kd_mode, int n_cu, hipStream_t stream, uint32_t* grid_out, bool launch) {
    const size_t lds = pt_render_lds_bytes(a.stack_lds_cap, TEX, (variant == PT_RUN_INTERP_PARK || variant == PT_RUN_INTERP_FORK) ? 1 : 0);
    switch (variant) {
    case PT_RUN_INTERP_PARK: return pt_launch_kernel<&pt_render_kernel<MODE, STATS, TEX, 1>>(lds, a, n_cu, stream, grid_out, launch);
    case PT_RUN_INTERP_FORK: return pt_launch_kernel<&pt_render_kernel<MODE, STATS, TEX, 3>>(lds, a, n_cu, stream, grid_out, launch);
    case PT_RUN_INTERP: return pt_launch_kernel<&pt_render_kernel<MODE, STATS, TEX, 0>>(lds, a, n_cu, stream, grid_out, launch);
#ifdef PT_KEEP_INTERP
    case PT_RUN_INTERP4:
        if constexpr (MODE != PT_MODE_KD && MODE != PT_MODE_KD_NOMESH && MODE != PT_MODE_KD_MESH) return pt_launch_kernel<&pt_render_kernel<MODE, STATS, TEX, 2>>(lds, a, n_cu, stream, grid_out, launch);
        return pt_launch_kernel<&pt_render_kernel<MODE, STATS, TEX, 0>>(lds, a, n_cu, stream, grid_out, launch);
#endif
    case PT_RUN_CHAIN:
        if constexpr (MODE != PT_MODE_KD)
            if (a.four_waves == 4) return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, 4, true>>(lds, a, n_cu, stream, grid_out, launch);
        return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, 3, true>>(lds, a, n_cu, stream, grid_out, launch);
    case PT_RUN_LINE5:  // mesh-free scenes in the flat_scene / hierarchical semantics: 96 registers, 5 waves per SIMD
        if constexpr (MODE == PT_MODE_FLAT_NOMESH || MODE == PT_MODE_HIER_NOMESH) return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, PT_LINE_TOP_WAVES>>(lds, a, n_cu, stream, grid_out, launch);
        if constexpr (MODE == PT_MODE_KD_NOMESH) return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, 5>>(lds, a, n_cu, stream, grid_out, launch);  // (its saved range bounds want the LDS rows: 5 at most)
        if constexpr (MODE == PT_MODE_FLAT || MODE == PT_MODE_HIER_MESH) return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, PT_MESH_TOP_WAVES>>(lds, a, n_cu, stream, grid_out, launch);  // scenes of very many triangles: their walks wait for node fetches
        [[fallthrough]];
    case PT_RUN_LINE4:  // the k-d tree semantics with mesh instances (per-lane walk through two levels of trees) have no 4-wave instantiation
        if constexpr (MODE != PT_MODE_KD) return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, 4>>(lds, a, n_cu, stream, grid_out, launch);
        [[fallthrough]];
    default: return pt_launch_kernel<&pt_render_simple_kernel<MODE, STATS, TEX, 3>>(lds, a, n_cu, stream, grid_out, launch);
    }
}

template <int MODE>
static hipError_t pt_dispatch_variant(const PtRenderArgs& a, int variant, bool stats, bool tex, int n_cu, hipStream_t stream, uint32_t* grid, bool launch) {
    if (tex) return stats ? pt_launch_variant<MODE, true, true>(a, variant, false, n_cu, stream, grid, launch) : pt_launch_variant<MODE, false, true>(a, variant, false, n_cu, stream, grid, launch);
    return stats ? pt_launch_variant<MODE, true, false>(a, variant, false, n_cu, stream, grid, launch) : pt_launch_variant<MODE, false, false>(a, variant, false, n_cu, stream, grid, launch);
}
