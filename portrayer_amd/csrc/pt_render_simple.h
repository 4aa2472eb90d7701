// The render kernel for scenes whose hits spawn no rays (no reflective material: material.rs:216 never takes the branch) -
// render.rs:127-150 and everything below it as ONE straight line per work item, no interpreter:
//
//     primary ray (camera.rs:48-84)  ->  nearest hit  ->  surface + material (flat_scene.rs:85-95, material.rs:109-144)
//     for every light in order (material.rs:149-210):  shadow ray -> any hit ? nothing : colour += (diffuse + specular) / attenuation
//     -> the sample's colour (ray.rs:139-148), summed with its chunk (the summation contract, pt_shade.h)
//
// Without recursion every sample of a work item goes through the same sequence of rays, so the 64 lanes of a wavefront stay in
// step by construction: what pt_lane_advance() (pt_shade.h, still the kernel of reflective scenes) does with a stage word, a
// dispatch loop and a frame in LDS per lane is here plain control flow, and a light's direction and distance are worked out once -
// for the shadow ray - and used again for the shading term (the interpreter recomputed them in its SHADE stage: the same
// expressions, so the same bits). The light record is wave-uniform and comes through the scalar cache.
//
// Registers: nothing but the accumulated colour, the distance to the light and the material tag is live across a shadow ray's
// walk besides the ray itself (the ray's origin IS the hit point, its direction IS the light direction); the shading normal and
// the incoming direction wait in the lane's LDS column.
#pragma once

#include "pt_shade.h"

// doubles at a wave-uniform address through the scalar cache (N = 2 or 4)
template <int N>
PT_HD void pt_sload_f64(const double* p, double* out) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (N == 4) {
        pt_u32x8 v = pt_sload8(p);
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = pt_f64_of(v[2 * k], v[2 * k + 1]);
    } else {
        pt_u32x4 v = pt_sload4(p);
#pragma unroll
        for (int k = 0; k < 2; k++) out[k] = pt_f64_of(v[2 * k], v[2 * k + 1]);
    }
#else
    for (int k = 0; k < N; k++) out[k] = p[k];
#endif
}

// The kernel's argument block seen AGAIN through a pointer the compiler cannot trace back to the kernarg segment: what is read
// through it is fetched (scalar loads, constant cache) where it is used instead of being kept in scalar registers from the top of
// the kernel on. The straight-line kernels keep ~200 wave-uniform values alive (scene pointers, camera, item geometry); what does
// not fit the 106 scalar registers is spilled into lanes of a vector register and comes back one v_readlane each - VALU issue slots,
// 180 of them per work item before this (38 for the camera alone, at the start of every item).
PT_HD const PtRenderArgs& pt_args_again(const PtRenderArgs& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) PtRenderArgs* ka = (const __attribute__((address_space(4))) PtRenderArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(const PtRenderArgs*)ka;
#else
    return a;
#endif
}

// Kernels whose lanes need traversal stacks of their own (KDMesh trees) split the block's LDS stack area: `rows` x 64 entries for each
// wavefront's own stack, the rest for the lanes' (which continue in HBM, PtStackSpill). The wavefront's stack has no HBM part: it gets as
// many rows as the deepest walk of THIS scene can have pending (scene.stack_cap bounds that), at the lanes' expense if need be - so a deep
// device-built tree costs lane rows, not the render (PT_ERR_TRAVERSAL is left for scenes beyond stack_lds_cap x 64 entries).
PT_HD int pt_wave_rows(const PtRenderArgs& a) {
    const int cap = a.stack_lds_cap;
    int rows = cap >= 16 ? 8 : (cap >= 8 ? 3 : (cap >= 4 ? 2 : 1));
    const int need = (a.scene.stack_cap + 63) / 64;
    if (need > rows) rows = need < cap ? need : cap;
    return rows;
}

// The block's LDS stack area in the k-d tree semantics (one walk per wavefront, pt_trace_packet_kd): `lane_rows` rows (PT_BLOCK words each)
// of per-lane stack columns - only scenes with KDMesh trees have any (pt_kdmesh_hit) -, then per wavefront `wrows` rows of 64 words:
// `srows` for its stack (a word per pending split + what a mesh's triangle tree can have pending), the rest two rows per tree level for
// the lanes' saved range bounds - the deepest levels; the top ones, touched once or twice per ray, in the lanes' HBM columns.
struct PtKdLayout { int lane_rows, wrows, srows, lds_levels, path_word; };  // path_word: where in the wavefront's region the path table starts (words)
template <int MODE>
PT_HD PtKdLayout pt_kd_layout(const PtRenderArgs& a) {
    PtKdLayout l;
    const int cap = a.stack_lds_cap;
    l.srows = (a.scene.stack_cap + 63) / 64;
    l.lane_rows = 0;
    if (MODE == PT_MODE_KD && a.scene.mkd) {
        int r = cap - l.srows - 2 * a.scene.kd_levels;
        r = r < 2 ? 2 : r;
        r = r > cap - l.srows ? cap - l.srows : r;
        l.lane_rows = r < 0 ? 0 : r;
    }
    l.wrows = cap - l.lane_rows;  // >= 1
    if (l.wrows < l.srows + 2 && 3 * a.scene.kd_levels > l.srows * 64 - a.scene.stack_cap) {  // room for the path table comes first (pt_render_common keeps cap >= srows + 2 in these semantics)
        l.wrows = cap < l.srows + 2 ? cap : l.srows + 2;
        l.lane_rows = cap - l.wrows;
    }
    // the levels that get no slot are recomputed from the path table (PtKdSav, pt_trace.h): 3 words per such level, in the slack behind the wavefront's stack
    // (its rows hold srows x 64 words, the walks need scene.stack_cap of them) or, where that is too small, in two rows taken from the level slots
    int lv = (l.wrows - l.srows) / 2;
    lv = lv < 0 ? 0 : (lv > a.scene.kd_levels ? a.scene.kd_levels : lv);
    const int slack = l.srows * 64 - a.scene.stack_cap;
    if (3 * (a.scene.kd_levels - lv) <= slack && l.wrows >= l.srows) {
        l.path_word = a.scene.stack_cap;
    } else {  // (3 x 32 levels = 96 words: two rows always do)
        lv = (l.wrows - l.srows - 2) / 2;
        lv = lv < 0 ? 0 : (lv > a.scene.kd_levels ? a.scene.kd_levels : lv);
        l.path_word = (l.srows + 2 * lv) * 64;
    }
    l.lds_levels = lv;
    return l;
}

// One ray kind for the whole wavefront: `tracing` lanes carry `ray`; result in `hit` (untouched for the other lanes).
template <int MODE, bool STATS>
PT_HD void pt_trace_wave(const PtRenderArgs& a, const PtRay& ray, bool tracing, bool any, PtHit& hit, const PtStackSpill& stk, uint32_t* lds, PtCounters* cnt) {
#ifdef PT_CYCLES  // profiles/cycles.sh: wave cycles inside the walks -> diag[0], calls -> diag[1]
    const unsigned long long cyc_t0 = __builtin_readcyclecounter();
    struct CycEnd { unsigned long long t0; PtCounters* c; __device__ ~CycEnd() { if (STATS && (threadIdx.x & 63u) == 0) { c->diag[0] += __builtin_readcyclecounter() - t0; c->diag[1]++; } } } cyc_end{cyc_t0, cnt};
#endif
#if defined(PT_ABLATE) && (PT_ABLATE == 5 || PT_ABLATE == 6)  // measurement builds only (profiles/light_slope.py): 5 = no shadow walks (nothing is ever in the way), 6 = no walks at all
    if (any || PT_ABLATE == 6) return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t wave = threadIdx.x >> 6;
#else
    const uint32_t wave = 0;
#endif
#ifndef PT_NO_PACKET
    // The wavefront's stack is a linear region of the block's LDS stack area: all of the wavefront's share where no lane needs a
    // stack of its own, else the part behind the lanes' rows (KDMesh trees are walked per lane, pt_kdmesh_hit).
    if (MODE == PT_MODE_FLAT_NOMESH || MODE == PT_MODE_HIER_NOMESH) {
        const int words = a.stack_lds_cap * 64;
        pt_trace_packet<STATS, MODE == PT_MODE_HIER_NOMESH>(a.scene, ray, tracing, any, hit, lds + (size_t)wave * words, words, a.overflow_flag, cnt);
    } else if (MODE == PT_MODE_FLAT || MODE == PT_MODE_HIER_MESH) {
        const int words = a.stack_lds_cap * 64;
        pt_trace_packet_mesh<STATS, false, MODE == PT_MODE_HIER_MESH>(a.scene, ray, tracing, any, hit, lds + (size_t)wave * words, words, stk, a.overflow_flag, cnt);
    } else if (MODE == PT_MODE_FLAT_KDMESH || MODE == PT_MODE_HIER) {
        const int rows = pt_wave_rows(a);  // of the wavefront's stack (64 entries each); the lanes' own stacks get the rest
        PtStackSpill lane_stk = stk;
        lane_stk.cap = a.stack_lds_cap - rows;
        pt_trace_packet_mesh<STATS, true, MODE == PT_MODE_HIER>(a.scene, ray, tracing, any, hit, lds + (size_t)lane_stk.cap * PT_BLOCK + (size_t)wave * rows * 64, rows * 64, lane_stk,
                                                                a.overflow_flag, cnt);
    } else if (MODE == PT_MODE_KD_NOMESH || MODE == PT_MODE_KD || MODE == PT_MODE_KD_MESH) {  // (pt_scene_upload refuses trees of more than PT_KD_WAVE_LEVELS levels)
        const PtKdLayout kl = pt_kd_layout<MODE>(a);
        const int lane_rows = kl.lane_rows, wrows = kl.wrows, srows = kl.srows, lds_levels = kl.lds_levels;
        PtStackSpill lane_stk = stk;
        lane_stk.cap = lane_rows;
        uint32_t* wbase = lds + (size_t)lane_rows * PT_BLOCK + (size_t)wave * wrows * 64;
        PtKdSav sav;
        sav.lds = wbase + (size_t)srows * 64;
        sav.top_levels = a.scene.kd_levels - lds_levels;
        sav.path = wbase + kl.path_word;
        int stack_words = (wrows < srows ? wrows : srows) * 64;
        if (kl.path_word < stack_words) stack_words = kl.path_word;  // (the path table sits in the slack of the stack's rows: the stack ends where it begins)
        pt_trace_packet_kd<STATS, MODE == PT_MODE_KD || MODE == PT_MODE_KD_MESH, MODE == PT_MODE_KD>(a.scene, ray, tracing, any, hit, wbase, stack_words, sav, lane_stk, a.overflow_flag, cnt);
    }
#else
    if (tracing) pt_trace<MODE, STATS>(a.scene, ray, any, hit, stk, cnt);
#endif
}

// The first words of the wavefront's own traversal stack (free between walks): the queue through which offered rays find takers
// (pt_render_kernel, FORK). At least 64 words: the wavefront's region is stack_lds_cap x 64 words, or `rows` x 64 behind the lanes' rows.
template <int MODE>
PT_HD uint32_t* pt_fork_queue(const PtRenderArgs& a, uint32_t* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t wave = threadIdx.x >> 6;
#else
    const uint32_t wave = 0;
#endif
    if (MODE == PT_MODE_FLAT_KDMESH || MODE == PT_MODE_HIER) {
        const int rows = pt_wave_rows(a);
        return lds + (size_t)(a.stack_lds_cap - rows) * PT_BLOCK + (size_t)wave * rows * 64;
    }
    if (MODE == PT_MODE_KD || MODE == PT_MODE_KD_NOMESH || MODE == PT_MODE_KD_MESH) { const PtKdLayout kl = pt_kd_layout<MODE>(a); return lds + (size_t)kl.lane_rows * PT_BLOCK + (size_t)wave * kl.wrows * 64; }  // the first row of the wavefront's region: its stack, free between walks
    return lds + (size_t)wave * a.stack_lds_cap * 64;
}

// CHAIN (round 3): scenes whose reflective materials are all OPAQUE (reflectivity > 0, no index of refraction - the mirror scene,
// glossy metal). Their recursion (material.rs:216-243, :312-316) is a chain, not a tree: every hit spawns at most ONE ray, so the
// samples of a work item still go through the same sequence of ray kinds - bounce by bounce - and the straight line only needs a
// loop over the depth: trace, shade (shadow rays as above), and where the material reflects, park {colour so far, reflectivity} in the
// lane's HBM line of that depth and go on with the reflected ray (glossy: its two draws follow the area lights' in the sampling
// contract's order, which is this loop's order). A chain that ends - background, a matte surface, depth 10 - is folded back at once,
// innermost first: value = colour_k + value * reflectivity_k, exactly the reference's `color += reflectivity * reflected_color`.
// Lanes whose chains have ended idle until the wavefront's longest chain has; no interpreter, no stage word, no frame in LDS.
template <int MODE, bool STATS, bool TEX, int WAVES, bool CHAIN = false>
__global__ void __launch_bounds__(PT_BLOCK, WAVES) pt_render_simple_kernel(PtRenderArgs a0) {
    constexpr bool HIER = MODE == PT_MODE_HIER || MODE == PT_MODE_HIER_NOMESH || MODE == PT_MODE_HIER_MESH;
    extern __shared__ uint32_t pt_lds[];
    const PtRenderArgs& a = a0;
    const uint32_t lane_global = blockIdx.x * PT_BLOCK + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    [[maybe_unused]] const PtSceneView& sc = a.scene;  // (shadowed per item by the re-read arguments' view unless -DPT_NO_ARGS_AGAIN)
    PtStackSpill stk;
    stk.base = pt_lds + threadIdx.x;
    stk.cap = a.stack_lds_cap;
    stk.total = a.scene.stack_cap;
    stk.gbase = a.stack_spill + lane_global;
    stk.gstride = a.n_lanes;
    stk.overflow = a.overflow_flag;
    PtFrameRef fr;
    fr.lds = reinterpret_cast<double*>(pt_lds + (size_t)a.stack_lds_cap * PT_BLOCK) + threadIdx.x;
    fr.park = fr.lds;
    fr.spill = CHAIN ? a.spill + (size_t)lane_global * (PT_SPILL_DEPTHS * PT_SPILL_STRIDE) : nullptr;
    fr.n_lanes = a.n_lanes;

    PtCounters cnt;
    if (STATS) memset(&cnt, 0, sizeof cnt);
    unsigned q_next = 0, q_end = 0, q_seen = 0;  // the wavefront's private batch of items (wave-uniform), as in pt_render_kernel
    if (a.fine_queues) q_next = blockIdx.x % a.fine_queues;
#ifdef PT_TIMELINE  // profiles/timeline.sh: when wavefronts start and end, and how long the longest item takes (100 MHz ticks), as in pt_render_kernel
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_item_max = 0, tl_first_item = 0;
#endif

    for (;;) {
        unsigned w;
        if (a.fine_queues == 0u) {  // guided batches from one counter (very long launches, the k-d tree semantics)
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
        } else {  // one item at a time from interleaved queues
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
        if (!tl_first_item) tl_first_item = tl_item;
#endif
#ifdef PT_CYCLES  // sections of an item outside the walks: diag[3] primary ray, diag[4] surface of the hit, diag[6] light + shadow ray set-up, diag[7] light term
        const unsigned long long cyc_item0 = __builtin_readcyclecounter();
        unsigned long long sec_t0 = cyc_item0;
#define PT_SEC_SKIP() sec_t0 = __builtin_readcyclecounter()
#define PT_SEC_END(k) do { if (STATS && lane == 0) cnt.diag[k] += __builtin_readcyclecounter() - sec_t0; sec_t0 = __builtin_readcyclecounter(); } while (0)
#else
#define PT_SEC_SKIP() do { } while (0)
#define PT_SEC_END(k) do { } while (0)
#endif
#ifndef PT_NO_ARGS_AGAIN  // (shadows the kernel's own view of its arguments for the rest of the item. The k-d semantics of mesh-free scenes are as fast without
        // - 30.5 against 30.7 ms on big-scene, c22 / c50 -; with mesh instances they gain like the others: mirror 8.7 -> 9.6 Gray/s, macho-cows 6.7 -> 7.2, c50)
        const PtRenderArgs& a = pt_args_again(a0);
        const PtSceneView& sc = a.scene;
#endif
        // ---- the primary ray of this lane's sample (render.rs:36-41, camera.rs:48-84)
        uint32_t x, y;
        bool mine;
        PtRay ray;
        ray.o = ray.d = pt_v3(0.0, 0.0, 0.0);
        {
            PtItemLane it;
            mine = pt_item_lane_fast(a, w, lane, &it, &x, &y);
            if (mine) {
                double jx = 0.5, jy = 0.5;
                if (a.jitter_mode == PT_JITTER_RNG) {  // render.rs:38-39: x drawn before y
                    const uint64_t key = pt_rng_key(a.seed, (uint64_t)y * a.width + x);
                    jx = pt_rng_draw(key, it.sample, 0);
                    jy = pt_rng_draw(key, it.sample, 1);
                }
                ray = pt_camera_ray(a.cam, (double)x + jx, (double)y + jy);
                if (STATS) cnt.primary++;
            }
        }
        bool live = mine;     // the lane's chain of rays goes on (without CHAIN: one round)
        uint32_t draw = 2;    // the two jitter draws came first. CHAIN: per lane (which lanes draw depends on what their rays hit) ...
        uint32_t udraw = 2;   // ... otherwise the same for every lane: kept where the compiler can see that it is wave-uniform
        for (int depth = 0; CHAIN ? __any(live) : depth == 0; depth++) {
        PtHit hit;
        hit.t = INFINITY; hit.node = PT_NO_HIT; hit.sub = 0;
        PT_SEC_END(3);
        pt_trace_wave<MODE, STATS>(a, ray, live, false, hit, stk, pt_lds, &cnt);
        PT_SEC_SKIP();

        // ---- ray.rs:139-148: the background where nothing was hit, else Material::hit_color (material.rs:91-243)
        const bool shaded = live && hit.node != PT_NO_HIT;
        bool ended = live && !shaded;  // this lane's chain ends in this round: `value` is set, the parked colours get folded in
        PtVec3 value = pt_v3(0.0, 0.0, 0.0);  // (not kept in registers across the rounds: a finished chain's colour goes to the lane's LDS column at once)
        if (ended) value = pt_background(a, x, y);
        if (__any(shaded)) {
            uint32_t mat = 0, ftag = 0;
            PtRay sray;  // o = the hit point for every light; d = the direction to the light being tested
            sray.o = sray.d = pt_v3(0.0, 0.0, 0.0);
            PtVec3 color = pt_v3(0.0, 0.0, 0.0);
            if (shaded) {
                if (STATS) cnt.hits++;
                PtVec3 N;
                pt_hit_surface<TEX, HIER>(sc, ray, hit, &sray.o, &N, &mat, &ftag);
                fr.set_l3(PT_L_N, N);
                fr.set_l3(PT_L_D, ray.d);
                const double* m = sc.materials + 10 * (size_t)mat;
                PtVec3 kd = pt_v3(m[0], m[1], m[2]);
                if (TEX && (ftag & PT_FS_TEXEL)) kd = pt_v3(sc.srgb_lut[ftag & 255u], sc.srgb_lut[(ftag >> 8) & 255u], sc.srgb_lut[(ftag >> 16) & 255u]);
                color = pt_v3(sc.ambient[0], sc.ambient[1], sc.ambient[2]) * kd;  // material.rs:148
#ifndef PT_NO_LDS_COLOR  // the colour so far and the distance to the light wait in the lane's LDS column (slots of P and the tag, unused here) while the shadow ray is walked: eight registers the walks need more
                fr.set_l3(PT_L_P, color);
#endif
            }
            PT_SEC_END(4);
            for (uint32_t li = 0; li < sc.n_lights; li++) {  // material.rs:149-210, one light after the other
                const double* light = sc.lights + 15 * (size_t)li;
                double lp[4], la[2], lb[4];  // position (+ colour.x), area_a.xy, area_a.z + area_b
                pt_sload_f64<4>(light, lp);
                pt_sload_f64<2>(light + 9, la);
                pt_sload_f64<4>(light + 11, lb);
                PtVec3 lpos = pt_v3(lp[0], lp[1], lp[2]);
                const PtVec3 aa = pt_v3(la[0], la[1], lb[0]), ab = pt_v3(lb[1], lb[2], lb[3]);
                const bool area = !((aa.x == 0.0 && aa.y == 0.0 && aa.z == 0.0) || (ab.x == 0.0 && ab.y == 0.0 && ab.z == 0.0));  // light.rs:51-53 (wave-uniform)
                double light_dist = 0.0;
                if (shaded) {
                    if (area) {  // light.rs:62-70, :87-90
                        PtItemLane it;
                        uint32_t x2, y2;
                        pt_item_lane_fast(a, w, lane, &it, &x2, &y2);
                        const uint64_t key = pt_rng_key(a.seed, (uint64_t)y * a.width + x);
                        const uint32_t d0 = CHAIN ? draw : udraw;
                        double a_coord = 2.0 * pt_rng_draw(key, it.sample, d0) - 1.0;
                        double b_coord = 2.0 * pt_rng_draw(key, it.sample, d0 + 1) - 1.0;
                        lpos = lpos + (aa * a_coord + ab * b_coord);
                        if (CHAIN) draw += 2;
                    }
                    PtVec3 hit_to_light = lpos - sray.o;
                    light_dist = pt_length(hit_to_light);
                    sray.d = hit_to_light / light_dist;
#ifndef PT_NO_LDS_COLOR
                    fr.l(PT_L_TAG) = light_dist;
#endif
                    if (STATS) cnt.shadow++;
                }
                if (area) udraw += 2;
                PtHit sh;
                sh.t = INFINITY; sh.node = PT_NO_HIT; sh.sub = 0;
                PT_SEC_END(6);
                pt_trace_wave<MODE, STATS>(a, sray, shaded, true, sh, stk, pt_lds, &cnt);  // material.rs:174-179 only asks whether anything is in the way
                PT_SEC_SKIP();
                if (shaded && sh.node == PT_NO_HIT) {
                    double lc[2], lf[4];  // colour.xy, colour.z + falloff
                    pt_sload_f64<2>(light + 3, lc);
                    pt_sload_f64<4>(light + 5, lf);
                    const double* m = sc.materials + 10 * (size_t)mat;
                    PtVec3 kd = pt_v3(m[0], m[1], m[2]), ks = pt_v3(m[3], m[4], m[5]);
                    if (TEX && (ftag & PT_FS_TEXEL)) kd = pt_v3(sc.srgb_lut[ftag & 255u], sc.srgb_lut[(ftag >> 8) & 255u], sc.srgb_lut[(ftag >> 16) & 255u]);
#ifndef PT_NO_LDS_COLOR
                    fr.set_l3(PT_L_P, fr.l3(PT_L_P) + pt_light_term(pt_v3(lc[0], lc[1], lf[0]), pt_v3(lf[1], lf[2], lf[3]), sray.d, fr.l(PT_L_TAG), fr.l3(PT_L_N), fr.l3(PT_L_D), kd, ks, m[6]));
#else
                    color = color + pt_light_term(pt_v3(lc[0], lc[1], lf[0]), pt_v3(lf[1], lf[2], lf[3]), sray.d, light_dist, fr.l3(PT_L_N), fr.l3(PT_L_D), kd, ks, m[6]);
#endif
                }
                PT_SEC_END(7);
            }
            if (shaded) {
#ifndef PT_NO_LDS_COLOR
                color = fr.l3(PT_L_P);
#endif
                const double* m = sc.materials + 10 * (size_t)mat;
                const double reflectivity = m[7], glossy = m[8];
                if (!CHAIN || !(reflectivity > 0.0)) {  // material.rs:216: nothing is reflected
                    value = color; ended = true;
                } else if (depth + 1 > PT_MAX_DEPTH) {  // the reflected ray would be a depth-11 ray: its colour is the background (material.rs:102-104), not traced
                    if (STATS) cnt.depth11_skipped++;
                    value = color + pt_background(a, x, y) * reflectivity;  // material.rs:312-316
                    ended = true;
                } else {
                    const PtVec3 ray_dir = fr.l3(PT_L_D), N = fr.l3(PT_L_N);
                    PtVec3 reflect_dir = ray_dir - (N * 2.0) * pt_dot(ray_dir, N);  // material.rs:218
                    if (glossy > 0.0) {  // material.rs:221-239 (not renormalised: quirk Q5)
                        PtVec3 off = (fabs(reflect_dir.x) < PT_EPSILON && fabs(reflect_dir.y) < PT_EPSILON) ? reflect_dir + pt_v3(0.0, 0.1, 0.0) : reflect_dir + pt_v3(0.0, 0.0, 0.1);
                        PtVec3 u_basis = pt_cross(reflect_dir, off);
                        PtVec3 v_basis = pt_cross(reflect_dir, u_basis);
                        PtItemLane it;
                        uint32_t x2, y2;
                        pt_item_lane_fast(a, w, lane, &it, &x2, &y2);
                        const uint64_t key = pt_rng_key(a.seed, (uint64_t)y * a.width + x);
                        double u_coord = -glossy / 2.0 + pt_rng_draw(key, it.sample, draw) * glossy;
                        double v_coord = -glossy / 2.0 + pt_rng_draw(key, it.sample, draw + 1) * glossy;
                        draw += 2;
                        reflect_dir = reflect_dir + (u_basis * u_coord + v_basis * v_coord);
                    }
                    double* line = fr.spill + (size_t)depth * PT_SPILL_STRIDE;  // parked until the chain ends: colour so far, reflectivity
                    line[0] = color.x; line[1] = color.y; line[2] = color.z; line[3] = reflectivity;
                    ray.o = sray.o; ray.d = reflect_dir;
                    if (STATS) cnt.reflect++;
                }
            }
        }
        if (CHAIN && ended) {  // fold the parked colours back, innermost first (material.rs:312-316 at every level)
            for (int k = depth - 1; k >= 0; k--) {
                const double* line = fr.spill + (size_t)k * PT_SPILL_STRIDE;
                value = pt_v3(line[0], line[1], line[2]) + value * line[3];
            }
        }
        if (ended) { fr.set_l3(PT_L_VALUE, value); live = false; }
        }

        // render.rs:36-43 under the summation contract: the chunk's samples in ascending order, added by the lane of the chunk's
        // first sample from its neighbours' LDS columns (same wavefront: program order suffices).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            PtItemLane it;
            uint32_t x2, y2;
            const bool mine2 = pt_item_lane_fast(a, w, lane, &it, &x2, &y2);
            if (mine2 && it.first) {
                PtVec3 sum = fr.l3(PT_L_VALUE);
#ifdef PT_UNBOUNDED_CHUNK_LOOP  // (A/B only)
                for (uint32_t k = 1; k < it.count; k++) {
#else
                for (uint32_t k = 1; k < it.count && k < PT_SAMPLE_CHUNK; k++) {  // (bounded by a constant as well: a corrupted count must not spin - profiles/r04/notes.md)
#endif
                    const double* o = fr.lds + k;
                    sum = sum + pt_v3(o[(PT_L_VALUE + 0) * PT_FRAME_STRIDE], o[(PT_L_VALUE + 1) * PT_FRAME_STRIDE], o[(PT_L_VALUE + 2) * PT_FRAME_STRIDE]);
                }
                // pixel-major: where a wavefront is one pixel (SAMPLES = 64) the eight chunk-first lanes' 24-byte stores fill one 192-byte run instead of eight
                // lines 1,536 bytes apart (round 3: 0.97 GB HBM-side per big-scene frame for 0.45 GB of payload, the write granularity's doing)
#ifdef PT_ACCUM_CHUNK_MAJOR  // (A/B only: round 3's layout)
                double* o = a.accum + 3 * ((size_t)(it.slot >> 6) * a.n_chunks * 64 + (size_t)it.chunk * 64 + (it.slot & 63u));
#else
                double* o = a.accum + 3 * ((size_t)it.slot * a.n_chunks + it.chunk);
#endif
                o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef PT_CYCLES  // the whole item -> diag[2] (the part outside the walks = diag[2] - diag[0])
        if (STATS && lane == 0) cnt.diag[2] += __builtin_readcyclecounter() - cyc_item0;
#endif
#ifdef PT_TIMELINE
        { const unsigned long long d = wall_clock64() - tl_item; if (d > tl_item_max) tl_item_max = d; }
#endif
    }
#ifdef PT_TIMELINE
    if (STATS && lane == 0) {
        const unsigned long long tl_end = wall_clock64();
        cnt.diag[0] = tl_end - tl_start; cnt.diag[1] = 1;                       // summed: wavefront lifetimes, wavefronts
        atomicMax(&a0.counters->diag[2], tl_item_max);                            // longest item
        atomicMax(&a0.counters->diag[4], ~tl_start);                              // ~(earliest start)
        atomicMax(&a0.counters->diag[5], tl_end);                                 // latest end
        cnt.diag[3] = tl_first_item ? tl_first_item - tl_start : 0;               // summed: start -> first item in hand
        atomicMax(&a0.counters->diag[6], ~tl_end);                                // ~(earliest end): the tail is latest end - earliest end
        cnt.diag[2] = cnt.diag[4] = cnt.diag[5] = cnt.diag[6] = 0;
    }
#endif
    if (STATS) pt_flush_counters(a.counters, cnt);
}
