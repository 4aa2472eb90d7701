// Per-lane ray state machine: primary-ray generation, Blinn-Phong + shadow rays, and the
// reflection / refraction recursion of the reference turned into iteration.
//
// Replaces render_single_pixel (src/render.rs:22-51), Camera::ray_at (src/camera.rs:48-84),
// Ray::color (src/ray.rs:139-148) and Material::hit_color (src/material.rs:91-320).
//
// A lane owns ONE sample of one pixel at a time (pt_render_kernel maps the 64 lanes of a wavefront to
// 64 / K neighbouring pixels x K samples of the same pixel). It is a small interpreter: every call to
// pt_lane_advance() consumes the result of the ray the lane traced last and runs until it has the NEXT
// ray to trace (primary, shadow, reflected or refracted) - so all 64 lanes of a wavefront meet again in
// the traversal loop whatever kind of ray each one carries.
//
// Where the state lives:
//  * the hit being shaded (its point P, unit normal N, the incoming direction D, material tag, texel
//    colour) in LDS, one column of doubles per lane next to the traversal stack. It is written once when
//    the hit is rebuilt and read when its shadow rays come back: nothing of it touches HBM;
//  * the reference recurses - hit_color(depth) -> Ray::color(depth + 1) -> hit_color ... - and folds the
//    child's colour into the parent AFTER the child returns (`color += reflectivity * child`,
//    material.rs:243,280,307-309,315); a forward "throughput" accumulation would round differently. So a
//    hit that spawns a reflected / refracted child parks what its parent frame still needs in a per-lane
//    stack in HBM and the frames are folded back innermost-first. A parked frame is one 128-byte line of its
//    lane's own region (lanes of a wavefront push and pop at different times and depths once their paths have
//    diverged: a structure-of-arrays layout over lanes cost one line per SLOT then - measured on the
//    transmission-refraction scene, profiles/r02/notes.md):
//      slots 0-2 colour so far, 3 {material, frame stage}, 4-6 refracted direction (later: the reflected
//      colour), 7-9 hit point (origin of the refracted ray), 10 Schlick reflectance, 11 reflectivity.
//    Hits that spawn nothing (every hit of a scene without reflective materials) never write it.
//    The YOUNGEST parked frame of a lane stays in LDS next to the hit frame (PARK = 1 instantiations): a frame goes
//    to its HBM line only when a deeper hit parks too and needs the slot (frames [L.lo, L.depth) are in LDS, older
//    ones in HBM). In a recursion tree most frames are near the leaves - the children of most parked frames hit
//    something that spawns nothing - so many pops never leave the CU (profiles/r02/notes.md).
//  * a finished sample's colour goes to the lane's LDS column; the lane of the pixel's first sample adds
//    the chunk's samples in ascending order (the summation contract) and writes the chunk sum.
#pragma once

#include <stdlib.h>

#include "pt_pow.h"
#include "pt_trace.h"

#define PT_MAX_DEPTH 10  // material.rs:12
#define PT_SPILL_SLOTS 11
#define PT_SPILL_STRIDE 16  // doubles per parked frame: one 128-byte line, so that a push or pop of one lane touches one line
#define PT_SPILL_DEPTHS (PT_MAX_DEPTH + 1)  // frames at depth 0..9 can have a child in flight; depth 10 only parks a colour between rounds of > 32 lights
#define PT_PARK_F64 12                // a parked frame: the PT_H_* slots 0..11
#define PT_LDS_FRAME_F64 10           // P, N, D, tag (material index + the texel's three bytes when the diffuse colour is a texel)
// Samples of a pixel are summed in chunks of PT_SAMPLE_CHUNK (8): each chunk sequentially (ascending
// sample index), then the chunk sums sequentially (ascending chunk index). This is the build's
// summation contract (the reference's rayon reduce has no fixed association, render.rs:36-43); it
// lets a chunk's samples run side by side in the lanes of one wavefront.
#ifndef PT_SAMPLE_CHUNK
#define PT_SAMPLE_CHUNK 8
#endif
// A compiler-only fence: stops hipcc from hoisting the loads of one interpreter state above the
// stores of the previous one (register pressure), costs no instruction.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_STATE_FENCE)
#define PT_FENCE asm volatile("" ::: "memory")
#else
#define PT_FENCE
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PT_THREAD_IN_BLOCK() (threadIdx.x)
#else
#define PT_THREAD_IN_BLOCK() 0u
#endif
#define PT_QUEUE_STRIDE 32  // words between two queue counters: one 128-byte line each
#define PT_FINE_QUEUES 64
#ifndef PT_WORK_BATCH_MAX
#define PT_WORK_BATCH_MAX 8  // most work items (64 lanes each) a wavefront takes from the global counter at a time (PtRenderArgs::batch_max): 32 -> 8 costs big-scene 0.4 % and gives the 1.25 M-triangle scenes 4-7 % (profiles/r02/notes.md)
#endif

enum { PT_JITTER_CENTRE = 0, PT_JITTER_RNG = 1 };
enum { PT_ST_NEW_SAMPLE = 0, PT_ST_CLOSEST_DONE = 1, PT_ST_LIGHT = 2, PT_ST_SHADOW_DONE = 3, PT_ST_SHADE = 4, PT_ST_DONE = 5, PT_ST_WAIT_FORK = 6 };
enum { PT_FS_WAIT_REFLECT = 1, PT_FS_WAIT_REFRACT = 2, PT_FS_STAGE_MASK = 3, PT_FS_HAVE_REFRACT = 4,  // parked frames
       PT_FS_FORKED = 8,      // the refracted subtree was handed to another lane (FORK instantiations): its colour arrives in the frame's mailbox
       PT_FS_SEQ_SHIFT = 8 }; // bits 8..31: the lane's fork sequence number when the frame was parked (what the mailbox ticket is made of)
// Fork / join of the reflective kernel (north_star: secondary rays compacted with ballot / popcount and re-enqueued through an LDS
// queue). A dielectric hit spawns TWO subtrees, reflected and refracted (material.rs:216-303); one lane walks them one after the
// other while lanes whose samples are finished idle until the wavefront's item ends. In FORK instantiations such a hit OFFERS its
// refracted ray when it parks its frame; after every interpreter pass the wavefront matches offers and idle lanes by their ranks in
// the two ballots (popcount of the lower lanes), the k-th offering lane writes its id to slot k of a queue in LDS, the k-th idle
// lane reads it, takes the ray out of the owner's parked frame (LDS) and walks the subtree as a task rooted at that depth; its colour
// goes to the owner's MAILBOX (slots 12..15 of the owner's HBM line for that depth: colour + ticket) and the owner, when its reflected
// subtree is back, waits for the ticket (PT_ST_WAIT_FORK) and folds the two exactly as it would have (material.rs:305-309: same
// operands, same order - which lane computed an operand does not change its bits). Only scenes whose recursion draws no random
// numbers fork (no area light, no glossy material: the draw indices of the sampling contract follow the depth-first order).
// A finished sample's colour waits in slots 12..14 of the lane's depth-10 line (frames of depth 10 never spawn, so that mailbox is
// free): the lane's LDS frame is reused by the tasks it takes.
enum { PT_H_MAIL = 12, PT_H_MAIL_TICKET = 15, PT_RESULT_DEPTH = PT_MAX_DEPTH };
#define PT_PRE_MAPS 0x40000000u   // pt_lane_maps ran for this hit; PT_PRE_NORMAL: it left the normal map's shading normal in the frame's PT_L_N
#define PT_PRE_NORMAL 0x20000000u
#define PT_FS_TEXEL 0x80000000u  // hit frame: bits 0..23 are the texel's R, G, B bytes; the diffuse colour is srgb_lut[] of them (texture.rs:162-168)
#define PT_LIGHT_ROUND 32  // shadow-ray results are kept as one bit per light, 32 lights at a time

// true when PORTRAYER_INTERP=1 asks for the interpreter kernel where the straight-line kernel would run (builds with -DPT_KEEP_INTERP only)
inline bool pt_interpreter_forced() {
#ifdef PT_KEEP_INTERP
    const char* e = getenv("PORTRAYER_INTERP");
    return e && atoi(e) > 0;
#else
    return false;
#endif
}

// Division of a wave-uniform index by a launch constant without a divider: q = (n * magic) >> shift, exact for every n < 2^31
// (magic = ceil(2^shift / d), shift = 31 + ceil(log2 d): the error of magic / 2^shift is below 2^-31 / d ... n e < 2^shift).
// The scalar unit has multiply-high and shifts but no division: hipcc turns `w / n_groups` into a float reciprocal on the VECTOR
// unit, six of them per pt_item_lane call.
struct PtFastDiv { uint32_t d, magic, shift; };
PT_HD uint32_t pt_fastdiv(uint32_t n, const PtFastDiv& f) { return (uint32_t)(((uint64_t)n * f.magic) >> f.shift); }
inline PtFastDiv pt_fastdiv_make(uint32_t d) {
    PtFastDiv f;
    f.d = d < 1u ? 1u : d;
    uint32_t l = 0;
    while ((1ull << l) < f.d) l++;
    f.shift = 31u + l;
    f.magic = (uint32_t)(((1ull << f.shift) + f.d - 1u) / f.d);
    return f;
}

enum { PT_RUN_INTERP = 0, PT_RUN_INTERP_PARK = 1, PT_RUN_LINE4 = 2, PT_RUN_LINE3 = 3, PT_RUN_INTERP4 = 4, PT_RUN_INTERP_FORK = 5, PT_RUN_LINE5 = 6, PT_RUN_CHAIN = 7 };  // PtRenderArgs::run_variant, explained in pt_render_kernel.h

#ifndef PT_LINE_TOP_WAVES
// Waves per SIMD of the mesh-free straight-line kernels' densest instantiation (flat_scene / hierarchical semantics): 6 since round 4 (80 registers, 26 KB of
// LDS a block; big-scene +3.1 %, hierarchical +2.3 %, 3840x2160x256 +3.3 %: profiles/r04/notes.md section 5 - round 3 measured the same gain and could not ship
// it because that build hung: section 2). -DPT_LINE_TOP_WAVES=5 builds round 3's.
#define PT_LINE_TOP_WAVES 6
#endif
#ifndef PT_MESH_TOP_WAVES
#define PT_MESH_TOP_WAVES 5   // ... and of scenes of very many triangles in plain Mesh instances (6 measured: profiles/r04/notes.md section 7)
#endif
struct PtRenderArgs {
    PtSceneView scene;
    PtCamera cam;
    const double* background;  // H x 3 (rows) or H x W x 3
    int32_t background_rows;
    uint32_t width, height;
    uint32_t x0, y0, x1, y1;   // inclusive slice (render.rs:115-138)
    uint32_t samples;
    uint64_t seed;
    int32_t jitter_mode;
    uint32_t tile_rank, tile_ranks;  // this launch renders 8x8 tiles t with t % tile_ranks == tile_rank
    uint32_t n_slots;                // pixel slots of this launch (own tiles x 64)
    uint32_t n_chunks;               // sample chunks per pixel: ceil(samples / PT_SAMPLE_CHUNK)
    uint32_t lane_samples;           // K = samples of a chunk that run side by side in a wavefront: 8, or the next power of two >= samples
    uint32_t lane_chunks;            // C = chunks of a pixel that run side by side (1, 2, 4 or 8); a wavefront covers 64 / (K C) pixels
    uint32_t n_items;                // wavefront work items of this launch = own tiles x ceil(n_chunks / C) x (K C)
    uint32_t k_log2, c_log2;         // K and C are powers of two: their logarithms, for pt_item_lane_fast
    PtFastDiv div_groups;            // by ceil(n_chunks / C): item -> (local tile, chunk group)
    PtFastDiv div_tiles_x;           // by the slice's tiles per row: tile -> (row, column)
    double* accum;                   // n_slots x n_chunks x 3: per (pixel slot, chunk) the sum of the chunk's samples
    int32_t compact;                 // 1: rgb is tile-major over own tiles; 0: rgb is the full H x W x 3 image
    uint8_t* rgb;
    double* linear;                  // optional, same indexing as rgb
    double* spill;                   // recursion frames, lane x PT_SPILL_DEPTHS x PT_SPILL_STRIDE
    uint32_t n_lanes;
    uint32_t* stack_spill;           // traversal-stack entries beyond the LDS part, entry x n_lanes
    int32_t stack_lds_cap;           // entries per lane kept in LDS; the rest (up to scene.stack_cap) in stack_spill
    uint32_t grid_share;             // host side only: the launch gets 1 / grid_share of the device's resident blocks (pt_node: ranks that share a GPU; 0 or 1 = all of them)
    int32_t park_slots;              // parked recursion frames per lane kept in LDS (0 or 1; selects the PARK instantiation); older ones in `spill`
    int32_t four_waves;              // the instantiation compiled for more than 3 waves per SIMD (scenes without reflective materials only): 0, or 4 / 5 = the waves
    int32_t run_variant;             // PT_RUN_* (pt_render_kernel.h): which kernel pt_render_common launches
    uint32_t launch_nonce;           // differs from launch to launch of a context (24 bits): mailbox tickets of an earlier launch never match
    unsigned int* work_counter;
    unsigned int* overflow_flag;     // set to 1 by any lane that runs out of traversal stack
    uint32_t work_div;               // a wavefront takes (remaining items / work_div) items from work_counter at a time
    uint32_t batch_max;              // most items a wavefront takes at a time
    uint32_t fine_queues;            // 0: batches from work_counter. N > 0: one item at a time from N interleaved queues (work_queues), see pt_render_kernel
    unsigned int* work_queues;       // N counters, PT_QUEUE_STRIDE words apart
    uint32_t item_stride;            // hand-out position q -> item (q * item_stride) mod n_items; 1 = in image order
    PtCounters* counters;
};

struct PtLane {
    uint32_t item;     // the wavefront's work item (wave-uniform)
    uint32_t x, y;     // its pixel; its sample index is worked out from `item` where it is needed (pt_lane_sample)
    uint32_t stage, light, draw, draw0, occluded;
    int32_t depth;
    int32_t lo;        // parked frames of depth [lo, depth) are in LDS, those of [0, lo) in HBM
    PtRay ray;
    bool has_ray, ray_any;
    // FORK instantiations only
    bool offer;          // the frame parked in this pass has a refracted ray another lane may take
    int32_t base;        // depth the lane's current task is rooted at: 0 = its own sample, d + 1 = a refracted subtree of a depth-d hit
    uint32_t owner;      // a taken task's owner: its thread index in the block | the frame's depth << 16
    uint32_t fork_seq;   // frames this lane has offered in this launch
    uint64_t ticket;     // a taken task: what goes into the owner's mailbox with the colour
    uint64_t wait_ticket;  // PT_ST_WAIT_FORK: what the lane waits for in its own mailbox (a lane walking a taken task may itself have forked)
};

// This lane's view of its two frame stores.
struct PtFrameRef {
    double* lds;       // column in the block's LDS frame area: slot s at lds[s * PT_FRAME_STRIDE]
    double* spill;     // column in the HBM recursion stack
    double* park;      // column in the block's LDS area for parked frames: frame k, slot s at park[(k * PT_PARK_F64 + s) * PT_FRAME_STRIDE]
    uint32_t n_lanes;
    PT_HD double& l(int slot) const { return lds[slot * PT_FRAME_STRIDE]; }
    PT_HD PtVec3 l3(int slot) const { return pt_v3(l(slot), l(slot + 1), l(slot + 2)); }
    PT_HD void set_l3(int slot, PtVec3 v) const { l(slot) = v.x; l(slot + 1) = v.y; l(slot + 2) = v.z; }
    PT_HD double& h(int depth, int slot) const { return spill[depth * PT_SPILL_STRIDE + slot]; }
    PT_HD PtVec3 h3(int depth, int slot) const { return pt_v3(h(depth, slot), h(depth, slot + 1), h(depth, slot + 2)); }
    PT_HD void set_h3(int depth, int slot, PtVec3 v) const { h(depth, slot) = v.x; h(depth, slot + 1) = v.y; h(depth, slot + 2) = v.z; }
    PT_HD double& p(int k, int slot) const { return park[(k * PT_PARK_F64 + slot) * PT_FRAME_STRIDE]; }
    // A parked frame as 12 doubles (the PT_H_* slots), wherever it lives. The HBM line is 128-byte aligned: 16-byte accesses.
    PT_HD void load_hbm(int depth, double* f) const {
        const double2* q = reinterpret_cast<const double2*>(spill + depth * PT_SPILL_STRIDE);
#pragma unroll
        for (int i = 0; i < PT_PARK_F64 / 2; i++) { double2 v = q[i]; f[2 * i] = v.x; f[2 * i + 1] = v.y; }
    }
    PT_HD void store_hbm(int depth, const double* f) const {
        double2* q = reinterpret_cast<double2*>(spill + depth * PT_SPILL_STRIDE);
#pragma unroll
        for (int i = 0; i < PT_PARK_F64 / 2; i++) { double2 v; v.x = f[2 * i]; v.y = f[2 * i + 1]; q[i] = v; }
    }
    PT_HD void load_lds(int k, double* f) const {
#pragma unroll
        for (int i = 0; i < PT_PARK_F64; i++) f[i] = p(k, i);
    }
    PT_HD void store_lds(int k, const double* f) const {
#pragma unroll
        for (int i = 0; i < PT_PARK_F64; i++) p(k, i) = f[i];
    }
    static PT_HD double pack_tag(uint32_t mat, uint32_t stage) { union { double d; uint32_t u[2]; } c; c.u[0] = mat; c.u[1] = stage; return c.d; }
    static PT_HD void unpack_tag(double t, uint32_t* mat, uint32_t* stage) { union { double d; uint32_t u[2]; } c; c.d = t; *mat = c.u[0]; *stage = c.u[1]; }
};
// LDS frame slots
enum { PT_L_P = 0, PT_L_N = 3, PT_L_D = 6, PT_L_TAG = 9, PT_L_VALUE = 0 /* a finished sample's colour reuses P */ };
// HBM spill slots
enum { PT_H_COLOR = 0, PT_H_TAG = 3, PT_H_DIR = 4, PT_H_P = 7, PT_H_SCHLICK = 10, PT_H_REFL = 11 /* the material's reflectivity: the pop needs no material fetch */ };

// 8x8 tiles over the slice rectangle, row-major over tiles; pixel slot p = local tile * 64 + j.
PT_HD bool pt_slot_to_pixel(const PtRenderArgs& a, uint32_t p, uint32_t* x, uint32_t* y) {
    uint32_t rw = a.x1 - a.x0 + 1;
    uint32_t tiles_x = (rw + 7) / 8;
    uint32_t tile = (p >> 6) * a.tile_ranks + a.tile_rank;
    uint32_t j = p & 63;
    uint32_t px = a.x0 + (tile % tiles_x) * 8 + (j & 7);
    uint32_t py = a.y0 + (tile / tiles_x) * 8 + (j >> 3);
    *x = px; *y = py;
    return px <= a.x1 && py <= a.y1;
}
// Wavefront work item: the 64 lanes are P neighbouring pixels of an 8x8 tile x C consecutive 8-sample chunks x K samples
// of a chunk (P C K = 64; K = lane_samples, C = lane_chunks), a chunk's samples in neighbouring lanes. With SAMPLES = 64
// that is ONE pixel's 64 samples (P = 1, C = 8, K = 8): every ray of the wavefront passes through the same pixel, walks the
// same part of the trees and takes about the same number of steps - which is what keeps the lanes of a wavefront busy
// inside the walk (profiles/r02/notes.md). Item w = ((local tile * n_groups + chunk group) * (64 / P)) + part.
// Pixels of a tile are taken in Morton order, so that P consecutive ones form a compact window (2x1, 2x2, 4x2, 4x4 ...).
PT_HD uint32_t pt_tile_order_to_slot(uint32_t q) {  // -> j = y * 8 + x inside the 8x8 tile
    uint32_t x = (q & 1u) | ((q >> 1) & 2u) | ((q >> 2) & 4u);
    uint32_t y = ((q >> 1) & 1u) | ((q >> 2) & 2u) | ((q >> 3) & 4u);
    return y * 8u + x;
}
struct PtItemLane {
    uint32_t slot;    // pixel slot of this launch (local tile * 64 + j)
    uint32_t chunk;
    uint32_t sample;  // absolute sample index
    uint32_t first;   // 1: this lane holds the first sample of its pixel's chunk (it sums the chunk)
    uint32_t count;   // samples of this chunk (<= K)
};
PT_HD bool pt_item_lane(const PtRenderArgs& a, uint32_t w, uint32_t lane, PtItemLane* it, uint32_t* x, uint32_t* y) {
    const uint32_t K = a.lane_samples, C = a.lane_chunks, P = 64u / (K * C), parts = 64u / P;
    const uint32_t n_groups = (a.n_chunks + C - 1u) / C;
    uint32_t part = w % parts, g = w / parts;
    uint32_t group = g % n_groups, tile_local = g / n_groups;
    uint32_t si = lane % K, ci = (lane / K) % C, pi = lane / (K * C);
    uint32_t j = pt_tile_order_to_slot(part * P + pi);
    uint32_t chunk = group * C + ci;
    it->slot = (tile_local << 6) | j;
    it->chunk = chunk;
    it->sample = chunk * PT_SAMPLE_CHUNK + si;
    it->first = si == 0u;
    uint32_t left = chunk < a.n_chunks ? a.samples - chunk * PT_SAMPLE_CHUNK : 0u;
    it->count = left < (uint32_t)PT_SAMPLE_CHUNK ? left : (uint32_t)PT_SAMPLE_CHUNK;
    return pt_slot_to_pixel(a, it->slot, x, y) && si < it->count;
}

// The same, the way the render kernels compute it: shifts and masks for the lane part (K, C and so P are powers of two), two
// multiply-high divisions on the scalar unit for the item part (pt_fill_work prepares them). pt_test_work_items replays every item
// of a launch through both and fails if they ever differ.
PT_HD bool pt_item_lane_fast(const PtRenderArgs& a, uint32_t w, uint32_t lane, PtItemLane* it, uint32_t* x, uint32_t* y) {
    const uint32_t kl = a.k_log2, cl = a.c_log2, parts_l = kl + cl, pl = 6u - parts_l;
    const uint32_t part = w & ((1u << parts_l) - 1u), g = w >> parts_l;
    const uint32_t tile_local = pt_fastdiv(g, a.div_groups), group = g - tile_local * a.div_groups.d;
    const uint32_t si = lane & ((1u << kl) - 1u), ci = (lane >> kl) & ((1u << cl) - 1u), pi = lane >> parts_l;
    const uint32_t j = pt_tile_order_to_slot((part << pl) + pi);
    const uint32_t chunk = (group << cl) + ci;
    it->slot = (tile_local << 6) | j;
    it->chunk = chunk;
    it->sample = chunk * PT_SAMPLE_CHUNK + si;
    it->first = si == 0u;
    const uint32_t left = chunk < a.n_chunks ? a.samples - chunk * PT_SAMPLE_CHUNK : 0u;
    it->count = left < (uint32_t)PT_SAMPLE_CHUNK ? left : (uint32_t)PT_SAMPLE_CHUNK;
    const uint32_t tile = tile_local * a.tile_ranks + a.tile_rank;
    const uint32_t ty = pt_fastdiv(tile, a.div_tiles_x), tx = tile - ty * a.div_tiles_x.d;
    const uint32_t px = a.x0 + tx * 8u + (j & 7u), py = a.y0 + ty * 8u + (j >> 3);
    *x = px; *y = py;
    return px <= a.x1 && py <= a.y1 && si < it->count;
}

// The lane's sample index, worked out from the wave-uniform item index where it is needed (the jitter draws at the start of
// the sample, area-light and glossy draws) instead of being carried through every tree walk in a register that hipcc then
// spills at the start of every item (measured: 2.3 -> 0.7 GB of scratch writes per 1920x1080x64 frame, +0.5-2 % on most
// workloads; doing the same for x and y cost big-scene 3 %). The empty asm stops the compiler from hoisting the computation
// out of the sample loop and keeping the result live after all.
PT_HD uint32_t pt_lane_sample(const PtRenderArgs& a, uint32_t item) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(item));
    const uint32_t lane = threadIdx.x & 63u;
#else
    const uint32_t lane = 0;
#endif
    PtItemLane it;
    uint32_t x, y;
    pt_item_lane_fast(a, item, lane, &it, &x, &y);
    return it.sample;
}

// -DPT_XY_ON_DEMAND: the pixel coordinates too (experiment; see profiles/r02/notes.md)
PT_HD void pt_lane_xy(const PtRenderArgs& a, uint32_t item, uint32_t* x, uint32_t* y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(item));
    const uint32_t lane = threadIdx.x & 63u;
#else
    const uint32_t lane = 0;
#endif
    PtItemLane it;
    pt_item_lane_fast(a, item, lane, &it, x, y);
}
#ifdef PT_XY_ON_DEMAND
#define PT_LANE_XY(a, L, X, Y) uint32_t X, Y; pt_lane_xy(a, (L).item, &X, &Y)
#else
#define PT_LANE_XY(a, L, X, Y) const uint32_t X = (L).x, Y = (L).y
#endif

PT_HD PtVec3 pt_background(const PtRenderArgs& a, uint32_t x, uint32_t y) {  // render.rs:31-34
    const double* b = a.background_rows ? a.background + 3 * (size_t)y : a.background + 3 * ((size_t)y * a.width + x);
    return pt_v3(b[0], b[1], b[2]);
}

PT_HD PtRay pt_camera_ray(const PtCamera& c, double x, double y) {  // camera.rs:48-84
    double ndc_y = y / c.height;
    double view_y = (1.0 - 2.0 * ndc_y) * c.fov_factor;
    double ndc_x = x / c.width;
    double view_x = (2.0 * ndc_x - 1.0) * c.aspect * c.fov_factor;
    PtVec3 eye = pt_v3(c.eye[0], c.eye[1], c.eye[2]);
    PtVec3 world = pt_xform_point(c.view_to_world, pt_v3(view_x, view_y, -1.0));
    PtRay r;
    r.o = eye;
    r.d = pt_normalized(world - eye);
    return r;
}

// pow is the only libm call on the path (gamma render.rs:47, specular material.rs:200): glibc's pow, restated operation by
// operation (pt_pow.h), so that linear colours carry the reference's bits. -DPT_POW_OCML keeps the device library's pow (within
// 1 ulp of glibc's). Out of line: one shared copy lets the register allocator size everything else for a higher occupancy.
#ifdef PT_POW_OCML
#define PT_POW_IMPL(x, y) pow(x, y)
#else
#define PT_POW_IMPL(x, y) pt_pow_glibc(x, y)
#endif
#ifdef PT_POW_INLINE
PT_HD double pt_pow(double x, double y) { return PT_POW_IMPL(x, y); }
#else
PT_NOINLINE double pt_pow(double x, double y) { return PT_POW_IMPL(x, y); }
#endif

PT_HD uint8_t pt_to_u8(double c) {  // render.rs:143-147: `as u8` saturates, NaN -> 0
    double v = c * 255.0;
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

PT_HD double pt_powi5(double x) {  // f64::powi(5): x * ((x*x) * (x*x))
    double x2 = x * x, x4 = x2 * x2;
    return x * x4;
}

// material.rs:27-48
PT_HD bool pt_refracted_direction(PtVec3 ray_dir, PtVec3 normal, double eta, PtVec3* out) {
    const double eta_outside = 1.00;
    double c = pt_dot(ray_dir, normal);
    double under_sqrt = 1.0 - eta_outside * eta_outside * (1.0 - c * c) / (eta * eta);
    if (under_sqrt < 0.0) return false;
    PtVec3 d1 = ((ray_dir - normal * c) * eta_outside) / eta;
    PtVec3 d2 = normal * sqrt(under_sqrt);
    *out = d1 - d2;
    return true;
}

// A pixel's chunk sums added up in ascending order (the summation contract). p = pixel slot.
PT_HD PtVec3 pt_pixel_sum(const PtRenderArgs& a, uint32_t p) {
#ifdef PT_ACCUM_CHUNK_MAJOR
    const double* acc = a.accum + 3 * ((size_t)(p >> 6) * a.n_chunks * 64 + (p & 63));
    PtVec3 sum = pt_v3(acc[0], acc[1], acc[2]);
    for (uint32_t k = 1; k < a.n_chunks; k++) sum = sum + pt_v3(acc[192 * (size_t)k], acc[192 * (size_t)k + 1], acc[192 * (size_t)k + 2]);
#else
    const double* acc = a.accum + 3 * (size_t)p * a.n_chunks;  // the pixel's chunk sums, side by side
    PtVec3 sum = pt_v3(acc[0], acc[1], acc[2]);
    for (uint32_t k = 1; k < a.n_chunks; k++) sum = sum + pt_v3(acc[3 * (size_t)k], acc[3 * (size_t)k + 1], acc[3 * (size_t)k + 2]);
#endif
    return sum;
}
// One pixel's sum -> mean, gamma, clamp, u8 (render.rs:45-50, :143-147). p = pixel slot.
PT_HD void pt_finish_pixel(const PtRenderArgs& a, uint32_t p, PtVec3 sum) {
    uint32_t x, y;
    if (!pt_slot_to_pixel(a, p, &x, &y)) return;
    PtVec3 color = sum / (double)a.samples;
    size_t idx = a.compact ? (size_t)p : (size_t)y * a.width + x;
    if (a.linear) { double* o = a.linear + 3 * idx; o[0] = color.x; o[1] = color.y; o[2] = color.z; }
    const double g = 1.0 / PT_GAMMA;
    double ch[3] = {pt_pow(color.x, g), pt_pow(color.y, g), pt_pow(color.z, g)};
    uint8_t* o = a.rgb + 3 * idx;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v = ch[k];
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        o[k] = pt_to_u8(v);
    }
}

// Position of light `li` as sampled for the current hit (light.rs:62-70, :87-90). `draw0` is the
// index of the first of the two draws an area light consumes; the generator is counter-based, so
// re-evaluating with the same indices after the shadow ray returns yields the same point.
PT_HD PtVec3 pt_light_position(const PtRenderArgs& a, const PtLane& L, const double* light, uint32_t draw0, bool* is_area) {
    PtVec3 pos = pt_v3(light[0], light[1], light[2]);
    PtVec3 aa = pt_v3(light[9], light[10], light[11]), ab = pt_v3(light[12], light[13], light[14]);
    bool empty = (aa.x == 0.0 && aa.y == 0.0 && aa.z == 0.0) || (ab.x == 0.0 && ab.y == 0.0 && ab.z == 0.0);  // light.rs:51-53
    *is_area = !empty;
    if (empty) return pos;
    const uint32_t sample = pt_lane_sample(a, L.item);
    PT_LANE_XY(a, L, lx, ly);
    uint64_t pixel = (uint64_t)ly * a.width + lx;
    double a_coord = 2.0 * pt_rng_f64(a.seed, pixel, sample, draw0) - 1.0;
    double b_coord = 2.0 * pt_rng_f64(a.seed, pixel, sample, draw0 + 1) - 1.0;
    return pos + (aa * a_coord + ab * b_coord);
}

// ------------------------------------------------------------------------------------------------
// Textures and normal maps (src/texture.rs, consumed at material.rs:109-144). Out of line and entered
// only for materials that carry a map, so the common untextured path keeps its register budget.
// ------------------------------------------------------------------------------------------------
PT_HD PtVec3 pt_texel(const PtTexInfo* tex, const uint8_t* rgb, int32_t id, double u, double v) {  // texture.rs:96-141
    long long width = tex[id].width, height = tex[id].height;
    double fx = u * (double)(width - 1), fy = v * (double)(height - 1);
    // Rust `as i64`: truncation toward zero, saturating, NaN -> 0
    long long x = fx != fx ? 0 : (fx >= 9.2233720368547758e18 ? 0x7fffffffffffffffLL : (fx <= -9.2233720368547758e18 ? (-0x7fffffffffffffffLL - 1) : (long long)fx));
    long long y = fy != fy ? 0 : (fy >= 9.2233720368547758e18 ? 0x7fffffffffffffffLL : (fy <= -9.2233720368547758e18 ? (-0x7fffffffffffffffLL - 1) : (long long)fy));
    x %= width; if (x < 0) x += width;  // rem_euclid
    y %= height; if (y < 0) y += height;
    const uint8_t* px = rgb + tex[id].offset + 3 * ((size_t)y * (size_t)width + (size_t)x);
    return pt_v3((double)px[0], (double)px[1], (double)px[2]);  // 0..255, not yet divided
}

struct PtMat3 {
    PtVec3 c0, c1, c2;  // columns (Mat3::from_col_arrays)
};
PT_HD PtVec3 pt_mat3_mul(const PtMat3& m, PtVec3 v) {  // row dot products, left to right
    return pt_v3((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z, (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z, (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z);
}

// type / sub / local / t identify the hit (as in PT_ST_CLOSEST_DONE); p, n = its model-space point and
// raw normal. Returns the diffuse colour (texel or *kd unchanged) and, when the material has a normal
// map, the shading normal (NOT multiplied by the node's normal_trans: quirk Q12 of the reference).
// Arguments and result travel in registers (29 and 13 VGPRs): a longer list or pointers to locals go through scratch memory,
// which on a scene with 196,608 resident lanes is HBM traffic (measured on transmission-refraction: half of the launch's
// HBM-side bytes, profiles/r02/notes.md).
struct PtMapsOut {
    double n[3];
    int32_t has_n;
    uint32_t texel;  // PT_FS_TEXEL | R | G << 8 | B << 16 when the material has a texture, else 0
};
#ifdef PT_MAPS_INLINE
#define PT_MAPS_ATTR PT_HD
#else
#define PT_MAPS_ATTR PT_NOINLINE
#endif
PT_HD PtMapsOut pt_apply_maps_body(const PtTexView* view, uint32_t mat, uint32_t type, uint32_t sub, double ox, double oy, double oz, double dx, double dy,
                                   double dz, double px, double py, double pz, double nx, double ny, double nz) {
    const PtTexInfo* tex = view->tex;
    const uint8_t* tex_rgb = view->tex_rgb;
    const double* uv_trans9 = view->uv_trans + 9 * (size_t)mat;
    const double* tri_v = view->tri_v;
    const double* tri_uv = view->tri_uv;
    const int32_t tex_id = view->mat_maps[2 * mat], nmap_id = view->mat_maps[2 * mat + 1];
    PtMapsOut out;
    out.texel = 0;
    out.n[0] = out.n[1] = out.n[2] = 0.0;
    PtVec3 p = pt_v3(px, py, pz), n = pt_v3(nx, ny, nz);
    double u = 0.0, v = 0.0;
    PtMat3 tbn;
    tbn.c0 = pt_v3(1.0, 0.0, 0.0); tbn.c1 = pt_v3(0.0, 1.0, 0.0); tbn.c2 = pt_v3(0.0, 0.0, 1.0);
    if (type == PT_SPHERE || type == PT_CUBE) {
        PtVec3 fn = n;  // sphere: normal = hit point; cube: the face normal
        if (type == PT_SPHERE) {  // sphere.rs:53-61
            const double PI = 3.14159265358979323846;
            u = (PI + atan2(-p.z, p.x)) / (2.0 * PI);
            v = acos(p.y) / PI;
        } else {  // cube.rs:46-66, :84-112
            const double ax_u[6] = {-1, 1, 1, 1, 1, -1}, ax_v[6] = {1, 1, -1, 1, 1, 1};
            const double off_u[6] = {1.0 / 2.0, 0.0, 1.0 / 4.0, 1.0 / 4.0, 1.0 / 4.0, 3.0 / 4.0};
            const double off_v[6] = {1.0 / 3.0, 1.0 / 3.0, 0.0, 2.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0};
            double fu, fv;
            if (fn.x != 0.0) { fu = p.z; fv = p.y; } else if (fn.y != 0.0) { fu = p.x; fv = p.z; } else { fu = p.x; fv = p.y; }
            double nu = fu * ax_u[sub] + 0.5, nv = 0.5 - fv * ax_v[sub];
            u = nu / 4.0 + off_u[sub];
            v = nv / 3.0 + off_v[sub];
        }
        // sphere.rs:75-96 / cube.rs:114-137
        PtVec3 to_top = pt_normalized(pt_v3(0.0, 1.0, 0.0) - p);
        if (fabs(to_top.x) < PT_EPSILON && fabs(to_top.z) < PT_EPSILON) {
            tbn.c0 = pt_v3(1.0, 0.0, 0.0); tbn.c1 = fn; tbn.c2 = fn.y > 0.0 ? pt_v3(0.0, 0.0, 1.0) : pt_v3(0.0, 0.0, -1.0);
        } else {
            PtVec3 ht = pt_cross(to_top, fn);
            tbn.c0 = ht; tbn.c1 = fn; tbn.c2 = pt_cross(fn, ht);
        }
    } else if (type == PT_PLANE) {  // plane.rs:40-46
        u = p.x + 0.5; v = p.z + 0.5;
    } else {  // triangle with texture coordinates: triangle.rs:90-138
        const double* tv = tri_v + 9 * (size_t)sub;
        const double* q = tri_uv + 6 * (size_t)sub;
        PtRay local; local.o = pt_v3(ox, oy, oz); local.d = pt_v3(dx, dy, dz);
        double t2, beta, gamma;
        pt_triangle_hit(tv, local, -INFINITY, INFINITY, &t2, &beta, &gamma);
        double alpha = 1.0 - beta - gamma;
        double uu = (q[0] * alpha + q[2] * beta) + q[4] * gamma, vv = (q[1] * alpha + q[3] * beta) + q[5] * gamma;
        u = uu; v = 1.0 - vv;
        PtVec3 A = pt_v3(tv[0], tv[1], tv[2]), B = pt_v3(tv[3], tv[4], tv[5]), C = pt_v3(tv[6], tv[7], tv[8]);
        PtVec3 e1 = B - A, e2 = C - A;
        double du1 = q[2] - q[0], dv1 = q[3] - q[1], du2 = q[4] - q[0], dv2 = q[5] - q[1];
        PtVec3 tangent = pt_v3(dv2 * e1.x - dv1 * e2.x, dv2 * e1.y - dv1 * e2.y, dv2 * e1.z - dv1 * e2.z);
        PtVec3 bitangent = pt_v3(-du2 * e1.x + du1 * e2.x, -du2 * e1.y + du1 * e2.y, -du2 * e1.z + du1 * e2.z);
        double coeff = du1 * dv2 - du2 * dv1;
        tbn.c0 = pt_normalized(tangent / coeff); tbn.c1 = pt_normalized(n); tbn.c2 = pt_normalized(bitangent / coeff);
    }
    // material.rs:113-117: uv_trans * (u, v, 1)
    double tu = (uv_trans9[0] * u + uv_trans9[1] * v) + uv_trans9[2] * 1.0;
    double tv2 = (uv_trans9[3] * u + uv_trans9[4] * v) + uv_trans9[5] * 1.0;
    out.has_n = 0;
    if (nmap_id >= 0) {  // texture.rs:192-221 + material.rs:126-132
        PtVec3 c = pt_texel(tex, tex_rgb, nmap_id, tu, tv2) / 255.0;
        PtVec3 norm = pt_v3(2.0 * c.x - 1.0, 2.0 * c.y - 1.0, -(2.0 * c.z - 1.0));
        PtVec3 tex_norm = pt_v3(norm.x, -norm.z, -norm.y);  // normal_to_rh * norm (a signed permutation: exact)
        PtVec3 shading = pt_mat3_mul(tbn, pt_normalized(tex_norm));
        out.n[0] = shading.x; out.n[1] = shading.y; out.n[2] = shading.z;
        out.has_n = 1;
    }
    if (tex_id >= 0) {  // texture.rs:162-168 through the host-built (k / 255)^2.2 table
        PtVec3 c = pt_texel(tex, tex_rgb, tex_id, tu, tv2);
        out.texel = PT_FS_TEXEL | (uint32_t)(int)c.x | ((uint32_t)(int)c.y << 8) | ((uint32_t)(int)c.z << 16);
    }
    return out;
}
PT_MAPS_ATTR PtMapsOut pt_apply_maps(const PtTexView* view, uint32_t mat, uint32_t type, uint32_t sub, double ox, double oy, double oz, double dx, double dy,
                                    double dz, double px, double py, double pz, double nx, double ny, double nz) {
    return pt_apply_maps_body(view, mat, type, sub, ox, oy, oz, dx, dy, dz, px, py, pz, nx, ny, nz);
}
// what pt_lane_maps (the interpreter's map stage in front of its state machine) calls: the out-of-line routine, or the body in place (PT_LANE_MAPS_INLINE)
#if defined(PT_LANE_MAPS_INLINE) || defined(PT_MAPS_INLINE)
#define PT_LANE_APPLY_MAPS pt_apply_maps_body
#else
#define PT_LANE_APPLY_MAPS pt_apply_maps
#endif

// flat_scene.rs:85-95 / scene.rs:100-112 + material.rs:109-144 for the winning candidate of a ray: the model-space hit is rebuilt
// with the reference's expressions, point and normal go to world space (HIER: level by level), the material's maps are applied.
// true when an identity transform leaves every component of v as it is, bit for bit: none is -0 (the sums would make it +0, or keep it, depending on the
// other components' signs) and none is infinite or NaN (0 x infinity)
PT_HD bool pt_identity_safe(PtVec3 v) {
    auto odd = [](double c) {
        union { double d; uint32_t u[2]; } b; b.d = c;
        return (b.u[1] == 0x80000000u && b.u[0] == 0u) || (b.u[1] & 0x7FF00000u) == 0x7FF00000u;
    };
    return !(odd(v.x) || odd(v.y) || odd(v.z));
}

// The hit in the node's model space: its point p and raw normal n, the ray there, the node's type and material.
template <bool HIER>
PT_HD void pt_hit_model(const PtSceneView& sc, const PtRay& ray, const PtHit& hit, uint32_t* type_out, uint32_t* mat_out, PtRay* local_out, PtVec3* p_out, PtVec3* n_out) {
    const uint32_t* info = sc.info + 4 * (size_t)hit.node;
    uint32_t type = info[0], flags = info[2];
    PtRay local = pt_node_local_ray<HIER>(sc, hit.node, ray);
    PtVec3 p, n;
    if (type == PT_TRIANGLE || type == PT_MESH || type == PT_KDMESH) {
        const double* v = sc.tri_v + 9 * (size_t)hit.sub;
        p = pt_ray_at(local, hit.t);
        bool smooth = (flags & 1u) != 0;
        if (smooth) {  // triangle.rs:82-86: re-derive beta / gamma with the same arithmetic
            double t2, beta, gamma;
            pt_triangle_hit(v, local, -INFINITY, INFINITY, &t2, &beta, &gamma);
            double alpha = 1.0 - beta - gamma;
            const double* vn = sc.tri_n + 9 * (size_t)hit.sub;
            n = (pt_v3(vn[0], vn[1], vn[2]) * alpha + pt_v3(vn[3], vn[4], vn[5]) * beta) + pt_v3(vn[6], vn[7], vn[8]) * gamma;
        } else {       // triangle.rs:87: (b - a) x (c - a)
            PtVec3 A = pt_v3(v[0], v[1], v[2]), B = pt_v3(v[3], v[4], v[5]), C = pt_v3(v[6], v[7], v[8]);
            n = pt_cross(B - A, C - A);
        }
    } else {
        pt_prim_surface(type, hit.sub, local, hit.t, &p, &n);
    }
    *type_out = type; *mat_out = info[3]; *local_out = local; *p_out = p; *n_out = n;
}

// Out: world-space point P, shading normal N (normalised geometric normal, or the normal map's), material index, texel tag.
// MAPS_LATER (the interpreter kernel): a material's texture / normal map is NOT applied here - pt_lane_maps has done it for the
// lane before the state machine was entered (the map code then sits outside the state machine, where the registers it needs
// are spilled around it and not around every pass: profiles/r04/notes.md section 6). Returns true when the material has a map.
// SKIP_IDENT: identity levels of a hierarchical path are skipped on the way up (the interpreter kernel: +2 % on the dielectric scenes; the straight-line kernels are
// as fast or faster applying every level, c51).
template <bool TEX, bool HIER, bool MAPS_LATER = false, bool SKIP_IDENT = false>
PT_HD bool pt_hit_surface(const PtSceneView& sc, const PtRay& ray, const PtHit& hit, PtVec3* P_out, PtVec3* N_out, uint32_t* mat_out, uint32_t* ftag_out) {
    uint32_t type, mat;
    PtRay local;
    PtVec3 p, n;
    pt_hit_model<HIER>(sc, ray, hit, &type, &mat, &local, &p, &n);
    PT_FENCE;
    PtVec3 P, Nw;
    if (HIER) {  // scene.rs:100-101, :111-112: every level on the way up applies its own trans / normal_trans
        P = p; Nw = n;
        const uint32_t* rec = sc.hier_rec + 8 * (size_t)hit.node;  // the node's path in one 32-byte line (pt_api.hip) instead of chain_off -> chain
        const uint32_t len = rec[0] & 255u;
        if (len == 255u) {  // more than seven levels
            for (uint32_t k = sc.chain_off[hit.node + 1]; k-- > sc.chain_off[hit.node];) {
                P = pt_xform_point(sc.g_fwd + 12 * (size_t)sc.chain[k], P);
                Nw = pt_xform_dir(sc.g_nrm + 9 * (size_t)sc.chain[k], 3, Nw);
            }
        } else {
            const uint32_t rec0 = rec[0];
            for (uint32_t k = len; k-- > 0u;) {
                // A level whose matrices are the identity (pt_api.hip marks them in the record: a group without a transform, like every reference scene's
                // root) changes no bit of a point or a direction whose components are finite and not -0: ((1 x + 0 y) + 0 z) + 0 is x. Skipped then (round 4).
                if (SKIP_IDENT && ((rec0 >> (8u + k)) & 1u) && pt_identity_safe(P) && pt_identity_safe(Nw)) continue;
                const uint32_t g = rec[1 + k];
                P = pt_xform_point(sc.g_fwd + 12 * (size_t)g, P);
                Nw = pt_xform_dir(sc.g_nrm + 9 * (size_t)g, 3, Nw);
            }
        }
    } else {
        P = pt_xform_point(sc.fwd + 12 * (size_t)hit.node, p);
        Nw = pt_xform_dir(sc.nrm + 9 * (size_t)hit.node, 3, n);
    }
    PtVec3 N = pt_normalized(Nw);  // material.rs:123-125
    uint32_t ftag = 0;
    bool later = false;
    if (TEX && sc.mat_maps && (sc.mat_maps[2 * mat] >= 0 || sc.mat_maps[2 * mat + 1] >= 0)) {  // material.rs:109-144
        if (MAPS_LATER) {
            later = true;
        } else {
            PtMapsOut mo = pt_apply_maps(sc.texview, mat, type, hit.sub, local.o.x, local.o.y, local.o.z, local.d.x, local.d.y, local.d.z,
                                         p.x, p.y, p.z, n.x, n.y, n.z);
            if (mo.has_n) N = pt_v3(mo.n[0], mo.n[1], mo.n[2]);
            ftag = mo.texel;
        }
    }
    *P_out = P; *N_out = N; *mat_out = mat; *ftag_out = ftag;
    return later;
}

// material.rs:179-210 for one light that is not occluded: (diffuse + specular) / attenuation. lcol = the light's colour,
// falloff = its three attenuation coefficients (light.rs:31-33), light_dir / light_dist as the shadow ray was set up.
PT_HD PtVec3 pt_light_term(PtVec3 lcol, PtVec3 falloff, PtVec3 light_dir, double light_dist, PtVec3 N, PtVec3 ray_dir, PtVec3 kd, PtVec3 ks,
                           double shininess) {
#if defined(PT_ABLATE) && PT_ABLATE == 2  // measurement builds only (profiles/light_slope.py): no light term at all
    return lcol;
#endif
    double attenuation = falloff.x + falloff.y * light_dist + falloff.z * light_dist * light_dist;  // light.rs:31-33
    double normal_light = fmax(pt_dot(N, light_dir), 0.0);
    PtVec3 diffuse = (kd * lcol) * normal_light;
    PtVec3 specular = pt_v3(0.0, 0.0, 0.0);
#if defined(PT_ABLATE) && PT_ABLATE == 1  // measurement builds only: no specular term (half vector, pow)
    if (false) {
#elif defined(PT_ABLATE) && PT_ABLATE == 3  // measurement builds only: the device library's pow instead of glibc's
    if (ks.x > PT_EPSILON || ks.y > PT_EPSILON || ks.z > PT_EPSILON) {
        PtVec3 view = -ray_dir;
        PtVec3 half = pt_normalized(view + light_dir);
        double nhs = pow(fmax(pt_dot(N, half), 0.0), 4.0 * shininess);
        specular = (ks * lcol) * nhs;
    } else if (false) {
#else
    if (ks.x > PT_EPSILON || ks.y > PT_EPSILON || ks.z > PT_EPSILON) {
#endif
        PtVec3 view = -ray_dir;
        PtVec3 half = pt_normalized(view + light_dir);
        double nhs = pt_pow(fmax(pt_dot(N, half), 0.0), 4.0 * shininess);
        specular = (ks * lcol) * nhs;
    }
    return (diffuse + specular) / attenuation;
}

// Runs the lane's interpreter until it needs a ray traced (L.has_ray) or its sample is finished
// (L.stage == PT_ST_DONE, colour in the lane's LDS column). `hit` is the result of the ray the lane traced last.
#ifdef PT_ADVANCE_NOINLINE  // measured slower at every occupancy (profiles/r01/notes.md)
#define PT_ADVANCE_ATTR PT_NOINLINE
#else
#define PT_ADVANCE_ATTR PT_HD
#endif
// TEX = false compiles the texture / normal-map path out (scenes without mapped materials).
PT_HD uint64_t pt_fork_ticket(uint32_t nonce, uint32_t seq, int32_t depth) { return ((uint64_t)(nonce & 0xFFFFFFu) << 40) | ((uint64_t)(seq & 0xFFFFFFu) << 8) | (uint64_t)(depth & 0xFF) | (1ull << 39); }
PT_HD uint64_t pt_mail_load(const double* slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(reinterpret_cast<const uint64_t*>(slot), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    uint64_t v; memcpy(&v, slot, 8); return v;
#endif
}
PT_HD void pt_mail_store(double* slot, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(reinterpret_cast<uint64_t*>(slot), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    memcpy(slot, &v, 8);
#endif
}

// The texture / normal map of the hit the lane is about to shade (material.rs:109-144), applied BEFORE pt_lane_advance runs its
// PT_ST_CLOSEST_DONE: returns PT_PRE_MAPS | the texel tag (| PT_PRE_NORMAL with the shading normal in the frame's PT_L_N slots).
// The hit's model-space point and normal are worked out here and again in pt_hit_surface - the same operations, the same bits.
PT_HD bool pt_hit_has_maps(const PtSceneView& sc, const PtHit& hit) {
    if (hit.node == PT_NO_HIT || !sc.mat_maps) return false;
    const uint32_t mat = sc.info[4 * (size_t)hit.node + 3];
    return sc.mat_maps[2 * mat] >= 0 || sc.mat_maps[2 * mat + 1] >= 0;
}
template <bool HIER>
PT_HD uint32_t pt_lane_maps(const PtSceneView& sc, const PtRay& ray, const PtHit& hit, const PtFrameRef& fr) {
    uint32_t type, mat;
    PtRay local;
    PtVec3 p, n;
    pt_hit_model<HIER>(sc, ray, hit, &type, &mat, &local, &p, &n);
    PtMapsOut mo = PT_LANE_APPLY_MAPS(sc.texview, mat, type, hit.sub, local.o.x, local.o.y, local.o.z, local.d.x, local.d.y, local.d.z, p.x, p.y, p.z, n.x, n.y, n.z);
    if (mo.has_n) fr.set_l3(PT_L_N, pt_v3(mo.n[0], mo.n[1], mo.n[2]));
    return PT_PRE_MAPS | (mo.has_n ? PT_PRE_NORMAL : 0u) | mo.texel;
}

template <bool STATS, bool TEX, bool HIER = false, int PARK = 0, bool FORK = false>
PT_ADVANCE_ATTR void pt_lane_advance(const PtRenderArgs& a, PtLane& L, const PtHit& hit, const PtFrameRef& fr, PtCounters* cnt, uint32_t pre = 0) {
    const PtSceneView& sc = a.scene;
    L.has_ray = false;
    PtVec3 value = pt_v3(0.0, 0.0, 0.0);  // colour being returned to the parent frame
    bool returning = false;
    for (;;) {
        if (returning) {
            // `value` = Ray::color() of the ray cast at depth L.depth
            returning = false;
            if (FORK && L.depth == L.base && L.base > 0) {  // a taken subtree is finished: its colour to the owner's mailbox, then the ticket
                double* mail = fr.spill + ((ptrdiff_t)(int32_t)(L.owner & 0xFFFFu) - (ptrdiff_t)PT_THREAD_IN_BLOCK()) * (PT_SPILL_DEPTHS * PT_SPILL_STRIDE) +
                               (size_t)(L.owner >> 16) * PT_SPILL_STRIDE + PT_H_MAIL;
                mail[0] = value.x; mail[1] = value.y; mail[2] = value.z;
                pt_mail_store(mail + 3, L.ticket);
                L.base = 0;
                L.stage = PT_ST_DONE;
                return;
            }
            if (L.depth == 0) {  // render.rs:36-43: this sample's colour, summed with its chunk by pt_render_kernel
                if (FORK) fr.set_h3(PT_RESULT_DEPTH, PT_H_MAIL, value);  // the LDS frame is reused by the tasks this lane may take
                else fr.set_l3(PT_L_VALUE, value);
                L.stage = PT_ST_DONE;
                return;
            }
            L.depth--;
            double f[PT_PARK_F64];
            const bool in_lds = PARK > 0 && L.depth >= L.lo;
            if (in_lds) fr.load_lds(0, f);
            else { fr.load_hbm(L.depth, f); if (PARK > 0) L.lo = L.depth; }  // nothing younger can be parked: the LDS slot is free
            uint32_t mat, fstage;
            PtFrameRef::unpack_tag(f[PT_H_TAG], &mat, &fstage);
            const double reflectivity = f[PT_H_REFL];
            PtVec3 color = pt_v3(f[PT_H_COLOR], f[PT_H_COLOR + 1], f[PT_H_COLOR + 2]);
            if ((fstage & PT_FS_STAGE_MASK) == PT_FS_WAIT_REFRACT) {  // material.rs:305-309
                PtVec3 reflected = pt_v3(f[PT_H_DIR], f[PT_H_DIR + 1], f[PT_H_DIR + 2]);
                double schlick = f[PT_H_SCHLICK];
                double transmittance = 1.0 - schlick;
                PtVec3 total = reflected * schlick + value * transmittance;
                value = color + total * reflectivity;
                returning = true;
                continue;
            }
            // PT_FS_WAIT_REFLECT: `value` is reflected_color (material.rs:242-243)
            if (!(fstage & PT_FS_HAVE_REFRACT)) {  // opaque (material.rs:312-316) or total internal reflection (:277-284)
                value = color + value * reflectivity;
                returning = true;
                continue;
            }
            // the refracted ray (material.rs:286-303); its direction was worked out when the hit was shaded
            const bool forked = FORK && (fstage & PT_FS_FORKED);
            if (!forked) {
                L.ray.o = pt_v3(f[PT_H_P], f[PT_H_P + 1], f[PT_H_P + 2]);
                L.ray.d = pt_v3(f[PT_H_DIR], f[PT_H_DIR + 1], f[PT_H_DIR + 2]);
            }
            // the frame waits again, now for the refracted subtree, with the reflected colour in place of the direction
            const double tag2 = PtFrameRef::pack_tag(mat, PT_FS_WAIT_REFRACT);
            if (in_lds) {
                fr.p(0, PT_H_DIR) = value.x; fr.p(0, PT_H_DIR + 1) = value.y; fr.p(0, PT_H_DIR + 2) = value.z;
                fr.p(0, PT_H_TAG) = tag2;
            } else if (PARK > 0) {  // it came from HBM and the LDS slot is free: it need not go back
                f[PT_H_DIR] = value.x; f[PT_H_DIR + 1] = value.y; f[PT_H_DIR + 2] = value.z; f[PT_H_TAG] = tag2;
                fr.store_lds(0, f);
            } else {
                fr.set_h3(L.depth, PT_H_DIR, value);
                fr.h(L.depth, PT_H_TAG) = tag2;
            }
            L.depth++;
            if (forked) {  // another lane walks that subtree: wait for its colour (the ticket in this frame's mailbox)
                L.wait_ticket = pt_fork_ticket(a.launch_nonce, fstage >> PT_FS_SEQ_SHIFT, L.depth - 1);
                L.stage = PT_ST_WAIT_FORK;
                return;
            }
            L.ray_any = false; L.has_ray = true; L.stage = PT_ST_CLOSEST_DONE;
            if (STATS) cnt->refract++;
            return;
        }
        if (FORK && L.stage == PT_ST_WAIT_FORK) {
            const double* mail = fr.spill + (size_t)(L.depth - 1) * PT_SPILL_STRIDE + PT_H_MAIL;
            if (pt_mail_load(mail + 3) != L.wait_ticket) return;  // not yet: no ray this pass
            value = pt_v3(mail[0], mail[1], mail[2]);
            returning = true;  // as if the refracted ray's Ray::color() had just returned (material.rs:305-309 follows)
            continue;
        }
        switch (L.stage) {
        case PT_ST_NEW_SAMPLE: {
            PT_FENCE;
            double jx = 0.5, jy = 0.5;
            PT_LANE_XY(a, L, lx, ly);
            if (a.jitter_mode == PT_JITTER_RNG) {  // render.rs:38-39: x drawn before y
                const uint32_t sample = pt_lane_sample(a, L.item);
                uint64_t pixel = (uint64_t)ly * a.width + lx;
                jx = pt_rng_f64(a.seed, pixel, sample, 0);
                jy = pt_rng_f64(a.seed, pixel, sample, 1);
            }
            L.draw = 2;
            L.ray = pt_camera_ray(a.cam, (double)lx + jx, (double)ly + jy);
            L.depth = 0;
            if (PARK > 0) L.lo = 0;
            L.ray_any = false; L.has_ray = true; L.stage = PT_ST_CLOSEST_DONE;
            if (STATS) cnt->primary++;
            return;
        }
        case PT_ST_CLOSEST_DONE: {  // ray.rs:139-148
            PT_FENCE;
            if (hit.node == PT_NO_HIT) { PT_LANE_XY(a, L, lx, ly); value = pt_background(a, lx, ly); returning = true; continue; }
            if (STATS) cnt->hits++;
            PtVec3 P, N;
            uint32_t mat, ftag;
#ifdef PT_MAPS_BEFORE
            pt_hit_surface<TEX, HIER, true, true>(sc, L.ray, hit, &P, &N, &mat, &ftag);
#else
            pt_hit_surface<TEX, HIER, false, true>(sc, L.ray, hit, &P, &N, &mat, &ftag);
#endif
            fr.set_l3(PT_L_P, P);
            PT_FENCE;
            if (TEX && (pre & PT_PRE_MAPS)) ftag = pre & (PT_FS_TEXEL | 0xFFFFFFu);  // pt_lane_maps ran for this hit
            if (!(TEX && (pre & PT_PRE_NORMAL))) fr.set_l3(PT_L_N, N);                // (else the normal map's normal is already there)
            fr.set_l3(PT_L_D, L.ray.d);
            fr.l(PT_L_TAG) = PtFrameRef::pack_tag(mat, ftag);
            L.light = 0;
            L.occluded = 0;
            L.draw0 = L.draw;
            L.stage = PT_ST_LIGHT;
            continue;
        }
        case PT_ST_LIGHT: {  // material.rs:149-179: one shadow ray per light, whatever the material
            PT_FENCE;
            if (L.light >= sc.n_lights) { L.stage = PT_ST_SHADE; continue; }  // a scene without lights
            const double* light = sc.lights + 15 * (size_t)L.light;
            bool is_area;
            PtVec3 lpos = pt_light_position(a, L, light, L.draw, &is_area);
            if (is_area) L.draw += 2;
            PtVec3 P = fr.l3(PT_L_P);
            PtVec3 hit_to_light = lpos - P;
            double light_dist = pt_length(hit_to_light);
            L.ray.o = P;
            L.ray.d = hit_to_light / light_dist;
            L.ray_any = true; L.has_ray = true; L.stage = PT_ST_SHADOW_DONE;
            if (STATS) cnt->shadow++;
            return;
        }
        case PT_ST_SHADOW_DONE: {  // material.rs:174-179 only asks whether anything is in the way
            if (hit.node != PT_NO_HIT) L.occluded |= 1u << (L.light % PT_LIGHT_ROUND);
            L.light++;
            L.stage = (L.light >= sc.n_lights || L.light % PT_LIGHT_ROUND == 0) ? PT_ST_SHADE : PT_ST_LIGHT;
            continue;
        }
        default: {  // PT_ST_SHADE: material.rs:148-243 for the lights whose shadow rays are back
            PT_FENCE;
            uint32_t mat, ftag;
            PtFrameRef::unpack_tag(fr.l(PT_L_TAG), &mat, &ftag);
            const double* m = sc.materials + 10 * (size_t)mat;
            PtVec3 ray_dir = fr.l3(PT_L_D), P = fr.l3(PT_L_P), N = fr.l3(PT_L_N);
            PtVec3 kd = pt_v3(m[0], m[1], m[2]), ks = pt_v3(m[3], m[4], m[5]);
            if (TEX && (ftag & PT_FS_TEXEL)) kd = pt_v3(sc.srgb_lut[ftag & 255u], sc.srgb_lut[(ftag >> 8) & 255u], sc.srgb_lut[(ftag >> 16) & 255u]);
            const uint32_t round_first = (L.light - 1u) / PT_LIGHT_ROUND * PT_LIGHT_ROUND;  // L.light > 0 here unless the scene has no light
            PtVec3 color;
            if (sc.n_lights == 0 || round_first == 0) color = pt_v3(sc.ambient[0], sc.ambient[1], sc.ambient[2]) * kd;  // material.rs:148
            else color = fr.h3(L.depth, PT_H_COLOR);  // a later round of a scene with > 32 lights
            uint32_t draw = L.draw0;
            for (uint32_t li = sc.n_lights ? round_first : 0u; li < L.light; li++) {  // material.rs:179-210
                const double* light = sc.lights + 15 * (size_t)li;
                bool is_area;
                PtVec3 lpos = pt_light_position(a, L, light, draw, &is_area);
                if (is_area) draw += 2;
                if ((L.occluded >> (li - round_first)) & 1u) continue;
                PtVec3 hit_to_light = lpos - P;
                double light_dist = pt_length(hit_to_light);
                PtVec3 light_dir = hit_to_light / light_dist;
                color = color + pt_light_term(pt_v3(light[3], light[4], light[5]), pt_v3(light[6], light[7], light[8]), light_dir, light_dist, N, ray_dir, kd, ks, m[6]);
            }
            if (L.light < sc.n_lights) {  // more than 32 lights: park the colour and do the next 32
                fr.set_h3(L.depth, PT_H_COLOR, color);
                L.occluded = 0;
                L.draw0 = L.draw;
                L.stage = PT_ST_LIGHT;
                continue;
            }
            PT_FENCE;
            // material.rs:216-243
            const double reflectivity = m[7], glossy = m[8], ior = m[9];
            if (!(reflectivity > 0.0)) { value = color; returning = true; continue; }
            PtVec3 reflect_dir = ray_dir - (N * 2.0) * pt_dot(ray_dir, N);  // material.rs:218
            if (glossy > 0.0) {  // material.rs:221-239 (not renormalised: quirk Q5)
                PtVec3 off = (fabs(reflect_dir.x) < PT_EPSILON && fabs(reflect_dir.y) < PT_EPSILON)
                                 ? reflect_dir + pt_v3(0.0, 0.1, 0.0) : reflect_dir + pt_v3(0.0, 0.0, 0.1);
                PtVec3 u_basis = pt_cross(reflect_dir, off);
                PtVec3 v_basis = pt_cross(reflect_dir, u_basis);
                const uint32_t sample = pt_lane_sample(a, L.item);
                PT_LANE_XY(a, L, lx, ly);
                uint64_t pixel = (uint64_t)ly * a.width + lx;
                double u_coord = -glossy / 2.0 + pt_rng_f64(a.seed, pixel, sample, L.draw) * glossy;
                double v_coord = -glossy / 2.0 + pt_rng_f64(a.seed, pixel, sample, L.draw + 1) * glossy;
                L.draw += 2;
                reflect_dir = reflect_dir + (u_basis * u_coord + v_basis * v_coord);
            }
            // The refracted ray (material.rs:245-303) is cast after the reflected subtree has returned; its direction and
            // the Schlick term depend only on this hit, so they are worked out now and parked with the frame.
            PtVec3 refract_dir = pt_v3(0.0, 0.0, 0.0);
            double schlick = 0.0;
            bool have = false;
            if (ior > 0.0) {
                double cos_incident = 0.0;
                if (pt_dot(ray_dir, N) < 0.0) {  // entering (material.rs:253-265)
                    if (pt_refracted_direction(ray_dir, N, ior, &refract_dir)) { cos_incident = pt_dot(-ray_dir, N); have = true; }
                } else if (pt_refracted_direction(ray_dir, -N, 1.0 / ior, &refract_dir)) {  // leaving (:266-276)
                    cos_incident = pt_dot(refract_dir, N); have = true;
                }
                // !have: total internal reflection (:277-284); also where the reference's expect() at :257-258 would panic
                if (have) {
                    double r0 = (ior - 1.0) * (ior - 1.0);
                    r0 = r0 / ((ior + 1.0) * (ior + 1.0));
                    schlick = r0 + (1.0 - r0) * pt_powi5(1.0 - cos_incident);
                }
            }
            if (L.depth + 1 > PT_MAX_DEPTH) {  // depth-11 rays: their colour is always the background (material.rs:102-104), not traced
                if (STATS) cnt->depth11_skipped++;
                PT_LANE_XY(a, L, lx, ly);
                PtVec3 bg = pt_background(a, lx, ly);
                if (!have) {
                    value = color + bg * reflectivity;
                } else {
                    if (STATS) cnt->depth11_skipped++;
                    double transmittance = 1.0 - schlick;
                    PtVec3 total = bg * schlick + bg * transmittance;
                    value = color + total * reflectivity;
                }
                returning = true;
                continue;
            }
            {
                double f[PT_PARK_F64];
                f[PT_H_COLOR] = color.x; f[PT_H_COLOR + 1] = color.y; f[PT_H_COLOR + 2] = color.z;
                uint32_t fs = PT_FS_WAIT_REFLECT | (have ? PT_FS_HAVE_REFRACT : 0);
                if (FORK && have) {  // the refracted ray may be taken by an idle lane (pt_render_kernel matches offers and takers)
                    L.fork_seq++;
                    fs |= L.fork_seq << PT_FS_SEQ_SHIFT;
                    L.offer = true;
                    // whatever an earlier launch (of this or another process) left in this frame's mailbox must not pass for a
                    // ticket: cleared here, by the owner, passes before any taker can write it
                    pt_mail_store(&fr.h(L.depth, PT_H_MAIL_TICKET), 0ull);
                }
                f[PT_H_TAG] = PtFrameRef::pack_tag(mat, fs);
                f[PT_H_DIR] = refract_dir.x; f[PT_H_DIR + 1] = refract_dir.y; f[PT_H_DIR + 2] = refract_dir.z;
                f[PT_H_P] = P.x; f[PT_H_P + 1] = P.y; f[PT_H_P + 2] = P.z;
                f[PT_H_SCHLICK] = schlick;
                f[PT_H_REFL] = reflectivity;
                if (PARK > 0) {
                    if (L.depth - L.lo == 1) {  // the LDS slot holds the parent's frame: that one goes to its HBM line
                        double old[PT_PARK_F64];
                        fr.load_lds(0, old);
                        fr.store_hbm(L.lo, old);
                        L.lo++;
                    }
                    fr.store_lds(0, f);
                } else {
                    fr.store_hbm(L.depth, f);
                }
            }
            L.ray.o = P;
            L.ray.d = reflect_dir;
            L.depth++;
            L.ray_any = false; L.has_ray = true; L.stage = PT_ST_CLOSEST_DONE;
            if (STATS) cnt->reflect++;
            return;
        }
        }
    }
}
