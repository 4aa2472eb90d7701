// One traversal mode's instantiations of pt_render_kernel (counting / plain, textured / untextured,
// the variants listed in pt_render_kernel.h) and their launcher. Compiled once per mode: -DPT_INST_MODE=1..8 (Makefile).
// The texture routine (pt_apply_maps) is out of line so that untextured hits keep their register budget - except in the flat_scene
// kernels of scenes with KDMesh trees, where inlining it measured +10 % (transmission-refraction 11.5 -> 12.6 Gray/s; the same scene
// in the hierarchical semantics loses 7 % inlined: profiles/r03/notes.md section 4). In that instantiation the interpreter also applies
// the maps BEFORE it enters its state machine (PT_MAPS_BEFORE, round 4: 12.9 -> 13.6 Gray/s, 148 -> 95 GB HBM-side per frame; the
// hierarchical instantiation loses 4 % that way and keeps the call inside pt_hit_surface: profiles/r04/notes.md section 6).
#if defined(PT_INST_MODE) && PT_INST_MODE == 4 && !defined(PT_MAPS_INLINE) && !defined(PT_MAPS_OUT_OF_LINE)
#define PT_MAPS_INLINE
#endif
#if defined(PT_INST_MODE) && PT_INST_MODE == 4 && !defined(PT_MAPS_BEFORE) && !defined(PT_MAPS_INSIDE)
#define PT_MAPS_BEFORE
#endif
// The mesh-free flat_scene kernels (mode 3: water-glass): the interpreter's map stage in front of the state machine too, with the routine's body in place THERE only
// (16.0 -> 17.3 Gray/s, c41 / c44; inline everywhere in that mode: 16.2; the hierarchical mesh-free mode 6 loses either way and keeps the call inside pt_hit_surface).
#if defined(PT_INST_MODE) && PT_INST_MODE == 3 && !defined(PT_MAPS_BEFORE) && !defined(PT_MAPS_INSIDE)
#define PT_MAPS_BEFORE
#define PT_LANE_MAPS_INLINE
#endif
#include "pt_render_kernel.h"
#include "pt_render_inst.h"

#ifndef PT_INST_MODE
#error "compile with -DPT_INST_MODE=<PT_MODE_*>"
#endif
#define PT_INST_CAT2(a, b) a##b
#define PT_INST_CAT(a, b) PT_INST_CAT2(a, b)

hipError_t PT_INST_CAT(pt_launch_mode_, PT_INST_MODE)(const PtRenderArgs& a, int variant, bool stats, bool tex, int n_cu, hipStream_t stream,
                                                      uint32_t* grid, bool launch) {
    return pt_dispatch_variant<PT_INST_MODE>(a, variant, stats, tex, n_cu, stream, grid, launch);
}
