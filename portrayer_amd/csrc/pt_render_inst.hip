// One traversal mode's instantiations of pt_render_kernel (counting / plain, textured / untextured,
// 3 / 4 waves per SIMD) and their launcher. Compiled once per mode: -DPT_INST_MODE=1..7 (Makefile).
#include "pt_render_kernel.h"
#include "pt_render_inst.h"

#ifndef PT_INST_MODE
#error "compile with -DPT_INST_MODE=<PT_MODE_*>"
#endif
#define PT_INST_CAT2(a, b) a##b
#define PT_INST_CAT(a, b) PT_INST_CAT2(a, b)

hipError_t PT_INST_CAT(pt_launch_mode_, PT_INST_MODE)(const PtRenderArgs& a, int waves, bool stats, bool tex, int n_cu, hipStream_t stream,
                                                      uint32_t* grid, bool launch) {
#if PT_INST_MODE == 3 || PT_INST_MODE == 6
    // the mesh-free FLAT / HIER kernels exist for 3 waves per SIMD only (round 1: big-scene 14.2 vs 13.0 Gray/s at 3 vs 4)
    (void)waves;
    return pt_dispatch_variant<PT_INST_MODE, 3>(a, stats, tex, n_cu, stream, grid, launch);
#else
    return waves == 4 ? pt_dispatch_variant<PT_INST_MODE, 4>(a, stats, tex, n_cu, stream, grid, launch)
                      : pt_dispatch_variant<PT_INST_MODE, 3>(a, stats, tex, n_cu, stream, grid, launch);
#endif
}
