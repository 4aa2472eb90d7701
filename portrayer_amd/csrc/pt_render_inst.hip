// One traversal mode's instantiations of pt_render_kernel (counting / plain, textured / untextured,
// the variants listed in pt_render_kernel.h) and their launcher. Compiled once per mode: -DPT_INST_MODE=1..7 (Makefile).
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
