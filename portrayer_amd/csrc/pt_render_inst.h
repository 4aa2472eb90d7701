// Launchers of the render kernel, one per traversal mode: each is defined in its own object
// (pt_render_inst.hip compiled with -DPT_INST_MODE=<mode>) so that the eight sets of kernel
// instantiations compile side by side.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_shade.h"

#define PT_DECLARE_MODE_LAUNCHER(n) \
    hipError_t pt_launch_mode_##n(const PtRenderArgs& a, int variant, bool stats, bool tex, int n_cu, hipStream_t stream, uint32_t* grid, bool launch)
PT_DECLARE_MODE_LAUNCHER(1);  // PT_MODE_FLAT
PT_DECLARE_MODE_LAUNCHER(2);  // PT_MODE_KD
PT_DECLARE_MODE_LAUNCHER(3);  // PT_MODE_FLAT_NOMESH
PT_DECLARE_MODE_LAUNCHER(4);  // PT_MODE_FLAT_KDMESH
PT_DECLARE_MODE_LAUNCHER(5);  // PT_MODE_HIER
PT_DECLARE_MODE_LAUNCHER(6);  // PT_MODE_HIER_NOMESH
PT_DECLARE_MODE_LAUNCHER(7);  // PT_MODE_KD_NOMESH
PT_DECLARE_MODE_LAUNCHER(8);  // PT_MODE_HIER_MESH
PT_DECLARE_MODE_LAUNCHER(9);  // PT_MODE_KD_MESH
