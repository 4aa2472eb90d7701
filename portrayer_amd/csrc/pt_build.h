// Device-side build of a mesh's triangle tree: parallel locally-ordered clustering
// (Meister & Bittner 2018) over the Morton-sorted triangles, run by HIP kernels straight from the
// triangle records that are already in HBM and written in the format pt_trace.h walks (PtBvhNode).
//
// Like the host build (pt_bvh.h) it has no counterpart in the reference and cannot change a result:
// FLAT mode's answer is "nearest hit over all candidates, lowest index on ties", whichever tree finds
// the candidates. It exists because the host build is the longest step of a render of a large mesh
// (1.25 M triangles: ~700 ms on the host) and the reference converts the scene inside every render
// call (render.rs:115-126).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pt_scene_view.h"

struct PtDeviceBuildResult {
    uint32_t root;   // packed reference of the tree's root
    int depth;       // inner nodes on the longest root-to-leaf path (+1): bound for the walk's stack
    float ms;        // device time of the build (HIP events)
    int rounds;      // clustering rounds it took
};

// Builds the tree of triangles [tri_first, tri_first + n) of `d_tri_v` (9 f64 each, model space) into
// d_nodes[node_base ...] (PT_DEVICE_TREE_NODES(n) nodes) and d_items[item_base ...]
// (PT_DEVICE_TREE_ITEMS(n, max_leaf) GLOBAL triangle indices: the triangles in Morton order, then one
// slot of max_leaf per node for the leaves that hold more than one); the caller allocates both.
// `lo` / `hi` = the mesh's bounds (any box containing all vertices); every triangle's box is grown by
// `pad` on all sides before it is rounded outward to f32 (pt_api pads the host build's boxes the same
// way). n > max_leaf required. Returns hipSuccess or the first error.
#define PT_DEVICE_TREE_NODES(n) ((n) - 1u)
#define PT_DEVICE_TREE_ITEMS(n, max_leaf) ((n) + (uint32_t)(max_leaf) * ((n) - 1u))
hipError_t pt_device_build_mesh_tree(const double* d_tri_v, uint32_t tri_first, uint32_t n, const double lo[3], const double hi[3], double pad, int max_leaf,
                                     PtBvhNode* d_nodes, uint32_t node_base, uint32_t* d_items, uint32_t item_base, hipStream_t stream,
                                     PtDeviceBuildResult* out);
